"""Does the torch-eager restatement of the reference show the same epoch-to-epoch validation swings (train-time
RandomResizedCrop vs plain eval images, BatchNorm running statistics)?  4 epochs, two seeds."""
import os, sys, tempfile
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mnist_parity as mp
from distributedpytorch_amd.data import mnist
path = tempfile.mkdtemp(prefix="dpa_mnist_")
mnist.generate_synthetic_mnist(path, n_train=60000, n_test=10000, seed=1234)
for seed in (1234, 7):
    _, sd0 = mp.run_engine(path, torch.float32, 1, 64, seed)
    print("torch seed", seed, flush=True)
    r = mp.run_torch(path, 4, 64, seed, sd0)
    print("  test acc", r["test_acc"], flush=True)
