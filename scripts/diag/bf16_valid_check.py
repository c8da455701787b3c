"""Is the bf16 validation dip after epoch 2 (93.8 %) a fluctuation?  Train 4 epochs, two seeds, print per-epoch metrics."""
import os, sys, tempfile
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mnist_parity as mp
from distributedpytorch_amd.data import mnist
path = tempfile.mkdtemp(prefix="dpa_mnist_")
mnist.generate_synthetic_mnist(path, n_train=60000, n_test=10000, seed=1234)
for seed in (1234, 7):
    for dt in (torch.bfloat16, torch.float32):
        print("seed", seed, dt, flush=True)
        r, _ = mp.run_engine(path, dt, 4, 64, seed)
        print("  test acc", r["test_acc"], flush=True)
