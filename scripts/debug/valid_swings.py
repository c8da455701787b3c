"""Count validation dips (valid loss > 0.1) over seeds and epochs: engine fp32 vs the torch-eager restatement, same data."""
import os, sys, tempfile
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mnist_parity as mp
from distributedpytorch_amd.data import mnist
path = tempfile.mkdtemp(prefix="dpa_mnist_")
mnist.generate_synthetic_mnist(path, n_train=60000, n_test=10000, seed=1234)
seeds = [int(s) for s in (sys.argv[1] if len(sys.argv) > 1 else "1,2,3").split(",")]
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
for seed in seeds:
    print("engine seed", seed, flush=True)
    r, sd0 = mp.run_engine(path, torch.float32, epochs, 64, seed)
    print("torch seed", seed, flush=True)
    mp.run_torch(path, epochs, 64, seed, sd0)
