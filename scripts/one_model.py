"""Run a few training steps of one zoo model (for rocprofv3): python scripts/one_model.py alexnet fp32 [batch] [steps]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributedpytorch_amd.models.factory import get_model, get_model_input_size
from distributedpytorch_amd.nn.loss import Criterion
from distributedpytorch_amd.nn.optim import ArenaOptimizer
from distributedpytorch_amd.parallel.ddp import DistributedModel
name = sys.argv[1]; dtype = torch.float32 if sys.argv[2] == "fp32" else torch.bfloat16
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64; steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
side = get_model_input_size(name)
net, _ = get_model(name, 10, device="cuda", dtype=dtype)
dm = DistributedModel(net, Criterion(), ArenaOptimizer(net.store, "adam", lr=1e-3))
x = torch.randn(batch, side, side, net.in_channels, device="cuda").to(dtype); y = torch.randint(0, 10, (batch,), device="cuda")
for _ in range(steps): dm.train_step(x, y)
torch.cuda.synchronize()
