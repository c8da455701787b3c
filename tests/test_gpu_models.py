"""T3: whole-model forward/backward on the HIP kernels vs torch.nn oracles (fp32 exact-MFMA path tight, bf16 loose)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(name, oracle, hw, n, dtype, small_input=False):
    from distributedpytorch_amd.models.factory import get_model
    from tests.helpers import framework_step, grads_by_name, oracle_step
    torch.manual_seed(0)
    net, _ = get_model(name, 10, device="cuda", dtype=dtype, small_input=small_input)
    net.load_state_dict(oracle.state_dict())
    x = torch.randn(n, 3, hw, hw)
    y = torch.randint(0, 10, (n,))
    logits, loss, acc = framework_step(net, x, y)
    # float64 oracle: torch's own fp32 backward is 100x noisier than the kernels under test (measured:
    # conv1.weight rel. error vs fp64 = 2.5e-6 ours, 3.6e-4 torch-cpu-fp32, 2.0e-3 torch-gpu-fp32)
    oracle = oracle.double()
    ref_logits, ref_loss = oracle_step(oracle, x.double(), y)
    ref_logits = ref_logits.float()
    g = grads_by_name(net)
    ref_g = {k: p.grad.float() for k, p in oracle.named_parameters()}
    if dtype == torch.float32:
        torch.testing.assert_close(logits, ref_logits, rtol=2e-3, atol=2e-4)
        assert abs(loss - ref_loss) < 1e-4 * max(1.0, abs(ref_loss))
        # A single ReLU sitting within rounding distance of zero flips its mask between fp32 and fp64 and shifts every
        # upstream gradient by ~1 % (measured; torch's own fp32 backward shows the same vs fp64), so the model-level
        # check is on the relative L2 error per parameter; digit-level checks are in test_gpu_ops / test_gpu_conv.
        for k, v in ref_g.items():
            rel = (g[k] - v).norm().item() / max(v.norm().item(), 1e-12)
            assert rel < 5e-2, (k, rel)
    else:
        assert abs(loss - ref_loss) < 0.05 * max(1.0, abs(ref_loss))
        # bf16 activations / MFMA operands, fp32 accumulation and master weights: every parameter's gradient within
        # 10 % relative L2 of the fp64 oracle (a wrong-by-10 % gradient must not pass)
        worst = max(((g[k] - v).norm().item() / max(v.norm().item(), 1e-12), k) for k, v in ref_g.items())
        assert worst[0] < 0.1, worst
    logits_e, _, _ = framework_step(net, x, y, train=False)
    ref_e, _ = oracle_step(oracle, x.double(), y, train=False)
    ref_e = ref_e.float()
    tol = dict(rtol=2e-3, atol=2e-3) if dtype == torch.float32 else dict(rtol=1e-1, atol=1e-1)
    torch.testing.assert_close(logits_e, ref_e, **tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_resnet18_gpu(dtype):
    from tests import torch_ref_models as ref
    _run("resnet", ref.resnet18(10), 64, 16, dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_resnet18_small_input_gpu(dtype):
    from tests import torch_ref_models as ref
    _run("resnet", ref.resnet18(10, small_input=True), 32, 16, dtype, small_input=True)


def test_resnet50_gpu_fp32():
    from tests import torch_ref_models as ref
    oracle = ref.resnet50(10)
    with torch.no_grad():
        for k, v in oracle.named_parameters():
            if k.endswith("bn3.weight"):
                v.fill_(0.2)
    _run("resnet50", oracle, 64, 8, torch.float32)
