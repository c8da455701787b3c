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
    bf16_ref = None
    if dtype == torch.bfloat16:          # torch's own bf16 execution (GPU autocast) of the same model and batch
        import copy
        tb = copy.deepcopy(oracle).cuda().train()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            tl = torch.nn.functional.cross_entropy(tb(x.cuda()), y.cuda())
        tl.backward()
        bf16_ref = {k: p.grad.float().cpu() for k, p in tb.named_parameters()}
        del tb
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
        # What bf16 can and cannot promise here (profiles/r2_bf16_grad_noise_resnet18.txt): on a randomly initialised
        # ResNet fed random data ANY bf16 execution is 0.2-0.45 relative L2 away from the fp64 gradient on most
        # parameters - torch's own bf16 autocast of the same model included (rounding noise amplified by ReLU-mask
        # flips and by the cancellation inside every gradient sum).  So the bound is two-fold:
        #  (a) per parameter, not worse than torch's bf16 execution of the same batch (1.5x + 0.05);
        #  (b) a bug is systematic, the noise is not: the projection <g, ref> / <ref, ref> of every large parameter must
        #      agree with the projection of torch's bf16 gradient within 5 % (both shrink a little, 0.93-0.99, because
        #      flipped ReLU masks decorrelate; measured on MI355X) - a gradient that is wrong by 10 % fails this however
        #      noisy the digits are.
        e_ours = {k: (g[k] - v).norm().item() / max(v.norm().item(), 1e-12) for k, v in ref_g.items()}
        e_torch = {k: (bf16_ref[k] - v).norm().item() / max(v.norm().item(), 1e-12) for k, v in ref_g.items()}
        for k in ref_g:
            assert e_ours[k] < 1.5 * e_torch[k] + 0.05, (k, e_ours[k], e_torch[k])
        assert sum(e_ours.values()) < 1.1 * sum(e_torch.values()) + 0.01 * len(e_ours)

        def proj(a, v):
            v = v.flatten().double()
            return (a.flatten().double() @ v).item() / (v @ v).item()
        for k, v in ref_g.items():
            if v.numel() >= 4096:
                po, pt = proj(g[k], v), proj(bf16_ref[k], v)
                assert abs(po - pt) < 0.05 and 0.85 < po < 1.1, (k, po, pt)
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
