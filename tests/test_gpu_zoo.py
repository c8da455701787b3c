"""T3: every architecture of the zoo on the HIP kernels, in both fp32 arithmetics, against fp64 torch oracles, plus a bf16
run of each for finiteness and loss agreement.

With 4-8 images per batch some ReLU pre-activations sit at the rounding level, and WHICH of them end up on the other side
of zero than in fp64 depends on the implementation's rounding: the exact-fp32 MFMA path flips the same ones as torch's own
fp32 (its per-parameter errors equal torch-fp32's digit for digit, scripts/diag/densenet_grad_diag.py), so there the
criterion "within twice torch-fp32's own distance to fp64" covers them; the x6 path rounds differently (not worse:
tests/test_gpu_x6.py) and flips a DIFFERENT handful - one flipped unit moves every gradient of that one layer by a few per
cent of its norm (measured: all six parameters of DenseNet's denseblock3.denselayer19 by 3.1-3.8 % for one of three
seeds, every other parameter of the network at 1e-5).  For x6 a handful of parameters may therefore miss the relative bound
gtol, provided each stays within 5 x gtol."""
import pytest
import torch

pytestmark = pytest.mark.gpu

ZOO = [("alexnet", "AlexNet", 64, 4, 2e-2), ("vgg", "VGG11BN", 32, 8, 2e-2), ("squeezenet", "SqueezeNet10", 64, 4, 2e-2),
       ("densenet", "DenseNet121", 64, 8, 3e-2), ("inception", "Inception3", 299, 4, 8e-2)]


@pytest.mark.parametrize("seed", [0, 1, 2])
@pytest.mark.parametrize("name,cls,hw,n,gtol", ZOO)
def test_zoo_gpu_fp32(name, cls, hw, n, gtol, seed, fp32_mode):
    """Three initialisations / inputs per architecture (not the one draw the autouse seeding fixture would pin)."""
    from distributedpytorch_amd.models.factory import get_model
    from tests import torch_ref_models as ref
    from tests.helpers import framework_step, grads_by_name, oracle_step
    from tests.test_models_cpu import _no_dropout, _zero_dropout
    torch.manual_seed(seed)
    oracle = _no_dropout(getattr(ref, cls)(10))
    net, _ = get_model(name, 10, device="cuda", dtype=torch.float32)
    _zero_dropout(net)
    net.load_state_dict(oracle.state_dict())
    x = torch.randn(n, 3, hw, hw)
    y = torch.randint(0, 10, (n,))
    logits, loss, acc = framework_step(net, x, y)
    oracle = oracle.double()
    ref_logits, ref_loss = oracle_step(oracle, x.double(), y)
    torch.testing.assert_close(logits, ref_logits.float(), rtol=5e-3, atol=5e-4)
    assert abs(loss - ref_loss) < 1e-3 * max(1.0, abs(ref_loss))
    g = grads_by_name(net)
    gmax = max(p.grad.norm().item() for p in oracle.parameters())
    # the same step in torch's own fp32 on the GPU: with 4-8 images per batch a ReLU whose pre-activation sits at the
    # rounding level flips between implementations, so a parameter may miss the fixed bound in ANY fp32 implementation;
    # such a parameter passes if our distance to fp64 is within twice torch-fp32's own distance to fp64
    import copy
    o32 = copy.deepcopy(oracle).float().cuda()
    for q in o32.parameters():
        q.grad = None
    oracle_step(o32, x.cuda(), y.cuda())
    g32 = {k: q.grad.detach().cpu() for k, q in o32.named_parameters()}
    flipped = []
    nparams = 0
    for k, p in oracle.named_parameters():
        v = p.grad.float()
        err = (g[k] - v).norm().item()
        err32 = (g32[k] - v).norm().item()
        nparams += 1
        if err / max(v.norm().item(), 1e-12) < gtol or err < 1e-5 * gmax or err < 2 * err32:
            continue
        rel = err / max(v.norm().item(), 1e-12)
        assert fp32_mode == "x6" and rel < 5 * gtol, (k, err, err32, v.norm().item(), gmax)
        flipped.append((k, rel))
    assert len(flipped) <= max(8, nparams // 50), flipped      # a handful of parameters of one or two layers, not a trend
    logits_e, _, _ = framework_step(net, x, y, train=False)
    ref_e, _ = oracle_step(oracle, x.double(), y, train=False)
    torch.testing.assert_close(logits_e, ref_e.float(), rtol=5e-3, atol=5e-3)


@pytest.mark.parametrize("name,cls,hw,n,gtol", ZOO)
def test_zoo_gpu_bf16_runs(name, cls, hw, n, gtol):
    from distributedpytorch_amd.models.factory import get_model
    from tests import torch_ref_models as ref
    from tests.helpers import framework_step, oracle_step
    from tests.test_models_cpu import _no_dropout, _zero_dropout
    torch.manual_seed(0)
    oracle = _no_dropout(getattr(ref, cls)(10))
    net, _ = get_model(name, 10, device="cuda", dtype=torch.bfloat16)
    _zero_dropout(net)
    net.load_state_dict(oracle.state_dict())
    x = torch.randn(n, 3, hw, hw)
    y = torch.randint(0, 10, (n,))
    logits, loss, acc = framework_step(net, x, y)
    _, ref_loss = oracle_step(oracle, x, y)
    assert torch.isfinite(logits).all() and torch.isfinite(net.store.g).all()
    assert abs(loss - ref_loss) < 0.08 * max(1.0, abs(ref_loss)), (loss, ref_loss)
