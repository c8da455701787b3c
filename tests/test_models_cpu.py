"""T0/T5: model forward/backward parity of the framework (CPU reference ops) against torch.nn oracles, and
state_dict key/shape parity (checkpoints must load strict=True into the reference's torchvision models)."""
import pytest
import torch

from distributedpytorch_amd.models.factory import get_model
from tests import torch_ref_models as ref
from tests.helpers import framework_step, grads_by_name, oracle_step


def test_resnet18_parity_small():
    _compare_zoo("resnet", ref.resnet18(10, small_input=True), 16, 16, small_input=True)


def test_resnet18_parity_imagenet_stem():
    _compare_zoo("resnet", ref.resnet18(10), 64, 8)


def test_resnet50_parity():
    oracle = ref.resnet50(10)
    # random-init ResNet-50 without zero-init residuals explodes (fp32 vs fp64 oracles disagree by several %):
    # damp the residual branches so the comparison is well conditioned
    with torch.no_grad():
        for k, v in oracle.named_parameters():
            if k.endswith("bn3.weight"):
                v.fill_(0.2)
    _compare_zoo("resnet50", oracle, 64, 8)


def _no_dropout(m):
    for mod in m.modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.p = 0.0
    return m


def _zero_dropout(net):
    from distributedpytorch_amd.nn.layers import Dropout
    def walk(o, seen):
        if id(o) in seen or isinstance(o, (torch.Tensor, str, int, float, type(None))):
            return
        seen.add(id(o))
        if isinstance(o, Dropout):
            o.p = 0.0
        elif isinstance(o, (list, tuple)):
            for x in o:
                walk(x, seen)
        elif isinstance(o, dict):
            for x in o.values():
                walk(x, seen)
        elif hasattr(o, "__dict__") and o.__class__.__module__.startswith("distributedpytorch_amd.models"):
            for x in vars(o).values():
                walk(x, seen)
    walk(net.net, set())


def _compare_zoo(name, oracle, hw, n, gtol=2e-2, small_input=False, seed=0):
    """Parity with dropout disabled on both sides (masks come from different RNG streams by design).
    `oracle`: a module, or a callable building one (then built under `seed`, so a test can be parametrised over several
    initialisations instead of relying on the one draw the autouse seeding fixture pins - ADVICE r2)."""
    torch.manual_seed(seed)
    if not isinstance(oracle, torch.nn.Module):
        oracle = oracle()
    net, _ = get_model(name, 10, device="cpu", small_input=small_input)
    _zero_dropout(net)
    _no_dropout(oracle)
    sd = oracle.state_dict()
    mine = net.state_dict(prefix="")
    assert list(mine.keys()) == list(sd.keys())
    for k in sd:
        assert tuple(mine[k].shape) == tuple(sd[k].shape), k
    net.load_state_dict(sd)
    x = torch.randn(n, 3, hw, hw)
    y = torch.randint(0, 10, (n,))
    logits, loss, acc = framework_step(net, x, y)
    oracle = oracle.double()
    ref_logits, ref_loss = oracle_step(oracle, x.double(), y)
    torch.testing.assert_close(logits, ref_logits.float(), rtol=2e-3, atol=2e-4)
    assert abs(loss - ref_loss) < 1e-4 * max(1.0, abs(ref_loss))
    g = grads_by_name(net)
    gmax = max(p.grad.norm().item() for p in oracle.parameters())
    # the same step in torch's OWN fp32: where the problem is ill-conditioned (a ReLU pre-activation at the rounding level,
    # BatchNorm over 8 values at 1x1 spatial size) any fp32 implementation misses a fixed bound for some draws - such a
    # parameter passes if our distance to fp64 is within twice torch-fp32's own distance to fp64
    import copy
    o32 = copy.deepcopy(oracle).float()
    for q in o32.parameters():
        q.grad = None
    oracle_step(o32, x, y)
    g32 = {k: q.grad.detach() for k, q in o32.named_parameters()}
    for k, p in oracle.named_parameters():
        v = p.grad.float()
        err = (g[k] - v).norm().item()
        err32 = (g32[k] - v).norm().item()
        # a conv bias followed by BatchNorm has an exactly-zero gradient: only absolute error is meaningful there
        assert err / max(v.norm().item(), 1e-12) < gtol or err < 1e-5 * gmax or err < 2 * err32, (k, err, err32, v.norm().item())
    # BatchNorm running statistics were updated like torch's, and eval mode uses them
    mine2 = net.state_dict(prefix="")
    for k, v in oracle.state_dict().items():
        if "running" in k or "num_batches" in k:
            torch.testing.assert_close(mine2[k].float(), v.float(), rtol=1e-4, atol=1e-5)
    logits_e, _, _ = framework_step(net, x, y, train=False)
    ref_e, _ = oracle_step(oracle, x.double(), y, train=False)
    torch.testing.assert_close(logits_e, ref_e.float(), rtol=2e-3, atol=2e-4)


def test_alexnet_parity():
    _compare_zoo("alexnet", ref.AlexNet(10), 64, 4)


def test_vgg11bn_parity():
    _compare_zoo("vgg", ref.VGG11BN(10), 32, 8)


def test_squeezenet_parity():
    _compare_zoo("squeezenet", ref.SqueezeNet10(10), 64, 4)


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4])
def test_densenet121_parity(seed):
    """Five initialisations / inputs, not one lucky draw: at 32x32 the last dense block runs BatchNorm over 8 values per
    channel (1x1 spatial, batch 8) - the test that failed one run in six before the bound was tied to torch-fp32's own
    distance to fp64."""
    _compare_zoo("densenet", lambda: ref.DenseNet121(10), 32, 8, seed=seed)


def test_inception_v3_parity():
    # torch's own fp32 backward differs from its fp64 backward by 2.1 % on this net at batch 4 (ReLU masks and 1x1-spatial
    # BatchNorm in the aux head are ill-conditioned), so the bound is looser than for the other architectures
    _compare_zoo("inception", ref.Inception3(10), 299, 4, gtol=8e-2)
