import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun / the driver's GPU tier)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(params=["x6", "exact"])
def fp32_mode(request):
    """Run a GPU kernel test under both fp32 convolution arithmetics (x6 is the default, exact the alternative)."""
    from distributedpytorch_amd.ops.native import set_fp32_mode
    prev = set_fp32_mode(request.param)
    yield request.param
    set_fp32_mode(prev)


@pytest.fixture(autouse=True)
def _x6p_on_small_shapes(request):
    """Kernel-level GPU tests use small problems; the pre-split fp32 kernel (conv_x6p.hip) is only dispatched for >= 96
    tiles in production, so those tests lower the threshold to 1 tile for their duration."""
    mod = request.module.__name__.rsplit(".", 1)[-1]
    if mod not in ("test_gpu_conv", "test_gpu_x6"):
        yield
        return
    import torch
    if not torch.cuda.is_available():
        yield
        return
    from distributedpytorch_amd.ops import native
    lib = native.hip_lib()
    prev = lib.dpa_x6p_min_tiles(1)
    yield
    lib.dpa_x6p_min_tiles(prev)


@pytest.fixture(autouse=True)
def _seeded():
    """Every test starts from the same torch RNG state: oracles built in a test's argument list (random init) and random
    inputs are then the same on every run - a parity bound that holds for one draw must not fail on another (the 32x32
    DenseNet parity test, 1x1-spatial BatchNorm at batch 8, failed about one run in six before this)."""
    import torch
    torch.manual_seed(20240229)
    yield
