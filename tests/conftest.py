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
