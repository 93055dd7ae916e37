import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "retinanet-tensorflow2.x_amd")
for p in (PKG, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from retinanet import _C
    assert _C.lib().rn_device_ok() == 1, "librnet_hip.so loaded but no gfx950 device"
    return torch.device("cuda:0")


@pytest.fixture(scope="session")
def params():
    from retinanet.cfg import default_params
    return default_params()
