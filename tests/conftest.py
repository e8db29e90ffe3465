import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def rel_l2(a, b):
    """||a - b|| / ||b|| on float64 copies."""
    import numpy as np
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session', autouse=True)
def _no_kernel_reported_a_failure():
    """After the whole run on a GPU box: no kernel may have left a bit in the device's asynchronous status word
    (icn_device_status; a lost stream-K partner would otherwise only show as NaNs somewhere)."""
    yield
    try:
        import torch
        if not torch.cuda.is_available():
            return
        from geniconet_amd import _lib
    except Exception:
        return
    assert _lib.device_status() == 0, 'a kernel reported an asynchronous failure during the test run'
