import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
ENVS = os.path.join(GOLDEN, 'envs')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def envs_dir():
    return ENVS


@pytest.fixture(scope='session', autouse=True)
def _one_hip_runtime():
    """PyTorch-ROCm bundles its own HIP runtime: it has to be the first one loaded in the process (pypownet_amd/_lib.py does the
    same for the product library; the test harness loads other libraries itself)."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass
    yield
