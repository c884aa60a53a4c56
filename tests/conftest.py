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


# ---- PPN_TEST_POISON=<pattern>: the whole GPU suite under the register poison ---------------------------------------------------------
# tools/ubench/register_poison.hip (build/libppn_poison.so) leaves <pattern> in every VGPR, AGPR and LDS byte of the chip in front of EVERY call into the
# product library that can launch a kernel.  A suite that is green under two different patterns has shown, entry point by entry point and against the
# oracle, that no result depends on what an earlier kernel left behind.  Not for the asynchronous session: its resident server kernel never lets the
# poison's device synchronisation return (those tests skip themselves under the poison).
_NO_POISON = ('ppn_create', 'ppn_destroy', 'ppn_last_error', 'ppn_version', 'ppn_dim', 'ppn_field_bytes', 'ppn_sync', 'ppn_wait', 'ppn_kernel_time',
              'ppn_observation_length')


class _PoisonedLibrary(object):
    def __init__(self, lib, poison, pattern):
        self.__dict__.update(_lib=lib, _poison=poison, _pattern=pattern)

    def __getattr__(self, name):
        f = getattr(self._lib, name)
        if not name.startswith('ppn_') or name in _NO_POISON:
            return f
        if 'async' in name or name in ('ppn_send', 'ppn_recv'):
            def skip(*a):
                pytest.skip('the asynchronous session keeps a kernel resident: no register poison next to it')
            return skip
        poison, pattern = self._poison, self._pattern

        def call(*a):
            rc = poison(pattern)
            assert rc == 0, 'ppn_poison: %d' % rc
            return f(*a)
        return call


@pytest.fixture(scope='session', autouse=True)
def _register_poison(_one_hip_runtime):
    pat = os.environ.get('PPN_TEST_POISON')
    if not pat:
        yield
        return
    import ctypes
    from pypownet_amd import _lib
    so = os.path.join(ROOT, 'build', 'libppn_poison.so')
    if not os.path.exists(so):
        import __graft_entry__ as ge
        ge.build_guards()
    poison = ctypes.CDLL(so).ppn_poison
    poison.argtypes = [ctypes.c_uint]
    poison.restype = ctypes.c_int
    real = _lib.load_library
    _lib.load_library = lambda: _PoisonedLibrary(real(), poison, int(pat, 0))
    try:
        yield
    finally:
        _lib.load_library = real
