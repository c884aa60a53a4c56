"""TEST INFRASTRUCTURE: drives libraries OTHER than the product's pypownet_amd/libppn.so through the product's Python
classes -- the g++ lane-serial emulation build of the kernel sources (build/libppn_emu.so), profiling / experimental
builds, and the C oracle (oracle/_build/liboracle.so, which exports the same signatures under an ``orc_`` prefix).

The product package has no notion of an alternative library: ``pypownet_amd._lib.load_library()`` takes no argument and
loads one fixed path.  Everything here works by temporarily replacing that function from the OUTSIDE; nothing under
pypownet_amd/ imports this module."""
import contextlib
import ctypes as C
import os

from pypownet_amd import _lib

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
ORACLE_LIB = os.path.join(ROOT, 'oracle', '_build', 'liboracle.so')
EMU_LIB = os.path.join(ROOT, 'build', 'libppn_emu.so')


class _Prefixed(object):
    """Attribute view that maps ppn_xxx onto <prefix>xxx."""

    def __init__(self, lib, prefix):
        self._lib, self._prefix = lib, prefix

    def __getattr__(self, name):
        if name.startswith('ppn_'):
            return getattr(self._lib, self._prefix + name[4:])
        return getattr(self._lib, name)


def bound_library(path, prefix=None):
    if not os.path.exists(path):
        raise ImportError('test harness: library %s is missing' % path)
    if prefix is None:
        prefix = 'orc_' if 'liboracle' in os.path.basename(path) else 'ppn_'
    lib = C.CDLL(path)
    if prefix != 'ppn_':
        lib = _Prefixed(lib, prefix)
    return _lib.bind_signatures(lib, full_abi=(prefix == 'ppn_'))


@contextlib.contextmanager
def library(path, prefix=None):
    """Objects of the product package created inside this context bind to ``path`` (None: the product library)."""
    if path is None:
        yield
        return
    lib = bound_library(path, prefix)
    saved = _lib.load_library
    _lib.load_library = lambda: lib
    try:
        yield
    finally:
        _lib.load_library = saved


def engine_with_library(path, *args, **kw):
    from pypownet_amd.engine import Engine
    with library(path):
        return Engine(*args, **kw)


def oracle_engine(*args, **kw):
    return engine_with_library(ORACLE_LIB, *args, **kw)
