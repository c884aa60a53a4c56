"""python run_ii_poison.py (in build/wt_ii or build/wt_ii_fixed): incident (ii)'s failing check -- the default118_soft reference replay on the GPU -- with
tools/ubench/register_poison.hip (build/libppn_poison.so of the main tree) run in front of every engine step, once per pattern.  A tree whose kernels are right
replays the run under every pattern; incident (ii)'s tree fails, and where it fails may move with the pattern."""
import sys, os, ctypes
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
torch.zeros(1, device='cuda')
import reference_replay as rr
from pypownet_amd.engine import Engine
poison = ctypes.CDLL(os.path.join(ROOT, '..', 'libppn_poison.so')).ppn_poison
poison.argtypes = [ctypes.c_uint]; poison.restype = ctypes.c_int
pattern = [None]
orig_step, orig_reset = Engine.step, Engine.reset
def step(self, *a, **k):
    if pattern[0] is not None: assert poison(pattern[0]) == 0
    return orig_step(self, *a, **k)
def reset(self, *a, **k):
    if pattern[0] is not None: assert poison(pattern[0]) == 0
    return orig_reset(self, *a, **k)
Engine.step, Engine.reset = step, reset
for pat in (None, 0x0, 0x7ff7a5a5, 0x3ff00000, 0xffffffff):
    pattern[0] = pat
    try:
        c = rr.replay_engine(None, 'default118_soft', batch=3)
        print('pattern %s: replayed (%s)' % ('none' if pat is None else hex(pat), {k: c[k] for k in sorted(c)[:4]}), flush=True)
    except AssertionError as ex:
        print('pattern %s: FAIL %s' % ('none' if pat is None else hex(pat), str(ex)[:120].replace('\n', ' ')), flush=True)
