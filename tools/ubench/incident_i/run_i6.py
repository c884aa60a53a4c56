"""python run_i6.py <lib> [...]: per library two engines, each with tools/ubench/register_poison.hip (build/libppn_poison.so of the main tree) run in front of every
engine call -- every vector register, AGPR and LDS byte of the chip holds the engine's pattern when its kernel starts -- with two different patterns (0 and 0x7ff7a5a5).
A kernel whose result depends on what an earlier kernel left there (incident (i)) gives two different answers, every time; a correct one cannot tell the patterns apart.
(Registers survive from one PROCESS to the next on this machine: a plain run is a run poisoned by whoever ran before.)"""
import sys, os, json, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from engine_checks import load_env, engine_with_library
from helpers import ENVS
from pypownet_amd import _lib
from pypownet_amd.batched import default_assignment
poison = ctypes.CDLL(os.path.join(ROOT, '..', 'libppn_poison.so')).ppn_poison
poison.argtypes = [ctypes.c_uint]; poison.restype = ctypes.c_int
batch, steps = 64, 6
case, cfg, chronics = load_env('default118', conf={'solver': 'newton'})
limits = np.asarray(json.load(open(os.path.join(ENVS, 'default118', 'bench_limits.json')))['limits_a'])
slots, t0 = default_assignment(np.arange(batch) * 5, chronics)
FIELDS = [f for f in _lib.FIELD_ID if f != 'OBSERVATION']
act = np.zeros((batch, case.action_length), dtype=np.uint8)
for lib in sys.argv[1:]:
    mk = lambda: engine_with_library(os.path.join(ROOT, lib), case, cfg, batch, chronics=chronics, thermal_limits=limits)
    for rep in range(3):
        p, q = mk(), mk()
        assert poison(0) == 0
        p.reset(chronic_slot=slots, t0=t0); p.sync()
        assert poison(0x7ff7a5a5) == 0
        q.reset(chronic_slot=slots, t0=t0); q.sync()
        bad = set()
        for t in range(steps):
            assert poison(0) == 0
            p.step(act, auto_reset=True); p.sync()
            assert poison(0x7ff7a5a5) == 0
            q.step(act, auto_reset=True); q.sync()
        for f in FIELDS:
            u, v = p.read(f), q.read(f)
            ne = ~((u == v) | (np.isnan(u) & np.isnan(v))) if u.dtype.kind == 'f' else (u != v)
            if ne.any(): bad.add(f)
        a = q.read('AMPS')
        print('%s rep %d: fields that differ between the two patterns: %d %s; finite AMPS rows under 0x7ff7a5a5 %d of %d' % (
            lib, rep, len(bad), sorted(bad)[:6], int(np.isfinite(a).all(axis=1).sum()), batch), flush=True)
        del p, q
