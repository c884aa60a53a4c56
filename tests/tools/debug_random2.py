"""Debug aid: replay random actions; at step T lift the thermal limits so that the step is a single solve, then compare."""
import os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import load_env
import engine_checks as ec
from pypownet_amd.engine import Engine

envname, T, batch, env = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
lib = sys.argv[5] if len(sys.argv) > 5 else None
case, cfg, chronics = load_env(envname, conf={'solver': 'newton'})
case.ntopo_offset_lines = case.nP + case.nL + 2 * case.nl
eng = Engine(case, cfg, batch, chronics=chronics, _lib_path=lib)
orc = Engine(case, cfg, batch, chronics=chronics, _lib_path=os.path.join(ROOT, 'oracle', '_build', 'liboracle.so'), _lib_prefix='orc_')
rng = np.random.default_rng(1234)
eng.reset(); orc.reset()
for t in range(T + 1):
    acts = ec.random_actions(case, rng, batch)
    if t == T:
        big = np.full(case.nl, 1e12)
        eng.set_thermal_limits(big); orc.set_thermal_limits(big)
    eng.step(acts, auto_reset=False); orc.step(acts, auto_reset=False)
    if t == T:
        e = env
        print('flags', eng.read('FLAG')[e], orc.read('FLAG')[e], 'iters', eng.read('N_ITERS')[e], orc.read('N_ITERS')[e])
        for f in ('VM', 'VA', 'AMPS', 'PF', 'QF'):
            a, b = eng.read(f)[e], orc.read(f)[e]
            d = np.abs(a - b)
            k = int(np.nanargmax(d))
            print('%-5s max|d| %.3e at %d (gpu %.9g orc %.9g)' % (f, np.nanmax(d), k, a[k], b[k]))
        a, b = eng.read('AMPS')[e], orc.read('AMPS')[e]
        print('amps line 61 gpu/orc', a[61], b[61], 'limit', eng.thermal_limits[61] if False else None)
        lims = ec  # noqa
    eng.process_game_over(); orc.process_game_over()
