"""python run_i3.py <bad lib> <good lib> [batch]: cells of SOFT_COUNT the bad library leaves at 0 after the first reset, with their flows and limits;
repeated over fresh engines to see whether the set of cells moves."""
import sys, os, json
import numpy as np
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from engine_checks import load_env, engine_with_library
from helpers import ENVS
from pypownet_amd.batched import default_assignment
bad, good = [os.path.join(ROOT, p) for p in sys.argv[1:3]]
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64
first = int(sys.argv[4]) if len(sys.argv) > 4 else 0
case, cfg, chronics = load_env('default118', conf={'solver': 'newton'})
limits = np.asarray(json.load(open(os.path.join(ENVS, 'default118', 'bench_limits.json')))['limits_a'])
mk = lambda lp: engine_with_library(lp, case, cfg, batch, chronics=chronics, thermal_limits=limits)
slots, t0 = default_assignment((first + np.arange(batch)) * 5, chronics)
print('mk good', flush=True); g = mk(good); print('reset good', flush=True); g.reset(chronic_slot=slots, t0=t0); g.sync(); print('good reset done', flush=True)
gs, ga = g.read('SOFT_COUNT'), g.read('AMPS')
sets = []
for rep in range(4):
    print('mk bad', flush=True); b = mk(bad); print('reset bad', flush=True); b.reset(chronic_slot=slots, t0=t0); b.sync(); print('bad reset done', flush=True)
    bs, ba = b.read('SOFT_COUNT'), b.read('AMPS')
    idx = np.argwhere(bs != gs)
    sets.append(set(map(tuple, idx.tolist())))
    print('rep %d: %d cells differ; AMPS equal in %d of %d envs' % (rep, len(idx), int((ba == ga).all(axis=1).sum()), batch))
    if rep == 0:
        for e, l in idx[:60]:
            print('  env %2d line %3d  bad soft %d good soft %d  amps bad %.9g good %.9g  limit %.9g  ratio %.6f' % (e, l, bs[e, l], gs[e, l], ba[e, l], ga[e, l], limits[l], ga[e, l] / limits[l]))
        over_good = (ga > limits[None, :])
        print('  good: over cells %d, soft==1 cells %d; bad soft==1 cells %d' % (over_good.sum(), (gs == 1).sum(), (bs == 1).sum()))
        byenv = {}
        for e, l in idx: byenv.setdefault(int(e), []).append(int(l))
        print('  by env:', byenv)
        print('  over lines per env (good), envs with diffs:', {e: np.flatnonzero(over_good[e]).tolist() for e in byenv})
    del b
print('common to all reps: %d, union: %d' % (len(set.intersection(*sets)), len(set.union(*sets))))
