#!/usr/bin/env python
"""Synthesises the thermal limits of the benchmark workload (BASELINE.json configs[2]).

The shipped default118 limits are a flat 2000 A that no flow of the chronics reaches, so the cascading-failure
loop would never run.  Rule used here (deterministic, data only):
    limit_k = max(50, round(Q_0.98 over t of I_k(t)))   [A]
where I_k(t) are the origin-side ampere flows of a do-nothing run WITHOUT limits over every timestep of every
shipped default118 chronic (steps that end in a game over are skipped).  Each line is therefore overflowed in
about 2 % of its timesteps: soft overflows accumulate, some turn into cuts and re-solves (mean ~1.35 solves/step).
Written to tests/golden/envs/default118/bench_limits.json.  Uses the C oracle (build container only).
"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import load_env  # noqa: E402
from harness import oracle_engine  # noqa: E402

if __name__ == '__main__':
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'oracle')])
    lib = os.path.join(ROOT, 'oracle', '_build', 'liboracle.so')
    case, conf, chronics = load_env('default118', conf={'solver': 'newton'})
    rows = []
    for slot in range(len(chronics)):
        eng = oracle_engine(case, conf, 1, chronics=chronics, thermal_limits=np.full(case.nl, 1e9))
        eng.reset(chronic_slot=[slot], t0=[0])
        rows.append(eng.read('AMPS')[0].copy())
        act = np.zeros((1, case.action_length), dtype=np.uint8)
        for _ in range(chronics[slot].n_timesteps - 1):
            eng.step(act)
            if eng.read('DONE')[0]:
                eng.process_game_over()
            else:
                rows.append(eng.read('AMPS')[0].copy())
    A = np.array(rows)
    lim = np.maximum(50.0, np.round(np.quantile(A, 0.98, axis=0)))
    out = os.path.join(ROOT, 'tests', 'golden', 'envs', 'default118', 'bench_limits.json')
    with open(out, 'w') as f:
        json.dump({'rule': 'max(50, round(q0.98_t I_k(t))) over do-nothing no-limit runs of chronics %s'
                           % [c.name for c in chronics], 'n_samples': int(A.shape[0]),
                   'limits_a': [float(v) for v in lim]}, f)
    print('wrote', out, 'min/max', lim.min(), lim.max())
    # SURVEY.md 8d "config 3" as written: limit_k = max(50, 1.10 x I_k(base case at t = 0)) rounded to integer A -- the second
    # headline variant bench.py reports (cascade-heavier, fewer game overs)
    base = rows[0]                       # first timestep of the first chronic, every line in service
    lim110 = np.maximum(50.0, np.round(1.10 * base))
    out = os.path.join(ROOT, 'tests', 'golden', 'envs', 'default118', 'bench_limits_110.json')
    with open(out, 'w') as f:
        json.dump({'rule': 'max(50, round(1.10 x I_k(t = 0))) with I_k(t = 0) the ampere flows of the first timestep of chronic %s, '
                           'every line in service (SURVEY.md 8d config 3)' % chronics[0].name,
                   'limits_a': [float(v) for v in lim110]}, f)
    print('wrote', out, 'min/max', lim110.min(), lim110.max())
