#!/usr/bin/env python
"""VERDICT r05 #4, first half: is there a PROVABLE early exit from a Newton solve that will not converge?  Once the iterate reaches a
state from which the remaining iterations are decided in floating point -- a NaN / infinite mismatch (the engine already leaves there),
an update that is exactly zero with the mismatch above the tolerance, or an iterate that repeats an earlier one bit for bit -- the
flag is known.  The C oracle (ORC_CYCLE_STATS) counts how many of the solves that run out of PF_MAX_IT on the bench workload qualify.
CPU only:  python tests/tools/divergence_fixed_points.py [batch] [steps]"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ['ORC_CYCLE_STATS'] = '1'
from harness import oracle_engine  # noqa: E402
from helpers import load_env, ENVS  # noqa: E402
from pypownet_amd.batched import default_assignment  # noqa: E402


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    for limits in ('bench_limits.json', 'bench_limits_110.json'):
        case, cfg, chronics = load_env('default118', conf={'solver': 'newton'})
        with open(os.path.join(ENVS, 'default118', limits)) as f:
            lim = np.asarray(json.load(f)['limits_a'])
        orc = oracle_engine(case, cfg, batch, chronics=chronics, thermal_limits=lim)
        slots, t0 = default_assignment(np.arange(batch), chronics)
        out = (C.c_longlong * 6)()
        orc._lib._lib.orc_debug_cycle_stats(out)
        base = list(out)
        orc.reset(chronic_slot=slots, t0=t0)
        act = np.zeros((batch, case.action_length), dtype=np.uint8)
        for _ in range(steps):
            orc.step(act, auto_reset=True)
        orc._lib._lib.orc_debug_cycle_stats(out)
        d = [int(b) - int(a) for a, b in zip(base, out)]
        print('%s: %d environments x %d steps: %d Newton solves, %d ran out of their iterations; of those: update exactly zero %d, '
              'iterate repeats an earlier one bit for bit %d, NaN / infinite mismatch (the engine leaves early there) %d' % (
                  limits, batch, steps, d[0], d[1], d[2], d[3], d[4]), flush=True)


if __name__ == '__main__':
    main()
