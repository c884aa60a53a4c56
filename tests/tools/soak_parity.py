#!/usr/bin/env python
"""Long-horizon lock-step of the GPU engine against the C oracle on the bench workload (do-nothing agent, cascade limits,
auto reset): flags, line status, counters and chronic positions bit-exact, voltages <= 1e-8, over many chronic roll-overs
and restarts.  Usage (GPU box): python tests/tools/soak_parity.py [batch] [steps] [check_every]"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pypownet_amd.engine import Engine  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from harness import oracle_engine  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
    every = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    case, conf, chronics = bench.load_workload()
    conf['solver'] = os.environ.get('PPN_SOAK_SOLVER', conf['solver'])      # (fdxb: the reference's own solver)
    lim = bench.bench_limits(case)
    eng = Engine(case, conf, B, chronics=chronics, thermal_limits=lim, max_active_buses=case.nS)
    orc = oracle_engine(case, conf, B, chronics=chronics, thermal_limits=lim)
    slots, t0 = bench.env_assignment(0, B, chronics)
    eng.reset(chronic_slot=slots, t0=t0)
    orc.reset(chronic_slot=slots, t0=t0)
    act = np.zeros((B, case.action_length), dtype=np.uint8)
    worst = 0.0
    AR = int(os.environ.get('PPN_SOAK_AUTO_RESET', '1'))      # (2 + PPN_RESTART_MEMO=1: the restart memo against the oracle, which knows none)
    for t in range(steps):
        eng.step(act, auto_reset=AR if AR == 2 else True)
        orc.step(act, auto_reset=True)
        if (t + 1) % every and t + 1 != steps:
            continue
        for f in ('DONE', 'FLAG', 'LINES_STATUS', 'RECONNECTABLE', 'SOFT_COUNT', 'CHRONIC_ROW', 'CHRONIC_SLOT', 'N_SOLVES',
                  'N_ITERS', 'CASCADE_DEPTH', 'N_LOADS_CUT', 'N_PRODS_CUT'):
            a, b = eng.read(f), orc.read(f)
            if not np.array_equal(a, b):
                bad = np.where((a != b).reshape(B, -1).any(axis=1))[0]
                raise SystemExit('step %d: %s differs for environments %s' % (t, f, bad[:10]))
        live = orc.read('BUS_TYPE') != 4
        dv = np.abs(eng.read('VM')[live] - orc.read('VM')[live]).max()
        da = np.abs(np.deg2rad(eng.read('VA')[live]) - np.deg2rad(orc.read('VA')[live])).max()
        worst = max(worst, dv, da)
        assert dv <= 1e-8 and da <= 1e-8, (t, dv, da)
    print('soak ok: %d environments x %d steps, %d solves, max |dV| %.2e%s'
          % (B, steps, int(orc.read('N_SOLVES').astype(np.int64).sum()), worst,
             ('; restart memo: %r' % eng.restart_memo_stats()) if os.environ.get('PPN_RESTART_MEMO') else ''))


if __name__ == '__main__':
    main()
