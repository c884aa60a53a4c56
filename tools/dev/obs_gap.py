#!/usr/bin/env python
"""Where does a closed-loop step WITH its observation spend its time?  60 x ppn_step_observe (bench.py's loop) -- run under
`rocprofv3 --kernel-trace` and pass the trace to `--trace` to get kernel durations and the gaps between consecutive kernels.
    python tools/dev/obs_gap.py [steps]            |  python tools/dev/obs_gap.py --trace <kernel_trace.csv>"""
import csv
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT)


def trace(path):
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r['Start_Timestamp']))
    rows = [r for r in rows if 'ppn_' in r['Kernel_Name'] or 'rocclr' in r['Kernel_Name']]
    tail = rows[-400:]
    by = {}
    prev_end = None
    for r in tail:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        k = r['Kernel_Name'][:50]
        d = by.setdefault(k, [0, 0.0, 0.0])
        d[0] += 1; d[1] += (e - s) / 1e3
        if prev_end is not None:
            d[2] += max(0, s - prev_end) / 1e3
        prev_end = e
    span = (int(tail[-1]['End_Timestamp']) - int(tail[0]['Start_Timestamp'])) / 1e3
    print('last %d kernels span %.1f us' % (len(tail), span))
    for k, (n, dur, gap) in sorted(by.items(), key=lambda kv: -kv[1][1]):
        print('  %-52s x %4d: %9.1f us running (avg %7.1f), %8.1f us idle in front of them (avg %5.1f)' % (k, n, dur, dur / n, gap, gap / n))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == '--trace':
        return trace(sys.argv[2])
    import torch
    import bench
    from pypownet_amd.engine import Engine
    K = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    B = 4096
    case, conf, chronics = bench.load_workload()
    eng = Engine(case, conf, B, device=0, chronics=chronics, thermal_limits=bench.bench_limits(case), max_active_buses=case.nS)
    slots, t0 = bench.env_assignment(0, B, chronics)
    eng.reset(chronic_slot=slots, t0=t0)
    act = torch.zeros((B, case.action_length), dtype=torch.uint8, device='cuda')
    obs = torch.empty((B, eng.observation_length('full')), dtype=torch.float64, device='cuda')
    nb = obs.numel() * 8
    torch.cuda.synchronize()
    for form in ('step (auto_reset=2)', 'step (auto_reset=1)', 'step_observe (auto_reset=1)'):
        for rep in range(2):
            eng.sync()
            c0 = int(eng.read('N_STEPS').astype(np.int64).sum())
            t = time.perf_counter()
            for _ in range(K):
                if form.startswith('step_observe'):
                    eng.step_observe_device(act.data_ptr(), obs.data_ptr(), nb, auto_reset=True, layout='full', dtype=np.float64)
                else:
                    eng.step_device(act.data_ptr(), auto_reset=2 if '=2' in form else 1)
            eng.sync()
            el = time.perf_counter() - t
            n = int(eng.read('N_STEPS').astype(np.int64).sum()) - c0
            print('%-28s %6.1f us per call, %.2f M env-steps/s executed (%d of %d)' % (form, el / K * 1e6, n / el / 1e6, n, B * K), flush=True)
    eng.close()


if __name__ == '__main__':
    main()
