#!/usr/bin/env python
"""Closed loop with an EXTERNAL (torch) policy on every environment's own clock: ppn_send / ppn_recv (include/ppn.h) on the bench
workload -- env-steps/s for a sweep of min_ready / server sizes, next to the synchronous ppn_step_observe loop.

    python tools/async_rate.py [batch] [steps] [min_ready,...] [workgroups,...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench  # noqa: E402


def main():
    import torch
    from pypownet_amd.engine import Engine
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    mrs = [int(v) for v in sys.argv[3].split(',')] if len(sys.argv) > 3 else [64, 256, 1024]
    wgs = [int(v) for v in sys.argv[4].split(',')] if len(sys.argv) > 4 else [0]
    case, conf, chronics = bench.load_workload()
    limits = bench.bench_limits(case)
    for wg in wgs:
        for mr in mrs:
            for mode in ('noop', 'reads_rows'):
                eng = Engine(case, conf, B, device=0, chronics=chronics, thermal_limits=limits, max_active_buses=case.nS)
                slots, t0 = bench.env_assignment(0, B, chronics)
                eng.reset(chronic_slot=slots, t0=t0)
                r = bench.async_rate(eng, case, B, K, min_ready=mr, workgroups=wg, mode=mode)
                print('batch %d steps %d min_ready %4d workgroups %4d (%4d resident) policy %-10s: %.3f M env-steps/s, %d receives, mean %.0f per receive, %d restarts' % (
                    B, K, mr, wg, r['workgroups'], mode, r['rate'] / 1e6, r['receives'], r['per_receive'], r['restarts']), flush=True)
                eng.close()


if __name__ == '__main__':
    main()
