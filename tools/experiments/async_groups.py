#!/usr/bin/env python
"""Experiment: the batch split into G groups of B/G environments, each group an engine of its own on its own HIP stream, every
group stepped K times without waiting for the others (asynchronous vectorised environments).  A launch lasts as long as its
longest chain; with G independent launch sequences a long chain only holds up its own group.

    python tools/experiments/async_groups.py [env] [batch] [steps] [G ...]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench  # noqa: E402


def run(envname, batch, steps, G, warmup=5):
    import torch
    from pypownet_amd.engine import Engine
    case, conf, chronics = bench.load_env_fixture(envname, 'newton')
    kw = {}
    limits = None
    if envname == 'default118':
        limits = bench.bench_limits(case)
        kw['max_active_buses'] = case.nS
    per = batch // G
    engs = []
    for g in range(G):
        e = Engine(case, conf, per, device=0, chronics=chronics, thermal_limits=limits, **kw)
        slots, t0 = bench.env_assignment(g * per, per, chronics)
        e.reset(chronic_slot=slots, t0=t0)
        engs.append(e)
    act = torch.zeros((per, case.action_length), dtype=torch.uint8, device='cuda:0')
    torch.cuda.synchronize()
    for _ in range(warmup):
        for e in engs:
            e.step_device(act.data_ptr(), auto_reset=2)
    for e in engs:
        e.sync()
    n0 = sum(int(e.read('N_STEPS').astype(np.int64).sum()) for e in engs)
    t = time.perf_counter()
    for _ in range(steps):
        for e in engs:
            e.step_device(act.data_ptr(), auto_reset=2)
    for e in engs:
        e.sync()
    el = time.perf_counter() - t
    n1 = sum(int(e.read('N_STEPS').astype(np.int64).sum()) for e in engs)
    for e in engs:
        e.close()
    return {'env': envname, 'batch': batch, 'groups': G, 'steps': steps, 'env_steps_per_s': (n1 - n0) / el,
            'ms_per_step_of_all_groups': 1e3 * el / steps}


if __name__ == '__main__':
    envname = sys.argv[1] if len(sys.argv) > 1 else 'default118'
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    Gs = [int(v) for v in sys.argv[4:]] or [1, 2, 4, 8, 16]
    for G in Gs:
        print(json.dumps(run(envname, batch, steps, G)), flush=True)
