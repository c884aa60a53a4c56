#!/usr/bin/env python
"""Developer aid (GPU box): the same workload through two builds of libppn.so -- e.g. the shipped one and an experimental
one with another register budget -- env-steps/s and step-kernel time of each.  Not part of the product: the library path
is injected through the test harness (tests/harness.py).

usage: python tests/tools/lib_compare.py <env> <solver> <batch> <steps> <lib.so> [<lib.so> ...]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    envname, solver, batch, steps = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    libs = sys.argv[5:] or [None]
    import torch
    from helpers import load_env
    from harness import engine_with_library
    import bench
    case, conf, chronics = load_env(envname, conf={'solver': solver})
    kw = {}
    if envname == 'default118':
        kw = dict(thermal_limits=bench.bench_limits(case), max_active_buses=case.nS)
    for lib in libs:
        eng = engine_with_library(None if lib in ('default', 'None') else lib, case, conf, batch, chronics=chronics, **kw)
        ids = np.arange(batch)
        slots = (ids % len(chronics)).astype(np.int32)
        T = np.array([c.n_timesteps for c in chronics])[slots]
        eng.reset(chronic_slot=slots, t0=((ids * 37) % T).astype(np.int32))
        act = torch.zeros((batch, case.action_length), dtype=torch.uint8, device='cuda')
        torch.cuda.synchronize()
        for _ in range(5):
            eng.step_device(act.data_ptr(), auto_reset=2)
        eng.sync()
        eng.kernel_time(reset=True)
        s0, i0 = eng.read('N_SOLVES').astype(np.int64).sum(), eng.read('N_ITERS').astype(np.int64).sum()
        t = time.perf_counter()
        for _ in range(steps):
            eng.step_device(act.data_ptr(), auto_reset=2)
        eng.sync()
        el = time.perf_counter() - t
        kms, kn = eng.kernel_time(reset=True)
        s1, i1 = eng.read('N_SOLVES').astype(np.int64).sum(), eng.read('N_ITERS').astype(np.int64).sum()
        print(json.dumps({'lib': lib, 'env': envname, 'solver': solver, 'batch': batch, 'env_steps_per_s': batch * steps / el,
                          'step_kernel_ms': kms / max(kn, 1), 'lds_bytes_per_env': eng.lds_bytes, 'envs_per_cu': eng.dim(16),
                          'solves_per_step': float(s1 - s0) / (batch * steps),
                          'iters_per_solve': float(i1 - i0) / max(float(s1 - s0), 1.0),
                          'checksum_vm': float(np.nansum(eng.read('VM')))}), flush=True)
        eng.close()


if __name__ == '__main__':
    main()
