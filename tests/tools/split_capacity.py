#!/usr/bin/env python
"""Developer aid (GPU box): the node-splitting workload of BASELINE.json configs[4] at several matrix capacities
(rules.lu_capacity): LDS bytes per environment, environments per CU, env-steps/s, capacity flags raised over the run.

usage: python tests/tools/split_capacity.py <batch> <steps> <lu_capacity> [<lu_capacity> ...]      (0 = default)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    batch, steps = int(sys.argv[1]), int(sys.argv[2])
    import torch
    import bench
    from pypownet_amd.engine import Engine
    case, conf, chronics = bench.load_env_fixture(bench.ENV_NAME, 'newton')
    lim = bench.bench_limits(case)
    for cap in [int(a) for a in sys.argv[3:]]:
        eng = Engine(case, conf, batch, chronics=chronics, thermal_limits=lim, lu_capacity=cap)
        slots, t0 = bench.env_assignment(0, batch, chronics)
        eng.reset(chronic_slot=slots, t0=t0)
        rng = np.random.default_rng(1234)
        acts = [torch.from_numpy(bench.random_node_splitting(case, rng, batch)).to('cuda') for _ in range(8)]
        flags = 0
        for k in range(30):      # warm-up, capacity flags watched every step
            eng.step_device(acts[k % 8].data_ptr(), auto_reset=2)
            flags += int((eng.read('FLAG') == 4).sum())
        eng.sync()
        eng.kernel_time(reset=True)
        t = time.perf_counter()
        for k in range(steps):
            eng.step_device(acts[k % 8].data_ptr(), auto_reset=2)
        eng.sync()
        el = time.perf_counter() - t
        kms, kn = eng.kernel_time(reset=True)
        for k in range(30):
            eng.step_device(acts[k % 8].data_ptr(), auto_reset=2)
            flags += int((eng.read('FLAG') == 4).sum())
        print(json.dumps({'lu_capacity': cap, 'batch': batch, 'lds_bytes_per_env': eng.lds_bytes, 'envs_per_cu_by_granules': 128 // -(-eng.lds_bytes // 1280),
                          'ECAP': eng.dim(12), 'QCAP': eng.dim(15), 'env_steps_per_s': batch * steps / el, 'step_kernel_ms': kms / max(kn, 1),
                          'capacity_flags_in_60_watched_steps': flags}), flush=True)
        eng.close()


if __name__ == '__main__':
    main()
