#!/usr/bin/env python
"""Rate of the observation gather (K_OBS) next to the step kernel: device-resident outputs, bench workload (GPU box).
    python tools/obs_rate.py [batch]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)


def main():
    import torch
    import bench
    from pypownet_amd.engine import Engine
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    case, conf, chronics = bench.load_workload()
    eng = Engine(case, conf, B, chronics=chronics, thermal_limits=bench.bench_limits(case), max_active_buses=case.nS)
    slots, t0 = bench.env_assignment(0, B, chronics)
    eng.reset(chronic_slot=slots, t0=t0)
    act = torch.zeros((B, case.action_length), dtype=torch.uint8, device='cuda')
    for name, lay, f32 in (('full f64', 0, 0), ('full f32', 0, 1), ('minimalist f32', 1, 1), ('ac_minimalist f32', 2, 1)):
        n = int(eng._lib.ppn_observation_length(eng._h, lay))
        out = torch.empty((B, n), dtype=torch.float32 if f32 else torch.float64, device='cuda')
        torch.cuda.synchronize()
        nb = out.numel() * out.element_size()
        for k in range(3):
            eng.step_device(act.data_ptr(), auto_reset=True)
            eng._check(eng._lib.ppn_read_observation(eng._h, lay, f32, out.data_ptr(), nb, 0, 0), 'obs')
        eng.sync()
        t = time.perf_counter()
        K = 20
        for k in range(K):
            eng.step_device(act.data_ptr(), auto_reset=True)
            eng._check(eng._lib.ppn_read_observation(eng._h, lay, f32, out.data_ptr(), nb, 0, 0), 'obs')
        eng.sync()
        el = (time.perf_counter() - t) / K
        print('%-18s %5d values/env: step + observation gather %.3f ms -> %.2f M env-steps/s (%.0f MB of observations per step)'
              % (name, n, 1e3 * el, B / el / 1e6, nb / 1e6))


if __name__ == '__main__':
    main()
