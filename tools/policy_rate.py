#!/usr/bin/env python
"""Closed-loop stepping without the batch barrier (GPU box): env-steps/s of ppn_rollout_policy (device-side policy, every environment
on its own clock) next to the synchronous forms on the bench workload.
    python tools/policy_rate.py [batch] [steps]"""
import json
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
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    case, conf, chronics = bench.load_workload()
    out = {'batch': B, 'steps': K}
    for policy, params in (('do_nothing', []), ('line_relief', [1.0])):
        eng = Engine(case, conf, B, chronics=chronics, thermal_limits=bench.bench_limits(case), max_active_buses=case.nS)
        slots, t0 = bench.env_assignment(0, B, chronics)
        eng.reset(chronic_slot=slots, t0=t0)
        buf = torch.zeros((B, case.action_length), dtype=torch.uint8, device='cuda')
        torch.cuda.synchronize()
        for _ in range(5):
            eng.policy_actions(policy, params, buf.data_ptr())
            eng.step_device(buf.data_ptr(), auto_reset=1)
        eng.sync()
        # synchronous closed loop: policy kernel + step kernel per step
        n0 = int(eng.read('N_STEPS').astype(np.int64).sum())
        t = time.perf_counter()
        for _ in range(K):
            eng.policy_actions(policy, params, buf.data_ptr())
            eng.step_device(buf.data_ptr(), auto_reset=1)
        eng.sync()
        el = time.perf_counter() - t
        n1 = int(eng.read('N_STEPS').astype(np.int64).sum())
        out[policy + '_stepped_env_steps_per_s'] = (n1 - n0) / el
        # the same closed loop in one launch
        eng.rollout_policy(policy, params, 4)
        eng.sync()
        n0 = int(eng.read('N_STEPS').astype(np.int64).sum())
        t = time.perf_counter()
        eng.rollout_policy(policy, params, K)
        eng.sync()
        el = time.perf_counter() - t
        n1 = int(eng.read('N_STEPS').astype(np.int64).sum())
        out[policy + '_rollout_policy_env_steps_per_s'] = (n1 - n0) / el
        out[policy + '_envs_done_last_step'] = int(eng.read('DONE').sum())
        if policy == 'do_nothing':      # the open-loop rollout of round 3 for comparison (same agent)
            n0 = n1
            t = time.perf_counter()
            eng.rollout_device(buf.zero_().data_ptr(), K, per_step_actions=False, auto_reset=1)
            eng.sync()
            el = time.perf_counter() - t
            out['do_nothing_open_loop_rollout_env_steps_per_s'] = (int(eng.read('N_STEPS').astype(np.int64).sum()) - n0) / el
        eng.close()
    print(json.dumps(out))


if __name__ == '__main__':
    main()
