#!/usr/bin/env python
"""Topology-action search on one GPU (ppn_simulate_candidates: K candidate actions per environment forked from its current state and
simulated in one launch -- what the reference's search agents do with one Game.simulate per candidate, agent.py:161-325):
simulated candidates per second on default118, every busbar may be active.
    python tools/search_rate.py [batch] [K] [rounds] [node|line]"""
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
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    R = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    kind = sys.argv[4] if len(sys.argv) > 4 else 'node'
    case, conf, chronics = bench.load_env_fixture(bench.ENV_NAME, 'newton')
    eng = Engine(case, conf, B, chronics=chronics, thermal_limits=bench.bench_limits(case))
    slots, t0 = bench.env_assignment(0, B, chronics)
    eng.reset(chronic_slot=slots, t0=t0)
    rng = np.random.default_rng(5)
    cands = np.zeros((B * K, case.action_length), dtype=np.uint8)
    for c in range(B * K):
        if kind == 'node':
            idx = np.asarray(case.mapping_array[int(rng.integers(case.nS))], dtype=int)
            cands[c, idx] = rng.integers(0, 2, size=len(idx))
        else:
            cands[c, case.nP + case.nL + 2 * case.nl + int(rng.integers(case.nl))] = 1
    d_c = torch.from_numpy(cands).cuda()
    env_ids = np.repeat(np.arange(B, dtype=np.int32), K)
    act = torch.zeros((B, case.action_length), dtype=torch.uint8, device='cuda')
    torch.cuda.synchronize()
    for _ in range(2):
        eng.simulate_candidates_device(d_c.data_ptr(), env_ids)
        eng.step_device(act.data_ptr(), auto_reset=1)
    eng.sync()
    t = time.perf_counter()
    for _ in range(R):                       # a search agent's step: simulate the SAME candidate set, then (mostly) do nothing
        eng.simulate_candidates_device(d_c.data_ptr(), env_ids)
        eng.step_device(act.data_ptr(), auto_reset=1)
    eng.sync()
    el = time.perf_counter() - t
    print(json.dumps({'batch': B, 'candidates_per_env': K, 'kind': kind, 'rounds': R, 'ms_per_round': 1e3 * el / R,
                      'simulated_candidates_per_s': B * K * R / el, 'flag_capacity': int((eng.read('FLAG', simulation=2) == 4).sum())}))


if __name__ == '__main__':
    main()
