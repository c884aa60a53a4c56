#!/usr/bin/env python
"""How much of the engine's capacities (LU block entries, schedule records) do node-splitting topologies use?
Drives random RandomNodeSplitting-style actions (reference pypownet/agent.py:116-158) for many steps and reports the
largest filled pattern / record counts any schedule rebuild produced, next to the capacities ppn_create derived.

  python tools/fill_survey.py [batch] [steps] [p_split]      [single]   (GPU box)
"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))


def main():
    import bench
    from bench_configs import random_node_splitting
    from pypownet_amd.engine import Engine
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    mass = not (len(sys.argv) > 3 and sys.argv[3] == 'single')   # 'single': one substation per step only (longer episodes)
    case, conf, chronics = bench.load_workload()
    conf = dict(conf)
    # no cooldowns, no limits on simultaneous actions: the survey wants as many busbars active as the rules can ever allow
    conf['n_timesteps_actionned_node_reactionable'] = 0
    conf['max_number_actionned_substations'] = case.nS
    conf['max_number_actionned_total'] = case.action_length
    eng = Engine(case, conf, B, chronics=chronics, max_active_buses=2 * case.nS)
    slots, t0 = bench.env_assignment(0, B, chronics)
    eng.reset(chronic_slot=slots, t0=t0)
    rng = np.random.default_rng(4321)
    mx = np.zeros(7, dtype=np.int64)
    cap_flags = 0
    nb_hist = []
    nbytes = int(eng._lib.ppn_field_bytes(eng._h, 101))
    blob = np.empty((B, nbytes), dtype=np.uint8)
    for k in range(steps):
        if mass and k % 3 == 0:   # every third step: re-draw the configuration of MANY substations at once
            act = np.zeros((B, case.action_length), dtype=np.uint8)
            for s in rng.choice(case.nS, size=(case.nS * (1 + (k // 3) % 4)) // 4, replace=False):
                idx = np.asarray(case.mapping_array[int(s)], dtype=int)
                act[:, idx] = rng.integers(0, 2, size=(B, len(idx)))
        else:
            act = random_node_splitting(case, rng, B)
        eng.step(act, auto_reset=False)
        eng._check(eng._lib.ppn_read(eng._h, 101, blob.ctypes.data, blob.nbytes, 1, 0), 'read schedule caches')
        hdr = blob[:, :64].copy().view(np.int32)
        mx = np.maximum(mx, hdr[:, :7].max(axis=0))
        cap_flags += int((eng.read('FLAG') == 4).sum())
        nb_hist.append(hdr[:, 1].copy())
        eng.process_game_over()
    print('max over %d env-steps: valid %d, active buses %d, filled block entries %d, Ybus entries %d, level records %d, pairs %d, triples %d'
          % ((B * steps,) + tuple(int(v) for v in mx)))
    nb = np.concatenate(nb_hist)
    print('active buses of the schedules: mean %.1f, p50 %d, p99 %d, max %d' % (nb.mean(), np.percentile(nb, 50), np.percentile(nb, 99), nb.max()))
    print('capacities: filled block entries %d (base case %d), pairs %d, triples %d, LDS %d bytes/env; capacity flags raised: %d'
          % (eng.dim(12), eng.dim(11), eng.dim(13), eng.dim(14), eng.lds_bytes, cap_flags))


if __name__ == '__main__':
    main()
