#!/usr/bin/env python
"""Side measurements on one GPU of the other workload configurations of BASELINE.json / SURVEY.md 8d (they are parity-test
cases, not the headline bench line): IEEE-14 AC Newton, IEEE-118 with the reference's fast-decoupled XB solver, IEEE-118
with a random node-splitting action per environment and step (every busbar may become active: W = 4 kernels, schedule
rebuilt whenever the node assignment changes).  One JSON line each.  Usage: python tools/bench_configs.py [steps]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)


def load(envname, solver):
    import yaml
    from pypownet_amd.case import Case
    from pypownet_amd.chronic import Chronic
    d = os.path.join(ROOT, 'tests', 'golden', 'envs', envname, 'level0')
    case = Case.from_file(os.path.join(d, 'reference_grid.json'))
    with open(os.path.join(d, 'configuration.yaml')) as f:
        conf = yaml.safe_load(f)
    conf['solver'] = solver
    cdir = os.path.join(d, 'chronics')
    chronics = [Chronic(os.path.join(cdir, c)) for c in sorted(os.listdir(cdir))]
    return case, conf, chronics


def random_node_splitting(case, rng, batch):
    """RandomNodeSplitting-style actions (reference pypownet/agent.py:116-158): one random substation per environment gets
    a random configuration of its switches."""
    acts = np.zeros((batch, case.action_length), dtype=np.uint8)
    subs = rng.integers(case.nS, size=batch)
    for b in range(batch):
        idx = np.asarray(case.mapping_array[int(subs[b])], dtype=int)
        acts[b, idx] = rng.integers(0, 2, size=len(idx))
    return acts


def run(name, envname, solver, batch, steps, limits=None, split=False, max_active=None, warmup=5):
    import torch
    from pypownet_amd.engine import Engine
    case, conf, chronics = load(envname, solver)
    kw = {}
    if max_active:
        kw['max_active_buses'] = max_active
    eng = Engine(case, conf, batch, chronics=chronics, thermal_limits=limits, **kw)
    ids = np.arange(batch)
    slots = (ids % len(chronics)).astype(np.int32)
    T = np.array([c.n_timesteps for c in chronics])[slots]
    eng.reset(chronic_slot=slots, t0=((ids * 37) % T).astype(np.int32))
    rng = np.random.default_rng(1234)
    n_act = 8 if split else 1
    acts = [torch.from_numpy(random_node_splitting(case, rng, batch) if split
                             else np.zeros((batch, case.action_length), dtype=np.uint8)).cuda() for _ in range(n_act)]
    torch.cuda.synchronize()
    for k in range(warmup):
        eng.step_device(acts[k % n_act].data_ptr(), auto_reset=True)
    eng.sync()
    s0, i0 = eng.read('N_SOLVES').astype(np.int64).sum(), eng.read('N_ITERS').astype(np.int64).sum()
    eng.kernel_time(reset=True)
    t = time.perf_counter()
    for k in range(steps):
        eng.step_device(acts[k % n_act].data_ptr(), auto_reset=True)
    eng.sync()
    el = time.perf_counter() - t
    kms, kn = eng.kernel_time(reset=True)
    s1, i1 = eng.read('N_SOLVES').astype(np.int64).sum(), eng.read('N_ITERS').astype(np.int64).sum()
    print(json.dumps({'config': name, 'env_steps_per_s': batch * steps / el, 'ms_per_step': 1e3 * el / steps,
                      'batch': batch, 'steps': steps, 'solver': solver, 'step_kernel_ms': kms / max(kn, 1),
                      'solves_per_step': float(s1 - s0) / (batch * steps), 'iters_per_solve': float(i1 - i0) / max(float(s1 - s0), 1.0),
                      'lds_bytes_per_env': eng.lds_bytes, 'illegal_fraction': float((eng.read('ILLEGAL') != 0).mean()),
                      'engine_capacity_flags_last_step': int((eng.read('FLAG') == 4).sum()),
                      'done_fraction_last_step': float(eng.read('DONE').mean())}), flush=True)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    import bench
    case118, _, _ = bench.load_workload()
    lim118 = bench.bench_limits(case118)
    run('configs[1]: default14 AC Newton-Raphson, batch 1024, do-nothing', 'default14', 'newton', 1024, steps)
    run('default14 AC fast-decoupled XB (the reference solver), batch 1024, do-nothing', 'default14', 'fdxb', 1024, steps)
    run('configs[2] with the reference solver: default118 fast-decoupled XB, batch 4096, cascade limits', 'default118', 'fdxb',
        4096, steps, limits=lim118, max_active=case118.nS)
    run('configs[4] share of one GPU: default118 AC Newton-Raphson, random node splitting every step, batch 1024, all 236 '
        'busbars may be active', 'default118', 'newton', 1024, steps, limits=lim118, split=True)
    run('the same with max_active_buses = 128 (W = 2 kernels; a topology with more live busbars would report flag 4)',
        'default118', 'newton', 1024, steps, limits=lim118, split=True, max_active=128)


if __name__ == '__main__':
    main()
