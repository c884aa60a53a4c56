"""Developer aid (GPU box): step kernel time with fused restart, alone / followed by the observation gather / as ppn_step_observe."""
import os, sys, time
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT)
import torch
import bench
from pypownet_amd.engine import Engine
case, conf, chronics = bench.load_workload()
B = 4096
eng = Engine(case, conf, B, chronics=chronics, thermal_limits=bench.bench_limits(case), max_active_buses=case.nS)
slots, t0 = bench.env_assignment(0, B, chronics)
eng.reset(chronic_slot=slots, t0=t0)
act = torch.zeros((B, case.action_length), dtype=torch.uint8, device='cuda')
n = eng.observation_length('full')
obs = torch.empty((B, n), dtype=torch.float64, device='cuda')
nb = obs.numel() * 8
torch.cuda.synchronize()
def run(mode, steps=60):
    for _ in range(5):
        eng.step_device(act.data_ptr(), auto_reset=1)
    eng.sync(); eng.kernel_time(reset=True)
    t = time.perf_counter()
    for _ in range(steps):
        if mode == 'step':
            eng.step_device(act.data_ptr(), auto_reset=1)
        elif mode == 'two':
            eng.step_device(act.data_ptr(), auto_reset=1); eng.observations_into_device(obs.data_ptr(), nb)
        else:
            eng.step_observe_device(act.data_ptr(), obs.data_ptr(), nb, auto_reset=True)
    eng.sync()
    el = time.perf_counter() - t
    k = eng.kernel_time(reset=True)
    print('%-6s ms/step %.4f   timed kernel ms %.4f (%d launches)' % (mode, 1e3 * el / steps, k[0] / max(k[1], 1), k[1]), flush=True)
for m in ('step', 'two', 'fused', 'step', 'two', 'fused'):
    run(m)
