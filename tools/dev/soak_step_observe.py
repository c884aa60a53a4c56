"""Developer aid (GPU box): ppn_step_observe against ppn_step + ppn_read_observation at scale (tests/engine_checks.check_step_observe)."""
import json, os, sys
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import engine_checks as ec
from helpers import ENVS, load_env
with open(os.path.join(ENVS, 'default118', 'bench_limits.json')) as f:
    lim = np.asarray(json.load(f)['limits_a'])
case, _, _ = load_env('default118')
n = ec.check_step_observe(None, 'default118', 4096, 40, 'newton', 'full', np.float64, thermal_limits=lim, max_active_buses=case.nS)
print('bench workload, two-word kernels: 4096 environments x 40 steps, full float64 rows identical; episodes ended on the way:', n, flush=True)
n = ec.check_step_observe(None, 'default118', 1024, 24, 'newton', 'minimalist', np.float32, thermal_limits=lim)
print('every busbar may be active (four-word kernels, two-capacity stepping): 1024 x 24, minimalist float32 rows identical; ended:', n, flush=True)
n = ec.check_step_observe(None, 'default118', 1024, 24, 'fdxb', 'ac_minimalist', np.float64, thermal_limits=lim, auto_reset=False)
print('fast-decoupled, four-word kernels, no auto reset: 1024 x 24, ac_minimalist rows identical; ended:', n, flush=True)
n = ec.check_step_observe(None, 'default14', 4096, 60, 'newton', 'full', np.float64)
print('default14 (one-word kernels): 4096 x 60 identical; ended:', n, flush=True)
