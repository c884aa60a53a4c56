"""Developer aid (GPU box, -DPPN_PROF build): if the K environments the launch-order key ranks heaviest ran on CUs of their own, how
long would the longest chain among the REST of a launch be?  Per step: body wall time of every environment against its key."""
import os, sys, numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tools'); sys.path.insert(0, ROOT + '/tests')
import bench
from harness import engine_with_library
lib = os.path.join(ROOT, 'build', 'libppn_prof.so')
case, conf, chronics = bench.load_workload()
B = 4096
lim = bench.bench_limits(case)
eng = engine_with_library(lib, case, conf, B, chronics=chronics, thermal_limits=lim, max_active_buses=case.nS)
slots, t0 = bench.env_assignment(0, B, chronics)
eng.reset(chronic_slot=slots, t0=t0)
act = np.zeros((B, case.action_length), dtype=np.uint8)
for _ in range(6):
    eng.step(act, auto_reset=2)
Ks = (32, 64, 128, 256, 512, 1024)
rows = []
for rep in range(16):
    eng.sync()
    prio = eng.read('LAUNCH_PRIO') if hasattr(eng, 'FIELDS') and 'LAUNCH_PRIO' in getattr(eng, 'FIELDS', {}) else None
    amps, st = eng.read('AMPS'), eng.read('LINES_STATUS')
    load = np.nan_to_num(np.where(st != 0, amps / lim[None, :], 0.0), nan=10.0, posinf=10.0).max(axis=1)
    dead = eng.read('DONE')      # (episodes that ended owe a restart: key 1.3)
    key = np.where(dead != 0, 1.3, load)
    zero = np.zeros((B, 32), dtype=np.int64)
    eng._check(eng._lib.ppn_write(eng._h, 100, zero.ctypes.data, zero.nbytes), 'w')
    eng.step(act, auto_reset=2)
    o = np.zeros((B, 32), dtype=np.int64)
    eng._check(eng._lib.ppn_read(eng._h, 100, o.ctypes.data, o.nbytes, 1, 0), 'r')
    w = o[:, 15] * 1e-2   # us
    order = np.argsort(-key, kind='stable')
    rank = np.empty(B, dtype=np.int64); rank[order] = np.arange(B)
    slow = np.argsort(-w)[:12]
    rows.append([w.max()] + [w[order[K:]].max() for K in Ks])
    print('step %2d: longest %4.0f us; ranks of the 12 slowest: %s; longest outside the top K %s: %s'
          % (rep, w.max(), rank[slow].tolist(), Ks, [int(w[order[K:]].max()) for K in Ks]), flush=True)
r = np.array(rows)
print('mean over steps: longest %.0f us; longest outside the top K %s: %s' % (r[:, 0].mean(), Ks, np.round(r[:, 1:].mean(axis=0)).tolist()))
