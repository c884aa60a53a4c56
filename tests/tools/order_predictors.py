"""Which quantity known BEFORE a step predicts the long environments of its launch?  (-DPPN_PROF build, GPU box.)
For a number of steps of the bench workload: per-environment body time of the step against (a) the largest line loading
left by the previous step (the key ppn_order_kernel sorts by), (b) the number of lines out of service, (c) whether the
chronic row about to be loaded carries a maintenance or a hazard.  Reported: how much of the total time of the slowest
10 % of the environments sits in the first quarter of the launch order under each key."""
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
    eng.step(act, auto_reset=True)
keys = {'soft_imminent': [], 'loading+soft': [], 'loading_one_step_old': [], 'low_vm': [], 'prev_iters': [], 'loading+vm': [], 'loading': [], 'lines_out': [], 'event_next': [], 'loading+event': [], 'loading+out': []}
times = []
load_prev = np.zeros(B)
it_prev = eng.read('N_ITERS').astype(np.int64)
for rep in range(12):
    amps, st = eng.read('AMPS'), eng.read('LINES_STATUS')
    load = np.nan_to_num(amps / lim[None, :], nan=10.0, posinf=10.0).max(axis=1)
    out_ = (st == 0).sum(axis=1)
    slot, row = eng.read('CHRONIC_SLOT'), eng.read('CHRONIC_ROW')
    ev = np.zeros(B)
    for e in range(B):
        c = chronics[int(slot[e])]
        r = min(int(row[e]) + 1, c.n_timesteps - 1)
        ev[e] = float((c.maintenance[r] > 0).any() or (c.hazards[r] > 0).any())
    keys['loading_one_step_old'].append(load_prev.copy()); load_prev = load
    soft = eng.read('SOFT_COUNT').astype(float)
    ll = np.nan_to_num(amps / lim[None, :], nan=10.0, posinf=10.0)
    imm = np.where(ll > 0.97, soft, 0.0).max(axis=1)
    keys['soft_imminent'].append(imm); keys['loading+soft'].append(load + 0.5 * imm)
    vm, bt = eng.read('VM'), eng.read('BUS_TYPE')
    lowv = -np.where(bt != 4, vm, 9.0).min(axis=1)
    it_now = eng.read('N_ITERS').astype(np.int64)
    keys['low_vm'].append(lowv); keys['prev_iters'].append((it_now - it_prev).astype(float)); keys['loading+vm'].append(load + 5.0 * (lowv + 1.0))
    it_prev = it_now
    zero = np.zeros((B, 32), dtype=np.int64)
    eng._check(eng._lib.ppn_write(eng._h, 100, zero.ctypes.data, zero.nbytes), 'w')
    eng.step(act, auto_reset=True); eng.sync()
    o = np.zeros((B, 32), dtype=np.int64)
    eng._check(eng._lib.ppn_read(eng._h, 100, o.ctypes.data, o.nbytes, 1, 0), 'r')
    w = o[:, 15] * 1e-2   # us
    times.append(w)
    keys['loading'].append(load); keys['lines_out'].append(out_.astype(float)); keys['event_next'].append(ev)
    keys['loading+event'].append(load + 2.0 * ev); keys['loading+out'].append(load + 0.05 * out_)
for name, ks in keys.items():
    cap, rho = [], []
    for k, w in zip(ks, times):
        slow = np.argsort(-w)[:B // 10]
        first = set(np.argsort(-k, kind='stable')[:B // 4].tolist())
        cap.append(sum(w[e] for e in slow if e in first) / w[slow].sum())
        rk, rw = np.argsort(np.argsort(k)), np.argsort(np.argsort(w))
        rho.append(np.corrcoef(rk, rw)[0, 1])
    print('%-14s share of the slowest decile\'s time started in the first quarter: %.2f   rank correlation with body time %.2f' % (name, np.mean(cap), np.mean(rho)))
w = np.concatenate(times)
print('body time us: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f' % (w.mean(), np.percentile(w, 50), np.percentile(w, 90), np.percentile(w, 99), w.max()))
