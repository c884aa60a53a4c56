"""Launch-order study (-DPPN_PROF build, GPU box): how long would a step launch last under other launch-order keys?
Every environment's body time of step t + 1 is recorded next to what was known at the end of step t (features the profiling
build leaves in its counters: largest line loading, the loading of in-service lines that the NEXT chronic row takes out, ...).
A greedy list schedule over the resident slots (256 CUs x 7) replays the launch under each key: environments are handed out
in key order, a slot takes the next one when it is free.  'oracle' sorts by the true body time (longest first).
The feature slots (23-28) are only written by a library built with -DPPN_PROF -DPPN_PROF_ORDER (a plain -DPPN_PROF build keeps
schedule_build's sub-phase counters there): PPN_PROF_FLAGS=-DPPN_PROF_ORDER PPN_REBUILD=1 python tools/profile_phases.py first.
Usage: python tests/tools/order_sim.py [steps]"""
import heapq, os, sys, numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/tools'); sys.path.insert(0, ROOT + '/tests')
import bench
from harness import engine_with_library
lib = os.environ.get('PPN_PROF_LIB', os.path.join(ROOT, 'build', 'libppn_prof.so'))
case, conf, chronics = bench.load_workload()
B, SLOTS = 4096, 1792
eng = engine_with_library(lib, case, conf, B, chronics=chronics, thermal_limits=bench.bench_limits(case), max_active_buses=case.nS)
slots, t0 = bench.env_assignment(0, B, chronics)
eng.reset(chronic_slot=slots, t0=t0)
act = np.zeros((B, case.action_length), dtype=np.uint8)
for _ in range(6):
    eng.step(act, auto_reset=2)


def prof():
    o = np.zeros((B, 32), dtype=np.int64)
    eng._check(eng._lib.ppn_read(eng._h, 100, o.ctypes.data, o.nbytes, 1, 0), 'r')
    return o


def makespan(order, w, starts=None):
    free = [0.0] * SLOTS
    heapq.heapify(free)
    end = 0.0
    for e in order:
        t0_ = heapq.heappop(free)
        if starts is not None:
            starts.append(t0_)
        t = t0_ + w[e]
        end = max(end, t)
        heapq.heappush(free, t)
    return end


steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
res = {}
feat = None
for rep in range(steps + 1):
    zero = np.zeros((B, 32), dtype=np.int64)
    eng._check(eng._lib.ppn_write(eng._h, 100, zero.ctypes.data, zero.nbytes), 'w')
    eng.kernel_time(reset=True)
    eng.step(act, auto_reset=2)
    kt = eng.kernel_time()
    o = prof()
    w = o[:, 15] * 1e-2      # body time of THIS step, us
    if feat is not None:
        ld, fmax, fsum, fcnt, fsec, dead = feat
        owes = dead == 2
        keys = {
            'current (loading; 1.3 if a restart is owed)': np.where(owes, 1.3, ld),
            'oracle (true body time)': w,
            'random': np.random.default_rng(rep).random(B),
            'loading + 0.5 x loading of the most loaded line going out': np.where(owes, 1.3, ld + 0.5 * fmax),
            'loading + 1.0 x ...': np.where(owes, 1.3, ld + 1.0 * fmax),
            'max(loading, 0.6 + loading of the most loaded line going out)': np.where(owes, 1.3, np.maximum(ld, 0.6 + fmax)),
            'loading + 0.3 x sum of loadings going out': np.where(owes, 1.3, ld + 0.3 * fsum),
            'loading + excess over 0.9 summed': np.where(owes, 1.3, ld + fsec),
            'restart owed at 1.0': np.where(owes, 1.0, ld),
            'restart owed at 2.0': np.where(owes, 2.0, ld),
        }
        for name, k in keys.items():
            order = np.argsort(-np.nan_to_num(k, nan=9.0), kind='stable')
            res.setdefault(name, []).append(makespan(order, w))
        # the same launch as the hardware hands it out: workgroup i goes to XCD i mod 8, every XCD schedules its own 32 CUs x 7 slots
        order = np.argsort(-np.nan_to_num(keys['current (loading; 1.3 if a restart is owed)'], nan=9.0), kind='stable')
        SL = SLOTS
        globals()['SLOTS'] = SL // 8
        res.setdefault('current key, 8 XCD queues of 224 slots (workgroup i -> XCD i mod 8)', []).append(max(makespan(order[x::8], w) for x in range(8)))
        res.setdefault('current key, 8 XCD queues, 6 slots per CU', []).append(0.0)
        globals()['SLOTS'] = 32 * 6
        res['current key, 8 XCD queues, 6 slots per CU'][-1] = max(makespan(order[x::8], w) for x in range(8))
        globals()['SLOTS'] = SL
        sim_starts = []
        makespan(order, w, sim_starts)
        ssim = np.sort(np.asarray(sim_starts))
        print('   step %d: the greedy replay would start #2500 at %.0f us, #3500 at %.0f, the last at %.0f' % (rep, ssim[2499], ssim[3499], ssim[-1]))
        st = (o[:, 13] - o[:, 13].min()) * 1e-2
        res.setdefault('measured: last body end - first body begin', []).append(float((st + w).max()))
        res.setdefault('measured: longest body', []).append(float(w.max()))
        res.setdefault('measured: start of the 1792nd / 2500th / 3500th / last environment (us)', []).append(0.0)
        ss = np.sort(st)
        print('   step %d: starts sorted: #1792 %.0f us, #2500 %.0f, #3500 %.0f, last %.0f; bodies started in the first 5 us: %d' % (rep, ss[1791], ss[2499], ss[3499], ss[-1], int((st < 5).sum())))
        res.setdefault('measured kernel time', []).append(kt[0] / kt[1] * 1e3)
    as_d = lambda col: o[:, col].copy().view(np.float64)
    feat = (as_d(23), as_d(24), as_d(25), o[:, 26].astype(float), as_d(27), o[:, 28].copy())
for name, v in res.items():
    print('%-66s simulated launch %.0f us (min %.0f max %.0f)' % (name, np.mean(v), np.min(v), np.max(v)))
