"""Distribution of per-environment step-kernel times on the bench workload (-DPPN_PROF build, GPU box): shows that a launch
lasts as long as its longest environment.  Usage: python tests/tools/chain_lengths.py"""
import os, sys, numpy as np
ROOT=os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT+'/tools'); sys.path.insert(0, ROOT+'/tests')
import bench
from harness import engine_with_library
lib=os.environ.get('PPN_PROF_LIB', os.path.join(ROOT,'build','libppn_prof.so'))
ENV=os.environ.get('PPN_PROF_ENV', bench.ENV_NAME)      # e.g. default14 (the one-word kernels; PPN_CHAIN_BATCH=1024)
if ENV == bench.ENV_NAME:
    case, conf, chronics = bench.load_workload(); LIM = bench.bench_limits(case)
else:
    case, conf, chronics = bench.load_env_fixture(ENV, 'newton'); LIM = None
B=int(os.environ.get('PPN_CHAIN_BATCH', '4096'))
eng=engine_with_library(lib, case, conf, B, chronics=chronics, thermal_limits=LIM, max_active_buses=case.nS)
slots,t0=bench.env_assignment(0,B,chronics)
eng.reset(chronic_slot=slots,t0=t0)
act=np.zeros((B,case.action_length),dtype=np.uint8)
AR = int(os.environ.get('PPN_BENCH_AUTO_RESET', '2'))
for _ in range(6): eng.step(act, auto_reset=AR)
for rep in range(4):
    zero=np.zeros((B,32),dtype=np.int64)
    eng._check(eng._lib.ppn_write(eng._h,100,zero.ctypes.data,zero.nbytes),'w')     # (field 100 neither settles owed restarts nor waits for them)
    st_before=eng.read('LINES_STATUS').copy()      # (analysis only: the read settles the owed restarts)
    ns0=eng.read('N_SOLVES').astype(np.int64); ni0=eng.read('N_ITERS').astype(np.int64)
    eng.kernel_time(reset=True)
    eng.step(act, auto_reset=AR)
    kt=eng.kernel_time()
    out=np.zeros((B,32),dtype=np.int64)
    eng._check(eng._lib.ppn_read(eng._h,100,out.ctypes.data,out.nbytes,1,0),'r')
    flag=eng.read('FLAG'); depth=eng.read('CASCADE_DEPTH')
    ns1=eng.read('N_SOLVES').astype(np.int64); ni1=eng.read('N_ITERS').astype(np.int64)
    w=out[:,15]*1e-8*1e6  # us
    print('auto_reset %d: kernel %.0f us | env body wall us: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f | sum/kernel = %.0f resident' % (AR, kt[0]/kt[1]*1e3, w.mean(), np.percentile(w,50), np.percentile(w,90), np.percentile(w,99), w.max(), w.sum()/(kt[0]/kt[1]*1e3)))
    # when did every environment START inside the launch, and which ones END it?  (prof[13]: wall ticks at body begin)
    start=(out[:,13]-out[:,13].min())*1e-8*1e6; end=start+w
    last=np.argsort(-end)[:8]
    print('   launch ends at %.0f us after the first body began; latest-ending environments (start + body = end us, rank of the launch-order key):' % end.max())
    prio_rank=None
    for e in last:
        print('      env %4d: start %5.0f + body %4.0f = end %4.0f   (cascade depth %d, flag %d)' % (e, start[e], w[e], end[e], depth[e], flag[e]))
    ev=eng.read('LINE_EVENTS'); outage=(((ev & 6)!=0)&(st_before!=0)).any(axis=1)      # a line that was ON gets a maintenance (2) / hazard (4) outage in this step
    med=(start>150)&(w>140)
    print('   late starters (> 150 us) with a body > 140 us: %d, of which with a maintenance / hazard event in this step: %d; all environments with such an event: %d, their mean body %.0f us (others %.0f us)'
          % (int(med.sum()), int((med&outage).sum()), int(outage.sum()), w[outage].mean() if outage.any() else 0, w[~outage].mean()))
    late=(start>50)&(w>0.7*w.max())
    print('   environments with a body > 70 %% of the longest that started later than 50 us: %d; longest body %.0f us, latest end %.0f us' % (int(late.sum()), w.max(), end.max()))
    top=np.argsort(-w)[:6]
    for e in top:
        print('   env %4d body %.0f us: prologue %.0f us, cascade %.0f us, restart (fused or owed) %.0f us (flag %d, cascade depth %d, solves this step %d, iterations %d)' % (e, w[e], out[e,9]/2370., out[e,10]/2370., out[e,11]/2370., flag[e], depth[e], ns1[e]-ns0[e], ni1[e]-ni0[e]))
