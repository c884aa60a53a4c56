#!/usr/bin/env python
"""Per-phase cycle breakdown of the step kernel (GPU box).  Builds pypownet_amd/csrc with -DPPN_PROF into
build/libppn_prof.so (never loaded by the product), runs the bench workload for a few steps and prints the share
of wave cycles spent in each phase of solve_loadflow / body_step."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
NAMES = ['connectivity (adjacency + BFS)', 'schedule check (+rebuild)', 'live buses/injections/types', 'Ybus', 'mismatch+Jacobian', 'LU factor', 'LU backward',
         'update/other', 'pfsoln+outputs', 'action+advance', 'cascade total (incl 0-8)', 'restart of ended episodes (incl its solves)',
         'cut flags + topology write-back', '(slot 13: body start time)']


def main():
    import bench
    from harness import engine_with_library      # (tests/harness.py: the profiling build is not the product library)
    lib = os.environ.get('PPN_PROF_LIB', os.path.join(ROOT, 'build', 'libppn_prof.so'))
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    if not os.path.exists(lib) or os.environ.get('PPN_REBUILD'):
        # (the parallel build of __graft_entry__: about a minute; build it in the development container -- build/ travels to the GPU box)
        import __graft_entry__ as ge
        built = ge.build_variant('prof', ['-DPPN_PROF'] + os.environ.get('PPN_PROF_FLAGS', '').split())
        if built != lib:
            os.replace(built, lib)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    split = len(sys.argv) > 3 and sys.argv[3] == 'split'     # random node-splitting actions, every busbar may be active
    envname = os.environ.get('PPN_PROF_ENV', bench.ENV_NAME)      # e.g. default14: the W = 1 kernels
    if envname == bench.ENV_NAME:
        case, conf, chronics = bench.load_workload()
        conf['solver'] = os.environ.get('PPN_PROF_SOLVER', 'newton')
        limits = bench.bench_limits(case)
    else:
        case, conf, chronics = bench.load_env_fixture(envname, os.environ.get('PPN_PROF_SOLVER', 'newton'))
        limits = None
    eng = engine_with_library(lib, case, conf, B, chronics=chronics, thermal_limits=limits,
                              max_active_buses=(int(sys.argv[4]) if len(sys.argv) > 4 else 2 * case.nS) if split else case.nS)
    slots, t0 = bench.env_assignment(0, B, chronics)
    eng.reset(chronic_slot=slots, t0=t0)
    act = np.zeros((B, case.action_length), dtype=np.uint8)
    if split:
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        from bench_configs import random_node_splitting
        rng = np.random.default_rng(1234)
        for _ in range(8):
            eng.step(random_node_splitting(case, rng, B), auto_reset=True)
        acts = [random_node_splitting(case, rng, B) for _ in range(8)]
    else:
        acts = [act]
    AR = int(os.environ.get('PPN_BENCH_AUTO_RESET', '2'))
    eng.step(acts[0], auto_reset=AR)
    zero = np.zeros((B, 32), dtype=np.int64)
    eng._check(eng._lib.ppn_write(eng._h, 100, zero.ctypes.data, zero.nbytes), 'write prof')
    eng.kernel_time(reset=True)
    s0, i0 = eng.read('N_SOLVES').sum(), eng.read('N_ITERS').sum()
    for k in range(steps):
        eng.step(acts[(k + 1) % len(acts)], auto_reset=AR)
    eng.sync()
    s1, i1 = eng.read('N_SOLVES').sum(), eng.read('N_ITERS').sum()
    out = np.zeros((B, 32), dtype=np.int64)
    eng._check(eng._lib.ppn_read(eng._h, 100, out.ctypes.data, out.nbytes, 1, 0), 'read prof')
    tot = out.sum(axis=0).astype(np.float64)
    nsolve, nit = float(s1 - s0), float(i1 - i0)
    print('B=%d steps=%d solves=%d iterations=%d' % (B, steps, nsolve, nit))
    for k, name in enumerate(NAMES):
        if k == 13:
            continue       # (slot 13: wall time at which the body began -- tests/tools/chain_lengths.py, order_sim.py)
        unit = 'iteration' if k in (4, 5, 6) else ('env-step' if k in (9, 10, 11, 12) else 'solve')
        per = tot[k] / {'iteration': nit, 'env-step': float(B * steps), 'solve': nsolve}[unit]
        print('%-28s total %.3e cyc  %8.0f cyc per %s' % (name, tot[k], per, unit))
    if conf['solver'] == 'fdxb':
        for k, name in enumerate(("B', B'' assembly", "factorisation of B', B'' (+ tail inverses, scaled L)", 'V = |V| e^{ja} (+ clears)',
                                  'mismatch over the Ybus entries', 'norm, right-hand side', 'forward substitution + tail', 'backward substitution + update')):
            per = tot[16 + k] / (nsolve if k < 2 else nit)
            print('fdxb: %-52s total %.3e cyc  %8.0f cyc per %s' % (name, tot[16 + k], per, 'solve' if k < 2 else 'half-iteration'))
        return
    for k, name in ((16, 'LU level bounds + record prefetch'), (17, 'LU phase 0 (invert, y\')'), (18, 'LU phase 1 (U\', forward)'), (19, 'LU phase 2 (Schur)')):
        print('%-28s total %.3e cyc  %8.0f cyc per iteration' % (name, tot[k], tot[k] / nit))
    print('%-28s total %.3e cyc  %8.0f cyc per iteration' % ('dense tail (inside LU factor)', tot[30], tot[30] / nit))
    for k, name in ((20, 'evaluation pass 0 (V, clears)'), (21, 'evaluation pass 1 (Ybus entries)'), (22, 'evaluation pass 2 (buses, norm)')):
        print('%-28s total %.3e cyc  %8.0f cyc per evaluation (iterations + solves)' % (name, tot[k], tot[k] / (nit + nsolve)))
    if split:
        for k, name in ((23, 'numbering, line ends'), (24, 'pivots per level'), (25, 'adjacency, Ybus row pointers'), (26, 'symbolic elimination'),
                        (27, 'entry numbering, pivot records, line / Ybus entry tables'), (28, 'fill-in list, pair / triple records'),
                        (29, 'level table, dense-tail map, header')):
            print('schedule_build: %-58s total %.3e cyc  %5.1f %% of the rebuilds  %8.0f cyc per env-step' % (name, tot[k], 100.0 * tot[k] / max(tot[23:30].sum(), 1.0), tot[k] / float(B * steps)))
    # whole kernel body per environment: shader cycles (clock64) and 100 MHz wall ticks (wall_clock64)
    kt = eng.kernel_time()
    body_c, body_w = tot[14], tot[15] * 1e-8
    print('step kernel body: %.0f cyc and %.1f us per env-step -> shader clock %.0f MHz under load'
          % (body_c / (B * steps), body_w / (B * steps) * 1e6, body_c / body_w * 1e-6))
    print('step kernel: %.3f ms per launch (HIP events, %d launches); mean resident environments = sum of body wall times / '
          'kernel time = %.0f (of %d slots = 256 CUs x floor(160 KiB / LDS per environment))' % (kt[0] / kt[1], kt[1], body_w * 1e3 / kt[0], 256 * (163840 // eng.lds_bytes)))


if __name__ == '__main__':
    main()
