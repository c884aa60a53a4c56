#!/usr/bin/env python
"""Where does an asynchronous session lose time?  Per-iteration host timestamps of the recv / send loop (tools/async_rate.py's loop),
the largest gaps, and the server-restart counter as it moves.   python tools/dev/async_debug.py [batch] [steps] [variant ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench  # noqa: E402


def hwid_report(out, n_wg):
    """-DPPN_PROF_HWID builds: slot 31 of profile row w = HW_ID | XCC_ID << 32 of workgroup w -- how many workgroups per CU / per SIMD?"""
    import collections
    v = out[:n_wg, 31]
    if not v.any():
        return
    hw = v & 0xFFFFFFFF
    xcc = (v >> 32) & 0xF
    simd = (hw >> 4) & 0x3; cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 0x1; se = (hw >> 13) & 0x7
    per_cu = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist()))
    per_simd = collections.Counter(zip(xcc.tolist(), se.tolist(), sh.tolist(), cu.tolist(), simd.tolist()))
    print('      placement of %d workgroups: %d CUs in use, workgroups per CU %s; per SIMD %s' % (
        n_wg, len(per_cu), dict(sorted(collections.Counter(per_cu.values()).items())), dict(sorted(collections.Counter(per_simd.values()).items()))), flush=True)


def run(variant, B, K):
    import torch
    from pypownet_amd.engine import Engine
    case, conf, chronics = bench.load_workload()
    lib = os.environ.get('PPN_DEBUG_LIB')      # (developer builds of the library, driven through the test harness)
    if lib:
        from harness import engine_with_library
        eng = engine_with_library(lib, case, conf, B, device=0, chronics=chronics, thermal_limits=bench.bench_limits(case), max_active_buses=case.nS)
    else:
        eng = Engine(case, conf, B, device=0, chronics=chronics, thermal_limits=bench.bench_limits(case), max_active_buses=case.nS)
    slots, t0 = bench.env_assignment(0, B, chronics)
    eng.reset(chronic_slot=slots, t0=t0)
    n_obs = eng.observation_length('full')
    obs_t = torch.zeros((B, n_obs), dtype=torch.float64, device='cuda')
    rep_t = torch.zeros((B, 3), dtype=torch.float64, device='cuda')
    ids_d = torch.zeros((B,), dtype=torch.int32, device='cuda')
    acts = torch.zeros((B, case.action_length), dtype=torch.uint8, device='cuda')
    acts_h = np.zeros((B, case.action_length), dtype=np.uint8)
    torch.cuda.synchronize()
    eng.sync()
    if os.environ.get('PPN_ASYNC_ANATOMY'):
        zero = np.zeros((B, 32), dtype=np.int64)
        eng._check(eng._lib.ppn_write(eng._h, 100, zero.ctypes.data, zero.nbytes), 'write prof')
        anat0 = (eng.read('N_SOLVES').sum(), eng.read('N_ITERS').sum())
    if 'memo' in variant:      # restart memo on, warmed by 40 fused steps (the engine's learning steps)
        eng.restart_memo(True)
        for _ in range(40):
            eng.step(acts_h, auto_reset=True)
        eng.sync()
        if os.environ.get('PPN_ASYNC_ANATOMY'):
            zero = np.zeros((B, 32), dtype=np.int64)
            eng._check(eng._lib.ppn_write(eng._h, 100, zero.ctypes.data, zero.nbytes), 'write prof')
            anat0 = (eng.read('N_SOLVES').sum(), eng.read('N_ITERS').sum())
    if 'rollout' in variant:      # reference point: the device-policy rollout kernel (no host in the loop, no observation)
        if 'hoststore' in variant:      # (-DPPN_ROLLOUT_HOST_STORE builds: a session started and stopped leaves its pinned rings to the rollout kernel)
            eng.async_start(0, 0, rep_t.data_ptr()); eng.async_stop()
        eng.rollout_policy('do_nothing', [], 3); eng.sync()
        load = None
        if 'load' in variant:      # `rollout+load`: while the rollout kernel runs, a second engine launches what a session launches per receive
            load = Engine(case, conf, 1024, device=0, chronics=chronics, thermal_limits=bench.bench_limits(case), max_active_buses=case.nS)
            load.reset()
            lobs = torch.zeros((1024, n_obs), dtype=torch.float64, device='cuda')
            lsrc = torch.zeros((1024, case.action_length), dtype=torch.uint8, device='cuda'); ldst = torch.zeros_like(lsrc)
            side = torch.cuda.Stream()
            load.observations_into_device(lobs.data_ptr(), lobs.numel() * 8); load.wait() if hasattr(load, 'wait') else load.sync()
            torch.cuda.synchronize()
        if 'single' in variant:      # `rollout+single`: K launches of ONE step each, waited for one by one (against the server with min_ready = batch)
            t_ = time.perf_counter()
            for _ in range(K):
                eng.rollout_policy('do_nothing', [], 1); eng.sync()
            print('%-16s B=%d K=%d: %.3f M env-steps/s (K launches of ppn_rollout_policy(1 step), each waited for)' % (variant, B, K, B * K / (time.perf_counter() - t_) / 1e6), flush=True)
            eng.close()
            return
        t_ = time.perf_counter(); eng.rollout_policy('do_nothing', [], K)
        n_load = 0
        if load is not None:
            t_end = time.perf_counter() + B * K / 16e6      # (about as long as the rollout lasts)
            while time.perf_counter() < t_end:
                load.observations_into_device(lobs.data_ptr(), lobs.numel() * 8)      # K_OBS over 1024 environments on the other engine's stream
                with torch.cuda.stream(side):
                    ldst.copy_(lsrc)                                                   # an enqueue-sized copy
                n_load += 1
                time.sleep(100e-6)
        ticks = [int(v[5:]) for v in variant.split('+') if v.startswith('ticks')]
        n_ticks = 0
        if ticks:      # `rollout+ticksN`: one kernel that moves nothing every N microseconds on another stream while the rollout kernel runs
            side = torch.cuda.Stream()
            tick = torch.zeros((1,), dtype=torch.float32, device='cuda')
            es = torch.cuda.ExternalStream(eng.stream_ptr(), device='cuda')
            nxt = time.perf_counter()
            with torch.cuda.stream(side):
                while not es.query():
                    tick.add_(1); n_ticks += 1
                    nxt += ticks[0] * 1e-6
                    while time.perf_counter() < nxt:
                        pass
        t_launch = time.perf_counter() - t_
        eng.sync()
        print('%-16s B=%d K=%d: %.3f M env-steps/s (ppn_rollout_policy, do-nothing)%s%s' % (variant, B, K, B * K / (time.perf_counter() - t_) / 1e6,
              (' with %d gather + copy launches beside it' % n_load) if load is not None else '',
              (' with %d empty launches beside it (%.0f per ms)' % (n_ticks, n_ticks / t_launch / 1e3)) if ticks else ''), flush=True)
        if os.environ.get('PPN_ASYNC_ANATOMY'):
            sys.path.insert(0, os.path.join(ROOT, 'tools'))
            from profile_phases import NAMES
            out = np.zeros((B, 32), dtype=np.int64)
            eng._check(eng._lib.ppn_read(eng._h, 100, out.ctypes.data, out.nbytes, 1, 0), 'read prof')
            tot = out.sum(axis=0).astype(np.float64)
            n = float(B * (K + 3))
            nsolve = float(eng.read('N_SOLVES').sum() - anat0[0]); nit = float(eng.read('N_ITERS').sum() - anat0[1])
            print('    solves %d iterations %d (%.2f / %.2f per step); shader cycles:' % (nsolve, nit, nsolve / n, nit / n))
            for k, name in enumerate(NAMES):
                if k == 13:
                    continue
                unit = 'iteration' if k in (4, 5, 6) else ('env-step' if k in (9, 10, 11, 12) else 'solve')
                print('      %-44s %8.0f cyc per %s' % (name, tot[k] / {'iteration': nit, 'env-step': n, 'solve': nsolve}[unit], unit))
            for k_, name_ in ((16, 'LU level bounds + record prefetch'), (18, "LU phase 1 (U', forward)"), (19, 'LU phase 2 (Schur)'), (30, 'dense tail (inside LU factor)')):
                print('      %-44s %8.0f cyc per iteration' % (name_, tot[k_] / nit))
            for k_, name_ in ((20, 'evaluation pass 0 (V, clears)'), (21, 'evaluation pass 1 (Ybus entries)'), (22, 'evaluation pass 2 (buses, norm)')):
                print('      %-44s %8.0f cyc per evaluation' % (name_, tot[k_] / (nit + nsolve)))
            print('      %-44s %8.0f cyc and %.1f us per env-step' % ('step body', tot[14] / n, tot[15] / n * 1e-2))
            hwid_report(out, int(os.environ.get('PPN_ROLLOUT_WORKGROUPS', '1792')))
            if 'memo' in variant:
                print('      restart memo:', eng.restart_memo_stats())
        eng.close()
        return
    wg = [int(v[2:]) for v in variant.split('+') if v.startswith('wg')]
    mr = [int(v[2:]) for v in variant.split('+') if v.startswith('mr')]
    mr = mr[0] if mr else 256
    noise = [int(v[5:]) for v in variant.split('+') if v.startswith('noise')]
    noise = noise[0] if noise else 0
    dummy = torch.zeros((1,), dtype=torch.float32, device='cuda')
    dummy.add_(1); torch.cuda.synchronize()
    if 'noobs' in variant:
        eng.async_start(0, 0, rep_t.data_ptr(), workgroups=wg[0] if wg else 0)
    else:
        eng.async_start(obs_t.data_ptr(), obs_t.numel() * 8, rep_t.data_ptr(), workgroups=wg[0] if wg else 0)
    st = torch.cuda.ExternalStream(eng.async_stream_ptr(), device='cuda')
    stamps = []
    t_begin = time.perf_counter()
    with torch.cuda.stream(st):
        if 'hostact' in variant:
            eng.send(np.arange(B, dtype=np.int32), acts_h, rows_by_env=True)
        else:
            eng.send_device(np.arange(B, dtype=np.int32), acts.data_ptr(), rows_by_env=True)
        total = 0
        while total < B * K:
            if 'hostsleep' in variant:      # (the host does NOT poll while the server works: is it the polling?)
                time.sleep(0.03)
            ta = time.perf_counter()
            ids = eng.recv(min_ready=mr, ids_device_ptr=ids_d.data_ptr() if 'idsdev' in variant else 0)
            tb = time.perf_counter()
            n = len(ids)
            for _ in range(noise):      # `noiseN`: N kernels that move nothing, on the session's stream -- what do kernel BOUNDARIES cost the resident server?
                dummy.add_(1)
            if 'hostact' in variant:
                eng.send(ids, acts_h, rows_by_env=True)
            else:
                eng.send_device(ids, acts.data_ptr(), rows_by_env=True)
            tc = time.perf_counter()
            total += n
            stamps.append((ta - t_begin, tb - ta, tc - tb, n, eng.async_stats()['server_restarts']))
        td = time.perf_counter()
        while eng.async_stats()['in_flight']:
            eng.recv(min_ready=eng.async_stats()['in_flight'])
        te = time.perf_counter()
    el = time.perf_counter() - t_begin
    s = eng.async_stats()
    eng.async_stop()
    if os.environ.get('PPN_ASYNC_ANATOMY'):      # (a -DPPN_PROF library: K_SERVE books 100 MHz ticks per served step into the profile rows)
        import ctypes as C
        out = np.zeros((B, 32), dtype=np.int64)
        eng._check(eng._lib.ppn_read(eng._h, 100, out.ctypes.data, out.nbytes, 1, 0), 'read prof')
        n = max(float(out[:, 27].sum()), 1.0)
        print('    served steps %d: per step  waiting for an item %.1f us | acquire + body_step %.1f us | observation + report %.1f us | release + record %.1f us'
              % (n, out[:, 23].sum() / n * 1e-2, out[:, 24].sum() / n * 1e-2, out[:, 25].sum() / n * 1e-2, out[:, 26].sum() / n * 1e-2), flush=True)
        sys.path.insert(0, os.path.join(ROOT, 'tools'))
        from profile_phases import NAMES
        tot = out.sum(axis=0).astype(np.float64)
        nsolve = float(eng.read('N_SOLVES').sum() - anat0[0]); nit = float(eng.read('N_ITERS').sum() - anat0[1])
        print('    solves %d iterations %d (%.2f / %.2f per step); shader cycles:' % (nsolve, nit, nsolve / n, nit / n))
        for k, name in enumerate(NAMES):
            if k == 13:
                continue
            unit = 'iteration' if k in (4, 5, 6) else ('env-step' if k in (9, 10, 11, 12) else 'solve')
            print('      %-44s %8.0f cyc per %s' % (name, tot[k] / {'iteration': nit, 'env-step': n, 'solve': nsolve}[unit], unit))
        for k_, name_ in ((16, 'LU level bounds + record prefetch'), (18, "LU phase 1 (U', forward)"), (19, 'LU phase 2 (Schur)'), (30, 'dense tail (inside LU factor)')):
            print('      %-44s %8.0f cyc per iteration' % (name_, tot[k_] / nit))
        for k_, name_ in ((20, 'evaluation pass 0 (V, clears)'), (21, 'evaluation pass 1 (Ybus entries)'), (22, 'evaluation pass 2 (buses, norm)')):
            print('      %-44s %8.0f cyc per evaluation' % (name_, tot[k_] / (nit + nsolve)))
        print('      %-44s %8.0f cyc and %.1f us per env-step' % ('step body', tot[14] / n, tot[15] / n * 1e-2))
        hwid_report(out, s['workgroups'])
        busy = (out[:, 24].sum() + out[:, 25].sum() + out[:, 26].sum()) * 1e-8
        print('    workgroup-seconds busy %.4f over %.4f s of session -> %.0f workgroups busy on average' % (busy, el, busy / el), flush=True)
    tf = time.perf_counter()
    a = np.array([(x[1], x[2], x[3]) for x in stamps])
    worst = sorted(stamps, key=lambda x: -x[1])[:4]
    print('%-22s B=%d K=%d: %.3f M env-steps/s overall; %d receives, recv wait mean %.1f us max %.1f ms, send mean %.1f us max %.1f ms, drain %.2f ms, stop %.2f ms, restarts %d republished %d, resident %d' % (
        variant, B, K, (total + B) / el / 1e6, len(stamps), a[:, 0].mean() * 1e6, a[:, 0].max() * 1e3, a[:, 1].mean() * 1e6, a[:, 1].max() * 1e3,
        (te - td) * 1e3, (tf - te) * 1e3, s['server_restarts'], s['republished'], s['workgroups']), flush=True)
    for w in worst:
        print('    at %.1f ms: recv waited %.2f ms for %d environments (restarts so far %d)' % (w[0] * 1e3, w[1] * 1e3, w[3], w[4]), flush=True)
    eng.close()


if __name__ == '__main__':
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    for v in (sys.argv[3:] or ['devact', 'devact+idsdev', 'hostact']):
        run(v, B, K)
