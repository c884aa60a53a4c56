#!/usr/bin/env python
"""What does ppn_recv hand out when an asynchronous session misbehaves?  Plays tools/async_rate.py's loop with full bookkeeping on the
host: every received id is checked against the set the host believes in flight; the first anomaly is printed with the session's
counters.   python tools/dev/async_dups.py [batch] [steps] [min_ready] [settle_at]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import bench  # noqa: E402


def main():
    import torch
    from pypownet_amd.engine import Engine
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    mr = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    settle_at = int(sys.argv[4]) if len(sys.argv) > 4 else -1
    flags = set(sys.argv[5:])
    host_rows = 'host' in flags
    read_rows = 'rows' in flags        # the receiver copies the rows of what it received to the host through torch's DEFAULT stream (tests do)
    idle_ms = [int(f[4:]) for f in flags if f.startswith('idle')]
    case, conf, chronics = bench.load_workload()
    eng = Engine(case, conf, B, device=0, chronics=chronics, thermal_limits=bench.bench_limits(case), max_active_buses=case.nS)
    slots, t0 = bench.env_assignment(0, B, chronics)
    eng.reset(chronic_slot=slots, t0=t0)
    rep_t = torch.zeros((B, 3), dtype=torch.float64, device='cuda')
    obs_t = torch.zeros((B, eng.observation_length('full')), dtype=torch.float64, device='cuda')
    acts = torch.zeros((B, case.action_length), dtype=torch.uint8, device='cuda')
    acts_h = np.zeros((B, case.action_length), dtype=np.uint8)
    torch.cuda.synchronize()
    eng.sync()
    eng.async_start(obs_t.data_ptr(), obs_t.numel() * 8, rep_t.data_ptr(), idle_timeout_ms=idle_ms[0] if idle_ms else 0)
    worst = 0.0
    inflight = np.zeros(B, dtype=bool)
    step_of = np.zeros(B, dtype=np.int64)

    def send(ids):
        assert not inflight[ids].any()
        inflight[ids] = True
        if host_rows:
            eng.send(ids, acts_h, rows_by_env=True)
        else:
            eng.send_device(ids, acts.data_ptr(), rows_by_env=True)
    send(np.arange(B, dtype=np.int32))
    total, n_recv, bad = 0, 0, 0
    t_begin = time.perf_counter()
    while total < B * K:
        t_a = time.perf_counter()
        ids = eng.recv(min_ready=mr).copy()
        n_recv += 1
        if read_rows:
            ix = torch.as_tensor(ids.astype(np.int64), device='cuda')
            o_ = obs_t[ix].cpu().numpy(); r_ = rep_t[ix].cpu().numpy()
        worst = max(worst, time.perf_counter() - t_a)
        u, c = np.unique(ids, return_counts=True)
        dup = u[c > 1]
        notin = np.array([e for e in u if not inflight[e]], dtype=np.int64)
        if len(ids) == 0 or len(dup) or len(notin):
            bad += 1
            print('receive %d (t = %.1f ms, %d steps so far): %d ids, %d duplicated %s, %d not in flight %s; stats %s; in flight on the host: %d'
                  % (n_recv, (time.perf_counter() - t_begin) * 1e3, total, len(ids), len(dup), dup[:8].tolist(), len(notin), notin[:8].tolist(),
                     eng.async_stats(), int(inflight.sum())), flush=True)
            pos = {int(e): np.nonzero(ids == e)[0].tolist() for e in dup[:4]}
            print('    positions of the duplicates inside the receive:', pos, flush=True)
            if bad >= 3:
                break
            ids = u[inflight[u]].astype(np.int32)
        inflight[ids] = False
        step_of[ids] += 1
        total += len(ids)
        if n_recv == settle_at:
            eng.read('N_STEPS')
        again = ids[step_of[ids] < K]
        if len(again):
            send(again.astype(np.int32))
    print('B=%d K=%d min_ready=%d settle_at=%d %s: %d receives, %d anomalies, %.3f M env-steps/s, longest receive (+ row copies) %.1f ms, stats %s' % (
        B, K, mr, settle_at, ' '.join(sorted(flags)) or 'device rows', n_recv, bad, total / (time.perf_counter() - t_begin) / 1e6, worst * 1e3, eng.async_stats()), flush=True)
    try:
        eng.async_stop()
    except Exception as ex:      # noqa: BLE001
        print('async_stop:', ex)
    eng.close()


if __name__ == '__main__':
    main()
