#!/usr/bin/env python
"""Headline benchmark: env-steps/s of the batched IEEE-118 AC load-flow step engine (BASELINE.json metric).

One "step" = ppn_step(auto_reset=1) over the whole batch: every environment applies its (do-nothing) action,
loads the next chronic row, runs the Newton-Raphson solve + the cascading-failure re-solve loop on the GPU,
updates the game counters, and environments that ended the step in game over are passed through
process_game_over -- all device resident, inputs (chronic tensors, state, actions) already in HBM.

Workload (BASELINE.json configs[2], SURVEY.md 8d "config 3"): default118, AC Newton (tol 1e-6, <=10 its),
4096 environments per GPU, do-nothing agent, thermal limits synthesised so that the cascade is exercised:
limit_k = max(50, round(Q0.98_t I_k(t))) A (tests/tools/make_bench_limits.py), hard coefficient 2.0, soft break after 3
consecutive overflowed steps (default118 YAML).
Environment e plays chronic (e mod n_chronics) from row t0 = (37 e) mod T.

Usage: python bench.py [--gpus N --steps K --warmup W]; for N > 1 launch with torch.distributed.run.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

ENV_NAME = 'default118'
BATCH_PER_GPU = 4096
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E peak (guides/MI355X_MICROARCH.md)
FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X FP64 vector peak (guides/MI355X_MICROARCH.md)
FLOP_PER_NR_ITERATION = 55.0e3   # useful flops of one Newton iteration @118 (LU 18 k, Jacobian 22 k, SpMV 4 k, solves 8 k)
B_IO = 21.6e3                    # compulsory per-step state bytes @118
B_IT_NR = 72.0e3                 # streaming-model bytes per Newton iteration @118


def load_workload():
    import yaml
    from pypownet_amd.case import Case
    from pypownet_amd.chronic import Chronic
    d = os.path.join(ROOT, 'tests', 'golden', 'envs', ENV_NAME, 'level0')
    case = Case.from_file(os.path.join(d, 'reference_grid.json'))
    with open(os.path.join(d, 'configuration.yaml')) as f:
        conf = yaml.safe_load(f)
    conf['solver'] = 'newton'
    cdir = os.path.join(d, 'chronics')
    chronics = [Chronic(os.path.join(cdir, c)) for c in sorted(os.listdir(cdir))]
    # (PPN_BENCH_CHRONICS=2: the two-chronic mix round 2 was measured on -- comparisons across rounds only)
    n_keep = int(os.environ.get('PPN_BENCH_CHRONICS', '0'))
    if n_keep > 0:
        chronics = chronics[:n_keep]
    return case, conf, chronics


def bench_limits(case):
    """Frozen synthetic limits of the workload (rule and generator: tests/tools/make_bench_limits.py)."""
    with open(os.path.join(ROOT, 'tests', 'golden', 'envs', ENV_NAME, 'bench_limits.json')) as f:
        lim = np.asarray(json.load(f)['limits_a'], dtype=np.float64)
    assert lim.shape == (case.nl,)
    return lim


def pmc_summary(batch):
    """The rocprofv3 PMC passes of this build (tools/collect_profiles.sh; summary committed under profiles/, newest round
    first): HBM bytes per step-kernel launch (FETCH_SIZE x 2 + WRITE_SIZE) and the SQ busy shares.  None when no summary
    matches this batch size -- counters cannot be collected from inside the timed run."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_summary.json')), reverse=True):
        try:
            with open(path) as f:
                p = json.load(f)
            if int(p['batch']) == int(batch):
                p['file'] = os.path.relpath(path, ROOT)
                return p
        except Exception:
            continue
    return None


def measured_traffic(batch):
    """(bytes per launch or None, where it comes from / why it is missing)."""
    p = pmc_summary(batch)
    if not p:
        return None, None
    want = p.get('library_sha256')
    try:
        have = library_sha256()
    except OSError:
        have = None
    if want is None or want != have:
        return None, 'dropped: %s was collected on another build (summary %s, this library %s)' % (
            p['file'], (want or 'unstamped')[:12], (have or '?')[:12])
    return float(p['hbm_bytes_per_launch']), p['file']


def env_assignment(first, count, chronics):
    ids = np.arange(first, first + count)
    slots = (ids % len(chronics)).astype(np.int32)
    T = np.array([c.n_timesteps for c in chronics])[slots]
    t0 = ((ids * 37) % T).astype(np.int32)
    return slots, t0


def cpu_baseline(case, conf, chronics, limits, budget_s=12.0):
    """The reference's CPU path, timed beside the GPU figure on this box's host cores.

    `value` = the REFERENCE-EQUIVALENT PYTHON BACKEND: oracle/game_np.py -- the numpy / scipy restatement of pypownet's Grid + Game
    over PYPOWER's algorithm (scipy.sparse + SuperLU, the library class the reference itself runs; PYPOWER is not installed here) --
    one environment on one host core, the same workload.  That is what the reference is: one Python process per environment
    (README.md:9 quotes ~25 steps/s on default14).
    `checker_port_*` = the C checker (oracle/ppn_oracle.c, OpenMP over environments, every host thread): reported because it is the
    fastest CPU code in this repository, NOT as "the CPU" -- it redoes the symbolic factorisation in every solve and was written to
    be compared against, not to be fast; a KLU-class refactorising solver would be several times faster."""
    import subprocess
    from harness import oracle_engine            # tests/harness.py: the checker library is driven from outside the product
    out = {'value': None, 'unit': 'env-steps/s', 'cores': 1, 'kind': 'port', 'sample': None}
    try:
        from oracle.game_np import OracleGame
        g = OracleGame(case, conf, chronics, thermal_limits=limits)
        a1 = np.zeros(case.action_length, dtype=np.int64)
        t = time.perf_counter()
        n1 = 0
        while time.perf_counter() - t < budget_s:
            if g.step(a1)[3]:
                g.process_game_over()
            n1 += 1
        el1 = time.perf_counter() - t
        out['value'] = n1 / el1
        out['sample'] = ('1 env x %d steps of the same workload (%.1f s), oracle/game_np.py: numpy + scipy SuperLU restatement of the '
                         'reference Python backend, 1 core' % (n1, el1))
        # (the keys round 4/5 lines carried, kept for comparisons across rounds)
        out['python_restatement_value'] = out['value']
        out['python_restatement_cores'] = 1
    except Exception as ex:
        out['sample'] = 'python restatement failed: %s' % ex
    try:
        lib = os.path.join(ROOT, 'oracle', '_build', 'liboracle.so')
        if not os.path.exists(lib):
            subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'oracle')])
        cores = os.cpu_count() or 1
        nb = max(256, 16 * cores)
        eng = oracle_engine(case, conf, nb, chronics=chronics, thermal_limits=limits)
        threads = max(1, eng.dim(12))
        slots, t0 = env_assignment(0, nb, chronics)
        eng.reset(chronic_slot=slots, t0=t0)
        act = np.zeros((nb, case.action_length), dtype=np.uint8)
        eng.step(act, auto_reset=True)
        t = time.perf_counter()
        steps = 0
        while True:
            eng.step(act, auto_reset=True)
            steps += 1
            el = time.perf_counter() - t
            if el > budget_s or steps >= 4000:
                break
        out['checker_port_value'] = nb * steps / el
        out['checker_port_cores'] = threads
        out['checker_port_sample'] = '%d envs x %d steps (%.1f s), C checker oracle/ppn_oracle.c, OpenMP over envs' % (nb, steps, el)
        out['checker_port_note'] = 'unoptimised checker (symbolic work redone per solve): not a tuned CPU solver'
    except Exception as ex:
        out['checker_port_value'] = None
        out['checker_port_sample'] = 'failed: %s' % ex
    return out


# SURVEY.md 8d algorithmic bytes per env-step of the other configurations: (B_io, B per Newton iteration, B per fast-decoupled
# half-iteration, B per fast-decoupled refactorisation)
BYTES_8D = {'default14': (2.6e3, 9.0e3, 9.0e3 * 22.0 / 72.0, 9.0e3 * 25.0 / 72.0),      # (@14 the FD figures are scaled from @118)
            'default118': (B_IO, B_IT_NR, 22.0e3, 25.0e3)}


def load_env_fixture(envname, solver):
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


def limits_110(case):
    with open(os.path.join(ROOT, 'tests', 'golden', 'envs', ENV_NAME, 'bench_limits_110.json')) as f:
        lim = np.asarray(json.load(f)['limits_a'], dtype=np.float64)
    assert lim.shape == (case.nl,)
    return lim


def random_node_splitting(case, rng, batch):
    """RandomNodeSplitting-style actions (reference pypownet/agent.py:116-158): one random substation
    per environment gets a random configuration of its switches."""
    acts = np.zeros((batch, case.action_length), dtype=np.uint8)
    subs = rng.integers(case.nS, size=batch)
    for b in range(batch):
        idx = np.asarray(case.mapping_array[int(subs[b])], dtype=int)
        acts[b, idx] = rng.integers(0, 2, size=len(idx))
    return acts


def philox_node_splitting(case, env_ids, k, seed=1234):
    """per environment and step a RandomNodeSplitting-style action (reference pypownet/agent.py:116-158) drawn
    from a counter-based generator philox(seed, env, step) -- substation ~ U{nS}, configuration ~ Bernoulli(0.5)^k_s.  A function of
    the GLOBAL environment id and the step alone: however the batch is sharded over ranks, environment g plays the same actions."""
    acts = np.zeros((len(env_ids), case.action_length), dtype=np.uint8)
    for b, g in enumerate(env_ids):
        rng = np.random.Generator(np.random.Philox(key=seed, counter=[int(k), int(g), 0, 0]))
        idx = np.asarray(case.mapping_array[int(rng.integers(case.nS))], dtype=int)
        acts[b, idx] = rng.integers(0, 2, size=len(idx))
    return acts


SPLIT_ACTION_MATRICES = 8      # the split workload cycles through this many pre-drawn action matrices (step s plays matrix s mod 8)


def library_sha256():
    """Hash of the library this process loads (pypownet_amd/libppn.so): stamped into profiles/r*_pmc_summary.json by
    tools/summarize_pmc.py, so that a PMC pass of ANOTHER build is not reported as this build's traffic (VERDICT r05 #7a)."""
    import hashlib
    h = hashlib.sha256()
    with open(os.path.join(ROOT, 'pypownet_amd', 'libppn.so'), 'rb') as f:
        for blk in iter(lambda: f.read(1 << 20), b''):
            h.update(blk)
    return h.hexdigest()


def async_rate(eng, case, B, steps, min_ready=256, workgroups=0, mode='noop', device=0, layout='full', f32=False):
    """CLOSED LOOP WITH AN EXTERNAL POLICY, NO BATCH BARRIER (include/ppn.h: ppn_async_start / ppn_send / ppn_recv): a torch "policy"
    on this GPU receives whichever environments have finished their step (>= min_ready of them, the observation rows in a device
    tensor), decides (do nothing) and sends exactly those again; stragglers keep running.  mode 'noop': the policy only passes the
    ids on; 'reads_rows': it gathers the observation rows of the ready environments and reduces every value of them (what a real
    policy's first layer costs in memory traffic).  Steps counted by PPN_F_N_STEPS, timed until the last one sent is complete."""
    import torch
    dev = 'cuda:%d' % device
    n_obs = eng.observation_length(layout)
    obs_t = torch.zeros((B, n_obs), dtype=torch.float32 if f32 else torch.float64, device=dev)
    rep_t = torch.zeros((B, 3), dtype=torch.float64, device=dev)
    ids_d = torch.zeros((B,), dtype=torch.int32, device=dev)
    acts = torch.zeros((B, case.action_length), dtype=torch.uint8, device=dev)
    sink = torch.zeros((B,), dtype=obs_t.dtype, device=dev)
    torch.cuda.synchronize()
    eng.sync()
    n0 = int(eng.read('N_STEPS').astype(np.int64).sum())
    eng.async_start(obs_t.data_ptr(), obs_t.numel() * obs_t.element_size(), rep_t.data_ptr(), layout=layout,
                    dtype=np.float32 if f32 else np.float64, workgroups=workgroups)
    st = torch.cuda.ExternalStream(eng.async_stream_ptr(), device=dev)
    aptr, iptr = acts.data_ptr(), ids_d.data_ptr()
    all_ids = np.arange(B, dtype=np.int32)
    receives = 0
    with torch.cuda.stream(st):
        def loop(target):
            nonlocal receives
            total = 0
            while total < target:
                ids = eng.recv(min_ready=min_ready, ids_device_ptr=iptr)
                n = len(ids)
                if mode == 'reads_rows':
                    ix = ids_d[:n].long()
                    sink.index_copy_(0, ix, obs_t.index_select(0, ix).sum(dim=1))
                eng.send_device(ids, aptr, rows_by_env=True)
                total += n
                receives += 1
            return total
        eng.send_device(all_ids, aptr, rows_by_env=True)
        warm = loop(3 * B)                            # warm-up (the server, torch's kernels, the allocator)
        receives = 0
        w0 = int(eng.async_stats()['in_flight'])      # (always B: everything received is sent again)
        t0 = time.perf_counter()
        timed_recv = loop(B * steps)
        while eng.async_stats()['in_flight']:         # the steps still in flight belong to the timed region
            eng.recv(min_ready=eng.async_stats()['in_flight'])
        el = time.perf_counter() - t0
    stats = eng.async_stats()
    eng.async_stop()
    n1 = int(eng.read('N_STEPS').astype(np.int64).sum())
    timed = timed_recv + w0                           # steps completed inside the timed region: everything received in it, and the drain
    assert n1 - n0 == B + warm + timed_recv, (n1 - n0, B, warm, timed_recv)      # every step sent was executed, once
    return {'rate': timed / el, 'receives': receives, 'per_receive': timed_recv / max(receives, 1), 'workgroups': stats['workgroups'],
            'restarts': stats['server_restarts']}


def side_config(name, envname, solver, batch, steps, device, auto_reset, limits=None, split=False, max_active=None, warmup=4, restart_memo=False,
                histogram=False, lu_capacity=0, watch_capacity=0, q_plane_auto=0):
    """One of the other single-GPU configurations of BASELINE.json, timed like the headline (device-resident actions, K steps
    between synchronisations, step-kernel time from HIP events) and priced with its own SURVEY.md 8d byte count."""
    import torch
    from pypownet_amd.engine import Engine
    case, conf, chronics = load_env_fixture(envname, solver)
    kw = {}
    if max_active:
        kw['max_active_buses'] = max_active
    if lu_capacity:
        kw['lu_capacity'] = lu_capacity
    if q_plane_auto:
        kw['q_plane_auto'] = 1
    eng = Engine(case, conf, batch, device=device, chronics=chronics, thermal_limits=limits, **kw)
    if restart_memo:      # (include/ppn.h: ppn_restart_memo -- side figures only, the headline computes every restart)
        eng.restart_memo(True)
    slots, t0 = env_assignment(0, batch, chronics)
    eng.reset(chronic_slot=slots, t0=t0)
    n_act = SPLIT_ACTION_MATRICES if split else 1
    acts = [torch.from_numpy(philox_node_splitting(case, np.arange(batch), k) if split else
                             np.zeros((batch, case.action_length), dtype=np.uint8)).to('cuda:%d' % device) for k in range(n_act)]
    torch.cuda.synchronize()
    for k in range(warmup):
        eng.step_device(acts[k % n_act].data_ptr(), auto_reset=auto_reset)
    eng.sync()
    s0, i0 = eng.read('N_SOLVES').astype(np.int64).sum(), eng.read('N_ITERS').astype(np.int64).sum()
    st0 = eng.read('N_STEPS').astype(np.int64).sum()
    eng.kernel_time(reset=True)
    n_done, depth_hist = 0, np.zeros(8, dtype=np.int64)
    nst = eng.read('N_STEPS').copy() if histogram else None
    t = time.perf_counter()
    for k in range(steps):
        eng.step_device(acts[k % n_act].data_ptr(), auto_reset=auto_reset)
        if histogram:      # (report fields: reading them does not settle the deferred restarts; it does synchronise)
            # an environment whose restarts keep diverging (PPN_F_DEAD = 3) executes no step and keeps the DONE / depth of its
            # last one: only environments whose PPN_F_N_STEPS moved are counted (include/ppn.h)
            now = eng.read('N_STEPS')
            stepped = now != nst
            nst = now.copy()
            n_done += int(eng.read('DONE')[stepped].sum())
            depth_hist += np.bincount(np.minimum(eng.read('CASCADE_DEPTH')[stepped], 7), minlength=8)
    eng.sync()
    el = time.perf_counter() - t
    kms, kn = eng.kernel_time(reset=True)
    s1, i1 = eng.read('N_SOLVES').astype(np.int64).sum(), eng.read('N_ITERS').astype(np.int64).sum()
    # env-steps = Game.step calls the engine EXECUTED (PPN_F_N_STEPS): an environment that is over and whose restart keeps
    # diverging (PPN_F_DEAD = 3) is launched but does not step, and is not counted
    executed = int(eng.read('N_STEPS').astype(np.int64).sum() - st0)
    dead_end = int((eng.read('DEAD') != 0).sum())
    n_solve, n_it = float(s1 - s0) / max(executed, 1), float(i1 - i0) / max(float(s1 - s0), 1.0)
    b_io, b_nr, b_fd, b_fact = BYTES_8D[envname]
    b_step = b_io + n_solve * (n_it * b_nr if solver == 'newton' else n_it * b_fd + b_fact)
    k_s = (kms / 1e3) / max(kn, 1)
    out = {'config': name, 'env': envname, 'solver': solver, 'batch': batch, 'steps': steps,
           'env_steps_per_s': executed / el, 'env_steps_executed': executed, 'env_steps_launched': batch * steps,
           'dead_envs_at_end': dead_end, 'ms_per_step': 1e3 * el / steps, 'step_kernel_ms': 1e3 * k_s,
           'solves_per_step': n_solve, 'iters_per_solve': n_it, 'lds_bytes_per_env': eng.lds_bytes,
           'algorithmic_bytes_per_env_step': b_step,
           'roofline_frac': (executed / float(steps)) * b_step / k_s / 1e9 / HBM_PEAK_GBS,
           'engine_capacity_flags_last_step': int((eng.read('FLAG') == 4).sum())}
    if restart_memo:
        ms_ = eng.restart_memo_stats()
        out['restart_memo'] = ms_
        out['note_restart_memo'] = ('restarts of ended episodes are computed once per chronic position and copied afterwards (bit-identical state and '
                                    'counters: tests check_restart_memo); solves_per_step / iters_per_solve and the roofline fraction count the served '
                                    'restarts as the solves they stand for -- read *_Msteps, not *_frac, for these entries')
    if split:
        out['illegal_fraction_last_step'] = float((eng.read('ILLEGAL') != 0).mean())
        out['q_plane_auto'] = int(q_plane_auto)
    if watch_capacity:     # untimed: the capacity flag of every environment looked at after every one of some more steps
        raised = 0
        for k in range(watch_capacity):
            eng.step_device(acts[k % n_act].data_ptr(), auto_reset=auto_reset)
            raised += int((eng.read('FLAG') == 4).sum())
        out['lu_capacity'] = lu_capacity
        out['engine_capacity_flags_in_%d_watched_steps' % watch_capacity] = raised
    if not split and not histogram:
        # the same K steps as ONE open-loop launch (ppn_rollout; do-nothing actions only): beside the stepped rate, never instead
        eng.sync()
        r0 = int(eng.read('N_STEPS').astype(np.int64).sum())
        t = time.perf_counter()
        eng.rollout_device(acts[0].data_ptr(), steps, per_step_actions=False, auto_reset=auto_reset)
        eng.sync()
        out['open_loop_rollout_env_steps_per_s'] = (int(eng.read('N_STEPS').astype(np.int64).sum()) - r0) / (time.perf_counter() - t)
    if histogram:
        out['note'] = 'DONE / CASCADE_DEPTH read back every step for the statistics: the rate includes that synchronisation'
        out['game_over_rate'] = n_done / float(max(executed, 1))
        out['cascade_depth_histogram'] = [int(v) for v in depth_hist]
    eng.close()
    return out


# keys of the flat `other_configs` map of the bench line, in the order other_configs() returns its entries: <key>_Msteps (M env-steps/s),
# <key>_frac (roofline fraction on the entry's own SURVEY.md 8d byte count), <key>_kernel_ms (step kernel, HIP events)
OTHER_CONFIG_KEYS = ['cfg1_d14_nr_b1024', 'cfg1_d14_nr_b16384', 'cfg2_d118_fdxb_b4096', 'cfg3_d118_nr_b32768',
                     'cfg4_split_b1024_safe', 'cfg4_split_b1024_tuned', 'cfg4_split_b8192_tuned', 'cfg2_rule110_b4096',
                     'cfg3_d118_nr_b32768_memo', 'cfg4_split_b1024_safe_memo', 'cfg4_split_b8192_tuned_memo', 'cfg2_rule110_b4096_memo']


def search_rate(device, batch=1024, k=8, rounds=8):
    """Topology-action search (ppn_simulate_candidates): k node-splitting candidates per environment forked from its
    current state and simulated in one call, followed by a do-nothing step -- a search agent's round; simulated candidates per second."""
    import torch
    from pypownet_amd.engine import Engine
    case, conf, chronics = load_env_fixture(ENV_NAME, 'newton')
    eng = Engine(case, conf, batch, device=device, chronics=chronics, thermal_limits=bench_limits(case))
    slots, t0 = env_assignment(0, batch, chronics)
    eng.reset(chronic_slot=slots, t0=t0)
    rng = np.random.default_rng(5)
    cands = np.zeros((batch * k, case.action_length), dtype=np.uint8)
    for c in range(batch * k):
        idx = np.asarray(case.mapping_array[int(rng.integers(case.nS))], dtype=int)
        cands[c, idx] = rng.integers(0, 2, size=len(idx))
    d_c = torch.from_numpy(cands).to('cuda:%d' % device)
    act = torch.zeros((batch, case.action_length), dtype=torch.uint8, device='cuda:%d' % device)
    env_ids = np.repeat(np.arange(batch, dtype=np.int32), k)
    torch.cuda.synchronize()
    for _ in range(2):
        eng.simulate_candidates_device(d_c.data_ptr(), env_ids)
        eng.step_device(act.data_ptr(), auto_reset=1)
    eng.sync()
    t = time.perf_counter()
    for _ in range(rounds):
        eng.simulate_candidates_device(d_c.data_ptr(), env_ids)
        eng.step_device(act.data_ptr(), auto_reset=1)
    eng.sync()
    el = time.perf_counter() - t
    eng.close()
    return batch * k * rounds / el


def other_configs(device, auto_reset, steps):
    """BASELINE.json configs[1], configs[2] with the reference's solver, the one-GPU share of configs[4], the large-batch point
    and the SURVEY.md 8d limit rule of configs[2] -- one entry each in the bench line (`other_configs`)."""
    case118, _, _ = load_env_fixture(ENV_NAME, 'newton')
    lim = bench_limits(case118)
    out = []

    def add(*a, **kw):
        try:
            out.append(side_config(*a, **kw))
        except Exception as ex:      # a side measurement must never take the bench line down
            out.append({'config': a[0], 'error': str(ex)})
    add('configs[1]: default14 AC Newton-Raphson, batch 1024, do-nothing', 'default14', 'newton', 1024, steps, device, auto_reset)
    add('configs[1] at a batch that fills the GPU: default14 AC Newton-Raphson, batch 16384', 'default14', 'newton', 16384, steps,
        device, auto_reset)
    add('configs[2] with the reference solver: default118 fast-decoupled XB, batch 4096, cascade limits', ENV_NAME, 'fdxb', 4096,
        steps, device, auto_reset, limits=lim, max_active=case118.nS)
    add('configs[2] / configs[3] at 32768 environments on one GPU', ENV_NAME, 'newton', 32768, max(8, steps // 3), device, auto_reset,
        limits=lim, max_active=case118.nS)
    add('configs[4] share of one GPU: default118 AC Newton-Raphson, random node splitting every step, batch 1024, every busbar may '
        'be active (W = 4 kernels); DEFAULT capacities: two-capacity stepping (small storage 1.48 x the base pattern, four environments per CU; large storage 2.15 x for schedules that do not fit)',
        ENV_NAME, 'newton', 1024, steps, device, auto_reset, limits=lim, split=True, watch_capacity=20)
    # matrix capacity 1.5 x the base pattern instead of the default 2.15 x (tools/fill_survey.py: the largest pattern over 10^6 random
    # topologies is 1.39 x): the working set drops under 40,960 bytes = 32 LDS granules, four environments per CU instead of three
    add('configs[4] share of one GPU with the capacity knobs: rules.lu_capacity = 3976 (pattern capacity 1.5 x base), '
        'rules.q_plane_auto = 1 (chronic-derived Q plane): 4 environments per CU, batch 1024; capacity flags watched',
        ENV_NAME, 'newton', 1024, steps, device, auto_reset, limits=lim, split=True, lu_capacity=3976, watch_capacity=20, q_plane_auto=1)
    add('configs[4] workload at a batch that fills the GPU: random node splitting every step, batch 8192 (the whole of configs[4] on '
        'one GPU), rules.lu_capacity = 3976, rules.q_plane_auto = 1', ENV_NAME, 'newton', 8192, max(8, steps // 3), device, auto_reset,
        limits=lim, split=True, lu_capacity=3976, watch_capacity=20, q_plane_auto=1)
    add('configs[2] with the limit rule of SURVEY.md 8d config 3: limit = max(50, 1.10 x I(t = 0))', ENV_NAME, 'newton', 4096, steps,
        device, auto_reset, limits=limits_110(case118), max_active=case118.nS, histogram=True)
    # ---- the same four with the RESTART MEMO on (ppn_restart_memo; round 6): restarts of ended episodes served from snapshots ----
    # (40 warm-up steps: the engine's learning passes -- the first 32 steps after the memo is set up -- are behind the timed region)
    add('cfg3 at 32768 environments, restart memo on', ENV_NAME, 'newton', 32768, max(8, steps // 3), device, auto_reset,
        limits=lim, max_active=case118.nS, restart_memo=True, warmup=40)
    add('configs[4] share of one GPU (batch 1024, default capacities), restart memo on', ENV_NAME, 'newton', 1024, steps, device, auto_reset,
        limits=lim, split=True, restart_memo=True, warmup=40)
    add('configs[4] workload at batch 8192 (capacity knobs), restart memo on', ENV_NAME, 'newton', 8192, max(8, steps // 3), device, auto_reset,
        limits=lim, split=True, lu_capacity=3976, q_plane_auto=1, restart_memo=True, warmup=40)
    add('configs[2] with the 110 % limit rule, restart memo on', ENV_NAME, 'newton', 4096, steps, device, auto_reset,
        limits=limits_110(case118), max_active=case118.nS, histogram=True, restart_memo=True, warmup=40)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=60)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=BATCH_PER_GPU, help='environments per GPU')
    ap.add_argument('--workload', choices=['cascade', 'split'], default='cascade',
                    help="cascade: BASELINE.json configs[2]/[3], do-nothing agent, cascade limits (the headline).  split: configs[4], "
                         'per-environment random node-splitting actions every step (philox(1234, env, step)), every busbar may be '
                         'active (four-word kernels, default capacities: two-capacity stepping) -- runs under --gpus N like the headline')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the side measurements of the other single-GPU configurations')
    ap.add_argument('--no-rollout', action='store_true',
                    help='skip the open-loop rollout side figure (profiling passes: its one long launch runs the same kernel symbol as '
                         'the timed step launches and would skew per-launch averages)')
    ap.add_argument('--headline-only', action='store_true',
                    help='only the timed headline loop: no closed-loop / host-boundary / rollout side figures, no other configurations, no '
                         'CPU baseline (profiling passes: every launch of the step kernel symbol then is a launch of the timed region or '
                         'its warm-up)')
    ap.add_argument('--single-controller', action='store_true',
                    help="BASELINE.json configs[3]'s exchange variant: every step rank 0 scatters the actions of ALL "
                         'environments and gathers (done, flag, reward) over RCCL; default: policy per GPU, no collective')
    args = ap.parse_args()
    if args.headline_only:
        args.no_rollout = args.no_other_configs = args.no_cpu_baseline = True

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # launched bare (`python bench.py --gpus N`): spawn the N ranks ourselves, one per GPU, the way the driver's launcher
        # does (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...); the exit code is
        # the launcher's -- non-zero unless all N ranks joined and finished
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(('127.0.0.1', 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('WORLD_SIZE (%d) != --gpus (%d): %d ranks joined' % (world, args.gpus, world))
    if os.environ.get('PPN_BENCH_BACKEND', 'nccl') == 'nccl' and torch.cuda.is_available() and torch.cuda.device_count() < world:
        raise SystemExit('--gpus %d needs %d GPUs, %d visible' % (args.gpus, world, torch.cuda.device_count()))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (the engine has no CPU fallback)')
    # PPN_BENCH_BACKEND=gloo: smoke test of the multi-rank path on a box with fewer GPUs than ranks (ranks share devices;
    # the numbers mean nothing then).  The driver's runs use the default: one GPU per rank, RCCL.
    backend = os.environ.get('PPN_BENCH_BACKEND', 'nccl')
    if backend != 'nccl':
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or os.environ.get('PPN_BENCH_FORCE_DIST') == '1'   # (the latter: 1-rank smoke test of the RCCL path)
    if use_dist:
        if backend == 'nccl':
            try:      # bind the communicator to this rank's GPU up front (no lazy device guess in barrier())
                dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
            except TypeError:
                dist.init_process_group(backend, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)   # nccl = RCCL over xGMI

    from pypownet_amd.engine import Engine
    case, conf, chronics = load_workload()
    limits = bench_limits(case)
    B = args.batch
    # the HIP events that bracket a step launch (roofline.avg_kernel_ms) cost ~7 us of stream time each pair -- 2 % of a step: the
    # engine brackets every 4th launch of the timed region only (every launch when K is small); `value` is wall-clock over all K
    timing_every = int(os.environ.get('PPN_KERNEL_TIMING_EVERY', '4' if args.steps >= 16 else '1'))
    os.environ['PPN_KERNEL_TIMING_EVERY'] = str(timing_every)
    SPLIT = args.workload == 'split'
    if SPLIT:      # configs[4]: node switches every step -- every busbar may be active, the engine's default capacities
        eng = Engine(case, conf, B, device=local_rank, chronics=chronics, thermal_limits=limits,
                     lu_capacity=int(os.environ.get('PPN_BENCH_LU_CAPACITY', '0')), q_plane_auto=int(os.environ.get('PPN_BENCH_Q_PLANE_AUTO', '0')))
    else:
        eng = Engine(case, conf, B, device=local_rank, chronics=chronics, thermal_limits=limits, max_active_buses=case.nS,
                     lu_capacity=int(os.environ.get('PPN_BENCH_LU_CAPACITY', '0')))   # (occupancy experiments only)
    slots, t0 = env_assignment(rank * B, B, chronics)
    eng.reset(chronic_slot=slots, t0=t0)
    import zlib
    my_crc = zlib.crc32(slots.tobytes() + t0.tobytes())      # which environments this rank plays (checked by the 2-rank test)
    dev_name = 'cuda:%d' % local_rank
    if SPLIT:
        host_acts = [philox_node_splitting(case, np.arange(rank * B, (rank + 1) * B), k) for k in range(SPLIT_ACTION_MATRICES)]
        for h_ in host_acts:
            my_crc = zlib.crc32(h_.tobytes(), my_crc)         # ... and which actions they play
        act_list = [torch.from_numpy(h_).to(dev_name) for h_ in host_acts]
    else:
        act_list = [torch.zeros((B, case.action_length), dtype=torch.uint8, device=dev_name)]
    crcs = [my_crc]
    if use_dist:
        crcs = [None] * world
        dist.all_gather_object(crcs, my_crc)
    actions = act_list[0]
    torch.cuda.synchronize()
    aptr = actions.data_ptr()
    aptrs = [a_.data_ptr() for a_ in act_list]
    step_no = [0]

    def next_actions():      # (the action matrix of the next step: the same for every step of the do-nothing workload)
        k_ = step_no[0] % len(aptrs)
        step_no[0] += 1
        return k_

    # auto_reset = 2 (include/ppn.h, ppn_step): an episode that ends is restarted at the head of the environment's NEXT step
    # launch instead of at the tail of this one; eng.sync() settles the restarts still owed, so that every restart of the K
    # timed steps is inside the timed region
    AUTO_RESET = int(os.environ.get('PPN_BENCH_AUTO_RESET', '2'))
    for _ in range(args.warmup):
        eng.step_device(aptrs[next_actions()], auto_reset=AUTO_RESET)
    eng.sync()
    exchange = None
    if args.single_controller and use_dist:
        dev = 'cuda:%d' % local_rank
        all_actions = [torch.zeros_like(actions) for _ in range(world)] if rank == 0 else None   # the controller's choice
        on_host = backend != 'nccl'      # (gloo smoke mode: the collectives run on host copies)
        if on_host and rank == 0:
            all_actions = [t_.cpu() for t_ in all_actions]

        # one [B x 3] report row block per step (PPN_F_STEP_REPORT: done, flag, reward sum -- written by the step kernel's epilogue),
        # stream waits instead of host synchronisations, two buffer sets alternating: the host only enqueues, step t + 1's scatter is
        # queued while step t's gather is in flight (round 4: two host synchronisations + three reads + a pack per step)
        ext = torch.cuda.ExternalStream(eng.stream_ptr(), device=dev)
        xb = [dict(act=torch.zeros_like(actions), res=torch.zeros((B, 3), dtype=torch.float64, device=dev), free=None,
                   out=([torch.empty((B, 3), dtype=torch.float64, device=dev) for _ in range(world)] if rank == 0 else None)) for _ in range(2)]
        if on_host and rank == 0:
            for b_ in xb:
                b_['out'] = [t_.cpu() for t_ in b_['out']]
        turn = [0]
        # the gather runs on a communicator and a stream of its own: the scatter of step t + 1 (default group, scatter stream) and
        # step t + 1 itself then do not queue behind the gather of step t -- what a controller whose policy does not need step t's
        # report to choose the actions of step t + 1 (the do-nothing controller of this bench) can overlap
        # (RCCL documents collectives issued concurrently on TWO communicators as deadlock-prone when their device-side
        #  order differs across ranks -- the second communicator is only used on a world of ONE rank, where there is no other rank
        #  to disagree with; with more ranks both collectives go through the default communicator, in program order on every rank)
        g_gather = dist.new_group(backend=backend) if (not on_host and world == 1) else None
        s_scatter, s_gather = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)) if not on_host else (None, None)

        def exchange():
            b_ = xb[turn[0]]
            turn[0] ^= 1
            if on_host:
                cur = torch.cuda.current_stream()
                recv = torch.empty(actions.shape, dtype=actions.dtype)
                dist.scatter(recv, all_actions, src=0)
                b_['act'].copy_(recv)
                ext.wait_stream(cur)
                eng.step_device(b_['act'].data_ptr(), auto_reset=AUTO_RESET)
                eng.read_into_device('STEP_REPORT', b_['res'].data_ptr(), 24 * B)
                eng.wait()
                dist.gather(b_['res'].cpu(), b_['out'], dst=0)
                return
            with torch.cuda.stream(s_scatter):
                if b_['free'] is not None:
                    s_scatter.wait_event(b_['free'])       # this buffer set's gather (two steps ago) is through -- and so is the step before it
                dist.scatter(b_['act'], all_actions, src=0)            # [B x action_length] u8 to every rank
                ev_sc = s_scatter.record_event()
            ext.wait_event(ev_sc)                          # the engine's stream waits for the scatter: an event, no host synchronisation
            eng.step_device(b_['act'].data_ptr(), auto_reset=AUTO_RESET)
            eng.read_into_device('STEP_REPORT', b_['res'].data_ptr(), 24 * B)
            ev_st = ext.record_event()
            with torch.cuda.stream(s_gather):
                s_gather.wait_event(ev_st)
                dist.gather(b_['res'], b_['out'], dst=0, group=g_gather)      # 24 B per environment back to the controller
                b_['free'] = s_gather.record_event()
        exchange()

    eng.sync()
    ns0, ni0 = eng.read('N_SOLVES').astype(np.int64).sum(), eng.read('N_ITERS').astype(np.int64).sum()
    nst0 = eng.read('N_STEPS').astype(np.int64).sum()
    eng.kernel_time(reset=True)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    eng.sync()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        if exchange is not None:
            exchange()
        else:
            eng.step_device(aptrs[next_actions()], auto_reset=AUTO_RESET)
    eng.sync()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    step_form = eng.last_step_form()

    kms, klaunch = eng.kernel_time(reset=True)
    ns1, ni1 = eng.read('N_SOLVES').astype(np.int64).sum(), eng.read('N_ITERS').astype(np.int64).sum()
    done_now = int(eng.read('DONE').sum())
    depth_now = float(eng.read('CASCADE_DEPTH').mean())
    executed = float(eng.read('N_STEPS').astype(np.int64).sum() - nst0)      # Game.step calls executed by this rank (PPN_F_N_STEPS)
    dead_now = float((eng.read('DEAD') != 0).sum())
    # the same loop once more over 60 steps (VERDICT r05 #7c: at the driver's K = 20 the timed region is 6.5 ms and the last digits of
    # `value` are noise): `value_k60`, reported beside `value`, never instead of it -- 20 ms of GPU time
    K60 = 60
    value_k60 = None
    if exchange is None and args.steps < K60 and not args.headline_only:
        e0 = float(eng.read('N_STEPS').astype(np.int64).sum())
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t60 = time.perf_counter()
        for _ in range(K60):
            eng.step_device(aptrs[next_actions()], auto_reset=AUTO_RESET)
        eng.sync()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        el60 = time.perf_counter() - t60
        st60 = torch.tensor([el60, float(eng.read('N_STEPS').astype(np.int64).sum()) - e0], dtype=torch.float64,
                            device=dev_name if backend == 'nccl' else 'cpu')
        if use_dist:
            mx60, sm60 = st60.clone(), st60.clone()
            dist.all_reduce(mx60, op=dist.ReduceOp.MAX)
            dist.all_reduce(sm60, op=dist.ReduceOp.SUM)
            value_k60 = float(sm60[1]) / float(mx60[0])
        else:
            value_k60 = float(st60[1]) / float(st60[0])
        eng.kernel_time(reset=True)
    stats = torch.tensor([elapsed, float(ns1 - ns0), float(ni1 - ni0), kms, float(klaunch), executed, dead_now], dtype=torch.float64,
                         device=('cuda:%d' % local_rank) if backend == 'nccl' else 'cpu')
    if use_dist:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0])
        n_solves, n_iters, executed_all, dead_all = float(sm[1]), float(sm[2]), float(sm[5]), float(sm[6])
    else:
        n_solves, n_iters, executed_all, dead_all = float(stats[1]), float(stats[2]), executed, dead_now

    if rank == 0:
        # env-steps = steps the engines EXECUTED; equals batch x ranks x K unless an environment sat out a launch waiting for a
        # restart that keeps diverging (PPN_F_DEAD = 3) -- `dead_envs_at_end` / `env_steps_launched` make that visible
        total_steps = executed_all
        solves_per_step = n_solves / total_steps
        iters_per_solve = n_iters / max(n_solves, 1.0)
        b_step = B_IO + solves_per_step * iters_per_solve * B_IT_NR          # algorithmic bytes per env-step
        avg_kernel_s = (kms / 1e3) / max(klaunch, 1)                          # rank 0's step kernel, HIP events
        per_launch = executed / float(args.steps)                              # env-steps rank 0's kernel executes per launch
        achieved = per_launch * b_step / avg_kernel_s / 1e9 if avg_kernel_s > 0 else None
        pmc = pmc_summary(B) if not SPLIT else None
        traffic, traffic_src = measured_traffic(B) if not SPLIT else (None, None)
        flop_step = solves_per_step * iters_per_solve * FLOP_PER_NR_ITERATION
        tflops = per_launch * flop_step / avg_kernel_s / 1e12 if avg_kernel_s > 0 else None
        sq = (pmc or {}).get('sq_shares_of_wave_cycles') or {}
        form_name, two_cap_form = step_form      # (what the engine launched: ppn_dim(19))
        # NOTE on the shape of this line: the driver's parser keeps scalars one level deep and cuts strings at ~130 characters
        # (round 4: nested dicts, lists and the long `other_configs` entries were dropped) -- everything below is flat and short;
        # the long form of the side measurements goes to PPN_BENCH_DETAILS (default gpurun_out/bench_details.json when that
        # directory exists) and into DESIGN.md section 5
        out = {
            'metric': 'env steps/sec, batched IEEE-118 AC load-flow',
            'value': total_steps / elapsed,
            'unit': 'env-steps/s',
            'value_k60': value_k60,      # the same loop over 60 steps in the same run (None when --steps >= 60: `value` is that)
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f64',
            'data': 'synthetic: IEEE-118 case + reference chronic series (fixture), synthetic thermal limits',
            'config': {'workload': ('default118 AC Newton (tol 1e-6), %d envs/GPU, cascade loop, per-env random node splitting every step '
                                    '(philox(1234, env, step)), every busbar may be active, auto reset' if SPLIT else
                                    'default118 AC Newton (tol 1e-6), %d envs/GPU, cascade loop, do-nothing agent, auto reset') % B,
                       'workload_name': args.workload,
                       'batch_per_gpu': B, 'solver': 'newton',
                       'parallelism': ('env-sharded x%d, single controller: RCCL scatter actions + gather report per step' if exchange is not None else
                                       'env-sharded x%d, no collective in the step loop') % world,
                       'solves_per_step': solves_per_step, 'iters_per_solve': iters_per_solve,
                       'lds_bytes_per_env': eng.lds_bytes, 'envs_done_at_end': done_now, 'auto_reset_mode': AUTO_RESET,
                       'env_steps_executed': int(executed_all), 'env_steps_launched': B * world * args.steps,
                       'dead_envs_at_end': int(dead_all), 'n_chronics': len(chronics),
                       'env_assignment_crc32': crcs,
                       'single_controller': exchange is not None, 'dist_backend': (backend if use_dist else None),
                       'mean_cascade_depth_last_step': depth_now},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': (achieved / HBM_PEAK_GBS) if achieved else None, 'traffic': traffic,
                         # `traffic` is NOT measured in this run (counters cannot be collected from inside the timed run): it is the
                         # rocprofv3 PMC pass of this build committed under profiles/ (FETCH_SIZE x 2 + WRITE_SIZE per launch)
                         'traffic_source': traffic_src,
                         'kernel': 'ppn_kernel<W=%d,%s,NT=1>%s' % (4 if SPLIT else 2, form_name,
                                                                   ' x 2 capacity classes + schedule pre-pass (one event bracket)' if two_cap_form else ''),
                         'avg_kernel_ms': 1e3 * avg_kernel_s,
                         'kernel_timing': 'HIP events on the engine stream, every %d%s step launch: %d launches' % (
                             timing_every, 'th' if timing_every > 1 else 'st', int(klaunch)),
                         'algorithmic_bytes_per_env_step': b_step,
                         # the compulsory state I/O alone (an LDS-resident solver legitimately moves fewer bytes
                         # than the streaming model; traffic / achieved bytes shows it)
                         'compulsory_io_bytes_per_env_step': B_IO,
                         'compulsory_io_frac': (per_launch * B_IO / avg_kernel_s / 1e9 / HBM_PEAK_GBS) if avg_kernel_s > 0 else None,
                         # what actually binds the kernel: the dependency chains of one wavefront per environment.  useful flops =
                         # SURVEY.md 8d's 55 kflop per Newton iteration x measured iterations; SQ shares from the committed PMC pass
                         'fp64_useful_flop_per_env_step': flop_step, 'fp64_achieved_tflops': tflops,
                         'fp64_peak_tflops': FP64_VECTOR_PEAK_TFLOPS,
                         'fp64_frac': (tflops / FP64_VECTOR_PEAK_TFLOPS) if tflops else None,
                         'sq_active_inst_any_frac': sq.get('SQ_ACTIVE_INST_ANY'), 'sq_wait_any_frac': sq.get('SQ_WAIT_ANY')},
            'cpu_baseline': None,
        }
        details = {}
        if world == 1 and not SPLIT and not args.no_rollout:
            # OPEN-LOOP rollout (ppn_rollout): the same K steps of the same do-nothing agent in ONE launch -- every environment
            # plays its K steps back to back instead of waiting, after every step, for the longest cascade of the batch.  Same
            # results bit for bit (tests: check_rollout_equals_steps); only usable when the actions do not depend on the
            # observations in between, so it is reported beside the headline, never as `value`
            eng.sync()
            r0 = int(eng.read('N_STEPS').astype(np.int64).sum())
            eng.kernel_time(reset=True)
            t_r = time.perf_counter()
            # (auto_reset = 1: the fused restart goes through the work-queue kernel since round 6 -- same trajectories as the deferred form)
            eng.rollout_device(aptr, args.steps, per_step_actions=False, auto_reset=1)
            eng.sync()
            el_r = time.perf_counter() - t_r
            r1 = int(eng.read('N_STEPS').astype(np.int64).sum())
            out['config']['open_loop_rollout_env_steps_per_s'] = (r1 - r0) / el_r      # (ppn_rollout, open-loop agents only; not the headline)
        if world == 1 and not SPLIT and exchange is None and not args.headline_only:
            # CLOSED LOOP WITH THE OBSERVATION (what RunEnv.step returns, environment.py:848-866): the same step followed by the
            # observation gather (K_OBS) into a device tensor, every step -- what a policy that lives on this GPU pays.  Full
            # Observation.as_array() in float64 (4 967 values = 39.7 KB per environment) and the minimalist layout in float32.
            # (fused restart, auto_reset = 1: looking at the state after every step settles a deferred restart with a launch of its
            #  own -- 0.18 ms on this workload, profiles/r05_bench_rocprofv3_kernel_stats.csv -- so the deferral only pays for agents
            #  that do not look; PPN_BENCH_OBS_AUTO_RESET=2 measures that form)
            # Since round 5 in ONE launch (ppn_step_observe: every environment's workgroup writes its row right behind its step);
            # the two-launch form (ppn_step, then ppn_read_observation) is measured beside it.
            OBS_AR = int(os.environ.get('PPN_BENCH_OBS_AUTO_RESET', '1'))
            for key, lay, f32 in (('closed_loop_with_observation_env_steps_per_s', 'full', False),
                                  ('closed_loop_with_minimalist_f32_observation_env_steps_per_s', 'minimalist', True)):
                if OBS_AR != 1:
                    break
                n_obs = eng.observation_length(lay)
                obs_t = torch.empty((B, n_obs), dtype=torch.float32 if f32 else torch.float64, device='cuda:%d' % local_rank)
                nb = obs_t.numel() * obs_t.element_size()
                torch.cuda.synchronize()
                for _ in range(3):
                    eng.step_observe_device(aptr, obs_t.data_ptr(), nb, auto_reset=True, layout=lay, dtype=np.float32 if f32 else np.float64)
                eng.sync()
                c0 = int(eng.read('N_STEPS').astype(np.int64).sum())
                t_o = time.perf_counter()
                for _ in range(args.steps):
                    eng.step_observe_device(aptr, obs_t.data_ptr(), nb, auto_reset=True, layout=lay, dtype=np.float32 if f32 else np.float64)
                eng.sync()
                out['config'][key] = (int(eng.read('N_STEPS').astype(np.int64).sum()) - c0) / (time.perf_counter() - t_o)
                del obs_t
            for key, lay, f32 in (('closed_loop_with_observation_two_launches_env_steps_per_s', 'full', False),
                                  ('closed_loop_with_minimalist_f32_observation_two_launches_env_steps_per_s', 'minimalist', True)):
                n_obs = eng.observation_length(lay)
                obs_t = torch.empty((B, n_obs), dtype=torch.float32 if f32 else torch.float64, device='cuda:%d' % local_rank)
                nb = obs_t.numel() * obs_t.element_size()
                torch.cuda.synchronize()
                for _ in range(3):
                    eng.step_device(aptr, auto_reset=OBS_AR)
                    eng.observations_into_device(obs_t.data_ptr(), nb, layout=lay, dtype=np.float32 if f32 else np.float64)
                eng.sync()
                c0 = int(eng.read('N_STEPS').astype(np.int64).sum())
                t_o = time.perf_counter()
                for _ in range(args.steps):
                    eng.step_device(aptr, auto_reset=OBS_AR)
                    eng.observations_into_device(obs_t.data_ptr(), nb, layout=lay, dtype=np.float32 if f32 else np.float64)
                eng.sync()
                out['config'][key] = (int(eng.read('N_STEPS').astype(np.int64).sum()) - c0) / (time.perf_counter() - t_o)
                del obs_t
        if world == 1 and not SPLIT and exchange is None and not args.no_rollout:
            # CLOSED LOOP WITHOUT THE BATCH BARRIER (ppn_rollout_policy, round 5): a policy that lives on the device -- the built-in
            # toy operator: reconnect a reconnectable line, else open the most loaded line beyond its limit -- and every environment
            # on its own clock: work items (step, environment) go to whichever workgroup is free, trajectories bit for bit those of
            # { ppn_policy_actions; ppn_step } round by round (tests: check_policy_rollout_equals_stepping).  Beside it: the same
            # closed loop stepped synchronously (policy kernel + step kernel per step).  Never `value`.
            pol_buf = torch.zeros((B, case.action_length), dtype=torch.uint8, device='cuda:%d' % local_rank)
            torch.cuda.synchronize()
            for _ in range(3):
                eng.policy_actions('line_relief', [1.0], pol_buf.data_ptr())
                eng.step_device(pol_buf.data_ptr(), auto_reset=1)
            eng.sync()
            p0 = int(eng.read('N_STEPS').astype(np.int64).sum())
            t_p = time.perf_counter()
            for _ in range(args.steps):
                eng.policy_actions('line_relief', [1.0], pol_buf.data_ptr())
                eng.step_device(pol_buf.data_ptr(), auto_reset=1)
            eng.sync()
            out['config']['closed_loop_device_policy_stepped_env_steps_per_s'] = (int(eng.read('N_STEPS').astype(np.int64).sum()) - p0) / (time.perf_counter() - t_p)
            eng.rollout_policy('line_relief', [1.0], 3)
            eng.sync()
            p0 = int(eng.read('N_STEPS').astype(np.int64).sum())
            t_p = time.perf_counter()
            eng.rollout_policy('line_relief', [1.0], args.steps)
            eng.sync()
            out['config']['closed_loop_device_policy_rollout_env_steps_per_s'] = (int(eng.read('N_STEPS').astype(np.int64).sum()) - p0) / (time.perf_counter() - t_p)
        if world == 1 and not SPLIT and exchange is None and not args.no_rollout:
            # CLOSED LOOP, EXTERNAL POLICY, NO BATCH BARRIER (round 6: ppn_send / ppn_recv, a resident step server): see async_rate
            try:
                for key, mode in (('closed_loop_async_external_policy_env_steps_per_s', 'noop'),
                                  ('closed_loop_async_external_policy_reading_every_row_env_steps_per_s', 'reads_rows')):
                    r_ = async_rate(eng, case, B, args.steps, min_ready=int(os.environ.get('PPN_BENCH_ASYNC_MIN_READY', '1024')), mode=mode,
                                    device=local_rank)
                    out['config'][key] = r_['rate']
                    out['config']['closed_loop_async_server_workgroups'] = r_['workgroups']
            except Exception as ex:
                out['config']['closed_loop_async_error'] = str(ex)[:120]
        if world == 1 and not SPLIT and not args.headline_only:
            # the same step through the host-buffer boundary (ppn_step with a host action matrix, done / flag / reward read
            # back every step): the PCIe-inclusive rate DESIGN.md quotes; never `value`
            host_actions = np.zeros((B, case.action_length), dtype=np.uint8)
            n_host = max(3, min(20, args.steps))
            eng.sync()
            t_h = time.perf_counter()
            for _ in range(n_host):
                eng.step(host_actions, auto_reset=AUTO_RESET)
                eng.read('DONE'); eng.read('FLAG'); eng.read('REWARD')
            eng.sync()
            out['config']['host_boundary_env_steps_per_s'] = B * n_host / (time.perf_counter() - t_h)
        if world == 1 and not SPLIT and exchange is None and not args.headline_only:
            # The same closed loops with the RESTART MEMO on (ppn_restart_memo, round 6; opt-in, never the headline): the fused restart of
            # an episode that ends sits behind the longest chain of a launch -- served from a snapshot it is a copy.  48 warm-up steps
            # (the engine's learning steps); every 16th step after that is a learning step as well and is inside the timed loop.
            try:
                eng.restart_memo(True)
                n_obs = eng.observation_length('full')
                obs_t = torch.empty((B, n_obs), dtype=torch.float64, device='cuda:%d' % local_rank)
                nb = obs_t.numel() * obs_t.element_size()
                for _ in range(48):
                    eng.step_observe_device(aptr, obs_t.data_ptr(), nb, auto_reset=True, layout='full', dtype=np.float64)
                eng.sync()
                c0 = int(eng.read('N_STEPS').astype(np.int64).sum())
                t_o = time.perf_counter()
                for _ in range(args.steps):
                    eng.step_observe_device(aptr, obs_t.data_ptr(), nb, auto_reset=True, layout='full', dtype=np.float64)
                eng.sync()
                out['config']['closed_loop_with_observation_restart_memo_env_steps_per_s'] = (int(eng.read('N_STEPS').astype(np.int64).sum()) - c0) / (time.perf_counter() - t_o)
                del obs_t
                if not args.no_rollout:
                    eng.rollout_policy('line_relief', [1.0], 3)
                    eng.sync()
                    p0 = int(eng.read('N_STEPS').astype(np.int64).sum())
                    t_p = time.perf_counter()
                    eng.rollout_policy('line_relief', [1.0], args.steps)
                    eng.sync()
                    out['config']['closed_loop_device_policy_rollout_restart_memo_env_steps_per_s'] = (int(eng.read('N_STEPS').astype(np.int64).sum()) - p0) / (time.perf_counter() - t_p)
                out['config']['restart_memo_served'] = eng.restart_memo_stats()['served']
            except Exception as ex:
                out['config']['restart_memo_error'] = str(ex)[:120]
            finally:
                eng.restart_memo(False)
        if world == 1 and not SPLIT and not args.no_other_configs and B == BATCH_PER_GPU and exchange is None:
            eng.close()
            long_form = other_configs(local_rank, AUTO_RESET, max(12, min(40, args.steps)))
            details['other_configs'] = long_form
            flat = {}
            for key, entry in zip(OTHER_CONFIG_KEYS, long_form):
                if 'error' in entry:
                    flat[key + '_error'] = str(entry['error'])[:100]
                    continue
                flat[key + '_Msteps'] = round(entry['env_steps_per_s'] / 1e6, 4)
                if not key.endswith('_memo'):      # (served restarts are counted as the solves they stand for: no roofline fraction for these)
                    flat[key + '_frac'] = round(entry['roofline_frac'], 4)
                flat[key + '_kernel_ms'] = round(entry['step_kernel_ms'], 4)
            try:      # topology-action search on the configs[4] engine: M simulated node-splitting candidates per second (1024 x 8)
                flat['cfg4_search_1024x8_Mcandidates'] = round(search_rate(local_rank) / 1e6, 4)
            except Exception as ex:
                flat['cfg4_search_error'] = str(ex)[:100]
            out['other_configs'] = flat
        if world == 1 and not args.no_cpu_baseline:
            try:
                out['cpu_baseline'] = cpu_baseline(case, conf, chronics, limits)
            except Exception as ex:   # the baseline must never take the bench line down
                out['cpu_baseline'] = {'value': None, 'unit': 'env-steps/s', 'cores': 0, 'kind': 'port',
                                       'sample': 'failed: %s' % ex}
        dpath = os.environ.get('PPN_BENCH_DETAILS') or (os.path.join(ROOT, 'gpurun_out', 'bench_details.json')
                                                        if os.path.isdir(os.path.join(ROOT, 'gpurun_out')) else None)
        if dpath and details:
            try:
                with open(dpath, 'w') as f:
                    json.dump(details, f, indent=1)
            except OSError:
                pass
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
