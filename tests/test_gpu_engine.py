"""GPU parity tests (-m gpu): the gfx950 build of libppn.so, called through the C ABI, against the CPU oracle on
the same seeded inputs.  Bars (BASELINE.json north_star): bit-exact line status / topology / counters / flags;
|dVm| <= 1e-6 p.u. and |dVa| <= 1e-6 rad (measured margins are ~1e-10, the tests use 1e-8 where the two sides
run the same algorithm)."""
import os

import numpy as np
import pytest

import engine_checks as ec
from test_oracle_known_answers import _basic_topology_policy

pytestmark = pytest.mark.gpu
HIP = None   # default library path: pypownet_amd/libppn.so
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


@pytest.mark.parametrize('solver', ['newton', 'fdxb'])
@pytest.mark.parametrize('env', ['default14_for_tests', 'default14_for_tests_hard_overflow'])
def test_gpu_do_nothing_matches_oracle(env, solver):
    ec.check_do_nothing(HIP, env, solver, steps=12, batch=4)


def test_gpu_dc_matches_oracle():
    ec.check_do_nothing(HIP, 'default14_for_tests_beta', 'fdxb', steps=8, batch=2)


@pytest.mark.parametrize('solver', ['newton', 'fdxb'])
def test_gpu_hard_overflow_scenario(solver):
    ec.check_hard_overflow_scenario(HIP, solver)


@pytest.mark.parametrize('solver', ['newton', 'fdxb'])
def test_gpu_k12_loss_error(solver):
    """The reference's own numeric known answer (tests/test_core.py:519-606): loss totals of three steps within 1e-3 MW."""
    ec.check_loss_error_scenario(HIP, solver)


@pytest.mark.parametrize('solver', ['newton', 'fdxb'])
def test_gpu_k3_node_splitting_all_substations(solver):
    nodes = list(range(1, 15))
    flags = ec.check_topology_scenarios(HIP, 'default14_for_tests_alpha', nodes, 7, _basic_topology_policy, solver)
    for node, f in zip(nodes, flags):
        exp = [0] * 7
        if node == 2:
            exp[6] = 1
        if node == 7:
            exp[0] = 1
        assert f == exp, (node, f)


def test_gpu_k3_dc():
    ec.check_topology_scenarios(HIP, 'default14_for_tests_beta', list(range(1, 15)), 7, _basic_topology_policy)


@pytest.mark.parametrize('solver', ['newton', 'fdxb'])
def test_gpu_default118_steps(solver):
    ec.check_do_nothing(HIP, 'default118', solver, steps=6, batch=3)


@pytest.mark.parametrize('solver', ['newton', 'fdxb'])
def test_gpu_auto_reset_and_cascade_118(solver):
    assert ec.check_auto_reset_and_cascade_118(HIP, steps=40, batch=64, solver=solver) > 0


@pytest.mark.parametrize('env,solver,steps,batch', [('default14_for_tests_alpha', 'newton', 60, 64),
                                                     ('default14_for_tests_alpha', 'fdxb', 60, 64),
                                                     ('default14_for_tests_beta', 'fdxb', 40, 32),
                                                     ('default30', 'newton', 50, 64),
                                                     ('default30', 'fdxb', 30, 32),
                                                     ('default118', 'newton', 40, 96),
                                                     ('default118', 'fdxb', 25, 64)])
def test_gpu_random_actions_vs_c_oracle(env, solver, steps, batch):
    """BASELINE.json configs[4] semantics (per-env random node splitting => dynamic Ybus rebuild) at parity size; the
    full-capacity (every busbar may be active) W=4 kernels are the ones exercised on default118."""
    st = ec.check_random_actions_vs_c_oracle(HIP, env, steps, batch, solver)
    assert st['split_buses'] > 0 and st['illegal'] > 0



def test_gpu_full_size_default118_4096_bench_workload():
    """BASELINE.json configs[2] at its full batch: 4096 environments of default118 (Newton, cascade limits, auto reset), 60 steps
    of lock-step with the C oracle (about 0.3 M solves; tests/tools/soak_parity.py runs the long version)."""
    st = ec.check_full_size_lockstep(HIP, 'default118', 4096, 60, 20, bench_limits=True, max_active_buses=118)
    assert st['done'] > 4096 and st['solves'] > 4096 * 60


@pytest.mark.parametrize('kw', [
    dict(steps=40, batch=2048),                                                                      # the bench workload (two-word Newton kernels), C oracle beside it
    dict(steps=24, batch=1024, limits_file='bench_limits_110.json'),                                 # the 110 % limit rule: restarts after restarts
    dict(steps=24, batch=768, random_acts=True, max_active_buses=0),                                 # node splitting: four-word kernels, two-capacity stepping
    dict(steps=24, batch=512, limits_file='bench_limits_110.json', solver='fdxb', hard=True),        # the reference's solver, hard game-over mode
    dict(envname='default14_for_tests_alpha', steps=80, batch=256, limits_file=None, max_active_buses=0, random_acts=True, oracle=False),
    dict(steps=24, batch=1024, limits_file='bench_limits_110.json', max_bytes=40 * 20000),           # a memo that fills up
])
def test_gpu_restart_memo(kw):
    """ppn_restart_memo (round 6): restarts of ended episodes served from snapshots -- one per chronic position, taken from the first
    computed restart -- leave every field, cumulative solve / Newton-iteration counts and epochs included, bit for bit what an
    engine that computes every restart shows, and what the C oracle (which knows no memo) counts."""
    st = ec.check_restart_memo(HIP, **kw)
    assert st['episodes_ended'] > 0 and st['snapshots'] > 0 and st['served'] > 0, st


@pytest.mark.parametrize('kw', [
    dict(steps=48, batch=2048),                                                                       # the bench workload: learning steps, then restarts served inside K_STEP_OBS
    dict(steps=40, batch=512, limits_file='bench_limits_110.json'),
    dict(steps=40, batch=512, random_acts=True, max_active_buses=0, layout='minimalist', dtype=np.float32),      # four-word kernels, two-capacity stepping
])
def test_gpu_restart_memo_under_the_fused_restart(kw):
    """The memo where the restart happens in the launch that ended the episode: ppn_step_observe(auto_reset = 1) and the closed-loop
    rollout kernel serve restarts from snapshots INSIDE the kernel (body_episode<APPLY>), learning steps are played as a deferred
    step + game-over pass + gather -- rows, every state field and the counters bit for bit those of an engine without the memo."""
    st = ec.check_restart_memo_fused(HIP, **kw)
    assert st['episodes_ended'] > 0 and st['snapshots'] > 0 and st['served'] > 0 and st['served_with_rollout'] > st['served'], st


def test_gpu_restart_memo_is_dropped_with_the_thermal_limits():
    held = ec.check_restart_memo_invalidation(HIP, batch=256, steps=16)
    assert all(h > 0 for h in held), held


def test_gpu_async_session_with_restart_memo():
    """K_SERVE with the memo: restarts of episodes that end inside the step server are served from snapshots learned before the
    session; rows, reports and final state bit for bit those of the stepped engine (which computes every restart)."""
    st = ec.check_async_equals_stepping(None, 'default118', batch=1024, n_steps=10, solver='newton', min_ready=256, max_active_buses=118,
                                        memo_warm=36)
    assert st['steps'] == 10240 and st['done'] > 0 and st['memo_served'] > 0, st


def test_gpu_full_size_bench_workload_with_restart_memo():
    """The 4096 x 60 lock-step of the headline workload against the C oracle with the memo on (PPN_RESTART_MEMO=1): flags, line
    status, counters, chronic positions, cumulative solves and Newton iterations bit-exact, voltages <= 1e-8 -- served restarts
    included (deferred restart, as bench.py steps)."""
    os.environ['PPN_RESTART_MEMO'] = '1'
    try:
        st = ec.check_full_size_lockstep(HIP, 'default118', 4096, 60, 20, bench_limits=True, max_active_buses=118, auto_reset=2)
    finally:
        os.environ.pop('PPN_RESTART_MEMO', None)
    assert st['done'] > 4096 and st['solves'] > 4096 * 60


@pytest.mark.parametrize('auto_reset', [True, 2])
def test_gpu_full_size_default118_4096_limit_rule_110(auto_reset):
    """configs[2] under the limit rule SURVEY.md 8d wrote down for it -- limit = max(50, 1.10 x I(t = 0)) -- at the
    full batch: ~98 % of the env-steps end in game over, ~8 solves per step, restart after restart (game.py:762-797 recurses), the
    64-attempt cap and the environments it leaves over (PPN_F_DEAD = 3), which the next launch takes up again BEFORE their step.
    4096 environments x 30 steps in lock-step with the C oracle, fused and deferred restart: N_STEPS, DEAD and EPOCH compared as
    well (who stepped, who still is over, how many restart attempts were made)."""
    st = ec.check_full_size_lockstep(HIP, 'default118', 4096, 30, 5, bench_limits=True, max_active_buses=118,
                                     limits_file='bench_limits_110.json', restarts=True, auto_reset=auto_reset)
    assert st['done'] > 0.6 * 4096 * 30 and st['solves'] > 8 * 4096 * 30 and st['stuck'] > 0, st


@pytest.mark.parametrize('solver', ['newton', 'fdxb'])
def test_gpu_soak_random_actions_w4_100k_solves(solver):
    """a soak of the four-word kernels (every busbar may be active; the kernels the three GPU-only incidents of
    rounds 2, 3 and 5 were seen on) inside the -m gpu run: 4096 environments x 24 steps of random node splitting / line switching in
    lock-step with the C oracle -- more than 10^5 load-flow solves per solver, the default two-capacity stepping + schedule pre-pass."""
    st = ec.check_random_actions_vs_c_oracle(HIP, 'default118', 24, 4096, solver, seed=2024, obs_every=12, excuse_vm=0.5,
                                             max_dropped=4096 // 16, count_solves=True)
    assert st['split_buses'] > 0 and st['illegal'] > 0 and st['solves'] > 100000, st


def test_gpu_persistent_step_kernel(monkeypatch):
    """The PERSISTENT form of the step kernel (K_STEP_PERSIST: as many workgroups as the GPU holds, each taking the next position
    of the launch order from a counter; the engine uses it from 4 environments per resident slot on -- 8192 environments and more
    on this workload) in lock-step with the C oracle: forced on at 4096 environments (PPN_PERSISTENT_ROUNDS=1: 1792 workgroups play
    4096 environments), Newton and fast-decoupled, and for the four-word kernels under random node splitting."""
    monkeypatch.setenv('PPN_PERSISTENT_ROUNDS', '1')
    st = ec.check_full_size_lockstep(HIP, 'default118', 4096, 30, 10, bench_limits=True, max_active_buses=118, auto_reset=2)
    assert st['done'] > 2048 and st['solves'] > 4096 * 30
    st = ec.check_full_size_lockstep(HIP, 'default118', 4096, 12, 6, solver='fdxb', bench_limits=True, max_active_buses=118)
    assert st['solves'] > 4096 * 12
    st = ec.check_random_actions_vs_c_oracle(HIP, 'default118', 15, 2048, 'newton', seed=909, obs_every=8)
    assert st['split_buses'] > 0


def test_gpu_persistent_kernel_is_the_one_running_at_8192():
    """... and unforced: at 8192 environments the engine picks the persistent form by itself (kernel time per launch is recorded for
    the step kernel whichever form ran; the lock-step is what matters)."""
    st = ec.check_full_size_lockstep(HIP, 'default118', 8192, 10, 5, bench_limits=True, max_active_buses=118, auto_reset=2)
    assert st['solves'] > 8192 * 10


def test_gpu_full_size_default118_4096_fdxb():
    """configs[2] with the solver the reference itself runs (PF_ALG = 2, grid.py:63) at the full batch."""
    st = ec.check_full_size_lockstep(HIP, 'default118', 4096, 30, 10, solver='fdxb', bench_limits=True, max_active_buses=118)
    assert st['done'] > 2048 and st['solves'] > 4096 * 30


def test_gpu_full_size_default118_4096_dc():
    """rundcpf (grid.py:227-229, loadflow_mode: DC) on IEEE-118 at the full batch, cascade limits."""
    st = ec.check_full_size_lockstep(HIP, 'default118', 4096, 30, 10, solver='fdxb', bench_limits=True, max_active_buses=118,
                                     conf={'loadflow_mode': 'DC'})
    assert st['solves'] > 4096 * 30


@pytest.mark.parametrize('env,solver,dc,n', [('default118', 'newton', False, 64), ('default118', 'fdxb', False, 64),
                                              ('default118', 'fdxb', True, 64), ('default14', 'newton', False, 48),
                                              ('default30', 'fdxb', False, 48)])
def test_gpu_runpf_arrays_solve_boundary(env, solver, dc, n):
    """the reference's seam `runpf(mpc, ppopt, '', '') -> (results, success)` (grid.py:226-229) in MATPOWER
    arrays with '666'-twin ids -- random states (split nodes, lines out, productions off) vs oracle/pf_np.runpf on the same
    arrays, <= 1e-8."""
    seen = ec.check_runpf_arrays(HIP, env, n, solver=solver, dc=dc)
    assert seen['ok'] >= n // 2 and seen['split'] and seen['lines_out'] and seen['prods_off'], seen


@pytest.mark.parametrize('solver', ['newton', 'fdxb'])
def test_gpu_soft_overflow_scenario_k2(solver):
    ec.check_soft_overflow_scenario(HIP, solver)


def test_gpu_restart_goes_on_after_the_attempt_cap():
    assert ec.check_restart_goes_on(HIP, batch=256, steps=12) >= 1


def test_gpu_rollout_with_environments_over_at_the_start():
    assert ec.check_rollout_dead_at_start(HIP, batch=256, n_steps=6) >= 1


def test_gpu_rollout_equals_steps():
    """ppn_rollout: n steps per environment in one launch = n ppn_step calls, bit for bit."""
    assert ec.check_rollout_equals_steps(HIP, 'default118', batch=512, n_steps=12) > 100
    assert ec.check_rollout_equals_steps(HIP, 'default14_for_tests_alpha', batch=64, n_steps=20, bench_limits=False, random_acts=True) > 0
    assert ec.check_rollout_equals_steps(HIP, 'default118', batch=64, n_steps=8, bench_limits=False, random_acts=True, modes=(1, 2)) >= 0


def test_gpu_full_size_default14_1024_newton():
    """BASELINE.json configs[1]: default14 (its own chronics: every chronic the fixture carries, start row (37 e) mod T),
    AC Newton-Raphson, 1024 environments, 200 steps against the C oracle."""
    st = ec.check_full_size_lockstep(HIP, 'default14', 1024, 200, 40)
    assert st['slots'] >= 2 and st['solves'] >= 1024 * 200


@pytest.mark.parametrize('solver', ['newton', 'fdxb'])
def test_gpu_hard_game_over_mode(solver):
    """game_over_mode='hard' against the numpy restatement (default14, random actions, chronic position after every restart)."""
    assert ec.check_hard_game_over_mode(HIP, solver=solver, steps=60, batch=8) >= 2


def test_gpu_hard_game_over_mode_118_auto_reset():
    """... and against the C oracle on default118 with the fused restart (auto_reset), bench limits, 512 environments."""
    st = ec.check_full_size_lockstep(HIP, 'default118', 512, 40, 10, bench_limits=True, max_active_buses=118, game_over_mode='hard')
    assert st['done'] > 100 and st['slots'] >= 2


def test_gpu_random_chronic_looping():
    ec.check_random_chronic_looping(HIP, batch=96, steps=40)


def test_gpu_deferred_restart_equals_fused():
    """ppn_step(auto_reset = 2), the mode bench.py steps in, against the fused restart and the C oracle."""
    assert ec.check_deferred_restart(HIP, steps=40, batch=256) > 100
    assert ec.check_deferred_restart(HIP, 'default14_for_tests_alpha', steps=60, batch=64, bench_limits=False, max_active_buses=0,
                                     random_acts=True) > 0


def test_gpu_config1_default14_dc_1000_steps():
    """BASELINE.json configs[0] through the HIP engine: default14 in DC mode, do-nothing, 1000 timesteps across the end of
    the first chronic, against the numpy restatement step by step."""
    ec.check_config1_default14_dc(HIP)


def test_gpu_launch_order_does_not_change_results(monkeypatch):
    """The loading-ordered launch (ppn_order_kernel, batches above the 1024 resident slots) only changes WHEN an
    environment runs: every field must be bit-identical with the ordering switched off (PPN_LAUNCH_ORDER=0)."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..'))
    import bench
    from pypownet_amd.engine import Engine
    case, conf, chronics = bench.load_workload()
    B = 2048
    slots, t0 = bench.env_assignment(0, B, chronics)
    act = np.zeros((B, case.action_length), dtype=np.uint8)
    states = []
    for order in ('1', '0'):
        monkeypatch.setenv('PPN_LAUNCH_ORDER', order)
        eng = Engine(case, conf, B, chronics=chronics, thermal_limits=bench.bench_limits(case), max_active_buses=case.nS)
        eng.reset(chronic_slot=slots, t0=t0)
        for _ in range(8):
            eng.step(act, auto_reset=True)
        states.append({f: eng.read(f).copy() for f in ('VM', 'VA', 'AMPS', 'PG', 'QG', 'LINES_STATUS', 'DONE', 'FLAG',
                                                       'SOFT_COUNT', 'RECONNECTABLE', 'CHRONIC_ROW', 'N_SOLVES', 'N_ITERS')})
        del eng
    for f in states[0]:
        assert np.array_equal(states[0][f], states[1][f], equal_nan=True), f
    assert int(states[0]['N_SOLVES'].sum()) > 8 * B      # cascades did happen


@pytest.mark.parametrize('env,steps,batch', [('default14_for_tests_alpha', 60, 64), ('default118', 12, 64)])
def test_gpu_device_reward_matches_restatement(env, steps, batch):
    ec.check_device_reward(HIP, env, steps, batch)


@pytest.mark.parametrize('env,batch,k', [('default14_for_tests_alpha', 48, 8), ('default118', 32, 6),
                                         ('default118', 192, 8),       # (1536 candidates: more than 1024, handed out by their parents' loading -- round 6)
                                         ('default118', 640, 8)])      # (5120: four per resident slot and more -- the persistent form of the step kernel plays them)
def test_gpu_candidate_search_equals_simulate(env, batch, k):
    ec.check_candidate_search(HIP, env, batch, k)


def test_gpu_reduced_observation_layouts():
    """Reduced / float32 observation layouts gathered on the GPU against the numpy oracle's field-by-field arrays."""
    ec.check_reduced_observation_layouts(HIP, 'default118')
    ec.check_reduced_observation_layouts(HIP, 'default14_for_tests', steps=6, batch=2)


@pytest.mark.parametrize('cap,solver', [(150, 'newton'), (128, 'newton'), (150, 'fdxb')])
def test_gpu_intermediate_busbar_capacities(cap, solver):
    """max_active_buses between the substation count and every busbar: the 4-word kernels below full capacity (150) and the
    2-word kernels with spare busbars (128), against the oracle under random node splitting; no environment may hit the capacity flag."""
    st = ec.check_random_actions_vs_c_oracle(HIP, 'default118', 20, 48, solver, seed=77, max_active_buses=cap)
    assert st['split_buses'] > 0


def test_gpu_tuned_pattern_capacity():
    """rules.lu_capacity = 3976 (pattern capacity 1.5 x the base pattern, every busbar may be active): the working set of the
    4-word kernels drops to 32 LDS granules -- four environments per CU, the configuration of bench.py's configs[4] lines -- and
    the results stay those of the oracle under random node splitting; no environment may hit the capacity flag."""
    st = ec.check_random_actions_vs_c_oracle(HIP, 'default118', 20, 48, 'newton', seed=78, lu_capacity=3976)
    assert st['split_buses'] > 0


def test_gpu_random_splitting_full_share_of_configs4_tuned():
    """BASELINE configs[4] in the configuration the bench line quotes -- default118, per-env random
    node splitting, the per-GPU share of 1024 environments, every busbar may be active (W = 4 kernels), lu_capacity = 3976,
    q_plane_auto = 1 -- 40 steps of lock-step with the C oracle, zero capacity flags (asserted inside the check)."""
    st = ec.check_random_actions_vs_c_oracle(HIP, 'default118', 40, 1024, 'newton', seed=404, lu_capacity=3976, q_plane_auto=1,
                                             obs_every=8)
    assert st['split_buses'] > 0 and st['illegal'] > 0 and st['done'] > 1024


def test_gpu_random_splitting_full_share_fdxb_w4():
    """the four-word FAST-DECOUPLED kernels (256 VGPRs + AGPRs; one code shape there gave wrong flows on the
    GPU only) in lock-step with the oracle under random node splits at the full per-GPU share."""
    st = ec.check_random_actions_vs_c_oracle(HIP, 'default118', 25, 1024, 'fdxb', seed=405, obs_every=8)
    assert st['split_buses'] > 0 and st['done'] > 512


@pytest.mark.parametrize('solver', ['newton', 'fdxb'])
def test_gpu_repacked_schedule(solver):
    """Pivots and Schur rounds of the shared schedule re-packed on the host vs the schedule as built, and the oracle (IEEE-118,
    cascade limits)."""
    assert ec.check_repacked_schedule(HIP, steps=20, batch=64, solver=solver) > 0


@pytest.mark.parametrize('tail_buses', [6, 8])
def test_gpu_dense_tail_unit(tail_buses):
    """The register-resident dense tail on its own (tools/ubench/dense_tail_test.hip): the Gauss-Jordan sweep (tail_gj_steps) of the kernel
    sources on random systems of 2 .. 2 * tail_buses rows with identity rows, against Gaussian elimination on the host.  Holds the
    DPP hazard the engine tests cannot see at the shipped tail size (a copy the register allocator may place in front of a DPP
    read): found with a 16-row tail."""
    import subprocess
    exe = os.path.join(ROOT, 'build', 'dense_tail_test_m%d' % (2 * tail_buses))
    src = os.path.join(ROOT, 'tools', 'ubench', 'dense_tail_test.hip')
    deps = [src] + [os.path.join(ROOT, 'pypownet_amd', 'csrc', f) for f in ('ppn_device.h', 'ppn_solve.inc')]
    if not os.path.exists(exe) or any(os.path.getmtime(d) > os.path.getmtime(exe) for d in deps):
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-w', '-DPPN_TAIL_BUSES=%d' % tail_buses,
                               src, '-o', exe])
    out = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    assert out.returncode == 0 and 'FAILED' not in out.stdout, out.stdout


class _DLPackOnly(object):
    """A device array that is not a torch tensor: only the DLPack protocol (what cupy / jax arrays offer)."""

    def __init__(self, t):
        self._t = t

    def __dlpack__(self, stream=None):
        return self._t.__dlpack__()

    def __dlpack_device__(self):
        return self._t.__dlpack_device__()


def test_gpu_batched_tensor_api():
    """BatchedRunEnv.step / search take device tensors (torch CUDA, or anything with __dlpack__) and answer with
    device tensors -- no PCIe crossing; same numbers as the host-array path."""
    import os
    import torch
    from helpers import ENVS
    from pypownet_amd.batched import BatchedRunEnv
    envdir = os.path.join(ENVS, 'default14')
    B = 64
    host = BatchedRunEnv(envdir, 'level0', B, device=0, config_overrides={'solver': 'newton'})
    dev = BatchedRunEnv(envdir, 'level0', B, device=0, config_overrides={'solver': 'newton'})
    host.reset()
    dev.reset()
    rng = np.random.default_rng(0)
    case = host.case
    case.ntopo_offset_lines = case.nP + case.nL + 2 * case.nl
    for t in range(12):
        a = ec.random_actions(case, rng, B)
        ta = torch.from_numpy(a).cuda()
        o1, d1, f1, i1 = host.step(a, auto_reset=True)
        o2, d2, f2, i2 = dev.step(ta if t % 2 else _DLPackOnly(ta), auto_reset=True)
        assert o2.is_cuda and d2.is_cuda and o2.dtype == torch.float64
        assert np.array_equal(o1, o2.cpu().numpy(), equal_nan=True)
        assert np.array_equal(d1, d2.cpu().numpy()) and np.array_equal(f1, f2.cpu().numpy()) and np.array_equal(i1, i2.cpu().numpy())
        o3 = dev.step(torch.zeros_like(ta), auto_reset=True, layout='minimalist', obs_dtype=np.float32)[0]
        host.step(np.zeros_like(a), auto_reset=True)
        assert o3.dtype == torch.float32 and o3.shape[1] == dev.engine.observation_length('minimalist')
    K = 5
    cand = np.stack([ec.random_actions(case, rng, B) for _ in range(K)], axis=1)
    r1, d1, f1, o1 = host.search(cand, want_obs=True)
    r2, d2, f2, o2 = dev.search(torch.from_numpy(cand).cuda(), want_obs=True)
    assert r2.is_cuda and np.array_equal(f1, f2.cpu().numpy()) and np.array_equal(d1, d2.cpu().numpy())
    np.testing.assert_allclose(r1, r2.cpu().numpy(), rtol=1e-13, atol=1e-13)      # (sums of the 5 components in another order)
    assert np.array_equal(o1, o2.cpu().numpy(), equal_nan=True)
    # open-loop rollout: a recorded action sequence from host memory and from a device tensor, against the stepped path
    seq = np.stack([ec.random_actions(case, rng, B) for _ in range(6)])
    ret0 = host.engine.read('RETURN').copy()
    for k in range(6):
        host.step(seq[k], auto_reset=True, want_obs=False)
    r_h = host.engine.read('RETURN') - ret0
    ret0 = dev.engine.read('RETURN').copy()
    ret, done, flag, nst = dev.rollout(torch.from_numpy(seq).cuda(), auto_reset=True)
    np.testing.assert_allclose(ret - ret0, r_h, rtol=1e-12, atol=1e-9)
    assert np.array_equal(done, host.engine.read('DONE').astype(bool)) and np.array_equal(flag, host.engine.read('FLAG'))
    assert np.array_equal(dev.engine.read('VM'), host.engine.read('VM'), equal_nan=True)
    with pytest.raises(ValueError):
        dev.step(torch.zeros((B, case.action_length), dtype=torch.uint8, device='cuda:0'), obs_dtype='int32')


def test_gpu_bench_spawns_its_ranks_when_launched_bare():
    """`python bench.py --gpus 2` without a launcher (how a driver may call it): bench.py spawns the two ranks itself and the line
    says n_gpus = 2; asking for more ranks than GPUs under RCCL is a non-zero exit, not a one-rank run."""
    import json
    import subprocess
    import sys
    B = 128
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--batch', str(B),
           '--no-cpu-baseline']
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    out = subprocess.run(cmd, env=dict(env, PPN_BENCH_BACKEND='gloo'), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert d['n_gpus'] == 2 and d['config']['env_steps_executed'] == 2 * B * 3
    import torch
    if torch.cuda.device_count() < 2:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)      # RCCL: one GPU per rank
        assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith('{')]


@pytest.mark.parametrize('single_controller,workload', [(False, 'cascade'), (True, 'cascade'), (False, 'split')])
def test_gpu_bench_two_ranks_share_one_gpu(single_controller, workload, tmp_path):
    """The N > 1 path of bench.py end to end: two ranks (PPN_BENCH_BACKEND=gloo lets them share this box's one GPU; on the
    8-GPU node the driver uses RCCL), launched the way the driver launches it.  Rank r must play environments [r B, (r+1) B) of
    the global assignment, the line must carry the aggregate of both ranks.  workload = 'split' (VERDICT r05 #5): BASELINE configs[4]
    -- per-environment random node splitting -- under --gpus 2: every rank's action matrices must be ITS SLICE of the matrices a
    single process draws for all 2 B environments (philox(1234, global env, step))."""
    import json
    import os
    import subprocess
    import sys
    import zlib
    from helpers import ROOT
    sys.path.insert(0, ROOT)
    import bench
    B = 256 if workload == 'cascade' else 96
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4', '--warmup', '2', '--batch', str(B),
           '--no-cpu-baseline', '--workload', workload] + (['--single-controller'] if single_controller else [])
    env = dict(os.environ, PPN_BENCH_BACKEND='gloo')
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
    d = json.loads(line)
    assert d['n_gpus'] == 2 and d['steps'] == 4 and d['value'] > 0 and d['scaling'] == 'weak'
    assert d['config']['workload_name'] == workload
    case, conf, chronics = bench.load_workload()
    want = []
    single = [bench.philox_node_splitting(case, np.arange(2 * B), k) for k in range(bench.SPLIT_ACTION_MATRICES)] if workload == 'split' else []
    for r in range(2):
        slots, t0 = bench.env_assignment(r * B, B, chronics)
        crc = zlib.crc32(slots.tobytes() + t0.tobytes())
        for m in single:
            crc = zlib.crc32(np.ascontiguousarray(m[r * B:(r + 1) * B]).tobytes(), crc)
        want.append(crc)
    assert d['config']['env_assignment_crc32'] == want
    if workload == 'split':
        assert single[0].any()
        assert 'W=4' in d['roofline']['kernel'] and d['config']['env_steps_executed'] == 2 * B * 4
    # the aggregate: both ranks' environments over the slowest rank's time
    assert abs(d['value'] - d['config']['env_steps_executed'] / (d['ms_per_step'] * 4 / 1e3)) < 1e-6 * d['value']
    if not single_controller:
        assert d['value_k60'] and d['value_k60'] > 0


def test_gpu_rccl_path_on_one_rank(tmp_path):
    """the RCCL ("nccl" backend) legs of the multi-GPU path on the hardware that exists -- ONE rank.
    (a) BatchedRunEnv.controller_step with device tensors on a world of one nccl rank: device-resident scatter -> Engine.step_device
        -> gather of device tensors, compared step by step with the host-array path on a second engine;
    (b) bench.py --gpus 1 --single-controller with PPN_BENCH_FORCE_DIST=1: the bench's own scatter / step / gather loop over RCCL;
        its line is kept as profiles/r04_bench_single_controller_1rank.json by tools/collect_profiles.sh."""
    import json
    import subprocess
    import sys
    port = 29500 + (os.getpid() % 2000)
    code = (
        "import os, sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import torch, torch.distributed as dist\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='%d', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1)\n"
        "from pypownet_amd.batched import BatchedRunEnv\n"
        "from helpers import ENVS\n"
        "B = 192\n"
        "kw = dict(config_overrides={'solver': 'newton'})\n"
        "a = BatchedRunEnv(os.path.join(ENVS, 'default118'), 'level0', B, rank=0, world_size=1, device=0, device_exchange=True, **kw)\n"
        "b = BatchedRunEnv(os.path.join(ENVS, 'default118'), 'level0', B, rank=0, world_size=1, device=0, **kw)\n"
        "a.reset(); b.reset()\n"
        "rng = np.random.default_rng(3)\n"
        "n_done = 0\n"
        "for t in range(12):\n"
        "    act = (rng.random((B, a.action_length)) < 0.004).astype(np.uint8)\n"
        "    done, flag, rew = a.controller_step(torch.from_numpy(act).to('cuda:0'), root=0, auto_reset=True)\n"
        "    assert done.is_cuda and flag.is_cuda and rew.is_cuda\n"
        "    b.engine.step(act, auto_reset=True)\n"
        "    assert np.array_equal(done.cpu().numpy(), b.engine.read('DONE').astype(bool))\n"
        "    assert np.array_equal(flag.cpu().numpy(), b.engine.read('FLAG'))\n"
        "    np.testing.assert_allclose(rew.cpu().numpy(), b.engine.read('REWARD').sum(axis=1), rtol=1e-12, atol=1e-9)\n"
        "    for f in ('VM', 'LINES_STATUS', 'PRODS_NODES', 'N_SOLVES'):\n"
        "        assert np.array_equal(a.engine.read(f), b.engine.read(f), equal_nan=True), f\n"
        "    n_done += int(done.sum())\n"
        "s = a.all_reduce_stats([1.0, float(n_done)])\n"
        "assert s[0] == 1.0\n"
        "dist.barrier(); dist.destroy_process_group()\n"
        "print('RCCL_ONE_RANK_OK', n_done)\n"
    ) % (ROOT, os.path.join(ROOT, 'tests'), port)
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and 'RCCL_ONE_RANK_OK' in out.stdout, out.stderr[-3000:]
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '10', '--warmup', '3', '--single-controller',
           '--no-cpu-baseline', '--no-other-configs']
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'PPN_BENCH_BACKEND')}
    env.update(PPN_BENCH_FORCE_DIST='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port + 1))
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert d['n_gpus'] == 1 and d['value'] > 0 and d['config'].get('single_controller') and d['config'].get('dist_backend') == 'nccl', d['config']
    keep = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(keep):
        with open(os.path.join(keep, 'r05_bench_single_controller_1rank.json'), 'w') as f:
            json.dump(d, f, indent=1)


def test_gpu_engine_library_first_then_torch():
    """One HIP runtime per process: loading libppn.so BEFORE torch is imported (pypownet_amd/_lib.py preloads the runtime torch
    bundles, not the framework) must leave torch able to see the GPU, and both must work side by side."""
    import subprocess
    import sys
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from pypownet_amd import _lib\n"
        "_lib.load_library()\n"
        "assert 'torch' not in sys.modules\n"
        "from helpers import load_env\n"
        "from pypownet_amd.engine import Engine\n"
        "case, cfg, chronics = load_env('default14_for_tests')\n"
        "eng = Engine(case, cfg, 4, chronics=chronics)\n"
        "eng.reset(); eng.step(np.zeros((4, case.action_length), dtype=np.uint8))\n"
        "import torch\n"
        "assert torch.cuda.is_available()\n"
        "t = torch.ones(8, device='cuda') * 2\n"
        "assert float(t.sum()) == 16.0 and not eng.read('DONE').any()\n"
        "print('ok')\n") % (ROOT, os.path.join(ROOT, 'tests'))
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and 'ok' in out.stdout, out.stderr[-2000:]


@pytest.mark.parametrize('nb', [118, 0])
def test_gpu_k1_style_rows_on_ieee118(nb):
    """tools/make_k1_rows.py: the reference's RunEnv on default118, do-nothing, int-truncated amperes of every line, 60 steps --
    two-word (nb = 118) and four-word (every busbar may be active) fast-decoupled kernels."""
    assert ec.check_k1_rows_118(HIP, max_active_buses=nb) >= 1


@pytest.mark.parametrize('solver,auto_reset,threads', [('newton', True, 256), ('newton', 2, 64), ('fdxb', 2, 256), ('fdxb', True, 64)])
def test_gpu_schedule_prepass_builds_the_same_tables(solver, auto_reset, threads):
    """Round 5: the schedule pre-pass (a four-wave workgroup per environment in front of the step kernel of the four-word
    engines) vs the build by the environment's own wavefront inside its solve (PPN_SCHED_PREPASS=0): states, reports and the
    schedule caches bit for bit the same over random node-splitting steps, and no environment builds a schedule inside its solve
    once the pre-pass runs."""
    st = ec.check_schedule_prepass(HIP, steps=30, batch=96, solver=solver, auto_reset=auto_reset, double_acts=True, threads=threads)
    assert st['illegal'] > 0 and st['split'] > 0 and st['done'] > 0, st


def test_gpu_step_report_field_mirrors_done_flag_reward():
    """PPN_F_STEP_REPORT (libppn 0.2): one [3] row per environment = (done, flag, sum of the reward components) of the last step."""
    from helpers import load_env
    from harness import engine_with_library
    case, cfg, chronics = load_env('default118', conf={'solver': 'newton'})
    case.ntopo_offset_lines = case.nP + case.nL + 2 * case.nl
    eng = engine_with_library(HIP, case, cfg, 64, chronics=chronics)
    eng.reset()
    rng = np.random.default_rng(9)
    seen = 0
    for t in range(25):
        eng.step(ec.random_actions(case, rng, 64), auto_reset=2 if t % 2 else True)
        rep = eng.read('STEP_REPORT')
        assert np.array_equal(rep[:, 0] != 0, eng.read('DONE').astype(bool))
        assert np.array_equal(rep[:, 1].astype(np.int32), eng.read('FLAG'))
        np.testing.assert_allclose(rep[:, 2], eng.read('REWARD').sum(axis=1), rtol=1e-13, atol=1e-10)
        seen += int(eng.read('DONE').sum())
    assert seen > 0
    eng.close()


@pytest.mark.parametrize('env,batch,steps,thr,kw', [('default118', 2048, 12, 0.9, dict(max_active_buses=118)),
                                                   ('default118', 4096, 24, 0.85, dict(max_active_buses=118)),      # the headline's size: 1 792 workgroups on 8 XCDs hand 4096 environments around (XCD-affine since round 6)
                                                   ('default118', 96, 10, 0.8, dict()),
                                                   ('default14', 512, 30, 0.5, dict())])
def test_gpu_policy_rollout_equals_stepping(env, batch, steps, thr, kw):
    """Round 5: closed-loop steps of a device-side policy with every environment on its own clock
    (ppn_rollout_policy: work items (step, environment) handed to whichever workgroup is free) -- trajectories bit for bit those of
    synchronous stepping with the same policy."""
    assert ec.check_policy_rollout_equals_stepping(HIP, env, batch=batch, n_steps=steps, params=(thr,), bench_limits=(env == 'default118'), **kw) > 0


@pytest.mark.parametrize('small_ecap,auto_reset,batch', [(0, 2, 1024), (664, True, 256), (656, 2, 2048)])
def test_gpu_two_capacity_stepping(small_ecap, auto_reset, batch):
    """Round 5: the default capacity of the four-word engines -- a small-storage launch (four environments per CU) for the environments
    whose schedule fits, a large-storage launch for the rest -- gives the states one large-storage launch gives, bit for bit; with a
    forced tiny small storage a good share of the environments goes through the second launch."""
    st = ec.check_two_capacity_stepping(HIP, steps=20, batch=batch, small_ecap=small_ecap, auto_reset=auto_reset)
    assert st['small'] > 0 and (st['big'] > 0 or not small_ecap), st


def test_gpu_batched_policy_api():
    """BatchedRunEnv.policy_actions / rollout_policy: the device policy stepped through the tensor API equals its one-launch rollout."""
    import torch
    from pypownet_amd.batched import BatchedRunEnv
    from helpers import ENVS
    envdir = os.path.join(ENVS, 'default14')
    a = BatchedRunEnv(envdir, 'level0', 96, device=0, config_overrides={'solver': 'newton'})
    b = BatchedRunEnv(envdir, 'level0', 96, device=0, config_overrides={'solver': 'newton'})
    for e in (a, b):
        e.reset()
        e.engine.process_game_over()
    for _ in range(25):
        acts = a.policy_actions('line_relief', (0.6,))
        assert acts.is_cuda and tuple(acts.shape) == (96, a.action_length) and int(acts.sum(dim=1).max()) <= 1
        a.step(acts, auto_reset=1, want_obs=False)
    ret, done, flag, nst = b.rollout_policy(25, 'line_relief', (0.6,))
    assert np.array_equal(nst, a.engine.read('N_STEPS')) and int(nst.sum()) == 96 * 25
    assert np.array_equal(ret, a.engine.read('RETURN')) and np.array_equal(flag, a.engine.read('FLAG'))
    for f in ('VM', 'LINES_STATUS', 'RECONNECTABLE', 'CHRONIC_ROW'):
        assert np.array_equal(a.engine.read(f), b.engine.read(f), equal_nan=True), f


def test_gpu_two_capacity_stepping_fast_decoupled():
    st = ec.check_two_capacity_stepping(HIP, steps=20, batch=512, small_ecap=660, auto_reset=2, solver='fdxb')
    assert st['small'] > 0 and st['big'] > 0, st
    st = ec.check_two_capacity_stepping(HIP, steps=12, batch=1024, small_ecap=0, auto_reset=2, solver='fdxb')
    assert st['small'] > 0, st


@pytest.mark.parametrize('envname,solver,layout,dtype,auto_reset,batch', [
    ('default14', 'newton', 'full', 'float64', True, 64), ('default14', 'fdxb', 'minimalist', 'float32', True, 64),
    ('default14_for_tests_alpha', 'newton', 'ac_minimalist', 'float64', False, 16), ('default30', 'dc', 'full', 'float32', True, 32),
    ('default118', 'newton', 'full', 'float64', True, 48)])
def test_gpu_step_observe_equals_step_then_read(envname, solver, layout, dtype, auto_reset, batch):
    """ppn_step_observe: the observation rows written by the step kernel's workgroups = ppn_step + ppn_read_observation, bit for bit."""
    import numpy as np
    ec.check_step_observe(HIP, envname, batch, 14, solver, layout, np.dtype(dtype), auto_reset=auto_reset)


def test_gpu_step_observe_bench_workload_2048():
    """... and on the bench workload (two-word kernels, launch order on: 2048 environments, cascade limits, do-nothing + switches)."""
    import json
    import os
    import numpy as np
    from helpers import ENVS
    with open(os.path.join(ENVS, 'default118', 'bench_limits.json')) as f:
        lim = np.asarray(json.load(f)['limits_a'])
    from helpers import load_env
    case, _, _ = load_env('default118')
    ec.check_step_observe(HIP, 'default118', 2048, 6, 'newton', 'full', np.float64, thermal_limits=lim, max_active_buses=case.nS)


@pytest.mark.parametrize('envname,batch,steps,solver,layout,dtype,min_ready,kw', [
    ('default14', 32, 30, 'newton', 'full', 'float64', 8, dict(bench_limits=False)),
    ('default118', 2048, 10, 'newton', 'full', 'float64', 512, dict(max_active_buses=118, settle_at=5)),
    ('default118', 512, 12, 'fdxb', 'minimalist', 'float32', 1, dict(max_active_buses=118, rows_by_env=True)),
    ('default118', 256, 10, 'newton', 'ac_minimalist', 'float64', 64, dict(device_actions=True)),       # four-word kernels
    ('default118', 4096, 6, 'newton', 'full', 'float64', 1024, dict(max_active_buses=118, device_actions=True))])
def test_gpu_async_send_recv_equals_stepping(envname, batch, steps, solver, layout, dtype, min_ready, kw):
    """the asynchronous batch boundary for EXTERNAL policies (ppn_async_start / ppn_send / ppn_recv over the C ABI:
    a resident step server, a completion ring in pinned memory) -- per environment the same trajectory, observation rows and report
    rows, bit for bit, as ppn_step(auto_reset = 1) with the same actions; host and device action rows, both row conventions, a call of
    another entry point in the middle of the session (it settles the session), one- to four-word kernels, up to the full 4096."""
    st = ec.check_async_equals_stepping(None, envname, batch=batch, n_steps=steps, solver=solver, layout=layout, dtype=np.dtype(dtype),
                                        min_ready=min_ready, **kw)
    assert st['steps'] == batch * steps and st['done'] > 0, st      # (st['restarts']: a host loop in Python may well be slower than the server's idle timeout)
    if batch >= 2048:
        assert st['apart'] >= 0 and st['receives'] > steps, st      # (more receives than synchronous steps: nobody waited for the batch)


def test_gpu_async_server_leaves_on_idle_timeout_and_is_restarted():
    """Liveness: the resident server must not outlive a host that went away -- it leaves when nothing has been published for its idle
    timeout -- and a host that was merely slow loses nothing: the next call finds the server gone, re-publishes what it had not started
    and launches it again.  Here: a 150 ms timeout and a host that sleeps 0.6 s in the middle of the session; same trajectories."""
    st = ec.check_async_equals_stepping(None, 'default118', batch=256, n_steps=8, solver='newton', min_ready=64, max_active_buses=118,
                                        idle_timeout_ms=150, pause_s=0.6)
    assert st['steps'] == 256 * 8 and st['restarts'] >= 1, st


def test_gpu_batched_send_recv_with_a_torch_policy():
    """BatchedRunEnv.send / recv: torch tensors in, torch tensors out, the policy's work ordered on the session's stream -- a closed
    loop in which every environment is stepped again as soon as a (torch) policy has looked at its observation row."""
    import torch
    from helpers import ENVS
    from pypownet_amd.batched import BatchedRunEnv
    B, K = 1024, 6
    env = BatchedRunEnv(os.path.join(ENVS, 'default118'), 'level0', B, device=0, config_overrides={'solver': 'newton'}, max_active_buses=118)
    ref = BatchedRunEnv(os.path.join(ENVS, 'default118'), 'level0', B, device=0, config_overrides={'solver': 'newton'}, max_active_buses=118)
    env.reset(); ref.reset()
    env.engine.process_game_over(); ref.engine.process_game_over()
    for _ in range(K):
        ref.engine.step(np.zeros((B, env.action_length), dtype=np.uint8), auto_reset=True)
    env.async_start(layout='minimalist', obs_dtype=torch.float32)
    stream = env.async_stream()
    with torch.cuda.stream(stream):
        acts = torch.zeros((B, env.action_length), dtype=torch.uint8, device='cuda:0')
        env.send(torch.arange(B, dtype=torch.int32), acts, rows_by_env=True)
        steps = torch.zeros(B, dtype=torch.int64)
        total, seen_nan = 0, 0
        while total < B * K:
            ids, obs, rep = env.recv(min_ready=128)
            assert obs.is_cuda and rep.is_cuda and obs.shape[0] == len(ids) and obs.dtype == torch.float32
            seen_nan += int(torch.isnan(obs).any())
            ids_c = ids.cpu().long()
            steps[ids_c] += 1
            total += len(ids_c)
            again = ids_c[steps[ids_c] < K]
            if len(again):
                env.send(again, acts, rows_by_env=True)      # the "policy": do nothing, in place
    env.async_stop()
    assert int(steps.min()) == K and int(steps.max()) == K
    for f in ('VM', 'LINES_STATUS', 'N_STEPS', 'N_SOLVES', 'CHRONIC_ROW', 'RETURN'):
        assert np.array_equal(env.engine.read(f), ref.engine.read(f), equal_nan=True), f


@pytest.mark.parametrize('solver,batch,k', [('newton', 128, 8), ('fdxb', 64, 4)])
def test_gpu_candidate_slots_keep_their_schedules(solver, batch, k):
    """Round 6: ppn_simulate_candidates' slots keep their schedules across calls -- same outcomes, bit for bit, as
    refilling every slot from its environment at every fork (PPN_CAND_CACHE=0)."""
    assert ec.check_candidate_schedule_cache(None, batch=batch, k=k, rounds=8, solver=solver) > 0


@pytest.mark.gpu
def test_gpu_wave_full_remedy_holds():
    """the compiler hazard behind round 5's GPU-only failure -- simplifycfg threads a work loop's back edge into the block of
    a convergent readfirstlane; lanes 1..63 then replay item 0 without lane 0 -- in twenty lines (tools/ubench/
    convergent_threading_repro.hip, built by __graft_entry__.build_guards).  With the statement the kernels carry at their loop heads
    (PPN_WAVE_FULL, FIX=3) every (item, lane) cell is played exactly once.  The unprotected form is run too and only reported: with
    this compiler it is wrong; a compiler that gets it right would not make the remedy wrong."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fixed = os.path.join(root, 'build', 'convergent_threading_repro_fix3')
    plain = os.path.join(root, 'build', 'convergent_threading_repro_fix0')
    if not os.path.exists(fixed):      # (build/ normally travels with the tree; the box has the same hipcc)
        import sys
        sys.path.insert(0, root)
        import __graft_entry__ as ge
        ge.build_guards()
    r = subprocess.run([fixed], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and '64000 of 64000' in r.stdout and '-> OK' in r.stdout, r.stdout + r.stderr
    if os.path.exists(plain):
        r0 = subprocess.run([plain], capture_output=True, text=True, timeout=60)
        print(r0.stdout.strip())


@pytest.mark.gpu
@pytest.mark.parametrize('solver,max_active_buses', [('newton', 118), ('fdxb', 118), ('newton', 0), ('fdxb', 0)])
def test_gpu_results_do_not_depend_on_stale_registers(solver, max_active_buses):
    """the differential register-poison check -- every VGPR, AGPR and LDS byte of the chip filled with 0 for one engine and with
    0x7ff7a5a5 for the other in front of every call; 1024 environments x 10 steps of random node splitting, two- and four-word kernels, both solvers.  On round 2's tree it tells the failing build from the
    passing ones every time (26 fields differ / none: profiles/r06_incident_i_register_poison.txt)."""
    st = ec.check_register_poison(HIP, 'default118', solver, batch=1024, steps=10, max_active_buses=max_active_buses)
    assert st['solves'] > 1024 * 10, st
