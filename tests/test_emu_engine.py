"""CPU-side logic tests of the HIP kernels.  The kernel source (pypownet_amd/csrc/*.inc) is compiled here
lane-serially with g++ (-DPPN_EMU) into build/libppn_emu.so -- a TEST-ONLY artefact that the package never loads --
and driven through the very same C ABI and Python wrapper as the GPU library, then compared with the oracle.
This is what lets host logic, the ABI plumbing and the game/solve control flow be checked without a GPU; the
parity tests proper (tests/test_gpu_*.py, -m gpu) run the real gfx950 build."""
import os
import subprocess

import numpy as np
import pytest

import engine_checks as ec
from helpers import ROOT
from test_oracle_known_answers import _basic_topology_policy

EMU = os.path.join(ROOT, 'build', 'libppn_emu.so')
SRC = os.path.join(ROOT, 'pypownet_amd', 'csrc')


@pytest.fixture(scope='session')
def emu_lib():
    srcs = [os.path.join(SRC, f) for f in os.listdir(SRC)] + [os.path.join(ROOT, 'include', 'ppn.h')]
    if not os.path.exists(EMU) or any(os.path.getmtime(s) > os.path.getmtime(EMU) for s in srcs):
        os.makedirs(os.path.dirname(EMU), exist_ok=True)
        subprocess.check_call(['g++', '-O2', '-std=c++17', '-DPPN_EMU', '-fPIC', '-shared', '-x', 'c++',
                               os.path.join(SRC, 'ppn_engine.hip'), '-o', EMU])
    return EMU


@pytest.mark.parametrize('solver', ['newton', 'fdxb'])
@pytest.mark.parametrize('env', ['default14_for_tests', 'default14_for_tests_hard_overflow'])
def test_emu_do_nothing_matches_oracle(emu_lib, env, solver):
    ec.check_do_nothing(emu_lib, env, solver)


def test_emu_dc_matches_oracle(emu_lib):
    ec.check_do_nothing(emu_lib, 'default14_for_tests_beta', 'fdxb', steps=8, batch=1)


@pytest.mark.parametrize('solver', ['newton', 'fdxb'])
def test_emu_hard_overflow_scenario(emu_lib, solver):
    ec.check_hard_overflow_scenario(emu_lib, solver)


@pytest.mark.parametrize('solver', ['newton', 'fdxb'])
def test_emu_k12_loss_error(emu_lib, solver):
    ec.check_loss_error_scenario(emu_lib, solver)


@pytest.mark.parametrize('solver', ['newton', 'fdxb'])
def test_emu_k3_node_splitting_all_substations(emu_lib, solver):
    nodes = list(range(1, 15))
    flags = ec.check_topology_scenarios(emu_lib, 'default14_for_tests_alpha', nodes, 7, _basic_topology_policy, solver)
    for node, f in zip(nodes, flags):
        exp = [0] * 7
        if node == 2:
            exp[6] = 1
        if node == 7:
            exp[0] = 1
        assert f == exp, (node, f)


def test_emu_k3_dc(emu_lib):
    nodes = list(range(1, 15))
    ec.check_topology_scenarios(emu_lib, 'default14_for_tests_beta', nodes, 7, _basic_topology_policy)


def test_emu_config1_default14_dc_1000_steps(emu_lib):
    ec.check_config1_default14_dc(emu_lib)


def test_emu_default118_few_steps(emu_lib):
    ec.check_do_nothing(emu_lib, 'default118', 'newton', steps=3, batch=1)


def test_emu_auto_reset_and_cascade_118(emu_lib):
    import subprocess
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'oracle')])
    assert ec.check_auto_reset_and_cascade_118(emu_lib, steps=20, batch=4) > 0


@pytest.mark.parametrize('env,solver,steps,batch', [('default14_for_tests_alpha', 'newton', 40, 6),
                                                     ('default14_for_tests_alpha', 'fdxb', 40, 6),
                                                     ('default14_for_tests_beta', 'fdxb', 30, 4),
                                                     ('default30', 'newton', 30, 8),
                                                     ('default30', 'fdxb', 20, 4),
                                                     ('default118', 'newton', 12, 3)])
def test_emu_random_actions_vs_c_oracle(emu_lib, env, solver, steps, batch):
    import subprocess
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'oracle')])
    st = ec.check_random_actions_vs_c_oracle(emu_lib, env, steps, batch, solver)
    assert st['split_buses'] > 0


@pytest.mark.parametrize('env,steps,batch', [('default14_for_tests_alpha', 40, 32), ('default118', 9, 24)])
def test_emu_device_reward_matches_restatement(emu_lib, env, steps, batch):
    seen = ec.check_device_reward(emu_lib, env, steps, batch)
    assert seen['ok'] > 0


@pytest.mark.parametrize('env,batch,k', [('default14_for_tests_alpha', 12, 5), ('default118', 6, 3)])
def test_emu_candidate_search_equals_simulate(emu_lib, env, batch, k):
    ec.check_candidate_search(emu_lib, env, batch, k)


def test_emu_intermediate_busbar_capacity(emu_lib):
    st = ec.check_random_actions_vs_c_oracle(emu_lib, 'default118', 8, 4, 'newton', seed=77, max_active_buses=150)
    assert st['split_buses'] > 0


@pytest.mark.parametrize('solver', ['newton', 'fdxb'])
def test_emu_hard_game_over_mode(emu_lib, solver):
    assert ec.check_hard_game_over_mode(emu_lib, solver=solver) >= 2


def test_emu_random_chronic_looping(emu_lib):
    import subprocess
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'oracle')])
    ec.check_random_chronic_looping(emu_lib)


def test_emu_reduced_observation_layouts(emu_lib):
    ec.check_reduced_observation_layouts(emu_lib, 'default14_for_tests', steps=6, batch=2)


def test_emu_full_size_check_at_small_size(emu_lib):
    """The full-size lock-step harness of the GPU tests (BASELINE.json configs[1] / configs[2]) on a few environments."""
    st = ec.check_full_size_lockstep(emu_lib, 'default14', 16, 30, 10)
    assert st['slots'] >= 2
    st = ec.check_full_size_lockstep(emu_lib, 'default118', 6, 12, 4, bench_limits=True, max_active_buses=118, game_over_mode='hard')
    assert st['done'] > 0


def test_emu_deferred_restart_equals_fused(emu_lib):
    assert ec.check_deferred_restart(emu_lib, steps=16, batch=6) > 0
    assert ec.check_deferred_restart(emu_lib, 'default14_for_tests_alpha', steps=40, batch=8, bench_limits=False, max_active_buses=0,
                                     random_acts=True) > 0


@pytest.mark.parametrize('kw', [
    dict(steps=30, batch=10, limits_file='bench_limits_110.json'),                                  # the 110 % limit rule: most steps end the episode, restarts after restarts
    dict(steps=24, batch=8),                                                                        # the bench workload's limits
    dict(steps=24, batch=8, random_acts=True, max_active_buses=0),                                  # node splitting: four-word kernels, schedules of their own
    dict(steps=30, batch=8, limits_file='bench_limits_110.json', solver='fdxb', hard=True),         # the reference's solver, hard game-over mode (next chronic)
    dict(envname='default14_for_tests_alpha', steps=60, batch=8, limits_file=None, max_active_buses=0, random_acts=True, oracle=False),
    dict(steps=30, batch=10, limits_file='bench_limits_110.json', max_bytes=3 * 20000),             # a memo that holds three snapshots: the rest is computed
])
def test_emu_restart_memo(emu_lib, kw):
    """ppn_restart_memo: restarts served from snapshots leave every field as the computed restart does."""
    st = ec.check_restart_memo(emu_lib, **kw)
    assert st['episodes_ended'] > 0 and st['snapshots'] > 0, st
    if kw.get('limits_file') == 'bench_limits_110.json' and not kw.get('max_bytes'):
        assert st['served'] > 0, st
    print(kw, st)


@pytest.mark.parametrize('kw', [
    dict(steps=44, batch=10, limits_file='bench_limits_110.json'),
    dict(steps=44, batch=8),
    dict(steps=40, batch=6, random_acts=True, max_active_buses=0, layout='minimalist', dtype=np.float32),
])
def test_emu_restart_memo_under_the_fused_restart(emu_lib, kw):
    """The memo where the restart happens in the launch that ended the episode (ppn_step_observe, ppn_rollout_policy): learning steps
    and served restarts against the plain fused launches."""
    st = ec.check_restart_memo_fused(emu_lib, **kw)
    assert st['episodes_ended'] > 0 and st['snapshots'] > 0, st
    print(kw, st)


def test_emu_restart_memo_is_dropped_with_the_thermal_limits(emu_lib):
    held = ec.check_restart_memo_invalidation(emu_lib)
    assert all(h > 0 for h in held), held


def test_emu_async_session_with_restart_memo(emu_lib):
    """The step server serves restarts from snapshots an earlier life of the engine learned (deferred steps with learning passes)."""
    st = ec.check_async_equals_stepping(emu_lib, 'default118', batch=12, n_steps=10, solver='newton', min_ready=3, max_active_buses=118,
                                        memo_warm=12)
    assert st['steps'] == 120 and st['done'] > 0, st


@pytest.mark.parametrize('solver', ['newton', 'fdxb'])
def test_emu_repacked_schedule(emu_lib, solver):
    """Pivots and Schur rounds of the shared schedule re-packed on the host vs the schedule as built, and the oracle."""
    assert ec.check_repacked_schedule(emu_lib, steps=6, batch=6, solver=solver) > 0


@pytest.mark.parametrize('env,solver,dc', [('default14', 'newton', False), ('default14', 'fdxb', False), ('default14', 'fdxb', True),
                                           ('default118', 'newton', False)])
def test_emu_runpf_arrays_solve_boundary(emu_lib, env, solver, dc):
    seen = ec.check_runpf_arrays(emu_lib, env, 32 if env == 'default118' else 48, solver=solver, dc=dc)
    assert seen['ok'] >= 16 and seen['split'] and seen['lines_out'] and seen['prods_off'], seen


@pytest.mark.parametrize('solver', ['newton', 'fdxb'])
def test_emu_soft_overflow_scenario_k2(emu_lib, solver):
    ec.check_soft_overflow_scenario(emu_lib, solver)


def test_emu_restart_goes_on_after_the_attempt_cap(emu_lib):
    assert ec.check_restart_goes_on(emu_lib, batch=64, steps=10) >= 1


def test_emu_rollout_equals_steps(emu_lib):
    assert ec.check_rollout_equals_steps(emu_lib, 'default14_for_tests_alpha', batch=12, n_steps=12, bench_limits=False, random_acts=True) > 0
    assert ec.check_rollout_equals_steps(emu_lib, 'default118', batch=6, n_steps=5, modes=(2,)) >= 0


def test_emu_rollout_with_environments_over_at_the_start(emu_lib):
    assert ec.check_rollout_dead_at_start(emu_lib, batch=24, n_steps=4) >= 1


@pytest.mark.parametrize('solver', ['newton', 'fdxb', 'dc'])
def test_emu_with_nan_poisoned_lds(emu_lib, solver, monkeypatch):
    """LDS is not zeroed between workgroups on the GPU.  The emulation normally fills it with 0xA5 bytes (tiny negative doubles);
    filled with 0xFF every double a kernel reads before writing it is a NaN and every u8 / u16 index is out of range: a step on
    the cascade workload (restarts, cascades, all solver flavours) must come out the same."""
    monkeypatch.setenv('PPN_EMU_LDS_FILL', '255')
    ec.check_auto_reset_and_cascade_118(emu_lib, steps=6, batch=4, solver=solver) if solver != 'dc' else \
        ec.check_do_nothing(emu_lib, 'default14_for_tests_beta', 'dc', steps=6, batch=2)


def test_emu_k1_style_rows_on_ieee118(emu_lib):
    assert ec.check_k1_rows_118(emu_lib, max_active_buses=118) >= 1


@pytest.mark.parametrize('solver,auto_reset', [('newton', True), ('fdxb', 2), ('newton', 2)])
def test_emu_schedule_prepass_is_a_pure_cache_warmer(emu_lib, solver, auto_reset):
    st = ec.check_schedule_prepass(emu_lib, steps=12, batch=8, solver=solver, auto_reset=auto_reset, double_acts=True)
    assert st['illegal'] > 0 and st['split'] > 0, st


def test_emu_step_report_field(emu_lib):
    from helpers import load_env
    from harness import engine_with_library
    case, cfg, chronics = load_env('default14_for_tests_alpha', conf={'solver': 'newton'})
    case.ntopo_offset_lines = case.nP + case.nL + 2 * case.nl
    eng = engine_with_library(emu_lib, case, cfg, 8, chronics=chronics)
    eng.reset()
    rng = np.random.default_rng(9)
    for t in range(20):
        eng.step(ec.random_actions(case, rng, 8), auto_reset=2 if t % 2 else True)
        rep = eng.read('STEP_REPORT')
        assert np.array_equal(rep[:, 0] != 0, eng.read('DONE').astype(bool))
        assert np.array_equal(rep[:, 1].astype(np.int32), eng.read('FLAG'))
        np.testing.assert_allclose(rep[:, 2], eng.read('REWARD').sum(axis=1), rtol=1e-13, atol=1e-10)
    eng.close()


def test_emu_policy_rollout_equals_stepping(emu_lib):
    assert ec.check_policy_rollout_equals_stepping(emu_lib, 'default118', batch=6, n_steps=14, params=(0.9,), max_active_buses=118) > 0
    assert ec.check_policy_rollout_equals_stepping(emu_lib, 'default14_for_tests_alpha', batch=8, n_steps=30, params=(0.5,), bench_limits=False) > 0


@pytest.mark.parametrize('small_ecap,auto_reset', [(0, 2), (664, True), (656, 2)])
def test_emu_two_capacity_stepping(emu_lib, small_ecap, auto_reset):
    st = ec.check_two_capacity_stepping(emu_lib, steps=14, batch=8, small_ecap=small_ecap, auto_reset=auto_reset)
    assert st['small'] > 0 and (st['big'] > 0 or not small_ecap), st


def test_emu_two_capacity_stepping_fast_decoupled(emu_lib):
    st = ec.check_two_capacity_stepping(emu_lib, steps=14, batch=8, small_ecap=660, auto_reset=True, solver='fdxb')
    assert st['small'] > 0 and st['big'] > 0, st


@pytest.mark.parametrize('envname,solver,layout,dtype,auto_reset', [
    ('default14', 'newton', 'full', 'float64', True), ('default14', 'fdxb', 'minimalist', 'float32', True),
    ('default14_for_tests_alpha', 'newton', 'ac_minimalist', 'float64', False), ('default30', 'dc', 'full', 'float32', True)])
def test_emu_step_observe_equals_step_then_read(emu_lib, envname, solver, layout, dtype, auto_reset):
    import numpy as np
    ended = ec.check_step_observe(emu_lib, envname, 5, 14, solver, layout, np.dtype(dtype), auto_reset=auto_reset)
    assert ended > 0 or envname != 'default14'


def test_emu_step_observe_two_capacity_118(emu_lib):
    """Four-word kernels with two-capacity stepping (both class launches write their environments' rows)."""
    import os
    os.environ['PPN_TWO_CAP_ECAP'] = '760'
    try:
        ec.check_step_observe(emu_lib, 'default118', 4, 4, 'newton', 'full')
    finally:
        del os.environ['PPN_TWO_CAP_ECAP']


@pytest.mark.parametrize('auto_reset', [True, 2])
def test_emu_limit_rule_110_restart_after_restart(emu_lib, auto_reset):
    """SURVEY.md 8d's limit rule for configs[2] (limit = max(50, 1.10 x I(t = 0))): nearly every step ends in game over and some
    restarts run out of their 64 attempts (PPN_F_DEAD = 3) -- in lock-step with the C oracle, fused and deferred restart, with
    N_STEPS / DEAD / EPOCH compared (the GPU test runs the full 4096 x 30)."""
    st = ec.check_full_size_lockstep(emu_lib, 'default118', 96, 14, 2, bench_limits=True, max_active_buses=118,
                                     limits_file='bench_limits_110.json', restarts=True, auto_reset=auto_reset)
    assert st['done'] > 800 and st['stuck'] > 0, st


@pytest.mark.parametrize('envname,solver,layout,dtype,min_ready,kw', [
    ('default118', 'newton', 'full', 'float64', 3, dict(max_active_buses=118)),
    ('default118', 'fdxb', 'minimalist', 'float32', 1, dict(max_active_buses=118)),
    ('default14_for_tests_alpha', 'newton', 'ac_minimalist', 'float64', 6, dict(bench_limits=False)),
    ('default118', 'newton', 'full', 'float64', 4, dict())])        # (four-word kernels: every busbar may be active)
def test_emu_async_send_recv_equals_stepping(emu_lib, envname, solver, layout, dtype, min_ready, kw):
    """ppn_send / ppn_recv (host logic, rings, settle-on-other-calls; the emulated server completes items in a shuffled order)."""
    st = ec.check_async_equals_stepping(emu_lib, envname, batch=7, n_steps=12, solver=solver, layout=layout, dtype=np.dtype(dtype),
                                        min_ready=min_ready, rows_by_env=(min_ready == 1), **kw)
    assert st['steps'] == 7 * 12 and st['settled'] >= 1


def test_emu_async_session_errors(emu_lib):
    from helpers import load_env
    from harness import engine_with_library
    from pypownet_amd.engine import EngineError
    case, cfg, chronics = load_env('default14_for_tests', conf={'solver': 'newton'})
    eng = engine_with_library(emu_lib, case, cfg, 4, chronics=chronics)
    eng.reset()
    act = np.zeros((4, case.action_length), dtype=np.uint8)
    with pytest.raises(EngineError):
        eng.send([0], act[:1])                       # no session
    eng.async_start()
    eng.send([0, 1], act[:2])
    with pytest.raises(EngineError):
        eng.send([1], act[:1])                       # in flight
    with pytest.raises(EngineError):
        eng.send([2, 7], act[:2])                    # out of range -- and environment 2 must not be left marked in flight
    eng.send([2], act[:1])
    got = set()
    while len(got) < 3:
        got |= set(int(v) for v in eng.recv(min_ready=1))
    assert got == {0, 1, 2} and len(eng.recv(min_ready=1, timeout_ms=0)) == 0
    assert eng.async_stats()['in_flight'] == 0
    eng.async_stop()
    eng.close()


@pytest.mark.parametrize('solver', ['newton', 'fdxb'])
def test_emu_candidate_slots_keep_their_schedules(emu_lib, solver):
    assert ec.check_candidate_schedule_cache(emu_lib, batch=5, k=4, rounds=6, solver=solver) > 0
