"""GPU parity tests (-m gpu): the gfx950 build of libppn.so, called through the C ABI, against the CPU oracle on
the same seeded inputs.  Bars (BASELINE.json north_star): bit-exact line status / topology / counters / flags;
|dVm| <= 1e-6 p.u. and |dVa| <= 1e-6 rad (measured margins are ~1e-10, the tests use 1e-8 where the two sides
run the same algorithm)."""
import numpy as np
import pytest

import engine_checks as ec
from test_oracle_known_answers import _basic_topology_policy

pytestmark = pytest.mark.gpu
HIP = None   # default library path: pypownet_amd/libppn.so


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


@pytest.mark.parametrize('env,batch,k', [('default14_for_tests_alpha', 48, 8), ('default118', 32, 6)])
def test_gpu_candidate_search_equals_simulate(env, batch, k):
    ec.check_candidate_search(HIP, env, batch, k)


def test_gpu_reduced_observation_layouts():
    """Reduced / float32 observation layouts gathered on the GPU are prefixes (resp. roundings) of Observation.as_array()."""
    import os
    from helpers import load_env
    from pypownet_amd.engine import Engine
    case, cfg, chronics = load_env('default118', conf={'solver': 'newton'})
    eng = Engine(case, cfg, 8, chronics=chronics)
    eng.reset()
    eng.step(np.zeros((8, case.action_length), dtype=np.uint8))
    full = eng.observations()
    for lay in ('minimalist', 'ac_minimalist', 'full'):
        o = eng.observations(layout=lay)
        assert np.array_equal(o, full[:, :o.shape[1]], equal_nan=True)
        o32 = eng.observations(layout=lay, dtype=np.float32)
        assert np.array_equal(o32, o.astype(np.float32), equal_nan=True)
    assert eng.observations(layout='minimalist').shape[1] < full.shape[1] // 2


@pytest.mark.parametrize('cap,solver', [(150, 'newton'), (128, 'newton'), (150, 'fdxb')])
def test_gpu_intermediate_busbar_capacities(cap, solver):
    """max_active_buses between the substation count and every busbar: the 4-word kernels below full capacity (150) and the
    2-word kernels with spare busbars (128), against the oracle under random node splitting; no environment may hit the capacity flag."""
    st = ec.check_random_actions_vs_c_oracle(HIP, 'default118', 20, 48, solver, seed=77, max_active_buses=cap)
    assert st['split_buses'] > 0
