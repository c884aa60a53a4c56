"""RunEnv drop-in API exercised end to end on the CPU emulation build (the same Python classes drive the GPU
library).  Scenarios follow the reference's own tests: obs array <-> object round trip (tests/test_core.py:44-85,
1430-1459, K11), simulate() leaves no trace (tests/test_simulate.py:371-536, K10), illegal-action payloads,
WrappedRunner protocol."""
import os

import numpy as np
import pytest

from helpers import ROOT, ENVS, oracle_game, do_nothing
from oracle.game_np import obs_as_array
from test_emu_engine import emu_lib  # noqa: F401  (fixture)


@pytest.fixture(autouse=True)
def _emulation_library(emu_lib):  # noqa: F811
    """Every RunEnv of this module (and the Game / Engine objects it re-creates on reset()) binds to the emulation build."""
    import harness
    with harness.library(emu_lib):
        yield


def make_env(emu_lib, name, **kw):
    from pypownet_amd.environment import RunEnv
    return RunEnv(os.path.join(ENVS, name), 'level0', **kw)


def test_obs_roundtrip_and_oracle_match(emu_lib):
    env = make_env(emu_lib, 'default14_for_tests')
    g = oracle_game('default14_for_tests')
    arr = env.get_observation()
    assert arr.shape == (env.game.case.observation_length,) == (538,)
    obs = env.observation_space.array_to_observation(arr)
    assert np.array_equal(obs.as_array(), arr)
    np.testing.assert_allclose(arr, obs_as_array(g.export_observation()), rtol=0, atol=1e-6)
    assert len(obs.as_minimalist().as_array()) + 0 < len(obs.as_ac_minimalist().as_array()) < len(arr)
    conf, types = obs.get_nodes_of_substation(2)
    assert list(conf) == [0] * 6 and len(types) == 6
    st, other = obs.get_lines_status_of_substation(1)
    assert list(st) == [1, 1] and other == [2, 5]


def test_step_returns_reference_tuple_and_illegal_payload(emu_lib):
    from pypownet_amd.environment import IllegalActionException
    env = make_env(emu_lib, 'default14_for_tests')
    a = env.action_space.get_do_nothing_action(as_class_Action=True)
    env.action_space.set_lines_status_switch_from_id(a, 18, 1)
    obs, reward, done, flag = env.step(a)
    assert flag is None and not done and isinstance(reward, float)
    o = env.observation_space.array_to_observation(obs)
    assert int(o.lines_status[18]) == 0 and int(o.timesteps_before_lines_reactionable[18]) == 2
    # switching it back immediately is on cooldown -> IllegalActionException returned (not raised), line stays off
    a = env.action_space.get_do_nothing_action(as_class_Action=True)
    env.action_space.set_lines_status_switch_from_id(a, 18, 1)
    obs, rew_list, done, flag = env.step(a, do_sum=False)
    assert isinstance(flag, IllegalActionException) and not done and len(rew_list) == 1
    assert flag.get_illegal_oncoolown_lines_switches()[18] and flag.get_illegal_broken_lines_reconnections() is None
    assert int(env.observation_space.array_to_observation(obs).lines_status[18]) == 0
    # too many activations (max 2 lines in this env) -> whole action dropped
    a = env.action_space.get_do_nothing_action(as_class_Action=True)
    for k in (1, 2, 3):
        env.action_space.set_lines_status_switch_from_id(a, k, 1)
    assert not env.is_action_valid(a)
    obs, r, done, flag = env.step(a)
    assert isinstance(flag, IllegalActionException) and flag.get_has_too_much_activations()
    assert list(env.observation_space.array_to_observation(obs).lines_status[:4].astype(int)) == [1, 1, 1, 1]
    with pytest.raises(ValueError):
        env.step(np.zeros(5))


def test_simulate_leaves_no_trace(emu_lib):
    env = make_env(emu_lib, 'default14_for_tests')
    ref = make_env(emu_lib, 'default14_for_tests')
    dn = env.action_space.get_do_nothing_action()
    for t in range(6):
        cand = env.action_space.get_do_nothing_action(as_class_Action=True)
        env.action_space.set_lines_status_switch_from_id(cand, (3 * t) % 20, 1)
        before = env.get_observation()
        sobs, srew, sdone, sflag = env.simulate(cand)
        assert np.array_equal(before, env.get_observation())          # nothing moved
        if sobs is not None:
            so = env.observation_space.array_to_observation(sobs)
            # simulate plays the PLANNED injections of the current entry (game.py:415-419)
            cur = env.observation_space.array_to_observation(before)
            assert np.array_equal(so.active_loads, cur.planned_active_loads)
        o1 = env.step(dn)
        o2 = ref.step(dn)
        assert np.array_equal(o1[0], o2[0]) and o1[1] == o2[1]


def test_game_over_and_process_game_over(emu_lib):
    from pypownet_amd.environment import TooManyProductionsCut
    env = make_env(emu_lib, 'default14_for_tests')
    env.process_game_over()
    a = env.action_space.get_do_nothing_action(as_class_Action=True)
    a.prods_switches_subaction[0] = 1
    obs, r, done, flag = env.step(a)
    assert not done
    a = env.action_space.get_do_nothing_action(as_class_Action=True)
    a.prods_switches_subaction[4] = 1
    obs, r, done, flag = env.step(a)
    assert done and obs is None and isinstance(flag, TooManyProductionsCut)
    arr = env.process_game_over()
    o = env.observation_space.array_to_observation(arr)
    assert list(o.productions_nodes.astype(int)) == [0] * 5


def test_default_reward_signal(emu_lib):
    from pypownet_amd.reward_signal import DefaultGridRewardSignal
    env = make_env(emu_lib, 'default14_for_tests')
    env.reward_signal = DefaultGridRewardSignal(14)
    obs, r, done, flag = env.step(env.action_space.get_do_nothing_action(), do_sum=False)
    assert len(r) == 5 and r[0] == 0 and r[1] == 0 and r[2] == 0 and r[3] == 0 and r[4] < 0
    o = env.observation_space.array_to_observation(obs)
    assert abs(r[4] + np.sum(np.square(o.ampere_flows / o.thermal_limits))) < 1e-12


def test_device_reward_equals_host_reward_signal(emu_lib):
    """The reward computed inside the step kernel (PPN_F_REWARD) equals what RunEnv.step returns when it evaluates the
    shipped five-component reward class on the Observation object (the reference's path, environment.py:866-874) -- for
    legal, illegal (repaired in place) and wholly rejected actions, for step and simulate."""
    from pypownet_amd.reward_signal import DefaultGridRewardSignal
    env = make_env(emu_lib, 'default14_for_tests')
    env.reward_signal = DefaultGridRewardSignal(14)
    sp = env.action_space
    rng = np.random.default_rng(7)
    kinds = set()
    for t in range(40):
        a = sp.get_do_nothing_action()
        if t % 2:
            sub = int(rng.choice(sp.substations_ids))
            n = sp.get_number_elements_of_substation(sub)
            sp.set_substation_switches_in_action(a, sub, rng.integers(0, 2, size=n))
        if t % 3 == 0:
            a[len(a) - env.game.case.nl + int(rng.integers(env.game.case.nl))] = 1     # a line-status switch
        if t % 7 == 6:
            a[-8:] = 1                      # beyond max_number_actionned_lines
        _, rs, _, _ = env.simulate(np.array(a), do_sum=False)
        np.testing.assert_allclose(env.game.engine.read('REWARD', simulation=True)[0], rs, rtol=1e-12, atol=1e-12)
        obs, r, done, flag = env.step(np.array(a), do_sum=False)
        np.testing.assert_allclose(env.game.engine.read('REWARD')[0], r, rtol=1e-12, atol=1e-12)
        kinds.add(type(flag).__name__)
        if done:
            env.process_game_over()
    assert 'IllegalActionException' in kinds and 'NoneType' in kinds, kinds


def test_reduced_observation_layouts_are_prefixes(emu_lib):
    """ppn_read_observation: MinimalistObservation / MinimalistACObservation arrays gathered on the device equal what the
    reference's as_minimalist() / as_ac_minimalist() give on the Observation object; float32 = the rounded float64."""
    env = make_env(emu_lib, 'default14_for_tests')
    for _ in range(3):
        obs, *_ = env.step(env.action_space.get_do_nothing_action())
    o = env.observation_space.array_to_observation(obs)
    eng = env.game.engine
    mini = eng.observations(layout='minimalist')[0]
    ac = eng.observations(layout='ac_minimalist')[0]
    assert np.array_equal(mini, o.as_minimalist().as_array())
    assert np.array_equal(ac, o.as_ac_minimalist().as_array())
    assert np.array_equal(mini, obs[:len(mini)]) and np.array_equal(ac, obs[:len(ac)])
    for lay, ref in (('minimalist', mini), ('ac_minimalist', ac), ('full', obs)):
        f32 = eng.observations(layout=lay, dtype=np.float32)[0]
        assert f32.dtype == np.float32 and np.array_equal(f32, ref.astype(np.float32))
