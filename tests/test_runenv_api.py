"""RunEnv drop-in API exercised end to end, twice: on the CPU emulation build of the kernels (runs everywhere) and --
marked gpu -- on the real pypownet_amd/libppn.so (the same Python classes, the product's own library path).  Scenarios
follow the reference's own tests: cooldowns and activation maxima (tests/test_basic.py:339-606, 730-893, K7), simulate()
leaves no trace (tests/test_simulate.py:371-536, K10), obs array <-> object round trip (tests/test_core.py:44-85,
1430-1459, K11), illegal-action payloads, game over / WrappedRunner protocol."""
import os

import numpy as np
import pytest

from helpers import ROOT, ENVS, oracle_game, do_nothing
from oracle.game_np import obs_as_array
import test_emu_engine

_emu_build = test_emu_engine.emu_lib      # the session fixture that compiles build/libppn_emu.so, under another name


@pytest.fixture(params=['emu', pytest.param('hip', marks=pytest.mark.gpu)])
def emu_lib(request):
    """Library the RunEnv objects of a test bind to: the emulation build (path), or None = the product's libppn.so."""
    return request.getfixturevalue('_emu_build') if request.param == 'emu' else None


@pytest.fixture(autouse=True)
def _bound_library(emu_lib):
    """Every RunEnv of this module (and the Game / Engine objects it re-creates on reset()) binds to that library."""
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


REWARD_FILE_AGAINST_REFERENCE_API = '''
import pypownet.environment
import pypownet.reward_signal


class CustomRewardSignal(pypownet.reward_signal.RewardSignal):
    """Written the way an environment folder of the reference writes it: against the package name `pypownet`."""

    def compute_reward(self, observation, action, flag):
        if flag is None:
            return [1., float(len(observation.ampere_flows))]
        if isinstance(flag, pypownet.environment.IllegalActionException):
            return [2., 0.]
        if isinstance(flag, pypownet.environment.DivergingLoadflowException):
            return [3., 0.]
        if isinstance(flag, (pypownet.environment.TooManyProductionsCut, pypownet.environment.TooManyConsumptionsCut)):
            return [4., 0.]
        return [-1., 0.]
'''


def test_environment_reward_file_written_against_reference_package_loads(emu_lib, tmp_path):
    """an unmodified reference environment folder ships a reward_signal.py that imports `pypownet.*`; it must
    load (not fall back to the [0.] base signal) and its isinstance tests must see the flags RunEnv.step returns."""
    import sys
    from pypownet_amd.environment import RunEnv
    folder = tmp_path / 'env_with_reward'
    folder.mkdir()
    os.symlink(os.path.join(ENVS, 'default14_for_tests', 'level0'), str(folder / 'level0'))
    (folder / 'reward_signal.py').write_text(REWARD_FILE_AGAINST_REFERENCE_API)
    env = RunEnv(str(folder), 'level0')
    assert type(env.reward_signal).__name__ == 'CustomRewardSignal'
    assert 'pypownet' not in sys.modules or not hasattr(sys.modules['pypownet'], '__path__') or sys.modules['pypownet'].__path__ != []
    obs, r, done, flag = env.step(env.action_space.get_do_nothing_action(), do_sum=False)
    assert r == [1., 20.] and flag is None
    a = env.action_space.get_do_nothing_action(as_class_Action=True)
    env.action_space.set_lines_status_switch_from_id(a, 18, 1)
    env.step(a)
    a = env.action_space.get_do_nothing_action(as_class_Action=True)
    env.action_space.set_lines_status_switch_from_id(a, 18, 1)       # on cooldown: IllegalActionException returned
    obs, r, done, flag = env.step(a, do_sum=False)
    assert r == [2., 0.] and not done
    a = env.action_space.get_do_nothing_action(as_class_Action=True)  # K5: productions 1 and 8 moved off -> too many cut
    a.prods_switches_subaction[0] = 1
    env.step(a)
    for _ in range(3):
        env.step(env.action_space.get_do_nothing_action())
    a = env.action_space.get_do_nothing_action(as_class_Action=True)
    a.prods_switches_subaction[4] = 1
    obs, r, done, flag = env.step(a, do_sum=False)
    assert done and r == [4., 0.]


def test_simulated_observation_equals_oracle_simulate(emu_lib):
    """Game.simulate does not advance current_timestep_entries (game.py:410-413), so the planned_*
    fields of a simulated observation are those of the CURRENT entry (they equal the simulated injections) while date
    and planned maintenance come from the next row.  Whole array against OracleGame.simulate."""
    env = make_env(emu_lib, 'default14_for_tests')
    g = oracle_game('default14_for_tests')
    for t in range(4):
        a = do_nothing(env.game.case)
        if t == 2:
            a[env.game.case.nP + env.game.case.nL + 3] = 1
        obs, _, done, _ = env.step(np.array(a))
        o, f, il, d = g.step(a.copy())
        assert not done and not d
        np.testing.assert_allclose(obs, obs_as_array(o), rtol=0, atol=1e-6)
        sim_obs, _, sdone, _ = env.simulate(np.array(a))
        so, sf, sil, sd = g.simulate(a.copy())
        assert sdone == sd
        if not sd:
            np.testing.assert_allclose(sim_obs, obs_as_array(so), rtol=0, atol=1e-6)
            ref = env.observation_space.array_to_observation(sim_obs)
            np.testing.assert_array_equal(ref.planned_active_loads, ref.active_loads)      # K9 (tests/test_simulate.py:275-327)


# ---- K7: cooldowns and activation maxima, the reference's scripted agents restated (tests/test_basic.py:339-606) ----------
def _wrapped_run(env, policy, n_iter):
    """tests/common_assets.py WrappedRunner.loop: process_game_over() first; (flags, game_overs) per step."""
    obs = env.process_game_over()
    flags, overs = [], []
    for i in range(1, n_iter + 1):
        o = env.observation_space.array_to_observation(obs)
        obs, reward, done, flag = env.step(policy(i, o))
        flags.append(flag)
        overs.append(done)
        if done:
            obs = env.process_game_over()
    return flags, overs


@pytest.mark.parametrize('line', [17, 18, 0, 1, 2])
def test_k7_line_cooldown(emu_lib, line):
    """n_timesteps_actionned_line_reactionable = 3 (tests/test_basic.py:508-606, 730-772)."""
    from pypownet_amd.environment import IllegalActionException
    env = make_env(emu_lib, 'default14_for_tests')
    sp = env.action_space
    assert sp.lines_status_subaction_length == 20

    def policy(i, o):
        a = sp.get_do_nothing_action(as_class_Action=True)
        if i in (2, 3, 4):
            assert o.lines_status[line] == 0
        if i in (5, 6):
            assert o.lines_status[line] == 1
        if i <= 5:
            sp.set_lines_status_switch_from_id(action=a, line_id=line, new_switch_value=1)
        return a
    flags, overs = _wrapped_run(env, policy, 6)
    assert overs == [False] * 6
    for i, f in enumerate(flags):
        assert (f is None) if i in (0, 3, 5) else isinstance(f, IllegalActionException), (i, f)


@pytest.mark.parametrize('sub', [1, 2, 3, 6, 8])
def test_k7_node_cooldown(emu_lib, sub):
    """n_timesteps_actionned_node_reactionable = 3 (tests/test_basic.py:607-716, 775-821)."""
    from pypownet_amd.environment import IllegalActionException
    env = make_env(emu_lib, 'default14_for_tests')
    sp = env.action_space

    def policy(i, o):
        a = sp.get_do_nothing_action(as_class_Action=True)
        n = sp.get_number_elements_of_substation(sub)
        conf, _ = o.get_nodes_of_substation(sub)
        target = np.zeros(n)
        if i == 1:
            target[0] = 1
        else:
            assert list(conf[:2]) == ([1, 1] if i == 5 else [1, 0]) and not conf[2:].any()
            if i < 5:
                target[1] = 1
        sp.set_substation_switches_in_action(action=a, substation_id=sub, new_values=target)
        cur, _ = sp.get_substation_switches_in_action(a, sub)
        assert np.all(cur == target)
        return a
    flags, overs = _wrapped_run(env, policy, 5)
    assert overs == [False] * 5
    for i, f in enumerate(flags):
        assert (f is None) if i in (0, 3, 4) else isinstance(f, IllegalActionException), (i, f)


def test_k7_max_number_actionned_lines(emu_lib):
    """max_number_actionned_lines = 2 (tests/test_basic.py:433-505, 824-857)."""
    from pypownet_amd.environment import IllegalActionException
    env = make_env(emu_lib, 'default14_for_tests')
    sp = env.action_space

    def policy(i, o):
        a = sp.get_do_nothing_action(as_class_Action=True)
        ls = o.lines_status
        if i == 2:
            assert (ls == 1).all()
        if i in (3, 4):
            assert ls[0] == 0 and ls[1] == 0 and (ls[2:15] == 1).all()
        for l in {1: (1, 0, 5), 2: (0, 1), 3: (2, 3, 4, 5)}.get(i, ()):
            sp.set_lines_status_switch_from_id(action=a, line_id=l, new_switch_value=1)
        return a
    flags, overs = _wrapped_run(env, policy, 5)
    assert overs == [False] * 5
    for i, f in enumerate(flags):
        if i in (1, 3, 4):
            assert f is None
        else:
            assert isinstance(f, IllegalActionException) and f.get_has_too_much_activations()


def test_k7_max_number_actionned_substations(emu_lib):
    """max_number_actionned_substations = 2 (tests/test_basic.py:339-430, 860-893)."""
    from pypownet_amd.environment import IllegalActionException
    env = make_env(emu_lib, 'default14_for_tests')
    sp = env.action_space

    def policy(i, o):
        a = sp.get_do_nothing_action(as_class_Action=True)
        if i == 1:
            sp.set_substation_switches_in_action(a, 4, [1, 0, 0, 0, 0, 0])
            sp.set_substation_switches_in_action(a, 5, [0, 1, 0, 0, 0])
            sp.set_substation_switches_in_action(a, 6, [0, 0, 1, 0, 0, 0])
        if i == 2:
            for s_, n_ in ((4, 6), (5, 5), (6, 6)):
                assert list(o.get_nodes_of_substation(s_)[0]) == [0] * n_
            sp.set_substation_switches_in_action(a, 4, [1, 0, 0, 0, 0, 0])
            sp.set_substation_switches_in_action(a, 5, [0, 1, 0, 0, 0])
        if i == 3:
            assert list(o.get_nodes_of_substation(4)[0]) == [1, 0, 0, 0, 0, 0]
            assert list(o.get_nodes_of_substation(5)[0]) == [0, 1, 0, 0, 0]
        return a
    flags, overs = _wrapped_run(env, policy, 3)
    assert overs == [False] * 3
    assert isinstance(flags[0], IllegalActionException) and flags[1] is None and flags[2] is None


def test_k10_cumulative_reward_unchanged_by_simulations(emu_lib):
    """tests/test_simulate.py:371-461: the cumulative reward of a run is the same with and without simulate() calls
    interleaved (single candidate and exhaustive line sweep) -- simulate leaves no trace in the game."""
    from pypownet_amd.reward_signal import DefaultGridRewardSignal

    def run(n_sim):
        env = make_env(emu_lib, 'default14_for_tests')
        env.reward_signal = DefaultGridRewardSignal(14)
        total = 0.0
        for t in range(8):
            for k in range(n_sim):
                cand = env.action_space.get_do_nothing_action(as_class_Action=True)
                env.action_space.set_lines_status_switch_from_id(cand, (t + k) % 20, 1)
                env.simulate(cand)
            a = env.action_space.get_do_nothing_action(as_class_Action=True)
            if t == 3:
                env.action_space.set_lines_status_switch_from_id(a, 7, 1)
            _, r, done, _ = env.step(a)
            total += r
            if done:
                env.process_game_over()
        return total, env.get_observation()
    base, obs0 = run(0)
    for n_sim in (1, 20):
        tot, obs = run(n_sim)
        assert tot == base and np.array_equal(obs, obs0)


def test_exception_texts_and_observation_types(emu_lib):
    """the IllegalActionException text carries the 'Ignoring ...' suffix of the repair (game.py:821-846), a solver
    that cannot run reports 'The grid is not connexe' (grid.py:231, 238), Game.export_observation hands out integer fields."""
    from pypownet_amd.environment import IllegalActionException, DivergingLoadflowException
    env = make_env(emu_lib, 'default14_for_tests')
    sp = env.action_space
    a = sp.get_do_nothing_action(as_class_Action=True)
    sp.set_lines_status_switch_from_id(a, 18, 1)
    t0 = env.game.timestep
    env.step(a)
    assert env.game.timestep == t0 + 1
    a = sp.get_do_nothing_action(as_class_Action=True)
    sp.set_lines_status_switch_from_id(a, 18, 1)
    _, _, _, flag = env.step(a)
    assert isinstance(flag, IllegalActionException)
    assert 'Trying to action on-cooldown line 18, must wait resp. 2 timesteps. ' in flag.text
    assert flag.text.endswith(' Ignoring action switches of on-cooldown lines: 18.')
    assert env.game.timestep == t0 + 3          # the repaired action is re-submitted: apply_action ran twice
    o = env.game.export_observation()
    assert isinstance(o.date_year, int) and o.lines_status.dtype.kind == 'i' and o.are_loads_cut.dtype.kind == 'i'
    assert '%d' % o.date_year == str(o.date_year)
    # K3: splitting substation 7 of default14_for_tests_alpha islands bus 6667 -> the solver cannot run
    env = make_env(emu_lib, 'default14_for_tests_alpha')
    env.process_game_over()
    sp = env.action_space
    a = sp.get_do_nothing_action(as_class_Action=True)
    n = sp.get_number_elements_of_substation(7)
    sp.set_substation_switches_in_action(a, 7, [1] + [0] * (n - 1))
    obs, r, done, flag = env.step(a)
    assert done and isinstance(flag, DivergingLoadflowException)
    assert flag.text == 'The grid is not connexe: cascading emulation of depth 0 has diverged', flag.text


def test_random_chronic_looping_mode_through_runenv(emu_lib):
    """RunEnv(chronic_looping_mode='random') is accepted like in the reference (chronic.py:266-291) and reproducible under `seed`."""
    names = []
    for seed in (3, 3, 4, 5, 6, 7):
        from pypownet_amd.environment import RunEnv
        env = RunEnv(os.path.join(ENVS, 'default14'), 'level0', chronic_looping_mode='random', game_over_mode='hard', seed=seed)
        seq = [env.game.get_current_chronic_name()]
        for _ in range(5):
            env.process_game_over()
            seq.append(env.game.get_current_chronic_name())
        names.append(tuple(seq))
    assert names[0] == names[1]
    seen = set(n for s_ in names for n in s_)
    assert len(set(names)) > 1 and len(seen) >= 8 and seen <= set('abcdefghijkl')
