"""Pins the numpy/scipy oracle (oracle/pf_np.py + oracle/game_np.py) against every known-answer the
reference's own tests hold for the load-flow path.  The expected values below are the
ones asserted (or documented) in /root/reference/tests/test_core.py and test_basic.py; each test cites them.
These are CPU tests (no GPU)."""
import numpy as np
import pytest

from helpers import (oracle_game, do_nothing, set_substation_switches, nodes_of_substation, set_line_switch,
                     differential, run_wrapped)
from oracle.game_np import FLAG_OK, FLAG_DIVERGED, FLAG_TOO_MANY_PRODS, FLAG_TOO_MANY_LOADS, ILL_BROKEN

K1_SERIES = [244, 210, 223, 214, 214, 237, 244, 286, 322, 347, 381, 310, 303, 324, 275]
ALL_ON = [1] * 20
L6_OFF = [1] * 6 + [0] + [1] * 13


@pytest.mark.parametrize('solver', ['fdxb', 'newton'])
def test_k1_line6_ampere_series(solver):
    """tests/test_core.py:919 -- int-truncated ampere flow of line 6 over 15 consecutive do-nothing steps."""
    g = oracle_game('default14_for_tests_hard_overflow', conf={'solver': solver}, without_overflow_cutoff=True)
    got = []

    def policy(i, obs):
        got.append(int(obs['ampere_flows'][6]))
        return do_nothing(g.case)

    dones, flags, ills = run_wrapped(g, policy, 15)
    assert got == K1_SERIES
    assert not any(dones)


@pytest.mark.parametrize('solver', ['fdxb', 'newton'])
def test_k1_hard_overflow_state_machine(solver):
    """tests/test_core.py:968-976, 1423-1427: limit 200 A x 1.5: line 6 hard-breaks when it reaches 322 A; the
    agent tries to reconnect at steps 9..14 -> IllegalAction at i in {8, 9, 11, 12}; all lines on at step 15."""
    g = oracle_game('default14_for_tests_hard_overflow', conf={'solver': solver})
    seen = {}

    def policy(i, obs):
        seen[i] = [int(v) for v in obs['lines_status']]
        a = do_nothing(g.case)
        if 9 <= i < 15:
            assert seen[i] == L6_OFF
            set_line_switch(g.case, a, 6, 1)
        if i == 15:
            assert seen[i] == ALL_ON
        return a

    dones, flags, ills = run_wrapped(g, policy, 15)
    assert dones == [False] * 15
    assert flags == [FLAG_OK] * 15
    for i, ill in enumerate(ills):
        if i in (8, 9, 11, 12):
            assert ill & ILL_BROKEN
        else:
            assert ill == 0


def test_k2_soft_overflow_break_and_reconnect():
    """tests/test_core.py:720-738, 784-811, 1322-1328 (default14_for_tests_alpha: line 6 limit 300 A, soft break
    after 2 consecutive overflowed steps, broken for 2): on at steps 9, 10; off at 11, 12; reconnect refused at
    i in {10, 11}, accepted at step 13, on again at 14."""
    g = oracle_game('default14_for_tests_alpha')

    def policy(i, obs):
        st = [int(v) for v in obs['lines_status']]
        a = do_nothing(g.case)
        if i in (9, 10, 14):
            assert st == ALL_ON
        if i in (11, 12, 13):
            assert st == L6_OFF
            set_line_switch(g.case, a, 6, 1)
        return a

    dones, flags, ills = run_wrapped(g, policy, 14)
    assert not any(dones)
    assert [k for k, v in enumerate(ills) if v] == [10, 11]


def _basic_topology_policy(case, node):
    n_el = int(case.n_elements[int(np.where(case.sub_ids == node)[0][0])])
    topos = [[1 if j == k else 0 for j in range(n_el)] for k in range(n_el)]

    def policy(i, obs):
        a = do_nothing(case)
        cur = nodes_of_substation(case, obs, node)
        if i == 1:
            set_substation_switches(case, a, node, topos[0])
        elif 2 <= i <= 6:
            if i == 2:
                assert cur == ([0, 0, 0] if node == 7 else topos[0])   # tests/test_basic.py:58-63
            if n_el > i - 1:
                set_substation_switches(case, a, node, differential(topos[i - 1], cur))
        return a
    return policy


@pytest.mark.parametrize('node', list(range(1, 15)))
def test_k3_node_splitting_divergence(node):
    """tests/test_basic.py:925-941 (default14_for_tests_alpha, AC): splitting each element of substation k in
    turn; substation 2 -> DivergingLoadflowException at i == 6 only, substation 7 -> at i == 0 only, others none."""
    g = oracle_game('default14_for_tests_alpha')
    dones, flags, ills = run_wrapped(g, _basic_topology_policy(g.case, node), 7)
    expected = [FLAG_OK] * 7
    if node == 2:
        expected[6] = FLAG_DIVERGED
    if node == 7:
        expected[0] = FLAG_DIVERGED
    assert flags == expected
    assert dones == [f != FLAG_OK for f in expected]


@pytest.mark.parametrize('node', list(range(1, 8)))
def test_k4_dc_back_and_forth(node):
    """tests/test_basic.py:944-982 (default14_for_tests_beta, DC mode): 13-step back-and-forth sweep of
    substations 1..7, no flag and no game over (the agent skips substation 7's first, islanding, configuration)."""
    g = oracle_game('default14_for_tests_beta')
    case = g.case
    n_el = int(case.n_elements[node - 1])
    topos = [[1 if j == k else 0 for j in range(n_el)] for k in range(n_el)]
    zeros = [0] * n_el

    def policy(i, obs):
        # step table of Agent_test_AdvancedSubstationTopologyChange (tests/test_basic.py:186-325)
        a = do_nothing(case)
        cur = nodes_of_substation(case, obs, node)
        go = None
        if i == 1 and node != 7:
            go = topos[0]
        elif i == 2:
            go = zeros
        elif i == 3:
            go = topos[1]
        elif i == 4 and n_el > 2:
            go = zeros
        elif i == 5 and n_el > 2:
            go = topos[2]
        elif i == 6 and n_el >= 3:
            go = zeros
        elif i == 7 and n_el >= 4:
            go = topos[3]
        elif i == 8 and n_el > 3:
            go = zeros
        elif i == 9 and n_el > 4:
            go = topos[4]
        elif i == 10 and n_el > 4:
            go = zeros
        elif i == 11 and n_el > 5 and node != 2:   # "there is a Game Over if we apply the last topo for node 2"
            go = topos[5]
        elif i == 12 and n_el > 5 and node != 2:
            go = zeros
        if go is not None:
            set_substation_switches(case, a, node, differential(go, cur))
        return a

    dones, flags, ills = run_wrapped(g, policy, 13)
    assert flags == [FLAG_OK] * 13
    assert not any(dones)


def test_k5_too_many_productions_cut():
    """tests/test_core.py:128-165, 1019-1024: isolating the productions of substations 1 then 8 ->
    game over [F, T, F] with TooManyProductionsCut at i == 1 and a reset topology afterwards."""
    g = oracle_game('default14_for_tests')
    case = g.case

    def policy(i, obs):
        a = do_nothing(case)
        if i == 1:
            a[0] = 1                                   # production of substation 1 -> busbar 1 (alone)
        if i == 2:
            a[4] = 1                                   # production of substation 8
        if i == 3:
            assert list(obs['productions_nodes']) == [0] * 5
        return a

    dones, flags, ills = run_wrapped(g, policy, 3)
    assert dones == [False, True, False]
    assert flags[1] == FLAG_TOO_MANY_PRODS


def test_k6_are_cut_masks():
    """tests/test_core.py:432, 508: moving one production / load alone on busbar 1 marks it cut."""
    g = oracle_game('default14_for_tests')
    case = g.case
    g.process_game_over()
    a = do_nothing(case)
    a[1] = 1
    obs, flag, ill, done = g.step(a)
    assert not done and list(obs['are_productions_cut']) == [0, 1, 0, 0, 0]
    a = do_nothing(case)
    a[case.nP + 3] = 1
    obs, flag, ill, done = g.step(a)
    assert not done and int(obs['are_loads_cut'][3]) == 1 and int(np.sum(obs['are_loads_cut'])) == 1


def test_k8_node_split_persists():
    """tests/test_core.py:862-864, 906-909: a line cut / node split persists over the following do-nothing steps."""
    g = oracle_game('default14_for_tests')
    case = g.case

    def policy(i, obs):
        a = do_nothing(case)
        if i == 1:
            set_line_switch(case, a, 18, 1)
        else:
            assert [int(v) for v in obs['lines_status']] == [1] * 18 + [0, 1]
        return a

    dones, flags, ills = run_wrapped(g, policy, 9)
    assert not any(dones) and flags == [FLAG_OK] * 9


def test_k9_chronic_passthrough_and_slack():
    """tests/test_core.py:294-317, 352-372: realised injections equal the chronic rows (float32 values) and the
    slack production recomputed by pfsoln stays within 1e-3 relative of ... the chronic's own balance."""
    g = oracle_game('default14_for_tests')
    g.process_game_over()
    ch = g.chronic
    for row in (2, 3):
        obs, flag, ill, done = g.step(do_nothing(g.case))
        assert np.array_equal(obs['active_loads'], ch.loads_p[row].astype(np.float64))
        assert np.array_equal(obs['reactive_loads'], ch.loads_q[row].astype(np.float64))
        assert np.array_equal(obs['active_productions'][1:], ch.prods_p[row][1:].astype(np.float64))
        assert abs(obs['active_productions'][0] - ch.prods_p[row][0]) < 1e-3   # the slack production pfsoln recomputed: the reference's own bound (tests/test_core.py:352-372)


@pytest.mark.parametrize('solver', ['fdxb', 'newton'])
def test_k12_loss_error_three_steps(solver):
    """Agent_test_Loss_Error (tests/test_core.py:519-606, 1200-1230): the loss total the agent sees at steps 1..3 of
    default14_for_tests -- sum of realised productions minus sum of loads, the slack production being what pfsoln recomputed --
    within 1e-3 MW of the one the chronic rows imply.  The reference's own numeric known answer for the load-flow result."""
    from engine_checks import K12_EXPECTED_PRODS, K12_EXPECTED_LOADS, k12_expected_losses
    g = oracle_game('default14_for_tests', conf={'solver': solver})
    g.process_game_over()
    # (the numbers the reference's test spells out ARE rows 1..3 of the fixture chronic)
    for i in range(3):
        assert np.allclose(g.chronic.prods_p[i + 1], K12_EXPECTED_PRODS[i], rtol=0, atol=1e-5)
        assert np.allclose(g.chronic.loads_p[i + 1], K12_EXPECTED_LOADS[i], rtol=0, atol=1e-5)
    exp = k12_expected_losses()
    for i in range(3):
        obs = g.export_observation()
        loss = float(np.sum(obs['active_productions']) - np.sum(obs['active_loads']))
        assert abs(loss - exp[i]) < 1e-3, (i + 1, loss, exp[i])
        _, flag, _, done = g.step(do_nothing(g.case))
        assert flag == FLAG_OK and not done


def test_k1_style_rows_on_ieee118_recorded_from_the_reference():
    """K1's kind of known answer on IEEE-118 (tools/make_k1_rows.py): int-truncated ampere flows of all 186 lines over 60
    do-nothing steps of the reference's own RunEnv on default118 (fast-decoupled XB, shipped limits, 8 game overs and their
    restarts inside).  The recording ran on the numpy restatement of PYPOWER; `python tools/make_k1_rows.py --check` repeats it
    with the real PYPOWER where that is installed.  Here: the oracle game reproduces every row."""
    import os
    from helpers import load_env, ROOT
    ref = np.load(os.path.join(ROOT, 'tests', 'golden', 'k1_rows', 'default118_do_nothing_k1_rows.npz'))
    game = oracle_game('default118')
    case, _, _ = load_env('default118')
    for t in range(int(ref['steps'])):
        done = game.step(do_nothing(case))[3]
        assert bool(done) == bool(ref['done'][t]), t
        if done:
            game.process_game_over()
        assert np.array_equal(game.extract_flows_a().astype(np.int64), ref['int_amps'][t]), t
