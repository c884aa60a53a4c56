"""TEST INFRASTRUCTURE (oracle): numpy restatement of the reward the reference's shipped environments compute,
``CustomRewardSignal.compute_reward`` (parameters/default14/reward_signal.py:45-118; default118 ships the same class with
``constant = 118``), as called by ``RunEnv.step`` (pypownet/environment.py:866-874) with the flag ``Game.step`` returned.
Only tests import this module; the product computes the reward on the device (ppn_game.inc:game_reward)."""
import numpy as np

FLAG_OK, FLAG_DIVERGED, FLAG_TOO_MANY_LOADS, FLAG_TOO_MANY_PRODS = 0, 1, 2, 3


def coefficients(constant):
    """reward_signal.py:8-43."""
    c = float(constant)
    return dict(line_usage=-1., distance=-.02, loads_cut=-c / 5., prods_cut=-c / 10., loadflow_exception=-c,
                illegal_broken=-c / 100., illegal_cd_line=-c / 100., illegal_cd_sub=-c / 100.,
                too_many_prods=-c, too_many_loads=-c, too_much_activated=-5 * c, line_switch=-.2, node_switch=-.1)


def action_cost(k, n_node_switches, n_line_switches):
    """reward_signal.py:120-140 (__get_action_cost): switches of the action object as RunEnv holds it after Game.step --
    the illegal-action repair edits it in place (game.py:816-846), a wholly rejected action is zeroed in place
    (set_as_do_nothing, game.py:813, 191-198)."""
    return k['node_switch'] * n_node_switches + k['line_switch'] * n_line_switches


def compute_reward(k, flag, illegal_bits, illegal_counts, n_node_switches, n_line_switches, n_loads_cut, n_prods_cut,
                   topology_bits, ampere_flows, thermal_limits):
    """Returns the 5-list [loads cut, prods cut, action cost, distance to the initial topology, line usage]."""
    cost = action_cost(k, n_node_switches, n_line_switches)
    if flag == FLAG_TOO_MANY_LOADS:                  # reward_signal.py:90-91
        return [k['too_many_loads'], 0., 0., 0., 0.]
    if flag == FLAG_TOO_MANY_PRODS:                  # :88-89
        return [0., k['too_many_prods'], 0., 0., 0.]
    if flag != FLAG_OK:                              # DivergingLoadflowException, :49-50
        return [0., 0., cost, k['loadflow_exception'], 0.]
    usage = np.divide(ampere_flows, thermal_limits)  # :142-147
    r = [k['loads_cut'] * float(n_loads_cut), k['prods_cut'] * float(n_prods_cut), cost,
         k['distance'] * float(np.sum(np.asarray(topology_bits) != 0)),      # :149-169 (initial topology: every element on node 0)
         k['line_usage'] * float(np.sum(np.square(usage)))]                  # :106-111
    if illegal_bits & 1:                             # too many activated elements, :57-59
        r[2] += k['too_much_activated']
    elif illegal_bits:                               # :60-86
        r[2] += k['illegal_broken'] * illegal_counts[0] + k['illegal_cd_line'] * illegal_counts[1] + \
            k['illegal_cd_sub'] * illegal_counts[2]
    return r
