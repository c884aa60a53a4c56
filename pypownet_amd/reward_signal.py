"""Base reward signal (reference: pypownet/reward_signal.py): a template returning ``[0.]``."""


class RewardSignal(object):
    def __init__(self):
        pass

    def compute_reward(self, observation, action, flag):
        return [0.]


class DefaultGridRewardSignal(RewardSignal):
    """Five-component reward used by the reference's shipped environments (parameters/default14/reward_signal.py:
    45-118; default118 uses constant=118): [load cut, prod cut, action cost, distance to the reference topology,
    -sum((ampere / limit)^2)], with fixed penalties on game-over / illegal-action flags.  Environments that ship
    their own ``reward_signal.py`` override it (pypownet/parameters.py:55-70)."""

    def __init__(self, constant=14):
        super().__init__()
        c = float(constant)
        self.k_line_usage, self.k_distance = -1., -.02
        self.k_loads_cut, self.k_prods_cut = -c / 5., -c / 10.
        self.loadflow_exception_reward = -c
        self.k_illegal = -c / 100.
        self.too_many_productions_cut = self.too_many_consumptions_cut = -c
        self.too_much_activated_elements = -5 * c
        self.k_line_switch, self.k_node_switch = -.2, -.1

    def as_engine_params(self):
        """The same coefficients in the layout of ``ppn_reward_params`` (include/ppn.h): Engine.set_reward."""
        return {'line_usage': self.k_line_usage, 'distance_initial_grid': self.k_distance,
                'number_loads_cut': self.k_loads_cut, 'number_prods_cut': self.k_prods_cut,
                'loadflow_exception': self.loadflow_exception_reward,
                'illegal_broken_line_switch': self.k_illegal, 'illegal_oncooldown_line_switch': self.k_illegal,
                'illegal_oncooldown_substation_switch': self.k_illegal,
                'too_many_productions_cut': self.too_many_productions_cut,
                'too_many_consumptions_cut': self.too_many_consumptions_cut,
                'too_much_activated_elements': self.too_much_activated_elements,
                'number_line_switches': self.k_line_switch, 'number_node_switches': self.k_node_switch}

    def _action_cost(self, action):
        import numpy as np
        return self.k_node_switch * float(np.sum(action.get_node_splitting_subaction())) + \
            self.k_line_switch * float(np.sum(action.get_lines_status_subaction()))

    def compute_reward(self, observation, action, flag):
        import numpy as np
        from . import game
        if flag is None:
            usage = np.divide(observation.ampere_flows, observation.thermal_limits)
            cur = np.concatenate((observation.productions_nodes, observation.loads_nodes, observation.lines_or_nodes,
                                  observation.lines_ex_nodes))
            ini = np.concatenate((observation.initial_productions_nodes, observation.initial_loads_nodes,
                                  observation.initial_lines_or_nodes, observation.initial_lines_ex_nodes))
            return [self.k_loads_cut * float(np.sum(observation.are_loads_cut)),
                    self.k_prods_cut * float(np.sum(observation.are_productions_cut)),
                    self._action_cost(action), self.k_distance * float(np.sum(ini != cur)),
                    self.k_line_usage * float(np.sum(np.square(usage)))]
        if isinstance(flag, game.DivergingLoadflowException):
            return [0., 0., self._action_cost(action), self.loadflow_exception_reward, 0.]
        if isinstance(flag, game.IllegalActionException):
            r = self.compute_reward(observation, action, None)
            if flag.get_has_too_much_activations():
                r[2] += self.too_much_activated_elements
            else:
                n = 0
                for m in (flag.get_illegal_broken_lines_reconnections(), flag.get_illegal_oncoolown_lines_switches(),
                          flag.get_illegal_oncoolown_substations_switches()):
                    if m is not None:
                        n += int(np.sum(m))
                r[2] += self.k_illegal * n
            return r
        if isinstance(flag, game.TooManyProductionsCut):
            return [0., self.too_many_productions_cut, 0., 0., 0.]
        if isinstance(flag, game.TooManyConsumptionsCut):
            return [self.too_many_consumptions_cut, 0., 0., 0., 0.]
        raise flag
