"""Gym-style public API, source compatible with ``pypownet.environment`` (reference environment.py:14-914):
``RunEnv`` (step / simulate / reset / process_game_over / get_observation / is_action_valid), ``ActionSpace``,
``ObservationSpace``, ``Observation`` (+ minimalist views, array <-> object round trip) and the exception wrappers.
The numerical work happens in the HIP engine (pypownet_amd/csrc); this module is host plumbing."""
from collections import OrderedDict
from copy import deepcopy
from enum import Enum

import numpy as np

from . import game as _game
from .spaces import MultiBinary, Box, Dict, Discrete


class IllegalActionException(_game.IllegalActionException):
    pass


class DivergingLoadflowException(_game.DivergingLoadflowException):
    pass


class TooManyProductionsCut(_game.TooManyProductionsCut):
    pass


class TooManyConsumptionsCut(_game.TooManyConsumptionsCut):
    pass


class ElementType(Enum):
    PRODUCTION = "production"
    CONSUMPTION = "consumption"
    ORIGIN_POWER_LINE = "origin of power line"
    EXTREMITY_POWER_LINE = "extremity of power line"


class ActionSpace(MultiBinary):
    """Action layout [prods | loads | lines_or | lines_ex | lines_status] (environment.py:46-112)."""

    def __init__(self, number_generators, number_consumers, number_power_lines, number_substations, substations_ids,
                 prods_subs_ids, loads_subs_ids, lines_or_subs_id, lines_ex_subs_id):
        self.prods_switches_subaction_length = number_generators
        self.loads_switches_subaction_length = number_consumers
        self.lines_or_switches_subaction_length = number_power_lines
        self.lines_ex_switches_subaction_length = number_power_lines
        self.lines_status_subaction_length = number_power_lines
        self.action_length = number_generators + number_consumers + 3 * number_power_lines
        super().__init__(self.action_length)
        self.substations_ids = substations_ids
        self.prods_subs_ids = prods_subs_ids
        self.loads_subs_ids = loads_subs_ids
        self.lines_or_subs_id = lines_or_subs_id
        self.lines_ex_subs_id = lines_ex_subs_id
        dn = self.get_do_nothing_action(as_class_Action=True)
        self._substations_n_elements = [len(self.get_substation_switches_in_action(dn, s)[1]) for s in substations_ids]

    def _new_action(self, parts):
        return _game.Action(parts[0], parts[1], parts[2], parts[3], parts[4], self.substations_ids, self.prods_subs_ids,
                            self.loads_subs_ids, self.lines_or_subs_id, self.lines_ex_subs_id, ElementType)

    def get_do_nothing_action(self, as_class_Action=False):
        a = self._new_action([np.zeros(n) for n in (self.prods_switches_subaction_length,
                                                    self.loads_switches_subaction_length,
                                                    self.lines_or_switches_subaction_length,
                                                    self.lines_ex_switches_subaction_length,
                                                    self.lines_status_subaction_length)])
        return a if as_class_Action else a.as_array()

    def array_to_action(self, array):
        if isinstance(array, _game.Action):
            return array
        if len(array) != self.action_length:
            raise ValueError('Expected action as a binary array of length %d, got %d' % (self.action_length, len(array)))
        array = np.asarray(array)
        cuts = np.cumsum([self.prods_switches_subaction_length, self.loads_switches_subaction_length,
                          self.lines_or_switches_subaction_length, self.lines_ex_switches_subaction_length])
        return self._new_action(np.split(array, cuts))

    def _verify_action_shape(self, action):
        if action is None:
            raise ValueError('Expected binary array of length %d, got None' % self.action_length)
        formatted = self.array_to_action(action) if not isinstance(action, _game.Action) else deepcopy(action)
        got = formatted.__len__(do_sum=False)
        exp = (self.prods_switches_subaction_length, self.loads_switches_subaction_length,
               self.lines_or_switches_subaction_length, self.lines_ex_switches_subaction_length)
        for name, g, e in zip(('prods_switches_subaction', 'loads_switches_subaction', 'lines_or_switches_subaction',
                               'lines_ex_subaction'), got, exp):
            if g and g != e:
                raise ValueError('Expected %s subaction of size %d, got %d' % (name, e, g))
        return formatted

    def get_number_elements_of_substation(self, substation_id):
        assert substation_id in self.substations_ids
        return self._substations_n_elements[np.where(self.substations_ids == substation_id)[0][0]]

    def get_substation_switches_in_action(self, action, substation_id, concatenated_output=True):
        action = self.array_to_action(action)
        assert substation_id in self.substations_ids, 'Substation with id %d does not exist' % substation_id
        return action.get_substation_switches(substation_id, concatenated_output)

    def set_substation_switches_in_action(self, action, substation_id, new_values):
        action = self.array_to_action(action)
        return action.set_substation_switches(substation_id, new_values)

    def get_lines_status_switches_of_substation(self, action, substation_id):
        assert substation_id in self.substations_ids, 'Substation with id %d does not exist' % substation_id
        mask = np.logical_or(self.lines_or_subs_id == substation_id, self.lines_ex_subs_id == substation_id)
        return action.lines_status_subaction[mask]

    def set_lines_status_switches_of_substation(self, action, substation_id, new_configuration):
        new_configuration = np.asarray(new_configuration)
        mask = np.logical_or(self.lines_or_subs_id == substation_id, self.lines_ex_subs_id == substation_id)
        assert mask.sum() == len(new_configuration), 'Expected configuration of size %d for substation %d, got %d' % (
            mask.sum(), substation_id, len(new_configuration))
        action.lines_status_subaction[mask] = new_configuration

    @staticmethod
    def get_lines_status_switch_from_id(action, line_id):
        return action.lines_status_subaction[line_id]

    @staticmethod
    def set_lines_status_switch_from_id(action, line_id, new_switch_value):
        action.lines_status_subaction[line_id] = new_switch_value


_MINIMALIST_FIELDS = ['active_loads', 'are_loads_cut', 'planned_active_loads', 'loads_nodes', 'active_productions',
                      'are_productions_cut', 'planned_active_productions', 'productions_nodes', 'lines_or_nodes',
                      'lines_ex_nodes', 'ampere_flows', 'lines_status', 'timesteps_before_lines_reconnectable',
                      'timesteps_before_lines_reactionable', 'timesteps_before_nodes_reactionable',
                      'timesteps_before_planned_maintenance', 'date_year', 'date_month', 'date_day', 'date_hour',
                      'date_minute', 'date_second']
_AC_FIELDS = ['reactive_loads', 'voltage_loads', 'reactive_productions', 'voltage_productions', 'active_flows_origin',
              'reactive_flows_origin', 'voltage_flows_origin', 'active_flows_extremity', 'reactive_flows_extremity',
              'voltage_flows_extremity', 'planned_reactive_loads', 'planned_voltage_productions']
_FULL_FIELDS = ['substations_ids', 'loads_substations_ids', 'productions_substations_ids', 'lines_or_substations_ids',
                'lines_ex_substations_ids', 'thermal_limits', 'initial_productions_nodes', 'initial_loads_nodes',
                'initial_lines_or_nodes', 'initial_lines_ex_nodes']


def _field_sizes(nP, nL, nl, nS):
    per = {'loads': nL, 'productions': nP}
    sizes = OrderedDict()
    for f in _MINIMALIST_FIELDS + _AC_FIELDS + _FULL_FIELDS:
        if f.startswith('date_'):
            sizes[f] = 1
        elif f in ('substations_ids', 'timesteps_before_nodes_reactionable'):
            sizes[f] = nS
        elif 'load' in f:
            sizes[f] = nL
        elif 'production' in f:
            sizes[f] = nP
        else:
            sizes[f] = nl
    return sizes


class ObservationSpace(Dict):
    def __init__(self, number_generators, number_consumers, number_power_lines, number_substations,
                 n_timesteps_horizon_maintenance):
        self.number_productions, self.number_loads = number_generators, number_consumers
        self.number_power_lines, self.number_substations = number_power_lines, number_substations
        self.n_timesteps_horizon_maintenance = n_timesteps_horizon_maintenance
        self.grid_number_of_elements = number_generators + number_consumers + 2 * number_power_lines
        sizes = _field_sizes(number_generators, number_consumers, number_power_lines, number_substations)
        binary = {'are_loads_cut', 'are_productions_cut', 'lines_status'}
        discrete = {'date_year': 3000, 'date_month': 12, 'date_day': 32, 'date_hour': 24, 'date_minute': 60,
                    'date_second': 60}

        def space(f):
            if f in discrete:
                return Discrete(discrete[f])
            if f in binary:
                return MultiBinary(sizes[f])
            return Box(-np.inf, np.inf, (sizes[f],), np.float32)
        minimalist = Dict(OrderedDict((f, space(f)) for f in _MINIMALIST_FIELDS))
        ac = OrderedDict([('MinimalistObservation', minimalist)] + [(f, space(f)) for f in _AC_FIELDS])
        full = OrderedDict([('MinimalistACObservation', Dict(ac))] + [(f, space(f)) for f in _FULL_FIELDS])
        super().__init__(full)
        self._sizes = sizes
        self.shape = tuple((1,) if f in discrete else (sizes[f],) for f in sizes)

    def array_to_observation(self, array):
        expected = sum(self._sizes.values())
        if len(array) != expected:
            raise ValueError('Expected observation array of length %d, got %d' % (expected, len(array)))
        return Observation._from_sizes(self._sizes, np.asarray(array))


class MinimalistObservation(object):
    def __init__(self, active_loads, active_productions, ampere_flows, lines_status, are_loads_cut,
                 are_productions_cut, timesteps_before_lines_reconnectable, timesteps_before_lines_reactionable,
                 timesteps_before_nodes_reactionable, timesteps_before_planned_maintenance, planned_active_loads,
                 planned_active_productions, date_year, date_month, date_day, date_hour, date_minute, date_second,
                 productions_nodes, loads_nodes, lines_or_nodes, lines_ex_nodes):
        loc = dict(locals())
        loc.pop('self')
        self.__dict__.update(loc)

    def _concat(self, fields):
        return np.concatenate([np.atleast_1d(np.asarray(getattr(self, f), dtype=np.float64)).flatten() for f in fields])

    def as_array(self):
        return self._concat(_MINIMALIST_FIELDS)

    @staticmethod
    def __keys__():
        return ['active_loads', 'are_loads_cut', 'loads_nodes', 'active_productions', 'are_productions_cut',
                'productions_nodes', 'lines_or_nodes', 'lines_ex_nodes', 'ampere_flows', 'lines_status',
                'timesteps_before_lines_reconnectable', 'timesteps_before_lines_reactionable',
                'timesteps_before_nodes_reactionable', 'timesteps_before_planned_maintenance', 'planned_active_loads',
                'planned_active_productions', 'datetime']

    def as_dict(self):
        return {k: v for k, v in self.__dict__.items() if k in MinimalistObservation.__keys__()}


class MinimalistACObservation(MinimalistObservation):
    def __init__(self, active_loads, reactive_loads, voltage_loads, active_productions, reactive_productions,
                 voltage_productions, active_flows_origin, reactive_flows_origin, voltage_flows_origin,
                 active_flows_extremity, reactive_flows_extremity, voltage_flows_extremity, ampere_flows, lines_status,
                 are_loads_cut, are_productions_cut, timesteps_before_lines_reconnectable,
                 timesteps_before_lines_reactionable, timesteps_before_nodes_reactionable,
                 timesteps_before_planned_maintenance, planned_active_loads, planned_reactive_loads,
                 planned_active_productions, planned_voltage_productions, date_year, date_month, date_day, date_hour,
                 date_minute, date_second, productions_nodes, loads_nodes, lines_or_nodes, lines_ex_nodes):
        super().__init__(active_loads, active_productions, ampere_flows, lines_status, are_loads_cut,
                         are_productions_cut, timesteps_before_lines_reconnectable, timesteps_before_lines_reactionable,
                         timesteps_before_nodes_reactionable, timesteps_before_planned_maintenance, planned_active_loads,
                         planned_active_productions, date_year, date_month, date_day, date_hour, date_minute,
                         date_second, productions_nodes, loads_nodes, lines_or_nodes, lines_ex_nodes)
        for f in _AC_FIELDS:
            setattr(self, f, locals()[f])

    def as_array(self):
        return self._concat(_MINIMALIST_FIELDS + _AC_FIELDS)

    @staticmethod
    def __keys__():
        return list(_AC_FIELDS)

    def as_dict(self):
        keys = MinimalistACObservation.__keys__() + MinimalistObservation.__keys__()
        return {k: v for k, v in self.__dict__.items() if k in keys}

    def as_minimalist(self):
        return super(MinimalistACObservation, self)


class Observation(MinimalistACObservation):
    def __init__(self, substations_ids, active_loads, reactive_loads, voltage_loads, active_productions,
                 reactive_productions, voltage_productions, active_flows_origin, reactive_flows_origin,
                 voltage_flows_origin, active_flows_extremity, reactive_flows_extremity, voltage_flows_extremity,
                 ampere_flows, thermal_limits, lines_status, are_loads_cut, are_productions_cut,
                 loads_substations_ids, productions_substations_ids, lines_or_substations_ids, lines_ex_substations_ids,
                 timesteps_before_lines_reconnectable, timesteps_before_lines_reactionable,
                 timesteps_before_nodes_reactionable, timesteps_before_planned_maintenance, planned_active_loads,
                 planned_reactive_loads, planned_active_productions, planned_voltage_productions, date_year,
                 date_month, date_day, date_hour, date_minute, date_second, productions_nodes,
                 loads_nodes, lines_or_nodes, lines_ex_nodes, initial_productions_nodes, initial_loads_nodes,
                 initial_lines_or_nodes, initial_lines_ex_nodes):
        super().__init__(active_loads, reactive_loads, voltage_loads, active_productions, reactive_productions,
                         voltage_productions, active_flows_origin, reactive_flows_origin, voltage_flows_origin,
                         active_flows_extremity, reactive_flows_extremity, voltage_flows_extremity, ampere_flows,
                         lines_status, are_loads_cut, are_productions_cut, timesteps_before_lines_reconnectable,
                         timesteps_before_lines_reactionable, timesteps_before_nodes_reactionable,
                         timesteps_before_planned_maintenance, planned_active_loads, planned_reactive_loads,
                         planned_active_productions, planned_voltage_productions, date_year, date_month, date_day,
                         date_hour, date_minute, date_second, productions_nodes, loads_nodes, lines_or_nodes,
                         lines_ex_nodes)
        for f in _FULL_FIELDS:
            setattr(self, f, locals()[f])

    @classmethod
    def _from_sizes(cls, sizes, array):
        kw, o = {}, 0
        for f, n in sizes.items():
            kw[f] = array[o:o + n]
            o += n
        return cls(**kw)

    _INT_FIELDS = ('lines_status', 'are_loads_cut', 'are_productions_cut', 'substations_ids', 'loads_substations_ids',
                   'productions_substations_ids', 'lines_or_substations_ids', 'lines_ex_substations_ids',
                   'productions_nodes', 'loads_nodes', 'lines_or_nodes', 'lines_ex_nodes')
    _DATE_FIELDS = ('date_year', 'date_month', 'date_day', 'date_hour', 'date_minute', 'date_second')

    @classmethod
    def from_array(cls, case, array, typed=False):
        """typed=False: plain slices of the array, what the reference's ObservationSpace.array_to_observation builds
        (environment.py:376-403).  typed=True: the types Game.export_observation hands out (grid.py:496-566,
        game.py:945-978): integer ids / status / cut masks / node bits, Python ints for the date fields."""
        o = cls._from_sizes(_field_sizes(case.nP, case.nL, case.nl, case.nS), np.asarray(array))
        if typed:
            for f in cls._INT_FIELDS:
                setattr(o, f, np.asarray(getattr(o, f)).astype(int))
            for f in cls._DATE_FIELDS:
                setattr(o, f, int(np.asarray(getattr(o, f)).reshape(-1)[0]))
        return o

    def as_dict(self):
        return self.__dict__

    def as_array(self):
        return self._concat(_MINIMALIST_FIELDS + _AC_FIELDS + _FULL_FIELDS)

    def as_ac_minimalist(self):
        return super(Observation, self)

    def as_minimalist(self):
        return super(Observation, self).as_minimalist()

    def _of_substation(self, substation_id, groups):
        assert substation_id in self.substations_ids, \
            'Substation with id {} does not exist; available substations: {}'.format(substation_id, self.substations_ids)
        values, types = [], []
        for arr, ids, t in groups:
            v = np.asarray(arr)[np.asarray(ids) == substation_id]
            values.append(v)
            types.extend([t] * len(v))
        return np.concatenate(values), types

    def get_nodes_of_substation(self, substation_id):
        return self._of_substation(substation_id, (
            (self.productions_nodes, self.productions_substations_ids, ElementType.PRODUCTION),
            (self.loads_nodes, self.loads_substations_ids, ElementType.CONSUMPTION),
            (self.lines_or_nodes, self.lines_or_substations_ids, ElementType.ORIGIN_POWER_LINE),
            (self.lines_ex_nodes, self.lines_ex_substations_ids, ElementType.EXTREMITY_POWER_LINE)))

    def get_lines_status_of_substation(self, substation_id):
        assert substation_id in self.substations_ids, \
            'Substation with id {} does not exist; available substations: {}'.format(substation_id, self.substations_ids)
        ori = np.asarray(self.lines_or_substations_ids) == substation_id
        ext = np.asarray(self.lines_ex_substations_ids) == substation_id
        concerned = np.logical_or(ori, ext)
        other = [int(self.lines_ex_substations_ids[i]) if ori[i] else int(self.lines_or_substations_ids[i])
                 for i in np.where(concerned)[0]]
        return np.asarray(self.lines_status)[concerned], other

    def get_lines_capacity_usage(self):
        return np.divide(self.ampere_flows, self.thermal_limits)

    def __str__(self):
        head = 'date: %d of %d of %d at %dh%dm%ds' % (self.date_year, self.date_month, self.date_day, self.date_hour,
                                                       self.date_minute, self.date_second)
        rows = ['  line %3d: sub %3d(n%d) -> sub %3d(n%d) on=%d  P=%8.1f Q=%7.1f  I=%8.1f / %6.0f A' % (
            k, self.lines_or_substations_ids[k], self.lines_or_nodes[k], self.lines_ex_substations_ids[k],
            self.lines_ex_nodes[k], self.lines_status[k], self.active_flows_origin[k], self.reactive_flows_origin[k],
            self.ampere_flows[k], self.thermal_limits[k]) for k in range(len(self.lines_status))]
        return '\n'.join([head] + rows)


class RunEnv(object):
    def __init__(self, parameters_folder, game_level, chronic_looping_mode='natural', start_id=0,
                 game_over_mode='soft', renderer_latency=None, without_overflow_cutoff=False, seed=None, device=0,
                 config_overrides=None):
        self.parameters_folder = parameters_folder
        self.game_level = game_level
        self.chronic_looping_mode = chronic_looping_mode
        self.start_id = start_id
        self.game_over_mode = game_over_mode
        self.renderer_latency = renderer_latency
        self.without_overflow_cutoff = without_overflow_cutoff
        self._extra = dict(device=device, config_overrides=config_overrides)
        self.game = None
        self.action_space = None
        self.observation_space = None
        self.reward_signal = None
        self.last_rewards = None
        if seed is not None:
            np.random.seed(seed)
        self.reset()

    def reset(self):
        if self.game is not None:
            self.game.engine.close()
        self.game = _game.Game(parameters_folder=self.parameters_folder, game_level=self.game_level,
                               chronic_looping_mode=self.chronic_looping_mode, chronic_starting_id=self.start_id,
                               game_over_mode=self.game_over_mode, renderer_frame_latency=self.renderer_latency,
                               without_overflow_cutoff=self.without_overflow_cutoff, **self._extra)
        g = self.game
        self.action_space = ActionSpace(*g.get_number_elements(), substations_ids=g.get_substations_ids(),
                                        prods_subs_ids=g.get_substations_ids_prods(),
                                        loads_subs_ids=g.get_substations_ids_loads(),
                                        lines_or_subs_id=g.get_substations_ids_lines_or(),
                                        lines_ex_subs_id=g.get_substations_ids_lines_ex())
        n_prods, n_loads, n_lines, n_substations = g.get_number_elements()
        self.observation_space = ObservationSpace(n_prods, n_loads, n_lines, n_substations,
                                                  g.n_timesteps_horizon_maintenance)
        self.reward_signal = g.get_reward_signal_class()
        self.last_rewards = []
        return self.get_observation(True)

    def get_observation(self, as_array=True):
        obs = self.game.export_observation()
        return obs.as_array() if as_array else obs

    def _get_obs(self):
        return self.get_observation(False)

    def is_action_valid(self, action):
        return self.game.is_action_valid(self.action_space.array_to_action(action))

    def _play(self, action, do_sum, simulate):
        submitted = self.action_space._verify_action_shape(action)
        observation, flag, done = self.game.simulate(submitted) if simulate else self.game.step(submitted)
        flag = self.__wrap_exception(flag)
        reward_aslist = self.reward_signal.compute_reward(observation=observation, action=submitted, flag=flag)
        self.last_rewards = reward_aslist
        return (observation.as_array() if observation is not None else observation,
                sum(reward_aslist) if do_sum else reward_aslist, done, flag)

    def step(self, action, do_sum=True):
        return self._play(action, do_sum, False)

    def simulate(self, action, do_sum=True):
        return self._play(action, do_sum, True)

    def process_game_over(self):
        self.game.process_game_over()
        return self.get_observation()

    def render(self, game_over=False):
        raise NotImplementedError('the pygame renderer of the reference is out of scope (SURVEY.md 2, item 13)')

    @staticmethod
    def __wrap_exception(flag):
        if isinstance(flag, _game.DivergingLoadflowException):
            return DivergingLoadflowException(flag.last_observation, flag.text)
        if isinstance(flag, _game.TooManyConsumptionsCut):
            return TooManyConsumptionsCut(flag.text)
        if isinstance(flag, _game.TooManyProductionsCut):
            return TooManyProductionsCut(flag.text)
        if isinstance(flag, _game.IllegalActionException):
            return IllegalActionException(flag.text, flag.get_has_too_much_activations(),
                                          flag.get_illegal_broken_lines_reconnections(),
                                          flag.get_illegal_oncoolown_lines_switches(),
                                          flag.get_illegal_oncoolown_substations_switches())
        return flag

    def get_current_chronic_name(self):
        return self.game.get_current_chronic_name()

    def get_current_datetime(self):
        return self.game.get_current_datetime()
