"""Game: single-environment host-side mirror of ``pypownet.game.Game`` on top of the batched HIP engine.

All rules (action legality and repair, timestep loading, cascading failure, game over) run on the GPU inside
``ppn_step``; this class only (a) keeps the reference's Python objects alive for agents -- ``Action``, the four
exception classes with their payloads, ``Observation`` -- and (b) maps the engine's integer flags back to them.
Reference: pypownet/game.py:21-252 (exceptions, Action), :254-340 (Game.__init__), :799-943 (step/simulate),
:762-797 (process_game_over), :945-978 (export_observation), :1088-1100 (get_changed_substations).
"""
import logging

import numpy as np

from .case import Case
from .chronic import Chronic, ChronicLooper, load_chronic
from .engine import (Engine, FLAG_DIVERGED, FLAG_TOO_MANY_LOADS, FLAG_TOO_MANY_PRODS, FLAG_ENGINE_CAPACITY,
                     ILL_TOO_MANY, ILL_BROKEN_LINE, ILL_LINE_COOLDOWN, ILL_NODE_COOLDOWN, SOLVE_NOT_CONNEXE)
from .parameters import Parameters


class NoMoreScenarios(Exception):
    pass


class IllegalActionException(Exception):
    def __init__(self, text, has_too_much_activations, illegal_lines_reconnections, illegal_unavailable_lines_switches,
                 illegal_oncoolown_substations_switches, *args):
        super(IllegalActionException, self).__init__(*args)
        self.text = text
        self.has_too_much_activations = has_too_much_activations
        self.illegal_broken_lines_reconnections = illegal_lines_reconnections
        self.illegal_oncooldown_lines_switches = illegal_unavailable_lines_switches
        self.illegal_oncoolown_substations_switches = illegal_oncoolown_substations_switches

    def get_has_too_much_activations(self): return self.has_too_much_activations
    def get_illegal_broken_lines_reconnections(self): return self.illegal_broken_lines_reconnections
    def get_illegal_oncoolown_lines_switches(self): return self.illegal_oncooldown_lines_switches
    def get_illegal_oncoolown_substations_switches(self): return self.illegal_oncoolown_substations_switches

    @property
    def is_empty(self):
        return self.has_too_much_activations is False and self.illegal_broken_lines_reconnections is None \
            and self.illegal_oncooldown_lines_switches is None and self.illegal_oncoolown_substations_switches is None


class DivergingLoadflowException(Exception):
    def __init__(self, last_observation, *args):
        super(DivergingLoadflowException, self).__init__(last_observation, *args)
        self.last_observation = last_observation
        self.text = args[0] if args else ''


class TooManyProductionsCut(Exception):
    def __init__(self, *args):
        super(TooManyProductionsCut, self).__init__(*args)
        self.text = args[0]


class TooManyConsumptionsCut(Exception):
    def __init__(self, *args):
        super(TooManyConsumptionsCut, self).__init__(*args)
        self.text = args[0]


class Action(object):
    """Five binary sub-vectors of switches (reference pypownet/game.py:74-252)."""

    def __init__(self, prods_switches_subaction, loads_switches_subaction, lines_or_switches_subaction,
                 lines_ex_switches_subaction, lines_status_subaction, substations_ids, prods_subs_ids, loads_subs_ids,
                 lines_or_subs_id, lines_ex_subs_id, elementtype):
        for name, v in (('prods_switches_subaction', prods_switches_subaction),
                        ('loads_switches_subaction', loads_switches_subaction),
                        ('lines_or_switches_subaction', lines_or_switches_subaction),
                        ('lines_ex_switches_subaction', lines_ex_switches_subaction),
                        ('lines_status_subaction', lines_status_subaction)):
            if v is None:
                raise ValueError('Expected %s to be array, got None' % name)
        self.prods_switches_subaction = np.asarray(prods_switches_subaction).astype(int)
        self.loads_switches_subaction = np.asarray(loads_switches_subaction).astype(int)
        self.lines_or_switches_subaction = np.asarray(lines_or_switches_subaction).astype(int)
        self.lines_ex_switches_subaction = np.asarray(lines_ex_switches_subaction).astype(int)
        self.lines_status_subaction = np.asarray(lines_status_subaction).astype(int)
        self._lengths = (len(self.prods_switches_subaction), len(self.loads_switches_subaction),
                         len(self.lines_or_switches_subaction), len(self.lines_ex_switches_subaction),
                         len(self.lines_status_subaction))
        self.substations_ids = substations_ids
        self.prods_subs_ids = prods_subs_ids
        self.loads_subs_ids = loads_subs_ids
        self.lines_or_subs_id = lines_or_subs_id
        self.lines_ex_subs_id = lines_ex_subs_id
        self.elementtype = elementtype

    def get_prods_switches_subaction(self): return self.prods_switches_subaction
    def get_loads_switches_subaction(self): return self.loads_switches_subaction
    def get_lines_or_switches_subaction(self): return self.lines_or_switches_subaction
    def get_lines_ex_switches_subaction(self): return self.lines_ex_switches_subaction
    def get_lines_status_subaction(self): return self.lines_status_subaction

    def get_node_splitting_subaction(self):
        return np.concatenate((self.prods_switches_subaction, self.loads_switches_subaction,
                               self.lines_or_switches_subaction, self.lines_ex_switches_subaction))

    def set_node_splitting_subaction(self, new):
        assert len(new) == sum(self._lengths[:4])
        o = 0
        parts = []
        for n in self._lengths[:4]:
            parts.append(np.asarray(new[o:o + n]).astype(int))
            o += n
        (self.prods_switches_subaction, self.loads_switches_subaction, self.lines_or_switches_subaction,
         self.lines_ex_switches_subaction) = parts

    def _sub_parts(self, substation_id):
        return ((self.prods_switches_subaction, self.prods_subs_ids, self.elementtype.PRODUCTION),
                (self.loads_switches_subaction, self.loads_subs_ids, self.elementtype.CONSUMPTION),
                (self.lines_or_switches_subaction, self.lines_or_subs_id, self.elementtype.ORIGIN_POWER_LINE),
                (self.lines_ex_switches_subaction, self.lines_ex_subs_id, self.elementtype.EXTREMITY_POWER_LINE))

    def get_substation_switches(self, substation_id, concatenated_output=True):
        assert substation_id in self.substations_ids, 'Substation with id %d does not exist' % substation_id
        values, types = [], []
        for arr, ids, t in self._sub_parts(substation_id):
            v = arr[np.asarray(ids) == substation_id]
            values.append(v)
            types.extend([t] * len(v))
        return (np.concatenate(values) if concatenated_output else tuple(values)), np.asarray(types)

    def set_substation_switches(self, substation_id, new_values):
        new_values = np.asarray(new_values)
        _, types = self.get_substation_switches(substation_id, concatenated_output=False)
        assert len(types) == len(new_values), 'Expected new_values of size %d for substation %d, got size %d' % (
            len(types), substation_id, len(new_values))
        for arr, ids, t in self._sub_parts(substation_id):
            arr[np.asarray(ids) == substation_id] = new_values[types == t]
        return self

    def set_as_do_nothing(self):
        for name in ('prods_switches_subaction', 'loads_switches_subaction', 'lines_or_switches_subaction',
                     'lines_ex_switches_subaction', 'lines_status_subaction'):
            setattr(self, name, np.zeros(len(getattr(self, name))).astype(int))
        return self

    def as_array(self):
        return np.concatenate((self.get_node_splitting_subaction(), self.lines_status_subaction))

    def __str__(self):
        return self.as_array().__str__()

    def __len__(self, do_sum=True):
        return sum(self._lengths) if do_sum else self._lengths

    def _locate(self, item):
        item %= len(self)
        for arr in (self.prods_switches_subaction, self.loads_switches_subaction, self.lines_or_switches_subaction,
                    self.lines_ex_switches_subaction, self.lines_status_subaction):
            if item < len(arr):
                return arr, item
            item -= len(arr)
        raise IndexError(item)

    def __setitem__(self, item, value):
        arr, k = self._locate(item)
        arr[k] = value

    def __getitem__(self, item):
        arr, k = self._locate(item)
        return arr[k]


class _TopologyView(object):
    """What agents read from ``game.grid``: ``get_topology()``-like access to mapping tables."""

    def __init__(self, case):
        self.mapping_array = case.mapping_array
        self._perm = case.zip_perm

    def mapping_permutation(self, array):
        a = np.asarray(array)
        return [int(a[c]) for c in self._perm]


class Game(object):
    def __init__(self, parameters_folder, game_level, chronic_looping_mode, chronic_starting_id, game_over_mode,
                 renderer_frame_latency=None, without_overflow_cutoff=False, device=0, config_overrides=None):
        self.logger = logging.getLogger('pypownet.' + __name__)
        self._parameters = Parameters(parameters_folder, game_level, overrides=config_overrides)
        conf = self._parameters.simulator_configuration
        self.is_mode_dc = self._parameters.is_dc_mode()
        p = self._parameters
        self.hard_overflow_coefficient = 1e9 if without_overflow_cutoff else p.get_hard_overflow_coefficient()
        self.n_timesteps_hard_overflow_is_broken = p.get_n_timesteps_hard_overflow_is_broken()
        self.n_timesteps_soft_overflow_is_broken = p.get_n_timesteps_soft_overflow_is_broken()
        self.n_timesteps_consecutive_soft_overflow_breaks = 1e12 if without_overflow_cutoff else \
            p.get_n_timesteps_consecutive_soft_overflow_breaks()
        self.n_timesteps_horizon_maintenance = p.get_n_timesteps_horizon_maintenance()
        self.max_number_prods_game_over = p.get_max_number_prods_game_over()
        self.max_number_loads_game_over = p.get_max_number_loads_game_over()
        self.n_timesteps_actionned_line_reactionable = p.get_n_timesteps_actionned_line_reactionable()
        self.n_timesteps_actionned_node_reactionable = p.get_n_timesteps_actionned_node_reactionable()
        self.max_number_actionned_substations = p.get_max_number_actionned_substations()
        self.max_number_actionned_lines = p.get_max_number_actionned_lines()
        self.max_number_actionned_total = p.get_max_number_actionned_total()
        self.game_over_mode = game_over_mode

        looper = ChronicLooper(p.get_chronics_path(), game_level, chronic_starting_id, chronic_looping_mode)
        # the engine holds every chronic of the level; slot order = play order starting at the first chronic played.
        # 'random' (chronic.py:266-291): the first chronic is drawn on the host like the reference does (np.random, so
        # RunEnv(seed=...) controls it), the later ones on the device from a counter-based generator seeded from the same
        # host stream (include/ppn.h, PPN_LOOP_RANDOM)
        n = len(looper.chronics)
        first = looper.next_chronic_id
        order = [first] if chronic_looping_mode == 'fixed' else [(first + k) % n for k in range(n)]
        self._chronics = [load_chronic(looper.chronics[k]) for k in order]
        self.case = Case.from_file(p.get_reference_grid_path())
        self.engine = Engine(self.case, conf, 1, device=device, chronics=self._chronics,
                             without_overflow_cutoff=without_overflow_cutoff, game_over_mode=game_over_mode,
                             looping_mode=chronic_looping_mode,
                             rng_seed=int(np.random.randint(0, 2 ** 31 - 1)) if chronic_looping_mode == 'random' else 0)
        self.substations_ids = self.case.sub_ids.astype(float)
        self.grid = self            # agents reach game.grid.get_topology().mapping_array etc.
        self.number_elements_per_substations = list(self.case.n_elements)
        self.n_nodes = 2 * self.case.nS
        self.n_prods, self.n_loads, self.n_lines = self.case.nP, self.case.nL, self.case.nl
        self.epoch = 1
        self.timestep = 1
        self.renderer = None
        self.last_action = None
        self.get_reward_signal_class = p.get_reward_signal_class()
        self.engine.reset()
        self._sync_done()

    # ---- reference accessors -------------------------------------------------------------------------
    def get_topology(self): return _TopologyView(self.case)
    def get_max_seconds_per_timestep(self): return self._parameters.get_max_seconds_per_timestep()
    def get_number_elements(self): return self.case.nP, self.case.nL, self.case.nl, self.case.nS
    def get_substations_ids(self): return self.substations_ids
    def get_substations_ids_prods(self): return self.case.sub_ids[self.case.gen_sub].astype(int)
    def get_substations_ids_loads(self): return self.case.sub_ids[self.case.load_sub].astype(int)
    def get_substations_ids_lines_or(self): return self.case.sub_ids[self.case.or_sub].astype(int)
    def get_substations_ids_lines_ex(self): return self.case.sub_ids[self.case.ex_sub].astype(int)
    def get_initial_topology(self): return (np.zeros(self.case.nP), np.zeros(self.case.nL), np.zeros(self.case.nl), np.zeros(self.case.nl))
    def parameters_environment_tostring(self): return self._parameters.__str__()

    def _current_chronic(self):
        return self._chronics[int(self.engine.read('CHRONIC_SLOT')[0])]

    def get_current_chronic_name(self): return self._current_chronic().name

    def get_current_timestep_id(self):
        return self._current_chronic().timestep_ids[int(self.engine.read('CHRONIC_ROW')[0])]

    def get_current_datetime(self):
        ch = self._current_chronic()
        return ch.get_timestep_entries(ch.timestep_ids[int(self.engine.read('CHRONIC_ROW')[0])]).get_datetime()

    def _sync_done(self):
        if bool(self.engine.read('DONE')[0]):     # initial state diverged: same recovery as the reference
            self.process_game_over()

    # ---- observation ------------------------------------------------------------------------------------
    def export_observation(self, simulation=False):
        from .environment import Observation
        return Observation.from_array(self.case, self.engine.observations(simulation=simulation)[0], typed=True)

    # ---- actions ----------------------------------------------------------------------------------------
    def get_changed_substations(self, action):
        topo = np.asarray(action.get_node_splitting_subaction()) != 0
        changed = np.zeros(self.case.nS, dtype=bool)
        np.logical_or.at(changed, self.case.elem_sub, topo)
        return changed

    def is_action_valid(self, action):
        if action is None:
            return False
        return bool(self.engine.is_action_valid(np.asarray(action.as_array())[None, :])[0])

    def _illegal_exception(self, action, bits, counters):
        """Rebuild the IllegalActionException payload (masks + text) and apply the reference's in-place repair of
        the Action object (game.py:809-846) -- the reward signal sees the repaired action."""
        rec, lcd, ncd = counters
        lines = np.asarray(action.get_lines_status_subaction()) == 1
        subs = self.get_changed_substations(action)
        if bits & ILL_TOO_MANY:
            e = IllegalActionException(
                'Action has too much activations simultaneously: {}/{} activated substations, {}/{} switched lines and '
                '{}/{} total switched elements (substations and lines).'.format(
                    int(subs.sum()), self.max_number_actionned_substations, int(lines.sum()),
                    self.max_number_actionned_lines, int(subs.sum() + lines.sum()), self.max_number_actionned_total),
                True, None, None, None)
            action.set_as_do_nothing()
            return e
        broken = np.logical_and(lines, rec > 0)
        line_cd = np.logical_and(lines, lcd > 0)
        node_cd = np.logical_and(subs, ncd > 0)
        text = ''
        if broken.any():
            text += 'Trying to reconnect broken/on-maintenance line%s %s, must wait %s timesteps.' % (
                's' if broken.sum() > 1 else '', ', '.join(map(str, np.where(broken)[0])),
                ('resp. ' if broken.sum() > 1 else '') + ', '.join(str(int(x)) for x in rec[broken]))
        if line_cd.any():
            text += 'Trying to action on-cooldown line%s %s, must wait resp. %s timesteps. ' % (
                's' if line_cd.sum() > 1 else '', ', '.join(map(str, np.where(line_cd)[0])),
                ('resp. ' if line_cd.sum() > 1 else '') + ', '.join(str(int(x)) for x in lcd[line_cd]))     # (sic, game.py:717-723)
        if node_cd.any():
            text += 'Trying to action on-cooldown substation%s %s, must wait resp. %s timesteps.' % (
                's' if node_cd.sum() > 1 else '', ', '.join(map(str, np.where(node_cd)[0])),
                ('resp. ' if node_cd.sum() > 1 else '') + ', '.join(str(int(x)) for x in ncd[node_cd]))
        e = IllegalActionException(text, False, broken if broken.any() else None, line_cd if line_cd.any() else None,
                                   node_cd if node_cd.any() else None)
        # the repair of Game.step (game.py:816-846) edits the Action object in place and extends the text
        if broken.any():
            action.lines_status_subaction[broken] = 0
            e.text += ' Ignoring action switches of broken/on-maintenance lines: %s.' % ', '.join(map(str, np.where(broken)[0]))
        if line_cd.any():
            action.lines_status_subaction[line_cd] = 0
            e.text += ' Ignoring action switches of on-cooldown lines: %s.' % ', '.join(map(str, np.where(line_cd)[0]))
        if node_cd.any():
            changed = self.substations_ids[node_cd]
            for sid in self.case.sub_ids[node_cd]:
                n_el = len(action.get_substation_switches(sid, False)[1])
                action.set_substation_switches(sid, np.zeros(n_el))
            # (the reference prints np.where() of the id list itself, game.py:845-846)
            e.text += ' Ignoring node switches of on-cooldown substations: %s.' % ', '.join(map(str, np.where(changed)[0]))
        return e

    def _flag_object(self, flag, sim=False):
        if flag == FLAG_DIVERGED:
            # grid.py:231, 238 ('The grid is not connexe': the solver raised) vs grid.py:264 ('Power grid outage': no
            # convergence or NaN), extended by the cascade (game.py:515)
            outcome = int(self.engine.read('SOLVE_OUTCOME', simulation=sim)[0])
            return DivergingLoadflowException(None, '%s: cascading emulation of depth %d has diverged' % (
                'The grid is not connexe' if outcome == SOLVE_NOT_CONNEXE else 'Power grid outage',
                int(self.engine.read('CASCADE_DEPTH', simulation=sim)[0])))
        if flag == FLAG_TOO_MANY_LOADS:
            return TooManyConsumptionsCut('There are %d isolated loads; at most %d tolerated' % (
                int(self.engine.read('N_LOADS_CUT')[0]), self.max_number_loads_game_over))
        if flag == FLAG_TOO_MANY_PRODS:
            return TooManyProductionsCut('There are %d isolated productions; at most %d tolerated' % (
                int(self.engine.read('N_PRODS_CUT')[0]), self.max_number_prods_game_over))
        if flag == FLAG_ENGINE_CAPACITY:
            raise RuntimeError('pypownet_amd engine capacity exceeded (raise max_active_buses / lu_capacity)')
        return None

    def step(self, action, _is_simulation=False):
        if action is None:
            raise ValueError('Cannot play None action')
        self.last_action = action
        self.timestep += 1
        eng = self.engine
        a = np.asarray(action.as_array())[None, :]
        # counters as the action meets them: only the payload of an IllegalActionException needs them, and an action
        # without a switch cannot be illegal
        counters = (eng.read('RECONNECTABLE')[0], eng.read('LINE_COOLDOWN')[0], eng.read('NODE_COOLDOWN')[0]) if a.any() else None
        if _is_simulation:
            eng.simulate(a)
        else:
            eng.step(a)
        sim = bool(_is_simulation)
        done = bool(eng.read('DONE', simulation=sim)[0])
        flag = self._flag_object(int(eng.read('FLAG', simulation=sim)[0]), sim)
        bits = int(eng.read('ILLEGAL', simulation=sim)[0])
        if bits:
            self.timestep += 1       # the repaired action is re-submitted: apply_action runs a second time (game.py:600, 849)
        illegal = self._illegal_exception(action, bits, counters) if bits else None
        if flag is None:
            flag = illegal
        obs = None if done else self.export_observation(simulation=sim)
        return obs, flag, done

    def simulate(self, action):
        return self.step(action, _is_simulation=True)

    def process_game_over(self):
        self.engine.force_game_over()
        # the reference recurses until a restart converges (game.py:776-780); one engine pass gives up after 64 attempts
        # (include/ppn.h, PPN_RESTART_ATTEMPTS): go on, up to about what Python's recursion limit allows the reference
        for _ in range(15):
            if int(self.engine.read('DEAD')[0]) != 3:
                break
            self.engine.process_game_over()
        # the reference increments epoch on every (recursive) attempt, game.py:767: so does the device counter
        self.epoch = int(self.engine.read('EPOCH')[0])
        if int(self.engine.read('DEAD')[0]) == 3:
            raise RecursionError('process_game_over: the restarted grid keeps diverging (1024 attempts = 16 passes of 64)')

    def reset_grid(self):
        raise NotImplementedError('reset_grid is internal to process_game_over on the device engine')
