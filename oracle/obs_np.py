"""ORACLE (test infrastructure, not product code): Observation arrays rebuilt on the host from an engine-shaped state.

The C oracle (oracle/ppn_oracle.c) exports the game state field by field but no observation; this module assembles
``Observation.as_array()`` -- and the reference's reduced layouts -- from those fields with the SAME numpy code the
single-environment oracle uses (``OracleGame.export_observation``, which follows pypownet/grid.py:496-566,
pypownet/game.py:945-978 and the field orders of pypownet/environment.py:451-466, 511-517, 583-595), so that the
lock-step tests can check the device-side gather (``ppn_read_observation``) under node splitting, line switching,
game overs and simulations against something that is not the HIP kernel.
Only tests/ may import this module."""
import numpy as np

from .game_np import OracleGame, Rules, obs_as_array, OBS_ARRAY_ORDER

_FIELDS = ('PRODS_NODES', 'LOADS_NODES', 'LINES_OR_NODES', 'LINES_EX_NODES', 'LINES_STATUS', 'VM', 'PD', 'QD', 'PG', 'QG',
           'VG', 'PF', 'QF', 'PT', 'QT', 'RECONNECTABLE', 'LINE_COOLDOWN', 'NODE_COOLDOWN', 'CHRONIC_SLOT', 'CHRONIC_ROW')

N_MINIMALIST_FIELDS = 22      # environment.py:451-466
N_AC_FIELDS = 12              # environment.py:511-517


def read_state(engine, simulation=False):
    """The fields an observation is made of, read through the engine-shaped wrapper (HIP engine or C oracle)."""
    return {f: engine.read(f, simulation=simulation) for f in _FIELDS}


def observation_dict(case, conf, chronics, thermal_limits, state, env, planned_from=None, without_overflow_cutoff=False):
    """``planned_from = (slot, row)``: chronic entry the planned_* series are taken from -- after a simulation the entry the
    simulation STARTED from (game.py:410-413: simulate does not advance current_timestep_entries; quirk q11); default:
    the entry the state stands on."""
    g = OracleGame.__new__(OracleGame)
    g.case = case
    g.rules = Rules(conf, without_overflow_cutoff)
    i = lambda k: np.asarray(state[k][env]).astype(np.int64)
    f = lambda k: np.asarray(state[k][env], dtype=np.float64)
    g.prods_nodes, g.loads_nodes, g.or_nodes, g.ex_nodes = i('PRODS_NODES'), i('LOADS_NODES'), i('LINES_OR_NODES'), i('LINES_EX_NODES')
    g.line_status = i('LINES_STATUS')
    g.vm, g.pd, g.qd, g.pg, g.qg, g.vg = f('VM'), f('PD'), f('QD'), f('PG'), f('QG'), f('VG')
    g.flows = np.stack([f('PF'), f('QF'), f('PT'), f('QT')], axis=1)
    g.thermal_limits = np.asarray(thermal_limits, dtype=np.float64)
    g.reconnectable, g.line_cooldown, g.node_cooldown = f('RECONNECTABLE'), f('LINE_COOLDOWN'), f('NODE_COOLDOWN')
    slot, row = int(state['CHRONIC_SLOT'][env]), max(int(state['CHRONIC_ROW'][env]), 0)
    g.chronic = chronics[slot]
    g.current_timestep_id = g.chronic.get_timestep_ids()[row]
    entries = g.chronic.get_timestep_entries(g.current_timestep_id)
    g.current_date = entries.get_datetime()
    if planned_from is not None:
        ps, pr = planned_from
        entries = chronics[int(ps)].get_timestep_entries(chronics[int(ps)].get_timestep_ids()[max(int(pr), 0)])
    g.current_entries = entries
    return g.export_observation()


def observation_array(*args, **kw):
    return obs_as_array(observation_dict(*args, **kw))


def reduced_array(obs_dict, layout):
    """layout 'minimalist' = MinimalistObservation.as_array(), 'ac_minimalist' = MinimalistACObservation.as_array(),
    'full' = Observation.as_array() (environment.py:451-466, 511-517, 583-595)."""
    n = {'minimalist': N_MINIMALIST_FIELDS, 'ac_minimalist': N_MINIMALIST_FIELDS + N_AC_FIELDS, 'full': len(OBS_ARRAY_ORDER)}[layout]
    return np.concatenate([np.atleast_1d(np.asarray(obs_dict[k], dtype=np.float64)).flatten() for k in OBS_ARRAY_ORDER[:n]])
