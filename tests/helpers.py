"""Shared test helpers: building oracle games from fixture env folders, scripted scenarios."""
import os

import numpy as np
import yaml

from pypownet_amd.case import Case
from pypownet_amd.chronic import Chronic

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
ENVS = os.path.join(ROOT, 'tests', 'golden', 'envs')


def load_env(envname, level='level0', conf=None):
    d = os.path.join(ENVS, envname, level)
    grid = os.path.join(d, 'reference_grid.json')
    case = Case.from_file(grid)
    with open(os.path.join(d, 'configuration.yaml')) as f:
        cfg = yaml.safe_load(f)
    if conf:
        cfg.update(conf)
    cdir = os.path.join(d, 'chronics')
    chronics = [Chronic(os.path.join(cdir, c)) for c in sorted(os.listdir(cdir))]
    return case, cfg, chronics


def oracle_game(envname, conf=None, **kw):
    from oracle.game_np import OracleGame
    case, cfg, chronics = load_env(envname, conf=conf)
    return OracleGame(case, cfg, chronics, **kw)


def do_nothing(case):
    return np.zeros(case.action_length, dtype=np.int64)


def set_substation_switches(case, action, sub_id, values):
    """ActionSpace.set_substation_switches_in_action (reference environment.py:201-239): ``values`` in the
    zipped per-substation order [prod?, load?, origins..., extremities...]."""
    s = int(np.where(case.sub_ids == sub_id)[0][0])
    idx = case.mapping_array[s]
    assert len(idx) == len(values)
    action[np.asarray(idx, dtype=int)] = np.asarray(values, dtype=np.int64)
    return action


def nodes_of_substation(case, obs, sub_id):
    """Observation.get_nodes_of_substation (reference environment.py:603-640)."""
    s = int(np.where(case.sub_ids == sub_id)[0][0])
    topo = np.concatenate([obs['productions_nodes'], obs['loads_nodes'], obs['lines_or_nodes'],
                           obs['lines_ex_nodes']])
    return [int(v) for v in topo[np.asarray(case.mapping_array[s], dtype=int)]]


def set_line_switch(case, action, line_id, value=1):
    action[case.nP + case.nL + 2 * case.nl + line_id] = value
    return action


def differential(new_conf, old_conf):
    return [1 if (a - b) != 0 else 0 for a, b in zip(new_conf, old_conf)]


def run_wrapped(game, policy, n_iter):
    """tests/common_assets.py WrappedRunner.loop protocol: process_game_over() first, then n steps; on
    ``done`` the env is reset with process_game_over().  ``policy(step_1based, obs)`` -> action array.
    Returns (dones, flags, illegal_bits)."""
    game.process_game_over()
    obs = game.export_observation()
    dones, flags, ills = [], [], []
    for i in range(1, n_iter + 1):
        action = policy(i, obs)
        o, flag, ill, done = game.step(action)
        if done:
            game.process_game_over()
            obs = game.export_observation()
        else:
            obs = o
        dones.append(done), flags.append(flag), ills.append(ill)
    return dones, flags, ills
