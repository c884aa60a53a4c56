"""Environment folder + ``configuration.yaml`` reader (reference: pypownet/parameters.py:10-153).

Same folder layout and the same 17 mandatory scalar keys.  Differences, all additive:
  * ``loadflow_backend`` additionally accepts ``hip`` (this engine); ``pypower``/``matpower`` are accepted and mapped
    onto ``hip`` because those runtimes are what this engine replaces.  There is no CPU backend;
  * optional keys ``solver`` (``newton``|``fdxb``, default ``fdxb`` = the reference's PF_ALG=2), ``tol``,
    ``max_it``;
  * the grid may be ``reference_grid.py`` (PYPOWER case format) or ``reference_grid.json``;
  * ``overrides`` lets a caller change keys without touching a read-only folder.
"""
import contextlib
import importlib.util
import logging
import os
import sys
import types

import yaml

from .reward_signal import RewardSignal

_MANDATORY_KEYS = [
    'loadflow_backend', 'loadflow_mode', 'max_seconds_per_timestep', 'hard_overflow_coefficient',
    'n_timesteps_hard_overflow_is_broken', 'n_timesteps_consecutive_soft_overflow_breaks',
    'n_timesteps_soft_overflow_is_broken', 'n_timesteps_horizon_maintenance', 'max_number_prods_game_over',
    'max_number_loads_game_over', 'n_timesteps_actionned_line_reactionable',
    'n_timesteps_actionned_node_reactionable', 'n_timesteps_pending_line_reactionable_when_overflowed',
    'n_timesteps_pending_node_reactionable_when_overflowed', 'max_number_actionned_substations',
    'max_number_actionned_lines', 'max_number_actionned_total']


@contextlib.contextmanager
def _reference_package_aliases():
    """An environment folder's ``reward_signal.py`` is written against the reference package: it imports
    ``pypownet.environment`` / ``pypownet.reward_signal`` and tests the step flag with ``isinstance(flag,
    pypownet.environment.DivergingLoadflowException)`` (parameters/default14/reward_signal.py:1-3, 48-60).  While such
    a file is executed, the names ``pypownet`` and ``pypownet.<module>`` resolve to this package's modules, so that the
    unmodified file loads here and its isinstance checks see the exception classes RunEnv.step actually returns -- also
    when the reference package itself happens to be installed."""
    from . import environment, reward_signal, game, chronic
    pkg = types.ModuleType('pypownet')
    pkg.__path__ = []
    mods = {'pypownet': pkg, 'pypownet.environment': environment, 'pypownet.reward_signal': reward_signal,
            'pypownet.game': game, 'pypownet.chronic': chronic, 'pypownet.parameters': sys.modules[__name__]}
    for k, m in mods.items():
        if k != 'pypownet':
            setattr(pkg, k.split('.', 1)[1], m)
    saved = {k: sys.modules.get(k) for k in mods}
    sys.modules.update(mods)
    try:
        yield
    finally:
        for k, m in saved.items():
            if m is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = m


class Parameters(object):
    def __init__(self, parameters_folder, game_level, overrides=None):
        self.__parameters_path = os.path.abspath(parameters_folder)
        self.logger = logging.getLogger('pypownet.' + __name__)
        if not os.path.exists(self.__parameters_path):
            raise FileNotFoundError('folder %s does not exist' % os.path.abspath(parameters_folder))
        self.level_folder = os.path.join(self.__parameters_path, game_level)
        if not os.path.exists(self.level_folder):
            level_folders = [os.path.join(self.__parameters_path, d) for d in os.listdir(self.__parameters_path)
                             if os.path.isdir(os.path.join(self.__parameters_path, d))]
            raise FileNotFoundError('Game level folder %s does not exist; level folders found in %s: %s' % (
                game_level, self.__parameters_path, '[' + ', '.join(level_folders) + ']'))
        fmt = lambda f: os.path.join(self.level_folder, f)
        for f in ['configuration.yaml', 'chronics/']:
            if not os.path.exists(fmt(f)):
                raise FileNotFoundError('Mandatory file/folder %s not found within %s' % (f, self.level_folder))
        grid = None
        for cand in ('reference_grid.py', 'reference_grid.json'):
            if os.path.exists(fmt(cand)):
                grid = fmt(cand)
                break
        if grid is None:
            raise FileNotFoundError('Mandatory file/folder reference_grid.py not found within %s' % self.level_folder)
        self.reference_grid_path = grid
        self.chronics_path = fmt('chronics/')
        self.configuration_path = fmt('configuration.yaml')
        with open(self.configuration_path, 'r') as stream:
            self.simulator_configuration = yaml.safe_load(stream)
        if overrides:
            self.simulator_configuration.update(overrides)
        for k in _MANDATORY_KEYS:
            if k not in self.simulator_configuration:
                raise KeyError(k)

        reward_path = os.path.join(self.__parameters_path, 'reward_signal.py')
        self.reward_signal_class = RewardSignal
        if os.path.exists(reward_path):
            try:
                with _reference_package_aliases():
                    spec = importlib.util.spec_from_file_location('reward_signal_%d' % abs(hash(reward_path)), reward_path)
                    mod = importlib.util.module_from_spec(spec)
                    spec.loader.exec_module(mod)
                self.reward_signal_class = getattr(mod, 'CustomRewardSignal')
            except Exception as e:  # same fallback as the reference on ImportError (parameters.py:55-70)
                self.logger.error('/!\\ Using default reward signal (%s)' % e)

    def get_reward_signal_class(self): return self.reward_signal_class
    def get_reference_grid_path(self, loadflow_backend=None): return self.reference_grid_path
    def get_chronics_path(self): return self.chronics_path
    def get_parameters_path(self): return self.__parameters_path

    def get_loadflow_backend(self):
        backend = str(self.simulator_configuration['loadflow_backend']).lower()
        if backend not in ['matpower', 'pypower', 'hip']:
            raise ValueError('loadflow_backend %s is not currently supported; supported backend: '
                             '"hip" ("pypower"/"matpower" map to "hip")' % backend)
        return 'hip'

    def _get_loadflow_mode(self):
        mode = str(self.simulator_configuration['loadflow_mode']).lower()
        if mode not in ['ac', 'dc']:
            raise ValueError('loadflow_mode value in configuration file should be either "AC" or "DC"')
        return mode

    def is_dc_mode(self): return self._get_loadflow_mode() == 'dc'

    def get_solver(self):
        s = str(self.simulator_configuration.get('solver', 'fdxb')).lower()
        if s not in ('fdxb', 'newton'):
            raise ValueError('solver should be "fdxb" (reference PF_ALG=2) or "newton"')
        return s

    def get_tol(self): return float(self.simulator_configuration.get('tol', 1e-6))

    def get_max_it(self):
        default = 25 if self.get_solver() == 'fdxb' else 10
        return int(self.simulator_configuration.get('max_it', default))

    def _g(self, k): return self.simulator_configuration[k]
    def get_max_seconds_per_timestep(self): return self._g('max_seconds_per_timestep')
    def get_hard_overflow_coefficient(self): return self._g('hard_overflow_coefficient')
    def get_n_timesteps_hard_overflow_is_broken(self): return self._g('n_timesteps_hard_overflow_is_broken')
    def get_n_timesteps_consecutive_soft_overflow_breaks(self): return self._g('n_timesteps_consecutive_soft_overflow_breaks')
    def get_n_timesteps_soft_overflow_is_broken(self): return self._g('n_timesteps_soft_overflow_is_broken')
    def get_n_timesteps_horizon_maintenance(self): return self._g('n_timesteps_horizon_maintenance')
    def get_max_number_prods_game_over(self): return self._g('max_number_prods_game_over')
    def get_max_number_loads_game_over(self): return self._g('max_number_loads_game_over')
    def get_n_timesteps_actionned_line_reactionable(self): return self._g('n_timesteps_actionned_line_reactionable')
    def get_n_timesteps_actionned_node_reactionable(self): return self._g('n_timesteps_actionned_node_reactionable')
    def get_n_timesteps_pending_line_reactionable_when_overflowed(self): return self._g('n_timesteps_pending_line_reactionable_when_overflowed')
    def get_n_timesteps_pending_node_reactionable_when_overflowed(self): return self._g('n_timesteps_pending_node_reactionable_when_overflowed')
    def get_max_number_actionned_substations(self): return self._g('max_number_actionned_substations')
    def get_max_number_actionned_lines(self): return self._g('max_number_actionned_lines')
    def get_max_number_actionned_total(self): return self._g('max_number_actionned_total')

    def __str__(self):
        params_str = ['    ' + k + ': ' + str(v) for k, v in self.simulator_configuration.items()]
        width = max(map(len, params_str))
        return '\n'.join(['  ' + '=' * width, ' ' * (width // 2 - 5) + 'GAME PARAMETERS', '  ' + '=' * width,
                          '\n'.join(params_str), '  ' + '=' * width])
