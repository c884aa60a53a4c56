"""Batched load-flow step engine: thin numpy-facing wrapper over the C ABI (include/ppn.h).

``Engine`` owns B independent grids on one GPU.  It replaces, for the whole batch at once, what one reference
``Game`` + ``Grid`` pair does per environment (pypownet/game.py, pypownet/grid.py): ``reset`` = Game.__init__'s
first load + cascade, ``step`` = Game.step, ``simulate`` = Game.simulate, ``process_game_over``,
``runpf`` = the bare pypower.runpf call.
"""
import ctypes as C

import numpy as np

from . import _lib
from .case import Case

MODE_AC, MODE_DC = 0, 1
SOLVER_NEWTON, SOLVER_FDXB = 1, 2
FLAG_OK, FLAG_DIVERGED, FLAG_TOO_MANY_LOADS, FLAG_TOO_MANY_PRODS, FLAG_ENGINE_CAPACITY = 0, 1, 2, 3, 4
ILL_TOO_MANY, ILL_BROKEN_LINE, ILL_LINE_COOLDOWN, ILL_NODE_COOLDOWN = 1, 2, 4, 8
EV_SWITCHED, EV_MAINTENANCE, EV_HAZARD, EV_HARD_OVERFLOW, EV_SOFT_OVERFLOW = 1, 2, 4, 8, 16      # LINE_EVENTS bits
SOLVE_CONVERGED, SOLVE_OUTAGE, SOLVE_NOT_CONNEXE, SOLVE_CAPACITY = 0, 1, 2, 4                      # SOLVE_OUTCOME


class EngineError(RuntimeError):
    pass


def rules_from_conf(conf, without_overflow_cutoff=False, game_over_mode='soft', looping_mode='natural',
                    max_active_buses=0, lu_capacity=0, rng_seed=0, q_plane_auto=0):
    """Build the ppn_rules struct from a parsed configuration.yaml (pypownet/parameters.py keys)."""
    r = _lib.PpnRules()
    r.mode = MODE_DC if str(conf.get('loadflow_mode', 'AC')).lower() == 'dc' else MODE_AC
    solver = str(conf.get('solver', 'fdxb')).lower()
    r.solver = SOLVER_NEWTON if solver == 'newton' else SOLVER_FDXB
    r.tol = float(conf.get('tol', 1e-6))
    r.max_it = int(conf.get('max_it', 10 if r.solver == SOLVER_NEWTON else 25))
    r.hard_overflow_coefficient = 1e9 if without_overflow_cutoff else float(conf['hard_overflow_coefficient'])
    r.n_timesteps_hard_overflow_is_broken = int(conf['n_timesteps_hard_overflow_is_broken'])
    r.n_timesteps_consecutive_soft_overflow_breaks = 1e12 if without_overflow_cutoff else float(
        conf['n_timesteps_consecutive_soft_overflow_breaks'])
    r.n_timesteps_soft_overflow_is_broken = int(conf['n_timesteps_soft_overflow_is_broken'])
    r.n_timesteps_horizon_maintenance = int(conf['n_timesteps_horizon_maintenance'])
    r.max_number_prods_game_over = int(conf['max_number_prods_game_over'])
    r.max_number_loads_game_over = int(conf['max_number_loads_game_over'])
    r.n_timesteps_actionned_line_reactionable = int(conf['n_timesteps_actionned_line_reactionable'])
    r.n_timesteps_actionned_node_reactionable = int(conf['n_timesteps_actionned_node_reactionable'])
    r.max_number_actionned_substations = int(conf['max_number_actionned_substations'])
    r.max_number_actionned_lines = int(conf['max_number_actionned_lines'])
    r.max_number_actionned_total = int(conf['max_number_actionned_total'])
    r.game_over_mode_hard = 1 if game_over_mode == 'hard' else 0
    if looping_mode not in ('natural', 'fixed', 'random'):
        raise ValueError('chronic looping mode should be "natural", "fixed" or "random"')
    r.chronic_looping = {'natural': 0, 'fixed': 1, 'random': 2}[looping_mode]
    r.rng_seed = int(rng_seed) & 0x7fffffff
    r.max_active_buses = int(max_active_buses)
    r.lu_capacity = int(lu_capacity)
    r.q_plane_auto = int(q_plane_auto)
    return r


class Engine(object):
    def __init__(self, case, conf, batch, device=0, chronics=None, thermal_limits=None, **rule_kw):
        self._lib = _lib.load_library()     # pypownet_amd/libppn.so; raises ImportError when it has not been built
        self.case = case if isinstance(case, Case) else Case(case)
        self.batch = int(batch)
        self._n_candidates = 0
        ppc = self.case.ppc
        self._bus = np.ascontiguousarray(ppc['bus'], dtype=np.float64)
        self._gen = np.ascontiguousarray(ppc['gen'], dtype=np.float64)
        self._branch = np.ascontiguousarray(ppc['branch'], dtype=np.float64)
        pc = _lib.PpnCase()
        pc.n_bus_rows, pc.bus_cols = self._bus.shape
        pc.n_gen, pc.gen_cols = self._gen.shape
        pc.n_branch, pc.branch_cols = self._branch.shape
        pc.base_mva = float(ppc['baseMVA'])
        dp = C.POINTER(C.c_double)
        pc.bus, pc.gen, pc.branch = self._bus.ctypes.data_as(dp), self._gen.ctypes.data_as(dp), \
            self._branch.ctypes.data_as(dp)
        self.rules = rules_from_conf(conf, **rule_kw)
        h = C.c_void_p()
        rc = self._lib.ppn_create(C.byref(pc), C.byref(self.rules), self.batch, int(device), C.byref(h))
        if rc != 0:
            raise EngineError('ppn_create failed (%d): %s' % (rc, self._lib.ppn_last_error(None).decode()))
        self._h = h
        self.n_chronics = 0
        if chronics:
            for ch in chronics:
                self.load_chronic(ch)
            limits = chronics[0].get_imaps() if thermal_limits is None else thermal_limits
            self.set_thermal_limits(limits)
        elif thermal_limits is not None:
            self.set_thermal_limits(thermal_limits)

    # ---- plumbing -----------------------------------------------------------------------------------
    def _check(self, rc, what):
        if rc != 0:
            raise EngineError('%s failed (%d): %s' % (what, rc, self._lib.ppn_last_error(self._h).decode()))

    def close(self):
        if getattr(self, '_h', None):
            self._lib.ppn_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def dim(self, which):
        return int(self._lib.ppn_dim(self._h, which))

    @property
    def lds_bytes(self):
        return self.dim(7)

    STEP_FORMS = {0: 'K_STEP', 1: 'K_STEP_PERSIST', 2: 'K_STEP_OBS', 3: 'K_ROLLOUT'}

    def last_step_form(self):
        """Kernel form the last step launch took (ppn_dim 19): ('K_STEP' | 'K_STEP_PERSIST' | 'K_STEP_OBS' | 'K_ROLLOUT', stepped in
        two capacity classes?) -- what the engine chose, not what a caller would guess from the batch size."""
        v = self.dim(19)
        return self.STEP_FORMS.get(v & 3, '?'), bool(v & 4)

    # ---- data ---------------------------------------------------------------------------------------
    def set_thermal_limits(self, limits):
        a = np.ascontiguousarray(limits, dtype=np.float64)
        assert a.shape == (self.case.nl,)
        self.thermal_limits = a
        self._check(self._lib.ppn_set_thermal_limits(self._h, a.ctypes.data_as(C.POINTER(C.c_double))), 'ppn_set_thermal_limits')

    def set_reward(self, params):
        """Coefficients of the device-side reward (PPN_F_REWARD): a dict with the keys of ``_lib.REWARD_PARAM_NAMES`` or a
        ``reward_signal.DefaultGridRewardSignal`` (the reference's shipped CustomRewardSignal)."""
        if hasattr(params, 'as_engine_params'):
            params = params.as_engine_params()
        p = _lib.PpnRewardParams(**{k: float(params[k]) for k in _lib.REWARD_PARAM_NAMES})
        self._check(self._lib.ppn_set_reward(self._h, C.byref(p)), 'ppn_set_reward')

    def load_chronic(self, ch):
        """Upload a pypownet_amd.chronic.Chronic into the next slot."""
        T = ch.n_timesteps
        c = _lib.PpnChronic()
        c.T = T
        keep = []
        fp = C.POINTER(C.c_float)

        def f32(a, n):
            a = np.ascontiguousarray(np.asarray(a)[:T].reshape(T, n), dtype=np.float32)
            keep.append(a)
            return a.ctypes.data_as(fp)
        cs = self.case
        c.prods_p, c.prods_v = f32(ch.prods_p, cs.nP), f32(ch.prods_v, cs.nP)
        c.loads_p, c.loads_q = f32(ch.loads_p, cs.nL), f32(ch.loads_q, cs.nL)
        c.prods_p_planned, c.prods_v_planned = f32(ch.prods_p_planned, cs.nP), f32(ch.prods_v_planned, cs.nP)
        c.loads_p_planned, c.loads_q_planned = f32(ch.loads_p_planned, cs.nL), f32(ch.loads_q_planned, cs.nL)
        c.maintenance, c.hazards = f32(ch.maintenance, cs.nl), f32(ch.hazards, cs.nl)
        ids = np.ascontiguousarray(ch.timestep_ids[:T], dtype=np.int32)
        dates = np.ascontiguousarray(ch.date_fields(), dtype=np.int32)
        keep += [ids, dates]
        c.ids = ids.ctypes.data_as(C.POINTER(C.c_int32))
        c.dates = dates.ctypes.data_as(C.POINTER(C.c_int32))
        self._check(self._lib.ppn_load_chronic(self._h, self.n_chronics, C.byref(c)), 'ppn_load_chronic')
        self.n_chronics += 1

    # ---- game ---------------------------------------------------------------------------------------
    def reset(self, env_ids=None, chronic_slot=None, t0=None):
        ip = C.POINTER(C.c_int32)
        n = self.batch if env_ids is None else len(env_ids)

        def arr(a):
            if a is None:
                return None, None
            a = np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.int32), (n,)))
            return a, a.ctypes.data_as(ip)
        e, ep = arr(env_ids)
        s, sp = arr(chronic_slot)
        t, tp = arr(t0)
        self._check(self._lib.ppn_reset(self._h, ep, n, sp, tp), 'ppn_reset')

    def _actions(self, actions):
        a = np.ascontiguousarray(np.asarray(actions).reshape(self.batch, self.case.action_length) != 0, dtype=np.uint8)
        return a

    def step(self, actions, auto_reset=False):
        """auto_reset: False | True (environments that end are restarted in the same launch) | 2 (same outcome, the restart
        deferred to the next step launch: include/ppn.h, ppn_step)."""
        a = self._actions(actions)
        self._check(self._lib.ppn_step(self._h, a.ctypes.data, 0, 0, int(auto_reset)), 'ppn_step')

    def step_device(self, actions_ptr, auto_reset=False):
        """actions_ptr: device address of a uint8 [batch x action_len] buffer (e.g. torch_tensor.data_ptr())."""
        self._check(self._lib.ppn_step(self._h, C.c_void_p(int(actions_ptr)), 1, 0, int(auto_reset)), 'ppn_step')

    def step_observe_device(self, actions_ptr, obs_ptr, nbytes, auto_reset=True, layout='full', dtype=np.float64):
        """RunEnv.step as the reference returns it -- the step AND the observation -- in one launch (include/ppn.h,
        ppn_step_observe): ``actions_ptr`` as for step_device, ``obs_ptr`` the device address of a
        ``[batch x observation_length(layout)]`` buffer of ``dtype`` (float64 / float32) that every environment's workgroup fills
        right behind its step.  Same rows as ``step_device`` + ``observations_into_device``."""
        self._check(self._lib.ppn_step_observe(self._h, C.c_void_p(int(actions_ptr)), 1, int(bool(auto_reset)), self.OBS_LAYOUTS[layout],
                                               1 if np.dtype(dtype) == np.float32 else 0, C.c_void_p(int(obs_ptr)), int(nbytes)),
                    'ppn_step_observe')

    def rollout(self, actions, n_steps=None, auto_reset=True):
        """Open-loop rollout (include/ppn.h, ppn_rollout): ``actions`` is ``[n_steps x batch x action_length]`` (one matrix per
        step) or ``[batch x action_length]`` replayed ``n_steps`` times (the do-nothing agent); one launch, every environment
        plays its steps back to back."""
        a = np.asarray(actions)
        if a.ndim == 3:
            n_steps = a.shape[0] if n_steps is None else n_steps
            a = np.ascontiguousarray(a.reshape(n_steps, self.batch, self.case.action_length) != 0, dtype=np.uint8)
            per_step = 1
        else:
            assert n_steps is not None
            a, per_step = self._actions(a), 0
        self._check(self._lib.ppn_rollout(self._h, a.ctypes.data, 0, int(n_steps), per_step, int(auto_reset)), 'ppn_rollout')

    def rollout_device(self, actions_ptr, n_steps, per_step_actions=False, auto_reset=True):
        """The same with the action matrix / sequence already on the device."""
        self._check(self._lib.ppn_rollout(self._h, C.c_void_p(int(actions_ptr)), 1, int(n_steps), 1 if per_step_actions else 0,
                                          int(auto_reset)), 'ppn_rollout')

    POLICIES = {'do_nothing': 0, 'line_relief': 1}

    def _policy_args(self, policy, params):
        pid = self.POLICIES[policy] if isinstance(policy, str) else int(policy)
        p = np.ascontiguousarray(params if params is not None else [], dtype=np.float64)
        return pid, p, p.ctypes.data_as(C.POINTER(C.c_double)), int(p.size)

    def policy_actions(self, policy, params, out_ptr):
        """The built-in device policy's action for the CURRENT state of every environment (include/ppn.h, ppn_policy_actions),
        written to the caller's DEVICE buffer u8 [batch x action_length] at address ``out_ptr``."""
        pid, keep, pp, n = self._policy_args(policy, params)
        self._check(self._lib.ppn_policy_actions(self._h, pid, pp, n, C.c_void_p(int(out_ptr))), 'ppn_policy_actions')

    def rollout_policy(self, policy, params, n_steps):
        """Closed-loop rollout of a built-in device policy (include/ppn.h, ppn_rollout_policy): ``n_steps`` rounds of
        { policy -> action; Game.step with the fused restart } per environment in ONE launch, no environment waiting for the
        batch between its steps; same trajectories as ``policy_actions`` + ``step_device(auto_reset=1)`` round by round."""
        pid, keep, pp, n = self._policy_args(policy, params)
        self._check(self._lib.ppn_rollout_policy(self._h, pid, pp, n, int(n_steps)), 'ppn_rollout_policy')

    # ---- asynchronous session: send / recv for policies that live outside the engine (include/ppn.h) ----------------------------
    LAYOUT_IDS = {'full': 0, 'minimalist': 1, 'ac_minimalist': 2}

    def async_start(self, obs_ptr=0, obs_bytes=0, report_ptr=0, layout='full', dtype=np.float64, workgroups=0, idle_timeout_ms=0):
        """Starts the step server (ppn_async_start).  ``obs_ptr`` / ``report_ptr``: DEVICE buffers of the caller, [batch x
        observation_length(layout)] of ``dtype`` and [batch x 3] float64 (done, flag, reward sum); 0 = not written."""
        cfg = _lib.PpnAsyncConfig()
        cfg.struct_size = C.sizeof(_lib.PpnAsyncConfig)
        cfg.layout = self.LAYOUT_IDS[layout] if isinstance(layout, str) else int(layout)
        cfg.as_f32 = 1 if np.dtype(dtype) == np.dtype(np.float32) else 0
        cfg.workgroups, cfg.idle_timeout_ms = int(workgroups), int(idle_timeout_ms)
        cfg.obs_device, cfg.obs_bytes, cfg.report_device = int(obs_ptr) or None, int(obs_bytes), int(report_ptr) or None
        self._recv_buf = np.empty(self.batch, dtype=np.int32)
        self._check(self._lib.ppn_async_start(self._h, C.byref(cfg)), 'ppn_async_start')

    def send(self, env_ids, actions, rows_by_env=False):
        """One step each for the listed environments (ppn_send).  ``actions``: a host uint8 array -- [len(env_ids) x action_length], or
        [batch x action_length] with ``rows_by_env`` -- or the integer address of such rows in DEVICE memory (see ``send_device``)."""
        ids = np.ascontiguousarray(env_ids, dtype=np.int32)
        a = np.ascontiguousarray(actions, dtype=np.uint8)
        assert a.shape == ((self.batch if rows_by_env else len(ids)), self.case.action_length), a.shape
        self._check(self._lib.ppn_send(self._h, ids.ctypes.data, len(ids), a.ctypes.data, 0, 1 if rows_by_env else 0), 'ppn_send')

    def send_device(self, env_ids, actions_ptr, rows_by_env=False):
        """The same with the action rows in DEVICE memory at ``actions_ptr`` -- complete on ``async_stream_ptr()`` and left alone until
        that stream has passed the send (include/ppn.h, "streams")."""
        ids = np.ascontiguousarray(env_ids, dtype=np.int32)
        self._check(self._lib.ppn_send(self._h, ids.ctypes.data, len(ids), C.c_void_p(int(actions_ptr)), 1, 1 if rows_by_env else 0), 'ppn_send')

    def recv(self, min_ready=1, max_n=None, timeout_ms=-1, ids_device_ptr=0):
        """Environments whose step is complete (ppn_recv): at least min(min_ready, in flight) of them unless the timeout strikes, in
        completion order, as an int32 array (a view of a buffer the next recv overwrites).  ``ids_device_ptr``: DEVICE int32 buffer that
        receives the same ids through an asynchronous copy on the session's stream."""
        n = C.c_int32(0)
        cap = self.batch if max_n is None else min(int(max_n), self.batch)
        self._check(self._lib.ppn_recv(self._h, int(min_ready), cap, int(timeout_ms), self._recv_buf.ctypes.data, C.byref(n),
                                       C.c_void_p(int(ids_device_ptr)) if ids_device_ptr else None), 'ppn_recv')
        return self._recv_buf[:n.value]

    def async_stop(self):
        self._check(self._lib.ppn_async_stop(self._h), 'ppn_async_stop')

    def async_stream_ptr(self):
        """hipStream_t of the session's device work as an integer (torch.cuda.ExternalStream wraps it); 0 outside a session."""
        return int(self._lib.ppn_async_stream(self._h) or 0)

    def async_stats(self):
        return dict(in_flight=int(self._lib.ppn_async_stat(self._h, 0)), workgroups=int(self._lib.ppn_async_stat(self._h, 1)),
                    server_restarts=int(self._lib.ppn_async_stat(self._h, 2)), republished=int(self._lib.ppn_async_stat(self._h, 3)))

    def restart_memo(self, enable=True, max_bytes=0):
        """Restart memo (include/ppn.h: ppn_restart_memo): restarts of ended episodes are computed once per chronic position and
        copied afterwards -- same fields, counters included, as if every restart had been computed."""
        self._check(self._lib.ppn_restart_memo(self._h, 1 if enable else 0, int(max_bytes)), 'ppn_restart_memo')

    def restart_memo_stats(self):
        g = lambda k: int(self._lib.ppn_restart_memo_stat(self._h, k))      # noqa: E731
        return dict(snapshots=g(0), served=g(1), not_eligible=g(2), capacity=g(3), bytes_per_snapshot=g(4))

    def simulate(self, actions):
        a = self._actions(actions)
        self._check(self._lib.ppn_step(self._h, a.ctypes.data, 0, 1, 0), 'ppn_step(simulate)')

    def simulate_candidates(self, actions, env_ids):
        """Topology-action search: candidate c plays ``Game.simulate(actions[c])`` from the CURRENT state of environment
        ``env_ids[c]`` (any number of candidates per environment, one launch).  Read the outcome with
        ``read(name, simulation=2)`` / ``observations(simulation=2)``: one row per candidate."""
        ids = np.ascontiguousarray(env_ids, dtype=np.int32)
        a = np.ascontiguousarray(actions, dtype=np.uint8)
        assert a.shape == (len(ids), self.case.action_length)
        self._n_candidates = len(ids)
        self._check(self._lib.ppn_simulate_candidates(self._h, a.ctypes.data, 0, ids.ctypes.data_as(C.POINTER(C.c_int32)),
                                                      len(ids)), 'ppn_simulate_candidates')

    def simulate_candidates_device(self, actions_ptr, env_ids):
        """The same with the candidate actions already on the device (uint8 [n x action_length] at ``actions_ptr``)."""
        ids = np.ascontiguousarray(env_ids, dtype=np.int32)
        self._n_candidates = len(ids)
        self._check(self._lib.ppn_simulate_candidates(self._h, C.c_void_p(int(actions_ptr)), 1,
                                                      ids.ctypes.data_as(C.POINTER(C.c_int32)), len(ids)), 'ppn_simulate_candidates')

    def process_game_over(self, env_mask=None):
        """Game.process_game_over for every dead environment, plus the live ones selected by env_mask."""
        if env_mask is None:
            self._check(self._lib.ppn_process_game_over(self._h, None), 'ppn_process_game_over')
        else:
            m = np.ascontiguousarray(np.broadcast_to(np.asarray(env_mask) != 0, (self.batch,)), dtype=np.uint8)
            self._check(self._lib.ppn_process_game_over(self._h, m.ctypes.data), 'ppn_process_game_over')

    def force_game_over(self):
        self.process_game_over(np.ones(self.batch, dtype=np.uint8))

    def is_action_valid(self, actions):
        a = self._actions(actions)
        out = np.zeros(self.batch, dtype=np.uint8)
        self._check(self._lib.ppn_is_action_valid(self._h, a.ctypes.data, out.ctypes.data), 'ppn_is_action_valid')
        return out.astype(bool)

    def runpf(self):
        """The bare solve on the CURRENT state of every environment (results through ``read``)."""
        self._check(self._lib.ppn_runpf_batch(self._h), 'ppn_runpf_batch')

    def runpf_arrays(self, bus, gen, branch):
        """``runpf(mpc, ppopt, '', '')`` / ``rundcpf`` of the reference's seam (pypownet/grid.py:226-229) for n <= batch cases at
        once: ``bus [n x 2nS x >=13]``, ``gen [n x nP x >=8]``, ``branch [n x nl x >=11]`` MATPOWER arrays with '666'-twin ids
        -> ``(bus, gen, branch[.. x 17], success[n], outcome[n])`` (include/ppn.h, ppn_runpf_arrays)."""
        bus = np.ascontiguousarray(bus, dtype=np.float64)
        gen = np.ascontiguousarray(gen, dtype=np.float64)
        branch = np.ascontiguousarray(branch, dtype=np.float64)
        if bus.ndim == 2:
            bus, gen, branch = bus[None], gen[None], branch[None]
        n = bus.shape[0]
        if bus.ndim != 3 or gen.ndim != 3 or branch.ndim != 3 or gen.shape[0] != n or branch.shape[0] != n:
            raise ValueError('runpf_arrays: bus / gen / branch must be [n x rows x columns] arrays of the same n')
        io = _lib.PpnMpcBatch()
        io.struct_size = C.sizeof(_lib.PpnMpcBatch)
        io.n, io.bus_cols, io.gen_cols, io.branch_cols = n, bus.shape[2], gen.shape[2], branch.shape[2]
        io.bus_rows, io.gen_rows, io.branch_rows = bus.shape[1], gen.shape[1], branch.shape[1]      # (checked against the case by the library)
        bus_o, gen_o = np.empty_like(bus), np.empty_like(gen)
        br_o = np.empty((n, branch.shape[1], 17), dtype=np.float64)
        ok, outcome = np.zeros(n, dtype=np.uint8), np.zeros(n, dtype=np.int32)
        dp = C.POINTER(C.c_double)
        io.bus, io.gen, io.branch = bus.ctypes.data_as(dp), gen.ctypes.data_as(dp), branch.ctypes.data_as(dp)
        io.bus_out, io.gen_out, io.branch_out = bus_o.ctypes.data_as(dp), gen_o.ctypes.data_as(dp), br_o.ctypes.data_as(dp)
        io.success, io.outcome = ok.ctypes.data_as(C.POINTER(C.c_uint8)), outcome.ctypes.data_as(C.POINTER(C.c_int32))
        self._check(self._lib.ppn_runpf_arrays(self._h, C.byref(io)), 'ppn_runpf_arrays')
        return bus_o, gen_o, br_o, ok.astype(bool), outcome

    def sync(self):
        self._check(self._lib.ppn_sync(self._h), 'ppn_sync')

    def stream_ptr(self):
        """The HIP stream the engine launches on (ppn_stream), as an integer: torch.cuda.ExternalStream(ptr) wraps it."""
        self._lib.ppn_stream.restype = C.c_void_p
        return int(self._lib.ppn_stream(self._h) or 0)

    def wait(self):
        """Blocks until the engine's stream is idle; unlike sync() it does not settle the restarts a deferred auto-reset owes."""
        self._check(self._lib.ppn_wait(self._h), 'ppn_wait')

    def kernel_time(self, reset=False):
        ms, n = C.c_double(0.0), C.c_int64(0)
        self._check(self._lib.ppn_kernel_time(self._h, 1 if reset else 0, C.byref(ms), C.byref(n)), 'ppn_kernel_time')
        return ms.value, n.value

    # ---- state --------------------------------------------------------------------------------------
    def read(self, name, simulation=False):
        fid = _lib.FIELD_ID[name]
        nbytes = int(self._lib.ppn_field_bytes(self._h, fid))
        dt = np.dtype(_lib.field_dtype(name))
        rows = self._n_candidates if int(simulation) == 2 else self.batch
        out = np.empty((rows, nbytes // dt.itemsize), dtype=dt)
        self._check(self._lib.ppn_read(self._h, fid, out.ctypes.data, out.nbytes, 1, int(simulation)), 'ppn_read')
        return out[:, 0] if out.shape[1] == 1 and name not in ('OBSERVATION',) and nbytes == dt.itemsize else out

    def schedule_builds_in_kernel(self):
        """Schedules that environments had to build INSIDE a solve since the engine was created (internal field 102).  With the
        schedule pre-pass on (four-word engines; PPN_SCHED_PREPASS=0 turns it off) that is what the pre-pass did not foresee."""
        out = np.empty(self.batch, dtype=np.int32)
        self._check(self._lib.ppn_read(self._h, 102, out.ctypes.data, out.nbytes, 1, 0), 'ppn_read')
        return out

    def capacity_classes(self):
        """Two-capacity stepping (four-word engines with the default matrix capacity): the class the schedule pre-pass gave every
        environment for its last step -- 0 small storage (four environments per CU), 1 large (internal field 103)."""
        out = np.empty(self.batch, dtype=np.uint8)
        self._check(self._lib.ppn_read(self._h, 103, out.ctypes.data, out.nbytes, 1, 0), 'ppn_read')
        return out

    def schedule_caches(self):
        """Raw schedule cache blobs of every environment (internal field 101: header, signature, index tables)."""
        n = int(self._lib.ppn_field_bytes(self._h, 101))
        out = np.empty((self.batch, n), dtype=np.uint8)
        self._check(self._lib.ppn_read(self._h, 101, out.ctypes.data, out.nbytes, 1, 0), 'ppn_read')
        return out

    def read_into_device(self, name, dev_ptr, nbytes, simulation=False):
        fid = _lib.FIELD_ID[name]
        self._check(self._lib.ppn_read(self._h, fid, C.c_void_p(int(dev_ptr)), nbytes, 0, int(simulation)), 'ppn_read')

    def observations_into_device(self, dev_ptr, nbytes, simulation=False, layout='full', dtype=np.float64):
        """Gathers the observations straight into a caller-owned DEVICE buffer ([rows x observation_length(layout)] of dtype):
        no host round trip (the policy of a batched agent lives on the GPU)."""
        self._check(self._lib.ppn_read_observation(self._h, self.OBS_LAYOUTS[layout], 1 if np.dtype(dtype) == np.float32 else 0,
                                                   C.c_void_p(int(dev_ptr)), int(nbytes), 0, int(simulation)), 'ppn_read_observation')

    def observation_length(self, layout='full'):
        return int(self._lib.ppn_observation_length(self._h, self.OBS_LAYOUTS[layout]))

    def write(self, name, values):
        fid = _lib.FIELD_ID[name]
        dt = np.dtype(_lib.field_dtype(name))
        a = np.ascontiguousarray(values, dtype=dt)
        self._check(self._lib.ppn_write(self._h, fid, a.ctypes.data, a.nbytes), 'ppn_write')

    OBS_LAYOUTS = {'full': 0, 'minimalist': 1, 'ac_minimalist': 2}

    def observations(self, simulation=False, layout='full', dtype=np.float64):
        """``Observation.as_array()`` of every environment (layout 'full'), or the reference's reduced layouts
        ``as_minimalist().as_array()`` / ``as_ac_minimalist().as_array()`` gathered directly on the device;
        dtype float64 (reference) or float32."""
        if layout == 'full' and np.dtype(dtype) == np.float64:
            return self.read('OBSERVATION', simulation=simulation)
        lay = self.OBS_LAYOUTS[layout]
        n = int(self._lib.ppn_observation_length(self._h, lay))
        rows = self._n_candidates if int(simulation) == 2 else self.batch
        out = np.empty((rows, n), dtype=np.dtype(dtype))
        assert out.dtype in (np.float32, np.float64)
        self._check(self._lib.ppn_read_observation(self._h, lay, 1 if out.dtype == np.float32 else 0, out.ctypes.data,
                                                   out.nbytes, 1, int(simulation)), 'ppn_read_observation')
        return out
