"""ORACLE (test infrastructure, not product code): single-environment restatement of pypownet's game
rules around the load-flow solve, in plain numpy, on top of oracle/pf_np.py.

Follows, function by function (reference file:line):
  OracleGame.__init__                 pypownet/game.py:255-340, pypownet/grid.py:40-95
  _sync_bus_types / _isolated_rows    pypownet/grid.py:141-209
  _build_mpc / compute_loadflow       pypownet/grid.py:212-264  (mpc in MATPOWER external format, 666-twin ids)
  extract_flows_a                     pypownet/grid.py:29-36, 112-138
  load_timestep_injections            pypownet/grid.py:266-311
  load_entries_from_timestep_id/next  pypownet/game.py:405-501
  compute_loadflow_cascading          pypownet/game.py:503-589
  verify_illegal_action/apply_action  pypownet/game.py:591-753, 1088-1100
  step / simulate / process_game_over pypownet/game.py:762-943
  export_observation                  pypownet/grid.py:496-566, pypownet/game.py:945-978

State is index based (substation index + node bit) instead of the reference's '666'-prefixed float ids;
the external ids are re-created for every solve so that the id convention itself is exercised.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import copy

import numpy as np

from . import pf_np

FLAG_OK, FLAG_DIVERGED, FLAG_TOO_MANY_LOADS, FLAG_TOO_MANY_PRODS = 0, 1, 2, 3
ILL_TOO_MANY, ILL_BROKEN, ILL_LINE_COOLDOWN, ILL_NODE_COOLDOWN = 1, 2, 4, 8


class Diverged(Exception):
    pass


class Illegal(Exception):
    def __init__(self, bits, broken=None, line_cd=None, node_cd=None):
        super(Illegal, self).__init__(bits)
        self.bits, self.broken, self.line_cd, self.node_cd = bits, broken, line_cd, node_cd


class Rules(object):
    """The 17 YAML scalars + solver knobs (pypownet/parameters.py:89-153)."""

    def __init__(self, conf, without_overflow_cutoff=False):
        g = conf.get
        self.dc = str(g('loadflow_mode', 'AC')).lower() == 'dc'
        self.solver = str(g('solver', 'fdxb')).lower()
        self.tol = float(g('tol', 1e-6))
        self.max_it = int(g('max_it', 25 if self.solver == 'fdxb' else 10))
        self.hard_coef = float(g('hard_overflow_coefficient'))
        self.n_hard_broken = int(g('n_timesteps_hard_overflow_is_broken'))
        self.n_soft_consecutive = float(g('n_timesteps_consecutive_soft_overflow_breaks'))
        self.n_soft_broken = int(g('n_timesteps_soft_overflow_is_broken'))
        self.horizon = int(g('n_timesteps_horizon_maintenance'))
        self.max_prods_cut = int(g('max_number_prods_game_over'))
        self.max_loads_cut = int(g('max_number_loads_game_over'))
        self.n_line_cooldown = int(g('n_timesteps_actionned_line_reactionable'))
        self.n_node_cooldown = int(g('n_timesteps_actionned_node_reactionable'))
        self.max_subs = int(g('max_number_actionned_substations'))
        self.max_lines = int(g('max_number_actionned_lines'))
        self.max_total = int(g('max_number_actionned_total'))
        if without_overflow_cutoff:  # game.py:268-275
            self.hard_coef = 1e9
            self.n_soft_consecutive = 1e12


class OracleGame(object):
    def __init__(self, case, conf, chronics, game_over_mode='soft', start_id=0, looping_mode='natural',
                 without_overflow_cutoff=False, thermal_limits=None):
        self.case = case
        self.rules = Rules(conf, without_overflow_cutoff)
        self.chronics = chronics
        self.looping_mode = looping_mode
        self.next_chronic_slot = start_id
        self.game_over_mode = game_over_mode
        c = case
        # grid state (Grid.__init__)
        self.prods_nodes = np.zeros(c.nP, dtype=np.int64)
        self.loads_nodes = np.zeros(c.nL, dtype=np.int64)
        self.or_nodes = np.zeros(c.nl, dtype=np.int64)
        self.ex_nodes = np.zeros(c.nl, dtype=np.int64)
        self.line_status = c.br_status0.astype(np.int64).copy()
        self.vm = c.vm0.copy()
        self.va = c.va0.copy()          # degrees, as stored in mpc
        self.pg = c.gen_pg0.copy()
        self.qg = c.gen_qg0.copy()
        self.vg = c.gen_vg0.copy()
        self.gen_status = c.gen_status0.astype(np.int64).copy()
        self.pd = c.load_pd0.copy()
        self.qd = c.load_qd0.copy()
        self.flows = np.zeros((c.nl, 4))
        self.bus_type = c.bus_type0.copy()

        self.chronic = self._get_next_chronic()
        # thermal limits from the FIRST chronic only (q1; game.py:301-304)
        self.thermal_limits = np.asarray(self.chronic.get_imaps() if thermal_limits is None else thermal_limits,
                                         dtype=np.float64)
        self.n_soft_overflowed = np.zeros(c.nl)
        self.current_timestep_id = None
        self.current_entries = None
        self.current_date = None
        self.initial_line_status = self.line_status.copy()
        self.initial_vm, self.initial_va = self.vm.copy(), self.va.copy()
        self.reconnectable = np.zeros(c.nl)
        self.line_cooldown = np.zeros(c.nl)
        self.node_cooldown = np.zeros(c.nS)
        self.epoch = 1
        self.last_depth = 0
        self.n_solves = 0
        self.last_info = None
        self.load_entries_from_next_timestep()
        self.compute_loadflow_cascading()

    # ---- chronic looping (chronic.py:283-292, game.py:395-400) ------------------------------------
    def _get_next_chronic(self):
        ch = self.chronics[self.next_chronic_slot]
        self.current_chronic_slot = self.next_chronic_slot
        if self.looping_mode == 'natural':
            self.next_chronic_slot = (self.next_chronic_slot + 1) % len(self.chronics)
        elif self.looping_mode == 'random':
            self.next_chronic_slot = np.random.choice(len(self.chronics))
        self.current_timestep_id = 0
        return ch

    # ---- grid helpers ------------------------------------------------------------------------------
    def _row(self, sub, node):
        return sub + node * self.case.nS

    def _online_rows(self):
        """bool[2nS]: bus row is an endpoint of at least one on-line line (grid.py:197-204)."""
        c = self.case
        on = self.line_status != 0
        touched = np.zeros(2 * c.nS, dtype=bool)
        touched[self._row(c.or_sub[on], self.or_nodes[on])] = True
        touched[self._row(c.ex_sub[on], self.ex_nodes[on])] = True
        return touched

    def isolated_masks(self):
        """(are_isolated_loads[nL], are_isolated_prods[nP]) in load / production order."""
        c = self.case
        iso = ~self._online_rows()
        return iso[self._row(c.load_sub, self.loads_nodes)], iso[self._row(c.gen_sub, self.prods_nodes)]

    def _sync_bus_types(self):
        c = self.case
        iso = ~self._online_rows()
        gen_rows = self._row(c.gen_sub, self.prods_nodes)
        has_gen_row = np.zeros(2 * c.nS, dtype=bool)
        has_gen_row[gen_rows] = True
        slack_row = c.slack_row
        slack_target = slack_row
        if iso[slack_row]:  # grid.py:159-160: first production bus that is not the slack id
            others = gen_rows[gen_rows != slack_row]
            slack_target = others[0]
        t = np.where(has_gen_row, 2, 1)
        if has_gen_row[slack_target]:
            t[slack_target] = 3
        t[iso] = 4
        self.bus_type = t.astype(np.int32)

    def _build_mpc(self):
        c = self.case
        nS = c.nS
        bus = np.zeros((2 * nS, 13))
        for r in range(2 * nS):
            bus[r, 0] = c.bus_row_id(r % nS, r // nS)
        bus[:, 1] = self.bus_type
        lrows = self._row(c.load_sub, self.loads_nodes)
        bus[lrows, 2] = self.pd
        bus[lrows, 3] = self.qd
        bus[:, 4], bus[:, 5] = c.bus_gs, c.bus_bs
        bus[:, 6] = 1
        bus[:, 7], bus[:, 8], bus[:, 9] = self.vm, self.va, c.bus_basekv
        gen = np.zeros((c.nP, 21))
        gen[:, 0] = bus[self._row(c.gen_sub, self.prods_nodes), 0]
        gen[:, 1], gen[:, 2], gen[:, 3], gen[:, 4] = self.pg, self.qg, c.gen_qmax, c.gen_qmin
        gen[:, 5], gen[:, 6], gen[:, 7] = self.vg, 100., self.gen_status
        br = np.zeros((c.nl, 17))
        br[:, 0] = bus[self._row(c.or_sub, self.or_nodes), 0]
        br[:, 1] = bus[self._row(c.ex_sub, self.ex_nodes), 0]
        br[:, 2], br[:, 3], br[:, 4] = c.br_r, c.br_x, c.br_b
        br[:, 5] = br[:, 6] = br[:, 7] = self.thermal_limits
        br[:, 8], br[:, 9], br[:, 10] = c.br_tap, c.br_shift, self.line_status
        br[:, 13:17] = self.flows
        return bus, gen, br

    def compute_loadflow(self):
        """grid.py:244-264.  The result overwrites the state BEFORE the success test (q7)."""
        self._sync_bus_types()
        bus, gen, br = self._build_mpc()
        r = self.rules
        info = pf_np.SolveInfo()
        alg = pf_np.ALG_NEWTON if r.solver == 'newton' else pf_np.ALG_FDXB
        self.n_solves += 1
        try:
            (bus, gen, br), success = pf_np.runpf(self.case.baseMVA, bus, gen, br, dc=r.dc, alg=alg, tol=r.tol,
                                                  max_it=r.max_it, info=info)
        except (RuntimeError, RuntimeWarning, IndexError, ValueError):
            raise Diverged('The grid is not connexe')
        self.last_info = info
        self.vm, self.va = bus[:, 7].copy(), bus[:, 8].copy()
        self.pg, self.qg = gen[:, 1].copy(), gen[:, 2].copy()
        self.flows = br[:, 13:17].copy()
        pd_rows = bus[:, 2]
        bad = lambda a: np.isnan(a).any() or np.any(a > 1e10)   # grid.py:103-110
        if (not success) or bad(bus[:, 7:9]) or bad(self.flows) or bad(pd_rows):
            raise Diverged('Power grid outage')

    def extract_flows_a(self):
        c = self.case
        rows = self._row(c.or_sub, self.or_nodes)
        v = self.vm[rows] * c.bus_basekv[rows]
        out = np.zeros(c.nl)
        on = self.line_status != 0
        p, q = self.flows[:, 0], self.flows[:, 1]
        out[on] = 1000. * np.sqrt(p[on] ** 2 + q[on] ** 2) / (3. ** .5 * v[on])
        return out

    def normalize_prods_voltages(self, voltages):
        """grid.py:266-271 incl. quirk q9: the divisor list is built in BUS ROW order."""
        c = self.case
        v = np.array(voltages, copy=True)
        v[v <= 0] = 0.
        gen_rows = np.sort(self._row(c.gen_sub, self.prods_nodes))
        return np.asarray(v / c.bus_basekv[gen_rows])

    def load_timestep_injections(self, entries, prods_p=None, prods_v=None, loads_p=None, loads_q=None):
        prods_p = entries.get_prods_p() if prods_p is None else prods_p
        prods_v = entries.get_prods_v() if prods_v is None else prods_v
        loads_p = entries.get_loads_p() if loads_p is None else loads_p
        loads_q = entries.get_loads_q() if loads_q is None else loads_q
        self.pg = np.asarray(prods_p, dtype=np.float64).copy()
        self.vg = self.normalize_prods_voltages(prods_v).astype(np.float64)
        self.gen_status = (np.asarray(prods_v) > 0).astype(np.int64)
        self.pd = np.asarray(loads_p, dtype=np.float64).copy()
        self.qd = np.asarray(loads_q, dtype=np.float64).copy()

    # ---- game.py:405-501 ---------------------------------------------------------------------------
    def load_entries_from_timestep_id(self, timestep_id, is_simulation=False):
        e = self.chronic.get_timestep_entries(timestep_id)
        if not is_simulation:
            self.current_entries = e
            self.load_timestep_injections(e)
        else:
            ce = self.current_entries
            self.load_timestep_injections(e, prods_p=ce.get_planned_prods_p(), prods_v=ce.get_planned_prods_v(),
                                          loads_p=ce.get_planned_loads_p(), loads_q=ce.get_planned_loads_q())
        m = np.asarray(e.get_maintenance())
        mask = m > 0
        self.line_status[mask] = 0
        self.reconnectable[mask] = np.maximum(self.reconnectable[mask], m[mask])
        if not is_simulation:
            h = np.asarray(e.get_hazards())
            mask = h > 0
            self.line_status[mask] = 0
            self.reconnectable[mask] = np.maximum(self.reconnectable[mask], h[mask])
        self.current_timestep_id = timestep_id
        self.current_date = e.get_datetime()

    def load_entries_from_next_timestep(self, is_simulation=False):
        ids = self.chronic.get_timestep_ids()   # NB: list of the chronic in place BEFORE a roll-over (q2)
        if self.current_timestep_id == ids[-1] and not is_simulation:
            self.chronic = self._get_next_chronic()
        if self.current_timestep_id is None:
            nxt = ids[0]
        else:
            nxt = ids[min(ids.index(self.current_timestep_id) + 1, len(ids) - 1)]
        if not is_simulation:
            self.reconnectable[self.reconnectable > 0] -= 1
            self.line_cooldown[self.line_cooldown > 0] -= 1
            self.node_cooldown[self.node_cooldown > 0] -= 1
        self.load_entries_from_timestep_id(nxt, is_simulation)

    # ---- game.py:503-589 ---------------------------------------------------------------------------
    def compute_loadflow_cascading(self):
        r = self.rules
        depth = 0
        over = np.full(self.case.nl, False)
        done = False
        while not done:
            done = True
            self.last_depth = depth
            self.compute_loadflow()   # raises Diverged
            flows_a = self.extract_flows_a()
            lim = self.thermal_limits
            over = flows_a > lim
            if np.sum(over) == 0:
                break
            hard = flows_a > r.hard_coef * lim
            if np.any(hard):
                self.line_status[hard] = 0
                self.reconnectable[hard] = r.n_hard_broken
                done = False
            over[hard] = False
            if np.any(over):
                soft = np.logical_and(over, self.n_soft_overflowed >= r.n_soft_consecutive)
                if np.any(soft):
                    self.line_status[soft] = 0
                    self.reconnectable[soft] = r.n_soft_broken
                    done = False
                    over[soft] = False
            depth += 1
        self.n_soft_overflowed[over] += 1
        self.n_soft_overflowed[~over] = 0

    # ---- actions -------------------------------------------------------------------------------------
    def split_action(self, action):
        c = self.case
        a = np.asarray(action).astype(np.int64)
        if len(a) != c.action_length:
            raise ValueError('Expected action as a binary array of length %d, got %d' % (c.action_length, len(a)))
        o = 0
        parts = []
        for n in (c.nP, c.nL, c.nl, c.nl, c.nl):
            parts.append(a[o:o + n].copy())
            o += n
        return parts

    def changed_substations(self, parts):
        c = self.case
        topo = np.concatenate(parts[:4])
        changed = np.zeros(c.nS, dtype=bool)
        np.logical_or.at(changed, c.elem_sub, topo != 0)
        return changed

    def verify_illegal_action(self, parts):
        r = self.rules
        subs = self.changed_substations(parts)
        lines = parts[4] == 1
        ns, nln = int(subs.sum()), int(lines.sum())
        if ns > r.max_subs or nln > r.max_lines or ns + nln > r.max_total:
            raise Illegal(ILL_TOO_MANY)
        bits = 0
        broken = np.logical_and(lines, self.reconnectable > 0)
        line_cd = np.logical_and(lines, self.line_cooldown > 0)
        node_cd = np.logical_and(subs, self.node_cooldown > 0)
        if broken.any():
            bits |= ILL_BROKEN
        if line_cd.any():
            bits |= ILL_LINE_COOLDOWN
        if node_cd.any():
            bits |= ILL_NODE_COOLDOWN
        return Illegal(bits, broken if broken.any() else None, line_cd if line_cd.any() else None,
                       node_cd if node_cd.any() else None)

    def is_action_valid(self, action):
        try:
            e = self.verify_illegal_action(self.split_action(action))
        except (Illegal, ValueError):
            return False
        return e.bits == 0

    def apply_action(self, parts):
        e = self.verify_illegal_action(parts)
        if e.bits:
            raise e
        r = self.rules
        ap, al, ao, ae, als = parts
        self.prods_nodes = np.where(ap, 1 - self.prods_nodes, self.prods_nodes)
        self.loads_nodes = np.where(al, 1 - self.loads_nodes, self.loads_nodes)
        self.or_nodes = np.where(ao, 1 - self.or_nodes, self.or_nodes)
        self.ex_nodes = np.where(ae, 1 - self.ex_nodes, self.ex_nodes)
        self.line_status = np.where(als, 1 - self.line_status, self.line_status)
        self.line_cooldown[als == 1] = r.n_line_cooldown
        self.node_cooldown[self.changed_substations(parts)] = r.n_node_cooldown

    # ---- step / simulate / game over ----------------------------------------------------------------
    def step(self, action, _is_simulation=False):
        parts = action if isinstance(action, list) else self.split_action(action)
        try:
            self.apply_action(parts)
        except Illegal as e:
            if e.bits & ILL_TOO_MANY:
                parts = [np.zeros_like(p) for p in parts]
            else:
                if e.broken is not None:
                    parts[4][e.broken] = 0
                if e.line_cd is not None:
                    parts[4][e.line_cd] = 0
                if e.node_cd is not None:
                    c = self.case
                    kill = e.node_cd[c.elem_sub]
                    o = 0
                    for k, n in enumerate((c.nP, c.nL, c.nl, c.nl)):
                        parts[k][kill[o:o + n]] = 0
                        o += n
            obs, flag, illegal, done = self.step(parts, _is_simulation=_is_simulation)
            return obs, flag, e.bits, done
        try:
            self.load_entries_from_next_timestep(is_simulation=_is_simulation)
            self.compute_loadflow_cascading()
        except Diverged:
            return None, FLAG_DIVERGED, 0, True
        iso_l, iso_p = self.isolated_masks()
        if np.sum(iso_l) > self.rules.max_loads_cut:      # q10: consumptions first
            return None, FLAG_TOO_MANY_LOADS, 0, True
        if np.sum(iso_p) > self.rules.max_prods_cut:
            return None, FLAG_TOO_MANY_PRODS, 0, True
        return self.export_observation(), FLAG_OK, 0, False

    _SNAP = ['prods_nodes', 'loads_nodes', 'or_nodes', 'ex_nodes', 'line_status', 'vm', 'va', 'pg', 'qg', 'vg',
             'gen_status', 'pd', 'qd', 'flows', 'bus_type', 'n_soft_overflowed', 'reconnectable', 'line_cooldown',
             'node_cooldown', 'current_timestep_id', 'current_date']

    def simulate(self, action):
        snap = {k: copy.deepcopy(getattr(self, k)) for k in self._SNAP}
        try:
            return self.step(action, _is_simulation=True)
        finally:
            for k, v in snap.items():
                setattr(self, k, v)

    def reset_grid(self):
        c = self.case
        self.reconnectable = np.zeros(c.nl)
        self.line_cooldown = np.zeros(c.nl)
        self.node_cooldown = np.zeros(c.nS)
        self.prods_nodes[:] = 0
        self.loads_nodes[:] = 0
        self.or_nodes[:] = 0
        self.ex_nodes[:] = 0
        self.gen_status[:] = 1
        self.line_status = self.initial_line_status.copy()
        self.va, self.vm = self.initial_va.copy(), self.initial_vm.copy()
        self.flows = np.zeros((c.nl, 4))   # mpc stripped to 5 keys; branch keeps 17 columns in practice
        # q3: n_soft_overflowed is NOT cleared

    def process_game_over(self):
        self.reset_grid()
        self.epoch += 1
        if self.game_over_mode == 'hard':
            self.current_timestep_id = None
            self.chronic = self._get_next_chronic()
        try:
            self.load_entries_from_next_timestep()
            self.compute_loadflow_cascading()
        except Diverged:
            self.process_game_over()

    # ---- observation --------------------------------------------------------------------------------
    def export_observation(self):
        c = self.case
        iso_l, iso_p = self.isolated_masks()
        or_rows = self._row(c.or_sub, self.or_nodes)
        ex_rows = self._row(c.ex_sub, self.ex_nodes)
        ce = self.current_entries
        d = self.current_date
        o = dict(
            substations_ids=c.sub_ids.astype(np.int64),
            active_loads=self.pd.copy(), reactive_loads=self.qd.copy(),
            voltage_loads=self.vm[self._row(c.load_sub, self.loads_nodes)],
            active_productions=self.pg.copy(), reactive_productions=self.qg.copy(),
            voltage_productions=self.vg.copy(),
            active_flows_origin=self.flows[:, 0].copy(), reactive_flows_origin=self.flows[:, 1].copy(),
            voltage_flows_origin=self.vm[or_rows],
            active_flows_extremity=self.flows[:, 2].copy(), reactive_flows_extremity=self.flows[:, 3].copy(),
            voltage_flows_extremity=self.vm[ex_rows],
            ampere_flows=self.extract_flows_a(), thermal_limits=self.thermal_limits.copy(),
            lines_status=self.line_status.astype(np.int64).copy(),
            are_loads_cut=iso_l.copy(), are_productions_cut=iso_p.copy(),
            loads_substations_ids=c.sub_ids[c.load_sub], productions_substations_ids=c.sub_ids[c.gen_sub],
            lines_or_substations_ids=c.sub_ids[c.or_sub], lines_ex_substations_ids=c.sub_ids[c.ex_sub],
            timesteps_before_lines_reconnectable=self.reconnectable.copy(),
            timesteps_before_lines_reactionable=self.line_cooldown.copy(),
            timesteps_before_nodes_reactionable=self.node_cooldown.copy(),
            timesteps_before_planned_maintenance=self.chronic.get_planned_maintenance(self.current_timestep_id,
                                                                                      self.rules.horizon),
            planned_active_loads=ce.get_planned_loads_p(), planned_reactive_loads=ce.get_planned_loads_q(),
            planned_active_productions=ce.get_planned_prods_p(),
            planned_voltage_productions=self.normalize_prods_voltages(ce.get_planned_prods_v()),
            date_year=d.year, date_month=d.month, date_day=d.day, date_hour=d.hour, date_minute=d.minute,
            date_second=d.second,
            productions_nodes=self.prods_nodes.copy(), loads_nodes=self.loads_nodes.copy(),
            lines_or_nodes=self.or_nodes.copy(), lines_ex_nodes=self.ex_nodes.copy(),
            initial_productions_nodes=np.zeros(c.nP), initial_loads_nodes=np.zeros(c.nL),
            initial_lines_or_nodes=np.zeros(c.nl), initial_lines_ex_nodes=np.zeros(c.nl))
        return o


OBS_ARRAY_ORDER = [
    # MinimalistObservation.as_array (environment.py:451-466)
    'active_loads', 'are_loads_cut', 'planned_active_loads', 'loads_nodes',
    'active_productions', 'are_productions_cut', 'planned_active_productions', 'productions_nodes',
    'lines_or_nodes', 'lines_ex_nodes', 'ampere_flows', 'lines_status',
    'timesteps_before_lines_reconnectable', 'timesteps_before_lines_reactionable',
    'timesteps_before_nodes_reactionable', 'timesteps_before_planned_maintenance',
    'date_year', 'date_month', 'date_day', 'date_hour', 'date_minute', 'date_second',
    # MinimalistACObservation.as_array (environment.py:511-517)
    'reactive_loads', 'voltage_loads', 'reactive_productions', 'voltage_productions',
    'active_flows_origin', 'reactive_flows_origin', 'voltage_flows_origin',
    'active_flows_extremity', 'reactive_flows_extremity', 'voltage_flows_extremity',
    'planned_reactive_loads', 'planned_voltage_productions',
    # Observation.as_array (environment.py:583-595)
    'substations_ids', 'loads_substations_ids', 'productions_substations_ids', 'lines_or_substations_ids',
    'lines_ex_substations_ids', 'thermal_limits', 'initial_productions_nodes', 'initial_loads_nodes',
    'initial_lines_or_nodes', 'initial_lines_ex_nodes']


def obs_as_array(o):
    return np.concatenate([np.atleast_1d(np.asarray(o[k], dtype=np.float64)).flatten() for k in OBS_ARRAY_ORDER])
