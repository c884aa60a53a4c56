#!/usr/bin/env python
"""Golden runs of the REFERENCE'S OWN game-rule code (build container only; needs /root/reference).

The reference's ``pypownet.environment/game/grid`` cannot be imported as shipped: ``gym`` (environment.py:9) and
``pypower`` (grid.py:62) are absent.  This script puts two in-memory stand-in MODULES into ``sys.modules`` --

  * ``gym.spaces``   four empty container classes (MultiBinary, Box, Dict, Discrete): the reference only subclasses them and
                     reads ``.spaces`` / ``.shape`` back;
  * ``pypower.api``  ``ppoption / loadcase / runpf / rundcpf / savecase`` routed to ``oracle/pf_np.py`` (the restatement of
                     PYPOWER 5.1.4, pinned by K1-K4), same call signatures and exception classes as the call sites
                     pypownet/grid.py:62-65, 227-231, 595;

imports ``/root/reference/pypownet`` IN PLACE (nothing is copied, nothing of it is written anywhere) and drives the reference's
``RunEnv`` -- its Grid / Game / Action / Observation / reward code, unmodified -- with seeded random actions in the style of
``RandomNodeSplitting`` / ``RandomLineSwitch`` (pypownet/agent.py:60-158).  What is recorded per step is DATA: the action, what
``RunEnv.step`` returned (observation array, reward list, done, flag class and its illegal-action masks), and the game state
behind it (line status, node vectors, the four counters, chronic name + timestep id, cut counts, bus types / voltages /
flows of ``grid.mpc``), before and after ``process_game_over``.  Output: ``tests/golden/reference_runs/<scenario>.npz``.

WHAT THIS PINS: the ~2 700 lines of game rules (game.py:405-885, grid.py:141-209, 266-566, environment.py:406-601, the shipped
reward signals) on default14 / default30 / default118, soft and hard game-over mode -- with the reference's own code as the
authority.  WHAT IT DOES NOT PIN: the numeric layer.  The solver underneath is oracle/pf_np.py, so voltages and flows in these
files are the restatement's; PYPOWER's arithmetic stays pinned by the reference-held K1-K4 only.

Every scenario runs in its own process (the reference caches ``reward_signal`` in sys.modules and seeds numpy globally).
Input folders are assembled under a temporary directory from symlinks into /root/reference/parameters so that a scenario sees
exactly the chronic set the committed fixture environment (tests/golden/envs/<env>) holds, optionally with other thermal
limits (the shipped flat 2000 A of default118 never cascades) or ``loadflow_mode: DC``.
"""
import json
import os
import subprocess
import sys
import tempfile
import types

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
REF = '/root/reference'
OUT = os.path.join(ROOT, 'tests', 'golden', 'reference_runs')
ENVS = os.path.join(ROOT, 'tests', 'golden', 'envs')

FLAG_NONE, FLAG_DIVERGED, FLAG_TOO_MANY_LOADS, FLAG_TOO_MANY_PRODS, FLAG_ILLEGAL = 0, 1, 2, 3, 4

# name -> spec.  ``env``: folder under /root/reference/parameters (or tests/parameters); ``fixture_env``: the committed
# environment folder a replay loads; ``limits``: key of tests/golden/envs/<fixture_env>/bench_limits.json or None
SCENARIOS = {
    # ``mix``: 'gentle' keeps episodes long (cooldowns, maintenance, soft-overflow counters get to act), 'wild' ends them fast
    'default14_soft':        dict(env='parameters/default14', fixture_env='default14', mode='soft', steps=400, seed=11),
    'default14_wild_soft':   dict(env='parameters/default14', fixture_env='default14', mode='soft', steps=250, seed=17,
                                  mix='wild'),
    'default14_hard':        dict(env='parameters/default14', fixture_env='default14', mode='hard', steps=300, seed=12),
    'default14_dc_soft':     dict(env='parameters/default14', fixture_env='default14', mode='soft', steps=250, seed=13,
                                  conf={'loadflow_mode': 'DC'}),
    'default14_newton_soft': dict(env='parameters/default14', fixture_env='default14', mode='soft', steps=250, seed=14,
                                  solver='newton'),
    'hard_overflow14_soft':  dict(env='tests/parameters/default14_for_tests_hard_overflow',
                                  fixture_env='default14_for_tests_hard_overflow', mode='soft', steps=200, seed=15),
    'alpha14_hard':          dict(env='tests/parameters/default14_for_tests_alpha', fixture_env='default14_for_tests_alpha',
                                  mode='hard', steps=200, seed=16, mix='wild'),
    'default30_soft':        dict(env='parameters/default30', fixture_env='default30', mode='soft', steps=250, seed=21),
    'default30_hard':        dict(env='parameters/default30', fixture_env='default30', mode='hard', steps=250, seed=22,
                                  mix='wild'),
    'default118_soft':       dict(env='parameters/default118', fixture_env='default118', mode='soft', steps=200, seed=31),
    'default118_wild_soft':  dict(env='parameters/default118', fixture_env='default118', mode='soft', steps=120, seed=38,
                                  mix='wild'),
    'default30_wild_hard':   dict(env='parameters/default30', fixture_env='default30', mode='hard', steps=150, seed=23,
                                  mix='wild'),
    'default118_hard':       dict(env='parameters/default118', fixture_env='default118', mode='hard', steps=200, seed=32,
                                  mix='wild'),
    'default118_tight_soft': dict(env='parameters/default118', fixture_env='default118', mode='soft', steps=200, seed=33,
                                  limits='limits_a'),
    'default118_tight_hard': dict(env='parameters/default118', fixture_env='default118', mode='hard', steps=200, seed=34,
                                  limits='limits_a'),
    'default118_tight_newton_soft': dict(env='parameters/default118', fixture_env='default118', mode='soft', steps=200,
                                         seed=35, limits='limits_a', solver='newton'),
    'default118_tight_newton_hard': dict(env='parameters/default118', fixture_env='default118', mode='hard', steps=200,
                                         seed=36, limits='limits_a', solver='newton', mix='wild'),
    'default118_dc_soft':    dict(env='parameters/default118', fixture_env='default118', mode='soft', steps=200, seed=37,
                                  limits='limits_a', conf={'loadflow_mode': 'DC'}),
    # the other constructor arguments of RunEnv (environment.py:789-791): a fixed chronic from a start id, overflow cut-off disabled
    'default14_fixed_start3_hard': dict(env='parameters/default14', fixture_env='default14', mode='hard', steps=200, seed=51,
                                        looping='fixed', start_id=3),
    'default14_natural_start7_hard': dict(env='parameters/default14', fixture_env='default14', mode='hard', steps=200, seed=54,
                                          start_id=7),
    'hard_overflow14_nocutoff_soft': dict(env='tests/parameters/default14_for_tests_hard_overflow',
                                          fixture_env='default14_for_tests_hard_overflow', mode='soft', steps=120, seed=53,
                                          without_overflow_cutoff=True),
    # Game.simulate (game.py:887-943) interleaved with the steps: a candidate action is simulated before about every third step
    # (planned injections of the CURRENT entry, no hazards, nothing may leak into the game) and its result recorded as well
    'default14_simulate_soft':  dict(env='parameters/default14', fixture_env='default14', mode='soft', steps=250, seed=41,
                                     simulate=True),
    'default30_simulate_hard':  dict(env='parameters/default30', fixture_env='default30', mode='hard', steps=200, seed=42,
                                     simulate=True),
    'default118_simulate_soft': dict(env='parameters/default118', fixture_env='default118', mode='soft', steps=150, seed=43,
                                     limits='limits_a', solver='newton', simulate=True),
}


# ---- stand-in modules ---------------------------------------------------------------------------------------------------
def install_stand_ins(solver):
    """gym.spaces and pypower.api as in-memory modules (never written to disk, never shipped)."""
    sys.path.insert(0, ROOT)
    from oracle import pf_np

    gym, spaces = types.ModuleType('gym'), types.ModuleType('gym.spaces')

    class MultiBinary(object):
        def __init__(self, n):
            self.n, self.shape = n, (n,)

    class Box(object):
        def __init__(self, low=None, high=None, shape=None, dtype=None):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

    class Discrete(object):
        def __init__(self, n):
            self.n, self.shape = n, ()

    class Dict(object):
        def __init__(self, spaces_):
            self.spaces = spaces_

    spaces.MultiBinary, spaces.Box, spaces.Discrete, spaces.Dict = MultiBinary, Box, Discrete, Dict
    gym.spaces = spaces
    sys.modules['gym'], sys.modules['gym.spaces'] = gym, spaces

    pypower, api = types.ModuleType('pypower'), types.ModuleType('pypower.api')

    def ppoption(**kw):                       # grid.py:63-64
        opt = dict(PF_ALG=1, PF_TOL=1e-8, PF_MAX_IT=10, PF_MAX_IT_FD=30, PF_DC=False, VERBOSE=1, OUT_ALL=-1)
        opt.update(kw)
        if solver == 'newton':                # the headline solver; the reference itself hard-wires PF_ALG=2
            opt['PF_ALG'] = 1
        return opt

    def loadcase(casefile, expect_gencost=True):   # grid.py:65
        import copy
        if isinstance(casefile, dict):
            return copy.deepcopy(casefile)
        name = os.path.splitext(os.path.basename(casefile))[0]
        scope = {'array': np.array}                 # PYPOWER's loadcase execs the file with numpy's array in scope
        with open(casefile if casefile.endswith('.py') else casefile + '.py') as f:
            exec(compile(f.read(), casefile, 'exec'), scope)
        return scope[name]()

    def _run(mpc, opt, dc):
        ppc = loadcase(mpc)
        alg = pf_np.ALG_NEWTON if opt['PF_ALG'] == 1 else pf_np.ALG_FDXB
        max_it = opt['PF_MAX_IT'] if alg == pf_np.ALG_NEWTON else opt['PF_MAX_IT_FD']
        (bus, gen, branch), success = pf_np.runpf(ppc['baseMVA'], ppc['bus'], ppc['gen'], ppc['branch'], dc=dc, alg=alg,
                                                  tol=opt['PF_TOL'], max_it=max_it)
        ppc['bus'], ppc['gen'], ppc['branch'] = bus, gen, branch
        ppc['success'] = 1 if success else 0
        ppc['et'] = 0.0
        return ppc, ppc['success']

    def runpf(casedata=None, ppopt=None, fname='', solvedcase=''):       # grid.py:227-229
        assert fname == '' and solvedcase == ''
        return _run(casedata, ppopt, bool(ppopt.get('PF_DC', False)))

    def rundcpf(casedata=None, ppopt=None, fname='', solvedcase=''):
        assert fname == '' and solvedcase == ''
        return _run(casedata, ppopt, True)

    def savecase(*a, **k):                                                # grid.py:595 (save_io is False)
        raise RuntimeError('savecase is not expected on this path')

    api.ppoption, api.loadcase, api.runpf, api.rundcpf, api.savecase = ppoption, loadcase, runpf, rundcpf, savecase
    pypower.api = api
    sys.modules['pypower'], sys.modules['pypower.api'] = pypower, api


# ---- input folders --------------------------------------------------------------------------------------------------------
def assemble_parameters(tmp, spec):
    """<tmp>/<envname>/{reward_signal.py, level0/{configuration.yaml, reference_grid.py, reference_grid.m, chronics/<c>/}}."""
    import yaml
    src_env = os.path.join(REF, spec['env'])
    src = os.path.join(src_env, 'level0')
    name = os.path.basename(spec['env'])
    dst_env = os.path.join(tmp, name)
    dst = os.path.join(dst_env, 'level0')
    os.makedirs(os.path.join(dst, 'chronics'))
    if os.path.exists(os.path.join(src_env, 'reward_signal.py')):
        os.symlink(os.path.join(src_env, 'reward_signal.py'), os.path.join(dst_env, 'reward_signal.py'))
    for f in ('reference_grid.py', 'reference_grid.m'):
        os.symlink(os.path.join(src, f), os.path.join(dst, f))
    with open(os.path.join(src, 'configuration.yaml')) as f:
        conf = yaml.safe_load(f)
    conf.update(spec.get('conf', {}))
    with open(os.path.join(dst, 'configuration.yaml'), 'w') as f:
        yaml.safe_dump(conf, f)
    fix_chron = os.path.join(ENVS, spec['fixture_env'], 'level0', 'chronics')
    names = sorted(os.path.splitext(c)[0] for c in os.listdir(fix_chron))
    limits = None
    if spec.get('limits'):
        with open(os.path.join(ENVS, spec['fixture_env'], 'bench_limits.json')) as f:
            limits = np.asarray(json.load(f)[spec['limits']], dtype=np.float64)
    for c in names:
        csrc = os.path.join(src, 'chronics', c)
        if limits is None:
            os.symlink(csrc, os.path.join(dst, 'chronics', c))
            continue
        cdst = os.path.join(dst, 'chronics', c)
        os.makedirs(cdst)
        for fn in os.listdir(csrc):
            if fn != '_N_imaps.csv':
                os.symlink(os.path.join(csrc, fn), os.path.join(cdst, fn))
        with open(os.path.join(csrc, '_N_imaps.csv')) as f:
            header = f.readline()
        with open(os.path.join(cdst, '_N_imaps.csv'), 'w') as f:
            f.write(header)
            f.write(';'.join('%g' % v for v in limits) + '\n')
    return dst_env, names, limits


# ---- the driver of one scenario ---------------------------------------------------------------------------------------------
def draw_action(rng, env, step, mix, recent):
    """Seeded action mix built with the reference's own ActionSpace helpers (agent.py:60-158 style).  ``recent``: the
    substations / lines touched lately, re-actioned now and then to run into the cooldown rules."""
    asp = env.action_space
    action = asp.get_do_nothing_action(as_class_Action=True)
    n_lines = asp.lines_status_subaction_length

    def split(sub, p):
        n = asp.get_number_elements_of_substation(sub)
        asp.set_substation_switches_in_action(action=action, substation_id=sub,
                                              new_values=(rng.rand(n) < p).astype(int))
        recent['subs'].append(sub)

    def switch(line):
        asp.set_lines_status_switch_from_id(action=action, line_id=int(line), new_switch_value=1)
        recent['lines'].append(int(line))

    u = rng.rand()
    if mix == 'gentle':
        if u < 0.30:
            return action
        if u < 0.55:                                         # sparse node splitting: few elements move
            split(rng.choice(asp.substations_ids), 0.2)
        elif u < 0.72:                                       # RandomLineSwitch.act
            switch(rng.randint(n_lines))
        elif u < 0.90:                                       # touch something touched a moment ago (cooldowns, broken lines)
            if recent['lines'] and rng.rand() < 0.6:
                switch(recent['lines'][-1 - rng.randint(min(3, len(recent['lines'])))])
            elif recent['subs']:
                split(recent['subs'][-1 - rng.randint(min(3, len(recent['subs'])))], 0.3)
        elif u < 0.96:                                       # one substation + one or two lines
            split(rng.choice(asp.substations_ids), 0.2)
            for _ in range(rng.randint(1, 3)):
                switch(rng.randint(n_lines))
        else:                                                # many elements at once: the activation maxima
            for sub in rng.choice(asp.substations_ids, size=min(rng.randint(3, 9), len(asp.substations_ids)), replace=False):
                split(sub, 0.2)
            for _ in range(rng.randint(0, 4)):
                switch(rng.randint(n_lines))
        return action
    if u < 0.10:
        return action
    if u < 0.62:                                             # RandomNodeSplitting.act
        split(rng.choice(asp.substations_ids), 0.5)
    elif u < 0.80:
        switch(rng.randint(n_lines))
    elif u < 0.92:
        split(rng.choice(asp.substations_ids), 0.5)
        for _ in range(rng.randint(1, 3)):
            switch(rng.randint(n_lines))
    else:
        for sub in rng.choice(asp.substations_ids, size=min(rng.randint(3, 9), len(asp.substations_ids)), replace=False):
            split(sub, 0.5)
        for _ in range(rng.randint(0, 4)):
            switch(rng.randint(n_lines))
    return action


def game_state(game, chron_names):
    """The reference Game / Grid fields a replay is compared with (all read, none written; COPIES -- the reference keeps
    writing into these arrays in place)."""
    grid = game.grid
    mpc = grid.mpc
    topo = grid.get_topology()
    branch = mpc['branch']
    flows = branch[:, 13:17] if branch.shape[1] >= 17 else np.zeros((branch.shape[0], 4))
    return dict(
        line_status=np.array(grid.get_lines_status(), dtype=np.int8),
        prods_nodes=np.array(topo.prods_nodes, dtype=np.int8), loads_nodes=np.array(topo.loads_nodes, dtype=np.int8),
        or_nodes=np.array(topo.lines_or_nodes, dtype=np.int8), ex_nodes=np.array(topo.lines_ex_nodes, dtype=np.int8),
        reconnectable=np.array(game.timesteps_before_lines_reconnectable, dtype=np.int32),
        line_cooldown=np.array(game.timesteps_before_lines_reactionable, dtype=np.int32),
        node_cooldown=np.array(game.timesteps_before_nodes_reactionable, dtype=np.int32),
        soft_count=np.array(game.n_timesteps_soft_overflowed_lines, dtype=np.int32),
        timestep_id=np.int64(-1 if game.current_timestep_id is None else game.current_timestep_id),
        chronic=np.int32(chron_names.index(game.get_current_chronic_name())),
        bus_ids=np.array(mpc['bus'][:, 0], dtype=np.int64), bus_type=np.array(mpc['bus'][:, 1], dtype=np.int8),
        vm=np.array(mpc['bus'][:, 7], dtype=np.float64), va=np.array(mpc['bus'][:, 8], dtype=np.float64),
        pd=np.array(mpc['bus'][:, 2], dtype=np.float64), qd=np.array(mpc['bus'][:, 3], dtype=np.float64),
        gen_bus=np.array(mpc['gen'][:, 0], dtype=np.int64), pg=np.array(mpc['gen'][:, 1], dtype=np.float64),
        qg=np.array(mpc['gen'][:, 2], dtype=np.float64), vg=np.array(mpc['gen'][:, 5], dtype=np.float64),
        gen_status=np.array(mpc['gen'][:, 7], dtype=np.int8),
        f_bus=np.array(branch[:, 0], dtype=np.int64), t_bus=np.array(branch[:, 1], dtype=np.int64),
        flows=np.array(flows, dtype=np.float64))


def run_scenario(name):
    spec = SCENARIOS[name]
    install_stand_ins(spec.get('solver', 'fdxb'))
    sys.path.insert(0, REF)
    import logging
    logging.disable(logging.CRITICAL)
    tmp = tempfile.mkdtemp(prefix='ppn_ref_')
    os.chdir(tmp)                                            # the reference creates tmp/ and log files in the CWD
    folder, chron_names, limits = assemble_parameters(tmp, spec)
    import pypownet.environment as renv
    import pypownet.game as rgame

    env = renv.RunEnv(parameters_folder=folder, game_level='level0', chronic_looping_mode=spec.get('looping', 'natural'),
                      start_id=spec.get('start_id', 0), game_over_mode=spec['mode'],
                      without_overflow_cutoff=spec.get('without_overflow_cutoff', False))
    game = env.game
    rng = np.random.RandomState(spec['seed'])
    nobs = len(env.get_observation())
    rec = dict(action=[], action_after=[], valid=[], done=[], flag=[], ill_too_many=[], ill_broken=[], ill_line_cd=[], ill_node_cd=[], reward=[],
               obs=[], obs_after=[], n_loads_cut=[], n_prods_cut=[], n_restarts=[])
    state_step, state_after = [], []
    initial = game_state(game, chron_names)
    initial_obs = env.get_observation()
    nl, ns = game.grid.n_lines, len(game.substations_ids)
    recent = dict(subs=[], lines=[])
    def classify(flag):
        code, too_many = FLAG_NONE, False
        broken, line_cd, node_cd = np.zeros(nl, bool), np.zeros(nl, bool), np.zeros(ns, bool)
        if isinstance(flag, renv.DivergingLoadflowException):
            code = FLAG_DIVERGED
        elif isinstance(flag, renv.TooManyConsumptionsCut):
            code = FLAG_TOO_MANY_LOADS
        elif isinstance(flag, renv.TooManyProductionsCut):
            code = FLAG_TOO_MANY_PRODS
        elif isinstance(flag, renv.IllegalActionException):
            code = FLAG_ILLEGAL
            too_many = bool(flag.get_has_too_much_activations())
            for dst, src in ((broken, flag.get_illegal_broken_lines_reconnections()),
                             (line_cd, flag.get_illegal_oncoolown_lines_switches()),
                             (node_cd, flag.get_illegal_oncoolown_substations_switches())):
                if src is not None:
                    dst[:] = np.asarray(src, dtype=bool)
        else:
            assert flag is None, flag
        return code, too_many, broken, line_cd, node_cd

    sim = dict(step=[], action=[], action_after=[], done=[], flag=[], ill_too_many=[], ill_counts=[], reward=[], obs=[])
    for step in range(spec['steps']):
        if spec.get('simulate') and rng.rand() < 0.35:
            cand = draw_action(rng, env, step, 'gentle', dict(subs=list(recent['subs']), lines=list(recent['lines'])))
            cand_bits = np.asarray(cand.as_array(), dtype=np.uint8).copy()
            sobs, srew, sdone, sflag = env.simulate(cand, do_sum=False)
            scode, stoo, sbr, slc, snc = classify(sflag)
            sim['step'].append(step), sim['action'].append(cand_bits), sim['done'].append(bool(sdone)), sim['flag'].append(scode)
            sim['action_after'].append(np.asarray(game.last_action.as_array(), dtype=np.uint8).copy())
            sim['ill_too_many'].append(stoo), sim['ill_counts'].append([int(sbr.sum()), int(slc.sum()), int(snc.sum())])
            r = np.full(5, np.nan)
            r[:len(srew)] = srew
            sim['reward'].append(r)
            sim['obs'].append(np.full(nobs, np.nan) if sobs is None else np.asarray(sobs, dtype=np.float64))
        action = draw_action(rng, env, step, spec.get('mix', 'gentle'), recent)
        submitted = np.asarray(action.as_array(), dtype=np.uint8).copy()     # BEFORE step: the repair edits the object
        rec['valid'].append(bool(env.is_action_valid(action)))               # Game.is_action_valid (game.py:755-760): no side effect
        obs, reward, done, flag = env.step(action, do_sum=False)
        code, too_many, broken, line_cd, node_cd = classify(flag)
        rec['action'].append(submitted)
        # the Action object RunEnv built from the submission (environment.py:860), repaired in place by Game.step
        # (game.py:809-846): what the reward signal is given
        rec['action_after'].append(np.asarray(game.last_action.as_array(), dtype=np.uint8).copy())
        rec['done'].append(bool(done))
        rec['flag'].append(code)
        rec['ill_too_many'].append(too_many)
        rec['ill_broken'].append(broken), rec['ill_line_cd'].append(line_cd), rec['ill_node_cd'].append(node_cd)
        r = np.full(5, np.nan)
        r[:len(reward)] = reward
        rec['reward'].append(r)
        rec['obs'].append(np.full(nobs, np.nan) if obs is None else np.asarray(obs, dtype=np.float64))
        rec['n_loads_cut'].append(int(game.n_loads_cut)), rec['n_prods_cut'].append(int(game.n_prods_cut))
        state_step.append(game_state(game, chron_names))
        epoch0 = game.epoch
        if done:                                               # Runner.step protocol (runner.py:81-84)
            env.process_game_over()
        rec['n_restarts'].append(game.epoch - epoch0)         # > 1: process_game_over recursed (game.py:776-780)
        rec['obs_after'].append(np.asarray(env.get_observation(), dtype=np.float64))
        state_after.append(game_state(game, chron_names))

    # Layout.  Integer game state: EVERY step ("step_*" right after RunEnv.step, "after_*" after the process_game_over of a
    # finished episode -- stored for the finished steps only, elsewhere it equals step_*).  Observation arrays and the
    # floating-point grid state: on the SAMPLED steps (every 4th / 10th, plus every step that returned an illegal-action
    # flag) and after the first few restarts.
    FLOATS = ('vm', 'va', 'pd', 'qd', 'pg', 'qg', 'vg', 'flows')
    out = {k: np.asarray(v) for k, v in rec.items() if k not in ('obs', 'obs_after')}
    out['action'] = np.packbits(out['action'].astype(np.uint8), axis=1)
    out['action_after'] = np.packbits(out['action_after'].astype(np.uint8), axis=1)
    steps = np.arange(spec['steps'])
    every = 4 if nobs < 2000 else 10                          # IEEE-118 observations are 4 967 doubles each
    sampled = np.where((steps % every == 0) | ((out['flag'] != 0) & ~out['done']))[0]
    ended = np.where(out['done'])[0]
    ended_sampled = ended[:40 if nobs < 2000 else 12]         # observation / float state after the restart: the first few
    out['sampled_steps'], out['ended_steps'] = sampled.astype(np.int32), ended.astype(np.int32)
    out['obs'] = np.asarray([rec['obs'][k] for k in sampled])
    out['ended_sampled_steps'] = ended_sampled.astype(np.int32)
    out['obs_after'] = np.asarray([rec['obs_after'][k] for k in ended_sampled]).reshape(len(ended_sampled), nobs)
    for k in initial:
        out['init_' + k] = initial[k]
        if k in FLOATS:
            out['step_' + k] = np.asarray([state_step[t][k] for t in sampled])
            out['after_' + k] = np.asarray([state_after[t][k] for t in ended_sampled]).reshape(
                (len(ended_sampled),) + initial[k].shape)
        else:
            out['step_' + k] = np.asarray([st[k] for st in state_step])
            out['after_' + k] = np.asarray([state_after[t][k] for t in ended]).reshape((len(ended),) + np.shape(initial[k]))
    out['init_obs'] = np.asarray(initial_obs, dtype=np.float64)
    # the reduced layouts, from the reference's OWN methods (environment.py:529-530, 597-601: Observation.as_minimalist() /
    # as_ac_minimalist(), each .as_array()): the first sampled observations that exist, through array_to_observation
    red = [k for k in sampled if not np.isnan(rec['obs'][k]).all()][:8]
    out['reduced_steps'] = np.asarray(red, dtype=np.int32)
    objs = [env.observation_space.array_to_observation(rec['obs'][k]) for k in red]
    out['obs_minimalist'] = np.asarray([np.asarray(o.as_minimalist().as_array(), dtype=np.float64) for o in objs])
    out['obs_ac_minimalist'] = np.asarray([np.asarray(o.as_ac_minimalist().as_array(), dtype=np.float64) for o in objs])
    if spec.get('simulate'):
        out['sim_step'] = np.asarray(sim['step'], dtype=np.int32)
        out['sim_action'] = np.packbits(np.asarray(sim['action'], dtype=np.uint8).reshape(len(sim['step']), -1), axis=1)
        out['sim_done'], out['sim_flag'] = np.asarray(sim['done']), np.asarray(sim['flag'])
        out['sim_action_after'] = np.packbits(np.asarray(sim['action_after'], dtype=np.uint8).reshape(len(sim['step']), -1), axis=1)
        out['sim_ill_too_many'] = np.asarray(sim['ill_too_many'])
        out['sim_ill_counts'] = np.asarray(sim['ill_counts'], dtype=np.int32).reshape(len(sim['step']), 3)
        out['sim_reward'] = np.asarray(sim['reward']).reshape(len(sim['step']), 5)
        keep = [k for k in range(len(sim['step'])) if k % 3 == 0 or nobs < 2000]      # simulated observations: all (small cases) / every third
        out['sim_obs_index'] = np.asarray(keep, dtype=np.int32)
        out['sim_obs'] = np.asarray([sim['obs'][k] for k in keep]).reshape(len(keep), nobs)
    out['action_length'] = np.int32(env.action_space.action_length)
    meta = dict(scenario=name, reference_env=spec['env'], fixture_env=spec['fixture_env'], game_over_mode=spec['mode'],
                solver=spec.get('solver', 'fdxb'), mix=spec.get('mix', 'gentle'), looping=spec.get('looping', 'natural'),
                start_id=spec.get('start_id', 0), without_overflow_cutoff=bool(spec.get('without_overflow_cutoff', False)), conf=spec.get('conf', {}), limits=spec.get('limits'),
                chronics=chron_names, seed=spec['seed'], steps=spec['steps'],
                flag_codes=dict(none=0, diverged=1, too_many_loads=2, too_many_prods=3, illegal=4),
                generated_by='tools/make_reference_fixtures.py: the reference\'s RunEnv/Game/Grid imported in place, '
                             'pypower.api routed to oracle/pf_np.py')
    out['meta'] = np.asarray(json.dumps(meta))
    if limits is not None:
        out['thermal_limits'] = limits
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **out)
    d, f = out['done'], out['flag']
    print('%-30s steps %d  game overs %d (div %d, loads %d, prods %d)  illegal %d  restarts>1 %d  %.0f KB' % (
        name, len(d), int(d.sum()), int((f == 1).sum()), int((f == 2).sum()), int((f == 3).sum()), int((f == 4).sum()),
        int((out['n_restarts'] > 1).sum()), os.path.getsize(os.path.join(OUT, name + '.npz')) / 1024.), flush=True)


if __name__ == '__main__':
    if len(sys.argv) == 3 and sys.argv[1] == '--one':
        run_scenario(sys.argv[2])
    else:
        wanted = sys.argv[1:] or sorted(SCENARIOS)
        procs = [(n, subprocess.Popen([sys.executable, os.path.abspath(__file__), '--one', n])) for n in wanted]
        bad = [n for n, p in procs if p.wait() != 0]
        if bad:
            sys.exit('scenarios failed: %s' % bad)
