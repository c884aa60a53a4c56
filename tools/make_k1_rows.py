#!/usr/bin/env python
"""K1-style known-answer rows on IEEE-118 (build container only; needs /root/reference).

The reference's own tests pin its numeric layer with ONE series: the int-truncated ampere flow of line 6 over 15 do-nothing
steps on IEEE-14 (`tests/test_core.py:919`).  This tool records the same kind of rows on default118: the
reference's unmodified RunEnv (imported in place, tools/make_reference_fixtures.py's stand-ins for gym and pypower.api), the
do-nothing agent, the reference's solver (fast-decoupled XB), shipped thermal limits, `int(ampere_flows)` of all 186 lines after
every step -> tests/golden/k1_rows/default118_do_nothing_k1_rows.npz (data only).

    python tools/make_k1_rows.py              # record (pypower.api = oracle/pf_np.py, as every recording here)
    python tools/make_k1_rows.py --check      # a container WITH PYPOWER 5.1.4 and gym installed: drive the REAL reference stack and
                                              # compare with the committed rows -- the one-command check of the numeric layer on
                                              # IEEE-118 this container cannot make
"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import make_reference_fixtures as mrf      # noqa: E402

OUT = os.path.join(mrf.ROOT, 'tests', 'golden', 'k1_rows', 'default118_do_nothing_k1_rows.npz')
STEPS = 60


def run(real_stack):
    spec = dict(env='parameters/default118', fixture_env='default118', mode='soft')
    if real_stack:
        import pypower.api      # noqa: F401  (raises ImportError where it is absent: this mode is for a PYPOWER-equipped container)
        import gym              # noqa: F401
    else:
        mrf.install_stand_ins('fdxb')
    sys.path.insert(0, mrf.REF)
    import logging
    logging.disable(logging.CRITICAL)
    tmp = tempfile.mkdtemp(prefix='ppn_k1_')
    os.chdir(tmp)
    folder, chron_names, _ = mrf.assemble_parameters(tmp, spec)
    import pypownet.environment as renv
    env = renv.RunEnv(parameters_folder=folder, game_level='level0', chronic_looping_mode='natural', start_id=0, game_over_mode='soft')
    rows, done_ = [], []
    act = env.action_space.get_do_nothing_action()
    for _ in range(STEPS):
        obs, _, done, _ = env.step(act)
        if done:
            obs = env.process_game_over()
        o = env.observation_space.array_to_observation(obs)
        rows.append(np.asarray(o.ampere_flows).astype(np.int64))          # int truncation, as tests/test_core.py:919 holds K1
        done_.append(bool(done))
    return np.asarray(rows), np.asarray(done_), chron_names


def main():
    check = '--check' in sys.argv[1:]
    rows, done, names = run(real_stack=check)
    if check:
        ref = np.load(OUT)
        same = np.array_equal(rows, ref['int_amps']) and np.array_equal(done, ref['done'])
        print('reference stack with PYPOWER vs committed rows: %s (largest |difference| %d A over %d x %d entries)'
              % ('IDENTICAL' if same else 'DIFFERENT', int(np.abs(rows - ref['int_amps']).max()), rows.shape[0], rows.shape[1]))
        sys.exit(0 if same else 1)
    np.savez_compressed(OUT, int_amps=rows.astype(np.int32), done=done, chronics=np.asarray(names), steps=np.int32(STEPS),
                        note=np.asarray('reference RunEnv, do-nothing, default118 level0, FDXB, shipped limits; pypower.api = oracle/pf_np.py'))
    print('%s: %d steps x %d lines, %d game overs, max %d A' % (os.path.basename(OUT), rows.shape[0], rows.shape[1], int(done.sum()), int(rows.max())))


if __name__ == '__main__':
    main()
