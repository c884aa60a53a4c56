#!/usr/bin/env python
"""Step rate of the drop-in single-environment API (RunEnv over an Engine with batch 1) on the GPU: do-nothing agent,
observation returned to the host every step like the reference.  Usage: python tools/runenv_rate.py [env] [steps]"""
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
from pypownet_amd.environment import RunEnv  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'default14'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
env = RunEnv(os.path.join(ROOT, 'tests', 'golden', 'envs', name), 'level0')
a = env.action_space.get_do_nothing_action()
env.step(a)
for how in ('process_game_over', 'reset'):
    # on a game over the reference's Runner calls process_game_over (runner.py:81-96); reset() re-instantiates the Game
    # (environment.py:814-821: parameters, case, chronics, engine) and is there for callers that do that instead
    t = time.perf_counter()
    n_done = 0
    for _ in range(steps):
        obs, r, done, flag = env.step(a)
        if done:
            n_done += 1
            getattr(env, how)()
    el = time.perf_counter() - t
    print('%s: %d RunEnv.step calls in %.2f s = %.0f steps/s (%d game overs, each followed by %s())' % (name, steps, el, steps / el, n_done, how))
