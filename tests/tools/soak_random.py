#!/usr/bin/env python
"""Long random-action lock-step of the GPU engine against the C oracle (node splitting, line switching, illegal-action
repair, islanding, restarts; every busbar may be active): tests/engine_checks.check_random_actions_vs_c_oracle at soak
size, in chunks with fresh seeds so that the (rare) environments dropped as numerically degenerate do not accumulate.
Usage (GPU box): python tests/tools/soak_random.py [env] [solver] [batch] [steps_per_chunk] [chunks]"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import engine_checks as ec  # noqa: E402


def main():
    env = sys.argv[1] if len(sys.argv) > 1 else 'default118'
    solver = sys.argv[2] if len(sys.argv) > 2 else 'newton'
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 60
    chunks = int(sys.argv[5]) if len(sys.argv) > 5 else 5
    lib = os.path.join(ROOT, 'pypownet_amd', 'libppn.so')
    tot = dict(done=0, illegal=0, split_buses=0, dropped=0, excused=0, rejoined=0)
    for c in range(chunks):
        st = ec.check_random_actions_vs_c_oracle(lib, env, steps, batch, solver, seed=1000 + 17 * c, max_dropped=batch // 16)
        for k in tot:
            tot[k] = max(tot[k], st[k]) if k == 'split_buses' else tot[k] + st[k]
        print('chunk %d: %s' % (c, st), flush=True)
    print('%s %s: %d environments x %d steps x %d chunks in lock-step with the oracle: %s' % (env, solver, batch, steps, chunks, tot))


if __name__ == '__main__':
    main()
