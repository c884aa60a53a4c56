#!/usr/bin/env python
"""ppn_rollout_policy against the stepped form at the headline's size, several times over: the closed-loop rollout kernel hands an
environment's state from workgroup to workgroup (round 6: XCD-affine, without an L2 write-back per step) -- a hand-over that is wrong
shows up as a trajectory that differs from { ppn_policy_actions; ppn_step } bit for bit.   python tests/tools/soak_policy_rollout.py
[batch] [steps] [repeats]"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import engine_checks as ec  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    R = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    acted = 0
    for r in range(R):
        for kw in (dict(max_active_buses=118), dict()):      # two-word kernels; four-word kernels (every busbar may be active)
            b = B if kw else max(256, B // 4)
            acted += ec.check_policy_rollout_equals_stepping(None, 'default118', batch=b, n_steps=K, params=(0.8 + 0.05 * r,), **kw)
            print('repeat %d %s: %d environments x %d steps identical to the stepped form' % (r, 'two-word' if kw else 'four-word', b, K), flush=True)
    print('soak ok: the policy acted %d times' % acted)


if __name__ == '__main__':
    main()
