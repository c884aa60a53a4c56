#!/usr/bin/env python
"""configs[4]'s workload on one GPU (random node splitting every step, every busbar may be active: four-word kernels), for A/B runs
and profiles:   [PPN_SCHED_PREPASS=0] python tools/split_rate.py [batch] [steps] [tuned|safe]"""
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    import bench
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    tuned = (sys.argv[3] if len(sys.argv) > 3 else 'tuned') == 'tuned'
    case, _, _ = bench.load_env_fixture(bench.ENV_NAME, 'newton')
    out = bench.side_config('configs[4] workload', bench.ENV_NAME, os.environ.get('PPN_SPLIT_SOLVER', 'newton'), B, K, 0, 2, limits=bench.bench_limits(case), split=True,
                            lu_capacity=3976 if tuned else 0, q_plane_auto=1 if tuned else 0)
    out['sched_prepass'] = os.environ.get('PPN_SCHED_PREPASS', '1') != '0'
    print(json.dumps(out))


if __name__ == '__main__':
    main()
