#!/usr/bin/env python
"""Data fixtures from the MATPOWER `.m` twin of every shipped reference grid (build container only: reads /root/reference).

The reference ships `parameters/<env>/level0/reference_grid.m` -- the plain MATPOWER case -- next to the `reference_grid.py` that
`Grid` loads (the same case run through parameters/make_reference_grid.py: '666'-twin rows, productions forced on, flat angles).
The arrays of the `.m` file are written as JSON next to the engine's copy of the `.py` arrays
(tests/golden/envs/<env>/level0/reference_grid_m.json; numbers only, no source text), so that tests/test_case_tooling.py can assert on
any box that the two describe the same case: the loss totals MATPOWER / PYPOWER print for `runpf(case14 / case30 / case118)`
(tests/physics_anchor.py) then are provably about the file the engine solves.

    python tools/make_m_fixtures.py [/root/reference/parameters]
"""
import json
import os
import re
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))


def parse_m(path):
    """bus / gen / branch matrices and baseMVA of a MATPOWER version-2 case file."""
    text = open(path).read()
    text = re.sub(r'%[^\n]*', '', text)          # comments
    out = {'baseMVA': float(re.search(r'mpc\.baseMVA\s*=\s*([0-9.eE+-]+)', text).group(1))}
    for name in ('bus', 'gen', 'branch'):
        m = re.search(r'mpc\.%s\s*=\s*\[(.*?)\]\s*;' % name, text, flags=re.S)
        rows = [r.strip() for r in m.group(1).replace('\n', ';').split(';')]
        out[name] = [[float(v) for v in r.split()] for r in rows if r]
        assert len(set(len(r) for r in out[name])) == 1, (path, name)
    return out


def main():
    params = sys.argv[1] if len(sys.argv) > 1 else '/root/reference/parameters'
    for env in sorted(os.listdir(os.path.join(ROOT, 'tests', 'golden', 'envs'))):
        src = os.path.join(params, env, 'level0', 'reference_grid.m')
        dst_dir = os.path.join(ROOT, 'tests', 'golden', 'envs', env, 'level0')
        if not os.path.exists(src) or not os.path.isdir(dst_dir):
            continue
        case = parse_m(src)
        with open(os.path.join(dst_dir, 'reference_grid_m.json'), 'w') as f:
            json.dump(case, f)
        print('%s: %d buses, %d productions, %d branches' % (env, len(case['bus']), len(case['gen']), len(case['branch'])))


if __name__ == '__main__':
    main()
