#!/usr/bin/env python
"""Turns a plain MATPOWER/PYPOWER case (.py or .json) into a pypownet reference grid: what the reference's tooling does
(parameters/make_reference_grid.py:9-64) -- rows sorted, bus ids renumbered 1..n, one artificial twin busbar "666<id>"
per substation (type 4, no load), every production and line in service, angles zeroed, baseKV defaulted to 100 when the
case has none -- written in the source's own format like the reference's savecase (a PYPOWER ``reference_grid.py`` for a
``.py`` case: what ``parameters/<env>/<level>/`` ships and ``loadcase`` reads, grid.py:65) or, for a ``.json`` source or an
explicit ``out.json``, as the JSON case format of this repository's fixtures.  (MATPOWER ``.m`` is the Octave backend's format: out of
scope with that backend.)  SURVEY.md 8f rank 4 (case tooling; not on the hot path).
Usage: python tools/make_reference_grid.py case.py|case.json [out.py|out.json]"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT)
from pypownet_amd.case import load_case_file, save_case_json, save_case_py  # noqa: E402

ARTIFICIAL_NODE_STARTING_STRING = '666'      # pypownet/__init__.py


def make_reference_grid(ppc):
    bus, gen, br = (np.array(ppc[k], dtype=np.float64, copy=True) for k in ('bus', 'gen', 'branch'))
    gen = gen[np.argsort(gen[:, 0], kind='stable')]
    bus = bus[np.argsort(bus[:, 0], kind='stable')]
    br = br[np.argsort(br[:, 1], kind='stable')]
    br = br[np.argsort(br[:, 0], kind='stable')]
    ids = bus[:, 0].copy()
    new_id = {float(v): float(i + 1) for i, v in enumerate(ids)}
    bus[:, 0] = [new_id[float(v)] for v in ids]
    gen[:, 0] = [new_id[float(v)] for v in gen[:, 0]]
    br[:, 0] = [new_id[float(v)] for v in br[:, 0]]
    br[:, 1] = [new_id[float(v)] for v in br[:, 1]]
    twins = bus.copy()
    twins[:, 0] = [float(ARTIFICIAL_NODE_STARTING_STRING + str(int(v))) for v in twins[:, 0]]
    twins[:, 1] = 4
    twins[:, 2] = 0.0
    twins[:, 3] = 0.0
    bus = np.concatenate((bus, twins), axis=0)
    gen[:, 7] = 1
    br[:, 10] = 1
    bus[:, 8] = 0
    if np.all(bus[:, 9] == 0):
        bus[:, 9] = 100
    return {'version': str(ppc.get('version', '2')), 'baseMVA': float(ppc['baseMVA']), 'bus': bus, 'gen': gen, 'branch': br}


if __name__ == '__main__':
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    src = sys.argv[1]
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(src), 'reference_grid.' + src.rsplit('.', 1)[-1])
    (save_case_py if out.endswith('.py') else save_case_json)(make_reference_grid(load_case_file(src)), out)
    print(out)
