"""Case tooling: the twin-busbar generator reproduces the shipped reference grids from their plain
halves (what parameters/make_reference_grid.py does to a MATPOWER case)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'tools'))
from make_reference_grid import make_reference_grid  # noqa: E402
from pypownet_amd.case import Case, load_case_file, save_case_py  # noqa: E402

ENVS = os.path.join(os.path.dirname(__file__), 'golden', 'envs')


@pytest.mark.parametrize('env', ['default14', 'default30', 'default118'])
def test_twin_generator_reproduces_shipped_reference_grid(env):
    ref = load_case_file(os.path.join(ENVS, env, 'level0', 'reference_grid.json'))
    n = ref['bus'].shape[0] // 2
    plain = dict(ref, bus=ref['bus'][:n][::-1].copy())          # the plain case, rows shuffled
    plain['gen'] = ref['gen'][::-1].copy()
    plain['gen'][:, 7] = 0                                      # out of service in the source: forced on
    out = make_reference_grid(plain)
    assert np.array_equal(out['bus'][:, :8], ref['bus'][:, :8]) and np.array_equal(out['bus'][:, 9:], ref['bus'][:, 9:])
    assert np.all(out['bus'][:, 8] == 0)
    assert np.array_equal(out['gen'][:, :7], ref['gen'][:, :7]) and np.all(out['gen'][:, 7] == 1)
    assert np.array_equal(np.sort(out['branch'][:, :2], axis=0), np.sort(ref['branch'][:, :2], axis=0))
    Case(out)      # accepted by the engine's case model (sorted ids, twins, one production / load per substation)


def test_reference_grid_py_round_trip(tmp_path):
    """The tool writes ``reference_grid.py`` (the format parameters/make_reference_grid.py:63 saves and grid.py:65 loads): read
    back through the ``loadcase`` contract it is the same case, bit for bit, and the engine's case model takes it."""
    import subprocess
    ref = load_case_file(os.path.join(ENVS, 'default30', 'level0', 'reference_grid.json'))
    n = ref['bus'].shape[0] // 2
    src = str(tmp_path / 'case30.py')
    save_case_py(dict(ref, bus=ref['bus'][:n]), src)
    tool = os.path.join(os.path.dirname(__file__), '..', 'tools', 'make_reference_grid.py')
    out = subprocess.check_output([sys.executable, tool, src]).decode().strip()
    assert out.endswith('reference_grid.py') and os.path.dirname(out) == str(tmp_path)
    back = load_case_file(out)
    for k in ('bus', 'gen'):
        assert np.array_equal(back[k], ref[k]), k
    rows = lambda a: a[np.lexsort(a.T[::-1])]      # (the shipped file keeps its own order among branches of equal origin)
    assert np.array_equal(rows(back['branch']), rows(ref['branch']))
    assert back['baseMVA'] == ref['baseMVA']
    Case(back)


@pytest.mark.parametrize('env', ['default14', 'default30', 'default118'])
def test_matpower_m_file_and_reference_grid_py_are_the_same_case(env):
    """The reference ships `reference_grid.m` (MATPOWER, what the Octave backend and a MATPOWER user would load) next to the
    `reference_grid.py` that `Grid` loads (pypownet/grid.py:65): bus and gen arrays bit for bit identical, branch arrays identical
    except RATE_A (column 5: 9900 in the .py, 0 = unlimited in the .m -- no part of a power flow).  The numeric anchors quoted
    from MATPOWER / PYPOWER printouts (tests/physics_anchor.py: losses of runpf(case14 / 30 / 118)) therefore are about the very
    arrays the engine solves.  Fixture: tools/make_m_fixtures.py (numbers parsed out of the .m files, no source text)."""
    import json
    d = os.path.join(ENVS, env, 'level0')
    py = load_case_file(os.path.join(d, 'reference_grid.json'))
    with open(os.path.join(d, 'reference_grid_m.json')) as f:
        m = json.load(f)
    assert float(m['baseMVA']) == float(py['baseMVA'])
    assert np.array_equal(np.asarray(m['bus']), py['bus'])
    assert np.array_equal(np.asarray(m['gen']), py['gen'])
    mb, pb = np.asarray(m['branch']), py['branch']
    assert mb.shape == pb.shape
    cols = [c for c in range(mb.shape[1]) if c != 5]
    assert np.array_equal(mb[:, cols], pb[:, cols])
    assert np.array_equal(mb[:, 5], pb[:, 5]) or (np.all(mb[:, 5] == 0) and np.all(pb[:, 5] == 9900.0))      # (default30: equal)
