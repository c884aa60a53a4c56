"""Anchors of the NUMERIC layer that do not route through the restatements under oracle/ (tests/physics_anchor.py): the shipped
case files solved as they are (published |V| of IEEE-14, the loss totals MATPOWER prints for case14 / case30 / case118) and a
pi-model written in the test itself that closes the power balance on the voltages the engine returns.

CPU (-m "not gpu"): the emulation build of the kernel sources, and the numpy oracle itself (so that the checker is pinned by the
same physics).  GPU (-m gpu): libppn.so through the C ABI, 1 000 random IEEE-118 states per AC solver."""
import os
import subprocess

import numpy as np
import pytest

import physics_anchor as pa

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
SRC = os.path.join(ROOT, 'pypownet_amd', 'csrc')
EMU = os.path.join(ROOT, 'build', 'libppn_emu.so')
ENVS3 = ['default14', 'default30', 'default118']


@pytest.fixture(scope='module')
def emu_lib():
    srcs = [os.path.join(SRC, f) for f in os.listdir(SRC)] + [os.path.join(ROOT, 'include', 'ppn.h')]
    if not os.path.exists(EMU) or any(os.path.getmtime(s) > os.path.getmtime(EMU) for s in srcs):
        os.makedirs(os.path.dirname(EMU), exist_ok=True)
        subprocess.check_call(['g++', '-O2', '-std=c++17', '-DPPN_EMU', '-fPIC', '-shared', '-x', 'c++',
                               os.path.join(SRC, 'ppn_engine.hip'), '-o', EMU])
    return EMU


@pytest.mark.parametrize('solver', ['newton', 'fdxb'])
@pytest.mark.parametrize('env', ENVS3)
def test_oracle_case_file_anchor(env, solver):
    """The numpy oracle on the raw case files: published loss totals, IEEE-14's published |V|, power balance of the
    test's own pi-model."""
    from oracle import pf_np
    base, bus, gen, br = pa.raw_case(env)
    alg = pf_np.ALG_NEWTON if solver == 'newton' else pf_np.ALG_FDXB
    (bo, go, ro), ok = pf_np.runpf(base, bus.copy(), gen.copy(), br.copy(), dc=False, alg=alg, tol=1e-6)
    assert ok
    act = bus[:, 1] != 4
    assert abs(go[:, 1].sum() - bus[act, 2].sum() - pa.KNOWN_LOSSES_MW[env]) < 2e-3
    if env == 'default14':
        pq = bus[:, 1] == 1
        assert np.abs(bo[pq, 7] - bus[pq, 7]).max() <= 1.5e-3
    mis, dflow, minloss = pa.ac_residuals(base, bus, gen, bo, go, ro)
    assert mis < 2e-6 and dflow < 1e-6 and minloss > -1e-9


@pytest.mark.parametrize('solver', ['newton', 'fdxb'])
@pytest.mark.parametrize('env', ENVS3)
def test_emu_case_file_anchor(emu_lib, env, solver):
    pa.check_case_file_anchor(emu_lib, env, solver)


@pytest.mark.parametrize('env,solver,n', [('default118', 'newton', 24), ('default30', 'fdxb', 24)])
def test_emu_physics_residuals(emu_lib, env, solver, n):
    seen = pa.check_physics_residuals(emu_lib, env, n, solver)
    assert seen['ok'] >= n // 2 and seen['split'] and seen['lines_out'], seen


def test_emu_dc_identities(emu_lib):
    assert pa.check_dc_identities(emu_lib, 'default118', 16) >= 8


# ---------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize('solver', ['newton', 'fdxb'])
@pytest.mark.parametrize('env', ENVS3)
def test_gpu_case_file_anchor(env, solver):
    """the raw default14 / 30 / 118 case files through ppn_runpf_arrays on the GPU."""
    pa.check_case_file_anchor(None, env, solver)


@pytest.mark.gpu
@pytest.mark.parametrize('solver', ['newton', 'fdxb'])
def test_gpu_physics_residuals_1000_states_default118(solver):
    """>= 1 000 random default118 states (split nodes, lines out, productions off); per-bus mismatch of
    an independent dense pi-model < 2e-6 p.u. on every state the GPU calls converged, returned flows = the pi-model's, P_f + P_t >= 0."""
    seen = pa.check_physics_residuals(None, 'default118', 1000, solver)
    assert seen['ok'] >= 600 and seen['split'] >= 100 and seen['lines_out'] >= 600, seen


@pytest.mark.gpu
def test_gpu_physics_residuals_small_cases():
    for env in ('default14', 'default30'):
        seen = pa.check_physics_residuals(None, env, 200, 'newton', seed=7)
        assert seen['ok'] >= 100, seen


@pytest.mark.gpu
def test_gpu_dc_identities():
    assert pa.check_dc_identities(None, 'default118', 256) >= 128
    assert pa.check_dc_identities(None, 'default14', 64) >= 32
