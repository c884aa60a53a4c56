"""Numeric anchors of the load-flow layer that do NOT route through oracle/ (missing #1).

PYPOWER is not under /root/reference, so bus voltages to 1e-6 on IEEE-30 / IEEE-118 used to rest on two restatements by one
author agreeing (oracle/pf_np.py == oracle/ppn_oracle.c == HIP).  Two anchors that need neither:

(a) the CASE FILES the reference ships (`parameters/<env>/level0/reference_grid.py`, re-emitted as
    tests/golden/envs/<env>/level0/reference_grid.json): solved through the `runpf` seam (ppn_runpf_arrays) exactly as they
    are -- their own Pd / Qd / Pg / Vg, every line in service, flat angles -- the stock IEEE cases must reproduce
    * the |V| column the IEEE-14 file keeps (the published solution to 3 decimals),
    * the totals MATPOWER / PYPOWER print for `runpf(case14 / case30 / case118)`: generation minus load = I^2 R losses
      of 13.393 / 2.444 / 132.863 MW (the 'Losses (I^2 * Z)' line of the printed summary; quoted from the published
      printouts, they are not in /root/reference);
(b) PHYSICS: from the voltages the engine returns, a dense pi-model written HERE from the MATPOWER branch equations
    (30 lines, no import from oracle/) recomputes S = V conj(Y V): the per-bus power balance against
    the returned Pg / Qg and the case's Pd / Qd must close to the solver's tolerance at every bus, the returned branch flows
    must be the pi-model's, P_f + P_t >= 0 on every branch with r >= 0; DC: P_f = -P_t, flows = (theta_f - theta_t) / x.

`engine_with_library` comes from tests/harness.py: lib_path None = the product library (GPU), else the emulation build."""
import json
import os

import numpy as np

from harness import engine_with_library
from helpers import ENVS, load_env

KNOWN_LOSSES_MW = {'default14': 13.393, 'default30': 2.444, 'default118': 132.863}


def dense_pi_model(base_mva, bus, branch):
    """Ybus, Yf, Yt (dense, external row order of `bus`) from MATPOWER-format arrays: branch k between rows f, t with
    series admittance ys = 1 / (r + jx), charging b, off-nominal tap t = tap e^{j shift}; out-of-service branches contribute
    nothing.  Bus shunts Gs + jBs in MW / MVAr at 1 p.u."""
    n = bus.shape[0]
    row_of = {int(b): i for i, b in enumerate(bus[:, 0])}
    Y = np.zeros((n, n), dtype=complex)
    Yf = np.zeros((branch.shape[0], n), dtype=complex)
    Yt = np.zeros((branch.shape[0], n), dtype=complex)
    for k, br in enumerate(branch):
        if br[10] == 0:
            continue
        f, t = row_of[int(br[0])], row_of[int(br[1])]
        ys = 1.0 / complex(br[2], br[3])
        tap = (br[8] if br[8] != 0 else 1.0) * np.exp(1j * np.deg2rad(br[9]))
        ytt = ys + 0.5j * br[4]
        yff = ytt / (tap * np.conj(tap))
        yft = -ys / np.conj(tap)
        ytf = -ys / tap
        Y[f, f] += yff; Y[f, t] += yft; Y[t, f] += ytf; Y[t, t] += ytt
        Yf[k, f] += yff; Yf[k, t] += yft; Yt[k, f] += ytf; Yt[k, t] += ytt
    Y[np.arange(n), np.arange(n)] += (bus[:, 4] + 1j * bus[:, 5]) / base_mva
    return Y, Yf, Yt, row_of


def ac_residuals(base_mva, bus_in, gen_in, bus_out, gen_out, branch_out):
    """(max per-bus |S mismatch| in p.u. over the buses in service, max |flow - pi-model| in MW, min P_f + P_t in MW)"""
    Y, Yf, Yt, row_of = dense_pi_model(base_mva, bus_out, branch_out)
    act = bus_out[:, 1] != 4
    V = np.where(act, bus_out[:, 7] * np.exp(1j * np.deg2rad(bus_out[:, 8])), 0.0)
    s_calc = V * np.conj(Y @ V)
    s_sched = -(bus_in[:, 2] + 1j * bus_in[:, 3]) / base_mva
    for g_in, g in zip(gen_in, gen_out):
        if g_in[7] > 0 and act[row_of[int(g[0])]]:
            s_sched[row_of[int(g[0])]] += (g[1] + 1j * g[2]) / base_mva
    mis = np.abs((s_calc - s_sched)[act]).max()
    on = branch_out[:, 10] != 0
    sf = (V[[row_of[int(b)] for b in branch_out[:, 0]]] * np.conj(Yf @ V)) * base_mva
    st = (V[[row_of[int(b)] for b in branch_out[:, 1]]] * np.conj(Yt @ V)) * base_mva
    dflow = max(np.abs(sf.real - branch_out[:, 13])[on].max(), np.abs(sf.imag - branch_out[:, 14])[on].max(),
                np.abs(st.real - branch_out[:, 15])[on].max(), np.abs(st.imag - branch_out[:, 16])[on].max())
    loss = (branch_out[:, 13] + branch_out[:, 15])[on & (branch_out[:, 2] >= 0)]
    assert np.all(branch_out[~on, 13:17] == 0)
    return mis, dflow, loss.min()


def raw_case(envname):
    with open(os.path.join(ENVS, envname, 'level0', 'reference_grid.json')) as f:
        d = json.load(f)
    return d['baseMVA'], np.array(d['bus'], dtype=float), np.array(d['gen'], dtype=float), np.array(d['branch'], dtype=float)


def check_case_file_anchor(lib_path, envname, solver):
    """(a): the shipped case file through ppn_runpf_arrays, as it is."""
    case, cfg, _ = load_env(envname, conf={'solver': solver})
    base, bus, gen, br = raw_case(envname)
    eng = engine_with_library(lib_path, case, cfg, 1)
    bo, go, ro, ok, outcome = eng.runpf_arrays(bus[None], gen[None], br[None])
    eng.close()
    assert ok[0] and outcome[0] == 0
    bo, go, ro = bo[0], go[0], ro[0]
    act = bus[:, 1] != 4
    losses = go[:, 1].sum() - bus[act, 2].sum()
    assert abs(losses - KNOWN_LOSSES_MW[envname]) < 2e-3, (envname, losses)                # the printed 'Losses (I^2 * Z)' total
    assert abs((ro[:, 13] + ro[:, 15]).sum() - losses) < 1e-3                               # ... is what the branches dissipate
    if envname == 'default14':     # the IEEE-14 file keeps the published solution (3 decimals); 30 is a flat start, 118 the CDF's
        pq = bus[:, 1] == 1
        assert np.abs(bo[pq, 7] - bus[pq, 7]).max() <= 1.5e-3
    held = (bus[:, 1] == 2) | (bus[:, 1] == 3)                                              # voltage-controlled buses sit at their Vg
    vg = {int(g[0]): g[5] for g in gen}
    assert all(abs(bo[i, 7] - vg[int(bus[i, 0])]) < 1e-12 for i in np.where(held)[0])
    mis, dflow, minloss = ac_residuals(base, bus, gen, bo, go, ro)
    assert mis < 2e-6 and dflow < 1e-6 and minloss > -1e-9, (mis, dflow, minloss)
    return losses, mis


_STATE_CACHE = {}


def _states(envname, n, seed, solver):
    """Random grid states are INPUTS: generated once per (case, n, seed) -- by numpy-oracle games driven with random actions --
    and shared by the solvers under test."""
    from engine_checks import random_grid_states
    key = (envname, n, seed)
    if key not in _STATE_CACHE:
        _STATE_CACHE[key] = random_grid_states(envname, n, seed, conf={'solver': 'newton'})
    case, cfg, states = _STATE_CACHE[key]
    cfg = dict(cfg)
    cfg['solver'] = solver
    return case, cfg, states


def check_physics_residuals(lib_path, envname, n, solver, seed=4242):
    """(b) on n random grid states (split nodes, lines out, productions off, warm starts): every solve the engine calls
    converged must close the power balance of the pi-model written in this file."""
    case, cfg, states = _states(envname, n, seed, solver)
    eng = engine_with_library(lib_path, case, cfg, min(n, 256))
    worst = [0.0, 0.0, 0.0]
    n_ok = n_split = n_out = 0
    for i0 in range(0, n, eng.batch):
        chunk = states[i0:i0 + eng.batch]
        bus = np.stack([s[0] for s in chunk]); gen = np.stack([s[1] for s in chunk]); br = np.stack([s[2] for s in chunk])
        bo, go, ro, ok, outcome = eng.runpf_arrays(bus, gen, br)
        for i in range(len(chunk)):
            if not ok[i]:
                continue
            mis, dflow, minloss = ac_residuals(case.baseMVA, bus[i], gen[i], bo[i], go[i], ro[i])
            # tolerance: the solver stops at |mismatch|_inf < 1e-6 p.u. (Newton: of S; fast-decoupled: of S / |V|, |V| <= ~1.1)
            assert mis < 2e-6, ('power balance of state %d does not close' % (i0 + i), mis)
            assert dflow < 1e-6 and minloss > -1e-9, (i0 + i, dflow, minloss)
            worst = [max(worst[0], mis), max(worst[1], dflow), min(worst[2], minloss)]
            n_ok += 1
            n_split += int((bus[i][case.nS:, 1] != 4).any())
            n_out += int((br[i][:, 10] == 0).any())
    eng.close()
    return dict(ok=n_ok, split=n_split, lines_out=n_out, worst_mismatch=worst[0], worst_flow=worst[1])


def check_dc_identities(lib_path, envname, n, seed=99):
    """rundcpf: lossless (P_t = -P_f, sum Pg = sum Pd + shunt G), flows = b (theta_f - theta_t), |V| = 1."""
    from engine_checks import random_grid_states
    conf = {'solver': 'fdxb', 'loadflow_mode': 'DC'}
    case, cfg, states = random_grid_states(envname, n, seed, conf=conf)
    eng = engine_with_library(lib_path, case, cfg, n)
    bus = np.stack([s[0] for s in states]); gen = np.stack([s[1] for s in states]); br = np.stack([s[2] for s in states])
    bo, go, ro, ok, outcome = eng.runpf_arrays(bus, gen, br)
    eng.close()
    n_ok = 0
    for i in range(n):
        if not ok[i]:
            continue
        n_ok += 1
        on = ro[i][:, 10] != 0
        act = bo[i][:, 1] != 4
        assert np.array_equal(ro[i][:, 15], -ro[i][:, 13]) and not ro[i][:, 14].any() and not ro[i][:, 16].any()
        assert np.all(bo[i][act, 7] == 1.0)
        row_of = {int(b): k for k, b in enumerate(bo[i][:, 0])}
        th = np.deg2rad(bo[i][:, 8])
        f = np.array([row_of[int(b)] for b in ro[i][:, 0]]); t = np.array([row_of[int(b)] for b in ro[i][:, 1]])
        tap = np.where(ro[i][:, 8] != 0, ro[i][:, 8], 1.0)
        pf = ((th[f] - th[t]) - np.deg2rad(ro[i][:, 9])) / ro[i][:, 3] / tap * case.baseMVA
        assert np.abs(pf - ro[i][:, 13])[on].max() < 1e-7
        gon = (gen[i][:, 7] > 0) & np.array([act[row_of[int(b)]] for b in gen[i][:, 0]])
        assert abs(go[i][gon, 1].sum() - bus[i][act, 2].sum() - bus[i][act, 4].sum()) < 1e-6
    return n_ok
