"""ORACLE (test infrastructure, not product code): numpy/scipy restatement of the power-flow solve that
pypownet delegates to PYPOWER 5.1.4 (``pypower.api.runpf`` / ``rundcpf``).

PYPOWER is a third-party dependency of the reference (requirements.txt:9, ``PYPOWER==5.1.4``) and is NOT
vendored under /root/reference, nor installed here.  This file restates its published algorithm
(MATPOWER/PYPOWER ``runpf, ext2int, bustypes, makeYbus, makeSbus, makeB, fdpf, newtonpf, dSbus_dV, pfsoln,
makeBdc, dcpf, int2ext``) with the same library class (scipy.sparse + SuperLU) and is
anchored on the reference's own call sites:
    pypownet/grid.py:63-64   ppoption(PF_ALG=2, PF_MAX_IT_FD=25, PF_TOL=1e-6, VERBOSE=0, OUT_ALL=0)
    pypownet/grid.py:227-229 runpf(mpc, opts, '', '') / rundcpf(...)
    pypownet/grid.py:228-231 RuntimeError/RuntimeWarning/IndexError/ValueError => "grid is not connexe"
Pinned by the reference's known-answer tests K1-K4 (tests/test_oracle_known_answers.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import warnings

import numpy as np
from scipy.sparse import csr_matrix, csc_matrix, hstack, vstack
from scipy.sparse.linalg import splu, spsolve, MatrixRankWarning

# MATPOWER v2 columns
BUS_I, BUS_TYPE, PD, QD, GS, BS, BUS_AREA, VM, VA, BASE_KV = range(10)
GEN_BUS, PG, QG, QMAX, QMIN, VG, MBASE, GEN_STATUS = range(8)
F_BUS, T_BUS, BR_R, BR_X, BR_B, RATE_A, RATE_B, RATE_C, TAP, SHIFT, BR_STATUS = range(11)
PF, QF, PT, QT = 13, 14, 15, 16
PQ, PV, REF, NONE = 1, 2, 3, 4

ALG_NEWTON, ALG_FDXB, ALG_FDBX = 1, 2, 3


class SolveInfo(object):
    def __init__(self):
        self.iterations = 0
        self.success = False


def _make_ybus(baseMVA, bus, branch):
    """A.2 makeYbus on internally numbered data."""
    nb, nl = bus.shape[0], branch.shape[0]
    stat = branch[:, BR_STATUS]
    Ys = stat / (branch[:, BR_R] + 1j * branch[:, BR_X])
    Bc = stat * branch[:, BR_B]
    tap = np.ones(nl)
    i = np.nonzero(branch[:, TAP])[0]
    tap[i] = branch[i, TAP]
    tap = tap * np.exp(1j * np.pi / 180 * branch[:, SHIFT])
    Ytt = Ys + 1j * Bc / 2
    Yff = Ytt / (tap * np.conj(tap))
    Yft = -Ys / np.conj(tap)
    Ytf = -Ys / tap
    Ysh = (bus[:, GS] + 1j * bus[:, BS]) / baseMVA
    f = branch[:, F_BUS].astype(int)
    t = branch[:, T_BUS].astype(int)
    r = np.arange(nl)
    Yf = csr_matrix((np.r_[Yff, Yft], (np.r_[r, r], np.r_[f, t])), (nl, nb))
    Yt = csr_matrix((np.r_[Ytf, Ytt], (np.r_[r, r], np.r_[f, t])), (nl, nb))
    Cf = csr_matrix((np.ones(nl), (r, f)), (nl, nb))
    Ct = csr_matrix((np.ones(nl), (r, t)), (nl, nb))
    Ybus = Cf.T * Yf + Ct.T * Yt + csr_matrix((Ysh, (np.arange(nb), np.arange(nb))), (nb, nb))
    return Ybus.tocsr(), Yf, Yt


def _make_sbus(baseMVA, bus, gen):
    """A.3 makeSbus."""
    nb = bus.shape[0]
    on = np.where(gen[:, GEN_STATUS] > 0)[0]
    gbus = gen[on, GEN_BUS].astype(int)
    Sg = np.zeros(nb, dtype=complex)
    np.add.at(Sg, gbus, gen[on, PG] + 1j * gen[on, QG])
    return (Sg - (bus[:, PD] + 1j * bus[:, QD])) / baseMVA


def _make_b(baseMVA, bus, branch, alg):
    """A.4 makeB: B' and B'' of the fast-decoupled method."""
    tb, tbr = bus.copy(), branch.copy()
    tb[:, BS] = 0
    tbr[:, BR_B] = 0
    tbr[:, TAP] = 1
    if alg == ALG_FDXB:
        tbr[:, BR_R] = 0
    Bp = -1 * _make_ybus(baseMVA, tb, tbr)[0].imag
    tbr = branch.copy()
    tbr[:, SHIFT] = 0
    if alg == ALG_FDBX:
        tbr[:, BR_R] = 0
    Bpp = -1 * _make_ybus(baseMVA, bus, tbr)[0].imag
    return Bp, Bpp


def _bustypes(bus, gen):
    nb, ng = bus.shape[0], gen.shape[0]
    has_gen = np.zeros(nb, dtype=bool)
    on = gen[:, GEN_STATUS] > 0
    has_gen[gen[on, GEN_BUS].astype(int)] = True
    ref = np.where((bus[:, BUS_TYPE] == REF) & has_gen)[0]
    pv = np.where((bus[:, BUS_TYPE] == PV) & has_gen)[0]
    pq = np.where((bus[:, BUS_TYPE] == PQ) | ~has_gen)[0]
    if len(ref) == 0:
        ref = np.zeros(1, dtype=int)
        ref[0] = pv[0]  # IndexError when there is no PV bus either -> "not connexe" in pypownet
        pv = pv[1:]
    return ref, pv, pq


def _fdpf(Ybus, Sbus, V0, Bp, Bpp, ref, pv, pq, tol, max_it, info):
    converged = False
    i = 0
    V = V0.copy()
    Va = np.angle(V)
    Vm = np.abs(V)
    pvpq = np.r_[pv, pq]
    mis = (V * np.conj(Ybus * V) - Sbus) / Vm
    P = mis[pvpq].real
    Q = mis[pq].imag
    normP = np.linalg.norm(P, np.inf)
    normQ = np.linalg.norm(Q, np.inf)  # ValueError on empty pq, as in PYPOWER
    if normP < tol and normQ < tol:
        converged = True
    Bp = csc_matrix(Bp[pvpq][:, pvpq])
    Bpp = csc_matrix(Bpp[pq][:, pq])
    Bp_solver = splu(Bp)      # RuntimeError("Factor is exactly singular") on islanded grids
    Bpp_solver = splu(Bpp)
    while not converged and i < max_it:
        i += 1
        dVa = -Bp_solver.solve(P)
        Va[pvpq] = Va[pvpq] + dVa
        V = Vm * np.exp(1j * Va)
        mis = (V * np.conj(Ybus * V) - Sbus) / Vm
        P = mis[pvpq].real
        Q = mis[pq].imag
        normP = np.linalg.norm(P, np.inf)
        normQ = np.linalg.norm(Q, np.inf)
        info.half_iterations += 1
        if normP < tol and normQ < tol:
            converged = True
            break
        dVm = -Bpp_solver.solve(Q)
        Vm[pq] = Vm[pq] + dVm
        V = Vm * np.exp(1j * Va)
        mis = (V * np.conj(Ybus * V) - Sbus) / Vm
        P = mis[pvpq].real
        Q = mis[pq].imag
        normP = np.linalg.norm(P, np.inf)
        normQ = np.linalg.norm(Q, np.inf)
        info.half_iterations += 1
        if normP < tol and normQ < tol:
            converged = True
            break
    info.iterations = i
    return V, converged


def _dSbus_dV(Ybus, V):
    ib = np.arange(len(V))
    Ibus = Ybus * V
    diagV = csr_matrix((V, (ib, ib)))
    diagIbus = csr_matrix((Ibus, (ib, ib)))
    diagVnorm = csr_matrix((V / np.abs(V), (ib, ib)))
    dS_dVm = diagV * np.conj(Ybus * diagVnorm) + np.conj(diagIbus) * diagVnorm
    dS_dVa = 1j * diagV * np.conj(diagIbus - Ybus * diagV)
    return dS_dVm, dS_dVa


def _newtonpf(Ybus, Sbus, V0, ref, pv, pq, tol, max_it, info):
    converged = False
    i = 0
    V = V0.copy()
    Va = np.angle(V)
    Vm = np.abs(V)
    pvpq = np.r_[pv, pq]
    npv, npq = len(pv), len(pq)
    j1, j2 = 0, npv
    j3, j4 = j2, j2 + npq
    j5, j6 = j4, j4 + npq
    mis = V * np.conj(Ybus * V) - Sbus
    F = np.r_[mis[pv].real, mis[pq].real, mis[pq].imag]
    normF = np.linalg.norm(F, np.inf)
    if normF < tol:
        converged = True
    while not converged and i < max_it:
        i += 1
        dS_dVm, dS_dVa = _dSbus_dV(Ybus, V)
        J11 = dS_dVa[pvpq][:, pvpq].real
        J12 = dS_dVm[pvpq][:, pq].real
        J21 = dS_dVa[pq][:, pvpq].imag
        J22 = dS_dVm[pq][:, pq].imag
        J = vstack([hstack([J11, J12]), hstack([J21, J22])], format='csc')
        with warnings.catch_warnings():
            # NR is not the algorithm the reference configures (grid.py:63 uses PF_ALG=2); a singular Jacobian
            # (islanded grid) is mapped onto the same outcome as SuperLU's "Factor is exactly singular"
            # RuntimeError of the fast-decoupled path, i.e. pypownet's "grid is not connexe" (grid.py:230).
            warnings.simplefilter('error', MatrixRankWarning)
            try:
                dx = -1 * spsolve(J, F)
            except MatrixRankWarning:
                raise RuntimeError('Jacobian is exactly singular')
        if npv:
            Va[pv] = Va[pv] + dx[j1:j2]
        if npq:
            Va[pq] = Va[pq] + dx[j3:j4]
            Vm[pq] = Vm[pq] + dx[j5:j6]
        V = Vm * np.exp(1j * Va)
        Vm = np.abs(V)
        Va = np.angle(V)
        mis = V * np.conj(Ybus * V) - Sbus
        F = np.r_[mis[pv].real, mis[pq].real, mis[pq].imag]
        normF = np.linalg.norm(F, np.inf)
        if normF < tol:
            converged = True
    info.iterations = i
    return V, converged


def _pfsoln(baseMVA, bus, gen, branch, Ybus, Yf, Yt, V, ref, pv, pq):
    """A.6 pfsoln (internal numbering; gens already restricted to on-line ones, sorted by bus)."""
    bus[:, VM] = np.abs(V)
    bus[:, VA] = np.angle(V) * 180 / np.pi
    on = np.where(gen[:, GEN_STATUS] > 0)[0]
    gbus = gen[on, GEN_BUS].astype(int)
    Sg = V[gbus] * np.conj(Ybus[gbus, :] * V)
    gen[:, QG] = 0
    gen[on, QG] = Sg.imag * baseMVA + bus[gbus, QD]
    if len(on) > 1:
        nb = bus.shape[0]
        ngon = len(on)
        Cg = csr_matrix((np.ones(ngon), (gbus, np.arange(ngon))), (nb, ngon))
        ngg = np.asarray(Cg.sum(1)).flatten()[gbus]  # number of on-line gens at each gen's bus
        gen[on, QG] = gen[on, QG] / ngg
        Cmin = csr_matrix((gen[on, QMIN], (np.arange(ngon), gbus)), (ngon, nb))
        Cmax = csr_matrix((gen[on, QMAX], (np.arange(ngon), gbus)), (ngon, nb))
        Qg_tot = np.asarray(Cg * gen[on, QG]).flatten()
        Qg_min = np.asarray(Cmin.sum(0)).flatten()
        Qg_max = np.asarray(Cmax.sum(0)).flatten()
        ig = np.where(Qg_min[gbus] == Qg_max[gbus])[0]
        Qg_save = gen[on[ig], QG].copy()
        eps = np.finfo(float).eps
        gen[on, QG] = gen[on, QMIN] + ((Qg_tot - Qg_min) / (Qg_max - Qg_min + eps))[gbus] * \
            (gen[on, QMAX] - gen[on, QMIN])
        gen[on[ig], QG] = Qg_save
    for k in range(len(ref)):
        refgen = np.where(gbus == ref[k])[0]
        gen[on[refgen[0]], PG] = Sg[refgen[0]].real * baseMVA + bus[ref[k], PD]
        if len(refgen) > 1:
            gen[on[refgen[0]], PG] -= np.sum(gen[on[refgen[1:]], PG])
    br = np.where(branch[:, BR_STATUS] != 0)[0]
    out = np.where(branch[:, BR_STATUS] == 0)[0]
    f = branch[br, F_BUS].astype(int)
    t = branch[br, T_BUS].astype(int)
    Sf = V[f] * np.conj(Yf[br, :] * V) * baseMVA
    St = V[t] * np.conj(Yt[br, :] * V) * baseMVA
    branch[br, PF], branch[br, QF], branch[br, PT], branch[br, QT] = Sf.real, Sf.imag, St.real, St.imag
    branch[out, PF:QT + 1] = 0
    return bus, gen, branch


def _make_bdc(baseMVA, bus, branch):
    nb, nl = bus.shape[0], branch.shape[0]
    stat = branch[:, BR_STATUS]
    b = stat / branch[:, BR_X]
    tap = np.ones(nl)
    i = np.nonzero(branch[:, TAP])[0]
    tap[i] = branch[i, TAP]
    b = b / tap
    f = branch[:, F_BUS].astype(int)
    t = branch[:, T_BUS].astype(int)
    r = np.arange(nl)
    Cft = csr_matrix((np.r_[np.ones(nl), -np.ones(nl)], (np.r_[r, r], np.r_[f, t])), (nl, nb))
    Bf = csr_matrix((np.r_[b, -b], (np.r_[r, r], np.r_[f, t])), (nl, nb))
    Bbus = Cft.T * Bf
    Pfinj = b * (-branch[:, SHIFT] * np.pi / 180)
    Pbusinj = Cft.T * Pfinj
    return Bbus.tocsr(), Bf, Pbusinj, Pfinj


def _connected(n, f, t, start):
    """Every kept bus reachable from `start` over the in-service branches."""
    adj = [[] for _ in range(n)]
    for a, b in zip(f, t):
        adj[a].append(b)
        adj[b].append(a)
    seen = np.zeros(n, dtype=bool)
    seen[start] = True
    stack = [start]
    while stack:
        u = stack.pop()
        for v in adj[u]:
            if not seen[v]:
                seen[v] = True
                stack.append(v)
    return bool(seen.all())


def runpf(baseMVA, bus, gen, branch, dc=False, alg=ALG_FDXB, tol=1e-6, max_it=None, info=None):
    """``(bus, gen, branch), success = runpf(...)`` on EXTERNAL MATPOWER-format arrays (copies are returned).

    Exceptions escape exactly where PYPOWER's would (singular factor -> RuntimeError, no PV/ref -> IndexError,
    empty PQ set -> ValueError); the caller maps them like pypownet/grid.py:228-231 does.
    """
    if info is None:
        info = SolveInfo()
    info.half_iterations = 0
    if max_it is None:
        max_it = 10 if alg == ALG_NEWTON else 25
    bus = np.array(bus, dtype=float, copy=True)
    gen = np.array(gen, dtype=float, copy=True)
    branch = np.array(branch, dtype=float, copy=True)
    if branch.shape[1] < QT + 1:
        branch = np.hstack([branch, np.zeros((branch.shape[0], QT + 1 - branch.shape[1]))])

    # ---- A.1 ext2int ----------------------------------------------------------------------------
    nb = bus.shape[0]
    ids = bus[:, BUS_I].astype(np.int64)
    bs = bus[:, BUS_TYPE] != NONE
    row_of = {int(v): k for k, v in enumerate(ids)}
    grow = np.array([row_of[int(v)] for v in gen[:, GEN_BUS]], dtype=int)
    frow = np.array([row_of[int(v)] for v in branch[:, F_BUS]], dtype=int)
    trow = np.array([row_of[int(v)] for v in branch[:, T_BUS]], dtype=int)
    gs = (gen[:, GEN_STATUS] > 0) & bs[grow]
    brs = (branch[:, BR_STATUS] != 0) & bs[frow] & bs[trow]
    bus_on = np.where(bs)[0]
    e2i = -np.ones(nb, dtype=int)
    e2i[bus_on] = np.arange(len(bus_on))
    gen_on = np.where(gs)[0]
    br_on = np.where(brs)[0]
    ibus = bus[bus_on].copy()
    ibus[:, BUS_I] = np.arange(len(bus_on))
    igen = gen[gen_on].copy()
    igen[:, GEN_BUS] = e2i[grow[gen_on]]
    gorder = np.argsort(igen[:, GEN_BUS], kind='stable')
    igen = igen[gorder]
    ibr = branch[br_on].copy()
    ibr[:, F_BUS] = e2i[frow[br_on]]
    ibr[:, T_BUS] = e2i[trow[br_on]]

    ref, pv, pq = _bustypes(ibus, igen)
    on = np.where(igen[:, GEN_STATUS] > 0)[0]
    gbus = igen[on, GEN_BUS].astype(int)

    if dc:
        # Islanded grid: B[pvpq, pvpq] is singular and what PYPOWER's spsolve returns then is SuperLU's rounding luck --
        # NaN (the reference's _contains_nan turns it into an outage, grid.py:103-110, 263-264), a RuntimeError on some
        # paths ("not connexe", grid.py:230), or finite garbage with which the game would go on.  The build defines the
        # outcome (SURVEY.md finding 5: exact connectivity test): an island without the reference bus is "not connexe".
        if not _connected(len(ibus), ibr[:, F_BUS].astype(int), ibr[:, T_BUS].astype(int), int(ref[0])):
            raise RuntimeError('grid is not connected')
        Va0 = ibus[:, VA] * (np.pi / 180)
        B, Bf, Pbusinj, Pfinj = _make_bdc(baseMVA, ibus, ibr)
        Pbus = _make_sbus(baseMVA, ibus, igen).real - Pbusinj - ibus[:, GS] / baseMVA
        pvpq = np.r_[pv, pq]
        Va = Va0.copy()
        with warnings.catch_warnings():
            warnings.simplefilter('ignore', MatrixRankWarning)
            Va[pvpq] = spsolve(csc_matrix(B[pvpq][:, pvpq]), Pbus[pvpq] - B[pvpq][:, ref] * Va0[ref])
        ibr[:, [QF, QT]] = 0
        ibr[:, PF] = (Bf * Va + Pfinj) * baseMVA
        ibr[:, PT] = -ibr[:, PF]
        ibus[:, VM] = 1
        ibus[:, VA] = Va * (180 / np.pi)
        for k in range(len(ref)):
            temp = np.where(gbus == ref[k])[0]
            rg = on[temp[0]]
            igen[rg, PG] = igen[rg, PG] + (B[ref[k], :] * Va - Pbus[ref[k]])[0] * baseMVA
        success = True
        info.iterations = 1
    else:
        V0 = ibus[:, VM] * np.exp(1j * np.pi / 180 * ibus[:, VA])
        vcb = np.ones(len(V0))
        vcb[pq] = 0
        k = np.where(vcb[gbus] != 0)[0]
        V0[gbus[k]] = igen[on[k], VG] / np.abs(V0[gbus[k]]) * V0[gbus[k]]
        Ybus, Yf, Yt = _make_ybus(baseMVA, ibus, ibr)
        Sbus = _make_sbus(baseMVA, ibus, igen)
        if alg == ALG_NEWTON:
            V, success = _newtonpf(Ybus, Sbus, V0, ref, pv, pq, tol, max_it, info)
        else:
            Bp, Bpp = _make_b(baseMVA, ibus, ibr, alg)
            V, success = _fdpf(Ybus, Sbus, V0, Bp, Bpp, ref, pv, pq, tol, max_it, info)
        ibus, igen, ibr = _pfsoln(baseMVA, ibus, igen, ibr, Ybus, Yf, Yt, V, ref, pv, pq)

    # ---- int2ext --------------------------------------------------------------------------------
    ibus[:, BUS_I] = ids[bus_on]
    bus[bus_on] = ibus
    unsort = np.empty_like(gorder)
    unsort[gorder] = np.arange(len(gorder))
    igen = igen[unsort]
    igen[:, GEN_BUS] = gen[gen_on, GEN_BUS]
    gen[gen_on] = igen
    ibr[:, F_BUS] = branch[br_on, F_BUS]
    ibr[:, T_BUS] = branch[br_on, T_BUS]
    branch[br_on] = ibr
    # zero result fields of out-of-service gens and branches (runpf tail)
    gen[~gs, PG] = 0
    gen[~gs, QG] = 0
    branch[~brs, PF:QT + 1] = 0
    info.success = bool(success)
    return (bus, gen, branch), bool(success)
