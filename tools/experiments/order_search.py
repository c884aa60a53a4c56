#!/usr/bin/env python
"""Prototype of the elimination-order search: randomised multiple-minimum-degree orders of the substation graph, scored with the
measured cycle model of the one-phase levels (per sparse level: 340 bookkeeping + 1215 per round of 64 triples + 840 per round
of 64 pairs in the backward pass; the last levels with <= 6 pivots in total go to the dense tail)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import load_env  # noqa: E402

TAIL = 6


def mmd_order(nS, f, t, rng, slack=1, p_skip=0.0):
    adj = np.zeros((nS, nS), dtype=bool)
    adj[f, t] = True
    adj[t, f] = True
    np.fill_diagonal(adj, False)
    gone = np.zeros(nS, dtype=bool)
    order = []
    while len(order) < nS:
        alive = ~gone
        deg = (adj & alive[None, :]).sum(axis=1)
        deg[gone] = 1 << 30
        dmin = deg.min()
        cand = [i for i in np.argsort(deg + (rng.random(nS) if rng is not None else 0) * 0.5, kind='stable') if deg[i] <= dmin + slack]
        blocked = np.zeros(nS, dtype=bool)
        picked = []
        for k in cand:
            if blocked[k]:
                continue
            if rng is not None and picked and rng.random() < p_skip:
                continue
            picked.append(k)
            blocked[k] = True
            blocked |= adj[k]
        for k in picked:
            gone[k] = True
            order.append(k)
            nb = np.where(adj[k] & ~gone)[0]
            adj[np.ix_(nb, nb)] = True
            np.fill_diagonal(adj, False)
    return order


def analyse(nS, f, t, order):
    pos = np.empty(nS, dtype=int)
    pos[order] = np.arange(nS)
    a = np.eye(nS, dtype=bool)
    a[pos[f], pos[t]] = True
    a[pos[t], pos[f]] = True
    level = np.zeros(nS, dtype=int)
    pairs, tris = np.zeros(nS, dtype=int), np.zeros(nS, dtype=int)
    for k in range(nS):
        nb = np.where(a[k, k + 1:])[0] + k + 1
        a[np.ix_(nb, nb)] = True
        level[nb] = np.maximum(level[nb], level[k] + 1)
        pairs[k], tris[k] = len(nb), len(nb) ** 2
    fill = int(a.sum())
    nlev = level.max() + 1
    per = [(int((level == lv).sum()), int(pairs[level == lv].sum()), int(tris[level == lv].sum())) for lv in range(nlev)]
    # dense tail: longest run of final levels with <= TAIL pivots in total
    tail_from = nlev
    tot = 0
    for lv in range(nlev - 1, -1, -1):
        if tot + per[lv][0] > TAIL:
            break
        tot += per[lv][0]
        tail_from = lv
    rounds = lambda c: (c + 63) // 64
    cost = sum(340 + 1215 * rounds(tr) + 840 * rounds(pr) for (_, pr, tr) in per[:tail_from])
    return dict(fill=fill, levels=nlev, sparse_levels=tail_from, cost=cost, per=per)


if __name__ == '__main__':
    env = sys.argv[1] if len(sys.argv) > 1 else 'default118'
    n_try = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    case, cfg, chronics = load_env(env)
    f, t = np.asarray(case.or_sub), np.asarray(case.ex_sub)
    base = analyse(case.nS, f, t, mmd_order(case.nS, f, t, None))
    print('deterministic MMD:', {k: base[k] for k in ('fill', 'levels', 'sparse_levels', 'cost')}, base['per'])
    rng = np.random.default_rng(0)
    best = {}
    for it in range(n_try):
        slack = int(rng.integers(0, 3))
        o = mmd_order(case.nS, f, t, rng, slack=slack, p_skip=float(rng.choice([0.0, 0.05, 0.15])))
        r = analyse(case.nS, f, t, o)
        key = r['fill']
        if key not in best or r['cost'] < best[key]['cost']:
            best[key] = r
    for key in sorted(best)[:25]:
        r = best[key]
        print('fill %4d  levels %2d  sparse %2d  cost %6d  %s' % (r['fill'], r['levels'], r['sparse_levels'], r['cost'], r['per']))
