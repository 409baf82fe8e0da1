#!/usr/bin/env python3
"""dev prototype: the two persistence sweeps of compute_dmt_graph (dmtgraph.py:277-314) as LEVELS of data-parallel steps.

Per level every super-node takes its first alive edge (smallest rank); if the node across is older, the younger side's root dies on
that edge (exact: the node is a closed Kruskal component at that time and the other side's root can only be older), the links are
pointer-jumped into trees, and edges inside a tree die.  Checked here against the sequential sweeps; prints the level counts.
"""
import sys
import numpy as np


def grid_edges(R, C):
    r, c = np.meshgrid(np.arange(R - 1), np.arange(C), indexing="ij"); a = (r * C + c).ravel(); va, vb = a, a + C
    t = 2 * (r * (C - 1) + c).ravel(); outer = 2 * (R - 1) * (C - 1)
    vf = np.where(c.ravel() == 0, outer, t - 1); vg = np.where(c.ravel() == C - 1, outer, t)
    r, c = np.meshgrid(np.arange(R), np.arange(C - 1), indexing="ij"); a = (r * C + c).ravel(); ha, hb = a, a + 1
    t = 2 * (r * (C - 1) + c).ravel()
    hf = np.where(r.ravel() == 0, outer, t - 2 * (C - 1) + 1); hg = np.where(r.ravel() == R - 1, outer, t)
    r, c = np.meshgrid(np.arange(R - 1), np.arange(C - 1), indexing="ij"); a = (r * C + c + 1).ravel(); da, db = a, a + C - 1
    df = 2 * (r * (C - 1) + c).ravel(); dg = df + 1
    return (np.concatenate([va, ha, da]), np.concatenate([vb, hb, db]), np.concatenate([vf, hf, df]), np.concatenate([vg, hg, dg]))


def seq_sweep(na, nb, age, active):
    """Kruskal with the elder rule over edges in array order; returns (is_tree, dead_node) per edge."""
    n = age.size
    p = np.arange(n)
    tree = np.zeros(na.size, bool); dead = np.full(na.size, -1)
    def find(v):
        while p[v] != v:
            p[v] = p[p[v]]; v = p[v]
        return v
    for i in range(na.size):
        if not active[i]: continue
        x, y = find(na[i]), find(nb[i])
        if x == y: continue
        if age[x] < age[y]: p[y] = x; dead[i] = y
        else: p[x] = y; dead[i] = x
        tree[i] = True
    return tree, dead


def level_sweep(na, nb, age, active):
    n = age.size; m = na.size
    cur = np.arange(n)
    alive = active.copy()
    tree = np.zeros(m, bool); dead = np.full(m, -1)
    rank = np.arange(m)
    levels = 0; jumps = 0; hist = []
    while True:
        X = cur[na]; Y = cur[nb]
        alive &= X != Y
        idx = np.nonzero(alive)[0]
        if idx.size == 0: break
        hist.append(idx.size)
        levels += 1
        first = np.full(n, m)
        np.minimum.at(first, X[idx], idx)
        np.minimum.at(first, Y[idx], idx)
        nodes = np.nonzero(first < m)[0]
        e = first[nodes]
        other = np.where(X[e] == nodes, Y[e], X[e])
        younger = age[other] < age[nodes]          # the node across is older: this node's root dies on e
        p = np.arange(n)
        p[nodes[younger]] = other[younger]
        tree[e[younger]] = True
        dead[e[younger]] = nodes[younger]
        while True:
            q = p[p]; jumps += 1
            if np.array_equal(q, p): break
            p = q
        cur = p[cur]
    return tree, dead, levels, jumps, hist


def run(field):
    img = -field.astype(np.float32)
    R, C = img.shape
    val = img.ravel()
    ea, eb, ef, eg = grid_edges(R, C)
    live = ~np.isclose(val, 0)
    keep = live[ea] & live[eb]
    ea, eb, ef, eg = ea[keep], eb[keep], ef[keep], eg[keep]
    ev = np.maximum(val[ea], val[eb])
    order = np.lexsort((np.arange(ea.size), ev))
    ea, eb, ef, eg, ev = ea[order], eb[order], ef[order], eg[order], ev[order]
    m = ea.size
    # primal: older = smaller (value, index)
    age = np.empty(val.size, np.int64); age[np.lexsort((np.arange(val.size), val))] = np.arange(val.size)
    act = np.ones(m, bool)
    t1, d1, lv1, j1, h1 = level_sweep(ea, eb, age, act)
    # dual: reverse order, older = larger (value, index); outer face +inf
    nT = 2 * (R - 1) * (C - 1)
    a, b, d, e_ = img[:-1, :-1], img[:-1, 1:], img[1:, :-1], img[1:, 1:]
    tv = np.empty(nT + 1, np.float32)
    tv[0:nT:2] = np.maximum(np.maximum(a, b), d).ravel(); tv[1:nT:2] = np.maximum(np.maximum(b, d), e_).ravel(); tv[nT] = np.inf
    aged = np.empty(nT + 1, np.int64); aged[np.lexsort((-np.arange(nT + 1), -tv))] = np.arange(nT + 1)
    rf, rg, ract = ef[::-1], eg[::-1], (~t1)[::-1]
    t2, d2, lv2, j2, h2 = level_sweep(rf, rg, aged, ract.copy())
    if "--check" in sys.argv:
        s1, sd1 = seq_sweep(ea, eb, age, act)
        assert np.array_equal(s1, t1) and np.array_equal(sd1, d1), "primal mismatch"
        s2, sd2 = seq_sweep(rf, rg, aged, (~s1)[::-1])
        assert np.array_equal(s2, t2) and np.array_equal(sd2, d2), "dual mismatch"
    return m, lv1, j1, h1, lv2, j2, h2


if __name__ == "__main__":
    d = np.load(sys.argv[1])
    for k in [a for a in sys.argv[2:] if not a.startswith("--")]:
        m, lv1, j1, h1, lv2, j2, h2 = run(d[k])
        print(f"{k}: {m} edges | primal {lv1} levels ({j1} jump rounds) alive {h1[:12]} | dual {lv2} levels ({j2} jump rounds) alive {h2[:12]}")
