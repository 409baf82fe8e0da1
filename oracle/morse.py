"""oracle/morse.py -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

Restatement of topology.MorseGraph downstream of compute_dmt_graph (reference
fl_tissue_model_tools/topology.py): graph build :530-539, chain smoothing :273-316 / :420-515,
two-pass trimming :588-706, BFS spanning forest :541-581, branch labels :181-222,
barcode :224-271, min-length filter :318-347, statistics :54-65 / :349-356.

The reference's results depend on CPython `set` iteration / pop order and on networkx adjacency
insertion order (hard part 2).  This oracle therefore uses real Python sets
and a real networkx.Graph, built by the same sequence of insertions, so that every
order-dependent choice falls out identically; the C++ product path (csrc/morse.cpp) emulates those
containers and is compared against this file.  Float semantics are those of the pinned numpy
1.26.4 (python-int + float32 promotes to float64): path lengths are float64 sums of float32 edge
lengths.  Pinned by tests/golden/morse.npz (tools/make_goldens.py morse, run under numpy 1.26.4).
"""
from __future__ import annotations

import math

import networkx as nx
import numpy as np

F32 = np.float32


def edge_len32(verts, a, b) -> float:
    """np.linalg.norm of a float32 2-vector: sqrt(dx*dx + dy*dy) in float32 (no FMA)."""
    d = verts[a] - verts[b]
    return float(np.sqrt(F32(F32(d[0] * d[0]) + F32(d[1] * d[1]))))


def graph_from_edges(edges) -> nx.Graph:
    g = nx.Graph()
    for a, b in edges:
        g.add_edge(a, b)        # keys are numpy int32 scalars, as in the reference
    return g


# ---- smoothing ---------------------------------------------------------------------------
def resample_uniform(pts: np.ndarray, n: int) -> np.ndarray:
    """:479-515"""
    seg = pts[1:] - pts[:-1]
    d = np.sqrt(seg[:, 0] * seg[:, 0] + seg[:, 1] * seg[:, 1])
    total = np.sum(d)
    acc = np.cumsum(np.concatenate(([0], d)))
    step = total / (n - 1)
    out = [pts[0]]
    for i in range(1, n - 1):
        s = i * step
        k = np.searchsorted(acc, s, side="right") - 1
        t = (s - acc[k]) / (acc[k + 1] - acc[k])
        out.append(pts[k] + (pts[k + 1] - pts[k]) * t)
    out.append(pts[-1])
    return np.array(out)


def moving_average_fixed_ends(A: np.ndarray, n: int) -> np.ndarray:
    """:421-476 -- end-weighted padding, float64 running mean, arc-length resampling."""
    n = min(n, math.ceil(len(A) / 2))
    if n == 1:
        return A
    core = A[n - 1 : -(n - 1)]
    for i in reversed(range(n - 1)):
        rep = n - i
        core = np.concatenate(([A[i]] * rep, core, [A[-i - 1]] * rep))
    cs = np.cumsum(core, axis=0, dtype=float)
    cs[n:] = cs[n:] - cs[:-n]
    return resample_uniform(cs[n - 1 :] / n, len(A))


def smooth_vertices(G: nx.Graph, verts: np.ndarray, window: int) -> np.ndarray:
    """:273-316"""
    if window <= 1:
        return verts
    verts = verts.copy()
    fixed = {v for v in G.nodes if G.degree[v] != 2}
    visited = set()
    for start in fixed:
        for base in G.neighbors(start):
            cur = base
            if cur in visited:
                continue
            chain = [start, cur]
            seen = set()
            while G.degree[cur] == 2:
                nb = list(G.neighbors(cur))
                nxt = nb[0] if nb[0] != cur else nb[1]
                if nxt in seen:
                    break
                cur = nxt
                seen.add(cur)
                chain.append(cur)
            verts[chain] = moving_average_fixed_ends(verts[chain], window)
            visited.update([chain[0], chain[-1]])
    return verts


# ---- trimming ----------------------------------------------------------------------------
def trim_graph(G: nx.Graph, verts, shape, min_len, max_len=None, pruning_mask=None, remove_isolated=False):
    """:588-706"""
    G = G.copy()
    if pruning_mask is None:
        pruning_mask = np.zeros(shape, dtype=bool)
    elif pruning_mask.dtype != bool:
        pruning_mask = pruning_mask > 0

    def bbox_diag(seg):
        p = verts[seg]
        lo = np.array([np.min(p[:, 0]), np.min(p[:, 1])])
        hi = np.array([np.max(p[:, 0]), np.max(p[:, 1])])
        return np.sqrt(np.sum((hi - lo) ** 2))

    pass_num, done = 1, False
    while not done:
        junctions = {n for n in G.nodes if G.degree[n] > 2}
        bases = {n for n in G.nodes if G.degree[n] == 1} if pass_num == 1 else junctions
        unmarked = {n for n in G.nodes if n not in junctions}
        keep, short, long_, isolated = [], [], [], []
        while bases:
            s0 = bases.pop()
            nbrs = {n for n in G.neighbors(s0) if n in unmarked}
            while nbrs:
                node = nbrs.pop()
                seg = [s0, node] if pass_num == 1 else [node]
                while True:
                    nxt = [n for n in G.neighbors(node) if n in unmarked]
                    if not nxt:
                        break
                    node = nxt[0]
                    seg.append(node)
                    unmarked.remove(node)
                n_leaf = (G.degree[seg[0]] == 1) + (G.degree[seg[-1]] == 1)
                if remove_isolated and n_leaf == 2 and not any(G.degree[v] > 2 for v in seg):
                    isolated.append(seg)
                elif n_leaf > 0:
                    if bbox_diag(seg) < min_len:
                        short.append(seg)
                    elif max_len and bbox_diag(seg) > max_len:
                        long_.append(seg)
                    else:
                        keep.append(seg)
                else:
                    keep.append(seg)
        if keep:
            pos = [np.round(np.median(verts[s], axis=0)).astype(int) for s in keep]
            hit = np.argwhere(pruning_mask[tuple(zip(*pos))]).flatten()
            doomed = [keep[i] for i in hit]
        else:
            doomed = []
        doomed += short + long_ + isolated
        for seg in doomed:
            G.remove_edges_from(set(G.edges(seg)))
            G.remove_nodes_from(seg)
        G.remove_nodes_from(list(nx.isolates(G)))
        done = pass_num == 2 and not doomed
        pass_num = 2 if pass_num == 1 else 1
    return G


# ---- forest, labels, barcode -------------------------------------------------------------
def spanning_forest(G: nx.Graph, verts, remove_isolated):
    """:541-581"""
    forest = nx.Graph()
    parent = {n: None for n in G.nodes()}
    dist = {}
    for g in [G.subgraph(c) for c in nx.connected_components(G)]:
        root, maxdeg = max(g.degree, key=lambda kv: kv[1])
        if remove_isolated and maxdeg <= 2:
            continue
        parent[root] = root
        dist[root] = 0.0
        queue = [root]
        while queue:
            v = queue.pop(0)
            for n in g.neighbors(v):
                if parent[n] is None:
                    forest.add_edge(v, n)
                    parent[n] = v
                    dist[n] = dist[v] + edge_len32(verts, v, n)
                    queue.append(n)
    return forest, parent, dist


def morse_stats(vertices, edges, shape, smoothing_window, min_branch_length, max_branch_length=None,
                remove_isolated_branches=False, pruning_mask=None):
    """MorseGraph.__init__ after compute_dmt_graph (:148-179, :48-50).  Returns
    (barcode list[(birth, death)], count, total, average)."""
    G = graph_from_edges(edges)
    verts = np.asarray(vertices).astype(np.float32)
    verts = smooth_vertices(G, verts, smoothing_window)
    G = trim_graph(G, verts, shape, min_branch_length, max_branch_length, pruning_mask, remove_isolated_branches)
    forest, parent, dist_root = spanning_forest(G, verts, remove_isolated_branches)

    # branch labels (:181-222)
    leaves = [n for n in forest.nodes if forest.degree[n] == 1]
    far = {v: -np.inf for v in forest.nodes}
    label = {}
    for leaf in leaves:
        cur, par = leaf, parent[leaf]
        far[leaf] = d = 0.0
        label[leaf] = leaf
        while par != cur:
            d += edge_len32(verts, par, cur)
            if d < far[par]:
                break
            cur, par = par, parent[par]
            far[cur] = d
            label[cur] = leaf
    # barcode (:224-271) and min-length filter (:318-347)
    bars = []
    for leaf in leaves:
        cur, lab, par, d = leaf, leaf, parent[leaf], 0.0
        while lab == leaf and cur != par:
            d += edge_len32(verts, par, cur)
            cur, par = par, parent[par]
            lab = label[cur]
        birth = -dist_root[leaf]
        bars.append((birth, birth + d))
    bars = [(b, e) for b, e in bars if e - b >= min_branch_length]
    # statistics (:54-65, :349-356)
    if bars:
        arr = np.array(bars)
        lens = arr[:, 1] - arr[:, 0]
        lens = lens[~np.isinf(lens)]
    else:
        lens = np.array([])
    total = float(np.sum(lens))
    avg = 0.0 if total == 0 else total / len(lens)
    return bars, len(bars), total, avg
