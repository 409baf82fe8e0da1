"""CPU: the product's host-side graph stages (tmat_dmt_graph, tmat_morse_stats; C-ABI, no GPU
needed) against the oracle restatements and the reference-generated goldens."""
from pathlib import Path

import numpy as np
import pytest

from make_goldens import DELTAS, DMT_SYNTH, MORSE_CASES, prune_mask, synth_field
from oracle import dmt as odmt, morse as omorse
from tmat_amd import _lib

GD = np.load(Path(__file__).parent / "golden" / "dmt.npz")


def all_fields():
    f = {n: synth_field(seed, shape) for n, seed, shape in DMT_SYNTH}
    for n in ("d5", "m1", "ties"):
        f[n] = GD["field_" + n].astype(np.float32)
    f["zero"] = np.zeros((24, 24), np.float32)
    f["const"] = np.full((24, 30), 7.0, np.float32)
    one = np.zeros((16, 16), np.float32); one[5, 9] = 200.0
    f["single"] = one
    return f


FIELDS = all_fields()


@pytest.mark.parametrize("name", sorted(FIELDS))
@pytest.mark.parametrize("deltas", DELTAS)
def test_product_dmt_matches_reference_golden(name, deltas):
    V, E = _lib.dmt_graph(FIELDS[name], *deltas)
    key = f"{name}_{deltas[0]}_{deltas[1]}"
    assert np.array_equal(V, GD[key + "_V"].reshape(-1, 2))
    assert np.array_equal(E, GD[key + "_E"].reshape(-1, 2))


@pytest.mark.parametrize("seed", range(6))
def test_product_dmt_matches_oracle_random(seed):
    rs = np.random.RandomState(100 + seed)
    shape = (int(rs.randint(3, 70)), int(rs.randint(3, 70)))
    f = synth_field(200 + seed, shape) if seed % 2 else np.round(rs.uniform(0, 6, shape)).astype(np.float32) * 40
    for d in DELTAS:
        V0, E0 = odmt.compute_dmt_graph(f, *d)
        V1, E1 = _lib.dmt_graph(f, *d)
        assert np.array_equal(V0, V1) and np.array_equal(E0, E1)


def _cmp_morse(f, case):
    d1, d2, sw, mn, mx, iso, um = case
    V, E = odmt.compute_dmt_graph(f, d1, d2)
    pm = prune_mask(f.shape) if um else None
    bars0, n0, tot0, avg0 = omorse.morse_stats(V, E, f.shape, sw, mn, mx, iso, pm)
    bars1, n1, tot1, avg1 = _lib.morse_stats(V, E, f.shape, sw, mn, mx, iso, pm)
    assert n0 == n1
    assert np.array_equal(np.array(bars0, np.float64).reshape(-1, 2), bars1)
    assert tot0 == tot1 and avg0 == avg1
    return n0


@pytest.mark.parametrize("name", ["s96", "s_rect", "s160", "d5", "m1", "zero", "ties"])
def test_product_morse_matches_oracle(name):
    total = 0
    for case in MORSE_CASES:
        total += _cmp_morse(FIELDS[name], case)
    if name not in ("zero",):
        assert total > 0


@pytest.mark.parametrize("seed", range(8))
def test_product_morse_matches_oracle_random(seed):
    rs = np.random.RandomState(300 + seed)
    shape = (int(rs.randint(40, 140)), int(rs.randint(40, 140)))
    f = synth_field(400 + seed, shape)
    case = (float(rs.choice([0.5, 2, 5])), float(rs.choice([0, 4, 10])), int(rs.randint(1, 14)), int(rs.randint(1, 14)),
            [None, 30, 60][rs.randint(3)], bool(rs.randint(2)), bool(rs.randint(2)))
    _cmp_morse(f, case)


# ---- size-independent properties of the graph stages ("DMT invariants") on random fields, product host path ----
def _random_field(seed, shape, zeros=0.3):
    rs = np.random.RandomState(seed)
    f = synth_field(seed, shape).astype(np.float32)
    if zeros:                                                       # holes: the reference drops edges that touch a ~0 pixel (dmtgraph.py:71-77)
        f[rs.uniform(size=shape) < zeros * rs.uniform()] = 0.0
    return f


@pytest.mark.parametrize("seed,shape", [(1, (40, 56)), (2, (97, 64)), (3, (128, 128)), (4, (384, 384))])
def test_dmt_graph_invariants(seed, shape):
    """every edge joins two distinct vertices that are neighbours in the anti-diagonal triangulation (dmtgraph.py:202-274: vertical, horizontal,
    (r, c + 1)-(r + 1, c)); vertices and edges are unique, inside the grid, and sit on non-zero pixels"""
    f = _random_field(seed, shape)
    for d in DELTAS:
        V, E = _lib.dmt_graph(f, *d)
        assert V.ndim == 2 and V.shape[1] == 2 and E.ndim == 2 and E.shape[1] == 2
        if len(V) == 0:
            assert len(E) == 0
            continue
        assert V[:, 0].min() >= 0 and V[:, 0].max() < shape[0] and V[:, 1].min() >= 0 and V[:, 1].max() < shape[1]
        assert len(np.unique(V[:, 0].astype(np.int64) * shape[1] + V[:, 1])) == len(V)
        assert np.all(f[V[:, 0], V[:, 1]] != 0)
        if len(E):
            assert E.min() >= 0 and E.max() < len(V)
            a, b = V[E[:, 0]].astype(np.int64), V[E[:, 1]].astype(np.int64)
            dr, dc = b[:, 0] - a[:, 0], b[:, 1] - a[:, 1]
            assert np.all((np.abs(dr) <= 1) & (np.abs(dc) <= 1) & ((dr != 0) | (dc != 0)))
            diag = (dr != 0) & (dc != 0)
            assert np.all(dr[diag] * dc[diag] == -1)                 # the anti-diagonal only
            lo, hi = np.minimum(E[:, 0], E[:, 1]).astype(np.int64), np.maximum(E[:, 0], E[:, 1]).astype(np.int64)
            assert len(np.unique(lo * len(V) + hi)) == len(E)
            used = np.zeros(len(V), bool); used[E.ravel()] = True
            assert used.all()                                        # no vertex without an edge is emitted


@pytest.mark.parametrize("seed", range(4))
def test_dmt_graph_scales_with_the_field(seed):
    """persistence is a difference of field values: multiplying the field AND both thresholds by a power of two (exact in float32)
    leaves the graph unchanged"""
    f = _random_field(10 + seed, (72, 90))
    for d in DELTAS:
        V0, E0 = _lib.dmt_graph(f, *d)
        for k in (0.5, 4.0):
            V1, E1 = _lib.dmt_graph(f * np.float32(k), d[0] * k, d[1] * k)
            assert np.array_equal(V0, V1) and np.array_equal(E0, E1)


@pytest.mark.parametrize("seed", range(4))
def test_morse_stats_are_consistent(seed):
    """count >= 0, one bar per counted branch at most, total = sum over the counted branches, average = total / count"""
    f = _random_field(20 + seed, (120, 96), zeros=0.0)
    V, E = _lib.dmt_graph(f, 5.0, 10.0)
    for sw, mn in ((1, 1), (12, 12), (5, 30)):
        bars, n, tot, avg = _lib.morse_stats(V, E, f.shape, sw, mn, None, False, None)
        assert n >= 0 and tot >= 0
        if n:
            assert abs(avg * n - tot) <= 1e-9 * max(1.0, tot)
        else:
            assert tot == 0
