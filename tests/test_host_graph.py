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
