"""CPU: oracle/dmt.c against golden vertices/edges produced by importing the reference's
dmtgraph.compute_dmt_graph (tools/make_goldens.py dmt)."""
from pathlib import Path

import numpy as np
import pytest

from make_goldens import DELTAS, DMT_SYNTH, synth_field
from oracle import dmt

G = np.load(Path(__file__).parent / "golden" / "dmt.npz")


def fields():
    f = {n: synth_field(seed, shape) for n, seed, shape in DMT_SYNTH}
    for n in ("d5", "m1", "ties"):
        f[n] = G["field_" + n].astype(np.float32)
    f["zero"] = np.zeros((24, 24), np.float32)
    f["const"] = np.full((24, 30), 7.0, np.float32)
    one = np.zeros((16, 16), np.float32); one[5, 9] = 200.0
    f["single"] = one
    return f


FIELDS = fields()


@pytest.mark.parametrize("name", sorted(FIELDS))
@pytest.mark.parametrize("deltas", DELTAS)
def test_dmt_graph_exact(name, deltas):
    V, E = dmt.compute_dmt_graph(FIELDS[name], *deltas)
    key = f"{name}_{deltas[0]}_{deltas[1]}"
    assert np.array_equal(V, G[key + "_V"].reshape(-1, 2))
    assert np.array_equal(E, G[key + "_E"].reshape(-1, 2))
