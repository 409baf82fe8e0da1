"""CPU: oracle/morse.py against goldens produced by running the reference's topology.MorseGraph
(tools/make_goldens.py morse, run under /opt/conda/bin/python3.9: numpy 1.26.4 -- the reference's pin, where
python-int + float32 promotes to float64, so path lengths accumulate in float64 -- and networkx 2.6.3, whose
_plain_bfs / subgraph-view iteration rules are those of the pinned 3.3).  Branch counts, every bar, the total and
the average must be EQUAL (==), not close."""
from pathlib import Path

import numpy as np
import pytest

from make_goldens import DMT_SYNTH, MORSE_CASES, prune_mask, synth_field
from oracle import dmt as odmt, morse as omorse

GM = np.load(Path(__file__).parent / "golden" / "morse.npz")
GD = np.load(Path(__file__).parent / "golden" / "dmt.npz")


def fields():
    f = {n: synth_field(seed, shape) for n, seed, shape in DMT_SYNTH}
    for n in ("d5", "m1"):
        f[n] = GD["field_" + n].astype(np.float32)
    f["zero"] = np.zeros((24, 24), np.float32)
    return f


FIELDS = fields()


@pytest.mark.parametrize("name", sorted(FIELDS))
@pytest.mark.parametrize("ci", range(len(MORSE_CASES)))
def test_morse_matches_reference(name, ci):
    d1, d2, sw, mn, mx, iso, um = MORSE_CASES[ci]
    f = FIELDS[name]
    V, E = odmt.compute_dmt_graph(f, d1, d2)
    bars, n, tot, avg = omorse.morse_stats(V, E, f.shape, sw, mn, mx, iso, prune_mask(f.shape) if um else None)
    key = f"{name}_c{ci}"
    assert n == int(GM[key + "_count"])
    gb = GM[key + "_bars"].reshape(-1, 2)
    assert np.array_equal(np.array(bars, np.float64).reshape(-1, 2), gb)
    assert tot == float(GM[key + "_total"])
    assert avg == float(GM[key + "_avg"])
