#!/usr/bin/env python3
"""dev soak (GPU box): the device DMT path (keys, sort, level sweeps) against the host-only execution on random fields of random
shapes -- smooth, tied, sparse, staircase-like -- singly and in batches.   python tools/dev/soak_dmt.py [cases]"""
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(REPO / "tissue-model-analysis-tools_amd"), str(REPO / "tools")]
import numpy as np
from scipy.ndimage import gaussian_filter
from tmat_amd import _lib

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
h = _lib.Handle(None, 0)
rs = np.random.RandomState(2026)
bad = 0


def field(kind, shape):
    if kind == 0:      # smooth: a few levels, long monotone slopes
        f = gaussian_filter(rs.uniform(0, 1, shape), rs.uniform(1, 6))
        f = (f - f.min()) / max(f.max() - f.min(), 1e-9) * 255
        f[f < rs.uniform(0, 120)] = 0
    elif kind == 1:    # few distinct values: ties everywhere
        f = np.round(rs.uniform(0, rs.randint(2, 9), shape)) * 30
        f[rs.uniform(size=shape) < rs.uniform(0, 0.4)] = 0
    elif kind == 2:    # noise: many minima
        f = rs.uniform(1, 255, shape)
    else:              # ramps with alternating ridges: many levels
        c = np.arange(shape[1]); row = np.where(c % 2 == 0, 1000.0 - c, 1.0 + c)
        f = row[None, :] + rs.uniform(0, 0.4, shape)
    return f.astype(np.float32)


for case in range(cases):
    shape = (int(rs.randint(2, 200)), int(rs.randint(2, 200)))
    n = int(rs.choice([1, 2, 5, 8, 9, 16, 33]))
    stack = np.stack([field(int(rs.randint(0, 4)), shape) for _ in range(n)])
    d = [(5.0, 10.0), (2.0, 4.0), (0.5, 0.0)][case % 3]
    got = _lib.dmt_graph_batch(stack, *d, handle=h)
    for k in range(n):
        V0, E0 = _lib.dmt_graph(stack[k], *d)
        if not (np.array_equal(V0, got[k][0]) and np.array_equal(E0, got[k][1])):
            bad += 1
            print("MISMATCH case", case, "field", k, shape, d, flush=True)
    if case % 20 == 19:
        print(f"{case + 1} cases, {bad} mismatching fields", flush=True)
print(f"soak done: {cases} cases, {bad} mismatching fields")
h.close()
sys.exit(1 if bad else 0)
