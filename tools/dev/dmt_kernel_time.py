#!/usr/bin/env python3
"""dev tool (GPU box, run under rocprofv3 --kernel-trace --stats): tmat_dmt_graph on the 384 x 384 golden fields, idle chip."""
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(REPO), str(REPO / "tissue-model-analysis-tools_amd")]
import numpy as np
from tmat_amd import _lib
d = np.load(REPO / "tests/golden/dmt.npz")
h = _lib.Handle(None, 0)
for _ in range(5):
    for k in ("field_d5", "field_m1"):
        _lib.dmt_graph(d[k].astype(np.float32), 5.0, 10.0, handle=h)
h.close()
