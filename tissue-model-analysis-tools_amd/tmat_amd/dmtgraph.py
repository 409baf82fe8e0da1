"""Mirror of fl_tissue_model_tools.dmtgraph.compute_dmt_graph (reference dmtgraph.py:38-99)."""
from __future__ import annotations

import numpy as np

from . import _lib


def compute_dmt_graph(img, delta1: float, delta2: float = 0.0):
    """-> (vertices (n, 2) int32 [row, col], edges (m, 2) int32)"""
    return _lib.dmt_graph(np.asarray(img, np.float32), delta1, delta2)
