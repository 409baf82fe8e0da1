"""Mirror of fl_tissue_model_tools.dmtgraph.compute_dmt_graph (reference dmtgraph.py:38-99)."""
from __future__ import annotations

import numpy as np

from . import _lib


def compute_dmt_graph(img, delta1: float, delta2: float = 0.0, handle=None):
    """-> (vertices (n, 2) int32 [row, col], edges (m, 2) int32).  With a handle (`_lib.Handle`) the edge keys, the lower-star sort and both
    persistence sweeps run on its GPU (same result); without one everything runs on the host, as the reference does."""
    return _lib.dmt_graph(np.asarray(img, np.float32), delta1, delta2, handle=handle)


def compute_dmt_graphs(imgs, delta1: float, delta2: float = 0.0, handle=None):
    """compute_dmt_graph for a stack of fields of one shape, (n, rows, cols): one launch per stage for all of them (tmat_dmt_graph_batch)
    -> list of (vertices, edges)"""
    return _lib.dmt_graph_batch(np.asarray(imgs, np.float32), delta1, delta2, handle=handle)
