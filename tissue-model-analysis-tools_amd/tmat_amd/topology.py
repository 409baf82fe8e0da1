"""Mirror of fl_tissue_model_tools.topology.MorseGraph (reference topology.py:15-65, 148-356).
Plotting methods of the reference (matplotlib) are outside the hot path and not provided."""
from __future__ import annotations

from numbers import Number
from typing import Optional, Tuple

import numpy as np

from . import _lib


class MorseGraph:
    """Morse skeleton of an image represented as a forest; exposes `.barcode`,
    `get_total_branch_length()`, `get_average_branch_length()`."""

    def __init__(self, img, thresholds: Tuple[Number, Number] = (1, 4), min_branch_length: int = 15,
                 max_branch_length: Optional[int] = None, remove_isolated_branches: bool = False,
                 smoothing_window: int = 15, pruning_mask=None, method=0):
        img = np.asarray(img)
        self.thresholds = thresholds
        self.min_branch_length = min_branch_length
        self.max_branch_length = max_branch_length
        self.remove_isolated_branches = remove_isolated_branches
        self.smoothing_window = smoothing_window
        self.pruning_mask = pruning_mask
        self._shape = img.shape[:2]
        V, E = _lib.dmt_graph(img.astype(np.float32), thresholds[0], thresholds[1])
        self._dmt_vertices, self._dmt_edges = V, E
        bars, n, tot, avg = _lib.morse_stats(V, E, self._shape, smoothing_window, min_branch_length, max_branch_length,
                                             remove_isolated_branches, pruning_mask)
        self.barcode = [(float(b), float(d)) for b, d in bars]
        self._total, self._avg = tot, avg

    def get_total_branch_length(self) -> float:
        return self._total

    def get_average_branch_length(self) -> float:
        return self._avg

    def plot_colored_barcode(self, *a, **k):
        raise NotImplementedError("visualisation is outside the accelerated path")

    plot_colored_tree = plot_colored_barcode
