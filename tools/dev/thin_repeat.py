#!/usr/bin/env python3
"""dev tool (GPU box): tmat_medial_axis_batch on the same batch over and over, compared with the host implementation"""
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO / "tissue-model-analysis-tools_amd"))
import numpy as np
from scipy import ndimage as ndi
from tmat_amd import _lib
rs = np.random.RandomState(1)
masks = np.stack([ndi.gaussian_filter(rs.normal(size=(640, 640)), s) > t for s, t in ((8, 0.0), (5, 0.01), (12, -0.005), (3, 0.02), (20, 0.0), (6, 0.0), (9, 0.01), (4, -0.01))])
h = _lib.Handle(None, 0)
ref = [_lib.host_medial_axis(m)[0] for m in masks]
bad = 0
for rep in range(30):
    sk, _ = h.medial_axis(masks)
    for i in range(len(masks)):
        if not np.array_equal(sk[i], ref[i]):
            bad += 1
            print("rep", rep, "image", i, "differs in", int((sk[i] != ref[i]).sum()), "pixels")
print("bad", bad)
h.close()
