"""oracle/dmt.py -- TEST INFRASTRUCTURE ONLY.  ctypes wrapper of oracle/dmt.c
(compute_dmt_graph restatement, reference dmtgraph.py:38-99)."""
import ctypes

import numpy as np

from . import unet as _u


def compute_dmt_graph(img, delta1, delta2=0.0):
    L = _u.lib()
    img = np.ascontiguousarray(img, np.float32)
    R, C = img.shape
    cap_v = R * C + 4
    cap_e = 3 * R * C + 4
    V = np.empty((cap_v, 2), np.int32)
    E = np.empty((cap_e, 2), np.int32)
    nv, ne = ctypes.c_int(), ctypes.c_int()
    rc = L.orc_dmt_graph(img.ctypes.data_as(ctypes.c_void_p), R, C, ctypes.c_float(delta1), ctypes.c_float(delta2),
                         V.ctypes.data_as(ctypes.c_void_p), cap_v, E.ctypes.data_as(ctypes.c_void_p), cap_e,
                         ctypes.byref(nv), ctypes.byref(ne))
    assert rc == 0, rc
    return V[: nv.value].copy(), E[: ne.value].copy()
