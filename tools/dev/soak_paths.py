#!/usr/bin/env python3
"""dev soak (GPU box): the shipped kernels (fused separable layers with the stem recomputed inside, activated block-output copies) against
the unfused reference forms of the same layers (separate stem / depthwise / pointwise / pooling kernels, ReLU on load), bit for bit, over many random batches --
two handles in one process (the switches are read when a handle is created).  Catches rare hazards (inline assembly, barriers, LDS aliasing)
that a single parity test can miss.   python tools/dev/soak_paths.py [iterations] [patches]"""
import os
import sys
import time
from pathlib import Path
REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO / "tissue-model-analysis-tools_amd"))
import numpy as np
from tmat_amd import synth, _lib

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
blob = synth.pack_weights(synth.synth_weights(0))
h_new = _lib.Handle(blob, 0, n)
REF = ("TMAT_FUSED_SEP", "TMAT_FUSED_POOL", "TMAT_STEM_FUSED", "TMAT_RELU_COPY")
for k in REF:
    os.environ[k] = "0"
h_ref = _lib.Handle(blob, 0, n)                      # separate stem + depthwise + pointwise + pooling kernels, ReLU on load
for k in REF:
    del os.environ[k]
rs = np.random.RandomState(12345)
bad = 0
t0 = time.time()
for it in range(iters):
    k = int(rs.randint(1, n + 1))
    x = rs.uniform(0, 1, (k, 320, 320)).astype(np.float32)
    if it % 7 == 0:
        x[0, : rs.randint(1, 320)] = 0.0
    a = h_new.unet_predict(x)
    b = h_ref.unet_predict(x)
    a2 = h_new.unet_predict(x)
    if not (np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.array_equal(a.view(np.uint32), a2.view(np.uint32))):
        bad += 1
        print(f"iteration {it}: {k} patches: shipped vs unfused differ in {(a != b).sum()} values, repeat differs in {(a != a2).sum()}", flush=True)
    if it % 50 == 49:
        print(f"{it + 1} iterations, {bad} mismatching, {time.time() - t0:.0f} s", flush=True)
print(f"soak done: {iters} iterations, {bad} mismatching")
h_new.close(); h_ref.close()
sys.exit(1 if bad else 0)
