#!/usr/bin/env python3
"""Quick on-GPU timing of the UNet patch batch (dev tool, not the bench contract)."""
import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO / "tissue-model-analysis-tools_amd"))
import numpy as np
from tmat_amd import synth, _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
w = synth.synth_weights(0)
h = _lib.Handle(synth.pack_weights(w), 0, n)
x = np.random.RandomState(0).uniform(0, 1, (n, 320, 320)).astype(np.float32)
h.unet_predict(x[:8])
h.prof_enable(True)
for r in range(reps):
    t = time.time(); y = h.unet_predict(x); dt = time.time() - t
    ms, k, fl = h.prof_read(True)
    print(f"rep {r}: wall {dt*1e3:.1f} ms for {n} patches ({dt/n*1e3:.3f} ms/patch incl. PCIe); "
          f"3x3 MFMA convs: {ms:.2f} ms over {k} launches = {fl/ms/1e9:.1f} TFLOP/s; "
          f"whole-net minimal-form rate {n*26.62e9/dt/1e12:.1f} TFLOP/s", flush=True)
print("out range", float(y.min()), float(y.max()))
