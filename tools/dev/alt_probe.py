import sys
from pathlib import Path
REPO = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(REPO), str(REPO / "tissue-model-analysis-tools_amd")]
import numpy as np, torch
from tmat_amd import synth, _lib
from oracle import unet as ou
w = synth.synth_weights(0)
blob = synth.pack_weights(w)
for n in (2, 4):
    x = np.random.RandomState(33).uniform(0, 1, (n, 320, 320)).astype(np.float32)
    ex = ou.forward_exact(w, x)
    r64 = np.asarray(ou.forward_torch(w, x, dtype=torch.float64), np.float64)
    print(n, "exact vs f64", np.abs(ex - r64).max())
    h = _lib.Handle(blob, 0, 1600)
    outs = {}
    for mode in ("f32", "bf16x3", "bf16x6", "f32", "bf16x3"):
        h.set_precision(mode)
        y = h.unet_predict(x)
        print(n, mode, "vs exact", np.abs(y.astype(np.float64) - ex).max(), "vs f64", np.abs(y - r64).max())
    h.close()
    for mode in ("bf16x3", "bf16x6"):
        h = _lib.Handle(blob, 0, 1600)
        h.set_precision(mode)
        y = h.unet_predict(x)
        print(n, "fresh", mode, "vs exact", np.abs(y.astype(np.float64) - ex).max())
        h.close()
