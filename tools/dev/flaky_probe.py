"""dev probe: fresh handles, first call, small n: how often does the f32 path miss the oracle bits?"""
import os, sys
from pathlib import Path
REPO = Path(__file__).resolve().parents[2]
sys.path[:0] = [str(REPO), str(REPO / "tissue-model-analysis-tools_amd")]
import numpy as np
from tmat_amd import synth, _lib
from oracle import unet as ou
w = synth.synth_weights(0)
blob = synth.pack_weights(w)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for n in (2, 3):
    x = np.random.RandomState(33 + n).uniform(0, 1, (n, 320, 320)).astype(np.float32)
    ex = ou.forward_exact(w, x)
    bad = []
    for it in range(iters):
        h = _lib.Handle(blob, 0, 1600 if it % 2 == 0 else 64)
        for rep in range(2):
            y = h.unet_predict(x)
            nb = int((y.view(np.uint32) != ex.view(np.uint32)).sum())
            if nb:
                bad.append((it, rep, nb, float(np.abs(y - ex).max())))
        h.close()
    print("TMAT_FUSED_SEP", os.environ.get("TMAT_FUSED_SEP", "1"), "n", n, "iters", iters, "bad", bad, flush=True)
