#!/usr/bin/env python3
"""dev tool (GPU box): TMAT_TRACE timings of the host stages per pass, with the DMT front end on the device and on the host"""
import os, subprocess, sys
from pathlib import Path
REPO = Path(__file__).resolve().parents[1]
code = '''
import sys, ctypes
sys.path.insert(0, r"%s/tissue-model-analysis-tools_amd")
import numpy as np
from tmat_amd import _lib, branches, synth
h = _lib.Handle(synth.pack_weights(synth.synth_weights(0)), 0, 1600)
imgs = np.stack([synth.synth_image(i, 1024) for i in range(16)])
cfg = dict(graph_thresh_1=5, graph_thresh_2=10, graph_smoothing_window=12, min_branch_length=12)
for rep in range(2):
    rows = branches.analyze_batch(h, imgs, cfg, 1000.0)
print("rows", rows[:3])
h.close()
''' % REPO
for dev in ("1", "0"):
    env = dict(os.environ, TMAT_TRACE="1", TMAT_DMT_DEVICE=dev)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(f"== TMAT_DMT_DEVICE={dev}")
    print("\n".join(l for l in r.stderr.splitlines() if "host pass" in l))
    print(r.stdout.strip()[-200:])
