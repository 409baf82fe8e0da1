#!/usr/bin/env python3
"""dev soak (GPU box): the whole batch path (three streams, host stages) on the same 8 images in permuted batch positions, many times;
every occurrence of an image must give the same row.   python tools/dev/soak_pipeline.py [repetitions] [images per repetition]"""
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO / "tissue-model-analysis-tools_amd"))
import numpy as np
from tmat_amd import _lib, branches, synth

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
per = int(sys.argv[2]) if len(sys.argv) > 2 else 64
h = _lib.Handle(synth.pack_weights(synth.synth_weights(0)), 0, 1600)
base = [synth.synth_image(i, 1024) for i in range(8)]
cfg = dict(graph_thresh_1=5, graph_thresh_2=10, graph_smoothing_window=12, min_branch_length=12)
ref, bad = None, 0
for rep in range(reps):
    order = np.random.RandomState(rep).permutation(per) % 8
    rows = branches.analyze_batch(h, np.stack([base[j] for j in order]), cfg, 1000.0)
    got = {}
    for j, r in zip(order, rows):
        got.setdefault(int(j), set()).add(r[1:])
    if ref is None:
        ref = {j: next(iter(v)) for j, v in got.items()}
    for j, v in got.items():
        if v != {ref[j]}:
            bad += 1
            print("MISMATCH repetition", rep, "image", j, sorted(v), "reference", ref[j], flush=True)
    if rep % 5 == 4:
        print(f"{rep + 1} repetitions of {per} images, {bad} mismatching", flush=True)
print(f"soak done: {reps * per} images, {bad} mismatching")
h.close()
sys.exit(1 if bad else 0)
