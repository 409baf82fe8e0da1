#!/usr/bin/env python3
"""Measurement of BASELINE config #5 on one MI355X: compute_cell_area and the compute_inv_depth ensemble on
synthetic images, host buffers in (PCIe inclusive).  Prints two JSON lines in bench.py's format:
  * cell area: images/s for 1024 x 1024 uint16 images -> 512 x 512 -> GMM threshold -> area (batches of 64);
  * invasion depth: Z slices/s for 32-slice 512 x 512 stacks through 3 ResNet50(conv4_block6_out) classifiers at 256 x 256.
cpu_baseline: the oracles (numpy / scikit-learn-equivalent EM; oracle/resnet.py through the C convolution) on a bounded sample.

    python tools/bench_config5.py [--images 256] [--stacks 8] [--steps 3] [--no-cpu]
"""
import argparse
import json
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tissue-model-analysis-tools_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--images", type=int, default=512, help="BASELINE config #5: 512 images")
    ap.add_argument("--stacks", type=int, default=16, help="16 stacks x 32 slices = 512 images for the invasion-depth tool")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    from tmat_amd import _lib, inv_depth, preprocessing, synth
    h = _lib.Handle(None, 0)
    base = np.stack([synth.synth_image(i, 1024, n_vessels=40) for i in range(8)])
    imgs = base[np.arange(a.images) % 8]

    def cell_step():
        out = []
        for i0 in range(0, a.images, 64):
            out.append(preprocessing.cell_area_batch(h, imgs[i0:i0 + 64], 512, 0.0)[0])
        return np.concatenate(out)
    area = cell_step()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        area = cell_step()
    dt = (time.perf_counter() - t0) / a.steps
    line = {"metric": "images/sec through compute_cell_area (1024x1024 uint16 -> 512 -> GMM threshold -> area)", "value": a.images / dt, "unit": "images/s",
            "n_gpus": 1, "steps": a.steps, "warmup": 1, "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64 (EM on the histogram) / u16", "data": "synthetic", "config": {"workload": f"{a.images} images, host buffers in, batches of 64", "sd_coef": 0.0}}
    # roofline of the tool as a whole: HBM-bound by construction -- algorithmic bytes per image = the 1024^2 u16 input (2 MB) read once
    # + the 512^2 u16 down-sampled image written and read twice (histogram, threshold: 1.5 MB) + the 512^2 u8 result (0.25 MB);
    # the host-pointer entry moves the same 2 MB over PCIe first, which is what bounds this measurement (63 GB/s spec)
    alg_bytes = 1024 * 1024 * 2 + 3 * 512 * 512 * 2 + 512 * 512
    line["roofline"] = {"bound": "hbm", "achieved": alg_bytes * a.images / dt / 1e9, "peak": 8000.0, "unit": "GB/s",
                        "frac": alg_bytes * a.images / dt / 1e9 / 8000.0, "traffic": None,
                        "note": "whole tool, PCIe inclusive (2 MB/image of H2D at <= 63 GB/s = at most 31 500 images/s); per kernel see profiles/r04_config5_kernel_stats.csv"}
    if not a.no_cpu:
        from oracle import cellarea as ca
        c0 = time.perf_counter()
        ref = [ca.cell_area(base[i], 512, 0.0)[0] for i in range(4)]
        c1 = time.perf_counter()
        line["cpu_baseline"] = {"value": 4 / (c1 - c0), "unit": "images/s", "cores": 1, "kind": "port", "sample": "4 images through oracle/cellarea.py"}
        line["parity"] = bool(np.array_equal(area[:4], ref))
    print(json.dumps(line), flush=True)

    stacks = [synth.synth_stack(i, 32, 512, 512, n_vessels=16) for i in range(2)]
    ws = [inv_depth.synth_resnet_weights(s) for s in range(3)]
    ens = inv_depth.InvDepthEnsemble(h, ws)
    probs = ens.predict_stack(stacks[0])
    batch = [stacks[k % 2] for k in range(a.stacks)]
    t0 = time.perf_counter()
    for _ in range(a.steps):
        all_probs = ens.predict_stacks(batch)                # stacks of one shape ride together, 128 slices per call
    dt = (time.perf_counter() - t0) / a.steps
    probs = all_probs[a.stacks - 1]
    nsl = a.stacks * 32
    flops = inv_depth.flops_per_slice(ws[0], 256) * 3 * nsl
    line = {"metric": "Z slices/sec through compute_inv_depth (3 x ResNet50 conv4_block6_out at 256x256)", "value": nsl / dt, "unit": "slices/s", "n_gpus": 1,
            "steps": a.steps, "warmup": 1, "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "config": {"workload": f"{a.stacks} stacks of 32 x 512 x 512 u16, host buffers in, 4 stacks per call; 3 ensemble members, random-init weights"},
            "roofline": {"bound": "mfma", "achieved": flops / dt / 1e12, "peak": 157.3, "unit": "TFLOP/s", "frac": flops / dt / 1e12 / 157.3, "traffic": None,
                         "note": "convolution FLOPs of the three ResNet50(conv4_block6_out) members (2 MACs, stem included) over the whole call: upload, resize, "
                                 "preparation, stem, pooling and head are inside the time"}}
    if not a.no_cpu:
        from oracle import resnet as orr
        c0 = time.perf_counter()
        ox = orr.prep_inv_depth_imgs(stacks[(a.stacks - 1) % 2][:4], 256)
        ref = np.stack([orr.forward(w, ox) for w in ws], axis=1)
        c1 = time.perf_counter()
        line["cpu_baseline"] = {"value": 4 / (c1 - c0), "unit": "slices/s", "cores": "all (OpenMP C convolution)", "kind": "port", "sample": "4 slices x 3 models through oracle/resnet.py"}
        line["parity"] = bool(np.array_equal(probs[:4].view(np.uint32), ref.view(np.uint32)))
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
