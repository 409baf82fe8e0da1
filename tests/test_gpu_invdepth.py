"""GPU parity: the invasion-depth classifier (reference scripts/compute_inv_depth.py, models.py:build_ResNet50_TL) through
tmat_resnet_load / tmat_resnet_predict / tmat_inv_depth_predict against oracle/resnet.py, bit-exact (float32).  PARITY UNPINNED
against Keras (TensorFlow absent; see oracle/resnet.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def plain():
    from tmat_amd import _lib
    h = _lib.Handle(None, 0)
    yield h
    h.close()


def test_small_trunk_bitexact(plain):
    """ResNet50 cut at conv3_block1_out (stem, pool, stage 2, one strided block with its projection shortcut) on 64 x 64 inputs:
    every kernel kind of the model, small enough for the CPU oracle"""
    from oracle import resnet as orr
    from tmat_amd import inv_depth
    w = inv_depth.synth_resnet_weights(3, "conv3_block1_out")
    ens = inv_depth.InvDepthEnsemble(plain, [w], size=64)
    rs = np.random.RandomState(2)
    x = (rs.uniform(0, 255, (3, 64, 64, 1)) - np.array([103.939, 116.779, 123.68])).astype(np.float32)
    got = ens.predict(x)
    ref = orr.forward(w, x)
    assert got.dtype == np.float32 and np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (got, ref)
    assert 0.01 < got.min() and got.max() < 0.99


def test_full_model_and_stack_pipeline(plain):
    """the configured model (conv4_block6_out, 256 x 256) on a small stack: data preparation equals the oracle bit for bit,
    probabilities of two ensemble members equal the oracle, the ensemble vote is the reference's rounding rule"""
    from oracle import resnet as orr
    from tmat_amd import inv_depth, synth
    ws = [inv_depth.synth_resnet_weights(s) for s in (0, 1)]
    ens = inv_depth.InvDepthEnsemble(plain, ws)
    stack = synth.synth_stack(9, 3, 300, 360, n_vessels=8)
    probs, x = ens.predict_stack(stack, return_input=True)
    ox = orr.prep_inv_depth_imgs(stack, 256)
    assert np.array_equal(x.view(np.uint32), ox.view(np.uint32))
    assert probs.shape == (3, 2)
    for m in range(2):
        ref = orr.forward(ws[m], ox)
        assert np.array_equal(probs[:, m].view(np.uint32), ref.view(np.uint32)), (m, probs[:, m], ref)
    assert inv_depth.ensemble_predictions(probs) == orr.ensemble(probs)
    assert len({round(float(p), 3) for p in probs.ravel()}) > 2            # the synthetic models do not saturate


def test_exact_halving_of_a_512_slice_takes_cv2s_area_path(plain):
    """a 512 x 512 slice at the configured 256 x 256 is an exact halving on both axes: cv2.resize then replaces INTER_LINEAR by
    INTER_AREA's integer mean (imgproc/src/resize.cpp), which oracle/cellarea.py:resize_linear_u16 restates (hand-derived vectors in
    tests/test_gpu_cellarea.py); the preparation of the invasion-depth tool must follow (tools/bench_config5.py's stacks are 512 x 512)"""
    from oracle import resnet as orr
    from tmat_amd import inv_depth, synth
    ens = inv_depth.InvDepthEnsemble(plain, [inv_depth.synth_resnet_weights(0, "conv2_block1_out")])
    stack = synth.synth_stack(4, 2, 512, 512, n_vessels=8)
    probs, x = ens.predict_stack(stack, return_input=True)
    ox = orr.prep_inv_depth_imgs(stack, 256)
    a = stack[0].astype(np.uint32)
    small = ((a[0::2, 0::2] + a[0::2, 1::2] + a[1::2, 0::2] + a[1::2, 1::2] + 2) >> 2).astype(np.float64)
    g = (small - small.min()) / (small.max() - small.min()) * 255.0
    assert np.array_equal(ox[0, :, :, 0], (g - 103.939).astype(np.float32))
    assert np.array_equal(x.view(np.uint32), ox.view(np.uint32))


def test_eight_bit_stacks_take_cv2s_fixed_point_bilinear(plain):
    """uint8 slices (data_prep.py:36: cv2.resize on the slice as loaded): the fixed-point bilinear path; the prepared input and the
    probabilities equal the oracle's, and differ from what the same pixels give as uint16"""
    from oracle import resnet as orr
    from tmat_amd import inv_depth, synth
    ens = inv_depth.InvDepthEnsemble(plain, [inv_depth.synth_resnet_weights(0, "conv2_block1_out")])
    stack16 = synth.synth_stack(5, 2, 300, 360, n_vessels=8)
    stack8 = (stack16 >> 8).astype(np.uint8)
    probs, x = ens.predict_stack(stack8, return_input=True)
    ox = orr.prep_inv_depth_imgs(stack8, 256)
    assert np.array_equal(x.view(np.uint32), ox.view(np.uint32))
    _, x16 = ens.predict_stack(stack8.astype(np.uint16), return_input=True)
    assert np.array_equal(x16.view(np.uint32), orr.prep_inv_depth_imgs(stack8.astype(np.uint16), 256).view(np.uint32))
    assert (x != x16).any()


def test_stacks_ride_together_without_changing_a_slice(plain):
    """predict_stacks: stacks of one shape and dtype go through the classifiers in one call (tmat_inv_depth_predict_multi: uploaded back
    to back, no concatenation on the host), stacks of another shape or dtype start a new call; every slice's probabilities equal what
    predict_stack gives for its stack alone, bit for bit"""
    from tmat_amd import inv_depth, synth
    ens = inv_depth.InvDepthEnsemble(plain, [inv_depth.synth_resnet_weights(s, "conv2_block1_out") for s in (0, 1)])
    a = synth.synth_stack(1, 3, 300, 360, n_vessels=8)
    b = synth.synth_stack(2, 2, 300, 360, n_vessels=8)
    c = synth.synth_stack(3, 2, 280, 300, n_vessels=8)
    d = (synth.synth_stack(4, 2, 280, 300, n_vessels=8) >> 8).astype(np.uint8)
    stacks = [a, b, a[:1], c, d, d]
    got = ens.predict_stacks(stacks)
    assert len(got) == len(stacks)
    for s, g in zip(stacks, got):
        ref = ens.predict_stack(s)
        assert g.shape == ref.shape == (len(s), 2) and np.array_equal(g.view(np.uint32), ref.view(np.uint32))
    # a group limited by max_slices splits where the next stack would not fit
    got2 = ens.predict_stacks(stacks, max_slices=4)
    for g, g2 in zip(got, got2):
        assert np.array_equal(g.view(np.uint32), g2.view(np.uint32))
    with pytest.raises(ValueError):
        ens.predict_stacks([np.zeros((2, 8, 8), np.float32)])


def test_bad_weights_and_arguments(plain):
    from tmat_amd import _lib, inv_depth
    w = inv_depth.synth_resnet_weights(0, "conv2_block1_out")
    bad = dict(w); del bad["fc.b"]
    with pytest.raises(_lib.TmatError):
        inv_depth.InvDepthEnsemble(plain, [bad])
    bad = dict(w); bad["s2b1.c2.w"] = np.zeros((3, 3, 32, 64), np.float32)
    with pytest.raises(_lib.TmatError):
        inv_depth.InvDepthEnsemble(plain, [bad])
    ens = inv_depth.InvDepthEnsemble(plain, [w], size=64)
    with pytest.raises(_lib.TmatError):
        ens.predict(np.zeros((1, 50, 50, 3), np.float32))                   # size must be a multiple of 32
    with pytest.raises(ValueError):
        ens.predict_stack(np.zeros((2, 8, 8), np.float32))


def test_script_end_to_end(tmp_path):
    """the drop-in CLI on two stacks stored as slice sequences, synthetic ensemble members (TMAT_SYNTHETIC_WEIGHTS=1): the CSV
    equals what the oracle computes for the same three best models"""
    import csv
    import os
    import subprocess
    import sys
    from pathlib import Path
    from PIL import Image
    from oracle import resnet as orr
    from tmat_amd import inv_depth, synth
    repo = Path(__file__).resolve().parents[1]
    ind, outd = tmp_path / "in", tmp_path / "out"
    ind.mkdir()
    stacks = {"gelA": synth.synth_stack(11, 3, 128, 160, n_vessels=6), "gelB": synth.synth_stack(12, 2, 128, 160, n_vessels=6)}
    for k, st in stacks.items():
        for z, sl in enumerate(st):
            Image.fromarray(sl).save(ind / f"{k}_z{z}.tif")
    script = repo / "tissue-model-analysis-tools_amd" / "scripts" / "compute_inv_depth.py"
    r = subprocess.run([sys.executable, str(script), str(ind), str(outd)], capture_output=True, text=True, timeout=600, env=dict(os.environ, TMAT_SYNTHETIC_WEIGHTS="1"))
    assert r.returncode == 0, r.stdout + r.stderr
    rows = list(csv.reader(open(outd / "invasion_depth_predictions.csv")))
    assert rows[0] == ["Z Slice ID", "Invasion Probability", "Invasion Prediction (0=no 1=yes)"]
    order = inv_depth.best_model_indices(repo / "tissue-model-analysis-tools_amd" / "model_training" / "best_ensemble", 5, 3)
    ws = [inv_depth.synth_resnet_weights(i) for i in order]
    want = {}
    for k, st in stacks.items():
        ox = orr.prep_inv_depth_imgs(st, 256)
        probs = np.stack([orr.forward(w, ox) for w in ws], axis=1)
        for z, (p, lab) in enumerate(orr.ensemble(probs)):
            want[f"{k}_z{z}"] = (str(np.float32(p)), str(lab))
    assert {r_[0]: (r_[1], r_[2]) for r_ in rows[1:]} == want
    r = subprocess.run([sys.executable, str(script), str(ind), str(outd)], capture_output=True, text=True, timeout=600)      # no weights, no opt-in
    assert r.returncode == 1 and "convert_keras_h5" in r.stdout
