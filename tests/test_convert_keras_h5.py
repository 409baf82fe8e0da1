"""CPU: tools/convert_keras_h5.py (the stand-in for models.py:622 `load_weights(checkpoint_N.h5)`).  The reference's
checkpoint is absent (.MISSING_LARGE_BLOBS), so a Keras-2.14-layout HDF5 file is written with h5py from known weights
(tests/helpers/write_keras_h5.py), converted, and the TMATW001 blob is compared tensor by tensor.  h5py only exists under
/opt/conda/bin/python3.9 in the build image: both steps run there; the test is skipped where that interpreter is missing."""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parents[1]
PY = Path("/opt/conda/bin/python3.9")


def _has_h5py():
    return PY.exists() and subprocess.run([str(PY), "-c", "import h5py"], capture_output=True).returncode == 0


@pytest.mark.skipif(not _has_h5py(), reason="no interpreter with h5py")
@pytest.mark.parametrize("nested", [False, True])
def test_h5_roundtrip(tmp_path, nested):
    from tmat_amd import synth
    rs = np.random.RandomState(3)
    w = synth.synth_weights(2)
    for k in w:                                         # every tensor distinct and non-trivial, including the BN statistics
        w[k] = (w[k] + rs.normal(0, 0.1, w[k].shape)).astype(np.float32)
    src = tmp_path / "in.tmatw"
    src.write_bytes(synth.pack_weights(w))
    h5, dst = tmp_path / "checkpoint_1.h5", tmp_path / "out.tmatw"
    r = subprocess.run([str(PY), str(REPO / "tests" / "helpers" / "write_keras_h5.py"), str(src), str(h5)] + (["--nested"] if nested else []),
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(PY), str(REPO / "tools" / "convert_keras_h5.py"), str(h5), str(dst)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = synth.unpack_weights(dst.read_bytes())
    assert list(got) == list(w)
    for k in w:
        assert got[k].shape == w[k].shape and np.array_equal(got[k], w[k]), k
    assert dst.read_bytes() == src.read_bytes()         # the container itself is reproduced byte for byte


@pytest.mark.skipif(not _has_h5py(), reason="no interpreter with h5py")
def test_resnet_h5_roundtrip(tmp_path):
    """the invasion-depth classifier's weight file (build_ResNet50_TL + save_weights layout) -> TMATW001, tensor by tensor"""
    from tmat_amd import inv_depth, synth
    w = inv_depth.synth_resnet_weights(4, "conv3_block2_out")
    src = tmp_path / "in.tmatw"
    src.write_bytes(inv_depth.pack_resnet(w))
    h5, dst = tmp_path / "best_finetune_weights_0.h5", tmp_path / "out.tmatw"
    r = subprocess.run([str(PY), str(REPO / "tests" / "helpers" / "write_keras_resnet_h5.py"), str(src), str(h5)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(PY), str(REPO / "tools" / "convert_keras_h5.py"), "--resnet", str(h5), str(dst)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = synth.unpack_weights(dst.read_bytes())
    assert list(got) == list(w)
    for k in w:
        assert got[k].shape == w[k].shape and np.array_equal(got[k], w[k]), k
