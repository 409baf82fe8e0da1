"""GPU parity: the device front end of compute_dmt_graph (csrc/dmt_kernels.hip: edge keys + stable lower-star radix sort
on the device, sweeps + collect on the host) through tmat_dmt_graph(handle, ...), against the goldens that
tools/make_goldens.py produced by running the reference's dmtgraph.compute_dmt_graph -- all 27 cases, exact vertices and
edges -- and against the oracle on random fields with heavy ties (ties are where sort stability is the result)."""
from pathlib import Path

import numpy as np
import pytest

from make_goldens import DELTAS, DMT_SYNTH, synth_field

pytestmark = pytest.mark.gpu
GD = np.load(Path(__file__).parent / "golden" / "dmt.npz")


def all_fields():
    f = {n: synth_field(seed, shape) for n, seed, shape in DMT_SYNTH}
    for n in ("d5", "m1", "ties"):
        f[n] = GD["field_" + n].astype(np.float32)
    f["zero"] = np.zeros((24, 24), np.float32)
    f["const"] = np.full((24, 30), 7.0, np.float32)
    one = np.zeros((16, 16), np.float32); one[5, 9] = 200.0
    f["single"] = one
    return f


FIELDS = all_fields()


@pytest.fixture(scope="module")
def plain():
    from tmat_amd import _lib
    h = _lib.Handle(None, 0)
    yield h
    h.close()


@pytest.mark.parametrize("name", sorted(FIELDS))
def test_device_front_end_matches_reference_goldens(plain, name):
    from tmat_amd import _lib
    for deltas in DELTAS:
        V, E = _lib.dmt_graph(FIELDS[name], *deltas, handle=plain)
        key = f"{name}_{deltas[0]}_{deltas[1]}"
        assert np.array_equal(V, GD[key + "_V"].reshape(-1, 2)), key
        assert np.array_equal(E, GD[key + "_E"].reshape(-1, 2)), key


@pytest.mark.parametrize("seed", range(5))
def test_device_front_end_matches_oracle_on_ties(plain, seed):
    from oracle import dmt as odmt
    from tmat_amd import _lib
    rs = np.random.RandomState(300 + seed)
    shape = (int(rs.randint(3, 90)), int(rs.randint(3, 90)))
    f = np.round(rs.uniform(0, 5, shape)).astype(np.float32) * 50          # six distinct values: long runs of equal keys
    f[rs.uniform(size=shape) < 0.2] = 0.0                                  # and dropped edges in between
    for d in DELTAS:
        V0, E0 = odmt.compute_dmt_graph(f, *d)
        V1, E1 = _lib.dmt_graph(f, *d, handle=plain)
        assert np.array_equal(V0, V1) and np.array_equal(E0, E1)


def test_full_size_field_device_equals_host(plain):
    """384 x 384 (the size the pipeline always runs, 440 833 edges): the device-sorted front end and the host-only
    execution (handle NULL) give the same graph"""
    from tmat_amd import _lib
    f = synth_field(77, (384, 384))
    V0, E0 = _lib.dmt_graph(f, 5.0, 10.0)
    V1, E1 = _lib.dmt_graph(f, 5.0, 10.0, handle=plain)
    assert len(V0) > 100 and np.array_equal(V0, V1) and np.array_equal(E0, E1)


def test_device_sweeps_match_reference_goldens_and_timing():
    """TMAT_DMT_SWEEP_DEVICE=1: the two persistence sweeps (dmtgraph.py:277-314) as a one-wave-per-image kernel
    (csrc/dmt_sweep_kernels.hip), `collect` on the host: all 27 golden cases exact.  Runs in a child process (the switch is
    read once per process) and prints the time of the 384 x 384 case next to the host sweeps' -- the kernel is NOT the
    default: it is slower than the host threads it would replace."""
    import os, subprocess, sys
    code = r'''
import sys, time
sys.path[:0] = [r"%(repo)s", r"%(repo)s/tissue-model-analysis-tools_amd", r"%(repo)s/tools", r"%(repo)s/tests"]
import numpy as np
from test_gpu_dmt import FIELDS, GD, DELTAS
from make_goldens import synth_field
from tmat_amd import _lib
h = _lib.Handle(None, 0)
for name in sorted(FIELDS):
    for deltas in DELTAS:
        V, E = _lib.dmt_graph(FIELDS[name], *deltas, handle=h)
        key = f"{name}_{deltas[0]}_{deltas[1]}"
        assert np.array_equal(V, GD[key + "_V"].reshape(-1, 2)) and np.array_equal(E, GD[key + "_E"].reshape(-1, 2)), key
f = synth_field(77, (384, 384))
_lib.dmt_graph(f, 5.0, 10.0, handle=h)
t0 = time.perf_counter()
for _ in range(3):
    V1, E1 = _lib.dmt_graph(f, 5.0, 10.0, handle=h)
print("ms_per_384_field", (time.perf_counter() - t0) / 3 * 1e3, len(V1), len(E1))
h.close()
'''
    repo = str(Path(__file__).resolve().parents[1])
    out = {}
    for dev in ("1", "0"):
        r = subprocess.run([sys.executable, "-c", code % dict(repo=repo)], env=dict(os.environ, TMAT_DMT_SWEEP_DEVICE=dev), capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0, r.stdout + r.stderr
        out[dev] = r.stdout.strip().splitlines()[-1].split()
    assert out["1"][2:] == out["0"][2:]
    print(f"\ntmat_dmt_graph on one 384 x 384 field, whole call: device sweeps {float(out['1'][1]):.1f} ms, host sweeps {float(out['0'][1]):.1f} ms")
