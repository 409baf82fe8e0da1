"""GPU parity: the device part of compute_dmt_graph (csrc/dmt_kernels.hip: edge keys + stable lower-star radix sort;
csrc/dmt_sweep_kernels.hip: the two persistence sweeps as levels of data-parallel steps; collect on the host) through
tmat_dmt_graph(handle, ...) and tmat_dmt_graph_batch, against the goldens that
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


def test_batches_run_the_sweeps_of_all_fields_in_one_launch(plain):
    """tmat_dmt_graph_batch: 8 and 16 fields per launch (8 = the pass size of the pipeline: the workgroups of one image then run on
    one XCD), 3 (they are spread over the XCDs) -- each field's graph equals the one the host-only execution gives, and the two
    golden fields equal the reference's output."""
    from tmat_amd import _lib
    fields = [GD["field_d5"].astype(np.float32), GD["field_m1"].astype(np.float32)] + [synth_field(500 + k, (384, 384)) for k in range(14)]
    fields[5] = np.zeros((384, 384), np.float32)                          # an empty image inside the batch
    want = [_lib.dmt_graph(f, 5.0, 10.0) for f in fields]
    for n in (8, 16, 3):
        got = _lib.dmt_graph_batch(np.stack(fields[:n]), 5.0, 10.0, handle=plain)
        for k in range(n):
            assert np.array_equal(got[k][0], want[k][0]) and np.array_equal(got[k][1], want[k][1]), (n, k)
    for k, name in enumerate(("d5", "m1")):
        assert np.array_equal(want[k][0], GD[f"{name}_5.0_10.0_V"].reshape(-1, 2)) and np.array_equal(want[k][1], GD[f"{name}_5.0_10.0_E"].reshape(-1, 2))


def test_many_levels_and_ties_in_batches(plain):
    """Fields built against the level formulation: a staircase whose minima get older to the left while its saddles get lower to the
    right (the sequential sweep merges right to left, one link per level: > 100 levels), and 8 random fields of six distinct values
    (long runs of equal keys, dropped edges) -- against the oracle."""
    from oracle import dmt as odmt
    from tmat_amd import _lib
    N = 260
    c = np.arange(N)
    row = np.where(c % 2 == 0, 1000.0 - c, 1.0 + c).astype(np.float32)     # img = -val: minima of val at even columns, saddles at odd ones
    stair = np.stack([row, row + 0.25, row + 0.5])
    for d in DELTAS:
        V0, E0 = odmt.compute_dmt_graph(stair, *d)
        V1, E1 = _lib.dmt_graph(stair, *d, handle=plain)
        assert np.array_equal(V0, V1) and np.array_equal(E0, E1)
    rs = np.random.RandomState(77)
    f = np.round(rs.uniform(0, 5, (8, 70, 83))).astype(np.float32) * 50
    f[rs.uniform(size=f.shape) < 0.2] = 0.0
    f[3] = np.stack([np.resize(row, 83)] * 70) + np.arange(70, dtype=np.float32)[:, None] * 0.125
    for d in DELTAS:
        got = _lib.dmt_graph_batch(f, *d, handle=plain)
        for k in range(8):
            V0, E0 = odmt.compute_dmt_graph(f[k], *d)
            assert np.array_equal(V0, got[k][0]) and np.array_equal(E0, got[k][1]), (d, k)
    # more fields than one launch of the sweep kernel takes (32): the call splits them
    f = np.round(rs.uniform(0, 9, (70, 21, 34))).astype(np.float32) * 25
    f[rs.uniform(size=f.shape) < 0.15] = 0.0
    got = _lib.dmt_graph_batch(f, 2.0, 4.0, handle=plain)
    for k in range(70):
        V0, E0 = odmt.compute_dmt_graph(f[k], 2.0, 4.0)
        assert np.array_equal(V0, got[k][0]) and np.array_equal(E0, got[k][1]), k


def test_device_sweeps_match_reference_goldens_and_timing():
    """The two persistence sweeps (dmtgraph.py:277-314) on the device as levels of data-parallel steps (csrc/dmt_sweep_kernels.hip,
    the default) and on host threads (TMAT_DMT_SWEEP_DEVICE=0), `collect` on the host either way: all 27 golden cases exact in
    both settings.  Runs in child processes (the switch is read when a handle is created) and prints the time of the 384 x 384
    case for both."""
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
