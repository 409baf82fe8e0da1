import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]
for p in (REPO, REPO / "tissue-model-analysis-tools_amd", REPO / "tools"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def weights():
    from tmat_amd import synth
    return synth.synth_weights(0)


@pytest.fixture(scope="session")
def handle(weights):
    from tmat_amd import synth, _lib
    h = _lib.Handle(synth.pack_weights(weights), 0, 256)
    yield h
    h.close()


def forward_torch_child(x, dtype="float32", seed=0):
    """oracle.unet.forward_torch(synth_weights(seed), x) evaluated in a CHILD process (CPU only).  The GPU test process never
    imports torch: torch's wheel carries its own ROCm runtime libraries, and once they are mapped a system librccl loaded later
    binds its HSA entry points to them and fails to initialise (seen as ncclCommInitRank -> 1, "pfn_hsa_system_get_info
    failed with 4107").  A host program picks ONE runtime; the tests do the same."""
    import subprocess
    import tempfile
    import numpy as np
    with tempfile.TemporaryDirectory() as d:
        np.save(f"{d}/x.npy", np.ascontiguousarray(x))
        code = (f"import sys; sys.path[:0] = [{str(REPO)!r}, {str(REPO / 'tissue-model-analysis-tools_amd')!r}]\n"
                "import numpy as np, torch\n"
                "from oracle import unet as ou\n"
                "from tmat_amd import synth\n"
                f"x = np.load({d + '/x.npy'!r})\n"
                f"y = ou.forward_torch(synth.synth_weights({int(seed)}), x, dtype=torch.{dtype})\n"
                f"np.save({d + '/y.npy'!r}, np.asarray(y))\n")
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900,
                           env=dict(__import__("os").environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES=""))
        assert r.returncode == 0, r.stdout + r.stderr
        return np.load(f"{d}/y.npy")
