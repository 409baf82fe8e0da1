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
