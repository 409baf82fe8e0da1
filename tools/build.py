#!/usr/bin/env python3
"""Build libtmat_hip.so (gfx950) in-tree and the oracle's C restatement.

    python tools/build.py            # incremental
    python tools/build.py --force

hipcc cross-compiles for gfx950 without a GPU; the .so is git-ignored but travels with gpurun.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

REPO = Path(__file__).resolve().parents[1]
PKG = REPO / "tissue-model-analysis-tools_amd"
CSRC = PKG / "csrc"
OUT = PKG / "tmat_amd" / "libtmat_hip.so"
OBJ = CSRC / "build"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result", "-pthread"]


def _newer(target: Path, deps) -> bool:
    return target.exists() and all(target.stat().st_mtime >= d.stat().st_mtime for d in deps)


def build_hip(force=False, verbose=True) -> Path:
    OBJ.mkdir(exist_ok=True)
    headers = sorted(CSRC.glob("*.h")) + [REPO / "include" / "tmat.h"]
    srcs = sorted(CSRC.glob("*.hip")) + sorted(CSRC.glob("*.cpp"))
    jobs = []
    objs = []
    for stale in OBJ.glob("*.o"):                 # objects of sources that no longer exist (tools/build_variant.sh links build/*.o)
        if stale.name[:-2] not in {s.name for s in srcs}:
            stale.unlink()
    for s in srcs:
        o = OBJ / (s.name + ".o")
        objs.append(o)
        if force or not _newer(o, [s] + headers):
            cmd = [HIPCC, "-x", "hip", "-c", str(s), "-o", str(o)] + FLAGS
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd[:6]), "...", flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or not OUT.exists():
        cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", str(OUT)] + [str(o) for o in objs] + ["-pthread", "-ldl"]
        run(cmd)
    return OUT


def build_oracle(force=False) -> Path:
    sys.path.insert(0, str(REPO))
    from oracle import unet as ou
    return ou.build(force)


if __name__ == "__main__":
    force = "--force" in sys.argv
    print(build_hip(force))
    print(build_oracle(force))
