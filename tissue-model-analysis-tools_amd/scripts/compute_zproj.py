#!/usr/bin/env python3
"""Compute Z projections of a directory of Z stacks -- MI355X drop-in for the reference's scripts/compute_zproj.py:
same positional arguments and flags, same output naming (`<stack id>_<method><ext>`, -2, -3 ... when the name is
taken), same exit behaviour (message + exit code 1).

    python compute_zproj.py IN_ROOT OUT_ROOT [-m {min,max,med,avg,fs}] [--channel N] [--time N]

IN_ROOT holds either multi-page image files (one stack per file) or slice images with a `z<number>` token in their
names (directly, or one folder per stack).  The projections run in HIP kernels (csrc/zproj_kernels.hip) through
tmat_zproj_batch; stacks of equal shape are projected together.  Differences: files are read with
Pillow (TIFF / PNG; single channel unless --channel picks one of an interleaved image); time-series files (--time)
and -a/--area (cell-area analysis after the projection) are outside the accelerated path and are refused.
"""
import argparse
import os
import sys
from glob import glob
from pathlib import Path

PKG = Path(__file__).resolve().parents[1]
if str(PKG) not in sys.path:
    sys.path.insert(0, str(PKG))

import numpy as np  # noqa: E402

FAIL = "\033[91m[FAILURE]\033[0m"
OK = "\033[92m[SUCCESS]\033[0m"
WARN = "\033[93m[WARNING]\033[0m"


def parse_zproj_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("in_root", type=str, help="Full path to root directory of input zstacks.")
    p.add_argument("out_root", type=str, help="Full path to root directory where output will be stored.")
    p.add_argument("--channel", type=int, default=None, help="Index of color channel (starting from 0) to read from images.")
    p.add_argument("--time", type=int, default=None, help="Index of time (starting from 0) to read from images.")
    p.add_argument("-m", "--method", type=str, default="max", choices=["min", "max", "med", "avg", "fs"],
                   help="Z projection method (min, max, med, avg, fs = focus stacking). Defaults to 'max'.")
    p.add_argument("-a", "--area", action="store_true", help="Compute cell area after computing Z projection.")
    args = p.parse_args(argv)
    for k, v in vars(args).items():
        if isinstance(v, str):
            setattr(args, k, v.strip("'\""))
    return args


def get_unique_output_filepath(file):
    file = Path(file)
    name, ext = os.path.splitext(file.name)
    n = 1
    while file.exists():
        n += 1
        file = file.parent / f"{name}-{n}{ext}"
    return file


def _pages(path):
    from PIL import Image
    with Image.open(path) as im:
        n = getattr(im, "n_frames", 1)
        out = []
        for i in range(n):
            im.seek(i)
            out.append(np.array(im))
    return out


def n_pages(path) -> int:
    from PIL import Image
    with Image.open(path) as im:
        return getattr(im, "n_frames", 1)


def load_stack(path_or_paths, channel=None) -> np.ndarray:
    """(Z, H, W) uint8 / uint16 stack from one multi-page file or from a list of slice files"""
    if isinstance(path_or_paths, (list, tuple)):
        pages = [pg for p in path_or_paths for pg in _pages(p)]
    else:
        pages = _pages(path_or_paths)
    sl = []
    for a in pages:
        if a.ndim == 3:
            if channel is None:
                raise ValueError("multi-channel image but no --channel was specified")
            a = a[..., channel] if a.shape[-1] <= 4 else a[channel]
        if a.ndim != 2:
            raise ValueError(f"expected 2-D slices, got shape {a.shape}")
        sl.append(a)
    st = np.stack(sl)
    if st.dtype not in (np.uint8, np.uint16):
        raise ValueError(f"expected uint8/uint16 pixels, got {st.dtype}")
    return st


def save_projection(path, img):
    from PIL import Image
    if img.dtype == np.float64:           # avg / med: the reference hands float64 to cv2.imwrite; written as float32 TIFF here
        img = img.astype(np.float32)
    Image.fromarray(img).save(path)


def main(args=None):
    if args is None:
        args = parse_zproj_args()
    if getattr(args, "area", False):
        print(f"{FAIL} -a/--area (cell area after the projection) is not part of the accelerated path.", flush=True)
        sys.exit(1)
    if getattr(args, "time", None) is not None:
        print(f"{FAIL} --time: time-series files are not part of the accelerated path.", flush=True)
        sys.exit(1)
    in_root = args.in_root
    if not os.path.isdir(in_root):
        print(f"{FAIL} Input data directory not found:{os.linesep}\t{in_root}", flush=True)
        sys.exit(1)
    entries = glob(os.path.join(in_root, "*"))
    files = [e for e in entries if os.path.isfile(e)]
    dirs = [e for e in entries if os.path.isdir(e)]
    if not files and not dirs:
        print(f"{FAIL} Input directory is empty: {in_root}", flush=True)
        sys.exit(1)
    if files and dirs:
        print(f"{FAIL} Input directory contains both files and subfolders: {in_root}", flush=True)
        sys.exit(1)

    from tmat_amd import zstacks as zs
    try:
        test_path = entries[0]
        if os.path.isdir(test_path) or n_pages(test_path) == 1:
            zstack_paths = zs.find_zstack_image_sequences(in_root)
        else:
            zstack_paths = zs.find_zstack_files(in_root)
    except zs.ZStackInputException as exc:
        print(f"{FAIL} {exc}", flush=True)
        sys.exit(1)

    out_root = args.out_root
    if not os.path.isdir(out_root):
        if os.path.isfile(out_root):
            print(f"{FAIL} Output path is a file: {out_root}", flush=True)
            sys.exit(1)
        try:
            os.makedirs(out_root, exist_ok=True)
        except PermissionError as error:
            print(f"{FAIL} {error}", flush=True)
            sys.exit(1)
    elif glob(os.path.join(out_root, "*")):
        print(f"{WARN}Output directory is not empty:{os.linesep}\t{out_root}", flush=True)

    print("Loading and computing Z stacks...", flush=True)
    handle = zs.default_handle()
    # project stacks of equal shape together (one launch per group), in bounded chunks: at most CHUNK_BYTES of stacks are
    # held in host memory / HBM at a time (the reference streams one stack at a time: compute_zproj.py:73-84)
    CHUNK_BYTES = 2 << 30

    def flush(groups):
        for members in groups.values():
            proj = handle.zproj(np.stack([m[2] for m in members]), args.method)
            for (zs_id, zs_path, _), img in zip(members, proj):
                out_ext = Path(np.atleast_1d(zs_path)[0]).suffix.lower()
                if out_ext not in (".tif", ".tiff", ".png"):
                    out_ext = ".tiff"
                if img.dtype == np.float64 and out_ext == ".png":
                    out_ext = ".tiff"
                save_path = get_unique_output_filepath(os.path.join(out_root, f"{zs_id}_{args.method}{out_ext}"))
                Path(save_path).parent.mkdir(parents=True, exist_ok=True)     # ids keep a folder part when stacks sit one per folder
                save_projection(save_path, img)
                print(f"Z projection saved to {save_path}", flush=True)

    groups, held = {}, 0
    for zs_id, zs_path in zstack_paths.items():
        print(f"Processing {zs_id}...", flush=True)
        try:
            st = load_stack(zs_path, args.channel)
        except (OSError, ValueError) as error:
            print(f"{FAIL}{error}", flush=True)
            sys.exit(1)
        groups.setdefault((st.shape, st.dtype.str), []).append((zs_id, zs_path, st))
        held += st.nbytes
        if held >= CHUNK_BYTES:
            flush(groups)
            groups, held = {}, 0
    flush(groups)
    print("... Projections saved.", flush=True)
    print(OK, flush=True)


if __name__ == "__main__":
    main()
