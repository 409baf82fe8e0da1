#!/usr/bin/env python3
"""Compute Z projections of a directory of Z stacks -- MI355X drop-in for the reference's scripts/compute_zproj.py:
same positional arguments and flags, same output naming (`<stack id>_<method><ext>`, -2, -3 ... when the name is
taken), same exit behaviour (message + exit code 1).

    python compute_zproj.py IN_ROOT OUT_ROOT [-m {min,max,med,avg,fs}] [--channel N] [--time N]

IN_ROOT holds either multi-page image files (one stack per file) or slice images with a `z<number>` token in their
names (directly, or one folder per stack).  The projections run in HIP kernels (csrc/zproj_kernels.hip) through
tmat_zproj_batch; stacks of equal shape are projected together.  Files are read with Pillow through tmat_amd.helper.load_image
(the reference's helper.load_image contract: --time / --channel select the T / C plane of OME / ImageJ hyperstacks and must be
given for time series / multi-channel files).  -a/--area runs the cell-area drop-in on the projections afterwards, with OUT_ROOT
as its input and output directory, as the reference does (compute_zproj.py:98-119); the options that describe the input stacks
(-m/--method, --time, --channel) are not forwarded to it (the reference forwards everything: its cell-area parser then rejects -m,
and a forwarded --time / --channel fails on the single-plane projection files).
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


def load_stack(path_or_paths, channel=None, time=None) -> np.ndarray:
    """(Z, H, W) uint8 / uint16 stack from one multi-page file or from a list of slice files (reference helper.load_image)"""
    from tmat_amd import helper
    st, _ = helper.load_image(list(path_or_paths) if isinstance(path_or_paths, (list, tuple)) else path_or_paths, time, channel)
    st = np.asarray(st)
    if st.ndim == 2:
        st = st[None]
    if st.ndim != 3:
        raise ValueError(f"expected a Z stack of 2-D slices, got shape {st.shape}")
    if st.dtype not in (np.uint8, np.uint16):
        raise ValueError(f"expected uint8/uint16 pixels, got {st.dtype}")
    return st


def save_projection(path, img):
    from PIL import Image
    if img.dtype == np.float64:           # avg / med: the reference hands float64 to cv2.imwrite; written as float32 TIFF here
        img = img.astype(np.float32)
    Image.fromarray(img).save(path)


def main(args=None):
    args_prespecified = args is not None and not isinstance(args, list)
    if args is None or isinstance(args, list):
        argv = sys.argv[1:] if args is None else list(args)
        args = parse_zproj_args(argv)
    else:
        argv = []
    compute_area_after_zproj = bool(getattr(args, "area", False))
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
            st = load_stack(zs_path, args.channel, getattr(args, "time", None))
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

    if compute_area_after_zproj:
        # compute_zproj.py:98-119: the cell-area tool on the projections, OUT_ROOT as both its input and its output directory
        import subprocess
        options, skip = [], False
        for arg in argv:
            if skip:
                skip = False
                continue
            if arg in ("-a", "--area", args.in_root, args.out_root):
                continue
            # options that describe the INPUT stacks: the projections written above are single-plane files
            if arg in ("-m", "--method", "--time", "--channel"):
                skip = True
                continue
            if arg.startswith(("--method=", "--time=", "--channel=")) or (arg.startswith("-m") and len(arg) > 2 and not arg.startswith("--")):
                continue
            options.append(arg)
        script_path = Path(__file__).resolve().parent / "compute_cell_area.py"
        subprocess.run([sys.executable, str(script_path), *options, str(out_root), str(out_root)], check=True)


if __name__ == "__main__":
    main()
