#!/usr/bin/env python3
"""Compute the cell-covered area of a directory of images -- MI355X drop-in for the reference's scripts/compute_cell_area.py:
same positional arguments, flags and config keys, same outputs (`thresholded/<id>_thresholded.png`,
`calculations/cell_area.csv` with columns image_id, area_pct; -2, -3 ... when a name is taken), same exit behaviour.

    python compute_cell_area.py IN_ROOT OUT_ROOT [--channel N] [--time N] [--sd-coef F] [-c CONFIG] [-w [--well-seed N]]

Z stacks (slice sequences or multi-page files) are max-projected first, as in the reference.  Differences
(INTEGRATION.md): images are thresholded in batches on the GPU (tmat_cell_area_batch); the gaussian-mixture fit is
deterministic, so `rs_seed` has no effect; -w/--detect-well takes an explicit --well-seed (the reference's superellipse search is
unseeded); --time / --channel follow the reference's helper.load_image; files are read and written with Pillow.
"""
import argparse
import csv
import json
import os
import sys
from glob import glob
from pathlib import Path

PKG = Path(__file__).resolve().parents[1]
if str(PKG) not in sys.path:
    sys.path.insert(0, str(PKG))

import numpy as np  # noqa: E402

DEFAULT_CONFIG_PATH = str(PKG / "config" / "default_cell_area_computation.json")
THRESH_SUBDIR = "thresholded"
CALC_SUBDIR = "calculations"
FAIL = "\033[91m[FAILURE]\033[0m"
OK = "\033[92m[SUCCESS]\033[0m"
WARN = "\033[93m[WARNING]\033[0m"


def parse_cell_area_args(argv=None):
    """same surface as the reference's script_util.parse_cell_area_args (script_util.py:208-298)"""
    p = argparse.ArgumentParser()
    p.add_argument("in_root", type=str)
    p.add_argument("out_root", type=str)
    p.add_argument("--channel", type=int, default=None)
    p.add_argument("--time", type=int, default=None)
    p.add_argument("-w", "--detect-well", action="store_true")
    p.add_argument("--well-seed", type=int, default=0,
                   help="seed of the random superellipse search of --detect-well (unseeded in the reference, well_mask_generation.py:35)")
    p.add_argument("--sd-coef", type=float, default=None)
    p.add_argument("-c", "--config", type=str, default=DEFAULT_CONFIG_PATH)
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


def main(argv=None):
    args = parse_cell_area_args(argv)
    from compute_branches import find_inputs, load_image_2d, load_stack            # the same discovery rule (script_util.py:506-553)
    in_root = Path(args.in_root)
    if not in_root.is_dir():
        print(f"{FAIL} Input directory {in_root} does not exist.", flush=True)
        sys.exit(1)
    if not glob(str(in_root / "*")):
        print(f"{FAIL}No images found in {in_root}", flush=True)
        sys.exit(1)
    paths, is_stack = find_inputs(in_root)
    if not paths:
        print(f"{FAIL}No images found in {in_root}", flush=True)
        sys.exit(1)
    out_root = Path(args.out_root)
    if out_root.is_file():
        print(f"{FAIL} Output path is a file: {out_root}", flush=True)
        sys.exit(1)
    try:
        (out_root / THRESH_SUBDIR).mkdir(parents=True, exist_ok=True)
        (out_root / CALC_SUBDIR).mkdir(parents=True, exist_ok=True)
    except PermissionError as error:
        print(f"{FAIL} {error}", flush=True)
        sys.exit(1)
    if not os.path.isfile(args.config):
        print(f"{FAIL} Config file not found: {args.config}", flush=True)
        sys.exit(1)
    with open(args.config, "r", encoding="utf8") as fp:
        config = json.load(fp)
    dsamp_size = config["dsamp_size"]
    sd_coef = config["sd_coef"] if args.sd_coef is None else args.sd_coef
    batch_size = int(config["batch_size"])
    if config.get("rs_seed") not in (None, 0):
        # the reference seeds sklearn's KMeans start with it (preprocessing.py:62-66); the fit here starts from the optimal 2-means split
        print(f"{WARN} rs_seed = {config['rs_seed']} has no effect: the mixture fit of the accelerated path is deterministic "
              "(it agrees with scikit-learn's to 0.1 percentage points of the area for any seed).", flush=True)

    from PIL import Image
    from tmat_amd import _lib, distributed, preprocessing, zstacks
    # one process per GPU under torch.distributed.run: images are independent, every rank takes a contiguous block of them,
    # rank 0 writes the CSV from the gathered rows; every rank writes the thresholded pictures of its own images
    ws, rank, local_rank = distributed.init_process_group_from_env()
    handle = _lib.Handle(None, local_rank)
    if is_stack:
        print(f"{WARN} Input images are Z stacks. Creating maximum intensity Z projections prior to cell area calculation.", flush=True)
    img_ids = sorted(paths)                     # a deterministic order: the ranks must agree on it (the reference keeps glob order)
    mine = [img_ids[int(i)] for i in distributed.shard_indices(len(img_ids), rank, ws)]
    areas, kept_all, wells = {}, {}, {}

    def flush(group):
        for (shape, dtype), items in group.items():          # one dtype per batch: np.stack would silently widen a mixed group
            batch = np.stack([im for _, im in items])
            if args.detect_well:
                area, kept, well = preprocessing.cell_area_batch_well(handle, batch, dsamp_size, sd_coef, args.well_seed)
                for (img_id, _), wm_ in zip(items, well):
                    wells[img_id] = wm_
            else:
                area, kept = preprocessing.cell_area_batch(handle, batch, dsamp_size, sd_coef)
            for (img_id, _), a, k in zip(items, area, kept):
                areas[img_id], kept_all[img_id] = a, k

    failed = False
    try:
        for i0 in range(0, len(mine), batch_size):
            group = {}
            for img_id in mine[i0:i0 + batch_size]:
                try:
                    if is_stack:
                        img = zstacks.proj_max(load_stack(paths[img_id], args.channel, args.time), handle=handle)     # compute_cell_area.py:50-52
                    else:
                        img = load_image_2d(paths[img_id], args.channel, args.time)
                except (OSError, ValueError) as error:
                    print(f"{FAIL}{error}", flush=True)
                    failed = True               # still enter the gather below: the other ranks are waiting in it
                    break
                group.setdefault((img.shape, img.dtype.str), []).append((img_id, img))
            if failed:
                break
            flush(group)
    except Exception:                           # noqa: BLE001 -- a library / HIP error in flush must not leave the other ranks in the gather
        import traceback
        traceback.print_exc()
        print(f"{FAIL}rank {rank}: the shard failed", flush=True)
        failed = True
    index_of = {img_id: i for i, img_id in enumerate(img_ids)}
    try:
        gathered = distributed.gather_rows([] if failed else [(index_of[i], 0, areas[i], 0.0) for i in mine], n_total=len(img_ids), failed=failed)
    except distributed.RankFailed:
        handle.close()
        distributed.finish_process_group()
        sys.exit(1)
    print("... Areas computed successfully.", flush=True)
    print(OK, flush=True)

    out_ids = [i.replace("/", "_").replace("\\", "_") for i in img_ids]
    for img_id in mine:
        oid = out_ids[img_ids.index(img_id)]
        if args.detect_well:                                   # compute_cell_area.py:301-306
            Image.fromarray(wells[img_id]).save(get_unique_output_filepath(out_root / THRESH_SUBDIR / f"{oid}_well_mask.png"))
        file = get_unique_output_filepath(out_root / THRESH_SUBDIR / f"{oid}_thresholded.png")
        Image.fromarray(kept_all[img_id]).save(file)
    areas = {img_ids[g[0]]: g[2] for g in gathered}
    if rank != 0:
        handle.close()
        distributed.finish_process_group()
        return
    if args.detect_well:
        print(f"... Well masks saved to:{os.linesep}\t{out_root}/{THRESH_SUBDIR}", flush=True)
    print(f"... Thresholded images saved to:{os.linesep}\t{out_root}/{THRESH_SUBDIR}", flush=True)
    area_out_path = get_unique_output_filepath(out_root / CALC_SUBDIR / "cell_area.csv")
    with open(area_out_path, "w", newline="") as f:           # pandas DataFrame.to_csv(index=False): header + repr of the floats
        wr = csv.writer(f, lineterminator="\n")
        wr.writerow(["image_id", "area_pct"])
        for img_id, oid in zip(img_ids, out_ids):
            wr.writerow([oid, repr(float(areas[img_id] * 100))])
    print(f"... Area calculations saved to:{os.linesep}\t{area_out_path}", flush=True)
    print(OK, flush=True)
    handle.close()
    distributed.finish_process_group()


if __name__ == "__main__":
    main()
