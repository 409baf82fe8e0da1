#!/usr/bin/env python3
"""Analyze microvessels in a directory of 2-D Z-projections or of Z stacks -- MI355X drop-in for the reference's
scripts/compute_branches.py: same positional arguments, flags, config keys, CSV (utf-16, same header), config.json and
exit behaviour (message + exit code 1).  Projections go through the segmentation model (the 2-D branch,
compute_branches.py:307-361); Z stacks (slice sequences with a z<number> token, or multi-page files) through the Sato
branch (:224-306), both on the GPU.

    python compute_branches.py IN_ROOT OUT_ROOT [--image-width-microns F] [--graph-thresh-1 F ...]
        [--graph-thresh-2 F ...] [--min-branch-length F] [--max-branch-length F]
        [--remove-isolated-branches] [--graph-smoothing-window F] [-c CONFIG] [--channel N] [--time N] [-w]

Differences (documented in INTEGRATION.md): images are analysed in batches on the GPU (one process
per GPU under torch.distributed.run; rows are gathered over RCCL and rank 0 writes the CSV);
--detect-well takes an explicit --well-seed (the reference's random search is unseeded); the PNG image dumps are opt-in (--visualizations; the
matplotlib barcode / tree plots are not reproduced); --sato-hessian picks the Hessian of skimage.filters.sato
(gaussian_derivatives = scikit-image >= 0.20, what the reference's pinned 0.22.0 runs; gradient = <= 0.19);
without --image-width-microns (or the config key) the width comes from OME / ImageJ TIFF metadata
(tmat_amd/helper.py), as in the reference.
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

DEFAULT_CONFIG_PATH = str(PKG / "config" / "default_branching_computation.json")
DOWNSAMPLE_WIDTH = 384
FAIL = "\033[91m[FAILURE]\033[0m"
OK = "\033[92m[SUCCESS]\033[0m"


def parse_branching_args(arg_defaults):
    """same surface as the reference's script_util.parse_branching_args (script_util.py:40-204)"""
    p = argparse.ArgumentParser()
    p.add_argument("in_root", type=str)
    p.add_argument("out_root", type=str)
    p.add_argument("--channel", type=int, default=None)
    p.add_argument("--time", type=int, default=None)
    p.add_argument("-w", "--detect-well", action="store_true")
    p.add_argument("--well-seed", type=int, default=0,
                   help="seed of the random superellipse search of --detect-well (the reference draws from numpy's global, unseeded "
                        "generator: well_mask_generation.py:35; a run here equals a reference run after numpy.random.seed(SEED))")
    p.add_argument("--image-width-microns", type=float, default=None)
    p.add_argument("--graph-thresh-1", nargs="+", type=float, default=None)
    p.add_argument("--graph-thresh-2", nargs="+", type=float, default=None)
    p.add_argument("--min-branch-length", type=float, default=None)
    p.add_argument("--max-branch-length", type=float, default=None)
    p.add_argument("--remove-isolated-branches", action="store_true")
    p.add_argument("--graph-smoothing-window", type=float, default=None)
    p.add_argument("-c", "--config", type=str, default=arg_defaults["default_config_path"])
    p.add_argument("--visualizations", action="store_true",
                   help="also write visualizations/<image>/{original_image,prediction,segmentation_mask,distance_transform}.png "
                        "(the reference always does; here it is opt-in: it re-runs the image through the staged entry points)")
    p.add_argument("--sato-hessian", choices=["gaussian_derivatives", "gradient"], default="gaussian_derivatives",
                   help="Z stacks: Hessian of skimage.filters.sato -- gaussian_derivatives (scikit-image >= 0.20, the reference's pinned "
                        "0.22.0) or gradient (scikit-image <= 0.19)")
    args = p.parse_args()
    if not args.remove_isolated_branches:
        args.remove_isolated_branches = None
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


def create_output_csv(output_file: Path):
    fields = ["Image", "Total # of branches", "Total branch length (µm)", "Average branch length (µm)"]
    with open(output_file, "w", encoding="utf-16") as f:
        csv.writer(f, lineterminator="\n").writerow(fields)


def load_image_2d(path: str, channel=None, time=None) -> np.ndarray:
    """One 2-D plane of an image file (reference helper.load_image :23-95 returns ZYX / YX for the chosen T and C; a time
    series needs --time, a multi-channel file --channel, as there).  .npy arrays: (H, W), or (C, H, W) / (H, W, C) with C <= 4."""
    from tmat_amd import helper
    if path.endswith(".npy"):
        a = np.load(path)
        if time not in (None, 0):
            raise ValueError(f"Time {time} is out of range for {path} with times: 0 - 0")
        if a.ndim == 3:
            cax = [ax for ax in (2, 0) if a.shape[ax] <= 4]
            if not cax:
                raise ValueError(f"{path}: cannot tell the channel axis of shape {a.shape}")
            if channel is None:
                if a.shape[cax[0]] != 1:
                    raise ValueError(f"{path} is a multi channel image but no color channel index was specified.")
                channel = 0
            if not 0 <= channel < a.shape[cax[0]]:
                raise ValueError(f"Color channel {channel} is out of range for {path} with color channels: 0 - {a.shape[cax[0]] - 1}")
            a = np.take(a, channel, axis=cax[0])
    else:
        a, _ = helper.load_image(path, time, channel)
        if a.ndim == 3:
            raise ValueError(f"{path}: multi-page file (Z stack); project it first with compute_zproj.py")
    if a.ndim != 2:
        raise ValueError(f"{path}: expected a single-channel 2-D image, got shape {a.shape}")
    if a.dtype not in (np.uint8, np.uint16):
        raise ValueError(f"{path}: expected uint8/uint16 pixels, got {a.dtype}")
    return a        # uint8 images are widened at the ABI and flagged (input_bits=8): cv2.resize saturates to the source depth


def n_planes(path: str) -> int:
    """helper.get_image_dims(path).Z of the reference for the formats read here: pages of a TIFF, leading axis of a 3-D .npy"""
    if path.endswith(".npy"):
        a = np.load(path, mmap_mode="r")
        return a.shape[0] if a.ndim == 3 and a.shape[0] > 4 and a.shape[-1] > 4 else 1
    from PIL import Image
    from tmat_amd import helper
    try:
        with Image.open(path) as im:
            n = getattr(im, "n_frames", 1)
            desc = dict(getattr(im, "tag_v2", {}) or {}).get(270, "")
    except OSError:
        return 1
    if isinstance(desc, (tuple, list)):
        desc = desc[0] if desc else ""
    if isinstance(desc, bytes):
        desc = desc.decode("utf8", "replace")
    return helper.page_layout(desc, n)[1]            # SizeZ: pages of a time series or of channels are not Z slices


def find_inputs(in_root: Path):
    """compute_branches.py:547-566 -> (paths, is_stack): {stack id: [slice files] | multi-page file} or {stem: 2-D image file}"""
    from tmat_amd import zstacks as zs
    entries = sorted(glob(str(in_root / "*")))
    test_path = entries[0]
    if os.path.isdir(test_path) or n_planes(test_path) == 1:
        try:
            img_paths = zs.find_zstack_image_sequences(str(in_root))
            if any(len(seq) == 1 for seq in img_paths.values()):
                img_paths = {}          # not z stacks. probably projections.
        except zs.ZStackInputException:
            img_paths = {}
    else:
        try:
            img_paths = zs.find_zstack_files(str(in_root))
        except zs.ZStackInputException as exc:
            print(f"{FAIL} {exc}", flush=True)
            sys.exit(1)
    if img_paths:
        return img_paths, True
    return {Path(fp).stem: fp for fp in entries if os.path.isfile(fp) and n_planes(fp) == 1}, False


def load_stack(files, channel=None, time=None) -> np.ndarray:
    """(Z, H, W) uint8 / uint16 stack of one Z-stack entry (reference helper.load_image with a list of slice files or one
    multi-page file)"""
    from tmat_amd import helper
    if isinstance(files, str) and files.endswith(".npy"):
        a = np.load(files)
    else:
        a, _ = helper.load_image(files, time, channel)
    if a.ndim != 3:
        raise ValueError(f"{np.atleast_1d(files)[0]}: expected a Z stack, got shape {a.shape}")
    if a.dtype not in (np.uint8, np.uint16):
        raise ValueError(f"{np.atleast_1d(files)[0]}: expected uint8/uint16 pixels, got {a.dtype}")
    if a.shape[0] < 2:
        raise ValueError(f"{np.atleast_1d(files)[0]}: a Z stack needs at least 2 slices")
    return a


def write_results(args, config, ids, gathered, out_root: Path, rank: int):
    """the CSV per threshold configuration and config.json (compute_branches.py:459-500, 596-600), written by rank 0"""
    from tmat_amd import branches
    created = set()
    for _, suffix in branches.threshold_grid(config):
        rows = gathered[suffix]
        if rank != 0:
            continue
        output_file = out_root / f"branching_analysis{suffix}.csv"
        n = 1
        while output_file.is_file() and str(output_file) not in created:
            n += 1
            output_file = out_root / f"branching_analysis{suffix}-{n}.csv"
        create_output_csv(output_file)
        created.add(str(output_file))
        with open(output_file, "a", encoding="utf-16") as f:
            wr = csv.writer(f, lineterminator="\n")
            for gidx, cnt, tot_um, avg_um in rows:
                wr.writerow([ids[gidx], cnt, tot_um, avg_um])
        print(f"Results saved to {output_file}.", flush=True)
    if rank == 0:
        config["time"], config["channel"] = getattr(args, "time", None), getattr(args, "channel", None)
        with open(get_unique_output_filepath(out_root / "config.json"), "w", encoding="utf8") as f:
            json.dump({k: v for k, v in config.items() if v is not None}, f, indent=4)
        print(f"{OK} Analysis complete.", flush=True)


def finish_distributed(ws: int):
    if ws > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def run_stacks(args, config, paths, out_root: Path, rank: int, ws: int, local_rank: int):
    """Z-stack entries: the Sato branch (compute_branches.py:224-306) on this rank's share of the stacks"""
    from tmat_amd import _lib, branches, distributed, helper, sato
    handle = _lib.Handle(None, local_rank)          # this branch needs no segmentation model
    hessian = getattr(args, "sato_hessian", "gaussian_derivatives")
    vis = bool(getattr(args, "visualizations", False))
    detect_well = bool(getattr(args, "detect_well", False))
    well_seed = int(getattr(args, "well_seed", 0) or 0)
    ids = sorted(paths)
    fields = {}

    def load_fn(img_id):
        try:
            return load_stack(paths[img_id], args.channel, args.time)
        except (OSError, ValueError) as error:
            print(f"{FAIL}{error}", flush=True)
            raise branches.InputError(str(error))

    def width_fn(img_id, st):
        width_um = config.get("image_width_microns")
        if width_um is None:
            px = helper.physical_pixel_sizes(np.atleast_1d(paths[img_id])[0]).X
            if px is None:
                print(f"{FAIL} The --image-width-microns parameter was not specified, and the pixel to micron conversion "
                      f"factor was not found in the image metadata ({img_id}). Specify --image-width-microns and try again. "
                      "Exiting...", flush=True)
                raise branches.InputError(img_id)
            width_um = st.shape[-1] * px
        return width_um

    def analyze_fn(batch, width_um, thresh, input_bits):
        # the vesselness image does not depend on the graph thresholds: one field per stack, swept over the grid
        sw_px, min_px, max_px = branches.graph_px_params(config, DOWNSAMPLE_WIDTH, width_um)
        rows = []
        if fields.get("batch") is not batch:            # run_sharded hands the same array to every configuration of the grid
            fields.clear()
            fields["batch"] = batch
        for i, st in enumerate(batch):
            if i not in fields:
                field = sato.stack_field(handle, st, DOWNSAMPLE_WIDTH, hessian)
                # --detect-well (compute_branches.py:231-243): the well mask of the resized max projection only prunes the graph here
                pruning = sato.stack_well_masks(handle, st, field.shape, well_seed)[1] if detect_well else None
                fields[i] = (field, pruning)
            rows.append((i,) + sato.field_stats(handle, fields[i][0], thresh[0], thresh[1], sw_px, min_px, max_px,
                                                bool(config.get("remove_isolated_branches", False)), pruning_mask=fields[i][1]))
        return rows

    def load_and_keep(img_id):
        st = load_fn(img_id)
        if vis:
            branches.save_stack_visualizations(handle, st, out_root / "visualizations" / img_id, hessian)
            if detect_well:
                from PIL import Image
                well = sato.stack_well_masks(handle, st, sato.dsamp_shape(st.shape, DOWNSAMPLE_WIDTH), well_seed)[0]
                Image.fromarray((well * 255).astype(np.uint8)).save(out_root / "visualizations" / img_id / "well_mask.png")
        return st

    # one stack per analysis call (chunk=1): a stack is the unit the reference streams, and it can be gigabytes
    try:
        gathered = branches.run_sharded(ids, load_and_keep, width_fn, analyze_fn, config, rank, ws, chunk=1, log=lambda m: print(m, flush=True))
    except distributed.RankFailed:
        handle.close()
        finish_distributed(ws)
        sys.exit(1)
    write_results(args, config, ids, gathered, out_root, rank)
    handle.close()
    finish_distributed(ws)


def main(args=None):
    if args is None:
        args = parse_branching_args({"default_config_path": DEFAULT_CONFIG_PATH})
        if not Path(args.config).is_file():
            print(f"{FAIL} Config file {args.config} does not exist.", flush=True)
            sys.exit(1)
        with open(args.config, "r", encoding="utf8") as fp:
            config = json.load(fp)
    else:
        config = {}
    ad = vars(args)
    for prm in ("image_width_microns", "graph_thresh_1", "graph_thresh_2", "graph_smoothing_window", "min_branch_length",
                "max_branch_length", "remove_isolated_branches"):
        if prm not in config or ad.get(prm) is not None:
            config[prm] = ad.get(prm)
    model_cfg_path = config.get("model_cfg_path")
    if not model_cfg_path:
        cfgs = sorted(glob(str(PKG / "model_training" / "binary_segmentation" / "configs" / "unet_patch_segmentor_*.json")),
                      key=lambda s: int(Path(s).stem.rsplit("_", 1)[1]))
        model_cfg_path = cfgs[-1] if cfgs else ""
    if not Path(model_cfg_path).is_file():
        print(f"{FAIL}Model config file {model_cfg_path} does not exist.", flush=True)
        sys.exit(1)
    in_root, out_root = Path(args.in_root), Path(args.out_root)
    if not in_root.is_dir():
        print(f"{FAIL} Input directory {in_root} does not exist.", flush=True)
        sys.exit(1)
    try:
        out_root.mkdir(parents=True, exist_ok=True)
    except PermissionError as e:
        print(f"{FAIL} {e}", flush=True)
        sys.exit(1)
    if not glob(str(in_root / "*")):
        print(f"{FAIL}No images found in {in_root}", flush=True)
        sys.exit(1)
    paths, is_stack = find_inputs(in_root)
    if not paths:
        print(f"{FAIL}No images found in {in_root}", flush=True)
        sys.exit(1)
    # image_width_microns: the option / config key, else per image from the file's metadata (reference
    # compute_branches.py:184-212: img.shape[-1] * PhysicalPixelSizes.X), else the reference's failure message
    from tmat_amd import branches, distributed, helper, models
    ws = distributed.world()[0]
    # host stages (thinning, DMT, MorseGraph) run on worker threads of every rank: share the cores between the ranks
    os.environ.setdefault("TMAT_HOST_THREADS", str(distributed.host_threads_per_rank(ws)))
    ws, rank, local_rank = distributed.init_process_group_from_env()
    if is_stack:
        run_stacks(args, config, paths, out_root, rank, ws, local_rank)
        return
    model = models.get_unet_patch_segmentor_from_cfg(model_cfg_path, device_id=local_rank)
    # models.py:636-637 normalises the image in predict(): the batched device path does it in front of the smooth prediction
    model.handle.set_input_norm(model.norm_mean, model.norm_std)
    detect_well = bool(getattr(args, "detect_well", False))
    well_seed = int(getattr(args, "well_seed", 0) or 0)

    # ids are file stems, as in the reference (compute_branches.py:565-569: two files with one stem are one entry there too)
    ids = sorted(paths)

    def load_fn(img_id):
        try:
            return load_image_2d(paths[img_id], args.channel, args.time)
        except (OSError, ValueError) as error:
            print(f"{FAIL}{error}", flush=True)
            raise branches.InputError(str(error))

    def width_fn(img_id, img):
        # image_width_microns: the option / config key, else per image from the file's metadata (reference
        # compute_branches.py:184-212: img.shape[-1] * PhysicalPixelSizes.X), else the reference's failure message
        width_um = config.get("image_width_microns")
        if width_um is None:
            px = helper.physical_pixel_sizes(paths[img_id]).X
            if px is None:
                print(f"{FAIL} The --image-width-microns parameter was not specified, and the pixel to micron conversion "
                      f"factor was not found in the image metadata ({img_id}). Specify --image-width-microns and try again. "
                      "Exiting...", flush=True)
                raise branches.InputError(img_id)
            width_um = img.shape[-1] * px
        return width_um

    well_cache = {}

    def analyze_fn(batch, width_um, thresh, input_bits):
        if not detect_well:
            return branches.analyze_batch(model.handle, batch, config, width_um, model.ds_ratio, thresh=thresh, input_bits=input_bits)
        # --detect-well (compute_branches.py:318-337): the fields do not depend on the graph thresholds -- one staged pass per
        # batch (run_sharded hands the same array to every configuration of the grid), then the graph stages per configuration
        if well_cache.get("batch") is not batch:
            well_cache.clear()
            well_cache["batch"] = batch
            well_cache["fields"] = branches.well_fields(model.handle, batch, model.ds_ratio, input_bits, well_seed,
                                                        warn=lambda m: print(f"\033[93m[WARNING]\033[0m {m}", flush=True))
        return branches.well_rows(model.handle, well_cache["fields"], config, width_um, thresh)

    vis = bool(getattr(args, "visualizations", False))

    def load_and_keep(img_id):
        img = load_fn(img_id)
        if vis:
            branches.save_visualizations(model.handle, img, out_root / "visualizations" / img_id, model.ds_ratio, 8 * img.dtype.itemsize)
        return img

    try:
        gathered = branches.run_sharded(ids, load_and_keep, width_fn, analyze_fn, config, rank, ws,
                                        log=lambda m: print(m, flush=True))
    except distributed.RankFailed:          # the failing rank has printed the reference's message; every rank exits with code 1
        model.handle.close()
        finish_distributed(ws)
        sys.exit(1)
    write_results(args, config, ids, gathered, out_root, rank)
    model.handle.close()
    finish_distributed(ws)


if __name__ == "__main__":
    main()
