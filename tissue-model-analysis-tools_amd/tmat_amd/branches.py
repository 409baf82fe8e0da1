"""Batched analyze_img (reference scripts/compute_branches.py:144-489, 2-D branch) over the C-ABI."""
from __future__ import annotations

import ctypes as C
from itertools import product

import numpy as np

from . import _lib

DOWNSAMPLE_WIDTH = 384


def pixels_to_microns(num_pixels: float, im_width_px: int, im_width_microns: float) -> float:
    return (im_width_microns / im_width_px) * num_pixels


def microns_to_pixels(num_microns: float, im_width_px: int, im_width_microns: float) -> float:
    return (im_width_px / im_width_microns) * num_microns


def graph_px_params(config: dict, field_width: int, image_width_microns: float):
    """compute_branches.py:401-415"""
    min_px = round(microns_to_pixels(config.get("min_branch_length", 12), field_width, image_width_microns))
    mx = config.get("max_branch_length")
    max_px = None if mx is None else round(max(1, microns_to_pixels(mx, field_width, image_width_microns)))
    sw_px = round(max(1, microns_to_pixels(config.get("graph_smoothing_window", 12), field_width, image_width_microns)))
    return sw_px, min_px, max_px


def threshold_grid(config: dict):
    """compute_branches.py:366-395: Cartesian grid over graph_thresh_1 x graph_thresh_2 with the file-name suffix."""
    params = {"thresh1": np.atleast_1d(config.get("graph_thresh_1", 5)).tolist(),
              "thresh2": np.atleast_1d(config.get("graph_thresh_2", 10)).tolist()}
    names, vals = zip(*params.items())
    cfgs = [dict(zip(names, comb)) for comb in product(*vals)]
    tuned = [k for k, v in params.items() if len(v) > 1]
    fmts = {}
    for k, v in params.items():
        if all(isinstance(x, (int, float)) for x in v):
            if all(isinstance(x, int) for x in v):
                fmts[k] = f"{{:0{max(len(str(x)) for x in v)}d}}"
            else:
                wl = max(str(float(x)).find(".") for x in v)
                wr = max(len(str(float(x)).split(".")[1]) for x in v)
                fmts[k] = f"{{:0{wl + 1 + wr}.{wr}f}}"
        else:
            fmts[k] = "{}"
    out = []
    for cfg in cfgs:
        s = "".join(f"_{k}_{fmts[k].format(v)}" for k, v in cfg.items() if k in tuned)
        out.append((cfg, f"_CONFIG{s}" if s else ""))
    return out


def analyze_batch(handle: _lib.Handle, imgs: np.ndarray, config: dict, image_width_microns: float, ds_ratio: float = 0.625,
                  thresh=(5.0, 10.0), first_index: int = 0, dev_ptr=None, input_bits: int = 16):
    """imgs (n, H, W) uint16 (host) or a device pointer + shape -> list of (index, count, total_px, avg_px).
    `input_bits` = 8 for images that were uint8 before widening (cv2.resize saturates to the source depth)."""
    if dev_ptr is None:
        imgs = np.ascontiguousarray(imgs, np.uint16)
        n, H, W = imgs.shape
    else:
        n, H, W = imgs       # shape tuple
    sw_px, min_px, max_px = graph_px_params(config, DOWNSAMPLE_WIDTH, image_width_microns)
    rows = (_lib.Row * n)()
    L = _lib.lib()
    _lib.check(L.tmat_set_input_depth(handle.raw, int(input_bits)), "tmat_set_input_depth")
    args = (n, H, W, float(ds_ratio), DOWNSAMPLE_WIDTH, float(thresh[0]), float(thresh[1]), int(sw_px), int(min_px),
            int(max_px or 0), int(bool(config.get("remove_isolated_branches", False))), int(first_index), rows)
    if dev_ptr is None:
        _lib.check(L.tmat_analyze_batch(handle.raw, _lib.ptr(imgs), *args), "tmat_analyze_batch")
    else:
        _lib.check(L.tmat_analyze_batch_dev(handle.raw, C.c_void_p(dev_ptr), *args), "tmat_analyze_batch_dev")
    return [(r.index, r.count, r.total_px, r.avg_px) for r in rows]


def dsamp_shape(img_shape, width: int = DOWNSAMPLE_WIDTH):
    """compute_branches.py:218-222: img_dsamp_res = round(shape * width / W)"""
    r = width / img_shape[1]
    return tuple(int(v) for v in np.round(np.multiply(img_shape[:2], r)).astype(int))


def well_fields(handle: _lib.Handle, imgs: np.ndarray, ds_ratio: float = 0.625, input_bits: int = 16, well_seed: int = 0, warn=print):
    """The 2-D branch of analyze_img with use_well_mask=True up to the vesselness field (compute_branches.py:309-361), image by
    image through the staged GPU entry points: Lanczos + rescale (tmat_preprocess_batch) -> make_well_mask on THAT image
    (:318-319, tmat_amd/well_mask_generation.py) -> predict(img * well_mask) (:328) -> (pred > 0.5) * well_mask ->
    filter_branch_seg_mask (:334-337) -> medial axis, centre-line weighting of the unmasked prediction, resize (:340-357).
    Returns [(field255 (fh, fw) f32, pruning_mask (fh, fw) bool, well_mask)] per image: the graph stages follow per threshold."""
    from . import well_mask_generation as wmg
    imgs = np.ascontiguousarray(imgs, np.uint16)
    n, H, W = imgs.shape
    hh, ww = int(round(W * ds_ratio)), int(round(H * ds_ratio))            # cv2 reads dsize as (width, height)
    L = _lib.lib()
    x = np.empty((n, hh, ww), np.float32)
    _lib.check(L.tmat_set_input_depth(handle.raw, int(input_bits)), "tmat_set_input_depth")
    _lib.check(L.tmat_preprocess_batch(handle.raw, _lib.ptr(imgs), n, H, W, float(ds_ratio), _lib.ptr(x)), "tmat_preprocess_batch")
    _lib.check(L.tmat_set_input_depth(handle.raw, 16), "tmat_set_input_depth")
    masks = [wmg.make_well_mask(x[i], handle=handle, seed=well_seed, warn=warn) for i in range(n)]
    well = np.stack([m[0] for m in masks])
    pred = handle.predict_smooth(x * well)                                  # img * well_mask: float32 * bool
    filt = handle.filter_mask((pred > 0.5) & well)                          # seg_mask * well_mask, footprint disk(2), remove_isolated
    skel, dist = handle.medial_axis(filt)
    # the reference resizes to img_dsamp_res computed from the ORIGINAL image shape (:218-222)
    fshape = dsamp_shape((H, W))
    _, f255 = handle.finish(pred, dist, skel, fshape)
    out = []
    for i in range(n):
        pruning = wmg._resize_nearest(np.logical_not(masks[i][1]), fshape).astype(bool)      # resize(order=0) (:359-361)
        out.append((f255[i], pruning, well[i]))
    return out


def well_rows(handle: _lib.Handle, fields, config: dict, image_width_microns: float, thresh=(5.0, 10.0), first_index: int = 0):
    """graph stages of the --detect-well form (compute_branches.py:391-457): DMT graph of the 0..255 field, MorseGraph with the
    pruning mask -> rows (index, count, total_px, avg_px)"""
    rows = []
    for i, (f255, pruning, _) in enumerate(fields):
        sw_px, min_px, max_px = graph_px_params(config, f255.shape[1], image_width_microns)
        V, E = _lib.dmt_graph(f255, thresh[0], thresh[1], handle=handle)
        _, cnt, tot, avg = _lib.morse_stats(V, E, f255.shape, sw_px, min_px, max_px, bool(config.get("remove_isolated_branches", False)),
                                            pruning)
        rows.append((first_index + i, cnt, tot, avg))
    return rows


class InputError(Exception):
    """an image of the run cannot be loaded / lacks its physical width: raised by run_sharded's callbacks (after they have
    printed the reference's message); the run then fails on EVERY rank after the gather instead of leaving the others blocked"""


def run_sharded(ids, load_fn, width_fn, analyze_fn, config: dict, rank: int = 0, world_size: int = 1, chunk: int = 64,
                log=print):
    """The per-run driver of scripts/compute_branches.py (reference :585-594 loops over the images one by one):
    rank `rank` of `world_size` takes a contiguous block of `ids`, loads its images in bounded chunks of at most `chunk`
    (images of equal shape, physical width and bit depth are analysed as one batch), and all ranks exchange their rows
    with one all-gather per threshold configuration.

    load_fn(img_id) -> uint8/uint16 (H, W) array; width_fn(img_id, img) -> image width in microns;
    analyze_fn(batch uint16 (n, H, W), width_um, thresh=(t1, t2), input_bits=8|16) -> [(i, count, total_px, avg_px)].
    Returns {file-name suffix: [(global index, count, total_um, avg_um)] sorted by index}, on every rank."""
    from . import distributed
    ids = list(ids)
    grid = threshold_grid(config)
    results = {suffix: [] for _, suffix in grid}
    mine = distributed.shard_indices(len(ids), rank, world_size)

    def flush(groups):
        for (shape, width_um, bits), items in groups.items():
            batch = np.stack([im for _, im in items]).astype(np.uint16)
            for cfg, suffix in grid:
                rows = analyze_fn(batch, width_um, thresh=(cfg["thresh1"], cfg["thresh2"]), input_bits=bits)
                for (gidx, _), r in zip(items, rows):
                    results[suffix].append((gidx, r[1], pixels_to_microns(r[2], DOWNSAMPLE_WIDTH, width_um),
                                            pixels_to_microns(r[3], DOWNSAMPLE_WIDTH, width_um)))

    # Whatever goes wrong on this rank -- an unreadable image (InputError), a HIP / library error out of analyze_fn, a shape
    # surprise in np.stack -- the rank must still enter the collective below: the other ranks are waiting in it and would block
    # until the launcher kills them.  The failure travels through the gather (distributed.gather_rows raises RankFailed everywhere).
    groups, held, failed = {}, 0, False
    try:
        for gidx in mine:
            img_id = ids[int(gidx)]
            log(f"Analyzing {img_id}...")
            try:
                img = load_fn(img_id)
                width_um = width_fn(img_id, img)
            except InputError:
                failed = True
                break
            groups.setdefault((img.shape, float(width_um), 8 * img.dtype.itemsize), []).append((int(gidx), img))
            held += 1
            if held >= chunk:           # bounded host / HBM footprint: the reference streams one image at a time
                flush(groups)
                groups, held = {}, 0
        if not failed:
            flush(groups)
    except Exception:                   # noqa: BLE001 -- reported here, re-raised as RankFailed by the gather on every rank
        import traceback
        traceback.print_exc()
        log(f"rank {rank}: the shard failed (traceback above); entering the gather with the failure marker")
        failed = True
    return {suffix: distributed.gather_rows(results[suffix], n_total=len(ids), failed=failed) for _, suffix in grid}


def save_visualizations(handle: _lib.Handle, img: np.ndarray, vis_dir, ds_ratio: float = 0.625, input_bits: int = 16):
    """The four image dumps of the reference's 2-D branch (compute_branches.py:74-78 save_vis = rescale_intensity to
    0..255 + cv2.imwrite; :315 original_image.png, :331 prediction.png, :347 segmentation_mask.png, :348
    distance_transform.png) for one image, through the staged entry points of the same GPU path (segment ->
    filter + EDT -> medial axis); the centre-line weighting of :341-344 is evaluated here with the scipy call the
    reference makes.  The matplotlib barcode / tree plots (:431-450) are not reproduced.  Returns the written paths."""
    import os
    from pathlib import Path
    from PIL import Image
    from scipy.ndimage import distance_transform_edt

    vis_dir = Path(vis_dir)
    vis_dir.mkdir(parents=True, exist_ok=True)
    img = np.ascontiguousarray(img, np.uint16)
    H, W = img.shape
    hh, ww = int(round(W * ds_ratio)), int(round(H * ds_ratio))            # cv2 reads dsize as (width, height)
    L = _lib.lib()
    pred = np.empty((1, hh, ww), np.float64)
    _lib.check(L.tmat_set_input_depth(handle.raw, int(input_bits)), "tmat_set_input_depth")
    _lib.check(L.tmat_segment_batch(handle.raw, _lib.ptr(img[None]), 1, H, W, float(ds_ratio), _lib.ptr(pred)), "tmat_segment_batch")
    _lib.check(L.tmat_set_input_depth(handle.raw, 16), "tmat_set_input_depth")
    filt, dist = handle.filter_edt(pred)
    skel, _ = _lib.host_medial_axis(filt[0])
    cdt = distance_transform_edt(np.logical_not(skel))
    with np.errstate(invalid="ignore", divide="ignore"):
        weighted = pred[0] * (dist[0] / (dist[0] + cdt))
    original = _lib.host_lanczos4_u16(img, (hh, ww))        # uint8 sources: float path, a 1-LSB-level difference in a picture

    def save_vis(a, name):
        a = np.asarray(a, np.float64)
        lo, hi = np.nanmin(a), np.nanmax(a)
        a = np.clip(a, lo, hi)
        a = (a - lo) / (hi - lo) * 255.0 if hi != lo else np.clip(a, 0, 255)
        file = vis_dir / name
        stem, ext = os.path.splitext(file.name)
        n = 1
        while file.exists():                                # helper.get_unique_output_filepath
            n += 1
            file = vis_dir / f"{stem}-{n}{ext}"
        Image.fromarray(np.rint(np.nan_to_num(a)).astype(np.uint8)).save(file)
        return str(file)
    return [save_vis(original, "original_image.png"), save_vis(pred[0], "prediction.png"),
            save_vis(filt[0].astype(np.float64), "segmentation_mask.png"), save_vis(weighted, "distance_transform.png")]


def save_stack_visualizations(handle: _lib.Handle, stack: np.ndarray, vis_dir, hessian: str = "gaussian_derivatives"):
    """The two image dumps of the reference's Z-stack branch (compute_branches.py:228-229 original_image.png = the max
    projection, :303 vesselness_image.png) for one stack, through the same GPU path.  Returns the written paths."""
    import os
    from pathlib import Path
    from PIL import Image
    from . import sato

    vis_dir = Path(vis_dir)
    vis_dir.mkdir(parents=True, exist_ok=True)
    field = sato.stack_field(handle, stack, DOWNSAMPLE_WIDTH, hessian)

    def save_vis(a, name):
        a = np.asarray(a, np.float64)
        lo, hi = a.min(), a.max()
        a = (a - lo) / (hi - lo) * 255.0 if hi != lo else np.clip(a, 0, 255)
        file = vis_dir / name
        stem, ext = os.path.splitext(file.name)
        n = 1
        while file.exists():                                # helper.get_unique_output_filepath
            n += 1
            file = vis_dir / f"{stem}-{n}{ext}"
        Image.fromarray(np.rint(a).astype(np.uint8)).save(file)
        return str(file)
    return [save_vis(np.asarray(stack).max(0), "original_image.png"), save_vis(field, "vesselness_image.png")]
