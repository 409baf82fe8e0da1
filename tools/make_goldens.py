#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING the reference's own Python modules from
/root/reference (read-only, this container only; the reference never travels to the GPU box).

Only inputs and expected outputs are stored -- no reference source text.
Run:  python tools/make_goldens.py [blend] [dmt] [morse] [filter]
The `filter` group needs scikit-image and therefore /opt/conda/bin/python3.9 (skimage 0.18.3).
Shims for absent third-party modules (numba.njit = identity decorator, cv2 colour stub used only
by plotting code) are created in a temp dir, exactly as SURVEY.md section 8c describes.
"""
import hashlib
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[1]
GOLD = REPO / "tests" / "golden"
REF = Path("/root/reference")


def _shims():
    d = Path(tempfile.mkdtemp(prefix="tmat_shims_"))
    (d / "numba.py").write_text(
        "def njit(*a, **k):\n"
        "    if a and callable(a[0]):\n        return a[0]\n"
        "    return lambda f: f\n")
    (d / "cv2.py").write_text(
        "COLOR_HSV2BGR = 54\nINTER_LANCZOS4 = 4\nDIST_L2 = 2\nINTER_AREA = 3\n"
        "def cvtColor(a, code):\n    return a\n")
    sys.path.insert(0, str(d))
    sys.path.insert(0, str(REF))


def pred_toy(batch, verbose=0):
    b = np.asarray(batch, np.float32)
    out = (0.6 * b + 0.3 * b[:, ::-1, :] * b[:, :, ::-1] + 0.05).astype(np.float32)
    return out[..., None]


def gen_blend():
    from fl_tissue_model_tools import smooth_tiled_predictions as stp

    out = {}
    out["window320"] = stp._spline_window(320)
    out["window64"] = stp._spline_window(64)
    rs = np.random.RandomState(7)
    for name, shape, ws in (("toy_a", (100, 90), 64), ("toy_b", (64, 64), 32), ("toy_c", (75, 130), 32)):
        img = rs.uniform(0, 1, shape).astype(np.float32)
        stp.cached_2d_windows.clear()
        res = stp.predict_img_with_smooth_windowing(img, ws, 2, pred_toy)
        out[name + "_in"] = img
        out[name + "_out"] = np.asarray(res)
        out[name + "_ws"] = np.int64(ws)
    img = np.random.RandomState(11).uniform(0, 1, (640, 640)).astype(np.float32)
    stp.cached_2d_windows.clear()
    res = stp.predict_img_with_smooth_windowing(img, 320, 2, lambda b, verbose=0: np.asarray(b)[..., None])
    out["ident640_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(res).tobytes()).digest(), np.uint8)
    out["ident640_maxerr"] = np.float64(np.abs(res - img).max())
    np.savez_compressed(GOLD / "blend.npz", **out)
    print("blend.npz", {k: getattr(v, "shape", None) for k, v in out.items()})




# --------------------------------------------------------------------------------------------
# filter group: run under /opt/conda/bin/python3.9 (scikit-image 0.18.3, networkx 2.6.3)
# --------------------------------------------------------------------------------------------
def _fixture_masks():
    from scipy import ndimage as ndi
    import tifffile
    masks = {}
    for name, fn in (("d5", "D5_1_ZProj_002_mask.tif"), ("m1", "mask.tif")):
        a = tifffile.imread(str(REF / "notebooks" / "topology" / "sample_data" / fn))
        if a.ndim == 3:
            a = a[..., 0]
        yy = (np.arange(640) * a.shape[0] / 640).astype(int)
        xx = (np.arange(640) * a.shape[1] / 640).astype(int)
        masks[name] = a[yy][:, xx] > 0
    rs = np.random.RandomState(21)
    masks["blobs"] = ndi.gaussian_filter(rs.normal(size=(200, 300)), 4) > 0.03
    masks["noise"] = rs.uniform(size=(96, 128)) > 0.45
    small = np.zeros((64, 80), bool)
    small[2, 2] = True                       # single pixel
    small[5:7, 5:7] = True                   # 2x2 block
    small[10, 10:40] = True                  # thin line, no fork
    small[20:50, 20:23] = True; small[33:36, 10:60] = True   # thick cross (forks)
    yy, xx = np.mgrid[:64, :80]
    small |= ((yy - 40) ** 2 + (xx - 66) ** 2) <= 81      # disc: circularity > 0.8
    masks["small"] = small
    masks["empty"] = np.zeros((40, 40), bool)
    masks["full"] = np.ones((40, 48), bool)
    return masks


def gen_filter():
    from scipy import ndimage as ndi
    from skimage.morphology import skeletonize, medial_axis, disk
    from skimage.measure import label, regionprops
    from fl_tissue_model_tools import transforms

    out = {}
    for name, m in _fixture_masks().items():
        out[name + "_shape"] = np.array(m.shape)
        out[name + "_mask"] = np.packbits(m)
        med = ndi.median_filter(m, footprint=disk(2), mode="nearest")
        out[name + "_median"] = np.packbits(med)
        out[name + "_skel"] = np.packbits(skeletonize(med))
        lab = label(med, connectivity=2)
        props = regionprops(lab)
        out[name + "_nlabels"] = np.int64(lab.max())
        out[name + "_areas"] = np.array([p.area for p in props], np.int64)
        out[name + "_perims"] = np.array([p.perimeter for p in props], np.float64)
        # the reference's own function (median applied by hand: skimage 0.18 names the kwarg selem)
        filt = transforms.filter_branch_seg_mask(med.copy(), None)
        out[name + "_filtered"] = np.packbits(filt.astype(bool))
        filt_keep = transforms.filter_branch_seg_mask(med.copy(), None, False)
        out[name + "_filtered_keepiso"] = np.packbits(filt_keep.astype(bool))
        sk, dist = medial_axis(filt.astype(float), return_distance=True)
        out[name + "_ma_skel"] = np.packbits(sk)
        out[name + "_ma_dist_sha"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(dist).tobytes()).digest(), np.uint8)
    np.savez_compressed(GOLD / "filter.npz", **out)
    print("filter.npz", len(out), "arrays,", sum(v.nbytes for v in out.values()), "bytes")


# --------------------------------------------------------------------------------------------
# dmt / morse groups: run under /opt/conda/bin/python3.9 (numpy 1.26.4 = the reference's pin)
# --------------------------------------------------------------------------------------------
def synth_field(seed, shape):
    """deterministic ridge-like f32 field in 0..255 (regenerated by the tests from the seed)."""
    from scipy import ndimage as ndi
    rs = np.random.RandomState(seed)
    a = ndi.gaussian_filter(rs.normal(size=shape), 3.0)
    b = np.exp(-(a / a.std()) ** 2 * 6.0)                 # ridges along the zero set of a
    b *= ndi.gaussian_filter(rs.uniform(size=shape), 6.0) > 0.5
    b = ndi.gaussian_filter(b, 1.0)
    b = (b - b.min()) / (b.max() - b.min() + 1e-12) * 255.0
    return b.astype(np.float32)


def real_fields():
    """384^2 vesselness-like fields from the two binary vessel masks the reference ships
    (notebooks/topology/sample_data/*.tif), stored as float16 (exact in f32)."""
    from scipy import ndimage as ndi
    import tifffile
    out = {}
    for name, fn in (("d5", "D5_1_ZProj_002_mask.tif"), ("m1", "mask.tif")):
        a = tifffile.imread(str(REF / "notebooks" / "topology" / "sample_data" / fn))
        if a.ndim == 3:
            a = a[..., 0]
        m = (a > 0).astype(np.float64)
        f = np.divide(m.shape, (384, 384))
        m = ndi.gaussian_filter(m, (f - 1) / 2, mode="mirror")
        m = ndi.zoom(m, 384 / np.array(m.shape), order=1, mode="mirror", grid_mode=True)
        m = np.clip(ndi.gaussian_filter(m, 1.2) * 1.3, 0, 1) * 255.0
        out[name] = m.astype(np.float16)
    return out


DMT_SYNTH = [("s96", 31, (96, 96)), ("s_rect", 32, (72, 96)), ("s160", 33, (160, 160))]
DELTAS = [(5.0, 10.0), (2.0, 4.0), (0.5, 0.0)]


def gen_dmt():
    from fl_tissue_model_tools.dmtgraph import compute_dmt_graph
    out = {}
    fields = {n: synth_field(seed, shape) for n, seed, shape in DMT_SYNTH}
    for n, f in real_fields().items():
        out["field_" + n] = f
        fields[n] = f.astype(np.float32)
    fields["zero"] = np.zeros((24, 24), np.float32)
    fields["const"] = np.full((24, 30), 7.0, np.float32)
    one = np.zeros((16, 16), np.float32); one[5, 9] = 200.0
    fields["single"] = one
    q = np.round(synth_field(34, (64, 64)) / 16.0) * 16.0      # heavy ties
    fields["ties"] = q.astype(np.float32)
    out["field_ties"] = fields["ties"].astype(np.float16)
    for name, f in fields.items():
        for d1, d2 in DELTAS:
            V, E = compute_dmt_graph(f, d1, d2)
            key = f"{name}_{d1}_{d2}"
            out[key + "_V"] = V.astype(np.int32)
            out[key + "_E"] = E.astype(np.int32)
            print(key, V.shape, E.shape, flush=True)
    np.savez_compressed(GOLD / "dmt.npz", **out)
    print("dmt.npz", sum(v.nbytes for v in out.values()), "bytes raw")


MORSE_CASES = [  # (delta1, delta2, smoothing, min_len, max_len, remove_isolated, use_mask)
    (5.0, 10.0, 5, 5, None, False, False),
    (2.0, 4.0, 5, 5, None, False, False),
    (5.0, 10.0, 12, 12, None, False, False),
    (0.5, 0.0, 1, 3, None, False, False),
    (5.0, 10.0, 5, 5, None, True, False),
    (5.0, 10.0, 5, 5, 40, False, False),
    (5.0, 10.0, 5, 5, None, False, True),
    (2.0, 4.0, 8, 3, 60, True, True),
]


def prune_mask(shape):
    yy, xx = np.mgrid[: shape[0], : shape[1]]
    r = min(shape) * 0.42
    return ((yy - shape[0] / 2) ** 2 + (xx - shape[1] / 2) ** 2) > r * r


def gen_morse():
    from fl_tissue_model_tools.topology import MorseGraph
    out = {}
    fields = {n: synth_field(seed, shape) for n, seed, shape in DMT_SYNTH}
    dmt_gold = np.load(GOLD / "dmt.npz")       # real-mask fields made by gen_dmt (needs tifffile)
    for n in ("d5", "m1"):
        fields[n] = dmt_gold["field_" + n].astype(np.float32)
    fields["zero"] = np.zeros((24, 24), np.float32)
    for name, f in fields.items():
        for ci, (d1, d2, sw, mn, mx, iso, um) in enumerate(MORSE_CASES):
            pm = prune_mask(f.shape) if um else None
            mg = MorseGraph(f, thresholds=(d1, d2), min_branch_length=mn, max_branch_length=mx,
                            remove_isolated_branches=iso, smoothing_window=sw, pruning_mask=pm)
            bars = np.array(mg.barcode, np.float64).reshape(-1, 2)
            key = f"{name}_c{ci}"
            out[key + "_bars"] = bars
            out[key + "_count"] = np.int64(len(mg.barcode))
            out[key + "_total"] = np.float64(mg.get_total_branch_length())
            out[key + "_avg"] = np.float64(mg.get_average_branch_length())
            print(key, len(mg.barcode), float(mg.get_total_branch_length()), flush=True)
    np.savez_compressed(GOLD / "morse.npz", **out)



# --------------------------------------------------------------------------------------------
# zstacks group: the pure-Python / numpy parts of zstacks.py (stack discovery, id clean-up, min / max / avg / med).
# proj_focus_stacking needs cv2.GaussianBlur / cv2.Laplacian, which the shim does not provide (OpenCV is absent).
# --------------------------------------------------------------------------------------------
ZSTACK_LAYOUTS = {
    "flat_two_wells": ["A1_z1_ch0.tif", "A1_z2_ch0.tif", "A1_z10_ch0.tif", "B2_z1_ch0.tif", "B2_z2_ch0.tif", "B2_z10_ch0.tif"],
    "upper_case_z": ["img_Z00.png", "img_Z01.png", "img_Z02.png"],
    "one_folder_per_stack": ["w1/img_z0.tif", "w1/img_z1.tif", "w2/img_z0.tif", "w2/img_z1.tif"],
    "folder_repeats_name": ["exp1_w1/exp1_w1_z0.tif", "exp1_w1/exp1_w1_z1.tif", "exp1_w2/exp1_w2_z0.tif", "exp1_w2/exp1_w2_z1.tif"],
    "two_numbers": ["s_z1_t_z5.tif", "s_z2_t_z5.tif", "s_z1_t_z6.tif", "s_z2_t_z6.tif"],
    "underscores": ["_a__z1.tif", "_a__z2.tif", "b_z1_.tif", "b_z2_.tif"],
    "duplicate_numbers": ["a_z1.tif", "a_Z1.tif"],
    "mixed_token_counts": ["a_z1.tif", "a_z2_z3.tif"],
    "no_z_token": ["a.tif", "b.tif"],
}
ZSTACK_ID_LISTS = [["exp/exp_a", "exp/exp_b"], ["a/b", "a_b"], ["_x_", "__x"], ["p__q", "p_q"], ["dir/name_1", "dir/name_2"],
                   ["alpha/alpha", "beta/beta"], ["s\\t", "u/v"]]


def gen_zstacks():
    import json
    import shutil
    from fl_tissue_model_tools import zstacks as zs
    out = {"layouts": {}, "clean_ids": [], "proj": {}}
    for name, files in ZSTACK_LAYOUTS.items():
        root = Path(tempfile.mkdtemp(prefix="tmat_zs_"))
        for f in files:
            (root / f).parent.mkdir(parents=True, exist_ok=True)
            (root / f).write_bytes(b"x")
        rec = {"files": files}
        try:
            got = zs.find_zstack_image_sequences(str(root))
            rec["sequences"] = {k: [os.path.relpath(p, root).replace(os.sep, "/") for p in v] for k, v in got.items()}
        except Exception as e:                      # ZStackInputException and whatever else the reference raises
            rec["error"] = type(e).__name__
        if all("/" not in f for f in files):
            rec["files_as_stacks"] = {k: os.path.relpath(v, root) for k, v in zs.find_zstack_files(str(root)).items()}
        out["layouts"][name] = rec
        shutil.rmtree(root)
    for ids in ZSTACK_ID_LISTS:
        out["clean_ids"].append({"in": ids, "out": zs.clean_zstack_ids(list(ids))})
    rs = np.random.RandomState(5)
    for Z in (1, 2, 5, 6):
        st = rs.randint(0, 65536, (Z, 4, 5)).astype(np.uint16)
        rec = {"stack": st.tolist()}
        for m in ("min", "max", "avg", "med"):
            r = getattr(zs, "proj_" + m)(st)
            rec[m] = {"dtype": str(r.dtype), "values": r.tolist()}
        out["proj"][str(Z)] = rec
    (GOLD / "zstacks.json").write_text(json.dumps(out, indent=1))
    print("zstacks.json", {k: len(v) for k, v in out.items()})



# --------------------------------------------------------------------------------------------
# sato group: the scikit-image functions of the Z-stack branch (compute_branches.py:224-306), stage by stage, from the
# scikit-image 0.18.3 of /opt/conda/bin/python3.9 (the reference pins 0.22.0, which is not importable anywhere here).
# --------------------------------------------------------------------------------------------
def sato_inputs():
    """deterministic small inputs, regenerated by the tests from the seeds"""
    from scipy import ndimage as ndi
    rs = np.random.RandomState(41)

    def tubes(shape, n, seed):
        r = np.random.RandomState(seed)
        a = np.zeros(shape)
        yy, xx = np.mgrid[: shape[0], : shape[1]]
        for _ in range(n):
            y0, x0, th = r.uniform(0, shape[0]), r.uniform(0, shape[1]), r.uniform(0, np.pi)
            d = np.abs((yy - y0) * np.cos(th) - (xx - x0) * np.sin(th))
            a += r.uniform(0.3, 1.0) * np.exp(-(d / r.uniform(1.0, 3.5)) ** 2)
        a += 0.05 * ndi.gaussian_filter(r.normal(size=shape), 1.0)
        a -= a.min()
        return (a / a.max()).astype(np.float32)
    imgs = {"t1": tubes((96, 128), 6, 1), "t2": tubes((80, 80), 3, 2)}
    vol = np.stack([tubes((64, 80), 4, 10) * w for w in (0.4, 0.8, 1.0, 0.7, 0.3)]).astype(np.float32)
    return imgs, vol


def gen_sato():
    from skimage.feature import canny
    from skimage.filters import gaussian, sato, unsharp_mask
    from skimage.morphology import closing, dilation, disk, square
    from fl_tissue_model_tools import transforms
    imgs, vol = sato_inputs()
    out = {}
    import warnings
    warnings.simplefilter("ignore")
    for k, im in imgs.items():
        out[k + "_sato"] = sato(im, sigmas=[1, 2, 3, 4, 5, 7, 9, 11, 13, 15], black_ridges=False).astype(np.float32)
        out[k + "_sato135"] = sato(im, sigmas=[1, 3, 5], black_ridges=False).astype(np.float32)
        out[k + "_gauss"] = gaussian(im)
    vess = np.stack([sato(np.maximum(vol[z], vol[z + 1]), sigmas=[1, 2, 3], black_ridges=False) for z in range(len(vol) - 1)]).astype(np.float32)
    sharp = unsharp_mask(vess, 2, 2)
    out["vol_sharp"] = sharp
    vessels = sharp.max(0)
    edges = canny(vessels, sigma=0)
    out["vol_edges"] = np.packbits(edges)
    out["t1_edges"] = np.packbits(canny(out["t1_sato"], sigma=0))
    from skimage.morphology import medial_axis
    skel = medial_axis(edges)
    out["vol_skel"] = np.packbits(skel)
    ecc = transforms.regionprops_image(skel, "eccentricity")
    dia = transforms.regionprops_image(skel, "equivalent_diameter")
    out["vol_eccdiam"] = (ecc * dia).astype(np.float64)
    rs = np.random.RandomState(3)
    m = rs.uniform(size=(60, 70)) > 0.7
    out["m_closed"] = np.packbits(closing(m.astype(np.int64), disk(2)).astype(bool))
    out["m_dil"] = np.packbits(dilation(m, square(3)))
    np.savez_compressed(GOLD / "sato.npz", **out)
    print("sato.npz", {k: getattr(v, "shape", None) for k, v in out.items()})


def cellarea_inputs():
    """deterministic small uint16 images: vessels over a noisy background, regenerated by the tests from the seeds"""
    sys.path.insert(0, str(REPO / "tissue-model-analysis-tools_amd"))
    from tmat_amd import synth
    return {f"c{i}": synth.synth_image(50 + i, 256, n_vessels=10 + 4 * i, scale=0.6)[:192, :224] for i in range(3)}


def gen_cellarea():
    """the reference's own preprocessing.exec_threshold (scikit-learn GaussianMixture inside) on rescaled images, as
    compute_cell_area.mask_and_threshold calls it: RandomState(0), sd_coef 0 and 0.5, no well mask"""
    import types
    if "dask" not in sys.modules:
        try:
            import dask  # noqa: F401
        except ImportError:
            sys.modules["dask"] = types.ModuleType("dask")
    # defs.py reads the install-time package.cfg (base_dir), which an uninstalled checkout does not have; the one constant
    # exec_threshold takes from it is MAX_UINT8 = np.iinfo(np.uint8).max
    import importlib
    import importlib.util
    pkg = importlib.import_module("fl_tissue_model_tools")
    dm = types.ModuleType("fl_tissue_model_tools.defs")
    dm.MAX_UINT8, dm.MAX_UINT16, dm.EPSILON = np.iinfo(np.uint8).max, np.iinfo(np.uint16).max, np.finfo(np.float32).eps
    sys.modules["fl_tissue_model_tools.defs"] = dm
    pkg.defs = dm
    from fl_tissue_model_tools import preprocessing as prep
    import sklearn
    out = {"sklearn_version": np.array(sklearn.__version__)}
    for k, img in cellarea_inputs().items():
        lo, hi = float(img.min()), float(img.max())
        x = ((img.astype(np.float64) - lo) / (hi - lo)).astype(np.float32)      # rescale_intensity(img, (0, 1)).astype(float32)
        for sd in (0.0, 0.5):
            rs = np.random.RandomState(0)
            res = prep.exec_threshold(x, None, sd, rs)
            kept = res > 0
            out[f"{k}_sd{sd}_bits"] = np.packbits(kept)
            out[f"{k}_sd{sd}_area"] = np.array(kept.sum() / kept.size)
    np.savez_compressed(GOLD / "cellarea.npz", **out)
    print("cellarea.npz", {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items()})


# --------------------------------------------------------------------------------------------
# wellmask group: run under /opt/conda/bin/python3.9 (scikit-image 0.18.3).  The reference module
# (fl_tissue_model_tools/well_mask_generation.py) is written against scikit-image 0.22; three API differences of the
# 0.18.3 copy are bridged HERE, in the generator, never in the reference:
#   * binary_erosion(image, footprint=...)  -> 0.18.3 spells the keyword `selem` (same function)
#   * rescale / resize(order=0, preserve_range=True) -> from 0.19 on these are scipy.ndimage.zoom(order=0, grid_mode=True,
#     mode="grid-constant"?) on the pixel-centre grid and keep a boolean input boolean (the module relies on that: it adds
#     the result to a boolean array in place); the bridge makes exactly that scipy call.  [recalled from the 0.19+ source,
#     which is not in this container]
#   * the random superellipse search draws from the GLOBAL numpy generator without a seed: the generator seeds it
#     (np.random.seed(seed)) before every call and stores the seed.
# --------------------------------------------------------------------------------------------
def wellmask_inputs():
    """float32 images in [0, 1] like the ones compute_branches.py:316-319 hands to make_well_mask"""
    from scipy import ndimage as ndi
    out = {}
    yy, xx = np.mgrid[0:640, 0:640].astype(np.float64)

    def finish(a, seed):
        rs = np.random.RandomState(seed)
        a = ndi.gaussian_filter(a + rs.normal(0, 0.03, a.shape), 2.0)
        a = (a - a.min()) / (a.max() - a.min())
        return a.astype(np.float32)
    out["round_bright"] = finish(((xx - 330) ** 2 + (yy - 310) ** 2 < 270 ** 2) * 0.6 + 0.2, 1)               # bright well, dark outside
    out["round_dark"] = finish(0.8 - ((xx - 300) ** 2 + (yy - 335) ** 2 < 280 ** 2) * 0.5, 2)                 # dark well, bright outside
    sq = (np.abs(xx - 320) / 285) ** 8 + (np.abs(yy - 320) / 285) ** 8 < 1
    out["square_bright"] = finish(sq * 0.5 + 0.25, 3)                                                          # rounded-square well
    a = np.full((480, 640), 0.5)
    a[(xx[:480] - 320) ** 2 / 300.0 ** 2 + (yy[:480] - 240) ** 2 / 225.0 ** 2 < 1] = 0.9
    out["nonsquare"] = finish(a, 4)
    out["blank"] = np.full((320, 320), 0.25, np.float32)                                                        # no structure: fallback paths
    # integer images, as compute_cell_area.py:127 hands them over (gaussian() then converts with img_as_float)
    out["round_u16"] = (out["round_dark"][::2, ::2] * 60000.0 + 1500.0).astype(np.uint16)
    out["square_u8"] = (out["square_bright"][::2, ::2] * 250.0).astype(np.uint8)
    return out


def gen_wellmask():
    import scipy.ndimage as ndi
    import skimage.morphology
    import skimage.transform
    _shims()
    _be = skimage.morphology.binary_erosion
    skimage.morphology.binary_erosion = lambda image, footprint=None, **kw: _be(image, selem=footprint, **kw)

    def _resize0(image, output_shape, order=None, preserve_range=False, **kw):
        assert order == 0 and preserve_range
        image = np.asarray(image)
        zoom = [o / i for o, i in zip(output_shape, image.shape)]
        src = image if image.dtype == bool else image.astype(np.float64)
        out = ndi.zoom(src.astype(np.float64), zoom, order=0, mode="nearest", grid_mode=True)
        return out.astype(bool) if image.dtype == bool else out

    def _rescale0(image, scale, order=None, preserve_range=False, **kw):
        shape = tuple(int(v) for v in np.round(np.asarray(image.shape) * scale))
        return _resize0(image, shape, order=order, preserve_range=preserve_range)
    skimage.transform.resize = _resize0
    skimage.transform.rescale = _rescale0
    from fl_tissue_model_tools import well_mask_generation as wm
    out = {}
    names = []
    for name, img in wellmask_inputs().items():
        for seed in (0, 7):
            key = f"{name}_s{seed}"
            out[key + "_thresh"] = np.packbits(wm.auto_threshold_well(img))
            np.random.seed(seed)
            res = wm.generate_well_mask(img, return_superellipse_params=True)
            if isinstance(res, tuple):
                mask, t, d, s_a, s_b, c_x, c_y, n = res
                out[key + "_params"] = np.array([t, d, s_a, s_b, c_x, c_y, n], np.float64)
            else:
                mask = res
                out[key + "_params"] = np.zeros(0)
            out[key + "_mask"] = np.packbits(np.asarray(mask) > 0)
            out[key + "_shape"] = np.array(img.shape, np.int64)
            # mask_val = 255 form of compute_cell_area.py:127
            np.random.seed(seed)
            m255 = wm.generate_well_mask(img, mask_val=255)
            assert np.array_equal(np.asarray(m255) > 0, np.asarray(mask) > 0)
            names.append(key)
            print(key, "coverage", float((np.asarray(mask) > 0).mean()), "params", out[key + "_params"])
    out["names"] = np.array(names)
    np.savez_compressed(GOLD / "wellmask.npz", **out)
    print("wellmask.npz", len(names), "cases")


if __name__ == "__main__":
    GOLD.mkdir(parents=True, exist_ok=True)
    _shims()
    groups = sys.argv[1:] or ["blend"]
    for g in groups:
        globals()["gen_" + g]()
