"""oracle/sato.py -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

CPU restatement of the Z-stack (Sato) branch of analyze_img (reference scripts/compute_branches.py:224-306): the
per-slice gaussian, the anti-aliased resize of the stack to the 384-wide grid, the 0..1 rescale, the Sato tubeness
filter over pairs of adjacent slices, unsharp masking of the volume, max projection, Canny edges, the medial axis of the
edges, the eccentricity x diameter test, masked blurring, region growing, closing, filter_branch_seg_mask(None, False),
dilation and the final gaussian that gives the 384-wide vesselness field MorseGraph then reads.

All arithmetic of this branch lives in scikit-image / scipy (third-party: scikit-image==0.22.0, scipy==1.13.1 in the
reference's setup.py:66-67), not under /root/reference.  scikit-image is not importable in the test interpreter; the
copy in /opt/conda (0.18.3) generated tests/golden/sato.npz (tools/make_goldens.py sato) stage by stage, and this file
restates those functions with numpy + scipy.ndimage so that the goldens pin it:
  filters.gaussian            -> ndi.gaussian_filter (mode 'nearest' default, truncate 4, float32 stays float32)
  transform.resize (3-D)      -> scikit-image >= 0.19 semantics, as for the 2-D path (oracle/morph.py:resize_aa): gaussian
                                 (mode 'mirror') over the scaled axes, grid-mode linear zoom, clip to the range of the WHOLE
                                 stack.  (0.18.3's own nD resize maps corner to corner -- ndi.zoom without grid_mode -- and is
                                 NOT what the pinned 0.22 does; this stage is therefore pinned to scipy, not to the goldens.)
  filters.sato (2-D)          -> two forms.  "gradient" (scikit-image <= 0.19, pinned by the goldens): 1 - image
                                 (util.invert), gaussian_filter(mode 'reflect') per sigma, np.gradient twice, sigma^2 scaling,
                                 closed-form eigenvalues (feature/corner.py:_image_orthogonal_matrix22_eigvals), max over
                                 sigma of max(l_max, 0); all in float32, in numpy's operation order.  "gaussian_derivatives"
                                 (>= 0.20, what the pinned 0.22.0 runs; the default of vessel_field): see sato2d_derivatives --
                                 restated from the published source, every primitive the real scipy call, the composition
                                 PARITY UNPINNED (no fixture of this branch in the reference, no scikit-image >= 0.20 here)
  filters.unsharp_mask        -> image + (image - gaussian(image, 2, mode 'reflect')) * 2, clipped to 0..1, on the 3-D volume
  feature.canny(sigma=0)      -> feature/_canny.py restated (sobel, 4-sector non-maximum suppression, 0.1 / 0.2 thresholds,
                                 hysteresis through 8-connected labels)
  measure.regionprops         -> eccentricity (inertia-tensor eigenvalues) and equivalent_diameter per 8-connected region
                                 ('equivalent_diameter_area' is the 0.19+ name of the same property)
  morphology.closing/dilation -> ndi.grey_dilation / grey_erosion with the footprint (mode 'reflect')
"""
from __future__ import annotations

import math

import numpy as np
from scipy import ndimage as ndi

from . import morph

SATO_SIGMAS = (1, 2, 3, 4, 5, 7, 9, 11, 13, 15)          # compute_branches.py:262
F32 = np.float32


def gaussian(img: np.ndarray, sigma=1.0, mode="nearest") -> np.ndarray:
    """skimage.filters.gaussian: float32 stays float32, everything else computes in float64 (preserve_range semantics
    are the caller's business here)"""
    a = img if img.dtype in (np.float32, np.float64) else img.astype(np.float64)
    return ndi.gaussian_filter(a, sigma, mode=mode, cval=0, truncate=4.0)


def stack_prepare(stack: np.ndarray, out_hw) -> np.ndarray:
    """compute_branches.py:247-257: per-slice gaussian written back into the INTEGER stack (C truncation), resize to
    (Z, out_h, out_w) -- scikit-image >= 0.19's resize is these two scipy calls on the 3-D array, the Z axis with sigma 0 and
    zoom 1 -- clip to the stack's range, rescale to 0..1 over the whole stack, float32"""
    st = np.array(stack, copy=True)
    for i in range(len(st)):
        st[i, :] = gaussian(st[i], 1.0)                                    # float64 -> integer dtype: truncation
    vol = st.astype(np.float64)
    out_shape = (len(vol),) + tuple(int(v) for v in out_hw)
    factors = np.divide(vol.shape, out_shape)
    sig = np.maximum(0, (factors - 1) / 2)
    filt = ndi.gaussian_filter(vol, sig, cval=0, mode="mirror")
    out = ndi.zoom(filt, [1 / f for f in factors], order=1, mode="mirror", cval=0, grid_mode=True)
    out = np.clip(out, vol.min(), vol.max())
    return morph.rescale_intensity(out, (0, 1)).astype(np.float32)


def hessian_eigmax(inv: np.ndarray, sigma: int) -> np.ndarray:
    """larger eigenvalue of sigma^2 * Hessian(gaussian_sigma(inv)), float32, numpy's operation order"""
    g = ndi.gaussian_filter(inv, sigma=sigma, mode="reflect", cval=0)      # float32 in -> float32 out
    gr, gc = np.gradient(g)
    hrr = np.gradient(gr, axis=0)
    hrc = np.gradient(gc, axis=0)
    hcc = np.gradient(gc, axis=1)
    s2 = F32(sigma ** 2)
    m00, m01, m11 = s2 * hrr, s2 * hrc, s2 * hcc
    return (m00 + m11) / 2 + np.sqrt(4 * m01 ** 2 + (m00 - m11) ** 2) / 2


def sato2d_gradient(im: np.ndarray, sigmas=SATO_SIGMAS) -> np.ndarray:
    """skimage.filters.sato(im, sigmas, black_ridges=False) of scikit-image <= 0.19 for a 2-D float32 image (pinned by the
    0.18.3 goldens): Hessian = np.gradient twice of the gaussian-smoothed inverted image"""
    inv = F32(1) - im.astype(np.float32)
    best = np.zeros(im.shape, np.float32)
    for s in sigmas:
        l1 = hessian_eigmax(inv, s)
        best = np.maximum(best, np.where(l1 > 0, np.abs(l1), F32(0)))
    return best


def sato2d_derivatives(im: np.ndarray, sigmas=SATO_SIGMAS) -> np.ndarray:
    """the same call under scikit-image >= 0.20 (the reference pins 0.22.0), restated from the published source
    (filters/ridges.py:sato, feature/corner.py:_hessian_matrix_with_gaussian, _symmetric_compute_eigenvalues): image = -image;
    Hessian by two successive first-order gaussian-derivative filters of sigma / sqrt(2) (truncate 8, or 100 when sigma <= 1,
    mode 'reflect'); eigenvalues (M00 + M11) / 2 +- sqrt(M01^2 + ((M00 - M11) / 2)^2); vesselness sigma^2 * max(l_max, 0).
    PARITY UNPINNED as a composition: no scikit-image >= 0.20 is importable here and the reference holds no fixture of this
    branch; each primitive below is the real scipy call."""
    img = -im.astype(np.float32)
    best = np.zeros(im.shape, np.float32)
    sq1_2 = 1 / math.sqrt(2)
    for s in sigmas:
        kw = dict(sigma=(sq1_2 * s, sq1_2 * s), mode="reflect", cval=0, truncate=8 if s > 1 else 100)
        g0 = ndi.gaussian_filter(img, order=[1, 0], **kw)
        g1 = ndi.gaussian_filter(img, order=[0, 1], **kw)
        m00 = ndi.gaussian_filter(g0, order=[1, 0], **kw)
        m01 = ndi.gaussian_filter(g0, order=[0, 1], **kw)
        m11 = ndi.gaussian_filter(g1, order=[0, 1], **kw)
        l1 = (m00 + m11) / 2 + np.sqrt(m01 ** 2 + ((m00 - m11) / 2) ** 2)
        best = np.maximum(best, F32(s ** 2) * np.maximum(l1, F32(0)))
    return best


def sato2d(im: np.ndarray, sigmas=SATO_SIGMAS, hessian="gradient") -> np.ndarray:
    return sato2d_gradient(im, sigmas) if hessian == "gradient" else sato2d_derivatives(im, sigmas)


def unsharp_mask(vol: np.ndarray, radius=2, amount=2) -> np.ndarray:
    """skimage.filters.unsharp_mask on a float32 array of any dimension (non-negative input: clipped to 0..1)"""
    blurred = gaussian(vol, radius, mode="reflect")
    res = vol + (vol - blurred) * amount
    lo, hi = ((-1.0, 1.0) if (vol < 0).any() else (0.0, 1.0))
    return np.clip(res, lo, hi).astype(vol.dtype)


def canny0(image: np.ndarray) -> np.ndarray:
    """skimage.feature.canny(image, sigma=0) with the default thresholds 0.1 / 0.2 (feature/_canny.py)"""
    sm = image.astype(image.dtype if image.dtype in (np.float32, np.float64) else np.float64) / (np.ones(image.shape) + np.finfo(float).eps)
    return canny_from_smoothed(sm)


def canny_from_smoothed(sm: np.ndarray) -> np.ndarray:
    """feature/_canny.py after the smoothing step: sobel, 4-sector non-maximum suppression, hysteresis (0.1 / 0.2)"""
    image = sm
    js = ndi.sobel(sm, axis=1)
    is_ = ndi.sobel(sm, axis=0)
    ai, aj = np.abs(is_), np.abs(js)
    mag = np.hypot(is_, js)
    er = ndi.binary_erosion(np.ones(image.shape, bool), ndi.generate_binary_structure(2, 2), border_value=0) & (mag > 0)
    lm = np.zeros(image.shape, bool)

    def sector(pp, pm, a, b):
        pts = er & (pp | pm)
        return pts

    # 0 .. 45 degrees
    pts = er & (((is_ >= 0) & (js >= 0) & (ai >= aj)) | ((is_ <= 0) & (js <= 0) & (ai >= aj)))
    c1 = mag[1:, :][pts[:-1, :]]; c2 = mag[1:, 1:][pts[:-1, :-1]]; m = mag[pts]; w = aj[pts] / ai[pts]
    cp = c2 * w + c1 * (1 - w) <= m
    c1 = mag[:-1, :][pts[1:, :]]; c2 = mag[:-1, :-1][pts[1:, 1:]]
    cm = c2 * w + c1 * (1 - w) <= m
    lm[pts] = cp & cm
    # 45 .. 90
    pts = er & (((is_ >= 0) & (js >= 0) & (ai <= aj)) | ((is_ <= 0) & (js <= 0) & (ai <= aj)))
    c1 = mag[:, 1:][pts[:, :-1]]; c2 = mag[1:, 1:][pts[:-1, :-1]]; m = mag[pts]; w = ai[pts] / aj[pts]
    cp = c2 * w + c1 * (1 - w) <= m
    c1 = mag[:, :-1][pts[:, 1:]]; c2 = mag[:-1, :-1][pts[1:, 1:]]
    cm = c2 * w + c1 * (1 - w) <= m
    lm[pts] = cp & cm
    # 90 .. 135
    pts = er & (((is_ <= 0) & (js >= 0) & (ai <= aj)) | ((is_ >= 0) & (js <= 0) & (ai <= aj)))
    c1 = mag[:, 1:][pts[:, :-1]]; c2 = mag[:-1, 1:][pts[1:, :-1]]; m = mag[pts]; w = ai[pts] / aj[pts]
    cp = c2 * w + c1 * (1.0 - w) <= m
    c1 = mag[:, :-1][pts[:, 1:]]; c2 = mag[1:, :-1][pts[:-1, 1:]]
    cm = c2 * w + c1 * (1.0 - w) <= m
    lm[pts] = cp & cm
    # 135 .. 180
    pts = er & (((is_ <= 0) & (js >= 0) & (ai >= aj)) | ((is_ >= 0) & (js <= 0) & (ai >= aj)))
    c1 = mag[:-1, :][pts[1:, :]]; c2 = mag[:-1, 1:][pts[1:, :-1]]; m = mag[pts]; w = aj[pts] / ai[pts]
    cp = c2 * w + c1 * (1 - w) <= m
    c1 = mag[1:, :][pts[:-1, :]]; c2 = mag[1:, :-1][pts[:-1, 1:]]
    cm = c2 * w + c1 * (1 - w) <= m
    lm[pts] = cp & cm
    high = lm & (mag >= 0.2)
    low = lm & (mag >= 0.1)
    lab, n = ndi.label(low, np.ones((3, 3), bool))
    if n == 0:
        return low
    sums = np.atleast_1d(ndi.sum(high, lab, np.arange(n, dtype=np.int32) + 1))
    good = np.zeros(n + 1, bool)
    good[1:] = sums > 0
    return good[lab]


def ecc_times_diameter(mask: np.ndarray) -> np.ndarray:
    """regionprops_image(mask, 'eccentricity') * regionprops_image(mask, 'equivalent_diameter[_area]')
    (reference transforms.py:291-303), per 8-connected region, 0 on the background"""
    lab, n = ndi.label(mask, np.ones((3, 3), bool))
    vals = np.zeros(n + 1)
    for i, sl in enumerate(ndi.find_objects(lab), 1):
        reg = lab[sl] == i
        rr, cc = np.nonzero(reg)
        area = float(len(rr))
        r0, c0 = rr.sum() / area, cc.sum() / area                          # centroid in the region's own frame
        dr, dc = rr - r0, cc - c0
        mu20, mu02, mu11 = (dr * dr).sum(), (dc * dc).sum(), (dr * dc).sum()   # mu[2,0] rows, mu[0,2] columns
        # inertia tensor (measure/_moments.py:inertia_tensor): T = [[mu02, -mu11], [-mu11, mu20]] / mu00
        t = np.array([[mu02, -mu11], [-mu11, mu20]]) / area
        ev = np.clip(np.linalg.eigvalsh(t), 0, None)
        l1, l2 = max(ev), min(ev)
        ecc = 0.0 if l1 == 0 else math.sqrt(1 - l2 / l1)
        diam = (2 * 2 * area / math.pi) ** 0.5
        vals[i] = ecc * diam
    return vals[lab]


def disk(r: int) -> np.ndarray:
    yy, xx = np.mgrid[-r:r + 1, -r:r + 1]
    return (yy * yy + xx * xx) <= r * r


def region_grow(mask: np.ndarray, vessels: np.ndarray, iters=10) -> np.ndarray:
    """compute_branches.py:283-294: a pixel joins when a mask neighbour is at least as bright as ... (flood towards the
    positive intensity gradient): it has a mask neighbour whose value is <= its own, none whose value is greater, and it
    is brighter than 0.01"""
    m = mask.astype(bool).copy()
    H, W = m.shape
    for _ in range(iters):
        lo = np.zeros((H, W), bool)
        hi = np.zeros((H, W), bool)
        for r in (-1, 0, 1):
            for c in (-1, 0, 1):
                if r == 0 and c == 0:
                    continue
                sl = {-1: slice(1, None), 0: slice(None, None), 1: slice(None, -1)}
                src = (sl[r], sl[c]); dst = (sl[-r], sl[-c])
                lt = vessels[dst] < vessels[src]
                lo[dst] |= m[src] & lt
                hi[dst] |= m[src] & ~lt
        m |= (vessels > 0.01) & hi & ~lo
    return m


def vessel_field(vol01: np.ndarray, return_stages=False, hessian="gaussian_derivatives"):
    """compute_branches.py:259-305 from the prepared float32 stack (Z, h, w) to the float32 vesselness image"""
    Z = len(vol01)
    vess = np.zeros((Z - 1,) + vol01.shape[1:], np.float32)
    for z in range(Z - 1):
        vess[z] = sato2d(np.maximum(vol01[z], vol01[z + 1]), SATO_SIGMAS, hessian)
    sharp = unsharp_mask(vess, 2, 2)
    vessels = sharp.max(0)
    edges = canny0(vessels)
    skel, _ = morph.medial_axis(edges)
    sel = skel & (ecc_times_diameter(skel) > 3.5)
    mask = sel
    v = vessels
    for _ in range(3):
        v = np.where(mask, gaussian(v), v)
    grown = region_grow(mask, v, 10)
    mask = grown & ~edges
    d2 = disk(2)
    closed = ndi.grey_erosion(ndi.grey_dilation(mask.astype(np.uint8), footprint=d2), footprint=d2).astype(bool)
    filt = morph.filter_branch_seg_mask(closed, use_median=False, remove_isolated=False)
    dil = ndi.grey_dilation(filt.astype(np.uint8), footprint=np.ones((3, 3), bool)).astype(bool)
    field = gaussian(np.where(dil, vessels, F32(0)).astype(np.float32))
    if return_stages:
        return field, dict(vess=vess, sharp=sharp, vessels=vessels, edges=edges, skel=skel, mask_sel=sel, grown=grown, closed=closed, filt=filt)
    return field


def resize_aa(img: np.ndarray, out_hw) -> np.ndarray:
    """skimage.transform.resize(img, out_hw, order=1, preserve_range=True, anti_aliasing=True) of a 2-D integer image (scikit-image
    >= 0.19: the two scipy calls of stack_prepare on the 2-D array, clipped to the input's range), float64"""
    a = np.asarray(img).astype(np.float64)
    factors = np.divide(a.shape, tuple(int(v) for v in out_hw))
    filt = ndi.gaussian_filter(a, np.maximum(0, (factors - 1) / 2), cval=0, mode="mirror")
    out = ndi.zoom(filt, [1 / f for f in factors], order=1, mode="mirror", cval=0, grid_mode=True)
    return np.clip(out, a.min(), a.max())


def analyze_stack(stack: np.ndarray, config: dict, image_width_microns: float, hessian="gaussian_derivatives", detect_well=False, well_seed=0,
                  return_masks=False):
    """(count, total_px, avg_px) of one Z stack through the Sato branch + the graph stages of the 2-D path.  detect_well
    (compute_branches.py:227-243): the well mask of the resized max projection; only its shrunken form enters, as MorseGraph's pruning mask"""
    from . import dmt, morse, pipeline, wellmask
    out_hw = morph.dsamp_shape(stack.shape[-2:], 384)
    pruning = well = None
    if detect_well:
        well, shrunken = wellmask.make_well_mask(resize_aa(stack.max(0), out_hw), seed=well_seed)
        pruning = np.logical_not(shrunken)
    field = vessel_field(stack_prepare(stack, out_hw), hessian=hessian)
    f255 = morph.rescale_intensity(field, (0, 255)).astype(np.float32)
    V, E = dmt.compute_dmt_graph(f255, float(config.get("graph_thresh_1", 5)), float(config.get("graph_thresh_2", 10)))
    sw, mn, mx = pipeline.px_params(config, 384, image_width_microns)
    _, n, tot, avg = morse.morse_stats(V, E, field.shape, sw, mn, mx, bool(config.get("remove_isolated_branches", False)), pruning)
    if return_masks:
        return (n, tot, avg), well, pruning
    return n, tot, avg
