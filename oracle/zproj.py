"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the Z-projection methods of the reference
(fl_tissue_model_tools/zstacks.py:134-249, driven by scripts/compute_zproj.py:73-84).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Focus stacking (zstacks.py:134-189) calls two OpenCV functions.  opencv-python (>=4.9,<4.10, setup.py:63) is absent from
this environment and the reference holds no test or fixture for them, so this restatement follows OpenCV's published
behaviour and is **parity unpinned** against cv2 itself:

* ``cv2.GaussianBlur(img, (5, 5), 0)``: for ksize <= 7 and sigma <= 0 OpenCV uses its fixed kernel
  [1, 4, 6, 4, 1] / 16 per axis; 8- and 16-bit images go through the fixed-point path, whose result is
  ``(sum_ij k_i k_j I + 128) >> 8`` (round half up); border BORDER_REFLECT_101; output dtype = input dtype.
* ``cv2.Laplacian(blurred, cv2.CV_64F, ksize=5)``: d2/dx2 + d2/dy2 with the Sobel kernels of aperture 5
  (derivative [1, 0, -2, 0, 1], smoothing [1, 4, 6, 4, 1]), scale 1, BORDER_REFLECT_101.  For 8/16-bit sources every
  intermediate is an integer below 2**24, so the float work type OpenCV uses is exact and so is integer arithmetic.

The projection itself (zstacks.py:176-187) keeps, per pixel, the value of the first slice with the strictly largest
|Laplacian|.
"""
import numpy as np

G5 = np.array([1, 4, 6, 4, 1], np.int64)
D5 = np.array([1, 0, -2, 0, 1], np.int64)
# 5x5 Laplacian kernel K[dy][dx] = d2/dx2 (D5 along x, G5 along y) + d2/dy2 (G5 along x, D5 along y)
K5 = np.outer(G5, D5) + np.outer(D5, G5)


def reflect101(i, n):
    """BORDER_REFLECT_101 index (gfedcb|abcdefgh|gfedcba); n == 1 maps everything to 0."""
    if n == 1:
        return np.zeros_like(i)
    p = 2 * (n - 1)
    i = np.mod(i, p)
    return np.where(i < n, i, p - i)


def _window_sum(img, kernel2d):
    """sum_{dy,dx} kernel2d[dy+2][dx+2] * img[reflect(y+dy)][reflect(x+dx)] in int64"""
    H, W = img.shape
    a = img.astype(np.int64)
    ys, xs = np.arange(H), np.arange(W)
    out = np.zeros((H, W), np.int64)
    for dy in range(-2, 3):
        ry = reflect101(ys + dy, H)
        for dx in range(-2, 3):
            k = int(kernel2d[dy + 2, dx + 2])
            if k:
                out += k * a[ry][:, reflect101(xs + dx, W)]
    return out


def gaussian_blur5(img):
    """cv2.GaussianBlur(img, (5, 5), 0) for uint8 / uint16   (zstacks.py:149)"""
    assert img.dtype in (np.uint8, np.uint16)
    return ((_window_sum(img, np.outer(G5, G5)) + 128) >> 8).astype(img.dtype)


def laplacian5(img):
    """cv2.Laplacian(img, cv2.CV_64F, ksize=5)   (zstacks.py:150)"""
    return _window_sum(img, K5).astype(np.float64)


def blur_and_lap(image, kernel_size=5):
    assert kernel_size == 5
    return laplacian5(gaussian_blur5(image))


def proj_focus_stacking(stack, axis=0):
    """zstacks.py:153-189"""
    if axis != 0:
        stack = np.moveaxis(stack, axis, 0)
    maxima = np.full(stack[0].shape, -np.inf, np.float32)
    zproj = stack[0].copy()
    for pos in stack:
        al = np.absolute(blur_and_lap(pos))
        m = al > maxima
        maxima[m] = al[m]
        zproj[m] = pos[m]
    return zproj


def proj_avg(stack, axis=0):
    return np.mean(stack, axis=axis)          # zstacks.py:192-204


def proj_med(stack, axis=0):
    return np.median(stack, axis=axis)        # zstacks.py:207-219


def proj_max(stack, axis=0):
    return np.max(stack, axis=axis)           # zstacks.py:222-235


def proj_min(stack, axis=0):
    return np.min(stack, axis=axis)           # zstacks.py:238-249
