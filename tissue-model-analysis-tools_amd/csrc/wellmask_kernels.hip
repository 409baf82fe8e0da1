// Well detection (--detect-well): device stages of the reference's fl_tissue_model_tools/well_mask_generation.py.
//
//   auto_threshold_well (:236-277): gaussian(sigma 1) -> rescale_intensity(0..255) -> uint8 -> medians of the four 5 % corners
//   decide whether the outside of the well is the bright or the dark side (invert) -> Otsu threshold -> binary erosion, disk(5).
//   The gaussian is the corr1d kernel of sato_kernels.hip (scipy's correlate1d order); what is new here: the uint8 rescale, one
//   histogram pass (whole image + the four corners, LDS-privatised), the decision kernel -- corner medians and
//   skimage.filters.threshold_otsu evaluated from the 256-bin histograms exactly as numpy does (sequential f64 cumulative
//   sums: exact integers below 2^53; correctly rounded divisions) -- thresholding and the 81-tap erosion.
//   canny(mask) (:165, :201): gaussian(sigma 1, mode="constant") of the mask and of an all-ones image by zero padding, their
//   quotient, then the sobel / non-maximum suppression / hysteresis kernels of sato_kernels.hip.
// oracle/wellmask.py restates the same steps with numpy / scipy and is pinned to the reference by tests/golden/wellmask.npz.
#include "tmat_internal.h"
#include "wellmask.h"

namespace tmat {

static inline dim3 wm_grid(size_t n) { const size_t b = (n + 255) / 256; return dim3((unsigned)(b < 4096 ? (b ? b : 1) : 4096)); }
#define WM_LOOP(p, n) for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < (n); p += (size_t)gridDim.x * blockDim.x)

// exposure.py:rescale_intensity on a float32 image: ((x - min) / float32(max - min)) * 255 in float32, widened, truncated
__global__ void wm_rescale_u8_kernel(const float *__restrict__ x, size_t n, const float *__restrict__ mn, const float *__restrict__ mx,
                                     uint8_t *__restrict__ out)
{
    const float lo = mn[0], hi = mx[0];
    const float rng = (float)((double)hi - (double)lo);
    WM_LOOP(p, n) {
        float v = fminf(fmaxf(x[p], lo), hi);
        if (lo != hi) {
            v = (v - lo) / rng;
            v = v * 255.0f + 0.0f;
        } else {
            v = fminf(fmaxf(v, 0.0f), 255.0f);
        }
        out[p] = (uint8_t)(int)(double)v;
    }
}
void launch_wm_rescale_u8(const float *blur, size_t n, const float *mn, const float *mx, uint8_t *out, hipStream_t s)
{
    hipLaunchKernelGGL(wm_rescale_u8_kernel, wm_grid(n), dim3(256), 0, s, blur, n, mn, mx, out);
}

// the float64 form (integer images reach gaussian() through img_as_float -> float64): extrema by one workgroup, then the rescale
__global__ __launch_bounds__(1024) void wm_minmax_f64_kernel(const double *__restrict__ x, size_t n, double *__restrict__ mm)
{
    __shared__ double smn[16], smx[16];
    double lo = __builtin_inf(), hi = -__builtin_inf();
    for (size_t p = threadIdx.x; p < n; p += 1024) { const double v = x[p]; lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
    for (int o = 32; o > 0; o >>= 1) { const double a = __shfl_down(lo, o), b = __shfl_down(hi, o); lo = a < lo ? a : lo; hi = b > hi ? b : hi; }
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = lo; smx[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; w++) { lo = smn[w] < lo ? smn[w] : lo; hi = smx[w] > hi ? smx[w] : hi; }
        mm[0] = lo; mm[1] = hi;
    }
}
__global__ void wm_rescale_u8_f64_kernel(const double *__restrict__ x, size_t n, const double *__restrict__ mm, uint8_t *__restrict__ out)
{
    const double lo = mm[0], hi = mm[1], rng = hi - lo;
    WM_LOOP(p, n) {
        double v = fmin(fmax(x[p], lo), hi);
        if (lo != hi) {
            v = (v - lo) / rng;
            v = v * 255.0 + 0.0;
        } else {
            v = fmin(fmax(v, 0.0), 255.0);
        }
        out[p] = (uint8_t)(int)v;
    }
}
void launch_wm_rescale_u8_f64(const double *blur, size_t n, double *mm, uint8_t *out, hipStream_t s)
{
    hipLaunchKernelGGL(wm_minmax_f64_kernel, dim3(1), dim3(1024), 0, s, blur, n, mm);
    hipLaunchKernelGGL(wm_rescale_u8_f64_kernel, wm_grid(n), dim3(256), 0, s, blur, n, mm, out);
}

__global__ __launch_bounds__(256) void wm_hist_kernel(const uint8_t *__restrict__ img, int H, int W, unsigned *__restrict__ hist)
{
    __shared__ unsigned h[5 * 256];
    for (int i = threadIdx.x; i < 5 * 256; i += 256) h[i] = 0;
    __syncthreads();
    const int xl = (int)(H * 0.05), xr = (int)(H * 0.95), yt = (int)(W * 0.05), yb = (int)(W * 0.95);
    WM_LOOP(p, (size_t)H * W) {
        const int y = (int)(p / W), x = (int)(p - (size_t)y * W);
        const unsigned v = img[p];
        atomicAdd(&h[v], 1u);
        const bool top = y < xl, bot = y >= xr, left = x < yt, right = x >= yb;
        if (top && left) atomicAdd(&h[256 + v], 1u);
        if (top && right) atomicAdd(&h[512 + v], 1u);
        if (bot && left) atomicAdd(&h[768 + v], 1u);
        if (bot && right) atomicAdd(&h[1024 + v], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 5 * 256; i += 256) if (h[i]) atomicAdd(&hist[i], h[i]);
}
void launch_wm_hist(const uint8_t *img, int H, int W, unsigned *hist, hipStream_t s)
{
    hipMemsetAsync(hist, 0, 5 * 256 * sizeof(unsigned), s);
    hipLaunchKernelGGL(wm_hist_kernel, wm_grid((size_t)H * W), dim3(256), 0, s, img, H, W, hist);
}

// np.median of the values counted in h[0..255] (n > 0): the middle value, or the mean of the two middle values
__device__ double wm_median(const unsigned *h)
{
    unsigned long long n = 0;
    for (int v = 0; v < 256; v++) n += h[v];
    if (n == 0) return __builtin_nan("");
    const unsigned long long k0 = (n - 1) / 2, k1 = n / 2;
    unsigned long long c = 0;
    int a = -1, b = -1;
    for (int v = 0; v < 256; v++) {
        c += h[v];
        if (a < 0 && c > k0) a = v;
        if (b < 0 && c > k1) { b = v; break; }
    }
    return ((double)a + (double)b) / 2.0;       // np.median: mean of the two middle elements (equal for odd n)
}

// one thread: the whole decision is a few thousand scalar operations on 1280 counters
__global__ void wm_decide_kernel(const unsigned *__restrict__ hist, int H, int W, int *__restrict__ decision)
{
    if (blockIdx.x || threadIdx.x) return;
    int lo = 0, hi = 255;
    while (lo < 255 && hist[lo] == 0) lo++;
    while (hi > 0 && hist[hi] == 0) hi--;
    double mmin = __builtin_inf(), mmax = -__builtin_inf();
    for (int c = 0; c < 4; c++) {
        const double m = wm_median(hist + 256 * (c + 1));
        mmin = m < mmin ? m : mmin;             // python min / max over the list (NaN never wins here: images are >= 20 px)
        mmax = m > mmax ? m : mmax;
    }
    const bool invert = fabs((double)lo - mmin) > fabs((double)hi - mmax);
    decision[1] = invert;
    // histogram of the image Otsu sees: v -> 255 - v when inverted
    const int ilo = invert ? 255 - hi : lo, ihi = invert ? 255 - lo : hi;
    if (ilo == ihi) { decision[0] = ilo; return; }      // a constant image: threshold_otsu returns that value
    const int nb = ihi - ilo + 1;
    // class weights and means for every split, numpy's cumulative sums in f64 (thresholding.py:threshold_otsu)
    double best = -1.0;
    int arg = 0;
    double total_w = 0.0, total_m = 0.0;
    for (int k = 0; k < nb; k++) {
        const double cnt = (double)hist[invert ? 255 - (ilo + k) : ilo + k];
        total_w += cnt;
    }
    // weight2 / mean2 come from cumulative sums of the REVERSED arrays: accumulate from the top
    // (two passes over 256 bins: keep the reversed cumulative sums in registers by walking k downwards first)
    double w2[256], s2[256];
    {
        double cw = 0.0, cs = 0.0;
        for (int k = nb - 1; k >= 0; k--) {
            const double cnt = (double)hist[invert ? 255 - (ilo + k) : ilo + k];
            cw += cnt;
            cs += cnt * (double)(ilo + k);
            w2[k] = cw;
            s2[k] = cs;
        }
    }
    double cw1 = 0.0, cs1 = 0.0;
    for (int k = 0; k < nb - 1; k++) {
        const double cnt = (double)hist[invert ? 255 - (ilo + k) : ilo + k];
        cw1 += cnt;
        cs1 += cnt * (double)(ilo + k);
        const double m1 = cs1 / cw1, m2 = s2[k + 1] / w2[k + 1];
        const double d = m1 - m2;
        const double var = cw1 * w2[k + 1] * (d * d);
        if (var > best) { best = var; arg = k; }        // np.argmax: the first maximum (NaN from 0 / 0 never compares greater)
    }
    (void)total_w; (void)total_m;
    decision[0] = ilo + arg;
}
void launch_wm_decide(const unsigned *hist, int H, int W, int *decision, hipStream_t s)
{
    hipLaunchKernelGGL(wm_decide_kernel, dim3(1), dim3(64), 0, s, hist, H, W, decision);
}

__global__ void wm_threshold_kernel(const uint8_t *__restrict__ img, size_t n, const int *__restrict__ decision, uint8_t *__restrict__ out)
{
    const int t = decision[0], inv = decision[1];
    WM_LOOP(p, n) {
        const int v = inv ? 255 - (int)img[p] : (int)img[p];
        out[p] = v >= t;
    }
}
void launch_wm_threshold(const uint8_t *img, size_t n, const int *decision, uint8_t *out, hipStream_t s)
{
    hipLaunchKernelGGL(wm_threshold_kernel, wm_grid(n), dim3(256), 0, s, img, n, decision, out);
}

__global__ void wm_erode_kernel(const uint8_t *__restrict__ m, int H, int W, const int *__restrict__ off, int noff, uint8_t *__restrict__ out)
{
    WM_LOOP(p, (size_t)H * W) {
        const int y = (int)(p / W), x = (int)(p - (size_t)y * W);
        bool acc = true;
        for (int k = 0; k < noff && acc; k++) {
            const int yy = y + off[2 * k], xx = x + off[2 * k + 1];
            if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) acc = m[(size_t)yy * W + xx] != 0;
        }
        out[p] = acc;
    }
}
void launch_wm_erode(const uint8_t *m, int H, int W, const int *off, int noff, uint8_t *out, hipStream_t s)
{
    hipLaunchKernelGGL(wm_erode_kernel, wm_grid((size_t)H * W), dim3(256), 0, s, m, H, W, off, noff, out);
}

__global__ void wm_pad_kernel(const uint8_t *__restrict__ m, int H, int W, int r, double *__restrict__ a, double *__restrict__ b)
{
    const int Hp = H + 2 * r, Wp = W + 2 * r;
    WM_LOOP(p, (size_t)Hp * Wp) {
        const int y = (int)(p / Wp) - r, x = (int)(p % Wp) - r;
        const bool in = (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W;
        a[p] = in && m[(size_t)y * W + x] ? 1.0 : 0.0;
        b[p] = in ? 1.0 : 0.0;
    }
}
void launch_wm_pad(const uint8_t *m, int H, int W, int r, double *img_pad, double *ones_pad, hipStream_t s)
{
    hipLaunchKernelGGL(wm_pad_kernel, wm_grid((size_t)(H + 2 * r) * (W + 2 * r)), dim3(256), 0, s, m, H, W, r, img_pad, ones_pad);
}
__global__ void wm_crop_div_kernel(const double *__restrict__ a, const double *__restrict__ b, int H, int W, int r, double *__restrict__ out)
{
    const int Wp = W + 2 * r;
    WM_LOOP(p, (size_t)H * W) {
        const int y = (int)(p / W), x = (int)(p - (size_t)y * W);
        const size_t q = (size_t)(y + r) * Wp + x + r;
        out[p] = a[q] / (b[q] + 2.220446049250313e-16);
    }
}
void launch_wm_crop_div(const double *img_pad, const double *ones_pad, int H, int W, int r, double *out, hipStream_t s)
{
    hipLaunchKernelGGL(wm_crop_div_kernel, wm_grid((size_t)H * W), dim3(256), 0, s, img_pad, ones_pad, H, W, r, out);
}

}  // namespace tmat
