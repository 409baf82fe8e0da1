// Cell-area tool on gfx950 (reference scripts/compute_cell_area.py:29-87, 164-178 and
// fl_tissue_model_tools/preprocessing.py:44-93): bilinear down-sampling, intensity histogram, a two-component gaussian
// mixture fitted to the histogram (2-means initialisation + scikit-learn's EM), threshold, area fraction.
//
// The mixture is fitted to the HISTOGRAM, not to the pixels: the image is 16-bit, so the 262 144 pixels of a 512 x 512
// image take at most 65 536 distinct values and every sum EM needs is a weighted sum over the bins.  One workgroup per
// image holds its 64 bins per thread in registers; an EM iteration is two block reductions of a few doubles.  The pixel
// kernels (resize, histogram, apply) are HBM-bound streaming passes of 0.5 - 2 MB per image: this tool is latency bound.
// Arithmetic contract: oracle/cellarea.py (same formulas in float64; block-tree instead of sequential summation).
#include "../../include/tmat.h"
#include "tmat_ctx.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace tmat {

// cv2.resize(INTER_LINEAR) of uint16 images: float32 weights, horizontal then vertical, round half to even, saturate
__global__ __launch_bounds__(256) void resize_linear_u16_kernel(const uint16_t *__restrict__ img, int H, int W, int oh, int ow, const int *__restrict__ r0,
                                                                const int *__restrict__ r1, const float *__restrict__ wr0, const float *__restrict__ wr1,
                                                                const int *__restrict__ c0, const int *__restrict__ c1, const float *__restrict__ wc0,
                                                                const float *__restrict__ wc1, uint16_t *__restrict__ out)
{
    const uint16_t *src = img + (size_t)blockIdx.y * H * W;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < oh * ow; p += gridDim.x * 256) {
        const int y = p / ow, x = p - y * ow;
        const float a = (float)src[(size_t)r0[y] * W + c0[x]] * wc0[x] + (float)src[(size_t)r0[y] * W + c1[x]] * wc1[x];
        const float b = (float)src[(size_t)r1[y] * W + c0[x]] * wc0[x] + (float)src[(size_t)r1[y] * W + c1[x]] * wc1[x];
        const float v = rintf(a * wr0[y] + b * wr1[y]);
        out[(size_t)blockIdx.y * oh * ow + p] = (uint16_t)fminf(fmaxf(v, 0.f), 65535.f);
    }
}

// cv2.resize(INTER_LINEAR) of uint8 images (widened to uint16 by the caller: values <= 255): OpenCV's FIXED-POINT path
// (imgproc/src/resize.cpp, resizeGeneric_ with fixed_pt for 8U): 11-bit coefficients a = cvRound(w * 2048) as short; horizontal pass
// into int32, S[c0] a0 + S[c1] a1; vertical pass of the 8-bit specialisation of VResizeLinear (what its SIMD lanes evaluate):
//     dst = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2.
// Reference call sites: scripts/compute_cell_area.py:54-57, fl_tissue_model_tools/data_prep.py:36 (cv2 is absent here: parity unpinned;
// oracle/cellarea.py:resize_linear_u8 restates the same source and tests/test_oracle_cellarea.py holds hand-derived vectors).
__global__ __launch_bounds__(256) void resize_linear_u8_kernel(const uint16_t *__restrict__ img, int H, int W, int oh, int ow, const int *__restrict__ r0,
                                                               const int *__restrict__ r1, const int *__restrict__ b0, const int *__restrict__ b1,
                                                               const int *__restrict__ c0, const int *__restrict__ c1, const int *__restrict__ a0,
                                                               const int *__restrict__ a1, uint16_t *__restrict__ out)
{
    const uint16_t *src = img + (size_t)blockIdx.y * H * W;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < oh * ow; p += gridDim.x * 256) {
        const int y = p / ow, x = p - y * ow;
        const int h0 = (int)src[(size_t)r0[y] * W + c0[x]] * a0[x] + (int)src[(size_t)r0[y] * W + c1[x]] * a1[x];
        const int h1 = (int)src[(size_t)r1[y] * W + c0[x]] * a0[x] + (int)src[(size_t)r1[y] * W + c1[x]] * a1[x];
        const int v = (((b0[y] * (h0 >> 4)) >> 16) + ((b1[y] * (h1 >> 4)) >> 16) + 2) >> 2;
        out[(size_t)blockIdx.y * oh * ow + p] = (uint16_t)(v < 0 ? 0 : v > 255 ? 255 : v);
    }
}

// cv::resize replaces INTER_LINEAR by INTER_AREA when both scale factors are exactly 2 (imgproc/src/resize.cpp: "if
// (interpolation == INTER_LINEAR && is_area_fast && iscale_x == 2 && iscale_y == 2) interpolation = INTER_AREA"); the
// INTER_AREA fast path for a 2 x 2 block is the integer mean rounded half up, (a + b + c + d + 2) >> 2
__global__ __launch_bounds__(256) void resize_area2_u16_kernel(const uint16_t *__restrict__ img, int H, int W, uint16_t *__restrict__ out)
{
    const int oh = H / 2, ow = W / 2;
    const uint16_t *src = img + (size_t)blockIdx.y * H * W;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < oh * ow; p += gridDim.x * 256) {
        const int y = p / ow, x = p - y * ow;
        const uint16_t *q = src + (size_t)(2 * y) * W + 2 * x;
        out[(size_t)blockIdx.y * oh * ow + p] = (uint16_t)(((unsigned)q[0] + q[1] + q[W] + q[W + 1] + 2u) >> 2);
    }
}

// mask (nullable): only pixels with mask != 0 are counted (exec_threshold fits the mixture to the pixels inside the well)
__global__ __launch_bounds__(256) void hist_u16_kernel(const uint16_t *__restrict__ img, const uint8_t *__restrict__ mask, int npx, unsigned *__restrict__ hist)
{
    const uint16_t *src = img + (size_t)blockIdx.y * npx;
    const uint8_t *m = mask ? mask + (size_t)blockIdx.y * npx : nullptr;
    unsigned *h = hist + (size_t)blockIdx.y * 65536;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < npx; p += gridDim.x * 256) if (!m || m[p]) atomicAdd(&h[src[p]], 1u);
}

// lowest / highest level of the WHOLE image (rescale_intensity runs before the mask is applied): lohi[2 i] = min, [2 i + 1] = max,
// initialised to 65536 / -1 by the host
__global__ __launch_bounds__(256) void lohi_u16_kernel(const uint16_t *__restrict__ img, int npx, int *__restrict__ lohi)
{
    const uint16_t *src = img + (size_t)blockIdx.y * npx;
    int lo = 65536, hi = -1;
    for (int p = blockIdx.x * 256 + threadIdx.x; p < npx; p += gridDim.x * 256) { const int v = src[p]; lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
    for (int o = 32; o > 0; o >>= 1) { const int a = __shfl_down(lo, o), b = __shfl_down(hi, o); lo = a < lo ? a : lo; hi = b > hi ? b : hi; }
    if ((threadIdx.x & 63) == 0) { atomicMin(&lohi[2 * blockIdx.y], lo); atomicMax(&lohi[2 * blockIdx.y + 1], hi); }
}

// ---- block reductions over 1024 threads (16 waves) ----
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double *scratch /* 16 * NV */)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; i++)
        for (int o = 32; o > 0; o >>= 1) v[i] += __shfl_down(v[i], o);
    __syncthreads();                                 // scratch may still be read from the previous reduction
    if (lane == 0)
        for (int i = 0; i < NV; i++) scratch[wave * NV + i] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; i++) {
        double t = 0.0;
        for (int w = 0; w < 16; w++) t += scratch[w * NV + i];
        v[i] = t;
    }
}

constexpr int GMM_BINS_PER_THREAD = 64;
constexpr double GMM_REG_COVAR = 1e-6, GMM_TOL = 1e-3;
constexpr int GMM_MAX_ITER = 100;

// params out per image: [0] threshold, [1..2] weights, [3..4] means, [5..6] variances, [7] iterations, [8] converged, [9] lo, [10] hi
// lohi (nullable): the image extrema when the histogram only covers a mask; a histogram without pixels gives NaN parameters
__global__ __launch_bounds__(1024) void gmm_hist_kernel(const unsigned *__restrict__ hist, double sd_coef, double *__restrict__ params, const int *__restrict__ lohi)
{
    __shared__ double scratch[16 * 6];
    __shared__ long long s_n[16], s_s[16];
    __shared__ double s_best[16];
    __shared__ int s_bestk[16];
    __shared__ int s_lo, s_hi;
    const unsigned *h = hist + (size_t)blockIdx.x * 65536;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int b0 = t * GMM_BINS_PER_THREAD;
    unsigned cnt[GMM_BINS_PER_THREAD];
    long long n_t = 0, s_t = 0;
    int lo_t = 65536, hi_t = -1;
#pragma unroll
    for (int i = 0; i < GMM_BINS_PER_THREAD; i++) {
        cnt[i] = h[b0 + i];
        n_t += cnt[i];
        s_t += (long long)cnt[i] * (b0 + i);
        if (cnt[i]) { lo_t = lo_t < b0 + i ? lo_t : b0 + i; hi_t = b0 + i; }
    }
    // lowest / highest occupied level (the image min / max of rescale_intensity)
    for (int o = 32; o > 0; o >>= 1) { const int l2 = __shfl_down(lo_t, o), h2 = __shfl_down(hi_t, o); lo_t = l2 < lo_t ? l2 : lo_t; hi_t = h2 > hi_t ? h2 : hi_t; }
    if (t == 0) { s_lo = 65536; s_hi = -1; }
    __syncthreads();
    if (lane == 0) { atomicMin(&s_lo, lo_t); atomicMax(&s_hi, hi_t); }
    // ---- 2-means optimum on the integer levels: exclusive prefix (n, s) per thread, then scan the thread's bins ----
    long long n_w = n_t, s_w = s_t;                                       // inclusive scan inside the wave
    for (int o = 1; o < 64; o <<= 1) {
        const long long n2 = __shfl_up(n_w, o), s2 = __shfl_up(s_w, o);
        if (lane >= o) { n_w += n2; s_w += s2; }
    }
    if (lane == 63) { s_n[wave] = n_w; s_s[wave] = s_w; }
    __syncthreads();
    long long n_pre = n_w - n_t, s_pre = s_w - s_t, N = 0, S = 0;
    for (int w = 0; w < 16; w++) { if (w < wave) { n_pre += s_n[w]; s_pre += s_s[w]; } N += s_n[w]; S += s_s[w]; }
    double best = -INFINITY;
    int bestk = 0;
    {
        long long n = n_pre, s = s_pre;
#pragma unroll
        for (int i = 0; i < GMM_BINS_PER_THREAD; i++) {
            n += cnt[i]; s += (long long)cnt[i] * (b0 + i);
            if (cnt[i] && n > 0 && n < N) {                                  // a present level that leaves both sides non-empty
                const double sd = (double)s, rd = (double)(S - s);
                const double between = sd * sd / (double)n + rd * rd / (double)(N - n);
                if (between > best) { best = between; bestk = b0 + i; }
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) {      // ties take the smallest level
        const double b2 = __shfl_down(best, o); const int k2 = __shfl_down(bestk, o);
        if (b2 > best || (b2 == best && k2 < bestk)) { best = b2; bestk = k2; }
    }
    if (lane == 0) { s_best[wave] = best; s_bestk[wave] = bestk; }
    __syncthreads();
    best = s_best[0]; bestk = s_bestk[0];
    for (int w = 1; w < 16; w++) if (s_best[w] > best || (s_best[w] == best && s_bestk[w] < bestk)) { best = s_best[w]; bestk = s_bestk[w]; }
    const int lo = lohi ? lohi[2 * blockIdx.x] : s_lo, hi = lohi ? lohi[2 * blockIdx.x + 1] : s_hi;
    const double dlo = (double)lo, drange = (double)hi - (double)lo;
    // pixel value of a level: rescale_intensity(img, (0, 1)).astype(float32)
    auto xval = [&](int level) -> double { return lo != hi ? (double)(float)(((double)level - dlo) / drange) : fmin(fmax((double)level, 0.0), 1.0); };
    const double n_tot = (double)N, eps10 = 10.0 * 2.220446049250313e-16;
    double w[2], mu[2], var[2];
    // M step from responsibilities resp(i, x, r0, r1) -> the level's log-likelihood term (0 for the one-hot start).  The E step's lower bound
    // is accumulated in the M step's first sweep (round 4: it needs the same log-sum-exp the responsibilities come from -- a sweep of its own
    // cost 2 exp + 1 log per level and iteration out of 13; the sums are separate accumulators with their own reductions, so nothing changes
    // in any of them)
    double ll_sum = 0.0;
    auto m_step = [&](auto resp) {
        double a[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < GMM_BINS_PER_THREAD; i++) {
            if (!cnt[i]) continue;
            const double c = (double)cnt[i], x = xval(b0 + i);
            double r0, r1;
            const double lse = resp(i, x, r0, r1);
            a[0] += r0 * c; a[1] += r1 * c; a[2] += r0 * (c * x); a[3] += r1 * (c * x);
            a[4] += c * lse;
        }
        block_sum<6>(a, scratch);
        ll_sum = a[4];
        const double nk0 = a[0] + eps10, nk1 = a[1] + eps10;
        mu[0] = a[2] / nk0; mu[1] = a[3] / nk1;
        double v[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < GMM_BINS_PER_THREAD; i++) {
            if (!cnt[i]) continue;
            const double c = (double)cnt[i], x = xval(b0 + i);
            double r0, r1;
            resp(i, x, r0, r1);
            v[0] += r0 * (c * ((x - mu[0]) * (x - mu[0]))); v[1] += r1 * (c * ((x - mu[1]) * (x - mu[1])));
        }
        block_sum<6>(v, scratch);
        var[0] = v[0] / nk0 + GMM_REG_COVAR; var[1] = v[1] / nk1 + GMM_REG_COVAR;
        w[0] = nk0 / n_tot; w[1] = nk1 / n_tot;
    };
    m_step([&](int i, double, double &r0, double &r1) { r0 = b0 + i <= bestk ? 1.0 : 0.0; r1 = 1.0 - r0; return 0.0; });
    double lower = -INFINITY;
    int iters = 0, converged = 0;
    const double log2pi = 1.8378770664093453;
    for (int it = 1; it <= GMM_MAX_ITER; it++) {
        iters = it;
        const double prev = lower;
        const double pc0 = 1.0 / sqrt(var[0]), pc1 = 1.0 / sqrt(var[1]);
        const double lw0 = log(w[0]), lw1 = log(w[1]), lp0 = log(pc0), lp1 = log(pc1);
        const double m0 = mu[0], m1 = mu[1];
        // E step: log responsibilities per level (kept as a closure: recomputed in the M step instead of stored); returns the level's
        // log-sum-exp, whose count-weighted mean is the lower bound
        auto e_resp = [&](int, double x, double &r0, double &r1) {
            const double y0 = (x - m0) * pc0, y1 = (x - m1) * pc1;
            const double a0 = -0.5 * (log2pi + y0 * y0) + lp0 + lw0, a1 = -0.5 * (log2pi + y1 * y1) + lp1 + lw1;
            const double mx = fmax(a0, a1);
            const double lse = mx + log(exp(a0 - mx) + exp(a1 - mx));
            r0 = exp(a0 - lse); r1 = exp(a1 - lse);
            return lse;
        };
        m_step(e_resp);
        lower = ll_sum / n_tot;
        if (fabs(lower - prev) < GMM_TOL) { converged = 1; break; }
    }
    if (t == 0) {
        const int fg = mu[1] > mu[0] ? 1 : 0;                    // np.argmax: the first of equal means
        double *p = params + (size_t)blockIdx.x * 11;
        p[0] = fmin(255.0, mu[fg] + sqrt(var[fg]) * sd_coef);
        p[1] = w[0]; p[2] = w[1]; p[3] = mu[0]; p[4] = mu[1]; p[5] = var[0]; p[6] = var[1];
        p[7] = (double)iters; p[8] = (double)converged; p[9] = dlo; p[10] = (double)hi;
    }
}

// gmm_masked = where(x <= thresh, 0, x); kept = gmm_masked > 0 -> 255 / 0, and the count of kept pixels
__global__ __launch_bounds__(256) void apply_threshold_kernel(const uint16_t *__restrict__ img, const uint8_t *__restrict__ mask, int npx,
                                                              const double *__restrict__ params, uint8_t *__restrict__ out, unsigned *__restrict__ kept)
{
    const double *p = params + (size_t)blockIdx.y * 11;
    const double thresh = p[0], lo = p[9], hi = p[10], range = hi - lo;
    const uint16_t *src = img + (size_t)blockIdx.y * npx;
    unsigned local = 0;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < npx; q += gridDim.x * 256) {
        const double lv = (double)src[q];
        const double x = lo != hi ? (double)(float)((lv - lo) / range) : fmin(fmax(lv, 0.0), 1.0);
        const bool k = !(x <= thresh) && x > 0.0 && (!mask || mask[(size_t)blockIdx.y * npx + q]);      // apply_mask: outside the well the image is 0
        if (out) out[(size_t)blockIdx.y * npx + q] = k ? 255 : 0;
        local += k;
    }
    for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(&kept[blockIdx.y], local);
}

// cv2's source coordinate and weights of INTER_LINEAR along one axis (float arithmetic, clamped to the image)
static void linear_axis(int n_src, int n_dst, std::vector<int> &i0, std::vector<int> &i1, std::vector<float> &w0, std::vector<float> &w1)
{
    const double scale = (double)n_src / (double)n_dst;
    i0.resize(n_dst); i1.resize(n_dst); w0.resize(n_dst); w1.resize(n_dst);
    for (int d = 0; d < n_dst; d++) {
        float fx = (float)(((double)d + 0.5) * scale - 0.5);
        int sx = (int)std::floor(fx);
        fx = fx - (float)sx;
        if (sx < 0) { sx = 0; fx = 0.f; }
        if (sx >= n_src - 1) { sx = n_src - 1; fx = 0.f; }
        i0[d] = sx; i1[d] = sx + 1 < n_src ? sx + 1 : n_src - 1;
        w0[d] = 1.0f - fx; w1[d] = fx;
    }
}

// cv2.resize(img, dsize) of the config-5 tools (bilinear, see the kernels) for n images already on the device: uploads the coefficient
// tables into `tab` ((oh + ow) * 4 dwords, device; must stay allocated until the stream has run the kernel) and launches the arithmetic of
// the SOURCE depth -- eight_bit: cv2's fixed-point path (tmat_set_input_depth(h, 8)), else the float path; an exact halving on both axes
// is INTER_AREA's integer mean for both depths.  Also used by tmat_inv_depth_predict (resnet_kernels.hip).
int launch_resize_linear_dev(const uint16_t *din, int n, int H, int W, int oh, int ow, bool eight_bit, int *tab, uint16_t *dsm, hipStream_t s)
{
    const int blocks = (oh * ow + 255) / 256;
    const dim3 grid(blocks < 1024 ? blocks : 1024, n);
    if (H == 2 * oh && W == 2 * ow) {
        hipLaunchKernelGGL(resize_area2_u16_kernel, grid, dim3(256), 0, s, din, H, W, dsm);
        return 0;
    }
    std::vector<int> r0, r1, c0, c1;
    std::vector<float> wr0, wr1, wc0, wc1;
    linear_axis(H, oh, r0, r1, wr0, wr1);
    linear_axis(W, ow, c0, c1, wc0, wc1);
    // one host block [r0 | r1 | c0 | c1 | wr0 | wr1 | wc0 | wc1]; eight_bit: the weights as cvRound(w * 2048) integers (saturate_cast<short>)
    std::vector<int> host((size_t)(oh + ow) * 4);
    int *hr0 = host.data(), *hr1 = hr0 + oh, *hc0 = hr1 + oh, *hc1 = hc0 + ow, *hw = hc1 + ow;
    for (int i = 0; i < oh; i++) { hr0[i] = r0[i]; hr1[i] = r1[i]; }
    for (int i = 0; i < ow; i++) { hc0[i] = c0[i]; hc1[i] = c1[i]; }
    auto put = [&](int *dst, const std::vector<float> &w) {
        for (size_t i = 0; i < w.size(); i++) {
            if (eight_bit) dst[i] = (int)std::nearbyint(w[i] * 2048.0f);        // exact product (power of two), round half to even
            else memcpy(&dst[i], &w[i], 4);
        }
    };
    put(hw, wr0); put(hw + oh, wr1); put(hw + 2 * oh, wc0); put(hw + 2 * oh + ow, wc1);
    // synchronous copy: `host` goes out of scope at return
    if (hipMemcpy(tab, host.data(), host.size() * 4, hipMemcpyHostToDevice) != hipSuccess) { set_error("resize: table upload failed"); return -2; }
    const int *dr0 = tab, *dr1 = dr0 + oh, *dc0 = dr1 + oh, *dc1 = dc0 + ow, *dw = dc1 + ow;
    if (eight_bit)
        hipLaunchKernelGGL(resize_linear_u8_kernel, grid, dim3(256), 0, s, din, H, W, oh, ow, dr0, dr1, dw, dw + oh, dc0, dc1, dw + 2 * oh, dw + 2 * oh + ow, dsm);
    else
        hipLaunchKernelGGL(resize_linear_u16_kernel, grid, dim3(256), 0, s, din, H, W, oh, ow, dr0, dr1, (const float *)dw, (const float *)(dw + oh), dc0, dc1,
                           (const float *)(dw + 2 * oh), (const float *)(dw + 2 * oh + ow), dsm);
    return 0;
}

}  // namespace tmat

using namespace tmat;

// masks (nullable): (n, oh, ow) u8 well masks on the DOWN-SAMPLED grid; small_out (nullable): the down-sampled images back to the host;
// area = kept pixels / pixels (the caller divides by the well area when it masks)
static int cell_area_impl(tmat_handle hd, const uint16_t *imgs, int n, int H, int W, int out_h, int out_w, double sd_coef, double *area,
                          uint8_t *thresholded, double *params, const uint8_t *masks, uint16_t *small_out, int fit)
{
    uint8_t *dmaskbuf = nullptr;
    int *lohibuf = nullptr;
    if (!hd || !imgs || !area || n < 0 || H < 1 || W < 1 || out_h < 0 || out_w < 0 || (out_h == 0) != (out_w == 0)) {
        set_error("tmat_cell_area_batch: bad argument");
        return TMAT_E_ARG;
    }
    if (n == 0) return TMAT_OK;
    Ctx *c = (Ctx *)hd;
    TMAT_HIP(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int oh = out_h ? out_h : H, ow = out_w ? out_w : W;
    const size_t nin = (size_t)n * H * W, nout = (size_t)n * oh * ow;
    uint16_t *din = nullptr, *dsm = nullptr;
    unsigned *hist = nullptr, *kept = nullptr;
    double *dpar = nullptr;
    uint8_t *dthr = nullptr;
    int *itab = nullptr;
    int rc = TMAT_OK;
    auto fail = [&](int code) { rc = code; };
    // workspaces live on the handle between calls (tmat_ctx.h:ws_get): slots 0..8 of this tool
    din = (uint16_t *)ws_get(c, 0, nin * 2); hist = (unsigned *)ws_get(c, 1, (size_t)n * 65536 * 4);
    kept = (unsigned *)ws_get(c, 2, (size_t)n * 4); dpar = (double *)ws_get(c, 3, (size_t)n * 11 * 8);
    if (thresholded) dthr = (uint8_t *)ws_get(c, 4, nout);
    if (!din || !hist || !kept || !dpar || (thresholded && !dthr)) fail(TMAT_E_HIP);
    if (!rc && !hip_ok(hipMemcpyAsync(din, imgs, nin * 2, hipMemcpyHostToDevice, s), "H2D")) fail(TMAT_E_HIP);
    const uint16_t *small = din;
    if (!rc && out_h) {
        dsm = (uint16_t *)ws_get(c, 5, nout * 2); itab = (int *)ws_get(c, 6, (size_t)(oh + ow) * 4 * 4);
        if (!dsm || !itab) fail(TMAT_E_HIP);
        // 8-bit sources (tmat_set_input_depth(h, 8)) take cv2's fixed-point arithmetic
        else if (launch_resize_linear_dev(din, n, H, W, oh, ow, c->input_sat == 255.f, itab, dsm, s)) fail(TMAT_E_HIP);
        else small = dsm;
    }
    if (!rc && !fit) {          // tmat_resize_linear_u16: only the down-sampled images are wanted
        if (!hip_ok(hipGetLastError(), "launch") || !hip_ok(hipMemcpyAsync(small_out, small, nout * 2, hipMemcpyDeviceToHost, s), "D2H") ||
            !hip_ok(hipStreamSynchronize(s), "sync")) { hipStreamSynchronize(s); fail(TMAT_E_HIP); }
    }
    if (!rc && fit) {
        const int npx = oh * ow, blocks = (npx + 255) / 256;
        const dim3 grid(blocks < 512 ? blocks : 512, n);
        if (!hip_ok(hipMemsetAsync(hist, 0, (size_t)n * 65536 * 4, s), "memset") || !hip_ok(hipMemsetAsync(kept, 0, n * 4, s), "memset")) fail(TMAT_E_HIP);
        else {
            if (small_out && !hip_ok(hipMemcpyAsync(small_out, small, nout * 2, hipMemcpyDeviceToHost, s), "D2H")) fail(TMAT_E_HIP);
            const uint8_t *dmask = nullptr;
            int *lohi = nullptr;
            if (!rc && masks) {
                std::vector<int> init((size_t)2 * n);
                for (int i = 0; i < n; i++) { init[2 * i] = 65536; init[2 * i + 1] = -1; }
                dmaskbuf = (uint8_t *)ws_get(c, 7, nout); lohibuf = (int *)ws_get(c, 8, (size_t)2 * n * 4);
                if (!dmaskbuf || !lohibuf ||
                    !hip_ok(hipMemcpyAsync(dmaskbuf, masks, nout, hipMemcpyHostToDevice, s), "H2D") ||
                    !hip_ok(hipMemcpy(lohibuf, init.data(), init.size() * 4, hipMemcpyHostToDevice), "H2D")) fail(TMAT_E_HIP);
                else {
                    dmask = dmaskbuf; lohi = lohibuf;
                    hipLaunchKernelGGL(lohi_u16_kernel, grid, dim3(256), 0, s, small, npx, lohi);
                }
            }
            hipLaunchKernelGGL(hist_u16_kernel, grid, dim3(256), 0, s, small, dmask, npx, hist);
            hipLaunchKernelGGL(gmm_hist_kernel, dim3(n), dim3(1024), 0, s, hist, sd_coef, dpar, lohi);
            hipLaunchKernelGGL(apply_threshold_kernel, grid, dim3(256), 0, s, small, dmask, npx, dpar, dthr, kept);
            std::vector<unsigned> kh(n);
            std::vector<double> ph((size_t)n * 11);
            if (!hip_ok(hipGetLastError(), "launch") || !hip_ok(hipMemcpyAsync(kh.data(), kept, n * 4, hipMemcpyDeviceToHost, s), "D2H") ||
                !hip_ok(hipMemcpyAsync(ph.data(), dpar, (size_t)n * 11 * 8, hipMemcpyDeviceToHost, s), "D2H") ||
                (thresholded && !hip_ok(hipMemcpyAsync(thresholded, dthr, nout, hipMemcpyDeviceToHost, s), "D2H")) ||
                !hip_ok(hipStreamSynchronize(s), "sync")) { hipStreamSynchronize(s); fail(TMAT_E_HIP); }      // nothing may still be copying into kh / ph when they go out of scope
            else {
                for (int i = 0; i < n; i++) area[i] = (double)kh[i] / (double)npx;          // compute_area_prop: np.sum(img > 0) / img.size
                if (params) for (int i = 0; i < n; i++) for (int k = 0; k < 9; k++) params[(size_t)i * 9 + k] = ph[(size_t)i * 11 + k];
            }
        }
    }
    if (rc) hipStreamSynchronize(s);      // nothing of a failed call stays in flight on the handle's workspaces
    return rc;
}

extern "C" int tmat_cell_area_batch(tmat_handle hd, const uint16_t *imgs, int n, int H, int W, int out_h, int out_w, double sd_coef, double *area,
                                    uint8_t *thresholded, double *params)
{
    return cell_area_impl(hd, imgs, n, H, W, out_h, out_w, sd_coef, area, thresholded, params, nullptr, nullptr, 1);
}

extern "C" int tmat_cell_area_masked(tmat_handle hd, const uint16_t *imgs, const uint8_t *masks, int n, int H, int W, double sd_coef, double *area,
                                     uint8_t *thresholded, double *params)
{
    if (!masks) { set_error("tmat_cell_area_masked: null masks"); return TMAT_E_ARG; }
    return cell_area_impl(hd, imgs, n, H, W, 0, 0, sd_coef, area, thresholded, params, masks, nullptr, 1);
}

extern "C" int tmat_resize_linear_u16(tmat_handle hd, const uint16_t *imgs, int n, int H, int W, int out_h, int out_w, uint16_t *out)
{
    if (!out || out_h < 1 || out_w < 1) { set_error("tmat_resize_linear_u16: bad argument"); return TMAT_E_ARG; }
    std::vector<double> area((size_t)(n > 0 ? n : 1));
    return cell_area_impl(hd, imgs, n, H, W, out_h, out_w, 0.0, area.data(), nullptr, nullptr, nullptr, out, 0);
}
