// Z-stack (Sato) branch of analyze_img on gfx950 (reference scripts/compute_branches.py:224-306), every pixel stage:
//   z1  per-slice gaussian(sigma 1, 'nearest') in f64 written back into the integer stack               :247-248
//   z2  skimage resize of the stack (gaussian 'mirror' over H, W; grid-mode bilinear zoom; clip)          :249-255
//   z3  rescale_intensity(0..1) over the whole stack -> f32                                               :256
//   z4  Sato tubeness of max(slice z, slice z+1) for 10 sigmas                                            :258-266
//   z5  unsharp_mask(volume, 2, 2), max projection                                                        :269-270
//   z6  canny(sigma=0): sobel, 4-sector non-maximum suppression, hysteresis                               :271
//   z7  medial_axis(edges) (thin_kernels.hip), eccentricity x equivalent diameter > 3.5 per component     :274-279
//   z8  3 masked blurs, 10 region-growing rounds, mask &= ~edges, closing(disk 2)                         :281-297
//   z9  filter_branch_seg_mask(mask, None, False) (morph_kernels.hip), dilation(square 3), gaussian        :299-302
// The arithmetic belongs to scipy.ndimage / scikit-image (third party); it is restated in oracle/sato.py, and every float
// expression here keeps that file's operation order (f64 accumulation tap by tap as scipy's correlate1d does, outputs rounded
// to the array dtype after every 1-D pass, f32 numpy expressions evaluated in f32 without contraction).
//
// Cost model: the only heavy stage is z4 -- per slice pair and sigma ten 1-D correlations of 2 r + 1 taps (r up to 85) in the
// gaussian-derivative form, 9340 taps per pixel over the ten sigmas: f64 adds/multiplies on L2-resident f32 planes, bound by
// vector f64 issue and the texture path, not by HBM.  Everything after z5 works on ONE 384-wide image: latency bound, tens of
// small launches.
#include "tmat_internal.h"
#include "morph.h"
#include "sato.h"

#include <algorithm>
#include <cmath>

namespace tmat {

static inline dim3 grid_for(size_t n) { const size_t b = (n + 255) / 256; return dim3((unsigned)(b < 16384 ? (b ? b : 1) : 16384)); }
#define FLAT_LOOP(p, n) for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < (n); p += (size_t)gridDim.x * blockDim.x)

__device__ __forceinline__ int ext_idx(int i, int n, int mode)
{
    if ((unsigned)i < (unsigned)n) return i;
    if (mode == EXT_NEAREST) return i < 0 ? 0 : n - 1;
    if (mode == EXT_REFLECT) { const int p = 2 * n; i %= p; if (i < 0) i += p; return i < n ? i : p - 1 - i; }      // d c b a | a b c d | d c b a
    if (n == 1) return 0;
    const int p = 2 * (n - 1); i %= p; if (i < 0) i += p; return i < n ? i : p - i;                                  // d c b | a b c d | c b a
}

// scipy.ndimage.correlate1d along the axis of length L and stride `inner` of a C-contiguous array seen as (outer, L, inner):
//   symmetric      t = x[l] w[c] + sum_{j=-r..-1} (x[l+j] + x[l-j]) w[c+j]
//   antisymmetric  t = x[l] w[c] + sum_{j=-r..-1} (x[l+j] - x[l-j]) w[c+j]
//   neither        t = x[l+r] w[c+r] + sum_{j=-r..r-1} x[l+j] w[c+j]        (ni_filters.c NI_Correlate1D, origin 0)
// accumulated in f64 in that order; the store rounds (float) or truncates (integer) like the NumPy output array does.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void corr1d_kernel(const TI *__restrict__ in, TO *__restrict__ out, size_t total, int L, int inner,
                                                     const double *__restrict__ w, int r, int sym, int mode)
{
    FLAT_LOOP(p, total) {
        const size_t oi = p / (size_t)inner;
        const int i = (int)(p - oi * inner);
        const size_t o = oi / (size_t)L;
        const int l = (int)(oi - o * L);
        const TI *line = in + o * (size_t)L * inner + i;
        double t;
        if (sym != 0) {
            t = (double)line[(size_t)l * inner] * w[r];
            if (l - r >= 0 && l + r < L) {
                if (sym > 0) for (int j = -r; j < 0; j++) t = t + ((double)line[(size_t)(l + j) * inner] + (double)line[(size_t)(l - j) * inner]) * w[r + j];
                else for (int j = -r; j < 0; j++) t = t + ((double)line[(size_t)(l + j) * inner] - (double)line[(size_t)(l - j) * inner]) * w[r + j];
            } else {
                for (int j = -r; j < 0; j++) {
                    const double v0 = (double)line[(size_t)ext_idx(l + j, L, mode) * inner], v1 = (double)line[(size_t)ext_idx(l - j, L, mode) * inner];
                    t = t + (sym > 0 ? v0 + v1 : v0 - v1) * w[r + j];
                }
            }
        } else {
            t = (double)line[(size_t)ext_idx(l + r, L, mode) * inner] * w[2 * r];
            for (int j = -r; j < r; j++) t = t + (double)line[(size_t)ext_idx(l + j, L, mode) * inner] * w[r + j];
        }
        out[p] = (TO)t;
    }
}

// The two fast forms for symmetric / antisymmetric kernels.  Both keep the accumulation order of every output exactly as above.
//
// Strided axis (inner > 1; lanes = neighbouring columns, so every load is one coalesced row segment): a thread owns CK
// consecutive outputs of its column and slides two register windows over the line -- x[l0 + k + j] and x[l0 + k - j],
// k < CK -- so each tap step costs two loads for CK outputs instead of 2 CK.  The windows are circular buffers whose slot
// arithmetic is static once the tap loop is unrolled by CK.
template <typename TI, typename TO, int CK>
__global__ __launch_bounds__(256) void corr1d_col_kernel(const TI *__restrict__ in, TO *__restrict__ out, size_t total_threads, int L, int inner, int groups,
                                                         const double *__restrict__ w, int r, int sym, int mode)
{
    FLAT_LOOP(p, total_threads) {
        const size_t og = p / (size_t)inner;
        const int i = (int)(p - og * inner);
        const size_t o = og / (size_t)groups;
        const int l0 = (int)(og - o * groups) * CK;
        const TI *line = in + o * (size_t)L * inner + i;
        TO *oline = out + o * (size_t)L * inner + i;
        auto at = [&](int l) { return (double)line[(size_t)ext_idx(l, L, mode) * inner]; };
        double t[CK], lo[CK], hi[CK];
#pragma unroll
        for (int k = 0; k < CK; k++) {
            t[k] = at(l0 + k) * w[r];                   // rows past the end are computed on extended data and not stored
            lo[k] = at(l0 + k - r);                     // slot (m + k) % CK at m = 0
            hi[k] = at(l0 + k + r);                     // slot (k - m) % CK at m = 0
        }
        int m = 0;
        for (; m + CK <= r; m += CK) {
#pragma unroll
            for (int u = 0; u < CK; u++) {
                const double wj = w[m + u];
#pragma unroll
                for (int k = 0; k < CK; k++) {
                    const double a = lo[(u + k) % CK], b = hi[(k - u + CK) % CK];
                    t[k] = t[k] + (sym > 0 ? a + b : a - b) * wj;
                }
                // step m + u -> m + u + 1: the low window gains x[l0 - r + (m + u) + CK], the high window x[l0 + r - (m + u) - 1]
                lo[u % CK] = at(l0 - r + m + u + CK);
                hi[(CK - 1 - u) % CK] = at(l0 + r - m - u - 1);
            }
        }
        for (; m < r; m++) {                            // fewer than CK taps left: plain loads
            const double wj = w[m];
#pragma unroll
            for (int k = 0; k < CK; k++) {
                const double a = at(l0 + k - r + m), b = at(l0 + k + r - m);
                t[k] = t[k] + (sym > 0 ? a + b : a - b) * wj;
            }
        }
#pragma unroll
        for (int k = 0; k < CK; k++) if (l0 + k < L) oline[(size_t)(l0 + k) * inner] = (TO)t[k];
    }
}

// Contiguous axis (inner == 1): a block stages one segment of a line plus its 2 r halo in LDS as f64 (boundary extension and
// the f32 -> f64 conversion happen once, at the fill), then every thread accumulates its outputs from LDS.
#define CORR_SEG 512
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void corr1d_row_kernel(const TI *__restrict__ in, TO *__restrict__ out, int L, int segs, const double *__restrict__ w, int r,
                                                         int sym, int mode)
{
    extern __shared__ double s_line[];
    const size_t lineno = blockIdx.x / segs;
    const int seg0 = (int)(blockIdx.x - lineno * segs) * CORR_SEG;
    const TI *line = in + lineno * (size_t)L;
    const int nseg = min(CORR_SEG, L - seg0);
    for (int q = threadIdx.x; q < nseg + 2 * r; q += 256) s_line[q] = (double)line[ext_idx(seg0 - r + q, L, mode)];
    __syncthreads();
    for (int q = threadIdx.x; q < nseg; q += 256) {
        const double *c = s_line + q + r;
        double t = c[0] * w[r];
        if (sym > 0) for (int j = -r; j < 0; j++) t = t + (c[j] + c[-j]) * w[r + j];
        else for (int j = -r; j < 0; j++) t = t + (c[j] - c[-j]) * w[r + j];
        out[lineno * (size_t)L + seg0 + q] = (TO)t;
    }
}

template <typename TI, typename TO>
static void launch_corr1d_t(const TI *in, TO *out, size_t outer, int L, int inner, const double *w, int r, int sym, int mode, hipStream_t s)
{
    const size_t total = outer * (size_t)L * inner;
    if (!total) return;
    if (sym != 0 && inner == 1 && (size_t)(CORR_SEG + 2 * r) * sizeof(double) <= 60 * 1024) {
        const int segs = (L + CORR_SEG - 1) / CORR_SEG;
        hipLaunchKernelGGL((corr1d_row_kernel<TI, TO>), dim3((unsigned)(outer * segs)), dim3(256), (size_t)(CORR_SEG + 2 * r) * sizeof(double), s, in, out, L,
                           segs, w, r, sym, mode);
        return;
    }
    if (sym != 0 && inner > 1 && r >= 4) {
        constexpr int CK = 8;
        const int groups = (L + CK - 1) / CK;
        const size_t threads = outer * (size_t)groups * inner;
        hipLaunchKernelGGL((corr1d_col_kernel<TI, TO, CK>), grid_for(threads), dim3(256), 0, s, in, out, threads, L, inner, groups, w, r, sym, mode);
        return;
    }
    hipLaunchKernelGGL((corr1d_kernel<TI, TO>), grid_for(total), dim3(256), 0, s, in, out, total, L, inner, w, r, sym, mode);
}
void launch_corr1d_f32(const float *in, float *out, size_t outer, int L, int inner, const double *w, int r, int sym, int mode, hipStream_t s) { launch_corr1d_t(in, out, outer, L, inner, w, r, sym, mode, s); }
void launch_corr1d_f64(const double *in, double *out, size_t outer, int L, int inner, const double *w, int r, int sym, int mode, hipStream_t s) { launch_corr1d_t(in, out, outer, L, inner, w, r, sym, mode, s); }
void launch_corr1d_u16_f64(const uint16_t *in, double *out, size_t outer, int L, int inner, const double *w, int r, int sym, int mode, hipStream_t s) { launch_corr1d_t(in, out, outer, L, inner, w, r, sym, mode, s); }
void launch_corr1d_f64_u16(const double *in, uint16_t *out, size_t outer, int L, int inner, const double *w, int r, int sym, int mode, hipStream_t s) { launch_corr1d_t(in, out, outer, L, inner, w, r, sym, mode, s); }

// ---- z2 / z3: bilinear zoom of Z slices, global min / max, clip, rescale ---------------------------------------------------
__device__ __forceinline__ unsigned long long f64_key(double v)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);          // order-preserving map of f64 onto u64
}
__device__ __forceinline__ double key_f64(unsigned long long k)
{
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}
__global__ void minmax_init_kernel(unsigned long long *mm) { mm[0] = ~0ull; mm[1] = 0ull; }
template <typename T>
__global__ __launch_bounds__(256) void minmax_all_kernel(const T *__restrict__ x, size_t n, unsigned long long *mm)
{
    unsigned long long lo = ~0ull, hi = 0ull;
    FLAT_LOOP(p, n) { const unsigned long long k = f64_key((double)x[p]); lo = k < lo ? k : lo; hi = k > hi ? k : hi; }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long l2 = __shfl_down(lo, o), h2 = __shfl_down(hi, o);
        lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi;
    }
    __shared__ unsigned long long slo[4], shi[4];
    if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {         // one pair of atomics per block (the grid is capped: a few thousand in all)
        for (int i = 1; i < 4; i++) { lo = slo[i] < lo ? slo[i] : lo; hi = shi[i] > hi ? shi[i] : hi; }
        atomicMin(&mm[0], lo); atomicMax(&mm[1], hi);
    }
}
__global__ void minmax_decode_kernel(const unsigned long long *mm, double *out) { out[0] = key_f64(mm[0]); out[1] = key_f64(mm[1]); }

// scipy zoom order 1 on every slice (the Z axis has zoom 1: its taps are (1, 0) and add exact zeros), clipped to lohi
__global__ void zoom_stack_kernel(const double *__restrict__ a, int H, int W, int oh, int ow, const int *__restrict__ r0,
                                  const int *__restrict__ r1, const double *__restrict__ wr0, const double *__restrict__ wr1,
                                  const int *__restrict__ c0, const int *__restrict__ c1, const double *__restrict__ wc0,
                                  const double *__restrict__ wc1, const double *__restrict__ lohi, double *__restrict__ out, size_t total)
{
    const double l = lohi[0], h = lohi[1];
    const size_t onpx = (size_t)oh * ow;
    FLAT_LOOP(p, total) {
        const size_t z = p / onpx;
        const int q = (int)(p - z * onpx);
        const int y = q / ow, x = q - y * ow;
        const double *src = a + z * (size_t)H * W;
        double t = (src[(size_t)r0[y] * W + c0[x]] * wr0[y]) * wc0[x];
        t = t + (src[(size_t)r0[y] * W + c1[x]] * wr0[y]) * wc1[x];
        t = t + (src[(size_t)r1[y] * W + c0[x]] * wr1[y]) * wc0[x];
        t = t + (src[(size_t)r1[y] * W + c1[x]] * wr1[y]) * wc1[x];
        out[p] = fmin(fmax(t, l), h);
    }
}
// rescale_intensity(out_range=(0, 1)) of an f64 array, then astype(f32)
__global__ void rescale01_f64_kernel(const double *__restrict__ x, size_t n, const double *__restrict__ lohi, float *__restrict__ out)
{
    const double lo = lohi[0], hi = lohi[1];
    const double d = hi - lo;
    FLAT_LOOP(p, n) {
        double v = fmin(fmax(x[p], lo), hi);
        v = lo != hi ? ((v - lo) / d) * 1.0 + 0.0 : fmin(fmax(v, 0.0), 1.0);
        out[p] = (float)v;
    }
}

int stack_zoom_rescale_dev(const double *filtered, const uint16_t *stack_after_gauss, int Z, int H, int W, int oh, int ow, const int *r0,
                           const int *r1, const double *wr0, const double *wr1, const int *c0, const int *c1, const double *wc0,
                           const double *wc1, double *zoomed, unsigned long long *mm, double *lohi, float *vol, hipStream_t s)
{
    const size_t nin = (size_t)Z * H * W, nout = (size_t)Z * oh * ow;
    hipLaunchKernelGGL(minmax_init_kernel, dim3(1), dim3(1), 0, s, mm);
    const dim3 rgrid_in((unsigned)std::min<size_t>(2048, (nin + 255) / 256)), rgrid_out((unsigned)std::min<size_t>(2048, (nout + 255) / 256));
    hipLaunchKernelGGL((minmax_all_kernel<uint16_t>), rgrid_in, dim3(256), 0, s, stack_after_gauss, nin, mm);
    hipLaunchKernelGGL(minmax_decode_kernel, dim3(1), dim3(1), 0, s, mm, lohi);
    hipLaunchKernelGGL(zoom_stack_kernel, grid_for(nout), dim3(256), 0, s, filtered, H, W, oh, ow, r0, r1, wr0, wr1, c0, c1, wc0, wc1, lohi, zoomed, nout);
    if (vol) {                // vol == nullptr: the caller wants the resized values themselves (tmat_resize_aa_u16)
        hipLaunchKernelGGL(minmax_init_kernel, dim3(1), dim3(1), 0, s, mm);
        hipLaunchKernelGGL((minmax_all_kernel<double>), rgrid_out, dim3(256), 0, s, zoomed, nout, mm);
        hipLaunchKernelGGL(minmax_decode_kernel, dim3(1), dim3(1), 0, s, mm, lohi + 2);
        hipLaunchKernelGGL(rescale01_f64_kernel, grid_for(nout), dim3(256), 0, s, zoomed, nout, lohi + 2, vol);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---- z4: Sato ----------------------------------------------------------------------------------------------------------------
// im = max(vol[z], vol[z+1]); gradient form (scikit-image <= 0.19): invert = 1 - im; gaussian-derivative form (>= 0.20): -im
__global__ void pairmax_kernel(const float *__restrict__ vol, size_t npx, size_t total, int negate, float *__restrict__ out)
{
    FLAT_LOOP(p, total) {
        const float m = fmaxf(vol[p], vol[p + npx]);
        out[p] = negate ? -m : 1.0f - m;
    }
}
__global__ void prep_single_kernel(const float *__restrict__ img, size_t total, int negate, float *__restrict__ out)
{
    FLAT_LOOP(p, total) out[p] = negate ? -img[p] : 1.0f - img[p];
}
// np.gradient (unit spacing, edge_order 1) of f32 planes along the axis of length L / stride inner
__global__ void gradient_kernel(const float *__restrict__ f, float *__restrict__ out, size_t total, int L, int inner)
{
    FLAT_LOOP(p, total) {
        const size_t oi = p / (size_t)inner;
        const int l = (int)(oi % (size_t)L);
        float v;
        if (l == 0) v = (f[p + inner] - f[p]) / 1.0f;
        else if (l == L - 1) v = (f[p] - f[p - inner]) / 1.0f;
        else v = (f[p + inner] - f[p - inner]) / 2.0f;
        out[p] = v;
    }
}
// scikit-image 0.18 (_image_orthogonal_matrix22_eigvals on sigma^2-scaled elements): l1 = (a + d) / 2 + sqrt(4 b^2 + (a - d)^2) / 2;
// best = max(best, l1 > 0 ? |l1| : 0)
__global__ void eig_gradient_kernel(const float *__restrict__ hrr, const float *__restrict__ hrc, const float *__restrict__ hcc, float s2,
                                    size_t total, float *__restrict__ best, int first)
{
    FLAT_LOOP(p, total) {
        const float m00 = s2 * hrr[p], m01 = s2 * hrc[p], m11 = s2 * hcc[p];
        const float q = m00 - m11;
        const float l1 = (m00 + m11) / 2.0f + sqrtf(4.0f * (m01 * m01) + q * q) / 2.0f;
        const float v = l1 > 0.0f ? fabsf(l1) : 0.0f;
        const float b = first ? 0.0f : best[p];
        best[p] = fmaxf(b, v);
    }
}
// scikit-image >= 0.19 (_symmetric_compute_eigenvalues): l1 = (a + d) / 2 + sqrt(b^2 + ((a - d) / 2)^2); vesselness = sigma^2 * max(l1, 0)
__global__ void eig_derivative_kernel(const float *__restrict__ hrr, const float *__restrict__ hrc, const float *__restrict__ hcc, float s2,
                                      size_t total, float *__restrict__ best, int first)
{
    FLAT_LOOP(p, total) {
        const float m00 = hrr[p], m01 = hrc[p], m11 = hcc[p];
        const float hd = (m00 - m11) / 2.0f;
        const float l1 = (m00 + m11) / 2.0f + sqrtf(m01 * m01 + hd * hd);
        const float v = s2 * fmaxf(l1, 0.0f);
        const float b = first ? 0.0f : best[p];
        best[p] = fmaxf(b, v);
    }
}
void launch_pairmax(const float *vol, int Zm1, size_t npx, int negate, float *out, hipStream_t s)
{
    const size_t total = (size_t)Zm1 * npx;
    hipLaunchKernelGGL(pairmax_kernel, grid_for(total), dim3(256), 0, s, vol, npx, total, negate, out);
}
void launch_prep_single(const float *img, size_t total, int negate, float *out, hipStream_t s)
{
    hipLaunchKernelGGL(prep_single_kernel, grid_for(total), dim3(256), 0, s, img, total, negate, out);
}
void launch_gradient(const float *f, float *out, size_t outer, int L, int inner, hipStream_t s)
{
    const size_t total = outer * (size_t)L * inner;
    hipLaunchKernelGGL(gradient_kernel, grid_for(total), dim3(256), 0, s, f, out, total, L, inner);
}
void launch_eig(int derivative_form, const float *hrr, const float *hrc, const float *hcc, float s2, size_t total, float *best, int first, hipStream_t s)
{
    if (derivative_form) hipLaunchKernelGGL(eig_derivative_kernel, grid_for(total), dim3(256), 0, s, hrr, hrc, hcc, s2, total, best, first);
    else hipLaunchKernelGGL(eig_gradient_kernel, grid_for(total), dim3(256), 0, s, hrr, hrc, hcc, s2, total, best, first);
}

// ---- z5: unsharp mask + max projection ------------------------------------------------------------------------------------
__global__ void unsharp_kernel(const float *__restrict__ v, const float *__restrict__ blurred, float amount, size_t total, float *__restrict__ out)
{
    FLAT_LOOP(p, total) {
        const float r = v[p] + (v[p] - blurred[p]) * amount;
        out[p] = fminf(fmaxf(r, 0.0f), 1.0f);
    }
}
__global__ void zmax_kernel(const float *__restrict__ v, int Z, size_t npx, float *__restrict__ out)
{
    FLAT_LOOP(p, npx) {
        float m = v[p];
        for (int z = 1; z < Z; z++) m = fmaxf(m, v[(size_t)z * npx + p]);
        out[p] = m;
    }
}
void launch_unsharp(const float *v, const float *blurred, float amount, size_t total, float *out, hipStream_t s)
{
    hipLaunchKernelGGL(unsharp_kernel, grid_for(total), dim3(256), 0, s, v, blurred, amount, total, out);
}
void launch_zmax(const float *v, int Z, size_t npx, float *out, hipStream_t s)
{
    hipLaunchKernelGGL(zmax_kernel, grid_for(npx), dim3(256), 0, s, v, Z, npx, out);
}

// ---- z6: canny(sigma = 0) -------------------------------------------------------------------------------------------------
// smoothed = image / (ones + eps) in f64 (feature/_canny.py: the bleed-over normalisation of an all-ones mask)
__global__ void canny_norm_kernel(const float *__restrict__ img, size_t n, double *__restrict__ out)
{
    FLAT_LOOP(p, n) out[p] = (double)img[p] / (1.0 + 2.220446049250313e-16);
}
// sqrt(x^2 + y^2) rounded from a double-double sum of squares (np.hypot = libm hypot: correctly rounded in all but freak cases)
__device__ __forceinline__ double hypot_dd(double x, double y)
{
    x = fabs(x); y = fabs(y);
    if (x == 0.0 && y == 0.0) return 0.0;
    const double px = x * x, ex = __fma_rn(x, x, -px), py = y * y, ey = __fma_rn(y, y, -py);
    const double sh = px + py;
    const double bb = sh - px;
    const double se = ((px - (sh - bb)) + (py - bb)) + (ex + ey);
    const double h = sqrt(sh);
    const double res = __fma_rn(-h, h, sh) + se;              // (sh + se) - h^2
    return h + res / (2.0 * h);
}
__global__ void canny_nms_kernel(const double *__restrict__ isob, const double *__restrict__ jsob, int H, int W, double *__restrict__ mag_out,
                                 uint8_t *__restrict__ low, uint8_t *__restrict__ high)
{
    const int npx = H * W;
    FLAT_LOOP(p, (size_t)npx) {
        const int y = (int)(p / W), x = (int)(p - (size_t)y * W);
        const double is_ = isob[p], js = jsob[p];
        const double m = hypot_dd(is_, js);
        mag_out[p] = m;
        bool lm = false;
        if (y > 0 && y < H - 1 && x > 0 && x < W - 1 && m > 0.0) {
            const double ai = fabs(is_), aj = fabs(js);
            auto mg = [&](int dy, int dx) { const size_t q = p + (ptrdiff_t)dy * W + dx; return hypot_dd(isob[q], jsob[q]); };
            const bool pp = is_ >= 0 && js >= 0, mm = is_ <= 0 && js <= 0, mp = is_ <= 0 && js >= 0, pm = is_ >= 0 && js <= 0;
            // the four sectors in the library's order; a pixel on a sector boundary takes the LAST sector that claims it
            if ((pp || mm) && ai >= aj) {                                   // 0 .. 45 degrees
                const double w = aj / ai;
                lm = (mg(1, 1) * w + mg(1, 0) * (1 - w) <= m) && (mg(-1, -1) * w + mg(-1, 0) * (1 - w) <= m);
            }
            if ((pp || mm) && ai <= aj) {                                   // 45 .. 90
                const double w = ai / aj;
                lm = (mg(1, 1) * w + mg(0, 1) * (1 - w) <= m) && (mg(-1, -1) * w + mg(0, -1) * (1 - w) <= m);
            }
            if ((mp || pm) && ai <= aj) {                                   // 90 .. 135
                const double w = ai / aj;
                lm = (mg(-1, 1) * w + mg(0, 1) * (1.0 - w) <= m) && (mg(1, -1) * w + mg(0, -1) * (1.0 - w) <= m);
            }
            if ((mp || pm) && ai >= aj) {                                   // 135 .. 180
                const double w = aj / ai;
                lm = (mg(-1, 1) * w + mg(-1, 0) * (1 - w) <= m) && (mg(1, -1) * w + mg(1, 0) * (1 - w) <= m);
            }
        }
        low[p] = lm && m >= 0.1;
        high[p] = lm && m >= 0.2;
    }
}
// hysteresis: components of `low` (labels L) that hold a `high` pixel
__global__ void flag_roots_kernel(const uint8_t *__restrict__ on, const int *__restrict__ L, int npx, int *__restrict__ flag)
{
    FLAT_LOOP(p, (size_t)npx) if (on[p] && L[p] >= 0) flag[L[p]] = 1;
}
__global__ void keep_flagged_kernel(const int *__restrict__ L, const int *__restrict__ flag, int npx, uint8_t *__restrict__ out)
{
    FLAT_LOOP(p, (size_t)npx) out[p] = L[p] >= 0 && flag[L[p]];
}

int canny0_dev(const float *img, int H, int W, const CannyWs &ws, uint8_t *edges, hipStream_t s)
{
    const size_t n = (size_t)H * W;
    hipLaunchKernelGGL(canny_norm_kernel, grid_for(n), dim3(256), 0, s, img, n, ws.sm);
    return canny_core_dev(H, W, ws, edges, s);
}

// feature/_canny.py after the smoothing step, from the smoothed image in ws.sm (f64)
int canny_core_dev(int H, int W, const CannyWs &ws, uint8_t *edges, hipStream_t s)
{
    const size_t n = (size_t)H * W;
    // ndi.sobel(axis): correlate1d [-1, 0, 1] along the axis, then [1, 2, 1] along the other; 'reflect'
    launch_corr1d_f64(ws.sm, ws.t0, (size_t)H, W, 1, ws.w_diff, 1, -1, EXT_REFLECT, s);       // jsobel: axis 1
    launch_corr1d_f64(ws.t0, ws.js, 1, H, W, ws.w_smooth, 1, 1, EXT_REFLECT, s);
    launch_corr1d_f64(ws.sm, ws.t0, 1, H, W, ws.w_diff, 1, -1, EXT_REFLECT, s);       // isobel: axis 0
    launch_corr1d_f64(ws.t0, ws.is_, (size_t)H, W, 1, ws.w_smooth, 1, 1, EXT_REFLECT, s);
    hipLaunchKernelGGL(canny_nms_kernel, grid_for(n), dim3(256), 0, s, ws.is_, ws.js, H, W, ws.mag, ws.low, ws.high);
    launch_ccl(ws.low, 1, H, W, ws.L, s);
    if (hipMemsetAsync(ws.flag, 0, n * sizeof(int), s) != hipSuccess) return -2;
    hipLaunchKernelGGL(flag_roots_kernel, grid_for(n), dim3(256), 0, s, ws.high, ws.L, (int)n, ws.flag);
    hipLaunchKernelGGL(keep_flagged_kernel, grid_for(n), dim3(256), 0, s, ws.L, ws.flag, (int)n, edges);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---- z7: eccentricity x equivalent diameter per 8-connected component ------------------------------------------------------
// integer moments per root label: [area, sum r, sum c, sum r^2, sum c^2, sum r c]
__global__ void moments_kernel(const int *__restrict__ L, int W, int npx, unsigned long long *__restrict__ mom)
{
    FLAT_LOOP(p, (size_t)npx) {
        const int l = L[p];
        if (l < 0) continue;
        const unsigned long long y = p / W, x = p - y * W;
        unsigned long long *m = mom + (size_t)l * 6;
        atomicAdd(&m[0], 1ull); atomicAdd(&m[1], y); atomicAdd(&m[2], x);
        atomicAdd(&m[3], y * y); atomicAdd(&m[4], x * x); atomicAdd(&m[5], y * x);
    }
}
// regionprops: inertia tensor [[mu02, -mu11], [-mu11, mu20]] / area, eccentricity sqrt(1 - l2 / l1), equivalent diameter
// sqrt(4 area / pi).  The central moments come from exact integer sums (u64 arithmetic; area * sum r^2 stays below 2^53 for any
// component of a 384-wide image that is not most of the frame), so each is the correctly rounded quotient.
__global__ void ecc_diam_kernel(const int *__restrict__ L, const unsigned long long *__restrict__ mom, int npx, double thresh,
                                const uint8_t *__restrict__ mask, uint8_t *__restrict__ out, double *__restrict__ val_out)
{
    FLAT_LOOP(p, (size_t)npx) {
        const int l = L[p];
        double v = 0.0;
        if (l >= 0) {
            const unsigned long long *m = mom + (size_t)l * 6;
            const double A = (double)m[0];
            const double mu20 = (double)(long long)(m[0] * m[3] - m[1] * m[1]) / A;
            const double mu02 = (double)(long long)(m[0] * m[4] - m[2] * m[2]) / A;
            const double mu11 = (double)((long long)(m[0] * m[5]) - (long long)(m[1] * m[2])) / A;
            const double a = mu02 / A, d = mu20 / A, b = -mu11 / A;
            const double mid = (a + d) / 2.0, rad = sqrt(((a - d) / 2.0) * ((a - d) / 2.0) + b * b);
            const double l1 = fmax(mid + rad, 0.0), l2 = fmax(mid - rad, 0.0);
            const double ecc = l1 == 0.0 ? 0.0 : sqrt(1.0 - l2 / l1);
            const double diam = sqrt(4.0 * A / 3.141592653589793);
            v = ecc * diam;
        }
        if (val_out) val_out[p] = v;
        out[p] = mask[p] && v > thresh;
    }
}
int ecc_diam_select_dev(const uint8_t *mask, int H, int W, double thresh, int *L, unsigned long long *mom, uint8_t *out, double *val_out, hipStream_t s)
{
    const size_t n = (size_t)H * W;
    launch_ccl(mask, 1, H, W, L, s);
    if (hipMemsetAsync(mom, 0, n * 6 * sizeof(unsigned long long), s) != hipSuccess) return -2;
    hipLaunchKernelGGL(moments_kernel, grid_for(n), dim3(256), 0, s, L, W, (int)n, mom);
    hipLaunchKernelGGL(ecc_diam_kernel, grid_for(n), dim3(256), 0, s, L, mom, (int)n, thresh, mask, out, val_out);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---- z8: masked blur, region growing, binary morphology ------------------------------------------------------------------
__global__ void where_f32_kernel(const uint8_t *__restrict__ m, const float *__restrict__ a, const float *__restrict__ b, size_t n, float *__restrict__ out)
{
    FLAT_LOOP(p, n) out[p] = m[p] ? a[p] : b[p];
}
__global__ void where_zero_kernel(const uint8_t *__restrict__ m, const float *__restrict__ a, size_t n, float *__restrict__ out)
{
    FLAT_LOOP(p, n) out[p] = m[p] ? a[p] : 0.0f;
}
// one round of compute_branches.py:283-294: a pixel joins when some mask neighbour is not brighter than it, none is brighter,
// and it is brighter than 0.01; all tests read the mask of the round's start
__global__ void grow_kernel(const uint8_t *__restrict__ m, const float *__restrict__ v, int H, int W, uint8_t *__restrict__ out)
{
    FLAT_LOOP(p, (size_t)H * W) {
        const int y = (int)(p / W), x = (int)(p - (size_t)y * W);
        bool lo = false, hi = false;
        const float c = v[p];
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                if (!dy && !dx) continue;
                const int yy = y + dy, xx = x + dx;
                if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                const size_t q = (size_t)yy * W + xx;
                if (!m[q]) continue;
                if (c < v[q]) lo = true; else hi = true;
            }
        out[p] = m[p] || (c > 0.01f && hi && !lo);
    }
}
__global__ void andnot_kernel(const uint8_t *__restrict__ a, const uint8_t *__restrict__ b, size_t n, uint8_t *__restrict__ out)
{
    FLAT_LOOP(p, n) out[p] = a[p] && !b[p];
}
// binary dilation (any) / erosion (all) over a footprint given as (dy, dx) offsets, borders by reflection (ndi.grey_dilation /
// grey_erosion default mode)
__global__ void morph_kernel(const uint8_t *__restrict__ m, int H, int W, const int *__restrict__ off, int noff, int erode, uint8_t *__restrict__ out)
{
    FLAT_LOOP(p, (size_t)H * W) {
        const int y = (int)(p / W), x = (int)(p - (size_t)y * W);
        bool acc = erode != 0;
        for (int k = 0; k < noff; k++) {
            const int yy = ext_idx(y + off[2 * k], H, EXT_REFLECT), xx = ext_idx(x + off[2 * k + 1], W, EXT_REFLECT);
            const bool b = m[(size_t)yy * W + xx] != 0;
            acc = erode ? (acc && b) : (acc || b);
        }
        out[p] = acc;
    }
}
void launch_where(const uint8_t *m, const float *a, const float *b, size_t n, float *out, hipStream_t s)
{
    if (b) hipLaunchKernelGGL(where_f32_kernel, grid_for(n), dim3(256), 0, s, m, a, b, n, out);
    else hipLaunchKernelGGL(where_zero_kernel, grid_for(n), dim3(256), 0, s, m, a, n, out);
}
void launch_grow(const uint8_t *m, const float *v, int H, int W, uint8_t *out, hipStream_t s)
{
    hipLaunchKernelGGL(grow_kernel, grid_for((size_t)H * W), dim3(256), 0, s, m, v, H, W, out);
}
void launch_andnot(const uint8_t *a, const uint8_t *b, size_t n, uint8_t *out, hipStream_t s)
{
    hipLaunchKernelGGL(andnot_kernel, grid_for(n), dim3(256), 0, s, a, b, n, out);
}
void launch_morph(const uint8_t *m, int H, int W, const int *off_dev, int noff, int erode, uint8_t *out, hipStream_t s)
{
    hipLaunchKernelGGL(morph_kernel, grid_for((size_t)H * W), dim3(256), 0, s, m, H, W, off_dev, noff, erode, out);
}

}  // namespace tmat
