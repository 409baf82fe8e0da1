// Vesselness-field stages after the (host, sequential) medial-axis thinning, on gfx950, batch per launch:
//   a15  centerline_dt = distance_transform_edt(~skel); relative_dt = dist / (dist + centerline_dt);
//        pred *= relative_dt                                                   compute_branches.py:341-344
//   a16  skimage resize(order=1, preserve_range, anti_aliasing): gaussian ('mirror', sigma = (f-1)/2, truncate 4)
//        along axis 0 then 1, bilinear zoom (grid_mode, 'mirror'), clip to the input range, -> f32   :351-357
//   a17  rescale_intensity(out_range=(0, 255)) in float32                                          :419
// HBM-bound f64 streaming kernels; every expression keeps the operation order of oracle/morph.py (resize_aa,
// rescale_intensity) so results are bit-identical to scipy / the host twin in postproc.cpp.
#include "tmat_internal.h"
#include "morph.h"
#include "postproc.h"
#include "tmat_ctx.h"

#include <cmath>

namespace tmat {

#define IMG_LOOP(p, n_px) for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < (n_px); p += gridDim.x * blockDim.x)

__global__ void invert_u8_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int npx)
{
    const size_t base = (size_t)blockIdx.y * npx;
    IMG_LOOP(p, npx) out[base + p] = !in[base + p];
}

__global__ void weight_kernel(const double *__restrict__ pred, const double *__restrict__ dist, const double *__restrict__ cdt,
                              double *__restrict__ wt, int npx)
{
    const size_t base = (size_t)blockIdx.y * npx;
    IMG_LOOP(p, npx) {
        const double d = dist[base + p];
        wt[base + p] = pred[base + p] * (d / (d + cdt[base + p]));
    }
}

// per-image min / max (one block per image)
template <typename T>
__global__ __launch_bounds__(256) void minmax_kernel(const T *__restrict__ x, size_t per, T *mn, T *mx)
{
    const T *p = x + (size_t)blockIdx.x * per;
    T lo = p[0], hi = p[0];
    for (size_t i = threadIdx.x; i < per; i += 256) { const T v = p[i]; lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
    for (int o = 32; o > 0; o >>= 1) {
        const T l2 = __shfl_down(lo, o), h2 = __shfl_down(hi, o);
        lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi;
    }
    __shared__ T slo[4], shi[4];
    if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; i++) { lo = slo[i] < lo ? slo[i] : lo; hi = shi[i] > hi ? shi[i] : hi; }
        mn[blockIdx.x] = lo; mx[blockIdx.x] = hi;
    }
}

__device__ __forceinline__ int mirror_idx(int i, int n)
{
    if (n == 1) return 0;
    const int p = 2 * (n - 1);
    i %= p; if (i < 0) i += p;
    return i < n ? i : p - i;
}

// scipy correlate1d, symmetric odd kernel w[0..2r], 'mirror':  x[l]*w[r] + sum_{j=-r..-1} (x[l+j] + x[l-j]) * w[r+j]
__global__ void gauss_axis_kernel(const double *__restrict__ a, double *__restrict__ out, int H, int W, int axis,
                                  const double *__restrict__ w, int r)
{
    const size_t base = (size_t)blockIdx.y * H * W;
    const int n = axis == 0 ? H : W;
    IMG_LOOP(p, H * W) {
        const int y = p / W, x = p - y * W;
        const int l = axis == 0 ? y : x;
        double t = a[base + p] * w[r];
        for (int j = -r; j < 0; j++) {
            const int i0 = mirror_idx(l + j, n), i1 = mirror_idx(l - j, n);
            const double v0 = axis == 0 ? a[base + (size_t)i0 * W + x] : a[base + (size_t)y * W + i0];
            const double v1 = axis == 0 ? a[base + (size_t)i1 * W + x] : a[base + (size_t)y * W + i1];
            t = t + (v0 + v1) * w[r + j];
        }
        out[base + p] = t;
    }
}

// scipy zoom order 1 (NI_ZoomShift) + clip to [lo, hi] + cast to f32; tables built on the host
__global__ void zoom_clip_kernel(const double *__restrict__ a, int H, int W, int oh, int ow, const int *__restrict__ r0,
                                 const int *__restrict__ r1, const double *__restrict__ wr0, const double *__restrict__ wr1,
                                 const int *__restrict__ c0, const int *__restrict__ c1, const double *__restrict__ wc0,
                                 const double *__restrict__ wc1, const double *__restrict__ lo, const double *__restrict__ hi,
                                 float *__restrict__ out)
{
    const int img = blockIdx.y;
    const double *src = a + (size_t)img * H * W;
    const double l = lo[img], h = hi[img];
    IMG_LOOP(p, oh * ow) {
        const int y = p / ow, x = p - y * ow;
        double t = (src[(size_t)r0[y] * W + c0[x]] * wr0[y]) * wc0[x];
        t = t + (src[(size_t)r0[y] * W + c1[x]] * wr0[y]) * wc1[x];
        t = t + (src[(size_t)r1[y] * W + c0[x]] * wr1[y]) * wc0[x];
        t = t + (src[(size_t)r1[y] * W + c1[x]] * wr1[y]) * wc1[x];
        t = fmin(fmax(t, l), h);
        out[(size_t)img * oh * ow + p] = (float)t;
    }
}

// skimage rescale_intensity(out_range=(0,255)) on a float32 image: ((x - min) / f32(max - min)) * 255 + 0
__global__ void rescale255_kernel(const float *__restrict__ x, int npx, const float *__restrict__ mn, const float *__restrict__ mx,
                                  float *__restrict__ out)
{
    const int img = blockIdx.y;
    const float lo = mn[img], hi = mx[img];
    const float d = (float)((double)hi - (double)lo);
    const size_t base = (size_t)img * npx;
    IMG_LOOP(p, npx) {
        const float v = x[base + p];
        out[base + p] = lo != hi ? ((v - lo) / d) * 255.0f + 0.0f : fminf(fmaxf(v, 0.0f), 255.0f);
    }
}

size_t finish_workspace_bytes(int k, int H, int W, int oh, int ow)
{
    const size_t n = (size_t)k * H * W;
    // nskel (u8) | g (int) | st (2 int) | cdt, wt, tmp (f64) | field (f32 out-size) | per-image scalars | tables
    return n + n * sizeof(int) * 3 + n * sizeof(double) * 3 + (size_t)k * oh * ow * sizeof(float) + (size_t)k * 64 +
           (size_t)(oh + ow) * 2 * (sizeof(int) + sizeof(double)) + 4096 * sizeof(double) + 8192;
}

// scipy zoom(order=1, mode='mirror', grid_mode=True) along one axis: source coordinate (j + 0.5) * (in/out) - 0.5, the two taps
// and their weights (1 - t, 1 - (1 - t))
void zoom_axis_table(int n_in, int n_out, std::vector<int> &i0, std::vector<int> &i1, std::vector<double> &a0, std::vector<double> &a1)
{
    auto mir = [](long i, int n) { if (n == 1) return 0; const long p = 2L * (n - 1); i %= p; if (i < 0) i += p; return (int)(i < n ? i : p - i); };
    const double zoom = (double)n_in / (double)n_out;
    i0.resize(n_out); i1.resize(n_out); a0.resize(n_out); a1.resize(n_out);
    for (int j = 0; j < n_out; j++) {
        const double cc = ((double)j + 0.5) * zoom - 0.5;
        const double fl = std::floor(cc), tt = cc - fl;
        a0[j] = 1.0 - tt; a1[j] = 1.0 - a0[j];
        i0[j] = mir((long)fl, n_in); i1[j] = mir((long)fl + 1, n_in);
    }
}
void launch_rescale255(const float *field, int k, int npx, float *mn, float *mx, float *out, hipStream_t s)
{
    const dim3 grid((npx + 255) / 256 < 1024 ? (npx + 255) / 256 : 1024, k);
    hipLaunchKernelGGL((minmax_kernel<float>), dim3(k), dim3(256), 0, s, field, (size_t)npx, mn, mx);
    hipLaunchKernelGGL(rescale255_kernel, grid, dim3(256), 0, s, field, npx, mn, mx, out);
}

// host-built tables, cached per geometry
struct FinishTables {
    int H = 0, W = 0, oh = 0, ow = 0;
    std::vector<double> w0, w1;           // gaussian kernels of axis 0 / 1 (empty: sigma == 0)
    std::vector<int> r0, r1, c0, c1;
    std::vector<double> wr0, wr1, wc0, wc1;
};
static void build_tables(FinishTables &t, int H, int W, int oh, int ow)
{
    t.H = H; t.W = W; t.oh = oh; t.ow = ow;
    auto gk = [](double sigma, std::vector<double> &w) {
        w.clear();
        if (!(sigma > 0)) return;
        const int r = (int)(4.0 * sigma + 0.5);
        w.resize(2 * r + 1);
        const double s2 = sigma * sigma;
        double tot = 0.0;
        for (int x = -r; x <= r; x++) w[x + r] = std::exp(-0.5 / s2 * (double)(x * x));
        tot = numpy_pairwise_sum(w.data(), (long)w.size());       // phi_x.sum() in scipy's _gaussian_kernel1d
        for (double &v : w) v = v / tot;
    };
    const double f0 = (double)H / (double)oh, f1 = (double)W / (double)ow;
    gk(std::max(0.0, (f0 - 1) / 2), t.w0);
    gk(std::max(0.0, (f1 - 1) / 2), t.w1);
    zoom_axis_table(H, oh, t.r0, t.r1, t.wr0, t.wr1);
    zoom_axis_table(W, ow, t.c0, t.c1, t.wc0, t.wc1);
}

// pred, dist (k, H, W) f64 device; skel (k, H, W) u8 device -> field (k, oh, ow) f32 (before a17) and f255 (after), device
int finish_dev(const double *pred, const double *dist, const uint8_t *skel, int k, int H, int W, int oh, int ow, void *workspace,
               float *field_out, float *f255_out, hipStream_t s)
{
    static thread_local FinishTables tab;
    if (tab.H != H || tab.W != W || tab.oh != oh || tab.ow != ow) build_tables(tab, H, W, oh, ow);
    if (tab.w0.size() > 2048 || tab.w1.size() > 2048) { set_error("finish: gaussian kernel too long"); return -1; }
    const int npx = H * W;
    const size_t n = (size_t)k * npx;
    uint8_t *nskel = (uint8_t *)workspace;
    int *g = (int *)(((uintptr_t)(nskel + n) + 15) & ~(uintptr_t)15);
    int *st = g + n;
    double *cdt = (double *)(((uintptr_t)(st + 2 * n) + 15) & ~(uintptr_t)15);
    double *wt = cdt + n, *tmp = wt + n;
    double *lo = tmp + n, *hi = lo + k;
    float *fmn = (float *)(hi + k), *fmx = fmn + k;
    int *anyz = (int *)(fmx + k);
    double *gw0 = (double *)(((uintptr_t)(anyz + k) + 15) & ~(uintptr_t)15), *gw1 = gw0 + 2048;
    double *dwr0 = gw1 + 2048, *dwr1 = dwr0 + oh, *dwc0 = dwr1 + oh, *dwc1 = dwc0 + ow;
    int *dr0 = (int *)(dwc1 + ow), *dr1 = dr0 + oh, *dc0 = dr1 + oh, *dc1 = dc0 + ow;
    auto up = [&](void *d, const void *h, size_t bytes) { return bytes == 0 || hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, s) == hipSuccess; };
    if (!up(gw0, tab.w0.data(), tab.w0.size() * 8) || !up(gw1, tab.w1.data(), tab.w1.size() * 8) || !up(dwr0, tab.wr0.data(), oh * 8) ||
        !up(dwr1, tab.wr1.data(), oh * 8) || !up(dwc0, tab.wc0.data(), ow * 8) || !up(dwc1, tab.wc1.data(), ow * 8) ||
        !up(dr0, tab.r0.data(), oh * 4) || !up(dr1, tab.r1.data(), oh * 4) || !up(dc0, tab.c0.data(), ow * 4) || !up(dc1, tab.c1.data(), ow * 4)) {
        set_error("finish: table upload failed");
        return -2;
    }
    const dim3 grid((npx + 255) / 256 < 1024 ? (npx + 255) / 256 : 1024, k), blk(256);
    hipLaunchKernelGGL(invert_u8_kernel, grid, blk, 0, s, skel, nskel, npx);
    launch_edt(nskel, k, H, W, g, st, anyz, cdt, s);
    hipLaunchKernelGGL(weight_kernel, grid, blk, 0, s, pred, dist, cdt, wt, npx);
    hipLaunchKernelGGL((minmax_kernel<double>), dim3(k), dim3(256), 0, s, wt, (size_t)npx, lo, hi);
    const double *cur = wt;
    double *a = cdt, *b = tmp;       // cdt is free after the weighting
    if (!tab.w0.empty()) { hipLaunchKernelGGL(gauss_axis_kernel, grid, blk, 0, s, cur, a, H, W, 0, gw0, (int)tab.w0.size() / 2); cur = a; }
    if (!tab.w1.empty()) { hipLaunchKernelGGL(gauss_axis_kernel, grid, blk, 0, s, cur, b, H, W, 1, gw1, (int)tab.w1.size() / 2); cur = b; }
    const int onpx = oh * ow;
    const dim3 ogrid((onpx + 255) / 256 < 1024 ? (onpx + 255) / 256 : 1024, k);
    hipLaunchKernelGGL(zoom_clip_kernel, ogrid, blk, 0, s, cur, H, W, oh, ow, dr0, dr1, dwr0, dwr1, dc0, dc1, dwc0, dwc1, lo, hi, field_out);
    hipLaunchKernelGGL((minmax_kernel<float>), dim3(k), dim3(256), 0, s, field_out, (size_t)onpx, fmn, fmx);
    hipLaunchKernelGGL(rescale255_kernel, ogrid, blk, 0, s, field_out, onpx, fmn, fmx, f255_out);
    if (hipGetLastError() != hipSuccess) { set_error("finish: kernel launch failed"); return -2; }
    return 0;
}

}  // namespace tmat
