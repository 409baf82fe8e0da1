// Pre-processing in front of the UNet, on device (inputs stay resident in HBM):
//   a1  cv2.resize(img, round(shape * ds_ratio), INTER_LANCZOS4) on uint16   (compute_branches.py:309-312)
//   a2  rescale_intensity(out_range=(0, 1)).astype(float32)                   (compute_branches.py:316)
// Separable 8-tap Lanczos4 (OpenCV interpolateLanczos4 coefficients, computed on the host in
// csrc/postproc.cpp:lanczos_axis and uploaded), replicate border, horizontal then vertical.
//   uint16 sources (cv2: HResizeLanczos4<ushort, float, float>): f32 accumulation left-to-right, round-half-even +
//   saturate.
//   uint8 sources (FIXED; cv2: HResizeLanczos4<uchar, int, short> + FixedPtCast<int, uchar, 22>): coefficients rounded
//   to 11-bit fixed point (rint(c * 2048), as saturate_cast<short> does), int32 accumulation in both passes (the
//   intermediate buffer holds int32 in the same 4 bytes), result (v + 2^21) >> 22 saturated to 0..255.
// HBM-bound: 2 B/px in, 4 B/px intermediate, 2+4 B/px out.  Same operation order as oracle/morph.py (no FMA contraction).
#include "tmat_internal.h"

namespace tmat {

template <bool FIXED>
__global__ __launch_bounds__(256) void lanczos_h_kernel(const uint16_t *__restrict__ img, int H, int W, int w,
                                                        const int *__restrict__ xi, const float *__restrict__ xc,
                                                        float *__restrict__ tmp)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int n = blockIdx.z;
    if (x >= w || y >= H) return;
    const uint16_t *row = img + ((size_t)n * H + y) * W;
    if (FIXED) {
        int acc = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) acc += (int)row[xi[x * 8 + k]] * (int)rintf(xc[x * 8 + k] * 2048.0f);
        reinterpret_cast<int *>(tmp)[((size_t)n * H + y) * w + x] = acc;
        return;
    }
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++) acc = acc + (float)row[xi[x * 8 + k]] * xc[x * 8 + k];
    tmp[((size_t)n * H + y) * w + x] = acc;
}

template <bool FIXED>
__global__ __launch_bounds__(256) void lanczos_v_kernel(const float *__restrict__ tmp, int H, int h, int w,
                                                        const int *__restrict__ yi, const float *__restrict__ yc,
                                                        uint16_t *__restrict__ out, float sat)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int n = blockIdx.z;
    if (x >= w || y >= h) return;
    if (FIXED) {
        const int *ib = reinterpret_cast<const int *>(tmp) + (size_t)n * H * w + x;
        int acc = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) acc += ib[(size_t)yi[y * 8 + k] * w] * (int)rintf(yc[y * 8 + k] * 2048.0f);
        const int v = (acc + (1 << 21)) >> 22;
        out[((size_t)n * h + y) * w + x] = (uint16_t)min(max(v, 0), 255);
        return;
    }
    const float *base = tmp + (size_t)n * H * w + x;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 8; k++) acc = acc + base[(size_t)yi[y * 8 + k] * w] * yc[y * 8 + k];
    float r = rintf(acc);
    r = fminf(fmaxf(r, 0.0f), sat);          // saturate_cast to the source depth: 65535, or 255 for 8-bit sources
    out[((size_t)n * h + y) * w + x] = (uint16_t)r;
}

// per-image min / max of u16 (one block per image), then (x - min) / (max - min) in f64 -> f32
__global__ __launch_bounds__(256) void minmax_u16_kernel(const uint16_t *__restrict__ x, size_t per, int *mn, int *mx)
{
    const uint16_t *p = x + (size_t)blockIdx.x * per;
    int lo = 65535, hi = 0;
    for (size_t i = threadIdx.x; i < per; i += 256) { int v = p[i]; lo = min(lo, v); hi = max(hi, v); }
    for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_down(lo, o)); hi = max(hi, __shfl_down(hi, o)); }
    __shared__ int slo[4], shi[4];
    if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; i++) { lo = min(lo, slo[i]); hi = max(hi, shi[i]); }
        mn[blockIdx.x] = lo; mx[blockIdx.x] = hi;
    }
}

__global__ __launch_bounds__(256) void rescale01_kernel(const uint16_t *__restrict__ x, size_t per, const int *mn,
                                                        const int *mx, float *__restrict__ out)
{
    const int n = blockIdx.y;
    const double imin = (double)mn[n], imax = (double)mx[n];
    const bool flat = mn[n] == mx[n];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < per; i += (size_t)gridDim.x * 256) {
        const double v = (double)x[(size_t)n * per + i];
        out[(size_t)n * per + i] = flat ? (float)fmin(fmax(v, 0.0), 1.0) : (float)(((v - imin) / (imax - imin)) * 1.0 + 0.0);
    }
}

void launch_lanczos(const uint16_t *img, int n, int H, int W, int h, int w, const int *xi, const float *xc, const int *yi,
                    const float *yc, float *tmp, uint16_t *out, float sat, hipStream_t s)
{
    if (sat < 256.f) {      // 8-bit sources: cv2's fixed-point path
        hipLaunchKernelGGL(lanczos_h_kernel<true>, dim3((w + 63) / 64, (H + 3) / 4, n), dim3(256), 0, s, img, H, W, w, xi, xc, tmp);
        hipLaunchKernelGGL(lanczos_v_kernel<true>, dim3((w + 63) / 64, (h + 3) / 4, n), dim3(256), 0, s, tmp, H, h, w, yi, yc, out, sat);
        return;
    }
    hipLaunchKernelGGL(lanczos_h_kernel<false>, dim3((w + 63) / 64, (H + 3) / 4, n), dim3(256), 0, s, img, H, W, w, xi, xc, tmp);
    hipLaunchKernelGGL(lanczos_v_kernel<false>, dim3((w + 63) / 64, (h + 3) / 4, n), dim3(256), 0, s, tmp, H, h, w, yi, yc, out, sat);
}

void launch_rescale01(const uint16_t *x, int n, size_t per, int *mn, int *mx, float *out, hipStream_t s)
{
    hipLaunchKernelGGL(minmax_u16_kernel, dim3(n), dim3(256), 0, s, x, per, mn, mx);
    int gx = (int)((per + 255) / 256 < 1024 ? (per + 255) / 256 : 1024);
    hipLaunchKernelGGL(rescale01_kernel, dim3(gx, n), dim3(256), 0, s, x, per, mn, mx, out);
}

}  // namespace tmat
