// Tiling / D4 test-time augmentation / squared-spline window blending on gfx950.
//
// Reference: fl_tissue_model_tools/smooth_tiled_predictions.py
//   _pad_img :68-79, _rotate_mirror_do :95-113, _windowed_subdivs :136-192,
//   _recreate_from_subdivs :195-217, _rotate_mirror_undo :116-133, driver :220-267.
//
// The reference materialises 8 padded copies, a 5-D tile array and 8 f64 canvases.  Here both
// directions are pure gathers (no atomics, no intermediates in HBM besides the patch batch):
//   extract: patch(g, a, b)[p][q] = D4_g(pad(x))[a*step + p][b*step + q]      (f32 copy)
//   blend  : out[y][x] = ( sum_g  ( sum_{tiles (a,b) of g covering the pixel, row-major}
//                                    f64(pred) * (w[p] * w[q]) ) / 4 ) / 8
// with exactly the reference's f64 operation order (sequential adds, product w[p]*w[q] formed
// first), so results are bit-identical to numpy.  Compiled with -ffp-contract=off.
#include "tmat_internal.h"

namespace tmat {

TileGeom make_geom(int hh, int ww, int ws)
{
    TileGeom g;
    g.hh = hh; g.ww = ww; g.ws = ws;
    g.step = ws / 2;                 // int(window_size / subdivisions), subdivisions = 2
    g.aug = (ws + 1) / 2;            // int(round(ws * (1 - 1/2))): ws is even for every model config
    g.Hp = hh + 2 * g.aug; g.Wp = ww + 2 * g.aug;
    int cntH = (g.Hp - ws) / g.step + 1, cntW = (g.Wp - ws) / g.step + 1;
    g.na[0] = cntH; g.nb[0] = cntW;   // even rotations: frame is Hp x Wp
    g.na[1] = cntW; g.nb[1] = cntH;   // odd rotations: frame is Wp x Hp
    int off = 0;
    for (int k = 0; k < 8; k++) { g.tile_off[k] = off; off += g.na[k & 1] * g.nb[k & 1]; }
    g.tiles_per_img = off;
    return g;
}

// frame coords (u, v) of orientation g  ->  padded-image coords (y, x)
__device__ __forceinline__ void frame_to_pad(int g, int u, int v, int Hp, int Wp, int &y, int &x)
{
    int k = g & 3, xp;
    if (k == 0) { y = u; xp = v; }
    else if (k == 1) { y = v; xp = Wp - 1 - u; }
    else if (k == 2) { y = Hp - 1 - u; xp = Wp - 1 - v; }
    else { y = Hp - 1 - v; xp = u; }
    x = (g & 4) ? Wp - 1 - xp : xp;
}
// padded-image coords -> frame coords of orientation g
__device__ __forceinline__ void pad_to_frame(int g, int y, int x, int Hp, int Wp, int &u, int &v)
{
    int k = g & 3;
    int xp = (g & 4) ? Wp - 1 - x : x;
    if (k == 0) { u = y; v = xp; }
    else if (k == 1) { u = Wp - 1 - xp; v = y; }
    else if (k == 2) { u = Hp - 1 - y; v = Wp - 1 - xp; }
    else { u = xp; v = Hp - 1 - y; }
}

// min / max of each image (float), one block per image; used for the pad value (x.min(), :77)
__global__ __launch_bounds__(256) void minmax_f32_kernel(const float *__restrict__ x, size_t per, float *mn, float *mx)
{
    const float *p = x + (size_t)blockIdx.x * per;
    float lo = INFINITY, hi = -INFINITY;
    for (size_t i = threadIdx.x; i < per; i += 256) { float v = p[i]; lo = fminf(lo, v); hi = fmaxf(hi, v); }
    for (int o = 32; o > 0; o >>= 1) { lo = fminf(lo, __shfl_down(lo, o)); hi = fmaxf(hi, __shfl_down(hi, o)); }
    __shared__ float slo[4], shi[4];
    if ((threadIdx.x & 63) == 0) { slo[threadIdx.x >> 6] = lo; shi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; i++) { lo = fminf(lo, slo[i]); hi = fmaxf(hi, shi[i]); }
        mn[blockIdx.x] = lo; mx[blockIdx.x] = hi;
    }
}
// models.py:636-637: x = (x - norm_mean) / norm_std, float32 arithmetic (numpy: a Python float next to a float32 array)
__global__ void norm_f32_kernel(float *__restrict__ x, size_t n, float mean, float sd)
{
    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) x[p] = (x[p] - mean) / sd;
}
void launch_norm_f32(float *x, size_t n, float mean, float sd, hipStream_t s)
{
    const size_t b = (n + 255) / 256;
    hipLaunchKernelGGL(norm_f32_kernel, dim3((unsigned)(b < 8192 ? (b ? b : 1) : 8192)), dim3(256), 0, s, x, n, mean, sd);
}

void launch_minmax_f32(const float *x, int n, size_t per, float *mn, float *mx, hipStream_t s)
{
    hipLaunchKernelGGL(minmax_f32_kernel, dim3(n), dim3(256), 0, s, x, per, mn, mx);
}

__global__ __launch_bounds__(256) void extract_tiles_kernel(const float *__restrict__ x, const float *__restrict__ padval,
                                                            TileGeom gm, float *__restrict__ patches)
{
    // grid: (ceil(ws*ws/256), tiles_per_img, n)
    const int img = blockIdx.z, tile = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= gm.ws * gm.ws) return;
    int g = 7;
#pragma unroll
    for (int k = 1; k < 8; k++) if (tile < gm.tile_off[k]) { g = k - 1; break; }
    const int tl = tile - gm.tile_off[g];
    const int nb = gm.nb[g & 1];
    const int a = tl / nb, b = tl - a * nb;
    const int p = pix / gm.ws, q = pix - p * gm.ws;
    int y, xx;
    frame_to_pad(g, a * gm.step + p, b * gm.step + q, gm.Hp, gm.Wp, y, xx);
    y -= gm.aug; xx -= gm.aug;
    float v = padval[img];
    if (y >= 0 && y < gm.hh && xx >= 0 && xx < gm.ww) v = x[((size_t)img * gm.hh + y) * gm.ww + xx];
    patches[(((size_t)img * gm.tiles_per_img + tile) * gm.ws + p) * gm.ws + q] = v;
}
void launch_extract_tiles(const float *x, const float *padval, int n, const TileGeom &g, float *patches, hipStream_t s)
{
    dim3 grid((g.ws * g.ws + 255) / 256, g.tiles_per_img, n);
    hipLaunchKernelGGL(extract_tiles_kernel, grid, dim3(256), 0, s, x, padval, g, patches);
}

__global__ __launch_bounds__(256) void blend_kernel(const float *__restrict__ pred, const double *__restrict__ win,
                                                    TileGeom gm, double *__restrict__ out)
{
    const int img = blockIdx.z;
    const int xo = blockIdx.x * 64 + (threadIdx.x & 63);
    const int yo = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (xo >= gm.ww || yo >= gm.hh) return;
    const int y = yo + gm.aug, x = xo + gm.aug;
    const float *pimg = pred + (size_t)img * gm.tiles_per_img * gm.ws * gm.ws;
    double total = 0.0;
#pragma unroll
    for (int g = 0; g < 8; g++) {
        int u, v;
        pad_to_frame(g, y, x, gm.Hp, gm.Wp, u, v);
        const int na = gm.na[g & 1], nb = gm.nb[g & 1];
        int a_hi = u / gm.step; if (a_hi > na - 1) a_hi = na - 1;
        int a_lo = u - gm.ws + gm.step; a_lo = a_lo > 0 ? a_lo / gm.step : 0;
        int b_hi = v / gm.step; if (b_hi > nb - 1) b_hi = nb - 1;
        int b_lo = v - gm.ws + gm.step; b_lo = b_lo > 0 ? b_lo / gm.step : 0;
        double acc = 0.0;
        for (int a = a_lo; a <= a_hi; a++) {
            const int p = u - a * gm.step;
            if (p >= gm.ws) continue;
            for (int b = b_lo; b <= b_hi; b++) {
                const int q = v - b * gm.step;
                if (q >= gm.ws) continue;
                const float pv = pimg[(((size_t)gm.tile_off[g] + a * nb + b) * gm.ws + p) * gm.ws + q];
                const double w2 = win[p] * win[q];
                acc = acc + (double)pv * w2;
            }
        }
        acc = acc / 4.0;
        total = g == 0 ? acc : total + acc;
    }
    out[((size_t)img * gm.hh + yo) * gm.ww + xo] = total / 8.0;
}
void launch_blend(const float *pred_patches, const double *win1d, int n, const TileGeom &g, double *out, hipStream_t s)
{
    dim3 grid((g.ww + 63) / 64, (g.hh + 3) / 4, n);
    hipLaunchKernelGGL(blend_kernel, grid, dim3(256), 0, s, pred_patches, win1d, g, out);
}

}  // namespace tmat
