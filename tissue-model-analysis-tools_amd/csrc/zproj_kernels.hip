// Z projections of image stacks for gfx950 (MI355X): focus stacking, min, max, mean, median.
//
// Reference: fl_tissue_model_tools/zstacks.py:134-249 (proj_focus_stacking, proj_avg, proj_med, proj_max, proj_min),
// driven by scripts/compute_zproj.py:73-84.  Arithmetic contract (shared with oracle/zproj.py, compared bit-exactly):
//   blur  B = (sum_{ij} g_i g_j I(reflect101) + 128) >> 8,  g = [1, 4, 6, 4, 1]        (cv2.GaussianBlur(5x5, sigma 0), fixed point)
//   focus L = | sum_{ij} (g_i d_j + d_i g_j) B(reflect101) |, d = [1, 0, -2, 0, 1]      (cv2.Laplacian(CV_64F, ksize 5))
//   projection = value of the first slice with the strictly largest L.
// Everything is integer arithmetic below 2^24, so it is exact in OpenCV's float work type and here.
//
// HBM-bound: every input sample is read once per tile that needs it (72 x 72 staged for 64 x 64 outputs, 1.27x through
// L2), the projection is written once; the blurred slice and the focus measure never leave LDS / registers.
// The two-stage integer filter chain costs about 45 VALU operations per pixel and slice, which bounds the kernel
// before HBM does.
#include "tmat_internal.h"
#include "morph.h"
#include "../../include/tmat.h"

#include <cstdint>

namespace tmat {

__device__ __forceinline__ int reflect101(int i, int n)
{
    if (n == 1) return 0;
    const int p = 2 * (n - 1);
    i %= p;
    if (i < 0) i += p;
    return i < n ? i : p - i;
}

constexpr int ZT = 64;              // output tile (64 x 64)
constexpr int ZI = ZT + 8;          // staged input tile (halo 4)
constexpr int ZB = ZT + 4;          // blurred tile (halo 2)
constexpr int ZBS = ZB + 4;         // row stride of the blurred tile in LDS (8-byte aligned rows)

// Two register-blocked stages per slice, both on 4 x 4 patches whose 8 x 8 source window is read from LDS with
// 8-byte loads (one LDS read per output instead of one per tap), separable in registers:
//   stage 1: input (72 x 72, reflect-101 gathered) -> blurred tile (68 x 68):  rows [1,4,6,4,1], columns [1,4,6,4,1], +128 >> 8
//   stage 2: blurred -> focus measure (64 x 64): rows with g and with d = [1,0,-2,0,1], columns d (on the g rows) + g (on the d rows)
// and the running arg-max over slices lives in registers (16 outputs per thread).
__device__ __forceinline__ void load8(const uint16_t *p, unsigned v[8])
{
    const uint2 a = *reinterpret_cast<const uint2 *>(p), b = *reinterpret_cast<const uint2 *>(p + 4);
    v[0] = a.x & 0xffffu; v[1] = a.x >> 16; v[2] = a.y & 0xffffu; v[3] = a.y >> 16;
    v[4] = b.x & 0xffffu; v[5] = b.x >> 16; v[6] = b.y & 0xffffu; v[7] = b.y >> 16;
}

// Round 4: the input tile is double-buffered and the NEXT slice travels global -> LDS by LDS-DMA (no registers, no ds_write) while the
// current slice is filtered.  Rounds 1-3 loaded a slice into LDS with 2-byte gathers and waited at a barrier: the counters showed the
// waves waiting 59 % of their cycles (profiles/r03_zproj_pmc.txt).  Interior tiles (no reflection, even W: every pixel pair of the 72 x 72
// window is 4 contiguous, aligned bytes) use the DMA: piece j = 256 bytes of the flat tile, 4 bytes per lane, its global offset recomputed
// per slice from the lane id (6 vector instructions; 11 pieces per wave) so that nothing lives in registers across the filter stages;
// border tiles keep the reflect-101 gather.  Two separate LDS objects for the two buffers: with one array hipcc cannot tell the DMA
// target from the ds_read source and waits for the DMA before the first read of the stage it should overlap with.
typedef __attribute__((address_space(3))) void zp_lds_void_t;
constexpr int ZIN = ZI * ZI + 64;       // the last DMA piece is written whole (256 bytes): pad the tile to 41 pieces

#ifndef ZP_WPS
#define ZP_WPS 4        // waves per SIMD the kernel is compiled for (128 registers; 3 values spill)
#endif
__global__ __launch_bounds__(256, ZP_WPS) void zproj_focus_kernel(const uint16_t *__restrict__ stacks, int Z, int H, int W,
                                                          uint16_t *__restrict__ out)
{
    __shared__ __attribute__((aligned(16))) uint16_t inA[ZIN];
    __shared__ __attribute__((aligned(16))) uint16_t inB[ZIN];
    __shared__ __attribute__((aligned(8))) uint16_t bl[ZB][ZBS];
    __shared__ int ry[ZI], rx[ZI];
    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int x0 = blockIdx.x * ZT, y0 = blockIdx.y * ZT;
    const size_t npx = (size_t)H * W;
    const uint16_t *st = stacks + (size_t)blockIdx.z * Z * npx;
    if (t < ZI) { ry[t] = reflect101(y0 - 4 + t, H); rx[t] = reflect101(x0 - 4 + t, W); }
    __syncthreads();
    // uniform per block: the whole window lies inside the image, pixel pairs are 4-byte aligned, the stack fits a buffer descriptor
    const bool fast = x0 >= 4 && x0 + ZT + 4 <= W && y0 >= 4 && y0 + ZT + 4 <= H && (W & 1) == 0 &&
                      (size_t)Z * npx * 2 < 0x7fffffffull && ((reinterpret_cast<uintptr_t>(st) & 3) == 0);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)st, 0, 0x7fffffff, 0x00020000);
    constexpr int NPIECE = (ZI * ZI * 2 + 255) / 256;                    // 41
    auto issue_dma = [&](uint16_t *dst, int z) __attribute__((always_inline)) {
        const int soff = (int)((size_t)z * npx * 2);
        int lane_o = lane;                                               // opaque: as loop invariants the 11 offsets would live in registers
        asm volatile("" : "+v"(lane_o));                                 // across both filter stages (156 instead of 128 VGPRs: 3 waves per SIMD)
#pragma unroll
        for (int i = 0; i < (NPIECE + 3) / 4; i++) {
            const int j = wave + 4 * i;                                  // wave-uniform piece index
            if (j < NPIECE) {
                const int pix = 128 * j + 2 * lane_o;
                const int iy = pix / ZI, ix = pix - iy * ZI;
                const unsigned vo = pix < ZI * ZI ? (unsigned)(((y0 - 4 + iy) * W + x0 - 4 + ix) * 2) : 0x80000000u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (zp_lds_void_t *)(dst + 128 * j), 4, vo, soff, 0, 0);
            }
        }
    };
    auto fill_gather = [&](uint16_t *dst, int z) __attribute__((always_inline)) {
        const uint16_t *sl = st + (size_t)z * npx;
        for (int e = t; e < ZI * ZI; e += 256) {
            const int iy = e / ZI, ix = e - iy * ZI;
            dst[e] = sl[(size_t)ry[iy] * W + rx[ix]];
        }
    };
    const int py = t >> 4, px = t & 15;                  // this thread's 4 x 4 output patch
    int best[4][4];
    unsigned val[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int c = 0; c < 4; c++) { best[i][c] = -1; val[i][c] = 0; }

    // one slice: `in` holds slice z, `nxt` receives slice z + 1 meanwhile
    auto slice = [&](const uint16_t *in, uint16_t *nxt, int z) __attribute__((always_inline)) {
        if (fast && z + 1 < Z) issue_dma(nxt, z + 1);    // in flight under the two filter stages of slice z
        // stage 1: 17 x 17 patches of the blurred tile
        for (int q = t; q < (ZB / 4) * (ZB / 4); q += 256) {
            const int qy = q / (ZB / 4), qx = q - qy * (ZB / 4);
            unsigned hb[8][4];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                unsigned v[8];
                load8(in + (4 * qy + r) * ZI + 4 * qx, v);
#pragma unroll
                for (int c = 0; c < 4; c++) hb[r][c] = v[c] + 4u * v[c + 1] + 6u * v[c + 2] + 4u * v[c + 3] + v[c + 4];
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                unsigned b[4];
#pragma unroll
                for (int c = 0; c < 4; c++)
                    b[c] = (hb[i][c] + 4u * hb[i + 1][c] + 6u * hb[i + 2][c] + 4u * hb[i + 3][c] + hb[i + 4][c] + 128u) >> 8;
                *reinterpret_cast<uint2 *>(&bl[4 * qy + i][4 * qx]) = make_uint2(b[0] | (b[1] << 16), b[2] | (b[3] << 16));
            }
        }
        __syncthreads();
        // stage 2: this thread's patch of the focus measure, arg-max update
        {
            int hg[8][4], hd[8][4];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                unsigned v[8];
                load8(&bl[4 * py + r][4 * px], v);
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    hg[r][c] = (int)(v[c] + 4u * v[c + 1] + 6u * v[c + 2] + 4u * v[c + 3] + v[c + 4]);
                    hd[r][c] = (int)v[c] - 2 * (int)v[c + 2] + (int)v[c + 4];
                }
            }
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint2 cw = *reinterpret_cast<const uint2 *>(in + (4 * py + 4 + i) * ZI + 4 * px + 4);      // centre input values
                const unsigned cv[4] = {cw.x & 0xffffu, cw.x >> 16, cw.y & 0xffffu, cw.y >> 16};
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    int s = hg[i][c] - 2 * hg[i + 2][c] + hg[i + 4][c] + hd[i][c] + 4 * hd[i + 1][c] + 6 * hd[i + 2][c] + 4 * hd[i + 3][c] + hd[i + 4][c];
                    s = s < 0 ? -s : s;
                    if (s > best[i][c]) { best[i][c] = s; val[i][c] = cv[c]; }
                }
            }
        }
        // border tiles: the next slice by the reflect-101 gather (nobody reads `nxt`: slice z - 1's readers passed the barrier above)
        if (!fast && z + 1 < Z) fill_gather(nxt, z + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces have landed ...
        __syncthreads();                                  // ... and everybody's: bl is free again, the next input tile is complete
    };

    if (fast) issue_dma(inA, 0); else fill_gather(inA, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int z = 0; z < Z; z += 2) {
        slice(inA, inB, z);
        if (z + 1 < Z) slice(inB, inA, z + 1);
    }
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int y = y0 + 4 * py + i, x = x0 + 4 * px + c;
            if (y < H && x < W) out[(size_t)blockIdx.z * npx + (size_t)y * W + x] = (uint16_t)val[i][c];
        }
}

// min / max / mean / median along z: one thread per pixel
constexpr int ZMED_MAX = 64;
__global__ __launch_bounds__(256) void zproj_reduce_kernel(const uint16_t *__restrict__ stacks, int Z, size_t npx, int method,
                                                           uint16_t *__restrict__ out16, double *__restrict__ out64)
{
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= npx) return;
    const uint16_t *st = stacks + (size_t)blockIdx.y * Z * npx + p;
    const size_t o = (size_t)blockIdx.y * npx + p;
    if (method == TMAT_ZPROJ_MIN || method == TMAT_ZPROJ_MAX) {
        unsigned m = st[0];
        for (int z = 1; z < Z; z++) { const unsigned v = st[(size_t)z * npx]; m = method == TMAT_ZPROJ_MIN ? (v < m ? v : m) : (v > m ? v : m); }
        out16[o] = (uint16_t)m;
    } else if (method == TMAT_ZPROJ_AVG) {
        double s = 0.0;                                  // integers: exact in any order (np.mean, f64 accumulator)
        for (int z = 0; z < Z; z++) s += (double)st[(size_t)z * npx];
        out64[o] = s / (double)Z;
    } else if (Z > ZMED_MAX) {                           // median of a deep stack: bitwise radix select, no slice limit
        // k-th smallest (0-based) of the Z values: walk the 16 bits from the top; among the values that share the
        // prefix chosen so far, count those with the current bit clear: the k-th lies among them iff k < count.
        auto kth = [&](int k) {
            unsigned prefix = 0;
            for (int bit = 15; bit >= 0; bit--) {
                int c0 = 0;
                for (int z = 0; z < Z; z++) {
                    const unsigned x = st[(size_t)z * npx];
                    c0 += ((x >> (bit + 1)) == (prefix >> (bit + 1)) && !((x >> bit) & 1u)) ? 1 : 0;
                }
                if (k >= c0) { prefix |= 1u << bit; k -= c0; }
            }
            return prefix;
        };
        out64[o] = (Z & 1) ? (double)kth(Z / 2) : ((double)kth(Z / 2 - 1) + (double)kth(Z / 2)) / 2.0;
    } else {                                             // median (np.median: mean of the middle value(s))
        uint16_t v[ZMED_MAX];
        for (int z = 0; z < Z; z++) {                    // insertion sort
            const uint16_t x = st[(size_t)z * npx];
            int i = z;
            while (i > 0 && v[i - 1] > x) { v[i] = v[i - 1]; i--; }
            v[i] = x;
        }
        out64[o] = (Z & 1) ? (double)v[Z / 2] : ((double)v[Z / 2 - 1] + (double)v[Z / 2]) / 2.0;
    }
}

int zproj_dev(const uint16_t *stacks, int n, int Z, int H, int W, int method, void *out, hipStream_t s)
{
    if (method == TMAT_ZPROJ_FS) {
        hipLaunchKernelGGL(zproj_focus_kernel, dim3((W + ZT - 1) / ZT, (H + ZT - 1) / ZT, n), dim3(256), 0, s, stacks, Z, H, W, (uint16_t *)out);
    } else {
        const size_t npx = (size_t)H * W;
        hipLaunchKernelGGL(zproj_reduce_kernel, dim3((unsigned)((npx + 255) / 256), n), dim3(256), 0, s, stacks, Z, npx, method,
                           (uint16_t *)out, (double *)out);
    }
    if (hipGetLastError() != hipSuccess) { set_error("zproj: kernel launch failed"); return -2; }
    return 0;
}

}  // namespace tmat
