// Fused SeparableConv2D (depthwise 3x3 -> pointwise 1x1 -> folded BN [-> ReLU]) for gfx950 (MI355X).  NHWC float32.
//
// Reference graph: fl_tissue_model_tools/models.py:126-136 (the two SeparableConv2D + BatchNormalization pairs of every
// down block), executed by keras Model.predict at smooth_tiled_predictions.py:179.
//
// Why fused: as separate launches the depthwise intermediate is written to HBM and read back by the pointwise GEMM
// (109 GB per 1600-patch pass).  Here a workgroup stages the spatial HALO tile of the input once per 16-channel block
// in LDS, every lane computes its own MFMA A fragment (the depthwise outputs of its pixel for its channels) in
// registers straight from the halo tile, and the pointwise contraction runs on v_mfma_f32_32x32x2_f32.  The depthwise
// tensor never exists in memory.
//
// Arithmetic contract (identical to the unfused pair dwconv_kernel -> conv_mfma_kernel<...,1,...>, and to
// oracle/unet_exact.c:orc_dwconv -> orc_conv): depthwise value = chain over the 9 taps in (ky, kx) order from +0.0,
// acc = fmaf(x, w, acc), zero padding, optional ReLU on load; pointwise = chain over the input channels in groups of 8
// in the order 0,4,1,5,2,6,3,7 (lanes 0-31 feed k = 0, lanes 32-63 k = 1 of each MFMA); epilogue fmaf(acc, scale,
// shift), optional ReLU.  Bit-exact with the unfused path by construction; tests/test_gpu_unet.py compares bits.
#include "dev_guard.h"
#include "tmat_internal.h"
#include "../../include/tmat.h"

#include <cstdlib>
#include <cstdio>
#include <vector>

namespace tmat {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_t;

// Geometry: one workgroup = 512 threads = 8 waves; one TILE = a 16 x 16 pixel tile (M = 256) x 128 output channels.
// Wave w owns tile rows 2w, 2w+1 (32 pixels) x 128 channels = four 32x32 accumulator tiles (64 VGPRs).
// One STEP = one K chunk of 16 input channels = two 8-channel planes (g = 0, 1: the channel groups of the two MFMA
// k-quads).  Workgroups are PERSISTENT: each walks a contiguous range of (pixel tile, channel tile) pairs as one
// stream of steps through a three-stage LDS ring, so the loads of the next tile run under the MFMAs of this one.
// LDS per stage, in 16-byte cells:
//   B    : [16 channels][128 output channels] f32 (n contiguous, filled from the [Cin][Cout] weights: one DMA piece =
//          two channel rows); an MFMA's B value (channel 8g + 4h + e, column n) is ONE float per lane, read with
//          ds_read_b32 (32 consecutive floats per half wave: conflict-free) one k step ahead of its MFMA
//   halo : plane g at cell g * 704: 18 x 18 pixels x 2 cells (channel quads h = 0, 1), linear pitch 18; the cell of
//          (pixel p, quad h) is 2 p + (h ^ ((p >> 3) & 1)); 11 DMA pieces of 64 cells per plane
//   dw   : 9 taps x 4 cells (the depthwise taps of this channel block, [tap][16 channels]); 1 DMA piece
// Every 16-lane service group of a ds_read_b128 ({0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32) must hit 16
// distinct cells mod 16, i.e. 16 pixels that are distinct mod 16.
// For the halo the MFMA row index r = lane & 31 is mapped to tile pixels so that each service group is 16 consecutive
// pixels of ONE tile row (PIXMAP below): their halo positions p are consecutive for every tap.
// The plane split keeps g out of the swizzle: the g = 1 fragment sits at a constant byte offset from the g = 0 one
// (an instruction immediate, no second set of address registers).
// Everything reaches LDS by LDS-DMA (buffer_load_dwordx4 ... lds, lane-linear destination; the swizzle is applied to
// the per-lane SOURCE address).
//
// Pipeline (per wave, step s in stage s % 3; one vmcnt(0) + barrier per step):
//   top of step s   : issue the DMA of step s + 2 into stage (s + 2) % 3 (last read in step s - 1, before that barrier)
//   8 segments      : segment k = the 4 MFMAs of k step k (k-quad g = k / 4, element e = k % 4) interleaved with 12
//                     depthwise FMAs; segments 0-2 finish the A fragment of THIS step's k-quad 1, segments 4-6 build
//                     the A fragment of the NEXT step's k-quad 0 (its halo landed before the previous barrier); each
//                     segment first issues the LDS reads the next segment consumes (two register sets), so no MFMA
//                     group waits on LDS latency and all depthwise VALU work sits between MFMAs.
//   end of step     : vmcnt(0) (the DMA issued at the top has had the whole step), barrier; after the last chunk of a
//                     tile the accumulators are stored straight from registers (4 B per lane: one instruction writes
//                     two full 128-byte lines) and cleared.
// Measured alternatives (tools/build_variant.sh, in-kernel s_memtime stamps): the 8 MFMA segments of a step take
// ~2500 cycles (2048 = the MFMAs alone).  Splitting the halo into four one-cell planes (no swizzle, one address
// register) made every DMA piece touch 64 cache lines instead of 32 and the vector-memory issue queue became the
// bottleneck (DMA issue 1300-3100 cycles per step); spreading the epilogue stores over the next tile's steps with a
// counted vmcnt cost more in VALU and registers than the burst it removed.
#ifndef SEP_WPS
#define SEP_WPS 2
#endif
#ifndef SEP_DEFAULT_WAVES
#define SEP_DEFAULT_WAVES 8
#endif
constexpr int SEP_KC = 16;
// NW = waves per workgroup: 8 (a 16 x 16 pixel tile, one workgroup per CU) or 4 (an 8-row x 16-pixel tile, TWO workgroups per
// CU with independent barriers: while one is stalled issuing its LDS-DMA or storing a tile, the other keeps the matrix pipe busy)
constexpr int sep_pp(int NW) { return ((2 * NW + 2) * 18 * 2 + 63) / 64; }         // 64-cell DMA pieces per halo plane: 11 / 6
constexpr int SEP_B_CELLS = 8 * 64;
constexpr int SEP_DW_CELLS = 64;
// stage = [B | dw | halo]; 31 KiB (NW = 8) / 21 KiB (NW = 4), every (stage base + constant) of stages 0 and 1 fits the 16-bit ds_read offset field
constexpr int SEP_B_OFF = 0, SEP_D_OFF = SEP_B_CELLS * 4, SEP_H_OFF = (SEP_B_CELLS + SEP_DW_CELLS) * 4;        // float offsets
constexpr int sep_stage_floats(int NW) { return (2 * sep_pp(NW) * 64 + SEP_B_CELLS + SEP_DW_CELLS) * 4; }

struct SepArgs {
    const float *in;      // (N, H, W, Cin)
    int N, H, W, Cin, Cout;
    const float *dwq;     // depthwise taps, [Cin / 16][9][16]
    const float *pw;      // pointwise weights [Cin][Cout] (the Keras layout)
    const float *scale, *shift;
    int relu_out;
    float *out;           // (N, H, W, Cout); POOL: (N, H/2, W/2, Cout) = pooled + resid, except row 7 / column 7 of every tile, which are
                          // left as partial maxima for pool_fix_add_kernel
    const float *resid;   // POOL only: (N, H/2, W/2, Cout)
    // POOL only: what the tiles above / left of this one need from it (pool_fix_add_kernel)
    float *strip_h;       // [tile][8][Cout]: row 0 of the tile, max over x in {2 px, 2 px + 1, 2 px + 2} (px = 7: without x = 16)
    float *strip_v;       // [tile][8][Cout]: column 0 of the tile, max over y in {2 py, 2 py + 1, 2 py + 2} (py = 7: without y = 16)
    float *corner;        // [tile][Cout]:   pixel (0, 0) of the tile
    long long *diag;      // SEP_DIAG builds only: [workgroup][wave][6] cycle sums
};

// PIXMAP: MFMA row rho (0..31) of wave w -> tile pixel (2w + yl, x)
__device__ __forceinline__ int pixmap_y(int r) { return ((r >= 4 && r < 12) || (r >= 16 && r < 20) || r >= 28) ? 1 : 0; }
__device__ __forceinline__ int pixmap_x(int r) { return r < 4 ? r : r < 12 ? r - 4 : r < 16 ? r - 8 : r < 20 ? r - 8 : r < 28 ? r - 12 : r - 16; }

template <bool RELU_IN, bool POOL, int NW>
__global__ __launch_bounds__(64 * NW, SEP_WPS) void sepconv_mfma_kernel(SepArgs a, int nMt, int nNt, int G)
{
    constexpr int PP = sep_pp(NW), SEP_HPLANE = PP * 64, SEP_STAGE_FLOATS = sep_stage_floats(NW), TR = 2 * NW;
    __shared__ __attribute__((aligned(16))) float stage0[SEP_STAGE_FLOATS];
    __shared__ __attribute__((aligned(16))) float stage1[SEP_STAGE_FLOATS];
    __shared__ __attribute__((aligned(16))) float stage2[SEP_STAGE_FLOATS];
    // POOL: what wave w + 1 hands to wave w at the end of a tile -- its upper row pooled along x (8 values) and that row's
    // first pixel, per output channel.  Its own object: the ring stages have LDS-DMA in flight around the epilogue.
    // (NW = 4: exchanged in two halves of 64 channels, so that two workgroups fit a CU's LDS)
    constexpr int XH = NW == 4 ? 2 : 1, XC = 128 / XH;           // exchange rounds per tile, channels per round
    __shared__ float xch[POOL ? NW * 9 * XC : 1];

    // Persistent ranges: blocks b and b + 8 share an XCD; give every XCD a contiguous super-range of the (pixel tile,
    // channel tile) pairs and every workgroup a contiguous piece of it, so the overlapping halos of neighbouring tiles
    // and the re-read of a pixel tile by its nNt channel tiles are served by that XCD's L2.
    const int b = blockIdx.x;
    const int bp = (b & 7) * (G >> 3) + (b >> 3);
    const long long P = (long long)nMt * nNt;
    const int j0 = (int)(P * bp / G), j1 = (int)(P * (bp + 1) / G);
    if (j0 >= j1) return;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int TW = a.W >> 4, TPP = (a.H / TR) * TW;          // tiles per row / per patch
    const int Cin = a.Cin;
    const int nchunks = Cin / SEP_KC;
    const int total = (j1 - j0) * nchunks;                   // steps of this workgroup
    constexpr unsigned OOB = 0x80000000u;

    // ---- DMA roles (tile-independent part) ------------------------------------------------------------------------
    // B piece of this wave: channel rows 2 wave, 2 wave + 1 of the block; lane -> (row lane >> 5, columns 4 (lane & 31) ..):
    // its offset is recomputed per step from the re-derived lane id (one register less across the loop)
    const int bchunk = SEP_KC * a.Cout * 4;                  // bytes between channel blocks of the weights
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void *)a.dwq, 0, 0x7fffffff, 0x00020000);

    // DMA iterator: the next step to load is (pair dj, chunk dc); its tile-dependent addressing:
    int dj = j0, dc = 0, dleft = total;
    unsigned hv[3];
    const float *dA = a.in, *dB = a.pw;
    auto dma_tile = [&](int jj) {
        const int mt = jj / nNt, nt = jj - mt * nNt;
        const int n = mt / TPP, tr = mt - n * TPP;
        const int ty0 = (tr / TW) * TR, tx0 = (tr % TW) * 16;
        dA = a.in + (size_t)n * a.H * a.W * Cin;
        dB = a.pw + nt * 128;
        // piece i * 8 + wave of 22: plane g = piece / 11, cells (piece % 11) * 64 + lane = (halo position, quad).  Recomputed
        // per tile from an opaque copy of the lane id: kept in registers across the loop these would spill.
        int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int i = 0; i < 3; i++) {
            const int piece = i * NW + wave;
            const int g = piece / PP, qq = (piece - g * PP) * 64 + ln;
            const int p = qq >> 1, d = qq & 1;
            const int hy = p / 18, hx = p - hy * 18;
            const int Y = ty0 - 1 + hy, X = tx0 - 1 + hx;
            const bool ok = piece < 2 * PP && p < 18 * (TR + 2) && Y >= 0 && Y < a.H && X >= 0 && X < a.W;
            hv[i] = ok ? (unsigned)((Y * a.W + X) * Cin + (2 * g + (d ^ ((p >> 3) & 1))) * 4) * 4u : OOB;
        }
    };
    dma_tile(dj);

#define SEP_ISSUE(stage_)                                                                                       \
    {                                                                                                           \
        float *st = (stage_);                                                                                   \
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)dA, 0, 0x7fffffff, 0x00020000); \
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void *)dB, 0, 0x7fffffff, 0x00020000); \
        const int so = dc * (SEP_KC * 4);                                                                       \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t *)(st + SEP_H_OFF + (0 * NW + wave) * 256), 16, hv[0], so, 0, 0); \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t *)(st + SEP_H_OFF + (1 * NW + wave) * 256), 16, hv[1], so, 0, 0); \
        if (2 * NW + wave < 2 * PP)                                                                             \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t *)(st + SEP_H_OFF + (2 * NW + wave) * 256), 16, hv[2], so, 0, 0); \
        const unsigned ln_ = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));                \
        _Pragma("unroll") for (int bj = 0; bj < 8 / NW; bj++) {     /* B pieces bj NW + wave: channel rows 2 piece, 2 piece + 1 */ \
            const unsigned bvo = ((2 * (bj * NW + wave) + (ln_ >> 5)) * a.Cout + (ln_ & 31) * 4) * 4u;          \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_t *)(st + SEP_B_OFF + (bj * NW + wave) * 256), 16, bvo, dc * bchunk, 0, 0); \
        }                                                                                                       \
        if (wave == NW - 2 + (dc & 1))                                                                          \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsD, (lds_void_t *)(st + SEP_D_OFF), 16, ln_ < 36 ? ln_ * 16u : OOB, dc * 576, 0, 0); \
        dleft--;                                                                                                \
        if (++dc == nchunks) { dc = 0; dj++; if (dleft > 0) dma_tile(dj); }                                     \
    }

// the same issue cut into parts that SEP_STEP places between its MFMA segments (SEP_VAR_SPREAD): every wave issuing its five
// DMA instructions at the top of the step queues 840 cache-line requests on the CU's texture path at once, and the waves that
// arrive last (4-7) wait ~1700 cycles for their turn with no MFMA of theirs in the pipe
#define SEP_ISSUE_PART(stage_, part_)                                                                           \
    {                                                                                                           \
        float *st = (stage_);                                                                                   \
        const int so = dc * (SEP_KC * 4);                                                                       \
        if (part_ < 3) {                                                                                        \
            const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)dA, 0, 0x7fffffff, 0x00020000); \
            if (part_ * NW + wave < 2 * PP)                                                                     \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t *)(st + SEP_H_OFF + (part_ * NW + wave) * 256), 16, hv[part_], so, 0, 0); \
        } else if (part_ == 3) {                                                                                \
            const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void *)dB, 0, 0x7fffffff, 0x00020000); \
            const unsigned ln_ = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));            \
            _Pragma("unroll") for (int bj = 0; bj < 8 / NW; bj++) {                                             \
                const unsigned bvo = ((2 * (bj * NW + wave) + (ln_ >> 5)) * a.Cout + (ln_ & 31) * 4) * 4u;      \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_t *)(st + SEP_B_OFF + (bj * NW + wave) * 256), 16, bvo, dc * bchunk, 0, 0); \
            }                                                                                                   \
            if (wave == NW - 2 + (dc & 1))                                                                      \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsD, (lds_void_t *)(st + SEP_D_OFF), 16, ln_ < 36 ? ln_ * 16u : OOB, dc * 576, 0, 0); \
        } else {                                                                                                \
            dleft--;                                                                                            \
            if (++dc == nchunks) { dc = 0; dj++; if (dleft > 0) dma_tile(dj); }                                 \
        }                                                                                                       \
    }

    // ---- fragment roles -------------------------------------------------------------------------------------------
    const int r = lane & 31, h = lane >> 5;
    const int yl = pixmap_y(r), x = pixmap_x(r);
    int hoff[9];            // BYTE offset of this lane's cell (quad h of plane 0) per tap; plane 1 is + SEP_HPLANE cells
#pragma unroll
    for (int tp = 0; tp < 9; tp++) {
        const int p = (2 * wave + yl + tp / 3) * 18 + x + tp % 3;
        hoff[tp] = (SEP_H_OFF + (p * 2 + (h ^ ((p >> 3) & 1))) * 4) * 4;
    }
    int boff = (SEP_B_OFF + 4 * h * 128 + r) * 4;        // bytes; + ((8 g + e) * 128 + 32 jn) * 4
    int doff = (SEP_D_OFF + h * 4) * 4;                  // bytes; + tap * 64 B, g = 1: + 32 B

    f32x16 acc[4];
#pragma unroll
    for (int jn = 0; jn < 4; jn++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[jn][e] = 0.f;
    // register sets: two (pixel quad, tap quad) triples for the depthwise rounds, two B quadruples for the k steps
    float4 tvA[3], twA[3], tvB[3], twB[3];
    float bqa[4], bqb[4];
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;

#ifdef SEP_ABL_NORD       // timing ablation: no halo / tap reads
#define SEP_RD(st, tv, tw, g, t0) _Pragma("unroll") for (int u = 0; u < 3; u++) { tv[u] = make_float4(1.f, 2.f, 3.f, 4.f); tw[u] = make_float4(.1f, .2f, .3f, .4f); }
#else
#define SEP_RD_REAL 1
#endif
#ifdef SEP_RD_REAL
#define SEP_RD(st, tv, tw, g, t0)                                                                               \
    _Pragma("unroll") for (int u = 0; u < 3; u++) {                                                             \
        tv[u] = *reinterpret_cast<const float4 *>(st + hoff[t0 + u] + g * (SEP_HPLANE * 16));                   \
        tw[u] = *reinterpret_cast<const float4 *>(st + doff + (t0 + u) * 64 + g * 32);                          \
    }
#endif
#ifdef SEP_ABL_NODW
#define SEP_FM(av, tv, tw) { av.x += tv[0].x + tw[0].x; }
#else
#define SEP_FM_REAL 1
#endif
#ifdef SEP_FM_REAL
#define SEP_FM(av, tv, tw)                                                                                      \
    _Pragma("unroll") for (int u = 0; u < 3; u++) {                                                             \
        if (RELU_IN) { tv[u].x = fmaxf(tv[u].x, 0.f); tv[u].y = fmaxf(tv[u].y, 0.f); tv[u].z = fmaxf(tv[u].z, 0.f); tv[u].w = fmaxf(tv[u].w, 0.f); } \
        av.x = fmaf(tv[u].x, tw[u].x, av.x); av.y = fmaf(tv[u].y, tw[u].y, av.y);                               \
        av.z = fmaf(tv[u].z, tw[u].z, av.z); av.w = fmaf(tv[u].w, tw[u].w, av.w);                               \
    }
#endif
#ifdef SEP_ABL_NORB       // timing ablation: no B reads
#define SEP_RB(st, bq, g, e) _Pragma("unroll") for (int jn = 0; jn < 4; jn++) bq[jn] = 1.0f + jn + e;
#else
#define SEP_RB_REAL 1
#endif
#ifdef SEP_RB_REAL
#define SEP_RB(st, bq, g, e)                                                                                    \
    _Pragma("unroll") for (int jn = 0; jn < 4; jn++)                                                            \
        bq[jn] = *reinterpret_cast<const float *>(st + boff + (((8 * g + e) * 128) + 32 * jn) * 4);
#endif
#ifdef SEP_ABL_NOMFMA     // timing ablations only (wrong results): no MFMAs / no depthwise FMAs / no DMA
#define SEP_MM(aval, bq) _Pragma("unroll") for (int jn = 0; jn < 4; jn++) acc[jn][0] += aval * bq[jn];
#else
#define SEP_MM(aval, bq)                                                                                        \
    _Pragma("unroll") for (int jn = 0; jn < 4; jn++) acc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(aval, bq[jn], acc[jn], 0, 0, 0);
#endif
#define SEP_PIN() __builtin_amdgcn_sched_barrier(0);
// inside a segment: LDS reads first, then MFMA / VALU alternating (one MFMA, a quarter of the segment's VALU work)
#if defined(SEP_VAR_MIX) && SEP_VAR_MIX == 0       // measured alternative: leave the order inside a segment to the compiler
#define SEP_MIX(nv)
#elif defined(SEP_VAR_MIX) && SEP_VAR_MIX == 1     // measured alternative: LDS reads, the four MFMAs back to back, then the VALU work
#define SEP_MIX(nv)                                                                                             \
    __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);                                                         \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                          \
    __builtin_amdgcn_sched_group_barrier(0x002, 4 * nv, 0);
#elif defined(SEP_VAR_MIX) && SEP_VAR_MIX == 2     // measured alternative: pairs
#define SEP_MIX(nv)                                                                                             \
    __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);                                                         \
    _Pragma("unroll") for (int q_ = 0; q_ < 2; q_++) {                                                          \
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                                                      \
        __builtin_amdgcn_sched_group_barrier(0x002, 2 * nv, 0);                                                 \
    }
#else
#define SEP_MIX(nv)                                                                                             \
    __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);                                                         \
    _Pragma("unroll") for (int q_ = 0; q_ < 4; q_++) {                                                          \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                      \
        __builtin_amdgcn_sched_group_barrier(0x002, nv, 0);                                                     \
    }
#endif
    constexpr int NV = RELU_IN ? 6 : 3;

    // ---- epilogue of one tile: straight from the accumulators.  C/D layout of the 32x32 tile: column = lane & 31
    // (output channel), row rho = (e & 3) + 8 (e >> 2) + 4 (lane >> 5) (the MFMA row, i.e. PIXMAP's r).  One store
    // instruction writes two full 128-byte lines (32 consecutive channels of two pixels).
    auto store_tile = [&](int jj) {
        const int mt = jj / nNt, nt = jj - mt * nNt;
        const int n = mt / TPP, tr = mt - n * TPP;
        const int ty0 = (tr / TW) * TR, tx0 = (tr % TW) * 16;
        const int n0 = nt * 128;
        float *obase = a.out + ((size_t)n * a.H * a.W + (size_t)(ty0 + 2 * wave) * a.W + tx0) * a.Cout + n0 + r;
#pragma unroll
        for (int jn = 0; jn < 4; jn++) {
            const float sc = a.scale[n0 + jn * 32 + r], sh = a.shift[n0 + jn * 32 + r];
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int rho = (e & 3) + 8 * (e >> 2) + 4 * h;
                float v = fmaf(acc[jn][e], sc, sh);
                if (a.relu_out) v = fmaxf(v, 0.f);
                obase[((size_t)pixmap_y(rho) * a.W + pixmap_x(rho)) * a.Cout + jn * 32] = v;
                acc[jn][e] = 0.f;
            }
        }
    };

    // ---- POOL epilogue: MaxPooling2D(3, strides 2, "same") of the tile (models.py:138; TF pads after, so output (py, px)
    // covers rows 2 py .. 2 py + 2, columns 2 px .. 2 px + 2) straight from the accumulators.  A wave holds tile rows
    // y0 = 2 w, y1 = 2 w + 1; per lane and channel group the 16 registers are 4 x-quads, and for every quad one half
    // wave holds the y0 pixels, the other half the y1 pixels (PIXMAP), so one cross-half exchange gives a half wave both
    // rows of "its" 9 columns 8 h .. 8 h + 8.  Row 2 w + 2 comes from wave w + 1 through LDS (already pooled along x).
    // What lies in the next tile (row 16, column 16) is missing here: the pooled values of row 7 / column 7 are partial,
    // and the tile's own row 0 / column 0 / corner go to the strips for pool_fix_add_kernel to finish its neighbours.
    auto store_tile_pool = [&](int jj) {
        const int mt = jj / nNt, nt = jj - mt * nNt;
        const int n = mt / TPP, tr = mt - n * TPP;
        const int ty = tr / TW, tx = tr - ty * TW;
        const int n0 = nt * 128;
        const int Wp = a.W >> 1;
        const size_t T = (size_t)n * TPP + tr;
        float hm[4][4], vs[4];
        constexpr float NEG = -__builtin_inff();
        constexpr int JR = 4 / XH;                          // channel groups per exchange round
        const size_t pix = (((size_t)n * (a.H >> 1) + ty * NW + wave) * Wp + tx * 8 + 4 * h) * a.Cout + n0 + r;
        float *pbase = a.out + pix;
        const float *rbase = a.resid + pix;
        float rv[4][4];
        if (wave < NW - 1) {        // the tile's last pooled row is finished (and gets its residual) in pool_fix_add_kernel
#pragma unroll
            for (int jn = 0; jn < 4; jn++)
#pragma unroll
                for (int q = 0; q < 4; q++) rv[jn][q] = (q < 3 || h == 0) ? rbase[(size_t)q * a.Cout + jn * 32] : 0.f;     // so is column 7
        }
#pragma unroll
        for (int xr = 0; xr < XH; xr++) {
        if (xr) __syncthreads();                            // the previous round's values have been read
#pragma unroll
        for (int jn = xr * JR; jn < (xr + 1) * JR; jn++) {
            const int jx = jn - xr * JR;                    // channel group inside the round's exchange buffer
            const float sc = a.scale[n0 + jn * 32 + r], sh = a.shift[n0 + jn * 32 + r];
            float v[16];
#pragma unroll
            for (int e = 0; e < 16; e++) {
                v[e] = fmaf(acc[jn][e], sc, sh);
                if (a.relu_out) v[e] = fmaxf(v[e], 0.f);
                acc[jn][e] = 0.f;
            }
            // columns 8 h + k, k = 0..7: own register 8 h + k; the other half's register of the same index holds the other row
            float y0[9], y1[9];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const float own = h ? v[8 + k] : v[k];
                const float oth = __shfl_xor(h ? v[k] : v[8 + k], 32);      // the other half sends what this one lacks
                y0[k] = k < 4 ? own : oth;                                   // quads 0, 2: this half holds y0; quads 1, 3: y1
                y1[k] = k < 4 ? oth : own;
            }
            {   // column 8 h + 8: x = 8 for h = 0 (register 8: this half holds its y1, the other half its y0), none for h = 1
                const float o8 = __shfl_xor(v[8], 32);
                y0[8] = h ? NEG : o8;
                y1[8] = h ? NEG : v[8];
            }
            float m01[9], h0[4];
#pragma unroll
            for (int k = 0; k < 9; k++) m01[k] = fmaxf(y0[k], y1[k]);
#pragma unroll
            for (int q = 0; q < 4; q++) {                   // pooled column 4 h + q
                hm[jn][q] = fmaxf(fmaxf(m01[2 * q], m01[2 * q + 1]), m01[2 * q + 2]);
                h0[q] = fmaxf(fmaxf(y0[2 * q], y0[2 * q + 1]), y0[2 * q + 2]);
                xch[(wave * 9 + 4 * h + q) * XC + jx * 32 + r] = h0[q];
                if (wave == 0) a.strip_h[(T * 8 + 4 * h + q) * a.Cout + n0 + jn * 32 + r] = h0[q];
            }
            vs[jn] = m01[0];                                // meaningful in half 0 (column 0)
            if (h == 0) {
                xch[(wave * 9 + 8) * XC + jx * 32 + r] = y0[0];
                if (wave == 0) a.corner[T * a.Cout + n0 + jn * 32 + r] = y0[0];
            }
        }
        __syncthreads();
#pragma unroll
        for (int jn = xr * JR; jn < (xr + 1) * JR; jn++) {
            const int jx = jn - xr * JR;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float pv = hm[jn][q];
                if (wave < NW - 1) {
                    pv = fmaxf(pv, xch[((wave + 1) * 9 + 4 * h + q) * XC + jx * 32 + r]);
                    if (q < 3 || h == 0) pv = pv + rv[jn][q];
                }
                pbase[(size_t)q * a.Cout + jn * 32] = pv;
            }
            if (h == 0) {
                float cv = vs[jn];
                if (wave < NW - 1) cv = fmaxf(cv, xch[((wave + 1) * 9 + 8) * XC + jx * 32 + r]);
                a.strip_v[(T * NW + wave) * a.Cout + n0 + jn * 32 + r] = cv;
            }
        }
        }
    };

#ifndef SEP_SPREAD
#define SEP_SPREAD 0
#endif
#ifdef SEP_ABL_NODMA
#define SEP_NODMA 1
#else
#define SEP_NODMA 0
#endif
#ifdef SEP_DIAG           // diagnostic build: s_memtime stamps at the step's phase boundaries, summed per wave (perturbs the LDS prefetch slightly)
#define SEP_STAMP(k) { const long long now_ = (long long)__builtin_readcyclecounter(); if (k) dsum[k] += now_ - dlast; dlast = now_; }
#else
#define SEP_STAMP(k)
#endif
#ifdef SEP_ABL_VMCNT      // timing ablation only (racy): leave the newest DMA group in flight across the step barrier
#define SEP_STEP_WAIT() asm volatile("s_waitcnt vmcnt(%0)" :: "n"(SEP_ABL_VMCNT) : "memory");
#else
#define SEP_STEP_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#define SEP_STEP(SA, SB, SC)                                                                                    \
    {                                                                                                           \
        const char *sa = reinterpret_cast<const char *>(SA);                                                    \
        const char *sb = reinterpret_cast<const char *>(SB);                                                    \
        SEP_STAMP(0)                                                                                            \
        const bool doiss = dleft > 0;                                                                           \
        if (SEP_SPREAD == 0 && doiss && !SEP_NODMA) SEP_ISSUE(SC) else if (SEP_SPREAD == 0 && doiss) { dleft--; if (++dc == nchunks) { dc = 0; dj++; } }                                                             \
        SEP_PIN()                                                                                               \
        SEP_STAMP(1)                                                                                            \
        a1 = make_float4(0.f, 0.f, 0.f, 0.f);                                                                   \
        /* seg 0 */ SEP_RD(sa, tvB, twB, 1, 3) SEP_RB(sa, bqb, 0, 1) SEP_MM(a0.x, bqa) SEP_FM(a1, tvA, twA) SEP_MIX(NV) SEP_PIN() \
        if (SEP_SPREAD && doiss) { SEP_ISSUE_PART(SC, 0) SEP_PIN() }                                            \
        /* seg 1 */ SEP_RD(sa, tvA, twA, 1, 6) SEP_RB(sa, bqa, 0, 2) SEP_MM(a0.y, bqb) SEP_FM(a1, tvB, twB) SEP_MIX(NV) SEP_PIN() \
        if (SEP_SPREAD && doiss) { SEP_ISSUE_PART(SC, 1) SEP_PIN() }                                            \
        /* seg 2 */ SEP_RB(sa, bqb, 0, 3) SEP_MM(a0.z, bqa) SEP_FM(a1, tvA, twA) SEP_MIX(NV) SEP_PIN()          \
        if (SEP_SPREAD && doiss) { SEP_ISSUE_PART(SC, 2) SEP_PIN() }                                            \
        /* seg 3 */ SEP_RD(sb, tvB, twB, 0, 0) SEP_RB(sa, bqa, 1, 0) SEP_MM(a0.w, bqb) SEP_PIN()                \
        if (SEP_SPREAD && doiss) { SEP_ISSUE_PART(SC, 3) SEP_PIN() }                                            \
        a0 = make_float4(0.f, 0.f, 0.f, 0.f);       /* from here on: the A fragment of the NEXT step's k-quad 0 */ \
        /* seg 4 */ SEP_RD(sb, tvA, twA, 0, 3) SEP_RB(sa, bqb, 1, 1) SEP_MM(a1.x, bqa) SEP_FM(a0, tvB, twB) SEP_MIX(NV) SEP_PIN() \
        if (SEP_SPREAD && doiss) { SEP_ISSUE_PART(SC, 4) SEP_PIN() }                                            \
        /* seg 5 */ SEP_RD(sb, tvB, twB, 0, 6) SEP_RB(sa, bqa, 1, 2) SEP_MM(a1.y, bqb) SEP_FM(a0, tvA, twA) SEP_MIX(NV) SEP_PIN() \
        /* seg 6 */ SEP_RD(sb, tvA, twA, 1, 0) SEP_RB(sa, bqb, 1, 3) SEP_MM(a1.z, bqa) SEP_FM(a0, tvB, twB) SEP_MIX(NV) SEP_PIN() \
        /* seg 7 */ SEP_RB(sb, bqa, 0, 0) SEP_MM(a1.w, bqb) SEP_PIN()                                           \
        SEP_STAMP(2)                                                                                            \
        SEP_STEP_WAIT()                                                                                         \
        SEP_STAMP(3)                                                                                            \
        __syncthreads();                                                                                        \
        SEP_STAMP(4)                                                                                            \
        if (++cc == nchunks) { cc = 0; if (POOL) store_tile_pool(cj); else store_tile(cj); cj++; }                                                  \
        SEP_STAMP(5)                                                                                            \
        if (--left == 0) break;                                                                                 \
    }

    // ---- prologue: steps 0 and 1 in flight, then the A fragment of step 0's k-quad 0, the first round of its k-quad 1
    // and the B values of its first k step
    SEP_ISSUE(stage0)
    if (dleft > 0) SEP_ISSUE(stage1)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    {
        const char *s0 = reinterpret_cast<const char *>(stage0);
        SEP_RD(s0, tvA, twA, 0, 0) SEP_FM(a0, tvA, twA)
        SEP_RD(s0, tvA, twA, 0, 3) SEP_FM(a0, tvA, twA)
        SEP_RD(s0, tvA, twA, 0, 6) SEP_FM(a0, tvA, twA)
        SEP_PIN()
        SEP_RD(s0, tvA, twA, 1, 0) SEP_RB(s0, bqa, 0, 0)
        SEP_PIN()
    }

    int cj = j0, cc = 0, left = total;      // compute iterator: pair, chunk, steps left
#ifdef SEP_DIAG
    long long dsum[6] = {0, 0, 0, 0, 0, 0}, dlast = 0;
#endif
#ifdef SEP_VAR_PRIO
    if (wave >= 4) __builtin_amdgcn_s_setprio(SEP_VAR_PRIO);       // static priority for the younger half (arbitration loser)
#endif
    for (;;) {
#pragma unroll
        for (int tp = 0; tp < 9; tp++) asm volatile("" : "+v"(hoff[tp]));
        asm volatile("" : "+v"(boff), "+v"(doff));
        SEP_STEP(stage0, stage1, stage2)
        SEP_STEP(stage1, stage2, stage0)
        SEP_STEP(stage2, stage0, stage1)
    }
#ifdef SEP_DIAG
    if (a.diag && lane == 0) {
        dsum[0] = total;
        for (int k = 0; k < 6; k++) a.diag[((size_t)blockIdx.x * 8 + wave) * 6 + k] = dsum[k];
    }
#endif
#undef SEP_STEP
#undef SEP_MIX
#undef SEP_RD
#undef SEP_FM
#undef SEP_RB
#undef SEP_MM
#undef SEP_PIN
#undef SEP_ISSUE
}

bool sepconv_supported(int H, int W, int Cin, int Cout)
{
    return H % 16 == 0 && W % 16 == 0 && Cin % 16 == 0 && Cout % 128 == 0 &&
           (long long)H * W * Cin * 4 < 0x7fffffffLL && (long long)16 * W * Cout * 4 < 0x7fffffffLL;
}

static int sep_cus()
{
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount < 8) n_cu = 256;
        else n_cu = prop.multiProcessorCount;
    }
    return n_cu;
}

// waves per workgroup of the launches: 8 = one 16 x 16 tile workgroup per CU, 4 = two 8 x 16 tile workgroups per CU (TMAT_SEP_WAVES)
static int sep_waves()
{
    static int nw = 0;
    if (!nw) { const char *e = getenv("TMAT_SEP_WAVES"); nw = (e && atoi(e) == 8) ? 8 : (e && atoi(e) == 4) ? 4 : SEP_DEFAULT_WAVES; }
    return nw;
}

template <bool POOL>
static void launch_sep_any(const SepArgs &a, int relu_in, int nw, hipStream_t s)
{
    const int TR = 2 * nw;
    const int nMt = a.N * (a.H / TR) * (a.W / 16), nNt = a.Cout / 128;
    const int G = (sep_cus() / 8) * 8 * (8 / nw);       // persistent workgroups: one (8 waves) or two (4 waves) per CU
    if (nw == 8) {
        if (relu_in) hipLaunchKernelGGL((sepconv_mfma_kernel<true, POOL, 8>), dim3(G), dim3(512), 0, s, a, nMt, nNt, G);
        else hipLaunchKernelGGL((sepconv_mfma_kernel<false, POOL, 8>), dim3(G), dim3(512), 0, s, a, nMt, nNt, G);
    } else {
        if (relu_in) hipLaunchKernelGGL((sepconv_mfma_kernel<true, POOL, 4>), dim3(G), dim3(256), 0, s, a, nMt, nNt, G);
        else hipLaunchKernelGGL((sepconv_mfma_kernel<false, POOL, 4>), dim3(G), dim3(256), 0, s, a, nMt, nNt, G);
    }
}

// in (N, H, W, Cin) -> out (N, H, W, Cout): depthwise 3x3 (taps dwq [Cin/16][9][16], optional ReLU on load) ->
// pointwise (pw [Cin][Cout]) -> fmaf(acc, scale, shift) -> optional ReLU
bool launch_sepconv(const float *in, int N, int H, int W, int Cin, int relu_in, const float *dwq, const float *pw, int Cout,
                    const float *scale, const float *shift, int relu_out, float *out, hipStream_t s)
{
    if (!sepconv_supported(H, W, Cin, Cout) || N <= 0 || (long long)N * (H / 8) * (W / 16) * (Cout / 128) > 0x3fffffffLL) {
        set_error("launch_sepconv: unsupported shape");
        return false;
    }
    SepArgs a{in, N, H, W, Cin, Cout, dwq, pw, scale, shift, relu_out, out, nullptr, nullptr, nullptr, nullptr, nullptr};
    const int nw = sep_waves();
#ifdef SEP_DIAG
    const int G = (sep_cus() / 8) * 8 * (8 / nw);
    static long long *diag = nullptr;
    if (!diag) hipMalloc((void **)&diag, (size_t)G * 8 * 6 * 8);
    a.diag = diag;
#endif
    launch_sep_any<false>(a, relu_in, nw, s);
#ifdef SEP_DIAG
    {
        std::vector<long long> h((size_t)G * 8 * 6);
        hipStreamSynchronize(s);
        hipMemcpy(h.data(), diag, h.size() * 8, hipMemcpyDeviceToHost);
        double sum[6] = {0, 0, 0, 0, 0, 0}, steps = 0;
        for (int b = 0; b < G; b++) for (int w = 0; w < nw; w++) { steps += (double)h[((size_t)b * 8 + w) * 6]; for (int k = 1; k < 6; k++) sum[k] += (double)h[((size_t)b * 8 + w) * 6 + k]; }
        fprintf(stderr, "[sepdiag] H %d Cin %d Cout %d relu %d: per step and wave (s_memtime ticks): issue %.0f segments %.0f vmcnt %.0f barrier %.0f epilogue %.0f | steps/wave %.0f\n",
                H, Cin, Cout, relu_in, sum[1] / steps, sum[2] / steps, sum[3] / steps, sum[4] / steps, sum[5] / steps, steps / (G * (double)nw));
        double bw[8] = {0}, sg[8] = {0}, st[8] = {0};
        for (int b = 0; b < G; b++) for (int w = 0; w < nw; w++) { st[w] += (double)h[((size_t)b * 8 + w) * 6]; bw[w] += (double)h[((size_t)b * 8 + w) * 6 + 4]; sg[w] += (double)h[((size_t)b * 8 + w) * 6 + 2]; }
        fprintf(stderr, "[sepdiag]   by wave: segments");
        for (int w = 0; w < nw; w++) fprintf(stderr, " %.0f", sg[w] / st[w]);
        fprintf(stderr, " | barrier");
        for (int w = 0; w < nw; w++) fprintf(stderr, " %.0f", bw[w] / st[w]);
        fprintf(stderr, "\n");
    }
#endif
    return true;
}

// Finishes the pooling the separable convolution started in its epilogue: the partial pooled values of a tile's last row /
// column take the missing row / column from the strips of the tile below / to the right (nothing at the patch border: TF pads
// with -inf) and get their residual, in place.  A tile has PR x 8 pooled pixels (PR = waves per workgroup of the convolution);
// one thread = 4 channels of one of its 8 + PR - 1 boundary pixels.
__global__ __launch_bounds__(256) void pool_fix_add_kernel(float *__restrict__ out, const float *__restrict__ strip_h, const float *__restrict__ strip_v,
                                                           const float *__restrict__ corner, const float *__restrict__ resid, int Hp, int Wp, int C,
                                                           int c4shift, int total, int PR)
{
    const int n = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    const int cq = e & ((1 << c4shift) - 1);
    const int bp = e >> c4shift;                     // boundary pixel: tile * NB + k
    const int NB = 8 + PR - 1;
    const int tile = bp / NB, k = bp - tile * NB;
    const int TH = Hp / PR, TW = Wp >> 3, ty = tile / TW, tx = tile - ty * TW;
    const int pl = k < 8 ? PR - 1 : k - 8, ql = k < 8 ? k : 7;
    const size_t o = (((size_t)n * Hp + ty * PR + pl) * Wp + tx * 8 + ql) * C + cq * 4;
    float4 m = *reinterpret_cast<const float4 *>(out + o);
    auto take = [&](const float *src) {
        const float4 v = *reinterpret_cast<const float4 *>(src + cq * 4);
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
    };
    const bool below = pl == PR - 1 && ty + 1 < TH, right = ql == 7 && tx + 1 < TW;
    if (below) take(strip_h + ((((size_t)n * TH + ty + 1) * TW + tx) * 8 + ql) * C);
    if (right) take(strip_v + ((((size_t)n * TH + ty) * TW + tx + 1) * PR + pl) * C);
    if (below && right) take(corner + (((size_t)n * TH + ty + 1) * TW + tx + 1) * C);
    const float4 rv = *reinterpret_cast<const float4 *>(resid + o);
    m.x = m.x + rv.x; m.y = m.y + rv.y; m.z = m.z + rv.z; m.w = m.w + rv.w;
    *reinterpret_cast<float4 *>(out + o) = m;
}

// finishes the tile edges of a pooled separable convolution (tiles of 2 nw rows x 16 columns; Cout / 4 a power of two)
void launch_pool_fix_add(float *out, const float *sh, const float *sv, const float *co, const float *resid, int N, int H, int W, int Cout, int nw,
                         hipStream_t s)
{
    int c4shift = 0;
    while ((1 << c4shift) < Cout / 4) c4shift++;
    const int total = (H / (2 * nw)) * (W / 16) * (8 + nw - 1) * (Cout / 4);
    hipLaunchKernelGGL(pool_fix_add_kernel, dim3((total + 255) / 256, N), dim3(256), 0, s, out, sh, sv, co, resid, H / 2, W / 2, Cout, c4shift, total, nw);
}

size_t sepconv_pool_scratch_floats(int N, int H, int W, int Cout)
{
    return (size_t)N * (H / 8) * (W / 16) * 17 * Cout;           // strips: 8 + PR + 1 pixels per tile, sized for the smaller tile
}

// The second separable convolution of a down block with MaxPooling2D(3, 2, "same") and the residual add fused behind it
// (models.py:134-144): in (N, H, W, Cin) -> out (N, H/2, W/2, Cout) = maxpool(bn(sepconv(in))) + resid.  `scratch` holds
// sepconv_pool_scratch_floats(...) floats (the tile-edge strips); the full-resolution tensor is never written.
bool launch_sepconv_pool(const float *in, int N, int H, int W, int Cin, int relu_in, const float *dwq, const float *pw, int Cout,
                         const float *scale, const float *shift, int relu_out, float *scratch, const float *resid, float *out, hipStream_t s)
{
    if (!sepconv_supported(H, W, Cin, Cout) || N <= 0 || (long long)N * (H / 8) * (W / 16) * (Cout / 128) > 0x3fffffffLL || Cout % 4) {
        set_error("launch_sepconv_pool: unsupported shape");
        return false;
    }
    const int nw = sep_waves();
    const size_t tiles = (size_t)N * (H / (2 * nw)) * (W / 16);
    float *sh = scratch, *sv = sh + tiles * 8 * Cout, *co = sv + tiles * nw * Cout;
    SepArgs a{in, N, H, W, Cin, Cout, dwq, pw, scale, shift, relu_out, out, resid, sh, sv, co, nullptr};
    if ((Cout / 4) & (Cout / 4 - 1)) { set_error("launch_sepconv_pool: Cout / 4 must be a power of two"); return false; }
    launch_sep_any<true>(a, relu_in, nw, s);
    launch_pool_fix_add(out, sh, sv, co, resid, N, H, W, Cout, nw, s);
    return true;
}

}  // namespace tmat
