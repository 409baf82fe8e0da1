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
#include "tmat_internal.h"
#include "../../include/tmat.h"

namespace tmat {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_t;

// Geometry: one workgroup = 512 threads = 8 waves = a 16 x 16 pixel tile (M = 256) x 128 output channels.
// Wave w owns tile rows 2w, 2w+1 (32 pixels) x 128 channels = four 32x32 accumulator tiles (64 VGPRs).
// K chunk = 16 input channels = two 8-channel planes (g = 0, 1: the channel groups of the two MFMA k-quads).  LDS per
// stage, in 16-byte cells:
//   halo : plane g at cell g * 704: 18 x 18 pixels x 2 cells (channel quads h = 0, 1), linear pitch 18; the cell of
//          (pixel p, quad h) is 2 p + (h ^ ((p >> 3) & 1)); 11 DMA pieces of 64 cells per plane
//   B    : plane g at cell g * 256: 128 output channels x 2 cells, cell of (row n, quad h) is 2 n + (h ^ ((n >> 3) & 1));
//          4 DMA pieces per plane
//   dw   : 9 taps x 4 cells (the depthwise taps of this channel block, [tap][16 channels]); 1 DMA piece
// Every 16-lane service group of a ds_read_b128 ({0-3,12-15,20-27}, {4-11,16-19,28-31} and the same + 32) must hit 16
// distinct cells mod 16, i.e. 16 pixels (rows) that are distinct mod 16.  For B the rows of a group are distinct mod 16.
// For the halo the MFMA row index r = lane & 31 is mapped to tile pixels so that each service group is 16 consecutive
// pixels of ONE tile row (PIXMAP below): their halo positions p are consecutive for every tap.
// The plane split keeps g out of the swizzle: the g = 1 fragment sits at a constant byte offset from the g = 0 one
// (an instruction immediate, no second set of address registers).
// Everything reaches LDS by LDS-DMA (buffer_load_dwordx4 ... lds, lane-linear destination; the swizzle is applied to
// the per-lane SOURCE address); the only synchronisation is vmcnt(0) + one barrier per chunk, two stages.
constexpr int SEP_KC = 16;
constexpr int SEP_HPLANE = 11 * 64;                    // cells per halo plane
constexpr int SEP_HALO_CELLS = 22 * 64;
constexpr int SEP_B_CELLS = 8 * 64;
constexpr int SEP_DW_CELLS = 64;
// stage = [B | halo | dw]; 31 KiB, so that every (stage base + constant) of stage 1 fits the 16-bit ds_read offset field
constexpr int SEP_B_OFF = 0, SEP_H_OFF = SEP_B_CELLS * 4, SEP_D_OFF = (SEP_B_CELLS + SEP_HALO_CELLS) * 4;      // float offsets
constexpr int SEP_STAGE_FLOATS = (SEP_HALO_CELLS + SEP_B_CELLS + SEP_DW_CELLS) * 4;

struct SepArgs {
    const float *in;      // (N, H, W, Cin)
    int N, H, W, Cin, Cout;
    const float *dwq;     // depthwise taps, [Cin / 16][9][16]
    const float *pw;      // pointwise weights [Cout][Cin]
    const float *scale, *shift;
    int relu_out;
    float *out;           // (N, H, W, Cout)
};

template <bool RELU_IN>
// 2 workgroups (16 waves) per CU: second launch-bounds argument = waves per SIMD = 4 (<= 128 VGPRs)
__global__ __launch_bounds__(512, 4) void sepconv_mfma_kernel(SepArgs a, int nMt, int nNt, int tiles_per_xcd)
{
    __shared__ __attribute__((aligned(16))) float stage0[SEP_STAGE_FLOATS];
    __shared__ __attribute__((aligned(16))) float stage1[SEP_STAGE_FLOATS];

    // XCD-aware mapping: blocks b and b + 8 share an XCD; every XCD walks a contiguous range of pixel tiles (with their
    // nNt channel tiles back to back), so the overlapping halos of neighbouring tiles are served by that XCD's L2.
    const int b = blockIdx.x;
    const int xcd = b & 7, j = b >> 3;
    const int nt = j % nNt;
    const int mt = xcd * tiles_per_xcd + j / nNt;
    if (mt >= nMt) return;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int TW = a.W >> 4, TPP = (a.H >> 4) * TW;          // tiles per row / per patch
    const int n = mt / TPP, tr = mt - n * TPP;
    const int ty0 = (tr / TW) * 16, tx0 = (tr % TW) * 16;
    const int n0 = nt * 128;
    const int Cin = a.Cin;
    const int nchunks = Cin / SEP_KC;
    constexpr unsigned OOB = 0x80000000u;

    // ---- DMA roles --------------------------------------------------------------------------------------------
    const __amdgpu_buffer_rsrc_t rsA =
        __builtin_amdgcn_make_buffer_rsrc((void *)(a.in + (size_t)n * a.H * a.W * Cin), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB =
        __builtin_amdgcn_make_buffer_rsrc((void *)(a.pw + (size_t)n0 * Cin), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc((void *)a.dwq, 0, 0x7fffffff, 0x00020000);
    unsigned hv[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int q = (i * 8 + wave) * 64 + lane;            // destination cell of this lane
        const int g = q >= SEP_HPLANE ? 1 : 0;
        const int qq = q - g * SEP_HPLANE;
        const int p = qq >> 1, d = qq & 1;
        const int hy = p / 18, hx = p - hy * 18;
        const int Y = ty0 - 1 + hy, X = tx0 - 1 + hx;
        const bool ok = p < 18 * 18 && Y >= 0 && Y < a.H && X >= 0 && X < a.W;
        const int sg = 2 * g + (d ^ ((p >> 3) & 1));
        hv[i] = ok ? (unsigned)((Y * a.W + X) * Cin + sg * 4) * 4u : OOB;
    }
    unsigned bvo;
    {
        const int q = wave * 64 + lane;
        const int g = q >> 8, qq = q & 255;
        const int nrow = qq >> 1, d = qq & 1;
        const int sg = 2 * g + (d ^ ((nrow >> 3) & 1));
        bvo = (unsigned)(nrow * Cin + sg * 4) * 4u;
    }
    const unsigned dvo = lane < 36 ? (unsigned)lane * 16u : OOB;

#define SEP_ISSUE(stage_, chunk_)                                                                               \
    {                                                                                                           \
        float *st = (stage_);                                                                                   \
        const int so = (chunk_) * (SEP_KC * 4);                                                                 \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t *)(st + SEP_H_OFF + (0 * 8 + wave) * 256), 16, hv[0], so, 0, 0); \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t *)(st + SEP_H_OFF + (1 * 8 + wave) * 256), 16, hv[1], so, 0, 0); \
        if (wave < 6)                                                                                           \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t *)(st + SEP_H_OFF + (2 * 8 + wave) * 256), 16, hv[2], so, 0, 0); \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_t *)(st + SEP_B_OFF + wave * 256), 16, bvo, so, 0, 0); \
        if (wave == 6 + ((chunk_) & 1))                                                                         \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsD, (lds_void_t *)(st + SEP_D_OFF), 16, dvo, (chunk_) * 576, 0, 0); \
    }

    // ---- fragment roles ---------------------------------------------------------------------------------------
    // PIXMAP: MFMA row r = lane & 31 -> (tile row 2 wave + yl, column x): rows of one ds_read_b128 service group are 16
    // consecutive pixels of one tile row.
    const int r = lane & 31, h = lane >> 5;
    const int yl = ((r >= 4 && r < 12) || (r >= 16 && r < 20) || r >= 28) ? 1 : 0;
    const int x = r < 4 ? r : r < 12 ? r - 4 : r < 16 ? r - 8 : r < 20 ? r - 8 : r < 28 ? r - 12 : r - 16;
    int hoff[9];            // BYTE offset of this lane's cell (quad h of plane 0) per tap; plane 1 is + SEP_HPLANE cells
#pragma unroll
    for (int tp = 0; tp < 9; tp++) {
        const int p = (2 * wave + yl + tp / 3) * 18 + x + tp % 3;
        hoff[tp] = (SEP_H_OFF + (p * 2 + (h ^ ((p >> 3) & 1))) * 4) * 4;
    }
    int boff = (SEP_B_OFF + (r * 2 + (h ^ ((r >> 3) & 1))) * 4) * 4;     // bytes; + jn * 1024 B, plane 1: + 4096 B
    int doff = (SEP_D_OFF + h * 4) * 4;                  // bytes; + tap * 64 B, g = 1: + 32 B

    f32x16 acc[4];
#pragma unroll
    for (int jn = 0; jn < 4; jn++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[jn][e] = 0.f;

// One tap at a time (read the pixel's quad and the tap's quad, four FMAs); the loop-invariant lane offsets are made
// opaque once per iteration so that hipcc keeps ONE set of them and folds the stage base and the plane / tap constants
// into the ds_read offset field (hoisted, it materialises base + offset per stage and tap and spills).
#define SEP_TAPS(st, g, t0)                                                                                     \
    _Pragma("unroll") for (int u = 0; u < 3; u++) {                                                             \
        float4 v = *reinterpret_cast<const float4 *>(st + hoff[t0 + u] + g * (SEP_HPLANE * 16));                 \
        const float4 w = *reinterpret_cast<const float4 *>(st + doff + (t0 + u) * 64 + g * 32);                  \
        if (RELU_IN) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); } \
        av.x = fmaf(v.x, w.x, av.x); av.y = fmaf(v.y, w.y, av.y);                                               \
        av.z = fmaf(v.z, w.z, av.z); av.w = fmaf(v.w, w.w, av.w);                                               \
    }
#define SEP_STEP(cur, nxt, c_, more)                                                                            \
    {                                                                                                           \
        const char *st = reinterpret_cast<const char *>(cur);                                                   \
        if (more) SEP_ISSUE(nxt, (c_) + 1)                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        _Pragma("unroll") for (int g = 0; g < 2; g++) {                                                         \
            float4 av = make_float4(0.f, 0.f, 0.f, 0.f);                                                        \
            SEP_TAPS(st, g, 0) SEP_TAPS(st, g, 3) SEP_TAPS(st, g, 6)                                            \
            float4 bv[4];                                                                                       \
            _Pragma("unroll") for (int jn = 0; jn < 4; jn++)                                                    \
                bv[jn] = *reinterpret_cast<const float4 *>(st + boff + g * 4096 + jn * 1024);                 \
            _Pragma("unroll") for (int jn = 0; jn < 4; jn++) acc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv[jn].x, acc[jn], 0, 0, 0); \
            _Pragma("unroll") for (int jn = 0; jn < 4; jn++) acc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv[jn].y, acc[jn], 0, 0, 0); \
            _Pragma("unroll") for (int jn = 0; jn < 4; jn++) acc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv[jn].z, acc[jn], 0, 0, 0); \
            _Pragma("unroll") for (int jn = 0; jn < 4; jn++) acc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv[jn].w, acc[jn], 0, 0, 0); \
            __builtin_amdgcn_sched_barrier(0);                                                                  \
        }                                                                                                       \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                        \
        __syncthreads();                                                                                        \
    }

    SEP_ISSUE(stage0, 0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // even chunks live in stage0, odd ones in stage1 (nchunks is even: host check)
    for (int c = 0; c < nchunks; c += 2) {
#pragma unroll
        for (int tp = 0; tp < 9; tp++) asm volatile("" : "+v"(hoff[tp]));
        asm volatile("" : "+v"(boff), "+v"(doff));
        SEP_STEP(stage0, stage1, c, true)
        SEP_STEP(stage1, stage0, c + 1, c + 2 < nchunks)
    }
#undef SEP_STEP
#undef SEP_TAPS
#undef SEP_ISSUE

    // ---- epilogue: straight from the accumulators.  C/D layout of the 32x32 tile: column = lane & 31 (output channel),
    // row rho = (e & 3) + 8 (e >> 2) + 4 (lane >> 5) (the MFMA row, i.e. PIXMAP's r).  One store instruction writes two
    // full 128-byte lines (32 consecutive channels of two pixels).
    {
        float *obase = a.out + ((size_t)n * a.H * a.W + (size_t)(ty0 + 2 * wave) * a.W + tx0) * a.Cout + n0 + r;
#pragma unroll
        for (int jn = 0; jn < 4; jn++) {
            const float sc = a.scale[n0 + jn * 32 + r], sh = a.shift[n0 + jn * 32 + r];
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const int rho = (e & 3) + 8 * (e >> 2) + 4 * h;
                const int pyl = ((rho >= 4 && rho < 12) || (rho >= 16 && rho < 20) || rho >= 28) ? 1 : 0;
                const int px = rho < 4 ? rho : rho < 12 ? rho - 4 : rho < 16 ? rho - 8 : rho < 20 ? rho - 8 : rho < 28 ? rho - 12 : rho - 16;
                float v = fmaf(acc[jn][e], sc, sh);
                if (a.relu_out) v = fmaxf(v, 0.f);
                obase[((size_t)pyl * a.W + px) * a.Cout + jn * 32] = v;
            }
        }
    }
}

bool sepconv_supported(int H, int W, int Cin, int Cout)
{
    return H % 16 == 0 && W % 16 == 0 && Cin % 32 == 0 && Cout % 128 == 0 && (long long)H * W * Cin * 4 < 0x7fffffffLL;
}

// in (N, H, W, Cin) -> out (N, H, W, Cout): depthwise 3x3 (taps dwq [Cin/16][9][16], optional ReLU on load) ->
// pointwise (pw [Cout][Cin]) -> fmaf(acc, scale, shift) -> optional ReLU
bool launch_sepconv(const float *in, int N, int H, int W, int Cin, int relu_in, const float *dwq, const float *pw, int Cout,
                    const float *scale, const float *shift, int relu_out, float *out, hipStream_t s)
{
    if (!sepconv_supported(H, W, Cin, Cout) || N <= 0 || (long long)N * (H / 16) * (W / 16) > 0x3fffffffLL / 8) {
        set_error("launch_sepconv: unsupported shape");
        return false;
    }
    SepArgs a{in, N, H, W, Cin, Cout, dwq, pw, scale, shift, relu_out, out};
    const int nMt = N * (H / 16) * (W / 16), nNt = Cout / 128;
    const int tpx = (nMt + 7) / 8;
    dim3 grid(8 * tpx * nNt);
    if (relu_in)
        hipLaunchKernelGGL(sepconv_mfma_kernel<true>, grid, dim3(512), 0, s, a, nMt, nNt, tpx);
    else
        hipLaunchKernelGGL(sepconv_mfma_kernel<false>, grid, dim3(512), 0, s, a, nMt, nNt, tpx);
    return true;
}

}  // namespace tmat
