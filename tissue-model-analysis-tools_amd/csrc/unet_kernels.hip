// UNet-Xception inference kernels for gfx950 (MI355X).  NHWC float32.
//
// Reference graph: fl_tissue_model_tools/models.py:110-166 (build_UNetXception), executed by
// keras Model.predict at smooth_tiled_predictions.py:179.
//
// Arithmetic contract (shared with oracle/unet_exact.c, compared bit-exactly):
//   every contraction is a chain acc = fmaf(a[k], w[k], acc); for the MFMA convolutions k walks the
//   input channels in blocks of 32, inside a block tap-major (ky, kx), then channel ascending
//   (depthwise / stem / final: tap-major, channel ascending); v_mfma_f32_32x32x2_f32 performs exactly
//   such a chain (one rounding per product, k ascending), so the dense 3x3 / 1x1 contractions run on
//   the matrix cores at full f32 precision.  Epilogues: v = fmaf(acc, scale, shift) (folded BN) or acc + bias,
//   optional residual add, optional ReLU.  Compiled with -ffp-contract=off.
#include "tmat_internal.h"
#include "../../include/tmat.h"

#include <cstdlib>

namespace tmat {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------
// implicit-GEMM convolution on MFMA:  C[m][co] = sum_{tap, ci} A(m, tap, ci) * W[tap][ci][co]
//   m = flattened (n, y, x) output pixel.  A is gathered on the fly (zero padding, optional
//   nearest-upsample of the stored input, optional ReLU on load) -> LDS (k-major, padded),
//   W chunk -> LDS.  256 threads = 4 waves arranged WM x WN, each wave owns
//   (BM/WM) x (BN/WN) outputs as TM x TN tiles of 32x32 (16 accumulator VGPRs each).
//   Register-prefetch double buffering: one barrier per K chunk.
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int KC, int KS>
__global__ __launch_bounds__(256, (BN == 64 ? 3 : 2)) void conv_mfma_kernel(ConvArgs a, int M, int Ho, int Wo, int nMt, int nNt)
{
    // KS (1, 2 or 3) is a template parameter so that the 3x3 and the pointwise instantiations are distinct kernels
    // (distinct names in rocprofv3 traces: the 3x3 <128,128,2,2,32,3> instantiation is the dominant kernel).
    // KS == 2 is the sub-pixel form of a 3x3 convolution over a 2x nearest-upsampled input: output pixel
    // (2i + py, 2j + px) only sees the 2x2 stored pixels (i + py - 1 + {0,1}, j + px - 1 + {0,1}), with the 3x3 taps that
    // fall on the same stored pixel pre-summed per parity class (weights [class][tap][Cin][Cout], class = blockIdx.y,
    // M enumerates the stored pixels).  4/9 of the multiply-adds of the as-written form.
    static_assert(WM * WN == 4, "4 waves");
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int LDA = BM + 1;
    constexpr int TPP = KC / 4;          // threads per pixel row of the A chunk (float4 each)
    constexpr int PPP = 256 / TPP;       // pixels per staging pass
    constexpr int NPA = BM / PPP;        // A passes
    constexpr int BV = BN / 4;           // float4 per B row
    constexpr int RPP = 256 / BV;        // B rows per pass
    constexpr int NPB = KC / RPP;        // B passes
    static_assert(NPA >= 1 && NPB >= 1, "tile config");

    // one LDS array: [2][KC*BN] weight chunks (16-byte aligned), then [2][KC*LDA] pixel chunks; the epilogue
    // reuses it as a [128][BN] output staging tile
    constexpr int EPR = 128;
    constexpr int SM_MAIN = 2 * KC * BN + 2 * KC * LDA;
    constexpr int SM = SM_MAIN > EPR * BN ? SM_MAIN : EPR * BN;
    __shared__ __attribute__((aligned(16))) float smem[SM];
    float(*Bs)[KC * BN] = reinterpret_cast<float(*)[KC * BN]>(smem);
    float(*As)[KC * LDA] = reinterpret_cast<float(*)[KC * LDA]>(smem + 2 * KC * BN);

    // XCD-aware tile mapping: blocks b and b+8 share an XCD (round-robin dispatch); give the
    // nNt column tiles of one pixel tile to the same XCD so its L2 serves the re-read A pixels.
    const int b = blockIdx.x;
    // Co-resident blocks of one CU (2 per CU here) are dispatched together and would run in lockstep: both in their
    // staging / barrier phase at the same time, both competing for the MFMA pipe at the same time.  Delaying every
    // second generation of 256 blocks by about half a K-chunk period keeps the pairs out of phase for the whole launch.
    if ((b >> 8) & 1)
        for (int i = 0; i < (a.stagger & 255); i++) __builtin_amdgcn_s_sleep(8);
    const int xcd = b & 7, j = b >> 3;
    const int nt = j % nNt;
    const int mt = (j / nNt) * 8 + xcd;
    if (mt >= nMt) return;
    const int m0 = mt * BM, n0 = nt * BN;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;

    const int H = a.h << a.up, W = a.w << a.up;
    const int spy = KS == 2 ? (int)(blockIdx.y >> 1) : 0, spx = KS == 2 ? (int)(blockIdx.y & 1) : 0;
    constexpr int taps = KS * KS;
    const int cchunks = a.Cin / KC;
    const int nchunks = taps * cchunks;

    // per-thread pixel bookkeeping for A staging: pointer to the centre tap's pixel, a 9-bit mask of the taps that fall
    // inside the image, and (for nearest-upsampled inputs) the parities of y and x
    const int c4 = (t % TPP) * 4;
    const float *pc[NPA];
    unsigned pm[NPA];
#pragma unroll
    for (int i = 0; i < NPA; i++) {
        const int m = m0 + i * PPP + t / TPP;
        const bool ok = m < M;
        const int mm = ok ? m : 0;
        const int n = mm / (Ho * Wo);
        const int r = mm - n * (Ho * Wo);
        const int yo = r / Wo;
        const int y = yo * a.stride, x = (r - yo * Wo) * a.stride;
        pc[i] = a.in + ((size_t)(n * a.h + (y >> a.up)) * a.w + (x >> a.up)) * a.Cin + c4;
        unsigned msk = 0;
        if (ok) {
            if (KS == 3) {
#pragma unroll
                for (int tp = 0; tp < 9; tp++) {
                    const int yy = y + tp / 3 - 1, xx = x + tp % 3 - 1;
                    if (yy >= 0 && yy < H && xx >= 0 && xx < W) msk |= 1u << tp;
                }
            } else if (KS == 2) {
#pragma unroll
                for (int tp = 0; tp < 4; tp++) {
                    const int yy = y + spy - 1 + (tp >> 1), xx = x + spx - 1 + (tp & 1);
                    if (yy >= 0 && yy < H && xx >= 0 && xx < W) msk |= 1u << tp;
                }
            } else msk = 1u;
        }
        pm[i] = msk | ((unsigned)(y & 1) << 9) | ((unsigned)(x & 1) << 10);
    }
    const int brow = t / BV, bcol = (t % BV) * 4;

    // Loads are unconditional (out-of-image taps read a valid dummy address and are zeroed when the
    // registers are written to LDS), so the next chunk's global loads stay in flight across the MFMA loop.
    // Two register sets (P, Q) hold the chunks c+1 and c+2 while chunk c is multiplied: a chunk's global loads
    // have two MFMA phases (~8k cycles) to land before they are written to LDS.
    // Loader state machine (chunks are loaded strictly in order): per staged pixel a base pointer for the current
    // tap -- the input pixel, or a zero row for out-of-image taps, so the loads are unconditional and need no
    // select afterwards -- recomputed only when the tap changes; the weight pointer just advances by KC rows
    // (W is [tap][Cin][Cout], i.e. chunk after chunk is contiguous).
    float4 pa0, pa1, pa2, pa3, pa4, pa5, pa6, pa7, pb0, pb1, pb2, pb3;
    float4 qa0, qa1, qa2, qa3, qa4, qa5, qa6, qa7, qb0, qb1, qb2, qb3;
    static_assert(NPA == 4 || NPA == 8 || NPA == 2, "A passes");
    static_assert(NPB == 4 || NPB == 2 || NPB == 1, "B passes");
    const float *ab0 = a.zeros, *ab1 = a.zeros, *ab2 = a.zeros, *ab3 = a.zeros, *ab4 = a.zeros, *ab5 = a.zeros, *ab6 = a.zeros,
                *ab7 = a.zeros;
    const float *wbase = a.W + (KS == 2 ? (size_t)blockIdx.y * 4 * a.Cin * a.Cout : (size_t)0) + n0 + bcol + (size_t)brow * a.Cout;
    // chunk order: 32-channel block major, then tap, then (KC = 16 only) the two halves of the block
    constexpr int HPB = 32 / KC;            // chunks per (block, tap)
    int ld_cb = 0, ld_tap = 0, ld_half = 0, ld_cc = 0;

#define TMAT_BASE(i, AB)                                                                                    \
    if (i < NPA) {                                                                                          \
        /* tap offset in stored pixels: (dy, dx) itself, or ((parity + d) >> 1) for a nearest-upsampled input */ \
        const int ddy = a.up ? (((int)((pm[i] >> 9) & 1u) + dy) >> 1) : dy;                                 \
        const int ddx = a.up ? (((int)((pm[i] >> 10) & 1u) + dx) >> 1) : dx;                                \
        const bool ok = (pm[i] >> ld_tap) & 1u;                                                             \
        AB = ok ? pc[i] + (ddy * a.w + ddx) * a.Cin : a.zeros + c4;                                         \
    }
#define TMAT_LOAD_CHUNK(S)                                                             \
    {                                                                                  \
        if (ld_half == 0 && ld_cc < nchunks) {                                         \
            const int dy = KS == 3 ? ld_tap / 3 - 1 : KS == 2 ? spy - 1 + (ld_tap >> 1) : 0; \
            const int dx = KS == 3 ? ld_tap % 3 - 1 : KS == 2 ? spx - 1 + (ld_tap & 1) : 0;  \
            TMAT_BASE(0, ab0) TMAT_BASE(1, ab1) TMAT_BASE(2, ab2) TMAT_BASE(3, ab3)    \
            TMAT_BASE(4, ab4) TMAT_BASE(5, ab5) TMAT_BASE(6, ab6) TMAT_BASE(7, ab7)    \
        }                                                                              \
        const int ld_c0 = ld_cb * 32 + ld_half * KC;                                   \
        const float *wp = wbase + ((size_t)ld_tap * a.Cin + ld_c0) * a.Cout;           \
        if (0 < NPA) S##a0 = *reinterpret_cast<const float4 *>(ab0 + ld_c0);           \
        if (1 < NPA) S##a1 = *reinterpret_cast<const float4 *>(ab1 + ld_c0);           \
        if (2 < NPA) S##a2 = *reinterpret_cast<const float4 *>(ab2 + ld_c0);           \
        if (3 < NPA) S##a3 = *reinterpret_cast<const float4 *>(ab3 + ld_c0);           \
        if (4 < NPA) S##a4 = *reinterpret_cast<const float4 *>(ab4 + ld_c0);           \
        if (5 < NPA) S##a5 = *reinterpret_cast<const float4 *>(ab5 + ld_c0);           \
        if (6 < NPA) S##a6 = *reinterpret_cast<const float4 *>(ab6 + ld_c0);           \
        if (7 < NPA) S##a7 = *reinterpret_cast<const float4 *>(ab7 + ld_c0);           \
        if (0 < NPB) S##b0 = *reinterpret_cast<const float4 *>(wp);                    \
        if (1 < NPB) S##b1 = *reinterpret_cast<const float4 *>(wp + (size_t)(1 * RPP) * a.Cout); \
        if (2 < NPB) S##b2 = *reinterpret_cast<const float4 *>(wp + (size_t)(2 * RPP) * a.Cout); \
        if (3 < NPB) S##b3 = *reinterpret_cast<const float4 *>(wp + (size_t)(3 * RPP) * a.Cout); \
        /* the loads are issued unconditionally (a guard would force the compiler to drain the prefetch at every   \
           LDS store); past the last chunk the state simply stops advancing and the last chunk is re-read, unused */ \
        ld_cc++;                                                                       \
        if (ld_cc < nchunks) {                                                         \
            if (++ld_half == HPB) { ld_half = 0; if (++ld_tap == taps) { ld_tap = 0; ld_cb++; } } \
        }                                                                              \
    }
#define TMAT_STORE_A(i, R)                                                             \
    if (i < NPA) {                                                                     \
        float *d = &As[bb][c4 * LDA + i * PPP + t / TPP];                              \
        if (a.relu_in) { d[0] = fmaxf(R.x, 0.f); d[LDA] = fmaxf(R.y, 0.f); d[2 * LDA] = fmaxf(R.z, 0.f); d[3 * LDA] = fmaxf(R.w, 0.f); } \
        else { d[0] = R.x; d[LDA] = R.y; d[2 * LDA] = R.z; d[3 * LDA] = R.w; }         \
    }
#define TMAT_STORE_B(i, R) \
    if (i < NPB) *reinterpret_cast<float4 *>(&Bs[bb][(i * RPP + brow) * BN + bcol]) = R;
#define TMAT_STORE_CHUNK(buf_, S)                                                      \
    {                                                                                  \
        const int bb = (buf_);                                                         \
        TMAT_STORE_A(0, S##a0) TMAT_STORE_A(1, S##a1) TMAT_STORE_A(2, S##a2) TMAT_STORE_A(3, S##a3) \
        TMAT_STORE_A(4, S##a4) TMAT_STORE_A(5, S##a5) TMAT_STORE_A(6, S##a6) TMAT_STORE_A(7, S##a7) \
        TMAT_STORE_B(0, S##b0) TMAT_STORE_B(1, S##b1) TMAT_STORE_B(2, S##b2) TMAT_STORE_B(3, S##b3) \
    }
#define TMAT_MFMA_CHUNK(buf_)                                                          \
    {                                                                                  \
        const float *Ab = &As[buf_][aoff];                                             \
        const float *Bb = &Bs[buf_][boff];                                             \
        _Pragma("unroll") for (int kk = 0; kk < KC / 2; kk++) {                        \
            float av[TM], bv[TN];                                                      \
            _Pragma("unroll") for (int i = 0; i < TM; i++) av[i] = Ab[kk * 2 * LDA + i * 32];   \
            _Pragma("unroll") for (int jn = 0; jn < TN; jn++) bv[jn] = Bb[kk * 2 * BN + jn * 32]; \
            _Pragma("unroll") for (int i = 0; i < TM; i++)                             \
                _Pragma("unroll") for (int jn = 0; jn < TN; jn++)                      \
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[jn], acc[i][jn], 0, 0, 0); \
        }                                                                              \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int jn = 0; jn < TN; jn++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][jn][r] = 0.f;

    const int aoff = (lane >> 5) * LDA + wm * (BM / WM) + (lane & 31);
    const int boff = (lane >> 5) * BN + wn * (BN / WN) + (lane & 31);

    TMAT_LOAD_CHUNK(p)
    TMAT_STORE_CHUNK(0, p)
    TMAT_LOAD_CHUNK(q)
    __syncthreads();

    // chunk c lives in LDS buffer c & 1; register set p carries even chunks, q odd ones (nchunks is even: host check)
    for (int c = 0; c < nchunks; c += 2) {
        TMAT_LOAD_CHUNK(p)
        __builtin_amdgcn_sched_barrier(0);      // keep the staging math of prefetched chunks out of the MFMA loop
        TMAT_MFMA_CHUNK(0)
        __builtin_amdgcn_sched_barrier(0);
        TMAT_STORE_CHUNK(1, q)                  // chunk c + 1
        __syncthreads();
        TMAT_LOAD_CHUNK(q)
        __builtin_amdgcn_sched_barrier(0);
        TMAT_MFMA_CHUNK(1)
        __builtin_amdgcn_sched_barrier(0);
        TMAT_STORE_CHUNK(0, p)                  // chunk c + 2
        __syncthreads();
    }
#undef TMAT_BASE
#undef TMAT_LOAD_CHUNK
#undef TMAT_STORE_A
#undef TMAT_STORE_B
#undef TMAT_STORE_CHUNK
#undef TMAT_MFMA_CHUNK

    // epilogue: accumulators -> LDS tile [128][BN] (C/D layout: col = lane & 31 is the output channel,
    // row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) the pixel) -> BN fold / bias, residual, ReLU on float4 rows ->
    // 16-byte coalesced stores (one wave writes 1 KiB contiguous).
    {
        constexpr int NPASS = BM / EPR;
        constexpr int V4 = BN / 4;
        constexpr int RPI = 256 / V4;            // rows per store iteration
        float *Cs = smem;
        const int cv = (t % V4) * 4, r0 = t / V4;
        const int co = n0 + cv;
        float4 sc = make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 sh = *reinterpret_cast<const float4 *>(a.shift + co);
        if (a.scale) sc = *reinterpret_cast<const float4 *>(a.scale + co);
        const int rH = Ho >> a.rs, rW = Wo >> a.rs;
        const int wrow0 = wm * (BM / WM);
#pragma unroll
        for (int pass = 0; pass < NPASS; pass++) {
            if (wrow0 / EPR == pass) {
#pragma unroll
                for (int i = 0; i < TM; i++)
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int row = wrow0 % EPR + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#pragma unroll
                        for (int jn = 0; jn < TN; jn++) Cs[row * BN + wn * (BN / WN) + jn * 32 + (lane & 31)] = acc[i][jn][r];
                    }
            }
            __syncthreads();
#pragma unroll 4
            for (int it = 0; it < EPR / RPI; it++) {
                const int row = it * RPI + r0;
                const int m = m0 + pass * EPR + row;
                if (m >= M) continue;
                float4 v = *reinterpret_cast<const float4 *>(Cs + row * BN + cv);
                if (a.scale) { v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w); }
                else { v.x = v.x + sh.x; v.y = v.y + sh.y; v.z = v.z + sh.z; v.w = v.w + sh.w; }
                if (a.resid) {
                    size_t ridx = (size_t)m;
                    if (a.rs) {
                        const int n = m / (Ho * Wo);
                        const int rr = m - n * (Ho * Wo);
                        const int y = rr / Wo, x = rr - y * Wo;
                        ridx = ((size_t)n * rH + (y >> a.rs)) * rW + (x >> a.rs);
                    }
                    const float4 rv = *reinterpret_cast<const float4 *>(a.resid + ridx * a.Cout + co);
                    v.x = v.x + rv.x; v.y = v.y + rv.y; v.z = v.z + rv.z; v.w = v.w + rv.w;
                }
                if (a.relu_out) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                size_t oidx = (size_t)m;
                if (KS == 2) {      // scatter to the parity class's pixels of the (2 Ho, 2 Wo) output
                    const int n = m / (Ho * Wo);
                    const int rr = m - n * (Ho * Wo);
                    const int y = rr / Wo, x = rr - y * Wo;
                    oidx = ((size_t)n * 2 * Ho + 2 * y + spy) * (2 * Wo) + 2 * x + spx;
                }
                *reinterpret_cast<float4 *>(a.out + oidx * a.Cout + co) = v;
            }
            if (pass + 1 < NPASS) __syncthreads();
        }
    }
}

template <int BM, int BN, int WM, int WN, int KC>
static void launch_conv_cfg(const ConvArgs &a, int M, int Ho, int Wo, hipStream_t s)
{
    int nMt = (M + BM - 1) / BM, nNt = a.Cout / BN;
    int grid = ((nMt + 7) / 8) * 8 * nNt;
    if (a.ksize == 3)
        hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, WM, WN, KC, 3>), dim3(grid), dim3(256), 0, s, a, M, Ho, Wo, nMt, nNt);
    else if (a.ksize == 2)
        hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, WM, WN, KC, 2>), dim3(grid, 4), dim3(256), 0, s, a, M, Ho, Wo, nMt, nNt);
    else
        hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, WM, WN, KC, 1>), dim3(grid), dim3(256), 0, s, a, M, Ho, Wo, nMt, nNt);
}

bool launch_conv(const ConvArgs &a, hipStream_t s)
{
    const int H = a.h << a.up, W = a.w << a.up;
    const int Ho = H / a.stride, Wo = W / a.stride;
    const long long Mll = (long long)a.N * Ho * Wo;
    if (!((a.ksize == 3 && a.stride == 1) || (a.ksize == 2 && a.stride == 1 && a.up == 0 && !a.resid) ||
          (a.ksize == 1 && (a.stride == 1 || a.stride == 2))) || a.Cin % 32 ||
        a.Cout % 64 || ((a.ksize * a.ksize * (a.Cin / (a.Cout % 128 == 0 ? 32 : 16))) & 1) || Mll <= 0 || Mll > 0x7fffffffLL / 2 || (a.resid && a.rs && ((Ho | Wo) & 1))) {
        set_error("launch_conv: unsupported shape");
        return false;
    }
    const int M = (int)Mll;
    static const int stagger = [] { const char *e = getenv("TMAT_STAGGER"); return e ? atoi(e) : 0; }();
    if (!a.zeros || a.Cin > 2048) { set_error("launch_conv: missing zero row or Cin > 2048"); return false; }
    ConvArgs b = a;
    b.stagger = stagger;
    if (a.Cout % 128 == 0)
        launch_conv_cfg<128, 128, 2, 2, 32>(b, M, Ho, Wo, s);
    else
        launch_conv_cfg<256, 64, 4, 1, 16>(b, M, Ho, Wo, s);
    return true;
}

// ---------------------------------------------------------------------------------------------
// depthwise 3x3 (SeparableConv2D's depthwise half, models.py:131,135): chain over the 9 taps in
// (ky, kx) order, zero padding, optional ReLU on load.  One thread = one pixel x 4 channels.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dwconv_kernel(const float *__restrict__ in, int H, int W, int C, int c4shift,
                                                     int relu_in, const float *__restrict__ Wd, float *__restrict__ out)
{
    // grid: (ceil(H * (W/4) * C4 / 256), N).  One thread = 4 consecutive pixels of a row x 4 channels: the 3 x 6
    // input window is loaded once (18 float4 instead of 36).  32-bit index math; C4 = C/4 is a power of two.
    const int n = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int C4 = 1 << c4shift;
    const int cq = e & (C4 - 1);
    const int g = e >> c4shift;                 // pixel group index
    const int WG = W >> 2;
    if (g >= H * WG) return;
    const int y = g / WG, x0 = (g - y * WG) * 4;
    const float *base = in + (size_t)n * H * W * C + cq * 4;
    const float lo = relu_in ? 0.f : -INFINITY;
    float4 win[3][6];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 6; c++) {
            const int yy = y + r - 1, xx = x0 + c - 1;
            const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
            float4 v = *reinterpret_cast<const float4 *>(base + (ok ? (yy * W + xx) * C : 0));
            v.x = ok ? fmaxf(v.x, lo) : 0.f; v.y = ok ? fmaxf(v.y, lo) : 0.f;
            v.z = ok ? fmaxf(v.z, lo) : 0.f; v.w = ok ? fmaxf(v.w, lo) : 0.f;
            win[r][c] = v;
        }
    float4 wt[9];
#pragma unroll
    for (int tp = 0; tp < 9; tp++) wt[tp] = *reinterpret_cast<const float4 *>(Wd + tp * C + cq * 4);
    float *obase = out + ((size_t)n * H * W + (size_t)y * W + x0) * C + cq * 4;
#pragma unroll
    for (int px = 0; px < 4; px++) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int tp = 0; tp < 9; tp++) {
            const float4 v = win[tp / 3][px + tp % 3];
            acc.x = fmaf(v.x, wt[tp].x, acc.x); acc.y = fmaf(v.y, wt[tp].y, acc.y);
            acc.z = fmaf(v.z, wt[tp].z, acc.z); acc.w = fmaf(v.w, wt[tp].w, acc.w);
        }
        *reinterpret_cast<float4 *>(obase + px * C) = acc;
    }
}

static int ilog2(int v) { int s = 0; while ((1 << s) < v) s++; return s; }

void launch_dwconv(const float *in, int N, int H, int W, int C, int relu_in, const float *Wd, float *out, hipStream_t s)
{
    const int C4 = C / 4;
    const int total = H * (W / 4) * C4;         // W % 4 == 0 for every level of the model (checked in tmat_create)
    hipLaunchKernelGGL(dwconv_kernel, dim3((total + 255) / 256, N), dim3(256), 0, s, in, H, W, C, ilog2(C4), relu_in, Wd, out);
}

// ---------------------------------------------------------------------------------------------
// stem: Conv2D(C, 3, strides=2, "same") + BN + ReLU on the single-channel patch (models.py:119-121).
// TF SAME with even H: taps read rows 2y .. 2y+2 (zero beyond the image).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stem_kernel(const float *__restrict__ x, int H, int W, const float *__restrict__ Ws,
                                                   int Cout, int c4shift, const float *__restrict__ scale,
                                                   const float *__restrict__ shift, float *__restrict__ out)
{
    const int Ho = H >> 1, Wo = W >> 1;
    const int n = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int cq = e & ((1 << c4shift) - 1);
    const int p = e >> c4shift;
    if (p >= Ho * Wo) return;
    const int yo = p / Wo, xo = p - yo * Wo;
    const float *xin = x + (size_t)n * H * W;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int tp = 0; tp < 9; tp++) {
        const int iy = 2 * yo + tp / 3, ix = 2 * xo + tp % 3;
        const float v = (iy < H && ix < W) ? xin[iy * W + ix] : 0.f;
        const float4 w = *reinterpret_cast<const float4 *>(Ws + tp * Cout + cq * 4);
        acc.x = fmaf(v, w.x, acc.x); acc.y = fmaf(v, w.y, acc.y);
        acc.z = fmaf(v, w.z, acc.z); acc.w = fmaf(v, w.w, acc.w);
    }
    const float4 sc = *reinterpret_cast<const float4 *>(scale + cq * 4);
    const float4 sh = *reinterpret_cast<const float4 *>(shift + cq * 4);
    float4 o;
    o.x = fmaxf(fmaf(acc.x, sc.x, sh.x), 0.f); o.y = fmaxf(fmaf(acc.y, sc.y, sh.y), 0.f);
    o.z = fmaxf(fmaf(acc.z, sc.z, sh.z), 0.f); o.w = fmaxf(fmaf(acc.w, sc.w, sh.w), 0.f);
    *reinterpret_cast<float4 *>(out + ((size_t)n * Ho * Wo + p) * Cout + cq * 4) = o;
}

void launch_stem(const float *x, int N, int H, int W, const float *Ws, int Cout, const float *scale,
                 const float *shift, float *out, hipStream_t s)
{
    const int total = (H / 2) * (W / 2) * (Cout / 4);
    hipLaunchKernelGGL(stem_kernel, dim3((total + 255) / 256, N), dim3(256), 0, s, x, H, W, Ws, Cout, ilog2(Cout / 4), scale, shift, out);
}

// ---------------------------------------------------------------------------------------------
// MaxPooling2D(3, strides=2, "same") + residual add (models.py:138-144)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_add_kernel(const float *__restrict__ p2, int H, int W, int C, int c4shift,
                                                          const float *__restrict__ r, float *__restrict__ out)
{
    const int Ho = H >> 1, Wo = W >> 1;
    const int n = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int cq = e & ((1 << c4shift) - 1);
    const int p = e >> c4shift;
    if (p >= Ho * Wo) return;
    const int yo = p / Wo, xo = p - yo * Wo;
    const float *base = p2 + (size_t)n * H * W * C + cq * 4;
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int tp = 0; tp < 9; tp++) {
        const int iy = 2 * yo + tp / 3, ix = 2 * xo + tp % 3;
        if (iy < H && ix < W) {
            const float4 v = *reinterpret_cast<const float4 *>(base + (iy * W + ix) * C);
            m.x = v.x > m.x ? v.x : m.x; m.y = v.y > m.y ? v.y : m.y;
            m.z = v.z > m.z ? v.z : m.z; m.w = v.w > m.w ? v.w : m.w;
        }
    }
    const size_t o = ((size_t)n * Ho * Wo + p) * C + cq * 4;
    const float4 rv = *reinterpret_cast<const float4 *>(r + o);
    m.x = m.x + rv.x; m.y = m.y + rv.y; m.z = m.z + rv.z; m.w = m.w + rv.w;
    *reinterpret_cast<float4 *>(out + o) = m;
}

void launch_maxpool_add(const float *p2, int N, int H, int W, int C, const float *r, float *out, hipStream_t s)
{
    const int total = (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(maxpool_add_kernel, dim3((total + 255) / 256, N), dim3(256), 0, s, p2, H, W, C, ilog2(C / 4), r, out);
}

// ---------------------------------------------------------------------------------------------
// final Conv2D(1, 3, "same") + sigmoid on the nearest-upsampled last block (models.py:158,166).
// Deterministic expf (same operation sequence as oracle/unet_exact.c: exp_det).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float exp_det(float x)
{
    x = fminf(fmaxf(x, -88.0f), 88.0f);
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = fmaf(p, r2, r) + 1.0f;
    return ldexpf(y, (int)n);
}

// One block = 32 x 8 output pixels of one patch.  The (6 x 18) low-resolution pixels they touch are staged in LDS
// (pixel stride padded to C + 4 floats so the 16-byte channel reads of neighbouring pixels do not collide), weights in LDS.
__global__ __launch_bounds__(256) void final_kernel(const float *__restrict__ S, int h, int w, int C,
                                                    const float *__restrict__ Wf, float bias, float *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int CP = C + 4;
    float *wsh = smem;                       // [9][C]
    float *tile = smem + 9 * C;              // [6][18][CP]
    const int H = 2 * h, W = 2 * w;
    const int n = blockIdx.z;
    const int x0 = blockIdx.x * 32, y0 = blockIdx.y * 8;
    const int ly0 = (y0 - 1) >> 1, lx0 = (x0 - 1) >> 1;     // floor division also for -1
    for (int i = threadIdx.x; i < 9 * C; i += 256) wsh[i] = Wf[i];
    const int C4 = C >> 2;
    for (int i = threadIdx.x; i < 6 * 18 * C4; i += 256) {
        const int cq = i % C4, p = i / C4;
        const int ly = ly0 + p / 18, lx = lx0 + p % 18;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ly >= 0 && ly < h && lx >= 0 && lx < w) v = *reinterpret_cast<const float4 *>(S + (((size_t)n * h + ly) * w + lx) * C + cq * 4);
        *reinterpret_cast<float4 *>(tile + p * CP + cq * 4) = v;
    }
    __syncthreads();
    const int x = x0 + (threadIdx.x & 31), y = y0 + (threadIdx.x >> 5);
    if (x >= W || y >= H) return;
    float acc = 0.f;
    for (int tp = 0; tp < 9; tp++) {
        const int iy = y + tp / 3 - 1, ix = x + tp % 3 - 1;
        const float *wr = wsh + tp * C;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
            const float *ip = tile + (((iy >> 1) - ly0) * 18 + ((ix >> 1) - lx0)) * CP;
            for (int c = 0; c < C; c += 4) {
                const float4 v = *reinterpret_cast<const float4 *>(ip + c);
                const float4 ww = *reinterpret_cast<const float4 *>(wr + c);
                acc = fmaf(v.x, ww.x, acc); acc = fmaf(v.y, ww.y, acc);
                acc = fmaf(v.z, ww.z, acc); acc = fmaf(v.w, ww.w, acc);
            }
        } else {
            for (int c = 0; c < C; c++) acc = fmaf(0.f, wr[c], acc);
        }
    }
    const float z = acc + bias;
    out[((size_t)n * H + y) * W + x] = 1.0f / (1.0f + exp_det(-z));
}

void launch_final(const float *S, int N, int h, int w, int C, const float *Wf, float bias, float *out, hipStream_t s)
{
    dim3 grid((2 * w + 31) / 32, (2 * h + 7) / 8, N);
    const size_t lds = (size_t)(9 * C + 6 * 18 * (C + 4)) * sizeof(float);
    hipLaunchKernelGGL(final_kernel, grid, dim3(256), lds, s, S, h, w, C, Wf, bias, out);
}

}  // namespace tmat
