// UNet-Xception inference kernels for gfx950 (MI355X).  NHWC float32.
//
// Reference graph: fl_tissue_model_tools/models.py:110-166 (build_UNetXception), executed by
// keras Model.predict at smooth_tiled_predictions.py:179.
//
// Arithmetic contract (shared with oracle/unet_exact.c, compared bit-exactly):
//   every contraction is a chain acc = fmaf(a[k], w[k], acc) with k = (ky, kx) tap-major and
//   input channel ascending; v_mfma_f32_32x32x2_f32 performs exactly that chain (one rounding
//   per product, k ascending), so the dense 3x3 / 1x1 contractions run on the matrix cores at
//   full f32 precision.  Epilogues: v = fmaf(acc, scale, shift) (folded BN) or acc + bias,
//   optional residual add, optional ReLU.  Compiled with -ffp-contract=off.
#include "tmat_internal.h"
#include "../../include/tmat.h"

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
template <int BM, int BN, int WM, int WN, int KC>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a, int M, int Ho, int Wo, int nMt, int nNt)
{
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

    __shared__ float As[2][KC * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][KC * BN];

    // XCD-aware tile mapping: blocks b and b+8 share an XCD (round-robin dispatch); give the
    // nNt column tiles of one pixel tile to the same XCD so its L2 serves the re-read A pixels.
    const int b = blockIdx.x;
    const int xcd = b & 7, j = b >> 3;
    const int nt = j % nNt;
    const int mt = (j / nNt) * 8 + xcd;
    if (mt >= nMt) return;
    const int m0 = mt * BM, n0 = nt * BN;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;

    const int H = a.h << a.up, W = a.w << a.up;
    const int taps = a.ksize * a.ksize;
    const int cchunks = a.Cin / KC;
    const int nchunks = taps * cchunks;

    // per-thread pixel bookkeeping for A staging
    const int c4 = (t % TPP) * 4;
    int py[NPA], px[NPA], pbase[NPA];
    bool pok[NPA];
#pragma unroll
    for (int i = 0; i < NPA; i++) {
        int m = m0 + i * PPP + t / TPP;
        pok[i] = m < M;
        int mm = pok[i] ? m : 0;
        int n = mm / (Ho * Wo);
        int r = mm - n * (Ho * Wo);
        int y = r / Wo;
        py[i] = y * a.stride;
        px[i] = (r - y * Wo) * a.stride;
        pbase[i] = n * a.h;
    }
    const int brow = t / BV, bcol = (t % BV) * 4;

    // Loads are unconditional (out-of-image taps read a valid dummy address and are zeroed when the
    // registers are written to LDS), so the next chunk's global loads stay in flight across the MFMA loop.
    float4 ra0, ra1, ra2, ra3, ra4, ra5, ra6, ra7, rb0, rb1, rb2, rb3;
    bool ok0 = false, ok1 = false, ok2 = false, ok3 = false, ok4 = false, ok5 = false, ok6 = false, ok7 = false;
    static_assert(NPA == 4 || NPA == 8 || NPA == 2, "A passes");
    static_assert(NPB == 4 || NPB == 2 || NPB == 1, "B passes");
    const float relu_lo = a.relu_in ? 0.f : -INFINITY;      // relu on load folded into one max

#define TMAT_LOAD_A(i, R, OK)                                                                               \
    if (i < NPA) {                                                                                          \
        const int yy = py[i] + dy, xx = px[i] + dx;                                                         \
        OK = pok[i] && yy >= 0 && yy < H && xx >= 0 && xx < W;                                              \
        const size_t off = OK ? ((size_t)(pbase[i] + (yy >> a.up)) * a.w + (xx >> a.up)) * a.Cin : (size_t)0; \
        R = *reinterpret_cast<const float4 *>(a.in + off + c0 + c4);                                        \
    }
#define TMAT_LOAD_B(i, R) \
    if (i < NPB) R = *reinterpret_cast<const float4 *>(wp + (size_t)(i * RPP + brow) * a.Cout);
#define TMAT_LOAD_CHUNK(cc)                                                            \
    {                                                                                  \
        const int tap = (cc) / cchunks;                                                \
        const int c0 = ((cc) - tap * cchunks) * KC;                                    \
        const int dy = a.ksize == 3 ? tap / 3 - 1 : 0;                                 \
        const int dx = a.ksize == 3 ? tap % 3 - 1 : 0;                                 \
        TMAT_LOAD_A(0, ra0, ok0) TMAT_LOAD_A(1, ra1, ok1) TMAT_LOAD_A(2, ra2, ok2) TMAT_LOAD_A(3, ra3, ok3) \
        TMAT_LOAD_A(4, ra4, ok4) TMAT_LOAD_A(5, ra5, ok5) TMAT_LOAD_A(6, ra6, ok6) TMAT_LOAD_A(7, ra7, ok7) \
        const float *wp = a.W + ((size_t)tap * a.Cin + c0) * a.Cout + n0 + bcol;       \
        TMAT_LOAD_B(0, rb0) TMAT_LOAD_B(1, rb1) TMAT_LOAD_B(2, rb2) TMAT_LOAD_B(3, rb3) \
    }
#define TMAT_STORE_A(i, R, OK)                                                         \
    if (i < NPA) {                                                                     \
        float *d = &As[bb][c4 * LDA + i * PPP + t / TPP];                              \
        d[0] = OK ? fmaxf(R.x, relu_lo) : 0.f; d[LDA] = OK ? fmaxf(R.y, relu_lo) : 0.f; \
        d[2 * LDA] = OK ? fmaxf(R.z, relu_lo) : 0.f; d[3 * LDA] = OK ? fmaxf(R.w, relu_lo) : 0.f; \
    }
#define TMAT_STORE_B(i, R) \
    if (i < NPB) *reinterpret_cast<float4 *>(&Bs[bb][(i * RPP + brow) * BN + bcol]) = R;
#define TMAT_STORE_CHUNK(buf_)                                                         \
    {                                                                                  \
        const int bb = (buf_);                                                         \
        TMAT_STORE_A(0, ra0, ok0) TMAT_STORE_A(1, ra1, ok1) TMAT_STORE_A(2, ra2, ok2) TMAT_STORE_A(3, ra3, ok3) \
        TMAT_STORE_A(4, ra4, ok4) TMAT_STORE_A(5, ra5, ok5) TMAT_STORE_A(6, ra6, ok6) TMAT_STORE_A(7, ra7, ok7) \
        TMAT_STORE_B(0, rb0) TMAT_STORE_B(1, rb1) TMAT_STORE_B(2, rb2) TMAT_STORE_B(3, rb3) \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int jn = 0; jn < TN; jn++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][jn][r] = 0.f;

    TMAT_LOAD_CHUNK(0)
    TMAT_STORE_CHUNK(0)
    __syncthreads();

    const int aoff = (lane >> 5) * LDA + wm * (BM / WM) + (lane & 31);
    const int boff = (lane >> 5) * BN + wn * (BN / WN) + (lane & 31);

    for (int c = 0; c < nchunks; c++) {
        const int buf = c & 1;
        // prefetch the next chunk (the last iteration re-loads the last chunk; it is never consumed)
        const int cn = c + 1 < nchunks ? c + 1 : c;
        TMAT_LOAD_CHUNK(cn)
        __builtin_amdgcn_sched_barrier(0);      // keep the LDS-store math of the prefetched chunk below the MFMA loop
        const float *Ab = &As[buf][aoff];
        const float *Bb = &Bs[buf][boff];
#pragma unroll
        for (int kk = 0; kk < KC / 2; kk++) {
            float av[TM], bv[TN];
#pragma unroll
            for (int i = 0; i < TM; i++) av[i] = Ab[kk * 2 * LDA + i * 32];
#pragma unroll
            for (int jn = 0; jn < TN; jn++) bv[jn] = Bb[kk * 2 * BN + jn * 32];
#pragma unroll
            for (int i = 0; i < TM; i++)
#pragma unroll
                for (int jn = 0; jn < TN; jn++)
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[jn], acc[i][jn], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        TMAT_STORE_CHUNK(buf ^ 1)
        __syncthreads();
    }
#undef TMAT_LOAD_A
#undef TMAT_LOAD_B
#undef TMAT_LOAD_CHUNK
#undef TMAT_STORE_A
#undef TMAT_STORE_B
#undef TMAT_STORE_CHUNK

    // epilogue: C/D layout col = lane & 31 (cout), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) (pixel)
    const int rH = Ho >> a.rs, rW = Wo >> a.rs;
#pragma unroll
    for (int i = 0; i < TM; i++) {
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            const int m = m0 + wm * (BM / WM) + i * 32 + row;
            if (m >= M) continue;
            size_t ridx = (size_t)m;
            if (a.resid && a.rs) {
                int n = m / (Ho * Wo);
                int rr = m - n * (Ho * Wo);
                int y = rr / Wo, x = rr - y * Wo;
                ridx = ((size_t)n * rH + (y >> a.rs)) * rW + (x >> a.rs);
            }
#pragma unroll
            for (int jn = 0; jn < TN; jn++) {
                const int co = n0 + wn * (BN / WN) + jn * 32 + (lane & 31);
                float v = acc[i][jn][r];
                v = a.scale ? fmaf(v, a.scale[co], a.shift[co]) : v + a.shift[co];
                if (a.resid) v = v + a.resid[ridx * a.Cout + co];
                if (a.relu_out) v = fmaxf(v, 0.f);
                a.out[(size_t)m * a.Cout + co] = v;
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int KC>
static void launch_conv_cfg(const ConvArgs &a, int M, int Ho, int Wo, hipStream_t s)
{
    int nMt = (M + BM - 1) / BM, nNt = a.Cout / BN;
    int grid = ((nMt + 7) / 8) * 8 * nNt;
    hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, WM, WN, KC>), dim3(grid), dim3(256), 0, s, a, M, Ho, Wo, nMt, nNt);
}

bool launch_conv(const ConvArgs &a, hipStream_t s)
{
    const int H = a.h << a.up, W = a.w << a.up;
    const int Ho = H / a.stride, Wo = W / a.stride;
    const long long Mll = (long long)a.N * Ho * Wo;
    if (!((a.ksize == 3 && a.stride == 1) || (a.ksize == 1 && (a.stride == 1 || a.stride == 2))) || a.Cin % 32 ||
        a.Cout % 64 || Mll <= 0 || Mll > 0x7fffffffLL / 2 || (a.resid && a.rs && ((Ho | Wo) & 1))) {
        set_error("launch_conv: unsupported shape");
        return false;
    }
    const int M = (int)Mll;
    if (a.Cout % 128 == 0)
        launch_conv_cfg<128, 128, 2, 2, 32>(a, M, Ho, Wo, s);
    else
        launch_conv_cfg<256, 64, 4, 1, 16>(a, M, Ho, Wo, s);
    return true;
}

// ---------------------------------------------------------------------------------------------
// depthwise 3x3 (SeparableConv2D's depthwise half, models.py:131,135): chain over the 9 taps in
// (ky, kx) order, zero padding, optional ReLU on load.  One thread = one pixel x 4 channels.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dwconv_kernel(const float *__restrict__ in, int N, int H, int W, int C,
                                                     int relu_in, const float *__restrict__ Wd, float *__restrict__ out,
                                                     size_t total4)
{
    const int C4 = C >> 2;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total4; idx += (size_t)gridDim.x * blockDim.x) {
        int cq = (int)(idx % C4);
        size_t p = idx / C4;
        int x = (int)(p % W);
        size_t q = p / W;
        int y = (int)(q % H);
        int n = (int)(q / H);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int tp = 0; tp < 9; tp++) {
            int yy = y + tp / 3 - 1, xx = x + tp % 3 - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                v = *reinterpret_cast<const float4 *>(in + (((size_t)n * H + yy) * W + xx) * C + cq * 4);
                if (relu_in) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            }
            float4 w = *reinterpret_cast<const float4 *>(Wd + tp * C + cq * 4);
            acc.x = fmaf(v.x, w.x, acc.x); acc.y = fmaf(v.y, w.y, acc.y);
            acc.z = fmaf(v.z, w.z, acc.z); acc.w = fmaf(v.w, w.w, acc.w);
        }
        *reinterpret_cast<float4 *>(out + idx * 4) = acc;
    }
}

void launch_dwconv(const float *in, int N, int H, int W, int C, int relu_in, const float *Wd, float *out, hipStream_t s)
{
    size_t total4 = (size_t)N * H * W * (C / 4);
    int grid = (int)((total4 + 255) / 256 < 16384 ? (total4 + 255) / 256 : 16384);
    hipLaunchKernelGGL(dwconv_kernel, dim3(grid), dim3(256), 0, s, in, N, H, W, C, relu_in, Wd, out, total4);
}

// ---------------------------------------------------------------------------------------------
// stem: Conv2D(C, 3, strides=2, "same") + BN + ReLU on the single-channel patch (models.py:119-121).
// TF SAME with even H: taps read rows 2y .. 2y+2 (zero beyond the image).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stem_kernel(const float *__restrict__ x, int N, int H, int W,
                                                   const float *__restrict__ Ws, int Cout,
                                                   const float *__restrict__ scale, const float *__restrict__ shift,
                                                   float *__restrict__ out, size_t total4)
{
    const int Ho = H >> 1, Wo = W >> 1, C4 = Cout >> 2;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total4; idx += (size_t)gridDim.x * blockDim.x) {
        int cq = (int)(idx % C4);
        size_t p = idx / C4;
        int xo = (int)(p % Wo);
        size_t q = p / Wo;
        int yo = (int)(q % Ho);
        int n = (int)(q / Ho);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int tp = 0; tp < 9; tp++) {
            int iy = 2 * yo + tp / 3, ix = 2 * xo + tp % 3;
            float v = (iy < H && ix < W) ? x[((size_t)n * H + iy) * W + ix] : 0.f;
            float4 w = *reinterpret_cast<const float4 *>(Ws + tp * Cout + cq * 4);
            acc.x = fmaf(v, w.x, acc.x); acc.y = fmaf(v, w.y, acc.y);
            acc.z = fmaf(v, w.z, acc.z); acc.w = fmaf(v, w.w, acc.w);
        }
        float4 sc = *reinterpret_cast<const float4 *>(scale + cq * 4);
        float4 sh = *reinterpret_cast<const float4 *>(shift + cq * 4);
        float4 o;
        o.x = fmaxf(fmaf(acc.x, sc.x, sh.x), 0.f); o.y = fmaxf(fmaf(acc.y, sc.y, sh.y), 0.f);
        o.z = fmaxf(fmaf(acc.z, sc.z, sh.z), 0.f); o.w = fmaxf(fmaf(acc.w, sc.w, sh.w), 0.f);
        *reinterpret_cast<float4 *>(out + idx * 4) = o;
    }
}

void launch_stem(const float *x, int N, int H, int W, const float *Ws, int Cout, const float *scale,
                 const float *shift, float *out, hipStream_t s)
{
    size_t total4 = (size_t)N * (H / 2) * (W / 2) * (Cout / 4);
    int grid = (int)((total4 + 255) / 256 < 16384 ? (total4 + 255) / 256 : 16384);
    hipLaunchKernelGGL(stem_kernel, dim3(grid), dim3(256), 0, s, x, N, H, W, Ws, Cout, scale, shift, out, total4);
}

// ---------------------------------------------------------------------------------------------
// MaxPooling2D(3, strides=2, "same") + residual add (models.py:138-144)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_add_kernel(const float *__restrict__ p2, int N, int H, int W, int C,
                                                          const float *__restrict__ r, float *__restrict__ out,
                                                          size_t total4)
{
    const int Ho = H >> 1, Wo = W >> 1, C4 = C >> 2;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total4; idx += (size_t)gridDim.x * blockDim.x) {
        int cq = (int)(idx % C4);
        size_t p = idx / C4;
        int xo = (int)(p % Wo);
        size_t q = p / Wo;
        int yo = (int)(q % Ho);
        int n = (int)(q / Ho);
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
        for (int tp = 0; tp < 9; tp++) {
            int iy = 2 * yo + tp / 3, ix = 2 * xo + tp % 3;
            if (iy < H && ix < W) {
                float4 v = *reinterpret_cast<const float4 *>(p2 + (((size_t)n * H + iy) * W + ix) * C + cq * 4);
                m.x = v.x > m.x ? v.x : m.x; m.y = v.y > m.y ? v.y : m.y;
                m.z = v.z > m.z ? v.z : m.z; m.w = v.w > m.w ? v.w : m.w;
            }
        }
        float4 rv = *reinterpret_cast<const float4 *>(r + idx * 4);
        m.x = m.x + rv.x; m.y = m.y + rv.y; m.z = m.z + rv.z; m.w = m.w + rv.w;
        *reinterpret_cast<float4 *>(out + idx * 4) = m;
    }
}

void launch_maxpool_add(const float *p2, int N, int H, int W, int C, const float *r, float *out, hipStream_t s)
{
    size_t total4 = (size_t)N * (H / 2) * (W / 2) * (C / 4);
    int grid = (int)((total4 + 255) / 256 < 16384 ? (total4 + 255) / 256 : 16384);
    hipLaunchKernelGGL(maxpool_add_kernel, dim3(grid), dim3(256), 0, s, p2, N, H, W, C, r, out, total4);
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

__global__ __launch_bounds__(256) void final_kernel(const float *__restrict__ S, int N, int h, int w, int C,
                                                    const float *__restrict__ Wf, float bias, float *__restrict__ out)
{
    extern __shared__ float wsh[];   // [9][C]
    for (int i = threadIdx.x; i < 9 * C; i += 256) wsh[i] = Wf[i];
    __syncthreads();
    const int H = 2 * h, W = 2 * w;
    const int x = blockIdx.x * 32 + (threadIdx.x & 31);
    const int y = blockIdx.y * 8 + (threadIdx.x >> 5);
    const int n = blockIdx.z;
    if (x >= W || y >= H) return;
    float acc = 0.f;
    for (int tp = 0; tp < 9; tp++) {
        int iy = y + tp / 3 - 1, ix = x + tp % 3 - 1;
        const float *wr = wsh + tp * C;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
            const float *ip = S + (((size_t)n * h + (iy >> 1)) * w + (ix >> 1)) * C;
            for (int c = 0; c < C; c += 4) {
                float4 v = *reinterpret_cast<const float4 *>(ip + c);
                acc = fmaf(v.x, wr[c], acc); acc = fmaf(v.y, wr[c + 1], acc);
                acc = fmaf(v.z, wr[c + 2], acc); acc = fmaf(v.w, wr[c + 3], acc);
            }
        } else {
            for (int c = 0; c < C; c++) acc = fmaf(0.f, wr[c], acc);
        }
    }
    float z = acc + bias;
    out[((size_t)n * H + y) * W + x] = 1.0f / (1.0f + exp_det(-z));
}

void launch_final(const float *S, int N, int h, int w, int C, const float *Wf, float bias, float *out, hipStream_t s)
{
    dim3 grid((2 * w + 31) / 32, (2 * h + 7) / 8, N);
    hipLaunchKernelGGL(final_kernel, grid, dim3(256), 9 * C * sizeof(float), s, S, N, h, w, C, Wf, bias, out);
}

}  // namespace tmat
