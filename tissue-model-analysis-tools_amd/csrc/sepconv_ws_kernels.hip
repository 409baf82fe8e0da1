// Fused SeparableConv2D, wave-specialised form (round 3) for gfx950 (MI355X).  NHWC float32.
//
// Reference graph: fl_tissue_model_tools/models.py:126-144 (SeparableConv2D + BatchNormalization pairs of every down
// block, MaxPooling2D(3, 2, "same") + residual add behind the second one), executed by keras Model.predict at
// smooth_tiled_predictions.py:179.
//
// Round 2's fused kernel (removed in round 4; git history: csrc/sepconv_kernels.hip) computed the depthwise values inside the MFMA
// waves (each lane built its own A fragment between its MFMAs); its counters showed the matrix pipe busy 0.53-0.65 of the time: the
// in-order waves cannot overlap their own vector work, LDS reads and LDS-DMA issue with their own MFMAs.  Here the two kinds of work
// live in DIFFERENT waves of one 12-wave workgroup (phases: see below -- the f32 MFMA and the vector ALU of a SIMD do not issue in
// the same cycles, so a step has an MFMA phase and a vector phase):
//   waves 0-7  consumers: a 16 x 16 pixel tile (M = 256) x 128 output channels on v_mfma_f32_32x32x2_f32; wave w owns
//              tile rows 2w, 2w+1 as four 32x32 accumulators.  Per 32-channel step: 20 ds_read_b128, 64 MFMAs, one
//              barrier.  No vector-memory instruction except the tile's stores.
//   waves 8-11 producers: wave p owns the 8 x 8 pixel quadrant (p >> 1, p & 1) of the tile.  Per step it (1) waits for
//              ITS OWN 10 x 10 pixel halo of the next step's 32 channels (LDS-DMA into a private LDS region: no other
//              wave reads it, so no barrier orders it, only the wave's own vmcnt), (2) slides a 3 x 3 register window
//              down its column (lane = channel quad x column: 3 ds_read_b128 and 36 FMAs per output quad), (3) writes
//              the depthwise values as the NEXT step's A operand, [MFMA row][32 channels] with the convolution kernel's
//              bank swizzle, (4) issues the LDS-DMA of the next step's pointwise weights and of its own next halo.
// Two A / B stages; the halo region is single (a producer refills it after its last read of the step).
//
// Arithmetic contract: identical to the unfused pair dwconv_kernel -> conv_mfma_kernel<.., 1, ..> and to oracle/unet_exact.c:orc_dwconv -> orc_conv (depthwise
// chain over the 9 taps in (ky, kx) order from +0.0, zero padding, optional ReLU on load; pointwise chain over the
// input channels in groups of 8 in the order 0,4,1,5,2,6,3,7; epilogue fmaf(acc, scale, shift), optional ReLU; the
// pooled form: max, then one add).  tests/test_gpu_unet.py compares bits.
#include "dev_guard.h"
#include "tmat_internal.h"
#include "../../include/tmat.h"

#include <cstdio>
#include <cstdlib>

namespace tmat {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// LDS map (floats): [A0 | X | A1 | B0 | B1 | H0 H1 H2 H3].  The pooling epilogue's exchange buffer (9216 floats) lies over
// the A stage the tile's last step has just consumed plus the gap X: [A0 | X] or [X | A1].
constexpr int WS_A = 256 * 32, WS_X = 1024, WS_B = 128 * 32, WS_NHP = 13, WS_HP = WS_NHP * 256;
constexpr int WS_A0 = 0, WS_A1 = WS_A + WS_X, WS_B0 = WS_A1 + WS_A, WS_B1 = WS_B0 + WS_B, WS_H = WS_B1 + WS_B;
constexpr int WS_TOTAL = WS_H + 4 * WS_HP;          // 38 912 floats = 152 KiB
static_assert(WS_TOTAL * 4 <= 160 * 1024, "LDS budget");
// STEM form: behind the halos, per producer the 21 x 22 window of the single-channel patch its 10 x 10 stem-output halo is computed from
constexpr int WS_WPITCH = 22, WS_WP = 21 * WS_WPITCH, WS_NWP = (WS_WP + 63) / 64, WS_WIN = WS_TOTAL, WS_TOTAL_STEM = WS_WIN + 4 * WS_NWP * 64;
static_assert(WS_TOTAL_STEM * 4 <= 160 * 1024, "LDS budget (stem form)");
static_assert(8 * 9 * 128 <= WS_A + WS_X, "exchange buffer fits an A stage + gap");

struct WsArgs {
    const float *in;      // (N, H, W, Cin)
    int N, H, W, Cin, Cout;
    const float *dw9;     // depthwise taps [9][Cin] (the Keras layout)
    const float *pwk;     // pointwise weights [Cout][Cin] (k contiguous)
    const float *scale, *shift;
    int relu_out;
    float *out;           // (N, H, W, Cout); POOL: (N, H/2, W/2, Cout): complete pooled pixels with their residual; a tile's last pooled row / column
                          // are left as partial maxima for pool_fix_add_kernel
    const float *resid;   // POOL only
    float *strip_h, *strip_v, *corner;      // POOL only (pool_fix_add_kernel finishes the tile edges)
    const float *stem_w, *stem_scale, *stem_shift;      // STEM only: `in` is then the (N, 2H, 2W) single-channel patch; taps [9][Cin], folded BN
};

__device__ __forceinline__ int ws_pixmap_y(int r) { return ((r >= 4 && r < 12) || (r >= 16 && r < 20) || r >= 28) ? 1 : 0; }
__device__ __forceinline__ int ws_pixmap_x(int r) { return r < 4 ? r : r < 12 ? r - 4 : r < 16 ? r - 8 : r < 20 ? r - 8 : r < 28 ? r - 12 : r - 16; }

// workgroup barrier that leaves LDS-DMA in flight (no vmcnt wait: __syncthreads() would drain it)
#define WS_BAR() { asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
#define WS_PIN() __builtin_amdgcn_sched_barrier(0);
#ifdef WS_DIAG            // diagnostic build (tools/build_variant.sh): s_memtime stamps at the phase boundaries, summed per wave
__device__ long long ws_diag[256 * 12 * 8];
#define WS_STAMP(k) { const long long now_ = (long long)__builtin_readcyclecounter(); dsum[k] += now_ - dlast; dlast = now_; }
#define WS_DIAG_DECL long long dsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dlast = (long long)__builtin_readcyclecounter();
#define WS_DIAG_FLUSH() if (lane == 0 && blockIdx.x < 256) { for (int k = 0; k < 8; k++) ws_diag[((size_t)blockIdx.x * 12 + wave) * 8 + k] = dsum[k]; }
__device__ long long ws_tl[12 * 2 * 14];         // absolute stamps of workgroup 0's waves in steps 13 (mid tile) and 15 (tile end)
#define WS_TL(k) if (lane == 0 && blockIdx.x == 0 && (s == 13 || s == 15)) ws_tl[(wave * 2 + (s == 15)) * 14 + (k)] = (long long)__builtin_readcyclecounter();
#else
#define WS_TL(k)
#define WS_STAMP(k)
#define WS_DIAG_DECL
#define WS_DIAG_FLUSH()
#endif

// PREC = 0: f32 MFMAs (the bit-exact path).  PREC = 1: the opt-in "bf16x3" mode of tmat_set_precision (unet_kernels.hip) for the
// POINTWISE contraction only: the depthwise values stay f32 (producers unchanged, A operand f32 in LDS) and are split into bf16 hi / lo in
// the consumers' registers; a.pwk then points to the host-split weights, two bf16 planes [plane][Cout][Cin]; three
// v_mfma_f32_32x32x16_bf16 (lo hi, hi lo, hi hi) replace eight f32 MFMAs.  Same accumulator layout, same epilogues.
//
// STEM (round 4): the layer's input is the stem's output, relu(bn(conv3x3 / stride 2 (x))) of the single-channel patch x (models.py:119-121),
// and is NOT read from memory: every producer recomputes its 10 x 10 halo of the step's 32 channels from a 21 x 22 window of x (LDS-DMA,
// once per tile) in the vector phase, with stem_kernel's own chain (taps (ky, kx) from +0.0, fmaf(acc, scale, shift), max with 0), and
// writes it where the LDS-DMA of the plain form would have put it.  The stem tensor (10.5 GB per pass of 1600 patches) is then neither
// written nor read: 9 multiply-adds per value are cheaper than 8 bytes of HBM traffic.  Halo pixels outside the image are the depthwise
// convolution's zero padding (zeros, not stem values).
template <bool RELU_IN, bool POOL, int PREC = 0, bool STEM = false>
__global__ __launch_bounds__(768, 3) void sepconv_ws_kernel(WsArgs a, int nMt, int nNt, int G)
{
    __shared__ __attribute__((aligned(16))) float smem[STEM ? WS_TOTAL_STEM : WS_TOTAL];

    // persistent ranges: blocks b and b + 8 share an XCD; every XCD gets a contiguous super-range of the (pixel tile, channel tile) pairs and
    // every workgroup a contiguous piece of it, so the overlapping halos of neighbouring tiles and the re-read of a pixel tile by its
    // channel tiles are served by that XCD's L2
    const int b = blockIdx.x;
    const int bp = (b & 7) * (G >> 3) + (b >> 3);
    const long long P = (long long)nMt * nNt;
    const int j0 = (int)(P * bp / G), j1 = (int)(P * (bp + 1) / G);
    if (j0 >= j1) return;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int TW = a.W >> 4, TPP = (a.H >> 4) * TW;          // tiles per row / per patch
    const int Cin = a.Cin;
    const int nchunks = Cin >> 5;
    const int total = (j1 - j0) * nchunks;                   // steps of this workgroup
    constexpr unsigned OOB = 0x80000000u;

    if (STEM) {
        // the depthwise taps [9][Cin] (Cin <= 113: launch check) go to the gap X once per workgroup: the producers read their quad's taps from
        // there at the start of every depthwise phase instead of holding them in 36 registers across the stem phase (whose own 44 weight
        // registers are dead by then) -- with both sets live the kernel spilled
        for (int e = t; e < 9 * Cin; e += 768) smem[WS_A + e] = a.dw9[e];
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        WS_BAR()
    }
    if (wave >= 8) {
        // ================================ producer ================================
        const int p = wave - 8;
#ifndef WS_PRODUCER_PRIO
#define WS_PRODUCER_PRIO 3
#endif
        // The producers are the youngest waves of their SIMDs and would lose every issue arbitration against the two MFMA
        // waves (priority, then age): at equal priority their vector instructions issued once per ~32 cycles.
        __builtin_amdgcn_s_setprio(WS_PRODUCER_PRIO);
        const int oy = (p >> 1) * 8, ox = (p & 1) * 8;
        float *Hp = smem + WS_H + p * WS_HP;
        const int q = lane & 7, xl = lane >> 3;
        // A rows of this lane's column: MFMA row r of consumer wave (oy + y) >> 1, r = PIXMAP^-1(y & 1, ox + xl)
        const int Xt = ox + xl;
        const int r0 = Xt < 4 ? Xt : Xt < 8 ? Xt + 8 : Xt + 12;
        const int r1 = Xt < 8 ? Xt + 4 : Xt < 12 ? Xt + 8 : Xt + 16;
        const int aoff0 = ((oy >> 1) * 32 + r0) * 32 + ((q ^ ((r0 >> 1) & 7)) * 4);
        const int aoff1 = ((oy >> 1) * 32 + r1) * 32 + ((q ^ ((r1 >> 1) & 7)) * 4);
        const int hoff = xl * 32 + q * 4;                    // halo pixel (hy, xl + dx): + (hy * 10 + dx) * 32
        // B pieces 4p .. 4p+3 (8 weight rows of 128 B each): row 32p + 8i + (lane >> 3), 16-byte slot lane & 7 holds the
        // channel group (lane & 7) ^ ((row >> 1) & 7) = (lane & 7) ^ ((4i + (lane >> 4)) & 7)
        const unsigned bvo0 = (unsigned)((32 * p + (lane >> 3)) * Cin * 4 + (((lane & 7) ^ (lane >> 4)) * 16));
        const unsigned bvo1 = (unsigned)((32 * p + 8 + (lane >> 3)) * Cin * 4 + (((lane & 7) ^ (4 + (lane >> 4))) * 16));
        const unsigned bvo_p = (unsigned)((64 * (p & 1) + (lane >> 2)) * Cin * 2 + (((lane & 3) ^ ((lane >> 4) & 3)) * 16));      // PREC = 1

        // Halo addressing.  Piece i of the 13 LDS-DMA pieces carries halo pixels 8i .. 8i+7 (hp = 8i + (lane >> 3), row hp / 10,
        // column hp % 10 of the 10 x 10 halo), 16 bytes of channel quad lane & 7 per lane.  rel[i] is the lane's byte offset from
        // the halo's first pixel (tile independent); per tile only the buffer base moves and the pieces that fall outside the
        // image are redirected to an out-of-range offset (the buffer range check then returns the zeros of the padding).
        // tile_setup is pure vector arithmetic, so it runs in the vector phase (see below), never next to the consumers' MFMAs.
        unsigned rel[WS_NHP], hv[WS_NHP];
        unsigned m_top = 0, m_bot = 0, m_left = 0, m_right = 0, m_inv = 0;      // bit i: piece i of this lane lies in halo row 0 / row 9 / column 0 / column 9 / past the halo
#pragma unroll
        for (int i = 0; i < WS_NHP; i++) {
            const int hp = 8 * i + (lane >> 3);
            const int hy = hp / 10, hx = hp - hy * 10;
            rel[i] = (unsigned)((hy * a.W + hx) * Cin + (lane & 7) * 4) * 4u;
            m_top |= (unsigned)(hy == 0) << i; m_bot |= (unsigned)(hy == 9) << i;
            m_left |= (unsigned)(hx == 0) << i; m_right |= (unsigned)(hx == 9) << i;
            m_inv |= (unsigned)(hp >= 100) << i;
        }
        // STEM: the window of the patch.  Piece i of its WS_NWP LDS-DMA pieces carries window elements 64 i + lane (element e = row e / 22,
        // column e % 22 of the 21 x 22 window whose first element is patch pixel (2 Y0, 2 X0)), 4 bytes per lane.  A halo at the image border
        // reaches 2 patch rows / columns past the patch on that side (window rows 0-1 resp. 18-20, columns 0-1 resp. 18-21): those lanes are
        // redirected out of range like the halo pieces of the plain form (their halo pixels are zero padding anyway).
        unsigned badw = 0;
        unsigned w_top = 0, w_bot = 0, w_left = 0, w_right = 0, w_inv = 0;
        unsigned badm = 0;                                   // STEM: bit i = halo piece i of this lane lies outside the image (zero padding)
        float *Wp = smem + WS_WIN + p * (WS_NWP * 64);
        if (STEM) {
#pragma unroll
            for (int i = 0; i < WS_NWP; i++) {
                const int e = 64 * i + lane;
                const int wy = e / WS_WPITCH, wx = e - wy * WS_WPITCH;
                w_top |= (unsigned)(wy < 2) << i; w_bot |= (unsigned)(wy >= 18) << i;
                w_left |= (unsigned)(wx < 2) << i; w_right |= (unsigned)(wx >= 18) << i;
                w_inv |= (unsigned)(e >= WS_WP) << i;
            }
        }
        const float *hA = a.in;
        auto tile_setup = [&](int jj) {
            const int mt = jj / nNt;
            const int n = mt / TPP, tr = mt - n * TPP;
            const int Y0 = (tr / TW) * 16 + oy - 1, X0 = (tr % TW) * 16 + ox - 1;      // image position of the halo's first pixel
            const unsigned bad = m_inv | (Y0 < 0 ? m_top : 0u) | (Y0 + 9 >= a.H ? m_bot : 0u) | (X0 < 0 ? m_left : 0u) | (X0 + 9 >= a.W ? m_right : 0u);
            if (STEM) {
                hA = a.in + ((long)n * a.H * a.W * 4 + (long)(2 * Y0) * (2 * a.W) + 2 * X0);       // may lie before the tensor for border tiles: those lanes are masked
                badw = w_inv | (Y0 < 0 ? w_top : 0u) | (Y0 + 9 >= a.H ? w_bot : 0u) | (X0 < 0 ? w_left : 0u) | (X0 + 9 >= a.W ? w_right : 0u);
                badm = bad;
            } else {
                hA = a.in + ((long)n * a.H * a.W + (long)Y0 * a.W + X0) * Cin;             // may lie before the tensor for border tiles: those lanes are masked
#pragma unroll
                for (int i = 0; i < WS_NHP; i++) hv[i] = ((bad >> i) & 1u) ? OOB : rel[i];
            }
        };
        int hj = j0, hc = 0;                                 // next halo to load: (pair, chunk)
        auto issue_halo = [&]() {
            const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void *)hA, 0, 0x7fffffff, 0x00020000);
            if (STEM) {
                if (hc == 0) {          // a new tile: its window (both chunks of the tile are computed from it)
                    // the per-lane offsets are recomputed here, once per tile, from a lane id the compiler cannot see through: as loop
                    // invariants they would sit in 8 registers for the whole kernel (which then spills)
                    int lane_o = lane;
                    asm volatile("" : "+v"(lane_o));
#pragma unroll
                    for (int i = 0; i < WS_NWP; i++) {
                        const int e = 64 * i + lane_o;
                        const int wy = e / WS_WPITCH, wx = e - wy * WS_WPITCH;
                        const unsigned relw = (unsigned)(wy * 2 * a.W + wx) * 4u;
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t *)(Wp + i * 64), 4, ((badw >> i) & 1u) ? OOB : relw, 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < WS_NHP; i++)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t *)(Hp + i * 256), 16, hv[i], hc * 128, 0, 0);
            }
            if (++hc == nchunks) { hc = 0; hj++; }
        };
        WS_DIAG_DECL
        int pj = j0, pc = 0;                                 // step being produced: (pair, chunk)
        float4 wt[9];                                        // its depthwise taps
        float4 sw[9], ssc, ssh;                              // STEM: the stem's taps and folded BN for the step's 32 channels (this lane's quad)
        const __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc((void *)a.dw9, 0, 0x7fffffff, 0x00020000);
        int cur_pc = 0;                                      // STEM: chunk of the step whose operands are being produced
        auto load_taps = [&]() {                             // buffer loads: per-lane offset fixed, (tap, chunk) in the scalar offset -- no vector arithmetic
            if (!STEM) {
#pragma unroll
                for (int tp = 0; tp < 9; tp++)
                    wt[tp] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsT, q * 16, (tp * Cin + pc * 32) * 4, 0));
            }
            if (STEM) {
                cur_pc = pc;
                const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void *)a.stem_w, 0, 0x7fffffff, 0x00020000);
                const __amdgpu_buffer_rsrc_t rsC = __builtin_amdgcn_make_buffer_rsrc((void *)a.stem_scale, 0, 0x7fffffff, 0x00020000);
                const __amdgpu_buffer_rsrc_t rsH = __builtin_amdgcn_make_buffer_rsrc((void *)a.stem_shift, 0, 0x7fffffff, 0x00020000);
#pragma unroll
                for (int tp = 0; tp < 9; tp++)
                    sw[tp] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsW, q * 16, (tp * Cin + pc * 32) * 4, 0));
                ssc = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsC, q * 16, pc * 128, 0));
                ssh = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsH, q * 16, pc * 128, 0));
            }
        };
        // The f32 MFMA and the vector ALU do not run side by side on a SIMD: next to two consumer waves that keep the matrix
        // pipe full, a producer's FMAs issued once per 20-40 cycles whatever its priority (measured: 6 000-10 000 cycles for
        // the 290-410 vector instructions of a step, the consumers waiting at the barrier for half of that).  So a step has
        // two phases separated by workgroup barriers:
        //   MFMA phase   consumers: the step's 64 MFMAs each.  Producers: everything that is NOT vector arithmetic for the
        //                next step -- LDS-DMA of its halo, taps and pointwise weights, the wait for them, the first four halo
        //                rows into registers.
        //   vector phase producers: the next step's depthwise values (dense: nothing else wants the pipe).  Consumers: at
        //                a tile's last step the epilogue (also vector work + stores), otherwise nothing.
        float4 win[4][3];                                    // halo rows y .. y+3 of this lane's column
        auto load_row = [&](int hy, float4 *row) {
            const float *hb = Hp + hoff;
#pragma unroll
            for (int dx = 0; dx < 3; dx++) row[dx] = *reinterpret_cast<const float4 *>(hb + (hy * 10 + dx) * 32);
        };
        auto relu_row = [&](float4 *row) {
            if (RELU_IN) {
#pragma unroll
                for (int dx = 0; dx < 3; dx++) {
                    row[dx].x = fmaxf(row[dx].x, 0.f); row[dx].y = fmaxf(row[dx].y, 0.f);
                    row[dx].z = fmaxf(row[dx].z, 0.f); row[dx].w = fmaxf(row[dx].w, 0.f);
                }
            }
        };
        auto fetch = [&](int stage) {
            load_taps();
            issue_halo();
            {
                const int nt = pj % nNt;
                float *Bs = smem + (stage ? WS_B1 : WS_B0);
                if (PREC == 0) {
                    const __amdgpu_buffer_rsrc_t rsB =
                        __builtin_amdgcn_make_buffer_rsrc((void *)(a.pwk + (size_t)nt * 128 * Cin), 0, 0x7fffffff, 0x00020000);
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_t *)(Bs + (4 * p + i) * 256), 16,
                                                                 (i & 1) ? bvo1 : bvo0, pc * 128 + (i >> 1) * 16 * Cin * 4, 0, 0);
                } else {
                    // two bf16 planes of [128 rows][32 channels] = 64-byte rows; piece q = 4 p + i (1 KiB = 16 rows) of the 16: plane q >> 3,
                    // rows 16 (q & 7) + (lane >> 2), 16-byte unit (lane & 3) ^ ((row >> 2) & 3) of the row's 64 bytes
                    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
                        (void *)(reinterpret_cast<const char *>(a.pwk) + (size_t)nt * 128 * Cin * 2), 0, 0x7fffffff, 0x00020000);
                    const int plane = (p >> 1) * a.Cout * Cin * 2;
#pragma unroll
                    for (int i = 0; i < 4; i++)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_t *)(Bs + (4 * p + i) * 256), 16, bvo_p,
                                                                 pc * 64 + plane + i * 16 * Cin * 2, 0, 0);
                }
            }
            if (++pc == nchunks) { pc = 0; pj++; }
            WS_STAMP(0)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // taps, halo, weights: all landed before the phase ends
            WS_STAMP(1)
            if (!STEM) {            // (STEM: the halo does not exist yet -- it is computed at the top of the vector phase)
                load_row(0, win[0]);
                load_row(1, win[1]);
                load_row(2, win[2]);
                load_row(3, win[3]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            WS_STAMP(2)
        };
        // STEM: this producer's 10 x 10 halo of the step's 32 channels from the patch window.  Item i = halo pixel 8 i + (lane >> 3) (the pixel
        // piece i of the plain form's LDS-DMA would carry), channel quad lane & 7: the 3 x 3 window pixels (stride 2: rows 2 hy .., columns
        // 2 hx ..) as three 8-byte + three 4-byte broadcast reads, 36 multiply-adds in stem_kernel's order, folded BN, ReLU, one 16-byte write.
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        auto stem_halo = [&]() {
            // window reads one item ahead of the arithmetic, pinned: left alone, the scheduler hoists the reads of all 13 items to the top
            // (117 live values: the kernel then spills, and scratch traffic shares vmcnt with the LDS-DMA).  The arithmetic is written on
            // 2-vectors so that it compiles to v_pk_fma_f32 (18 per item for the taps, 2 for the folded BN); the zero padding outside the
            // image is a multiplication by a 0 / 1 lane mask (exact: the values are finite and >= +0) instead of four selects.
            float xw[2][9];
            int xl_o = xl;                  // opaque: the 13 window addresses are recomputed per step instead of living in 13 registers
            asm volatile("" : "+v"(xl_o));
            auto read_item = [&](int i, float *x9) {
                int hp = 8 * i + xl_o;
                if (i == WS_NHP - 1) hp = hp < 100 ? hp : 99;        // pieces past the halo (zeroed below) must not read past the window
                const int hy = (hp * 205) >> 11;                      // hp / 10 for hp < 1024
                const int hx = hp - hy * 10;
                const float *wb = Wp + (2 * hy) * WS_WPITCH + 2 * hx;
#pragma unroll
                for (int ky = 0; ky < 3; ky++) {
                    const float2 t2 = *reinterpret_cast<const float2 *>(wb + ky * WS_WPITCH);
                    x9[ky * 3] = t2.x; x9[ky * 3 + 1] = t2.y; x9[ky * 3 + 2] = wb[ky * WS_WPITCH + 2];
                }
            };
            const f32x2 sc01 = {ssc.x, ssc.y}, sc23 = {ssc.z, ssc.w}, sh01 = {ssh.x, ssh.y}, sh23 = {ssh.z, ssh.w};
            read_item(0, xw[0]);
#pragma unroll
            for (int i = 0; i < WS_NHP; i++) {
                if (i + 1 < WS_NHP) read_item(i + 1, xw[(i + 1) & 1]);
                WS_PIN()
                const float *x9 = xw[i & 1];
                f32x2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
#pragma unroll
                for (int tp = 0; tp < 9; tp++) {
                    const f32x2 vv = {x9[tp], x9[tp]};
                    const f32x2 w01 = {sw[tp].x, sw[tp].y}, w23 = {sw[tp].z, sw[tp].w};
                    a01 = __builtin_elementwise_fma(vv, w01, a01);
                    a23 = __builtin_elementwise_fma(vv, w23, a23);
                }
                a01 = __builtin_elementwise_fma(a01, sc01, sh01);
                a23 = __builtin_elementwise_fma(a23, sc23, sh23);
                const float keep = ((badm >> i) & 1u) ? 0.f : 1.f;       // outside the image: the depthwise convolution's zero padding
                const f32x2 kk = {keep, keep};
                f32x2 o01 = {fmaxf(a01.x, 0.f), fmaxf(a01.y, 0.f)}, o23 = {fmaxf(a23.x, 0.f), fmaxf(a23.y, 0.f)};
                o01 = o01 * kk; o23 = o23 * kk;
                *reinterpret_cast<float4 *>(Hp + i * 256 + lane * 4) = make_float4(o01.x, o01.y, o23.x, o23.y);
                WS_PIN()
            }
        };
        // One round = two output rows (acc0: row y, acc1: row y + 1) from the window rows y .. y + 3, each chain in (ky, kx)
        // order.  (Tried and rejected, +3.6 ms over the four layers: requesting rows y + 4 / y + 5 in the middle of a round, as
        // soon as rows y / y + 1 are dead, so that they land under the remaining taps -- at tile-end steps the producers then
        // queue their LDS reads behind the consumers' epilogue traffic twice per round instead of once.)
        auto taps3 = [&](float4 &acc, const float4 *row, int ky) {
#pragma unroll
            for (int kx = 0; kx < 3; kx++) {
                const float4 v = row[kx], w = wt[ky * 3 + kx];
                acc.x = fmaf(v.x, w.x, acc.x); acc.y = fmaf(v.y, w.y, acc.y);
                acc.z = fmaf(v.z, w.z, acc.z); acc.w = fmaf(v.w, w.w, acc.w);
            }
        };
        auto compute = [&](int stage) {
            float *As = smem + (stage ? WS_A1 : WS_A0);
            if (STEM) {
                stem_halo();
                WS_PIN()
#pragma unroll
                for (int tp = 0; tp < 9; tp++) wt[tp] = *reinterpret_cast<const float4 *>(smem + WS_A + tp * Cin + cur_pc * 32 + q * 4);
                load_row(0, win[0]);        // LDS operations of one wave execute in order: these reads see the writes above
                load_row(1, win[1]);
                load_row(2, win[2]);
                load_row(3, win[3]);
            }
#pragma unroll
            for (int y = 0; y < 8; y += 2) {
                if (y) {
                    load_row(y + 2, win[(y + 2) & 3]);
                    load_row(y + 3, win[(y + 3) & 3]);
                }
                WS_PIN()
                if (y == 0) { relu_row(win[0]); relu_row(win[1]); }
                relu_row(win[(y + 2) & 3]);
                relu_row(win[(y + 3) & 3]);
                float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0;
                taps3(acc0, win[y & 3], 0);
                taps3(acc0, win[(y + 1) & 3], 1);
                taps3(acc0, win[(y + 2) & 3], 2);
                taps3(acc1, win[(y + 1) & 3], 0);
                taps3(acc1, win[(y + 2) & 3], 1);
                taps3(acc1, win[(y + 3) & 3], 2);
                *reinterpret_cast<float4 *>(As + aoff0 + (y >> 1) * 1024) = acc0;
                *reinterpret_cast<float4 *>(As + aoff1 + (y >> 1) * 1024) = acc1;
                WS_PIN()
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // A written (and the halo reads done: the next fetch refills it)
            if (hc == 0 && hj < j1) tile_setup(hj);                   // the next fetch starts a new tile
        };

        tile_setup(j0);
        fetch(0);
        compute(0);
        WS_BAR()
        int cc = 0;
        for (int s = 0; s < total; s++) {
            WS_TL(0)
#ifdef WS_VAR_SLEEP
            __builtin_amdgcn_s_sleep(WS_VAR_SLEEP);
#endif
#ifdef WS_VAR_FETCHPRIO
            __builtin_amdgcn_s_setprio(WS_VAR_FETCHPRIO);
#endif
            if (s + 1 < total) fetch((s + 1) & 1);
            WS_STAMP(3)
            WS_TL(1)
            WS_BAR()                                         // consumers: MFMAs of step s issued
#ifdef WS_VAR_FETCHPRIO
            __builtin_amdgcn_s_setprio(WS_PRODUCER_PRIO);
#endif
            WS_STAMP(4)
            WS_TL(2)
            if (s + 1 < total) compute((s + 1) & 1);
            WS_STAMP(5)
            WS_TL(3)
            if (++cc == nchunks) {
                cc = 0;
                if (POOL) WS_BAR()                           // inside the consumers' pooling epilogue
            }
            WS_BAR()                                         // step s + 1's operands are in LDS
            WS_STAMP(6)
            WS_TL(4)
        }
#ifdef WS_DIAG
        dsum[7] = total;
#endif
        WS_DIAG_FLUSH()
        return;
    }

    // ================================ consumers ================================
    const int r = lane & 31, h = lane >> 5;
    const int key = (r >> 1) & 7;
    const int aoffc = (32 * wave + r) * 32, boffc = r * 32;

    f32x16 acc[4];
#pragma unroll
    for (int jn = 0; jn < 4; jn++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[jn][e] = 0.f;

    // Folded-BN scale / shift of this lane's 4 output channels and (POOL) the residual values of its pooled pixels are loaded
    // BEFORE the MFMAs of a tile's steps, never inside the epilogue: a load there is followed by s_waitcnt vmcnt(0), which
    // also waits for every store the wave has in flight -- the epilogue would drain its own stores four times per tile
    // instead of leaving them to complete under the next tile's MFMAs.
    float scv[4], shv[4];
    auto load_scale_shift = [&](int jj) {
        const int n0 = (jj % nNt) * 128;
#pragma unroll
        for (int jn = 0; jn < 4; jn++) { scv[jn] = a.scale[n0 + jn * 32 + r]; shv[jn] = a.shift[n0 + jn * 32 + r]; }
    };
    // Epilogue addressing without vector arithmetic: buffer loads / stores whose per-lane offset is a kernel constant (this
    // lane's channel and tile row), the tile in the buffer base and (pixel, channel group) in the scalar offset -- in the
    // vector phase every vector instruction of the 8 consumer waves delays the producers' depthwise arithmetic and vice versa.
    const int CoutB = a.Cout * 4;
    const float relu_lo = a.relu_out ? 0.f : -__builtin_inff();        // fmaxf(v, relu_lo): ReLU or identity without a select per value
    const unsigned vo_pool = (unsigned)(4 * h * a.Cout + r) * 4u;        // pooled pixel 4h (+ q), channel r (+ 32 jn)
    float rv[4][4];
    auto load_resid = [&](int jj) {
        const int mt = jj / nNt, nt = jj - mt * nNt;
        const int n = mt / TPP, tr = mt - n * TPP;
        const int ty = tr / TW, tx = tr - ty * TW;
        const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(a.resid + (((size_t)n * (a.H >> 1) + ty * 8 + wave) * (a.W >> 1) + tx * 8) * a.Cout + nt * 128), 0, 0x7fffffff, 0x00020000);
        if (wave < 7) {             // the tile's last pooled row is finished (and gets its residual) in pool_fix_add_kernel
#pragma unroll
            for (int jn = 0; jn < 4; jn++)
#pragma unroll
                for (int qq = 0; qq < 4; qq++)      // h = 1, q = 3 is column 7 (finished in pool_fix_add_kernel too): loaded but not used
                    rv[jn][qq] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsR, vo_pool, qq * CoutB + jn * 128, 0));
        }
    };

    // ---- epilogue of one tile.  C/D layout of a 32x32 accumulator: column = lane & 31 = output channel, register e = MFMA row
    // rho = (e & 3) + 8 (e >> 2) + 4 h = tile pixel (y, x) with x = e and y = h for e in {0-3, 12-15}, 1 - h for e in 4..11 (PIXMAP).
    // Plain epilogue: 64 one-dword stores per wave and tile (two 128-byte lines each) kept the CU's one texture-address path busy
    // for ~8 000 cycles per tile (512 store instructions, ~16 cycles each).  The values go through the wave's PRIVATE 4 KiB of the
    // A stage the tile's last step has just consumed instead -- [32 tile pixels][32 channels], one 32-channel slab (jn) at a time --
    // and leave as 16-byte stores: 16 store instructions per wave and tile, each 8 pixels x one full 128-byte line.  No barrier:
    // LDS operations of one wave execute in order.
    unsigned vo_t[4];            // tile pixel 8 i + (lane >> 3) of this wave's two rows, channel quad lane & 7
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int pix = 8 * i + (lane >> 3);
        vo_t[i] = (unsigned)(((pix >> 4) * a.W + (pix & 15)) * a.Cout + (lane & 7) * 4) * 4u;
    }
    auto store_tile = [&](int jj, float *xw) {
        const int mt = jj / nNt, nt = jj - mt * nNt;
        const int n = mt / TPP, tr = mt - n * TPP;
        const int ty0 = (tr / TW) * 16, tx0 = (tr % TW) * 16;
        const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(a.out + ((size_t)n * a.H * a.W + (size_t)(ty0 + 2 * wave) * a.W + tx0) * a.Cout + nt * 128), 0, 0x7fffffff, 0x00020000);
        float *xl = xw + r;
        const float *xr = xw + lane * 4;
#pragma unroll
        for (int jn = 0; jn < 4; jn++) {
            const float sc = scv[jn], sh = shv[jn];
#pragma unroll
            for (int e = 0; e < 16; e++) {      // register e = tile pixel (y, x = e), y = h for e in {0-3, 12-15}, 1 - h for e in 4..11 (PIXMAP)
                const int yy = (e >= 4 && e < 12) ? 1 - h : h;
                xl[(yy * 16 + e) * 32] = fmaxf(fmaf(acc[jn][e], sc, sh), relu_lo);
                acc[jn][e] = 0.f;
            }
            float4 o[4];
#pragma unroll
            for (int i = 0; i < 4; i++) o[i] = *reinterpret_cast<const float4 *>(xr + i * 256);
#pragma unroll
            for (int i = 0; i < 4; i++)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, o[i]), rsO, vo_t[i], jn * 128, 0);
        }
    };

    // ---- POOL epilogue: MaxPooling2D(3, 2, "same") of the tile straight from the accumulators (TF pads AFTER: pooled pixel (py, px) covers
    // rows 2 py .. 2 py + 2).  PIXMAP puts, for every x-quad, tile row 2w in one half wave and row 2w + 1 in the other, so one cross-half
    // exchange gives a half wave both rows of its 9 columns; row 2w + 2 comes from wave w + 1 through the exchange buffer (already pooled
    // along x).  Row 7 / column 7 of a tile need row 16 / column 16 of the next tile: they are stored as partial maxima, the tile's own
    // row 0 / column 0 / corner go to three strips, and pool_fix_add_kernel finishes the 15 boundary pixels per tile in place.
    int tl_s = -1;          // step index for the WS_DIAG timeline stamps inside the epilogue
    auto store_tile_pool = [&](int jj, float *xch) {
#ifdef WS_DIAG
        const int s = tl_s;
#endif
        constexpr int NW = 8, XC = 128;
        const int mt = jj / nNt, nt = jj - mt * nNt;
        const int n = mt / TPP, tr = mt - n * TPP;
        const int ty = tr / TW, tx = tr - ty * TW;
        const int n0 = nt * 128;
        const size_t T = (size_t)n * TPP + tr;
        float hm[4][4], vs[4];
        constexpr float NEG = -__builtin_inff();
        const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(a.out + (((size_t)n * (a.H >> 1) + ty * NW + wave) * (a.W >> 1) + tx * 8) * a.Cout + n0), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsSH = __builtin_amdgcn_make_buffer_rsrc((void *)(a.strip_h + T * 8 * a.Cout + n0), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsSV = __builtin_amdgcn_make_buffer_rsrc((void *)(a.strip_v + (T * NW + wave) * a.Cout + n0), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsCO = __builtin_amdgcn_make_buffer_rsrc((void *)(a.corner + T * a.Cout + n0), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int jn = 0; jn < 4; jn++) {
            const float sc = scv[jn], sh = shv[jn];
            float v[16];
            if (a.relu_out) {             // uniform branch: the network's pooled layers have no ReLU, and a max per value is matrix time
#pragma unroll
                for (int e = 0; e < 16; e++) v[e] = fmaxf(fmaf(acc[jn][e], sc, sh), 0.f);
            } else {
#pragma unroll
                for (int e = 0; e < 16; e++) v[e] = fmaf(acc[jn][e], sc, sh);
            }
#pragma unroll
            for (int e = 0; e < 16; e++) acc[jn][e] = 0.f;
            // columns 8 h + k, k = 0..7: own register 8 h + k; the other half wave's register of the same index holds the other tile
            // row.  v_permlane32_swap_b32 A, B exchanges A's upper half wave with B's lower one: with A = v[k], B = v[8 + k] a lane of
            // half 0 then holds (own v[k], the other half's v[k]) and a lane of half 1 (the other half's v[8 + k], own v[8 + k]) -- both rows
            // of its column in (A, B), one vector instruction instead of two selects and a ds_bpermute.  Inline assembly: the ROCm 7.2
            // builtin returned wrong values in this register pattern (tools/dev/permlane_probe.hip checks this form against the shuffles).
#ifndef WS_POOL_SHUFFLE
            const float v8 = v[8], o8 = __shfl_xor(v8, 32);      // column 8 (half 0 only): taken before the swaps
            float y0[9], m01[9];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                float A = v[k], B = v[8 + k];
                asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(A), "+v"(B));
                m01[k] = fmaxf(A, B);
                y0[k] = ((h != 0) != (k >= 4)) ? B : A;          // tile row 2 w: own for quads 0, 2 of the half, the other's for quads 1, 3
            }
            y0[8] = h ? NEG : o8;       // column 8 h + 8: x = 8 for h = 0 (register 8: this half holds its y1, the other half its y0), none for h = 1
            m01[8] = h ? NEG : fmaxf(o8, v8);
#else
            float y0[9], y1[9];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const float own = h ? v[8 + k] : v[k];
                const float oth = __shfl_xor(h ? v[k] : v[8 + k], 32);      // the other half sends what this one lacks
                y0[k] = k < 4 ? own : oth;                   // quads 0, 2: this half holds y0; quads 1, 3: y1
                y1[k] = k < 4 ? oth : own;
            }
            {   // column 8 h + 8: x = 8 for h = 0 (register 8: this half holds its y1, the other half its y0), none for h = 1
                const float o8 = __shfl_xor(v[8], 32);
                y0[8] = h ? NEG : o8;
                y1[8] = h ? NEG : v[8];
            }
            float m01[9];
#pragma unroll
            for (int k = 0; k < 9; k++) m01[k] = fmaxf(y0[k], y1[k]);
#endif
            float h0[4];
#pragma unroll
            for (int qq = 0; qq < 4; qq++) {
                hm[jn][qq] = fmaxf(fmaxf(m01[2 * qq], m01[2 * qq + 1]), m01[2 * qq + 2]);
                h0[qq] = fmaxf(fmaxf(y0[2 * qq], y0[2 * qq + 1]), y0[2 * qq + 2]);
                xch[(wave * 9 + 4 * h + qq) * XC + jn * 32 + r] = h0[qq];
                if (wave == 0) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, h0[qq]), rsSH, vo_pool, qq * CoutB + jn * 128, 0);
            }
            vs[jn] = m01[0];
            if (h == 0) {
                xch[(wave * 9 + 8) * XC + jn * 32 + r] = y0[0];
                if (wave == 0) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y0[0]), rsCO, r * 4, jn * 128, 0);
            }
        }
        WS_TL(9)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        WS_TL(10)
        WS_BAR()
        WS_TL(11)
#pragma unroll
        for (int jn = 0; jn < 4; jn++) {
#pragma unroll
            for (int qq = 0; qq < 4; qq++) {
                float pv = hm[jn][qq];
                if (wave < NW - 1) {
                    pv = fmaxf(pv, xch[((wave + 1) * 9 + 4 * h + qq) * XC + jn * 32 + r]);
                    if (qq < 3 || h == 0) pv = pv + rv[jn][qq];
                }
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, pv), rsP, vo_pool, qq * CoutB + jn * 128, 0);
            }
            if (h == 0) {
                float cv = vs[jn];
                if (wave < NW - 1) cv = fmaxf(cv, xch[((wave + 1) * 9 + 8) * XC + jn * 32 + r]);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, cv), rsSV, r * 4, jn * 128, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // exchange values read before the step's closing barrier
        WS_TL(12)
    };

    WS_DIAG_DECL
    WS_BAR()                                                 // step 0's operands are in LDS
    // Fragment reads.  The pointwise weights of step s + 1 are complete in LDS when the MFMA phase of step s ends (the producers
    // wait for their LDS-DMA before barrier X), the A operand only when its vector phase ends (barrier Y).  The consumers idle
    // through the vector phase, so they read the weight fragments of the next step's first two k groups THERE; after barrier Y
    // only the two A reads stand between a wave and its first MFMA.
    float4 av[2], bv[2][4];
#define WS_READ_A(g) av[(g) & 1] = *reinterpret_cast<const float4 *>(As + ((((2 * (g)) + h) ^ key) * 4));
#define WS_READ_B(g, Bp)                                                                                        \
        {                                                                                                       \
            const int slot = (((2 * (g)) + h) ^ key) * 4;                                                       \
            _Pragma("unroll") for (int jn = 0; jn < 4; jn++) bv[(g) & 1][jn] = *reinterpret_cast<const float4 *>((Bp) + jn * 1024 + slot); \
        }
#define WS_MM(g)                                                                                                \
        {                                                                                                       \
            _Pragma("unroll") for (int jn = 0; jn < 4; jn++) acc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(g) & 1].x, bv[(g) & 1][jn].x, acc[jn], 0, 0, 0); \
            _Pragma("unroll") for (int jn = 0; jn < 4; jn++) acc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(g) & 1].y, bv[(g) & 1][jn].y, acc[jn], 0, 0, 0); \
            _Pragma("unroll") for (int jn = 0; jn < 4; jn++) acc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(g) & 1].z, bv[(g) & 1][jn].z, acc[jn], 0, 0, 0); \
            _Pragma("unroll") for (int jn = 0; jn < 4; jn++) acc[jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[(g) & 1].w, bv[(g) & 1][jn].w, acc[jn], 0, 0, 0); \
        }
    // PREC = 1: weight fragments of one k step (16 channels): plane pl, column block jn -> 8 bf16 of row jn 32 + r, unit (2 tk + h) ^ ((r >> 2) & 3)
    bf16x8 wq[2][4];
    const int boffp = r * 16, keyb = (r >> 2) & 3;
#define WS_READ_BP(tk, Bp)                                                                                      \
        _Pragma("unroll") for (int pl = 0; pl < 2; pl++)                                                        \
            _Pragma("unroll") for (int jn = 0; jn < 4; jn++)                                                    \
                wq[pl][jn] = *reinterpret_cast<const bf16x8 *>((Bp) + pl * 2048 + jn * 512 + boffp + ((((2 * (tk)) + h) ^ keyb) * 4));
    if (PREC == 0) {
        const float *B0 = smem + WS_B0 + boffc;
        WS_READ_B(0, B0)
        WS_READ_B(1, B0)
    } else {
        const float *B0 = smem + WS_B0;
        WS_READ_BP(0, B0)
    }
    int cj = j0, cc = 0;
    for (int s = 0; s < total; s++) {
        const int stage = s & 1;
        WS_STAMP(0)
        // (scale / shift of a tile and the residual values of its pooled pixels are requested in the vector phase BEFORE the step that
        // needs them: at the top of a step the producers have just queued their 26 LDS-DMA / buffer loads, and a consumer's 16 residual
        // loads behind them held its first MFMA back by ~3 300 cycles -- WS_DIAG timeline, tile-end step)
        if (s == 0) load_scale_shift(cj);
        if (POOL && nchunks == 1) load_resid(cj);
        const float *As = smem + (stage ? WS_A1 : WS_A0) + aoffc;
        const float *Bs = smem + (stage ? WS_B1 : WS_B0) + boffc;
        WS_TL(0)
        if (PREC == 1) {
            // k step tk covers channels 16 tk .. 16 tk + 15; a lane's 8 values are channels 16 tk + 8 h + j: 16-byte units 4 tk + 2 h and
            // 4 tk + 2 h + 1 of its f32 row
            float4 af[2][2];
#pragma unroll
            for (int tk = 0; tk < 2; tk++) {
                af[tk][0] = *reinterpret_cast<const float4 *>(As + (((4 * tk + 2 * h) ^ key) * 4));
                af[tk][1] = *reinterpret_cast<const float4 *>(As + (((4 * tk + 2 * h + 1) ^ key) * 4));
            }
            WS_PIN()
#pragma unroll
            for (int tk = 0; tk < 2; tk++) {
                if (tk == 1) {              // the second k step's weights: behind the first one's MFMAs, into the same registers
                    const float *Bq = smem + (stage ? WS_B1 : WS_B0);
                    WS_READ_BP(1, Bq)
                    WS_PIN()
                }
                const float xs[8] = {af[tk][0].x, af[tk][0].y, af[tk][0].z, af[tk][0].w, af[tk][1].x, af[tk][1].y, af[tk][1].z, af[tk][1].w};
                bf16x8 ahi, alo;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const __bf16 q = (__bf16)xs[j];
                    ahi[j] = q;
                    alo[j] = (__bf16)(xs[j] - (float)q);
                }
#pragma unroll
                for (int jn = 0; jn < 4; jn++) {        // smallest products first: lo hi, hi lo, hi hi
                    acc[jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, wq[0][jn], acc[jn], 0, 0, 0);
                    acc[jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, wq[1][jn], acc[jn], 0, 0, 0);
                    acc[jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, wq[0][jn], acc[jn], 0, 0, 0);
                }
                WS_PIN()
            }
        } else {
        WS_READ_A(0)
        WS_READ_A(1)
        WS_PIN()
#ifdef WS_DIAG
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        WS_TL(1)
        WS_PIN()
#endif
        WS_MM(0)
        WS_PIN()
        WS_TL(2)
        WS_READ_A(2)
        WS_READ_B(2, Bs)
        WS_PIN()
        WS_MM(1)
        WS_PIN()
        WS_TL(3)
        WS_READ_A(3)
        WS_READ_B(3, Bs)
        WS_PIN()
        WS_MM(2)
        WS_PIN()
        WS_TL(4)
        WS_MM(3)
        WS_PIN()
        }
        WS_TL(5)
        WS_STAMP(1)
        WS_BAR()                                             // vector phase: the producers compute the next step's A operand
        WS_STAMP(2)
        WS_TL(6)
        if (++cc == nchunks) {
            cc = 0;
            tl_s = s;
            if (POOL) store_tile_pool(cj, smem + (stage ? WS_A : 0));
            else store_tile(cj, smem + (stage ? WS_A1 : WS_A0) + wave * 1024);
            cj++;
        }
        WS_PIN()
        if (s + 1 < total) {
            if (cc == 0) load_scale_shift(cj);                                   // the next step starts tile cj
            if (POOL && nchunks > 1 && cc == nchunks - 1) load_resid(cj);       // the next step ends tile cj
        }
        WS_PIN()
        if (s + 1 < total) {
            if (PREC == 0) {
                const float *Bn = smem + (stage ? WS_B0 : WS_B1) + boffc;
                WS_READ_B(0, Bn)
                WS_READ_B(1, Bn)
            } else {
                const float *Bn = smem + (stage ? WS_B0 : WS_B1);
                WS_READ_BP(0, Bn)
            }
        }
        WS_PIN()
        WS_STAMP(3)
        WS_TL(7)
        WS_BAR()
        WS_STAMP(4)
        WS_TL(8)
    }
#undef WS_READ_A
#undef WS_READ_B
#undef WS_READ_BP
#undef WS_MM
#ifdef WS_DIAG
    dsum[7] = total;
#endif
    WS_DIAG_FLUSH()
}

bool sepconv_ws_supported(int H, int W, int Cin, int Cout)
{
    return H % 16 == 0 && W % 16 == 0 && Cin % 32 == 0 && Cout % 128 == 0 &&
           (long long)H * W * Cin * 4 < 0x7fffffffLL && (long long)Cout * Cin * 4 < 0x7fffffffLL;
}

static int ws_cus()
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

template <bool POOL>
static void launch_ws_any(const WsArgs &a, int relu_in, int prec, hipStream_t s)
{
    const int nMt = a.N * (a.H / 16) * (a.W / 16), nNt = a.Cout / 128;
    const int G = (ws_cus() / 8) * 8;                       // one persistent 12-wave workgroup per CU (152 KiB of LDS)
    if (prec == 1) {
        if (relu_in) hipLaunchKernelGGL((sepconv_ws_kernel<true, POOL, 1>), dim3(G), dim3(768), 0, s, a, nMt, nNt, G);
        else hipLaunchKernelGGL((sepconv_ws_kernel<false, POOL, 1>), dim3(G), dim3(768), 0, s, a, nMt, nNt, G);
    } else if (relu_in) hipLaunchKernelGGL((sepconv_ws_kernel<true, POOL>), dim3(G), dim3(768), 0, s, a, nMt, nNt, G);
    else hipLaunchKernelGGL((sepconv_ws_kernel<false, POOL>), dim3(G), dim3(768), 0, s, a, nMt, nNt, G);
#ifdef WS_DIAG
    {
        static long long hbuf[256 * 12 * 8];
        hipStreamSynchronize(s);
        hipMemcpyFromSymbol(hbuf, HIP_SYMBOL(ws_diag), sizeof(hbuf));
        double cs[8] = {0}, ps[8] = {0};
        const int nb = G < 256 ? G : 256;
        for (int b = 0; b < nb; b++)
            for (int w = 0; w < 12; w++)
                for (int k = 0; k < 8; k++) (w < 8 ? cs : ps)[k] += (double)hbuf[((size_t)b * 12 + w) * 8 + k];
        const double cst = cs[7], pst = ps[7];
        fprintf(stderr, "[wsdiag] H %d Cin %d Cout %d pool %d | consumer per step: top %.0f mfma %.0f barX %.0f epilogue %.0f barY %.0f | producer per step: issue %.0f vmcnt0 %.0f rows %.0f barX %.0f compute %.0f barY %.0f | steps/wave %.0f\n",
                a.H, a.Cin, a.Cout, (int)POOL, cs[0] / cst, cs[1] / cst, cs[2] / cst, cs[3] / cst, cs[4] / cst,
                ps[0] / pst, ps[1] / pst, ps[2] / pst, (ps[3] + ps[4]) / pst, ps[5] / pst, ps[6] / pst, cst / (nb * 8.0));
        static long long tl[12 * 2 * 14];
        hipMemcpyFromSymbol(tl, HIP_SYMBOL(ws_tl), sizeof(tl));
        for (int st = 0; st < 2; st++) {
            long long t0 = tl[st * 14];
            for (int w = 0; w < 12; w++) if (tl[(w * 2 + st) * 14] < t0) t0 = tl[(w * 2 + st) * 14];
            for (int w = 0; w < 12; w++) {
                fprintf(stderr, "[wstl] step %d wave %2d:", st ? 15 : 13, w);
                for (int k = 0; k < (w < 8 ? 13 : 5); k++) fprintf(stderr, " %6lld", tl[(w * 2 + st) * 14 + k] - t0);
                fprintf(stderr, "\n");
            }
        }
    }
#endif
}

bool launch_sepconv_ws(const float *in, int N, int H, int W, int Cin, int relu_in, const float *dw9, const float *pwk, int Cout,
                       const float *scale, const float *shift, int relu_out, float *out, hipStream_t s, int prec)
{
    if (!sepconv_ws_supported(H, W, Cin, Cout) || N <= 0 || (long long)N * (H / 16) * (W / 16) * (Cout / 128) > 0x3fffffffLL) {
        set_error("launch_sepconv_ws: unsupported shape");
        return false;
    }
    WsArgs a{in, N, H, W, Cin, Cout, dw9, pwk, scale, shift, relu_out, out, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    launch_ws_any<false>(a, relu_in, prec, s);
    return true;
}

bool launch_sepconv_ws_stem(const float *x, int N, int H, int W, int Cin, const float *stem_w, const float *stem_scale, const float *stem_shift,
                            const float *dw9, const float *pwk, int Cout, const float *scale, const float *shift, int relu_out, float *out,
                            hipStream_t s)
{
    // the patch window offsets are 32-bit byte offsets from the tile's first window pixel: (21 rows of 2 W floats) always fits; the patch
    // itself is addressed through a 64-bit base
    if (!sepconv_ws_supported(H, W, Cin, Cout) || N <= 0 || (long long)N * (H / 16) * (W / 16) * (Cout / 128) > 0x3fffffffLL || 9 * Cin > WS_X) {
        set_error("launch_sepconv_ws_stem: unsupported shape");
        return false;
    }
    WsArgs a{x, N, H, W, Cin, Cout, dw9, pwk, scale, shift, relu_out, out, nullptr, nullptr, nullptr, nullptr, stem_w, stem_scale, stem_shift};
    const int nMt = N * (H / 16) * (W / 16), nNt = Cout / 128;
    const int G = (ws_cus() / 8) * 8;
    hipLaunchKernelGGL((sepconv_ws_kernel<false, false, 0, true>), dim3(G), dim3(768), 0, s, a, nMt, nNt, G);
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
static void launch_pool_fix_add(float *out, const float *sh, const float *sv, const float *co, const float *resid, int N, int H, int W, int Cout, int nw,
                         hipStream_t s)
{
    int c4shift = 0;
    while ((1 << c4shift) < Cout / 4) c4shift++;
    const int total = (H / (2 * nw)) * (W / 16) * (8 + nw - 1) * (Cout / 4);
    hipLaunchKernelGGL(pool_fix_add_kernel, dim3((total + 255) / 256, N), dim3(256), 0, s, out, sh, sv, co, resid, H / 2, W / 2, Cout, c4shift, total, nw);
}

// strips of the pooled form: row 0 / column 0 of every tile, already pooled along the row / column, and its corner pixel (8 + 8 + 1 pixels per 16 x 16 tile)
size_t sepconv_pool_scratch_floats(int N, int H, int W, int Cout)
{
    return (size_t)N * (H / 16) * (W / 16) * 17 * Cout;
}


bool launch_sepconv_pool_ws(const float *in, int N, int H, int W, int Cin, int relu_in, const float *dw9, const float *pwk, int Cout,
                            const float *scale, const float *shift, int relu_out, float *scratch, const float *resid, float *out, hipStream_t s, int prec)
{
    if (!sepconv_ws_supported(H, W, Cin, Cout) || N <= 0 || (long long)N * (H / 16) * (W / 16) * (Cout / 128) > 0x3fffffffLL ||
        ((Cout / 4) & (Cout / 4 - 1))) {
        set_error("launch_sepconv_pool_ws: unsupported shape");
        return false;
    }
    const size_t tiles = (size_t)N * (H / 16) * (W / 16);
    float *sh = scratch, *sv = sh + tiles * 8 * Cout, *co = sv + tiles * 8 * Cout;
    WsArgs a{in, N, H, W, Cin, Cout, dw9, pwk, scale, shift, relu_out, out, resid, sh, sv, co, nullptr, nullptr, nullptr};
    launch_ws_any<true>(a, relu_in, prec, s);
    launch_pool_fix_add(out, sh, sv, co, resid, N, H, W, Cout, 8, s);
    return true;
}

}  // namespace tmat
