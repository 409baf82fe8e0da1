// UNet-Xception inference kernels for gfx950 (MI355X).  NHWC float32.
//
// Reference graph: fl_tissue_model_tools/models.py:110-166 (build_UNetXception), executed by
// keras Model.predict at smooth_tiled_predictions.py:179.
//
// Arithmetic contract (shared with oracle/unet_exact.c, compared bit-exactly):
//   every contraction is a chain acc = fmaf(a[k], w[k], acc); for the MFMA convolutions k walks the
//   input channels in blocks of 32, inside a block tap-major (ky, kx), then the block's channels in the order
//   0,4,1,5,2,6,3,7, 8,12,9,13,... (the order ds_read_b128 fragments feed the MFMA; see conv_mfma_kernel)
//   (depthwise / stem / final: tap-major, channel ascending); v_mfma_f32_32x32x2_f32 performs exactly
//   such a chain (one rounding per product, k ascending), so the dense 3x3 / 1x1 contractions run on
//   the matrix cores at full f32 precision.  Epilogues: v = fmaf(acc, scale, shift) (folded BN) or acc + bias,
//   optional residual add, optional ReLU.  Compiled with -ffp-contract=off.
#include "dev_guard.h"
#include "tmat_internal.h"
#include "../../include/tmat.h"

#include <cstdlib>

namespace tmat {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------
// implicit-GEMM convolution on MFMA:  C[m][co] = sum_{cb, tap, k in block} A(m, tap, cb*32+k) * W[tap][co][cb*32+k]
//   m = flattened (n, y, x) output pixel.  Both operands live in LDS as [row][32 channels] images (128-byte rows,
//   k contiguous -- the NHWC layout of the activations and the [tap][Cout][Cin] layout of the weights), filled by
//   LDS-DMA (buffer_load_dwordx4 ... lds: no VGPR staging, no ds_write pass).  The 16-byte slot p of row r holds the
//   channel group p ^ ((r >> 1) & 7): the DMA destination is lane-linear, so the swizzle is applied to the per-lane SOURCE
//   address, and to the slot index on the read side; fragments are read with ds_read_b128.  A ds_read_b128 is served in
//   16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31} (+32); the 16 rows of a group are distinct mod 16, so
//   (row & 1, slot ^ ((row >> 1) & 7)) names 16 distinct 16-byte bank groups: conflict-free (with the key r & 7 of round 1,
//   rows 12 and 20 of a group shared a bank group: SQ_LDS_BANK_CONFLICT was half of SQ_LDS_IDX_ACTIVE).
//   One K chunk = one (32-channel block, tap); out-of-image taps are out-of-range buffer offsets (zeros).  Two LDS stages: the DMA of chunk
//   c+1 is issued before the MFMAs of chunk c and retired (vmcnt(0)) at the single barrier that ends chunk c.
//   256 threads = 4 waves arranged WM x WN, each wave owns (BM/WM) x (BN/WN) outputs as TM x TN tiles of 32x32.
//   A lane's b128 fragment holds channels 8g+4h .. 8g+4h+3 (h = lane >> 5) of its row, so the four MFMAs of a group
//   consume k = (8g+e, 8g+4+e), e = 0..3: the chain order inside a 32-channel block is 0,4,1,5,2,6,3,7, 8,12,9,13, ...
//   (oracle/unet_exact.c:conv_block walks the same order).
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_t;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// Exact division of a non-negative int (< 2^31) by a launch constant d >= 2 as one v_mul_hi_u32 and one shift: with
// l = ceil(log2 d) and mul = ceil(2^(31 + l) / d) < 2^32, floor(n mul / 2^(31 + l)) = floor(n / d) for every n < 2^31 (the
// excess n e / (d 2^(31 + l)), e = mul d - 2^(31 + l) < d, stays below 2^-l <= 1 / d).  A runtime `/` costs ~35 vector
// instructions, and vector instructions do not run beside the other workgroup's f32 MFMAs on the SIMD: the pixel <-> (n, y, x)
// conversions of the prologue and of the sub-pixel / residual epilogue were 3-8 % of a short-K tile.
struct FastDiv {
    unsigned mul;
    int sh;
};
static FastDiv make_fastdiv(int d)
{
    int l = 0;
    while ((1ll << l) < d) l++;
    const unsigned long long num = 1ull << (31 + l);
    return FastDiv{(unsigned)((num + (unsigned long long)d - 1) / (unsigned long long)d), l - 1};
}
__device__ __forceinline__ int fdiv(int n, FastDiv f) { return (int)(__umulhi((unsigned)n, f.mul) >> f.sh); }

// ReLU as one v_max_i32: as a signed integer a negative float (and -0.0) is negative, so max with 0 gives x > 0 ? x : +0.0 exactly
// (no NaNs on this path), without fmaxf's choice between the two zeros
__device__ __forceinline__ float relu_pos0(float v) { return __builtin_bit_cast(float, max(__builtin_bit_cast(int, v), 0)); }

// PREC = 0: f32 operands on v_mfma_f32_32x32x2_f32 (the bit-exact path, everything above).
// PREC = 1, 2: split precision on the bf16 matrix cores, opt-in (tmat_set_precision), f32 accumulation:
//   1 "bf16x3": x = hi + lo (hi = rne_bf16(x), lo = rne_bf16(x - hi)); a w ~ a_lo w_hi + a_hi w_lo + a_hi w_hi: three
//     v_mfma_f32_32x32x16_bf16 do the work of eight f32 MFMAs at half the cycles each.  About 2^-16 relative error per product.
//   2 "bf16x6": x = hi + mid + lo (24 bits: the whole f32 mantissa); the six products of total order <= 2
//     (lo hi, hi lo, mid mid, mid hi, hi mid, hi hi): dropped terms are 2^-24 relative, i.e. f32-level error, at 3/8 of the
//     matrix-pipe time of the f32 path.
//   Neither is bit-exact with the oracle: tests/test_gpu_alt_precision.py gates them by the tolerance of BASELINE.json's
//   north_star (counts equal, lengths within 1e-4) and bench.py reports them as a separate "alt" block.
//   Activations stay f32 in HBM and in LDS and are split in registers after the fragment read (the vector ALU and the bf16
//   matrix pipe run side by side, unlike the f32 MFMA).  The WEIGHTS are split once on the host (tmat_api.cpp:split_bf16)
//   into NPL = 2 / 3 bf16 PLANES [plane][tap][Cout][Cin]; a stage holds the A rows as in the f32 path and NPL weight planes
//   of [BN rows][32 bf16] (64-byte rows, 16-byte unit u of row r in slot u ^ ((r >> 2) & 3): conflict-free ds_read_b128).
#ifdef TMAT_DIAG
__device__ long long conv_diag[2048 * 8 * 8];      // [workgroup < 2048][wave][work, dma wait, barrier, fill, chunks, 1, epilogue]
#endif
template <int BM, int BN, int WM, int WN, int KS, bool RELU, int PREC = 0>
#ifndef TMAT_CONV_WPS
#define TMAT_CONV_WPS 4     // waves per SIMD the 8-wave conv kernel is compiled for (VGPR budget 512 / this)
#endif
#ifndef TMAT_CONV64_WPS
#define TMAT_CONV64_WPS TMAT_CONV_WPS     // the same for the 8-wave 128 x 64 tile of the Cout = 64 layers (3 workgroups per CU need 6)
#endif
#ifndef TMAT_CONV4_WPS
#define TMAT_CONV4_WPS 2    // 4-wave workgroups
#endif
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 16 ? 4 : (BM + BN) * 256 > 65536 ? 2 : WM * WN == 8 ? (BM == 128 && BN == 64 ? TMAT_CONV64_WPS : TMAT_CONV_WPS) : TMAT_CONV4_WPS)) void conv_mfma_kernel(ConvArgs a, int M, int Ho, int Wo, int nMt, int nNt, FastDiv dHW, FastDiv dW, int nstat)
{
    // KS (1, 2 or 3) is a template parameter so that the 3x3 and the pointwise instantiations are distinct kernels
    // (distinct names in rocprofv3 traces: the 3x3 <128,128,2,2,3,*> instantiations are the dominant kernel).
    // KS == 2 is the sub-pixel form of a 3x3 convolution over a 2x nearest-upsampled input: output pixel
    // (2i + py, 2j + px) only sees the 2x2 stored pixels (i + py - 1 + {0,1}, j + px - 1 + {0,1}), with the 3x3 taps that
    // fall on the same stored pixel pre-summed per parity class (weights [class][tap][Cout][Cin], class = blockIdx.y,
    // M enumerates the stored pixels).  4/9 of the multiply-adds of the as-written form.
    static_assert(WM * WN == 4 || WM * WN == 8 || WM * WN == 16, "4, 8 or 16 waves");
    constexpr int NT = 64 * WM * WN;                 // threads
    constexpr int RP = NT / 8;                       // rows per DMA pass: NT lanes x 16 B = RP rows of 128 B
    constexpr int KC = 32;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int NPA = BM / RP, NPB = BN / RP;      // DMA passes
    constexpr int NPL = PREC == 0 ? 0 : PREC + 1;                            // bf16 weight planes of the split-precision forms
    constexpr int NPLA = NPL > 0 ? NPL : 1;                                  // array extent (the f32 instantiation never runs that code)
    constexpr int STAGE = PREC == 0 ? (BM + BN) * KC : BM * KC + NPL * BN * 16;   // floats per stage: A rows, then B rows / B planes
    static_assert(NPA >= 1 && NPA <= 8 && NPA * RP == BM, "A passes");
    static_assert(NPB >= 1 && NPB <= 4 && NPB * RP == BN, "B passes");
    static_assert(TM >= 1 && TN >= 1, "wave tile");
    // Two separate LDS objects, one per stage: hipcc can then tell that the DMA in flight into one stage does not alias
    // the ds_reads of the other (with a single array it waits vmcnt(0) before the first ds_read of every chunk, which
    // serialises the DMA with the MFMAs).
    __shared__ __attribute__((aligned(16))) float stage0[STAGE];
    __shared__ __attribute__((aligned(16))) float stage1[STAGE];

    // XCD-aware tile mapping: blocks b and b+8 share an XCD (round-robin dispatch); give the
    // nNt column tiles of one pixel tile to the same XCD so its L2 serves the re-read A pixels.
    const int b = blockIdx.x;
    const int xcd = b & 7, j = b >> 3;
    // Two mappings.  Default: the nNt column tiles of one pixel tile go to the same XCD (its L2 serves the re-read A pixels; every XCD
    // streams ALL weights).  N-stationary (nstat, chosen by the host when the layer's weights exceed what an XCD's 4 MB L2 keeps, i.e.
    // the 512 -> 512 3x3 layers: 9.4 MB): XCD x keeps column tile x % nNt -- 2.4 MB of weights, L2-resident -- and shares the pixel
    // tiles with the 8 / nNt - 1 other XCDs of its group; A is then fetched nNt times from the fabric.  Measured (FETCH_SIZE as
    // read, per launch of 1600 patches, 1.31 GB of input): 512 -> 512 at 20 x 20: 15.7 GB default (6.5 in another run: the weight
    // stream thrashes the L2 differently from run to run), 5.0 GB N-stationary; 256 -> 256 at 40 x 40 (2.4 MB of weights): 3.9 GB
    // default, 5.3 GB N-stationary.  The launch times do not differ (the pipeline hides the latency either way).
    const int nt = nstat ? xcd % nNt : j % nNt;
    const int mt = nstat ? j * (8 / nNt) + xcd / nNt : (j / nNt) * 8 + xcd;
    if (mt >= nMt) return;
    const int m0 = mt * BM, n0 = nt * BN;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int spy = KS == 2 ? (int)(blockIdx.y >> 1) : 0, spx = KS == 2 ? (int)(blockIdx.y & 1) : 0;
    constexpr int taps = KS * KS;
    const int cchunks = a.Cin / KC;
    const int nchunks = taps * cchunks;

    // DMA role of this lane: row (t >> 3) of every RP-row pass, slot t & 7, i.e. channel group (t & 7) ^ ((row >> 1) & 7).
    // The DMA is a raw buffer load (buffer_load_dwordx4 ... lds): address = buffer base + per-lane voffset + scalar
    // soffset.  The A buffer starts (w + 1) stored pixels before the tile's first pixel, the per-lane voffset is the
    // pixel's distance from that first pixel (fixed for the whole kernel), and the tap / channel-block displacement
    // ((dy + 1) w + dx + 1) Cin + 32 cb is the scalar soffset: no per-chunk 64-bit address math.  Out-of-image taps (and
    // rows past M) use a voffset beyond num_records: the buffer range check makes the load return zeros.
    const int srow = t >> 3;
    const int c4 = ((t & 7) ^ ((srow >> 1) & 7)) * 4;
    constexpr unsigned OOB = 0x80000000u;
    // pointwise, stride 1: the stored pixel of output m is m itself -- no index arithmetic at all (these layers have
    // K = Cin as small as 64, i.e. two chunks per tile, so a division-heavy prologue would show)
    const bool flat = KS == 1 && a.stride == 1;
    auto stored_pixel = [&](int m) {     // linear index of the stored pixel that output pixel m is centred on
        const int n = fdiv(m, dHW);
        const int r = m - n * (Ho * Wo);
        const int yo = fdiv(r, dW);
        return (n * a.h + yo * a.stride) * a.w + (r - yo * Wo) * a.stride;
    };
    const int p0 = flat ? m0 : __builtin_amdgcn_readfirstlane(stored_pixel(m0));
    // Row-uniform form (round 4): a wave stages 8 consecutive output pixels per pass (rows 8 wave .. 8 wave + 7 of the pass), and with
    // Wo % 8 == 0 those lie in ONE image row: (n, y) and the first column are wave-uniform, so the pixel -> (n, y, x) conversion, the row
    // masks and the "past M" test run on the SCALAR unit (s_mul_hi_u32), and a lane only adds its column.  The generic form below costs
    // ~35 vector instructions per pass, a dozen of them quarter-rate (v_mul_hi / v_mul_lo / v_mad_u64), and every vector instruction of an
    // f32-MFMA kernel is matrix time: with K = 576 (the Cout = 64 layers) the prologue was 3 % of a tile.
    const bool rowfast = !flat && (Wo & 7) == 0;
    unsigned pv[NPA];       // voffset (bytes) of the staged pixel's channel group
    unsigned pm[NPA];       // generic form: mask of the taps that fall inside the image
    unsigned pvt[taps][NPA];
    if (rowfast) {
        // the lane's share of x and of the offset, made once (opaque to the optimiser, which otherwise folds them back into per-pass multiplies)
        int lx = (lane >> 3) * a.stride;
        unsigned lane_pv = (unsigned)(lx * a.Cin + c4) * 4u;
        asm volatile("" : "+v"(lx), "+v"(lane_pv));
#pragma unroll
        for (int i = 0; i < NPA; i++) {
            const int mb = m0 + i * RP + wave * 8;           // uniform; M % 8 == 0, so the 8 pixels are inside M or past it together
            const bool ok = mb < M;
            const int mm = ok ? mb : m0;
            const int n = fdiv(mm, dHW);
            const int r = mm - n * (Ho * Wo);
            const int yo = fdiv(r, dW);
            const int y = yo * a.stride, xs = (r - yo * Wo) * a.stride;
            const int x = xs + lx;
            pv[i] = (unsigned)(((n * a.h + y) * a.w + xs - p0) * a.Cin) * 4u + lane_pv;
            if (KS == 3) {
                const unsigned pl = x > 0 ? pv[i] : OOB, pr = x + 1 < a.w ? pv[i] : OOB;
                const bool y0 = ok && y > 0, y2 = ok && y + 1 < a.h;
                pvt[0][i] = y0 ? pl : OOB; pvt[1][i] = y0 ? pv[i] : OOB; pvt[2][i] = y0 ? pr : OOB;
                pvt[3 % taps][i] = ok ? pl : OOB; pvt[4 % taps][i] = ok ? pv[i] : OOB; pvt[5 % taps][i] = ok ? pr : OOB;
                pvt[6 % taps][i] = y2 ? pl : OOB; pvt[7 % taps][i] = y2 ? pv[i] : OOB; pvt[8 % taps][i] = y2 ? pr : OOB;
            } else if (KS == 2) {       // tap tp: row y + spy - 1 + (tp >> 1), column x + spx - 1 + (tp & 1)
                const unsigned pl = x + spx > 0 ? pv[i] : OOB, pr = x + spx < a.w ? pv[i] : OOB;
                const bool y0 = ok && y + spy > 0, y1 = ok && y + spy < a.h;
                pvt[0][i] = y0 ? pl : OOB; pvt[1 % taps][i] = y0 ? pr : OOB;
                pvt[2 % taps][i] = y1 ? pl : OOB; pvt[3 % taps][i] = y1 ? pr : OOB;
            } else {
                pvt[0][i] = ok ? pv[i] : OOB;
            }
        }
    } else {
#pragma unroll
    for (int i = 0; i < NPA; i++) {
        const int m = m0 + i * RP + srow;
        const bool ok = m < M;
        if (flat) {
            pv[i] = (unsigned)((i * RP + srow) * a.Cin + c4) * 4u;
            pm[i] = ok ? 1u : 0u;
            continue;
        }
        const int mm = ok ? m : m0;
        const int n = fdiv(mm, dHW);
        const int r = mm - n * (Ho * Wo);
        const int yo = fdiv(r, dW);
        const int y = yo * a.stride, x = (r - yo * Wo) * a.stride;
        pv[i] = (unsigned)(((n * a.h + y) * a.w + x - p0) * a.Cin + c4) * 4u;
        // tap masks without branches: bit ky * 3 + kx is set when row y + ky - 1 and column x + kx - 1 lie inside the image
        unsigned msk = 1u;
        if (KS == 3) {
            const unsigned ym = (y > 0 ? 0x007u : 0u) | 0x038u | (y + 1 < a.h ? 0x1C0u : 0u);
            const unsigned xm = (x > 0 ? 0x049u : 0u) | 0x092u | (x + 1 < a.w ? 0x124u : 0u);
            msk = ym & xm;
        } else if (KS == 2) {       // tap tp: row y + spy - 1 + (tp >> 1), column x + spx - 1 + (tp & 1)
            const unsigned ym = (y + spy > 0 ? 0x3u : 0u) | (y + spy < a.h ? 0xCu : 0u);
            const unsigned xm = (x + spx > 0 ? 0x5u : 0u) | (x + spx < a.w ? 0xAu : 0u);
            msk = ym & xm;
        }
        if (!ok) msk = 0u;
        pm[i] = msk;
    }
    // Per-tap scalars and per-(tap, pass) lane offsets, made ONCE per tile (round 4).  Rounds 1-3 derived them per chunk from a runtime tap
    // counter: ~25 scalar instructions (tap -> (dy, dx) -> offsets, wrap) and, per A pass, v_and / v_cmp / v_cndmask to pick the lane's offset
    // or the out-of-range constant -- 6 vector instructions per chunk next to 16-32 MFMAs, and vector instructions do not issue beside f32
    // MFMAs.  The K loop below is unrolled over the taps (18 chunks for 3 x 3: two channel blocks, so that the stage parity
    // repeats; 4 for the sub-pixel form; 2 for 1 x 1), which makes the tap of every DMA a compile-time constant: its scalar offsets are
    // two s_add, its lane offsets live in registers (taps x NPA <= 18; the kernels use 58-86 of their 128).
#pragma unroll
    for (int tp = 0; tp < taps; tp++)
#pragma unroll
        for (int i = 0; i < NPA; i++) pvt[tp][i] = ((pm[i] >> tp) & 1u) ? pv[i] : OOB;
    }
    const __amdgpu_buffer_rsrc_t rsA =
        __builtin_amdgcn_make_buffer_rsrc((void *)(a.in + ((long)p0 - a.w - 1) * a.Cin), 0, 0x7fffffff, 0x00020000);
    // weights [tap][Cout][Cin] (KS == 2: [class][tap][Cout][Cin]): row n0 + pass * RP + srow, this lane's channel group
    const size_t wofs = (KS == 2 ? (size_t)blockIdx.y * 4 * a.Cin * a.Cout : (size_t)0) + (size_t)n0 * a.Cin;      // elements
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(
        PREC == 0 ? (void *)(a.W + wofs) : (void *)((const char *)a.W + wofs * 2), 0, 0x7fffffff, 0x00020000);
    const unsigned wv = (unsigned)(srow * a.Cin + c4) * 4u;
    // split-precision planes: lane t loads the 16-byte unit ((t & 3) ^ ((row >> 2) & 3)) of row t >> 2 (64 bytes of bf16 per row and chunk);
    // the buffer base of a plane row is in bf16 elements what rsB's is in floats, so the same descriptor serves with halved offsets
    const unsigned wvp = (unsigned)((t >> 2) * a.Cin * 2 + (((t & 3) ^ ((t >> 4) & 3)) * 16));
    const int plane_bytes = (KS == 2 ? 16 : taps) * a.Cout * a.Cin * 2;
    const int wtap = a.Cout * a.Cin * 4;             // bytes per tap
    const int wpass = RP * a.Cin * 4;                // bytes per DMA pass of weight rows
    const int ldsw = wave * 8 * KC;                  // this wave's 8 rows (1 KiB) inside an RP-row pass

    // (pvt[tap][pass], the per-(tap, pass) lane offsets of the A DMA, were made above, once per tile)
    // scalar offsets of the chunk to load next, advanced by compile-time-selected steps (tap -> next tap in the row, next row, wrap to the
    // next channel block): A: ((dy + 1) w + dx + 1) Cin + 32 cb floats from the descriptor base; B: tap * wtap + 32 cb
    const int a_cstep = a.Cin * 4;                                                   // next tap of the same row
    const int a_rstep = (a.w - (KS == 3 ? 2 : 1)) * a.Cin * 4;                        // first tap of the next row
    const int a_wrap = KC * 4 - (KS == 3 ? 2 * a.w + 2 : KS == 2 ? a.w + 1 : 0) * a.Cin * 4;      // last tap -> tap 0 of the next channel block
    const int b_wrap = KC * 4 - (taps - 1) * wtap;
    int soA = (KS == 3 ? 0 : KS == 2 ? spy * a.w + spx : a.w + 1) * a.Cin * 4, soB = 0, soBp = 0;
#ifdef TMAT_ABL_A9        // timing ablation (wrong results): the A tile is fetched for one tap per channel block only -- what a halo tile could save at most
#define TMAT_ABL_A9_COND(T_) ((T_) == 0)
#else
#define TMAT_ABL_A9_COND(T_) true
#endif
#define TMAT_DMA_A(i, T_)                                                                                    \
    if (i < NPA) {        /* (the offset through a local: with pvt[T_][i] as the argument itself, the HOST pass silently drops the kernel's stub) */ \
        const unsigned vo_ = pvt[T_][i];                                                                     \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_void_t *)(st + i * RP * KC + ldsw), 16, vo_, soA, 0, 0); \
    }
#define TMAT_DMA_B(i)                                                                                        \
    if (PREC == 0 && i < NPB)                                                                                \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_t *)(st + (BM + i * RP) * KC + ldsw), 16, wv, soB + i * wpass, 0, 0); \
    if (PREC != 0 && i < NPL && wave * 16 < BN)      /* plane i: this wave's 16 rows of 64 bytes */          \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_void_t *)(st + BM * KC + i * BN * 16 + wave * 256), 16, wvp, soBp + i * plane_bytes, 0, 0);
// T_: the tap of the chunk being loaded -- a compile-time constant at every use (the K loop is unrolled over the taps)
#define TMAT_ISSUE_CHUNK(stage_, T_)                                                   \
    {                                                                                  \
        float *st = (stage_);                                                          \
        if (TMAT_ABL_A9_COND(T_)) {                                                    \
        TMAT_DMA_A(0, T_) TMAT_DMA_A(1, T_) TMAT_DMA_A(2, T_) TMAT_DMA_A(3, T_)        \
        TMAT_DMA_A(4, T_) TMAT_DMA_A(5, T_) TMAT_DMA_A(6, T_) TMAT_DMA_A(7, T_)        \
        }                                                                              \
        TMAT_DMA_B(0) TMAT_DMA_B(1) TMAT_DMA_B(2) TMAT_DMA_B(3)                        \
        if ((T_) == taps - 1) { soA += a_wrap; soB += b_wrap; soBp += (b_wrap >> 1); } \
        else {                                                                         \
            soA += (KS == 3 ? (T_) % 3 == 2 : KS == 2 ? ((T_) & 1) == 1 : false) ? a_rstep : a_cstep; \
            soB += wtap; soBp += (wtap >> 1);                                          \
        }                                                                              \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; i++)
#pragma unroll
        for (int jn = 0; jn < TN; jn++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][jn][r] = 0.f;

    // fragment addressing: row = tile base + (lane & 31) (tile bases are multiples of 32, so the swizzle key is (lane >> 1) & 7),
    // slot of channel group 2g + (lane >> 5) is (2g + (lane >> 5)) ^ key
    const int arow = (wm * (BM / WM) + (lane & 31)) * KC;
    const int brow = (BM + wn * (BN / WN) + (lane & 31)) * KC;
    const int hi = lane >> 5, key = (lane >> 1) & 7;

    // ReLU on a fragment register: as a signed integer a negative float (and -0.0) is negative, so one v_max_i32 with 0
    // gives relu(x) = x > 0 ? x : +0.0 exactly (no NaNs here), without fmaxf's canonicalisation.
#define TMAT_RELU(v) __builtin_bit_cast(float, max(__builtin_bit_cast(int, (v)), 0))
    // One step = one K chunk: fragment reads, the DMA of the next chunk into the other stage and the chunk's MFMAs in the order given
    // at TMAT_ORDER below, then the wave's vmcnt(0) and the barrier.  sched_barrier(0) pins that order: left alone, hipcc sinks the
    // reads next to their MFMAs and hoists the barrier.
#if defined(TMAT_VAR_SETPRIO)
#define TMAT_PRIO(x) __builtin_amdgcn_s_setprio(x);
#else
#define TMAT_PRIO(x)
#endif
#if defined(TMAT_VAR_NOPIN)
#define TMAT_PIN()
#else
#define TMAT_PIN() __builtin_amdgcn_sched_barrier(0);
#endif
#ifdef TMAT_ABL_NODMA
#define TMAT_LOOP_ISSUE(st, T_)
#else
#define TMAT_LOOP_ISSUE(st, T_) TMAT_ISSUE_CHUNK(st, T_)
#endif
#ifdef TMAT_ABL_NOBAR
#define TMAT_LOOP_SYNC()
#else
// every wave retires its own DMA (explicitly: the ordering of LDS-DMA data for the readers is this wait followed by the
// barrier, and must not depend on what hipcc chooses to put in front of a barrier), then the workgroup barrier
#ifdef TMAT_DIAG      // diagnostic build: cycles a wave spends in the chunk's DMA wait, in the barrier, and in the rest of a step
#define TMAT_LOOP_SYNC() { const long long d0_ = (long long)__builtin_readcyclecounter(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      \
                           const long long d1_ = (long long)__builtin_readcyclecounter(); __syncthreads();                                       \
                           const long long d2_ = (long long)__builtin_readcyclecounter(); dg_work += d0_ - dg_last; dg_vm += d1_ - d0_; dg_bar += d2_ - d1_; dg_last = d2_; }
#else
#define TMAT_LOOP_SYNC() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
#endif
#endif
#define TMAT_READ_FRAGS(stage_) TMAT_READ_FRAGS_RANGE(stage_, 0, 4)
#define TMAT_READ_FRAGS_RANGE(stage_, G0, G1)                                          \
        _Pragma("unroll") for (int g = (G0); g < (G1); g++) {                          \
            const int slot = ((2 * g + hi) ^ key) * 4;                                 \
            _Pragma("unroll") for (int i = 0; i < TM; i++) av[g][i] = *reinterpret_cast<const float4 *>((stage_) + arow + i * 32 * KC + slot); \
            _Pragma("unroll") for (int jn = 0; jn < TN; jn++) bv[g][jn] = *reinterpret_cast<const float4 *>((stage_) + brow + jn * 32 * KC + slot); \
        }
#define TMAT_MFMAS() TMAT_MFMAS_RANGE(0, 4)
#define TMAT_MFMAS_RANGE(G0, G1)                                                       \
        _Pragma("unroll") for (int g = (G0); g < (G1); g++) {                          \
            if (RELU) {                                                                \
                _Pragma("unroll") for (int i = 0; i < TM; i++) {                       \
                    av[g][i].x = TMAT_RELU(av[g][i].x); av[g][i].y = TMAT_RELU(av[g][i].y); \
                    av[g][i].z = TMAT_RELU(av[g][i].z); av[g][i].w = TMAT_RELU(av[g][i].w); \
                }                                                                      \
            }                                                                          \
            _Pragma("unroll") for (int i = 0; i < TM; i++)                             \
                _Pragma("unroll") for (int jn = 0; jn < TN; jn++) {                    \
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g][i].x, bv[g][jn].x, acc[i][jn], 0, 0, 0); \
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g][i].y, bv[g][jn].y, acc[i][jn], 0, 0, 0); \
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g][i].z, bv[g][jn].z, acc[i][jn], 0, 0, 0); \
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g][i].w, bv[g][jn].w, acc[i][jn], 0, 0, 0); \
                }                                                                      \
        }
// Order of one step (R = the fragment reads of a k group: TM + TN ds_read_b128, M = its 8 TM TN MFMAs, D = the LDS-DMA of the next chunk):
//     R0 M0 D R1 M1 R2 M2 R3 M3      then the wave's vmcnt(0) and the workgroup barrier            (round 4)
//     R0 R1 M0 R2 D M1 R3 M2 M3      was round 3's order (TMAT_VAR_ORDER=1).  With the K loop unrolled over the taps the DMA issue is four
//     s_mov m0 / buffer_load pairs and nothing else, and reads one group ahead are enough: measured per pass of 1600 patches (network
//     kernels) 241.1 (round-3 order) / 239.8 (R0 M0 R1 D M1 ..) / 239.2 (this) / 239.3 (R0 D M0 R1 ..) / 240.2 (D behind M1) / 240.1 (D first).
// After a barrier all eight waves of a workgroup stand at the same instruction, so whatever comes first is what the matrix pipe waits
// for unless the CU's other workgroup happens to be in its MFMA stretch.  Round 1-2 issued all sixteen reads, then the DMA (4 instructions
// + ~30 scalar ones per wave), then the 32 MFMAs: the first MFMA of the LAST wave waited for 8 x 16 KB of reads to drain through the LDS
// pipe and for its own DMA issue.  Reads two groups ahead and the DMA behind the first group (measured per pass of 1600 patches):
// 255.3 ms (old order) -> 252.1 (DMA behind M0) -> 248.2 (split reads); the orders R0 R1 M0 D R2 ..., R0 M0 R1 D M1 R2 ... measure the
// same, the DMA behind M1 or later is slower (it lands late for the closing wait).  Bit-exact: the MFMA order is unchanged.
// (TMAT_ABL_* / TMAT_VAR_*: timing experiments of tools/gpu_variants.sh; ablations produce wrong results and nothing of this is defined
// in the shipped build.)
#define R_(c_, g) TMAT_READ_FRAGS_RANGE(c_, g, g + 1) TMAT_PIN()
#define M_(g) TMAT_MFMAS_RANGE(g, g + 1) TMAT_PIN()
#define D_(n_, m_, T_) if (m_) TMAT_LOOP_ISSUE(n_, T_) TMAT_PIN()
#if defined(TMAT_VAR_ORDER) && TMAT_VAR_ORDER == 0       // the round-2 order
#define TMAT_ORDER(c_, n_, m_, T_) R_(c_, 0) R_(c_, 1) R_(c_, 2) R_(c_, 3) D_(n_, m_, T_) M_(0) M_(1) M_(2) M_(3)
#elif defined(TMAT_VAR_ORDER) && TMAT_VAR_ORDER == 2
#define TMAT_ORDER(c_, n_, m_, T_) R_(c_, 0) R_(c_, 1) M_(0) D_(n_, m_, T_) R_(c_, 2) M_(1) R_(c_, 3) M_(2) M_(3)
#elif defined(TMAT_VAR_ORDER) && TMAT_VAR_ORDER == 3
#define TMAT_ORDER(c_, n_, m_, T_) R_(c_, 0) M_(0) R_(c_, 1) D_(n_, m_, T_) M_(1) R_(c_, 2) M_(2) R_(c_, 3) M_(3)
#elif defined(TMAT_VAR_ORDER) && TMAT_VAR_ORDER == 1       // the round-3 order
#define TMAT_ORDER(c_, n_, m_, T_) R_(c_, 0) R_(c_, 1) M_(0) R_(c_, 2) D_(n_, m_, T_) M_(1) R_(c_, 3) M_(2) M_(3)
#elif defined(TMAT_VAR_ORDER) && TMAT_VAR_ORDER == 5
#define TMAT_ORDER(c_, n_, m_, T_) R_(c_, 0) R_(c_, 1) M_(0) D_(n_, m_, T_) M_(1) R_(c_, 2) M_(2) R_(c_, 3) M_(3)
#elif defined(TMAT_VAR_ORDER) && TMAT_VAR_ORDER == 6
#define TMAT_ORDER(c_, n_, m_, T_) R_(c_, 0) D_(n_, m_, T_) M_(0) R_(c_, 1) M_(1) R_(c_, 2) M_(2) R_(c_, 3) M_(3)
#elif defined(TMAT_VAR_ORDER) && TMAT_VAR_ORDER == 7
#define TMAT_ORDER(c_, n_, m_, T_) R_(c_, 0) M_(0) R_(c_, 1) M_(1) D_(n_, m_, T_) R_(c_, 2) M_(2) R_(c_, 3) M_(3)
#elif defined(TMAT_VAR_ORDER) && TMAT_VAR_ORDER == 8
#define TMAT_ORDER(c_, n_, m_, T_) D_(n_, m_, T_) R_(c_, 0) M_(0) R_(c_, 1) M_(1) R_(c_, 2) M_(2) R_(c_, 3) M_(3)
#else
#define TMAT_ORDER(c_, n_, m_, T_) R_(c_, 0) M_(0) D_(n_, m_, T_) R_(c_, 1) M_(1) R_(c_, 2) M_(2) R_(c_, 3) M_(3)
#endif
#define TMAT_STEP(cur, nxt, more, T_)                                                  \
    {                                                                                  \
        float4 av[4][TM], bv[4][TN];                                                   \
        TMAT_ORDER(cur, nxt, more, T_)                                                 \
        TMAT_LOOP_SYNC()                                                               \
    }

    // split-precision step: k step tk (0, 1) of the chunk covers channels 16 tk .. 16 tk + 15; a lane's 8 values are channels
    // 16 tk + 8 h + j (A: 16-byte units 4 tk + 2 h, 4 tk + 2 h + 1 of its f32 row; B: unit 2 tk + h of its row in every plane)
    const int browp = BM * KC + (wn * (BN / WN) + (lane & 31)) * 16;
    const int keyb = (lane >> 2) & 3;
#define TMAT_STEP_BF16(cur, nxt, more, T_)                                             \
    {                                                                                  \
        float4 af[2][TM][2];                                                           \
        _Pragma("unroll") for (int tk = 0; tk < 2; tk++) {                             \
            const int sa = ((4 * tk + 2 * hi) ^ key) * 4, sb = ((4 * tk + 2 * hi + 1) ^ key) * 4; \
            _Pragma("unroll") for (int i = 0; i < TM; i++) {                           \
                af[tk][i][0] = *reinterpret_cast<const float4 *>((cur) + arow + i * 32 * KC + sa); \
                af[tk][i][1] = *reinterpret_cast<const float4 *>((cur) + arow + i * 32 * KC + sb); \
            }                                                                          \
        }                                                                              \
        bf16x8 bp[2][NPLA][TN];                                                        \
        /* three planes: the second k step's weight fragments are read behind the first one's MFMAs (128-VGPR budget) */ \
        _Pragma("unroll") for (int tk = 0; tk < (NPL == 3 ? 1 : 2); tk++)              \
            _Pragma("unroll") for (int pl = 0; pl < NPL; pl++)                         \
                _Pragma("unroll") for (int jn = 0; jn < TN; jn++)                      \
                    bp[tk][pl][jn] = *reinterpret_cast<const bf16x8 *>((cur) + browp + pl * BN * 16 + jn * 32 * 16 + (((2 * tk + hi) ^ keyb) * 4)); \
        TMAT_PIN()                                                                     \
        /* the next chunk's DMA: six products -- behind the first k step's MFMAs, as in the f32 step (209.8 -> 204.3 ms per pass); three */ \
        /* products -- in front (a chunk is only 384 cycles of matrix time per wave: issued later, the DMA lands late: +4 ms per pass) */ \
        if (NPL == 2) { if (more) TMAT_LOOP_ISSUE(nxt, T_) }                           \
        TMAT_PIN()                                                                     \
        _Pragma("unroll") for (int tk = 0; tk < 2; tk++) {                             \
            if (NPL == 3 && tk == 1) {                                                 \
                TMAT_PIN()                                                             \
                if (more) TMAT_LOOP_ISSUE(nxt, T_)                                     \
                TMAT_PIN()                                                             \
            }                                                                          \
            if (NPL == 3 && tk == 1) {                                                 \
                _Pragma("unroll") for (int pl = 0; pl < NPL; pl++)                     \
                    _Pragma("unroll") for (int jn = 0; jn < TN; jn++)                  \
                        bp[1][pl][jn] = *reinterpret_cast<const bf16x8 *>((cur) + browp + pl * BN * 16 + jn * 32 * 16 + (((2 + hi) ^ keyb) * 4)); \
            }                                                                          \
            bf16x8 ap[NPLA][TM];                                                       \
            _Pragma("unroll") for (int i = 0; i < TM; i++) {                           \
                float xs[8] = {af[tk][i][0].x, af[tk][i][0].y, af[tk][i][0].z, af[tk][i][0].w, \
                               af[tk][i][1].x, af[tk][i][1].y, af[tk][i][1].z, af[tk][i][1].w}; \
                _Pragma("unroll") for (int j = 0; j < 8; j++) {                        \
                    float rem = RELU ? TMAT_RELU(xs[j]) : xs[j];                       \
                    _Pragma("unroll") for (int pl = 0; pl < NPL; pl++) {               \
                        const __bf16 q = (__bf16)rem;                                  \
                        ap[pl][i][j] = q;                                              \
                        rem = rem - (float)q;                                          \
                    }                                                                  \
                }                                                                      \
            }                                                                          \
            _Pragma("unroll") for (int i = 0; i < TM; i++)                             \
                _Pragma("unroll") for (int jn = 0; jn < TN; jn++) {                    \
                    /* smallest products first; NPL == 2: lo hi, hi lo, hi hi; NPL == 3: lo hi, hi lo, mid mid, mid hi, hi mid, hi hi */ \
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[NPLA - 1][i], bp[tk][0][jn], acc[i][jn], 0, 0, 0); \
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0][i], bp[tk][NPLA - 1][jn], acc[i][jn], 0, 0, 0); \
                    if (NPL == 3) {                                                    \
                        constexpr int MID = NPL == 3 ? 1 : 0;                          \
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[MID][i], bp[tk][MID][jn], acc[i][jn], 0, 0, 0); \
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[MID][i], bp[tk][0][jn], acc[i][jn], 0, 0, 0); \
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0][i], bp[tk][MID][jn], acc[i][jn], 0, 0, 0); \
                    }                                                                  \
                    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[0][i], bp[tk][0][jn], acc[i][jn], 0, 0, 0); \
                }                                                                      \
        }                                                                              \
        TMAT_PIN()                                                                     \
        TMAT_LOOP_SYNC()                                                               \
    }

#ifdef TMAT_DIAG
    long long dg_work = 0, dg_vm = 0, dg_bar = 0;
    const long long dg_t0 = (long long)__builtin_readcyclecounter();
#endif
    TMAT_ISSUE_CHUNK(stage0, 0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef TMAT_DIAG
    long long dg_last = (long long)__builtin_readcyclecounter();
    const long long dg_fill = dg_last - dg_t0;
#endif

    // even chunks live in stage0, odd ones in stage1.  The loop body is UNR chunks, unrolled: UNR is the smallest even multiple of the tap
    // count (nchunks = taps x Cin / 32 is a multiple of it: host check), so chunk c + u has tap u % taps and stage u & 1, and the DMA issued
    // in its step is that of chunk c + u + 1: tap (u + 1) % taps, at compile time.
    constexpr int UNR = (taps & 1) ? 2 * taps : taps;
    if (PREC == 0) {
        for (int c = 0; c < nchunks; c += UNR) {
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                const bool more = u + 1 < UNR || c + UNR < nchunks;
                if (u & 1) TMAT_STEP(stage1, stage0, more, (u + 1) % taps)
                else TMAT_STEP(stage0, stage1, more, (u + 1) % taps)
            }
        }
    } else {
        for (int c = 0; c < nchunks; c += UNR) {
#pragma unroll
            for (int u = 0; u < UNR; u++) {
                const bool more = u + 1 < UNR || c + UNR < nchunks;
                if (u & 1) TMAT_STEP_BF16(stage1, stage0, more, (u + 1) % taps)
                else TMAT_STEP_BF16(stage0, stage1, more, (u + 1) % taps)
            }
        }
    }
#undef TMAT_STEP_BF16
#undef TMAT_STEP
#undef TMAT_ORDER
#undef R_
#undef M_
#undef D_
#undef TMAT_READ_FRAGS
#undef TMAT_READ_FRAGS_RANGE
#undef TMAT_MFMAS
#undef TMAT_MFMAS_RANGE
#undef TMAT_PIN
#undef TMAT_PRIO
#undef TMAT_LOOP_ISSUE
#undef TMAT_LOOP_SYNC
#undef TMAT_RELU
#undef TMAT_DMA_A
#undef TMAT_DMA_B
#undef TMAT_ISSUE_CHUNK

#ifdef TMAT_DIAG
    long long *dg_slot = conv_diag + ((size_t)(blockIdx.x < 2048 ? blockIdx.x : 0) * 8 + wave) * 8;
    const bool dg_on = lane == 0 && KS == 3 && BN == 128 && !RELU && PREC == 0 && blockIdx.x < 2048;
    if (dg_on) { dg_slot[0] = dg_work; dg_slot[1] = dg_vm; dg_slot[2] = dg_bar; dg_slot[3] = dg_fill; dg_slot[4] = nchunks; dg_slot[5] = 1; }
    const long long dg_e0 = (long long)__builtin_readcyclecounter();
#endif
#ifdef TMAT_ABL_NOEPI      // timing ablation only (results are wrong): one store per lane instead of the epilogue
    { float sacc = 0.f;
      for (int i = 0; i < TM; i++) for (int jn = 0; jn < TN; jn++) for (int r = 0; r < 16; r++) sacc += acc[i][jn][r];
      if (m0 + t < M) a.out[(size_t)(m0 + t) * a.Cout + n0] = sacc;
      return; }
#endif
    // epilogue, wave-private: every wave takes ITS OWN (32 TM) x (32 TN) block of outputs through a private slab of the two stages (all
    // fragment reads of the last chunk are behind the loop's closing barrier, no DMA is in flight) -- accumulators in (C/D layout: col =
    // lane & 31 the output channel, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) the pixel), float4 rows out, BN fold / bias, residual, ReLU,
    // 16-byte stores.  No workgroup barrier: LDS operations of one wave execute in order, so a wave whose MFMAs have drained stores while
    // the others still compute (the shared [128][BN] tile of rounds 1-2 needed two to three barriers per tile; measured difference small:
    // 249.4 -> 248.2 ms per pass).
    {
        constexpr int WROWS = TM * 32, WCOLS = TN * 32;      // the wave's block
        constexpr int V4W = WCOLS / 4;                       // float4 per row
        constexpr int RPW = 64 / V4W;                        // rows per store iteration of one wave
        constexpr int PW = WROWS * WCOLS;
        constexpr int WPS = (WM * WN) / 2;                   // waves per stage
        static_assert(PW * WPS <= STAGE, "wave-private epilogue slabs fit the two stages");
        float *Ws = (wave < WPS ? stage0 : stage1) + (wave % WPS) * PW;
        constexpr int NIT = WROWS / RPW;                     // store iterations: RPW consecutive output pixels each
        const int quad = lane % V4W, rsub = lane / V4W;
        const int cbase = n0 + wn * (BN / WN);
        const int co = cbase + quad * 4;
        // no scale (plain bias): fmaf(v, 1, shift) is v + shift bit for bit (the product is exact, one rounding), so the loop below has ONE
        // form -- with `a.scale ? fmaf : add` hipcc evaluated both and selected per value (2 v_pk_add + 4 v_cndmask per 16-byte row, and
        // every vector instruction of an f32-MFMA kernel is matrix time: DESIGN 4)
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f);
        const float4 sh = *reinterpret_cast<const float4 *>(a.shift + co);
        if (a.scale) sc = *reinterpret_cast<const float4 *>(a.scale + co);
        const int rH = Ho >> a.rs, rW = Wo >> a.rs;
        const int mw = m0 + wm * (BM / WM);                  // first output pixel of the wave's block
        const bool full = mw + WROWS <= M;
        const unsigned vo = (unsigned)(rsub * a.Cout + quad * 4) * 4u;
        const size_t tbase = (size_t)mw * a.Cout + cbase;
        char *const obase = reinterpret_cast<char *>(a.out + tbase);
        char *const o2base = reinterpret_cast<char *>(a.out_relu + tbase);                // used when a.out_relu only
        const char *const rbase = reinterpret_cast<const char *>(a.resid + tbase);       // used for rs == 0 only
        // Row-uniform form of the two index-heavy paths (the low-resolution residual, rs != 0, and the sub-pixel scatter, KS == 2; round 4):
        // an iteration handles RPW consecutive output pixels starting at a multiple of RPW; with Wo % RPW == 0 they lie in one image row
        // (and, M being a multiple of RPW then, inside M or past it together), so pixel -> (n, y, x) is SCALAR arithmetic (s_mul_hi_u32) on
        // the iteration's first pixel and a lane adds a kernel-constant offset.  The per-lane form cost ~23 vector instructions per iteration,
        // ten of them quarter-rate 32 / 64-bit multiplies: 4-5 % of a K = 576 tile (every vector instruction here is matrix time).
        const bool efast = (Wo % RPW) == 0 && (RPW >> a.rs) >= 1;
        const unsigned vo_r = (unsigned)((rsub >> a.rs) * a.Cout + quad * 4) * 4u;         // low-resolution residual: column (x0 + rsub) >> rs
        const unsigned vo_s = (unsigned)(2 * rsub * a.Cout + quad * 4) * 4u;               // sub-pixel scatter: column 2 (x0 + rsub) + spx
        // The residual rows are requested FIRST, all NIT of them, before the accumulators go through LDS: one wait for the lot under the
        // transposition instead of a vmcnt(0) per iteration (which also waited for the previous iteration's store).
        float4 rvs[NIT];
        if (a.resid) {
#pragma unroll
            for (int it = 0; it < NIT; it++) {
                float4 rv = make_float4(0.f, 0.f, 0.f, 0.f);
                const int m_it = mw + it * RPW;
                if (a.rs) {
                    if (efast) {
                        if (full || m_it < M) {
                            const int un = fdiv(m_it, dHW);
                            const int rr = m_it - un * (Ho * Wo);
                            const int uy = fdiv(rr, dW), ux = rr - uy * Wo;
                            const size_t ridx = ((size_t)un * rH + (uy >> a.rs)) * rW + (ux >> a.rs);
                            rv = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(a.resid + ridx * a.Cout + cbase) + vo_r);
                        }
                    } else if (full || m_it + rsub < M) {
                        const int m = m_it + rsub;
                        const int n = fdiv(m, dHW);
                        const int rr = m - n * (Ho * Wo);
                        const int y = fdiv(rr, dW), x = rr - y * Wo;
                        const size_t ridx = ((size_t)n * rH + (y >> a.rs)) * rW + (x >> a.rs);
                        rv = *reinterpret_cast<const float4 *>(a.resid + ridx * a.Cout + co);
                    }
                } else if (full || m_it + rsub < M) rv = *reinterpret_cast<const float4 *>(rbase + (size_t)(it * RPW) * a.Cout * 4 + vo);
                rvs[it] = rv;
            }
        }
#pragma unroll
        for (int i = 0; i < TM; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
#pragma unroll
                for (int jn = 0; jn < TN; jn++) Ws[row * WCOLS + jn * 32 + (lane & 31)] = acc[i][jn][r];
            }
#pragma unroll
        for (int it = 0; it < NIT; it++) {
            const int row = it * RPW + rsub;
            const int m = mw + row;
            const bool ok = full || m < M;
            const size_t so = (size_t)(it * RPW) * a.Cout * 4;
            float4 v = *reinterpret_cast<const float4 *>(Ws + row * WCOLS + quad * 4);
            v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y); v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
            if (a.resid) { const float4 rv = rvs[it]; v.x = v.x + rv.x; v.y = v.y + rv.y; v.z = v.z + rv.z; v.w = v.w + rv.w; }
            if (a.relu_out) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }      // uniform branch
            if (KS == 2) {      // scatter to the parity class's pixels of the (2 Ho, 2 Wo) output
                if (efast) {
                    const int m_it = mw + it * RPW;
                    if (full || m_it < M) {
                        const int un = fdiv(m_it, dHW);
                        const int rr = m_it - un * (Ho * Wo);
                        const int uy = fdiv(rr, dW), ux = rr - uy * Wo;
                        const size_t oidx = ((size_t)un * 2 * Ho + 2 * uy + spy) * (2 * Wo) + 2 * ux + spx;
                        *reinterpret_cast<float4 *>(reinterpret_cast<char *>(a.out + oidx * a.Cout + cbase) + vo_s) = v;
                    }
                } else if (ok) {
                    const int n = fdiv(m, dHW);
                    const int rr = m - n * (Ho * Wo);
                    const int y = fdiv(rr, dW), x = rr - y * Wo;
                    const size_t oidx = ((size_t)n * 2 * Ho + 2 * y + spy) * (2 * Wo) + 2 * x + spx;
                    *reinterpret_cast<float4 *>(a.out + oidx * a.Cout + co) = v;
                }
            }
            else if (ok) {
#ifdef TMAT_VAR_BUFSTORE       // round-3 incident (ii): the rows through raw_buffer_store_b128 (tile in the descriptor, iteration in soffset); see DESIGN "Incidents"
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v),
                                                       __builtin_amdgcn_make_buffer_rsrc((void *)obase, 0, 0x7fffffff, 0x00020000), vo, (int)so, 0);
#else
                *reinterpret_cast<float4 *>(obase + so + vo) = v;
#endif
                // the activated copy for the next block's first convolution (uniform branch): four v_max per 16-byte store here instead of
                // sixteen v_max_i32 per K chunk and wave in the consumer's RELU instantiation (0.7 ms per launch)
                if (a.out_relu) {       // relu_pos0: negative values and -0.0 become +0.0, exactly what the RELU instantiation's load-side max does
                    v.x = relu_pos0(v.x); v.y = relu_pos0(v.y); v.z = relu_pos0(v.z); v.w = relu_pos0(v.w);
                    *reinterpret_cast<float4 *>(o2base + so + vo) = v;
                }
            }
        }
    }
#ifdef TMAT_DIAG
    if (dg_on) dg_slot[6] = (long long)__builtin_readcyclecounter() - dg_e0;
#endif
}

template <int BM, int BN, int WM, int WN, int KS>
static void launch_conv_ks(const ConvArgs &a, int M, int Ho, int Wo, hipStream_t s)
{
    int nMt = (M + BM - 1) / BM, nNt = a.Cout / BN;
    dim3 grid(((nMt + 7) / 8) * 8 * nNt, KS == 2 ? 4 : 1);
    const FastDiv dHW = make_fastdiv(Ho * Wo), dW = make_fastdiv(Wo);
    // N-stationary block mapping (see the kernel) for layers whose weights do not stay in an XCD's L2
    const int nstat = (nNt > 1 && nNt <= 8 && 8 % nNt == 0 && (size_t)(KS == 2 ? 4 : KS * KS) * a.Cin * a.Cout * 4 > ((size_t)3 << 20)) ? 1 : 0;
    if (a.prec == 1) {
        if (a.relu_in)
            hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, WM, WN, KS, true, 1>), grid, dim3(64 * WM * WN), 0, s, a, M, Ho, Wo, nMt, nNt, dHW, dW, nstat);
        else
            hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, WM, WN, KS, false, 1>), grid, dim3(64 * WM * WN), 0, s, a, M, Ho, Wo, nMt, nNt, dHW, dW, nstat);
        return;
    }
    if (a.prec == 2) {
        if (a.relu_in)
            hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, WM, WN, KS, true, 2>), grid, dim3(64 * WM * WN), 0, s, a, M, Ho, Wo, nMt, nNt, dHW, dW, nstat);
        else
            hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, WM, WN, KS, false, 2>), grid, dim3(64 * WM * WN), 0, s, a, M, Ho, Wo, nMt, nNt, dHW, dW, nstat);
        return;
    }
    if (a.relu_in)
        hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, WM, WN, KS, true>), grid, dim3(64 * WM * WN), 0, s, a, M, Ho, Wo, nMt, nNt, dHW, dW, nstat);
    else
        hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, WM, WN, KS, false>), grid, dim3(64 * WM * WN), 0, s, a, M, Ho, Wo, nMt, nNt, dHW, dW, nstat);
#ifdef TMAT_DIAG
    if (KS == 3 && BN == 128 && !a.relu_in) {
        static long long zz[2048 * 8 * 8];
        long long z[8] = {0};
        (void)hipStreamSynchronize(s);
        (void)hipMemcpyFromSymbol(zz, HIP_SYMBOL(conv_diag), sizeof(zz));
        const int nb = grid.x < 2048 ? (int)grid.x : 2048;
        for (int b = 0; b < nb; b++) for (int w8 = 0; w8 < 8; w8++) for (int k = 0; k < 7; k++) z[k] += zz[((size_t)b * 8 + w8) * 8 + k];
        const double w = (double)z[5], ch = (double)z[4];
        fprintf(stderr, "[convdiag] Cin %d h %d: per chunk and wave: work %.0f  dma wait %.0f  barrier %.0f cycles | per tile: fill %.0f  epilogue %.0f | chunks per tile %.0f\n",
                a.Cin, a.h, z[0] / ch, z[1] / ch, z[2] / ch, z[3] / w, z[6] / w, ch / w);
    }
#endif
}

template <int BM, int BN, int WM, int WN>
static void launch_conv_cfg(const ConvArgs &a, int M, int Ho, int Wo, hipStream_t s)
{
    if (a.ksize == 3) launch_conv_ks<BM, BN, WM, WN, 3>(a, M, Ho, Wo, s);
    else if (a.ksize == 2) launch_conv_ks<BM, BN, WM, WN, 2>(a, M, Ho, Wo, s);
    else launch_conv_ks<BM, BN, WM, WN, 1>(a, M, Ho, Wo, s);
}

#ifndef TMAT_WM
#define TMAT_WM 4       // waves per block = TMAT_WM x TMAT_WN: 8 waves (32 x 64 outputs each) measured 2-4 % faster than 2 x 2
#define TMAT_WN 2
#endif
#ifndef TMAT_BM
#define TMAT_BM 128
#endif
#ifndef TMAT_64_BM
#define TMAT_64_BM 128  // tile of the Cout = 64 layers
#define TMAT_64_WM 4
#define TMAT_64_WN 2
#endif
#ifndef TMAT_64S_BM
#define TMAT_64S_BM 256 // tile of the sub-pixel Cout = 64 layer
#define TMAT_64S_WM 8
#define TMAT_64S_WN 1
#endif

bool launch_conv(const ConvArgs &a, hipStream_t s)
{
    const int Ho = a.h / a.stride, Wo = a.w / a.stride;
    const long long Mll = (long long)a.N * Ho * Wo;
    if (!((a.ksize == 3 && a.stride == 1) || (a.ksize == 2 && a.stride == 1 && !a.resid) ||
          (a.ksize == 1 && (a.stride == 1 || a.stride == 2))) || a.Cin % 32 || a.Cout % 64 ||
        ((a.ksize == 2 ? 4 : a.ksize * a.ksize) * (a.Cin / 32)) % ((a.ksize & 1) ? 2 * a.ksize * a.ksize : 4) || Mll <= 0 ||
        Mll > 0x7fffffffLL / 2 || (a.resid && a.rs && ((Ho | Wo) & 1)) || Wo < 2 || (a.out_relu && a.ksize == 2)) {
        set_error("launch_conv: unsupported shape");
        return false;
    }
    const int M = (int)Mll;
    if (a.Cout % 128 == 0)
        launch_conv_cfg<TMAT_BM, 128, TMAT_WM, TMAT_WN>(a, M, Ho, Wo, s);
    else if (a.ksize == 2)
        // the sub-pixel layer with 64 output channels (16 chunks per tile, four parity classes): 256 x 64 tiles of eight 32 x 64 wave blocks
        // -- 32 MFMAs per wave and chunk instead of 16 -- measure 20.6 against 21.5 ms; the 3 x 3 layer prefers 128 x 64 (22.7 against 23.2)
        launch_conv_ks<TMAT_64S_BM, 64, TMAT_64S_WM, TMAT_64S_WN, 2>(a, M, Ho, Wo, s);
    else if (a.ksize == 3 && a.prec != 0)          // split precision: a chunk is short, the A-tile traffic per MFMA decides (12.5 -> 11.4 ms in bf16x3)
        launch_conv_ks<256, 64, 8, 1, 3>(a, M, Ho, Wo, s);
    else
        launch_conv_cfg<TMAT_64_BM, 64, TMAT_64_WM, TMAT_64_WN>(a, M, Ho, Wo, s);
    return true;
}

// ---------------------------------------------------------------------------------------------
// depthwise 3x3 (SeparableConv2D's depthwise half, models.py:131,135): chain over the 9 taps in
// (ky, kx) order, zero padding, optional ReLU on load.
// ---------------------------------------------------------------------------------------------
constexpr int DW_ROWS = 8;      // output rows per thread (a 3-row window slides down the column strip)
#ifndef TMAT_DW_WPS
#define TMAT_DW_WPS 1
#endif
__global__ __launch_bounds__(256, TMAT_DW_WPS) void dwconv_kernel(const float *__restrict__ in, int N, int H, int W, int C, int c4shift,
                                                     int relu_in, const float *__restrict__ Wd, float *__restrict__ out, int bpp)
{
    // grid: (ceil((H/8) * (W/4) * C4 / 256), N).  One thread = a strip of 4 consecutive pixels x 8 rows x 4 channels.
    // The 3 x 6 input window slides down the strip: every new output row costs 6 float4 loads for 4 float4 stores
    // (1.9 loads per store over the strip instead of 4.5 for an isolated 3 x 6 window).  Lanes that are adjacent in the
    // channel-group index read adjacent 16 B, so every load instruction moves whole 128-byte lines.
    // 32-bit index math; C4 = C/4 is a power of two.
    // 1-D grid, XCD-aware (round 4): block b runs on XCD b & 7; the bpp blocks of one patch go to one XCD, consecutively, so the rows
    // two vertically adjacent strips share (10 input rows per 8 output rows) are fetched by one L2 only
    const int bj = blockIdx.x >> 3;
    const int n = (bj / bpp) * 8 + (blockIdx.x & 7);
    if (n >= N) return;
    const int e = (bj % bpp) * 256 + threadIdx.x;
    const int C4 = 1 << c4shift;
    const int cq = e & (C4 - 1);
    const int g = e >> c4shift;                 // strip index
    const int WG = W >> 2;
    if (g >= (H / DW_ROWS) * WG) return;
    const int ys = (g / WG) * DW_ROWS, x0 = (g % WG) * 4;
    const float *base = in + (size_t)n * H * W * C + cq * 4;
    const float lo = relu_in ? 0.f : -INFINITY;
    float4 wt[9];
#pragma unroll
    for (int tp = 0; tp < 9; tp++) wt[tp] = *reinterpret_cast<const float4 *>(Wd + tp * C + cq * 4);
    bool xok[6];
#pragma unroll
    for (int c = 0; c < 6; c++) xok[c] = x0 + c - 1 >= 0 && x0 + c - 1 < W;
    auto load_row = [&](int yy, float4 *row) {
        const bool yok = yy >= 0 && yy < H;
#pragma unroll
        for (int c = 0; c < 6; c++) {
            const bool ok = yok && xok[c];
            float4 v = *reinterpret_cast<const float4 *>(base + (ok ? (yy * W + x0 + c - 1) * C : 0));
            v.x = ok ? fmaxf(v.x, lo) : 0.f; v.y = ok ? fmaxf(v.y, lo) : 0.f;
            v.z = ok ? fmaxf(v.z, lo) : 0.f; v.w = ok ? fmaxf(v.w, lo) : 0.f;
            row[c] = v;
        }
    };
    float4 win[3][6];
    load_row(ys - 1, win[0]);
    load_row(ys, win[1]);
    float *obase = out + ((size_t)n * H * W + (size_t)ys * W + x0) * C + cq * 4;
#pragma unroll
    for (int r = 0; r < DW_ROWS; r++) {
        // rows (r, r + 1, r + 2) mod 3 of `win` hold input rows ys + r - 1, ys + r, ys + r + 1
        load_row(ys + r + 1, win[(r + 2) % 3]);
#pragma unroll
        for (int px = 0; px < 4; px++) {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int tp = 0; tp < 9; tp++) {
                const float4 v = win[(r + tp / 3) % 3][px + tp % 3];
                acc.x = fmaf(v.x, wt[tp].x, acc.x); acc.y = fmaf(v.y, wt[tp].y, acc.y);
                acc.z = fmaf(v.z, wt[tp].z, acc.z); acc.w = fmaf(v.w, wt[tp].w, acc.w);
            }
            *reinterpret_cast<float4 *>(obase + ((size_t)r * W + px) * C) = acc;
        }
    }
}

static int ilog2(int v) { int s = 0; while ((1 << s) < v) s++; return s; }

void launch_dwconv(const float *in, int N, int H, int W, int C, int relu_in, const float *Wd, float *out, hipStream_t s)
{
    const int C4 = C / 4;
    const int total = (H / DW_ROWS) * (W / 4) * C4;     // H % 8 == 0 and W % 4 == 0 for every level of the model (checked in tmat_create)
    const int bpp = (total + 255) / 256;
    hipLaunchKernelGGL(dwconv_kernel, dim3((unsigned)(((N + 7) / 8) * 8 * bpp)), dim3(256), 0, s, in, N, H, W, C, ilog2(C4), relu_in, Wd, out, bpp);
}

// ---------------------------------------------------------------------------------------------
// stem: Conv2D(C, 3, strides=2, "same") + BN + ReLU on the single-channel patch (models.py:119-121).
// TF SAME with even H: taps read rows 2y .. 2y+2 (zero beyond the image).
// ---------------------------------------------------------------------------------------------
constexpr int STEM_PX = 8;      // output pixels (of one row) per thread
__global__ __launch_bounds__(256) void stem_kernel(const float *__restrict__ x, int H, int W, const float *__restrict__ Ws,
                                                   int Cout, int c4shift, const float *__restrict__ scale,
                                                   const float *__restrict__ shift, float *__restrict__ out)
{
    // One thread = 8 consecutive output pixels of a row x 4 channels: the 9 tap weights, scale and shift are loaded once
    // per thread and the 3 x 17 input window once per strip (7.5 loads per 16-byte store instead of 18).
    const int Ho = H >> 1, Wo = W >> 1;
    const int n = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int cq = e & ((1 << c4shift) - 1);
    const int g = e >> c4shift;
    const int WG = Wo / STEM_PX;
    if (g >= Ho * WG) return;
    const int yo = g / WG, xo0 = (g - yo * WG) * STEM_PX;
    const float *xin = x + (size_t)n * H * W;
    float4 wt[9];
#pragma unroll
    for (int tp = 0; tp < 9; tp++) wt[tp] = *reinterpret_cast<const float4 *>(Ws + tp * Cout + cq * 4);
    const float4 sc = *reinterpret_cast<const float4 *>(scale + cq * 4);
    const float4 sh = *reinterpret_cast<const float4 *>(shift + cq * 4);
    float win[3][2 * STEM_PX + 1];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 2 * STEM_PX + 1; c++) {
            const int iy = 2 * yo + r, ix = 2 * xo0 + c;
            win[r][c] = (iy < H && ix < W) ? xin[iy * W + ix] : 0.f;
        }
    float *obase = out + ((size_t)n * Ho * Wo + (size_t)yo * Wo + xo0) * Cout + cq * 4;
#pragma unroll
    for (int px = 0; px < STEM_PX; px++) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int tp = 0; tp < 9; tp++) {
            const float v = win[tp / 3][2 * px + tp % 3];
            acc.x = fmaf(v, wt[tp].x, acc.x); acc.y = fmaf(v, wt[tp].y, acc.y);
            acc.z = fmaf(v, wt[tp].z, acc.z); acc.w = fmaf(v, wt[tp].w, acc.w);
        }
        float4 o;
        o.x = fmaxf(fmaf(acc.x, sc.x, sh.x), 0.f); o.y = fmaxf(fmaf(acc.y, sc.y, sh.y), 0.f);
        o.z = fmaxf(fmaf(acc.z, sc.z, sh.z), 0.f); o.w = fmaxf(fmaf(acc.w, sc.w, sh.w), 0.f);
        *reinterpret_cast<float4 *>(obase + (size_t)px * Cout) = o;
    }
}

// The stem at the EVEN output pixels only: out (N, H/4, W/4, Cout), out[yo][xo] = stem(x)[2 yo][2 xo].  When the stem itself is
// recomputed inside the first separable convolution's depthwise producers (sepconv_ws_kernel<..., STEM>), the only other reader of the
// stem tensor is block 0's 1x1 / stride-2 residual convolution (models.py:140), which samples exactly these pixels: a quarter of the
// tensor (2.6 instead of 10.5 GB per pass of 1600 patches) and a unit-stride 1x1 convolution behind it.  Same chain as stem_kernel.
constexpr int SE_PX = 4;        // output pixels per thread (measured on one box: 4 pixels 1.05 ms, 2 pixels -- fewer registers, more waves -- 1.22 ms)
__global__ __launch_bounds__(256) void stem_even_kernel(const float *__restrict__ x, int H, int W, const float *__restrict__ Ws,
                                                        int Cout, int c4shift, const float *__restrict__ scale,
                                                        const float *__restrict__ shift, float *__restrict__ out)
{
    // One thread = SE_PX consecutive output pixels of a row x 4 channels: the 3 rows x 4 SE_PX input columns they touch as aligned
    // 16-byte loads shared by the 16 channel-quad lanes of a pixel group, the taps and the folded BN once per thread.
    const int Ho = H >> 2, Wo = W >> 2, WG = Wo / SE_PX;        // (W / 4) % 4 == 0: the patch size is a multiple of 64
    const int n = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int cq = e & ((1 << c4shift) - 1);
    const int g = e >> c4shift;
    if (g >= Ho * WG) return;
    const int yo = g / WG, xg = g - yo * WG;
    const float *xin = x + (size_t)n * H * W + (size_t)(4 * yo) * W + 4 * SE_PX * xg;      // rows 4 yo .. 4 yo + 2 < H, columns < W
    float win[3][4 * SE_PX];
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c4 = 0; c4 < SE_PX; c4++) {
            const float4 v = *reinterpret_cast<const float4 *>(xin + r * W + c4 * 4);
            win[r][c4 * 4] = v.x; win[r][c4 * 4 + 1] = v.y; win[r][c4 * 4 + 2] = v.z; win[r][c4 * 4 + 3] = v.w;
        }
    float4 wt[9];
#pragma unroll
    for (int tp = 0; tp < 9; tp++) wt[tp] = *reinterpret_cast<const float4 *>(Ws + tp * Cout + cq * 4);
    const float4 sc = *reinterpret_cast<const float4 *>(scale + cq * 4);
    const float4 sh = *reinterpret_cast<const float4 *>(shift + cq * 4);
    float *obase = out + ((size_t)n * Ho * Wo + (size_t)yo * Wo + SE_PX * xg) * Cout + cq * 4;
#pragma unroll
    for (int px = 0; px < SE_PX; px++) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int tp = 0; tp < 9; tp++) {
            const float v = win[tp / 3][4 * px + tp % 3];
            acc.x = fmaf(v, wt[tp].x, acc.x); acc.y = fmaf(v, wt[tp].y, acc.y); acc.z = fmaf(v, wt[tp].z, acc.z); acc.w = fmaf(v, wt[tp].w, acc.w);
        }
        float4 o;
        o.x = fmaxf(fmaf(acc.x, sc.x, sh.x), 0.f); o.y = fmaxf(fmaf(acc.y, sc.y, sh.y), 0.f);
        o.z = fmaxf(fmaf(acc.z, sc.z, sh.z), 0.f); o.w = fmaxf(fmaf(acc.w, sc.w, sh.w), 0.f);
        *reinterpret_cast<float4 *>(obase + (size_t)px * Cout) = o;
    }
}

void launch_stem_even(const float *x, int N, int H, int W, const float *Ws, int Cout, const float *scale,
                      const float *shift, float *out, hipStream_t s)
{
    const int total = (H / 4) * (W / 4 / SE_PX) * (Cout / 4);
    hipLaunchKernelGGL(stem_even_kernel, dim3((total + 255) / 256, N), dim3(256), 0, s, x, H, W, Ws, Cout, ilog2(Cout / 4), scale, shift, out);
}

void launch_stem(const float *x, int N, int H, int W, const float *Ws, int Cout, const float *scale,
                 const float *shift, float *out, hipStream_t s)
{
    const int total = (H / 2) * (W / 2 / STEM_PX) * (Cout / 4);     // (W / 2) % 8 == 0: patch size is a multiple of 64
    hipLaunchKernelGGL(stem_kernel, dim3((total + 255) / 256, N), dim3(256), 0, s, x, H, W, Ws, Cout, ilog2(Cout / 4), scale, shift, out);
}

// ---------------------------------------------------------------------------------------------
// MaxPooling2D(3, strides=2, "same") + residual add (models.py:138-144)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_add_kernel(const float *__restrict__ p2, int H, int W, int C, int c4shift,
                                                          const float *__restrict__ r, float *__restrict__ out, float *__restrict__ out_relu)
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
    if (out_relu) {         // activated copy for the first up block's convolution (see conv_mfma_kernel's epilogue)
        m.x = relu_pos0(m.x); m.y = relu_pos0(m.y); m.z = relu_pos0(m.z); m.w = relu_pos0(m.w);
        *reinterpret_cast<float4 *>(out_relu + o) = m;
    }
}

void launch_maxpool_add(const float *p2, int N, int H, int W, int C, const float *r, float *out, hipStream_t s, float *out_relu)
{
    const int total = (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(maxpool_add_kernel, dim3((total + 255) / 256, N), dim3(256), 0, s, p2, H, W, C, ilog2(C / 4), r, out, out_relu);
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

// final Conv2D(1, 3, sigmoid) over the 2x nearest-upsampled tensor, in sub-pixel form (see conv_mfma_kernel, KS == 2):
// output (2i + py, 2j + px) = sum over the stored pixels (i + py - 1 + a, j + px - 1 + b) of pre-summed taps.
// One thread = the 2 x 2 outputs above stored pixel (i, j): its 3 x 3 stored neighbourhood is read once from LDS per
// 4-channel group and feeds the 16 (class, slot) products.  Chain order per output (shared with oracle orc_final):
// 4-channel group ascending, inside a group slot-major (a, b), then channel ascending.  Weights Wq laid out
// [C/4][4 classes][4 slots][4] are wave-uniform: hipcc reads them with scalar loads, so LDS only carries the pixels.
// One block = 8 x 16 stored pixels (16 x 32 outputs); the 10 x 18 stored pixels they touch are staged in LDS.
//
// LDS layout (round 4): pixel stride CP = CB + 4 floats (36 for CB = 32: the 16 pixels of a tile row then name 16 distinct 16-byte
// bank groups) and a ROW pitch that is a multiple of 64 floats.  A ds_read_b128 is served in the 16-lane groups {0-3, 12-15, 20-27},
// {4-11, 16-19, 28-31}: with 16 threads per tile row a group takes columns {0-3, 12-15} from one tile row and {4-11} from the next,
// so the rows must start at the same bank -- with the natural pitch 18 CP = 648 floats (8 banks further) two of every 16 reads
// collided (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.595, profiles/r03_pmc_layers.txt).
// CB is a template parameter (round 4): with a runtime CB the tile fill divided by CB / 4 and by 18 at run time (~35 vector
// instructions per division, 4 divisions per element) -- 1.3 of the kernel's 2.0 G vector instructions were index arithmetic, and the
// kernel sat at its vector-issue time (3.7 ms) instead of its 11.1 GB of reads.
//
// Block -> tile mapping (round 4): 1-D grid; block b runs on XCD b & 7 (round-robin dispatch), and all tiles of one patch go to ONE
// XCD (patch = 8 (j / tpp) + xcd, tile = j % tpp, j = b >> 3), consecutively: the 10 x 18 halos of neighbouring tiles overlap by
// 41 %, and with the (x, y, patch) grid of rounds 1-3 neighbours sat on different XCDs, so every XCD's L2 fetched the overlap again
// (15.6 GB of fabric reads for an 11.1 GB tensor).
template <int CB>
__global__ __launch_bounds__(128) void final_kernel(const float *__restrict__ S, int N, int h, int w, int C, int tw, int tpp,
                                                    const float *__restrict__ Wq, float bias, float *__restrict__ out)
{
    // The channels go through LDS in blocks of CB (the chain order -- 4-channel groups ascending -- is unchanged, the four
    // accumulators live in registers across the blocks): a workgroup holds 15 KiB at CB = 16 (28 at 32, 49 for all 64 channels),
    // so eight of them (16 waves) fit a CU and the fill of one overlaps the arithmetic of the others; the kernel then runs at the
    // streaming rate of its 11.1 GB of reads (4.4 TB/s).
    constexpr int CP = CB + 4, B4 = CB / 4;
    constexpr int RPITCH = ((18 * CP + 63) / 64) * 64;
    static_assert((B4 & (B4 - 1)) == 0, "channel quads per block: a power of two");
    __shared__ __attribute__((aligned(16))) float tile[10 * RPITCH];    // [10][18 (+ pad)][CB + 4]
    const int bj = blockIdx.x >> 3;
    const int n = (bj / tpp) * 8 + (blockIdx.x & 7);
    if (n >= N) return;
    const int tile_id = bj % tpp;
    const int j0 = (tile_id % tw) * 16, i0 = (tile_id / tw) * 8;
    const int qy = threadIdx.x >> 4, qx = threadIdx.x & 15;
    const int i = i0 + qy, jj = j0 + qx;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};             // classes (py, px) = (0,0), (0,1), (1,0), (1,1)
    const float *t0 = tile + qy * RPITCH + qx * CP;  // stored pixel (i - 1, j - 1)
    constexpr int total = 10 * 18 * B4;
    const float *Sn = S + (size_t)n * h * w * C;
    for (int cb = 0; cb < C; cb += CB) {
        if (cb) __syncthreads();                     // the previous block of channels has been consumed
        // tile fill in batches of 8 loads per thread (all in flight together), then 8 LDS stores
        for (int e0 = threadIdx.x; e0 < total; e0 += 128 * 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int e = e0 + u * 128;
                const int cq = e & (B4 - 1), p = e / B4;
                const int py = p / 18, px = p - py * 18;
                const int ly = i0 - 1 + py, lx = j0 - 1 + px;
                const bool ok = e < total && ly >= 0 && ly < h && lx >= 0 && lx < w;
                v[u] = *reinterpret_cast<const float4 *>(Sn + (ok ? ((size_t)ly * w + lx) * C + cb + cq * 4 : (size_t)0));
                if (!ok) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int e = e0 + u * 128;
                const int cq = e & (B4 - 1), p = e / B4;
                const int py = p / 18, px = p - py * 18;
                if (e < total) *reinterpret_cast<float4 *>(tile + py * RPITCH + px * CP + cq * 4) = v[u];
            }
        }
        __syncthreads();
        if (i < h && jj < w) {
#pragma unroll 2
            for (int cg = 0; cg < B4; cg++) {
                float4 v[3][3];
#pragma unroll
                for (int r = 0; r < 3; r++)
#pragma unroll
                    for (int c = 0; c < 3; c++) v[r][c] = *reinterpret_cast<const float4 *>(t0 + r * RPITCH + c * CP + cg * 4);
                const float *wq = Wq + ((cb >> 2) + cg) * 64;
#pragma unroll
                for (int cls = 0; cls < 4; cls++)
#pragma unroll
                    for (int tp = 0; tp < 4; tp++) {
                        const float4 x = v[(cls >> 1) + (tp >> 1)][(cls & 1) + (tp & 1)];
                        const float *ww = wq + (cls * 4 + tp) * 4;
                        acc[cls] = fmaf(x.x, ww[0], acc[cls]); acc[cls] = fmaf(x.y, ww[1], acc[cls]);
                        acc[cls] = fmaf(x.z, ww[2], acc[cls]); acc[cls] = fmaf(x.w, ww[3], acc[cls]);
                    }
            }
        }
    }
    if (i >= h || jj >= w) return;
    const int W = 2 * w;
    float *o = out + ((size_t)n * 2 * h + 2 * i) * W + 2 * jj;
    float2 r0, r1;
    r0.x = 1.0f / (1.0f + exp_det(-(acc[0] + bias))); r0.y = 1.0f / (1.0f + exp_det(-(acc[1] + bias)));
    r1.x = 1.0f / (1.0f + exp_det(-(acc[2] + bias))); r1.y = 1.0f / (1.0f + exp_det(-(acc[3] + bias)));
    *reinterpret_cast<float2 *>(o) = r0;
    *reinterpret_cast<float2 *>(o + W) = r1;
}

// Wq: [C/4][4][4][4] (tmat_api.cpp:final_subpixel_weights)
void launch_final(const float *S, int N, int h, int w, int C, const float *Wq, float bias, float *out, hipStream_t s)
{
    const int tw = (w + 15) / 16, tpp = tw * ((h + 7) / 8);
    const dim3 grid((unsigned)(((N + 7) / 8) * 8 * tpp));           // N <= max_patches and tpp = 200 at 160 x 160: far below 2^31
    // channels per LDS block: 32 when the channel count allows (every shipped model: C = 64), else 16 / 8 / 4 (C % 4 == 0: tmat_create)
#ifndef TMAT_FINAL_CB
#define TMAT_FINAL_CB 16    // measured per launch of 1600 patches: 16 channels per LDS block 2.51 ms (15 KB of LDS: 8 workgroups = 16 waves per CU), 32: 3.22 ms (5 workgroups), 8: 4.16 ms
#endif
    if (C % 32 == 0) hipLaunchKernelGGL(final_kernel<TMAT_FINAL_CB>, grid, dim3(128), 0, s, S, N, h, w, C, tw, tpp, Wq, bias, out);
    else if (C % 16 == 0) hipLaunchKernelGGL(final_kernel<16>, grid, dim3(128), 0, s, S, N, h, w, C, tw, tpp, Wq, bias, out);
    else if (C % 8 == 0) hipLaunchKernelGGL(final_kernel<8>, grid, dim3(128), 0, s, S, N, h, w, C, tw, tpp, Wq, bias, out);
    else hipLaunchKernelGGL(final_kernel<4>, grid, dim3(128), 0, s, S, N, h, w, C, tw, tpp, Wq, bias, out);
}

}  // namespace tmat
