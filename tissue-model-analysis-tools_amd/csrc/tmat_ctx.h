// Handle state of libtmat_hip.so.
#pragma once
#include "tmat_internal.h"
#include "gauss_tables.h"

#include <map>

namespace tmat {

struct DownBlock {
    int cin = 0, cout = 0;
    float *dw[2] = {nullptr, nullptr};      // [9][C]
    float *pw[2] = {nullptr, nullptr};      // [Cout][Cin] (k contiguous: conv_mfma_kernel and the fused separable kernel)
    float *scale[2] = {nullptr, nullptr}, *shift[2] = {nullptr, nullptr};
    float *res_w = nullptr, *res_b = nullptr;
};
struct UpBlock {
    int cin = 0, cout = 0;
    float *ct[2] = {nullptr, nullptr};      // conv-form taps [9][Cin][Cout]
    float *ct_sub = nullptr;                // first conv in sub-pixel form [4][4][Cin][Cout] (blocks after the first)
    float *scale[2] = {nullptr, nullptr}, *shift[2] = {nullptr, nullptr};
    float *res_w = nullptr, *res_b = nullptr;
};
struct ProfEv { hipEvent_t e0, e1; double flops; };
// a device / pinned workspace a forward or a pass writes before it reads (tmat_debug_poison fills exactly these)
struct WsEnt { void *p; size_t bytes; bool host; };
struct ConvWHost { std::vector<float> w; int cin; };        // host copy of an MFMA convolution's weights ([rows][cin])

// per-geometry buffers of the batch pipeline (pipeline.cpp)
struct PassBuf {
    int K = 0, H = 0, W = 0, h = 0, w = 0;
    int *xi = nullptr, *yi = nullptr;
    float *xc = nullptr, *yc = nullptr;
    float *tmp = nullptr, *x = nullptr;
    uint16_t *small = nullptr;
    int *mn = nullptr, *mx = nullptr;
    double *pred[2] = {nullptr, nullptr};
    double *pred_host[2] = {nullptr, nullptr};
    // GPU morphology outputs (morph_kernels.hip): filtered mask, its EDT, per-image convergence flags
    void *morph_ws = nullptr;
    uint8_t *filt[2] = {nullptr, nullptr};
    double *dist[2] = {nullptr, nullptr};
    uint8_t *filt_host[2] = {nullptr, nullptr};
    double *dist_host[2] = {nullptr, nullptr};
    int *conv_host[2] = {nullptr, nullptr};
    // finish stage (finish_kernels.hip): skeleton from the host thinning, vesselness field back to the host
    int fh = 0, fw = 0;
    void *finish_ws = nullptr;
    uint8_t *skel[2] = {nullptr, nullptr};
    uint8_t *skel_host[2] = {nullptr, nullptr};
    float *field[2] = {nullptr, nullptr}, *f255[2] = {nullptr, nullptr};
    float *f255_host[2] = {nullptr, nullptr};
    // ordered thinning on the device (thin_kernels.hip): foreground counts, RandomState(0) permutations (host-made), scratch
    void *thin_ws = nullptr;
    int *nfg[2] = {nullptr, nullptr};
    int *nfg_host[2] = {nullptr, nullptr};
    uint32_t *tie = nullptr;
    uint32_t *tie_host[2] = {nullptr, nullptr};
    // DMT front end (dmt_kernels.hip): lower-star sorted edge ids of every image of the pass, counts of kept edges
    void *dmt_ws = nullptr;
    int32_t *dmt_ids[2] = {nullptr, nullptr};
    int32_t *dmt_ids_host[2] = {nullptr, nullptr};
    int *dmt_m[2] = {nullptr, nullptr};
    int *dmt_m_host[2] = {nullptr, nullptr};
    // the two persistence sweeps on the device (dmt_sweep_kernels.hip): pairing kind and persistence of every sorted edge
    void *dmt_sweep_ws = nullptr;
    uint8_t *dmt_kind[2] = {nullptr, nullptr};
    uint8_t *dmt_kind_host[2] = {nullptr, nullptr};
    float *dmt_pers[2] = {nullptr, nullptr};
    float *dmt_pers_host[2] = {nullptr, nullptr};
    hipEvent_t done[2] = {nullptr, nullptr};
    std::vector<WsEnt> ws;                                  // every scratch allocation above with its size (not the Lanczos tables: they are constants)
};

// weight container ("TMATW001", tmat_amd/synth.py:pack_weights): tensors by name, pointing into the caller's blob
struct Tensor { std::vector<int> shape; const float *data; size_t count; };
bool parse_blob(const void *blob, size_t nbytes, std::map<std::string, Tensor> &out, int &patch);
std::vector<float> k_contiguous(const float *w, int taps, int I, int O);        // [taps][I][O] -> [taps][O][I]

// ResNet50 classifier of the invasion-depth tool (resnet_kernels.hip)
struct ResConv { int cin = 0, cout = 0, ksize = 1, stride = 1; float *w = nullptr, *scale = nullptr, *shift = nullptr; };
struct ResBlock { ResConv c1, c2, c3, sc; bool has_sc = false; };
struct ResNetModel {
    float *stem_w = nullptr, *stem_scale = nullptr, *stem_shift = nullptr;      // [147][64], folded BN
    std::vector<ResBlock> blocks;
    float *fc_w = nullptr;
    float fc_b = 0.f;
    int feat = 0;                                                               // channels of the last block
    std::vector<void *> owned;
};

struct GaussKey {
    double sigma; int order, radius;
    bool operator<(const GaussKey &o) const { return sigma != o.sigma ? sigma < o.sigma : order != o.order ? order < o.order : radius < o.radius; }
};

struct Ctx {
    int device = 0;
    std::vector<ResNetModel> resnets;                        // invasion-depth classifiers loaded on this handle (tmat_resnet_load)
    std::map<GaussKey, GaussTable> gauss;                    // gaussian kernel tables (gauss_tables.cpp)
    std::map<const GaussTable *, double *> gauss_dev;        // their device copies (stack_pipeline.cpp)
    hipStream_t stream = nullptr;
    int patch = 0, max_patches = 0;
    int f0 = 0, f_last = 0;
    float *stem_w = nullptr, *stem_scale = nullptr, *stem_shift = nullptr;
    std::vector<DownBlock> down;
    std::vector<UpBlock> up;
    float *final_w = nullptr;
    float final_b = 0.f;
    std::vector<void *> owned;              // weight allocations
    float *buf[4] = {nullptr, nullptr, nullptr, nullptr};    // down-path ping-pong activations
    float *ubuf[4] = {nullptr, nullptr, nullptr, nullptr};   // up-path activations
    float *dout[2] = {nullptr, nullptr};                     // down-path output (P/16)^2 x f_deep, double-buffered
    // activated copies (max(x, +0)) of the tensors a block's first convolution reads: the bottleneck (written by the last pooling) and
    // the outputs of up blocks 0 .. n-2 (written by conv_mfma_kernel's epilogue); TMAT_RELU_COPY=0: ReLU on load instead
    bool relu_copy = true;
    float *dout_relu[2] = {nullptr, nullptr};
    bool dout_relu_ok[2] = {false, false};                   // the last down-path call wrote dout_relu[i] (only the unfused last block does)
    float *urelu[2] = {nullptr, nullptr};
    size_t urelu_bytes[2] = {0, 0};
    hipStream_t stream2 = nullptr;                           // second stream: down path of pass p+1 overlaps up path of pass p
    hipStream_t stream3 = nullptr;                           // third stream: finish stage of pass p-1 (after the host thinning)
    hipEvent_t ev_down[2] = {nullptr, nullptr};
    hipEvent_t ev_pre[2] = {nullptr, nullptr};              // front end of a pass (Lanczos ... tile gather) done: patch_in_of(slot) is complete
    bool down_pending[2] = {false, false};
    hipEvent_t ev_up[2] = {nullptr, nullptr}, ev_blend[2] = {nullptr, nullptr};      // the tail of a pass (blend, mask filter, EDT) on the second stream
    bool blend_pending[2] = {false, false};
    float *patch_in = nullptr, *patch_out = nullptr;
    float *patch_in2 = nullptr;                              // second input buffer of the batch path: the front end of pass p + 2 runs beside pass p + 1's network (pipeline.cpp)
    bool pre_side = true;                                    // that front end on the second stream (TMAT_PRE_STREAM=0: on the main stream)
    size_t buf_bytes[4] = {0, 0, 0, 0}, ubuf_bytes[4] = {0, 0, 0, 0}, dout_bytes = 0;      // sizes of the activation workspaces (tmat_debug_poison)
    float input_sat = 65535.f;                               // Lanczos saturation: 65535, or 255 for 8-bit sources (tmat_set_input_depth)
    int patch_cap = 0;                                       // patches patch_in / patch_out hold (>= max_patches)
    void *scratch = nullptr;
    size_t scratch_bytes = 0;
    double *win1d = nullptr;
    std::vector<double> win_host;
    PassBuf pass;
    void free_pass()
    {
        PassBuf &b = pass;
        void *dev[] = {b.xi, b.yi, b.xc, b.yc, b.tmp, b.x, b.small, b.mn, b.mx, b.pred[0], b.pred[1], b.morph_ws,
                       b.filt[0], b.filt[1], b.dist[0], b.dist[1], b.finish_ws, b.skel[0], b.skel[1], b.field[0], b.field[1],
                       b.f255[0], b.f255[1], b.dmt_ws, b.dmt_ids[0], b.dmt_ids[1], b.dmt_m[0], b.dmt_m[1], b.thin_ws, b.nfg[0], b.nfg[1], b.tie,
                       b.dmt_sweep_ws, b.dmt_kind[0], b.dmt_kind[1], b.dmt_pers[0], b.dmt_pers[1]};
        for (void *p : dev) if (p) hipFree(p);
        for (int i = 0; i < 2; i++) {
            if (b.pred_host[i]) hipHostFree(b.pred_host[i]);
            if (b.filt_host[i]) hipHostFree(b.filt_host[i]);
            if (b.dist_host[i]) hipHostFree(b.dist_host[i]);
            if (b.conv_host[i]) hipHostFree(b.conv_host[i]);
            if (b.skel_host[i]) hipHostFree(b.skel_host[i]);
            if (b.f255_host[i]) hipHostFree(b.f255_host[i]);
            if (b.dmt_ids_host[i]) hipHostFree(b.dmt_ids_host[i]);
            if (b.dmt_m_host[i]) hipHostFree(b.dmt_m_host[i]);
            if (b.dmt_kind_host[i]) hipHostFree(b.dmt_kind_host[i]);
            if (b.dmt_pers_host[i]) hipHostFree(b.dmt_pers_host[i]);
            if (b.nfg_host[i]) hipHostFree(b.nfg_host[i]);
            if (b.tie_host[i]) hipHostFree(b.tie_host[i]);
            if (b.done[i]) hipEventDestroy(b.done[i]);
        }
        pass = PassBuf();
    }
    bool thin_device = true;                                 // ordered medial-axis thinning on the device (TMAT_THIN_DEVICE=0: host threads)
    uint32_t *ma_table = nullptr;                            // its 512-entry decision table, 16 words
    bool dmt_device = true;                                  // DMT key build + lower-star sort on the device (TMAT_DMT_DEVICE=0: host)
    bool dmt_sweep_device = true;                            // the two persistence sweeps on the device as well (TMAT_DMT_SWEEP_DEVICE=0: host threads)
    bool fused_pool = true;                                  // max-pool + residual add fused behind the second separable convolution (TMAT_FUSED_POOL=0: separate kernel)
    bool norm_on = false;                                    // models.py:636-637 input normalisation in front of the smooth prediction (tmat_set_input_norm)
    float norm_mean = 0.f, norm_std = 1.f;
    int precision = 0;                                       // TMAT_PRECISION_F32 (bit-exact contract) or TMAT_PRECISION_BF16X3 / _BF16X6 (opt-in, tmat_set_precision)
    std::map<const float *, ConvWHost> conv_w_host;          // device pointer of every MFMA convolution weight tensor -> its host copy
    std::map<int, std::map<const float *, float *>> wsplit;  // precision mode -> (... -> its split-precision copy on the device, made on first use)
    bool sep_bf16 = true;                                    // bf16x3 mode also runs the separable layers' pointwise part on the bf16 cores (TMAT_SEP_BF16=0: f32)
    bool stem_fused = true;                                  // the stem recomputed inside block 0's first separable convolution (TMAT_STEM_FUSED=0: stem_kernel writes its tensor)
    bool fused_sep = true;                                   // fused depthwise -> pointwise kernel (sepconv_ws_kernel) where the level allows (TMAT_FUSED_SEP=0: separate kernels)
    // call-scoped device workspaces of the side tools (cell area, invasion depth), kept between calls: with the reference's default batch of 4 images a
    // hipMalloc / hipFree pair per buffer and call costs more than the batch's kernels.  Slot = a fixed id per buffer (ws_get below).
    static constexpr int N_TOOL_WS = 24;                     // 0-8 cell area, 9-10 Z projection, 12-22 invasion depth
    void *tool_ws[N_TOOL_WS] = {};
    size_t tool_ws_bytes[N_TOOL_WS] = {};
    std::multimap<size_t, void *> ws_pool;                   // released call-scoped blocks of the Z-stack tool, by size (stack_pipeline.cpp:Arena)
    // profiling of the dominant kernel family
    bool prof_on = false;
    std::vector<ProfEv> ev_open;
    std::vector<hipEvent_t> ev_pool;
    double prof_ms = 0, prof_flops = 0;
    int64_t prof_launches = 0;
};

int unet_forward_dev(Ctx *c, const float *X, int n, float *Y, hipStream_t s);
int ensure_patch_io(Ctx *c, int n_patches);
int unet_down_dev(Ctx *c, const float *X, int n, float *dout, hipStream_t s);
int unet_up_dev(Ctx *c, const float *dout, int n, float *Y, hipStream_t s);
int predict_smooth_dev(Ctx *c, float *x_dev, int n, int hh, int ww, double *pred_dev);      // x_dev is normalised in place when tmat_set_input_norm is on

// numpy's pairwise summation of a contiguous f64 vector (np.add.reduce / np.mean inner loop):
// 8 interleaved partial sums on blocks <= 128, recursive halves (rounded down to a multiple of 8) above.
inline double numpy_pairwise_sum(const double *a, long n)
{
    if (n < 8) {
        double res = 0.0;
        for (long i = 0; i < n; i++) res += a[i];
        return res;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; j++) r[j] = a[j];
        long i;
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    }
    long n2 = n / 2;
    n2 -= n2 % 8;
    return numpy_pairwise_sum(a, n2) + numpy_pairwise_sum(a + n2, n - n2);
}

// device workspace `slot` of the handle with room for `bytes` (grown by reallocation: the caller's stream must not have work in flight on
// the old buffer -- every user synchronises before it returns); nullptr + set_error on failure
inline void *ws_get(Ctx *c, int slot, size_t bytes)
{
    if (c->tool_ws_bytes[slot] >= bytes && c->tool_ws[slot]) return c->tool_ws[slot];
    if (c->tool_ws[slot]) { hipFree(c->tool_ws[slot]); c->tool_ws[slot] = nullptr; c->tool_ws_bytes[slot] = 0; }
    if (!hip_ok(hipMalloc(&c->tool_ws[slot], bytes ? bytes : 16), "hipMalloc(tool workspace)")) { c->tool_ws[slot] = nullptr; return nullptr; }
    c->tool_ws_bytes[slot] = bytes ? bytes : 16;
    return c->tool_ws[slot];
}

// handles made by tmat_create_plain carry no model
inline bool has_model(const Ctx *c) { return c && !c->up.empty(); }
}  // namespace tmat
