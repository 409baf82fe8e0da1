// C-ABI entry points of libtmat_hip.so (see include/tmat.h) -- context, weights, UNet driver,
// smooth tiled prediction.  gfx950 only; there is deliberately no CPU fallback: every compute
// entry point needs a HIP device and fails loudly without one.
#include "../../include/tmat.h"
#include "tmat_ctx.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>

namespace tmat {

// One process-wide message (the last error of any thread): entry points run stages on worker threads, and a message
// recorded there must reach the thread that called the entry point.
static std::mutex g_err_mu;
static std::string g_err;
void set_error(const std::string &msg)
{
    std::lock_guard<std::mutex> lk(g_err_mu);
    g_err = msg;
}
bool hip_ok(hipError_t e, const char *what)
{
    if (e == hipSuccess) return true;
    set_error(std::string(what) + ": " + hipGetErrorString(e));
    return false;
}

// ---------------------------------------------------------------------------------------------
// weight container ("TMATW001", tmat_amd/synth.py:pack_weights)
// ---------------------------------------------------------------------------------------------

bool parse_blob(const void *blob, size_t nbytes, std::map<std::string, Tensor> &out, int &patch)
{
    const uint8_t *p = (const uint8_t *)blob;
    if (nbytes < 16 || memcmp(p, "TMATW001", 8)) { set_error("weights: bad magic"); return false; }
    uint32_t n, ps;
    memcpy(&n, p + 8, 4); memcpy(&ps, p + 12, 4);
    patch = (int)ps;
    size_t pos = 16;
    for (uint32_t i = 0; i < n; i++, pos += 84) {
        if (pos + 84 > nbytes) { set_error("weights: truncated table"); return false; }
        char name[49]; memcpy(name, p + pos, 48); name[48] = 0;
        uint32_t ndim, d[4]; uint64_t off, cnt;
        memcpy(&ndim, p + pos + 48, 4); memcpy(d, p + pos + 52, 16);
        memcpy(&off, p + pos + 68, 8); memcpy(&cnt, p + pos + 76, 8);
        // overflow-safe bounds: off inside the blob, cnt floats fit behind it (no off + cnt * 4 that could wrap)
        if (ndim > 4 || off % 4 || off > nbytes || cnt > (nbytes - off) / 4) { set_error("weights: bad entry"); return false; }
        Tensor t; t.data = (const float *)(p + off); t.count = cnt;
        uint64_t prod = 1;
        for (uint32_t k = 0; k < ndim; k++) {
            if (d[k] == 0 || d[k] > (1u << 24) || prod > (1ull << 40)) { set_error("weights: bad dimension"); return false; }
            t.shape.push_back((int)d[k]); prod *= d[k];
        }
        if (prod != cnt) { set_error("weights: shape/count mismatch"); return false; }
        out[name] = t;
    }
    return true;
}

static bool upload(Ctx *c, const std::vector<float> &v, float **dev)
{
    if (!hip_ok(hipMalloc((void **)dev, v.size() * sizeof(float)), "hipMalloc(weights)")) return false;
    c->owned.push_back(*dev);
    return hip_ok(hipMemcpy(*dev, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy(weights)");
}
// weights of an MFMA convolution ([taps][Cout][Cin], k contiguous): the host copy is kept so that tmat_set_precision can
// make the split-precision copy later
static bool upload_conv(Ctx *c, const std::vector<float> &v, int cin, float **dev)
{
    if (!upload(c, v, dev)) return false;
    c->conv_w_host[*dev] = ConvWHost{v, cin};
    return true;
}

// round-to-nearest-even f32 -> bf16 (finite inputs: weights)
static uint16_t bf16_rne(float x)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf16_to_f32(uint16_t b)
{
    const uint32_t u = (uint32_t)b << 16;
    float x;
    memcpy(&x, &u, 4);
    return x;
}
// Split-precision copy of conv weights for conv_mfma_kernel<..., PREC = 1 | 2> (unet_kernels.hip): npl bf16 planes of the
// tensor's own layout, plane 0 = rne(w), plane 1 = rne(w - p0), plane 2 = rne(w - p0 - p1) (each difference is exact in f32).
// Returned as a float vector used as a byte container (npl * w.size() bf16 values).
static std::vector<float> split_bf16(const std::vector<float> &w, int npl)
{
    std::vector<float> out((w.size() * npl + 1) / 2);
    uint16_t *o = reinterpret_cast<uint16_t *>(out.data());
    const size_t n = w.size();
    for (size_t i = 0; i < n; i++) {
        float rem = w[i];
        for (int pl = 0; pl < npl; pl++) {
            const uint16_t q = bf16_rne(rem);
            o[(size_t)pl * n + i] = q;
            rem = rem - bf16_to_f32(q);
        }
    }
    return out;
}

// scale = f32(gamma / sqrt(var + eps)); shift = f32(beta + (bias - mean) * scale_f64)   (BN folding)
static void fold_bn(const Tensor &bn, const Tensor &bias, std::vector<float> &scale, std::vector<float> &shift)
{
    int C = bn.shape[1];
    scale.resize(C); shift.resize(C);
    for (int i = 0; i < C; i++) {
        double g = bn.data[i], b = bn.data[C + i], m = bn.data[2 * C + i], v = bn.data[3 * C + i];
        double sd = g / std::sqrt(v + 1e-3);
        scale[i] = (float)sd;
        shift[i] = (float)(b + ((double)bias.data[i] - m) * sd);
    }
}

// Keras Conv2DTranspose kernel (kh, kw, out, in), stride 1, SAME -> conv taps
// Wc[a][b][in][out] = K[2-a][2-b][out][in]
static std::vector<float> convt_as_conv(const Tensor &k)
{
    int O = k.shape[2], I = k.shape[3];
    std::vector<float> w((size_t)9 * I * O);
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++)
            for (int i = 0; i < I; i++)
                for (int o = 0; o < O; o++)
                    w[(((size_t)a * 3 + b) * I + i) * O + o] = k.data[(((size_t)(2 - a) * 3 + (2 - b)) * O + o) * I + i];
    return w;
}

// Sub-pixel form of a 3x3 convolution over a 2x nearest-upsampled tensor (unet_kernels.hip, KS == 2): per output
// parity class (py, px) the taps that land on the same stored pixel are summed -- f32 adds from +0.0 in (ky, kx)
// order, as oracle/unet.py:subpixel_weights does.  [9][I][O] -> [4 classes][4 slots][I][O]
static std::vector<float> subpixel_weights(const std::vector<float> &w9, int I, int O)
{
    const size_t IO = (size_t)I * O;
    std::vector<float> w(16 * IO, 0.0f);
    for (int py = 0; py < 2; py++)
        for (int px = 0; px < 2; px++)
            for (int ky = 0; ky < 3; ky++)
                for (int kx = 0; kx < 3; kx++) {
                    const int a = ((py + ky - 1) >> 1) - (py - 1), b = ((px + kx - 1) >> 1) - (px - 1);
                    float *d = &w[((size_t)(py * 2 + px) * 4 + a * 2 + b) * IO];
                    const float *s = &w9[(size_t)(ky * 3 + kx) * IO];
                    for (size_t i = 0; i < IO; i++) d[i] = d[i] + s[i];
                }
    return w;
}

// [taps][I][O] -> [taps][O][I]: the MFMA convolution reads both operands k-contiguous (unet_kernels.hip)
std::vector<float> k_contiguous(const float *w, int taps, int I, int O)
{
    std::vector<float> r((size_t)taps * I * O);
    for (int t = 0; t < taps; t++)
        for (int i = 0; i < I; i++)
            for (int o = 0; o < O; o++) r[((size_t)t * O + o) * I + i] = w[((size_t)t * I + i) * O + o];
    return r;
}

static bool need(const std::map<std::string, Tensor> &m, const std::string &k, Tensor &t)
{
    auto it = m.find(k);
    if (it == m.end()) { set_error("weights: missing tensor " + k); return false; }
    t = it->second;
    return true;
}
// shape checks of what build_model indexes: a BatchNormalization block is (4, C) [gamma, beta, mean, var], a bias is (C)
static bool bn_ok(const Tensor &bn, const Tensor &bias, int C)
{
    if (bn.shape.size() != 2 || bn.shape[0] != 4 || bn.shape[1] != C || (int)bias.count != C) {
        set_error("weights: BatchNormalization / bias tensor does not match its convolution");
        return false;
    }
    return true;
}

static bool build_model(Ctx *c, const std::map<std::string, Tensor> &m)
{
    Tensor w, b, bn;
    std::vector<float> sc, sh;
    if (!need(m, "stem.w", w) || !need(m, "stem.b", b) || !need(m, "stem.bn", bn)) return false;
    if (w.shape.size() != 4 || w.shape[2] != 1) { set_error("weights: only channels=1 supported"); return false; }
    c->f0 = w.shape[3];
    if (w.shape[0] != 3 || w.shape[1] != 3) { set_error("weights: stem must be 3x3"); return false; }
    if (!bn_ok(bn, b, c->f0)) return false;
    fold_bn(bn, b, sc, sh);
    if (!upload(c, std::vector<float>(w.data, w.data + w.count), &c->stem_w) || !upload(c, sc, &c->stem_scale) ||
        !upload(c, sh, &c->stem_shift)) return false;
    int cin = c->f0;
    for (int i = 0;; i++) {
        std::string p = "down" + std::to_string(i);
        if (!m.count(p + ".sep1.dw")) break;
        DownBlock d; d.cin = cin;
        Tensor dw, pw, rb;
        for (int s = 0; s < 2; s++) {
            std::string sp = p + (s ? ".sep2" : ".sep1");
            if (!need(m, sp + ".dw", dw) || !need(m, sp + ".pw", pw) || !need(m, sp + ".b", b) ||
                !need(m, p + (s ? ".bn2" : ".bn1"), bn)) return false;
            if (dw.shape.size() != 3 || pw.shape.size() != 2 || dw.count != (size_t)9 * pw.shape[0] || pw.shape[0] != (s ? d.cout : d.cin) ||
                !bn_ok(bn, b, pw.shape[1])) {
                set_error("weights: separable convolution tensors of " + sp + " have unexpected shapes");
                return false;
            }
            fold_bn(bn, b, sc, sh);
            if (!upload(c, std::vector<float>(dw.data, dw.data + dw.count), &d.dw[s]) ||
                !upload_conv(c, k_contiguous(pw.data, 1, pw.shape[0], pw.shape[1]), pw.shape[0], &d.pw[s]) || !upload(c, sc, &d.scale[s]) ||
                !upload(c, sh, &d.shift[s])) return false;
            d.cout = pw.shape[1];
        }
        if (!need(m, p + ".res.w", w) || !need(m, p + ".res.b", rb)) return false;
        if (w.count != (size_t)d.cin * d.cout || (int)rb.count != d.cout) { set_error("weights: residual convolution of " + p + " has unexpected shape"); return false; }
        if (!upload_conv(c, k_contiguous(w.data, 1, d.cin, d.cout), d.cin, &d.res_w) ||
            !upload(c, std::vector<float>(rb.data, rb.data + rb.count), &d.res_b)) return false;
        c->down.push_back(d);
        cin = d.cout;
    }
    for (int j = 0;; j++) {
        std::string p = "up" + std::to_string(j);
        if (!m.count(p + ".ct1.w")) break;
        UpBlock u; u.cin = cin;
        Tensor rb;
        for (int s = 0; s < 2; s++) {
            std::string sp = p + (s ? ".ct2" : ".ct1");
            if (!need(m, sp + ".w", w) || !need(m, sp + ".b", b) || !need(m, p + (s ? ".bn2" : ".bn1"), bn)) return false;
            if (w.shape.size() != 4 || w.shape[0] != 3 || w.shape[1] != 3) { set_error("weights: ConvT must be 3x3"); return false; }
            if (w.shape[3] != (s ? w.shape[2] : u.cin) || !bn_ok(bn, b, w.shape[2])) { set_error("weights: transposed convolution tensors of " + sp + " have unexpected shapes"); return false; }
            fold_bn(bn, b, sc, sh);
            const std::vector<float> w9 = convt_as_conv(w);
            const int I = w.shape[3], O = w.shape[2];
            if (!upload_conv(c, k_contiguous(w9.data(), 9, I, O), I, &u.ct[s]) || !upload(c, sc, &u.scale[s]) || !upload(c, sh, &u.shift[s])) return false;
            // blocks after the first read a 2x nearest-upsampled tensor in their first convolution: sub-pixel form
            if (s == 0 && j > 0 && !upload_conv(c, k_contiguous(subpixel_weights(w9, I, O).data(), 16, I, O), I, &u.ct_sub)) return false;
            u.cout = w.shape[2];
        }
        if (!need(m, p + ".res.w", w) || !need(m, p + ".res.b", rb)) return false;
        if (w.count != (size_t)u.cin * u.cout || (int)rb.count != u.cout) { set_error("weights: residual convolution of " + p + " has unexpected shape"); return false; }
        if (!upload_conv(c, k_contiguous(w.data, 1, u.cin, u.cout), u.cin, &u.res_w) ||
            !upload(c, std::vector<float>(rb.data, rb.data + rb.count), &u.res_b)) return false;
        c->up.push_back(u);
        cin = u.cout;
    }
    if (!need(m, "final.w", w) || !need(m, "final.b", b)) return false;
    {   // final 3x3 conv over the upsampled tensor: sub-pixel form, laid out [C/4][class][slot][4] for unet_kernels.hip:final_kernel
        const int C = cin;
        if ((int)w.count != 9 * C || C % 4) { set_error("weights: final conv must be 3x3xCx1"); return false; }
        const std::vector<float> sub = subpixel_weights(std::vector<float>(w.data, w.data + w.count), C, 1);   // [4][4][C]
        std::vector<float> q((size_t)16 * C);
        for (int cg = 0; cg < C / 4; cg++)
            for (int ct = 0; ct < 16; ct++)
                for (int e = 0; e < 4; e++) q[((size_t)cg * 16 + ct) * 4 + e] = sub[(size_t)ct * C + cg * 4 + e];
        if (!upload(c, q, &c->final_w)) return false;
    }
    c->final_b = b.data[0];
    c->f_last = cin;
    if (c->down.size() != c->up.size() - 1 || c->down.empty()) { set_error("weights: unexpected block structure"); return false; }
    int P = c->patch;
    if (P % (8 << c->down.size()) || P <= 0) { set_error("weights: patch size must be divisible by 2^(blocks+1)"); return false; }
    return true;
}

// ---------------------------------------------------------------------------------------------
// UNet forward on device patches X (n, P, P) -> Y (n, P, P); n <= max_patches.
// Buffer plan (4 ping-pong activation buffers) is documented in DESIGN.md.
// ---------------------------------------------------------------------------------------------
static void prof_begin(Ctx *c, double flops, hipStream_t st)
{
    if (!c->prof_on) return;
    hipEvent_t e0, e1;
    if (c->ev_pool.size() >= 2) { e0 = c->ev_pool.back(); c->ev_pool.pop_back(); e1 = c->ev_pool.back(); c->ev_pool.pop_back(); }
    else { hipEventCreate(&e0); hipEventCreate(&e1); }
    hipEventRecord(e0, st);
    c->ev_open.push_back({e0, e1, flops});
}
static void prof_end(Ctx *c, hipStream_t st)
{
    if (!c->prof_on) return;
    hipEventRecord(c->ev_open.back().e1, st);
}

static bool conv(Ctx *c, ConvArgs a, hipStream_t st)
{
    bool dom = a.ksize == 3 && a.Cout % 128 == 0 && !a.relu_in;     // the conv_mfma_kernel<128,128,4,2,3,false> instantiation
    if (dom) {
        double H = (double)a.h, W = (double)a.w;
        prof_begin(c, 2.0 * a.N * H * W * 9.0 * a.Cin * a.Cout, st);
    }
    if (c->precision != TMAT_PRECISION_F32) {        // opt-in split precision: the same launch on the split copy of the weights
        auto &mp = c->wsplit[c->precision];
        auto it = mp.find(a.W);
        if (it != mp.end()) { a.W = it->second; a.prec = c->precision; }
    }
    bool ok = launch_conv(a, st);
    if (dom) prof_end(c, st);
    return ok;
}

// does down block bi (input resolution P / 2 >> bi) run on the wave-specialised fused separable kernel?
static bool down_block_ws(const Ctx *c, size_t bi)
{
    const DownBlock &d = c->down[bi];
    const int H = (c->patch / 2) >> bi;
    return c->fused_sep && sepconv_ws_supported(H, H, d.cin, d.cout) && sepconv_ws_supported(H, H, d.cout, d.cout) &&
           ((d.cout / 4) & (d.cout / 4 - 1)) == 0;
}

// Down path (memory-bound kernels + small pointwise GEMMs): X (n, P, P) -> dout (n, P/16, P/16, f_deep)
int unet_down_dev(Ctx *c, const float *X, int n, float *dout, hipStream_t s)
{
    if (n <= 0) return TMAT_OK;
    if (n > c->max_patches) { set_error("unet_down_dev: n > max_patches"); return TMAT_E_ARG; }
    const int P = c->patch;
    float *b0 = c->buf[0], *b1 = c->buf[1], *b2 = c->buf[2], *b3 = c->buf[3];
    const int di = dout == c->dout[1] ? 1 : 0;
    c->dout_relu_ok[di] = false;
    // Block 0 on the wave-specialised separable kernel: the stem tensor is never materialised -- the first separable convolution's
    // producers recompute it from the patch (sepconv_ws_kernel<..., STEM>), and the stride-2 residual convolution, which samples the
    // even pixels only, reads those from stem_even_kernel's quarter-size tensor as a unit-stride 1x1 convolution.
    const bool stem_in_sep = c->stem_fused && !c->down.empty() && down_block_ws(c, 0) && c->down[0].cin == c->f0 &&
                             !(c->precision == TMAT_PRECISION_BF16X3 && c->sep_bf16);
    if (stem_in_sep) launch_stem_even(X, n, P, P, c->stem_w, c->f0, c->stem_scale, c->stem_shift, b0, s);
    else launch_stem(X, n, P, P, c->stem_w, c->f0, c->stem_scale, c->stem_shift, b0, s);
    int H = P / 2;
    for (size_t bi = 0; bi < c->down.size(); bi++) {
        auto &d = c->down[bi];
        // prev = b0 (n, H, H, cin)
        if (down_block_ws(c, bi)) {
            // fused separable kernels (sepconv_ws_kernels.hip): depthwise producers + MFMA consumers in one workgroup.
            // bf16x3 mode: the pointwise contraction on the split weights (bf16x6 keeps these layers in f32: three planes do not fit the LDS budget)
            const float *pw0 = d.pw[0], *pw1 = d.pw[1];
            int sprec = 0;
            if (c->precision == TMAT_PRECISION_BF16X3 && c->sep_bf16) {
                auto &mp = c->wsplit[c->precision];
                auto i0 = mp.find(d.pw[0]), i1 = mp.find(d.pw[1]);
                if (i0 != mp.end() && i1 != mp.end()) { pw0 = i0->second; pw1 = i1->second; sprec = 1; }
            }
            const bool stem_here = bi == 0 && stem_in_sep;
            if (stem_here) {
                if (!launch_sepconv_ws_stem(X, n, H, H, d.cin, c->stem_w, c->stem_scale, c->stem_shift, d.dw[0], pw0, d.cout, d.scale[0], d.shift[0], 1, b2, s)) return TMAT_E_ARG;
            } else if (!launch_sepconv_ws(b0, n, H, H, d.cin, bi > 0, d.dw[0], pw0, d.cout, d.scale[0], d.shift[0], 1, b2, s, sprec)) return TMAT_E_ARG;
            ConvArgs r{};
            r.in = b0; r.N = n; r.h = H; r.w = H; r.Cin = d.cin; r.ksize = 1; r.stride = 2; r.W = d.res_w; r.Cout = d.cout;
            r.scale = nullptr; r.shift = d.res_b; r.out = b1;
            if (stem_here) { r.h = H / 2; r.w = H / 2; r.stride = 1; }      // b0 holds the even pixels only
            if (!conv(c, r, s)) return TMAT_E_ARG;
            float *nxt = bi + 1 == c->down.size() ? dout : b0;
            if (c->fused_pool) {
                if (!launch_sepconv_pool_ws(b2, n, H, H, d.cout, 0, d.dw[1], pw1, d.cout, d.scale[1], d.shift[1], 0, b3, b1, nxt, s, sprec)) return TMAT_E_ARG;
            } else {
                if (!launch_sepconv_ws(b2, n, H, H, d.cout, 0, d.dw[1], pw1, d.cout, d.scale[1], d.shift[1], 0, b3, s, sprec)) return TMAT_E_ARG;
                launch_maxpool_add(b3, n, H, H, d.cout, b1, nxt, s);
            }
            H /= 2;
            continue;
        }
        launch_dwconv(b0, n, H, H, d.cin, 1, d.dw[0], b1, s);
        ConvArgs a{};
        a.in = b1; a.N = n; a.h = H; a.w = H; a.Cin = d.cin; a.relu_in = 0; a.ksize = 1; a.stride = 1;
        a.W = d.pw[0]; a.Cout = d.cout; a.scale = d.scale[0]; a.shift = d.shift[0]; a.resid = nullptr; a.rs = 0;
        a.relu_out = 1; a.out = b2;
        if (!conv(c, a, s)) return TMAT_E_ARG;
        launch_dwconv(b2, n, H, H, d.cout, 0, d.dw[1], b3, s);
        a.in = b3; a.Cin = d.cout; a.W = d.pw[1]; a.scale = d.scale[1]; a.shift = d.shift[1]; a.relu_out = 0; a.out = b2;
        if (!conv(c, a, s)) return TMAT_E_ARG;
        ConvArgs r{};
        r.in = b0; r.N = n; r.h = H; r.w = H; r.Cin = d.cin; r.ksize = 1; r.stride = 2; r.W = d.res_w; r.Cout = d.cout;
        r.scale = nullptr; r.shift = d.res_b; r.out = b1;
        if (!conv(c, r, s)) return TMAT_E_ARG;
        const bool last = bi + 1 == c->down.size();
        float *dr = (last && c->relu_copy && dout == c->dout[di]) ? c->dout_relu[di] : nullptr;      // activated copy for the first up block
        launch_maxpool_add(b2, n, H, H, d.cout, b1, last ? dout : b0, s, dr);
        if (dr) c->dout_relu_ok[di] = true;
        H /= 2;
    }
    TMAT_HIP(hipGetLastError());
    return TMAT_OK;
}

// Up path (the MFMA-bound 3x3 transposed convolutions) + final conv: dout -> Y (n, P, P)
int unet_up_dev(Ctx *c, const float *dout, int n, float *Y, hipStream_t s)
{
    if (n <= 0) return TMAT_OK;
    // S = stored tensor at resolution Hs; logical block input = Up^up(S).  `dout` is never recycled.
    const float *S = dout;
    int Hs = c->patch >> (1 + c->down.size()), up = 0;
    float *t1 = c->ubuf[0], *rr = c->ubuf[1];
    int so_idx = 2;
    // S_act: the activated copy of S when its producer wrote one (the block's first convolution then needs no ReLU on load)
    const int di = dout == c->dout[1] ? 1 : 0;
    const float *S_act = (c->relu_copy && dout == c->dout[di] && c->dout_relu_ok[di]) ? c->dout_relu[di] : nullptr;
    for (size_t j = 0; j < c->up.size(); j++) {
        auto &u = c->up[j];
        float *so = c->ubuf[so_idx];
        ConvArgs a{};
        a.in = S_act ? S_act : S; a.N = n; a.h = Hs; a.w = Hs; a.Cin = u.cin; a.relu_in = S_act ? 0 : 1; a.ksize = 3; a.stride = 1;
        a.W = u.ct[0];
        if (up) { a.ksize = 2; a.W = u.ct_sub; }      // 4 taps per output parity class instead of 9
        a.Cout = u.cout; a.scale = u.scale[0]; a.shift = u.shift[0]; a.relu_out = 1; a.out = t1;
        if (!conv(c, a, s)) return TMAT_E_ARG;
        ConvArgs r{};
        r.in = S; r.N = n; r.h = Hs; r.w = Hs; r.Cin = u.cin; r.ksize = 1; r.stride = 1; r.W = u.res_w; r.Cout = u.cout;
        r.scale = nullptr; r.shift = u.res_b; r.out = rr;
        if (!conv(c, r, s)) return TMAT_E_ARG;
        const int Hl = Hs << up;
        ConvArgs b{};
        b.in = t1; b.N = n; b.h = Hl; b.w = Hl; b.Cin = u.cout; b.relu_in = 0; b.ksize = 3; b.stride = 1;
        b.W = u.ct[1]; b.Cout = u.cout; b.scale = u.scale[1]; b.shift = u.shift[1]; b.resid = rr; b.rs = up; b.relu_out = 0;
        b.out = so;
        // the next block's first convolution reads relu(so): written here as a second output (the last block's output feeds the final conv as is)
        float *so_act = (c->relu_copy && j + 1 < c->up.size()) ? c->urelu[j & 1] : nullptr;
        b.out_relu = so_act;
        if (!conv(c, b, s)) return TMAT_E_ARG;
        S = so; S_act = so_act; so_idx = so_idx == 2 ? 3 : 2;
        Hs = Hl; up = 1;
    }
    launch_final(S, n, Hs, Hs, c->f_last, c->final_w, c->final_b, Y, s);
    TMAT_HIP(hipGetLastError());
    return TMAT_OK;
}

// n may exceed max_patches (an image that needs more patches than the activation workspace holds): the patch list
// is then walked in chunks of max_patches, all on stream `s`
int unet_forward_dev(Ctx *c, const float *X, int n, float *Y, hipStream_t s)
{
    const size_t pp = (size_t)c->patch * c->patch;
    for (int i0 = 0; i0 < n; i0 += c->max_patches) {
        const int k = std::min(c->max_patches, n - i0);
        int rc = unet_down_dev(c, X + (size_t)i0 * pp, k, c->dout[0], s);
        if (rc) return rc;
        rc = unet_up_dev(c, c->dout[0], k, Y + (size_t)i0 * pp, s);
        if (rc) return rc;
    }
    return TMAT_OK;
}

// patch_in / patch_out hold max_patches patches; an image that needs more gets larger ones (2 x 0.41 MB per patch).
// Only called between entry points (nothing in flight uses the old buffers).
int ensure_patch_io(Ctx *c, int n_patches)
{
    if (n_patches <= c->patch_cap) return TMAT_OK;
    TMAT_HIP(hipStreamSynchronize(c->stream));
    TMAT_HIP(hipStreamSynchronize(c->stream2));
    if (c->patch_in) hipFree(c->patch_in);
    if (c->patch_in2) hipFree(c->patch_in2);
    if (c->patch_out) hipFree(c->patch_out);
    c->patch_in = c->patch_in2 = c->patch_out = nullptr;
    c->patch_cap = 0;
    const size_t bytes = (size_t)c->patch * c->patch * n_patches * sizeof(float);
    TMAT_HIP(hipMalloc((void **)&c->patch_in, bytes));
    TMAT_HIP(hipMalloc((void **)&c->patch_out, bytes));
    c->patch_cap = n_patches;
    return TMAT_OK;
}

// predict_img_with_smooth_windowing on device: x_dev (n, hh, ww) f32 -> pred_dev (n, hh, ww) f64
int predict_smooth_dev(Ctx *c, float *x_dev, int n, int hh, int ww, double *pred_dev)
{
    if (c->norm_on) launch_norm_f32(x_dev, (size_t)n * hh * ww, c->norm_mean, c->norm_std, c->stream);
    const int P = c->patch;
    TileGeom g = make_geom(hh, ww, P);
    const int per_pass = std::max(1, c->max_patches / g.tiles_per_img);
    { int rc = ensure_patch_io(c, per_pass * g.tiles_per_img); if (rc) return rc; }
    size_t need_pv = (size_t)std::min(n, per_pass) * 2 * sizeof(float);
    if (need_pv > c->scratch_bytes) { set_error("predict_smooth: scratch too small"); return TMAT_E_ARG; }
    float *mn = (float *)c->scratch, *mx = mn + std::min(n, per_pass);
    for (int i0 = 0; i0 < n; i0 += per_pass) {
        int k = std::min(per_pass, n - i0);
        const float *xi = x_dev + (size_t)i0 * hh * ww;
        launch_minmax_f32(xi, k, (size_t)hh * ww, mn, mx, c->stream);
        launch_extract_tiles(xi, mn, k, g, c->patch_in, c->stream);
        int rc = unet_forward_dev(c, c->patch_in, k * g.tiles_per_img, c->patch_out, c->stream);
        if (rc) return rc;
        launch_blend(c->patch_out, c->win1d, k, g, pred_dev + (size_t)i0 * hh * ww, c->stream);
    }
    TMAT_HIP(hipGetLastError());
    return TMAT_OK;
}

}  // namespace tmat

using namespace tmat;

extern "C" {

const char *tmat_last_error(void)
{
    static thread_local std::string copy;       // stays valid for the caller until its next call on this thread
    std::lock_guard<std::mutex> lk(g_err_mu);
    copy = g_err;
    return copy.c_str();
}
int tmat_version(void) { return 0x000100; }

// A handle without a model: device + stream only, for the entry points that need no weights (tmat_zproj_*,
// tmat_filter_edt_batch, tmat_finish_batch).  Model entry points refuse it.
int tmat_create_plain(int device_id, tmat_handle *out)
{
    if (!out) { set_error("tmat_create_plain: null argument"); return TMAT_E_ARG; }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("tmat_create_plain: no HIP device available (libtmat_hip has no CPU fallback)");
        return TMAT_E_HIP;
    }
    if (device_id < 0 || device_id >= ndev) { set_error("tmat_create_plain: bad device id"); return TMAT_E_ARG; }
    TMAT_HIP(hipSetDevice(device_id));
    Ctx *c = new Ctx();
    c->device = device_id;
    if (!hip_ok(hipStreamCreate(&c->stream), "hipStreamCreate")) { tmat_destroy((tmat_handle)c); return TMAT_E_HIP; }
    *out = (tmat_handle)c;
    return TMAT_OK;
}

int tmat_create(int device_id, const void *weights_blob, size_t n_bytes, int max_patches, tmat_handle *out)
{
    if (!out || !weights_blob) { set_error("tmat_create: null argument"); return TMAT_E_ARG; }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("tmat_create: no HIP device available (libtmat_hip has no CPU fallback)");
        return TMAT_E_HIP;
    }
    if (device_id < 0 || device_id >= ndev) { set_error("tmat_create: bad device id"); return TMAT_E_ARG; }
    TMAT_HIP(hipSetDevice(device_id));
    std::map<std::string, Tensor> m;
    int patch = 0;
    if (!parse_blob(weights_blob, n_bytes, m, patch)) return TMAT_E_WEIGHTS;
    Ctx *c = new Ctx();
    c->device = device_id;
    c->patch = patch;
    c->max_patches = max_patches > 0 ? max_patches : 400;
    if (const char *e = getenv("TMAT_FUSED_SEP")) c->fused_sep = atoi(e) != 0;
    if (const char *e = getenv("TMAT_FUSED_POOL")) c->fused_pool = atoi(e) != 0;
    if (const char *e = getenv("TMAT_STEM_FUSED")) c->stem_fused = atoi(e) != 0;
    if (const char *e = getenv("TMAT_SEP_BF16")) c->sep_bf16 = atoi(e) != 0;
    const char *prec_env = getenv("TMAT_PRECISION");
    if (prec_env && strcmp(prec_env, "f32") && strcmp(prec_env, "bf16x3") && strcmp(prec_env, "bf16x6")) {
        set_error("tmat_create: TMAT_PRECISION must be f32, bf16x3 or bf16x6");
        delete c;
        return TMAT_E_ARG;
    }
    if (const char *e = getenv("TMAT_DMT_DEVICE")) c->dmt_device = atoi(e) != 0;
    if (const char *e = getenv("TMAT_DMT_SWEEP_DEVICE")) c->dmt_sweep_device = atoi(e) != 0;
    if (const char *e = getenv("TMAT_PRE_STREAM")) c->pre_side = atoi(e) != 0;
    if (const char *e = getenv("TMAT_THIN_DEVICE")) c->thin_device = atoi(e) != 0;
    // the UNet stream gets the highest priority, the side stream of the post-processing stages (thinning rounds, finish,
    // DMT front end: many short launches that only have to be done before the next pass ends) the lowest: they fill the
    // gaps instead of taking workgroup slots from the MFMA kernels
    int prio_least = 0, prio_greatest = 0;
    hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
    if (!hip_ok(hipStreamCreateWithPriority(&c->stream, hipStreamDefault, prio_greatest), "hipStreamCreate") || !build_model(c, m)) { tmat_destroy((tmat_handle)c); return TMAT_E_WEIGHTS; }
    // activation workspace: per patch (P/2)^2 * f0 floats for buf0/buf1 and twice that for buf2/buf3
    const size_t unit = (size_t)(patch / 2) * (patch / 2) * c->f0;
    const size_t sizes[4] = {unit, unit, 2 * unit, 2 * unit};
    for (int i = 0; i < 4; i++) {
        c->buf_bytes[i] = sizes[i] * c->max_patches * sizeof(float);
        if (!hip_ok(hipMalloc((void **)&c->buf[i], c->buf_bytes[i]), "hipMalloc(activations)")) {
            tmat_destroy((tmat_handle)c); return TMAT_E_HIP;
        }
    }
    // up-path buffers: t1 (unit), hoisted residual (unit/4), block outputs alternate (unit/2, unit); down-path output x2
    const size_t usizes[4] = {unit, unit / 4, unit / 2, unit};
    for (int i = 0; i < 4; i++) {
        c->ubuf_bytes[i] = usizes[i] * c->max_patches * sizeof(float);
        if (!hip_ok(hipMalloc((void **)&c->ubuf[i], c->ubuf_bytes[i]), "hipMalloc(up buffers)")) {
            tmat_destroy((tmat_handle)c); return TMAT_E_HIP;
        }
    }
    const size_t dsz = (size_t)(patch >> (1 + c->down.size())) * (patch >> (1 + c->down.size())) * c->up[0].cin;
    c->dout_bytes = dsz * c->max_patches * sizeof(float);
    for (int i = 0; i < 2; i++)
        if (!hip_ok(hipMalloc((void **)&c->dout[i], c->dout_bytes), "hipMalloc(dout)")) {
            tmat_destroy((tmat_handle)c); return TMAT_E_HIP;
        }
    if (const char *e = getenv("TMAT_RELU_COPY")) c->relu_copy = atoi(e) != 0;
    if (c->relu_copy) {
        // up block j's output has (P/16 << j)^2 x cout_j values per patch: the larger of the even / odd blocks sizes each buffer
        size_t need[2] = {0, 0};
        for (size_t j = 0; j + 1 < c->up.size(); j++) {
            const size_t side = (size_t)(patch >> (1 + c->down.size())) << j;
            need[j & 1] = std::max(need[j & 1], side * side * c->up[j].cout);
        }
        for (int i = 0; i < 2; i++) {
            c->urelu_bytes[i] = need[i] * c->max_patches * sizeof(float);
            if (!hip_ok(hipMalloc((void **)&c->dout_relu[i], c->dout_bytes), "hipMalloc(dout_relu)") ||
                (need[i] && !hip_ok(hipMalloc((void **)&c->urelu[i], c->urelu_bytes[i]), "hipMalloc(urelu)"))) { tmat_destroy((tmat_handle)c); return TMAT_E_HIP; }
        }
    }
    if (!hip_ok(hipStreamCreate(&c->stream2), "hipStreamCreate(2)") || !hip_ok(hipStreamCreateWithPriority(&c->stream3, hipStreamDefault, prio_least), "hipStreamCreate(3)")) {
        tmat_destroy((tmat_handle)c); return TMAT_E_HIP;
    }
    for (int i = 0; i < 2; i++)
        if (!hip_ok(hipEventCreateWithFlags(&c->ev_down[i], hipEventDisableTiming), "hipEventCreate") ||
            !hip_ok(hipEventCreateWithFlags(&c->ev_up[i], hipEventDisableTiming), "hipEventCreate") ||
            !hip_ok(hipEventCreateWithFlags(&c->ev_pre[i], hipEventDisableTiming), "hipEventCreate") ||
            !hip_ok(hipEventCreateWithFlags(&c->ev_blend[i], hipEventDisableTiming), "hipEventCreate")) { tmat_destroy((tmat_handle)c); return TMAT_E_HIP; }
    const size_t pp = (size_t)patch * patch * c->max_patches * sizeof(float);
    c->scratch_bytes = 64 << 20;
    if (!hip_ok(hipMalloc((void **)&c->patch_in, pp), "hipMalloc(patch_in)") ||
        !hip_ok(hipMalloc((void **)&c->patch_out, pp), "hipMalloc(patch_out)") ||
        !hip_ok(hipMalloc((void **)&c->scratch, c->scratch_bytes), "hipMalloc(scratch)")) { tmat_destroy((tmat_handle)c); return TMAT_E_HIP; }
    c->patch_cap = c->max_patches;
    // squared-spline window (smooth_tiled_predictions.py:26-41), f64, on host then uploaded
    {
        const int ws = patch;
        std::vector<double> tri(ws), wind(ws);
        for (int i = 0; i < ws / 2; i++) { tri[i] = (2.0 * (i + 1) - 1.0) / ws; tri[ws - 1 - i] = tri[i]; }
        const int isec = ws / 4;
        double sum = 0;
        for (int i = 0; i < ws; i++) {
            double a2 = std::fabs(2 * tri[i]); double outer = (a2 * a2) / 2;
            if (i >= isec && i < ws - isec) outer = 0;
            double b2 = std::fabs(2 * (tri[i] - 1)); double inner = 1 - (b2 * b2) / 2;
            if (i < isec || i >= ws - isec) inner = 0;
            wind[i] = inner + outer;
        }
        // np.average = np.mean: pairwise summation in numpy; ws <= 8*128 blocks... restate numpy's
        // pairwise add.reduce for n < 8*PW_BLOCKSIZE? (n=320 > 128 -> pairwise halves)
        c->win_host = wind;
        sum = numpy_pairwise_sum(wind.data(), ws);
        double avg = sum / ws;
        for (int i = 0; i < ws; i++) wind[i] = wind[i] / avg;
        if (!hip_ok(hipMalloc((void **)&c->win1d, ws * sizeof(double)), "hipMalloc(win)") ||
            !hip_ok(hipMemcpy(c->win1d, wind.data(), ws * sizeof(double), hipMemcpyHostToDevice), "hipMemcpy(win)")) {
            tmat_destroy((tmat_handle)c); return TMAT_E_HIP;
        }
        c->win_host = wind;
    }
    *out = (tmat_handle)c;
    if (prec_env && strcmp(prec_env, "f32")) {
        const int rc = tmat_set_precision((tmat_handle)c, !strcmp(prec_env, "bf16x3") ? TMAT_PRECISION_BF16X3 : TMAT_PRECISION_BF16X6);
        if (rc) { tmat_destroy((tmat_handle)c); *out = nullptr; return rc; }
    }
    return TMAT_OK;
}

void tmat_destroy(tmat_handle h)
{
    Ctx *c = (Ctx *)h;
    if (!c) return;
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    for (void *p : c->owned) hipFree(p);
    if (c->stream2) hipStreamSynchronize(c->stream2);
    for (int i = 0; i < 4; i++) if (c->buf[i]) hipFree(c->buf[i]);
    for (int i = 0; i < 4; i++) if (c->ubuf[i]) hipFree(c->ubuf[i]);
    for (int i = 0; i < 2; i++) { if (c->dout_relu[i]) hipFree(c->dout_relu[i]); if (c->urelu[i]) hipFree(c->urelu[i]); }
    for (int i = 0; i < 2; i++) { if (c->dout[i]) hipFree(c->dout[i]); if (c->ev_down[i]) hipEventDestroy(c->ev_down[i]); if (c->ev_up[i]) hipEventDestroy(c->ev_up[i]); if (c->ev_pre[i]) hipEventDestroy(c->ev_pre[i]); if (c->ev_blend[i]) hipEventDestroy(c->ev_blend[i]); }
    if (c->stream2) hipStreamDestroy(c->stream2);
    if (c->stream3) { hipStreamSynchronize(c->stream3); hipStreamDestroy(c->stream3); }
    if (c->patch_in) hipFree(c->patch_in);
    if (c->patch_in2) hipFree(c->patch_in2);
    if (c->patch_out) hipFree(c->patch_out);
    if (c->scratch) hipFree(c->scratch);
    if (c->win1d) hipFree(c->win1d);
    if (c->ma_table) hipFree(c->ma_table);
    for (void *p : c->tool_ws) if (p) hipFree(p);
    for (auto &b : c->ws_pool) hipFree(b.second);
    for (auto &g : c->gauss_dev) hipFree(g.second);
    for (auto &m : c->resnets) for (void *p : m.owned) hipFree(p);
    c->free_pass();
    for (auto &e : c->ev_open) { hipEventDestroy(e.e0); hipEventDestroy(e.e1); }
    for (auto e : c->ev_pool) hipEventDestroy(e);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

int tmat_sync(tmat_handle h)
{
    Ctx *c = (Ctx *)h;
    if (!c) { set_error("null handle"); return TMAT_E_ARG; }
    if (c->stream2) TMAT_HIP(hipStreamSynchronize(c->stream2));
    if (c->stream3) TMAT_HIP(hipStreamSynchronize(c->stream3));
    TMAT_HIP(hipStreamSynchronize(c->stream));
    return TMAT_OK;
}

int tmat_unet_predict(tmat_handle h, const float *x, int n, float *y)
{
    Ctx *c = (Ctx *)h;
    if (c && !has_model(c)) { set_error("tmat_unet_predict: this handle has no model (tmat_create_plain)"); return TMAT_E_ARG; }
    if (!c || !x || !y || n < 0) { set_error("tmat_unet_predict: bad argument"); return TMAT_E_ARG; }
    TMAT_HIP(hipSetDevice(c->device));
    const size_t per = (size_t)c->patch * c->patch;
    for (int i0 = 0; i0 < n; i0 += c->max_patches) {
        int k = std::min(c->max_patches, n - i0);
        TMAT_HIP(hipMemcpyAsync(c->patch_in, x + i0 * per, k * per * sizeof(float), hipMemcpyHostToDevice, c->stream));
        int rc = unet_forward_dev(c, c->patch_in, k, c->patch_out, c->stream);
        if (rc) return rc;
        TMAT_HIP(hipMemcpyAsync(y + i0 * per, c->patch_out, k * per * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        TMAT_HIP(hipStreamSynchronize(c->stream));
    }
    return TMAT_OK;
}

int tmat_predict_smooth(tmat_handle h, const float *x, int n, int hh, int ww, double *pred)
{
    Ctx *c = (Ctx *)h;
    if (c && !has_model(c)) { set_error("tmat_predict_smooth: this handle has no model (tmat_create_plain)"); return TMAT_E_ARG; }
    if (!c || !x || !pred || n < 0 || hh <= 0 || ww <= 0) { set_error("tmat_predict_smooth: bad argument"); return TMAT_E_ARG; }
    if (n == 0) return TMAT_OK;
    TMAT_HIP(hipSetDevice(c->device));
    const size_t per = (size_t)hh * ww;
    float *xd = nullptr; double *pd = nullptr;
    TMAT_HIP(hipMalloc((void **)&xd, n * per * sizeof(float)));
    if (!hip_ok(hipMalloc((void **)&pd, n * per * sizeof(double)), "hipMalloc(pred)")) { hipFree(xd); return TMAT_E_HIP; }
    int rc = TMAT_OK;
    if (!hip_ok(hipMemcpyAsync(xd, x, n * per * sizeof(float), hipMemcpyHostToDevice, c->stream), "H2D")) rc = TMAT_E_HIP;
    if (!rc) rc = predict_smooth_dev(c, xd, n, hh, ww, pd);
    if (!rc && !hip_ok(hipMemcpyAsync(pred, pd, n * per * sizeof(double), hipMemcpyDeviceToHost, c->stream), "D2H")) rc = TMAT_E_HIP;
    if (!hip_ok(hipStreamSynchronize(c->stream), "sync") && !rc) rc = TMAT_E_HIP;
    hipFree(xd); hipFree(pd);
    return rc;
}

int tmat_dev_alloc(tmat_handle h, size_t bytes, void **dev_ptr)
{
    Ctx *c = (Ctx *)h;
    if (!c || !dev_ptr) { set_error("tmat_dev_alloc: bad argument"); return TMAT_E_ARG; }
    TMAT_HIP(hipSetDevice(c->device));
    TMAT_HIP(hipMalloc(dev_ptr, bytes));
    return TMAT_OK;
}
int tmat_dev_free(tmat_handle h, void *dev_ptr)
{
    Ctx *c = (Ctx *)h;
    if (!c) { set_error("null handle"); return TMAT_E_ARG; }
    TMAT_HIP(hipSetDevice(c->device));
    TMAT_HIP(hipFree(dev_ptr));
    return TMAT_OK;
}
int tmat_dev_upload(tmat_handle h, void *dev_dst, const void *host_src, size_t bytes)
{
    Ctx *c = (Ctx *)h;
    if (!c || !dev_dst || !host_src) { set_error("tmat_dev_upload: bad argument"); return TMAT_E_ARG; }
    TMAT_HIP(hipSetDevice(c->device));
    TMAT_HIP(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, c->stream));
    TMAT_HIP(hipStreamSynchronize(c->stream));
    return TMAT_OK;
}

int tmat_set_input_depth(tmat_handle h, int bits)
{
    Ctx *c = (Ctx *)h;
    if (!c || (bits != 8 && bits != 16)) { set_error("tmat_set_input_depth: bits must be 8 or 16"); return TMAT_E_ARG; }
    c->input_sat = bits == 8 ? 255.f : 65535.f;
    return TMAT_OK;
}

int tmat_set_input_norm(tmat_handle h, int on, double mean, double sd)
{
    Ctx *c = (Ctx *)h;
    if (!c || (on && !(sd != 0.0))) { set_error("tmat_set_input_norm: bad argument (norm_std must not be 0)"); return TMAT_E_ARG; }
    c->norm_on = on != 0;
    c->norm_mean = (float)mean;
    c->norm_std = (float)sd;
    return TMAT_OK;
}

int tmat_set_precision(tmat_handle h, int mode)
{
    Ctx *c = (Ctx *)h;
    if (!c || !has_model(c) || (mode != TMAT_PRECISION_F32 && mode != TMAT_PRECISION_BF16X3 && mode != TMAT_PRECISION_BF16X6)) {
        set_error("tmat_set_precision: needs a model handle and mode TMAT_PRECISION_F32, TMAT_PRECISION_BF16X3 or TMAT_PRECISION_BF16X6");
        return TMAT_E_ARG;
    }
    TMAT_HIP(hipSetDevice(c->device));
    if (mode != TMAT_PRECISION_F32) {
        auto &mp = c->wsplit[mode];
        // bf16x6 keeps the fused separable layers in f32 (three planes do not fit their LDS budget): no split copy of their pointwise weights
        std::vector<const float *> skip;
        if (mode == TMAT_PRECISION_BF16X6)
            for (size_t bi = 0; bi < c->down.size(); bi++)
                if (down_block_ws(c, bi)) { skip.push_back(c->down[bi].pw[0]); skip.push_back(c->down[bi].pw[1]); }
        for (auto &kv : c->conv_w_host) {
            if (mp.count(kv.first) || std::find(skip.begin(), skip.end(), kv.first) != skip.end()) continue;
            float *dev = nullptr;
            if (!upload(c, split_bf16(kv.second.w, mode + 1), &dev)) return TMAT_E_HIP;
            mp[kv.first] = dev;
        }
    }
    { const int rc = tmat_sync(h); if (rc) return rc; }      // all three streams: nothing in flight reads the old mode's weights
    c->precision = mode;
    return TMAT_OK;
}

// Test-only (tests/test_gpu_poison.py): fill every scratch workspace of the handle -- the activation ping-pong sets, the pooled-tile
// strips that live in them, patch_in / patch_out, the scratch block and every per-pass device / pinned buffer -- with `byte_pattern`
// (0xFF: NaN as f32 / f64, -1 as int).  Weights, the spline window and the Lanczos tables are constants and stay.  A forward or a pass
// that reads a location it has not written in the same call then produces NaNs or a mismatch against the oracle deterministically,
// instead of depending on what a recycled allocation happens to hold.
int tmat_debug_poison(tmat_handle h, int byte_pattern)
{
    Ctx *c = (Ctx *)h;
    if (!c) { set_error("null handle"); return TMAT_E_ARG; }
    TMAT_HIP(hipSetDevice(c->device));
    { const int rc = tmat_sync(h); if (rc) return rc; }
    std::vector<WsEnt> all;
    for (int i = 0; i < 4; i++) { if (c->buf[i]) all.push_back({c->buf[i], c->buf_bytes[i], false}); if (c->ubuf[i]) all.push_back({c->ubuf[i], c->ubuf_bytes[i], false}); }
    for (int i = 0; i < 2; i++) if (c->dout[i]) all.push_back({c->dout[i], c->dout_bytes, false});
    for (int i = 0; i < 2; i++) { if (c->dout_relu[i]) all.push_back({c->dout_relu[i], c->dout_bytes, false}); if (c->urelu[i]) all.push_back({c->urelu[i], c->urelu_bytes[i], false}); }
    const size_t pio = (size_t)c->patch * c->patch * c->patch_cap * sizeof(float);
    if (c->patch_in) all.push_back({c->patch_in, pio, false});
    if (c->patch_in2) all.push_back({c->patch_in2, pio, false});
    if (c->patch_out) all.push_back({c->patch_out, pio, false});
    if (c->scratch) all.push_back({c->scratch, c->scratch_bytes, false});
    for (int i = 0; i < Ctx::N_TOOL_WS; i++) if (c->tool_ws[i]) all.push_back({c->tool_ws[i], c->tool_ws_bytes[i], false});
    for (auto &b : c->ws_pool) all.push_back({b.second, b.first, false});
    all.insert(all.end(), c->pass.ws.begin(), c->pass.ws.end());
    for (const WsEnt &e : all) {
        if (e.host) memset(e.p, byte_pattern, e.bytes);
        else TMAT_HIP(hipMemsetAsync(e.p, byte_pattern, e.bytes, c->stream));
    }
    TMAT_HIP(hipStreamSynchronize(c->stream));
    return TMAT_OK;
}

int tmat_prof_enable(tmat_handle h, int on)
{
    Ctx *c = (Ctx *)h;
    if (!c) { set_error("null handle"); return TMAT_E_ARG; }
    c->prof_on = on != 0;
    return TMAT_OK;
}
int tmat_prof_read(tmat_handle h, double *ms, int64_t *launches, double *flops, int reset)
{
    Ctx *c = (Ctx *)h;
    if (!c) { set_error("null handle"); return TMAT_E_ARG; }
    TMAT_HIP(hipSetDevice(c->device));
    TMAT_HIP(hipStreamSynchronize(c->stream));
    for (auto &e : c->ev_open) {
        float t = 0;
        if (hipEventElapsedTime(&t, e.e0, e.e1) == hipSuccess) { c->prof_ms += t; c->prof_launches++; c->prof_flops += e.flops; }
        c->ev_pool.push_back(e.e0); c->ev_pool.push_back(e.e1);
    }
    c->ev_open.clear();
    if (ms) *ms = c->prof_ms;
    if (launches) *launches = c->prof_launches;
    if (flops) *flops = c->prof_flops;
    if (reset) { c->prof_ms = 0; c->prof_launches = 0; c->prof_flops = 0; }
    return TMAT_OK;
}

}  // extern "C"
