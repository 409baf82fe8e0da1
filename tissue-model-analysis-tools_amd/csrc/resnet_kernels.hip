// Invasion-depth tool on gfx950 (reference scripts/compute_inv_depth.py:96-172, fl_tissue_model_tools/models.py:33-82,
// data_prep.py:17-61): per Z slice a ResNet50 trunk up to conv4_block6_out (keras.applications.resnet50), global average
// pooling, one dense unit, sigmoid; an ensemble of such models votes by the mean probability.
//
// The trunk is 1x1 and 3x3 convolutions with folded BatchNormalization: they run on the f32-MFMA implicit-GEMM kernel of the
// branching path (unet_kernels.hip: conv_mfma_kernel, 1x1 stride 1 / 2, 3x3 SAME, residual add + ReLU in the epilogue), so the
// two tools share their hot kernel and its arithmetic contract (oracle/unet_exact.c:orc_conv).  New here: the data
// preparation (bilinear resize, rescale to 0..255, caffe-style mean subtraction), the 7x7 stride-2 stem, the 3x3 stride-2
// max-pool and the head.  Keras details: every Conv2D has a bias, BatchNormalization eps = 1.001e-5, ZeroPadding2D(3) before
// the stem convolution and ZeroPadding2D(1) before the pool (both 'valid'), the stride of a stage sits in the first 1x1
// convolution of its first block and in that block's projection shortcut.
// The 7x7 stride-2 stem runs on the same MFMA kernel (round 4): resnet_im2col_kernel lays the 147 taps of every output pixel out as
// a 192-channel pixel (k = (ky 7 + kx) 3 + c, zeros outside the image and for k >= 147: six K chunks of 32) and the convolution
// becomes a 1x1 layer with K = 192; round 3's direct kernel (one thread per output, 147 LDS + 147 cached global reads each) took
// 18 % of the tool's GPU time at 12 TFLOP/s.
// Arithmetic of the new kernels (shared with oracle/resnet.py): float32, multiply and add kept separate (no FMA) in pool and
// head; stem: orc_conv's chain over k (the MFMA order) on the im2col tensor, fmaf(acc, scale, shift), ReLU; head: per channel sum over the 256 pixels in raster order, times 1/256, dot product over
// the channels in order, plus bias, 1 / (1 + exp_det(-z)).
#include "../../include/tmat.h"
#include "tmat_ctx.h"

#include <cmath>
#include <cstring>

namespace tmat {

// (N, S, S) u16 slices (already resized) -> (N, S, S, 3) f32: rescale_intensity(img, (0, 255)) in f64 (per image), the three
// identical channels minus the caffe means of B, G, R (resnet50.preprocess_input), cast to f32 (tf.convert_to_tensor)
__global__ __launch_bounds__(256) void inv_prep_kernel(const uint16_t *__restrict__ img, int npx, const int *__restrict__ mn, const int *__restrict__ mx,
                                                       float *__restrict__ out)
{
    const int n = blockIdx.y;
    const double lo = (double)mn[n], hi = (double)mx[n];
    for (int p = blockIdx.x * 256 + threadIdx.x; p < npx; p += gridDim.x * 256) {
        const double x = (double)img[(size_t)n * npx + p];
        double g = fmin(fmax(x, lo), hi);
        g = lo != hi ? ((g - lo) / (hi - lo)) * 255.0 + 0.0 : fmin(fmax(g, 0.0), 255.0);
        float *o = out + ((size_t)n * npx + p) * 3;
        o[0] = (float)(g - 103.939); o[1] = (float)(g - 116.779); o[2] = (float)(g - 123.68);
    }
}
__global__ __launch_bounds__(256) void minmax_u16_img_kernel(const uint16_t *__restrict__ img, int npx, int *__restrict__ mn, int *__restrict__ mx)
{
    const uint16_t *p = img + (size_t)blockIdx.x * npx;
    int lo = 65535, hi = 0;
    for (int i = threadIdx.x; i < npx; i += 256) { const int v = p[i]; lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
    for (int o = 32; o > 0; o >>= 1) { const int l2 = __shfl_down(lo, o), h2 = __shfl_down(hi, o); lo = l2 < lo ? l2 : lo; hi = h2 > hi ? h2 : hi; }
    __shared__ int sl[4], sh[4];
    if ((threadIdx.x & 63) == 0) { sl[threadIdx.x >> 6] = lo; sh[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; i++) { lo = sl[i] < lo ? sl[i] : lo; hi = sh[i] > hi ? sh[i] : hi; }
        mn[blockIdx.x] = lo; mx[blockIdx.x] = hi;
    }
}

// conv1: ZeroPadding2D(3) + Conv2D(64, 7, strides 2, valid) + BN + ReLU, as im2col + a 1x1 convolution on conv_mfma_kernel.
// x (N, S, S, 3) -> col (N, S/2, S/2, 192): col[k] = x[2 yo + ky - 3][2 xo + kx - 3][c] for k = (ky 7 + kx) 3 + c < 147 (0 outside the
// image: the zero padding), 0 for 147 <= k < 192.  One thread = 4 consecutive k of one output pixel (one 16-byte store).
constexpr int STEM_TAPS = 147, STEM_K = 192;
__global__ __launch_bounds__(256) void resnet_im2col_kernel(const float *__restrict__ x, int S, float *__restrict__ col, size_t total)
{
    const int So = S >> 1;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int q = (int)(e % (STEM_K / 4));
        const size_t pix = e / (STEM_K / 4);
        const int xo = (int)(pix % So), yo = (int)((pix / So) % So);
        const size_t n = pix / ((size_t)So * So);
        const float *xi = x + n * S * S * 3;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int k = 4 * q + j;
            const int tap = k / 3, c = k - tap * 3;
            const int ky = tap / 7, kx = tap - ky * 7;
            const int iy = 2 * yo + ky - 3, ix = 2 * xo + kx - 3;
            const bool in = k < STEM_TAPS && iy >= 0 && iy < S && ix >= 0 && ix < S;
            v[j] = in ? xi[((size_t)iy * S + ix) * 3 + c] : 0.0f;
        }
        *reinterpret_cast<float4 *>(col + e * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// pool1: ZeroPadding2D(1) + MaxPooling2D(3, strides 2, valid): (N, S, S, C) -> (N, S/2, S/2, C); the padding is ZEROS
__global__ __launch_bounds__(256) void resnet_pool_kernel(const float *__restrict__ x, int S, int C, float *__restrict__ out, size_t total)
{
    const int So = S >> 1;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int c = (int)(e % C);
        const size_t q = e / C;
        const int xo = (int)(q % So), yo = (int)((q / So) % So);
        const size_t n = q / ((size_t)So * So);
        float m = -INFINITY;
        for (int ky = 0; ky < 3; ky++)
            for (int kx = 0; kx < 3; kx++) {
                const int iy = 2 * yo + ky - 1, ix = 2 * xo + kx - 1;
                const float v = (iy >= 0 && iy < S && ix >= 0 && ix < S) ? x[((n * S + iy) * S + ix) * C + c] : 0.0f;
                m = v > m ? v : m;
            }
        out[e] = m;
    }
}

__device__ __forceinline__ float exp_det_r(float x)        // the deterministic expf of unet_kernels.hip / oracle/unet_exact.c
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

// GlobalAveragePooling2D + Dense(1) + sigmoid: feat (N, P, C) -> prob (N); one block per image, thread = channel (C <= 1024)
__global__ __launch_bounds__(1024) void resnet_head_kernel(const float *__restrict__ feat, int P, int C, const float *__restrict__ w, float b, float *__restrict__ prob)
{
    __shared__ float sm[1024];
    const int c = threadIdx.x;
    if (c < C) {
        const float *f = feat + (size_t)blockIdx.x * P * C + c;
        float s = 0.0f;
        for (int p = 0; p < P; p++) s = s + f[(size_t)p * C];
        sm[c] = s * (1.0f / (float)P);
    }
    __syncthreads();
    if (c == 0) {
        float z = 0.0f;
        for (int k = 0; k < C; k++) z = z + sm[k] * w[k];
        z = z + b;
        prob[blockIdx.x] = 1.0f / (1.0f + exp_det_r(-z));
    }
}

static bool up(ResNetModel &m, const std::vector<float> &v, float **dev)
{
    if (!hip_ok(hipMalloc((void **)dev, v.size() * sizeof(float)), "hipMalloc(resnet weights)")) return false;
    m.owned.push_back(*dev);
    return hip_ok(hipMemcpy(*dev, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice), "hipMemcpy(resnet weights)");
}
// scale = f32(gamma / sqrt(var + 1.001e-5)); shift = f32(beta + (bias - mean) * scale_f64)
static void fold_bn_keras(const Tensor &bn, const Tensor &bias, std::vector<float> &scale, std::vector<float> &shift)
{
    const int C = bn.shape[1];
    scale.resize(C); shift.resize(C);
    for (int i = 0; i < C; i++) {
        const double g = bn.data[i], b = bn.data[C + i], mu = bn.data[2 * C + i], v = bn.data[3 * C + i];
        const double sd = g / std::sqrt(v + 1.001e-5);
        scale[i] = (float)sd;
        shift[i] = (float)(b + ((double)bias.data[i] - mu) * sd);
    }
}
static bool load_conv(ResNetModel &m, const std::map<std::string, Tensor> &t, const std::string &name, int ksize, int stride, int cin_expect, ResConv &c)
{
    auto w = t.find(name + ".w"), b = t.find(name + ".b"), bn = t.find(name + ".bn");
    if (w == t.end() || b == t.end() || bn == t.end()) { set_error("resnet weights: missing " + name); return false; }
    const Tensor &W = w->second;
    if (W.shape.size() != 4 || W.shape[0] != ksize || W.shape[1] != ksize || W.shape[2] != cin_expect || W.shape[3] % 64 || cin_expect % 32) {
        set_error("resnet weights: bad shape of " + name + ".w");
        return false;
    }
    c.cin = W.shape[2]; c.cout = W.shape[3]; c.ksize = ksize; c.stride = stride;
    if (bn->second.shape.size() != 2 || bn->second.shape[0] != 4 || bn->second.shape[1] != c.cout || (int)b->second.count != c.cout) {
        set_error("resnet weights: BatchNormalization / bias of " + name + " does not match");
        return false;
    }
    std::vector<float> sc, sh;
    fold_bn_keras(bn->second, b->second, sc, sh);
    return up(m, k_contiguous(W.data, ksize * ksize, c.cin, c.cout), &c.w) && up(m, sc, &c.scale) && up(m, sh, &c.shift);
}

static bool run_conv(const ResConv &c, const float *in, int N, int h, const float *resid, int relu, float *out, hipStream_t s)
{
    ConvArgs a{};
    a.in = in; a.N = N; a.h = h; a.w = h; a.Cin = c.cin; a.relu_in = 0; a.ksize = c.ksize; a.stride = c.stride; a.W = c.w; a.Cout = c.cout;
    a.scale = c.scale; a.shift = c.shift; a.resid = resid; a.rs = 0; a.relu_out = relu; a.out = out;
    return launch_conv(a, s);
}

// x (N, S, S, 3) f32 on the device -> prob (N) on the device; bufs: 4 activation buffers of N * (S/2)^2 * 64 floats; col: N * (S/2)^2 * 192
static int resnet_forward_dev(const ResNetModel &m, const float *x, int N, int S, float *const bufs[4], float *col, float *prob, hipStream_t s)
{
    float *a = bufs[0], *b = bufs[1], *t1 = bufs[2], *t2 = bufs[3];
    const int S2 = S / 2, S4 = S / 4;
    {
        const size_t quads = (size_t)N * S2 * S2 * (STEM_K / 4);
        hipLaunchKernelGGL(resnet_im2col_kernel, dim3((unsigned)std::min<size_t>((quads + 255) / 256, 1u << 20)), dim3(256), 0, s, x, S, col, quads);
        ConvArgs st{};
        st.in = col; st.N = N; st.h = S2; st.w = S2; st.Cin = STEM_K; st.relu_in = 0; st.ksize = 1; st.stride = 1; st.W = m.stem_w; st.Cout = 64;
        st.scale = m.stem_scale; st.shift = m.stem_shift; st.resid = nullptr; st.rs = 0; st.relu_out = 1; st.out = a;
        if (!launch_conv(st, s)) return TMAT_E_ARG;
    }
    const size_t ptotal = (size_t)N * S4 * S4 * 64;
    hipLaunchKernelGGL(resnet_pool_kernel, dim3((unsigned)std::min<size_t>((ptotal + 255) / 256, 16384)), dim3(256), 0, s, a, S2, 64, b, ptotal);
    float *cur = b, *nxt = a;
    int h = S4;
    for (const ResBlock &k : m.blocks) {
        const int ho = h / k.c1.stride;
        const float *shortcut = cur;
        if (k.has_sc) {
            if (!run_conv(k.sc, cur, N, h, nullptr, 0, t2, s)) return TMAT_E_ARG;
            shortcut = t2;
        }
        if (!run_conv(k.c1, cur, N, h, nullptr, 1, t1, s)) return TMAT_E_ARG;
        if (!run_conv(k.c2, t1, N, ho, nullptr, 1, nxt, s)) return TMAT_E_ARG;
        // c3 writes over t1 (its input is nxt), then the roles rotate: out -> cur
        if (!run_conv(k.c3, nxt, N, ho, shortcut, 1, t1, s)) return TMAT_E_ARG;
        float *old = cur;
        cur = t1; t1 = old;
        h = ho;
    }
    if (m.feat > 1024) { set_error("resnet: head supports at most 1024 channels"); return TMAT_E_ARG; }
    hipLaunchKernelGGL(resnet_head_kernel, dim3(N), dim3(1024), 0, s, cur, h * h, m.feat, m.fc_w, m.fc_b, prob);
    return hipGetLastError() == hipSuccess ? TMAT_OK : TMAT_E_HIP;
}

int launch_resize_linear_dev(const uint16_t *din, int n, int H, int W, int oh, int ow, bool eight_bit, int *tab, uint16_t *dsm, hipStream_t s);      // cellarea_kernels.hip

}  // namespace tmat

using namespace tmat;

extern "C" {

int tmat_resnet_load(tmat_handle hd, const void *weights_blob, size_t n_bytes, int *model_id)
{
    Ctx *c = (Ctx *)hd;
    if (!c || !weights_blob || !model_id) { set_error("tmat_resnet_load: bad argument"); return TMAT_E_ARG; }
    TMAT_HIP(hipSetDevice(c->device));
    std::map<std::string, Tensor> t;
    int patch = 0;
    if (!parse_blob(weights_blob, n_bytes, t, patch)) return TMAT_E_WEIGHTS;
    ResNetModel m;
    auto fail = [&]() { for (void *p : m.owned) hipFree(p); return TMAT_E_WEIGHTS; };
    // stem: BN folded with the convolution's bias
    auto sw = t.find("conv1.w"), sb = t.find("conv1.b"), sbn = t.find("conv1.bn");
    if (sw == t.end() || sb == t.end() || sbn == t.end() || sw->second.shape != std::vector<int>({7, 7, 3, 64}) || sbn->second.shape != std::vector<int>({4, 64}) ||
        sb->second.count != 64) { set_error("resnet weights: conv1 missing or malformed"); return fail(); }
    {
        std::vector<float> sc, sh;
        fold_bn_keras(sbn->second, sb->second, sc, sh);
        // (7, 7, 3, 64) = [147][64] as stored -> [64][192], k contiguous (conv_mfma_kernel's weight layout), zeros for k >= 147
        std::vector<float> wk((size_t)64 * STEM_K, 0.0f);
        for (int k = 0; k < STEM_TAPS; k++) for (int o = 0; o < 64; o++) wk[(size_t)o * STEM_K + k] = sw->second.data[(size_t)k * 64 + o];
        if (!up(m, wk, &m.stem_w) || !up(m, sc, &m.stem_scale) || !up(m, sh, &m.stem_shift)) return fail();
    }
    int cin = 64;
    for (int stage = 2;; stage++) {
        if (!t.count("s" + std::to_string(stage) + "b1.c1.w")) break;
        for (int blk = 1;; blk++) {
            const std::string p = "s" + std::to_string(stage) + "b" + std::to_string(blk);
            if (!t.count(p + ".c1.w")) break;
            ResBlock k;
            const int stride = (blk == 1 && stage > 2) ? 2 : 1;
            k.has_sc = blk == 1;
            if (!load_conv(m, t, p + ".c1", 1, stride, cin, k.c1) || !load_conv(m, t, p + ".c2", 3, 1, k.c1.cout, k.c2) || !load_conv(m, t, p + ".c3", 1, 1, k.c2.cout, k.c3)) return fail();
            if (k.has_sc) { if (!load_conv(m, t, p + ".c0", 1, stride, cin, k.sc) || k.sc.cout != k.c3.cout) { set_error("resnet weights: bad shortcut " + p); return fail(); } }
            else if (k.c3.cout != cin) { set_error("resnet weights: identity block changes the channel count: " + p); return fail(); }
            cin = k.c3.cout;
            m.blocks.push_back(k);
        }
    }
    auto fw = t.find("fc.w"), fb = t.find("fc.b");
    if (m.blocks.empty() || fw == t.end() || fb == t.end() || (int)fw->second.count != cin || fb->second.count != 1) { set_error("resnet weights: head missing or malformed"); return fail(); }
    m.feat = cin;
    m.fc_b = fb->second.data[0];
    if (!up(m, std::vector<float>(fw->second.data, fw->second.data + cin), &m.fc_w)) return fail();
    c->resnets.push_back(std::move(m));
    *model_id = (int)c->resnets.size() - 1;
    return TMAT_OK;
}

int tmat_resnet_predict(tmat_handle hd, int model_id, const float *x, int n, int size, float *prob)
{
    Ctx *c = (Ctx *)hd;
    if (!c || !x || !prob || n < 0 || size < 32 || size % 32 || model_id < 0 || model_id >= (int)c->resnets.size()) { set_error("tmat_resnet_predict: bad argument"); return TMAT_E_ARG; }
    if (n == 0) return TMAT_OK;
    TMAT_HIP(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const size_t nx = (size_t)n * size * size * 3, nb = (size_t)n * (size / 2) * (size / 2) * 64;
    float *dx = nullptr, *dp = nullptr, *col = nullptr, *bufs[4] = {nullptr, nullptr, nullptr, nullptr};
    int rc = TMAT_OK;
    if (!hip_ok(hipMalloc((void **)&dx, nx * 4), "hipMalloc") || !hip_ok(hipMalloc((void **)&dp, n * 4), "hipMalloc") ||
        !hip_ok(hipMalloc((void **)&col, nb * 3 * 4), "hipMalloc")) rc = TMAT_E_HIP;      // 192 = 3 x 64 values per stem output pixel
    for (int i = 0; i < 4 && !rc; i++) if (!hip_ok(hipMalloc((void **)&bufs[i], nb * 4), "hipMalloc")) rc = TMAT_E_HIP;
    if (!rc && !hip_ok(hipMemcpyAsync(dx, x, nx * 4, hipMemcpyHostToDevice, s), "H2D")) rc = TMAT_E_HIP;
    if (!rc) rc = resnet_forward_dev(c->resnets[model_id], dx, n, size, bufs, col, dp, s);
    if (!rc && (!hip_ok(hipMemcpyAsync(prob, dp, n * 4, hipMemcpyDeviceToHost, s), "D2H") || !hip_ok(hipStreamSynchronize(s), "sync"))) rc = TMAT_E_HIP;
    hipFree(dx); hipFree(dp); hipFree(col);
    for (float *b : bufs) hipFree(b);
    return rc;
}

}  // extern "C"

// stacks[k] (Zs[k], H, W) u16 host, k < n_stacks: uploaded back to back (Z = their sum) -- every step of the tool is per slice
static int inv_depth_impl(tmat_handle hd, const int *model_ids, int n_models, const uint16_t *const *stacks, const int *Zs, int n_stacks, int Z, int H, int W,
                          int size, float *probs, float *x_out)
{
    Ctx *c = (Ctx *)hd;
    for (int i = 0; i < n_models; i++) if (model_ids[i] < 0 || model_ids[i] >= (int)c->resnets.size()) { set_error("tmat_inv_depth_predict: unknown model id"); return TMAT_E_ARG; }
    if (Z == 0) return TMAT_OK;
    TMAT_HIP(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int CH = 128;                                    // slices per forward: M = 128 x 16^2 rows at the deepest stage (round 2: 32, launch bound)
    const size_t npx = (size_t)size * size;
    uint16_t *din = nullptr, *dsm = nullptr;
    int *itab = nullptr, *mnmx = nullptr;
    float *dx = nullptr, *dp = nullptr, *col = nullptr, *bufs[4] = {nullptr, nullptr, nullptr, nullptr};
    // cv2.resize(img, img_hw, cv2.INTER_LANCZOS4) (data_prep.py:36): the third positional parameter is `dst`: bilinear
    int rc = TMAT_OK;
    const int nb = std::min(Z, CH);
    // workspaces live on the handle between calls (tmat_ctx.h:ws_get): slots 12..22 of this tool
    din = (uint16_t *)ws_get(c, 12, (size_t)Z * H * W * 2); dsm = (uint16_t *)ws_get(c, 13, (size_t)Z * npx * 2);
    itab = (int *)ws_get(c, 14, (size_t)size * 2 * 4 * 4); mnmx = (int *)ws_get(c, 15, (size_t)Z * 2 * 4);
    dx = (float *)ws_get(c, 16, (size_t)Z * npx * 3 * 4); dp = (float *)ws_get(c, 17, (size_t)Z * n_models * 4);
    for (int i = 0; i < 4; i++) bufs[i] = (float *)ws_get(c, 18 + i, (size_t)nb * (size / 2) * (size / 2) * 64 * 4);
    col = (float *)ws_get(c, 22, (size_t)nb * (size / 2) * (size / 2) * STEM_K * 4);
    if (!din || !dsm || !itab || !mnmx || !dx || !dp || !bufs[0] || !bufs[1] || !bufs[2] || !bufs[3] || !col) rc = TMAT_E_HIP;
    if (!rc) {
        // cv::resize takes INTER_AREA's integer mean for an exact halving on both axes (a 512 x 512 slice at the configured 256 x 256);
        // 8-bit sources (tmat_set_input_depth(h, 8)) take its fixed-point bilinear arithmetic
        bool ok = true;
        {
            size_t z0 = 0;
            for (int k = 0; k < n_stacks && ok; k++) {
                ok = hipMemcpyAsync(din + z0 * H * W, stacks[k], (size_t)Zs[k] * H * W * 2, hipMemcpyHostToDevice, s) == hipSuccess;
                z0 += (size_t)Zs[k];
            }
        }
        ok = ok && launch_resize_linear_dev(din, Z, H, W, size, size, c->input_sat == 255.f, itab, dsm, s) == 0;
        if (!ok) { set_error("tmat_inv_depth_predict: upload failed"); rc = TMAT_E_HIP; }
        else {
            const int blocks = (int)((npx + 255) / 256);
            hipLaunchKernelGGL(minmax_u16_img_kernel, dim3(Z), dim3(256), 0, s, dsm, (int)npx, mnmx, mnmx + Z);
            hipLaunchKernelGGL(inv_prep_kernel, dim3(blocks < 1024 ? blocks : 1024, Z), dim3(256), 0, s, dsm, (int)npx, mnmx, mnmx + Z, dx);
            for (int mi = 0; mi < n_models && !rc; mi++)
                for (int z0 = 0; z0 < Z && !rc; z0 += CH) {
                    const int k = std::min(CH, Z - z0);
                    rc = resnet_forward_dev(c->resnets[model_ids[mi]], dx + (size_t)z0 * npx * 3, k, size, bufs, col, dp + (size_t)mi * Z + z0, s);
                }
            std::vector<float> ph((size_t)Z * n_models);
            if (!rc && (!hip_ok(hipMemcpyAsync(ph.data(), dp, ph.size() * 4, hipMemcpyDeviceToHost, s), "D2H") ||
                        (x_out && !hip_ok(hipMemcpyAsync(x_out, dx, (size_t)Z * npx * 3 * 4, hipMemcpyDeviceToHost, s), "D2H")) ||
                        !hip_ok(hipStreamSynchronize(s), "sync"))) { hipStreamSynchronize(s); rc = TMAT_E_HIP; }      // drain before ph goes out of scope
            if (!rc) for (int z = 0; z < Z; z++) for (int mi = 0; mi < n_models; mi++) probs[(size_t)z * n_models + mi] = ph[(size_t)mi * Z + z];     // (Z, n_models)
        }
    }
    if (rc) hipStreamSynchronize(s);      // nothing of a failed call stays in flight on the handle's workspaces
    return rc;
}

extern "C" {

int tmat_inv_depth_predict(tmat_handle hd, const int *model_ids, int n_models, const uint16_t *stack, int Z, int H, int W, int size, float *probs, float *x_out)
{
    Ctx *c = (Ctx *)hd;
    if (!c || !model_ids || !stack || !probs || n_models < 1 || Z < 0 || H < 1 || W < 1 || size < 32 || size % 32) { set_error("tmat_inv_depth_predict: bad argument"); return TMAT_E_ARG; }
    return inv_depth_impl(hd, model_ids, n_models, &stack, &Z, 1, Z, H, W, size, probs, x_out);
}

int tmat_inv_depth_predict_multi(tmat_handle hd, const int *model_ids, int n_models, const uint16_t *const *stacks, const int *Zs, int n_stacks, int H, int W,
                                 int size, float *probs)
{
    Ctx *c = (Ctx *)hd;
    if (!c || !model_ids || !stacks || !Zs || !probs || n_models < 1 || n_stacks < 0 || H < 1 || W < 1 || size < 32 || size % 32) {
        set_error("tmat_inv_depth_predict_multi: bad argument"); return TMAT_E_ARG;
    }
    long long Z = 0;
    for (int k = 0; k < n_stacks; k++) {
        if (Zs[k] < 0 || (Zs[k] > 0 && !stacks[k])) { set_error("tmat_inv_depth_predict_multi: bad stack"); return TMAT_E_ARG; }
        Z += Zs[k];
    }
    if (Z > 0x7fffffffLL / ((long long)size * size)) { set_error("tmat_inv_depth_predict_multi: too many slices"); return TMAT_E_ARG; }
    return inv_depth_impl(hd, model_ids, n_models, stacks, Zs, n_stacks, (int)Z, H, W, size, probs, nullptr);
}

}  // extern "C"
