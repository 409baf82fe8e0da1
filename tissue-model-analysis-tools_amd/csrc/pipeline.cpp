// Batch drivers of the branching hot path (include/tmat.h): tmat_segment_batch,
// tmat_postprocess_batch, tmat_analyze_batch(_dev).
//
// Reference control flow: scripts/compute_branches.py:585-594 runs analyze_img one image at a time.
// Here a run is cut into passes of K images (K = max_patches / patches-per-image).  Per pass the GPU
// does Lanczos4 + rescale (preproc_kernels.hip), tile extraction, the UNet over K*200 patches
// (unet_kernels.hip) and the f64 window blend (blend_kernels.hip) on the handle's stream; the
// probability maps come back through pinned memory, and host worker threads run the sequential
// graph stages (postproc.cpp, dmt.cpp, morse.cpp), one image per thread, WHILE the GPU already
// works on the next pass.  Nothing is shared between images, so this is also how the path shards
// across GPUs (one process per GPU, see tmat_amd/distributed.py).
#include "../../include/tmat.h"
#include "tmat_ctx.h"
#include "postproc.h"
#include "morph.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <chrono>
#include <cstdio>
#include <thread>

namespace tmat {

void launch_lanczos(const uint16_t *img, int n, int H, int W, int h, int w, const int *xi, const float *xc, const int *yi,
                    const float *yc, float *tmp, uint16_t *out, float sat, hipStream_t s);
void launch_rescale01(const uint16_t *x, int n, size_t per, int *mn, int *mx, float *out, hipStream_t s);

static int round_half_even(double v) { return (int)std::nearbyint(v); }

// device / pinned buffers for one image geometry, cached on the handle
static int ensure_pass_buffers(Ctx *c, int K, int H, int W, int h, int w, int fh, int fw)
{
    PassBuf &b = c->pass;
    if (b.K >= K && b.H == H && b.W == W && b.h == h && b.w == w && b.fh == fh && b.fw == fw) return TMAT_OK;
    c->free_pass();
    std::vector<int> xi, yi;
    std::vector<float> xc, yc;
    lanczos_axis(W, w, xi, xc);
    lanczos_axis(H, h, yi, yc);
    TMAT_HIP(hipMalloc((void **)&b.xi, xi.size() * 4)); TMAT_HIP(hipMalloc((void **)&b.xc, xc.size() * 4));
    TMAT_HIP(hipMalloc((void **)&b.yi, yi.size() * 4)); TMAT_HIP(hipMalloc((void **)&b.yc, yc.size() * 4));
    TMAT_HIP(hipMemcpy(b.xi, xi.data(), xi.size() * 4, hipMemcpyHostToDevice));
    TMAT_HIP(hipMemcpy(b.xc, xc.data(), xc.size() * 4, hipMemcpyHostToDevice));
    TMAT_HIP(hipMemcpy(b.yi, yi.data(), yi.size() * 4, hipMemcpyHostToDevice));
    TMAT_HIP(hipMemcpy(b.yc, yc.data(), yc.size() * 4, hipMemcpyHostToDevice));
    // every scratch buffer is recorded with its size: tmat_debug_poison (test-only) fills them between calls
#define DALLOC(ptr_, bytes_) { const size_t nb_ = (bytes_); TMAT_HIP(hipMalloc((void **)&(ptr_), nb_)); b.ws.push_back(WsEnt{(void *)(ptr_), nb_, false}); }
#define HALLOC(ptr_, bytes_) { const size_t nb_ = (bytes_); TMAT_HIP(hipHostMalloc((void **)&(ptr_), nb_, hipHostMallocDefault)); b.ws.push_back(WsEnt{(void *)(ptr_), nb_, true}); }
    DALLOC(b.tmp, (size_t)K * H * w * sizeof(float));
    DALLOC(b.small, (size_t)K * h * w * sizeof(uint16_t));
    DALLOC(b.x, (size_t)K * h * w * sizeof(float));
    DALLOC(b.mn, (size_t)K * sizeof(int)); DALLOC(b.mx, (size_t)K * sizeof(int));
    DALLOC(b.morph_ws, morph_workspace_bytes(K, h, w));
    DALLOC(b.finish_ws, finish_workspace_bytes(K, h, w, fh, fw));
    for (int i = 0; i < 2; i++) {
        DALLOC(b.skel[i], (size_t)K * h * w);
        HALLOC(b.skel_host[i], (size_t)K * h * w);
        DALLOC(b.field[i], (size_t)K * fh * fw * sizeof(float));
        DALLOC(b.f255[i], (size_t)K * fh * fw * sizeof(float));
        HALLOC(b.f255_host[i], (size_t)K * fh * fw * sizeof(float));
    }
    if (thin_dev_supported(h, w)) {
        DALLOC(b.thin_ws, thin_workspace_bytes(K, h, w));
        DALLOC(b.tie, (size_t)K * h * w * sizeof(uint32_t));
        for (int i = 0; i < 2; i++) {
            DALLOC(b.nfg[i], (size_t)K * sizeof(int));
            HALLOC(b.nfg_host[i], (size_t)K * sizeof(int));
            HALLOC(b.tie_host[i], (size_t)K * h * w * sizeof(uint32_t));
        }
    }
    if (fh >= 2 && fw >= 2) {
        const size_t nE = dmt_edge_count(fh, fw);
        DALLOC(b.dmt_ws, dmt_workspace_bytes(K, fh, fw));
        for (int i = 0; i < 2; i++) {
            DALLOC(b.dmt_ids[i], (size_t)K * nE * sizeof(int32_t));
            HALLOC(b.dmt_ids_host[i], (size_t)K * nE * sizeof(int32_t));
            DALLOC(b.dmt_m[i], (size_t)K * sizeof(int));
            HALLOC(b.dmt_m_host[i], (size_t)K * sizeof(int));
        }
        if (c->dmt_device && c->dmt_sweep_device) {
            DALLOC(b.dmt_sweep_ws, dmt_sweep_workspace_bytes(K, fh, fw));
            for (int i = 0; i < 2; i++) {
                DALLOC(b.dmt_kind[i], (size_t)K * nE);
                HALLOC(b.dmt_kind_host[i], (size_t)K * nE);
                DALLOC(b.dmt_pers[i], (size_t)K * nE * sizeof(float));
                HALLOC(b.dmt_pers_host[i], (size_t)K * nE * sizeof(float));
            }
        }
    }
    b.fh = fh; b.fw = fw;
    for (int i = 0; i < 2; i++) {
        DALLOC(b.pred[i], (size_t)K * h * w * sizeof(double));
        HALLOC(b.pred_host[i], (size_t)K * h * w * sizeof(double));
        DALLOC(b.filt[i], (size_t)K * h * w);
        DALLOC(b.dist[i], (size_t)K * h * w * sizeof(double));
        HALLOC(b.filt_host[i], (size_t)K * h * w);
        HALLOC(b.dist_host[i], (size_t)K * h * w * sizeof(double));
        HALLOC(b.conv_host[i], (size_t)K * sizeof(int));
        TMAT_HIP(hipEventCreateWithFlags(&b.done[i], hipEventDisableTiming));
    }
#undef DALLOC
#undef HALLOC
    b.K = K; b.H = H; b.W = W; b.h = h; b.w = w;
    return TMAT_OK;
}

// the decision table of the ordered thinning, uploaded once per handle
static int ensure_ma_table(Ctx *c)
{
    if (c->ma_table) return TMAT_OK;
    uint32_t bits[16];
    medial_table_bits(bits);
    TMAT_HIP(hipMalloc((void **)&c->ma_table, sizeof(bits)));
    TMAT_HIP(hipMemcpy(c->ma_table, bits, sizeof(bits), hipMemcpyHostToDevice));
    return TMAT_OK;
}
static bool thin_on_device(const Ctx *c) { return c->thin_device && c->pass.thin_ws != nullptr; }

// GPU part of one pass: imgs_dev (k, H, W) u16 -> pred (k, h, w) f64 in b.pred[slot], copied to pinned host
static int enqueue_segment(Ctx *c, const uint16_t *imgs_dev, int k, int slot)
{
    PassBuf &b = c->pass;
    launch_lanczos(imgs_dev, k, b.H, b.W, b.h, b.w, b.xi, b.xc, b.yi, b.yc, b.tmp, b.small, c->input_sat, c->stream);
    launch_rescale01(b.small, k, (size_t)b.h * b.w, b.mn, b.mx, b.x, c->stream);
    int rc = predict_smooth_dev(c, b.x, k, b.h, b.w, b.pred[slot]);
    if (rc) return rc;
    TMAT_HIP(hipMemcpyAsync(b.pred_host[slot], b.pred[slot], (size_t)k * b.h * b.w * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TMAT_HIP(hipEventRecord(b.done[slot], c->stream));
    return TMAT_OK;
}

// Two-stream form used by tmat_analyze_batch*: the memory-bound front half of pass p+1 (Lanczos, rescale, tile
// gather, UNet down path) runs on stream2 while the MFMA-bound back half of pass p (UNet up path, final conv, blend,
// D2H) runs on the main stream.
static bool use_one_stream()
{
    static const bool one = [] { const char *e = getenv("TMAT_STREAMS"); return !(e && atoi(e) == 2); }();
    return one;
}
// the tail of a pass (blend, mask filter, EDT, copies) on the second stream: see enqueue_back
static bool tail_on_side_stream()
{
    static const bool side = [] { const char *e = getenv("TMAT_TAIL_STREAM"); return !(e && atoi(e) == 0); }() && use_one_stream();
    return side;
}
// The front end of a pass (Lanczos, rescale, tile gather: ~2 ms of memory-bound kernels) runs on the SECOND stream, ahead of time: the
// one of pass p + 2 is queued when pass p's tail has finished, beside pass p + 1's network, into the input buffer pass p has released
// (patch_in / patch_in2 alternate), so that on the main stream one pass's network follows the other's directly (TMAT_PRE_STREAM=0 keeps
// it on the main stream in front of its down path).  An image that needs more patches than the activation workspace holds takes the
// old route: everything on the main stream, one input buffer.
static bool pre_on_side_stream(const Ctx *c, const TileGeom &g) { return c->pre_side && tail_on_side_stream() && g.tiles_per_img <= c->max_patches && c->patch_in2; }
static float *patch_in_of(Ctx *c, int slot, const TileGeom &g) { return pre_on_side_stream(c, g) && (slot & 1) ? c->patch_in2 : c->patch_in; }
static int enqueue_pre(Ctx *c, const uint16_t *imgs_dev, int k, int slot, const TileGeom &g)
{
    PassBuf &b = c->pass;
    const bool oversize = g.tiles_per_img > c->max_patches;
    const bool side = pre_on_side_stream(c, g);
    hipStream_t s = side ? c->stream2 : (use_one_stream() || oversize) ? c->stream : c->stream2;
    // the down path of the pass before last read this buffer (long finished: its whole pass has ended; stated for the record)
    if (side && c->down_pending[slot]) TMAT_HIP(hipStreamWaitEvent(s, c->ev_down[slot], 0));
    launch_lanczos(imgs_dev, k, b.H, b.W, b.h, b.w, b.xi, b.xc, b.yi, b.yc, b.tmp, b.small, c->input_sat, s);
    launch_rescale01(b.small, k, (size_t)b.h * b.w, b.mn, b.mx, b.x, s);
    if (c->norm_on) launch_norm_f32(b.x, (size_t)k * b.h * b.w, c->norm_mean, c->norm_std, s);      // models.py:636-637
    float *mn = (float *)c->scratch, *mx = mn + k;
    launch_minmax_f32(b.x, k, (size_t)b.h * b.w, mn, mx, s);
    launch_extract_tiles(b.x, mn, k, g, patch_in_of(c, slot, g), s);
    if (side) TMAT_HIP(hipEventRecord(c->ev_pre[slot], s));
    return TMAT_OK;
}
static int enqueue_down(Ctx *c, int k, int slot, const TileGeom &g)
{
    // Default: the network on the main stream.  TMAT_STREAMS=2 puts the front end and the down path on the second stream; measured +1.8 %
    // images/s, but every kernel of the MFMA half then shares the CUs with a memory-bound one and its own duration
    // (the roofline measurement) stretches by 20 %, so the overlap is opt-in.
    // an image that needs more patches than the activation workspace holds: the whole network runs here, chunk by
    // chunk, on the main stream (enqueue_back then only blends)
    const bool oversize = g.tiles_per_img > c->max_patches;
    hipStream_t s = (use_one_stream() || oversize) ? c->stream : c->stream2;
    if (pre_on_side_stream(c, g)) TMAT_HIP(hipStreamWaitEvent(s, c->ev_pre[slot], 0));
    // oversize: the whole network runs HERE and writes the single patch_out, which the blend of the previous pass may still be
    // reading on the second stream (enqueue_back's own wait on ev_blend comes too late: it is issued after this forward)
    if (oversize && tail_on_side_stream() && c->blend_pending[slot ^ 1]) TMAT_HIP(hipStreamWaitEvent(s, c->ev_blend[slot ^ 1], 0));
    float *pin = patch_in_of(c, slot, g);
    int rc = oversize ? unet_forward_dev(c, pin, k * g.tiles_per_img, c->patch_out, s)
                      : unet_down_dev(c, pin, k * g.tiles_per_img, c->dout[slot], s);
    if (rc) return rc;
    TMAT_HIP(hipEventRecord(c->ev_down[slot], s));
    c->down_pending[slot] = true;
    return TMAT_OK;
}
static int enqueue_back(Ctx *c, int k, int slot, const TileGeom &g)
{
    PassBuf &b = c->pass;
    hipStream_t s = c->stream;
    TMAT_HIP(hipStreamWaitEvent(s, c->ev_down[slot], 0));
    // The tail of a pass -- blend, mask filter, EDT, the copies: ~8 ms of small kernels (83 Zhang launches among them) that leave most
    // of the chip idle -- runs on the second stream, so that the next pass's network follows this pass's network directly on the main
    // stream: 31.28 -> 31.58 images/s (TMAT_TAIL_STREAM=0 keeps everything on the main stream).  patch_out is single: the next up path
    // waits for this pass's blend.  (Round 2 tried the same with a low-priority stream and saw nothing; the second stream has normal priority.)
    const bool tail_side = tail_on_side_stream();
    if (tail_side && c->blend_pending[slot ^ 1]) TMAT_HIP(hipStreamWaitEvent(s, c->ev_blend[slot ^ 1], 0));
    int rc = g.tiles_per_img > c->max_patches ? TMAT_OK : unet_up_dev(c, c->dout[slot], k * g.tiles_per_img, c->patch_out, s);
    if (rc) return rc;
    if (tail_side) {
        TMAT_HIP(hipEventRecord(c->ev_up[slot], s));
        s = c->stream2;
        TMAT_HIP(hipStreamWaitEvent(s, c->ev_up[slot], 0));
    }
    launch_blend(c->patch_out, c->win1d, k, g, b.pred[slot], s);
    if (tail_side) { TMAT_HIP(hipEventRecord(c->ev_blend[slot], s)); c->blend_pending[slot] = true; }
    // binary morphology on the GPU: threshold, median, labelling, perimeter, thinning, fork test, EDT (remove_isolated=True
    // is filter_branch_seg_mask's default, compute_branches.py:337)
    rc = filter_edt_dev(b.pred[slot], k, b.h, b.w, 1, b.morph_ws, b.filt[slot], b.dist[slot], s);
    if (rc) return TMAT_E_HIP;
    const size_t npx = (size_t)k * b.h * b.w;
    if (thin_on_device(c)) {
        // the ordered thinning runs on the device too: the host only needs the foreground counts (for the permutations)
        if (thin_count_dev(b.filt[slot], k, b.h, b.w, b.nfg[slot], s)) return TMAT_E_HIP;
        TMAT_HIP(hipMemcpyAsync(b.nfg_host[slot], b.nfg[slot], k * sizeof(int), hipMemcpyDeviceToHost, s));
    } else {
        TMAT_HIP(hipMemcpyAsync(b.filt_host[slot], b.filt[slot], npx, hipMemcpyDeviceToHost, s));
        TMAT_HIP(hipMemcpyAsync(b.dist_host[slot], b.dist[slot], npx * sizeof(double), hipMemcpyDeviceToHost, s));
    }
    TMAT_HIP(hipMemcpyAsync(b.conv_host[slot], morph_done_flags(b.morph_ws, k, b.h, b.w), k * sizeof(int), hipMemcpyDeviceToHost, s));
    TMAT_HIP(hipEventRecord(b.done[slot], s));
    return TMAT_OK;
}

struct GraphParams {
    int fh, fw;
    float t1, t2;
    int smooth, min_len, max_len, remove_isolated;
};

// host part for one image: pred (h, w) f64 -> row
static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static bool trace_on() { static int t = -1; if (t < 0) { const char *e = getenv("TMAT_TRACE"); t = e && atoi(e) > 0; } return t; }

static int n_workers(int k)
{
    int hw = (int)std::thread::hardware_concurrency();
    if (hw <= 0) hw = 4;
    const char *e = getenv("TMAT_HOST_THREADS");
    if (e && atoi(e) > 0) hw = atoi(e);
    return std::max(1, std::min(hw, k));
}

template <class F>
static void parallel_images(int k, F f)
{
    std::atomic<int> next{0};
    std::vector<std::thread> th;
    const int nt = n_workers(k);
    for (int t = 0; t < nt; t++)
        th.emplace_back([&]() { for (;;) { const int i = next.fetch_add(1); if (i >= k) break; f(i); } });
    for (auto &t : th) t.join();
}

// Host coordinator of one pass (runs in its own thread while the GPU works on the next pass):
//   1. ordered medial-axis thinning per image (sequential by construction)            host threads
//   2. EDT(~skeleton), weighting, resize, rescale on the third stream                 GPU (finish_kernels.hip)
//   3. DMT sweeps + collect + MorseGraph statistics per image                         host threads
struct PassJob {
    std::thread th;
    std::atomic<int> rc{0};
    void join() { if (th.joinable()) th.join(); }
};

static void run_pass_host(Ctx *c, int slot, int k, const GraphParams gp, tmat_row *rows, PassJob *job)
{
    PassBuf &b = c->pass;
    const int h = b.h, w = b.w;
    const size_t per = (size_t)h * w, fper = (size_t)gp.fh * gp.fw;
    const double t0 = now_s();
    double t_perm = t0;
    hipSetDevice(c->device);
    hipStream_t s = c->stream3;
    bool ok = true;
    if (thin_on_device(c)) {
        // host part of the medial axis: the RandomState(0) tie-break permutation of every image's foreground count
        parallel_images(k, [&](int i) {
            std::vector<uint32_t> perm;
            legacy_permutation(0, (size_t)b.nfg_host[slot][i], perm);
            std::memcpy(b.tie_host[slot] + i * per, perm.data(), perm.size() * sizeof(uint32_t));
        });
        t_perm = now_s();
        for (int i = 0; i < k && ok; i++)
            ok = hipMemcpyAsync(b.tie + i * per, b.tie_host[slot] + i * per, (size_t)b.nfg_host[slot][i] * sizeof(uint32_t), hipMemcpyHostToDevice, s) == hipSuccess;
        ok = ok && thin_dev(b.filt[slot], b.dist[slot], b.tie, b.nfg[slot], k, h, w, b.thin_ws, c->ma_table, b.skel[slot], s) == 0;
    } else {
        parallel_images(k, [&](int i) { medial_axis_thin(b.filt_host[slot] + i * per, b.dist_host[slot] + i * per, h, w, b.skel_host[slot] + i * per); });
        ok = hipMemcpyAsync(b.skel[slot], b.skel_host[slot], k * per, hipMemcpyHostToDevice, s) == hipSuccess;
    }
    const double t1 = now_s();
    ok = ok && finish_dev(b.pred[slot], b.dist[slot], b.skel[slot], k, h, w, gp.fh, gp.fw, b.finish_ws, b.field[slot], b.f255[slot], s) == 0;
    ok = ok && hipMemcpyAsync(b.f255_host[slot], b.f255[slot], k * fper * sizeof(float), hipMemcpyDeviceToHost, s) == hipSuccess;
    // DMT front end on the device: edge keys + lower-star sort from the field in HBM; the sorted edge ids come back
    const bool dmt_dev = c->dmt_device && b.dmt_ws && gp.fh >= 2 && gp.fw >= 2;
    const size_t nE = dmt_dev ? dmt_edge_count(gp.fh, gp.fw) : 0;
    if (dmt_dev) {
        ok = ok && dmt_sorted_edges_dev(b.f255[slot], k, gp.fh, gp.fw, b.dmt_ws, b.dmt_ids[slot], b.dmt_m[slot], s) == 0;
        ok = ok && hipMemcpyAsync(b.dmt_ids_host[slot], b.dmt_ids[slot], k * nE * sizeof(int32_t), hipMemcpyDeviceToHost, s) == hipSuccess;
        ok = ok && hipMemcpyAsync(b.dmt_m_host[slot], b.dmt_m[slot], k * sizeof(int), hipMemcpyDeviceToHost, s) == hipSuccess;
    }
    // ... and the two persistence sweeps (one workgroup per image, one launch): pairing kind + persistence per sorted edge
    const bool sweep_dev = dmt_dev && b.dmt_sweep_ws;
    if (sweep_dev) {
        ok = ok && dmt_sweeps_dev(b.f255[slot], b.dmt_ids[slot], b.dmt_m[slot], k, gp.fh, gp.fw, b.dmt_sweep_ws, b.dmt_kind[slot], b.dmt_pers[slot], s) == 0;
        ok = ok && hipMemcpyAsync(b.dmt_kind_host[slot], b.dmt_kind[slot], k * nE, hipMemcpyDeviceToHost, s) == hipSuccess;
        ok = ok && hipMemcpyAsync(b.dmt_pers_host[slot], b.dmt_pers[slot], k * nE * sizeof(float), hipMemcpyDeviceToHost, s) == hipSuccess;
    }
    ok = ok && hipStreamSynchronize(s) == hipSuccess;
    if (!ok) { job->rc = TMAT_E_HIP; return; }
    const double t2 = now_s();
    parallel_images(k, [&](int i) {
        const int cap_v = (int)fper + 4, cap_e = 3 * (int)fper + 4;
        std::vector<int32_t> V((size_t)cap_v * 2), E((size_t)cap_e * 2);
        int nv = 0, ne = 0;
        int rc = dmt_graph_host_sorted(b.f255_host[slot] + i * fper, gp.fh, gp.fw, gp.t1, gp.t2, dmt_dev ? b.dmt_ids_host[slot] + i * nE : nullptr,
                                       dmt_dev ? b.dmt_m_host[slot][i] : 0, V.data(), cap_v, E.data(), cap_e, &nv, &ne,
                                       sweep_dev ? b.dmt_kind_host[slot] + i * nE : nullptr, sweep_dev ? b.dmt_pers_host[slot] + i * nE : nullptr);
        if (!rc)
            rc = tmat_morse_stats(V.data(), nv, E.data(), ne, gp.fh, gp.fw, gp.smooth, gp.min_len, gp.max_len, gp.remove_isolated, nullptr,
                                  &rows[i].count, &rows[i].total_px, &rows[i].avg_px, nullptr, 0);
        if (rc) job->rc = rc;
    });
    if (trace_on())
        fprintf(stderr, "[tmat] host pass (%d images): ordered thinning %.1f ms (host permutations %.1f, the rest = its launches and convergence polls on the low-priority stream), finish + DMT on the GPU %.1f ms, collect + Morse %.1f ms\n", k,
                (t1 - t0) * 1e3, (t_perm - t0) * 1e3, (t2 - t1) * 1e3, (now_s() - t2) * 1e3);
}

static int analyze_dev(Ctx *c, const uint16_t *imgs_dev, int n, int H, int W, double ds_ratio, int ds_width, GraphParams gp,
                       int64_t first_index, tmat_row *rows)
{
    // compute_branches.py:309-312 hands target_shape = round(shape * ds_ratio) = (round(H r), round(W r)) to cv2.resize as
    // dsize, which cv2 reads as (width, height): the resized image has round(W r) rows and round(H r) columns.  Square
    // images are unaffected; for the others the network sees the reference's (anisotropically scaled) image and the
    // later resize to (fh, fw) restores the aspect ratio, exactly as in the reference.
    const int h = round_half_even((double)W * ds_ratio), w = round_half_even((double)H * ds_ratio);
    if (h < 1 || w < 1) { set_error("analyze: target shape is empty"); return TMAT_E_ARG; }
    gp.fh = round_half_even((double)H * ((double)ds_width / (double)W));
    gp.fw = round_half_even((double)W * ((double)ds_width / (double)W));
    TileGeom g = make_geom(h, w, c->patch);
    const int K = std::min(n, std::max(1, c->max_patches / g.tiles_per_img));      // one image per pass when it needs > max_patches
    { int rc0 = ensure_patch_io(c, K * g.tiles_per_img); if (rc0) return rc0; }
    int rc = ensure_pass_buffers(c, K, H, W, h, w, gp.fh, gp.fw);
    if (!rc) rc = ensure_ma_table(c);
    if (rc) return rc;
    for (int i = 0; i < n; i++) { rows[i].index = first_index + i; rows[i].count = 0; rows[i].total_px = 0; rows[i].avg_px = 0; }
    const int P = (n + K - 1) / K;
    PassJob jobs[2];
    auto cnt = [&](int p) { return std::min(K, n - p * K); };
    auto img_at = [&](int p) { return imgs_dev + (size_t)p * K * H * W; };
    // the caller may have queued work that produces the images on the main stream (tmat_zproj_dev, tmat_dev_upload)
    if (!use_one_stream() && !hip_ok(hipStreamSynchronize(c->stream), "hipStreamSynchronize")) return TMAT_E_HIP;
    // second input buffer for the front end that runs ahead on the second stream (enqueue_pre)
    if (c->pre_side && tail_on_side_stream() && g.tiles_per_img <= c->max_patches && !c->patch_in2)
        TMAT_HIP(hipMalloc((void **)&c->patch_in2, (size_t)c->patch * c->patch * c->patch_cap * sizeof(float)));
    if (pre_on_side_stream(c, g)) {         // what the caller queued on the main stream (the images) comes first on the second one too
        TMAT_HIP(hipEventRecord(c->ev_pre[0], c->stream));
        TMAT_HIP(hipStreamWaitEvent(c->stream2, c->ev_pre[0], 0));
    }
    // (order of the calls = order on the second stream: the front end of pass p + 2 in front of the tail of pass p + 1, which only
    // starts when that pass's up path has ended)
    c->down_pending[0] = c->down_pending[1] = false;
    rc = enqueue_pre(c, img_at(0), cnt(0), 0, g);
    if (!rc) rc = enqueue_down(c, cnt(0), 0, g);
    if (!rc && P > 1) rc = enqueue_pre(c, img_at(1), cnt(1), 1, g);
    if (!rc) rc = enqueue_back(c, cnt(0), 0, g);
    if (!rc && P > 1) rc = enqueue_down(c, cnt(1), 1, g);
    for (int p = 0; p < P && !rc; p++) {
        const int slot = p & 1;
        const double tw0 = now_s();
        if (!hip_ok(hipEventSynchronize(c->pass.done[slot]), "hipEventSynchronize")) { rc = TMAT_E_HIP; break; }
        const double tw1 = now_s();
        if (p >= 1) { jobs[slot ^ 1].join(); if (jobs[slot ^ 1].rc) rc = jobs[slot ^ 1].rc; }
        if (trace_on())
            fprintf(stderr, "[tmat] pass %d/%d (%d images): waited %.1f ms for the GPU, %.1f ms for host jobs of the previous pass\n",
                    p + 1, P, cnt(p), (tw1 - tw0) * 1e3, (now_s() - tw1) * 1e3);
        if (p + 2 < P && !rc) rc = enqueue_pre(c, img_at(p + 2), cnt(p + 2), slot, g);
        if (p + 1 < P && !rc) rc = enqueue_back(c, cnt(p + 1), slot ^ 1, g);
        if (p + 2 < P && !rc) rc = enqueue_down(c, cnt(p + 2), slot, g);
        for (int i = 0; i < cnt(p) && !rc; i++)
            if (!c->pass.conv_host[slot][i]) { set_error("analyze: Zhang thinning did not converge within its launch budget"); rc = TMAT_E_HIP; }
        if (!rc) jobs[slot].th = std::thread(run_pass_host, c, slot, cnt(p), gp, rows + (size_t)p * K, &jobs[slot]);
    }
    for (auto &j : jobs) { j.join(); if (j.rc && !rc) rc = j.rc; }
    hipStreamSynchronize(c->stream2);
    hipStreamSynchronize(c->stream);
    return rc;
}

// medial_axis on the device for k masks that are in HBM with their EDT: foreground counts -> host permutations -> keys,
// sort, ordered thinning (thin_kernels.hip).  Synchronises the stream once (the counts come back to the host).
int medial_thin_batch_dev(Ctx *c, const uint8_t *mask_dev, const double *dist_dev, int k, int hh, int ww, uint8_t *skel_dev, hipStream_t s)
{
    int rc = ensure_ma_table(c);
    if (rc) return rc;
    const size_t per = (size_t)hh * ww;
    int *nfg = nullptr; uint32_t *tie = nullptr; void *ws = nullptr;
    std::vector<int> nfg_host(k, 0);
    if (!hip_ok(hipMalloc((void **)&nfg, k * sizeof(int)), "hipMalloc") || !hip_ok(hipMalloc((void **)&tie, k * per * sizeof(uint32_t)), "hipMalloc") ||
        !hip_ok(hipMalloc(&ws, thin_workspace_bytes(k, hh, ww)), "hipMalloc")) rc = TMAT_E_HIP;
    if (!rc && thin_count_dev(mask_dev, k, hh, ww, nfg, s)) rc = TMAT_E_HIP;
    if (!rc && (!hip_ok(hipMemcpyAsync(nfg_host.data(), nfg, k * sizeof(int), hipMemcpyDeviceToHost, s), "D2H") || !hip_ok(hipStreamSynchronize(s), "sync"))) rc = TMAT_E_HIP;
    std::vector<std::vector<uint32_t>> perms(k);
    if (!rc) parallel_images(k, [&](int i) { legacy_permutation(0, (size_t)nfg_host[i], perms[i]); });
    for (int i = 0; i < k && !rc; i++)
        if (!perms[i].empty() && !hip_ok(hipMemcpyAsync(tie + i * per, perms[i].data(), perms[i].size() * sizeof(uint32_t), hipMemcpyHostToDevice, s), "H2D")) rc = TMAT_E_HIP;
    if (!rc && thin_dev(mask_dev, dist_dev, tie, nfg, k, hh, ww, ws, c->ma_table, skel_dev, s)) { set_error("medial axis: device thinning failed"); rc = TMAT_E_HIP; }
    if (!rc && !hip_ok(hipStreamSynchronize(s), "sync")) rc = TMAT_E_HIP;      // perms / scratch are released below
    hipFree(nfg); hipFree(tie); hipFree(ws);
    return rc;
}

// tmat_dmt_graph / tmat_dmt_graph_batch with a handle: key build + sort + the two persistence sweeps of all n fields on the handle's
// device (one launch each), `collect` per field on host threads.  Outputs of field i start at verts + 2 i cap_v / edges + 2 i cap_e.
int dmt_graph_device_batch(void *handle, const float *imgs, int n, int R, int C, float delta1, float delta2, int32_t *verts, int cap_v,
                           int32_t *edges, int cap_e, int *n_verts, int *n_edges)
{
    Ctx *c = (Ctx *)handle;
    TMAT_HIP(hipSetDevice(c->device));
    const size_t nE = dmt_edge_count(R, C), npx = (size_t)R * C;
    float *df = nullptr; void *ws = nullptr; int32_t *ids = nullptr; int *m = nullptr;
    std::vector<int32_t> ids_host((size_t)n * nE);
    std::vector<int> m_host(n, 0);
    int rc = TMAT_OK;
    if (!hip_ok(hipMalloc((void **)&df, n * npx * sizeof(float)), "hipMalloc") || !hip_ok(hipMalloc(&ws, dmt_workspace_bytes(n, R, C)), "hipMalloc") ||
        !hip_ok(hipMalloc((void **)&ids, n * nE * sizeof(int32_t)), "hipMalloc") || !hip_ok(hipMalloc((void **)&m, n * sizeof(int)), "hipMalloc")) rc = TMAT_E_HIP;
    if (!rc && !hip_ok(hipMemcpyAsync(df, imgs, n * npx * sizeof(float), hipMemcpyHostToDevice, c->stream), "H2D")) rc = TMAT_E_HIP;
    if (!rc && dmt_sorted_edges_dev(df, n, R, C, ws, ids, m, c->stream)) { set_error("tmat_dmt_graph: device front end failed"); rc = TMAT_E_HIP; }
    // the two persistence sweeps on the device too (dmt_sweep_kernels.hip; TMAT_DMT_SWEEP_DEVICE=0: on the host); `collect` stays on the host
    const bool sweep_dev = c->dmt_sweep_device;
    std::vector<uint8_t> kind_host;
    std::vector<float> pers_host;
    void *sws = nullptr; uint8_t *dkind = nullptr; float *dpers = nullptr;
    if (!rc && sweep_dev) {
        kind_host.resize((size_t)n * nE); pers_host.resize((size_t)n * nE);
        if (!hip_ok(hipMalloc(&sws, dmt_sweep_workspace_bytes(n, R, C)), "hipMalloc") || !hip_ok(hipMalloc((void **)&dkind, n * nE), "hipMalloc") ||
            !hip_ok(hipMalloc((void **)&dpers, n * nE * sizeof(float)), "hipMalloc")) rc = TMAT_E_HIP;
        if (!rc && dmt_sweeps_dev(df, ids, m, n, R, C, sws, dkind, dpers, c->stream)) { set_error("tmat_dmt_graph: device sweeps failed"); rc = TMAT_E_HIP; }
        if (!rc && (!hip_ok(hipMemcpyAsync(kind_host.data(), dkind, n * nE, hipMemcpyDeviceToHost, c->stream), "D2H") ||
                    !hip_ok(hipMemcpyAsync(pers_host.data(), dpers, n * nE * sizeof(float), hipMemcpyDeviceToHost, c->stream), "D2H"))) rc = TMAT_E_HIP;
    }
    if (!rc && (!hip_ok(hipMemcpyAsync(ids_host.data(), ids, n * nE * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream), "D2H") ||
                !hip_ok(hipMemcpyAsync(m_host.data(), m, n * sizeof(int), hipMemcpyDeviceToHost, c->stream), "D2H"))) rc = TMAT_E_HIP;
    if (!hip_ok(hipStreamSynchronize(c->stream), "sync") && !rc) rc = TMAT_E_HIP;
    hipFree(df); hipFree(ws); hipFree(ids); hipFree(m); hipFree(sws); hipFree(dkind); hipFree(dpers);
    if (rc) return rc;
    std::vector<int> rcs(n, TMAT_OK);
    parallel_images(n, [&](int i) {
        rcs[i] = dmt_graph_host_sorted(imgs + i * npx, R, C, delta1, delta2, ids_host.data() + i * nE, m_host[i], verts + (size_t)i * 2 * cap_v, cap_v,
                                       edges + (size_t)i * 2 * cap_e, cap_e, n_verts + i, n_edges + i, sweep_dev ? kind_host.data() + i * nE : nullptr,
                                       sweep_dev ? pers_host.data() + i * nE : nullptr);
    });
    for (int i = 0; i < n; i++) if (rcs[i]) return rcs[i];
    return TMAT_OK;
}

int dmt_graph_device_front(void *handle, const float *img, int R, int C, float delta1, float delta2, int32_t *verts, int cap_v,
                           int32_t *edges, int cap_e, int *n_verts, int *n_edges)
{
    return dmt_graph_device_batch(handle, img, 1, R, C, delta1, delta2, verts, cap_v, edges, cap_e, n_verts, n_edges);
}

}  // namespace tmat

using namespace tmat;

extern "C" {

int tmat_segment_batch(tmat_handle hd, const uint16_t *imgs, int n, int H, int W, double ds_ratio, double *pred)
{
    Ctx *c = (Ctx *)hd;
    if (c && !has_model(c)) { set_error("tmat_segment_batch: this handle has no model (tmat_create_plain)"); return TMAT_E_ARG; }
    if (!c || !imgs || !pred || n < 0 || H < 1 || W < 1) { set_error("tmat_segment_batch: bad argument"); return TMAT_E_ARG; }
    if (n == 0) return TMAT_OK;
    TMAT_HIP(hipSetDevice(c->device));
    const int h = round_half_even((double)W * ds_ratio), w = round_half_even((double)H * ds_ratio);    // see analyze_dev
    if (h < 1 || w < 1) { set_error("tmat_segment_batch: target shape is empty"); return TMAT_E_ARG; }
    TileGeom g = make_geom(h, w, c->patch);
    const int K = std::min(n, std::max(1, c->max_patches / g.tiles_per_img));
    int rc = ensure_pass_buffers(c, K, H, W, h, w, std::max(1, c->pass.fh), std::max(1, c->pass.fw));
    if (rc) return rc;
    uint16_t *dimg = nullptr;
    TMAT_HIP(hipMalloc((void **)&dimg, (size_t)K * H * W * sizeof(uint16_t)));
    for (int i0 = 0; i0 < n && !rc; i0 += K) {
        const int k = std::min(K, n - i0);
        if (!hip_ok(hipMemcpyAsync(dimg, imgs + (size_t)i0 * H * W, (size_t)k * H * W * 2, hipMemcpyHostToDevice, c->stream), "H2D")) { rc = TMAT_E_HIP; break; }
        rc = enqueue_segment(c, dimg, k, 0);
        if (rc) break;
        if (!hip_ok(hipStreamSynchronize(c->stream), "sync")) { rc = TMAT_E_HIP; break; }
        std::memcpy(pred + (size_t)i0 * h * w, c->pass.pred_host[0], (size_t)k * h * w * sizeof(double));
    }
    hipFree(dimg);
    return rc;
}

int tmat_preprocess_batch(tmat_handle hd, const uint16_t *imgs, int n, int H, int W, double ds_ratio, float *x)
{
    Ctx *c = (Ctx *)hd;
    if (c && !has_model(c)) { set_error("tmat_preprocess_batch: this handle has no model (tmat_create_plain)"); return TMAT_E_ARG; }
    if (!c || !imgs || !x || n < 0 || H < 1 || W < 1) { set_error("tmat_preprocess_batch: bad argument"); return TMAT_E_ARG; }
    if (n == 0) return TMAT_OK;
    TMAT_HIP(hipSetDevice(c->device));
    const int h = round_half_even((double)W * ds_ratio), w = round_half_even((double)H * ds_ratio);    // see analyze_dev
    if (h < 1 || w < 1) { set_error("tmat_preprocess_batch: target shape is empty"); return TMAT_E_ARG; }
    TileGeom g = make_geom(h, w, c->patch);
    const int K = std::min(n, std::max(1, c->max_patches / g.tiles_per_img));
    int rc = ensure_pass_buffers(c, K, H, W, h, w, std::max(1, c->pass.fh), std::max(1, c->pass.fw));
    if (rc) return rc;
    uint16_t *dimg = nullptr;
    TMAT_HIP(hipMalloc((void **)&dimg, (size_t)K * H * W * sizeof(uint16_t)));
    PassBuf &b = c->pass;
    for (int i0 = 0; i0 < n && !rc; i0 += K) {
        const int k = std::min(K, n - i0);
        if (!hip_ok(hipMemcpyAsync(dimg, imgs + (size_t)i0 * H * W, (size_t)k * H * W * 2, hipMemcpyHostToDevice, c->stream), "H2D")) { rc = TMAT_E_HIP; break; }
        launch_lanczos(dimg, k, b.H, b.W, b.h, b.w, b.xi, b.xc, b.yi, b.yc, b.tmp, b.small, c->input_sat, c->stream);
        launch_rescale01(b.small, k, (size_t)b.h * b.w, b.mn, b.mx, b.x, c->stream);
        if (!hip_ok(hipMemcpyAsync(x + (size_t)i0 * h * w, b.x, (size_t)k * h * w * sizeof(float), hipMemcpyDeviceToHost, c->stream), "D2H") ||
            !hip_ok(hipStreamSynchronize(c->stream), "sync")) { hipStreamSynchronize(c->stream); rc = TMAT_E_HIP; break; }
    }
    hipFree(dimg);
    return rc;
}

int tmat_filter_edt_batch(tmat_handle hd, const double *pred, int n, int hh, int ww, uint8_t *filtered, double *dist)
{
    Ctx *c = (Ctx *)hd;
    if (!c || !pred || !filtered || !dist || n < 0 || hh < 1 || ww < 1) { set_error("tmat_filter_edt_batch: bad argument"); return TMAT_E_ARG; }
    if (n == 0) return TMAT_OK;
    TMAT_HIP(hipSetDevice(c->device));
    const size_t npx = (size_t)n * hh * ww;
    double *dp = nullptr, *dd = nullptr; uint8_t *df = nullptr; void *ws = nullptr;
    int rc = TMAT_OK;
    std::vector<int> conv(n, 0);
    if (!hip_ok(hipMalloc((void **)&dp, npx * 8), "hipMalloc") || !hip_ok(hipMalloc((void **)&dd, npx * 8), "hipMalloc") ||
        !hip_ok(hipMalloc((void **)&df, npx), "hipMalloc") || !hip_ok(hipMalloc(&ws, morph_workspace_bytes(n, hh, ww)), "hipMalloc")) rc = TMAT_E_HIP;
    if (!rc && !hip_ok(hipMemcpyAsync(dp, pred, npx * 8, hipMemcpyHostToDevice, c->stream), "H2D")) rc = TMAT_E_HIP;
    if (!rc && filter_edt_dev(dp, n, hh, ww, 1, ws, df, dd, c->stream)) rc = TMAT_E_HIP;
    if (!rc && (!hip_ok(hipMemcpyAsync(filtered, df, npx, hipMemcpyDeviceToHost, c->stream), "D2H") ||
                !hip_ok(hipMemcpyAsync(dist, dd, npx * 8, hipMemcpyDeviceToHost, c->stream), "D2H") ||
                !hip_ok(hipMemcpyAsync(conv.data(), morph_done_flags(ws, n, hh, ww), n * sizeof(int), hipMemcpyDeviceToHost, c->stream), "D2H") ||
                !hip_ok(hipStreamSynchronize(c->stream), "sync"))) rc = TMAT_E_HIP;
    for (int i = 0; i < n && !rc; i++) if (!conv[i]) { set_error("tmat_filter_edt_batch: thinning did not converge"); rc = TMAT_E_HIP; }
    hipFree(dp); hipFree(dd); hipFree(df); hipFree(ws);
    return rc;
}

int tmat_filter_mask_batch(tmat_handle hd, const uint8_t *mask, int n, int hh, int ww, int use_median, int remove_isolated, uint8_t *filtered)
{
    Ctx *c = (Ctx *)hd;
    if (!c || !mask || !filtered || n < 0 || hh < 1 || ww < 1) { set_error("tmat_filter_mask_batch: bad argument"); return TMAT_E_ARG; }
    if (n == 0) return TMAT_OK;
    TMAT_HIP(hipSetDevice(c->device));
    const size_t npx = (size_t)n * hh * ww;
    uint8_t *dm = nullptr, *df = nullptr; void *ws = nullptr;
    int rc = TMAT_OK;
    std::vector<int> conv(n, 0);
    if (!hip_ok(hipMalloc((void **)&dm, npx), "hipMalloc") || !hip_ok(hipMalloc((void **)&df, npx), "hipMalloc") ||
        !hip_ok(hipMalloc(&ws, morph_workspace_bytes(n, hh, ww)), "hipMalloc")) rc = TMAT_E_HIP;
    if (!rc && !hip_ok(hipMemcpyAsync(dm, mask, npx, hipMemcpyHostToDevice, c->stream), "H2D")) rc = TMAT_E_HIP;
    if (!rc && filter_mask_dev(nullptr, dm, n, hh, ww, use_median != 0, remove_isolated != 0, ws, df, nullptr, c->stream)) rc = TMAT_E_HIP;
    if (!rc && (!hip_ok(hipMemcpyAsync(filtered, df, npx, hipMemcpyDeviceToHost, c->stream), "D2H") ||
                !hip_ok(hipMemcpyAsync(conv.data(), morph_done_flags(ws, n, hh, ww), n * sizeof(int), hipMemcpyDeviceToHost, c->stream), "D2H") ||
                !hip_ok(hipStreamSynchronize(c->stream), "sync"))) rc = TMAT_E_HIP;
    for (int i = 0; i < n && !rc; i++) if (!conv[i]) { set_error("tmat_filter_mask_batch: thinning did not converge"); rc = TMAT_E_HIP; }
    hipFree(dm); hipFree(df); hipFree(ws);
    return rc;
}

int tmat_finish_batch(tmat_handle hd, const double *pred, const double *dist, const uint8_t *skel, int n, int hh, int ww, int out_h,
                      int out_w, float *field, float *field255)
{
    Ctx *c = (Ctx *)hd;
    if (!c || !pred || !dist || !skel || !field || !field255 || n < 0 || hh < 1 || ww < 1 || out_h < 1 || out_w < 1) {
        set_error("tmat_finish_batch: bad argument");
        return TMAT_E_ARG;
    }
    if (n == 0) return TMAT_OK;
    TMAT_HIP(hipSetDevice(c->device));
    const size_t npx = (size_t)n * hh * ww, onpx = (size_t)n * out_h * out_w;
    double *dp = nullptr, *dd = nullptr; uint8_t *ds = nullptr; void *ws = nullptr; float *df = nullptr, *d255 = nullptr;
    int rc = TMAT_OK;
    if (!hip_ok(hipMalloc((void **)&dp, npx * 8), "hipMalloc") || !hip_ok(hipMalloc((void **)&dd, npx * 8), "hipMalloc") ||
        !hip_ok(hipMalloc((void **)&ds, npx), "hipMalloc") || !hip_ok(hipMalloc(&ws, finish_workspace_bytes(n, hh, ww, out_h, out_w)), "hipMalloc") ||
        !hip_ok(hipMalloc((void **)&df, onpx * 4), "hipMalloc") || !hip_ok(hipMalloc((void **)&d255, onpx * 4), "hipMalloc")) rc = TMAT_E_HIP;
    if (!rc && (!hip_ok(hipMemcpyAsync(dp, pred, npx * 8, hipMemcpyHostToDevice, c->stream), "H2D") ||
                !hip_ok(hipMemcpyAsync(dd, dist, npx * 8, hipMemcpyHostToDevice, c->stream), "H2D") ||
                !hip_ok(hipMemcpyAsync(ds, skel, npx, hipMemcpyHostToDevice, c->stream), "H2D"))) rc = TMAT_E_HIP;
    if (!rc && finish_dev(dp, dd, ds, n, hh, ww, out_h, out_w, ws, df, d255, c->stream)) rc = TMAT_E_HIP;
    if (!rc && (!hip_ok(hipMemcpyAsync(field, df, onpx * 4, hipMemcpyDeviceToHost, c->stream), "D2H") ||
                !hip_ok(hipMemcpyAsync(field255, d255, onpx * 4, hipMemcpyDeviceToHost, c->stream), "D2H") ||
                !hip_ok(hipStreamSynchronize(c->stream), "sync"))) rc = TMAT_E_HIP;
    hipFree(dp); hipFree(dd); hipFree(ds); hipFree(ws); hipFree(df); hipFree(d255);
    return rc;
}

int tmat_zproj_dev(tmat_handle hd, const uint16_t *stacks_dev, int n, int Z, int H, int W, int method, void *out_dev)
{
    Ctx *c = (Ctx *)hd;
    if (!c || !stacks_dev || !out_dev || n < 0 || Z < 1 || H < 1 || W < 1 || method < TMAT_ZPROJ_FS || method > TMAT_ZPROJ_MED) {
        set_error("tmat_zproj_dev: bad argument");
        return TMAT_E_ARG;
    }
    if (n == 0) return TMAT_OK;
    TMAT_HIP(hipSetDevice(c->device));
    const int rc = zproj_dev(stacks_dev, n, Z, H, W, method, out_dev, c->stream);
    return rc == 0 ? TMAT_OK : rc == -1 ? TMAT_E_ARG : TMAT_E_HIP;
}

int tmat_zproj_batch(tmat_handle hd, const uint16_t *stacks, int n, int Z, int H, int W, int method, void *out)
{
    Ctx *c = (Ctx *)hd;
    if (!c || !stacks || !out || n < 0 || Z < 1 || H < 1 || W < 1 || method < TMAT_ZPROJ_FS || method > TMAT_ZPROJ_MED) {
        set_error("tmat_zproj_batch: bad argument");
        return TMAT_E_ARG;
    }
    if (n == 0) return TMAT_OK;
    TMAT_HIP(hipSetDevice(c->device));
    const size_t npx = (size_t)H * W, per_in = (size_t)Z * npx * sizeof(uint16_t);
    const size_t osz = (method == TMAT_ZPROJ_AVG || method == TMAT_ZPROJ_MED) ? sizeof(double) : sizeof(uint16_t);
    const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)n, ((size_t)1 << 30) / per_in));   // <= 1 GiB of stacks at a time
    uint16_t *din = nullptr; void *dout = nullptr;
    int rc = TMAT_OK;
    // the staging buffers stay on the handle between calls (tmat_ctx.h:ws_get, slots 9 and 10)
    din = (uint16_t *)ws_get(c, 9, (size_t)chunk * per_in); dout = ws_get(c, 10, (size_t)chunk * npx * osz);
    if (!din || !dout) rc = TMAT_E_HIP;
    for (int i0 = 0; i0 < n && !rc; i0 += chunk) {
        const int k = std::min(chunk, n - i0);
        if (!hip_ok(hipMemcpyAsync(din, stacks + (size_t)i0 * Z * npx, (size_t)k * per_in, hipMemcpyHostToDevice, c->stream), "H2D")) { rc = TMAT_E_HIP; break; }
        const int r = zproj_dev(din, k, Z, H, W, method, dout, c->stream);
        if (r) { rc = r == -1 ? TMAT_E_ARG : TMAT_E_HIP; break; }
        if (!hip_ok(hipMemcpyAsync((char *)out + (size_t)i0 * npx * osz, dout, (size_t)k * npx * osz, hipMemcpyDeviceToHost, c->stream), "D2H") ||
            !hip_ok(hipStreamSynchronize(c->stream), "sync")) rc = TMAT_E_HIP;
    }
    if (rc) hipStreamSynchronize(c->stream);
    return rc;
}

// The post-processing of compute_branches.py:334-357 for a batch of probability maps, with the same split as the batch
// pipeline: GPU (threshold, filter_branch_seg_mask, EDT) -> host (ordered medial-axis thinning, sequential by
// construction) -> GPU (EDT of the skeleton, weighting, anti-aliased resize).  Chunked to bound device memory.
int tmat_postprocess_batch(tmat_handle hd, const double *pred, int n, int hh, int ww, int out_h, int out_w, float *field)
{
    Ctx *c = (Ctx *)hd;
    if (!c || !pred || !field || n < 0 || hh < 1 || ww < 1 || out_h < 1 || out_w < 1) { set_error("tmat_postprocess_batch: bad argument"); return TMAT_E_ARG; }
    if (n == 0) return TMAT_OK;
    TMAT_HIP(hipSetDevice(c->device));
    const size_t per = (size_t)hh * ww, oper = (size_t)out_h * out_w;
    const int K = (int)std::max<size_t>(1, std::min<size_t>((size_t)n, ((size_t)256 << 20) / (per * 8)));      // <= 256 MiB of f64 maps per chunk
    double *dp = nullptr, *dd = nullptr; uint8_t *df = nullptr, *dsk = nullptr; void *ws = nullptr, *fws = nullptr; float *dfield = nullptr, *d255 = nullptr;
    std::vector<uint8_t> filt((size_t)K * per), skel((size_t)K * per);
    std::vector<double> dist((size_t)K * per);
    std::vector<int> conv(K, 0);
    int rc = TMAT_OK;
    if (!hip_ok(hipMalloc((void **)&dp, K * per * 8), "hipMalloc") || !hip_ok(hipMalloc((void **)&dd, K * per * 8), "hipMalloc") ||
        !hip_ok(hipMalloc((void **)&df, K * per), "hipMalloc") || !hip_ok(hipMalloc((void **)&dsk, K * per), "hipMalloc") ||
        !hip_ok(hipMalloc(&ws, morph_workspace_bytes(K, hh, ww)), "hipMalloc") ||
        !hip_ok(hipMalloc(&fws, finish_workspace_bytes(K, hh, ww, out_h, out_w)), "hipMalloc") ||
        !hip_ok(hipMalloc((void **)&dfield, K * oper * 4), "hipMalloc") || !hip_ok(hipMalloc((void **)&d255, K * oper * 4), "hipMalloc")) rc = TMAT_E_HIP;
    hipStream_t s = c->stream;
    for (int i0 = 0; i0 < n && !rc; i0 += K) {
        const int k = std::min(K, n - i0);
        if (!hip_ok(hipMemcpyAsync(dp, pred + (size_t)i0 * per, k * per * 8, hipMemcpyHostToDevice, s), "H2D")) { rc = TMAT_E_HIP; break; }
        if (filter_edt_dev(dp, k, hh, ww, 1, ws, df, dd, s)) { rc = TMAT_E_HIP; break; }
        if (!hip_ok(hipMemcpyAsync(filt.data(), df, k * per, hipMemcpyDeviceToHost, s), "D2H") ||
            !hip_ok(hipMemcpyAsync(dist.data(), dd, k * per * 8, hipMemcpyDeviceToHost, s), "D2H") ||
            !hip_ok(hipMemcpyAsync(conv.data(), morph_done_flags(ws, k, hh, ww), k * sizeof(int), hipMemcpyDeviceToHost, s), "D2H") ||
            !hip_ok(hipStreamSynchronize(s), "sync")) { rc = TMAT_E_HIP; break; }
        for (int i = 0; i < k && !rc; i++) if (!conv[i]) { set_error("tmat_postprocess_batch: thinning did not converge"); rc = TMAT_E_HIP; }
        if (rc) break;
        if (c->thin_device && thin_dev_supported(hh, ww)) {
            rc = medial_thin_batch_dev(c, df, dd, k, hh, ww, dsk, s);
            if (rc) break;
        } else {        // images too large for the LDS-resident thinning kernel: host threads
            parallel_images(k, [&](int i) { medial_axis_thin(filt.data() + i * per, dist.data() + i * per, hh, ww, skel.data() + i * per); });
            if (!hip_ok(hipMemcpyAsync(dsk, skel.data(), k * per, hipMemcpyHostToDevice, s), "H2D")) { rc = TMAT_E_HIP; break; }
        }
        if (finish_dev(dp, dd, dsk, k, hh, ww, out_h, out_w, fws, dfield, d255, s)) { rc = TMAT_E_HIP; break; }
        if (!hip_ok(hipMemcpyAsync(field + (size_t)i0 * oper, dfield, k * oper * 4, hipMemcpyDeviceToHost, s), "D2H") ||
            !hip_ok(hipStreamSynchronize(s), "sync")) rc = TMAT_E_HIP;
    }
    hipFree(dp); hipFree(dd); hipFree(df); hipFree(dsk); hipFree(ws); hipFree(fws); hipFree(dfield); hipFree(d255);
    return rc;
}

int tmat_medial_axis_batch(tmat_handle hd, const uint8_t *mask, int n, int hh, int ww, uint8_t *skel, double *dist)
{
    Ctx *c = (Ctx *)hd;
    if (!c || !mask || !skel || !dist || n < 0 || hh < 1 || ww < 1) { set_error("tmat_medial_axis_batch: bad argument"); return TMAT_E_ARG; }
    if (n == 0) return TMAT_OK;
    if (!thin_dev_supported(hh, ww)) { set_error("tmat_medial_axis_batch: image too large for the device thinning kernel (use tmat_host_medial_axis)"); return TMAT_E_ARG; }
    TMAT_HIP(hipSetDevice(c->device));
    const size_t npx = (size_t)n * hh * ww;
    uint8_t *dm = nullptr, *dsk = nullptr; double *dd = nullptr; int *g = nullptr, *anyz = nullptr;
    int rc = TMAT_OK;
    hipStream_t s = c->stream;
    if (!hip_ok(hipMalloc((void **)&dm, npx), "hipMalloc") || !hip_ok(hipMalloc((void **)&dsk, npx), "hipMalloc") ||
        !hip_ok(hipMalloc((void **)&dd, npx * 8), "hipMalloc") || !hip_ok(hipMalloc((void **)&g, npx * sizeof(int)), "hipMalloc") ||
        !hip_ok(hipMalloc((void **)&anyz, n * sizeof(int)), "hipMalloc")) rc = TMAT_E_HIP;
    if (!rc && !hip_ok(hipMemcpyAsync(dm, mask, npx, hipMemcpyHostToDevice, s), "H2D")) rc = TMAT_E_HIP;
    if (!rc) launch_edt(dm, n, hh, ww, g, nullptr, anyz, dd, s);
    if (!rc) rc = medial_thin_batch_dev(c, dm, dd, n, hh, ww, dsk, s);
    if (!rc && (!hip_ok(hipMemcpyAsync(skel, dsk, npx, hipMemcpyDeviceToHost, s), "D2H") ||
                !hip_ok(hipMemcpyAsync(dist, dd, npx * 8, hipMemcpyDeviceToHost, s), "D2H") || !hip_ok(hipStreamSynchronize(s), "sync"))) rc = TMAT_E_HIP;
    hipFree(dm); hipFree(dsk); hipFree(dd); hipFree(g); hipFree(anyz);
    return rc;
}

int tmat_analyze_batch_dev(tmat_handle hd, const uint16_t *imgs_dev, int n, int H, int W, double ds_ratio, int ds_width,
                           float graph_thresh_1, float graph_thresh_2, int smoothing_window_px, int min_branch_length_px,
                           int max_branch_length_px, int remove_isolated, int64_t first_index, tmat_row *rows)
{
    Ctx *c = (Ctx *)hd;
    if (c && !has_model(c)) { set_error("tmat_analyze_batch_dev: this handle has no model (tmat_create_plain)"); return TMAT_E_ARG; }
    if (!c || !imgs_dev || !rows || n < 0 || H < 1 || W < 1 || ds_width < 1) { set_error("tmat_analyze_batch_dev: bad argument"); return TMAT_E_ARG; }
    if (n == 0) return TMAT_OK;
    TMAT_HIP(hipSetDevice(c->device));
    GraphParams gp{0, 0, graph_thresh_1, graph_thresh_2, smoothing_window_px, min_branch_length_px, max_branch_length_px, remove_isolated};
    return analyze_dev(c, imgs_dev, n, H, W, ds_ratio, ds_width, gp, first_index, rows);
}

int tmat_analyze_batch(tmat_handle hd, const uint16_t *imgs, int n, int H, int W, double ds_ratio, int ds_width,
                       float graph_thresh_1, float graph_thresh_2, int smoothing_window_px, int min_branch_length_px,
                       int max_branch_length_px, int remove_isolated, int64_t first_index, tmat_row *rows)
{
    Ctx *c = (Ctx *)hd;
    if (c && !has_model(c)) { set_error("tmat_analyze_batch: this handle has no model (tmat_create_plain)"); return TMAT_E_ARG; }
    if (!c || !imgs || !rows || n < 0) { set_error("tmat_analyze_batch: bad argument"); return TMAT_E_ARG; }
    if (n == 0) return TMAT_OK;
    TMAT_HIP(hipSetDevice(c->device));
    uint16_t *dimg = nullptr;
    TMAT_HIP(hipMalloc((void **)&dimg, (size_t)n * H * W * sizeof(uint16_t)));
    int rc = TMAT_OK;
    if (!hip_ok(hipMemcpy(dimg, imgs, (size_t)n * H * W * 2, hipMemcpyHostToDevice), "H2D")) rc = TMAT_E_HIP;
    if (!rc) rc = tmat_analyze_batch_dev(hd, dimg, n, H, W, ds_ratio, ds_width, graph_thresh_1, graph_thresh_2, smoothing_window_px,
                                         min_branch_length_px, max_branch_length_px, remove_isolated, first_index, rows);
    hipFree(dimg);
    return rc;
}

}  // extern "C"
