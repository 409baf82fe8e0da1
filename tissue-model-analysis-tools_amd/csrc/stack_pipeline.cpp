// Driver of the Z-stack (Sato) branch of analyze_img (reference scripts/compute_branches.py:224-306 and the common tail
// :391-457) behind include/tmat.h: tmat_gaussian_f32, tmat_sato_batch, tmat_stack_prepare, tmat_vessel_field,
// tmat_analyze_stack.  The pixel stages are the kernels of sato_kernels.hip plus the medial axis (thin_kernels.hip), the
// mask filter (morph_kernels.hip) and the DMT front end (dmt_kernels.hip) of the 2-D path; the host contributes the gaussian
// tables (gauss_tables.cpp), the medial axis' tie-break permutation and the sequential graph sweeps (dmt.cpp, morse.cpp).
#include "../../include/tmat.h"
#include "tmat_ctx.h"
#include "postproc.h"
#include "morph.h"
#include "sato.h"
#include "wellmask.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace tmat {

int medial_thin_batch_dev(Ctx *c, const uint8_t *mask_dev, const double *dist_dev, int k, int hh, int ww, uint8_t *skel_dev, hipStream_t s);
void zoom_axis_table(int n_in, int n_out, std::vector<int> &i0, std::vector<int> &i1, std::vector<double> &a0, std::vector<double> &a1);

namespace {

// device allocations of one call, released together -- back into the handle's pool (Ctx::ws_pool: blocks by size), from which the next call
// takes them again: a Z-stack call makes some thirty allocations of a dozen sizes, and a hipMalloc / hipFree pair per block was a tenth of
// its time.  The pool only ever holds what one call of each geometry needs; tmat_destroy frees it.
struct Arena {
    Ctx *c;
    std::vector<std::pair<void *, size_t>> blocks;
    std::vector<void *> ptrs;               // blocks that are not the pool's (morph / DMT workspaces allocated by the caller): freed
    bool ok = true;
    explicit Arena(Ctx *ctx) : c(ctx) {}
    template <typename T> T *get(size_t count)
    {
        if (!ok) return nullptr;
        const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
        return (T *)get_bytes(bytes);
    }
    void *get_bytes(size_t bytes)
    {
        if (!ok) return nullptr;
        auto it = c->ws_pool.lower_bound(bytes);
        if (it != c->ws_pool.end() && it->first <= bytes + bytes / 4 + 4096) {          // close enough in size: reuse
            void *p = it->second;
            blocks.push_back({p, it->first});
            c->ws_pool.erase(it);
            return p;
        }
        void *p = nullptr;
        if (!hip_ok(hipMalloc(&p, bytes), "hipMalloc")) { ok = false; return nullptr; }
        blocks.push_back({p, bytes});
        return p;
    }
    hipStream_t drain = nullptr;            // synchronised before anything is released: an early error return must not leave async work behind
    ~Arena()
    {
        if (drain) hipStreamSynchronize(drain);
        for (auto &b : blocks) c->ws_pool.insert({b.second, b.first});
        for (void *p : ptrs) hipFree(p);
    }
};

// device copy of a gaussian table, made on first use (synchronous copy: a few hundred doubles)
const double *table_dev(Ctx *c, const GaussTable &t)
{
    auto it = c->gauss_dev.find(&t);
    if (it != c->gauss_dev.end()) return it->second;
    double *d = nullptr;
    if (!hip_ok(hipMalloc((void **)&d, t.w.size() * sizeof(double)), "hipMalloc")) return nullptr;
    if (!hip_ok(hipMemcpy(d, t.w.data(), t.w.size() * sizeof(double), hipMemcpyHostToDevice), "H2D")) { hipFree(d); return nullptr; }
    c->gauss_dev[&t] = d;
    return d;
}

struct Pass { double sigma; int order; int L, inner; size_t outer; };

// one 1-D pass of ndi.gaussian_filter on f32 data
bool gauss_pass_f32(Ctx *c, const float *in, float *out, const Pass &p, double truncate, int mode, hipStream_t s)
{
    const GaussTable &t = gauss_table(c, p.sigma, p.order, gauss_radius(p.sigma, truncate));
    const double *w = table_dev(c, t);
    if (!w) return false;
    launch_corr1d_f32(in, out, p.outer, p.L, p.inner, w, t.r, t.sym, mode, s);
    return true;
}

// ndi.gaussian_filter(x, sigma, order=(o0, o1), mode, truncate) on n images (h, w): axis 0 then axis 1, f32 after each pass
bool gauss2d_f32(Ctx *c, const float *in, float *tmp, float *out, int n, int h, int w, double sigma, int o0, int o1, double truncate, int mode,
                 hipStream_t s)
{
    return gauss_pass_f32(c, in, tmp, Pass{sigma, o0, h, w, (size_t)n}, truncate, mode, s) &&
           gauss_pass_f32(c, tmp, out, Pass{sigma, o1, w, 1, (size_t)n * h}, truncate, mode, s);
}

// skimage.filters.sato(im, sigmas, black_ridges=False) on n images; x = the prepared input (1 - im or -im), best = result.
// bufs: 7 arrays of n h w floats
bool sato_dev(Ctx *c, const float *x, int n, int h, int w, const double *sigmas, int nsig, int form, float *const bufs[7], float *best, hipStream_t s)
{
    const size_t total = (size_t)n * h * w;
    float *T = bufs[0], *G0 = bufs[1], *G1 = bufs[2], *H0 = bufs[3], *H1 = bufs[4], *H2 = bufs[5], *T2 = bufs[6];
    if (nsig == 0) return hip_ok(hipMemsetAsync(best, 0, total * sizeof(float), s), "memset");
    for (int k = 0; k < nsig; k++) {
        const double sg = sigmas[k];
        const float s2 = (float)(sg * sg);
        if (form == TMAT_SATO_GAUSSIAN_DERIVATIVES) {
            // hessian_matrix(use_gaussian_derivatives=True): sigma / sqrt(2) twice, truncate 8 (100 when sigma <= 1), 'reflect'
            const double sq1_2 = 1.0 / std::sqrt(2.0);
            const double ss = sq1_2 * sg, tr = sg > 1.0 ? 8.0 : 100.0;
            if (!gauss2d_f32(c, x, T, G0, n, h, w, ss, 1, 0, tr, EXT_REFLECT, s) || !gauss2d_f32(c, x, T, G1, n, h, w, ss, 0, 1, tr, EXT_REFLECT, s) ||
                !gauss2d_f32(c, G0, T, H0, n, h, w, ss, 1, 0, tr, EXT_REFLECT, s) || !gauss2d_f32(c, G0, T, H1, n, h, w, ss, 0, 1, tr, EXT_REFLECT, s) ||
                !gauss2d_f32(c, G1, T, H2, n, h, w, ss, 0, 1, tr, EXT_REFLECT, s)) return false;
            launch_eig(1, H0, H1, H2, s2, total, best, k == 0, s);
        } else {
            // hessian_matrix of scikit-image <= 0.19: gaussian (truncate 4, 'reflect'), np.gradient twice, elements scaled by sigma^2
            if (!gauss2d_f32(c, x, T, G0, n, h, w, sg, 0, 0, 4.0, EXT_REFLECT, s)) return false;
            launch_gradient(G0, G1, (size_t)n, h, w, s);              // d/dr
            launch_gradient(G0, H0, (size_t)n * h, w, 1, s);          // d/dc
            launch_gradient(G1, H1, (size_t)n, h, w, s);              // Hrr
            launch_gradient(H0, H2, (size_t)n, h, w, s);              // Hrc = d/dr of d/dc
            launch_gradient(H0, T2, (size_t)n * h, w, 1, s);          // Hcc
            launch_eig(0, H1, H2, T2, s2, total, best, k == 0, s);
        }
    }
    return hipGetLastError() == hipSuccess;
}

const double SATO_SIGMAS[10] = {1, 2, 3, 4, 5, 7, 9, 11, 13, 15};       // compute_branches.py:262

bool d2h(void *dst, const void *src, size_t bytes, hipStream_t s) { return !dst || hip_ok(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s), "D2H"); }

}  // namespace

// vol (Z, h, w) f32 on the device -> field (h, w) f32 on the device; stage copies go to the host arrays of `st` that are non-null
int vessel_field_dev(Ctx *c, const float *vol, int Z, int h, int w, int form, float *field, const tmat_vessel_stages *st, hipStream_t s)
{
    if (Z < 2) { set_error("vessel field: a Z stack needs at least 2 slices"); return TMAT_E_ARG; }
    if (!thin_dev_supported(h, w)) { set_error("vessel field: image too large for the device thinning kernel"); return TMAT_E_ARG; }
    const int D = Z - 1;
    const size_t npx = (size_t)h * w, nv = (size_t)D * npx;
    Arena A(c);
    A.drain = s;
    float *x = A.get<float>(nv), *vess = A.get<float>(nv), *bufs[7];
    for (float *&b : bufs) b = A.get<float>(nv);
    float *vessels = A.get<float>(npx), *vcur = A.get<float>(npx), *vblur = A.get<float>(npx), *vtmp = A.get<float>(npx);
    CannyWs cw{};
    cw.sm = A.get<double>(npx); cw.t0 = A.get<double>(npx); cw.is_ = A.get<double>(npx); cw.js = A.get<double>(npx); cw.mag = A.get<double>(npx);
    cw.low = A.get<uint8_t>(npx); cw.high = A.get<uint8_t>(npx); cw.L = A.get<int>(npx); cw.flag = A.get<int>(npx);
    double *tabs = A.get<double>(6);
    uint8_t *edges = A.get<uint8_t>(npx), *skel = A.get<uint8_t>(npx), *m0 = A.get<uint8_t>(npx), *m1 = A.get<uint8_t>(npx), *filt = A.get<uint8_t>(npx);
    double *dist = A.get<double>(npx);
    int *edt_g = A.get<int>(npx), *anyz = A.get<int>(4);
    unsigned long long *mom = A.get<unsigned long long>(npx * 6);
    int *offs = A.get<int>(2 * (13 + 9));
    void *mws = nullptr;
    mws = A.get_bytes(morph_workspace_bytes(1, h, w));
    if (!A.ok) return TMAT_E_HIP;
    {
        const double t[6] = {-1.0, 0.0, 1.0, 1.0, 2.0, 1.0};
        int o[2 * 22], k = 0;
        for (int dy = -2; dy <= 2; dy++) for (int dx = -2; dx <= 2; dx++) if (dy * dy + dx * dx <= 4) { o[2 * k] = dy; o[2 * k + 1] = dx; k++; }   // disk(2): 13
        for (int dy = -1; dy <= 1; dy++) for (int dx = -1; dx <= 1; dx++) { o[2 * k] = dy; o[2 * k + 1] = dx; k++; }                                // square(3): 9
        TMAT_HIP(hipMemcpyAsync(tabs, t, sizeof(t), hipMemcpyHostToDevice, s));
        TMAT_HIP(hipMemcpyAsync(offs, o, sizeof(o), hipMemcpyHostToDevice, s));
        TMAT_HIP(hipStreamSynchronize(s));           // the two host arrays leave scope
    }
    cw.w_diff = tabs; cw.w_smooth = tabs + 3;
    const bool deriv = form == TMAT_SATO_GAUSSIAN_DERIVATIVES;

    // z4: Sato of max(slice z, slice z + 1), all D pairs in one launch per pass
    launch_pairmax(vol, D, npx, deriv ? 1 : 0, x, s);
    if (!sato_dev(c, x, D, h, w, SATO_SIGMAS, 10, form, bufs, vess, s)) { set_error("vessel field: Sato stage failed"); return TMAT_E_HIP; }
    if (st && !d2h(st->vess, vess, nv * 4, s)) return TMAT_E_HIP;
    // z5: unsharp_mask(volume, radius 2, amount 2): 3-D gaussian ('reflect', truncate 4) over Z, rows, columns
    float *sharp = x;                                                        // x is free from here on
    if (!gauss_pass_f32(c, vess, bufs[0], Pass{2.0, 0, D, (int)npx, 1}, 4.0, EXT_REFLECT, s) ||
        !gauss_pass_f32(c, bufs[0], bufs[1], Pass{2.0, 0, h, w, (size_t)D}, 4.0, EXT_REFLECT, s) ||
        !gauss_pass_f32(c, bufs[1], bufs[0], Pass{2.0, 0, w, 1, (size_t)D * h}, 4.0, EXT_REFLECT, s)) return TMAT_E_HIP;
    launch_unsharp(vess, bufs[0], 2.0f, nv, sharp, s);
    launch_zmax(sharp, D, npx, vessels, s);
    if (st && (!d2h(st->sharp, sharp, nv * 4, s) || !d2h(st->vessels, vessels, npx * 4, s))) return TMAT_E_HIP;
    // z6: canny
    if (canny0_dev(vessels, h, w, cw, edges, s)) { set_error("vessel field: canny stage failed"); return TMAT_E_HIP; }
    if (st && !d2h(st->edges, edges, npx, s)) return TMAT_E_HIP;
    // z7: medial axis of the edges; keep skeleton components with eccentricity * equivalent diameter > 3.5
    launch_edt(edges, 1, h, w, edt_g, nullptr, anyz, dist, s);
    { int rc = medial_thin_batch_dev(c, edges, dist, 1, h, w, skel, s); if (rc) return rc; }
    if (ecc_diam_select_dev(skel, h, w, 3.5, cw.L, mom, m0, nullptr, s)) return TMAT_E_HIP;
    if (st && (!d2h(st->skel, skel, npx, s) || !d2h(st->mask_sel, m0, npx, s))) return TMAT_E_HIP;
    // z8: three masked blurs of the projection
    TMAT_HIP(hipMemcpyAsync(vcur, vessels, npx * 4, hipMemcpyDeviceToDevice, s));
    for (int it = 0; it < 3; it++) {
        if (!gauss2d_f32(c, vcur, vtmp, vblur, 1, h, w, 1.0, 0, 0, 4.0, EXT_NEAREST, s)) return TMAT_E_HIP;
        launch_where(m0, vblur, vcur, npx, vtmp, s);
        std::swap(vcur, vtmp);
    }
    //     ten region-growing rounds, mask &= ~edges, closing with disk(2)
    uint8_t *ma = m0, *mb = m1;
    for (int it = 0; it < 10; it++) { launch_grow(ma, vcur, h, w, mb, s); std::swap(ma, mb); }
    if (st && !d2h(st->grown, ma, npx, s)) return TMAT_E_HIP;
    launch_andnot(ma, edges, npx, mb, s);
    launch_morph(mb, h, w, offs, 13, 0, ma, s);
    launch_morph(ma, h, w, offs, 13, 1, mb, s);
    if (st && !d2h(st->closed, mb, npx, s)) return TMAT_E_HIP;
    // z9: filter_branch_seg_mask(mask, None, False), dilation with square(3), mask the sharpened projection, gaussian
    if (filter_mask_dev(nullptr, mb, 1, h, w, 0, 0, mws, filt, nullptr, s)) return TMAT_E_HIP;
    int conv = 0;
    TMAT_HIP(hipMemcpyAsync(&conv, morph_done_flags(mws, 1, h, w), sizeof(int), hipMemcpyDeviceToHost, s));
    launch_morph(filt, h, w, offs + 26, 9, 0, ma, s);
    launch_where(ma, vessels, nullptr, npx, vtmp, s);
    if (!gauss2d_f32(c, vtmp, vblur, field, 1, h, w, 1.0, 0, 0, 4.0, EXT_NEAREST, s)) return TMAT_E_HIP;
    if (st && !d2h(st->filt, filt, npx, s)) return TMAT_E_HIP;
    TMAT_HIP(hipStreamSynchronize(s));
    if (hipGetLastError() != hipSuccess) { set_error("vessel field: kernel launch failed"); return TMAT_E_HIP; }
    if (!conv) { set_error("vessel field: thinning did not converge"); return TMAT_E_HIP; }
    return TMAT_OK;
}

// stack (Z, H, W) u16 on the device (overwritten by its per-slice gaussian) -> vol (Z, oh, ow) f32 on the device
// skimage.transform.resize(stack, (Z, oh, ow), order=1, preserve_range=True, anti_aliasing=True) of an integer stack (scikit-image
// >= 0.19: ndi.gaussian_filter(sigma (factor - 1) / 2, 'mirror') + ndi.zoom(order 1, grid_mode) on the 3-D array, the Z axis with sigma 0
// and zoom 1), clipped to the stack's range: the float64 values in `zoomed` (device, Z oh ow); with `vol` also rescale_intensity(0..1) over
// the whole stack as float32
int stack_resize_aa_dev(Ctx *c, const uint16_t *stack, int Z, int H, int W, int oh, int ow, double *zoomed, float *vol, hipStream_t s)
{
    const size_t nin = (size_t)Z * H * W;
    Arena A(c);
    A.drain = s;
    double *fa = A.get<double>(nin), *fb = A.get<double>(nin), *lohi = A.get<double>(4);
    unsigned long long *mm = A.get<unsigned long long>(2);
    std::vector<int> r0, r1, c0, c1;
    std::vector<double> wr0, wr1, wc0, wc1;
    zoom_axis_table(H, oh, r0, r1, wr0, wr1);
    zoom_axis_table(W, ow, c0, c1, wc0, wc1);
    int *dr0 = A.get<int>(oh), *dr1 = A.get<int>(oh), *dc0 = A.get<int>(ow), *dc1 = A.get<int>(ow);
    double *dwr0 = A.get<double>(oh), *dwr1 = A.get<double>(oh), *dwc0 = A.get<double>(ow), *dwc1 = A.get<double>(ow);
    if (!A.ok) return TMAT_E_HIP;
    TMAT_HIP(hipMemcpyAsync(dr0, r0.data(), oh * 4, hipMemcpyHostToDevice, s)); TMAT_HIP(hipMemcpyAsync(dr1, r1.data(), oh * 4, hipMemcpyHostToDevice, s));
    TMAT_HIP(hipMemcpyAsync(dc0, c0.data(), ow * 4, hipMemcpyHostToDevice, s)); TMAT_HIP(hipMemcpyAsync(dc1, c1.data(), ow * 4, hipMemcpyHostToDevice, s));
    TMAT_HIP(hipMemcpyAsync(dwr0, wr0.data(), oh * 8, hipMemcpyHostToDevice, s)); TMAT_HIP(hipMemcpyAsync(dwr1, wr1.data(), oh * 8, hipMemcpyHostToDevice, s));
    TMAT_HIP(hipMemcpyAsync(dwc0, wc0.data(), ow * 8, hipMemcpyHostToDevice, s)); TMAT_HIP(hipMemcpyAsync(dwc1, wc1.data(), ow * 8, hipMemcpyHostToDevice, s));
    // anti-aliasing gaussian over rows and columns ('mirror', sigma (factor - 1) / 2), zoom, clip to the stack's range
    const double f0 = (double)H / (double)oh, f1 = (double)W / (double)ow;
    const double s0 = std::max(0.0, (f0 - 1) / 2), s1 = std::max(0.0, (f1 - 1) / 2);
    const double *cur = nullptr;
    if (s0 > 1e-15) {          // scipy skips an axis whose sigma is <= 1e-15
        const GaussTable &t = gauss_table(c, s0, 0, gauss_radius(s0, 4.0));
        const double *wd = table_dev(c, t);
        if (!wd) return TMAT_E_HIP;
        launch_corr1d_u16_f64(stack, fa, (size_t)Z, H, W, wd, t.r, t.sym, EXT_MIRROR, s);
        cur = fa;
    }
    if (s1 > 1e-15) {
        const GaussTable &t = gauss_table(c, s1, 0, gauss_radius(s1, 4.0));
        const double *wd = table_dev(c, t);
        if (!wd) return TMAT_E_HIP;
        if (cur) launch_corr1d_f64(cur, fb, (size_t)Z * H, W, 1, wd, t.r, t.sym, EXT_MIRROR, s);
        else launch_corr1d_u16_f64(stack, fb, (size_t)Z * H, W, 1, wd, t.r, t.sym, EXT_MIRROR, s);
        cur = fb;
    }
    if (!cur) {               // no smoothing at all (an image at most as wide as the target): the zoom reads the stack as f64
        const double one = 1.0;
        double *w_id = lohi;   // borrowed for a moment: a 1-tap identity kernel
        TMAT_HIP(hipMemcpyAsync(w_id, &one, 8, hipMemcpyHostToDevice, s));
        TMAT_HIP(hipStreamSynchronize(s));
        launch_corr1d_u16_f64(stack, fa, (size_t)Z * H, W, 1, w_id, 0, 1, EXT_MIRROR, s);
        cur = fa;
    }
    if (stack_zoom_rescale_dev(cur, stack, Z, H, W, oh, ow, dr0, dr1, dwr0, dwr1, dc0, dc1, dwc0, dwc1, zoomed, mm, lohi, vol, s)) {
        set_error("stack resize: kernel launch failed");
        return TMAT_E_HIP;
    }
    TMAT_HIP(hipStreamSynchronize(s));       // the host tables leave scope
    return TMAT_OK;
}

int stack_prepare_dev(Ctx *c, uint16_t *stack, int Z, int H, int W, int oh, int ow, float *vol, hipStream_t s)
{
    const size_t nin = (size_t)Z * H * W, nout = (size_t)Z * oh * ow;
    Arena A(c);
    A.drain = s;
    double *fa = A.get<double>(nin), *zoomed = A.get<double>(nout);
    if (!A.ok) return TMAT_E_HIP;
    // z1: gaussian(slice, sigma 1, 'nearest') in f64, written back into the integer stack (C truncation)
    const GaussTable &g1 = gauss_table(c, 1.0, 0, gauss_radius(1.0, 4.0));
    const double *w1 = table_dev(c, g1);
    if (!w1) return TMAT_E_HIP;
    launch_corr1d_u16_f64(stack, fa, (size_t)Z, H, W, w1, g1.r, g1.sym, EXT_NEAREST, s);
    launch_corr1d_f64_u16(fa, stack, (size_t)Z * H, W, 1, w1, g1.r, g1.sym, EXT_NEAREST, s);
    // z2 + z3: anti-aliased resize, rescale to 0..1 over the whole stack
    return stack_resize_aa_dev(c, stack, Z, H, W, oh, ow, zoomed, vol, s);
}

}  // namespace tmat

using namespace tmat;

extern "C" {

int tmat_gaussian_f32(tmat_handle hd, const float *x, int d0, int d1, int d2, double sigma, int mode, float *out)
{
    Ctx *c = (Ctx *)hd;
    if (!c || !x || !out || d0 < 1 || d1 < 1 || d2 < 1 || !(sigma > 0) || mode < 0 || mode > 2) { set_error("tmat_gaussian_f32: bad argument"); return TMAT_E_ARG; }
    TMAT_HIP(hipSetDevice(c->device));
    const size_t n = (size_t)d0 * d1 * d2;
    hipStream_t s = c->stream;
    Arena A(c);
    A.drain = s;
    float *a = A.get<float>(n), *b = A.get<float>(n);
    if (!A.ok) return TMAT_E_HIP;
    TMAT_HIP(hipMemcpyAsync(a, x, n * 4, hipMemcpyHostToDevice, s));
    // ndi.gaussian_filter: every axis in order (the caller folds leading axes of length 1 away by passing d0 = 1 -> two passes)
    float *src = a, *dst = b;
    if (d0 > 1) { if (!gauss_pass_f32(c, src, dst, Pass{sigma, 0, d0, d1 * d2, 1}, 4.0, mode, s)) return TMAT_E_HIP; std::swap(src, dst); }
    if (!gauss_pass_f32(c, src, dst, Pass{sigma, 0, d1, d2, (size_t)d0}, 4.0, mode, s)) return TMAT_E_HIP;
    std::swap(src, dst);
    if (!gauss_pass_f32(c, src, dst, Pass{sigma, 0, d2, 1, (size_t)d0 * d1}, 4.0, mode, s)) return TMAT_E_HIP;
    TMAT_HIP(hipMemcpyAsync(out, dst, n * 4, hipMemcpyDeviceToHost, s));
    TMAT_HIP(hipStreamSynchronize(s));
    return TMAT_OK;
}

/* well_mask_generation.auto_threshold_well (reference :236-277) on the device: img (H, W) -> thresholded, eroded mask u8.
 * is_f64 = 0: a float32 image (what compute_branches.py hands over); 1: float64 (integer images after img_as_float) */
static int well_threshold(tmat_handle hd, const void *img, int is_f64, int H, int W, uint8_t *out)
{
    Ctx *c = (Ctx *)hd;
    if (!c || !img || !out || H < 20 || W < 20 || (long long)H * W > (1LL << 28)) { set_error("tmat_well_threshold: bad argument (images of at least 20 x 20)"); return TMAT_E_ARG; }
    TMAT_HIP(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const size_t n = (size_t)H * W;
    Arena A(c);
    A.drain = s;
    float *a = A.get<float>(is_f64 ? 1 : n), *b = A.get<float>(is_f64 ? 1 : n), *mm = A.get<float>(2);
    double *da = A.get<double>(is_f64 ? n : 1), *db = A.get<double>(is_f64 ? n : 1), *dmm = A.get<double>(2);
    uint8_t *u8 = A.get<uint8_t>(n), *th = A.get<uint8_t>(n), *er = A.get<uint8_t>(n);
    unsigned *hist = A.get<unsigned>(5 * 256);
    int *decision = A.get<int>(2), *offs = A.get<int>(2 * 81);
    if (!A.ok) return TMAT_E_HIP;
    int o[2 * 81], k = 0;
    for (int dy = -5; dy <= 5; dy++) for (int dx = -5; dx <= 5; dx++) if (dy * dy + dx * dx <= 25) { o[2 * k] = dy; o[2 * k + 1] = dx; k++; }     // disk(5): 81
    TMAT_HIP(hipMemcpyAsync(offs, o, sizeof(int) * 2 * k, hipMemcpyHostToDevice, s));
    // gaussian(image, sigma=1): ndi.gaussian_filter, mode 'nearest', truncate 4, the image's float type after each axis
    if (is_f64) {
        TMAT_HIP(hipMemcpyAsync(da, img, n * 8, hipMemcpyHostToDevice, s));
        const GaussTable &gt = gauss_table(c, 1.0, 0, gauss_radius(1.0, 4.0));
        const double *w = table_dev(c, gt);
        if (!w) return TMAT_E_HIP;
        launch_corr1d_f64(da, db, 1, H, W, w, gt.r, gt.sym, EXT_NEAREST, s);
        launch_corr1d_f64(db, da, (size_t)H, W, 1, w, gt.r, gt.sym, EXT_NEAREST, s);
        launch_wm_rescale_u8_f64(da, n, dmm, u8, s);
    } else {
        TMAT_HIP(hipMemcpyAsync(a, img, n * 4, hipMemcpyHostToDevice, s));
        if (!gauss_pass_f32(c, a, b, Pass{1.0, 0, H, W, 1}, 4.0, EXT_NEAREST, s) || !gauss_pass_f32(c, b, a, Pass{1.0, 0, W, 1, (size_t)H}, 4.0, EXT_NEAREST, s))
            return TMAT_E_HIP;
        launch_minmax_f32(a, 1, n, mm, mm + 1, s);
        launch_wm_rescale_u8(a, n, mm, mm + 1, u8, s);
    }
    launch_wm_hist(u8, H, W, hist, s);
    launch_wm_decide(hist, H, W, decision, s);
    launch_wm_threshold(u8, n, decision, th, s);
    launch_wm_erode(th, H, W, offs, k, er, s);
    TMAT_HIP(hipGetLastError());
    TMAT_HIP(hipMemcpyAsync(out, er, n, hipMemcpyDeviceToHost, s));
    TMAT_HIP(hipStreamSynchronize(s));
    return TMAT_OK;
}

int tmat_well_threshold(tmat_handle hd, const float *img, int H, int W, uint8_t *out) { return well_threshold(hd, img, 0, H, W, out); }
int tmat_well_threshold_f64(tmat_handle hd, const double *img, int H, int W, uint8_t *out) { return well_threshold(hd, img, 1, H, W, out); }

/* skimage.feature.canny(mask, sigma) of a boolean image with the default thresholds (reference calls:
 * well_mask_generation.py:165, :201): mask (H, W) u8 -> edges u8 */
int tmat_canny_mask(tmat_handle hd, const uint8_t *mask, int H, int W, double sigma, uint8_t *edges)
{
    Ctx *c = (Ctx *)hd;
    if (!c || !mask || !edges || H < 3 || W < 3 || !(sigma > 0) || sigma > 16 || (long long)H * W > (1LL << 26)) { set_error("tmat_canny_mask: bad argument"); return TMAT_E_ARG; }
    TMAT_HIP(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const int r = gauss_radius(sigma, 4.0);
    const int Hp = H + 2 * r, Wp = W + 2 * r;
    const size_t npx = (size_t)H * W, npad = (size_t)Hp * Wp;
    Arena A(c);
    A.drain = s;
    uint8_t *m = A.get<uint8_t>(npx), *e = A.get<uint8_t>(npx);
    double *pa = A.get<double>(npad), *pb = A.get<double>(npad), *pt = A.get<double>(npad);
    CannyWs cw{};
    cw.sm = A.get<double>(npx); cw.t0 = A.get<double>(npx); cw.is_ = A.get<double>(npx); cw.js = A.get<double>(npx); cw.mag = A.get<double>(npx);
    cw.low = A.get<uint8_t>(npx); cw.high = A.get<uint8_t>(npx); cw.L = A.get<int>(npx); cw.flag = A.get<int>(npx);
    double *tabs = A.get<double>(6);
    if (!A.ok) return TMAT_E_HIP;
    const double t[6] = {-1.0, 0.0, 1.0, 1.0, 2.0, 1.0};
    TMAT_HIP(hipMemcpyAsync(tabs, t, sizeof(t), hipMemcpyHostToDevice, s));
    cw.w_diff = tabs; cw.w_smooth = tabs + 3;
    TMAT_HIP(hipMemcpyAsync(m, mask, npx, hipMemcpyHostToDevice, s));
    // gaussian(x, sigma, mode='constant') of the image and of the all-ones mask (0.18.3 smooth_with_function_and_mask): zero padding
    // by the kernel radius makes the boundary mode irrelevant
    launch_wm_pad(m, H, W, r, pa, pb, s);
    const GaussTable &gt = gauss_table(c, sigma, 0, r);
    const double *w = table_dev(c, gt);
    if (!w) return TMAT_E_HIP;
    launch_corr1d_f64(pa, pt, 1, Hp, Wp, w, gt.r, gt.sym, EXT_NEAREST, s);
    launch_corr1d_f64(pt, pa, (size_t)Hp, Wp, 1, w, gt.r, gt.sym, EXT_NEAREST, s);
    launch_corr1d_f64(pb, pt, 1, Hp, Wp, w, gt.r, gt.sym, EXT_NEAREST, s);
    launch_corr1d_f64(pt, pb, (size_t)Hp, Wp, 1, w, gt.r, gt.sym, EXT_NEAREST, s);
    launch_wm_crop_div(pa, pb, H, W, r, cw.sm, s);
    if (canny_core_dev(H, W, cw, e, s)) { set_error("tmat_canny_mask: kernel launch failed"); return TMAT_E_HIP; }
    TMAT_HIP(hipMemcpyAsync(edges, e, npx, hipMemcpyDeviceToHost, s));
    TMAT_HIP(hipStreamSynchronize(s));
    return TMAT_OK;
}

int tmat_sato_batch(tmat_handle hd, const float *imgs, int n, int hh, int ww, const double *sigmas, int n_sigmas, int hessian, float *out)
{
    Ctx *c = (Ctx *)hd;
    if (!c || !imgs || !out || n < 0 || hh < 2 || ww < 2 || n_sigmas < 0 || (n_sigmas && !sigmas) ||
        (hessian != TMAT_SATO_GAUSSIAN_DERIVATIVES && hessian != TMAT_SATO_GRADIENT)) { set_error("tmat_sato_batch: bad argument"); return TMAT_E_ARG; }
    for (int k = 0; k < n_sigmas; k++) if (!(sigmas[k] > 0)) { set_error("tmat_sato_batch: sigmas must be positive"); return TMAT_E_ARG; }
    if (n == 0) return TMAT_OK;
    TMAT_HIP(hipSetDevice(c->device));
    const size_t total = (size_t)n * hh * ww;
    hipStream_t s = c->stream;
    Arena A(c);
    A.drain = s;
    float *raw = A.get<float>(total), *x = A.get<float>(total), *best = A.get<float>(total), *bufs[7];
    for (float *&b : bufs) b = A.get<float>(total);
    if (!A.ok) return TMAT_E_HIP;
    TMAT_HIP(hipMemcpyAsync(raw, imgs, total * 4, hipMemcpyHostToDevice, s));
    launch_prep_single(raw, total, hessian == TMAT_SATO_GAUSSIAN_DERIVATIVES, x, s);
    if (!sato_dev(c, x, n, hh, ww, sigmas, n_sigmas, hessian, bufs, best, s)) { set_error("tmat_sato_batch: kernel launch failed"); return TMAT_E_HIP; }
    TMAT_HIP(hipMemcpyAsync(out, best, total * 4, hipMemcpyDeviceToHost, s));
    TMAT_HIP(hipStreamSynchronize(s));
    return TMAT_OK;
}

int tmat_stack_prepare(tmat_handle hd, const uint16_t *stack, int Z, int H, int W, int out_h, int out_w, float *vol)
{
    Ctx *c = (Ctx *)hd;
    if (!c || !stack || !vol || Z < 1 || H < 1 || W < 1 || out_h < 1 || out_w < 1) { set_error("tmat_stack_prepare: bad argument"); return TMAT_E_ARG; }
    TMAT_HIP(hipSetDevice(c->device));
    const size_t nin = (size_t)Z * H * W, nout = (size_t)Z * out_h * out_w;
    Arena A(c);
    A.drain = c->stream;
    uint16_t *ds = A.get<uint16_t>(nin);
    float *dv = A.get<float>(nout);
    if (!A.ok) return TMAT_E_HIP;
    TMAT_HIP(hipMemcpyAsync(ds, stack, nin * 2, hipMemcpyHostToDevice, c->stream));
    int rc = stack_prepare_dev(c, ds, Z, H, W, out_h, out_w, dv, c->stream);
    if (rc) return rc;
    TMAT_HIP(hipMemcpyAsync(vol, dv, nout * 4, hipMemcpyDeviceToHost, c->stream));
    TMAT_HIP(hipStreamSynchronize(c->stream));
    return TMAT_OK;
}

int tmat_vessel_field(tmat_handle hd, const float *vol, int Z, int hh, int ww, int hessian, float *field, const tmat_vessel_stages *stages)
{
    Ctx *c = (Ctx *)hd;
    if (!c || !vol || !field || Z < 2 || hh < 2 || ww < 2 || (hessian != TMAT_SATO_GAUSSIAN_DERIVATIVES && hessian != TMAT_SATO_GRADIENT)) {
        set_error("tmat_vessel_field: bad argument");
        return TMAT_E_ARG;
    }
    TMAT_HIP(hipSetDevice(c->device));
    const size_t nvol = (size_t)Z * hh * ww, npx = (size_t)hh * ww;
    Arena A(c);
    A.drain = c->stream;
    float *dv = A.get<float>(nvol), *df = A.get<float>(npx);
    if (!A.ok) return TMAT_E_HIP;
    TMAT_HIP(hipMemcpyAsync(dv, vol, nvol * 4, hipMemcpyHostToDevice, c->stream));
    int rc = vessel_field_dev(c, dv, Z, hh, ww, hessian, df, stages, c->stream);
    if (rc) return rc;
    TMAT_HIP(hipMemcpyAsync(field, df, npx * 4, hipMemcpyDeviceToHost, c->stream));
    TMAT_HIP(hipStreamSynchronize(c->stream));
    return TMAT_OK;
}

// common tail of analyze_img from the vesselness image in HBM: rescale_intensity(0..255) (:419), DMT graph, MorseGraph statistics
static int field_stats_dev(Ctx *c, const float *field, int fh, int fw, float t1, float t2, int smooth, int min_len, int max_len, int remove_isolated,
                           const uint8_t *pruning_mask, int64_t index, tmat_row *row, hipStream_t s)
{
    const size_t npx = (size_t)fh * fw, nE = dmt_edge_count(fh, fw);
    Arena A(c);
    A.drain = s;
    float *f255 = A.get<float>(npx), *mnmx = A.get<float>(2);
    int32_t *ids = A.get<int32_t>(nE);
    int *m = A.get<int>(1);
    void *dws = nullptr;
    dws = A.get_bytes(dmt_workspace_bytes(1, fh, fw));
    if (!A.ok) return TMAT_E_HIP;
    launch_rescale255(field, 1, (int)npx, mnmx, mnmx + 1, f255, s);
    std::vector<float> f255_host(npx);
    std::vector<int32_t> ids_host(nE);
    int m_host = 0;
    if (dmt_sorted_edges_dev(f255, 1, fh, fw, dws, ids, m, s)) { set_error("field stats: DMT front end failed"); return TMAT_E_HIP; }
    // the two persistence sweeps on the device as well (dmt_sweep_kernels.hip; TMAT_DMT_SWEEP_DEVICE=0: inside dmt_graph_host_sorted)
    std::vector<uint8_t> kind_host;
    std::vector<float> pers_host;
    if (c->dmt_sweep_device) {
        uint8_t *dkind = A.get<uint8_t>(nE);
        float *dpers = A.get<float>(nE);
        void *sws = nullptr;
        sws = A.get_bytes(dmt_sweep_workspace_bytes(1, fh, fw));
        if (!A.ok) return TMAT_E_HIP;
        if (dmt_sweeps_dev(f255, ids, m, 1, fh, fw, sws, dkind, dpers, s)) { set_error("field stats: device sweeps failed"); return TMAT_E_HIP; }
        kind_host.resize(nE); pers_host.resize(nE);
        TMAT_HIP(hipMemcpyAsync(kind_host.data(), dkind, nE, hipMemcpyDeviceToHost, s));
        TMAT_HIP(hipMemcpyAsync(pers_host.data(), dpers, nE * sizeof(float), hipMemcpyDeviceToHost, s));
    }
    TMAT_HIP(hipMemcpyAsync(f255_host.data(), f255, npx * 4, hipMemcpyDeviceToHost, s));
    TMAT_HIP(hipMemcpyAsync(ids_host.data(), ids, nE * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    TMAT_HIP(hipMemcpyAsync(&m_host, m, sizeof(int), hipMemcpyDeviceToHost, s));
    TMAT_HIP(hipStreamSynchronize(s));
    const int cap_v = (int)npx + 4, cap_e = 3 * (int)npx + 4;
    std::vector<int32_t> V((size_t)cap_v * 2), E((size_t)cap_e * 2);
    int nv = 0, ne = 0;
    int rc = dmt_graph_host_sorted(f255_host.data(), fh, fw, t1, t2, ids_host.data(), m_host, V.data(), cap_v, E.data(), cap_e, &nv, &ne,
                                   kind_host.empty() ? nullptr : kind_host.data(), pers_host.empty() ? nullptr : pers_host.data());
    row->index = index; row->count = 0; row->total_px = 0; row->avg_px = 0;
    if (!rc)
        rc = tmat_morse_stats(V.data(), nv, E.data(), ne, fh, fw, smooth, min_len, max_len, remove_isolated, pruning_mask, &row->count, &row->total_px,
                              &row->avg_px, nullptr, 0);
    return rc;
}

int tmat_field_stats_pruned(tmat_handle hd, const float *field, int fh, int fw, float graph_thresh_1, float graph_thresh_2, int smoothing_window_px,
                            int min_branch_length_px, int max_branch_length_px, int remove_isolated, const uint8_t *pruning_mask, int64_t index,
                            tmat_row *row)
{
    Ctx *c = (Ctx *)hd;
    if (!c || !field || !row || fh < 2 || fw < 2) { set_error("tmat_field_stats: bad argument"); return TMAT_E_ARG; }
    TMAT_HIP(hipSetDevice(c->device));
    Arena A(c);
    A.drain = c->stream;
    float *df = A.get<float>((size_t)fh * fw);
    if (!A.ok) return TMAT_E_HIP;
    TMAT_HIP(hipMemcpyAsync(df, field, (size_t)fh * fw * 4, hipMemcpyHostToDevice, c->stream));
    return field_stats_dev(c, df, fh, fw, graph_thresh_1, graph_thresh_2, smoothing_window_px, min_branch_length_px, max_branch_length_px, remove_isolated,
                           pruning_mask, index, row, c->stream);
}

int tmat_field_stats(tmat_handle hd, const float *field, int fh, int fw, float graph_thresh_1, float graph_thresh_2, int smoothing_window_px,
                     int min_branch_length_px, int max_branch_length_px, int remove_isolated, int64_t index, tmat_row *row)
{
    return tmat_field_stats_pruned(hd, field, fh, fw, graph_thresh_1, graph_thresh_2, smoothing_window_px, min_branch_length_px, max_branch_length_px,
                                   remove_isolated, nullptr, index, row);
}

int tmat_resize_aa_u16(tmat_handle hd, const uint16_t *imgs, int n, int H, int W, int out_h, int out_w, double *out)
{
    Ctx *c = (Ctx *)hd;
    if (!c || !imgs || !out || n < 1 || H < 1 || W < 1 || out_h < 1 || out_w < 1) { set_error("tmat_resize_aa_u16: bad argument"); return TMAT_E_ARG; }
    TMAT_HIP(hipSetDevice(c->device));
    const size_t nin = (size_t)n * H * W, nout = (size_t)n * out_h * out_w;
    Arena A(c);
    A.drain = c->stream;
    uint16_t *ds = A.get<uint16_t>(nin);
    double *dz = A.get<double>(nout);
    if (!A.ok) return TMAT_E_HIP;
    TMAT_HIP(hipMemcpyAsync(ds, imgs, nin * 2, hipMemcpyHostToDevice, c->stream));
    int rc = stack_resize_aa_dev(c, ds, n, H, W, out_h, out_w, dz, nullptr, c->stream);
    if (rc) return rc;
    TMAT_HIP(hipMemcpyAsync(out, dz, nout * 8, hipMemcpyDeviceToHost, c->stream));
    TMAT_HIP(hipStreamSynchronize(c->stream));
    return TMAT_OK;
}

int tmat_analyze_stack(tmat_handle hd, const uint16_t *stack, int Z, int H, int W, int ds_width, int hessian, float graph_thresh_1,
                       float graph_thresh_2, int smoothing_window_px, int min_branch_length_px, int max_branch_length_px, int remove_isolated,
                       int64_t index, tmat_row *row, float *field_out)
{
    Ctx *c = (Ctx *)hd;
    if (!c || !stack || !row || Z < 2 || H < 2 || W < 2 || ds_width < 2 || (hessian != TMAT_SATO_GAUSSIAN_DERIVATIVES && hessian != TMAT_SATO_GRADIENT)) {
        set_error("tmat_analyze_stack: bad argument");
        return TMAT_E_ARG;
    }
    TMAT_HIP(hipSetDevice(c->device));
    // img_dsamp_res = round(shape * ds_width / width), half to even (compute_branches.py:218-222)
    const int fh = (int)std::nearbyint((double)H * ((double)ds_width / (double)W)), fw = (int)std::nearbyint((double)W * ((double)ds_width / (double)W));
    if (fh < 2 || fw < 2) { set_error("tmat_analyze_stack: downsampled shape is empty"); return TMAT_E_ARG; }
    const size_t nin = (size_t)Z * H * W, npx = (size_t)fh * fw;
    hipStream_t s = c->stream;
    Arena A(c);
    A.drain = s;
    uint16_t *ds = A.get<uint16_t>(nin);
    float *vol = A.get<float>((size_t)Z * npx), *field = A.get<float>(npx);
    if (!A.ok) return TMAT_E_HIP;
    TMAT_HIP(hipMemcpyAsync(ds, stack, nin * 2, hipMemcpyHostToDevice, s));
    int rc = stack_prepare_dev(c, ds, Z, H, W, fh, fw, vol, s);
    if (!rc) rc = vessel_field_dev(c, vol, Z, fh, fw, hessian, field, nullptr, s);
    if (rc) return rc;
    if (field_out) TMAT_HIP(hipMemcpyAsync(field_out, field, npx * 4, hipMemcpyDeviceToHost, s));
    return field_stats_dev(c, field, fh, fw, graph_thresh_1, graph_thresh_2, smoothing_window_px, min_branch_length_px, max_branch_length_px,
                           remove_isolated, nullptr, index, row, s);
}

int tmat_host_gaussian_kernel1d(double sigma, int order, int radius, double *weights)
{
    if (!weights || !(sigma > 0) || order < 0 || order > 1 || radius < 0) { set_error("tmat_host_gaussian_kernel1d: bad argument"); return TMAT_E_ARG; }
    std::vector<double> w;
    gaussian_kernel1d(sigma, order, radius, w);
    std::memcpy(weights, w.data(), w.size() * sizeof(double));
    return TMAT_OK;
}

}  // extern "C"
