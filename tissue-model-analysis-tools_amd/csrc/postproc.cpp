// Pixel post-processing between the UNet probability map and the DMT field
// (reference scripts/compute_branches.py:334-361; fl_tissue_model_tools/transforms.py:209-361) and the
// pre-processing in front of the UNet (compute_branches.py:309-316).
//
//   a1  cv2.resize(INTER_LANCZOS4) on uint16                      -> lanczos4_resize_u16
//   a2  rescale_intensity(out_range=(0,1)).astype(f32)             -> rescale01_u16
//   a11 pred > 0.5                                                 -> postprocess_image
//   a12 filter_branch_seg_mask: 13-tap binary median, 8-connected labels, area / 4-neighbourhood
//       perimeter / circularity, Zhang thinning, skeleton components with fork test -> filter_mask
//   a14 medial_axis (exact EDT + ordered thinning, RandomState(0) tie-break)       -> medial_axis
//   a15 centre-line weighting dist / (dist + EDT(~skeleton))
//   a16 anti-aliased bilinear resize (gaussian 'mirror' + zoom grid_mode)           -> resize_aa
//   a17 rescale_intensity(out_range=(0,255)) in float32                             -> rescale255_f32
//
// Round-1 status: these stages run on the host (C++, one image per worker thread, overlapped with
// the GPU segmenting the next batch); they are the only implementation of these stages (not a
// fallback).  The thinning loops (Zhang sub-iterations to convergence, medial-axis ordered
// thinning) and component labelling are the parts that do not map to one-pass kernels; moving the
// one-pass stages (median, EDT, resize) into HIP is tracked in DESIGN.md.
// All arithmetic is ordered exactly as in oracle/morph.py (compiled with -ffp-contract=off).
#include "../../include/tmat.h"
#include "tmat_ctx.h"
#include "postproc.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace tmat {

// ---------------------------------------------------------------------------------------------
// a1: OpenCV Lanczos4 (interpolateLanczos4 + separable resize, f32 work type for 16U)
// ---------------------------------------------------------------------------------------------
static void lanczos4_coeffs(float x, float *co)
{
    static const double s45 = 0.70710678118654752440084436210485;
    static const double cs[8][2] = {{1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45}, {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
    const double y0 = -((double)x + 3) * M_PI * 0.25, s0 = std::sin(y0), c0 = std::cos(y0);
    float sum = 0.f;
    for (int i = 0; i < 8; i++) {
        const float y0_ = x + (float)(3 - i);
        if (std::fabs((double)y0_) >= 1e-6) {
            const double y = -(double)y0_ * M_PI * 0.25;
            co[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
        } else {
            co[i] = 1e30f;
        }
        sum = sum + co[i];
    }
    const float inv = 1.0f / sum;
    for (int i = 0; i < 8; i++) co[i] = co[i] * inv;
}

void lanczos_axis(int n_src, int n_dst, std::vector<int> &idx, std::vector<float> &co)
{
    idx.resize((size_t)n_dst * 8); co.resize((size_t)n_dst * 8);
    const double scale = 1.0 / ((double)n_dst / (double)n_src);
    for (int d = 0; d < n_dst; d++) {
        float fx = (float)((d + 0.5) * scale - 0.5);
        const int sx = (int)std::floor((double)fx);
        fx = fx - (float)sx;
        lanczos4_coeffs(fx, &co[(size_t)d * 8]);
        for (int k = 0; k < 8; k++) idx[(size_t)d * 8 + k] = std::min(std::max(sx - 3 + k, 0), n_src - 1);
    }
}

void lanczos4_resize_u16(const uint16_t *img, int H, int W, int h, int w, uint16_t *out)
{
    std::vector<int> xi, yi;
    std::vector<float> xc, yc;
    lanczos_axis(W, w, xi, xc);
    lanczos_axis(H, h, yi, yc);
    std::vector<float> tmp((size_t)H * w);
    for (int y = 0; y < H; y++) {
        const uint16_t *row = img + (size_t)y * W;
        for (int x = 0; x < w; x++) {
            float acc = 0.f;
            for (int k = 0; k < 8; k++) acc = acc + (float)row[xi[(size_t)x * 8 + k]] * xc[(size_t)x * 8 + k];
            tmp[(size_t)y * w + x] = acc;
        }
    }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            float acc = 0.f;
            for (int k = 0; k < 8; k++) acc = acc + tmp[(size_t)yi[(size_t)y * 8 + k] * w + x] * yc[(size_t)y * 8 + k];
            float r = std::nearbyint(acc);
            r = std::min(std::max(r, 0.0f), 65535.0f);
            out[(size_t)y * w + x] = (uint16_t)r;
        }
}

// a2: (x - min) / (max - min) in f64 -> f32 (skimage rescale_intensity on an integer image)
void rescale01_u16(const uint16_t *img, size_t n, float *out)
{
    uint16_t lo = 65535, hi = 0;
    for (size_t i = 0; i < n; i++) { lo = std::min(lo, img[i]); hi = std::max(hi, img[i]); }
    const double imin = lo, imax = hi;
    if (lo != hi)
        for (size_t i = 0; i < n; i++) out[i] = (float)((((double)img[i] - imin) / (imax - imin)) * 1.0 + 0.0);
    else
        for (size_t i = 0; i < n; i++) out[i] = (float)std::min(std::max((double)img[i], 0.0), 1.0);
}

// a17: float32 arithmetic (scalars are weak in numpy): ((x - min) / f32(max - min)) * 255 + 0
void rescale255_f32(const float *img, size_t n, float *out)
{
    float lo = std::numeric_limits<float>::infinity(), hi = -lo;
    for (size_t i = 0; i < n; i++) { lo = std::min(lo, img[i]); hi = std::max(hi, img[i]); }
    if (lo != hi) {
        const float d = (float)((double)hi - (double)lo);
        for (size_t i = 0; i < n; i++) { float t = (img[i] - lo) / d; out[i] = t * 255.0f + 0.0f; }
    } else {
        for (size_t i = 0; i < n; i++) out[i] = std::min(std::max(img[i], 0.0f), 255.0f);
    }
}

// ---------------------------------------------------------------------------------------------
// a12
// ---------------------------------------------------------------------------------------------
void median13(const uint8_t *m, int H, int W, uint8_t *out)
{
    static const int dy[13] = {-2, -1, -1, -1, 0, 0, 0, 0, 0, 1, 1, 1, 2};
    static const int dx[13] = {0, -1, 0, 1, -2, -1, 0, 1, 2, -1, 0, 1, 0};
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            int c = 0;
            for (int k = 0; k < 13; k++) {
                const int yy = std::min(std::max(y + dy[k], 0), H - 1), xx = std::min(std::max(x + dx[k], 0), W - 1);
                c += m[(size_t)yy * W + xx] != 0;
            }
            out[(size_t)y * W + x] = c >= 7;
        }
}

// 8-connected labelling, labels 1..n in raster order of each component's first pixel
int label8(const uint8_t *m, int H, int W, std::vector<int32_t> &lab)
{
    lab.assign((size_t)H * W, 0);
    std::vector<int32_t> stack;
    int n = 0;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const size_t p = (size_t)y * W + x;
            if (!m[p] || lab[p]) continue;
            n++;
            lab[p] = n;
            stack.clear(); stack.push_back((int32_t)p);
            while (!stack.empty()) {
                const int32_t q = stack.back(); stack.pop_back();
                const int qy = q / W, qx = q % W;
                for (int a = -1; a <= 1; a++)
                    for (int b = -1; b <= 1; b++) {
                        const int yy = qy + a, xx = qx + b;
                        if ((a | b) == 0 || yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                        const size_t r = (size_t)yy * W + xx;
                        if (m[r] && !lab[r]) { lab[r] = n; stack.push_back((int32_t)r); }
                    }
            }
        }
    return n;
}

// skimage's 2-D skeletonize table (recovered from scikit-image's compiled _fast_skeletonize by
// exhaustive small-image probing, tools/recover_skel_lut.py): neighbour weights NW=1, N=2, NE=4,
// E=8, SE=16, S=32, SW=64, W=128; 1: removable in the first sub-iteration, 2: second, 3: both.
#include "skel_lut.inc"

void skeletonize_zhang(const uint8_t *m, int H, int W, uint8_t *out)
{
    const int Wp = W + 2;
    std::vector<uint8_t> sk((size_t)(H + 2) * Wp, 0);
    std::vector<int32_t> fg, kill, nxt;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            if (m[(size_t)y * W + x]) { sk[(size_t)(y + 1) * Wp + x + 1] = 1; fg.push_back((y + 1) * Wp + x + 1); }
    bool removed = true;
    while (removed) {
        removed = false;
        for (int pass = 0; pass < 2; pass++) {
            kill.clear(); nxt.clear();
            for (int32_t p : fg) {
                const uint8_t *s = &sk[p];
                const int code = s[-Wp - 1] + 2 * s[-Wp] + 4 * s[-Wp + 1] + 8 * s[1] + 16 * s[Wp + 1] + 32 * s[Wp] +
                                 64 * s[Wp - 1] + 128 * s[-1];
                const int v = SKEL_LUT[code];
                if (v == 3 || (v == 1 && pass == 0) || (v == 2 && pass == 1)) kill.push_back(p);
                else nxt.push_back(p);
            }
            if (!kill.empty()) {
                removed = true;
                for (int32_t p : kill) sk[p] = 0;
                fg.swap(nxt);
            }
        }
    }
    std::memset(out, 0, (size_t)H * W);
    for (int32_t p : fg) out[(size_t)(p / Wp - 1) * W + (p % Wp - 1)] = 1;
}

// transforms.py:306-361
void filter_mask(const uint8_t *mask_in, int H, int W, bool use_median, bool remove_isolated, uint8_t *out)
{
    const size_t n = (size_t)H * W;
    std::vector<uint8_t> m(n);
    if (use_median) median13(mask_in, H, W, m.data());
    else for (size_t i = 0; i < n; i++) m[i] = mask_in[i] != 0;
    std::vector<int32_t> lab;
    const int nl = label8(m.data(), H, W, lab);
    // area + perimeter codes per label
    std::vector<int64_t> area(nl + 1, 0), n1(nl + 1, 0), n2(nl + 1, 0), n3(nl + 1, 0);
    auto L = [&](int y, int x) -> int32_t { return (y < 0 || y >= H || x < 0 || x >= W) ? 0 : lab[(size_t)y * W + x]; };
    auto border = [&](int y, int x, int32_t l) -> bool {
        if (L(y, x) != l) return false;
        return L(y - 1, x) != l || L(y + 1, x) != l || L(y, x - 1) != l || L(y, x + 1) != l;
    };
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const int32_t l = lab[(size_t)y * W + x];
            if (!l) continue;
            area[l]++;
            if (!border(y, x, l)) continue;
            const int code = 1 + 2 * (border(y - 1, x, l) + border(y + 1, x, l) + border(y, x - 1, l) + border(y, x + 1, l)) +
                             10 * (border(y - 1, x - 1, l) + border(y - 1, x + 1, l) + border(y + 1, x - 1, l) + border(y + 1, x + 1, l));
            if (code == 5 || code == 7 || code == 15 || code == 17 || code == 25 || code == 27) n1[l]++;
            else if (code == 21 || code == 33) n2[l]++;
            else if (code == 13 || code == 23) n3[l]++;
        }
    const double SQ2 = std::sqrt(2.0);
    std::vector<double> circ(nl + 1, 0.0);
    for (int l = 1; l <= nl; l++) {
        const double per = (double)n1[l] + (double)n2[l] * SQ2 + (double)n3[l] * ((1 + SQ2) / 2);
        circ[l] = 4 * M_PI * (double)area[l] / (per * per + 1e-7);
    }
    // skeleton, its 8-connected components, fork test
    std::vector<uint8_t> sk(n);
    skeletonize_zhang(m.data(), H, W, sk.data());
    std::vector<int32_t> slab;
    const int ns = label8(sk.data(), H, W, slab);
    std::vector<uint8_t> has_fork(ns + 1, 0);
    std::vector<int32_t> first_lab(ns + 1, 0);
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const size_t p = (size_t)y * W + x;
            const int32_t c = slab[p];
            if (!c) continue;
            if (!first_lab[c]) first_lab[c] = lab[p];
            int deg = 0;
            for (int a = -1; a <= 1; a++)
                for (int b = -1; b <= 1; b++) {
                    const int yy = y + a, xx = x + b;
                    if ((a | b) == 0 || yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                    deg += sk[(size_t)yy * W + xx];
                }
            if (deg > 2) has_fork[c] = 1;
        }
    std::vector<uint8_t> drop(nl + 1, 0);
    for (int c = 1; c <= ns; c++) {
        const int32_t l = first_lab[c];
        if ((remove_isolated && !has_fork[c]) || circ[l] > 0.8) drop[l] = 1;
    }
    for (size_t i = 0; i < n; i++) out[i] = m[i] && !drop[lab[i]];
}

// ---------------------------------------------------------------------------------------------
// exact Euclidean distance transform (distance to the nearest zero pixel), Meijster's integer
// two-phase algorithm; sqrt of the exact squared distance == scipy.ndimage.distance_transform_edt
// ---------------------------------------------------------------------------------------------
void edt(const uint8_t *m, int H, int W, double *dist)
{
    const int64_t INF = 1 << 20;
    bool any_zero = false;
    for (size_t i = 0, n = (size_t)H * W; i < n; i++) if (!m[i]) { any_zero = true; break; }
    if (!any_zero) {   // scipy quirk for an image without background: distance to the virtual pixel (-1, 0)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) dist[(size_t)y * W + x] = std::sqrt((double)((int64_t)(y + 1) * (y + 1) + (int64_t)x * x));
        return;
    }
    std::vector<int64_t> g((size_t)H * W);
    for (int x = 0; x < W; x++) {
        g[x] = m[x] ? INF : 0;
        for (int y = 1; y < H; y++) g[(size_t)y * W + x] = m[(size_t)y * W + x] ? std::min(INF, g[(size_t)(y - 1) * W + x] + 1) : 0;
        for (int y = H - 2; y >= 0; y--)
            if (g[(size_t)(y + 1) * W + x] < g[(size_t)y * W + x]) g[(size_t)y * W + x] = std::min(g[(size_t)y * W + x], g[(size_t)(y + 1) * W + x] + 1);
    }
    std::vector<int> s(W), t(W);
    for (int y = 0; y < H; y++) {
        const int64_t *gr = &g[(size_t)y * W];
        auto f = [&](int x, int i) -> int64_t { return (int64_t)(x - i) * (x - i) + gr[i] * gr[i]; };
        auto sep = [&](int i, int u) -> int64_t {
            return ((int64_t)u * u - (int64_t)i * i + gr[u] * gr[u] - gr[i] * gr[i]) / (2 * (int64_t)(u - i));
        };
        int q = 0;
        s[0] = 0; t[0] = 0;
        for (int u = 1; u < W; u++) {
            while (q >= 0 && f(t[q], s[q]) > f(t[q], u)) q--;
            if (q < 0) { q = 0; s[0] = u; }
            else {
                const int64_t w = 1 + sep(s[q], u);
                if (w < W) { q++; s[q] = u; t[q] = (int)w; }
            }
        }
        for (int u = W - 1; u >= 0; u--) {
            dist[(size_t)y * W + u] = std::sqrt((double)f(u, s[q]));
            if (u == t[q]) q--;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// numpy RandomState(seed).permutation(arange(n)): MT19937 + legacy masked-rejection shuffle
// ---------------------------------------------------------------------------------------------
struct MT19937 {
    uint32_t key[624];
    int pos;
    explicit MT19937(uint32_t seed)
    {
        for (int i = 0; i < 624; i++) { key[i] = seed; seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)i + 1; }
        pos = 624;
    }
    uint32_t next()
    {
        if (pos == 624) {
            int i;
            uint32_t y;
            for (i = 0; i < 624 - 397; i++) { y = (key[i] & 0x80000000u) | (key[i + 1] & 0x7fffffffu); key[i] = key[i + 397] ^ (y >> 1) ^ ((y & 1) ? 0x9908b0dfu : 0); }
            for (; i < 623; i++) { y = (key[i] & 0x80000000u) | (key[i + 1] & 0x7fffffffu); key[i] = key[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1) ? 0x9908b0dfu : 0); }
            y = (key[623] & 0x80000000u) | (key[0] & 0x7fffffffu);
            key[623] = key[396] ^ (y >> 1) ^ ((y & 1) ? 0x9908b0dfu : 0);
            pos = 0;
        }
        uint32_t y = key[pos++];
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        return y;
    }
};

void legacy_permutation(uint32_t seed, size_t n, std::vector<uint32_t> &perm)
{
    perm.resize(n);
    for (size_t i = 0; i < n; i++) perm[i] = (uint32_t)i;
    MT19937 rng(seed);
    for (size_t i = n; i-- > 1;) {
        uint64_t mask = i;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
        uint64_t v;
        while ((v = (rng.next() & mask)) > i) {}
        std::swap(perm[i], perm[v]);
    }
}

// a14: skimage.morphology.medial_axis(mask, return_distance=True), 0.18.3 tie-break
static const uint8_t *medial_table()
{
    static uint8_t table[512];
    static bool init = false;
    if (!init) {
        for (int idx = 0; idx < 512; idx++) {
            auto ncc = [](int pat) {      // number of 8-connected components of the 3x3 pattern
                int lab[9] = {0}, n = 0;
                for (int s = 0; s < 9; s++) {
                    if (!((pat >> s) & 1) || lab[s]) continue;
                    int stack[9], sp = 0;
                    stack[sp++] = s; lab[s] = ++n;
                    while (sp) {
                        int q = stack[--sp], qy = q / 3, qx = q % 3;
                        for (int a = -1; a <= 1; a++)
                            for (int b = -1; b <= 1; b++) {
                                int yy = qy + a, xx = qx + b;
                                if (yy < 0 || yy > 2 || xx < 0 || xx > 2) continue;
                                int r = yy * 3 + xx;
                                if (((pat >> r) & 1) && !lab[r]) { lab[r] = n; stack[sp++] = r; }
                            }
                    }
                }
                return n;
            };
            const bool center = idx & 16;
            const bool c2 = ncc(idx) != ncc(idx & ~16);
            const bool c3 = __builtin_popcount(idx) < 3;
            table[idx] = center && (c2 || c3);
        }
        init = true;
    }
    return table;
}

void medial_table_bits(uint32_t out[16])
{
    const uint8_t *t = medial_table();
    for (int i = 0; i < 16; i++) out[i] = 0;
    for (int i = 0; i < 512; i++) if (t[i]) out[i >> 5] |= 1u << (i & 31);
}

void medial_axis(const uint8_t *m, int H, int W, uint8_t *skel, double *dist)
{
    edt(m, H, W, dist);
    medial_axis_thin(m, dist, H, W, skel);
}

// the ordered thinning of medial_axis given the mask's EDT (sequential by construction: every decision reads the
// current state of the 3x3 neighbourhood)
void medial_axis_thin(const uint8_t *m, const double *dist, int H, int W, uint8_t *skel)
{
    const uint8_t *table = medial_table();
    const int Wp = W + 2;
    std::vector<uint8_t> res((size_t)(H + 2) * Wp, 0);
    std::vector<int32_t> pos;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++)
            if (m[(size_t)y * W + x]) { res[(size_t)(y + 1) * Wp + x + 1] = 1; pos.push_back((y + 1) * Wp + x + 1); }
    const size_t nfg = pos.size();
    std::vector<uint32_t> tie;
    legacy_permutation(0, nfg, tie);
    // sort key: (distance^2 as exact integer, corner score, tiebreak) ascending
    struct Key { uint64_t k; int32_t p; };
    std::vector<Key> keys(nfg);
    for (size_t i = 0; i < nfg; i++) {
        const int32_t p = pos[i];
        const uint8_t *s = &res[p];
        const int cnt = s[-Wp - 1] + s[-Wp] + s[-Wp + 1] + s[-1] + 1 + s[1] + s[Wp - 1] + s[Wp] + s[Wp + 1];
        const int y = p / Wp - 1, x = p % Wp - 1;
        const double d = dist[(size_t)y * W + x];
        const uint64_t d2 = (uint64_t)std::llround(d * d);
        keys[i].k = (d2 << 40) | ((uint64_t)(9 - cnt) << 32) | tie[i];
        keys[i].p = p;
    }
    std::sort(keys.begin(), keys.end(), [](const Key &a, const Key &b) { return a.k < b.k; });
    for (const Key &kk : keys) {
        uint8_t *s = &res[kk.p];
        const int acc = 16 + s[-Wp - 1] + 2 * s[-Wp] + 4 * s[-Wp + 1] + 8 * s[-1] + 32 * s[1] + 64 * s[Wp - 1] + 128 * s[Wp] + 256 * s[Wp + 1];
        *s = table[acc];
    }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) skel[(size_t)y * W + x] = res[(size_t)(y + 1) * Wp + x + 1];
}

// ---------------------------------------------------------------------------------------------
// a16: skimage resize(order=1, anti_aliasing=True, preserve_range=True), scikit-image >= 0.19
// ---------------------------------------------------------------------------------------------
static inline int mirror_idx(long i, int n)
{
    if (n == 1) return 0;
    const long p = 2L * (n - 1);
    i %= p; if (i < 0) i += p;
    return (int)(i < n ? i : p - i);
}

static void gauss_mirror_axis(const std::vector<double> &a, int H, int W, int axis, double sigma, std::vector<double> &out)
{
    const int r = (int)(4.0 * sigma + 0.5);
    std::vector<double> w(2 * r + 1);
    const double s2 = sigma * sigma;
    double tot = 0.0;
    for (int x = -r; x <= r; x++) { w[x + r] = std::exp(-0.5 / s2 * (double)(x * x)); }
    for (int k = 0; k < 2 * r + 1; k++) tot += w[k];
    for (int k = 0; k < 2 * r + 1; k++) w[k] = w[k] / tot;
    out.resize(a.size());
    const int n = axis == 0 ? H : W;
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const int l = axis == 0 ? y : x;
            auto at = [&](int i) -> double { return axis == 0 ? a[(size_t)i * W + x] : a[(size_t)y * W + i]; };
            double t = at(l) * w[r];
            for (int j = -r; j < 0; j++) t = t + (at(mirror_idx((long)l + j, n)) + at(mirror_idx((long)l - j, n))) * w[r + j];
            out[(size_t)y * W + x] = t;
        }
}

void resize_aa(const double *img, int H, int W, int oh, int ow, float *out)
{
    const size_t n = (size_t)H * W;
    std::vector<double> cur(img, img + n), tmp;
    double lo = std::numeric_limits<double>::infinity(), hi = -lo;
    for (size_t i = 0; i < n; i++) { lo = std::min(lo, img[i]); hi = std::max(hi, img[i]); }
    const double f0 = (double)H / (double)oh, f1 = (double)W / (double)ow;
    const double sg0 = std::max(0.0, (f0 - 1) / 2), sg1 = std::max(0.0, (f1 - 1) / 2);
    if (sg0 > 0) { gauss_mirror_axis(cur, H, W, 0, sg0, tmp); cur.swap(tmp); }
    if (sg1 > 0) { gauss_mirror_axis(cur, H, W, 1, sg1, tmp); cur.swap(tmp); }
    auto axis_tab = [](int n_in, int n_out, std::vector<int> &i0, std::vector<int> &i1, std::vector<double> &w0, std::vector<double> &w1) {
        const double zoom = (double)n_in / (double)n_out;
        i0.resize(n_out); i1.resize(n_out); w0.resize(n_out); w1.resize(n_out);
        for (int j = 0; j < n_out; j++) {
            const double cc = ((double)j + 0.5) * zoom - 0.5;
            const double fl = std::floor(cc), t = cc - fl;
            w0[j] = 1.0 - t; w1[j] = 1.0 - w0[j];
            i0[j] = mirror_idx((long)fl, n_in); i1[j] = mirror_idx((long)fl + 1, n_in);
        }
    };
    std::vector<int> r0, r1, c0, c1;
    std::vector<double> wr0, wr1, wc0, wc1;
    axis_tab(H, oh, r0, r1, wr0, wr1);
    axis_tab(W, ow, c0, c1, wc0, wc1);
    for (int y = 0; y < oh; y++)
        for (int x = 0; x < ow; x++) {
            double t = (cur[(size_t)r0[y] * W + c0[x]] * wr0[y]) * wc0[x];
            t = t + (cur[(size_t)r0[y] * W + c1[x]] * wr0[y]) * wc1[x];
            t = t + (cur[(size_t)r1[y] * W + c0[x]] * wr1[y]) * wc0[x];
            t = t + (cur[(size_t)r1[y] * W + c1[x]] * wr1[y]) * wc1[x];
            t = std::min(std::max(t, lo), hi);
            out[(size_t)y * ow + x] = (float)t;
        }
}

// compute_branches.py:340-357 for one image, given the filtered mask and its EDT from the GPU (morph_kernels.hip)
void postprocess_from_filtered(const double *pred, const uint8_t *filt, const double *dist, int H, int W, int oh, int ow, float *field)
{
    const size_t n = (size_t)H * W;
    std::vector<uint8_t> skel(n), nskel(n);
    std::vector<double> cdt(n), wt(n);
    medial_axis_thin(filt, dist, H, W, skel.data());
    for (size_t i = 0; i < n; i++) nskel[i] = !skel[i];
    edt(nskel.data(), H, W, cdt.data());
    for (size_t i = 0; i < n; i++) wt[i] = pred[i] * (dist[i] / (dist[i] + cdt[i]));
    resize_aa(wt.data(), H, W, oh, ow, field);
}

// compute_branches.py:334-357 for one image (no well mask), host only
void postprocess_image(const double *pred, int H, int W, int oh, int ow, float *field)
{
    const size_t n = (size_t)H * W;
    std::vector<uint8_t> seg(n), filt(n), skel(n), nskel(n);
    for (size_t i = 0; i < n; i++) seg[i] = pred[i] > 0.5;
    filter_mask(seg.data(), H, W, true, true, filt.data());
    std::vector<double> dist(n), cdt(n), wt(n);
    medial_axis(filt.data(), H, W, skel.data(), dist.data());
    for (size_t i = 0; i < n; i++) nskel[i] = !skel[i];
    edt(nskel.data(), H, W, cdt.data());
    for (size_t i = 0; i < n; i++) wt[i] = pred[i] * (dist[i] / (dist[i] + cdt[i]));
    resize_aa(wt.data(), H, W, oh, ow, field);
}

}  // namespace tmat

using namespace tmat;

// Host-stage entry points (no handle): used by the pipeline driver and exposed for stage-wise parity tests.
extern "C" {
int tmat_host_lanczos4_u16(const uint16_t *img, int H, int W, int h, int w, uint16_t *out)
{
    if (!img || !out || H < 1 || W < 1 || h < 1 || w < 1) { set_error("tmat_host_lanczos4_u16: bad argument"); return TMAT_E_ARG; }
    lanczos4_resize_u16(img, H, W, h, w, out);
    return TMAT_OK;
}
int tmat_host_rescale01_u16(const uint16_t *img, size_t n, float *out) { rescale01_u16(img, n, out); return TMAT_OK; }
int tmat_host_rescale255_f32(const float *img, size_t n, float *out) { rescale255_f32(img, n, out); return TMAT_OK; }
int tmat_host_filter_mask(const uint8_t *mask, int H, int W, int use_median, int remove_isolated, uint8_t *out)
{
    if (!mask || !out || H < 1 || W < 1) { set_error("tmat_host_filter_mask: bad argument"); return TMAT_E_ARG; }
    filter_mask(mask, H, W, use_median != 0, remove_isolated != 0, out);
    return TMAT_OK;
}
int tmat_host_skeletonize(const uint8_t *mask, int H, int W, uint8_t *out) { skeletonize_zhang(mask, H, W, out); return TMAT_OK; }
int tmat_host_medial_axis(const uint8_t *mask, int H, int W, uint8_t *skel, double *dist) { medial_axis(mask, H, W, skel, dist); return TMAT_OK; }
int tmat_host_permutation(uint32_t seed, int n, uint32_t *out)
{
    std::vector<uint32_t> p;
    legacy_permutation(seed, (size_t)n, p);
    std::memcpy(out, p.data(), sizeof(uint32_t) * (size_t)n);
    return TMAT_OK;
}
int tmat_host_postprocess(const double *pred, int H, int W, int oh, int ow, float *field)
{
    if (!pred || !field || H < 1 || W < 1 || oh < 1 || ow < 1) { set_error("tmat_host_postprocess: bad argument"); return TMAT_E_ARG; }
    postprocess_image(pred, H, W, oh, ow, field);
    return TMAT_OK;
}
}
