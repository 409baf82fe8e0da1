// Binary morphology of the segmentation mask on gfx950 (batch of images, one launch per step):
//   a11  seg = pred > 0.5                                              compute_branches.py:334
//   a12  filter_branch_seg_mask (transforms.py:306-361):
//        13-tap binary median (disk(2), 'nearest') -> 8-connected labelling (union-find, root = smallest pixel
//        index) -> area / 4-neighbourhood perimeter codes per label -> circularity -> Zhang thinning with
//        scikit-image's table (two parallel sub-iterations to convergence) -> skeleton components + fork test ->
//        drop components without a fork or with circularity > 0.8
//   a14a exact EDT of the filtered mask (Meijster, integer squared distances -> sqrt in f64), the `distance`
//        of medial_axis (compute_branches.py:340); the ordered thinning itself is sequential and stays on the host.
// All of it is integer / byte work: HBM-bound streaming kernels plus atomics on per-label counters.
// Results are identical to csrc/postproc.cpp (host twin used by the stage-wise C-ABI entry points) and to
// oracle/morph.py: the label ids differ (root pixel index instead of raster rank) but no output depends on them.
#include "tmat_internal.h"
#include "dev_guard.h"
#include "morph.h"

namespace tmat {

#include "skel_lut.inc"
__constant__ unsigned char d_skel_lut[256];

static bool g_lut_uploaded = false;
static bool upload_lut()
{
    if (g_lut_uploaded) return true;
    if (hipMemcpyToSymbol(HIP_SYMBOL(d_skel_lut), SKEL_LUT, 256) != hipSuccess) return false;
    g_lut_uploaded = true;
    return true;
}

#define IMG_LOOP(p, n_px) for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < (n_px); p += gridDim.x * blockDim.x)

__global__ void binarize_kernel(const uint8_t *__restrict__ m, uint8_t *__restrict__ seg, int npx)
{
    const size_t base = (size_t)blockIdx.y * npx;
    IMG_LOOP(p, npx) seg[base + p] = m[base + p] != 0;
}

__global__ void threshold_kernel(const double *__restrict__ pred, uint8_t *__restrict__ seg, int npx)
{
    const size_t base = (size_t)blockIdx.y * npx;
    IMG_LOOP(p, npx) seg[base + p] = pred[base + p] > 0.5;
}

__global__ void median13_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int H, int W)
{
    const size_t base = (size_t)blockIdx.y * H * W;
    IMG_LOOP(p, H * W) {
        const int y = p / W, x = p - y * W;
        int c = 0;
#pragma unroll
        for (int dy = -2; dy <= 2; dy++) {
            const int r = 2 - (dy < 0 ? -dy : dy);
            const int yy = min(max(y + dy, 0), H - 1);
            for (int dx = -r; dx <= r; dx++) c += in[base + (size_t)yy * W + min(max(x + dx, 0), W - 1)];
        }
        out[base + p] = c >= 7;
    }
}

// ---- 8-connected labelling: union-find with atomicMin, root = smallest pixel index of the component ----
__device__ __forceinline__ int uf_find(const int *L, int x)
{
    for (;;) {
        const int p = __hip_atomic_load(&L[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p == x) return x;
        x = p;
    }
}
__device__ __forceinline__ void uf_union(int *L, int a, int b)
{
    for (;;) {
        a = uf_find(L, a); b = uf_find(L, b);
        if (a == b) return;
        if (a < b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(&L[a], b);      // a > b: hang root a under b
        if (old == a) return;
        a = old;
    }
}
// Initial label = the first pixel of the pixel's horizontal run inside its 64-pixel chunk (a wave covers 64 consecutive
// flattened indices, so the run start comes from one ballot): runs are already merged, and every label is <= its index.
__global__ void ccl_init_kernel(const uint8_t *__restrict__ m, int *__restrict__ L, int H, int W)
{
    const size_t base = (size_t)blockIdx.y * H * W;
    IMG_LOOP(p, H * W) {
        const bool set = m[base + p] != 0;
        const unsigned long long on = __ballot(set);
        const int lane = p & 63;                               // blockDim and the grid stride are multiples of 64
        const int x = p % W;
        const int first = lane - (lane < x ? lane : x);        // first lane of this chunk that lies in the same row
        // zeros among the lanes [first, lane): the run starts just after the highest of them
        const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
        const unsigned long long span = below & (first ? ~(~0ull >> (64 - first)) : ~0ull);
        const unsigned long long zeros = ~on & span;
        const int start = zeros ? 64 - __clzll(zeros) : first;
        L[base + p] = set ? p - (lane - start) : -1;
    }
}
// Unions that the run initialisation and the neighbours' own unions do not already imply:
//   west, only across a chunk boundary;  north unless west and north-west are set (then W-NW-N already connects);
//   without north: north-west unless west is set (W's north is NW), and north-east.
__global__ void ccl_merge_kernel(const uint8_t *__restrict__ m, int *__restrict__ L, int H, int W)
{
    const size_t base = (size_t)blockIdx.y * H * W;
    const uint8_t *mm = m + base;
    int *LL = L + base;
    IMG_LOOP(p, H * W) {
        if (!mm[p]) continue;
        const int y = p / W, x = p - y * W;
        const bool w_ = x > 0 && mm[p - 1];
        if (w_ && (p & 63) == 0) uf_union(LL, p, p - 1);
        if (y > 0) {
            const bool n_ = mm[p - W], nw = x > 0 && mm[p - W - 1], ne = x < W - 1 && mm[p - W + 1];
            if (n_) { if (!(w_ && nw)) uf_union(LL, p, p - W); }
            else {
                if (nw && !w_) uf_union(LL, p, p - W - 1);
                if (ne) uf_union(LL, p, p - W + 1);
            }
        }
    }
}
__global__ void ccl_compress_kernel(int *__restrict__ L, int npx)
{
    const size_t base = (size_t)blockIdx.y * npx;
    IMG_LOOP(p, npx) { if (L[base + p] >= 0) L[base + p] = uf_find(L + base, p); }
}

// 8-connected labels of k masks: L = smallest pixel index of the component (per image), -1 on the background
void launch_ccl(const uint8_t *mask, int k, int H, int W, int *L, hipStream_t s)
{
    const int npx = H * W;
    const dim3 grid((npx + 255) / 256 < 1024 ? (npx + 255) / 256 : 1024, k), blk(256);
    hipLaunchKernelGGL(ccl_init_kernel, grid, blk, 0, s, mask, L, H, W);
    hipLaunchKernelGGL(ccl_merge_kernel, grid, blk, 0, s, mask, L, H, W);
    hipLaunchKernelGGL(ccl_compress_kernel, grid, blk, 0, s, L, npx);
}

// ---- per-label area and perimeter code counts (skimage regionprops.perimeter, 4-neighbourhood) ----
__device__ __forceinline__ int lab_at(const int *L, int H, int W, int y, int x) { return (y < 0 || y >= H || x < 0 || x >= W) ? -1 : L[y * W + x]; }
__device__ __forceinline__ bool is_border(const int *L, int H, int W, int y, int x, int l)
{
    if (lab_at(L, H, W, y, x) != l) return false;
    return lab_at(L, H, W, y - 1, x) != l || lab_at(L, H, W, y + 1, x) != l || lab_at(L, H, W, y, x - 1) != l || lab_at(L, H, W, y, x + 1) != l;
}
__global__ void region_stats_kernel(const int *__restrict__ L, int H, int W, int *__restrict__ area, int *__restrict__ n1,
                                    int *__restrict__ n2, int *__restrict__ n3)
{
    const size_t base = (size_t)blockIdx.y * H * W;
    const int *LL = L + base;
    IMG_LOOP(p, H * W) {
        const int l = LL[p];
        // area: the lanes that share the first active lane's label add once (blob interiors: one atomic per wave)
        const unsigned long long act = __ballot(l >= 0);
        if (!act) continue;
        const int lead = __ffsll((long long)act) - 1;
        const int l0 = __shfl(l, lead);
        const unsigned long long same = __ballot(l == l0);
        if (l < 0) continue;
        if (l == l0) { if ((p & 63) == lead) atomicAdd(&area[base + l0], __popcll(same)); }
        else atomicAdd(&area[base + l], 1);
        const int y = p / W, x = p - y * W;
        if (!is_border(LL, H, W, y, x, l)) continue;
        const int code = 1 + 2 * (is_border(LL, H, W, y - 1, x, l) + is_border(LL, H, W, y + 1, x, l) + is_border(LL, H, W, y, x - 1, l) + is_border(LL, H, W, y, x + 1, l)) +
                         10 * (is_border(LL, H, W, y - 1, x - 1, l) + is_border(LL, H, W, y - 1, x + 1, l) + is_border(LL, H, W, y + 1, x - 1, l) + is_border(LL, H, W, y + 1, x + 1, l));
        if (code == 5 || code == 7 || code == 15 || code == 17 || code == 25 || code == 27) atomicAdd(&n1[base + l], 1);
        else if (code == 21 || code == 33) atomicAdd(&n2[base + l], 1);
        else if (code == 13 || code == 23) atomicAdd(&n3[base + l], 1);
    }
}

// ---- Zhang thinning, ZH_IT full iterations (2 ZH_IT parallel sub-iterations) per launch ----
// A sub-iteration moves information by one pixel, so a block that stages its ZH_TI x ZH_TI tile with a halo of
// ZH_R = 2 ZH_IT pixels in LDS can run 2 ZH_IT sub-iterations locally and still hold the exact state of its inner
// tile (the ring that would need pixels outside the staged tile shrinks inward by one pixel per sub-iteration and never
// reaches the inner tile).  changed[launch][img] is raised when an inner-tile pixel is removed; a launch that removes
// nothing leaves out == in, so every later launch of that image can return immediately (both buffers hold the result).
constexpr int ZH_IT = 4, ZH_R = 2 * ZH_IT, ZH_TI = 48, ZH_T = ZH_TI + 2 * ZH_R;     // 64 x 64 staged pixels
//
// Tiles (round 4): tflags[launch % 3][img][tile] is raised with it for the TILE.  A tile whose 3 x 3 tile neighbourhood removed nothing
// in the previous launch has nothing to do in this one: a 144 x 144 region around it stood still through 4 full iterations (so its
// last iteration was a fixed point there), and what changed further out is >= 40 pixels from the tile's staged area while a launch
// moves information 8 pixels.  The tile itself did not change in that launch either, so both buffers already hold its pixels and
// skipping the write is exact.  It wakes up again as soon as a neighbour removes something.  (The slot of the next launch is
// cleared here: nobody reads it during this launch.)
__global__ __launch_bounds__(256) void zhang_tile_kernel(const uint8_t *__restrict__ in, uint8_t *__restrict__ out, int H, int W,
                                                         int tiles_x, const int *__restrict__ prev_changed, int *__restrict__ changed,
                                                         int *__restrict__ tflags, int it)
{
    const int img = blockIdx.y, ntiles = gridDim.x, k = gridDim.y;
    int *tf_w = tflags + ((size_t)(it % 3) * k + img) * ntiles;
    const int *tf_r = tflags + ((size_t)((it + 2) % 3) * k + img) * ntiles;
    if (threadIdx.x == 0) tflags[((size_t)((it + 1) % 3) * k + img) * ntiles + blockIdx.x] = 0;
    if (prev_changed && !prev_changed[img]) return;          // converged in an earlier launch
    if (it > 0) {
        const int tiles_y = ntiles / tiles_x, ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
#ifdef ZH_VAR_NOSKIP
        int act = 1;
#else
        int act = 0;
#endif
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                const int yy = ty + dy, xx = tx + dx;
                if (yy >= 0 && yy < tiles_y && xx >= 0 && xx < tiles_x) act |= tf_r[yy * tiles_x + xx];
            }
        if (!act) return;
    }
    __shared__ uint8_t buf[2][ZH_T * ZH_T];
    const size_t base = (size_t)img * H * W;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y0 = ty * ZH_TI - ZH_R, x0 = tx * ZH_TI - ZH_R;
    const int t = threadIdx.x;
    for (int idx = t; idx < ZH_T * ZH_T; idx += 256) {
        const int y = y0 + idx / ZH_T, x = x0 + idx % ZH_T;
        buf[0][idx] = (y >= 0 && y < H && x >= 0 && x < W) ? in[base + (size_t)y * W + x] : 0;
    }
    __syncthreads();
    bool any = false;
    int cur = 0;
#pragma unroll 1
    for (int sub = 0; sub < 2 * ZH_IT; sub++) {
        const uint8_t *s = buf[cur];
        uint8_t *d = buf[cur ^ 1];
        const int pass = sub & 1;
        for (int idx = t; idx < ZH_T * ZH_T; idx += 256) {
            const int ly = idx / ZH_T, lx = idx % ZH_T;
            uint8_t v = s[idx];
            if (v && ly > 0 && ly < ZH_T - 1 && lx > 0 && lx < ZH_T - 1) {
                const uint8_t *q = s + idx;
                const int code = q[-ZH_T - 1] + 2 * q[-ZH_T] + 4 * q[-ZH_T + 1] + 8 * q[1] + 16 * q[ZH_T + 1] + 32 * q[ZH_T] +
                                 64 * q[ZH_T - 1] + 128 * q[-1];
                const int tt = d_skel_lut[code];
                if (tt == 3 || (tt == 1 && pass == 0) || (tt == 2 && pass == 1)) {
                    v = 0;
                    if (ly >= ZH_R && ly < ZH_R + ZH_TI && lx >= ZH_R && lx < ZH_R + ZH_TI) any = true;
                }
            }
            d[idx] = v;
        }
        cur ^= 1;
        __syncthreads();
    }
    for (int idx = t; idx < ZH_TI * ZH_TI; idx += 256) {
        const int ly = idx / ZH_TI, lx = idx % ZH_TI;
        const int y = y0 + ZH_R + ly, x = x0 + ZH_R + lx;
        if (y < H && x < W) out[base + (size_t)y * W + x] = buf[cur][(ly + ZH_R) * ZH_T + lx + ZH_R];
    }
    if (__any(any) && (t & 63) == 0) { atomicOr(&changed[img], 1); tf_w[blockIdx.x] = 1; }
}

// converged <=> the last launch that could run removed nothing (its flag stayed 0)
__global__ void zhang_done_kernel(const int *__restrict__ last_changed, int *__restrict__ done, int k)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < k) done[i] = !last_changed[i];
}

// ---- skeleton components: fork flag and decision per skeleton root -> drop flag per mask root ----
__global__ void skel_fork_kernel(const uint8_t *__restrict__ sk, const int *__restrict__ SL, int H, int W, int *__restrict__ has_fork)
{
    const size_t base = (size_t)blockIdx.y * H * W;
    const uint8_t *s = sk + base;
    IMG_LOOP(p, H * W) {
        if (!s[p]) continue;
        const int y = p / W, x = p - y * W;
        int deg = 0;
        for (int a = -1; a <= 1; a++)
            for (int b = -1; b <= 1; b++) {
                const int yy = y + a, xx = x + b;
                if ((a | b) == 0 || yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
                deg += s[yy * W + xx];
            }
        if (deg > 2) atomicOr(&has_fork[base + SL[base + p]], 1);
    }
}
__global__ void decide_kernel(const int *__restrict__ SL, const int *__restrict__ ML, const int *__restrict__ has_fork,
                              const int *__restrict__ area, const int *__restrict__ n1, const int *__restrict__ n2,
                              const int *__restrict__ n3, int npx, int remove_isolated, int *__restrict__ drop)
{
    const size_t base = (size_t)blockIdx.y * npx;
    const double SQ2 = sqrt(2.0);
    IMG_LOOP(p, npx) {
        if (SL[base + p] != p) continue;                 // one decision per skeleton component (its root pixel)
        const int l = ML[base + p];                      // mask component of that pixel
        const double per = (double)n1[base + l] + (double)n2[base + l] * SQ2 + (double)n3[base + l] * ((1 + SQ2) / 2);
        const double circ = 4 * M_PI * (double)area[base + l] / (per * per + 1e-7);
        if ((remove_isolated && !has_fork[base + p]) || circ > 0.8) drop[base + l] = 1;
    }
}
__global__ void apply_drop_kernel(const uint8_t *__restrict__ m, const int *__restrict__ ML, const int *__restrict__ drop,
                                  uint8_t *__restrict__ out, int npx)
{
    const size_t base = (size_t)blockIdx.y * npx;
    IMG_LOOP(p, npx) out[base + p] = m[base + p] && !drop[base + ML[base + p]];
}

// ---- exact EDT (Meijster): columns, then rows; squared distances are exact integers ----
__global__ void edt_cols_kernel(const uint8_t *__restrict__ m, int *__restrict__ g, int H, int W, int *__restrict__ any_zero)
{
    const int img = blockIdx.y;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= W) return;
    const uint8_t *mm = m + (size_t)img * H * W;
    int *gg = g + (size_t)img * H * W;
    const int INF = 1 << 20;
    bool zero = false;
    int prev = mm[x] ? INF : 0;
    zero |= !mm[x];
    gg[x] = prev;
    for (int y = 1; y < H; y++) {
        const bool fg = mm[(size_t)y * W + x];
        zero |= !fg;
        prev = fg ? min(INF, prev + 1) : 0;
        gg[(size_t)y * W + x] = prev;
    }
    for (int y = H - 2; y >= 0; y--) {
        const int below = gg[(size_t)(y + 1) * W + x], cur = gg[(size_t)y * W + x];
        if (below < cur) gg[(size_t)y * W + x] = min(cur, below + 1);
    }
    if (zero) atomicOr(&any_zero[img], 1);
}
// Row phase: dist2(y, x) = min over i of (x - i)^2 + g(y, i)^2 -- the lower envelope Meijster's scan builds, evaluated
// directly: one block per row, g^2 of the row in LDS, every thread scans outward from its own column and stops once the
// horizontal offset alone exceeds the best value (exact integers, so the minimum is the same number).
__global__ __launch_bounds__(256) void edt_rows_kernel(const int *__restrict__ g, int H, int W, const int *__restrict__ any_zero,
                                                       double *__restrict__ dist)
{
    extern __shared__ long long g2[];           // [W]
    const int img = blockIdx.y, y = blockIdx.x;
    double *dd = dist + ((size_t)img * H + y) * W;
    if (!any_zero[img]) {     // scipy quirk without background: distance to the virtual pixel (-1, 0)
        for (int x = threadIdx.x; x < W; x += 256) dd[x] = sqrt((double)((long long)(y + 1) * (y + 1) + (long long)x * x));
        return;
    }
    const int *gr = g + ((size_t)img * H + y) * W;
    for (int x = threadIdx.x; x < W; x += 256) { const long long v = gr[x]; g2[x] = v * v; }
    __syncthreads();
    for (int x = threadIdx.x; x < W; x += 256) {
        long long best = g2[x];
        for (int d = 1; (long long)d * d < best && (x - d >= 0 || x + d < W); d++) {
            const long long dd2 = (long long)d * d;
            if (x - d >= 0) { const long long v = dd2 + g2[x - d]; best = v < best ? v : best; }
            if (x + d < W) { const long long v = dd2 + g2[x + d]; best = v < best ? v : best; }
        }
        dd[x] = sqrt((double)best);
    }
}

// exact EDT of `mask` (distance of every pixel to the nearest zero pixel): g (n ints), st (2n ints), any_zero (k ints) scratch
void launch_edt(const uint8_t *mask, int k, int H, int W, int *g, int *st, int *any_zero, double *dist, hipStream_t s)
{
    hipMemsetAsync(any_zero, 0, k * sizeof(int), s);
    hipLaunchKernelGGL(edt_cols_kernel, dim3((W + 63) / 64, k), dim3(64), 0, s, mask, g, H, W, any_zero);
    (void)st;
    hipLaunchKernelGGL(edt_rows_kernel, dim3(H, k), dim3(256), (size_t)W * sizeof(long long), s, g, H, W, any_zero, dist);
}

// ---------------------------------------------------------------------------------------------
// driver
// ---------------------------------------------------------------------------------------------
size_t morph_workspace_bytes(int k, int H, int W)
{
    const size_t npx = (size_t)k * H * W;
    // seg, med, skA, skB (u8) + ML, SL, g, area, n1, n2, n3, fork, drop (int) + st (2 ints) + flags
    const int launches = ((H > W ? H : W) / 2 + 8 + ZH_IT - 1) / ZH_IT + 1;      // thinning launches (filter_edt_dev)
    const size_t tiles = (size_t)((W + ZH_TI - 1) / ZH_TI) * ((H + ZH_TI - 1) / ZH_TI);
    return npx * 4 + npx * sizeof(int) * 11 + (3 + (size_t)launches + 3 * tiles) * k * sizeof(int) + 4096;
}
const int *morph_done_flags(void *workspace, int k, int H, int W)
{
    const size_t n = (size_t)k * H * W;
    uint8_t *skB = (uint8_t *)workspace + 3 * n;
    int *ML = (int *)(((uintptr_t)(skB + n) + 15) & ~(uintptr_t)15);
    return ML + 11 * n + 2 * k;
}

int filter_edt_dev(const double *pred, int k, int H, int W, int remove_isolated, void *workspace, uint8_t *filt_out,
                   double *dist_out, hipStream_t s)
{
    return filter_mask_dev(pred, nullptr, k, H, W, 1, remove_isolated, workspace, filt_out, dist_out, s);
}

// General form: the mask comes from `pred > 0.5` or, when `mask_in` is given, from a u8 mask; the 13-tap median is
// optional (filter_branch_seg_mask(mask, footprint=None)); dist_out may be null (no EDT).
int filter_mask_dev(const double *pred, const uint8_t *mask_in, int k, int H, int W, int use_median, int remove_isolated, void *workspace,
                    uint8_t *filt_out, double *dist_out, hipStream_t s)
{
    if (!upload_lut()) { set_error("morph: cannot upload the skeletonize table"); return -2; }
    const int npx = H * W;
    const size_t n = (size_t)k * npx;
    uint8_t *seg = (uint8_t *)workspace, *med = seg + n, *skA = med + n, *skB = skA + n;
    int *ML = (int *)(((uintptr_t)(skB + n) + 15) & ~(uintptr_t)15);
    int *SL = ML + n, *g = SL + n, *area = g + n, *n1 = area + n, *n2 = n1 + n, *n3 = n2 + n, *fork = n3 + n, *drop = fork + n;
    int *st = drop + n;                 // 2n ints
    int *flags = st + 2 * n;            // [0,k): zhang changed, [k,2k): any_zero, [2k,3k): zhang done
    const dim3 grid((npx + 255) / 256 < 1024 ? (npx + 255) / 256 : 1024, k), blk(256);

    if (mask_in) hipLaunchKernelGGL(binarize_kernel, grid, blk, 0, s, mask_in, seg, npx);
    else hipLaunchKernelGGL(threshold_kernel, grid, blk, 0, s, pred, seg, npx);
    if (use_median) hipLaunchKernelGGL(median13_kernel, grid, blk, 0, s, seg, med, H, W);
    else if (hipMemcpyAsync(med, seg, n, hipMemcpyDeviceToDevice, s) != hipSuccess) { set_error("morph: copy"); return -2; }
    // labels of the median-filtered mask
    hipLaunchKernelGGL(ccl_init_kernel, grid, blk, 0, s, med, ML, H, W);
    hipLaunchKernelGGL(ccl_merge_kernel, grid, blk, 0, s, med, ML, H, W);
    hipLaunchKernelGGL(ccl_compress_kernel, grid, blk, 0, s, ML, npx);
    if (hipMemsetAsync(area, 0, n * sizeof(int) * 6, s) != hipSuccess) { set_error("morph: memset"); return -2; }   // area,n1,n2,n3,fork,drop
    hipLaunchKernelGGL(region_stats_kernel, grid, blk, 0, s, ML, H, W, area, n1, n2, n3);
    // Zhang thinning to convergence: (first, second) sub-iteration pairs until a whole pair removes nothing
    if (hipMemcpyAsync(skA, med, n, hipMemcpyDeviceToDevice, s) != hipSuccess) { set_error("morph: copy"); return -2; }
    // No host round trip: changed[launch][image] flags live on the device; once a launch removes nothing for an image
    // the remaining launches return at once.  A component of width w needs about w/2 iterations;
    // max(H, W)/2 + 8 always suffice (checked by the caller through flags[2k..3k) == 1, copied back with the results).
    int *done = flags + 2 * k;
    const int max_pairs = (H > W ? H : W) / 2 + 8;
    const int launches = (max_pairs + ZH_IT - 1) / ZH_IT + 1;
    int *chg = flags + 3 * k;            // [launches][k]
    const int tiles_x = (W + ZH_TI - 1) / ZH_TI, tiles_y = (H + ZH_TI - 1) / ZH_TI;
    int *tflags = chg + (size_t)launches * k;          // [3][k][tiles]: per-tile "removed something" of the launches it - 1, it, it + 1
    if (hipMemsetAsync(flags, 0, (3 + (size_t)launches + 3 * (size_t)tiles_x * tiles_y) * k * sizeof(int), s) != hipSuccess) { set_error("morph: memset"); return -2; }
    const dim3 zgrid(tiles_x * tiles_y, k);
    for (int it = 0; it < launches; it++) {
        const uint8_t *src = (it & 1) ? skB : skA;
        uint8_t *dst = (it & 1) ? skA : skB;
        hipLaunchKernelGGL(zhang_tile_kernel, zgrid, blk, 0, s, src, dst, H, W, tiles_x, it ? chg + (size_t)(it - 1) * k : nullptr,
                           chg + (size_t)it * k, tflags, it);
    }
    hipLaunchKernelGGL(zhang_done_kernel, dim3((k + 63) / 64), dim3(64), 0, s, chg + (size_t)(launches - 1) * k, done, k);
    // skeleton components, fork test, decision, filtered mask
    hipLaunchKernelGGL(ccl_init_kernel, grid, blk, 0, s, skA, SL, H, W);
    hipLaunchKernelGGL(ccl_merge_kernel, grid, blk, 0, s, skA, SL, H, W);
    hipLaunchKernelGGL(ccl_compress_kernel, grid, blk, 0, s, SL, npx);
    hipLaunchKernelGGL(skel_fork_kernel, grid, blk, 0, s, skA, SL, H, W, fork);
    hipLaunchKernelGGL(decide_kernel, grid, blk, 0, s, SL, ML, fork, area, n1, n2, n3, npx, remove_isolated, drop);
    hipLaunchKernelGGL(apply_drop_kernel, grid, blk, 0, s, med, ML, drop, filt_out, npx);
    // exact EDT of the filtered mask
    if (dist_out) launch_edt(filt_out, k, H, W, g, st, flags + k, dist_out, s);
    if (hipGetLastError() != hipSuccess) { set_error("morph: kernel launch failed"); return -2; }
    return 0;
}

}  // namespace tmat
