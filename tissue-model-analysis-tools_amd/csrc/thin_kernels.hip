// Ordered thinning of skimage.morphology.medial_axis (reference call: compute_branches.py:340; scikit-image 0.18.3
// _skeletonize.py:436-512 with the RandomState(0) tie-break -- SURVEY hard part 4) on the device, a batch of images per
// launch, fed from the filtered mask and its exact EDT that morph_kernels.hip left in HBM.
//
// As written the algorithm is sequential: foreground pixels are visited in the order of (distance, corner score, random
// tie-break) and every visit rewrites the pixel from the CURRENT state of its 3x3 neighbourhood through a 512-entry
// table.  But a visit only reads the pixel's 8 neighbours, so the total order matters only between NEIGHBOURS: pixel p
// needs exactly those neighbours that precede it in the order to have been visited (they then hold their final value:
// every pixel is visited once) and those that follow it to be untouched.  That is a dependency DAG with at most 8
// predecessors per pixel, and it is evaluated here as a wavefront -- no sort, no sequential walk:
//   ma_keys_kernel  : raster-order index i of every foreground pixel (chunk counts, a scan, ballot ranks) and its 64-bit
//                     order key  dist^2 << 36 | (9 - count3x3) << 32 | tie[i]  (dist^2 is an exact integer;
//                     tie = numpy's legacy RandomState(0).permutation(n), a Mersenne-Twister shuffle that stays on the
//                     host: it depends only on n).  Keys are distinct.
//   ma_dep_kernel   : per pixel one byte: which of its 8 neighbours are foreground with a SMALLER key (its predecessors).
//   ma_round_kernel : tiles of 64 x 64 pixels with a 16-pixel halo staged in LDS as (state, done) bytes + the predecessor
//                     bytes (27 KB); 16 rounds per launch: a pixel whose predecessors are all done takes
//                     table[current 3x3 state] and becomes done.  Rounds are strict Jacobi steps (two LDS copies of the
//                     bytes): information moves exactly one pixel per round, so what a workgroup finishes in its inner tile
//                     never depends on pixels beyond its 16-pixel halo -- with in-place updates a workgroup could resolve a
//                     longer chain in one launch than its neighbour can see, and write back a pixel whose predecessor (in
//                     the neighbour's tile) is still undone in global memory.  Launches
//                     ping-pong between two copies of the bytes (every workgroup stages the same snapshot) and repeat until no image has an undone pixel (a flag per image; the depth of the DAG is about the largest
//                     distance value: a few launches of ~30 us on all CUs instead of one wave per image for 20 ms).
#include "tmat_internal.h"
#include "dev_guard.h"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace tmat {

__global__ __launch_bounds__(256) void ma_count_kernel(const uint8_t *__restrict__ mask, size_t per, int *__restrict__ nfg)
{
    const uint8_t *m = mask + (size_t)blockIdx.y * per;
    int c = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < per; i += (size_t)gridDim.x * 256) c += m[i] ? 1 : 0;
    for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(&nfg[blockIdx.y], c);
}

// raster-order rank of every foreground pixel = exclusive prefix of the foreground counts of the 1024-pixel chunks in front
// of its chunk + its ballot rank inside the chunk: ma_chunk_count_kernel, ma_chunk_scan_kernel, ma_keys_kernel
__global__ __launch_bounds__(1024) void ma_chunk_count_kernel(const uint8_t *__restrict__ mask, size_t per, int nchunk, int *__restrict__ cnt)
{
    __shared__ int wtot[16];
    const int img = blockIdx.y, t = threadIdx.x;
    const size_t p = (size_t)blockIdx.x * 1024 + t;
    const bool fg = p < per && mask[(size_t)img * per + p];
    const unsigned long long bal = __ballot(fg);
    if ((t & 63) == 0) wtot[t >> 6] = __popcll(bal);
    __syncthreads();
    if (t == 0) { int s = 0; for (int w = 0; w < 16; w++) s += wtot[w]; cnt[(size_t)img * nchunk + blockIdx.x] = s; }
}

__global__ __launch_bounds__(1024) void ma_chunk_scan_kernel(int *__restrict__ cnt, int nchunk)
{
    // exclusive scan of one image's chunk counts in place (nchunk is a few hundred: one workgroup, serial over 1024-wide slabs)
    __shared__ int wsum[16];
    __shared__ int carry;
    int *c = cnt + (size_t)blockIdx.x * nchunk;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int i0 = 0; i0 < nchunk; i0 += 1024) {
        const int i = i0 + t;
        const int v = i < nchunk ? c[i] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o); if (lane >= o) incl += u; }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int off = carry;
        for (int w = 0; w < wave; w++) off += wsum[w];
        if (i < nchunk) c[i] = off + incl - v;
        __syncthreads();
        if (t == 0) { int s = 0; for (int w = 0; w < 16; w++) s += wsum[w]; carry += s; }
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void ma_keys_kernel(const uint8_t *__restrict__ mask, const double *__restrict__ dist, int H, int W,
                                                       const uint32_t *__restrict__ tie, const int *__restrict__ chunk_off, int nchunk,
                                                       uint64_t *__restrict__ keys)
{
    __shared__ int wtot[16];
    const int img = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const size_t per = (size_t)H * W;
    const uint8_t *m = mask + img * per;
    const size_t p = (size_t)blockIdx.x * 1024 + t;
    const bool fg = p < per && m[p];
    const unsigned long long bal = __ballot(fg);
    const int rank = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wtot[wave] = __popcll(bal);
    __syncthreads();
    int off = chunk_off[(size_t)img * nchunk + blockIdx.x];
    for (int w = 0; w < wave; w++) off += wtot[w];
    if (p >= per) return;
    uint64_t key = ~0ull;
    if (fg) {
        const int y = (int)(p / W), x = (int)(p - (size_t)y * W);
        int cnt = 0;
#pragma unroll
        for (int dy = -1; dy <= 1; dy++)
#pragma unroll
            for (int dx = -1; dx <= 1; dx++) {
                const int yy = y + dy, xx = x + dx;
                cnt += (yy >= 0 && yy < H && xx >= 0 && xx < W && m[(size_t)yy * W + xx]) ? 1 : 0;
            }
        const double dv = dist[img * per + p];
        const unsigned long long d2 = (unsigned long long)__double2ll_rn(dv * dv);
        key = (d2 << 36) | ((unsigned long long)(9 - cnt) << 32) | tie[img * per + off + rank];
    }
    keys[img * per + p] = key;
}

// sd: bit 0 = current value of the pixel, bit 1 = done (background: value 0, done); dep: bit k = neighbour k (raster order
// of the 3x3 window without its centre) is a predecessor
__global__ __launch_bounds__(256) void ma_dep_kernel(const uint64_t *__restrict__ keys, int H, int W, uint8_t *__restrict__ sd, uint8_t *__restrict__ dep)
{
    const size_t per = (size_t)H * W;
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= per) return;
    const uint64_t *k = keys + (size_t)blockIdx.y * per;
    const uint64_t me = k[p];
    unsigned d = 0;
    if (me != ~0ull) {
        const int y = (int)(p / W), x = (int)(p - (size_t)y * W);
        int bit = 0;
#pragma unroll
        for (int dy = -1; dy <= 1; dy++)
#pragma unroll
            for (int dx = -1; dx <= 1; dx++) {
                if (dy == 0 && dx == 0) continue;
                const int yy = y + dy, xx = x + dx;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W && k[(size_t)yy * W + xx] < me) d |= 1u << bit;
                bit++;
            }
    }
    sd[(size_t)blockIdx.y * per + p] = me != ~0ull ? 1 : 2;
    dep[(size_t)blockIdx.y * per + p] = (uint8_t)d;
}

constexpr int MA_T = 64, MA_R = 16, MA_S = MA_T + 2 * MA_R;      // tile, halo = rounds per launch, staged side
__global__ __launch_bounds__(256) void ma_round_kernel(const uint8_t *__restrict__ sd, uint8_t *__restrict__ sd_out, const uint8_t *__restrict__ dep,
                                                       int H, int W, const uint32_t *__restrict__ table, int *__restrict__ undone,
                                                       const int *__restrict__ active, uint8_t *__restrict__ tile_final, int *__restrict__ tprog,
                                                       uint8_t *__restrict__ tile_left, int launch)
{
    __shared__ uint8_t s_a[MA_S * MA_S];
    __shared__ uint8_t s_b[MA_S * MA_S];
    __shared__ uint8_t s_dp[MA_S * MA_S];
    __shared__ unsigned tbl[16];
    __shared__ int any;
    const int img = blockIdx.z;
    const size_t per = (size_t)H * W;
    // A tile whose pixels were all done in the INPUT snapshot of some launch has been copied to the other snapshot by that
    // launch: both copies are final and every later launch skips it (its neighbours read final values from either copy).
    const int ntiles = gridDim.x * gridDim.y, tile = blockIdx.y * gridDim.x + blockIdx.x;
    uint8_t *final_flag = tile_final + (size_t)img * ntiles + tile;
    // Tile activity (round 4): tprog[launch % 3][img][tile] = "a staged pixel of this tile became done in that launch".  A tile none of
    // whose 3 x 3 neighbours (itself included) saw that in the previous launch has no ready pixel now -- readiness only changes when a
    // predecessor becomes done, information moves 16 pixels per launch and the tiles are 64 wide -- and wrote nothing new then, so both
    // snapshots already hold its pixels: it returns without staging anything, only repeating what it last said about undone pixels.
    // (The slot of the next launch is cleared here: nobody reads it during this one.)
    int *tp_w = tprog + ((size_t)(launch % 3) * gridDim.z + img) * ntiles;
    const int *tp_r = tprog + ((size_t)((launch + 2) % 3) * gridDim.z + img) * ntiles;
    if (threadIdx.x == 0) tprog[((size_t)((launch + 1) % 3) * gridDim.z + img) * ntiles + tile] = 0;
    if (*final_flag) return;
    uint8_t *left_flag = tile_left + (size_t)img * ntiles + tile;
    if (launch > 0) {
#ifdef MA_VAR_NOSKIP
        int act = 1;
#else
        int act = 0;
#endif
        for (int dy = -1; dy <= 1; dy++)
            for (int dx = -1; dx <= 1; dx++) {
                const int yy = (int)blockIdx.y + dy, xx = (int)blockIdx.x + dx;
                if (yy >= 0 && yy < (int)gridDim.y && xx >= 0 && xx < (int)gridDim.x) act |= tp_r[yy * gridDim.x + xx];
            }
        if (!act) {
            if (threadIdx.x == 0 && *left_flag) atomicOr(&undone[img], 1);
            return;
        }
    }
    const uint8_t *g_sd = sd + img * per;
    uint8_t *g_out = sd_out + img * per;
    const uint8_t *g_dp = dep + img * per;
    // Launches ping-pong between two copies of the (value, done) bytes: every workgroup of a launch stages the SAME
    // snapshot.
    const bool idle = active && !active[img];        // every pixel of this image was done before this launch: copy through
    const int y0 = blockIdx.y * MA_T - MA_R, x0 = blockIdx.x * MA_T - MA_R;
    const int t = threadIdx.x;
    if (t < 16) tbl[t] = table[t];
    if (t == 0) any = 0;
    for (int i = t; i < MA_S * MA_S; i += 256) {
        const int ly = i / MA_S, lx = i - ly * MA_S, y = y0 + ly, x = x0 + lx;
        const bool in = y >= 0 && y < H && x >= 0 && x < W;
        const uint8_t v = in ? g_sd[(size_t)y * W + x] : 2;          // outside the image: value 0, done
        s_a[i] = v; s_b[i] = v;                                       // the rim is never updated: both copies keep it
        s_dp[i] = in ? g_dp[(size_t)y * W + x] : 0;
    }
    __syncthreads();
    // offsets of the 8 neighbours in raster order of the window without its centre
    const int noff[8] = {-MA_S - 1, -MA_S, -MA_S + 1, -1, 1, MA_S - 1, MA_S, MA_S + 1};
    uint8_t *cur = s_a, *nxt = s_b;
    // a staged area without an undone pixel (most tiles after the first launches) only copies through
    bool undone_here = false, undone_inner = false;
    for (int i = t; i < MA_S * MA_S; i += 256) {
        const bool u = !(s_a[i] & 2);
        const int ly = i / MA_S, lx = i - ly * MA_S;
        undone_here |= u;
        undone_inner |= u && ly >= MA_R && ly < MA_R + MA_T && lx >= MA_R && lx < MA_R + MA_T;
    }
    const bool inner_done_at_input = !__syncthreads_or(undone_inner);
    const bool work = __syncthreads_or(undone_here) && !idle;
    bool progress = false;
    for (int r = 0; r < (work ? MA_R : 0); r++) {
        bool changed = false;
        // pixels on the rim of the staged area have neighbours that are not staged: they keep waiting (conservative)
        for (int i = t; i < (MA_S - 2) * (MA_S - 2); i += 256) {
            const int ly = i / (MA_S - 2) + 1, lx = i - (ly - 1) * (MA_S - 2) + 1, c = ly * MA_S + lx;
            const unsigned me = cur[c];
            unsigned out = me;
            if (!(me & 2)) {
                const unsigned d = s_dp[c];
                bool ready = true;
                unsigned acc = 16;
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const unsigned nb = cur[c + noff[k]];
                    if (((d >> k) & 1) && !(nb & 2)) ready = false;
                    acc |= (nb & 1) << (k < 4 ? k : k + 1);
                }
                if (ready) { out = 2 | ((tbl[acc >> 5] >> (acc & 31)) & 1u); changed = true; }
            }
            nxt[c] = (uint8_t)out;
        }
        const bool any_changed = __syncthreads_or(changed);
        uint8_t *tp = cur; cur = nxt; nxt = tp;
        if (!any_changed) break;            // a round that finishes nothing: nothing can become ready later in this launch
        progress = true;
    }
    const uint8_t *s_sd = cur;
    bool left = false;
    for (int i = t; i < MA_T * MA_T; i += 256) {
        const int ly = i / MA_T + MA_R, lx = i % MA_T + MA_R, y = y0 + ly, x = x0 + lx;
        if (y < H && x < W) {
            const uint8_t v = s_sd[ly * MA_S + lx];
            g_out[(size_t)y * W + x] = v;
            left |= !(v & 2);
        }
    }
    if (left) any = 1;
    __syncthreads();
    if (t == 0 && any) atomicOr(&undone[img], 1);
    if (t == 0) { *left_flag = (uint8_t)(any != 0); if (progress) tp_w[tile] = 1; }
    if (t == 0 && inner_done_at_input) *final_flag = 1;
}

__global__ __launch_bounds__(256) void ma_finish_kernel(const uint8_t *__restrict__ sd, size_t total, uint8_t *__restrict__ skel)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) skel[i] = sd[i] & 1;
}

bool thin_dev_supported(int H, int W) { return H >= 1 && W >= 1 && (unsigned long long)H * H + (unsigned long long)W * W < (1ull << 27); }

size_t thin_workspace_bytes(int n, int H, int W)
{
    const size_t per = (size_t)H * W;
    const size_t tiles = (size_t)((H + MA_T - 1) / MA_T) * ((W + MA_T - 1) / MA_T);
    return (size_t)n * per * (8 + 1 + 1 + 1) + (size_t)n * 2 * sizeof(int) * 64 + (size_t)n * ((per + 1023) / 1024) * sizeof(int) + n * tiles * (2 + 3 * sizeof(int)) + 1024;
}

int thin_count_dev(const uint8_t *mask, int n, int H, int W, int *nfg, hipStream_t s)
{
    if (hipMemsetAsync(nfg, 0, n * sizeof(int), s) != hipSuccess) return -2;
    const size_t per = (size_t)H * W;
    hipLaunchKernelGGL(ma_count_kernel, dim3((unsigned)std::min<size_t>((per + 255) / 256, 256), n), dim3(256), 0, s, mask, per, nfg);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

// mask (n, H, W) u8, dist (n, H, W) f64 (its exact EDT), tie (n, H*W) u32: per image the legacy permutation of its
// foreground count (first nfg[i] entries) -> skel (n, H, W) u8.  table_dev: 16 words.  Synchronises `s` (the round
// launches repeat until a flag read back from the device says that every pixel is done).
int thin_dev(const uint8_t *mask, const double *dist, const uint32_t *tie, const int * /*nfg*/, int n, int H, int W, void *ws,
             const uint32_t *table_dev, uint8_t *skel, hipStream_t s)
{
    if (n <= 0) return 0;
    if (!thin_dev_supported(H, W)) return -1;
    const size_t per = (size_t)H * W;
    uint64_t *keys = (uint64_t *)ws;
    uint8_t *sd = (uint8_t *)(keys + (size_t)n * per), *sd2 = sd + (size_t)n * per, *dep = sd2 + (size_t)n * per;
    int *flags = (int *)(((uintptr_t)(dep + (size_t)n * per) + 15) & ~(uintptr_t)15);       // [launch][image] undone flags
    int *chunk = flags + (size_t)n * 2 * 64;                                                // per-chunk foreground counts -> offsets
    const int nchunk = (int)((per + 1023) / 1024);
    const size_t tiles = (size_t)((H + MA_T - 1) / MA_T) * ((W + MA_T - 1) / MA_T);
    uint8_t *tile_final = (uint8_t *)(chunk + (size_t)n * nchunk);                          // [image][tile]: both snapshots hold the tile's final values
    uint8_t *tile_left = tile_final + (size_t)n * tiles;                                    // [image][tile]: the tile had undone pixels when it last ran
    int *tprog = (int *)(((uintptr_t)(tile_left + (size_t)n * tiles) + 15) & ~(uintptr_t)15);      // [3][image][tile]: progress flags of launches l - 1, l, l + 1
    if (hipMemsetAsync(tile_final, 0, (size_t)n * tiles * 2 + 16 + 3 * (size_t)n * tiles * sizeof(int), s) != hipSuccess) return -2;
    hipLaunchKernelGGL(ma_chunk_count_kernel, dim3(nchunk, n), dim3(1024), 0, s, mask, per, nchunk, chunk);
    hipLaunchKernelGGL(ma_chunk_scan_kernel, dim3(n), dim3(1024), 0, s, chunk, nchunk);
    hipLaunchKernelGGL(ma_keys_kernel, dim3(nchunk, n), dim3(1024), 0, s, mask, dist, H, W, tie, chunk, nchunk, keys);
    hipLaunchKernelGGL(ma_dep_kernel, dim3((unsigned)((per + 255) / 256), n), dim3(256), 0, s, keys, H, W, sd, dep);
    const dim3 grid((W + MA_T - 1) / MA_T, (H + MA_T - 1) / MA_T, n);
    const int max_launches = (H + W) / MA_R + 8;            // a chain of predecessors cannot be longer than the pixel count of a
                                                            // monotone path; in practice the depth is about the largest distance
    std::vector<int> host(n);
    constexpr int GROUP = 4;                                // launches between two looks at the flags
    for (int l = 0;; l += GROUP) {
        if (hipMemsetAsync(flags, 0, (size_t)GROUP * n * sizeof(int), s) != hipSuccess) return -2;
        for (int g = 0; g < GROUP; g++) {                   // GROUP is even: the current copy is `sd` again after a group
            hipLaunchKernelGGL(ma_round_kernel, grid, dim3(256), 0, s, (g & 1) ? sd2 : sd, (g & 1) ? sd : sd2, dep, H, W, table_dev,
                               flags + (size_t)g * n, g ? flags + (size_t)(g - 1) * n : (const int *)nullptr, tile_final, tprog, tile_left, l + g);
        }
        if (hipMemcpyAsync(host.data(), flags + (size_t)(GROUP - 1) * n, n * sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess ||
            hipStreamSynchronize(s) != hipSuccess) return -2;
        bool left = false;
        for (int i = 0; i < n; i++) left |= host[i] != 0;
        if (!left) break;
        if (l + GROUP > 64 * max_launches) return -3;       // cannot happen: every launch finishes at least one pixel per undone image
    }
    hipLaunchKernelGGL(ma_finish_kernel, dim3(1024), dim3(256), 0, s, sd, (size_t)n * per, skel);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace tmat
