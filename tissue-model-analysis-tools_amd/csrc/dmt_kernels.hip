// Device front end of compute_dmt_graph (reference fl_tissue_model_tools/dmtgraph.py:57-93): negate the field, enumerate
// the edges of the anti-diagonal triangulation, drop those with an endpoint |v| <= 1e-8 (:71-77), and order the rest by
// (max endpoint value, position in the filtered list) (:80-88, np.lexsort) -- a batch of images per launch, fed from the
// 0..255 field that the finish stage left in HBM.  What crosses PCIe is the sorted edge-id list; the two union-find
// sweeps and `collect` (:277-453) are Kruskal-ordered and stay on host threads (csrc/dmt.cpp).
//
//   dmt_keys_kernel : one thread per edge id e of the canonical enumeration (vertical, horizontal, anti-diagonal; the
//                     order of dmtgraph.py:create_edges): key = order-preserving uint32 image of max(val[a], val[b]), or
//                     0xFFFFFFFF for a dropped edge (sorts behind every live key); wave-aggregated count of live edges.
//   dmt_sort_kernel : stable LSD radix sort (4 passes of 8 bits) of (key, e) pairs, one 1024-thread workgroup per image
//                     (block_radix_sort.h).  Stability IS the reference's tie-break (position in the filtered list =
//                     canonical order).  HBM-bound in principle
//                     (16 B per pair and pass); with one workgroup per image it is latency-bound (~0.3 ms per pass) and
//                     runs on the finish stream under the UNet of the next pass.
#include "tmat_internal.h"
#include "block_radix_sort.h"

#include <cstdlib>
#include <utility>

namespace tmat {

__device__ __forceinline__ uint32_t dmt_sort_key(float v)
{
    const uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__global__ __launch_bounds__(256) void dmt_keys_kernel(const float *__restrict__ field, int R, int C, int nE, uint32_t *__restrict__ keys,
                                                       int32_t *__restrict__ ids, int *__restrict__ m_out)
{
    const int img = blockIdx.y;
    const int e = blockIdx.x * 256 + threadIdx.x;
    const float *f = field + (size_t)img * R * C;
    const int nVert = (R - 1) * C, nHor = R * (C - 1);
    bool live = false;
    uint32_t key = 0xFFFFFFFFu;
    if (e < nE) {
        int a, b;
        if (e < nVert) { a = e; b = e + C; }
        else if (e < nVert + nHor) { const int q = e - nVert, r = q / (C - 1), c = q - r * (C - 1); a = r * C + c; b = a + 1; }
        else { const int q = e - nVert - nHor, r = q / (C - 1), c = q - r * (C - 1); a = r * C + c + 1; b = a + C - 1; }
        const float va = -f[a], vb = -f[b];
        live = !(fabs((double)va) <= 1e-8) && !(fabs((double)vb) <= 1e-8);
        if (live) key = dmt_sort_key(va > vb ? va : vb);
        keys[(size_t)img * nE + e] = key;
        ids[(size_t)img * nE + e] = e;
    }
    const unsigned long long bal = __ballot(live);
    if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&m_out[img], __popcll(bal));
}

__global__ __launch_bounds__(BRS_T) void dmt_sort_kernel(uint32_t *__restrict__ k0, int32_t *__restrict__ v0, uint32_t *__restrict__ k1,
                                                         int32_t *__restrict__ v1, int n)
{
    const size_t off = (size_t)blockIdx.x * n;
    block_radix_sort<uint32_t, int32_t, 4>(k0 + off, v0 + off, k1 + off, v1 + off, n);
}

// ---- the same sort across workgroups (round 3) -----------------------------------------------------------------------------
// One 1024-thread workgroup per image walks 4 x 431 chunks with four barriers each: 3.6 ms for one 384 x 384 field, a third
// of the Z-stack branch's GPU time.  Here a pass is three launches over tiles of 4096 pairs (108 tiles per 384^2 image):
//   ms_hist    per tile: digit histogram (LDS atomics) -> hist[img][digit][tile]
//   ms_scan    per image: exclusive scan of the 256 x tiles counters in (digit, tile) order = where every tile's run of every
//              digit starts in the output
//   ms_scatter per tile: its four waves own 1024 consecutive pairs each; per-wave digit counts, prefixed over the waves on
//              top of the tile's bases, then every wave walks its pairs in order in groups of 64: a lane's rank among the
//              lanes of the group with the same digit (8 ballots + popcount) plus the wave's running base of that digit.
// Input order is preserved at every level (tiles, waves, groups, lanes), so the sort is stable -- the reference's tie-break.
constexpr int MS_T = 256, MS_TILE = 4096, MS_PER_WAVE = MS_TILE / (MS_T / 64);

__global__ __launch_bounds__(MS_T) void ms_hist_kernel(const uint32_t *__restrict__ keys, int n, int sh, int ntiles, unsigned *__restrict__ hist)
{
    __shared__ unsigned h[256];
    const int img = blockIdx.y, tile = blockIdx.x;
    h[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t *k = keys + (size_t)img * n;
    const int lo = tile * MS_TILE, hi = min(lo + MS_TILE, n);
    for (int i = lo + threadIdx.x; i < hi; i += MS_T) atomicAdd(&h[(k[i] >> sh) & 255u], 1u);
    __syncthreads();
    hist[((size_t)img * 256 + threadIdx.x) * ntiles + tile] = h[threadIdx.x];
}

__global__ __launch_bounds__(1024) void ms_scan_kernel(unsigned *__restrict__ hist, int total)
{
    // exclusive scan of `total` = 256 x tiles counters of one image, in place: a contiguous run per thread, block scan of the sums
    __shared__ unsigned wsum[16];
    unsigned *h = hist + (size_t)blockIdx.x * total;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int per = (total + 1023) / 1024, lo = t * per, hi = min(lo + per, total);
    unsigned s = 0;
    for (int i = lo; i < hi; i++) s += h[i];
    unsigned incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned u = __shfl_up(incl, o); if (lane >= o) incl += u; }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    unsigned base = incl - s;
    for (int w = 0; w < wave; w++) base += wsum[w];
    for (int i = lo; i < hi; i++) { const unsigned c = h[i]; h[i] = base; base += c; }
}

__global__ __launch_bounds__(MS_T) void ms_scatter_kernel(const uint32_t *__restrict__ kin, const int32_t *__restrict__ vin, uint32_t *__restrict__ kout,
                                                          int32_t *__restrict__ vout, int n, int sh, int ntiles, const unsigned *__restrict__ hist)
{
    __shared__ unsigned wcnt[MS_T / 64][256];
    const int img = blockIdx.y, tile = blockIdx.x;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const size_t off = (size_t)img * n;
    for (int j = t; j < (MS_T / 64) * 256; j += MS_T) (&wcnt[0][0])[j] = 0;
    __syncthreads();
    const int w_lo = tile * MS_TILE + wave * MS_PER_WAVE;
    for (int g = 0; g < MS_PER_WAVE; g += 64) {
        const int i = w_lo + g + lane;
        if (i < n) atomicAdd(&wcnt[wave][(kin[off + i] >> sh) & 255u], 1u);
    }
    __syncthreads();
    {   // digit t: the waves' counts -> their output bases
        unsigned run = hist[((size_t)img * 256 + t) * ntiles + tile];
#pragma unroll
        for (int w = 0; w < MS_T / 64; w++) { const unsigned c = wcnt[w][t]; wcnt[w][t] = run; run += c; }
    }
    __syncthreads();
    for (int g = 0; g < MS_PER_WAVE; g += 64) {
        const int i = w_lo + g + lane;
        const bool valid = i < n;
        const uint32_t key = valid ? kin[off + i] : 0u;
        const int32_t val = valid ? vin[off + i] : 0;
        const unsigned d = (key >> sh) & 255u;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const bool bit = (d >> b) & 1;
            const unsigned long long bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        const unsigned rank = __popcll(peers & ((1ull << lane) - 1ull));
        unsigned pos = 0;
        if (valid) pos = wcnt[wave][d] + rank;
        __builtin_amdgcn_wave_barrier();             // every lane of the group has read the base before its first lane advances it
        if (valid && rank == 0) wcnt[wave][d] += (unsigned)__popcll(peers);
        __builtin_amdgcn_wave_barrier();
        if (valid) { kout[off + pos] = key; vout[off + pos] = val; }
    }
}

static int dmt_tiles(int nE) { return (nE + MS_TILE - 1) / MS_TILE; }

size_t dmt_workspace_bytes(int n, int R, int C)
{
    const size_t nE = (size_t)(R - 1) * C + (size_t)R * (C - 1) + (size_t)(R - 1) * (C - 1);
    return (size_t)n * nE * 16 + 256 + (size_t)n * 256 * dmt_tiles((int)nE) * sizeof(unsigned);
}

// field (n, R, C) f32 on the device -> ids (n, nE) int32: per image the edge ids in lower-star order (the first m[i] are
// the kept edges), m (n) their counts.  ws: dmt_workspace_bytes.  Asynchronous on `s`.
int dmt_sorted_edges_dev(const float *field, int n, int R, int C, void *ws, int32_t *ids, int *m, hipStream_t s)
{
    if (n <= 0) return 0;
    if (R < 2 || C < 2) return -1;
    const int nE = (R - 1) * C + R * (C - 1) + (R - 1) * (C - 1);
    uint32_t *k0 = (uint32_t *)ws, *k1 = k0 + (size_t)n * nE;
    int32_t *v1 = (int32_t *)(k1 + (size_t)n * nE);
    if (hipMemsetAsync(m, 0, n * sizeof(int), s) != hipSuccess) return -2;
    hipLaunchKernelGGL(dmt_keys_kernel, dim3((nE + 255) / 256, n), dim3(256), 0, s, field, R, C, nE, k0, ids, m);
    static const bool one_wg = [] { const char *e = getenv("TMAT_DMT_SORT_ONE_WG"); return e && atoi(e) != 0; }();
    if (one_wg) {
        hipLaunchKernelGGL(dmt_sort_kernel, dim3(n), dim3(BRS_T), 0, s, k0, ids, k1, v1, nE);     // 4 passes: the result is back in (k0, ids)
        return hipGetLastError() == hipSuccess ? 0 : -2;
    }
    const int ntiles = dmt_tiles(nE);
    unsigned *hist = (unsigned *)((char *)ws + (size_t)n * nE * 16 + 256);
    uint32_t *kin = k0, *kout = k1;
    int32_t *vin = ids, *vout = v1;
    for (int pass = 0; pass < 4; pass++) {          // 4 passes of 8 bits: the result is back in (k0, ids)
        hipLaunchKernelGGL(ms_hist_kernel, dim3(ntiles, n), dim3(MS_T), 0, s, kin, nE, 8 * pass, ntiles, hist);
        hipLaunchKernelGGL(ms_scan_kernel, dim3(n), dim3(1024), 0, s, hist, 256 * ntiles);
        hipLaunchKernelGGL(ms_scatter_kernel, dim3(ntiles, n), dim3(MS_T), 0, s, kin, vin, kout, vout, nE, 8 * pass, ntiles, hist);
        std::swap(kin, kout);
        std::swap(vin, vout);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace tmat
