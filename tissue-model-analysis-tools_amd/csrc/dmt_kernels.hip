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

size_t dmt_workspace_bytes(int n, int R, int C)
{
    const size_t nE = (size_t)(R - 1) * C + (size_t)R * (C - 1) + (size_t)(R - 1) * (C - 1);
    return (size_t)n * nE * 16 + 256;
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
    hipLaunchKernelGGL(dmt_sort_kernel, dim3(n), dim3(BRS_T), 0, s, k0, ids, k1, v1, nE);     // 4 passes: the result is back in (k0, ids)
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace tmat
