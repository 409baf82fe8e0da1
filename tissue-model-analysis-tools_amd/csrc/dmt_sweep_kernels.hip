// The two persistence sweeps of compute_dmt_graph (reference fl_tissue_model_tools/dmtgraph.py:277-314, compute_persistence_1 /
// _2: Kruskal order with the elder rule on the primal and on the dual graph) as a device kernel -- ONE WAVE PER IMAGE, opt-in
// (TMAT_DMT_SWEEP_DEVICE=1): the default keeps them on host threads (csrc/dmt.cpp), where they are hidden under the next pass
// (DESIGN.md holds the measurement).
//
// Kruskal order is sequential: every union changes the roots the next edge sees.  What a wave can do in parallel is the
// FINDs.  The sorted edges are taken 64 at a time: every lane chases the parent pointers of its edge's two endpoints (with
// path halving: concurrent lanes only ever replace a parent by an ancestor, so the forest stays valid), then the unions of
// the batch are applied in edge order through wave broadcasts -- the lane whose turn it is publishes (dead root, surviving
// root, value of the survivor) and every later lane whose stale root equals the dead one renames it.  Lanes whose two roots
// already agree are skipped (two thirds of the edges close a cycle).  The parent arrays (147 456 vertices, 293 379
// triangles at 384 x 384) live in HBM / L2: they do not fit the LDS.
#include "tmat_internal.h"

namespace tmat {

struct SweepGrid {
    int R, C, nVert, nHor;
    __device__ void endpoints(int e, int &a, int &b) const
    {
        if (e < nVert) { a = e; b = e + C; return; }
        e -= nVert;
        if (e < nHor) { const int r = e / (C - 1), c = e - r * (C - 1); a = r * C + c; b = a + 1; return; }
        e -= nHor;
        const int r = e / (C - 1), c = e - r * (C - 1);
        a = r * C + c + 1; b = a + C - 1;
    }
    __device__ void faces(int e, int &f, int &g) const
    {
        const int outer = 2 * (R - 1) * (C - 1);
        if (e < nVert) {
            const int r = e / C, c = e - r * C, t = 2 * (r * (C - 1) + c);
            f = c == 0 ? outer : t - 1;
            g = c == C - 1 ? outer : t;
            return;
        }
        e -= nVert;
        if (e < nHor) {
            const int r = e / (C - 1), c = e - r * (C - 1), t = 2 * (r * (C - 1) + c);
            f = r == 0 ? outer : t - 2 * (C - 1) + 1;
            g = r == R - 1 ? outer : t;
            return;
        }
        e -= nHor;
        f = 2 * e; g = f + 1;
    }
};

// val = -field (dmtgraph.py:57); tv[t] = value of dual vertex t (max over the triangle's corners), tv[nT] = +inf (outer face)
__global__ void dmt_prep_kernel(const float *__restrict__ field, int R, int C, float *__restrict__ val, float *__restrict__ tv,
                                int *__restrict__ p1, int *__restrict__ p2)
{
    const int nV = R * C, nT = 2 * (R - 1) * (C - 1);
    const size_t io = (size_t)blockIdx.y;
    field += io * nV; val += io * nV; tv += io * (nT + 1); p1 += io * nV; p2 += io * (nT + 1);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= nT || i < nV; i += gridDim.x * blockDim.x) {
        if (i < nV) { val[i] = -field[i]; p1[i] = i; }
        if (i <= nT) {
            p2[i] = i;
            if (i == nT) tv[i] = __builtin_inff();
            else {
                const int q = i >> 1, r = q / (C - 1), c = q - r * (C - 1);
                const float a = -field[r * C + c], b = -field[r * C + c + 1], d = -field[(r + 1) * C + c], e = -field[(r + 1) * C + c + 1];
                tv[i] = (i & 1) ? fmaxf(fmaxf(b, d), e) : fmaxf(fmaxf(a, b), d);
            }
        }
    }
}

__device__ __forceinline__ int sweep_find(int *p, int v)
{
    int pv = p[v];
    while (pv != v) {
        const int g = p[pv];
        if (g != pv) p[v] = g;          // path halving (benign race between lanes: always an ancestor)
        v = pv; pv = g;
    }
    return v;
}

// one wave per image: kind[i] (0 unpaired, 1 vertex-edge, 2 edge-triangle) and pers[i] of the sorted kept edges
__global__ __launch_bounds__(64) void dmt_sweep_kernel(const int32_t *__restrict__ ids_all, const int *__restrict__ m_all, int nE, int R, int C,
                                                       const float *__restrict__ val_all, const float *__restrict__ tv_all, int *__restrict__ p1_all,
                                                       int *__restrict__ p2_all, uint8_t *__restrict__ kind_all, float *__restrict__ pers_all)
{
    const int img = blockIdx.x, lane = threadIdx.x;
    const int nV = R * C, nT = 2 * (R - 1) * (C - 1);
    const SweepGrid gd{R, C, (R - 1) * C, R * (C - 1)};
    const int32_t *ids = ids_all + (size_t)img * nE;
    const int m = m_all[img];
    const float *val = val_all + (size_t)img * nV, *tv = tv_all + (size_t)img * (nT + 1);
    int *p1 = p1_all + (size_t)img * nV, *p2 = p2_all + (size_t)img * (nT + 1);
    uint8_t *kind = kind_all + (size_t)img * nE;
    float *pers = pers_all + (size_t)img * nE;

    // ---- ascending sweep: elder rule on vertices (the younger root -- larger value, ties: larger index -- dies) ----
    for (int base = 0; base < m; base += 64) {
        const int i = base + lane;
        const bool valid = i < m;
        int x = -1, y = -1;
        float vx = 0.f, vy = 0.f, ev = 0.f;
        if (valid) {
            int a, b;
            gd.endpoints(ids[i], a, b);
            const float va = val[a], vb = val[b];
            ev = va > vb ? va : vb;
            x = sweep_find(p1, a); y = sweep_find(p1, b);
            vx = val[x]; vy = val[y];
        }
        uint8_t k = 0;
        float pr = __builtin_inff();
        unsigned long long todo = __ballot(valid && x != y);
        while (todo) {
            const int j = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int xj = __shfl(x, j), yj = __shfl(y, j);
            if (xj == yj) continue;                              // an earlier union of this batch joined them
            const float vxj = __shfl(vx, j), vyj = __shfl(vy, j);
            const bool x_older = vxj < vyj || (vxj == vyj && xj < yj);
            const int dead = x_older ? yj : xj, keep = x_older ? xj : yj;
            const float vdead = x_older ? vyj : vxj, vkeep = x_older ? vxj : vyj;
            if (lane == j) { p1[dead] = keep; k = 1; pr = ev - vdead; }
            if (x == dead) { x = keep; vx = vkeep; }
            if (y == dead) { y = keep; vy = vkeep; }
        }
        if (valid) { kind[i] = k; pers[i] = pr; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // this batch's parent updates before the next batch's finds
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    // ---- descending sweep on the dual graph over the edges the first sweep left unpaired; the outer face always survives ----
    for (int top = m; top > 0; top -= 64) {
        const int i = top - 1 - lane;                            // lane 0 = the last (largest) edge of the batch
        const bool valid = i >= 0;
        int x = -1, y = -1;
        float vx = 0.f, vy = 0.f, ev = 0.f;
        bool cand = false;
        if (valid && kind[i] == 0) {
            const int e = ids[i];
            int a, b, f, g;
            gd.endpoints(e, a, b);
            const float va = val[a], vb = val[b];
            ev = va > vb ? va : vb;
            gd.faces(e, f, g);
            x = sweep_find(p2, f); y = sweep_find(p2, g);
            vx = tv[x]; vy = tv[y];
            cand = x != y;
        }
        uint8_t k = 0;
        float pr = 0.f;
        unsigned long long todo = __ballot(cand);
        while (todo) {
            const int j = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int xj = __shfl(x, j), yj = __shfl(y, j);
            if (xj == yj) continue;
            const float vxj = __shfl(vx, j), vyj = __shfl(vy, j);
            const bool x_wins = vxj > vyj || (vxj == vyj && xj > yj);
            const int dead = x_wins ? yj : xj, keep = x_wins ? xj : yj;
            const float vdead = x_wins ? vyj : vxj, vkeep = x_wins ? vxj : vyj;
            if (lane == j) { p2[dead] = keep; k = 2; pr = vdead - ev; }
            if (x == dead) { x = keep; vx = vkeep; }
            if (y == dead) { y = keep; vy = vkeep; }
        }
        if (k) { kind[i] = k; pers[i] = pr; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}

size_t dmt_sweep_workspace_bytes(int n, int R, int C)
{
    const size_t nV = (size_t)R * C, nT1 = 2 * (size_t)(R - 1) * (C - 1) + 1;
    return (size_t)n * (nV * 8 + nT1 * 8) + 256;
}

// ids (n, nE) sorted kept edges, m (n) their counts (device) -> kind (n, nE) u8, pers (n, nE) f32 (device).  Asynchronous on s.
int dmt_sweeps_dev(const float *field, const int32_t *ids, const int *m, int n, int R, int C, void *ws, uint8_t *kind, float *pers, hipStream_t s)
{
    if (n <= 0) return 0;
    const size_t nV = (size_t)R * C, nT1 = 2 * (size_t)(R - 1) * (C - 1) + 1;
    const int nE = (int)dmt_edge_count(R, C);
    float *val = (float *)ws, *tv = val + (size_t)n * nV;
    int *p1 = (int *)(tv + (size_t)n * nT1), *p2 = p1 + (size_t)n * nV;
    const int blocks = (int)((nT1 + 255) / 256);
    hipLaunchKernelGGL(dmt_prep_kernel, dim3(blocks < 512 ? blocks : 512, n), dim3(256), 0, s, field, R, C, val, tv, p1, p2);
    hipLaunchKernelGGL(dmt_sweep_kernel, dim3(n), dim3(64), 0, s, ids, m, nE, R, C, val, tv, p1, p2, kind, pers);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace tmat
