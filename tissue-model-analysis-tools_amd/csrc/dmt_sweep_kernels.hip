// The two persistence sweeps of compute_dmt_graph (reference fl_tissue_model_tools/dmtgraph.py:277-314, compute_persistence_1 /
// _2: Kruskal order with the elder rule on the primal and on the dual graph) on the device, as LEVELS of data-parallel steps.
//
// As written the sweeps are sequential: every union changes the roots the next edge sees.  What the result consists of, though,
// is per edge (a) whether it joins two components and (b) the value of the younger root that dies on it, and both can be decided
// without walking the edges in order.  Nodes carry a strict age order (primal: smaller (value, index) = older; dual: larger),
// edges a strict rank (their position in the lower-star order).  A super-node is a set of nodes with its oldest member as root;
// an edge is alive while its endpoints lie in different super-nodes.  One level:
//
//   A  every alive edge looks up the super-nodes X, Y of its endpoints (find with path halving) and posts its rank to both
//      (atomicMin): first[X] = the first alive edge that touches X;
//   B  the edge that is first[X] decides for X: if the root across is OLDER, X's root dies on this edge -- the edge is a tree
//      edge with persistence |value(edge) - value(root X)| -- and X is linked under Y.  Otherwise nothing is decided for X yet.
//
// Why B is exact (induction over the levels; the invariant: just before an alive edge is processed by the sequential sweep, each
// endpoint is already connected to the root of its super-node): let e = first[X] = (a in X, b in Y).  No alive edge of smaller
// rank touches X, so at e's Kruskal time the component of a lies inside X and contains X's root: its oldest node IS root(X).  The
// component of b contains root(Y), so its oldest node is at least as old as root(Y), hence older than root(X): whatever else has
// been merged over there, root(X) is the one that dies, and e joins two different components.  Links go to strictly older roots
// (a forest); the edge that links X is processed before every other alive edge of X, so the invariant carries over to the
// linked trees, and an edge that ends up inside a tree without being a link closes a cycle at its Kruskal time (both ends are
// connected to the tree's root by then).  The globally first alive edge always links, so the levels terminate; measured 5-15
// levels per sweep on 384 x 384 fields (alive edges 193 000 -> 54 000 -> 15 000 -> ...; tools/dev/dmt_levels_proto.py checks the
// formulation against the sequential sweeps edge by edge).
//
// LV_WG workgroups of LV_THREADS threads per image run both sweeps in ONE launch per pass (on the lowest-priority stream every
// launch waits for free CUs): the phases of a level are separated by a barrier across the image's workgroups -- a monotonic
// arrival counter in memory (how the shared state is kept coherent: see below).  A launch has at most 32 x LV_WG workgroups
// of 8 waves, an eighth of what the chip holds at once, and a workgroup waits for nothing but the arrival of its image's other
// workgroups, which need no resource a waiting workgroup holds: every wave reaches the exit.
// The parent arrays (147 456 vertices, 293 379 triangles at 384 x 384), the first[] array and the alive lists live in HBM / L2.
#include "tmat_internal.h"
#include "dev_guard.h"

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

// val = -field (dmtgraph.py:57); tv[t] = value of dual vertex t (max over the triangle's corners), tv[nT] = +inf (outer face);
// the image's control words (barrier arrivals, list counters) are cleared
__global__ void dmt_prep_kernel(const float *__restrict__ field, int R, int C, float *__restrict__ val, float *__restrict__ tv, unsigned *__restrict__ ctl)
{
    const int nV = R * C, nT = 2 * (R - 1) * (C - 1);
    const size_t io = (size_t)blockIdx.y;
    field += io * nV; val += io * nV; tv += io * (nT + 1);
    if (blockIdx.x == 0 && threadIdx.x < 4) ctl[io * 4 + threadIdx.x] = 0u;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i <= nT || i < nV; i += gridDim.x * blockDim.x) {
        if (i < nV) val[i] = -field[i];
        if (i <= nT) {
            if (i == nT) tv[i] = __builtin_inff();
            else {
                const int q = i >> 1, r = q / (C - 1), c = q - r * (C - 1);
                const float a = -field[r * C + c], b = -field[r * C + c + 1], d = -field[(r + 1) * C + c], e = -field[(r + 1) * C + c + 1];
                tv[i] = (i & 1) ? fmaxf(fmaxf(b, d), e) : fmaxf(fmaxf(a, b), d);
            }
        }
    }
}

#ifndef LV_WG
#define LV_WG 8                           // workgroups per image
#endif
constexpr int LV_THREADS = 512;
constexpr int LV_STRIDE = LV_WG * LV_THREADS;
constexpr unsigned LV_NONE = 0xffffffffu;
struct LvItem { int i, x, y; };           // sorted position of the edge, roots of its endpoints when it was last looked at

// Control words of an image (4): [0] barrier arrivals, [1] [2] the list counters of alternating levels.
//
// How the workgroups of an image see each other's writes.  Loads of shared, changing data (parent pointers, first[], the lists,
// kind[], the control words) are relaxed agent-scope atomic loads: never served from a CU's L1.  first[] and the control words are
// the targets of read-modify-writes and are also only stored to by agent-scope atomics.  The rest is stored plainly (write-through
// L1, write-back L2) and made visible by the barrier: every wave waits for its own stores, then ONE wave per workgroup writes the
// XCD's L2 back before it signals the arrival and invalidates it after the wait (agent-scope release / acquire: the workgroups of
// an image may run on different XCDs, each with its own L2).  One wave, because the maintenance is per XCD, not per wave, and every
// such operation costs the convolution kernels running on that XCD cached weights: with all 8 waves of every workgroup fencing at
// every barrier the sweeps took 0.7 % off the pass, with one wave nothing measurable (32.85-32.87 against 32.86-32.92 images/s).
// Seen to fail on the way here and therefore not used: `buffer_inv sc0` to drop only the L1 when an image's workgroups share an
// XCD (it does not, for a workgroup that is not split over CUs); a workgroup-scope `fetch_add(p, 0)` as an L1-bypassing read
// (compiled to an sc0 load, which spun on a stale L1 line for ever); plain loads of words that read-modify-writes had changed
// (stale: the RMWs are performed past the L1); atomic-only accesses with no cache maintenance at all across XCDs (wrong results).
// pers[] is only written (the host reads it after the kernel); val / tv / ids are constants.
__device__ __forceinline__ unsigned lv_add(unsigned *p, unsigned v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lv_min(unsigned *p, unsigned v) { __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T> __device__ __forceinline__ T lv_ld(const T *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void lv_poke(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename T> __device__ __forceinline__ void lv_st(T *p, T v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// barrier across the LV_WG workgroups of one image: the arrival counter only grows; `due` = arrivals after this barrier
__device__ __forceinline__ void lv_barrier(unsigned *arrive, unsigned &due)
{
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // this wave's stores have reached the L2
    __syncthreads();
    due += LV_WG;
    if (threadIdx.x < 64) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (threadIdx.x == 0) {
            lv_add(arrive, 1u);
            while (lv_ld(arrive) < due) __builtin_amdgcn_s_sleep(4);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// find with path halving: inside a phase every wave reads and writes the parent pointers, but a parent is only ever replaced by
// an ancestor, so whatever value a load returns is valid; links are made in phase B only, with a barrier before the next finds
__device__ __forceinline__ int lv_find(int *p, int v)
{
    int pv = lv_ld(p + v);
    while (pv != v) {
        const int g = lv_ld(p + pv);
        if (g != pv) lv_st(p + v, g);
        v = pv; pv = g;
    }
    return v;
}

// one sweep.  DUAL = false: vertices, edges in ascending order, the root with the larger (value, index) dies, pers = edge - root;
// DUAL = true: triangles + outer face, edges in descending order over those the first sweep left unpaired, the root with the
// smaller (value, index) dies, pers = root - edge.  Levels 0 and 1 walk the sorted edges themselves (an edge takes part while its
// kind is 0; at level 0 every node is its own root); from level 1 on the survivors are kept as a list.  Returns true if a level
// made no progress (which the formulation excludes).
template <bool DUAL>
__device__ bool lv_sweep(const SweepGrid gd, const int32_t *__restrict__ ids, int m, const float *__restrict__ val, const float *__restrict__ nval,
                         int *par, unsigned *first, LvItem *la, LvItem *lb, uint8_t *kind, float *pers, unsigned *ctl, unsigned &due)
{
    const int g = blockIdx.y * LV_THREADS + threadIdx.x, lane = threadIdx.x & 63;
    int n_in = m, cur = 0;                  // ctl[1 + cur] counts this level's survivors, ctl[1 + (cur ^ 1)] is cleared for the next one
    int level = 0, n_prev = 0x7fffffff;
    auto from_edges = [&](int k, int &i, int &a, int &b) -> bool {
        i = DUAL ? m - 1 - k : k;
        if (lv_ld(kind + i) != 0) return false;
        if (DUAL) gd.faces(ids[i], a, b);
        else gd.endpoints(ids[i], a, b);
        return true;
    };
    auto decide = [&](int i, int x, int y) {
        const unsigned rk = (unsigned)(DUAL ? m - 1 - i : i);
        const bool fx = lv_ld(first + x) == rk;
        const bool fy = lv_ld(first + y) == rk;
        if (!(fx || fy)) return;
        const float vx = nval[x], vy = nval[y];
        const bool x_older = DUAL ? (vx > vy || (vx == vy && x > y)) : (vx < vy || (vx == vy && x < y));
        int ea, eb;
        gd.endpoints(ids[i], ea, eb);
        const float va = val[ea], vb = val[eb], ev = va > vb ? va : vb;
        if (fx) {
            if (!x_older) { lv_st(par + x, y); lv_st(kind + i, (uint8_t)(DUAL ? 2 : 1)); pers[i] = DUAL ? vx - ev : ev - vx; }
            lv_poke(first + x, LV_NONE);
        }
        if (fy) {
            if (x_older) { lv_st(par + y, x); lv_st(kind + i, (uint8_t)(DUAL ? 2 : 1)); pers[i] = DUAL ? vy - ev : ev - vy; }
            lv_poke(first + y, LV_NONE);
        }
    };
    for (;;) {
        // ---- phase A: roots of the alive edges, first[] of the super-nodes, the next level's list
        for (int base = 0; base < n_in; base += LV_STRIDE) {
            const int k = base + g;
            bool valid = k < n_in;
            int i = 0, a = 0, b = 0;
            if (valid) {
                if (level < 2) valid = from_edges(k, i, a, b);
                else { i = lv_ld(&la[k].i); a = lv_ld(&la[k].x); b = lv_ld(&la[k].y); }
            }
            int X = a, Y = b;
            if (valid && level) { X = lv_find(par, a); Y = lv_find(par, b); }
            const bool alive = valid && X != Y;
            const unsigned long long bal = __ballot(alive);
            if (bal) {
                const int lead = __ffsll((long long)bal) - 1;
                unsigned pos = 0;
                if (lane == lead) pos = lv_add(ctl + 1 + cur, (unsigned)__popcll(bal));
                if (level) pos = __shfl(pos, lead) + __popcll(bal & ((1ull << lane) - 1ull));
                if (alive) {
                    if (level) { lv_st(&lb[pos].i, i); lv_st(&lb[pos].x, X); lv_st(&lb[pos].y, Y); }
                    const unsigned rk = (unsigned)(DUAL ? m - 1 - i : i);
                    lv_min(first + X, rk);
                    lv_min(first + Y, rk);
                }
            }
        }
        if (g == 0) lv_poke(ctl + 1 + (cur ^ 1), 0u);
        lv_barrier(ctl, due);
        const int n_out = (int)lv_ld(ctl + 1 + cur);
        if (n_out == 0) break;
        // the first alive edge of all always links, so the count must fall from level to level: anything else is a defect (or a
        // memory-ordering assumption that does not hold on this part) and must not spin -- kind[0] = 0xFF tells the host, which
        // fails the call (dmt.cpp)
        if (n_out >= n_prev) {
            if (g == 0) lv_poke(ctl + 1 + cur, 0u);
            lv_barrier(ctl, due);
            return true;
        }
        n_prev = n_out;
        // ---- phase B: the first edge of a super-node decides for it
        if (level == 0) {
            for (int k = g; k < m; k += LV_STRIDE) {
                int i, a, b;
                if (from_edges(k, i, a, b) && a != b) decide(i, a, b);
            }
        } else {
            for (int k = g; k < n_out; k += LV_STRIDE) decide(lv_ld(&lb[k].i), lv_ld(&lb[k].x), lv_ld(&lb[k].y));
        }
        lv_barrier(ctl, due);
        if (level) { LvItem *sw = la; la = lb; lb = sw; n_in = n_out; }
        cur ^= 1;
        level++;
    }
    // the last level left ctl[1 + cur] at 0 and cleared the other counter: both are 0 for the next sweep
    return false;
}

// LV_WG workgroups per image: kind[i] (0 unpaired, 1 vertex-edge, 2 edge-triangle) and pers[i] of the sorted kept edges
__global__ __launch_bounds__(LV_THREADS) void dmt_levels_kernel(const int32_t *__restrict__ ids_all, const int *__restrict__ m_all, int nE, int R, int C,
                                                                const float *__restrict__ val_all, const float *__restrict__ tv_all, int *par_all,
                                                                unsigned *first_all, LvItem *list_all, unsigned *ctl_all, uint8_t *kind_all,
                                                                float *pers_all)
{
    const int img = blockIdx.x, g = blockIdx.y * LV_THREADS + threadIdx.x;     // (image, workgroup): a pass of 8 images puts an image's workgroups on one XCD (its atomics meet in one L2)
    const int nV = R * C, nT1 = 2 * (R - 1) * (C - 1) + 1, nN = nV > nT1 ? nV : nT1;
    const SweepGrid gd{R, C, (R - 1) * C, R * (C - 1)};
    const int32_t *ids = ids_all + (size_t)img * nE;
    const int m = m_all[img];
    const float *val = val_all + (size_t)img * nV, *tv = tv_all + (size_t)img * nT1;
    int *par = par_all + (size_t)img * nN;
    unsigned *first = first_all + (size_t)img * nN;
    LvItem *la = list_all + (size_t)img * 2 * nE, *lb = la + nE;
    unsigned *ctl = ctl_all + (size_t)img * 4;
    uint8_t *kind = kind_all + (size_t)img * nE;
    float *pers = pers_all + (size_t)img * nE;
    unsigned due = 0;

    for (int i = g; i < m; i += LV_STRIDE) { kind[i] = 0; pers[i] = __builtin_inff(); }
    for (int v = g; v < nN; v += LV_STRIDE) { par[v] = v; lv_poke(first + v, LV_NONE); }
    lv_barrier(ctl, due);
    bool bad = lv_sweep<false>(gd, ids, m, val, val, par, first, la, lb, kind, pers, ctl, due);
    for (int v = g; v < nN; v += LV_STRIDE) { par[v] = v; lv_poke(first + v, LV_NONE); }
    lv_barrier(ctl, due);
    bad |= lv_sweep<true>(gd, ids, m, val, tv, par, first, la, lb, kind, pers, ctl, due);
    if (bad && g == 0 && m > 0) kind[0] = 0xFF;             // (every sweep ends behind a barrier: nobody writes kind[] any more)
}

size_t dmt_sweep_workspace_bytes(int n, int R, int C)
{
    const size_t nV = (size_t)R * C, nT1 = 2 * (size_t)(R - 1) * (C - 1) + 1, nN = nV > nT1 ? nV : nT1;
    return (size_t)n * ((nV + nT1) * 4 + nN * 8 + 2 * dmt_edge_count(R, C) * sizeof(LvItem) + 16) + 256;
}

// ids (n, nE) sorted kept edges, m (n) their counts (device) -> kind (n, nE) u8, pers (n, nE) f32 (device).  Asynchronous on s.
int dmt_sweeps_dev(const float *field, const int32_t *ids, const int *m, int n, int R, int C, void *ws, uint8_t *kind, float *pers, hipStream_t s)
{
    if (n <= 0) return 0;
    const size_t nV = (size_t)R * C, nT1 = 2 * (size_t)(R - 1) * (C - 1) + 1, nN = nV > nT1 ? nV : nT1;
    const int nE = (int)dmt_edge_count(R, C);
    float *val = (float *)ws, *tv = val + (size_t)n * nV;
    int *par = (int *)(tv + (size_t)n * nT1);
    unsigned *first = (unsigned *)(par + (size_t)n * nN);
    LvItem *lists = (LvItem *)(first + (size_t)n * nN);
    unsigned *ctl = (unsigned *)(lists + (size_t)n * 2 * nE);
    const int blocks = (int)((nT1 + 255) / 256);
    hipLaunchKernelGGL(dmt_prep_kernel, dim3(blocks < 512 ? blocks : 512, n), dim3(256), 0, s, field, R, C, val, tv, ctl);
    // At most LV_CHUNK images per launch: the workgroups of an image wait for each other, so all LV_WG x images of a launch must be able
    // to be resident at once (workgroups are dispatched wg 0 of every image first: with more images than the chip has room for, the
    // first workgroups would fill it and wait for partners that can never start).  32 x 8 workgroups of 8 waves are an eighth of it.
    constexpr int LV_CHUNK = 32;
    for (int i0 = 0; i0 < n; i0 += LV_CHUNK) {
        const int k = n - i0 < LV_CHUNK ? n - i0 : LV_CHUNK;
        hipLaunchKernelGGL(dmt_levels_kernel, dim3(k, LV_WG), dim3(LV_THREADS), 0, s, ids + (size_t)i0 * nE, m + i0, nE, R, C, val + (size_t)i0 * nV,
                           tv + (size_t)i0 * nT1, par + (size_t)i0 * nN, first + (size_t)i0 * nN, lists + (size_t)i0 * 2 * nE, ctl + (size_t)i0 * 4,
                           kind + (size_t)i0 * nE, pers + (size_t)i0 * nE);
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

}  // namespace tmat
