// tmat_dmt_graph: Discrete-Morse-Theory graph of a 2-D scalar field.
//
// Reference: fl_tissue_model_tools/dmtgraph.py:38-99 (compute_dmt_graph) and helpers :102-453 --
// the numba port of the pydmtgraph C++ extension that the reference no longer vendors.
//
// Structure here (not the reference's row-record arrays): simplices are never materialised.
// An edge is its id in the canonical enumeration (vertical, then horizontal, then anti-diagonal),
// endpoints / dual triangles are decoded from the id; the lower-star order is a stable LSD radix
// sort of order-preserving uint32 images of the float32 keys (stability == the reference's
// tie-break by filtered-edge index); both persistence sweeps are union-find with path halving.
// The sweeps and `collect` are inherently sequential per image (Kruskal order); batches run them
// on host threads, one image per thread, while the GPU segments the next batch (tmat_analyze_*).
#include "../../include/tmat.h"
#include "tmat_internal.h"
#include "postproc.h"

#include <cmath>
#include <cstring>
#include <limits>

namespace tmat {

struct Grid {
    int R, C;
    int nVert, nHor, nDiag;   // edge-class sizes
    Grid(int r, int c) : R(r), C(c), nVert((r - 1) * c), nHor(r * (c - 1)), nDiag((r - 1) * (c - 1)) {}
    int n_edges() const { return nVert + nHor + nDiag; }
    int n_tri() const { return 2 * nDiag; }          // outer face gets index n_tri()
    inline void endpoints(int e, int &a, int &b) const
    {
        if (e < nVert) { a = e; b = e + C; return; }                       // (r,c)-(r+1,c): e = r*C + c
        e -= nVert;
        if (e < nHor) { int r = e / (C - 1), c = e - r * (C - 1); a = r * C + c; b = a + 1; return; }
        e -= nHor;
        int r = e / (C - 1), c = e - r * (C - 1);
        a = r * C + c + 1; b = a + C - 1;                                  // (r,c+1)-(r+1,c)
    }
    inline void faces(int e, int &f, int &g) const
    {
        const int outer = n_tri();
        if (e < nVert) {
            int r = e / C, c = e - r * C, t = 2 * (r * (C - 1) + c);
            f = c == 0 ? outer : t - 1;
            g = c == C - 1 ? outer : t;
            return;
        }
        e -= nVert;
        if (e < nHor) {
            int r = e / (C - 1), c = e - r * (C - 1), t = 2 * (r * (C - 1) + c);
            f = r == 0 ? outer : t - 2 * (C - 1) + 1;
            g = r == R - 1 ? outer : t;
            return;
        }
        e -= nHor;
        f = 2 * e; g = f + 1;
    }
};

static inline uint32_t f32_sort_key(float v)
{
    uint32_t u;
    memcpy(&u, &v, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

static inline int uf_find(int32_t *p, int v)
{
    while (p[v] != v) { p[v] = p[p[v]]; v = p[v]; }
    return v;
}

// lower-star order of the kept edges on the host (dmtgraph.py:71-93): sorted edge ids
static void sorted_edges_host(const float *val, const Grid &gd, std::vector<int32_t> &out)
{
    const int nV = gd.R * gd.C, nE = gd.n_edges();
    std::vector<uint8_t> live(nV);
    for (int i = 0; i < nV; i++) live[i] = !(std::fabs((double)val[i]) <= 1e-8);
    // filtered edge list in canonical order with keys
    std::vector<int32_t> eid; eid.reserve(nE);
    std::vector<uint32_t> key; key.reserve(nE);
    for (int e = 0; e < nE; e++) {
        int a, b; gd.endpoints(e, a, b);
        if (live[a] && live[b]) { eid.push_back(e); key.push_back(f32_sort_key(val[a] > val[b] ? val[a] : val[b])); }
    }
    const int m = (int)eid.size();
    // stable LSD radix sort (4 x 8 bit) of (key, position)
    std::vector<int32_t> ord(m), tmp(m);
    for (int i = 0; i < m; i++) ord[i] = i;
    for (int pass = 0; pass < 4; pass++) {
        size_t cnt[257] = {0};
        const int sh = 8 * pass;
        for (int i = 0; i < m; i++) cnt[((key[ord[i]] >> sh) & 255) + 1]++;
        for (int i = 0; i < 256; i++) cnt[i + 1] += cnt[i];
        for (int i = 0; i < m; i++) tmp[cnt[(key[ord[i]] >> sh) & 255]++] = ord[i];
        ord.swap(tmp);
    }
    out.resize(m);
    for (int i = 0; i < m; i++) out[i] = eid[ord[i]];
}

// `sorted` / m: the kept edges in lower-star order when the device front end (csrc/dmt_kernels.hip) produced them; NULL =
// filter and sort here
// kind_in / pers_in (nullable, m_in entries): the result of the two sweeps when the device ran them (dmt_sweep_kernels.hip)
int dmt_graph_host_sorted(const float *img, int R, int C, float delta1, float delta2, const int32_t *sorted, int m_in, int32_t *verts,
                          int cap_v, int32_t *edges, int cap_e, int *n_verts, int *n_edges, const uint8_t *kind_in, const float *pers_in)
{
    *n_verts = 0; *n_edges = 0;
    if (R < 1 || C < 1) { set_error("tmat_dmt_graph: empty image"); return TMAT_E_ARG; }
    const Grid gd(R, C);
    const int nV = R * C, nT = gd.n_tri();
    std::vector<float> val(nV);
    for (int i = 0; i < nV; i++) val[i] = -img[i];
    std::vector<int32_t> own;
    if (!sorted) { sorted_edges_host(val.data(), gd, own); sorted = own.data(); m_in = (int)own.size(); }
    const int m = m_in;
    const int32_t *se = sorted;
    // sorted edges: endpoints, key value, pairing state
    std::vector<int32_t> ea(m), eb(m);
    std::vector<float> ev(m), pers(m, std::numeric_limits<float>::infinity());
    std::vector<uint8_t> kind(m, 0);      // 0 unpaired, 1 vertex-edge, 2 edge-triangle
    for (int i = 0; i < m; i++) {
        int a, b; gd.endpoints(se[i], a, b);
        ea[i] = a; eb[i] = b; ev[i] = val[a] > val[b] ? val[a] : val[b];
    }
    // the device kernel marks a sweep whose levels stopped making progress -- which the formulation excludes -- with kind[0] = 0xFF:
    // that is an error, not something to paper over (TMAT_DMT_SWEEP_DEVICE=0 runs the sweeps below instead)
    if (kind_in && pers_in && m > 0 && kind_in[0] == 0xFF) {
        set_error("tmat_dmt_graph: the device persistence sweeps made no progress (dmt_sweep_kernels.hip); TMAT_DMT_SWEEP_DEVICE=0 runs them on the host");
        return TMAT_E_HIP;
    }
    if (kind_in && pers_in) {
        for (int i = 0; i < m; i++) { kind[i] = kind_in[i]; pers[i] = pers_in[i]; }
    } else {
    // ---- ascending sweep: elder rule on vertices (younger = larger value, ties by larger index, dies) ----
    {
        std::vector<int32_t> p(nV);
        for (int i = 0; i < nV; i++) p[i] = i;
        for (int i = 0; i < m; i++) {
            int x = uf_find(p.data(), ea[i]), y = uf_find(p.data(), eb[i]);
            if (x == y) continue;
            const bool x_older = val[x] < val[y] || (val[x] == val[y] && x < y);
            const int dead = x_older ? y : x, keep = x_older ? x : y;
            p[dead] = keep;
            pers[i] = ev[i] - val[dead];
            kind[i] = 1;
        }
    }
    // ---- descending sweep on the dual graph; the outer face (+inf, largest index) always survives ----
    {
        std::vector<float> tv(nT + 1);
        for (int r = 0; r < R - 1; r++)
            for (int c = 0; c < C - 1; c++) {
                const float a = val[r * C + c], b = val[r * C + c + 1], d = val[(r + 1) * C + c], e = val[(r + 1) * C + c + 1];
                const int t = 2 * (r * (C - 1) + c);
                tv[t] = std::fmax(std::fmax(a, b), d);
                tv[t + 1] = std::fmax(std::fmax(b, d), e);
            }
        tv[nT] = std::numeric_limits<float>::infinity();
        std::vector<int32_t> p(nT + 1);
        for (int i = 0; i <= nT; i++) p[i] = i;
        for (int i = m - 1; i >= 0; i--) {
            if (kind[i]) continue;
            int f, g; gd.faces(se[i], f, g);
            int x = uf_find(p.data(), f), y = uf_find(p.data(), g);
            if (x == y) continue;
            const bool x_wins = tv[x] > tv[y] || (tv[x] == tv[y] && x > y);
            const int dead = x_wins ? y : x, keep = x_wins ? x : y;
            p[dead] = keep;
            pers[i] = tv[dead] - ev[i];
            kind[i] = 2;
        }
    }
    }
    // ---- collect: low-persistence tree edges, at most 4 links per vertex, in descending edge order ----
    std::vector<int32_t> link(4 * (size_t)nV, -1);
    auto add_link = [&](int a, int b) {
        int32_t *s = &link[4 * (size_t)a];
        for (int k = 0; k < 4; k++) if (s[k] < 0) { s[k] = b; return; }
    };
    for (int i = m - 1; i >= 0; i--)
        if (kind[i] == 1 && pers[i] < delta1) { add_link(ea[i], eb[i]); add_link(eb[i], ea[i]); }
    // Morse cancellation: each link-component is re-rooted at its minimum (value, index)
    std::vector<int32_t> up(nV, -1), mark(nV, -1), q;
    q.reserve(1024);
    for (int s = 0; s < nV; s++) {
        if (up[s] != -1) continue;
        q.clear(); q.push_back(s);
        int best = s;
        for (size_t h = 0; h < q.size(); h++) {
            const int cur = q[h];
            mark[cur] = s;
            if (val[cur] < val[best] || (val[cur] == val[best] && cur < best)) best = cur;
            const int32_t *nb = &link[4 * (size_t)cur];
            for (int k = 0; k < 4 && nb[k] >= 0; k++) if (mark[nb[k]] != s) q.push_back(nb[k]);
        }
        up[best] = best;
        q.clear(); q.push_back(best);
        for (size_t h = 0; h < q.size(); h++) {
            const int cur = q[h];
            const int32_t *nb = &link[4 * (size_t)cur];
            for (int k = 0; k < 4 && nb[k] >= 0; k++) if (up[nb[k]] == -1) { up[nb[k]] = cur; q.push_back(nb[k]); }
        }
    }
    // unstable 1-manifolds of the significant saddles
    std::vector<uint8_t> onpath(nV, 0);
    std::vector<int32_t> newid(nV, -1);
    int nv = 0, ne = 0;
    auto vid = [&](int v) -> int {
        if (newid[v] < 0) {
            if (nv >= cap_v) return -1;
            newid[v] = nv; verts[2 * nv] = v / C; verts[2 * nv + 1] = v % C; nv++;
        }
        return newid[v];
    };
    auto emit = [&](int a, int b) -> bool {
        int ia = vid(a); if (ia < 0) return false;
        int ib = vid(b); if (ib < 0) return false;
        if (ne >= cap_e) return false;
        edges[2 * ne] = ia; edges[2 * ne + 1] = ib; ne++;
        return true;
    };
    for (int i = m - 1; i >= 0; i--) {
        if (!(pers[i] > delta1 && ev[i] < -delta2)) continue;
        const int ends[2] = {ea[i], eb[i]};
        for (int s = 0; s < 2; s++) {
            int cur = ends[s];
            while (!onpath[cur] && up[cur] != cur && up[cur] != -1) {
                onpath[cur] = 1;
                if (!emit(cur, up[cur])) { set_error("tmat_dmt_graph: output capacity too small"); return TMAT_E_CAP; }
                cur = up[cur];
            }
        }
        if (!emit(ea[i], eb[i])) { set_error("tmat_dmt_graph: output capacity too small"); return TMAT_E_CAP; }
    }
    *n_verts = nv; *n_edges = ne;
    return TMAT_OK;
}

int dmt_graph_host(const float *img, int R, int C, float delta1, float delta2, int32_t *verts, int cap_v, int32_t *edges,
                   int cap_e, int *n_verts, int *n_edges)
{
    return dmt_graph_host_sorted(img, R, C, delta1, delta2, nullptr, 0, verts, cap_v, edges, cap_e, n_verts, n_edges, nullptr, nullptr);
}

}  // namespace tmat

extern "C" int tmat_dmt_graph(tmat_handle hd, const float *img, int rows, int cols, float delta1, float delta2, int32_t *verts,
                              int cap_v, int32_t *edges, int cap_e, int *n_verts, int *n_edges)
{
    if (!img || !verts || !edges || !n_verts || !n_edges || cap_v < 0 || cap_e < 0) {
        tmat::set_error("tmat_dmt_graph: bad argument");
        return TMAT_E_ARG;
    }
    if (!hd || rows < 2 || cols < 2)      // no handle: host-only execution (key build and sort included)
        return tmat::dmt_graph_host(img, rows, cols, delta1, delta2, verts, cap_v, edges, cap_e, n_verts, n_edges);
    return tmat::dmt_graph_device_front(hd, img, rows, cols, delta1, delta2, verts, cap_v, edges, cap_e, n_verts, n_edges);
}

extern "C" int tmat_dmt_graph_batch(tmat_handle hd, const float *imgs, int n, int rows, int cols, float delta1, float delta2, int32_t *verts,
                                    int cap_v, int32_t *edges, int cap_e, int *n_verts, int *n_edges)
{
    if (!imgs || !verts || !edges || !n_verts || !n_edges || n < 0 || cap_v < 0 || cap_e < 0) {
        tmat::set_error("tmat_dmt_graph_batch: bad argument");
        return TMAT_E_ARG;
    }
    if (n == 0) return TMAT_OK;
    if (hd && rows >= 2 && cols >= 2)
        return tmat::dmt_graph_device_batch(hd, imgs, n, rows, cols, delta1, delta2, verts, cap_v, edges, cap_e, n_verts, n_edges);
    for (int i = 0; i < n; i++) {
        const int rc = tmat::dmt_graph_host(imgs + (size_t)i * rows * cols, rows, cols, delta1, delta2, verts + (size_t)i * 2 * cap_v, cap_v,
                                            edges + (size_t)i * 2 * cap_e, cap_e, n_verts + i, n_edges + i);
        if (rc) return rc;
    }
    return TMAT_OK;
}
