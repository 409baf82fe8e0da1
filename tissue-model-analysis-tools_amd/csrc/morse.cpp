// tmat_morse_stats: topology.MorseGraph downstream of compute_dmt_graph, in C++.
//
// Reference: fl_tissue_model_tools/topology.py
//   graph build :530-539, smoothing :273-316 + :420-515, trimming :588-706, spanning forest
//   :541-581, branch labels :181-222, barcode :224-271, min-length filter :318-347,
//   statistics :54-65 / :349-356.
//
// The reference's output depends on CPython `set` iteration / pop order and on networkx's
// insertion-ordered adjacency (hard part 2).  This file reproduces both:
//   * PySet    -- CPython 3.10 setobject.c for small non-negative int keys (hash(k) == k):
//                 open addressing, 9 linear probes, perturb >> 5, grow x4 (x2 above 50000 used)
//                 when fill*5 >= mask*3, iteration in slot order, pop() with the search finger.
//   * OrdGraph -- networkx.Graph semantics: node order = first mention, per-node neighbour order =
//                 edge insertion order, copy() re-inserting edges node by node,
//                 remove_edges_from / remove_nodes_from / isolates / connected_components (_plain_bfs) /
//                 subgraph view iteration rule (FilterAtlas: set order if 2*|c| < |G| else node order).
// Float semantics follow numpy 1.26 (the reference's pin): float32 vertex positions and edge
// lengths, float64 accumulation, numpy pairwise summation in np.sum.
// This is host code by design (sequential, order-dependent graph logic on ~5k vertices); it is not
// a fallback for any GPU kernel.
#include "../../include/tmat.h"
#include "tmat_ctx.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace tmat {
namespace {

// ---------------------------------------------------------------------------------------------
struct PySet {
    std::vector<int32_t> key;
    std::vector<uint8_t> st;   // 0 unused, 1 active, 2 dummy
    size_t mask = 7, fill = 0, used = 0, finger = 0;
    PySet() : key(8, 0), st(8, 0) {}

    static void insert_clean(std::vector<int32_t> &k, std::vector<uint8_t> &s, size_t mask, int32_t v)
    {
        size_t perturb = (size_t)v, i = (size_t)v & mask;
        for (;;) {
            size_t e = i;
            int probes = (i + 9 <= mask) ? 9 : 0;
            do {
                if (s[e] == 0) { k[e] = v; s[e] = 1; return; }
                e++;
            } while (probes--);
            perturb >>= 5;
            i = (i * 5 + 1 + perturb) & mask;
        }
    }
    void resize(size_t minused)
    {
        size_t ns = 8;
        while (ns <= minused) ns <<= 1;
        std::vector<int32_t> nk(ns, 0);
        std::vector<uint8_t> nst(ns, 0);
        for (size_t e = 0; e <= mask; e++)
            if (st[e] == 1) insert_clean(nk, nst, ns - 1, key[e]);
        key.swap(nk); st.swap(nst);
        mask = ns - 1; fill = used;
    }
    bool contains(int32_t v) const
    {
        size_t perturb = (size_t)v, i = (size_t)v & mask;
        for (;;) {
            size_t e = i;
            int probes = (i + 9 <= mask) ? 9 : 0;
            do {
                if (st[e] == 0) return false;
                if (st[e] == 1 && key[e] == v) return true;
                e++;
            } while (probes--);
            perturb >>= 5;
            i = (i * 5 + 1 + perturb) & mask;
        }
    }
    void add(int32_t v)
    {
        size_t perturb = (size_t)v, i = (size_t)v & mask;
        long freeslot = -1;
        for (;;) {
            size_t e = i;
            int probes = (i + 9 <= mask) ? 9 : 0;
            do {
                if (st[e] == 0) {
                    if (freeslot >= 0) { key[freeslot] = v; st[freeslot] = 1; used++; return; }
                    key[e] = v; st[e] = 1; fill++; used++;
                    if (fill * 5 >= mask * 3) resize(used > 50000 ? used * 2 : used * 4);
                    return;
                }
                if (st[e] == 1 && key[e] == v) return;
                if (st[e] == 2) freeslot = (long)e;
                e++;
            } while (probes--);
            perturb >>= 5;
            i = (i * 5 + 1 + perturb) & mask;
        }
    }
    bool discard(int32_t v)
    {
        size_t perturb = (size_t)v, i = (size_t)v & mask;
        for (;;) {
            size_t e = i;
            int probes = (i + 9 <= mask) ? 9 : 0;
            do {
                if (st[e] == 0) return false;
                if (st[e] == 1 && key[e] == v) { st[e] = 2; used--; return true; }
                e++;
            } while (probes--);
            perturb >>= 5;
            i = (i * 5 + 1 + perturb) & mask;
        }
    }
    int32_t pop()
    {
        size_t e = finger & mask;
        while (st[e] != 1) { e++; if (e > mask) e = 0; }
        st[e] = 2; used--;
        finger = e + 1;
        return key[e];
    }
    template <class F> void for_each(F f) const
    {
        for (size_t e = 0; e <= mask; e++) if (st[e] == 1) f(key[e]);
    }
    std::vector<int32_t> items() const
    {
        std::vector<int32_t> r; r.reserve(used);
        for_each([&](int32_t v) { r.push_back(v); });
        return r;
    }
};

// ---------------------------------------------------------------------------------------------
struct OrdGraph {
    // node ids are 0..n-1 (DMT vertex indices); `order` = dict insertion order of _node
    std::vector<int32_t> order;
    std::vector<uint8_t> alive;
    std::vector<std::vector<int32_t>> adj;
    size_t n_alive = 0;
    explicit OrdGraph(size_t n) : alive(n, 0), adj(n) {}

    void add_node(int32_t u) { if (!alive[u]) { alive[u] = 1; order.push_back(u); n_alive++; } }
    static bool has(const std::vector<int32_t> &v, int32_t x) { return std::find(v.begin(), v.end(), x) != v.end(); }
    void add_edge(int32_t u, int32_t v)
    {
        add_node(u); add_node(v);
        if (!has(adj[u], v)) adj[u].push_back(v);
        if (!has(adj[v], u)) adj[v].push_back(u);
    }
    void remove_edge(int32_t u, int32_t v)
    {
        if (!alive[u]) return;
        auto it = std::find(adj[u].begin(), adj[u].end(), v);
        if (it == adj[u].end()) return;
        adj[u].erase(it);
        if (u != v) { auto jt = std::find(adj[v].begin(), adj[v].end(), u); if (jt != adj[v].end()) adj[v].erase(jt); }
    }
    void remove_node(int32_t n)
    {
        if (!alive[n]) return;
        alive[n] = 0; n_alive--;
        for (int32_t u : adj[n]) { auto jt = std::find(adj[u].begin(), adj[u].end(), n); if (jt != adj[u].end()) adj[u].erase(jt); }
        adj[n].clear();
    }
    size_t degree(int32_t n) const { return adj[n].size(); }
    template <class F> void for_nodes(F f) const { for (int32_t u : order) if (alive[u]) f(u); }
    void compact() { std::vector<int32_t> o; for (int32_t u : order) if (alive[u]) o.push_back(u); order.swap(o); }
    // networkx Graph.copy(): nodes in order, then edges re-added node by node
    OrdGraph copy() const
    {
        OrdGraph g(alive.size());
        for_nodes([&](int32_t u) { g.add_node(u); });
        for_nodes([&](int32_t u) { for (int32_t v : adj[u]) g.add_edge(u, v); });
        return g;
    }
    void remove_isolates()
    {
        std::vector<int32_t> iso;
        for_nodes([&](int32_t u) { if (adj[u].empty()) iso.push_back(u); });
        for (int32_t u : iso) remove_node(u);
    }
};

// ---------------------------------------------------------------------------------------------
struct V2f { float x, y; };
struct V2d { double x, y; };

inline double edge_len32(const std::vector<V2f> &v, int32_t a, int32_t b)
{
    float dx = v[a].x - v[b].x, dy = v[a].y - v[b].y;
    float s = dx * dx;
    float t = dy * dy;
    return (double)std::sqrt(s + t);
}

// topology.py:479-515
static std::vector<V2d> resample_uniform(const std::vector<V2d> &p, int n)
{
    const int m = (int)p.size();
    std::vector<double> d(m - 1), acc(m);
    for (int i = 0; i + 1 < m; i++) {
        double dx = p[i + 1].x - p[i].x, dy = p[i + 1].y - p[i].y;
        double sx = dx * dx, sy = dy * dy;
        d[i] = std::sqrt(sx + sy);
    }
    const double total = numpy_pairwise_sum(d.data(), m - 1);
    acc[0] = 0.0;
    for (int i = 1; i < m; i++) acc[i] = acc[i - 1] + d[i - 1];
    const double step = total / (double)(n - 1);
    std::vector<V2d> out;
    out.reserve(n);
    out.push_back(p[0]);
    for (int i = 1; i < n - 1; i++) {
        const double s = (double)i * step;
        // np.searchsorted(acc, s, side="right") - 1
        int k = (int)(std::upper_bound(acc.begin(), acc.end(), s) - acc.begin()) - 1;
        if (k < 0) k = 0;
        if (k > m - 2) k = m - 2;          // the reference would raise IndexError here; unreachable for sane input
        const double t = (s - acc[k]) / (acc[k + 1] - acc[k]);
        V2d q;
        q.x = p[k].x + (p[k + 1].x - p[k].x) * t;
        q.y = p[k].y + (p[k + 1].y - p[k].y) * t;
        out.push_back(q);
    }
    out.push_back(p[m - 1]);
    return out;
}

// topology.py:421-476; returns false when the input is returned unchanged (n == 1)
static bool moving_average_fixed_ends(const std::vector<V2f> &A, int n, std::vector<V2d> &out)
{
    const int L = (int)A.size();
    n = std::min(n, (L + 1) / 2);
    if (n <= 1) return false;
    // A_transformed = [A[0]]*n, [A[1]]*(n-1), ..., [A[n-2]]*2, A[n-1 : L-(n-1)], [A[-(n-1)]]*2, ..., [A[-1]]*n
    std::vector<V2d> T;
    for (int i = 0; i <= n - 2; i++)
        for (int r = 0; r < n - i; r++) T.push_back({(double)A[i].x, (double)A[i].y});
    for (int i = n - 1; i < L - (n - 1); i++) T.push_back({(double)A[i].x, (double)A[i].y});
    for (int i = n - 2; i >= 0; i--)
        for (int r = 0; r < n - i; r++) T.push_back({(double)A[L - 1 - i].x, (double)A[L - 1 - i].y});
    const int M = (int)T.size();
    std::vector<V2d> cs(M);
    cs[0] = T[0];
    for (int i = 1; i < M; i++) { cs[i].x = cs[i - 1].x + T[i].x; cs[i].y = cs[i - 1].y + T[i].y; }
    std::vector<V2d> ma(M - n + 1);
    for (int i = n - 1; i < M; i++) {
        V2d v = cs[i];
        if (i >= n) { v.x = cs[i].x - cs[i - n].x; v.y = cs[i].y - cs[i - n].y; }
        ma[i - (n - 1)] = {v.x / (double)n, v.y / (double)n};
    }
    out = resample_uniform(ma, L);
    return true;
}

// topology.py:273-316
static void smooth_vertices(const OrdGraph &G, std::vector<V2f> &verts, int window)
{
    if (window <= 1) return;
    PySet fixed;
    G.for_nodes([&](int32_t v) { if (G.degree(v) != 2) fixed.add(v); });
    std::vector<uint8_t> visited(verts.size(), 0);
    std::vector<uint32_t> seen_stamp(verts.size(), 0);
    uint32_t stamp = 0;
    std::vector<int32_t> chain;
    std::vector<V2f> pos;
    std::vector<V2d> sm;
    for (int32_t start : fixed.items()) {
        const std::vector<int32_t> nb0 = G.adj[start];     // neighbors(start) (adjacency is not mutated here)
        for (int32_t base : nb0) {
            int32_t cur = base;
            if (visited[cur]) continue;
            chain.clear(); chain.push_back(start); chain.push_back(cur);
            stamp++;
            while (G.degree(cur) == 2) {
                const auto &nb = G.adj[cur];
                int32_t nxt = nb[0] != cur ? nb[0] : nb[1];
                if (seen_stamp[nxt] == stamp) break;
                cur = nxt;
                seen_stamp[cur] = stamp;
                chain.push_back(cur);
            }
            pos.resize(chain.size());
            for (size_t i = 0; i < chain.size(); i++) pos[i] = verts[chain[i]];
            if (moving_average_fixed_ends(pos, window, sm))
                for (size_t i = 0; i < chain.size(); i++) verts[chain[i]] = {(float)sm[i].x, (float)sm[i].y};
            visited[chain.front()] = 1; visited[chain.back()] = 1;
        }
    }
}

static float median_f32(std::vector<float> v)
{
    std::sort(v.begin(), v.end());
    size_t k = v.size();
    if (k & 1) return v[k / 2];
    float s = v[k / 2 - 1] + v[k / 2];
    return s / 2.0f;
}

// topology.py:588-706
static OrdGraph trim_graph(const OrdGraph &G0, const std::vector<V2f> &verts, int rows, int cols, int min_len, int max_len,
                           const uint8_t *pruning_mask, bool remove_isolated)
{
    OrdGraph G = G0.copy();
    auto bbox_diag = [&](const std::vector<int32_t> &seg) -> float {
        float lx = verts[seg[0]].x, hx = lx, ly = verts[seg[0]].y, hy = ly;
        for (int32_t v : seg) {
            lx = std::min(lx, verts[v].x); hx = std::max(hx, verts[v].x);
            ly = std::min(ly, verts[v].y); hy = std::max(hy, verts[v].y);
        }
        float dx = hx - lx, dy = hy - ly;
        float sx = dx * dx, sy = dy * dy;
        return std::sqrt(sx + sy);
    };
    int pass_num = 1;
    bool done = false;
    std::vector<uint8_t> unmarked(verts.size());
    while (!done) {
        PySet junctions, leaves;
        G.for_nodes([&](int32_t n) { if (G.degree(n) > 2) junctions.add(n); });
        if (pass_num == 1) G.for_nodes([&](int32_t n) { if (G.degree(n) == 1) leaves.add(n); });
        std::fill(unmarked.begin(), unmarked.end(), 0);
        G.for_nodes([&](int32_t n) { if (!junctions.contains(n)) unmarked[n] = 1; });
        PySet &bases = pass_num == 1 ? leaves : junctions;
        std::vector<std::vector<int32_t>> keep, shortv, longv, isolated;
        while (bases.used) {
            const int32_t s0 = bases.pop();
            PySet nbrs;
            for (int32_t n : G.adj[s0]) if (unmarked[n]) nbrs.add(n);
            while (nbrs.used) {
                int32_t node = nbrs.pop();
                std::vector<int32_t> seg;
                if (pass_num == 1) seg.push_back(s0);
                seg.push_back(node);
                for (;;) {
                    int32_t nxt = -1;
                    for (int32_t n : G.adj[node]) if (unmarked[n]) { nxt = n; break; }
                    if (nxt < 0) break;
                    node = nxt;
                    seg.push_back(node);
                    unmarked[node] = 0;
                }
                const int n_leaf = (G.degree(seg.front()) == 1) + (G.degree(seg.back()) == 1);
                bool any_junction = false;
                for (int32_t v : seg) if (G.degree(v) > 2) { any_junction = true; break; }
                if (remove_isolated && n_leaf == 2 && !any_junction) isolated.push_back(seg);
                else if (n_leaf > 0) {
                    const float len = bbox_diag(seg);
                    if (len < (float)min_len) shortv.push_back(seg);
                    else if (max_len > 0 && len > (float)max_len) longv.push_back(seg);
                    else keep.push_back(seg);
                } else keep.push_back(seg);
            }
        }
        std::vector<std::vector<int32_t>> doomed;
        if (pruning_mask)
            for (auto &seg : keep) {
                std::vector<float> xs, ys;
                for (int32_t v : seg) { xs.push_back(verts[v].x); ys.push_back(verts[v].y); }
                long r = (long)std::nearbyint(median_f32(xs)), c = (long)std::nearbyint(median_f32(ys));
                if (r < 0) r += rows;               // numpy negative index wrap
                if (c < 0) c += cols;
                if (r >= 0 && r < rows && c >= 0 && c < cols && pruning_mask[(size_t)r * cols + c]) doomed.push_back(seg);
            }
        for (auto &s : shortv) doomed.push_back(s);
        for (auto &s : longv) doomed.push_back(s);
        for (auto &s : isolated) doomed.push_back(s);
        for (auto &seg : doomed)
            for (int32_t v : seg) G.remove_node(v);   // remove_edges_from(edges(seg)) + remove_nodes_from(seg)
        G.remove_isolates();
        done = pass_num == 2 && doomed.empty();
        pass_num = pass_num == 1 ? 2 : 1;
    }
    G.compact();
    return G;
}

}  // namespace
}  // namespace tmat

using namespace tmat;

extern "C" int tmat_morse_stats(const int32_t *verts_in, int n_verts, const int32_t *edges, int n_edges, int rows, int cols,
                                int smoothing_window, int min_branch_length, int max_branch_length,
                                int remove_isolated_branches, const uint8_t *pruning_mask, int64_t *count,
                                double *total_px, double *avg_px, double *bars, int cap)
{
    if (n_verts < 0 || n_edges < 0 || (n_verts && !verts_in) || (n_edges && !edges) || !count || !total_px || !avg_px) {
        set_error("tmat_morse_stats: bad argument");
        return TMAT_E_ARG;
    }
    for (int i = 0; i < 2 * n_edges; i++)
        if (edges[i] < 0 || edges[i] >= n_verts) { set_error("tmat_morse_stats: edge index out of range"); return TMAT_E_ARG; }
    std::vector<V2f> verts(n_verts);
    for (int i = 0; i < n_verts; i++) verts[i] = {(float)verts_in[2 * i], (float)verts_in[2 * i + 1]};
    OrdGraph G((size_t)n_verts);
    for (int i = 0; i < n_edges; i++) G.add_edge(edges[2 * i], edges[2 * i + 1]);

    smooth_vertices(G, verts, smoothing_window);
    OrdGraph T = trim_graph(G, verts, rows, cols, min_branch_length, max_branch_length, pruning_mask,
                            remove_isolated_branches != 0);

    // ---- spanning forest (:541-581): connected_components (_plain_bfs) -> subgraph views -> BFS from the
    //      first max-degree node in view order ----
    const size_t N = (size_t)n_verts;
    std::vector<int32_t> parent(N, -1);
    std::vector<double> dist_root(N, 0.0);
    OrdGraph forest(N);
    {
        std::vector<uint8_t> seen_all(N, 0);
        std::vector<PySet> comps;
        const size_t n_total = T.n_alive;
        size_t n_seen = 0;
        T.for_nodes([&](int32_t v) {
            if (seen_all[v]) return;
            // _plain_bfs(G, n - len(seen), v)
            const size_t target = n_total - n_seen;
            PySet seen;
            seen.add(v);
            std::vector<int32_t> next{v};
            bool full = false;
            while (!next.empty() && !full) {
                std::vector<int32_t> cur;
                cur.swap(next);
                for (int32_t a : cur) {
                    for (int32_t w : T.adj[a]) if (!seen.contains(w)) { seen.add(w); next.push_back(w); }
                    if (seen.used == target) { full = true; break; }
                }
            }
            seen.for_each([&](int32_t k) { seen_all[k] = 1; });
            n_seen += seen.used;
            comps.push_back(std::move(seen));
        });
        for (const PySet &c : comps) {
            // show_nodes(nbunch_iter(c)).nodes = set(<generator over c>)
            PySet B;
            c.for_each([&](int32_t k) { B.add(k); });
            std::vector<int32_t> view;
            if (2 * B.used < n_total) view = B.items();
            else T.for_nodes([&](int32_t u) { if (B.contains(u)) view.push_back(u); });
            int32_t root = -1;
            size_t maxdeg = 0;
            for (int32_t u : view) if (root < 0 || T.degree(u) > maxdeg) { root = u; maxdeg = T.degree(u); }
            if (root < 0) continue;
            if (remove_isolated_branches && maxdeg <= 2) continue;
            parent[root] = root;
            dist_root[root] = 0.0;
            std::vector<int32_t> queue{root};
            for (size_t qh = 0; qh < queue.size(); qh++) {
                const int32_t v = queue[qh];
                for (int32_t n : T.adj[v])
                    if (parent[n] < 0) {
                        forest.add_edge(v, n);
                        parent[n] = v;
                        dist_root[n] = dist_root[v] + edge_len32(verts, v, n);
                        queue.push_back(n);
                    }
            }
        }
    }

    // ---- branch labels (:181-222) ----
    std::vector<int32_t> leaves;
    forest.for_nodes([&](int32_t n) { if (forest.degree(n) == 1) leaves.push_back(n); });
    std::vector<double> far(N, -std::numeric_limits<double>::infinity());
    std::vector<int32_t> label(N, -1);
    for (int32_t leaf : leaves) {
        int32_t cur = leaf, par = parent[leaf];
        double d = 0.0;
        far[leaf] = 0.0;
        label[leaf] = leaf;
        while (par != cur) {
            d += edge_len32(verts, par, cur);
            if (d < far[par]) break;
            cur = par; par = parent[cur];
            far[cur] = d;
            label[cur] = leaf;
        }
    }
    // ---- barcode (:224-271) + min-length filter (:318-347) ----
    std::vector<double> births, deaths;
    for (int32_t leaf : leaves) {
        int32_t cur = leaf, lab = leaf, par = parent[leaf];
        double d = 0.0;
        while (lab == leaf && cur != par) {
            d += edge_len32(verts, par, cur);
            cur = par; par = parent[cur];
            lab = label[cur];
        }
        const double birth = -dist_root[leaf];
        const double death = birth + d;
        if (death - birth >= (double)min_branch_length) { births.push_back(birth); deaths.push_back(death); }
    }
    // ---- statistics (:54-65, :349-356) ----
    std::vector<double> lens;
    for (size_t i = 0; i < births.size(); i++) { double l = deaths[i] - births[i]; if (!std::isinf(l)) lens.push_back(l); }
    const double total = lens.empty() ? 0.0 : numpy_pairwise_sum(lens.data(), (long)lens.size());
    *count = (int64_t)births.size();
    *total_px = total;
    *avg_px = total == 0.0 ? 0.0 : total / (double)lens.size();
    if (bars) {
        if ((size_t)cap < births.size()) { set_error("tmat_morse_stats: bars capacity too small"); return TMAT_E_CAP; }
        for (size_t i = 0; i < births.size(); i++) { bars[2 * i] = births[i]; bars[2 * i + 1] = deaths[i]; }
    }
    return TMAT_OK;
}
