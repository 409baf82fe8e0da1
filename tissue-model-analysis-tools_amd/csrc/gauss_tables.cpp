// Gaussian kernel tables of scipy.ndimage.gaussian_filter1d (scipy/ndimage/_filters.py:_gaussian_kernel1d), the one piece of
// every gaussian / Sato / resize stage that is computed on the host: at most a few hundred doubles per sigma.
//
// scipy forms them with numpy.exp, whose float64 loop is CPU-dispatched (the AVX512 path differs from libm's exp in the last
// bit at some taps).  The tables made here use libm; a Python host that wants the reference's exact tables on its machine
// computes them with numpy and hands them over through tmat_set_gaussian_table (tmat_amd/sato.py does), after which every
// filter below is bit-identical to scipy on that host.
#include "../../include/tmat.h"
#include "gauss_tables.h"
#include "tmat_ctx.h"

#include <cfloat>
#include <cmath>

namespace tmat {

void gaussian_kernel1d(double sigma, int order, int radius, std::vector<double> &w)
{
    const int n = 2 * radius + 1;
    w.assign(n, 0.0);
    const double sigma2 = sigma * sigma;
    for (int x = -radius; x <= radius; x++) w[x + radius] = std::exp(-0.5 / sigma2 * (double)((long)x * x));
    const double tot = numpy_pairwise_sum(w.data(), n);
    for (double &v : w) v = v / tot;
    if (order == 0) return;
    // q(x) phi(x) with q from `order` applications of (D + P): for order 1 q = [0, -1/sigma2], for order 2 [-1/sigma2, 0, 1/sigma2^2]
    std::vector<double> q(order + 1, 0.0), nq(order + 1);
    q[0] = 1.0;
    for (int it = 0; it < order; it++) {
        for (int i = 0; i <= order; i++) {
            // row i of Q_deriv = D + P:  D[i][i+1] = i + 1,  P[i][i-1] = 1 / -sigma2;  numpy's dot sums j = 0 .. order
            double acc = 0.0;
            for (int j = 0; j <= order; j++) {
                double m = 0.0;
                if (j == i + 1) m = (double)(i + 1);
                if (j == i - 1) m = 1.0 / -sigma2;
                acc += m * q[j];
            }
            nq[i] = acc;
        }
        q = nq;
    }
    std::vector<double> k(n);
    for (int x = -radius; x <= radius; x++) {
        double acc = 0.0, xp = 1.0;
        for (int e = 0; e <= order; e++) { acc += xp * q[e]; xp *= (double)x; }
        k[x + radius] = acc * w[x + radius];
    }
    for (int i = 0; i < n; i++) w[i] = k[n - 1 - i];         // gaussian_filter1d reverses the kernel for correlate1d
}

int correlate_symmetry(const std::vector<double> &w)
{
    const int n = (int)w.size();
    if (!(n & 1)) return 0;
    const int c = n / 2;
    int sym = 1;
    for (int i = 1; i <= c; i++) if (std::fabs(w[c + i] - w[c - i]) > DBL_EPSILON) { sym = 0; break; }
    if (sym == 0) {
        sym = -1;
        for (int i = 1; i <= c; i++) if (std::fabs(w[c + i] + w[c - i]) > DBL_EPSILON) { sym = 0; break; }
    }
    return sym;
}

const GaussTable &gauss_table(Ctx *c, double sigma, int order, int radius)
{
    const GaussKey key{sigma, order, radius};
    auto it = c->gauss.find(key);
    if (it != c->gauss.end()) return it->second;
    GaussTable t;
    gaussian_kernel1d(sigma, order, radius, t.w);
    t.r = radius;
    t.sym = correlate_symmetry(t.w);
    return c->gauss.emplace(key, std::move(t)).first->second;
}

}  // namespace tmat

using namespace tmat;

extern "C" int tmat_set_gaussian_table(tmat_handle hd, double sigma, int order, int radius, const double *weights)
{
    Ctx *c = (Ctx *)hd;
    if (!c || !weights || !(sigma > 0) || order < 0 || radius < 0 || radius > 100000) { set_error("tmat_set_gaussian_table: bad argument"); return TMAT_E_ARG; }
    GaussTable t;
    t.w.assign(weights, weights + 2 * radius + 1);
    t.r = radius;
    t.sym = correlate_symmetry(t.w);
    GaussTable &slot = c->gauss[GaussKey{sigma, order, radius}];
    auto dev = c->gauss_dev.find(&slot);
    if (dev != c->gauss_dev.end()) {                 // a device copy of the table this one replaces
        hipSetDevice(c->device);
        hipStreamSynchronize(c->stream);
        hipFree(dev->second);
        c->gauss_dev.erase(dev);
    }
    slot = std::move(t);
    return TMAT_OK;
}
