// scipy.ndimage gaussian kernel tables (host): what scipy's gaussian_filter1d hands to correlate1d.
#pragma once
#include <vector>
namespace tmat {
struct Ctx;
struct GaussTable {
    std::vector<double> w;      // 2 r + 1 weights in correlate1d order (already reversed as gaussian_filter1d does)
    int r = 0;
    int sym = 0;                // correlate1d's classification: 1 symmetric, -1 antisymmetric, 0 neither
};
// scipy.ndimage._filters._gaussian_kernel1d(sigma, order, radius)[::-1] with libm's exp and numpy's pairwise sum
void gaussian_kernel1d(double sigma, int order, int radius, std::vector<double> &w);
// correlate1d's symmetry test on a weight vector (ni_filters.c: |w[c+i] -/+ w[c-i]| <= DBL_EPSILON)
int correlate_symmetry(const std::vector<double> &w);
// the table for (sigma, order, radius): a host-supplied one (tmat_set_gaussian_table) when the handle holds it, else computed here
const GaussTable &gauss_table(Ctx *c, double sigma, int order, int radius);
inline int gauss_radius(double sigma, double truncate) { return (int)(truncate * sigma + 0.5); }
}  // namespace tmat
