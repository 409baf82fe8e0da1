// GPU binary morphology (morph_kernels.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
namespace tmat {
size_t morph_workspace_bytes(int k, int H, int W);
// device pointer to the k per-image "thinning converged" flags inside the workspace (1 = converged)
const int *morph_done_flags(void *workspace, int k, int H, int W);
// pred (k, H, W) f64 device -> filtered mask (k, H, W) u8 device, EDT of it (k, H, W) f64 device; all async on `s`
int filter_edt_dev(const double *pred, int k, int H, int W, int remove_isolated, void *workspace, uint8_t *filt_out,
                   double *dist_out, hipStream_t s);
int filter_mask_dev(const double *pred, const uint8_t *mask_in, int k, int H, int W, int use_median, int remove_isolated,
                    void *workspace, uint8_t *filt_out, double *dist_out, hipStream_t s);
void launch_ccl(const uint8_t *mask, int k, int H, int W, int *L, hipStream_t s);
void launch_edt(const uint8_t *mask, int k, int H, int W, int *g, int *st, int *any_zero, double *dist, hipStream_t s);
// finish_kernels.hip: EDT(~skel), centre-line weighting, anti-aliased resize, rescale to 0..255
size_t finish_workspace_bytes(int k, int H, int W, int oh, int ow);
int finish_dev(const double *pred, const double *dist, const uint8_t *skel, int k, int H, int W, int oh, int ow, void *workspace,
               float *field_out, float *f255_out, hipStream_t s);
void launch_rescale255(const float *field, int k, int npx, float *mn, float *mx, float *out, hipStream_t s);
// zproj_kernels.hip: Z projection of n stacks (n, Z, H, W) u16 device -> (n, H, W) u16 (fs / min / max) or f64 (avg / med)
int zproj_dev(const uint16_t *stacks, int n, int Z, int H, int W, int method, void *out, hipStream_t s);
}  // namespace tmat
