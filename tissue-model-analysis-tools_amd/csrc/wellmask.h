// Well detection (--detect-well): device stages of fl_tissue_model_tools/well_mask_generation.py (wellmask_kernels.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
namespace tmat {
// rescale_intensity(blur, out_range=(0, 255)).astype(uint8) of a float32 image whose extrema are mn[0], mx[0] (device)
void launch_wm_rescale_u8(const float *blur, size_t n, const float *mn, const float *mx, uint8_t *out, hipStream_t s);
// the same for a float64 image; mm: 2 doubles of device scratch (extrema are computed here)
void launch_wm_rescale_u8_f64(const double *blur, size_t n, double *mm, uint8_t *out, hipStream_t s);
// hist: 5 x 256 counts (whole image, then the four 5 % corners: top-left, top-right, bottom-left, bottom-right); zeroed here
void launch_wm_hist(const uint8_t *img, int H, int W, unsigned *hist, hipStream_t s);
// decision[0] = Otsu threshold of the (possibly inverted) image, decision[1] = invert flag (auto_threshold_well :244-273)
void launch_wm_decide(const unsigned *hist, int H, int W, int *decision, hipStream_t s);
void launch_wm_threshold(const uint8_t *img, size_t n, const int *decision, uint8_t *out, hipStream_t s);
// binary erosion with the offsets `off` (noff pairs dy, dx), pixels outside the image count as set (border_value=True)
void launch_wm_erode(const uint8_t *m, int H, int W, const int *off, int noff, uint8_t *out, hipStream_t s);
// canny's smoothing of a 0 / 1 mask with mode="constant": pad (zeros, and a second padded image of ones inside), crop + divide
void launch_wm_pad(const uint8_t *m, int H, int W, int r, double *img_pad, double *ones_pad, hipStream_t s);
void launch_wm_crop_div(const double *img_pad, const double *ones_pad, int H, int W, int r, double *out, hipStream_t s);
}  // namespace tmat
