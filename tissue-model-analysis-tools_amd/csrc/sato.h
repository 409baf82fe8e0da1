// Z-stack (Sato) branch: device stages (sato_kernels.hip) and their driver (stack_pipeline.cpp)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
namespace tmat {
enum { EXT_NEAREST = 0, EXT_REFLECT = 1, EXT_MIRROR = 2 };      // scipy.ndimage boundary modes 'nearest' / 'reflect' / 'mirror'

void launch_corr1d_f32(const float *in, float *out, size_t outer, int L, int inner, const double *w, int r, int sym, int mode, hipStream_t s);
void launch_corr1d_f64(const double *in, double *out, size_t outer, int L, int inner, const double *w, int r, int sym, int mode, hipStream_t s);
void launch_corr1d_u16_f64(const uint16_t *in, double *out, size_t outer, int L, int inner, const double *w, int r, int sym, int mode, hipStream_t s);
void launch_corr1d_f64_u16(const double *in, uint16_t *out, size_t outer, int L, int inner, const double *w, int r, int sym, int mode, hipStream_t s);
int stack_zoom_rescale_dev(const double *filtered, const uint16_t *stack_after_gauss, int Z, int H, int W, int oh, int ow, const int *r0,
                           const int *r1, const double *wr0, const double *wr1, const int *c0, const int *c1, const double *wc0,
                           const double *wc1, double *zoomed, unsigned long long *mm, double *lohi, float *vol, hipStream_t s);
void launch_pairmax(const float *vol, int Zm1, size_t npx, int negate, float *out, hipStream_t s);
void launch_prep_single(const float *img, size_t total, int negate, float *out, hipStream_t s);
void launch_gradient(const float *f, float *out, size_t outer, int L, int inner, hipStream_t s);
void launch_eig(int derivative_form, const float *hrr, const float *hrc, const float *hcc, float s2, size_t total, float *best, int first, hipStream_t s);
void launch_unsharp(const float *v, const float *blurred, float amount, size_t total, float *out, hipStream_t s);
void launch_zmax(const float *v, int Z, size_t npx, float *out, hipStream_t s);
struct CannyWs {
    double *sm, *t0, *is_, *js, *mag;       // H W f64 each
    uint8_t *low, *high;
    int *L, *flag;
    const double *w_diff, *w_smooth;        // device tables [-1, 0, 1] and [1, 2, 1]
};
int canny0_dev(const float *img, int H, int W, const CannyWs &ws, uint8_t *edges, hipStream_t s);
int canny_core_dev(int H, int W, const CannyWs &ws, uint8_t *edges, hipStream_t s);      // ws.sm holds the smoothed image
int ecc_diam_select_dev(const uint8_t *mask, int H, int W, double thresh, int *L, unsigned long long *mom, uint8_t *out, double *val_out, hipStream_t s);
void launch_where(const uint8_t *m, const float *a, const float *b, size_t n, float *out, hipStream_t s);     // b null: zeros
void launch_grow(const uint8_t *m, const float *v, int H, int W, uint8_t *out, hipStream_t s);
void launch_andnot(const uint8_t *a, const uint8_t *b, size_t n, uint8_t *out, hipStream_t s);
void launch_morph(const uint8_t *m, int H, int W, const int *off_dev, int noff, int erode, uint8_t *out, hipStream_t s);
}  // namespace tmat
