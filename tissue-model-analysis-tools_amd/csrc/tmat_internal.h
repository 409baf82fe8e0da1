// Internal declarations shared by the translation units of libtmat_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

namespace tmat {

void set_error(const std::string &msg);
bool hip_ok(hipError_t e, const char *what);
#define TMAT_HIP(call) do { if (!::tmat::hip_ok((call), #call)) return TMAT_E_HIP; } while (0)

// ---- UNet layer launchers (unet_kernels.hip); all tensors NHWC f32 on device ----------------
struct ConvArgs {
    const float *in;      // stored tensor (N, h, w, Cin)
    int N, h, w, Cin;
    int relu_in;          // relu applied on load
    int ksize;            // 1 or 3; 2 = sub-pixel form of a 3x3 over the 2x nearest-upsampled stored tensor (out is 2h x 2w)
    int stride;           // 1, or 2 with ksize 1 (TF SAME 1x1 s2 samples even indices)
    const float *W;       // [ksize*ksize][Cout][Cin] (k contiguous); ksize 2: [4 parity classes][4][Cout][Cin]
    int Cout;
    const float *scale;   // nullable: v = scale ? fmaf(acc, scale, shift) : acc + shift
    const float *shift;
    const float *resid;   // nullable: v += resid[n][y >> rs][x >> rs][co]
    int rs;
    int relu_out;
    float *out;           // (N, h/stride, w/stride, Cout); ksize 2: (N, 2h, 2w, Cout)
    float *out_relu;      // nullable (ksize 1 / 3 only): a second, activated copy max(v, +0) of the output -- for a consumer that would otherwise apply ReLU on load
    int prec;             // 0: f32 MFMA (bit-exact contract); 1: bf16x3 split precision, W then points to the split copy of the weights
};
// returns false (and sets the error) on unsupported shapes
bool launch_conv(const ConvArgs &a, hipStream_t s);
// strips of the pooled separable convolution (floats): sepconv_ws_kernels.hip
size_t sepconv_pool_scratch_floats(int N, int H, int W, int Cout);
// fused SeparableConv2D, wave-specialised (sepconv_ws_kernels.hip): depthwise taps dw9 [9][Cin], pointwise pwk [Cout][Cin] (k contiguous);
// the pooled form fuses MaxPooling2D(3, 2, "same") + the residual add behind it: out (N, H/2, W/2, Cout), scratch: sepconv_pool_scratch_floats
bool sepconv_ws_supported(int H, int W, int Cin, int Cout);
bool launch_sepconv_ws(const float *in, int N, int H, int W, int Cin, int relu_in, const float *dw9, const float *pwk, int Cout,
                       const float *scale, const float *shift, int relu_out, float *out, hipStream_t s, int prec = 0);
bool launch_sepconv_pool_ws(const float *in, int N, int H, int W, int Cin, int relu_in, const float *dw9, const float *pwk, int Cout,
                            const float *scale, const float *shift, int relu_out, float *scratch, const float *resid, float *out, hipStream_t s, int prec = 0);
void launch_dwconv(const float *in, int N, int H, int W, int C, int relu_in, const float *Wd, float *out, hipStream_t s);
void launch_stem(const float *x, int N, int H, int W, const float *Ws, int Cout, const float *scale,
                 const float *shift, float *out, hipStream_t s);
// the stem at the even output pixels only: out (N, H/4, W/4, Cout); the input of block 0's stride-2 residual convolution when the stem
// itself is recomputed inside the first separable convolution (launch_sepconv_ws_stem)
void launch_stem_even(const float *x, int N, int H, int W, const float *Ws, int Cout, const float *scale,
                      const float *shift, float *out, hipStream_t s);
// SeparableConv2D over the stem's output WITHOUT the stem tensor in memory: x (N, 2H, 2W) single-channel patches; the depthwise
// producers recompute stem = relu(bn(conv3x3/s2(x))) (H x W x Cin) for their halo.  Same results as launch_stem + launch_sepconv_ws.
bool launch_sepconv_ws_stem(const float *x, int N, int H, int W, int Cin, const float *stem_w, const float *stem_scale, const float *stem_shift,
                            const float *dw9, const float *pwk, int Cout, const float *scale, const float *shift, int relu_out, float *out,
                            hipStream_t s);
void launch_maxpool_add(const float *p2, int N, int H, int W, int C, const float *r, float *out, hipStream_t s, float *out_relu = nullptr);
void launch_final(const float *S, int N, int h, int w, int C, const float *Wf, float bias, float *out, hipStream_t s);

// ---- tiling / blending (blend_kernels.hip) -------------------------------------------------
// x: (n, hh, ww) f32; padval[n]; patches out: (n, 8, na_g, nb_g, ws, ws)
struct TileGeom {
    int hh, ww;     // image
    int ws, step, aug;
    int Hp, Wp;     // padded
    int na[2], nb[2];   // tile counts for even (k=0,2) / odd (k=1,3) rotations
    int tiles_per_img;  // sum over 8 orientations
    int tile_off[8];    // first tile index of orientation g within the image
};
TileGeom make_geom(int hh, int ww, int ws);
void launch_minmax_f32(const float *x, int n, size_t per, float *mn, float *mx, hipStream_t s);
void launch_norm_f32(float *x, size_t n, float mean, float sd, hipStream_t s);          // x = (x - mean) / sd in place
void launch_extract_tiles(const float *x, const float *padval, int n, const TileGeom &g, float *patches, hipStream_t s);
void launch_blend(const float *pred_patches, const double *win1d, int n, const TileGeom &g, double *out, hipStream_t s);

// ---- ordered medial-axis thinning on the device (thin_kernels.hip) ------------------------------------
bool thin_dev_supported(int H, int W);
size_t thin_workspace_bytes(int n, int H, int W);
int thin_count_dev(const uint8_t *mask, int n, int H, int W, int *nfg, hipStream_t s);
int thin_dev(const uint8_t *mask, const double *dist, const uint32_t *tie, const int *nfg, int n, int H, int W, void *ws,
             const uint32_t *table_dev, uint8_t *skel, hipStream_t s);

// ---- DMT front end on the device (dmt_kernels.hip) -------------------------------------------------
inline size_t dmt_edge_count(int R, int C) { return (size_t)(R - 1) * C + (size_t)R * (C - 1) + (size_t)(R - 1) * (C - 1); }
size_t dmt_workspace_bytes(int n, int R, int C);
int dmt_sorted_edges_dev(const float *field, int n, int R, int C, void *ws, int32_t *ids, int *m, hipStream_t s);
// the two persistence sweeps on the device as levels of data-parallel steps, one workgroup per image (dmt_sweep_kernels.hip;
// TMAT_DMT_SWEEP_DEVICE=0 keeps them on host threads)
size_t dmt_sweep_workspace_bytes(int n, int R, int C);
int dmt_sweeps_dev(const float *field, const int32_t *ids, const int *m, int n, int R, int C, void *ws, uint8_t *kind, float *pers, hipStream_t s);

}  // namespace tmat
