// host pixel stages (postproc.cpp)
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <vector>
namespace tmat {
void lanczos_axis(int n_src, int n_dst, std::vector<int> &idx, std::vector<float> &co);
void lanczos4_resize_u16(const uint16_t *img, int H, int W, int h, int w, uint16_t *out);
void rescale01_u16(const uint16_t *img, size_t n, float *out);
void rescale255_f32(const float *img, size_t n, float *out);
void median13(const uint8_t *m, int H, int W, uint8_t *out);
int label8(const uint8_t *m, int H, int W, std::vector<int32_t> &lab);
void skeletonize_zhang(const uint8_t *m, int H, int W, uint8_t *out);
void filter_mask(const uint8_t *mask_in, int H, int W, bool use_median, bool remove_isolated, uint8_t *out);
void edt(const uint8_t *m, int H, int W, double *dist);
void legacy_permutation(uint32_t seed, size_t n, std::vector<uint32_t> &perm);
void medial_axis(const uint8_t *m, int H, int W, uint8_t *skel, double *dist);
void medial_table_bits(uint32_t out[16]);          // the 512 decisions of the medial-axis table, bit idx of word idx >> 5
void medial_axis_thin(const uint8_t *m, const double *dist, int H, int W, uint8_t *skel);
void postprocess_from_filtered(const double *pred, const uint8_t *filt, const double *dist, int H, int W, int oh, int ow, float *field);
void resize_aa(const double *img, int H, int W, int oh, int ow, float *out);
void postprocess_image(const double *pred, int H, int W, int oh, int ow, float *field);
int dmt_graph_host(const float *img, int R, int C, float delta1, float delta2, int32_t *verts, int cap_v, int32_t *edges,
                   int cap_e, int *n_verts, int *n_edges);
int dmt_graph_host_sorted(const float *img, int R, int C, float delta1, float delta2, const int32_t *sorted, int m, int32_t *verts,
                          int cap_v, int32_t *edges, int cap_e, int *n_verts, int *n_edges,
                          const uint8_t *kind_in = nullptr, const float *pers_in = nullptr);
// with a handle: key build + lower-star sort + the persistence sweeps on its device (csrc/dmt_kernels.hip, dmt_sweep_kernels.hip),
// collect on the host (pipeline.cpp)
int dmt_graph_device_batch(void *handle, const float *imgs, int n, int R, int C, float delta1, float delta2, int32_t *verts, int cap_v,
                           int32_t *edges, int cap_e, int *n_verts, int *n_edges);
int dmt_graph_device_front(void *handle, const float *img, int R, int C, float delta1, float delta2, int32_t *verts, int cap_v,
                           int32_t *edges, int cap_e, int *n_verts, int *n_edges);
}  // namespace tmat
