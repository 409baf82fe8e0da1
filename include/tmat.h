/*
 * tmat.h -- C-ABI of libtmat_hip.so: the MI355X (gfx950) implementation of the 2-D
 * microvessel-branching hot path of fogg-lab/tissue-model-analysis-tools
 * (scripts/compute_branches.py:307-361,391-426 and the fl_tissue_model_tools functions it drives).
 *
 * The reference is pure Python; a maintainer binds this library with ctypes (INTEGRATION.md shows
 * the stub).  All buffers are caller-owned, contiguous, row-major.  Every function returns 0 on
 * success or a negative code; tmat_last_error() returns a message for the calling thread.
 * No torch / Python types cross this boundary.  Functions taking a `tmat_handle` run on that
 * handle's HIP device and stream; a handle is not thread-safe, separate handles are independent.
 *
 * Pointer suffix convention:  *_dev arguments are HIP device pointers on the handle's device,
 * everything else is host memory.
 */
#ifndef TMAT_H
#define TMAT_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tmat_ctx *tmat_handle;

#define TMAT_OK 0
#define TMAT_E_ARG (-1)     /* bad argument / unsupported shape            */
#define TMAT_E_HIP (-2)     /* HIP runtime error (message has the detail)  */
#define TMAT_E_WEIGHTS (-3) /* malformed weight blob                       */
#define TMAT_E_CAP (-4)     /* caller-provided output capacity too small   */

const char *tmat_last_error(void);
/* 0x00MMmmpp */
int tmat_version(void);

/*
 * Model life-cycle.  Replaces models.get_unet_patch_segmentor_from_cfg /
 * UNetXceptionPatchSegmentor.__init__ (reference models.py:600-622, 656-684): builds the
 * UNet-Xception of models.py:85-171 and loads its weights once per run (compute_branches.py:576).
 * `weights_blob` is the "TMATW001" container (tmat_amd/synth.py:pack_weights; Keras tensor
 * layouts) -- the offline stand-in for checkpoint_1.h5.
 * `max_patches` sizes the activation workspace, i.e. the UNet batch held in HBM at once (39 MB per patch; 0 = default
 * 400 = two 1024^2 images per pass).  It is not a limit on the image size: an image that needs more patches runs its
 * patch list in chunks, one image per pass.
 */
int tmat_create(int device_id, const void *weights_blob, size_t n_bytes, int max_patches, tmat_handle *out);
/* A handle without weights (device + stream only) for the entry points that need no model: tmat_zproj_*,
 * tmat_filter_edt_batch, tmat_finish_batch.  Model entry points return TMAT_E_ARG on it. */
int tmat_create_plain(int device_id, tmat_handle *out);
void tmat_destroy(tmat_handle h);
/* blocks until all work queued on the handle's stream is done */
int tmat_sync(tmat_handle h);

/* Bit depth of the images the next segment / analyze calls receive (16, the default, or 8).  Images always cross the
 * ABI as uint16; cv2.resize saturates its result to the depth of its input (compute_branches.py:309-312), so 8-bit
 * sources (widened by the caller) must saturate the Lanczos overshoot at 255 instead of 65535. */
int tmat_set_input_depth(tmat_handle h, int bits);

/*
 * keras_model.predict on a patch batch (reference models.py:644 as called from
 * smooth_tiled_predictions.py:179,182): x (n, P, P) f32 -> y (n, P, P) f32 sigmoid output
 * (the trailing channel axis of size 1 is implicit).  P = patch size of the blob (320).
 */
int tmat_unet_predict(tmat_handle h, const float *x, int n, float *y);

/*
 * UNetXceptionPatchSegmentor.predict(x, auto_resample=False) (reference models.py:624-653) =
 * predict_img_with_smooth_windowing(x, window_size=P, subdivisions=2, pred_func=model.predict)
 * (smooth_tiled_predictions.py:220-267): pad, 8 D4 copies, 50 %-overlap tiles, UNet, squared-spline
 * window, f64 overlap-add, D4 undo + mean, unpad.  x: (n, h, w) f32; pred: (n, h, w) f64.
 */
int tmat_predict_smooth(tmat_handle h, const float *x, int n, int hh, int ww, double *pred);

/*
 * compute_branches.py:309-328 for a batch: cv2.resize(img, round(shape * ds_ratio), INTER_LANCZOS4)
 * -> rescale_intensity(out_range=(0,1)).astype(f32) -> model.predict(auto_resample=False).
 * imgs: (n, H, W) u16; pred: (n, round(W*ds_ratio), round(H*ds_ratio)) f64 -- the reference passes its target shape to
 * cv2.resize as dsize, which cv2 reads as (width, height), so rows and columns swap roles for non-square images
 * (the later resize to the 384-wide field restores the aspect ratio); square images are unaffected.
 */
int tmat_segment_batch(tmat_handle h, const uint16_t *imgs, int n, int H, int W, double ds_ratio, double *pred);

/*
 * compute_branches.py:334-361 for a batch: pred > 0.5 -> filter_branch_seg_mask
 * (transforms.py:306-361) -> medial_axis + EDT centre-line weighting -> skimage resize to
 * img_dsamp_res = (out_h, out_w) (compute_branches.py:218-222: round(orig_shape * 384 / orig_width)),
 * order 1, anti-aliased -> f32.   pred: (n, h, w) f64; field: (n, out_h, out_w) f32.
 * Runs on the handle's device like the batch pipeline: threshold, filter, EDT, ordered medial-axis thinning (only its
 * tie-break permutation is made on the host), EDT of the skeleton, weighting, resize.
 * A handle from tmat_create_plain is enough.
 */
int tmat_postprocess_batch(tmat_handle h, const double *pred, int n, int hh, int ww, int out_h, int out_w, float *field);

/*
 * The GPU part of the above (csrc/morph_kernels.hip): pred > 0.5 -> filter_branch_seg_mask
 * (transforms.py:306-361, remove_isolated=True) -> EDT of the filtered mask (the `distance` output of
 * medial_axis(seg_mask, return_distance=True), compute_branches.py:340).
 * pred (n, h, w) f64; filtered (n, h, w) u8; dist (n, h, w) f64.  Needs a handle (runs on its device).
 */
int tmat_filter_edt_batch(tmat_handle h, const double *pred, int n, int hh, int ww, uint8_t *filtered, double *dist);

/*
 * The two pixel stages in front of the UNet on their own (compute_branches.py:309-316): cv2.resize(img, round(shape *
 * ds_ratio), INTER_LANCZOS4) and rescale_intensity(out_range=(0, 1)).astype(float32), on the device.  imgs (n, H, W) u16
 * host; x (n, round(W ds_ratio), round(H ds_ratio)) f32 host.  This is the image `make_well_mask` sees when --detect-well
 * is given (:318-319); tmat_predict_smooth continues from it.
 */
int tmat_preprocess_batch(tmat_handle h, const uint16_t *imgs, int n, int H, int W, double ds_ratio, float *x);

/*
 * Well detection (--detect-well), device stages of fl_tissue_model_tools/well_mask_generation.py:
 *   tmat_well_threshold = auto_threshold_well (:236-277): skimage gaussian(sigma 1) -> rescale_intensity(0..255) -> uint8 ->
 *     corner medians decide whether to invert -> skimage.filters.threshold_otsu (256-bin histogram kernel + a decision
 *     kernel that evaluates the inter-class variance exactly as numpy does) -> binary_erosion(footprint=disk(5)).
 *     img (H, W) f32 host -> out (H, W) u8 host.  H, W >= 20.
 *   tmat_canny_mask = skimage.feature.canny(mask, sigma) of a boolean image, default thresholds (:165, :201): mask, edges
 *     (H, W) u8 host.
 * The small host steps between them (nearest-neighbour rescale to <= 200 px, scipy.spatial.ConvexHull, the hull mask, the
 * seeded random superellipse search :16-91) are Python host code in tmat_amd/well_mask_generation.py.
 * A handle from tmat_create_plain is enough.
 */
int tmat_well_threshold(tmat_handle h, const float *img, int H, int W, uint8_t *out);
/* the same for a float64 image: integer images (compute_cell_area.py:127) enter skimage's gaussian through img_as_float */
int tmat_well_threshold_f64(tmat_handle h, const double *img, int H, int W, uint8_t *out);
int tmat_canny_mask(tmat_handle h, const uint8_t *mask, int H, int W, double sigma, uint8_t *edges);

/*
 * transforms.filter_branch_seg_mask(mask, footprint, remove_isolated) (transforms.py:306-361) on the GPU for a batch of
 * uint8 masks: use_median 1 = footprint disk(2) (the default), 0 = footprint None (compute_branches.py:293).
 * mask, filtered (n, h, w) u8.  A handle from tmat_create_plain is enough.
 */
int tmat_filter_mask_batch(tmat_handle h, const uint8_t *mask, int n, int hh, int ww, int use_median, int remove_isolated,
                           uint8_t *filtered);

/*
 * skimage.morphology.medial_axis(mask, return_distance=True) (reference call compute_branches.py:340; scikit-image
 * 0.18.3 semantics with the RandomState(0) tie-break) for a batch, on the device: exact EDT (morph_kernels.hip), then the
 * ordered thinning (thin_kernels.hip: keys from distance / corner score / tie-break; a pixel's decision depends only on
 * its 8 neighbours with smaller keys, so the removal order is resolved as a dependency wavefront in strict Jacobi rounds).
 * Only the tie-break permutation (a Mersenne-Twister shuffle that depends on the foreground COUNT alone) is produced on
 * the host.  mask, skel (n, h, w) u8; dist (n, h, w) f64.  Images with h^2 + w^2 >= 2^27 are refused here (the squared
 * distance no longer fits its key field); the pipeline runs those through the host implementation
 * (tmat_host_medial_axis).  A handle from tmat_create_plain is enough.
 */
int tmat_medial_axis_batch(tmat_handle h, const uint8_t *mask, int n, int hh, int ww, uint8_t *skel, double *dist);

/*
 * The GPU stages after the medial-axis thinning (csrc/finish_kernels.hip): centerline_dt = EDT(~skel),
 * pred *= dist / (dist + centerline_dt) (compute_branches.py:341-344), skimage resize to (out_h, out_w)
 * (order 1, anti-aliased, :351-357) -> field f32, and rescale_intensity(field, (0, 255)) (:419) -> field255 f32.
 * pred, dist (n, h, w) f64; skel (n, h, w) u8; field, field255 (n, out_h, out_w) f32.
 */
int tmat_finish_batch(tmat_handle h, const double *pred, const double *dist, const uint8_t *skel, int n, int hh, int ww,
                      int out_h, int out_w, float *field, float *field255);

/*
 * fl_tissue_model_tools.dmtgraph.compute_dmt_graph(img, delta1, delta2) (reference
 * dmtgraph.py:38-99; the contract the un-vendored pydmtgraph C++ extension exposed).
 * img: (rows, cols) f32.  verts: (cap_v, 2) int32 [row, col]; edges: (cap_e, 2) int32.
 * With a handle the edge keys, the lower-star sort and both persistence sweeps run on its device
 * (TMAT_DMT_SWEEP_DEVICE=0: the sweeps on the host), `collect` on the host; `h` may be NULL
 * (host-only execution, sequential sweeps).
 */
int tmat_dmt_graph(tmat_handle h, const float *img, int rows, int cols, float delta1, float delta2,
                   int32_t *verts, int cap_v, int32_t *edges, int cap_e, int *n_verts, int *n_edges);

/*
 * The same for n fields of one shape in one call -- the loop over images around compute_dmt_graph
 * (reference topology.py:148-160 per image, scripts/compute_branches.py:585-594 over images): edge keys,
 * the lower-star sort and the two persistence sweeps of all fields run as one launch each
 * (dmtgraph.py:57-93, :277-314), `collect` (dmtgraph.py:317-453) per field on host threads.
 * imgs: (n, rows, cols) f32.  verts: (n, cap_v, 2) int32, edges: (n, cap_e, 2) int32; n_verts, n_edges: (n).
 */
int tmat_dmt_graph_batch(tmat_handle h, const float *imgs, int n, int rows, int cols, float delta1, float delta2,
                         int32_t *verts, int cap_v, int32_t *edges, int cap_e, int *n_verts, int *n_edges);

/*
 * topology.MorseGraph(img, thresholds, min_branch_length, max_branch_length,
 * remove_isolated_branches, smoothing_window, pruning_mask) downstream of compute_dmt_graph
 * (reference topology.py:148-179, 181-356): smoothing, trimming, spanning forest, branch labels,
 * barcode, min-length filter.  verts/edges are compute_dmt_graph's outputs.
 * max_branch_length <= 0 means None.  pruning_mask (rows, cols) u8 or NULL.
 * Outputs: *count = len(barcode); *total_px = get_total_branch_length();
 * *avg_px = get_average_branch_length(); bars (cap, 2) f64 [birth, death] (may be NULL).
 */
int tmat_morse_stats(const int32_t *verts, int n_verts, const int32_t *edges, int n_edges, int rows, int cols,
                     int smoothing_window, int min_branch_length, int max_branch_length,
                     int remove_isolated_branches, const uint8_t *pruning_mask, int64_t *count,
                     double *total_px, double *avg_px, double *bars, int cap);

/* One result row per image, the payload gathered across ranks. */
typedef struct tmat_row {
    int64_t index;   /* image index in the run                      */
    int64_t count;   /* Total # of branches                         */
    double total_px; /* total branch length in 384-wide pixels      */
    double avg_px;   /* average branch length in 384-wide pixels    */
} tmat_row;

/*
 * The whole per-image compute of analyze_img's 2-D branch (compute_branches.py:307-361, 391-426,
 * 455-457) for a batch with inputs ALREADY RESIDENT IN HBM: imgs_dev (n, H, W) u16 device pointer.
 * rows: n host records (index field = first_index + i).  graph_thresh_1/2, smoothing_window_px,
 * min_branch_length_px, max_branch_length_px (<=0: none), remove_isolated: as computed at
 * compute_branches.py:401-426.
 */
int tmat_analyze_batch_dev(tmat_handle h, const uint16_t *imgs_dev, int n, int H, int W, double ds_ratio,
                           int ds_width, float graph_thresh_1, float graph_thresh_2, int smoothing_window_px,
                           int min_branch_length_px, int max_branch_length_px, int remove_isolated,
                           int64_t first_index, tmat_row *rows);

/* host-pointer convenience wrapper of the above (uploads imgs first) */
int tmat_analyze_batch(tmat_handle h, const uint16_t *imgs, int n, int H, int W, double ds_ratio, int ds_width,
                       float graph_thresh_1, float graph_thresh_2, int smoothing_window_px,
                       int min_branch_length_px, int max_branch_length_px, int remove_isolated,
                       int64_t first_index, tmat_row *rows);

/* ---------------------------------------------------------------------------------------------------------------
 * Z-stack (Sato) branch of analyze_img: reference scripts/compute_branches.py:224-306 (csrc/sato_kernels.hip,
 * csrc/stack_pipeline.cpp).  Its arithmetic lives in scikit-image / scipy.ndimage (reference setup.py pins
 * scikit-image==0.22.0, scipy==1.13.1); oracle/sato.py restates it.  A handle from tmat_create_plain is enough.
 * --------------------------------------------------------------------------------------------------------------- */

/* Hessian of skimage.filters.sato: 0 = gaussian derivatives (scikit-image >= 0.20, what the pinned 0.22.0 runs:
 * hessian_matrix(use_gaussian_derivatives=True), -image, vesselness sigma^2 * max(l1, 0));  1 = gradient of the smoothed
 * image (scikit-image <= 0.19: gaussian, np.gradient twice, 1 - image, sigma^2-scaled elements). */
#define TMAT_SATO_GAUSSIAN_DERIVATIVES 0
#define TMAT_SATO_GRADIENT 1
/* scipy.ndimage boundary modes of tmat_gaussian_f32 */
#define TMAT_EXT_NEAREST 0
#define TMAT_EXT_REFLECT 1
#define TMAT_EXT_MIRROR 2

/*
 * Replace the kernel table of scipy's gaussian_filter1d for (sigma, order, radius) on this handle: weights = the 2 radius + 1
 * values scipy hands to correlate1d (ndimage/_filters.py:_gaussian_kernel1d(sigma, order, radius)[::-1]).  Without it the
 * library forms the table with libm's exp; numpy's exp is CPU-dispatched and differs from libm in the last bit at some taps,
 * so a host that wants scipy's exact results on its machine hands over numpy-made tables (tmat_amd/sato.py does).
 */
int tmat_set_gaussian_table(tmat_handle h, double sigma, int order, int radius, const double *weights);
/* the library's own table (host only): scipy's formula with libm's exp and numpy's pairwise sum; order 0 or 1 */
int tmat_host_gaussian_kernel1d(double sigma, int order, int radius, double *weights);

/* skimage.filters.gaussian(x, sigma, mode) on a float32 array (d0, d1, d2) (= ndi.gaussian_filter, truncate 4, f64
 * accumulation, f32 after every axis); d0 = 1 filters a 2-D image (d1, d2).  compute_branches.py:248, 269, 282, 302 */
int tmat_gaussian_f32(tmat_handle h, const float *x, int d0, int d1, int d2, double sigma, int mode, float *out);

/* skimage.filters.sato(img, sigmas, black_ridges=False) on n float32 images (n, h, w) (compute_branches.py:261-263) */
int tmat_sato_batch(tmat_handle h, const float *imgs, int n, int hh, int ww, const double *sigmas, int n_sigmas, int hessian,
                    float *out);

/* compute_branches.py:247-256: per-slice gaussian written back into the integer stack, skimage resize of the stack to
 * (Z, out_h, out_w) (order 1, anti-aliased, preserve_range), rescale_intensity(0..1) over the stack.
 * stack (Z, H, W) u16 host (8-bit stacks widened); vol (Z, out_h, out_w) f32 host. */
int tmat_stack_prepare(tmat_handle h, const uint16_t *stack, int Z, int H, int W, int out_h, int out_w, float *vol);

/* optional host copies of the stages of tmat_vessel_field (any member may be NULL) */
typedef struct tmat_vessel_stages {
    float *vess;       /* (Z-1, h, w) Sato response of every slice pair          :258-266 */
    float *sharp;      /* (Z-1, h, w) unsharp_mask(vess, 2, 2)                   :269     */
    float *vessels;    /* (h, w) its max projection                               :270     */
    uint8_t *edges;    /* (h, w) canny(vessels, sigma=0)                          :271     */
    uint8_t *skel;     /* (h, w) medial_axis(edges)                               :274     */
    uint8_t *mask_sel; /* (h, w) components with eccentricity * diameter > 3.5    :276-279 */
    uint8_t *grown;    /* (h, w) after the 10 region-growing rounds               :283-294 */
    uint8_t *closed;   /* (h, w) closing(mask & ~edges, disk(2))                  :296-297 */
    uint8_t *filt;     /* (h, w) filter_branch_seg_mask(closed, None, False)      :299     */
} tmat_vessel_stages;

/* compute_branches.py:258-302: prepared stack vol (Z, h, w) f32 in 0..1 -> vesselness image field (h, w) f32 */
int tmat_vessel_field(tmat_handle h, const float *vol, int Z, int hh, int ww, int hessian, float *field,
                      const tmat_vessel_stages *stages);

/*
 * The whole per-image compute of analyze_img for a Z stack (compute_branches.py:224-306, 391-426, 455-457; no well
 * mask): stack (Z, H, W) u16 host -> one result row.  The field is (round(H ds_width / W), ds_width); the graph
 * parameters are those of tmat_analyze_batch.  field_out (nullable) receives the vesselness image.
 * The device blocks a Z-stack call allocates (this and the stage-wise entry points above) go back to a pool on the handle
 * when it returns and are taken from there by the next call; tmat_destroy frees the pool.
 */
int tmat_analyze_stack(tmat_handle h, const uint16_t *stack, int Z, int H, int W, int ds_width, int hessian,
                       float graph_thresh_1, float graph_thresh_2, int smoothing_window_px, int min_branch_length_px,
                       int max_branch_length_px, int remove_isolated, int64_t index, tmat_row *row, float *field_out);

/* The common tail of analyze_img from a vesselness image (compute_branches.py:391-426, 455-457): rescale to 0..255, DMT
 * graph, MorseGraph statistics -> one row.  field (fh, fw) f32 host.  Lets a host that sweeps the graph_thresh grid
 * (:366-395) compute the field once. */
int tmat_field_stats(tmat_handle h, const float *field, int fh, int fw, float graph_thresh_1, float graph_thresh_2,
                     int smoothing_window_px, int min_branch_length_px, int max_branch_length_px, int remove_isolated,
                     int64_t index, tmat_row *row);

/* The same with MorseGraph's pruning_mask (topology.py: branches that end inside the mask are trimmed): the Z-stack
 * branch with --detect-well passes the inverse of the shrunken well mask (compute_branches.py:243, 412-420).
 * pruning_mask (fh, fw) u8 host, nonzero = prune; NULL = none. */
int tmat_field_stats_pruned(tmat_handle h, const float *field, int fh, int fw, float graph_thresh_1, float graph_thresh_2,
                            int smoothing_window_px, int min_branch_length_px, int max_branch_length_px, int remove_isolated,
                            const uint8_t *pruning_mask, int64_t index, tmat_row *row);

/* skimage.transform.resize(x, (n, out_h, out_w), order=1, preserve_range=True, anti_aliasing=True) of n integer images
 * (scikit-image >= 0.19: anti-aliasing gaussian + grid-mode linear zoom, clipped to the input's range), float64 out.
 * compute_branches.py:232-238 resizes the max projection of a Z stack with it before make_well_mask.
 * imgs (n, H, W) u16 host (8-bit images widened); out (n, out_h, out_w) f64 host. */
int tmat_resize_aa_u16(tmat_handle h, const uint16_t *imgs, int n, int H, int W, int out_h, int out_w, double *out);

/* ---------------------------------------------------------------------------------------------------------------
 * Cell-area tool: reference scripts/compute_cell_area.py:29-87, 164-178 and
 * fl_tissue_model_tools/preprocessing.py:44-93 (csrc/cellarea_kernels.hip).  A handle from tmat_create_plain is enough.
 * --------------------------------------------------------------------------------------------------------------- */

/*
 * For n uint16 images (n, H, W) (Z stacks: project first, tmat_zproj_batch method max): cv2.resize to (out_h, out_w)
 * with bilinear interpolation (compute_cell_area.py:57 passes INTER_AREA in the position of `dst`, so the default
 * interpolation applies; out_h = out_w = 0: no resize), rescale_intensity to 0..1 as float32 (:79),
 * preprocessing.exec_threshold (:44-93: two-component gaussian mixture of the pixel intensities -- fitted here to the
 * intensity histogram, from the optimal 2-means partition, with scikit-learn's EM update and stopping rule -- and the
 * threshold foreground mean + sd_coef * foreground sd), compute_area_prop (:164-178).
 * area: n fractions of pixels kept.  thresholded (nullable): (n, out_h, out_w) u8, 255 where kept (:87).
 * params (nullable): n x 9 doubles [threshold, weight0, weight1, mean0, mean1, var0, var1, EM iterations, converged].
 * The call's device workspaces (sized by the largest batch seen) stay on the handle until tmat_destroy: the reference's
 * default batch is 4 images, for which a hipMalloc / hipFree pair per buffer and call cost more than the kernels.
 * tmat_inv_depth_predict keeps its workspaces the same way.
 */
int tmat_cell_area_batch(tmat_handle h, const uint16_t *imgs, int n, int H, int W, int out_h, int out_w, double sd_coef,
                         double *area, uint8_t *thresholded, double *params);

/*
 * The --detect-well form of the same tool (compute_cell_area.py:117-130, 273-286; preprocessing.exec_threshold with
 * mask_idx): imgs (n, H, W) u16 are the ALREADY down-sampled images, masks (n, H, W) u8 their well masks
 * (well_mask_generation.generate_well_mask(img, mask_val=255) > 0).  rescale_intensity uses the extrema of the whole image,
 * the mixture is fitted to the pixels inside the mask only, pixels outside count as background.  area[i] = kept pixels /
 * (H W): the caller divides by the well's pixel count (compute_area_prop with well_pix_area).
 * tmat_resize_linear_u16 = the down-sampling step alone (compute_cell_area.py:54-57, cv2.resize with the default
 * INTER_LINEAR; exactly halving both axes takes cv2's INTER_AREA shortcut (a + b + c + d + 2) >> 2): imgs (n, H, W) ->
 * out (n, out_h, out_w), host buffers.
 */
int tmat_cell_area_masked(tmat_handle h, const uint16_t *imgs, const uint8_t *masks, int n, int H, int W, double sd_coef, double *area,
                          uint8_t *thresholded, double *params);
int tmat_resize_linear_u16(tmat_handle h, const uint16_t *imgs, int n, int H, int W, int out_h, int out_w, uint16_t *out);

/* ---------------------------------------------------------------------------------------------------------------
 * Invasion-depth tool: reference scripts/compute_inv_depth.py:96-172, models.py:33-82 (build_ResNet50_TL),
 * data_prep.py:17-61 (csrc/resnet_kernels.hip; the 1x1 / 3x3 convolutions run on the conv_mfma_kernel of the branching
 * path).  A handle from tmat_create_plain is enough.
 * --------------------------------------------------------------------------------------------------------------- */

/* One classifier = keras.applications ResNet50 up to a `conv{N}_block{K}_out` layer + GlobalAveragePooling2D + Dense(1) +
 * sigmoid.  weights_blob: "TMATW001" container (tmat_amd/inv_depth.py:pack_resnet) with Keras layouts -- conv1.{w,b,bn},
 * s{stage}b{block}.c{1,2,3}.{w,b,bn} (+ .c0 for the projection shortcut of block 1), fc.{w,b}; bn = (4, C) gamma, beta,
 * moving mean, moving variance, eps 1.001e-5.  Returns the model's id on this handle. */
int tmat_resnet_load(tmat_handle h, const void *weights_blob, size_t n_bytes, int *model_id);
/* model.predict(x): x (n, size, size, 3) float32 host (already preprocessed), prob (n) float32.  size % 32 == 0 */
int tmat_resnet_predict(tmat_handle h, int model_id, const float *x, int n, int size, float *prob);
/* compute_inv_depth.py:150-156 for one Z stack: data_prep.prep_inv_depth_imgs (cv2.resize to size x size -- bilinear: the
 * interpolation argument sits in the position of `dst` --, rescale_intensity 0..255, three identical channels,
 * resnet50.preprocess_input) and every model's prediction per slice.  stack (Z, H, W) u16 host; probs (Z, n_models);
 * x_out (nullable): the prepared input (Z, size, size, 3) f32. */
int tmat_inv_depth_predict(tmat_handle h, const int *model_ids, int n_models, const uint16_t *stack, int Z, int H, int W,
                           int size, float *probs, float *x_out);
/* The same for n_stacks Z stacks of one (H, W) in ONE call (compute_inv_depth.py:123-166 loops over the stacks; every step is per
 * slice, so stacks ride together through the classifiers): stacks[k] (Zs[k], H, W) u16 host -- uploaded back to back, no
 * concatenation on the host --, probs (sum of Zs, n_models) in stack order. */
int tmat_inv_depth_predict_multi(tmat_handle h, const int *model_ids, int n_models, const uint16_t *const *stacks, const int *Zs,
                                 int n_stacks, int H, int W, int size, float *probs);

/* device memory helpers so a ctypes host can stage inputs in HBM without torch */
int tmat_dev_alloc(tmat_handle h, size_t bytes, void **dev_ptr);
int tmat_dev_free(tmat_handle h, void *dev_ptr);
int tmat_dev_upload(tmat_handle h, void *dev_dst, const void *host_src, size_t bytes);

/*
 * Host pixel stages, exposed one by one for stage-wise parity tests (csrc/postproc.cpp): host twins of the device
 * stages.  The batch entry points above run these stages on the GPU (morph_kernels.hip, thin_kernels.hip,
 * finish_kernels.hip); the twins are reached only through these test entry points, TMAT_THIN_DEVICE=0 and images with
 * h^2 + w^2 >= 2^27 -- they are not a fallback for a missing device.
 *   lanczos4: cv2.resize(INTER_LANCZOS4) on u16 (compute_branches.py:312); rescale01: rescale_intensity
 *   (0,1) of an integer image -> f32 (:316); rescale255: rescale_intensity (0,255) in f32 (:419);
 *   filter_mask: transforms.filter_branch_seg_mask (transforms.py:306-361); skeletonize: skimage
 *   skeletonize (transforms.py:331); medial_axis: skimage medial_axis(return_distance=True)
 *   (compute_branches.py:340); permutation: numpy RandomState(seed).permutation(arange(n));
 *   postprocess: compute_branches.py:334-357 for one image.
 */
int tmat_host_lanczos4_u16(const uint16_t *img, int H, int W, int h, int w, uint16_t *out);
int tmat_host_rescale01_u16(const uint16_t *img, size_t n, float *out);
int tmat_host_rescale255_f32(const float *img, size_t n, float *out);
int tmat_host_filter_mask(const uint8_t *mask, int H, int W, int use_median, int remove_isolated, uint8_t *out);
int tmat_host_skeletonize(const uint8_t *mask, int H, int W, uint8_t *out);
int tmat_host_medial_axis(const uint8_t *mask, int H, int W, uint8_t *skel, double *dist);
int tmat_host_permutation(uint32_t seed, int n, uint32_t *out);
int tmat_host_postprocess(const double *pred, int H, int W, int out_h, int out_w, float *field);

/*
 * The one collective of the path (compute_branches.py:585-594 writes its rows from one process; with one process per
 * GPU the rows are gathered first): all-gather of result rows over RCCL.  `rccl_comm` is an ncclComm_t of the caller,
 * `hip_stream` a hipStream_t (NULL = default stream); rows_dev (n_local) and out_dev (world * n_local) are device
 * buffers; every rank passes the same n_local (pad short shards).  Asynchronous on the stream.  The Python host uses
 * torch.distributed's all_gather instead, which is the same RCCL call on torch's communicator.
 */
int tmat_gather_rows(void *rccl_comm, const tmat_row *rows_dev, int n_local, tmat_row *out_dev, void *hip_stream);

/*
 * Z projection of image stacks -- what scripts/compute_zproj.py:73-84 calls through proj_methods
 * (fl_tissue_model_tools/zstacks.py): "fs" proj_focus_stacking (zstacks.py:153-189: cv2.GaussianBlur 5x5 sigma 0, then
 * cv2.Laplacian(CV_64F, ksize 5), per pixel the value of the first slice with the strictly largest |Laplacian|),
 * "min" / "max" (np.min / np.max, :222-249), "avg" (np.mean, :192-204), "med" (np.median, :207-219).
 * stacks (n, Z, H, W) u16 (uint8 stacks are widened by the caller: the arithmetic is the same); out (n, H, W):
 * uint16 for fs / min / max, float64 for avg / med.  tmat_zproj_batch takes host pointers and streams the stacks
 * through the device in chunks; tmat_zproj_dev takes device pointers and is asynchronous on the handle's stream
 * (chain it in front of tmat_analyze_batch_dev).
 */
enum { TMAT_ZPROJ_FS = 0, TMAT_ZPROJ_MIN = 1, TMAT_ZPROJ_MAX = 2, TMAT_ZPROJ_AVG = 3, TMAT_ZPROJ_MED = 4 };
int tmat_zproj_batch(tmat_handle h, const uint16_t *stacks, int n, int Z, int H, int W, int method, void *out);
int tmat_zproj_dev(tmat_handle h, const uint16_t *stacks_dev, int n, int Z, int H, int W, int method, void *out_dev);

/*
 * Arithmetic of the dense convolutions of the UNet (reference: keras Model.predict in float32, models.py:615-622, 644).
 *   TMAT_PRECISION_F32    (default) f32 operands on v_mfma_f32_32x32x2_f32: bit-exact with oracle/unet_exact.c.
 *   TMAT_PRECISION_BF16X3 opt-in split precision: every f32 operand as a bf16 hi + bf16 lo pair, three bf16 MFMAs per
 *                         product (lo*hi + hi*lo + hi*hi) with f32 accumulation; about 2^-16 relative error per product.
 *   TMAT_PRECISION_BF16X6 opt-in: hi + mid + lo (the whole 24-bit mantissa), the six products of order <= 2 (lo*hi, hi*lo,
 *                         mid*mid, mid*hi, hi*mid, hi*hi): 2^-24-level error per product at 3/8 of the f32 path's
 *                         matrix-pipe time.  NOT f32-equivalent at row level: see the measured outcome below.
 *   Neither opt-in mode is bit-exact with the oracle, and NEITHER MEETS the north-star tolerance (branch counts equal, lengths
 *   within 1e-4 relative) at row level on the 256 bench images: measured against the f32 rows (BENCH_r03 "alt" blocks,
 *   profiles/r03_bench.json) bf16x3 flips the branch count of 1 image, bf16x6 of 4 images, and where counts agree the largest
 *   relative length difference is 0.072 in both -- the graph is a discontinuous function of the probability map, so any change of
 *   summation order at the 1e-6 level flips a few rows.  bench.py reports both as a separate "alt" block with "pass": false;
 *   tests/test_gpu_alt_precision.py bounds the probability-map error and records the row flips.  The modes apply to the 3x3 /
 *   sub-pixel / 1x1 convolutions of conv_mfma_kernel; in BF16X3 mode the POINTWISE contraction of the fused separable
 *   convolutions also runs on the bf16 cores (two weight planes; TMAT_SEP_BF16=0 keeps those layers in f32), in BF16X6 mode the
 *   separable layers stay f32 (three planes do not fit the kernel's LDS budget); depthwise taps, stem and final layer are always f32.
 * Takes effect for the calls that follow (all three streams of the handle are drained first).  The environment variable
 * TMAT_PRECISION=f32|bf16x3|bf16x6 selects the mode at tmat_create.
 */
#define TMAT_PRECISION_F32 0
#define TMAT_PRECISION_BF16X3 1
#define TMAT_PRECISION_BF16X6 2
int tmat_set_precision(tmat_handle h, int mode);

/*
 * UNetXceptionPatchSegmentor's optional input normalisation, x = (x - norm_mean) / norm_std in float32 (reference
 * models.py:600-612, 636-637; keys norm_mean / norm_std of the model config), applied on the device in front of
 * predict_img_with_smooth_windowing by every entry point that runs it (tmat_predict_smooth, tmat_segment_batch,
 * tmat_analyze_batch*).  Off by default (the shipped unet_patch_segmentor_1.json has no such keys).
 */
int tmat_set_input_norm(tmat_handle h, int on, double norm_mean, double norm_std);

/*
 * Timing hook for bench.py's roofline line: accumulated HIP-event time (ms) and launch count of
 * the dominant kernel -- tmat::conv_mfma_kernel<128, 128, 4, 2, 3, false>, the 3x3 implicit-GEMM MFMA
 * convolution instantiation that runs 4 of the 8 transposed-conv layers -- since the last reset, measured
 * on the stream it is launched on.  flops = algorithmic FLOPs of those launches.
 */
int tmat_prof_enable(tmat_handle h, int on);
int tmat_prof_read(tmat_handle h, double *ms, int64_t *launches, double *flops, int reset);

/*
 * Test-only (no reference counterpart): fill every scratch workspace the handle owns -- activation buffers, pooled-tile strips,
 * patch_in / patch_out, the scratch block, every per-pass device buffer and pinned mirror -- with `byte_pattern` (0xFF reads as
 * NaN in f32 / f64 and -1 as int).  Constants (weights, window, resize tables) are left alone.  A call that reads a location it
 * has not written itself then fails its bit comparison deterministically (tests/test_gpu_poison.py) instead of depending on
 * what a recycled allocation holds.  Drains the handle's streams before and after.
 */
int tmat_debug_poison(tmat_handle h, int byte_pattern);

#ifdef __cplusplus
}
#endif
#endif
