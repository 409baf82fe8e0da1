"""tmat_amd -- MI355X-native drop-in for the 2-D microvessel-branching hot path of
fogg-lab/tissue-model-analysis-tools (scripts/compute_branches.py and the fl_tissue_model_tools
functions it drives).  Host side: Python mirroring the reference's module / function names;
compute: libtmat_hip.so (hand-written HIP for gfx950 + C++ host graph stages) through ctypes.

    models.UNetXceptionPatchSegmentor / get_unet_patch_segmentor_from_cfg   (reference models.py:597-684)
    smooth_tiled_predictions.predict_img_with_smooth_windowing               (smooth_tiled_predictions.py:220)
    transforms.filter_branch_seg_mask                                        (transforms.py:306)
    dmtgraph.compute_dmt_graph                                               (dmtgraph.py:38)
    topology.MorseGraph                                                      (topology.py:15)
    branches.analyze_batch                                                   (compute_branches.py:144, batched)
"""
__version__ = "0.1.0"
