"""ctypes binding of libapr_hip.so (the C ABI declared in include/apr_hip.h).

This is the same stub a maintainer of the reference would add.
torch is imported first so the library binds to the HIP runtime torch already
loaded (both export SONAME libamdhip64.so.7); torch tensors only supply device
pointers and the current stream.

There is NO CPU fallback: if the library is missing or a call fails, an
exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL so the HIP runtime is shared)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libapr_hip.so")

_p = C.c_void_p
_i32, _i64, _u64, _f32, _f64, _sz = C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_double, C.c_size_t

class PairDesc(C.Structure):
    """apr_pair_desc (include/apr_hip.h)."""
    _fields_ = [("f0", C.c_void_p), ("n0", C.c_int64), ("f1", C.c_void_p), ("n1", C.c_int64),
                ("xyz0", C.c_void_p), ("xyz1", C.c_void_p), ("seed", C.c_uint64)]


class SpconvDesc(C.Structure):
    """struct apr_spconv_desc (include/apr_hip.h)."""
    _fields_ = [("inp", C.c_void_p), ("ldi", C.c_int64), ("nbr", C.c_void_p), ("n_out", C.c_int64),
                ("K", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32), ("relu", C.c_int32),
                ("w_packed", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p),
                ("residual", C.c_void_p), ("ldr", C.c_int64), ("out", C.c_void_p), ("ldo", C.c_int64),
                ("counters", C.c_void_p), ("plist", C.c_void_p), ("prod_scratch", C.c_void_p),
                ("plist_bytes", C.c_int64), ("w_bf3", C.c_void_p),
                ("os_pairs", C.c_void_p), ("os_rows", C.c_int64), ("os_build_bytes", C.c_int64),
                ("os_n_in", C.c_int64), ("l2norm", C.c_int32), ("ws3", C.c_int32)]


class ResunetLayer(C.Structure):
    """struct apr_resunet_layer (include/apr_hip.h)."""
    _fields_ = [("K", C.c_int32), ("cin", C.c_int32), ("cout", C.c_int32), ("relu", C.c_int32),
                ("w_packed", C.c_void_p), ("w_bf3", C.c_void_p), ("scale", C.c_void_p), ("shift", C.c_void_p)]


class ResunetPlan(C.Structure):
    """struct apr_resunet_plan (include/apr_hip.h)."""
    _fields_ = [("layer", ResunetLayer * 23), ("conv1_w", C.c_void_p), ("conv1_ks", C.c_int32), ("normalize", C.c_int32),
                ("ws_conv", C.c_uint32), ("ws_block", C.c_uint32), ("os_block", C.c_uint32),
                ("ws3", C.c_int32), ("ws3_cin128", C.c_int32), ("os_min_rows", C.c_int32), ("occ_kernel_map", C.c_int32),
                ("ws3_max_rows_128", C.c_int64)]


class LevelMap(C.Structure):
    """struct apr_level_map (include/apr_hip.h)."""
    _fields_ = [("coords", C.c_void_p), ("keys", C.c_void_p), ("vals", C.c_void_p), ("cap", C.c_int64), ("n", C.c_int64)]


class Pyramid(C.Structure):
    """struct apr_pyramid (include/apr_hip.h)."""
    _fields_ = [("lv", LevelMap * 4), ("first", C.c_void_p), ("pts", C.c_void_p), ("header", C.c_void_p),
                ("header_ints", C.c_int32), ("compact", C.c_int32), ("compact_rows", C.c_int64),
                ("counters", C.c_void_p), ("n_counter_slots", C.c_int32)]


class GcnLayer(C.Structure):
    """struct apr_gcn_layer (include/apr_hip.h)."""
    _fields_ = [("kind", C.c_int32), ("k", C.c_int32), ("heads", C.c_int32),
                ("eps1", C.c_float), ("eps2", C.c_float), ("eps3", C.c_float),
                ("w1", C.c_void_p), ("w2", C.c_void_p), ("w3", C.c_void_p), ("b1", C.c_void_p), ("b2", C.c_void_p),
                ("wq", C.c_void_p), ("wk", C.c_void_p), ("wv", C.c_void_p), ("wm", C.c_void_p),
                ("bq", C.c_void_p), ("bk", C.c_void_p), ("bv", C.c_void_p), ("bm", C.c_void_p)]


class GcnDesc(C.Structure):
    """struct apr_gcn_desc (include/apr_hip.h)."""
    _fields_ = [("n_layers", C.c_int32), ("c", C.c_int32), ("layer", GcnLayer * 8)]


# name -> (restype, argtypes); every symbol include/apr_hip.h declares
PROTOTYPES = {
    "apr_col_sums": (C.c_int, [_p, _i64, _i64, _i32, _p, _p, _sz, _p]),
    "apr_mha_train_forward": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p, _p, _p]),
    "apr_mha_train_backward": (C.c_int, [_p, _p, _p, _p, _p, _i32, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _p]),
    "apr_edge_features_backward": (C.c_int, [_p, _i32, _i32, _i32, _p, _i64, _p, _p]),
    "apr_gcn_scratch_bytes": (_sz, [_p, _i32, _i32]),
    "apr_gcn_forward": (C.c_int, [_p, _p, _i32, _p, _i32, _p, _i64, _p, _i64, _p, _i64, _p, _i64, _p, _sz, _p]),
    "apr_voxel_pyramid_scratch_bytes": (_sz, [_i64, _i32]),
    "apr_voxel_pyramid": (C.c_int, [_p, _p, _i32, _f32, _p, _sz, _p, _p]),
    "apr_resunet_encode_supported": (C.c_int, [_p, _p, _p]),
    "apr_resunet_encode_scratch_bytes": (_sz, [_p, _p, _p]),
    "apr_resunet_encode": (C.c_int, [_p, _p, _p, _p, _i32, _p, _sz, _p, _i64, _p]),
    "apr_last_error": (C.c_char_p, []),
    "apr_version": (C.c_int, []),
    "apr_device_count": (C.c_int, []),
    "apr_struct_sizes": (_i32, [_p, _i32]),
    "apr_ransac_set_screen": (C.c_int, [_i32]),
    "apr_ransac_set_option": (C.c_int, [_i32, _i32]),
    "apr_ransac_sampling_launches": (C.c_int, [_p]),
    "apr_event_wait": (C.c_int, [_p, _i32]),
    "apr_event_wait_timeout": (C.c_int, [_p, _i32, _i64]),
    "apr_hash_capacity": (_i64, [_i64]),
    "apr_map_scratch_bytes": (_sz, [_i64]),
    "apr_voxelize": (C.c_int, [_p, _i64, _f32, _i32, _p, _p]),
    "apr_map_build": (C.c_int, [_p, _i64, _p, _i32, _p, _p, _i64, _p, _p, _p, _p, _p, _sz, _p]),
    "apr_kernel_map_transpose": (C.c_int, [_p, _i64, _i32, _i64, _p, _p]),
    "apr_voxelize_segments": (C.c_int, [_p, _i64, C.c_float, _p, _i32, _p, _p]),
    "apr_segment_counts": (C.c_int, [_p, _p, _p, _i32, _p, _p]),
    "apr_kernel_map_same": (C.c_int, [_p, _i64, _p, _p, _i64, _i32, _i32, _p, _p]),
    "apr_kernel_map": (C.c_int, [_p, _i64, _p, _p, _p, _i64, _i32, _i32, _p, _p]),
    "apr_spconv_packed_size": (_i64, [_i32, _i32, _i32]),
    "apr_spconv_pack_weights": (C.c_int, [_p, _i32, _i32, _i32, _p, _p]),
    "apr_spconv_fwd": (C.c_int, [_p, _i64, _p, _i64, _i32, _i32, _i32, _p, _p, _p, _p, _i64, _i32, _p, _i64, _p]),
    "apr_pairlist_counter_ints": (_i32, []),
    "apr_pairlist_bytes": (_sz, [_i64, _i32]),
    "apr_pairlist_build": (C.c_int, [_p, _i64, _i32, _p, _p, _sz, _p]),
    "apr_pairlist3_bytes": (_sz, [_i64]),
    "apr_pairlist3_build": (C.c_int, [_p, _i64, _i32, _p, _p, _sz, _p]),
    "apr_spconv_ws3_supported": (C.c_int, [_i32, _i32, _i32]),
    "apr_spconv_ws3_fwd_bf3": (C.c_int, [_p, _i64, _p, _p, _i64, _i32, _i32, _p, _p, _p, _p, _i64, _i32, _p, _i64, _p, _p]),
    "apr_spconv_ws_fwd": (C.c_int, [_p, _i64, _p, _p, _i64, _i32, _i32, _i32, _p, _p, _p, _p, _i64, _i32, _p, _i64, _p, _p]),
    "apr_spconv_packed_bf3_bytes": (_i64, [_i32, _i32, _i32]),
    "apr_spconv_pack_weights_bf3": (C.c_int, [_p, _i32, _i32, _i32, _p, _p]),
    "apr_spconv_pack_weights_bf3_ex": (C.c_int, [_p, _i32, _i32, _i32, _i32, _i32, _p, _p]),
    "apr_spconv_ws_fwd_bf3": (C.c_int, [_p, _i64, _p, _p, _i64, _i32, _i32, _i32, _p, _p, _p, _p, _p, _i64, _i32, _p, _i64,
                                        _p, _p]),
    "apr_spconv_wgrad_scratch_bytes": (_sz, [_i64, _i32, _i32, _i32]),
    "apr_spconv_wgrad": (C.c_int, [_p, _i64, _p, _i64, _p, _i64, _i32, _i32, _i32, _p, _p, _sz, _p]),
    "apr_spconv_wgrad_same_level": (C.c_int, [_p, _i64, _p, _i64, _p, _i64, _i32, _i32, _i32, _p, _p, _sz, _p]),
    "apr_spconv_fwd_batch": (C.c_int, [_p, _i32, _p]),
    "apr_spconv_fwd_batch_timed": (C.c_int, [_p, _i32, _p, _p]),
    "apr_bn_stats": (C.c_int, [_p, _i64, _i64, _i32, _p, _p, _p, _sz, _p]),
    "apr_norm_params": (C.c_int, [_p, _i64, _i64, _i32, _f32, _p, _p, _p, _sz, _p]),
    "apr_bn_stats_scratch_bytes": (_sz, [_i64, _i32]),
    "apr_instance_norm_act": (C.c_int, [_p, _i64, _i64, _i32, _f32, _p, _i64, _i32, _f32, _p, _i64, _p, _sz, _p]),
    "apr_spconv_os_tile_rows": (_i32, [_i64, _i32, _i32]),
    "apr_spconv_os_pairs_bytes": (_sz, [_i64, _i32, _i32]),
    "apr_spconv_os_pairs_build": (C.c_int, [_p, _i64, _i64, _i32, _i32, _p, _sz, _p]),
    "apr_spconv_os_trace": (C.c_int, [_p, _i32]),
    "apr_spconv_os_set_debug": (C.c_int, [_i32]),
    "apr_spconv_os_fwd": (C.c_int, [_p, _i64, _p, _i64, _i32, _i32, _i32, _i32, _p, _p, _p, _p, _i64, _i32, _p, _i64, _p]),
    "apr_norm_backward_scratch_bytes": (_sz, [_i64, _i32]),
    "apr_norm_backward": (C.c_int, [_p, _i64, _p, _i64, _i64, _i32, _p, _p, _p, _p, _i64, _p, _p, _p, _sz, _p]),
    "apr_bn_train_fwd": (C.c_int, [_p, _i64, _i64, _i32, _p, _p, C.c_float, C.c_float, _p, _p, _p, _i64, _i32, _p, _i64, _p, _p,
                                   _p, _p, _i32, _p, _sz, _p]),
    "apr_bn_train_bwd": (C.c_int, [_p, _i64, _p, _i64, _p, _i64, _i64, _i32, _p, _p, _p, _i32, _p, _i64, _p, _i64, _p, _p,
                                   _p, _i32, _p, _sz, _p]),
    "apr_weights_flip_transpose": (C.c_int, [_p, _i32, _i32, _i32, _i32, _p, _p]),
    "apr_dense_gemm_bf3": (C.c_int, [_p, _i64, _i64, _i32, _i32, _p, _p, _p, _p, _i64, _i32, _p, _i64, _p]),
    "apr_dense_rows_bf3_ok": (_i32, [_i32, _i32]),
    "apr_dense_rows_bf3_route": (_i32, [_i64, _i32, _i32]),
    "apr_dense_rows_bf3": (C.c_int, [_p, _i64, _i64, _i32, _i32, _p, _p, _p, _p, _i64, _i32, _i32, _p, _i64, _p]),
    "apr_dense_gemm_bf3_norm_scratch_bytes": (_sz, [_i64, _i32, _i32]),
    "apr_dense_gemm_bf3_norm_act": (C.c_int, [_p, _i64, _i64, _i32, _i32, _p, _f32, _p, _i64, _i32, _f32, _p, _i64, _p, _i32, _p,
                                              _sz, _p]),
    "apr_weighted_choice_round": (C.c_int64, [_p, _i64, _p, _i64, _i64, _p, _i64, _p, _p, _i32]),
    "apr_instance_norm_act_seg": (C.c_int, [_p, _i64, _i64, _i32, _f32, _p, _i64, _i32, _f32, _p, _i64, _p, _i32, _p, _sz, _p]),
    "apr_affine_act": (C.c_int, [_p, _i64, _i64, _i32, _p, _p, _p, _i64, _i32, _f32, _p, _i64, _p]),
    "apr_act_backward": (C.c_int, [_p, _i64, _p, _i64, _i64, _i32, _i32, _f32, _p, _i64, _p]),
    "apr_l2_normalize": (C.c_int, [_p, _i64, _i64, _i32, _p, _i64, _p]),
    "apr_l2_normalize_backward": (C.c_int, [_p, _i64, _p, _i64, _i64, _i32, _p, _i64, _p]),
    "apr_feature_nn": (C.c_int, [_p, _i64, _p, _i64, _i32, _p, _p]),
    "apr_feature_nn_fast_scratch_bytes": (_sz, [_i64, _i64, _i32]),
    "apr_feature_nn_fast": (C.c_int, [_p, _i64, _p, _i64, _i32, _p, _p, _sz, _p]),
    "apr_nn_unpack": (C.c_int, [_p, _i64, _p, _p, _p]),
    "apr_ransac_scratch_bytes": (_sz, [_i64, _i64]),
    "apr_ransac_pose": (C.c_int, [_p, _i64, _p, _i64, _p, _f64, _f64, _i64, _u64, _p, _sz, _p, _p]),
    "apr_ransac_geometric_scratch_bytes": (_sz, [_i64, _i64, _i64]),
    "apr_ransac_pose_geometric": (C.c_int, [_p, _i64, _p, _i64, _p, _f64, _f64, _i64, _i64, _u64, _p, _sz, _p, _p]),
    "apr_ransac_raw_bytes": (_sz, []),
    "apr_ransac_pose_geometric_async": (C.c_int, [_p, _i64, _p, _i64, _p, _f64, _f64, _i64, _i64, _u64, _p, _sz, _p, _p]),
    "apr_ransac_decode": (C.c_int, [_p, _p]),
    "apr_irls_pose": (C.c_int, [_p, _p, _p, _i64, _p, _p, _sz, _p]),
    "apr_match_pose_batch_scratch_bytes": (_sz, [_i32, _i64, _i64, _i32, _i64]),
    "apr_match_pose_batch": (C.c_int, [_p, _i32, _i32, C.c_double, C.c_double, _i64, _p, _sz, _p, _p]),
    "apr_match_pose_batch_slot_bytes": (_sz, [_i32]),
    "apr_match_pose_set_lanes": (C.c_int, [_i32]),
    "apr_match_pose_batch_enqueue": (C.c_int, [_p, _i32, _i32, C.c_double, C.c_double, _i64, _p, _sz, _p, _p]),
    "apr_match_pose_batch_finish": (C.c_int, [_p, _i32, _i32, C.c_double, C.c_double, _i64, _p, _sz, _p, _p, _p]),
    "apr_irls_scratch_bytes": (_sz, [_i64]),
    "apr_contrastive_reduce": (C.c_int, [_p, _p, _i32, _i32, _p, _p, _p, _p, _p, _p, _p, _i32, _i64, _f32, _f32, _p, _p]),
    "apr_grid_subsample_scratch_bytes": (_sz, [_i64]),
    "apr_grid_subsample": (C.c_int, [_p, _i64, _p, _i32, _f32, _p, _i32, _p, _p, _p, _p, _sz, _p]),
    "apr_grid_subsample_async": (C.c_int, [_p, _i64, _p, _i32, _f32, _p, _i32, _p, _p, _p, _p, _sz, _p]),
    "apr_radius_scratch_bytes": (_sz, [_i64, _i64]),
    "apr_radius_neighbors_async": (C.c_int, [_p, _i64, _p, _i64, _p, _p, _i32, _f32, _i32, _p, _i64, _p, _p, _sz, _p]),
    "apr_radius_neighbors_regrid_async": (C.c_int, [_p, _i64, _p, _i64, _p, _p, _i32, _f32, _i32, _p, _i64, _p, _p, _sz, _p]),
    "apr_radius_neighbors": (C.c_int, [_p, _i64, _p, _i64, _p, _p, _i32, _f32, _i32, _p, _i64, _p, _p, _sz, _p]),
    "apr_knn": (C.c_int, [_p, _i32, _i32, _i32, _p, _p]),
    "apr_row_sums": (C.c_int, [_p, _i64, _i64, _i32, _p, _p]),
    "apr_kpconv_weighted": (C.c_int, [_p, _i64, _p, _i64, _p, _i32, _p, _i64, _i32, _p, _i32, _f32, _p, _p, _i64, _p]),
    "apr_kpconv_dfeat": (C.c_int, [_p, _i64, _p, _i64, _p, _i32, _p, _i64, _i32, _p, _i32, _f32, _p, _p, _i64, _p]),
    "apr_kpconv_dfeat_contrib": (C.c_int, [_p, _i64, _p, _i64, _p, _i32, _p, _i64, _i32, _p, _i32, _f32, _p, _p, _p]),
    "apr_reverse_table_scratch_bytes": (_sz, [_i64, _i32, _i64]),
    "apr_reverse_table_build": (C.c_int, [_p, _i64, _i32, _i64, _p, _p, _p, _sz, _p]),
    "apr_reverse_gather": (C.c_int, [_p, _i32, _p, _p, _i64, _p, _i64, _p]),
    "apr_reverse_gather_range": (C.c_int, [_p, _i32, _p, _p, _i64, _i64, _i64, _i32, _p, _i64, _p]),
    "apr_gather_pool": (C.c_int, [_p, _i64, _i64, _i32, _p, _i32, _i64, _i32, _p, _i64, _p]),
    "apr_gather_pool_argmax": (C.c_int, [_p, _i64, _i64, _i32, _p, _i32, _i64, _p, _i64, _p, _p]),
    "apr_gather_pool_backward": (C.c_int, [_p, _i64, _i32, _p, _p, _i64, _i32, _p, _i32, _p, _i64, _p]),
    "apr_edge_features": (C.c_int, [_p, _i64, _i32, _i32, _p, _i32, _p, _p]),
    "apr_group_max": (C.c_int, [_p, _i64, _i32, _i32, _i32, _p, _p, _f32, _p, _i64, _p]),
    "apr_coords_bbox": (C.c_int, [_p, _i64, _p, _p]),
    "apr_kp_resnet_scratch_bytes": (_sz, [_p]),
    "apr_kp_resnet_block": (C.c_int, [_p, _p]),
    "apr_voxelize_frames": (C.c_int, [_p, _p, _i32, _f32, _p, _p, _p]),
    "apr_gather_frame_points": (C.c_int, [_p, _p, _i32, _p, _p, _i64, _p, _p]),
    "apr_pack_i32": (C.c_int, [_p, _p, _i32, _p, _p, _i64, _p]),
    "apr_fill_bytes": (C.c_int, [_p, _i32, _sz, _p]),
    "apr_kernel_map_transpose_prefilled": (C.c_int, [_p, _i64, _i32, _i64, _p, _p]),
    "apr_occ_conv_scratch_bytes": (_sz, [_p, _i32]),
    "apr_occ_conv_pays": (C.c_int, [_p, _i32, _i64]),
    "apr_occ_conv": (C.c_int, [_p, _i64, _p, _i32, _p, _i32, _p, _p, _p, _i64, _i32, _p, _i64, _p, _sz, _p]),
    "apr_kernel_map_occ": (C.c_int, [_p, _i64, _p, _p, _p, _i64, _i32, _i32, _p, _i32, _p, _p, _p]),
    "apr_mha": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _i32, _p, _p]),
    "apr_mha_headmajor": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _i32, _p, _p]),
    "apr_softmax_matvec": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _f32, _p, _p]),
    "apr_softmax_matvec_bt": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _f32, _p, _p]),
    "apr_softmax_matvec_mfma": (C.c_int, [_p, _p, _p, _i32, _i32, _i32, _f32, _p, _p]),
    "apr_score_head": (C.c_int, [_p, _i64, _i64, _p, _p]),
    "apr_transform_points": (C.c_int, [_p, _i64, _p, _p, _p]),
    "apr_crop_scratch_bytes": (_sz, [_i64]),
    "apr_crop_to_radius": (C.c_int, [_p, _i64, _p, _i64, _p, _p, _p, _sz, _p]),
    "apr_chamfer_sum": (C.c_int, [_p, _i64, _p, _i64, _p, _p, _sz, _p]),
    "apr_nn3_scratch_bytes": (_sz, [_i64, _i64]),
    "apr_nn3": (C.c_int, [_p, _i64, _p, _i64, C.c_float, _p, _p, _p, _sz, _p]),
    "apr_nn3_batch": (C.c_int, [_p, _p, _p, _p, _i32, C.c_float, _p, _p, _p, _p, _sz, _p]),
}

class KpResnetDesc(C.Structure):
    """apr_kp_resnet_desc (include/apr_hip.h)."""
    _fields_ = [("x", _p), ("ldx", _i64), ("n_in", _i64),
                ("in_dim", _i32), ("mid", _i32), ("out_dim", _i32), ("strided", _i32),
                ("q_pts", _p), ("s_pts", _p), ("n_out", _i64),
                ("nbr", _p), ("H", _i32), ("n_kp", _i32),
                ("kernel_points", _p), ("extent", _f32), ("eps", _f32), ("slope", _f32), ("nseg", _i32),
                ("w_unary1", _p), ("w_kpconv", _p), ("w_unary2", _p), ("w_shortcut", _p),
                ("seg_in", _p), ("seg_out", _p),
                ("out", _p), ("ldo", _i64),
                ("scratch", _p), ("scratch_bytes", _sz)]


_lib = None


class AprHipError(RuntimeError):
    pass


def load():
    """Load libapr_hip.so; raises if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AprHipError(
            f"{LIB_PATH} is missing: build it with `python -m apr_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        msg = load().apr_last_error().decode()
        raise AprHipError(f"libapr_hip error {rc}: {msg}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


def stream():
    """Raw hipStream_t of torch's current stream (fast path: no Python Stream object)."""
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch.cuda.current_device()))


def require_gpu_tensor(t, dtype, name):
    if not t.is_cuda:
        raise AprHipError(f"{name}: expected a GPU tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise AprHipError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise AprHipError(f"{name}: tensor must be contiguous")
    return t
