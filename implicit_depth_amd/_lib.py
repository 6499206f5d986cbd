"""ctypes binding of liblidf_hip.so — the C ABI declared in include/lidf_hip.h.

This module is the only place that touches the shared library. It never falls back to a CPU
implementation: if the library is missing, `lib()` raises, and every op raises on non-CUDA input
(mirroring the reference extension's CHECK_CUDA, extensions/ray_aabb/ray_aabb_cuda.cpp:16-18).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LIDF_HIP_LIB") or os.path.join(_HERE, "csrc", "liblidf_hip.so")

LIDF_OK = 0
# The ABI this binding is written against: LIDF_ABI_VERSION of include/lidf_hip.h (the struct mirrors
# below follow that header's layouts; tests/test_host.py compares both with gcc's view of the header).
# lib() refuses a liblidf_hip.so that answers another number — the library is git-ignored and travels
# outside history, so a stale build must fail loudly, not be driven with wrong struct offsets.
ABI = 12


class LidfDecoder(C.Structure):
    """struct LidfDecoder (include/lidf_hip.h)."""
    _fields_ = [
        ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p),
        ("w3", C.c_void_p), ("b3", C.c_void_p), ("w4", C.c_void_p), ("b4", C.c_void_p),
        ("wenc", C.c_void_p), ("benc", C.c_void_p),
        ("is_ief", C.c_int32), ("n_iter", C.c_int32), ("init_offset", C.c_float),
        ("use_sigmoid", C.c_int32),
    ]


class LidfQueryArgs(C.Structure):
    """struct LidfQueryArgs (include/lidf_hip.h)."""
    _fields_ = [
        ("n_rays", C.c_int64), ("ray_dir", C.c_void_p), ("ray_pix", C.c_void_p),
        ("ray_bid", C.c_void_p), ("ray_flat", C.c_void_p),
        ("n_pairs", C.c_int64), ("pair_off", C.c_void_p), ("pair_ray", C.c_void_p),
        ("pair_vox", C.c_void_p), ("pair_t", C.c_void_p),
        ("batch", C.c_int32), ("height", C.c_int32), ("width", C.c_int32),
        ("feat_grid", C.c_void_p),
        ("n_vox", C.c_int64), ("vox_feat", C.c_void_p), ("vox_center", C.c_void_p),
        ("prob", C.POINTER(LidfDecoder)), ("off", C.POINTER(LidfDecoder)),
        ("multires", C.c_int32), ("multires_views", C.c_int32), ("roi_inp_bbox", C.c_int32),
        ("pos_rel", C.c_int32),
        ("offset_range0", C.c_float), ("offset_range1", C.c_float), ("part_size", C.c_float),
        ("pred_offset", C.c_void_p), ("pred_prob", C.c_void_p), ("pair_pred_pos", C.c_void_p),
        ("pred_prob_softmax", C.c_void_p), ("max_pair_id", C.c_void_p), ("pred_pos", C.c_void_p),
        ("depth", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("rayfeat_out", C.c_void_p),
        ("precision", C.c_int32), ("packed", C.c_void_p),
        ("offsets_selected", C.c_int32),
    ]


class LidfDecoderGrads(C.Structure):
    """struct LidfDecoderGrads (include/lidf_hip.h)."""
    _fields_ = [(k, C.c_void_p) for k in ("w1", "b1", "w2", "b2", "w3", "b3", "w4", "b4", "wenc", "benc")]


class LidfQueryTrainArgs(C.Structure):
    """struct LidfQueryTrainArgs (include/lidf_hip.h)."""
    _fields_ = [
        ("n_pairs", C.c_int64), ("n_rays", C.c_int64), ("n_vox", C.c_int64),
        ("pair_off", C.c_void_p), ("pair_ray", C.c_void_p), ("pair_vox", C.c_void_p),
        ("pe", C.c_void_p), ("multires", C.c_int32), ("multires_views", C.c_int32),
        ("vox_feat", C.c_void_p), ("rayfeat", C.c_void_p), ("dec", C.POINTER(LidfDecoder)),
    ]


class LidfPointNet(C.Structure):
    """struct LidfPointNet (include/lidf_hip.h)."""
    _fields_ = [(n, C.c_void_p) for n in (
        "w_p1", "b_p1", "w_p2", "b_p2", "w_v1", "b_v1", "w_p3", "b_p3", "w_p4", "b_p4", "w_v2", "b_v2",
        "packed")]


class LidfPointNetGrads(C.Structure):
    """struct LidfPointNetGrads (include/lidf_hip.h)."""
    _fields_ = [(n, C.c_void_p) for n in (
        "w_p1", "b_p1", "w_p2", "b_p2", "w_v1", "b_v1", "w_p3", "b_p3", "w_p4", "b_p4", "w_v2", "b_v2")]


class LidfRefineArgs(C.Structure):
    """struct LidfRefineArgs (include/lidf_hip.h)."""
    _fields_ = [
        ("n_rays", C.c_int64), ("ray_dir", C.c_void_p), ("ray_bid", C.c_void_p),
        ("ray_flat", C.c_void_p), ("pred_pos", C.c_void_p), ("max_pair_id", C.c_void_p),
        ("pair_vox", C.c_void_p), ("n_pairs", C.c_int64), ("n_vox", C.c_int64),
        ("voxel_bound", C.c_void_p), ("voxel_bid", C.c_void_p), ("rgb_img", C.c_void_p),
        ("batch", C.c_int32), ("height", C.c_int32), ("width", C.c_int32),
        ("rayfeat", C.c_void_p), ("n_valid", C.c_int64), ("valid_inp", C.c_void_p),
        ("valid_vox", C.c_void_p), ("pnet", C.POINTER(LidfPointNet)),
        ("off", C.POINTER(LidfDecoder)), ("multires", C.c_int32), ("multires_views", C.c_int32),
        ("pos_rel", C.c_int32), ("pnet_pos_rel", C.c_int32),
        ("offset_range0", C.c_float), ("offset_range1", C.c_float),
        ("pred_pos_out", C.c_void_p), ("end_voxel_id", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("precision", C.c_int32), ("pnet_select", C.c_void_p), ("packed", C.c_void_p),
        ("ray_l1", C.c_void_p), ("ray_l1_ready", C.c_int32),
        ("voxel_coord", C.c_void_p), ("grid_res", C.c_int32 * 3), ("grid_xmin", C.c_float * 3),
        ("grid_part", C.c_float), ("cell_table", C.c_void_p), ("cell_table_ready", C.c_int32),
    ]


class LidfFrameArgs(C.Structure):
    """struct LidfFrameArgs (include/lidf_hip.h)."""
    _fields_ = [
        ("batch", C.c_int32), ("height", C.c_int32), ("width", C.c_int32),
        ("rgb", C.c_void_p), ("xyz_corrupt", C.c_void_p), ("valid_mask", C.c_void_p),
        ("miss_mask", C.c_void_p), ("intr", C.c_void_p), ("feat_grid", C.c_void_p),
        ("xmin", C.c_float * 3), ("res", C.c_int32 * 3), ("part_size", C.c_float),
        ("valid_stride", C.c_int32),
        ("pnet", C.POINTER(LidfPointNet)), ("prob", C.POINTER(LidfDecoder)), ("off", C.POINTER(LidfDecoder)),
        ("packed_query", C.c_void_p),
        ("multires", C.c_int32), ("multires_views", C.c_int32), ("roi_inp_bbox", C.c_int32),
        ("pos_rel", C.c_int32), ("offset_range0", C.c_float), ("offset_range1", C.c_float),
        ("refine_times", C.c_int32), ("pnet_refine", C.POINTER(LidfPointNet)),
        ("off_refine", C.POINTER(LidfDecoder)), ("packed_refine", C.c_void_p),
        ("refine_pos_rel", C.c_int32), ("refine_pnet_pos_rel", C.c_int32),
        ("refine_use_all_pix", C.c_int32),
        ("refine_offset_range0", C.c_float), ("refine_offset_range1", C.c_float),
        ("precision", C.c_int32),
        ("max_pairs", C.c_int64), ("lds_voxels", C.c_int32),
        ("counts", C.c_void_p),
        ("valid_bid", C.c_void_p), ("valid_flat", C.c_void_p), ("valid_xyz", C.c_void_p),
        ("valid_rgb", C.c_void_p), ("occ_bid_coord", C.c_void_p), ("voxel_bound", C.c_void_p),
        ("valid_v_pid", C.c_void_p), ("revidx", C.c_void_p), ("valid_v_rel_coord", C.c_void_p),
        ("pnet_inp", C.c_void_p), ("occ_voxel_feat", C.c_void_p),
        ("ray_bid", C.c_void_p), ("ray_flat", C.c_void_p), ("ray_pix", C.c_void_p),
        ("ray_dir", C.c_void_p), ("pair_off", C.c_void_p), ("pair_ray", C.c_void_p),
        ("pair_vox", C.c_void_p), ("pair_t", C.c_void_p),
        ("pred_offset", C.c_void_p), ("pred_prob", C.c_void_p), ("pred_prob_softmax", C.c_void_p),
        ("pair_pred_pos", C.c_void_p), ("max_pair_id", C.c_void_p), ("pred_pos", C.c_void_p),
        ("rayfeat", C.c_void_p), ("pred_depth", C.c_void_p), ("pred_pos_refine", C.c_void_p),
        ("end_voxel_id", C.c_void_p), ("pred_depth_refine", C.c_void_p),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
        ("valid_idx_bid", C.c_void_p), ("valid_idx_flat", C.c_void_p), ("n_valid_idx", C.c_int64),
        ("pack_blob", C.c_void_p), ("pack_blob_bytes", C.c_size_t), ("pack_guard", C.c_void_p),
        ("pack_mode", C.c_int32), ("offsets_selected", C.c_int32),
        ("aux_stream", C.c_void_p), ("ev_fork", C.c_void_p), ("ev_join", C.c_void_p),
        ("fail_after", C.c_int32),
        ("profile_events", C.POINTER(C.c_void_p)),
    ]


_P, _I64, _I, _SZ = C.c_void_p, C.c_int64, C.c_int, C.c_size_t

# name -> (restype, argtypes); every symbol include/lidf_hip.h declares
SIGNATURES = {
    "lidf_version": (C.c_int, []),
    "lidf_strerror": (C.c_char_p, [C.c_int]),
    "lidf_embed_f32": (C.c_int, [_P, _I64, _I, _P, _P]),
    "lidf_decoders_workspace_bytes": (_SZ, [_I64, _I]),
    "lidf_decoders_f32": (C.c_int, [_P, _I64, _I, _I64, C.POINTER(LidfDecoder),
                                    C.POINTER(LidfDecoder), _P, _P, _P, _SZ, _P]),
    "lidf_decoders_split_f32": (C.c_int, [_P, _I64, _I, _I64, C.POINTER(LidfDecoder),
                                    C.POINTER(LidfDecoder), _P, _P, _P, _SZ, _P]),
    "lidf_query_workspace_bytes": (_SZ, [_I64, _I64, _I64]),
    "lidf_query_pack_bytes": (_SZ, []),
    "lidf_query_pack_f32": (C.c_int, [C.POINTER(LidfDecoder), C.POINTER(LidfDecoder), _I, _I, _I, _P, _SZ, _P]),
    "lidf_pack_guard_bytes": (_SZ, []),
    "lidf_query_pack_guarded_f32": (C.c_int, [C.POINTER(LidfDecoder), C.POINTER(LidfDecoder), _I, _I, _I, _P,
                                              _SZ, _P, _P]),
    "lidf_pointnet_pack_guarded_f32": (C.c_int, [C.POINTER(LidfPointNet), _P, _SZ, _P, _P]),
    "lidf_refine_pack_guarded_f32": (C.c_int, [C.POINTER(LidfDecoder), _I, _I, _P, _SZ, _P, _P]),
    "lidf_query_f32": (C.c_int, [C.POINTER(LidfQueryArgs), _P]),
    "lidf_query_profile_f32": (C.c_int, [C.POINTER(LidfQueryArgs), _P, _P, _P]),
    "lidf_ray_features_workspace_bytes": (_SZ, [_I, _I, _I, _I64]),
    "lidf_ray_features_f32": (C.c_int, [_P, _I, _I, _I, _P, _P, _P, _I64, _I, _I, _P, _P, _SZ, _P]),
    "lidf_ray_reduce_f32": (C.c_int, [_P, _P, _P, _I64, _I64, _P, _P, _I64, _P, _P, _P, _P, _P]),
    "lidf_ray_dirs_f32": (C.c_int, [_P, _I, _I, _I, _P, _P]),
    "lidf_miss_ray_workspace_bytes": (_SZ, [_I64]),
    "lidf_miss_ray_count": (C.c_int, [_P, _I, _I64, _P, _P, _SZ, _P]),
    "lidf_miss_ray_fill_f32": (C.c_int, [_P, _I, _P, _I, _I, _I, _P, _SZ, _P, _P, _P, _P, _P, _P, _P, _P]),
    "lidf_ray_aabb_dense_f32": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _P, _P, _P]),
    "lidf_ray_aabb_count_f32": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _P, _P]),
    "lidf_ray_aabb_fill_f32": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _P, _P, _P, _P, _P]),
    "lidf_ray_aabb_grid_workspace_bytes": (_SZ, [_I, _I, _I, _I]),
    "lidf_ray_aabb_grid_build_f32": (C.c_int, [_P, _P, _P, _I64, _I, _I, _I, _I, _P, _SZ, _P]),
    "lidf_ray_aabb_grid_count_f32": (C.c_int, [_P, _P, _I64, _I, _I, _I, _I, _P, _SZ, _P, _P]),
    "lidf_ray_aabb_grid_fill_f32": (C.c_int, [_P, _P, _I64, _I, _I, _I, _I, _P, _SZ, _P, _P, _P, _P, _P]),
    "lidf_exclusive_scan_workspace_bytes": (_SZ, [_I64]),
    "lidf_exclusive_scan_i32": (C.c_int, [_P, _I64, _P, _P, _SZ, _P]),
    "lidf_pcl_aabb_dense_f32": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _P, _P]),
    "lidf_pcl_aabb_last_f32": (C.c_int, [_P, _P, _P, _P, _I64, _I64, _P, _P]),
    "lidf_voxelize_workspace_bytes": (_SZ, [_I64, _I64]),
    "lidf_voxelize_f32": (C.c_int, [_P, _P, _I64, _I, C.POINTER(C.c_float), C.POINTER(C.c_int32),
                                    C.c_float, _P, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    "lidf_pointnet_workspace_bytes": (_SZ, [_I64, _I64]),
    "lidf_pointnet_pack_bytes": (_SZ, []),
    "lidf_pointnet_pack_f32": (C.c_int, [C.POINTER(LidfPointNet), _P, _SZ, _P]),
    "lidf_pointnet_f32": (C.c_int, [C.POINTER(LidfPointNet), _P, _P, _I64, _I64, _P, _P, _SZ, _P]),
    "lidf_refine_workspace_bytes": (_SZ, [_I64, _I64, _I64]),
    "lidf_refine_f32": (C.c_int, [C.POINTER(LidfRefineArgs), _P]),
    "lidf_refine_profile_f32": (C.c_int, [C.POINTER(LidfRefineArgs), _P, _P, _P, _P, _P]),
    "lidf_refine_pack_bytes": (_SZ, [_I, _I]),
    "lidf_refine_pack_f32": (C.c_int, [C.POINTER(LidfDecoder), _I, _I, _P, _SZ, _P]),
    "lidf_frame_workspace_bytes": (_SZ, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), _I64,
                                         C.c_int32, C.c_int32]),
    "lidf_frame_f32": (C.c_int, [C.POINTER(LidfFrameArgs), _P]),
    "lidf_event_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "lidf_event_destroy": (C.c_int, [_P]),
    "lidf_frame_pack_bytes": (_SZ, []),
    "lidf_frame_pack_guard_bytes": (_SZ, []),
    "lidf_depth_metrics_workspace_bytes": (_SZ, []),
    "lidf_depth_metrics_f32": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _SZ, _P]),
    "lidf_build_rows_f32": (C.c_int, [_P, _P, _P, _P, _P, _I, _P, _P, _I, _I, _I64, _P, _P]),
    "lidf_rows_backward_f32": (C.c_int, [_P, _P, _P, _I64, _I64, _I64, _I, _I, _P, _P, _P]),
    "lidf_ray_features_backward_f32": (C.c_int, [_P, _P, _P, _I64, _I, _I, _I, _I, _I, _P, _P, C.c_size_t, _P]),
    "lidf_pe_rows_f32": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I64, _P, _P]),
    "lidf_query_decoder_act_floats": (C.c_size_t, [_I64, _I64, _I64, _I]),
    "lidf_query_decoder_workspace_bytes": (C.c_size_t, [_I64, _I64, _I64]),
    "lidf_query_decoder_forward_train_f32": (C.c_int, [C.POINTER(LidfQueryTrainArgs), _P, _P, _P,
                                                       C.c_size_t, _P]),
    "lidf_query_forward_train_workspace_bytes": (C.c_size_t, [_I64, _I64]),
    "lidf_query_forward_train_f32": (C.c_int, [C.POINTER(LidfQueryTrainArgs), C.POINTER(LidfDecoder), _P, _P,
                                               _P, _I, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "lidf_query_decoder_backward_f32": (C.c_int, [C.POINTER(LidfQueryTrainArgs), _P, _P, _P, _P, _I,
                                                  C.POINTER(LidfDecoderGrads), _P, C.c_size_t, _P]),
    "lidf_query_decoder_rows_workspace_bytes": (_SZ, [_I64, _I64, C.c_int32, C.c_int32]),
    "lidf_query_decoder_backward_rows_f32": (C.c_int, [C.POINTER(LidfQueryTrainArgs), _P, C.c_int32, _P, _P, _P, _P,
                                                       C.c_float, _P, _P, _I, C.POINTER(LidfDecoderGrads), _P,
                                                       C.c_size_t, _P]),
    "lidf_query_forward_train_selected_f32": (C.c_int, [C.POINTER(LidfQueryTrainArgs), C.POINTER(LidfDecoder), _P, _P,
                                                        _P, C.c_int32, C.c_float, C.c_float, C.c_float, _P, _P, _P, _P,
                                                        _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "lidf_embed_backward_f32": (C.c_int, [_P, _P, _I64, _I, _P, _P]),
    "lidf_pointnet_train_act_floats": (_SZ, [_I64, _I64]),
    "lidf_pointnet_train_workspace_bytes": (_SZ, [_I64, _I64]),
    "lidf_pointnet_forward_train_f32": (C.c_int, [C.POINTER(LidfPointNet), _P, _P, _I64, _I64, _P, _P, _P, _SZ, _P]),
    "lidf_pointnet_backward_f32": (C.c_int, [C.POINTER(LidfPointNet), _P, _P, _I64, _I64, _P, _P, _P,
                                             C.POINTER(LidfPointNetGrads), _P, _SZ, _P]),
    "lidf_refine_train_act_bytes": (_SZ, [_I64, _I64, _I64, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "lidf_refine_train_workspace_bytes": (_SZ, [_I64, _I64, _I64, C.c_int32]),
    "lidf_refine_train_forward_f32": (C.c_int, [C.POINTER(LidfRefineArgs), C.c_int32, _P, _SZ, _P]),
    "lidf_refine_train_backward_f32": (C.c_int, [C.POINTER(LidfRefineArgs), C.c_int32, _P, _SZ, _P, _P, _P,
                                                 C.POINTER(LidfPointNetGrads), C.POINTER(LidfDecoderGrads), _P]),
    "lidf_query_tail_f32": (C.c_int, [_P, _P, _P, _P, _P, _P, _I64, _I64, C.c_float, C.c_float, C.c_float,
                                      _P, _P, _P, _P, _P, _P]),
    "lidf_query_tail_backward_f32": (C.c_int, [_P, _P, _P, _P, _P, _I64, _I64, C.c_float, C.c_float,
                                               C.c_float, _P, _P]),
    "lidf_roi_align_f32": (_I, [_P, _I, _I, _I, _I, _P, _P, _I64, _I, _I, _P, _I64, _P]),
    "lidf_wgrad_workspace_bytes": (_SZ, []),
    "lidf_wgrad_f32": (_I, [_P, _I64, _I, _P, _I64, _I, _I64, _P, _I64, _P, _P, _SZ, _P]),
    "lidf_linear_workspace_bytes": (_SZ, [_I]),
    "lidf_linear_f32": (_I, [_P, _I64, _I64, _I, _P, _I64, _P, _I, _I, C.c_float, _P, _P, _I64, _P, _I64, _P, _P,
                            _I64, _P, _SZ, _P]),
    "lidf_linear_gather2_f32": (_I, [_P, _I64, _I64, _I, _P, _I64, _P, _I, _I, C.c_float, _P, _P, _I64, _P, _P, _I64,
                                    _P, _I64, _P, _SZ, _P]),
    "lidf_decoder_train_act_floats": (C.c_size_t, [_I64, _I]),
    "lidf_decoder_train_workspace_bytes": (C.c_size_t, [_I64, _I]),
    "lidf_decoder_forward_train_f32": (C.c_int, [_P, _I64, _I, _I64, C.POINTER(LidfDecoder), _P, _P,
                                                 _P, C.c_size_t, _P]),
    "lidf_decoder_backward_f32": (C.c_int, [_P, _I64, _I, _I64, C.POINTER(LidfDecoder), _P, _P, _P,
                                            _I64, C.POINTER(LidfDecoderGrads), _P, C.c_size_t, _P]),
    "lidf_decoder_chain_workspace_bytes": (C.c_size_t, [_I, _I]),
    "lidf_decoder_chain_f32": (C.c_int, [C.POINTER(LidfDecoder), _I, _I, _P, _I64, _I, _I, _I64, _P, _P, _P, _P, _P,
                                         _I, _P, C.c_size_t, _P]),
    "lidf_decoder_pair_workspace_bytes": (C.c_size_t, [_I64, _I]),
    "lidf_decoder_pair_workspace_offset": (C.c_size_t, [_I64, _I, _I]),
    "lidf_decoder_pair_backward_f32": (C.c_int, [_P, _I64, _I, _I64, C.POINTER(LidfDecoder), C.POINTER(LidfDecoder),
                                                 _P, _P, _P, _P, _P, _I64, C.POINTER(LidfDecoderGrads),
                                                 C.POINTER(LidfDecoderGrads), _P, C.c_size_t, _P]),
}

_lib = None

# Packed weight streams per module (query.py / pointnet.py): kept beside the modules, not on them
# — an entry holds a device blob and a torch.cuda.Event, which must not travel with
# copy.deepcopy(module) or pickling of the module — and dropped with the module.
import weakref  # noqa: E402
PACK_CACHE = weakref.WeakKeyDictionary()
PACK_CACHE_REFINE = weakref.WeakKeyDictionary()   # stage-2 IEF (lidf_refine_pack_f32), keyed by the module
FROZEN = weakref.WeakSet()                        # modules whose packed streams the caller froze


MAX_STREAM_ENTRIES = 16   # packed entries kept per module (one per stream that used it)


class PackedEntry:
    """One packed blob with its device-side guard (lidf_*_pack_guarded_f32). `key` holds only
    host-side configuration (never a parameter version: torch's version counter misses `p.data`
    writes — the guard compares the parameters' CONTENTS on the device) and the stream the entry
    belongs to: a blob and its guard are only ever touched by launches of ONE stream, so calls on
    different streams (frames pipelined over two streams) share nothing and need no cross-stream
    ordering — each stream keeps its own ~1 MB copy of a module's streams."""
    __slots__ = ("key", "blob", "guard", "frozen_ready")

    def __init__(self, key, blob, guard):
        self.key, self.blob, self.guard, self.frozen_ready = key, blob, guard, False


def packed_entry(cache, owner, key, nbytes, device):
    """The entry of `owner` in `cache` for (`key`, current stream), created with a zero-filled guard
    when absent. A changed configuration key drops the owner's entries of every stream."""
    import torch
    per = cache.get(owner)
    if per is None or per[0] != key:
        per = (key, {})
        cache[owner] = per
    sid = torch.cuda.current_stream(device).cuda_stream
    e = per[1].get(sid)
    if e is None:
        while len(per[1]) >= MAX_STREAM_ENTRIES:   # short-lived streams must not pile up ~1 MB entries
            per[1].pop(next(iter(per[1])))          # (the oldest; a dropped stream's entry is simply re-built)
        blob = torch.empty((nbytes,), dtype=torch.uint8, device=device)
        guard = torch.zeros((lib().lidf_pack_guard_bytes(),), dtype=torch.uint8, device=device)
        e = PackedEntry(key, blob, guard)
        per[1][sid] = e
    return e


def packed_entries(cache, owner):
    """Every PackedEntry of `owner` (one per stream that used it); tests and diagnostics."""
    per = cache.get(owner)
    return list(per[1].values()) if per is not None else []


def freeze_packed(*modules):
    """Opt out of the per-call fingerprint check for these modules: their packed weight streams are
    built on the next call and then trusted until invalidate_packed() — for inference loops that
    never touch the parameters and want the last microseconds (the check costs one small launch
    plus the early-exit pack launches per call)."""
    for m in modules:
        FROZEN.add(m)


def invalidate_packed(*modules):
    """Drop the packed weight streams of these modules (and un-freeze them): the next call re-packs."""
    for m in modules:
        FROZEN.discard(m)
        PACK_CACHE.pop(m, None)
        PACK_CACHE_REFINE.pop(m, None)


def lib():
    """Load liblidf_hip.so (built by implicit_depth_amd/csrc/build.py). Raises if absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "liblidf_hip.so not found at %s — run `python -c 'import __graft_entry__ as g; "
                "g.build()'` (there is no CPU fallback)" % LIB_PATH)
        handle = C.CDLL(LIB_PATH)
        try:
            handle.lidf_version.restype, handle.lidf_version.argtypes = C.c_int, []
            have = handle.lidf_version()
        except AttributeError:
            have = None
        if have != ABI:
            raise RuntimeError(
                "%s was built for ABI %s, this binding (implicit_depth_amd/_lib.py) is written against ABI %d "
                "of include/lidf_hip.h — rebuild it: `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or implicit_depth_amd/csrc/build.py --force)" % (LIB_PATH, have, ABI))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(status):
    """Turn a non-zero lidf_status into RuntimeError (the reference's TORCH_CHECK behaviour)."""
    if status != LIDF_OK:
        raise RuntimeError(lib().lidf_strerror(status).decode())


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream(device=None):
    import torch
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_cuda(*tensors, names=None):
    """CHECK_INPUT of the reference bindings: CUDA + contiguous, else RuntimeError."""
    for i, t in enumerate(tensors):
        if t is None:
            continue
        name = names[i] if names else "tensor %d" % i
        if not t.is_cuda:
            raise RuntimeError("%s must be a CUDA tensor" % name)
        if not t.is_contiguous():
            raise RuntimeError("%s must be contiguous" % name)


def hip_event():
    """A raw hipEvent_t (timing disabled) for the C-ABI's event arguments (LidfFrameArgs.ev_fork / ev_join),
    created by the HIP runtime liblidf_hip.so itself is linked against (lidf_event_create): torch.cuda.Event
    creates its handle lazily at the first record(), the library needs it up front, and a second copy of the
    runtime opened from here would hand out handles that are foreign to the library's hipEventRecord.
    Release it with hip_event_destroy()."""
    e = C.c_void_p()
    check(lib().lidf_event_create(C.byref(e)))
    return e.value


def hip_event_destroy(e):
    if e:
        lib().lidf_event_destroy(C.c_void_p(e))
