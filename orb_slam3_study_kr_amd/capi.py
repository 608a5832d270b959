"""ctypes mirror of include/orbslam3_hip.h and loader of the HIP shared library.

The product path has NO CPU fallback: if the gfx950 library is missing or fails
to load, :func:`load_library` raises.  (The CPU oracle under ``oracle/`` is test
infrastructure and is never imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

PKG_DIR = Path(__file__).resolve().parent
REPO_ROOT = PKG_DIR.parent
LIB_PATH = PKG_DIR / "csrc" / "liborbslam3_hip.so"

OSH_OK = 0
OSH_ERR_INVALID = -1
OSH_ERR_DEVICE = -2
OSH_ERR_UNSUPPORTED = -3
OSH_ERR_NO_DEVICE = -4
OSH_EDGE_MONO = 0
OSH_EDGE_STEREO = 1
OSH_EDGE_BODY = 2
OSH_EDGE_RIGHT = 2   # LocalInertialBA: EdgeMono(1), the right camera of a fisheye rig
OSH_LBA_MAX_TRACE = 128
OSH_K_COUNT = 11
KERNEL_NAMES = ["linearize", "pose_hess", "schur", "solve", "backsub", "residual", "control", "schur_reduce", "schur_cross", "lin_aux", "lin_pose"]

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_uint8_p = C.POINTER(C.c_uint8)
c_int64_p = C.POINTER(C.c_int64)
c_float_p = C.POINTER(C.c_float)


class LbaProblem(C.Structure):
    """``osh_lba_problem`` (include/orbslam3_hip.h)."""

    _fields_ = [
        ("n_free", C.c_int32), ("n_fixed", C.c_int32), ("n_points", C.c_int32), ("n_edges", C.c_int32),
        ("pose_qt", c_double_p), ("pose_cam", c_double_p), ("points", c_double_p),
        ("edge_pose", c_int32_p), ("edge_point", c_int32_p), ("edge_kind", c_uint8_p),
        ("edge_obs", c_double_p), ("edge_info", c_double_p),
        ("huber_mono", C.c_double), ("huber_stereo", C.c_double),
        ("lambda_init", C.c_double), ("max_iterations", C.c_int32),
        ("stop_flag", c_uint8_p), ("kb8", c_double_p), ("cam2", c_double_p), ("trl", c_double_p),
    ]


class LbaResult(C.Structure):
    """``osh_lba_result`` (include/orbslam3_hip.h)."""

    _fields_ = [
        ("pose_qt", c_double_p), ("points", c_double_p), ("edge_chi2", c_double_p), ("edge_depth_pos", c_uint8_p),
        ("status", C.c_int32), ("iterations", C.c_int32), ("trials", C.c_int32), ("n_trace", C.c_int32),
        ("chi2_trace", C.c_double * OSH_LBA_MAX_TRACE),
        ("lambda_trace", C.c_double * OSH_LBA_MAX_TRACE),
        ("trials_trace", C.c_int32 * OSH_LBA_MAX_TRACE),
        ("chi2_initial", C.c_double),
    ]


class PoseProblem(C.Structure):
    """``osh_pose_problem`` (include/orbslam3_hip.h)."""

    _fields_ = [
        ("n_edges", C.c_int32), ("pose_qt", c_double_p), ("cam", c_double_p), ("points", c_double_p),
        ("edge_kind", c_uint8_p), ("edge_obs", c_double_p), ("edge_info", c_double_p),
        ("huber_mono", C.c_double), ("huber_stereo", C.c_double),
        ("chi2_mono", C.c_float * 4), ("chi2_stereo", C.c_float * 4), ("iterations", C.c_int32 * 4),
        ("kb8", c_double_p), ("cam2", c_double_p), ("trl", c_double_p),
    ]


class PoseResult(C.Structure):
    """``osh_pose_result`` (include/orbslam3_hip.h)."""

    _fields_ = [
        ("pose_qt", C.c_double * 7), ("outlier", c_uint8_p), ("edge_chi2", c_double_p),
        ("n_bad", C.c_int32), ("rounds", C.c_int32), ("iterations", C.c_int32 * 4), ("chi2_final", C.c_double * 4),
        ("status", C.c_int32),
    ]


OSH_PREINT_FLOATS = 72


class LibaProblem(C.Structure):
    """``osh_liba_problem`` (include/orbslam3_hip.h)."""

    _fields_ = [
        ("n_opt", C.c_int32), ("n_fixed_imu", C.c_int32), ("n_fixed", C.c_int32),
        ("n_points", C.c_int32), ("n_edges", C.c_int32), ("n_links", C.c_int32),
        ("pose_Rcw", c_double_p), ("pose_tcw", c_double_p), ("pose_Rwb", c_double_p), ("pose_twb", c_double_p),
        ("Rcb", c_double_p), ("tcb", c_double_p), ("tbc", c_double_p), ("cam", c_double_p),
        ("vel", c_double_p), ("bias_g", c_double_p), ("bias_a", c_double_p), ("points", c_double_p),
        ("edge_pose", c_int32_p), ("edge_point", c_int32_p), ("edge_kind", c_uint8_p),
        ("edge_obs", c_double_p), ("edge_info", c_double_p),
        ("link_prev", c_int32_p), ("link_cur", c_int32_p), ("link_preint", c_float_p),
        ("link_info", c_double_p), ("link_info_g", c_double_p), ("link_info_a", c_double_p), ("link_robust", c_uint8_p),
        ("huber_mono", C.c_double), ("huber_stereo", C.c_double), ("huber_inertial", C.c_double),
        ("lambda_init", C.c_double), ("max_iterations", C.c_int32), ("kb8", c_double_p),
        ("cam2", c_double_p), ("trl", c_double_p), ("link_bias", c_int32_p),
    ]


class PoseiProblem(C.Structure):
    """``osh_posei_problem`` (include/orbslam3_hip.h)."""

    _fields_ = [
        ("mode", C.c_int32), ("n_edges", C.c_int32), ("rec_init", C.c_int32),
        ("Rcw", c_double_p), ("tcw", c_double_p), ("Rwb", c_double_p), ("twb", c_double_p),
        ("vel", c_double_p), ("bias_g", c_double_p), ("bias_a", c_double_p),
        ("prev_Rwb", c_double_p), ("prev_twb", c_double_p), ("prev_vel", c_double_p), ("prev_bias_g", c_double_p), ("prev_bias_a", c_double_p),
        ("Rcb", c_double_p), ("tcb", c_double_p), ("tbc", c_double_p), ("cam", c_double_p), ("kb8", c_double_p), ("cam2", c_double_p),
        ("trl", c_double_p), ("preint", c_float_p), ("info_inertial", c_double_p), ("info_g", c_double_p), ("info_a", c_double_p),
        ("prior_Rwb", c_double_p), ("prior_twb", c_double_p), ("prior_vel", c_double_p), ("prior_bg", c_double_p), ("prior_ba", c_double_p),
        ("prior_H", c_double_p), ("points", c_double_p), ("edge_kind", c_uint8_p), ("edge_obs", c_double_p), ("edge_info", c_double_p),
        ("edge_close", c_uint8_p), ("huber_mono", C.c_double), ("huber_stereo", C.c_double), ("huber_prior", C.c_double),
        ("chi2_mono", C.c_float * 4), ("chi2_stereo", C.c_float * 4), ("iterations", C.c_int32 * 4),
    ]


class PoseiResult(C.Structure):
    """``osh_posei_result`` (include/orbslam3_hip.h)."""

    _fields_ = [
        ("Rcw", C.c_double * 9), ("tcw", C.c_double * 3), ("Rwb", C.c_double * 9), ("twb", C.c_double * 3),
        ("vel", C.c_double * 3), ("bias_g", C.c_double * 3), ("bias_a", C.c_double * 3),
        ("outlier", c_uint8_p), ("edge_chi2", c_double_p),
        ("n_bad", C.c_int32), ("n_inliers", C.c_int32), ("rounds", C.c_int32), ("status", C.c_int32),
        ("H", C.c_double * 900),
    ]


class LibaResult(C.Structure):
    """``osh_liba_result`` (include/orbslam3_hip.h)."""

    _fields_ = [
        ("pose_Rcw", c_double_p), ("pose_tcw", c_double_p), ("pose_Rwb", c_double_p), ("pose_twb", c_double_p),
        ("vel", c_double_p), ("bias_g", c_double_p), ("bias_a", c_double_p), ("points", c_double_p),
        ("edge_chi2", c_double_p), ("edge_depth_pos", c_uint8_p),
        ("status", C.c_int32), ("iterations", C.c_int32), ("trials", C.c_int32), ("n_trace", C.c_int32),
        ("chi2_trace", C.c_double * OSH_LBA_MAX_TRACE),
        ("lambda_trace", C.c_double * OSH_LBA_MAX_TRACE),
        ("trials_trace", C.c_int32 * OSH_LBA_MAX_TRACE),
        ("chi2_initial", C.c_double), ("chi2_final", C.c_double),
    ]


class OrbBatch(C.Structure):
    """``osh_orb_batch`` (include/orbslam3_hip.h)."""

    _fields_ = [
        ("n_pairs", C.c_int32), ("n_query", C.c_int32), ("n_train", C.c_int32),
        ("query_desc", c_uint8_p), ("train_desc", c_uint8_p), ("train_level", c_int32_p),
        ("cand_off", c_int32_p), ("cand_idx", c_int32_p), ("pair_cand_base", c_int64_p),
    ]


class OrbGrid(C.Structure):
    """``osh_orb_grid`` (include/orbslam3_hip.h)."""

    _fields_ = [
        ("train_xy", c_float_p), ("train_uright", c_float_p), ("train_skip", c_uint8_p),
        ("min_x", C.c_float), ("min_y", C.c_float), ("cell_w_inv", C.c_float), ("cell_h_inv", C.c_float),
        ("cols", C.c_int32), ("rows", C.c_int32),
        ("query_window", c_float_p), ("query_levels", c_int32_p), ("query_uright", c_float_p),
    ]


class FrustumFrame(C.Structure):
    """``osh_frustum_frame`` (include/orbslam3_hip.h)."""

    _fields_ = [
        ("Rcw", C.c_float * 9), ("tcw", C.c_float * 3), ("Ow", C.c_float * 3),
        ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("bf", C.c_float),
        ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float), ("max_y", C.c_float),
        ("log_scale_factor", C.c_float), ("n_scale_levels", C.c_int32), ("viewing_cos_limit", C.c_float),
        ("fisheye", C.c_int32), ("kb8", C.c_float * 4),
    ]


class FrustumPoints(C.Structure):
    """``osh_frustum_points`` (include/orbslam3_hip.h)."""

    _fields_ = [("n", C.c_int32), ("pos", c_float_p), ("normal", c_float_p), ("min_dist", c_float_p), ("max_dist", c_float_p)]


class FrustumResult(C.Structure):
    """``osh_frustum_result`` (include/orbslam3_hip.h)."""

    _fields_ = [("stage", c_uint8_p), ("proj_x", c_float_p), ("proj_y", c_float_p), ("proj_xr", c_float_p),
                ("depth", c_float_p), ("view_cos", c_float_p), ("level", c_int32_p)]


def ptr(a, typ):
    """Pointer of ctypes type `typ` to the data of numpy array `a` (None -> NULL)."""
    if a is None:
        return C.cast(None, typ)
    assert a.flags["C_CONTIGUOUS"], "array must be C-contiguous"
    return a.ctypes.data_as(typ)


# Every symbol include/orbslam3_hip.h declares, with its signature.
_SIGNATURES = {
    "osh_last_error": (C.c_char_p, []),
    "osh_version": (C.c_char_p, []),
    "osh_device_count": (C.c_int, []),
    "osh_lba_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "osh_lba_destroy": (None, [C.c_void_p]),
    "osh_lba_upload": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(LbaProblem)]),
    "osh_lba_optimize": (C.c_int, [C.c_void_p]),
    "osh_lba_download": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(LbaResult)]),
    "osh_lba_solve": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(LbaProblem), C.POINTER(LbaResult)]),
    "osh_lba_linearize": (C.c_int, [C.c_void_p, C.c_int32] + [c_double_p] * 7),
    "osh_lba_debug_trial": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, c_double_p, c_double_p, c_double_p]),
    "osh_liba_solve": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(LibaProblem), C.POINTER(LibaResult)]),
    "osh_lba_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "osh_lba_get_profile": (C.c_int, [C.c_void_p, c_int64_p, c_double_p]),
    "osh_lba_kernel_name": (C.c_char_p, [C.c_int]),
    "osh_pose_optimize": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(PoseProblem), C.POINTER(PoseResult)]),
    "osh_lba_get_plan_stats": (C.c_int, [C.c_void_p, c_int64_p]),
    "osh_lba_get_upload_times": (C.c_int, [C.c_void_p, c_double_p]),
    "osh_lba_get_pack_profile": (C.c_int, [C.c_void_p, c_double_p]),
    "osh_lba_set_pack_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "osh_lba_pack_compare": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(LbaProblem), c_int64_p]),
    "osh_lba_schur_plan_stats": (C.c_int, [C.POINTER(LbaProblem), c_int64_p]),
    "osh_lba_pack_check": (C.c_int, [C.c_int32, C.POINTER(LbaProblem), C.c_int32, c_int64_p, c_double_p]),
    "osh_orb_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "osh_orb_destroy": (None, [C.c_void_p]),
    "osh_orb_upload": (C.c_int, [C.c_void_p, C.POINTER(OrbBatch)]),
    "osh_orb_upload_grid": (C.c_int, [C.c_void_p, C.POINTER(OrbBatch), C.POINTER(OrbGrid)]),
    "osh_orb_frustum": (C.c_int, [C.c_void_p, C.POINTER(FrustumFrame), C.POINTER(FrustumPoints), C.POINTER(FrustumResult)]),
    "osh_posei_optimize": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(PoseiProblem), C.POINTER(PoseiResult)]),
    "osh_orb_match": (C.c_int, [C.c_void_p]),
    "osh_orb_match_local_points": (C.c_int, [C.c_void_p, C.c_float, C.c_int32, c_uint8_p, c_uint8_p, c_int32_p, c_int32_p, c_int32_p, c_int32_p]),
    "osh_orb_download": (C.c_int, [C.c_void_p] + [c_int32_p] * 6),
    "osh_orb_get_profile": (C.c_int, [C.c_void_p, c_int64_p, c_double_p]),
    "osh_orb_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "osh_liba_get_profile": (C.c_int, [C.POINTER(C.c_int32), c_int64_p]),
    "osh_orb_list_distances": (C.c_int, [C.c_void_p, c_int32_p]),
    "osh_orb_get_resolve_profile": (C.c_int, [C.c_void_p, c_int64_p, c_double_p]),
    "osh_orb_distance_matrix": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, c_uint8_p, c_uint8_p, c_int32_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

# include/orbslam3_hip_host.h (C wrappers of the C++ host layer)
c_float_p = C.POINTER(C.c_float)
_HOST_SIGNATURES = {
    "osh_host_graph_create": (C.c_void_p, [C.c_int32, c_int64_p, c_float_p, c_float_p, c_float_p, C.c_int32, C.c_int32, c_int64_p,
                                           c_float_p, C.c_int32, c_int32_p, c_int32_p, c_float_p, c_int32_p, C.c_int64, C.c_int32]),
    "osh_host_graph_destroy": (None, [C.c_void_p]),
    "osh_host_last_call_ms": (C.c_double, []),
    "osh_host_search_by_sim3": (C.c_int, [C.c_void_p, C.c_void_p, c_float_p, C.c_float, C.c_int32, c_float_p, c_uint8_p, c_float_p, c_uint8_p, c_int32_p,
                                          C.c_int32, c_float_p, c_uint8_p, c_float_p, c_uint8_p, c_int32_p, c_int32_p, c_int32_p]),
    "osh_host_graph_set_fisheye": (None, [C.c_void_p, c_float_p]),
    "osh_host_graph_set_rig": (C.c_int, [C.c_void_p, c_float_p, c_float_p, C.c_int32, c_int32_p, c_int32_p, c_float_p, c_int32_p]),
    "osh_host_last_pack_rig": (C.c_int, [C.c_void_p, c_double_p, c_double_p]),
    "osh_host_last_pack_kb8": (C.c_int, [C.c_void_p, c_double_p]),
    "osh_host_graph_set_covisible": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, c_int32_p]),
    "osh_host_pack_lba": (C.c_int, [C.c_void_p, C.c_int32, c_int32_p, c_double_p, c_double_p, c_double_p, c_int32_p, c_int32_p,
                                    c_uint8_p, c_double_p, c_double_p, c_int64_p, c_int64_p]),
    "osh_host_run_lba": (C.c_int, [C.c_void_p, C.c_int32, c_uint8_p, c_int32_p]),
    "osh_host_get_kf_pose": (None, [C.c_void_p, C.c_int32, c_float_p]),
    "osh_host_get_mp_pos": (None, [C.c_void_p, C.c_int32, c_float_p]),
    "osh_host_mp_num_observations": (C.c_int, [C.c_void_p, C.c_int32]),
    "osh_host_mp_is_bad": (C.c_int, [C.c_void_p, C.c_int32]),
    "osh_host_kf_num_matches": (C.c_int, [C.c_void_p, C.c_int32]),
    "osh_host_kf_observes": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "osh_host_map_change_index": (C.c_int, [C.c_void_p]),
    "osh_host_kf_pose_sets": (C.c_int, [C.c_void_p, C.c_int32]),
    "osh_host_graph_set_inertial": (C.c_int, [C.c_void_p, C.c_int32, c_int32_p, c_int32_p, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p]),
    "osh_host_pack_liba": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(LibaProblem), c_int64_p, c_int64_p]),
    "osh_host_run_liba": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "osh_host_pack_full_inertial": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.POINTER(LibaProblem), c_int64_p, c_int64_p, c_int32_p]),
    "osh_host_run_full_inertial": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_float, C.c_float]),
    "osh_host_get_kf_inertial_gba": (C.c_int64, [C.c_void_p, C.c_int32, c_float_p, c_float_p]),
    "osh_host_pack_merge_inertial": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(LibaProblem), c_int64_p, c_int64_p, c_int32_p, c_int64_p, c_int64_p]),
    "osh_host_run_merge_inertial": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, c_int64_p, c_double_p]),
    "osh_host_get_kf_velocity": (None, [C.c_void_p, C.c_int32, c_float_p]),
    "osh_host_get_kf_bias": (None, [C.c_void_p, C.c_int32, c_float_p]),
    "osh_host_preintegrate": (C.c_int, [C.c_int32, c_float_p, c_float_p, C.c_float, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p]),
    "osh_host_inertial_information": (C.c_int, [c_float_p, c_double_p]),
    "osh_host_frame_create": (C.c_void_p, [C.c_int32, c_float_p, c_int32_p, c_float_p, c_float_p, c_uint8_p, c_float_p, c_float_p,
                                           C.c_float, C.c_float, C.c_int32, C.c_float]),
    "osh_host_frame_set_rig": (C.c_int, [C.c_void_p, C.c_int32, c_int32_p, c_int32_p, c_float_p]),
    "osh_host_search_local_points_rig": (C.c_int, [C.c_void_p, C.c_int32, c_uint8_p, c_uint8_p, c_float_p, c_int32_p, c_float_p, c_uint8_p,
                                         c_float_p, c_int32_p, c_float_p, c_int32_p, C.c_float, C.c_float, c_int32_p]),
    "osh_host_frame_set_fisheye": (None, [C.c_void_p, c_float_p]),
    "osh_host_frame_destroy": (None, [C.c_void_p]),
    "osh_host_frame_search_local_points_projected": (C.c_int, [C.c_void_p, C.c_int32, c_float_p, c_float_p, c_float_p, c_float_p, C.c_float,
                                                              c_uint8_p, c_float_p, c_float_p, c_float_p, c_float_p, c_int32_p, c_uint8_p,
                                                              c_int32_p, C.c_float, C.c_float, c_int32_p, c_int32_p]),
    "osh_host_search_local_points": (C.c_int, [C.c_void_p, C.c_int32, c_uint8_p, c_float_p, c_float_p, c_int32_p, c_float_p, c_float_p,
                                               c_int32_p, C.c_float, C.c_float, c_int32_p]),
    "osh_host_search_last_frame": (C.c_int, [C.c_void_p, C.c_void_p, c_int32_p, C.c_int32, c_float_p, c_uint8_p, C.c_float, C.c_int32,
                                             C.c_int32, c_int32_p]),
    "osh_host_frame_set_camera2": (C.c_int, [C.c_void_p, c_float_p]),
    "osh_host_search_by_bow_kf": (C.c_int, [C.c_int32, c_uint8_p, c_float_p, c_uint8_p, C.c_int32, c_int32_p, c_int32_p, c_int32_p,
                                            C.c_int32, c_uint8_p, c_float_p, c_uint8_p, C.c_int32, c_int32_p, c_int32_p, c_int32_p,
                                            C.c_float, C.c_int32, c_int32_p]),
    "osh_host_search_by_bow": (C.c_int, [C.c_void_p, C.c_int32, c_uint8_p, c_float_p, c_uint8_p, C.c_int32, c_int32_p, c_int32_p, c_int32_p,
                                         C.c_int32, c_int32_p, c_int32_p, c_int32_p, C.c_float, C.c_int32, c_int32_p]),
    "osh_host_frame_pose_optimization": (C.c_int, [C.c_void_p, C.c_int32, c_float_p, c_int32_p, c_float_p, C.c_int32, c_float_p, c_uint8_p]),
    "osh_host_posei_create": (C.c_void_p, [C.c_int32, C.c_int32, c_float_p, c_int32_p, c_float_p, C.c_int32, c_float_p, c_float_p, c_float_p, c_float_p,
                                           c_float_p, c_float_p, C.c_int32, c_float_p, c_uint8_p, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p,
                                           c_float_p, c_float_p, c_float_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p]),
    "osh_host_posei_destroy": (None, [C.c_void_p]),
    "osh_host_posei_pack": (C.c_int, [C.c_void_p, C.c_int32, C.POINTER(PoseiProblem), c_int32_p]),
    "osh_host_posei_run": (C.c_int, [C.c_void_p, C.c_int32, c_float_p, c_float_p, c_float_p, c_float_p, c_float_p, c_uint8_p, c_double_p, c_int32_p]),
    "osh_host_search_sim3": (C.c_int, [C.c_void_p, c_float_p, C.c_int32, c_float_p, c_uint8_p, c_float_p, c_float_p, c_uint8_p, c_int32_p,
                                       C.c_int32, C.c_float, C.c_int32, c_int32_p, c_int32_p]),
    "osh_host_fuse": (C.c_int, [C.c_void_p, C.c_int32, c_float_p, c_uint8_p, c_float_p, c_float_p, c_uint8_p, c_uint8_p, c_int32_p, C.c_int32,
                                c_int32_p, c_int32_p, c_uint8_p, C.c_float, c_int32_p, c_uint8_p, c_int32_p, c_int32_p, c_uint8_p, c_int32_p, c_int32_p]),
    "osh_host_fuse_sim3": (C.c_int, [C.c_void_p, c_float_p, C.c_int32, c_float_p, c_uint8_p, c_float_p, c_float_p, c_uint8_p, c_int32_p, c_int32_p,
                                     C.c_int32, c_int32_p, c_uint8_p, C.c_float, c_int32_p, c_int32_p, c_int32_p]),
    "osh_host_search_for_triangulation": (C.c_int, [c_float_p, C.c_int32, C.c_float, C.c_int32, c_float_p, c_int32_p, c_uint8_p, c_uint8_p, c_float_p,
                                                    C.c_int32, c_int32_p, c_int32_p, c_int32_p, C.c_int32, c_float_p, c_int32_p, c_uint8_p, c_uint8_p,
                                                    c_float_p, C.c_int32, c_int32_p, c_int32_p, c_int32_p, C.c_int32, C.c_int32, C.c_int32, c_int32_p]),
    "osh_host_search_for_initialization": (C.c_int, [C.c_void_p, C.c_void_p, c_float_p, C.c_int32, C.c_float, C.c_int32, c_int32_p]),
    "osh_host_pack_gba": (C.c_int, [C.c_void_p, c_int32_p, c_double_p, c_double_p, c_double_p, c_int32_p, c_int32_p, c_uint8_p, c_double_p,
                                    c_double_p, c_int64_p, c_int64_p]),
    "osh_host_pack_welding": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, c_int32_p, C.c_int32, c_int32_p, c_int32_p, c_double_p, c_double_p,
                                        c_double_p, c_int32_p, c_int32_p, c_uint8_p, c_double_p, c_double_p, c_int64_p, c_int64_p]),
    "osh_host_run_welding": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, c_int32_p, C.c_int32, c_int32_p, c_uint8_p]),
    "osh_host_run_gba": (C.c_int, [C.c_void_p, C.c_int32, c_uint8_p, C.c_int64, C.c_int32]),
    "osh_host_get_kf_pose_gba": (C.c_int64, [C.c_void_p, C.c_int32, c_float_p]),
    "osh_host_get_mp_pos_gba": (C.c_int64, [C.c_void_p, C.c_int32, c_float_p]),
    "osh_host_mp_normal_updates": (C.c_int, [C.c_void_p, C.c_int32]),
    "osh_host_set_bad": (None, [C.c_void_p, C.c_int32, C.c_int32]),
    "osh_host_search_keyframe": (C.c_int, [C.c_void_p, C.c_int32, c_float_p, c_int32_p, C.c_int32, c_float_p, c_uint8_p, c_float_p,
                                           c_uint8_p, c_uint8_p, c_int32_p, C.c_float, C.c_int32, C.c_int32, c_int32_p]),
}
HOST_EXPORTED_SYMBOLS = tuple(_HOST_SIGNATURES)

_lib = None


def load_library(path: os.PathLike | None = None) -> C.CDLL:
    """Load liborbslam3_hip.so (built by ``__graft_entry__.build()``); raise if absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    # ORBSLAM3_HIP_LIB: developer aid for A/B runs of an alternative build of the same library
    p = Path(path) if path else Path(os.environ.get("ORBSLAM3_HIP_LIB", LIB_PATH))
    if not p.exists():
        raise RuntimeError(
            f"{p} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the product path)."
        )
    lib = C.CDLL(str(p))
    for name, (res, args) in list(_SIGNATURES.items()) + list(_HOST_SIGNATURES.items()):
        fn = getattr(lib, name)  # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def last_error(lib=None) -> str:
    lib = lib or load_library()
    s = lib.osh_last_error()
    return s.decode() if s else ""


class OshError(RuntimeError):
    def __init__(self, code: int, where: str, msg: str):
        super().__init__(f"{where} failed with code {code}: {msg}")
        self.code = code


def check(code: int, where: str, lib=None):
    if code != OSH_OK:
        raise OshError(code, where, last_error(lib))


def np_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)
