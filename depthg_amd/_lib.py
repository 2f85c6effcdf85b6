"""ctypes loader for libdepthg_hip.so (C ABI: include/depthg_corr.h).  Fails loudly when the
library is missing: there is no CPU or eager fallback for the product path."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DEPTHG_LIB") or os.path.join(_HERE, "lib", "libdepthg_hip.so")   # DEPTHG_LIB: developer A/B builds

DG_OUT_COUNT = 9
DG_OUT_TOTAL = 8
DG_VERSION = 118                     # must match include/depthg_corr.h: a stale library is refused
DG_POINTWISE, DG_ZERO_CLAMP, DG_STABALIZE, DG_DEPTH_TERM, DG_NEED_GRAD, DG_SHARED_COORDS, DG_IDENTITY_GRID, DG_LINE_GRID, \
    DG_EXACT_MASKS, DG_FEATS_UNIT = (1 << i for i in range(10))

EXPORTS = ["dg_version", "dg_last_error", "dg_corr_workspace_bytes", "dg_corr_forward", "dg_corr_backward",
           "dg_corr_materialize", "dg_corr_relaunch_main", "dg_fps_workspace_bytes", "dg_fps_coords", "dg_fps_coords_pair", "dg_rand_coords_state", "dg_rand_keep_state", "dg_super_perms",
           "dg_salience_coords", "dg_simple_depth_coords", "dg_confusion_update", "dg_topk_rows", "dg_lhp_forward", "dg_lhp_backward", "dg_super_perms_seeded", "dg_super_perms_state",
           "dg_lhp_map_forward", "dg_lhp_map_backward", "dg_corr_forward_draw", "dg_corr_forward_masked",
           "dg_corr_backward_total", "dg_corr_main_kernel_name", "dg_corr_intra_folded",
           "dg_head_forward", "dg_head_workspace_bytes", "dg_head_weights_bytes", "dg_head_backward", "dg_head_forward_pair",
           "dg_head_backward_pair", "dg_cluster_lookup_forward",
           "dg_cluster_lookup_backward", "dg_probe_ce_forward", "dg_probe_ce_backward", "dg_knn_similarities",
           "dg_prof_main_span", "dg_corr_materialize_shared", "dg_normalize_split", "dg_sampled_sumsq",
           "dg_corr_forward_extnorm"]


class CorrDesc(ctypes.Structure):
    """struct dg_corr_desc"""
    _fields_ = [("B", ctypes.c_int32), ("C", ctypes.c_int32), ("D", ctypes.c_int32), ("h", ctypes.c_int32),
                ("w", ctypes.c_int32), ("S", ctypes.c_int32), ("n_neg", ctypes.c_int32),
                ("depth_h", ctypes.c_int32), ("depth_w", ctypes.c_int32), ("flags", ctypes.c_uint32),
                ("shift_intra", ctypes.c_float), ("shift_inter", ctypes.c_float), ("shift_neg", ctypes.c_float),
                ("shift_depth", ctypes.c_float),
                ("w_intra", ctypes.c_float), ("w_inter", ctypes.c_float), ("w_neg", ctypes.c_float),
                ("w_depth", ctypes.c_float), ("code_h", ctypes.c_int32), ("code_w", ctypes.c_int32)]


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"depthg_amd: {LIB_PATH} not found. Build it with `make -C depthg_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no fallback path.")
    lib = ctypes.CDLL(LIB_PATH)
    vp, cp = ctypes.c_void_p, ctypes.POINTER(CorrDesc)
    i32, f32 = ctypes.c_int32, ctypes.c_float
    lib.dg_version.restype = ctypes.c_int
    if lib.dg_version() != DG_VERSION:
        raise RuntimeError(f"depthg_amd: {LIB_PATH} is version {lib.dg_version()}, the Python layer expects {DG_VERSION}; "
                           "rebuild it with `make -C depthg_amd/csrc`")
    lib.dg_last_error.restype = ctypes.c_char_p
    lib.dg_corr_workspace_bytes.restype = ctypes.c_size_t
    lib.dg_corr_workspace_bytes.argtypes = [cp]
    lib.dg_corr_forward.restype = ctypes.c_int
    lib.dg_corr_forward.argtypes = [cp] + [vp] * 10 + [ctypes.c_size_t, vp]
    lib.dg_corr_forward_draw.restype = ctypes.c_int
    lib.dg_corr_forward_draw.argtypes = [cp] + [vp] * 8 + [ctypes.c_uint64, vp, vp, vp, ctypes.c_size_t, vp]
    lib.dg_corr_forward_masked.restype = ctypes.c_int
    lib.dg_corr_forward_masked.argtypes = [cp] + [vp] * 8 + [i32, ctypes.c_uint64, vp, vp, vp, f32, vp, vp, ctypes.c_size_t, vp]
    lib.dg_head_forward.restype = ctypes.c_int
    lib.dg_head_forward.argtypes = [i32] * 4 + [vp] * 10 + [f32] + [vp] * 5
    lib.dg_head_weights_bytes.restype = ctypes.c_size_t
    lib.dg_head_weights_bytes.argtypes = [i32] * 2
    lib.dg_head_workspace_bytes.restype = ctypes.c_size_t
    lib.dg_head_workspace_bytes.argtypes = [i32] * 4
    lib.dg_head_backward.restype = ctypes.c_int
    lib.dg_head_backward.argtypes = [i32] * 4 + [vp] * 3 + [f32] + [vp] * 10 + [ctypes.c_size_t, vp]
    lib.dg_head_forward_pair.restype = ctypes.c_int
    lib.dg_head_forward_pair.argtypes = [i32] * 4 + [vp] * 11 + [f32] + [vp] * 7
    lib.dg_head_backward_pair.restype = ctypes.c_int
    lib.dg_head_backward_pair.argtypes = [i32] * 4 + [vp] * 4 + [f32] + [vp] * 11 + [ctypes.c_size_t, vp]
    lib.dg_cluster_lookup_forward.restype = ctypes.c_int
    lib.dg_cluster_lookup_forward.argtypes = [vp, vp, f32] + [i32] * 4 + [vp] * 6
    lib.dg_cluster_lookup_backward.restype = ctypes.c_int
    lib.dg_cluster_lookup_backward.argtypes = [vp, vp, vp, f32, vp] + [i32] * 4 + [vp] * 4
    lib.dg_probe_ce_forward.restype = ctypes.c_int
    lib.dg_probe_ce_forward.argtypes = [vp, vp] + [i32] * 6 + [vp] * 3
    lib.dg_probe_ce_backward.restype = ctypes.c_int
    lib.dg_probe_ce_backward.argtypes = [vp] * 4 + [i32] * 6 + [vp] * 2
    lib.dg_corr_main_kernel_name.restype = ctypes.c_char_p
    lib.dg_corr_main_kernel_name.argtypes = [cp]
    lib.dg_corr_intra_folded.restype = ctypes.c_int
    lib.dg_corr_intra_folded.argtypes = [cp]
    lib.dg_corr_backward.restype = ctypes.c_int
    lib.dg_corr_backward.argtypes = [cp] + [vp] * 7 + [ctypes.c_size_t, vp]
    lib.dg_corr_backward_total.restype = ctypes.c_int
    lib.dg_corr_backward_total.argtypes = [cp] + [vp] * 7 + [ctypes.c_size_t, vp]
    lib.dg_corr_materialize.restype = ctypes.c_int
    lib.dg_corr_materialize.argtypes = [cp, ctypes.c_int32, vp, vp, vp, ctypes.c_size_t, vp]
    lib.dg_corr_materialize_shared.restype = ctypes.c_int
    lib.dg_corr_materialize_shared.argtypes = [cp, ctypes.c_int32, vp, vp, vp, vp, ctypes.c_size_t, vp]
    lib.dg_corr_relaunch_main.restype = ctypes.c_int
    lib.dg_corr_relaunch_main.argtypes = [cp, vp, vp, ctypes.c_size_t, vp]
    lib.dg_fps_workspace_bytes.restype = ctypes.c_size_t
    lib.dg_fps_workspace_bytes.argtypes = [ctypes.c_int32] * 3
    lib.dg_fps_coords.restype = ctypes.c_int
    lib.dg_fps_coords.argtypes = [vp] + [ctypes.c_int32] * 6 + [vp, vp, vp, ctypes.c_size_t, vp]
    lib.dg_sampled_sumsq.restype = ctypes.c_int
    lib.dg_sampled_sumsq.argtypes = [ctypes.c_int32] * 6 + [vp, vp, vp, ctypes.c_int32, vp, vp]
    lib.dg_corr_forward_extnorm.restype = ctypes.c_int
    lib.dg_corr_forward_extnorm.argtypes = [cp] + [vp] * 11 + [ctypes.c_size_t, vp]
    lib.dg_normalize_split.restype = ctypes.c_int
    lib.dg_normalize_split.argtypes = [ctypes.c_int32] * 4 + [vp, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p), vp]
    lib.dg_rand_coords_state.restype = ctypes.c_int
    lib.dg_rand_coords_state.argtypes = [vp, ctypes.c_int64, vp, vp]
    lib.dg_rand_keep_state.restype = ctypes.c_int
    lib.dg_rand_keep_state.argtypes = [vp, ctypes.c_int64, ctypes.c_float, vp, vp]
    lib.dg_fps_coords_pair.restype = ctypes.c_int
    lib.dg_fps_coords_pair.argtypes = [vp, vp] + [ctypes.c_int32] * 6 + [vp, vp, vp, ctypes.c_size_t, vp]
    lib.dg_super_perms.restype = ctypes.c_int
    lib.dg_super_perms.argtypes = [vp, ctypes.c_int32, ctypes.c_int32, vp, vp]
    lib.dg_salience_coords.restype = ctypes.c_int
    lib.dg_salience_coords.argtypes = [vp] + [ctypes.c_int32] * 4 + [vp, vp, vp, vp]
    lib.dg_simple_depth_coords.restype = ctypes.c_int
    lib.dg_simple_depth_coords.argtypes = [vp] + [ctypes.c_int32] * 6 + [vp, vp, vp, vp]
    lib.dg_confusion_update.restype = ctypes.c_int
    lib.dg_confusion_update.argtypes = [vp, vp, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, vp, vp]
    lib.dg_topk_rows.restype = ctypes.c_int
    lib.dg_topk_rows.argtypes = [vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, vp, vp, vp]
    lib.dg_knn_similarities.restype = ctypes.c_int
    lib.dg_knn_similarities.argtypes = [vp, vp, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int64, ctypes.c_int64, vp, ctypes.c_int64, vp]
    lib.dg_lhp_forward.restype = ctypes.c_int
    lib.dg_lhp_forward.argtypes = [vp, vp] + [ctypes.c_int32] * 6 + [vp, vp, vp, vp]
    lib.dg_lhp_backward.restype = ctypes.c_int
    lib.dg_lhp_backward.argtypes = [vp, vp, vp] + [ctypes.c_int32] * 4 + [vp, vp]
    lib.dg_lhp_map_forward.restype = ctypes.c_int
    lib.dg_lhp_map_forward.argtypes = [ctypes.c_int32, vp, vp, vp, vp] + [ctypes.c_int32] * 7 + [vp, vp, vp, vp]
    lib.dg_lhp_map_backward.restype = ctypes.c_int
    lib.dg_lhp_map_backward.argtypes = [ctypes.c_int32, vp, vp, vp] + [ctypes.c_int32] * 4 + [vp, vp]
    lib.dg_super_perms_seeded.restype = ctypes.c_int
    lib.dg_super_perms_seeded.argtypes = [ctypes.c_uint64, ctypes.c_int32, ctypes.c_int32, vp, vp]
    lib.dg_super_perms_state.restype = ctypes.c_int
    lib.dg_super_perms_state.argtypes = [vp, ctypes.c_int32, ctypes.c_int32, vp, vp]
    lib.dg_prof_main_span.restype = ctypes.c_int
    lib.dg_prof_main_span.argtypes = [vp]
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().dg_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"depthg_amd: {what} failed ({rc}): {msg}")
