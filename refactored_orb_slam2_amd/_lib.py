"""ctypes loader of the in-tree liborbfe.so (HIP, gfx950).  Fails loudly: there is no CPU fallback."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(CSRC, "liborbfe.so")

OK, ERR_INVALID, ERR_CAPACITY, ERR_NO_DEVICE, ERR_HIP, ERR_EMPTY = 0, -1, -2, -3, -4, -5
MAX_LEVELS = 16
STAGES = ("pyramid", "fast", "octree", "blur", "describe")

KP_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
     ("octave", "<i4"), ("class_id", "<i4")]
)
QUERY_DTYPE = np.dtype(
    [("u", "<f4"), ("v", "<f4"), ("u_r", "<f4"), ("radius", "<f4"), ("min_level", "<i4"),
     ("max_level", "<i4"), ("valid", "<i4"), ("blocks", "<i4"), ("angle", "<f4"), ("desc", "u1", (32,))]
)
CAND_DTYPE = np.dtype([("idx", "<i4"), ("dist", "<i4")])
# orbfe_frustum / orbfe_map_point / orbfe_track (include/orbfe.h)
FRUSTUM_DTYPE = np.dtype(
    [("Rcw", "<f4", (9,)), ("tcw", "<f4", (3,)), ("Ow", "<f4", (3,)), ("fx", "<f4"), ("fy", "<f4"), ("cx", "<f4"),
     ("cy", "<f4"), ("mbf", "<f4"), ("min_x", "<f4"), ("max_x", "<f4"), ("min_y", "<f4"), ("max_y", "<f4"),
     ("log_scale_factor", "<f4"), ("n_levels", "<i4"), ("scale_factors", "<f4", (16,))]
)
MAP_POINT_DTYPE = np.dtype(
    [("pos", "<f4", (3,)), ("normal", "<f4", (3,)), ("min_distance", "<f4"), ("max_distance", "<f4"), ("skip", "<i4"),
     ("observed", "<i4"), ("desc", "u1", (32,))]
)
TRACK_DTYPE = np.dtype(
    [("in_view", "<i4"), ("proj_x", "<f4"), ("proj_y", "<f4"), ("proj_xr", "<f4"), ("level", "<i4"), ("view_cos", "<f4")]
)
EPIPOLAR_DTYPE = np.dtype([("F12", "<f4", (9,)), ("ex", "<f4"), ("ey", "<f4"), ("scale_factors", "<f4", (16,)), ("level_sigma2", "<f4", (16,))])
UNPROJECT_CAM_DTYPE = np.dtype([("Rwc", "<f4", (9,)), ("Ow", "<f4", (3,)), ("cx", "<f4"), ("cy", "<f4"), ("invfx", "<f4"), ("invfy", "<f4")])
LAST_POINT_DTYPE = np.dtype([("pos", "<f4", (3,)), ("valid", "<i4"), ("observed", "<i4"), ("octave", "<i4"), ("angle", "<f4"), ("desc", "u1", (32,))])
TRACK_POSE_DTYPE = np.dtype(
    [("Rcw", "<f4", (9,)), ("tcw", "<f4", (3,)), ("fx", "<f4"), ("fy", "<f4"), ("cx", "<f4"), ("cy", "<f4"), ("mbf", "<f4"),
     ("min_x", "<f4"), ("max_x", "<f4"), ("min_y", "<f4"), ("max_y", "<f4"), ("forward", "<i4"), ("backward", "<i4"),
     ("th", "<f4"), ("scale_factors", "<f4", (16,))]
)
assert UNPROJECT_CAM_DTYPE.itemsize == 64 and LAST_POINT_DTYPE.itemsize == 60 and TRACK_POSE_DTYPE.itemsize == 160
assert FRUSTUM_DTYPE.itemsize == 168 and MAP_POINT_DTYPE.itemsize == 72 and TRACK_DTYPE.itemsize == 24
BF_DTYPE = np.dtype([("best_idx", "<i4"), ("best_dist", "<i4"), ("second_dist", "<i4")])
# orbfe_kf_camera / orbfe_kf_point / orbfe_kf_result: the whole-function keyframe-rate searches (orbfe_kf_search)
KF_CAMERA_DTYPE = np.dtype(
    [("R", "<f4", (9,)), ("t", "<f4", (3,)), ("R2", "<f4", (9,)), ("t2", "<f4", (3,)), ("Ow", "<f4", (3,)), ("fx", "<f4"), ("fy", "<f4"),
     ("cx", "<f4"), ("cy", "<f4"), ("mbf", "<f4"), ("min_x", "<f4"), ("max_x", "<f4"), ("min_y", "<f4"), ("max_y", "<f4"),
     ("log_scale_factor", "<f4"), ("n_levels", "<i4"), ("th", "<f4"), ("scale_factors", "<f4", (16,))]
)
KF_POINT_DTYPE = np.dtype(
    [("pos", "<f4", (3,)), ("normal", "<f4", (3,)), ("min_distance", "<f4"), ("max_distance", "<f4"), ("skip", "<i4"),
     ("angle", "<f4"), ("desc", "u1", (32,))]
)
KF_RESULT_DTYPE = np.dtype([("best_idx", "<i4"), ("best_dist", "<i4"), ("level", "<i4"), ("u", "<f4"), ("v", "<f4"), ("u_r", "<f4")])
assert KF_CAMERA_DTYPE.itemsize == 220 and KF_POINT_DTYPE.itemsize == 72 and KF_RESULT_DTYPE.itemsize == 24
KF_FUSE, KF_FUSE_SIM3, KF_SIM3, KF_LOOP, KF_RELOC = 1, 2, 3, 4, 5


class OrbfeError(RuntimeError):
    def __init__(self, code: int, where: str, text: str):
        super().__init__(f"{where} failed with code {code}: {text}")
        self.code = code


class Params(C.Structure):
    _fields_ = [("n_features", C.c_int32), ("scale_factor", C.c_float), ("n_levels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32)]


class FeatVecNode(C.Structure):
    _fields_ = [("node_id", C.c_int32), ("start", C.c_int32), ("count", C.c_int32)]


class FrameView(C.Structure):
    _fields_ = [("n", C.c_int32), ("keys_un", C.c_void_p), ("desc", C.c_void_p), ("u_right", C.c_void_p),
                ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float), ("max_y", C.c_float)]


# every symbol include/orbfe.h declares (tests check the .so exports all of them)
EXPORTS = [
    "orbfe_last_error", "orbfe_device_count", "orbfe_set_device", "orbfe_device_malloc", "orbfe_device_free", "orbfe_device_upload",
    "orbfe_device_download", "orbfe_thread_release", "orbfe_extractor_prepare", "orbfe_frontend_prepare",
    "orbfe_shard_range", "orbfe_gather_unique_id", "orbfe_gather_create", "orbfe_gather_create_all", "orbfe_gather_destroy",
    "orbfe_gather_rank", "orbfe_gather_records", "orbfe_gather_sync", "orbfe_extractor_create", "orbfe_extractor_destroy",
    "orbfe_extractor_levels", "orbfe_extractor_scale_factors", "orbfe_extractor_inv_scale_factors",
    "orbfe_extractor_sigma2", "orbfe_extractor_inv_sigma2", "orbfe_extractor_features_per_level",
    "orbfe_extractor_max_keypoints", "orbfe_extract", "orbfe_pyramid_level", "orbfe_pyramid_level_size", "orbfe_pyramid_levels",
    "orbfe_extract_batch", "orbfe_extract_batch_device", "orbfe_device_pyramid", "orbfe_device_pyramid_layout", "orbfe_sync",
    "orbfe_device_status", "orbfe_debug_candidates", "orbfe_debug_blurred", "orbfe_debug_blur_kernel", "orbfe_debug_pyramid",
    "orbfe_debug_level_keypoints", "orbfe_profile_enable", "orbfe_stage_times", "orbfe_stage_intervals",
    "orbfe_matcher_create", "orbfe_matcher_destroy", "orbfe_matcher_sync", "orbfe_proj_match_batch_device",
    "orbfe_hamming_matrix_device", "orbfe_hamming_bf_device", "orbfe_proj_candidates",
    "orbfe_search_by_projection_points", "orbfe_search_by_projection_frame", "orbfe_stereo_match_device", "orbfe_stereo_match",
    "orbfe_search_for_initialization", "orbfe_search_by_bow", "orbfe_search_by_bow_kf", "orbfe_search_for_triangulation", "orbfe_proj_best", "orbfe_kf_search", "orbfe_search_by_projection_keyframe", "orbfe_search_local_points",
    "orbfe_search_local_points_batch_device", "orbfe_unproject_stereo_device", "orbfe_track_queries_device", "orbfe_track_queries_stereo_device",
    "orbfe_vocabulary_create", "orbfe_vocabulary_load_text", "orbfe_vocabulary_load_binary", "orbfe_vocabulary_destroy", "orbfe_vocabulary_info",
    "orbfe_bow_transform_device", "orbfe_compute_bow", "orbfe_png_info", "orbfe_png_info2", "orbfe_png_read_gray", "orbfe_png_read_gray2", "orbfe_png_read_gray16",
    "orbfe_pipeline_create", "orbfe_pipeline_destroy", "orbfe_pipeline_input", "orbfe_pipeline_submit", "orbfe_pipeline_wait",
    "orbfe_pipeline_output", "orbfe_pipeline_device_records", "orbfe_pipeline_stream", "orbfe_pipeline_gather", "orbfe_pipeline_gather_wait",
    "orbfe_pipeline_device_input", "orbfe_pipeline_submit_resident", "orbfe_debug_pipeline_streams",
]


def build(force: bool = False) -> str:
    """Compile liborbfe.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC]
    if force:
        args.append("-B")
    r = subprocess.run(args, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building liborbfe.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    return LIB_PATH


_lib = None


def lib():
    """Loads liborbfe.so.  Raises if it is missing -- the product path has no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    # One HIP runtime per process: PyTorch bundles its own libamdhip64 (same soname as /opt/rocm's).  Whichever is
    # loaded first serves both; torch cannot find the GPU if the system copy got there first, so when torch is
    # installed let it load its runtime before liborbfe.so pulls in libamdhip64.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, ci, cf, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    pi = C.POINTER(ci)
    L.orbfe_last_error.restype = C.c_char_p
    L.orbfe_device_count.argtypes = [pi]
    L.orbfe_extractor_create.argtypes = [C.POINTER(Params), ci, C.POINTER(vp)]
    L.orbfe_extractor_destroy.argtypes = [vp]
    L.orbfe_extractor_levels.argtypes = [vp, pi]
    for n in ("scale_factors", "inv_scale_factors", "sigma2", "inv_sigma2", "features_per_level"):
        getattr(L, "orbfe_extractor_" + n).argtypes = [vp, vp]
    L.orbfe_extractor_max_keypoints.argtypes = [vp, ci, ci, pi]
    L.orbfe_extract.argtypes = [vp, vp, ci, ci, ci, vp, vp, ci, pi]
    L.orbfe_pyramid_level.argtypes = [vp, ci, vp, ci, pi, pi]
    L.orbfe_pyramid_level_size.argtypes = [vp, ci, ci, ci, pi, pi]
    L.orbfe_pyramid_levels.argtypes = [vp, vp, vp]
    L.orbfe_extract_batch.argtypes = [vp, vp, ci, ci, ci, ci, vp, vp, ci, vp]
    L.orbfe_extract_batch_device.argtypes = [vp, vp, ci, ci, ci, ci, sz, vp, vp, ci, vp, vp]
    L.orbfe_device_pyramid.argtypes = [vp, ci, ci, C.POINTER(vp), pi, pi, pi]
    L.orbfe_sync.argtypes = [vp]
    L.orbfe_device_status.argtypes = [vp]
    L.orbfe_debug_candidates.argtypes = [vp, ci, ci, vp, vp, vp, ci, pi]
    L.orbfe_debug_blurred.argtypes = [vp, ci, ci, vp, ci]
    L.orbfe_debug_pyramid.argtypes = [vp, ci, ci, vp, ci]
    L.orbfe_debug_level_keypoints.argtypes = [vp, ci, ci, vp, vp, vp, ci, pi]
    L.orbfe_profile_enable.argtypes = [vp, ci]
    L.orbfe_stage_times.argtypes = [vp, vp, vp, ci]
    L.orbfe_stage_intervals.argtypes = [vp, vp, vp, vp, vp, ci, vp]
    L.orbfe_hamming_matrix_device.argtypes = [vp, ci, vp, ci, vp, vp]
    L.orbfe_hamming_bf_device.argtypes = [vp, vp, ci, ci, vp, vp, ci, vp, vp, vp, ci, vp, vp]
    L.orbfe_matcher_create.argtypes = [ci, C.POINTER(vp)]
    L.orbfe_matcher_destroy.argtypes = [vp]
    L.orbfe_matcher_sync.argtypes = [vp]
    L.orbfe_proj_match_batch_device.argtypes = [vp, ci, vp, vp, vp, vp, ci, cf, cf, cf, cf, vp, vp, ci, ci, cf, ci,
                                                vp, vp, vp, vp]
    L.orbfe_proj_candidates.argtypes = [C.POINTER(FrameView), vp, ci, vp, vp, ci]
    L.orbfe_search_by_projection_points.argtypes = [C.POINTER(FrameView), vp, ci, cf, vp, vp, pi]
    L.orbfe_search_by_projection_frame.argtypes = [C.POINTER(FrameView), vp, ci, ci, vp, vp, pi]
    L.orbfe_unproject_stereo_device.argtypes = [ci, vp, vp, vp, vp, ci, vp, ci, vp, vp]
    L.orbfe_track_queries_device.argtypes = [ci, vp, vp, vp, ci, ci, vp, vp, vp]
    L.orbfe_track_queries_stereo_device.argtypes = [ci, vp, vp, vp, vp, ci, vp, ci, vp, vp, vp, vp, vp, vp, ci, vp, vp, vp]
    L.orbfe_search_local_points.argtypes = [C.POINTER(FrameView), vp, vp, ci, cf, cf, vp, vp, vp, pi, pi]
    L.orbfe_search_local_points_batch_device.argtypes = [vp, ci, vp, vp, vp, vp, ci, cf, cf, cf, cf, vp, vp, vp, ci, cf, cf,
                                                         vp, vp, vp, vp, vp, vp]
    L.orbfe_search_by_projection_keyframe.argtypes = [C.POINTER(FrameView), vp, ci, ci, ci, vp, vp, pi]
    L.orbfe_vocabulary_create.argtypes = [ci, ci, ci, ci, ci, vp, vp, vp, vp, ci, C.POINTER(vp)]
    L.orbfe_vocabulary_load_text.argtypes = [C.c_char_p, ci, C.POINTER(vp)]
    L.orbfe_vocabulary_load_binary.argtypes = [C.c_char_p, ci, C.POINTER(vp)]
    L.orbfe_vocabulary_destroy.argtypes = [vp]
    L.orbfe_vocabulary_info.argtypes = [vp, pi, pi, pi, pi]
    L.orbfe_bow_transform_device.argtypes = [vp, vp, ci, ci, vp, vp, vp, vp]
    L.orbfe_compute_bow.argtypes = [vp, vp, ci, ci, vp, vp, vp, vp, vp, pi, vp, vp, pi]
    L.orbfe_proj_best.argtypes = [C.POINTER(FrameView), vp, ci, ci, vp, ci, vp, vp]
    L.orbfe_kf_search.argtypes = [C.POINTER(FrameView), vp, vp, vp, ci, ci, ci, ci, vp, vp, pi]
    L.orbfe_search_for_triangulation.argtypes = [vp, vp, vp, vp, ci, vp, ci, vp, vp, vp, vp, vp, ci, vp, ci, vp, vp, ci, ci, vp, pi]
    L.orbfe_search_by_bow_kf.argtypes = [vp, vp, vp, ci, vp, ci, vp, vp, vp, vp, ci, vp, ci, vp, cf, ci, vp, pi]
    L.orbfe_search_by_bow.argtypes = [vp, vp, vp, ci, vp, ci, vp, vp, vp, ci, vp, ci, vp, cf, ci, vp, pi]
    L.orbfe_search_for_initialization.argtypes = [C.POINTER(FrameView), C.POINTER(FrameView), vp, ci, cf, ci, vp, pi]
    L.orbfe_stereo_match_device.argtypes = [vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, ci, cf, cf, vp, vp, vp, vp]
    L.orbfe_stereo_match.argtypes = [vp, vp, vp, vp, ci, vp, vp, ci, cf, cf, vp, vp, pi]
    for name in EXPORTS:
        if name != "orbfe_last_error":
            getattr(L, name).restype = ci
    _lib = L
    return L


def check(rc: int, where: str):
    if rc != OK:
        raise OrbfeError(rc, where, lib().orbfe_last_error().decode(errors="replace"))


def stream_handle(stream) -> C.c_void_p:
    """torch stream -> hipStream_t.  None selects the handle's own stream.  torch's default stream is the NULL
    stream, which the C ABI also reads as "the handle's own stream" -- refuse it instead of racing silently."""
    if stream is None:
        return C.c_void_p(None)
    h = int(stream.cuda_stream)
    if h == 0:
        raise ValueError("pass an explicit torch.cuda.Stream (or None for the handle's own stream), not the default stream")
    return C.c_void_p(h)


def ptr(a) -> C.c_void_p:
    """Host numpy array or torch tensor -> void*."""
    if a is None:
        return C.c_void_p(None)
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    return C.c_void_p(a.data_ptr())
