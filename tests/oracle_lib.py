"""ctypes binding of oracle/_build/liborboracle.so (the CPU restatement; TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "liborboracle.so")

KP_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
     ("octave", "<i4"), ("class_id", "<i4")]
)
assert KP_DTYPE.itemsize == 28

QUERY_DTYPE = np.dtype(
    [("u", "<f4"), ("v", "<f4"), ("u_r", "<f4"), ("radius", "<f4"), ("min_level", "<i4"),
     ("max_level", "<i4"), ("valid", "<i4"), ("blocks", "<i4"), ("angle", "<f4"), ("desc", "u1", (32,))]
)
assert QUERY_DTYPE.itemsize == 68

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
assert FRUSTUM_DTYPE.itemsize == 168 and MAP_POINT_DTYPE.itemsize == 72 and TRACK_DTYPE.itemsize == 24

GRID_COLS, GRID_ROWS = 64, 48
MAX_LEVELS = 16


class OOFrame(C.Structure):
    _fields_ = [
        ("n", C.c_int), ("keys_un", C.c_void_p), ("desc", C.c_void_p), ("u_right", C.c_void_p),
        ("min_x", C.c_float), ("max_x", C.c_float), ("min_y", C.c_float), ("max_y", C.c_float),
        ("grid_w_inv", C.c_float), ("grid_h_inv", C.c_float), ("n_levels", C.c_int),
        ("scale_factors", C.c_void_p), ("cell_start", C.c_int32 * (GRID_COLS * GRID_ROWS + 1)),
        ("cell_idx", C.c_void_p),
    ]


class OOPyramidView(C.Structure):
    _fields_ = [("n_levels", C.c_int), ("data", C.c_void_p * MAX_LEVELS), ("stride", C.c_int * MAX_LEVELS),
                ("w", C.c_int * MAX_LEVELS), ("h", C.c_int * MAX_LEVELS)]


class FeatVecNode(C.Structure):
    _fields_ = [("node_id", C.c_int32), ("start", C.c_int32), ("count", C.c_int32)]


def build(force: bool = False) -> str:
    src = [os.path.join(ORACLE_DIR, f) for f in ("orb_oracle.c", "orb_oracle.h")]
    stale = force or not os.path.exists(LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in src if os.path.exists(s))
    if stale:
        subprocess.run(["make", "-C", ORACLE_DIR, "-B"], check=True, capture_output=True)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(LIB_PATH)
    vp, ci, cf = C.c_void_p, C.c_int, C.c_float
    L.oo_cvround.restype = ci; L.oo_cvround.argtypes = [C.c_double]
    L.oo_cvroundf.restype = ci; L.oo_cvroundf.argtypes = [cf]
    L.oo_fast_atan2.restype = cf; L.oo_fast_atan2.argtypes = [cf, cf]
    L.oo_sinf.restype = cf; L.oo_sinf.argtypes = [cf]
    L.oo_cosf.restype = cf; L.oo_cosf.argtypes = [cf]
    L.oo_resize_linear_u8.argtypes = [vp, ci, ci, ci, vp, ci, ci, ci]
    L.oo_resize_tables.argtypes = [ci, ci, vp, vp]
    L.oo_gauss_taps7.argtypes = [vp]
    L.oo_gaussian_blur7_u8.argtypes = [vp, ci, ci, ci, vp, ci]
    L.oo_copy_make_border_reflect101.argtypes = [vp, ci, ci, ci, vp, ci, ci]
    L.oo_fast9_16.restype = ci; L.oo_fast9_16.argtypes = [vp, ci, ci, ci, ci, ci, ci, vp, vp, vp]
    L.oo_fast_corner_score.restype = ci; L.oo_fast_corner_score.argtypes = [vp, ci, ci]
    L.oo_extractor_create.restype = vp; L.oo_extractor_create.argtypes = [ci, cf, ci, ci, ci]
    L.oo_extractor_destroy.argtypes = [vp]
    L.oo_extractor_levels.restype = ci; L.oo_extractor_levels.argtypes = [vp]
    for name in ("scale_factors", "inv_scale_factors", "sigma2", "inv_sigma2"):
        f = getattr(L, "oo_extractor_" + name); f.restype = C.POINTER(cf); f.argtypes = [vp]
    L.oo_extractor_features_per_level.restype = C.POINTER(ci); L.oo_extractor_features_per_level.argtypes = [vp]
    L.oo_extractor_umax.restype = C.POINTER(ci); L.oo_extractor_umax.argtypes = [vp]
    L.oo_pattern.restype = C.POINTER(C.c_int8)
    L.oo_extract.restype = ci; L.oo_extract.argtypes = [vp, vp, ci, ci, ci, vp, vp, ci, C.POINTER(ci)]
    L.oo_level_size.restype = ci; L.oo_level_size.argtypes = [vp, ci, C.POINTER(ci), C.POINTER(ci)]
    L.oo_level_pixels.restype = vp; L.oo_level_pixels.argtypes = [vp, ci, C.POINTER(ci)]
    L.oo_level_blurred.restype = vp; L.oo_level_blurred.argtypes = [vp, ci, C.POINTER(ci)]
    L.oo_level_candidates.restype = ci
    L.oo_level_candidates.argtypes = [vp, ci, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.oo_level_keypoints.restype = ci; L.oo_level_keypoints.argtypes = [vp, ci, C.POINTER(vp)]
    L.oo_distribute_octree.restype = ci
    L.oo_distribute_octree.argtypes = [vp, vp, vp, ci, ci, ci, ci, ci, ci, vp]
    L.oo_ic_angle.restype = cf; L.oo_ic_angle.argtypes = [vp, ci, ci, ci, vp]
    L.oo_orb_descriptor.argtypes = [vp, ci, ci, ci, cf, vp]
    L.oo_descriptor_distance.restype = ci; L.oo_descriptor_distance.argtypes = [vp, vp]
    L.oo_frame_build_grid.argtypes = [C.POINTER(OOFrame)]
    L.oo_features_in_area.restype = ci
    L.oo_features_in_area.argtypes = [C.POINTER(OOFrame), cf, cf, cf, ci, ci, vp]
    L.oo_search_by_projection_points.restype = ci
    L.oo_search_by_projection_points.argtypes = [C.POINTER(OOFrame), vp, ci, cf, vp, vp]
    L.oo_unproject_stereo.restype = None
    L.oo_unproject_stereo.argtypes = [vp, vp, cf, vp, ci, vp]
    L.oo_track_query.restype = None
    L.oo_track_query.argtypes = [vp, vp, vp]
    L.oo_unproject_stereo_n.restype = None
    L.oo_unproject_stereo_n.argtypes = [vp, vp, vp, vp, ci, ci, vp]
    L.oo_track_queries_n.restype = None
    L.oo_track_queries_n.argtypes = [vp, vp, ci, vp]
    L.oo_logf.restype = cf
    L.oo_logf.argtypes = [cf]
    L.oo_predict_scale.restype = ci
    L.oo_predict_scale.argtypes = [cf, cf, cf, ci]
    L.oo_is_in_frustum.restype = ci
    L.oo_is_in_frustum.argtypes = [vp, vp, cf, vp]
    L.oo_search_local_points.restype = ci
    L.oo_search_local_points.argtypes = [C.POINTER(OOFrame), vp, vp, ci, cf, cf, vp, vp, vp, C.POINTER(C.c_int)]
    L.oo_search_by_projection_keyframe.restype = ci
    L.oo_search_by_projection_keyframe.argtypes = [C.POINTER(OOFrame), vp, ci, ci, ci, vp, vp]
    L.oo_search_by_projection_frame.restype = ci
    L.oo_search_by_projection_frame.argtypes = [C.POINTER(OOFrame), vp, ci, ci, vp, vp]
    L.oo_search_by_bow.restype = ci
    L.oo_search_by_bow.argtypes = [vp, vp, vp, vp, ci, vp, vp, vp, ci, vp, ci, vp, cf, ci, vp]
    L.oo_proj_best.restype = None
    L.oo_proj_best.argtypes = [C.POINTER(OOFrame), vp, ci, ci, vp, vp, vp]
    L.oo_search_for_triangulation.restype = ci
    L.oo_search_for_triangulation.argtypes = [vp, vp, vp, vp, ci, vp, ci, vp, vp, vp, vp, vp, ci, vp, ci, vp, vp, ci, ci, vp]
    L.oo_search_by_bow_kf.restype = ci
    L.oo_search_by_bow_kf.argtypes = [vp, vp, vp, ci, vp, ci, vp, vp, vp, vp, ci, vp, ci, vp, cf, ci, vp]
    L.oo_search_for_initialization.restype = ci
    L.oo_search_for_initialization.argtypes = [vp, vp, ci, C.POINTER(OOFrame), vp, ci, cf, ci, vp]
    L.oo_three_maxima.argtypes = [vp, ci, C.POINTER(ci), C.POINTER(ci), C.POINTER(ci)]
    L.oo_compute_stereo_matches.restype = ci
    L.oo_compute_stereo_matches.argtypes = [vp, vp, ci, vp, vp, ci, C.POINTER(OOPyramidView),
                                            C.POINTER(OOPyramidView), vp, vp, cf, cf, vp, vp]
    L.oo_vocab_create.restype = vp; L.oo_vocab_create.argtypes = [ci, ci, ci, ci, ci, vp, vp, vp, vp]
    L.oo_vocab_load_text.restype = vp; L.oo_vocab_load_text.argtypes = [C.c_char_p]
    L.oo_vocab_load_binary.restype = vp; L.oo_vocab_load_binary.argtypes = [C.c_char_p]
    L.oo_vocab_destroy.argtypes = [vp]
    L.oo_vocab_nodes.restype = ci; L.oo_vocab_nodes.argtypes = [vp]
    L.oo_vocab_words.restype = ci; L.oo_vocab_words.argtypes = [vp]
    L.oo_vocab_transform_feature.argtypes = [vp, vp, ci, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_double)]
    L.oo_vocab_transform.restype = ci
    L.oo_vocab_transform.argtypes = [vp, vp, ci, ci, vp, vp, C.POINTER(ci), vp, vp, C.POINTER(ci)]
    _lib = L
    return L


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class OracleExtractor:
    """Mirror of ORB_SLAM2::ORBextractor over the C oracle."""

    def __init__(self, nfeatures=2000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        self.L = lib()
        self.h = self.L.oo_extractor_create(nfeatures, scale_factor, nlevels, ini_th, min_th)
        if not self.h:
            raise ValueError("bad extractor parameters")
        self.nfeatures, self.nlevels = nfeatures, nlevels

    def __del__(self):
        if getattr(self, "h", None):
            self.L.oo_extractor_destroy(self.h)
            self.h = None

    def _farr(self, fn, n):
        return np.array([fn(self.h)[i] for i in range(n)], dtype=np.float32)

    @property
    def scale_factors(self):
        return self._farr(self.L.oo_extractor_scale_factors, self.nlevels)

    @property
    def inv_scale_factors(self):
        return self._farr(self.L.oo_extractor_inv_scale_factors, self.nlevels)

    @property
    def sigma2(self):
        return self._farr(self.L.oo_extractor_sigma2, self.nlevels)

    @property
    def inv_sigma2(self):
        return self._farr(self.L.oo_extractor_inv_sigma2, self.nlevels)

    @property
    def features_per_level(self):
        p = self.L.oo_extractor_features_per_level(self.h)
        return [p[i] for i in range(self.nlevels)]

    @property
    def umax(self):
        p = self.L.oo_extractor_umax(self.h)
        return [p[i] for i in range(16)]

    def __call__(self, img: np.ndarray):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        cap = self.nfeatures + 4 * self.nlevels + 64
        kps = np.zeros(cap, dtype=KP_DTYPE)
        desc = np.zeros((cap, 32), dtype=np.uint8)
        n = C.c_int(0)
        rc = self.L.oo_extract(self.h, _p(img), w, h, img.strides[0], _p(kps), _p(desc), cap, C.byref(n))
        if rc != 0:
            raise RuntimeError(f"oo_extract rc={rc} n={n.value}")
        return kps[: n.value].copy(), desc[: n.value].copy()

    def level_size(self, level):
        w, h = C.c_int(), C.c_int()
        self.L.oo_level_size(self.h, level, C.byref(w), C.byref(h))
        return w.value, h.value

    def _plane(self, fn, level):
        w, h = self.level_size(level)
        s = C.c_int()
        p = fn(self.h, level, C.byref(s))
        if not p:
            return None
        buf = (C.c_uint8 * (s.value * h)).from_address(p)
        return np.frombuffer(buf, dtype=np.uint8).reshape(h, s.value)[:, :w].copy()

    def level_pixels(self, level):
        return self._plane(self.L.oo_level_pixels, level)

    def level_blurred(self, level):
        return self._plane(self.L.oo_level_blurred, level)

    def level_candidates(self, level):
        x, y, s = C.c_void_p(), C.c_void_p(), C.c_void_p()
        n = self.L.oo_level_candidates(self.h, level, C.byref(x), C.byref(y), C.byref(s))
        if n == 0:
            z = np.zeros(0, np.int32)
            return z, z, z
        mk = lambda p: np.frombuffer((C.c_int * n).from_address(p.value), dtype=np.int32).copy()
        return mk(x), mk(y), mk(s)

    def level_keypoints(self, level):
        p = C.c_void_p()
        n = self.L.oo_level_keypoints(self.h, level, C.byref(p))
        if n == 0:
            return np.zeros(0, KP_DTYPE)
        return np.frombuffer((C.c_uint8 * (28 * n)).from_address(p.value), dtype=KP_DTYPE).copy()


def resize_linear(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().oo_resize_linear_u8(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dw, dh, dw)
    return dst


def gaussian_blur7(src: np.ndarray) -> np.ndarray:
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros_like(src)
    lib().oo_gaussian_blur7_u8(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dst.strides[0])
    return dst


def fast9_16(img: np.ndarray, th: int, nonmax: bool = True):
    img = np.ascontiguousarray(img, np.uint8)
    cap = img.size
    x = np.zeros(cap, np.int32); y = np.zeros(cap, np.int32); s = np.zeros(cap, np.int32)
    n = lib().oo_fast9_16(_p(img), img.strides[0], img.shape[1], img.shape[0], th, int(nonmax), cap, _p(x), _p(y), _p(s))
    return x[:n], y[:n], s[:n]


def distribute_octree(x, y, score, minX, maxX, minY, maxY, N):
    x = np.ascontiguousarray(x, np.int32); y = np.ascontiguousarray(y, np.int32)
    score = np.ascontiguousarray(score, np.int32)
    out = np.zeros(max(len(x), 1), np.int32)
    n = lib().oo_distribute_octree(_p(x), _p(y), _p(score), len(x), minX, maxX, minY, maxY, N, _p(out))
    return out[:n]


def descriptor_distance(a, b) -> int:
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return lib().oo_descriptor_distance(_p(a), _p(b))


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


def kf_search(frame: "OracleFrame", cam, pts, mode, inv_level_sigma2=None, matched=None, th_low=50, check_orientation=True):
    """Whole loops of Fuse (mode 1), Fuse(Sim3) (2), one direction of SearchBySim3 (3), SearchByProjection(KF,Scw) (4) and
    SearchByProjection(Frame,KF,sAlreadyFound,th,ORBdist) (5) on the oracle.  Returns (n, results, matched)."""
    cam = np.ascontiguousarray(cam, KF_CAMERA_DTYPE).reshape(-1)[:1]
    pts = np.ascontiguousarray(pts, KF_POINT_DTYPE)
    res = np.zeros(len(pts), KF_RESULT_DTYPE)
    L = lib()
    for fn in (L.oo_fuse, L.oo_fuse_sim3, L.oo_search_by_sim3_dir, L.oo_reloc_query):
        fn.restype = None
    L.oo_search_by_projection_loop.restype = C.c_int
    if mode == 1:
        inv = np.ascontiguousarray(inv_level_sigma2, np.float32)
        L.oo_fuse(C.byref(frame.f), _p(inv), _p(cam), _p(pts), len(pts), _p(res))
        return int((res["best_idx"] >= 0).sum()), res, None
    if mode == 2:
        L.oo_fuse_sim3(C.byref(frame.f), _p(cam), _p(pts), len(pts), _p(res))
        return int((res["best_idx"] >= 0).sum()), res, None
    if mode == 3:
        L.oo_search_by_sim3_dir(C.byref(frame.f), _p(cam), _p(pts), len(pts), _p(res))
        return int((res["best_idx"] >= 0).sum()), res, None
    fn_ = len(frame.kps)
    m = np.zeros(max(fn_, 1), np.uint8) if matched is None else np.ascontiguousarray(matched, np.uint8).copy()
    if mode == 4:
        nm = L.oo_search_by_projection_loop(C.byref(frame.f), _p(cam), _p(pts), len(pts), int(th_low), _p(m), _p(res))
        return nm, res, m[:fn_]
    q = np.zeros(len(pts), QUERY_DTYPE)
    for i in range(len(pts)):
        L.oo_reloc_query(_p(cam), _p(pts[i:i + 1]), _p(q[i:i + 1]))
    nm, assigned, m2 = frame.search_by_projection_keyframe(q, check_orientation, int(th_low), m[:fn_])
    res["best_idx"] = -1
    for j, a in enumerate(assigned):
        if a >= 0:
            res["best_idx"][a] = j
    return nm, res, m2


_POPC = np.array([bin(i).count("1") for i in range(256)], np.uint16)


def hamming_bf(A, B, groupA=None, groupB=None):
    """Best / second-best of every row of A over B as the loops of SearchByBoW walk them (ORBmatcher.cc:201-222, distance
    :1542-1556): first minimum in index order, second-smallest distance; optional node-id groups.  Returns
    (best_idx [-1 = none], best_dist [256], second_dist [256]).  numpy restatement (byte arithmetic only)."""
    A = np.ascontiguousarray(A, np.uint8).reshape(-1, 32); B = np.ascontiguousarray(B, np.uint8).reshape(-1, 32)
    bi = np.full(len(A), -1, np.int32); bd = np.full(len(A), 256, np.int32); sd = np.full(len(A), 256, np.int32)
    if len(A) == 0 or len(B) == 0:
        return bi, bd, sd
    for i0 in range(0, len(A), 256):
        a = A[i0:i0 + 256]
        d = _POPC[a[:, None, :] ^ B[None, :, :]].sum(axis=2).astype(np.int32)
        if groupA is not None:
            d = np.where(np.asarray(groupA)[i0:i0 + 256, None] == np.asarray(groupB)[None, :], d, 1 << 20)
        order = np.argsort(d, axis=1, kind="stable")[:, :2]
        rows = np.arange(len(a))
        b0 = d[rows, order[:, 0]]
        ok = b0 < (1 << 20)
        bi[i0:i0 + 256] = np.where(ok, order[:, 0], -1); bd[i0:i0 + 256] = np.where(ok, b0, 256)
        if B.shape[0] > 1:
            b1 = d[rows, order[:, 1]]
            sd[i0:i0 + 256] = np.where(b1 < (1 << 20), b1, 256)
    return bi, bd, sd


class OracleFrame:
    """POD view of the Frame members the matcher reads (mvKeysUn, mDescriptors, mvuRight, grid)."""

    def __init__(self, kps, desc, scale_factors, min_x, max_x, min_y, max_y, u_right=None):
        self.kps = np.ascontiguousarray(kps, KP_DTYPE)
        self.desc = np.ascontiguousarray(desc, np.uint8)
        self.sf = np.ascontiguousarray(scale_factors, np.float32)
        self.u_right = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
        self.cell_idx = np.zeros(max(len(self.kps), 1), np.int32)
        f = OOFrame()
        f.n = len(self.kps)
        f.keys_un = self.kps.ctypes.data
        f.desc = self.desc.ctypes.data
        f.u_right = None if self.u_right is None else self.u_right.ctypes.data
        f.min_x, f.max_x, f.min_y, f.max_y = min_x, max_x, min_y, max_y
        f.grid_w_inv = np.float32(GRID_COLS) / (np.float32(max_x) - np.float32(min_x))
        f.grid_h_inv = np.float32(GRID_ROWS) / (np.float32(max_y) - np.float32(min_y))
        f.n_levels = len(self.sf)
        f.scale_factors = self.sf.ctypes.data
        f.cell_idx = self.cell_idx.ctypes.data
        self.f = f
        lib().oo_frame_build_grid(C.byref(f))

    @property
    def cell_start(self):
        return np.array(self.f.cell_start[:], dtype=np.int32)

    def features_in_area(self, x, y, r, min_level=-1, max_level=-1):
        out = np.zeros(max(self.f.n, 1), np.int32)
        n = lib().oo_features_in_area(C.byref(self.f), x, y, r, min_level, max_level, _p(out))
        return out[:n]

    def search_by_projection_points(self, queries, nnratio, blocked=None):
        q = np.ascontiguousarray(queries, QUERY_DTYPE)
        blocked = np.zeros(self.f.n, np.uint8) if blocked is None else np.ascontiguousarray(blocked, np.uint8).copy()
        assigned = np.full(self.f.n, -1, np.int32)
        nm = lib().oo_search_by_projection_points(C.byref(self.f), _p(q), len(q), nnratio, _p(blocked), _p(assigned))
        return nm, assigned, blocked

    def search_by_projection_frame(self, queries, check_orientation=True, blocked=None):
        q = np.ascontiguousarray(queries, QUERY_DTYPE)
        blocked = np.zeros(self.f.n, np.uint8) if blocked is None else np.ascontiguousarray(blocked, np.uint8).copy()
        assigned = np.full(self.f.n, -1, np.int32)
        nm = lib().oo_search_by_projection_frame(C.byref(self.f), _p(q), len(q), int(check_orientation), _p(blocked), _p(assigned))
        return nm, assigned, blocked


def _kf(self, queries, check_orientation=True, orb_dist=100, blocked=None):
    q = np.ascontiguousarray(queries, QUERY_DTYPE)
    blocked = np.zeros(self.f.n, np.uint8) if blocked is None else np.ascontiguousarray(blocked, np.uint8).copy()
    assigned = np.full(self.f.n, -1, np.int32)
    nm = lib().oo_search_by_projection_keyframe(C.byref(self.f), _p(q), len(q), int(check_orientation), int(orb_dist), _p(blocked), _p(assigned))
    return nm, assigned, blocked


OracleFrame.search_by_projection_keyframe = _kf


def _proj_best(self, queries, inv_level_sigma2=None):
    q = np.ascontiguousarray(queries, QUERY_DTYPE)
    bi = np.full(len(q), -1, np.int32); bd = np.full(len(q), 256, np.int32)
    inv = None if inv_level_sigma2 is None else np.ascontiguousarray(inv_level_sigma2, np.float32)
    lib().oo_proj_best(C.byref(self.f), _p(q), len(q), 2 if inv is not None else 1, None if inv is None else _p(inv), _p(bi), _p(bd))
    return bi, bd


OracleFrame.proj_best = _proj_best


def unproject_stereo(cam, keys, desc, depth, observed=1):
    cam = np.ascontiguousarray(cam, UNPROJECT_CAM_DTYPE).reshape(1)
    keys = np.ascontiguousarray(keys, KP_DTYPE); desc = np.ascontiguousarray(desc, np.uint8)
    out = np.zeros(len(keys), LAST_POINT_DTYPE)
    depth = np.ascontiguousarray(depth[: len(keys)], np.float32)
    lib().oo_unproject_stereo_n(_p(cam), _p(keys), _p(depth), _p(desc), len(keys), observed, _p(out))
    return out


def track_queries(pose, points):
    pose = np.ascontiguousarray(pose, TRACK_POSE_DTYPE).reshape(1)
    pts = np.ascontiguousarray(points, LAST_POINT_DTYPE)
    q = np.zeros(len(pts), QUERY_DTYPE)
    lib().oo_track_queries_n(_p(pose), _p(pts), len(pts), _p(q))
    return q


def is_in_frustum(frustum, points, viewing_cos_limit=0.5):
    fr = np.ascontiguousarray(frustum, FRUSTUM_DTYPE).reshape(1)
    mp = np.ascontiguousarray(points, MAP_POINT_DTYPE)
    track = np.zeros(len(mp), TRACK_DTYPE)
    for i in range(len(mp)):
        lib().oo_is_in_frustum(_p(fr), mp[i:i + 1].ctypes.data, viewing_cos_limit, track[i:i + 1].ctypes.data)
    return track


def _slp(self, frustum, points, th, nnratio, blocked=None):
    fr = np.ascontiguousarray(frustum, FRUSTUM_DTYPE).reshape(1)
    mp = np.ascontiguousarray(points, MAP_POINT_DTYPE)
    track = np.zeros(len(mp), TRACK_DTYPE)
    blocked = np.zeros(self.f.n, np.uint8) if blocked is None else np.ascontiguousarray(blocked, np.uint8).copy()
    assigned = np.full(self.f.n, -1, np.int32)
    ntm = C.c_int(0)
    nm = lib().oo_search_local_points(C.byref(self.f), _p(fr), _p(mp), len(mp), th, nnratio, _p(track), _p(blocked), _p(assigned),
                                      C.byref(ntm))
    return ntm.value, nm, track, assigned, blocked


OracleFrame.search_local_points = _slp


def search_for_initialization(keys1, desc1, frame2: "OracleFrame", prev_xy, window, nnratio, check_orientation=True):
    keys1 = np.ascontiguousarray(keys1, KP_DTYPE); desc1 = np.ascontiguousarray(desc1, np.uint8)
    prev = np.ascontiguousarray(prev_xy, np.float32).reshape(-1, 2).copy()
    m12 = np.full(len(keys1), -1, np.int32)
    nm = lib().oo_search_for_initialization(_p(keys1), _p(desc1), len(keys1), C.byref(frame2.f), _p(prev), int(window),
                                            nnratio, int(check_orientation), _p(m12))
    return nm, m12, prev


def featvec_arrays(groups: dict):
    """{node_id: [indices]} -> (nodes array of FeatVecNode, idx int32 array), nodes sorted by id."""
    ids = sorted(groups)
    nodes = (FeatVecNode * max(len(ids), 1))()
    idx = []
    for k, nid in enumerate(ids):
        nodes[k].node_id = nid; nodes[k].start = len(idx); nodes[k].count = len(groups[nid])
        idx.extend(groups[nid])
    return nodes, len(ids), np.asarray(idx if idx else [0], np.int32)


def search_by_bow(descA, angleA, validA, groupsA, descB, angleB, groupsB, nnratio=0.7, check_orientation=True):
    descA = np.ascontiguousarray(descA, np.uint8); descB = np.ascontiguousarray(descB, np.uint8)
    angleA = np.ascontiguousarray(angleA, np.float32); angleB = np.ascontiguousarray(angleB, np.float32)
    validA = np.ascontiguousarray(validA, np.uint8)
    nA, nnA, iA = featvec_arrays(groupsA)
    nB, nnB, iB = featvec_arrays(groupsB)
    matchB = np.full(len(descB), -1, np.int32)
    nm = lib().oo_search_by_bow(_p(descA), _p(angleA), _p(validA), C.cast(nA, C.c_void_p), nnA, _p(iA),
                                _p(descB), _p(angleB), len(descB), C.cast(nB, C.c_void_p), nnB, _p(iB),
                                nnratio, int(check_orientation), _p(matchB))
    return nm, matchB


def search_by_bow_kf(descA, angleA, validA, groupsA, descB, angleB, validB, groupsB, nnratio=0.8, check_orientation=True):
    descA = np.ascontiguousarray(descA, np.uint8); descB = np.ascontiguousarray(descB, np.uint8)
    angleA = np.ascontiguousarray(angleA, np.float32); angleB = np.ascontiguousarray(angleB, np.float32)
    validA = np.ascontiguousarray(validA, np.uint8); validB = np.ascontiguousarray(validB, np.uint8)
    nA, nnA, iA = featvec_arrays(groupsA)
    nB, nnB, iB = featvec_arrays(groupsB)
    matchA = np.full(len(descA), -1, np.int32)
    nm = lib().oo_search_by_bow_kf(_p(descA), _p(angleA), _p(validA), len(descA), C.cast(nA, C.c_void_p), nnA, _p(iA), _p(descB),
                                   _p(angleB), _p(validB), len(descB), C.cast(nB, C.c_void_p), nnB, _p(iB), nnratio,
                                   int(check_orientation), _p(matchA))
    return nm, matchA


def search_for_triangulation(keysA, descA, u_rightA, has_mpA, groupsA, keysB, descB, u_rightB, has_mpB, groupsB, epipolar,
                             only_stereo=False, check_orientation=True):
    keysA = np.ascontiguousarray(keysA, KP_DTYPE); keysB = np.ascontiguousarray(keysB, KP_DTYPE)
    descA = np.ascontiguousarray(descA, np.uint8); descB = np.ascontiguousarray(descB, np.uint8)
    urA = None if u_rightA is None else np.ascontiguousarray(u_rightA, np.float32)
    urB = None if u_rightB is None else np.ascontiguousarray(u_rightB, np.float32)
    hA = np.ascontiguousarray(has_mpA, np.uint8); hB = np.ascontiguousarray(has_mpB, np.uint8)
    ep = np.ascontiguousarray(epipolar, EPIPOLAR_DTYPE).reshape(1)
    nA, nnA, iA = featvec_arrays(groupsA)
    nB, nnB, iB = featvec_arrays(groupsB)
    matchA = np.full(len(descA), -1, np.int32)
    nm = lib().oo_search_for_triangulation(_p(keysA), _p(descA), None if urA is None else _p(urA), _p(hA), len(descA), C.cast(nA, C.c_void_p),
                                           nnA, _p(iA), _p(keysB), _p(descB), None if urB is None else _p(urB), _p(hB), len(descB),
                                           C.cast(nB, C.c_void_p), nnB, _p(iB), _p(ep), int(only_stereo), int(check_orientation), _p(matchA))
    return nm, matchA


def pyramid_view(planes):
    v = OOPyramidView()
    v.n_levels = len(planes)
    keep = []
    for i, p in enumerate(planes):
        p = np.ascontiguousarray(p, np.uint8)
        keep.append(p)
        v.data[i] = p.ctypes.data
        v.stride[i] = p.strides[0]
        v.w[i] = p.shape[1]
        v.h[i] = p.shape[0]
    return v, keep


def compute_stereo_matches(kL, dL, kR, dR, planesL, planesR, sf, isf, mbf, mb):
    kL = np.ascontiguousarray(kL, KP_DTYPE); kR = np.ascontiguousarray(kR, KP_DTYPE)
    dL = np.ascontiguousarray(dL, np.uint8); dR = np.ascontiguousarray(dR, np.uint8)
    sf = np.ascontiguousarray(sf, np.float32); isf = np.ascontiguousarray(isf, np.float32)
    vL, keepL = pyramid_view(planesL)
    vR, keepR = pyramid_view(planesR)
    ur = np.zeros(len(kL), np.float32); depth = np.zeros(len(kL), np.float32)
    n = lib().oo_compute_stereo_matches(_p(kL), _p(dL), len(kL), _p(kR), _p(dR), len(kR), C.byref(vL), C.byref(vR),
                                        _p(sf), _p(isf), mbf, mb, _p(ur), _p(depth))
    return n, ur, depth


class OracleVocabulary:
    """DBoW2 vocabulary tree + transform over the C oracle."""

    def __init__(self, handle):
        if not handle:
            raise ValueError("vocabulary could not be created / loaded")
        self.h = handle

    @classmethod
    def load_text(cls, path):
        return cls(lib().oo_vocab_load_text(path.encode()))

    @classmethod
    def load_binary(cls, path):
        return cls(lib().oo_vocab_load_binary(path.encode()))

    @classmethod
    def from_arrays(cls, k, L, parent, is_leaf, desc, weight, scoring=0, weighting=0):
        parent = np.ascontiguousarray(parent, np.int32); is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
        desc = np.ascontiguousarray(desc, np.uint8); weight = np.ascontiguousarray(weight, np.float64)
        return cls(lib().oo_vocab_create(k, L, scoring, weighting, len(parent), _p(parent), _p(is_leaf), _p(desc), _p(weight)))

    def __del__(self):
        if getattr(self, "h", None):
            lib().oo_vocab_destroy(self.h)
            self.h = None

    def info(self):
        return lib().oo_vocab_nodes(self.h), lib().oo_vocab_words(self.h)

    def transform(self, descriptors, levelsup=4):
        d = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
        n = len(d)
        word = np.zeros(n, np.int32); node = np.zeros(n, np.int32); weight = np.zeros(n, np.float64)
        for i in range(n):
            w, nd, wt = C.c_int32(), C.c_int32(), C.c_double()
            lib().oo_vocab_transform_feature(self.h, _p(d[i]), levelsup, C.byref(w), C.byref(nd), C.byref(wt))
            word[i], node[i], weight[i] = w.value, nd.value, wt.value
        bow_ids = np.zeros(max(n, 1), np.int32); bow_vals = np.zeros(max(n, 1), np.float64)
        fv_nodes = (FeatVecNode * max(n, 1))(); fv_idx = np.zeros(max(n, 1), np.int32)
        nb, nf = C.c_int(0), C.c_int(0)
        lib().oo_vocab_transform(self.h, _p(d), n, levelsup, _p(bow_ids), _p(bow_vals), C.byref(nb), C.cast(fv_nodes, C.c_void_p),
                                 _p(fv_idx), C.byref(nf))
        bow = {int(bow_ids[i]): float(bow_vals[i]) for i in range(nb.value)}
        fv = {int(fv_nodes[i].node_id): fv_idx[fv_nodes[i].start: fv_nodes[i].start + fv_nodes[i].count].tolist() for i in range(nf.value)}
        return bow, fv, (word, node, weight)


def synthetic_vocabulary(k=10, L=3, seed=0, stop_fraction=0.05, ragged=False):
    """Random k-ary tree of depth L in loadFromTextFile order (breadth first).  Returns arrays + the text lines."""
    rng = np.random.default_rng(seed)
    parent = [0]; leaf = [0]; depth = [0]
    frontier = [0]
    for lvl in range(1, L + 1):
        nxt = []
        for p in frontier:
            if ragged and lvl > 1 and rng.random() < 0.15:
                continue  # this node stays a leaf above the last level
            kk = k if not ragged else int(rng.integers(2, k + 1))
            for _ in range(kk):
                parent.append(p); leaf.append(0); depth.append(lvl); nxt.append(len(parent) - 1)
        frontier = nxt
    n = len(parent)
    has_child = np.zeros(n, bool)
    for i in range(1, n):
        has_child[parent[i]] = True
    leaf = (~has_child).astype(np.uint8); leaf[0] = 0
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    weight = np.where(leaf > 0, rng.uniform(0.5, 9.0, n), 0.0)
    weight[(rng.random(n) < stop_fraction) & (leaf > 0)] = 0.0  # stopped words
    return np.array(parent, np.int32), leaf, desc, weight


def write_vocabulary_binary(path, k, L, parent, leaf, desc, weight, scoring=0, weighting=0):
    """what ORBVocabulary::saveToBinaryFile writes (ORBVocabulary.cc:217-243)"""
    import struct
    with open(path, "wb") as f:
        f.write(struct.pack("<IIiiii", len(parent), 4 + 32 + 4 + 1, k, L, scoring, weighting))
        for i in range(1, len(parent)):
            f.write(struct.pack("<I", int(parent[i])) + bytes(np.asarray(desc[i], np.uint8)) + struct.pack("<f", float(weight[i])) +
                    struct.pack("<?", bool(leaf[i])))


def write_vocabulary_text(path, k, L, parent, leaf, desc, weight, scoring=0, weighting=0):
    with open(path, "w") as f:
        f.write(f"{k} {L} {scoring} {weighting}\n")
        for i in range(1, len(parent)):
            f.write(f"{parent[i]} {int(leaf[i])} " + " ".join(str(int(b)) for b in desc[i]) + f" {float(weight[i])!r}\n")
