"""Host-side mirror of ORB_SLAM2::ORBmatcher (+ the Frame helpers it reads) over the C ABI of liborbfe.so.

Method names and argument meaning follow Source/Libraries/ORB_SLAM2/include/ORBmatcher.h:34-114; SLAM
objects (Frame, MapPoint, KeyFrame) are replaced by plain arrays: a FrameView carries mvKeysUn /
mDescriptors / mvuRight / image bounds, queries carry one projected map point each (``QUERY_DTYPE``).
PyTorch is used only to hold device memory.  All distance / window / assignment work runs in HIP kernels, the
order-dependent passes (greedy assignment, SearchByBoW's per-node replay, the rotation histogram) included.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import BF_DTYPE, CAND_DTYPE, FRUSTUM_DTYPE, KP_DTYPE, MAP_POINT_DTYPE, QUERY_DTYPE, TRACK_DTYPE

TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 50, 30


class FrameView:
    """mvKeysUn, mDescriptors, mvuRight and mnMin/Max of a Frame (host arrays)."""

    def __init__(self, keys_un, desc, min_x, max_x, min_y, max_y, u_right=None):
        self.keys = np.ascontiguousarray(keys_un, KP_DTYPE)
        self.desc = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        self.u_right = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
        self.bounds = (float(min_x), float(max_x), float(min_y), float(max_y))
        v = _lib.FrameView()
        v.n = len(self.keys)
        v.keys_un = self.keys.ctypes.data
        v.desc = self.desc.ctypes.data
        v.u_right = None if self.u_right is None else self.u_right.ctypes.data
        v.min_x, v.max_x, v.min_y, v.max_y = self.bounds
        self.c = v

    @property
    def n(self):
        return len(self.keys)


def make_queries(n: int) -> np.ndarray:
    q = np.zeros(n, QUERY_DTYPE)
    q["valid"] = 1
    q["min_level"] = -1
    q["max_level"] = -1
    return q


def make_frustum(Rcw, tcw, fx, fy, cx, cy, mbf, bounds, scale_factor, n_levels) -> np.ndarray:
    """The Frame members Frame::isInFrustum reads (Frame.cc:284-339), computed as Frame does: mOw = -Rcw.t()*tcw
    (Frame.cc:274-279, float arithmetic), mfLogScaleFactor = logf(mfScaleFactor) (Frame.cc:80), mvScaleFactors from the
    extractor (ORBextractor.cc:416-420)."""
    fr = np.zeros(1, FRUSTUM_DTYPE)
    R = np.asarray(Rcw, np.float32).reshape(3, 3); t = np.asarray(tcw, np.float32).reshape(3)
    fr["Rcw"][0] = R.reshape(9); fr["tcw"][0] = t
    Rt = R.T
    fr["Ow"][0] = [-(np.float32(np.float32(Rt[r, 0] * t[0]) + np.float32(Rt[r, 1] * t[1])) + np.float32(Rt[r, 2] * t[2])) for r in range(3)]
    for k, v in (("fx", fx), ("fy", fy), ("cx", cx), ("cy", cy), ("mbf", mbf)):
        fr[k] = np.float32(v)
    fr["min_x"], fr["max_x"], fr["min_y"], fr["max_y"] = [np.float32(b) for b in bounds]
    fr["log_scale_factor"] = np.log(np.float32(scale_factor), dtype=np.float32)
    fr["n_levels"] = n_levels
    sf = np.ones(n_levels, np.float32)
    for i in range(1, n_levels):
        sf[i] = np.float32(np.float64(sf[i - 1]) * np.float64(np.float32(scale_factor)))   # float * double -> float (ORBextractor.cc:417)
    fr["scale_factors"][0, :len(sf)] = sf
    return fr


class ORBmatcher:
    TH_HIGH, TH_LOW, HISTO_LENGTH = TH_HIGH, TH_LOW, HISTO_LENGTH

    def __init__(self, nnratio: float = 0.6, checkOri: bool = True):
        self.mfNNratio = float(np.float32(nnratio))
        self.mbCheckOrientation = bool(checkOri)
        self._L = _lib.lib()

    # ---- DescriptorDistance for all pairs (device)
    @staticmethod
    def DescriptorDistanceMatrix(A, B) -> np.ndarray:
        import torch
        L = _lib.lib()
        dA = torch.from_numpy(np.ascontiguousarray(A, np.uint8).reshape(-1, 32)).cuda()
        dB = torch.from_numpy(np.ascontiguousarray(B, np.uint8).reshape(-1, 32)).cuda()
        out = torch.zeros((dA.shape[0], dB.shape[0]), dtype=torch.int16, device="cuda")
        torch.cuda.synchronize()  # inputs were produced on torch's default stream; the kernel runs on the NULL stream
        _lib.check(L.orbfe_hamming_matrix_device(_lib.ptr(dA), dA.shape[0], _lib.ptr(dB), dB.shape[0], _lib.ptr(out),
                                                 C.c_void_p(None)), "orbfe_hamming_matrix_device")
        torch.cuda.synchronize()
        return out.cpu().numpy().astype(np.int32)

    @staticmethod
    def DescriptorDistance(a, b) -> int:
        return int(ORBmatcher.DescriptorDistanceMatrix(np.asarray(a).reshape(1, 32), np.asarray(b).reshape(1, 32))[0, 0])

    @staticmethod
    def BruteForce(A, B, groupA=None, groupB=None, maskB=None) -> np.ndarray:
        """Best / second-best of every row of A over B (device); returns BF_DTYPE records."""
        import torch
        L = _lib.lib()
        A = np.ascontiguousarray(A, np.uint8).reshape(-1, 32); B = np.ascontiguousarray(B, np.uint8).reshape(-1, 32)
        dA, dB = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda()
        nA = torch.tensor([len(A)], dtype=torch.int32, device="cuda")
        nB = torch.tensor([len(B)], dtype=torch.int32, device="cuda")
        gA = None if groupA is None else torch.from_numpy(np.ascontiguousarray(groupA, np.int32)).cuda()
        gB = None if groupB is None else torch.from_numpy(np.ascontiguousarray(groupB, np.int32)).cuda()
        mB = None if maskB is None else torch.from_numpy(np.ascontiguousarray(maskB, np.uint8)).cuda()
        out = torch.zeros((max(len(A), 1), 3), dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        _lib.check(L.orbfe_hamming_bf_device(_lib.ptr(dA), _lib.ptr(nA), max(len(A), 1), len(A), _lib.ptr(dB), _lib.ptr(nB),
                                             max(len(B), 1), _lib.ptr(gA), _lib.ptr(gB), _lib.ptr(mB), 1, _lib.ptr(out),
                                             C.c_void_p(None)), "orbfe_hamming_bf_device")
        torch.cuda.synchronize()
        return out.cpu().numpy()[: len(A)].copy().view(BF_DTYPE).reshape(-1)

    # ---- window query (Frame::GetFeaturesInArea + distances)
    def ProjCandidates(self, frame: FrameView, queries: np.ndarray, max_cand: int = 64):
        q = np.ascontiguousarray(queries, QUERY_DTYPE)
        cand = np.zeros((len(q), max_cand), CAND_DTYPE)
        n = np.zeros(len(q), np.int32)
        _lib.check(self._L.orbfe_proj_candidates(C.byref(frame.c), _lib.ptr(q), len(q), _lib.ptr(cand), _lib.ptr(n),
                                                 max_cand), "orbfe_proj_candidates")
        return cand, n

    # ---- SearchByProjection(Frame&, const vector<MapPoint*>&, th)
    def SearchByProjection(self, frame: FrameView, queries: np.ndarray, blocked=None, assigned=None):
        q = np.ascontiguousarray(queries, QUERY_DTYPE)
        blocked = np.zeros(frame.n, np.uint8) if blocked is None else np.ascontiguousarray(blocked, np.uint8).copy()
        assigned = np.full(frame.n, -1, np.int32) if assigned is None else np.ascontiguousarray(assigned, np.int32).copy()
        nm = C.c_int(0)
        _lib.check(self._L.orbfe_search_by_projection_points(C.byref(frame.c), _lib.ptr(q), len(q), self.mfNNratio,
                                                             _lib.ptr(blocked), _lib.ptr(assigned), C.byref(nm)),
                   "orbfe_search_by_projection_points")
        return nm.value, assigned, blocked

    # ---- SearchByProjection(Frame& cur, const Frame& last, th, bMono)
    def SearchByProjectionFrame(self, cur: FrameView, queries: np.ndarray, blocked=None, assigned=None):
        q = np.ascontiguousarray(queries, QUERY_DTYPE)
        blocked = np.zeros(cur.n, np.uint8) if blocked is None else np.ascontiguousarray(blocked, np.uint8).copy()
        assigned = np.full(cur.n, -1, np.int32) if assigned is None else np.ascontiguousarray(assigned, np.int32).copy()
        nm = C.c_int(0)
        _lib.check(self._L.orbfe_search_by_projection_frame(C.byref(cur.c), _lib.ptr(q), len(q),
                                                            int(self.mbCheckOrientation), _lib.ptr(blocked),
                                                            _lib.ptr(assigned), C.byref(nm)),
                   "orbfe_search_by_projection_frame")
        return nm.value, assigned, blocked

    # ---- Tracking::SearchLocalPoints second half: isInFrustum(pMP, 0.5) + SearchByProjection(F, vpMapPoints, th)
    def SearchLocalPoints(self, frame: FrameView, frustum: np.ndarray, points: np.ndarray, th: float = 1.0, blocked=None):
        """points: MAP_POINT_DTYPE records (mvpLocalMapPoints).  Returns (nToMatch, nmatches, track, assigned, blocked);
        track[i] holds what isInFrustum leaves in point i (TRACK_DTYPE)."""
        fr = np.ascontiguousarray(frustum, FRUSTUM_DTYPE).reshape(1)
        mp = np.ascontiguousarray(points, MAP_POINT_DTYPE)
        track = np.zeros(len(mp), TRACK_DTYPE)
        blocked = np.zeros(frame.n, np.uint8) if blocked is None else np.ascontiguousarray(blocked, np.uint8).copy()
        assigned = np.full(frame.n, -1, np.int32)
        ntm, nm = C.c_int(0), C.c_int(0)
        _lib.check(self._L.orbfe_search_local_points(C.byref(frame.c), _lib.ptr(fr), _lib.ptr(mp), len(mp), float(np.float32(th)),
                                                     self.mfNNratio, _lib.ptr(track), _lib.ptr(blocked), _lib.ptr(assigned),
                                                     C.byref(ntm), C.byref(nm)), "orbfe_search_local_points")
        return ntm.value, nm.value, track, assigned, blocked

    # ---- SearchByProjection(Frame& cur, KeyFrame* pKF, sAlreadyFound, th, ORBdist): ORBmatcher.cc:1385-1504
    def SearchByProjectionKeyFrame(self, cur: FrameView, queries: np.ndarray, ORBdist: int, blocked=None):
        """blocked[i2] = cur.mvpMapPoints[i2] is not None; queries carry blocks=1.  Returns (nmatches, assigned, blocked)."""
        q = np.ascontiguousarray(queries, QUERY_DTYPE)
        blocked = np.zeros(cur.n, np.uint8) if blocked is None else np.ascontiguousarray(blocked, np.uint8).copy()
        assigned = np.full(cur.n, -1, np.int32)
        nm = C.c_int(0)
        _lib.check(self._L.orbfe_search_by_projection_keyframe(C.byref(cur.c), _lib.ptr(q), len(q), int(self.mbCheckOrientation),
                                                               int(ORBdist), _lib.ptr(blocked), _lib.ptr(assigned), C.byref(nm)),
                   "orbfe_search_by_projection_keyframe")
        return nm.value, assigned, blocked

    # ---- candidate loop of Fuse / Fuse(Sim3) / SearchBySim3: independent first-minimum per query
    def ProjBest(self, keyframe: FrameView, queries: np.ndarray, inv_level_sigma2=None):
        """inv_level_sigma2 given -> Fuse's chi-square gate (ORBmatcher.cc:833-854), else no gate.  Returns (best_idx, best_dist)."""
        q = np.ascontiguousarray(queries, QUERY_DTYPE)
        bi = np.full(len(q), -1, np.int32); bd = np.full(len(q), 256, np.int32)
        inv = None if inv_level_sigma2 is None else np.ascontiguousarray(inv_level_sigma2, np.float32)
        _lib.check(self._L.orbfe_proj_best(C.byref(keyframe.c), _lib.ptr(q), len(q), 2 if inv is not None else 1, _lib.ptr(inv),
                                           0 if inv is None else len(inv), _lib.ptr(bi), _lib.ptr(bd)), "orbfe_proj_best")
        return bi, bd

    # ---- Fuse / Fuse(Sim3) / SearchBySim3 / SearchByProjection(KF,Scw) / SearchByProjection(Frame,KF,...) from the projection on
    def KeyFrameSearch(self, keyframe: FrameView, camera: np.ndarray, points: np.ndarray, mode: int, inv_level_sigma2=None,
                       blocked=None, max_dist: int = TH_LOW):
        """orbfe_kf_search: projects `points` (KF_POINT_DTYPE) with `camera` (KF_CAMERA_DTYPE), applies the gates of the
        reference function named by `mode` (_lib.KF_*), predicts the level, queries the keyframe's window and picks the
        best descriptor -- all on the device.  Returns (n_matches, results [KF_RESULT_DTYPE], blocked)."""
        cam = np.ascontiguousarray(camera, _lib.KF_CAMERA_DTYPE).reshape(-1)[:1]
        pts = np.ascontiguousarray(points, _lib.KF_POINT_DTYPE)
        res = np.zeros(len(pts), _lib.KF_RESULT_DTYPE)
        inv = None if inv_level_sigma2 is None else np.ascontiguousarray(inv_level_sigma2, np.float32)
        blk = None
        if mode in (_lib.KF_LOOP, _lib.KF_RELOC):
            blk = np.zeros(max(keyframe.n, 1), np.uint8) if blocked is None else np.ascontiguousarray(blocked, np.uint8).copy()
        nm = C.c_int(0)
        _lib.check(self._L.orbfe_kf_search(C.byref(keyframe.c), _lib.ptr(inv), _lib.ptr(cam), _lib.ptr(pts), len(pts), int(mode),
                                           int(self.mbCheckOrientation), int(max_dist), _lib.ptr(blk), _lib.ptr(res), C.byref(nm)),
                   "orbfe_kf_search")
        return nm.value, res, (None if blk is None else blk[:keyframe.n])

    # ---- SearchForInitialization(F1, F2, vbPrevMatched, vnMatches12, windowSize): ORBmatcher.cc:388-492
    def SearchForInitialization(self, f1: FrameView, f2: FrameView, prev_matched: np.ndarray, windowSize: int = 10):
        """Returns (nmatches, vnMatches12, updated vbPrevMatched)."""
        prev = np.ascontiguousarray(prev_matched, np.float32).reshape(-1, 2).copy()
        m12 = np.full(f1.n, -1, np.int32)
        nm = C.c_int(0)
        _lib.check(self._L.orbfe_search_for_initialization(C.byref(f1.c), C.byref(f2.c), _lib.ptr(prev), int(windowSize),
                                                           self.mfNNratio, int(self.mbCheckOrientation), _lib.ptr(m12),
                                                           C.byref(nm)), "orbfe_search_for_initialization")
        return nm.value, m12, prev

    # ---- SearchByBoW(KeyFrame*, Frame&, vector<MapPoint*>&): ORBmatcher.cc:161-273
    def SearchByBoW(self, descA, angleA, validA, groupsA: dict, descB, angleB, groupsB: dict):
        """groupsA/groupsB: {node_id: [feature indices]} (DBoW2::FeatureVector of the keyframe / the frame).
        Runs entirely on the device (one wave per common vocabulary node).  Returns (nmatches, matchB)."""
        descA = np.ascontiguousarray(descA, np.uint8).reshape(-1, 32)
        descB = np.ascontiguousarray(descB, np.uint8).reshape(-1, 32)
        angleA = np.ascontiguousarray(angleA, np.float32); angleB = np.ascontiguousarray(angleB, np.float32)
        validA = np.ascontiguousarray(validA, np.uint8)
        nA, nnA, iA = featvec_arrays(groupsA)
        nB, nnB, iB = featvec_arrays(groupsB)
        matchB = np.full(len(descB), -1, np.int32)
        nm = C.c_int(0)
        _lib.check(self._L.orbfe_search_by_bow(
            _lib.ptr(descA), _lib.ptr(angleA), _lib.ptr(validA), len(descA), C.cast(nA, C.c_void_p), nnA, _lib.ptr(iA),
            _lib.ptr(descB), _lib.ptr(angleB), len(descB), C.cast(nB, C.c_void_p), nnB, _lib.ptr(iB), self.mfNNratio,
            int(self.mbCheckOrientation), _lib.ptr(matchB), C.byref(nm)), "orbfe_search_by_bow")
        return nm.value, matchB


def search_by_bow_kf(descA, angleA, validA, groupsA, descB, angleB, validB, groupsB, nnratio=0.8, check_orientation=True):
    """SearchByBoW(KeyFrame*, KeyFrame*, ...) (ORBmatcher.cc:494-612) on the device.  Returns (nmatches, matchA)."""
    L = _lib.lib()
    descA = np.ascontiguousarray(descA, np.uint8).reshape(-1, 32); descB = np.ascontiguousarray(descB, np.uint8).reshape(-1, 32)
    angleA = np.ascontiguousarray(angleA, np.float32); angleB = np.ascontiguousarray(angleB, np.float32)
    validA = np.ascontiguousarray(validA, np.uint8); validB = np.ascontiguousarray(validB, np.uint8)
    nA, nnA, iA = featvec_arrays(groupsA)
    nB, nnB, iB = featvec_arrays(groupsB)
    matchA = np.full(len(descA), -1, np.int32)
    nm = C.c_int(0)
    _lib.check(L.orbfe_search_by_bow_kf(_lib.ptr(descA), _lib.ptr(angleA), _lib.ptr(validA), len(descA), C.cast(nA, C.c_void_p), nnA,
                                        _lib.ptr(iA), _lib.ptr(descB), _lib.ptr(angleB), _lib.ptr(validB), len(descB),
                                        C.cast(nB, C.c_void_p), nnB, _lib.ptr(iB), float(np.float32(nnratio)), int(check_orientation),
                                        _lib.ptr(matchA), C.byref(nm)), "orbfe_search_by_bow_kf")
    return nm.value, matchA


def search_for_triangulation(keysA, descA, u_rightA, has_mpA, groupsA, keysB, descB, u_rightB, has_mpB, groupsB, epipolar,
                             only_stereo=False, check_orientation=True):
    """SearchForTriangulation (ORBmatcher.cc:614-764) on the device.  Returns (nmatches, matchA); vMatchedPairs are the
    pairs (i, matchA[i]) with matchA[i] >= 0."""
    from ._lib import EPIPOLAR_DTYPE
    L = _lib.lib()
    keysA = np.ascontiguousarray(keysA, KP_DTYPE); keysB = np.ascontiguousarray(keysB, KP_DTYPE)
    descA = np.ascontiguousarray(descA, np.uint8).reshape(-1, 32); descB = np.ascontiguousarray(descB, np.uint8).reshape(-1, 32)
    urA = None if u_rightA is None else np.ascontiguousarray(u_rightA, np.float32)
    urB = None if u_rightB is None else np.ascontiguousarray(u_rightB, np.float32)
    hA = np.ascontiguousarray(has_mpA, np.uint8); hB = np.ascontiguousarray(has_mpB, np.uint8)
    ep = np.ascontiguousarray(epipolar, EPIPOLAR_DTYPE).reshape(1)
    nA, nnA, iA = featvec_arrays(groupsA)
    nB, nnB, iB = featvec_arrays(groupsB)
    matchA = np.full(len(descA), -1, np.int32)
    nm = C.c_int(0)
    _lib.check(L.orbfe_search_for_triangulation(_lib.ptr(keysA), _lib.ptr(descA), _lib.ptr(urA), _lib.ptr(hA), len(descA),
                                                C.cast(nA, C.c_void_p), nnA, _lib.ptr(iA), _lib.ptr(keysB), _lib.ptr(descB), _lib.ptr(urB),
                                                _lib.ptr(hB), len(descB), C.cast(nB, C.c_void_p), nnB, _lib.ptr(iB), _lib.ptr(ep),
                                                int(only_stereo), int(check_orientation), _lib.ptr(matchA), C.byref(nm)),
               "orbfe_search_for_triangulation")
    return nm.value, matchA


def unproject_stereo_batch(kps, desc, n, depth, cams, observed, points, stream):
    """Frame::UnprojectStereo for every keypoint of a batch (torch CUDA tensors): kps (F,cap,28) u8, desc (F,cap,32) u8,
    n (F) i32, depth (F,cap) f32, cams (F,64) u8 [UNPROJECT_CAM_DTYPE], points (F,cap,60) u8 out [LAST_POINT_DTYPE]."""
    F, cap = desc.shape[0], desc.shape[1]
    _lib.check(_lib.lib().orbfe_unproject_stereo_device(F, _lib.ptr(kps), _lib.ptr(desc), _lib.ptr(n), _lib.ptr(depth), cap,
                                                        _lib.ptr(cams), int(observed), _lib.ptr(points), _lib.stream_handle(stream)),
               "orbfe_unproject_stereo_device")


def track_queries_batch(poses, points, n_points, frame_shift, queries, nq, stream):
    """Projection part of SearchByProjection(cur, last) (ORBmatcher.cc:1257-1308): poses (F,128) u8 [TRACK_POSE_DTYPE],
    points (F,pcap,60) u8, n_points (F) i32 -> queries (F,pcap,68) u8, nq (F) i32."""
    F, pcap = points.shape[0], points.shape[1]
    _lib.check(_lib.lib().orbfe_track_queries_device(F, _lib.ptr(poses), _lib.ptr(points), _lib.ptr(n_points), pcap, int(frame_shift),
                                                     _lib.ptr(queries), _lib.ptr(nq), _lib.stream_handle(stream)),
               "orbfe_track_queries_device")


def track_queries_stereo_batch(kps, desc, n, depth, cams, observed, poses, frame_shift, queries, nq, stream, carry=None):
    """unproject_stereo_batch + track_queries_batch in one pass (orbfe_track_queries_stereo_device): the queries of frame f from the
    keypoints / stereo depth of frame f - frame_shift, no point records through memory.  `carry` = (kps (cap,28), desc (cap,32),
    n (1), depth (cap), cam (64)) of the frame in front of the batch, or None: the batch's own tail (index mod F)."""
    F, cap = desc.shape[0], desc.shape[1]
    c = [_lib.ptr(t) for t in carry] if carry is not None else [None] * 5
    _lib.check(_lib.lib().orbfe_track_queries_stereo_device(F, _lib.ptr(kps), _lib.ptr(desc), _lib.ptr(n), _lib.ptr(depth), cap,
                                                            _lib.ptr(cams), int(observed), c[0], c[1], c[2], c[3], c[4], _lib.ptr(poses),
                                                            int(frame_shift), _lib.ptr(queries), _lib.ptr(nq), _lib.stream_handle(stream)),
               "orbfe_track_queries_stereo_device")


def featvec_arrays(groups: dict):
    """{node_id: [indices]} -> (ctypes array of orbfe_featvec_node sorted by id, count, flat int32 index array)."""
    ids = sorted(groups)
    nodes = (_lib.FeatVecNode * max(len(ids), 1))()
    idx = []
    for k, nid in enumerate(ids):
        nodes[k].node_id = nid
        nodes[k].start = len(idx)
        nodes[k].count = len(groups[nid])
        idx.extend(groups[nid])
    return nodes, len(ids), np.asarray(idx if idx else [0], np.int32)


def three_maxima(sizes):
    """ORBmatcher::ComputeThreeMaxima (ORBmatcher.cc:1506-1538)."""
    max1 = max2 = max3 = 0
    i1 = i2 = i3 = -1
    for i, s in enumerate(sizes):
        if s > max1:
            max3, max2, max1 = max2, max1, s
            i3, i2, i1 = i2, i1, i
        elif s > max2:
            max3, max2 = max2, s
            i3, i2 = i2, i
        elif s > max3:
            max3, i3 = s, i
    if max2 < np.float32(0.1) * np.float32(max1):
        i2 = i3 = -1
    elif max3 < np.float32(0.1) * np.float32(max1):
        i3 = -1
    return i1, i2, i3


class Matcher:
    """Work-space handle for the device-resident batch entry points (orbfe_matcher)."""

    def __init__(self, device: int = -1):
        self._L = _lib.lib()
        self._h = C.c_void_p(None)
        _lib.check(self._L.orbfe_matcher_create(device, C.byref(self._h)), "orbfe_matcher_create")

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.orbfe_matcher_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync(self):
        _lib.check(self._L.orbfe_matcher_sync(self._h), "orbfe_matcher_sync")

    def proj_match_batch(self, kps, desc, n, u_right, bounds, queries, nq, mode, nnratio, check_ori, blocked, assigned,
                         n_matches, stream=None):
        """All arguments are torch CUDA tensors: kps (F,cap,28) u8, desc (F,cap,32) u8, n (F) i32, u_right
        (F,cap) f32 or None, queries (F,qcap,68) u8, nq (F) i32, blocked (F,cap) u8, assigned (F,cap) i32,
        n_matches (F) i32."""
        F, cap = desc.shape[0], desc.shape[1]
        s = _lib.stream_handle(stream)
        _lib.check(self._L.orbfe_proj_match_batch_device(
            self._h, F, _lib.ptr(kps), _lib.ptr(desc), _lib.ptr(n), _lib.ptr(u_right), cap, bounds[0], bounds[1], bounds[2],
            bounds[3], _lib.ptr(queries), _lib.ptr(nq), queries.shape[1], mode, nnratio, int(check_ori), _lib.ptr(blocked),
            _lib.ptr(assigned), _lib.ptr(n_matches), s), "orbfe_proj_match_batch_device")

    @staticmethod
    def hamming_bf_batch(descA, nA, descB, nB, groupA, groupB, out, stream=None):
        """The brute force inside SearchByBoW (ORBmatcher.cc:201-222) for a batch of frame pairs, device-resident
        (torch CUDA tensors): descA / descB (S,cap,32) u8, nA / nB (S) i32, groupA / groupB (S,cap) i32 node ids or None,
        out (S,cap,12) u8 = orbfe_bf_match records (best index, best and second-best distance) per row of A."""
        S, cap = descA.shape[0], descA.shape[1]
        _lib.check(_lib.lib().orbfe_hamming_bf_device(_lib.ptr(descA), _lib.ptr(nA), cap, cap, _lib.ptr(descB), _lib.ptr(nB),
                                                      descB.shape[1], _lib.ptr(groupA), _lib.ptr(groupB), None, S, _lib.ptr(out),
                                                      _lib.stream_handle(stream)), "orbfe_hamming_bf_device")

    def search_local_points_batch(self, kps, desc, n, u_right, bounds, frustums, points, n_points, th, nnratio, track,
                                  blocked, assigned, n_to_match, n_matches, stream=None):
        """Tracking::SearchLocalPoints for a batch of frames, everything device-resident (torch CUDA tensors):
        frustums (F,168) u8, points (F,pcap,72) u8, n_points (F) i32, track (F,pcap,24) u8; the rest as proj_match_batch."""
        F, cap = desc.shape[0], desc.shape[1]
        s = _lib.stream_handle(stream)
        _lib.check(self._L.orbfe_search_local_points_batch_device(
            self._h, F, _lib.ptr(kps), _lib.ptr(desc), _lib.ptr(n), _lib.ptr(u_right), cap, bounds[0], bounds[1], bounds[2],
            bounds[3], _lib.ptr(frustums), _lib.ptr(points), _lib.ptr(n_points), points.shape[1], float(np.float32(th)),
            float(np.float32(nnratio)), _lib.ptr(track), _lib.ptr(blocked), _lib.ptr(assigned), _lib.ptr(n_to_match),
            _lib.ptr(n_matches), s), "orbfe_search_local_points_batch_device")

    def stereo_match(self, ex_left, ex_right, kps_l, desc_l, n_l, kps_r, desc_r, n_r, mbf, mb, u_right, depth, n_matched,
                     stream=None):
        """Frame::ComputeStereoMatches for a batch of pairs (torch CUDA tensors, see proj_match_batch)."""
        P, cap = desc_l.shape[0], desc_l.shape[1]
        s = _lib.stream_handle(stream)
        _lib.check(self._L.orbfe_stereo_match_device(
            self._h, ex_left._h, ex_right._h, P, _lib.ptr(kps_l), _lib.ptr(desc_l), _lib.ptr(n_l), _lib.ptr(kps_r),
            _lib.ptr(desc_r), _lib.ptr(n_r), cap, mbf, mb, _lib.ptr(u_right), _lib.ptr(depth), _lib.ptr(n_matched), s),
            "orbfe_stereo_match_device")


def compute_stereo_matches(ex_left, ex_right, kps_l, desc_l, kps_r, desc_r, mbf, mb):
    """Frame::ComputeStereoMatches (L/src/Frame.cc:477-646) for the one pair `ex_left` / `ex_right` extracted last: host numpy
    keypoints / descriptors in, (n_matched, mvuRight, mvDepth) out.  The pyramids of the two extractors are read in HBM."""
    L = _lib.lib()
    kps_l = np.ascontiguousarray(kps_l); kps_r = np.ascontiguousarray(kps_r)
    desc_l = np.ascontiguousarray(desc_l, np.uint8); desc_r = np.ascontiguousarray(desc_r, np.uint8)
    ur = np.empty(len(kps_l), np.float32); depth = np.empty(len(kps_l), np.float32)
    nm = C.c_int(0)
    _lib.check(L.orbfe_stereo_match(ex_left._h, ex_right._h, _lib.ptr(kps_l), _lib.ptr(desc_l), len(kps_l), _lib.ptr(kps_r),
                                    _lib.ptr(desc_r), len(kps_r), float(np.float32(mbf)), float(np.float32(mb)), _lib.ptr(ur),
                                    _lib.ptr(depth), C.byref(nm)), "orbfe_stereo_match")
    return nm.value, ur, depth
