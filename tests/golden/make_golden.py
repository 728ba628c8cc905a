#!/usr/bin/env python3
"""Generates the golden fixtures in tests/golden/ (committed; run from the repo root).

The reference ships no golden vectors and cannot be built here (no OpenCV), so these pin the ORACLE
(oracle/orb_oracle.c, cross-checked stage by stage against tests/np_restatement.py when the fixture is
made): inputs = seeded synthetic images from refactored_orb_slam2_amd.synth (stored, so the fixture does
not depend on numpy's RNG staying stable), outputs = keypoints + descriptors of oo_extract, the FAST
candidate and octree selections of one level, and small matcher vectors with planted ties.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from refactored_orb_slam2_amd import synth  # noqa: E402
from tests import np_restatement as nr  # noqa: E402
from tests import oracle_lib as ol  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def extractor_fixture(name, w, h, nfeat, seq, f):
    img = synth.frame(w, h, seq=seq, f=f)
    e = ol.OracleExtractor(nfeat, 1.2, 8, 20, 7)
    k, d = e(img)
    data = {"image": img, "nfeatures": nfeat, "keypoints": k, "descriptors": d}
    for l in range(8):
        x, y, s = e.level_candidates(l)
        # independent restatement must agree before anything is written
        lv = e.level_pixels(l)
        nx, ny, ns = nr.fast_candidates(lv)
        assert np.array_equal(x, nx) and np.array_equal(y, ny) and np.array_equal(s, ns), (name, l)
        if l > 0:
            assert np.array_equal(nr.resize_linear(e.level_pixels(l - 1), lv.shape[1], lv.shape[0]), lv)
        assert np.array_equal(nr.gaussian_blur7(lv), e.level_blurred(l))
        sel = nr.distribute_octree(x, y, s, 16, lv.shape[1] - 16, 16, lv.shape[0] - 16, e.features_per_level[l])
        kl = e.level_keypoints(l)
        assert np.array_equal(x[sel] + 16, kl["x"].astype(np.int64)) and np.array_equal(y[sel] + 16, kl["y"].astype(np.int64))
        data[f"cand_n_{l}"] = np.int32(len(x))
        data[f"kp_n_{l}"] = np.int32(len(kl))
        if l in (0, 7):
            data[f"cand_{l}"] = np.stack([x, y, s]).astype(np.int16)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **data)
    print(name, img.shape, len(k), "keypoints")
    return k, d, e.scale_factors


def matcher_fixture(k0, d0, k1, d1, sf, w, h):
    rng = np.random.default_rng(77)
    q = np.zeros(len(k0), ol.QUERY_DTYPE)
    q["u"] = k0["x"] - np.float32(2.0) + rng.normal(0, 0.5, len(k0)).astype(np.float32)
    q["v"] = k0["y"] + rng.normal(0, 0.5, len(k0)).astype(np.float32)
    q["u_r"] = q["u"] - np.float32(15)
    q["radius"] = np.float32(15.0) * sf[k0["octave"]]
    q["min_level"] = k0["octave"] - 1
    q["max_level"] = k0["octave"] + 1
    q["valid"] = (rng.random(len(k0)) < 0.95).astype(np.int32)
    q["blocks"] = (rng.random(len(k0)) < 0.6).astype(np.int32)
    q["angle"] = k0["angle"]
    q["desc"] = d0
    ur = np.where(rng.random(len(k1)) < 0.5, k1["x"] - np.float32(15) + rng.normal(0, 3, len(k1)).astype(np.float32), -1).astype(np.float32)
    of = ol.OracleFrame(k1, d1, sf, 0, w, 0, h, ur)
    nm1, a1, b1 = of.search_by_projection_frame(q, True)
    q2 = q.copy()
    q2["max_level"] = k0["octave"]
    nm0, a0, b0 = of.search_by_projection_points(q2, np.float32(0.8))
    # window enumeration order for the first 64 queries
    wins = [of.features_in_area(float(q["u"][i]), float(q["v"][i]), float(q["radius"][i]), int(q["min_level"][i]), int(q["max_level"][i])) for i in range(64)]
    win_n = np.array([len(x) for x in wins], np.int32)
    win_idx = np.concatenate(wins).astype(np.int32) if wins else np.zeros(0, np.int32)
    np.savez_compressed(os.path.join(OUT, "matcher_tum.npz"), k0=k0, d0=d0, k1=k1, d1=d1, sf=sf, w=w, h=h, queries=q, u_right=ur,
                        frame_nm=np.int32(nm1), frame_assigned=a1, frame_blocked=b1, points_nm=np.int32(nm0), points_assigned=a0,
                        points_blocked=b0, win_n=win_n, win_idx=win_idx, cell_start=of.cell_start)
    print("matcher_tum", nm1, nm0)


def tracking_fixture():
    """Tracking-side functions added after the first fixtures (isInFrustum / SearchLocalPoints, UnprojectStereo + the
    projection of SearchByProjection(cur,last), the relocalisation search, SearchByBoW(KF,KF)); inputs are stored, the
    keypoints come from the committed TUM extraction fixtures.  Run: python tests/golden/make_golden.py tracking"""
    from refactored_orb_slam2_amd.matcher import make_frustum
    g0 = np.load(os.path.join(OUT, "extract_tum_640x480_1000_f0.npz")); g1 = np.load(os.path.join(OUT, "extract_tum_640x480_1000_f1.npz"))
    k0, d0, k1, d1 = g0["keypoints"], g0["descriptors"], g1["keypoints"], g1["descriptors"]
    sf = ol.OracleExtractor(1000).scale_factors
    w, h = 640, 480
    rng = np.random.default_rng(2024)
    R, t = synth.camera_pose(31)
    fr = make_frustum(R, t, 517.3, 516.5, 318.6, 255.3, 40.0, (0, w, 0, h), 1.2, 8).astype(ol.FRUSTUM_DTYPE)
    mp = synth.local_map(k1, d1, fr, 32, n_extra=200).astype(ol.MAP_POINT_DTYPE)
    blocked0 = (rng.random(len(k1)) < 0.1).astype(np.uint8)
    ur = np.where(rng.random(len(k1)) < 0.3, k1["x"] - np.float32(25), -1).astype(np.float32)
    of = ol.OracleFrame(k1, d1, sf, 0, w, 0, h, ur)
    ntm, nm, track, assigned, blocked = of.search_local_points(fr, mp, np.float32(3.0), np.float32(0.8), blocked0)
    # UnprojectStereo of frame 0 + projection into frame 1
    cam = np.zeros(1, ol.UNPROJECT_CAM_DTYPE); pose = np.zeros(1, ol.TRACK_POSE_DTYPE)
    cam["Rwc"][0] = R.T.reshape(9); cam["Ow"][0] = -(R.T @ t); cam["cx"] = 318.6; cam["cy"] = 255.3
    cam["invfx"] = np.float32(1) / np.float32(517.3); cam["invfy"] = np.float32(1) / np.float32(516.5)
    depth = np.where(rng.random(len(k0)) < 0.75, rng.uniform(1, 30, len(k0)), -1).astype(np.float32)
    pts = ol.unproject_stereo(cam, k0, d0, depth)
    pose["Rcw"][0] = R.reshape(9); pose["tcw"][0] = t + np.array([0.01, 0.0, 0.02], np.float32)
    pose["fx"] = 517.3; pose["fy"] = 516.5; pose["cx"] = 316.6; pose["cy"] = 255.3; pose["mbf"] = 40.0
    pose["max_x"] = w; pose["max_y"] = h; pose["th"] = 15.0; pose["scale_factors"][0, :len(sf)] = sf
    tq = ol.track_queries(pose, pts)
    t_nm, t_assigned, _ = ol.OracleFrame(k1, d1, sf, 0, w, 0, h, None).search_by_projection_frame(tq, True)
    # relocalisation search with the same queries, caller's distance bound
    k_nm, k_assigned, k_blocked = of.search_by_projection_keyframe(tq, True, 64, blocked0)
    # SearchByBoW(KF, KF) on 64 synthetic vocabulary buckets
    ga = {}; gb = {}
    for i, d in enumerate(d0):
        ga.setdefault(int(d[0]) % 64, []).append(i)
    for i, d in enumerate(d1):
        gb.setdefault(int(d[0]) % 64, []).append(i)
    validA = (rng.random(len(d0)) < 0.9).astype(np.uint8); validB = (rng.random(len(d1)) < 0.9).astype(np.uint8)
    b_nm, b_matchA = ol.search_by_bow_kf(d0, k0["angle"], validA, ga, d1, k1["angle"], validB, gb, np.float32(0.9), True)
    np.savez_compressed(os.path.join(OUT, "tracking_tum.npz"), frustum=fr, map_points=mp, blocked0=blocked0, u_right=ur,
                        lp_n_to_match=np.int32(ntm), lp_nm=np.int32(nm), lp_track=track, lp_assigned=assigned, lp_blocked=blocked,
                        cam=cam, pose=pose, depth=depth, last_points=pts, track_queries=tq, track_nm=np.int32(t_nm),
                        track_assigned=t_assigned, kf_nm=np.int32(k_nm), kf_assigned=k_assigned, kf_blocked=k_blocked,
                        validA=validA, validB=validB, bow_nm=np.int32(b_nm), bow_matchA=b_matchA)
    print("tracking_tum", ntm, nm, t_nm, k_nm, b_nm)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "tracking":
        tracking_fixture()
        sys.exit(0)
    extractor_fixture("extract_kitti_1241x376_2000", 1241, 376, 2000, seq=11, f=0)
    k0, d0, sf = extractor_fixture("extract_tum_640x480_1000_f0", 640, 480, 1000, seq=12, f=0)
    k1, d1, _ = extractor_fixture("extract_tum_640x480_1000_f1", 640, 480, 1000, seq=12, f=1)
    matcher_fixture(k0, d0, k1, d1, sf, 640, 480)
