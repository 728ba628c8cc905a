"""GPU parity of the matcher path (HIP kernels through the C ABI) against the CPU oracle.  Bit-exact bar:
distances, candidate lists and their order, assignments, match counts; stereo u_right/depth floats equal.
"""
import numpy as np
import pytest

from refactored_orb_slam2_amd import ORBextractor, synth
from refactored_orb_slam2_amd.matcher import FrameView, Matcher, ORBmatcher, make_queries
from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu


def _np_dist(A, B):
    x = np.bitwise_xor(A[:, None, :], B[None, :, :])
    return np.unpackbits(x, axis=2).sum(axis=2).astype(np.int32)


def _rand_desc(rng, n):
    return rng.integers(0, 256, size=(n, 32), dtype=np.uint8)


def test_hamming_matrix_and_known_answers():
    rng = np.random.default_rng(1)
    A = _rand_desc(rng, 300)
    B = _rand_desc(rng, 517)
    A[0] = 0
    B[0] = 255          # all-zeros vs all-ones = 256
    B[1] = A[5]         # planted duplicate
    D = ORBmatcher.DescriptorDistanceMatrix(A, B)
    assert D[0, 0] == 256 and D[5, 1] == 0
    np.testing.assert_array_equal(D, _np_dist(A, B))
    for i, j in [(0, 0), (5, 1), (17, 300), (299, 516)]:
        assert D[i, j] == ol.descriptor_distance(A[i], B[j])
    assert ORBmatcher.DescriptorDistance(A[3], B[4]) == ol.descriptor_distance(A[3], B[4])


def _seq_best_second(Drow, js):
    b1, b2, bi = 256, 256, -1
    for j in js:
        d = int(Drow[j])
        if d < b1:
            b2, b1, bi = b1, d, j
        elif d < b2:
            b2 = d
    return bi, b1, b2


@pytest.mark.parametrize("nA,nB", [(1, 1), (33, 700), (1000, 1000), (257, 2049)])
def test_brute_force_best_second(nA, nB):
    rng = np.random.default_rng(nA * 7 + nB)
    A = _rand_desc(rng, nA)
    B = _rand_desc(rng, nB)
    if nB > 40:  # ties: identical descriptors at several j -> first index must win, second == best
        B[30] = B[7]
        B[nB - 1] = B[7]
        A[0] = B[7]
    D = _np_dist(A, B)
    out = ORBmatcher.BruteForce(A, B)
    for i in range(nA):
        bi, b1, b2 = _seq_best_second(D[i], range(nB))
        assert (out[i]["best_idx"], out[i]["best_dist"], out[i]["second_dist"]) == (bi, b1, b2), i
    # grouped (vocabulary node ids) + mask of already matched
    gA = rng.integers(0, 10, nA).astype(np.int32)
    gB = rng.integers(0, 10, nB).astype(np.int32)
    mB = (rng.random(nB) < 0.2).astype(np.uint8)
    out = ORBmatcher.BruteForce(A, B, gA, gB, mB)
    for i in range(nA):
        js = [j for j in range(nB) if gB[j] == gA[i] and not mB[j]]
        bi, b1, b2 = _seq_best_second(D[i], js)
        assert (out[i]["best_idx"], out[i]["best_dist"], out[i]["second_dist"]) == (bi, b1, b2), i


@pytest.mark.parametrize("nA,nB,ngroups", [(1000, 1000, 100), (600, 3000, 2500), (300, 9000, 64), (257, 700, 1)])
def test_brute_force_grouped_by_node_id(nA, nB, ngroups):
    """The SearchByBoW use of the brute force: vocabulary node ids as groups.  Wide id range (hash collisions inside the kernel's
    bucket table), many more groups than buckets, a B set beyond the in-LDS sort (plain scan with the group test), one group."""
    rng = np.random.default_rng(nA + 3 * nB + ngroups)
    A = _rand_desc(rng, nA); B = _rand_desc(rng, nB)
    ids = rng.integers(-2**31, 2**31 - 1, ngroups, dtype=np.int64).astype(np.int32)   # arbitrary ints, negative ones included
    gA = ids[rng.integers(0, ngroups, nA)]; gB = ids[rng.integers(0, ngroups, nB)]
    A[:50] = B[rng.integers(0, nB, 50)]                      # exact duplicates: ties on distance 0 across and inside groups
    B[nB // 2] = B[3]; gB[nB // 2] = gB[3]
    mB = (rng.random(nB) < 0.1).astype(np.uint8)
    D = _np_dist(A, B)
    for mask in (None, mB):
        out = ORBmatcher.BruteForce(A, B, gA, gB, mask)
        for i in range(nA):
            js = [j for j in np.nonzero(gB == gA[i])[0] if mask is None or not mask[j]]
            bi, b1, b2 = _seq_best_second(D[i], js)
            assert (out[i]["best_idx"], out[i]["best_dist"], out[i]["second_dist"]) == (bi, b1, b2), (i, mask is None)


def _two_frames(w, h, nfeat, with_right=False, seed=5):
    ex = ORBextractor(nfeat)
    seq = synth.sequence(w, h, 2, seq=seed)
    (k0, d0), (k1, d1) = ex.extract_batch(seq)
    sf = ex.GetScaleFactors()
    ex.close()
    return k0, d0, k1, d1, sf


def _queries_from_last(k0, d0, sf, th, shift=(-2.0, 0.0), rng=None, blocks_p=0.7):
    rng = rng or np.random.default_rng(3)
    q = make_queries(len(k0))
    q["u"] = k0["x"] + np.float32(shift[0]) + rng.normal(0, 0.7, len(k0)).astype(np.float32)
    q["v"] = k0["y"] + np.float32(shift[1]) + rng.normal(0, 0.7, len(k0)).astype(np.float32)
    q["u_r"] = q["u"] - np.float32(20.0)
    q["radius"] = np.float32(th) * sf[k0["octave"]]
    q["min_level"] = k0["octave"] - 1
    q["max_level"] = k0["octave"] + 1
    q["valid"] = (rng.random(len(k0)) < 0.95).astype(np.int32)
    q["blocks"] = (rng.random(len(k0)) < blocks_p).astype(np.int32)
    q["angle"] = k0["angle"]
    q["desc"] = d0
    return q


@pytest.mark.parametrize("geom", [(1241, 376, 2000), (640, 480, 1000)])
def test_proj_candidates_order_and_distances(geom):
    w, h, nf = geom
    k0, d0, k1, d1, sf = _two_frames(w, h, nf)
    rng = np.random.default_rng(11)
    ur = np.where(rng.random(len(k1)) < 0.6, k1["x"] - np.float32(20.0) + rng.normal(0, 4, len(k1)).astype(np.float32), np.float32(-1)).astype(np.float32)
    q = _queries_from_last(k0, d0, sf, 15.0, rng=rng)
    q["min_level"][::7] = 0
    q["max_level"][::7] = -1     # bCheckLevels false
    q["max_level"][3::11] = -1   # only a lower bound
    fv = FrameView(k1, d1, 0, w, 0, h, ur)
    of = ol.OracleFrame(k1, d1, sf, 0, w, 0, h, ur)
    m = ORBmatcher()
    cand, n = m.ProjCandidates(fv, q, max_cand=256)
    tot = 0
    for i in range(len(q)):
        if not q["valid"][i]:
            assert n[i] == 0
            continue
        idx = of.features_in_area(float(q["u"][i]), float(q["v"][i]), float(q["radius"][i]), int(q["min_level"][i]), int(q["max_level"][i]))
        keep = [j for j in idx if not (ur[j] > 0 and abs(np.float32(q["u_r"][i]) - ur[j]) > q["radius"][i])]
        assert n[i] == len(keep), i
        np.testing.assert_array_equal(cand[i, : n[i]]["idx"], np.asarray(keep, np.int32), err_msg=f"query {i}")
        for c in range(0, n[i], 5):
            assert cand[i, c]["dist"] == ol.descriptor_distance(q["desc"][i], d1[keep[c]])
        tot += n[i]
    assert tot > len(q)  # the test exercises real windows


@pytest.mark.parametrize("n", [1500, 9000, 20500])
def test_grid_of_crowded_and_large_frames(n):
    """The Frame grid (Frame.cc:250-263) from random keypoints instead of extracted ones: a quarter of them piled into two cells
    (long cell lists: the rank-based placement), some outside the image bounds (no cell), 20 500 keypoints (beyond the LDS-resident
    form: the kernel that sorts the cell lists in global memory).  Windows must list the same indices in the same order as the
    oracle's GetFeaturesInArea."""
    rng = np.random.default_rng(n)
    w, h = 1241, 376
    k = np.zeros(n, ol.KP_DTYPE)
    k["x"] = rng.uniform(-8, w + 8, n).astype(np.float32); k["y"] = rng.uniform(-8, h + 8, n).astype(np.float32)
    pile = rng.random(n) < (0.25 if n < 20000 else 0.03)   # (the global-memory form sorts a cell list serially: keep its piles moderate)
    k["x"][pile] = (np.where(rng.random(pile.sum()) < 0.5, 300.0, 911.5) + rng.uniform(-6, 6, pile.sum())).astype(np.float32)
    k["y"][pile] = (150.0 + rng.uniform(-3, 3, pile.sum())).astype(np.float32)
    k["octave"] = rng.integers(0, 8, n); k["angle"] = rng.uniform(0, 360, n).astype(np.float32); k["size"] = 31; k["class_id"] = -1
    d = _rand_desc(rng, n)
    sf = np.array([np.float32(1.2) ** i for i in range(8)], np.float32)
    fv = FrameView(k, d, 0, w, 0, h); of = ol.OracleFrame(k, d, sf, 0, w, 0, h)
    nq = 96
    q = make_queries(nq)
    q["u"] = np.concatenate([rng.uniform(0, w, nq - 16), np.full(8, 300.0), np.full(8, 911.5)]).astype(np.float32)
    q["v"] = np.concatenate([rng.uniform(0, h, nq - 16), np.full(16, 150.0)]).astype(np.float32)
    q["u_r"] = q["u"]; q["radius"] = rng.uniform(4, 30, nq).astype(np.float32)
    q["min_level"] = -1; q["max_level"] = -1
    q["min_level"][::3] = 2; q["max_level"][::3] = 5
    q["valid"] = 1; q["blocks"] = 1; q["desc"] = _rand_desc(rng, nq)
    cand, cnt = ORBmatcher().ProjCandidates(fv, q, max_cand=4096)
    for i in range(nq):
        idx = of.features_in_area(float(q["u"][i]), float(q["v"][i]), float(q["radius"][i]), int(q["min_level"][i]), int(q["max_level"][i]))
        assert cnt[i] == len(idx), i
        m_ = min(len(idx), 4096)
        np.testing.assert_array_equal(cand[i, :m_]["idx"], np.asarray(idx[:m_], np.int32), err_msg=f"query {i}")
    assert cnt.max() > 100


@pytest.mark.parametrize("geom,th", [((1241, 376, 2000), 7.0), ((640, 480, 1000), 15.0), ((640, 480, 1000), 40.0)])
def test_search_by_projection_frame(geom, th):
    w, h, nf = geom
    k0, d0, k1, d1, sf = _two_frames(w, h, nf)
    rng = np.random.default_rng(5)
    ur = np.where(rng.random(len(k1)) < 0.5, k1["x"] - np.float32(20.0) + rng.normal(0, 3, len(k1)).astype(np.float32), np.float32(-1)).astype(np.float32)
    q = _queries_from_last(k0, d0, sf, th, rng=rng, blocks_p=0.5)
    blocked0 = (rng.random(len(k1)) < 0.05).astype(np.uint8)
    fv = FrameView(k1, d1, 0, w, 0, h, ur)
    of = ol.OracleFrame(k1, d1, sf, 0, w, 0, h, ur)
    for check in (True, False):
        nm, assigned, blocked = ORBmatcher(0.9, check).SearchByProjectionFrame(fv, q, blocked0)
        onm, oassigned, oblocked = of.search_by_projection_frame(q, check, blocked0)
        assert nm == onm
        np.testing.assert_array_equal(assigned, oassigned)
        np.testing.assert_array_equal(blocked, oblocked)
        assert nm > 0.3 * len(k1)


def _frustum_and_map(k1, d1, w, h, seed, n_extra=500):
    from refactored_orb_slam2_amd.matcher import make_frustum
    from refactored_orb_slam2_amd import synth
    R, t = synth.camera_pose(seed)
    fr = make_frustum(R, t, 718.856, 718.856, w / 2 + 3.2, h / 2 - 1.7, 386.1448, (0, w, 0, h), 1.2, 8)
    return fr, synth.local_map(k1, d1, fr, seed + 1, n_extra)


@pytest.mark.parametrize("geom,th", [((1241, 376, 2000), 1.0), ((640, 480, 1000), 3.0), ((752, 480, 1200), 5.0)])
def test_search_local_points(geom, th):
    """Tracking::SearchLocalPoints: isInFrustum + PredictScale + SearchByProjection(F, points, th), queries built on the device"""
    w, h, nf = geom
    _, _, k1, d1, sf = _two_frames(w, h, nf)
    rng = np.random.default_rng(31)
    ur = np.where(rng.random(len(k1)) < 0.3, k1["x"] - np.float32(30.0) + rng.normal(0, 3, len(k1)).astype(np.float32), np.float32(-1)).astype(np.float32)
    fr, mp = _frustum_and_map(k1, d1, w, h, 77)
    blocked0 = (rng.random(len(k1)) < 0.1).astype(np.uint8)
    fv = FrameView(k1, d1, 0, w, 0, h, ur)
    of = ol.OracleFrame(k1, d1, sf, 0, w, 0, h, ur)
    ntm, nm, track, assigned, blocked = ORBmatcher(0.8).SearchLocalPoints(fv, fr, mp, th, blocked0)
    ontm, onm, otrack, oassigned, oblocked = of.search_local_points(fr, mp, np.float32(th), np.float32(0.8), blocked0)
    assert track.tobytes() == otrack.tobytes()          # u, v, u_r, level, viewCos bit for bit
    assert ntm == ontm and nm == onm
    np.testing.assert_array_equal(assigned, oassigned)
    np.testing.assert_array_equal(blocked, oblocked)
    # the synthetic map exercises every rejection branch and the level clamp
    assert 0.3 * len(mp) < ntm < 0.9 * len(mp) and nm > 0.2 * len(k1)
    assert len(np.unique(track["level"][track["in_view"] == 1])) == 8


def test_search_local_points_edge_cases():
    w, h = 640, 480
    _, _, k1, d1, sf = _two_frames(w, h, 1000)
    fr, mp = _frustum_and_map(k1, d1, w, h, 5, n_extra=50)
    fv = FrameView(k1, d1, 0, w, 0, h)
    of = ol.OracleFrame(k1, d1, sf, 0, w, 0, h, None)
    m = ORBmatcher(0.8)
    # no points / all skipped / nothing in view
    assert m.SearchLocalPoints(fv, fr, mp[:0])[:2] == (0, 0)
    sk = mp.copy(); sk["skip"] = 1
    r = m.SearchLocalPoints(fv, fr, sk)
    assert r[:2] == (0, 0) and not r[2]["in_view"].any() and np.all(r[3] == -1)
    # degenerate geometry: dist == 0 with min_distance 0, max_distance 0 / inf / nan, point on the camera plane
    deg = mp[:8].copy(); deg["skip"] = 0
    deg["pos"][0] = fr["Ow"][0]; deg["min_distance"][0] = 0
    deg["max_distance"][1] = 0; deg["min_distance"][1] = 0
    deg["max_distance"][2] = np.inf
    deg["max_distance"][3] = np.nan
    deg["normal"][4] = 0
    got = m.SearchLocalPoints(fv, fr, deg)
    exp = of.search_local_points(fr, deg, np.float32(1.0), np.float32(0.8))
    assert got[2].tobytes() == exp[2].tobytes() and got[0] == exp[0] and got[1] == exp[1]
    # more pyramid levels than the frustum record holds are rejected, not read out of bounds
    from refactored_orb_slam2_amd import _lib
    bad = fr.copy(); bad["n_levels"] = 17
    with pytest.raises(_lib.OrbfeError):
        m.SearchLocalPoints(fv, bad, mp)
    # a frame without keypoints still gets its track records
    fv0 = FrameView(k1[:0], d1[:0], 0, w, 0, h)
    of0 = ol.OracleFrame(k1[:0], d1[:0], sf, 0, w, 0, h, None)
    got = m.SearchLocalPoints(fv0, fr, mp)
    exp = of0.search_local_points(fr, mp, np.float32(1.0), np.float32(0.8))
    assert got[2].tobytes() == exp[2].tobytes() and got[0] == exp[0] > 0 and got[1] == 0


def test_search_local_points_batch_device():
    """batched form: per-frame pose and map, everything resident in HBM, on a caller stream"""
    import torch
    from refactored_orb_slam2_amd.matcher import Matcher
    from refactored_orb_slam2_amd._lib import TRACK_DTYPE
    w, h, F = 752, 480, 3
    frames = [_two_frames(w, h, 1200, seed=s)[2:] for s in range(F)]
    cap = max(len(f[0]) for f in frames) + 5
    maps, frs = [], []
    for i, (k1, d1, sf) in enumerate(frames):
        fr, mp = _frustum_and_map(k1, d1, w, h, 100 + i, n_extra=300 + 100 * i)
        maps.append(mp); frs.append(fr)
    pcap = max(len(m) for m in maps) + 3
    kps = np.zeros((F, cap), ol.KP_DTYPE); desc = np.zeros((F, cap, 32), np.uint8); n = np.zeros(F, np.int32)
    pts = np.zeros((F, pcap), maps[0].dtype); npts = np.zeros(F, np.int32)
    for i, (k1, d1, sf) in enumerate(frames):
        kps[i, :len(k1)] = k1; desc[i, :len(k1)] = d1; n[i] = len(k1)
        pts[i, :len(maps[i])] = maps[i]; npts[i] = len(maps[i])
    dev = lambda a: torch.from_numpy(a.view(np.uint8).reshape(a.shape + (-1,)) if a.dtype.names else a).cuda()
    t_kps, t_desc, t_n, t_pts, t_np = dev(kps), dev(desc), dev(n), dev(pts), dev(npts)
    t_fr = dev(np.concatenate(frs))
    t_track = torch.zeros((F, pcap, 24), dtype=torch.uint8, device="cuda")
    t_blocked = torch.zeros((F, cap), dtype=torch.uint8, device="cuda")
    t_assigned = torch.full((F, cap), -1, dtype=torch.int32, device="cuda")
    t_ntm = torch.zeros(F, dtype=torch.int32, device="cuda"); t_nm = torch.zeros(F, dtype=torch.int32, device="cuda")
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    m = Matcher()
    with torch.cuda.stream(s):
        m.search_local_points_batch(t_kps, t_desc, t_n, None, (0, w, 0, h), t_fr, t_pts, t_np, 1.0, 0.8, t_track, t_blocked,
                                    t_assigned, t_ntm, t_nm, stream=s)
    s.synchronize()
    for i, (k1, d1, sf) in enumerate(frames):
        of = ol.OracleFrame(k1, d1, sf, 0, w, 0, h, None)
        ontm, onm, otrack, oassigned, oblocked = of.search_local_points(frs[i], maps[i], np.float32(1.0), np.float32(0.8))
        gtrack = t_track[i].cpu().numpy().reshape(-1).view(TRACK_DTYPE)[: len(maps[i])]
        assert gtrack.tobytes() == otrack.tobytes()
        assert int(t_ntm[i]) == ontm and int(t_nm[i]) == onm and onm > 100
        np.testing.assert_array_equal(t_assigned[i].cpu().numpy()[: len(k1)], oassigned)
        np.testing.assert_array_equal(t_blocked[i].cpu().numpy()[: len(k1)], oblocked)
    m.close()


@pytest.mark.parametrize("mode", ["window", "forward", "backward"])
def test_unproject_and_track_queries_device(mode):
    """Frame::UnprojectStereo + the projection part of SearchByProjection(cur, last) on the device, then the search itself:
    records, queries and matches equal the oracle's"""
    import torch
    from refactored_orb_slam2_amd import synth
    from refactored_orb_slam2_amd.matcher import Matcher, unproject_stereo_batch, track_queries_batch
    from refactored_orb_slam2_amd._lib import LAST_POINT_DTYPE, QUERY_DTYPE
    w, h, F = 752, 480, 3
    frames = [_two_frames(w, h, 1200, seed=s)[2:] for s in range(F)]
    cap = max(len(f[0]) for f in frames) + 7
    rng = np.random.default_rng(4)
    kps = np.zeros((F, cap), ol.KP_DTYPE); desc = np.zeros((F, cap, 32), np.uint8); n = np.zeros(F, np.int32)
    depth = np.full((F, cap), -1, np.float32)
    cams = np.zeros(F, ol.UNPROJECT_CAM_DTYPE); poses = np.zeros(F, ol.TRACK_POSE_DTYPE)
    sf = frames[0][2]
    for i, (k1, d1, _) in enumerate(frames):
        kps[i, :len(k1)] = k1; desc[i, :len(k1)] = d1; n[i] = len(k1)
        z = rng.uniform(2, 60, len(k1)).astype(np.float32); z[rng.random(len(k1)) < 0.25] = -1   # no stereo match
        z[:3] = [0.0, 1e-30, 1e30]
        depth[i, :len(k1)] = z
        R, t = synth.camera_pose(200 + i)
        cams["Rwc"][i] = R.T.reshape(9); cams["Ow"][i] = -(R.T @ t)
        cams["cx"][i] = 370.0; cams["cy"][i] = 236.5; cams["invfx"][i] = np.float32(1) / np.float32(458.654); cams["invfy"][i] = np.float32(1) / np.float32(457.296)
        # the current frame of frame i's points is frame i+1: nearly the same pose (small motion), so most points project inside
        R2, t2 = synth.camera_pose(200 + i)
        t2 = t2 + np.array([0.05, -0.02, 0.3 if mode == "forward" else (-0.3 if mode == "backward" else 0.0)], np.float32)
        j = (i + 1) % F
        poses["Rcw"][j] = R2.reshape(9); poses["tcw"][j] = t2
        poses["fx"][j] = 458.654; poses["fy"][j] = 457.296; poses["cx"][j] = 367.215; poses["cy"][j] = 248.375; poses["mbf"][j] = 47.9
        poses["min_x"][j] = 0; poses["max_x"][j] = w; poses["min_y"][j] = 0; poses["max_y"][j] = h
        poses["forward"][j] = mode == "forward"; poses["backward"][j] = mode == "backward"
        poses["th"][j] = 15.0; poses["scale_factors"][j, :len(sf)] = sf
    dev = lambda a: torch.from_numpy(a.view(np.uint8).reshape(a.shape + (-1,)) if a.dtype.names else a).cuda()
    t_kps, t_desc, t_n, t_depth, t_cams, t_poses = dev(kps), dev(desc), dev(n), dev(depth), dev(cams), dev(poses)
    t_pts = torch.zeros((F, cap, 60), dtype=torch.uint8, device="cuda")
    t_q = torch.zeros((F, cap, 68), dtype=torch.uint8, device="cuda")
    t_nq = torch.zeros(F, dtype=torch.int32, device="cuda")
    t_blocked = torch.zeros((F, cap), dtype=torch.uint8, device="cuda")
    t_assigned = torch.full((F, cap), -1, dtype=torch.int32, device="cuda")
    t_nm = torch.zeros(F, dtype=torch.int32, device="cuda")
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    m = Matcher()
    with torch.cuda.stream(s):
        unproject_stereo_batch(t_kps, t_desc, t_n, t_depth, t_cams, 1, t_pts, s)
        track_queries_batch(t_poses, t_pts, t_n, 1, t_q, t_nq, s)
        m.proj_match_batch(t_kps, t_desc, t_n, None, (0.0, float(w), 0.0, float(h)), t_q, t_nq, 1, 0.9, True, t_blocked, t_assigned,
                           t_nm, stream=s)
    s.synchronize()
    g_pts = t_pts.cpu().numpy().reshape(F, cap * 60).view(LAST_POINT_DTYPE).reshape(F, cap)
    g_q = t_q.cpu().numpy().reshape(F, cap * 68).view(QUERY_DTYPE).reshape(F, cap)
    tot_valid = 0
    for i, (k1, d1, _) in enumerate(frames):
        opts = ol.unproject_stereo(cams[i:i + 1], k1, d1, depth[i])
        assert g_pts[i, :len(k1)].tobytes() == opts.tobytes()
        j = (i + 1) % F
        oq = ol.track_queries(poses[j:j + 1], opts)
        assert int(t_nq[j]) == len(k1)
        assert g_q[j, :len(k1)].tobytes() == oq.tobytes()
        tot_valid += int(oq["valid"].sum())
        kj, dj, _ = frames[j]
        of = ol.OracleFrame(kj, dj, sf, 0, w, 0, h, None)
        onm, oassigned, oblocked = of.search_by_projection_frame(oq, True)
        assert int(t_nm[j]) == onm
        np.testing.assert_array_equal(t_assigned[j].cpu().numpy()[: len(kj)], oassigned)
    assert tot_valid > 0.4 * int(n.sum())
    # ---- the same queries in ONE pass (orbfe_track_queries_stereo_device: no point records in between): every byte of the query array,
    #      rows behind a frame's count included.  Frame 0's source: the batch's tail (as above), or a carry frame holding the same data.
    from refactored_orb_slam2_amd.matcher import track_queries_stereo_batch
    for carry in (None, (t_kps[F - 1].clone(), t_desc[F - 1].clone(), t_n[F - 1:].clone(), t_depth[F - 1].clone(), t_cams[F - 1].clone())):
        t_q2 = torch.full((F, cap, 68), 0xAB, dtype=torch.uint8, device="cuda")
        t_nq2 = torch.full((F,), -5, dtype=torch.int32, device="cuda")
        with torch.cuda.stream(s):
            track_queries_stereo_batch(t_kps, t_desc, t_n, t_depth, t_cams, 1, t_poses, 1, t_q2, t_nq2, s, carry=carry)
        s.synchronize()
        assert torch.equal(t_q2, t_q) and torch.equal(t_nq2, t_nq)
        for i, (k1, d1, _) in enumerate(frames):   # ... and against the oracle directly
            j = (i + 1) % F
            oq = ol.track_queries(poses[j:j + 1], ol.unproject_stereo(cams[i:i + 1], k1, d1, depth[i]))
            assert t_q2[j].cpu().numpy().reshape(cap * 68).view(QUERY_DTYPE)[:len(k1)].tobytes() == oq.tobytes()
    # frame_shift 0: every frame's own keypoints with its own pose
    with torch.cuda.stream(s):
        track_queries_batch(t_poses, t_pts, t_n, 0, t_q, t_nq, s)
        t_q2.fill_(0xCD); t_nq2.fill_(-7)
        track_queries_stereo_batch(t_kps, t_desc, t_n, t_depth, t_cams, 1, t_poses, 0, t_q2, t_nq2, s)
    s.synchronize()
    assert torch.equal(t_q2, t_q) and torch.equal(t_nq2, t_nq)
    m.close()


@pytest.mark.parametrize("th,orb_dist", [(10.0, 100), (3.0, 64), (10.0, 30)])
def test_search_by_projection_keyframe(th, orb_dist):
    """Relocalization's SearchByProjection(Frame&, KeyFrame*, set, th, ORBdist): no stereo gate (u_right present in the
    view must be ignored), caller's distance bound, every match blocks"""
    w, h = 640, 480
    k0, d0, k1, d1, sf = _two_frames(w, h, 1000)
    rng = np.random.default_rng(17)
    ur = np.where(rng.random(len(k1)) < 0.5, k1["x"] - np.float32(200.0), np.float32(-1)).astype(np.float32)  # gate would reject
    q = _queries_from_last(k0, d0, sf, th, rng=rng, blocks_p=1.0)
    q["min_level"] = k0["octave"] - 1; q["max_level"] = k0["octave"] + 1
    q["valid"] = q["valid"] * (rng.random(len(q)) < 0.8).astype(np.int32)          # sAlreadyFound / isBad / out of frustum
    blocked0 = (rng.random(len(k1)) < 0.3).astype(np.uint8)
    fv = FrameView(k1, d1, 0, w, 0, h, ur)
    of = ol.OracleFrame(k1, d1, sf, 0, w, 0, h, ur)
    for check in (True, False):
        nm, assigned, blocked = ORBmatcher(0.9, check).SearchByProjectionKeyFrame(fv, q, orb_dist, blocked0)
        onm, oassigned, oblocked = of.search_by_projection_keyframe(q, check, orb_dist, blocked0)
        assert nm == onm and onm > 20
        np.testing.assert_array_equal(assigned, oassigned)
        np.testing.assert_array_equal(blocked, oblocked)


@pytest.mark.parametrize("geom,th,ratio", [((1241, 376, 2000), 1.0, 0.8), ((640, 480, 1000), 3.0, 0.8), ((752, 480, 1200), 5.0, 0.6),
                                           ((640, 480, 1000), 3.0, 0.3), ((640, 480, 1000), 5.0, 0.95), ((640, 480, 1000), 3.0, 1.25)])
def test_search_by_projection_points(geom, th, ratio):
    """ratios 0.3 / 0.95 / 1.25: the resolver drops candidates whose distance cannot matter (beyond TH_HIGH and nnratio * d >=
    TH_HIGH); the cut moves with the ratio -- no cut at all at 0.3, at TH_HIGH itself from 1.0 on -- and must never change a result"""
    w, h, nf = geom
    k0, d0, k1, d1, sf = _two_frames(w, h, nf)
    rng = np.random.default_rng(9)
    q = _queries_from_last(k0, d0, sf, 4.0 * th, rng=rng)
    q["min_level"] = k0["octave"] - 1
    q["max_level"] = k0["octave"]
    # duplicate a few queries so that later ones meet blocked candidates
    q = np.concatenate([q, q[:200]])
    fv = FrameView(k1, d1, 0, w, 0, h, None)
    of = ol.OracleFrame(k1, d1, sf, 0, w, 0, h, None)
    nm, assigned, blocked = ORBmatcher(ratio).SearchByProjection(fv, q)
    onm, oassigned, oblocked = of.search_by_projection_points(q, np.float32(ratio))
    assert nm == onm
    np.testing.assert_array_equal(assigned, oassigned)
    np.testing.assert_array_equal(blocked, oblocked)
    assert nm > 0.3 * len(k1)


def test_search_by_bow_grouped():
    k0, d0, k1, d1, sf = _two_frames(640, 480, 1000)
    # stand-in vocabulary: 100 buckets from a hash of the first two descriptor bytes
    bucket = lambda d: (d[:, 0].astype(np.int64) * 256 + d[:, 1]) % 100
    gA, gB = {}, {}
    for i, b in enumerate(bucket(d0)):
        gA.setdefault(int(b), []).append(i)
    for i, b in enumerate(bucket(d1)):
        gB.setdefault(int(b), []).append(i)
    # one big shared bucket with near-duplicates so that matches actually occur
    rng = np.random.default_rng(2)
    dA = d0.copy(); dB = d1.copy()
    for t in range(150):
        dB[t] = dA[t]
        flip = rng.integers(0, 256, 6)
        for f in flip:
            dB[t, f // 8] ^= np.uint8(1 << (f % 8))
    gA = {7: list(range(150)), **{k + 1000: v for k, v in gA.items()}}
    gB = {7: list(range(150)), **{k + 1000: v for k, v in gB.items()}}
    valid = (rng.random(len(dA)) < 0.9).astype(np.uint8)
    # (1) as built above features 0..149 are listed under node 7 AND their hash bucket: nodes depend on each other
    #     (sequential device path); (2) a proper FeatureVector: every feature under exactly one node (parallel path)
    gA2 = {7: list(range(150)), **{k: [i for i in v if i >= 150] for k, v in gA.items() if k != 7}}
    gB2 = {7: list(range(150)), **{k: [i for i in v if i >= 150] for k, v in gB.items() if k != 7}}
    for ga, gb in ((gA, gB), (gA2, gB2)):
        for ratio, check in ((0.7, True), (0.9, False)):
            nm, matchB = ORBmatcher(ratio, check).SearchByBoW(dA, k0["angle"], valid, ga, dB, k1["angle"], gb)
            onm, omatchB = ol.search_by_bow(dA, k0["angle"], valid, ga, dB, k1["angle"], gb, np.float32(ratio), check)
            assert nm == onm
            np.testing.assert_array_equal(matchB, omatchB)
            assert onm > 50
            # SearchByBoW(KF, KF): validity mask on both sides, strict TH_LOW, result per feature of the first keyframe
            from refactored_orb_slam2_amd.matcher import search_by_bow_kf
            validB = (np.random.default_rng(8).random(len(dB)) < 0.85).astype(np.uint8)
            nm2, mA = search_by_bow_kf(dA, k0["angle"], valid, ga, dB, k1["angle"], validB, gb, ratio, check)
            onm2, omA = ol.search_by_bow_kf(dA, k0["angle"], valid, ga, dB, k1["angle"], validB, gb, np.float32(ratio), check)
            assert nm2 == onm2 and onm2 > 40
            np.testing.assert_array_equal(mA, omA)


@pytest.mark.parametrize("geom", [(1241, 376, 2000), (752, 480, 1200)])
def test_stereo_matches_batch(geom):
    import torch
    w, h, nf = geom
    P = 3
    pairs = synth.sequence(w, h, P, seq=8, stereo=True)
    exL, exR = ORBextractor(nf), ORBextractor(nf)
    cap = exL.max_keypoints(w, h)
    dev = "cuda"
    L = torch.from_numpy(np.stack([p[0] for p in pairs])).to(dev)
    R = torch.from_numpy(np.stack([p[1] for p in pairs])).to(dev)
    mk = lambda: (torch.zeros((P, cap, 28), dtype=torch.uint8, device=dev), torch.zeros((P, cap, 32), dtype=torch.uint8, device=dev),
                  torch.zeros(P, dtype=torch.int32, device=dev))
    kl, dl, nl = mk()
    kr, dr, nr = mk()
    ur = torch.zeros((P, cap), dtype=torch.float32, device=dev)
    depth = torch.zeros((P, cap), dtype=torch.float32, device=dev)
    nmatched = torch.zeros(P, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()  # torch's default stream does not order against the handles' own streams
    exL.extract_batch_device(L, kl, dl, nl)
    exR.extract_batch_device(R, kr, dr, nr)
    exL.sync(); exR.sync()
    mbf, fx = 386.1448, 718.856
    mb = mbf / fx
    m = Matcher()
    m.stereo_match(exL, exR, kl, dl, nl, kr, dr, nr, mbf, mb, ur, depth, nmatched)
    m.sync()
    sf, isf = exL.GetScaleFactors(), exL.GetInverseScaleFactors()
    from refactored_orb_slam2_amd._lib import KP_DTYPE
    for p in range(P):
        n_l, n_r = int(nl[p]), int(nr[p])
        kL = kl[p].cpu().numpy().view(KP_DTYPE).reshape(-1)[:n_l]
        kR = kr[p].cpu().numpy().view(KP_DTYPE).reshape(-1)[:n_r]
        dL = dl[p].cpu().numpy()[:n_l]
        dR = dr[p].cpu().numpy()[:n_r]
        oL, oR = ol.OracleExtractor(nf), ol.OracleExtractor(nf)
        okL, odL = oL(pairs[p][0])
        okR, odR = oR(pairs[p][1])
        np.testing.assert_array_equal(dL, odL)
        np.testing.assert_array_equal(dR, odR)
        planesL = [oL.level_pixels(l) for l in range(8)]
        planesR = [oR.level_pixels(l) for l in range(8)]
        on, our, odepth = ol.compute_stereo_matches(okL, odL, okR, odR, planesL, planesR, sf, isf, mbf, mb)
        gur = ur[p].cpu().numpy()[:n_l]
        gdepth = depth[p].cpu().numpy()[:n_l]
        np.testing.assert_array_equal(gur, our)
        np.testing.assert_array_equal(gdepth, odepth)
        assert int(nmatched[p]) == on
        assert on > 0.2 * n_l


def test_hip_reproduces_golden_matcher():
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "matcher_tum.npz"))
    fv = FrameView(g["k1"], g["d1"], 0, int(g["w"]), 0, int(g["h"]), g["u_right"])
    q = g["queries"]
    nm, a, b = ORBmatcher(0.9, True).SearchByProjectionFrame(fv, q)
    assert nm == int(g["frame_nm"])
    np.testing.assert_array_equal(a, g["frame_assigned"]); np.testing.assert_array_equal(b, g["frame_blocked"])
    q2 = q.copy(); q2["max_level"] = g["k0"]["octave"]
    nm, a, b = ORBmatcher(0.8).SearchByProjection(fv, q2)
    assert nm == int(g["points_nm"])
    np.testing.assert_array_equal(a, g["points_assigned"]); np.testing.assert_array_equal(b, g["points_blocked"])
    # window enumeration order: without the stereo gate the candidate list == GetFeaturesInArea
    fv2 = FrameView(g["k1"], g["d1"], 0, int(g["w"]), 0, int(g["h"]), None)
    cand, n = ORBmatcher().ProjCandidates(fv2, q[:64], max_cand=512)
    pos = 0
    for i in range(64):
        exp = g["win_idx"][pos:pos + int(g["win_n"][i])] if q["valid"][i] else np.zeros(0, np.int32)
        pos += int(g["win_n"][i])
        np.testing.assert_array_equal(cand[i, : n[i]]["idx"], exp)


@pytest.mark.parametrize("th", [3.0, 7.5])
def test_proj_best_fuse_and_sim3(th):
    """candidate loop of Fuse (chi-square gate) and of Fuse(Sim3) / SearchBySim3 (no gate): independent first minimum"""
    w, h = 640, 480
    k0, d0, k1, d1, sf = _two_frames(w, h, 1000)
    rng = np.random.default_rng(23)
    ur = np.where(rng.random(len(k1)) < 0.5, k1["x"] - np.float32(20.0) + rng.normal(0, 1, len(k1)).astype(np.float32), np.float32(-1)).astype(np.float32)
    q = _queries_from_last(k0, d0, sf, th, rng=rng)
    q["max_level"] = k0["octave"]                 # nPredictedLevel-1 .. nPredictedLevel
    q["u_r"] = q["u"] - np.float32(20.0)
    inv_sigma2 = (np.float32(1) / (sf * sf)).astype(np.float32)
    fv = FrameView(k1, d1, 0, w, 0, h, ur)
    of = ol.OracleFrame(k1, d1, sf, 0, w, 0, h, ur)
    m = ORBmatcher()
    for inv in (inv_sigma2, None):
        bi, bd = m.ProjBest(fv, q, inv)
        obi, obd = of.proj_best(q, inv)
        np.testing.assert_array_equal(bi, obi); np.testing.assert_array_equal(bd, obd)
        assert (bi >= 0).sum() > 0.4 * len(q)
    # the gate matters: without it more queries find a keypoint (or a closer one)
    g_bi, g_bd = m.ProjBest(fv, q, inv_sigma2)
    n_bi, n_bd = m.ProjBest(fv, q, None)
    assert ((n_bd < g_bd) | ((g_bi < 0) & (n_bi >= 0))).any()
    # mono keyframe (no mvuRight): the 5.99 branch everywhere
    fvm = FrameView(k1, d1, 0, w, 0, h, None); ofm = ol.OracleFrame(k1, d1, sf, 0, w, 0, h, None)
    bi, bd = m.ProjBest(fvm, q, inv_sigma2); obi, obd = ofm.proj_best(q, inv_sigma2)
    np.testing.assert_array_equal(bi, obi); np.testing.assert_array_equal(bd, obd)


@pytest.mark.parametrize("only_stereo,mono", [(False, False), (True, False), (False, True)])
def test_search_for_triangulation(only_stereo, mono):
    """SearchForTriangulation: BoW node groups, epipole / epipolar-line gates, last-minimum tie rule, rotation histogram"""
    from refactored_orb_slam2_amd.matcher import search_for_triangulation
    w, h = 640, 480
    k0, d0, k1, d1, sf = _two_frames(w, h, 1000)
    rng = np.random.default_rng(41)
    # second view = first view shifted by the synthetic image motion (2 px in x): F12 of a pure x-translation between two
    # identical cameras is [t]_x up to scale, i.e. the epipolar line of (x1, y1) is the row y = y1
    ep = np.zeros(1, ol.EPIPOLAR_DTYPE)
    ep["F12"][0] = np.array([0, 0, 0, 0, 0, -1, 0, 1, 0], np.float32) * np.float32(0.37)
    ep["ex"] = 9000.0 if not mono else 300.0; ep["ey"] = 240.0      # mono: some keypoints fall inside the epipole exclusion disc
    ep["scale_factors"][0, :len(sf)] = sf
    ep["level_sigma2"][0, :len(sf)] = (sf * sf).astype(np.float32)
    urA = None if mono else np.where(rng.random(len(k0)) < 0.5, k0["x"] - np.float32(20), -1).astype(np.float32)
    urB = None if mono else np.where(rng.random(len(k1)) < 0.5, k1["x"] - np.float32(20), -1).astype(np.float32)
    hasA = (rng.random(len(k0)) < 0.3).astype(np.uint8); hasB = (rng.random(len(k1)) < 0.3).astype(np.uint8)
    grp = lambda desc: {k: [i for i, d in enumerate(desc) if int(d[1]) % 24 == k] for k in set(int(d[1]) % 24 for d in desc)}
    ga, gb = grp(d0), grp(d1)
    dB = d1.copy()
    dB[gb[3][1]] = dB[gb[3][0]]          # two identical candidates in one node: the later one must win a tie
    for check in (True, False):
        nm, mA = search_for_triangulation(k0, d0, urA, hasA, ga, k1, dB, urB, hasB, gb, ep, only_stereo, check)
        onm, omA = ol.search_for_triangulation(k0, d0, urA, hasA, ga, k1, dB, urB, hasB, gb, ep, only_stereo, check)
        assert nm == onm and onm > (15 if only_stereo else 60)
        np.testing.assert_array_equal(mA, omA)
        assert not hasA[mA >= 0].any() and not hasB[mA[mA >= 0]].any()
    # a pKF1 feature listed under two nodes: the reference visits it twice (sequential replay on the device)
    ga2 = {k: list(v) for k, v in ga.items()}
    ga2[5] = ga2[5] + ga2[4][:3]
    nm, mA = search_for_triangulation(k0, d0, urA, hasA, ga2, k1, dB, urB, hasB, gb, ep, only_stereo, True)
    onm, omA = ol.search_for_triangulation(k0, d0, urA, hasA, ga2, k1, dB, urB, hasB, gb, ep, only_stereo, True)
    assert nm == onm and np.array_equal(mA, omA)


def test_hip_reproduces_golden_tracking():
    import os
    from refactored_orb_slam2_amd.matcher import search_by_bow_kf
    G = os.path.join(os.path.dirname(__file__), "golden")
    g = np.load(os.path.join(G, "tracking_tum.npz"))
    g0 = np.load(os.path.join(G, "extract_tum_640x480_1000_f0.npz")); g1 = np.load(os.path.join(G, "extract_tum_640x480_1000_f1.npz"))
    k0, d0, k1, d1 = g0["keypoints"], g0["descriptors"], g1["keypoints"], g1["descriptors"]
    fv = FrameView(k1, d1, 0, 640, 0, 480, g["u_right"])
    ntm, nm, track, assigned, blocked = ORBmatcher(0.8).SearchLocalPoints(fv, g["frustum"], g["map_points"], 3.0, g["blocked0"])
    assert (ntm, nm) == (int(g["lp_n_to_match"]), int(g["lp_nm"])) and track.tobytes() == g["lp_track"].tobytes()
    np.testing.assert_array_equal(assigned, g["lp_assigned"]); np.testing.assert_array_equal(blocked, g["lp_blocked"])
    k_nm, k_assigned, k_blocked = ORBmatcher(0.9, True).SearchByProjectionKeyFrame(fv, g["track_queries"], 64, g["blocked0"])
    assert k_nm == int(g["kf_nm"]); np.testing.assert_array_equal(k_assigned, g["kf_assigned"]); np.testing.assert_array_equal(k_blocked, g["kf_blocked"])
    grp = lambda desc: {k: [i for i, d in enumerate(desc) if int(d[0]) % 64 == k] for k in set(int(d[0]) % 64 for d in desc)}
    b_nm, b_matchA = search_by_bow_kf(d0, k0["angle"], g["validA"], grp(d0), d1, k1["angle"], g["validB"], grp(d1), 0.9, True)
    assert b_nm == int(g["bow_nm"]); np.testing.assert_array_equal(b_matchA, g["bow_matchA"])
    t_nm, t_assigned, _ = ORBmatcher(0.9, True).SearchByProjectionFrame(FrameView(k1, d1, 0, 640, 0, 480, None), g["track_queries"])
    assert t_nm == int(g["track_nm"]); np.testing.assert_array_equal(t_assigned, g["track_assigned"])


def test_proj_overflowing_candidate_lists_are_reenumerated():
    """windows holding more than the stored 64 candidates: the resolver re-enumerates them (same result)"""
    k0, d0, k1, d1, sf = _two_frames(640, 480, 1000)
    rng = np.random.default_rng(21)
    q = _queries_from_last(k0, d0, sf, 120.0, rng=rng, blocks_p=0.5)   # huge windows
    q["min_level"] = 0; q["max_level"] = -1
    fv = FrameView(k1, d1, 0, 640, 0, 480, None)
    of = ol.OracleFrame(k1, d1, sf, 0, 640, 0, 480, None)
    _, n = ORBmatcher().ProjCandidates(fv, q[:50], max_cand=8)
    assert n.max() > 64
    for mode in (0, 1):
        if mode:
            got = ORBmatcher(0.9, True).SearchByProjectionFrame(fv, q)
            exp = of.search_by_projection_frame(q, True)
        else:
            got = ORBmatcher(0.7).SearchByProjection(fv, q)
            exp = of.search_by_projection_points(q, np.float32(0.7))
        assert got[0] == exp[0]
        np.testing.assert_array_equal(got[1], exp[1]); np.testing.assert_array_equal(got[2], exp[2])


def test_empty_inputs():
    fv = FrameView(np.zeros(0, ol.KP_DTYPE), np.zeros((0, 32), np.uint8), 0, 640, 0, 480)
    q = make_queries(5)
    nm, a, b = ORBmatcher().SearchByProjectionFrame(fv, q)
    assert nm == 0 and len(a) == 0
    k0, d0, k1, d1, sf = _two_frames(640, 480, 1000)
    fv = FrameView(k1, d1, 0, 640, 0, 480)
    nm, a, b = ORBmatcher().SearchByProjection(fv, make_queries(0))
    assert nm == 0 and np.all(a == -1)
    q = _queries_from_last(k0, d0, sf, 7.0)
    q["valid"] = 0
    nm, a, b = ORBmatcher().SearchByProjectionFrame(fv, q)
    assert nm == 0 and np.all(a == -1)
    q["valid"] = 1; q["u"] = 5000.0   # every projection outside the grid
    nm, a, b = ORBmatcher().SearchByProjectionFrame(fv, q)
    assert nm == 0


def test_bench_pipeline_matches_oracle():
    """The exact device-resident sequence bench.py times (2x extract -> stereo -> queries -> SearchByProjection
    (cur,last)) on a short KITTI-geometry sequence, every frame checked against the oracle."""
    import torch
    import bench
    from refactored_orb_slam2_amd._lib import KP_DTYPE, QUERY_DTYPE
    W, H, NF, F = bench.W, bench.H, bench.NFEAT, 4
    pairs = synth.sequence(W, H, F, seq=0, stereo=True)
    dev = "cuda"
    dL = torch.from_numpy(np.stack([p[0] for p in pairs])).to(dev)
    dR = torch.from_numpy(np.stack([p[1] for p in pairs])).to(dev)
    exL, exR, mt = ORBextractor(NF), ORBextractor(NF), Matcher()
    cap = exL.max_keypoints(W, H)
    mk = lambda: (torch.zeros((F, cap, 28), dtype=torch.uint8, device=dev), torch.zeros((F, cap, 32), dtype=torch.uint8, device=dev),
                  torch.zeros(F, dtype=torch.int32, device=dev))
    kl, dl, nl = mk(); kr, dr, nr = mk()
    ur = torch.zeros((F, cap), dtype=torch.float32, device=dev); depth = torch.zeros_like(ur)
    n_st = torch.zeros(F, dtype=torch.int32, device=dev)
    blocked = torch.zeros((F, cap), dtype=torch.uint8, device=dev)
    assigned = torch.full((F, cap), -1, dtype=torch.int32, device=dev)
    n_tr = torch.zeros(F, dtype=torch.int32, device=dev)
    sf = exL.GetScaleFactors(); isf = exL.GetInverseScaleFactors()
    mb = bench.MBF / bench.FX
    from refactored_orb_slam2_amd.matcher import track_queries_batch, unproject_stereo_batch
    cams_np, poses_np = bench.camera_records(F, sf)
    t_cams = torch.from_numpy(cams_np.view(np.uint8).reshape(F, -1)).to(dev)
    t_poses = torch.from_numpy(poses_np.view(np.uint8).reshape(F, -1)).to(dev)
    pts = torch.zeros((F, cap, 60), dtype=torch.uint8, device=dev)
    q = torch.zeros((F, cap, 68), dtype=torch.uint8, device=dev)
    nq = torch.zeros(F, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    exL.extract_batch_device(dL, kl, dl, nl); exR.extract_batch_device(dR, kr, dr, nr)
    exL.sync(); exR.sync()
    cur = torch.cuda.Stream()
    with torch.cuda.stream(cur):
        mt.stereo_match(exL, exR, kl, dl, nl, kr, dr, nr, bench.MBF, mb, ur, depth, n_st, stream=cur)
        unproject_stereo_batch(kl, dl, nl, depth, t_cams, 1, pts, cur)
        track_queries_batch(t_poses, pts, nl, 1, q, nq, cur)
        mt.proj_match_batch(kl, dl, nl, ur, (0.0, float(W), 0.0, float(H)), q, nq, 1, 0.9, True, blocked, assigned, n_tr, stream=cur)
    torch.cuda.synchronize()
    qh = q.cpu().numpy().reshape(F, cap * 68).view(QUERY_DTYPE).reshape(F, cap)
    for f in range(F):
        n = int(nl[f])
        k = kl[f].cpu().numpy().view(KP_DTYPE).reshape(-1)[:n]
        d = dl[f].cpu().numpy()[:n]
        u = ur[f].cpu().numpy()[:n]
        of = ol.OracleFrame(k, d, sf, 0, W, 0, H, u)
        # the queries of frame f: the stereo points of frame f-1 (oracle UnprojectStereo) projected with frame f's camera
        fp = (f - 1) % F
        npv = int(nl[fp])
        kp = kl[fp].cpu().numpy().view(KP_DTYPE).reshape(-1)[:npv]
        opts = ol.unproject_stereo(cams_np[fp:fp + 1], kp, dl[fp].cpu().numpy()[:npv], depth[fp].cpu().numpy()[:npv])
        oq = ol.track_queries(poses_np[f:f + 1], opts)
        assert int(nq[f]) == npv and qh[f, :npv].tobytes() == oq.tobytes()
        # the projection reproduces the known image motion of the synthetic sequence
        ok = oq["valid"] == 1
        assert np.abs(oq["u"][ok] - (kp["x"][ok] + np.float32(bench.SHIFT_X))).max() < 2e-3 and np.array_equal(oq["v"][ok] > 0, ok[ok])
        onm, oa, ob = of.search_by_projection_frame(oq, True)
        assert int(n_tr[f]) == onm, f
        np.testing.assert_array_equal(assigned[f].cpu().numpy()[:n], oa)
        np.testing.assert_array_equal(blocked[f].cpu().numpy()[:n], ob)
        assert onm > 500


@pytest.mark.parametrize("window,ratio", [(100, 0.9), (30, 0.9), (100, 0.6)])
def test_search_for_initialization(window, ratio):
    """mono initialisation matcher incl. match stealing (vMatchedDistance) and huge windows (list overflow)"""
    ex = ORBextractor(2000)
    a, b = synth.sequence(640, 480, 2, seq=15)
    (k0, d0), (k1, d1) = ex.extract_batch([a, b])
    sf = ex.GetScaleFactors()
    ex.close()
    prev = np.stack([k0["x"], k0["y"]], axis=1).astype(np.float32)  # vbPrevMatched starts at F1's own keypoints
    f1 = FrameView(k0, d0, 0, 640, 0, 480)
    f2 = FrameView(k1, d1, 0, 640, 0, 480)
    of2 = ol.OracleFrame(k1, d1, sf, 0, 640, 0, 480)
    for check in (True, False):
        nm, m12, p2 = ORBmatcher(ratio, check).SearchForInitialization(f1, f2, prev, window)
        onm, om12, op2 = ol.search_for_initialization(k0, d0, of2, prev, window, np.float32(ratio), check)
        assert nm == onm
        np.testing.assert_array_equal(m12, om12)
        np.testing.assert_array_equal(p2, op2)
        assert nm > 100 and np.all(m12[k0["octave"] > 0] == -1)


@pytest.mark.parametrize("k,L,ragged,weighting,scoring", [(10, 3, False, 0, 0), (10, 4, False, 0, 0), (7, 5, True, 0, 0),
                                                           (10, 3, False, 1, 1), (10, 3, False, 2, 5), (10, 3, False, 3, 0)])
def test_compute_bow_transform(tmp_path, k, L, ragged, weighting, scoring):
    """Frame::ComputeBoW: tree descent on the device, BowVector / FeatureVector equal to the DBoW2 restatement."""
    from refactored_orb_slam2_amd.vocabulary import ORBVocabulary
    parent, leaf, vdesc, weight = ol.synthetic_vocabulary(k, L, seed=k * 10 + L, ragged=ragged)
    path = str(tmp_path / "voc.txt")
    ol.write_vocabulary_text(path, k, L, parent, leaf, vdesc, weight, scoring, weighting)
    voc = ORBVocabulary()
    assert voc.loadFromTextFile(path)
    ov = ol.OracleVocabulary.load_text(path)
    assert voc.info()[2:] == ov.info() == (len(parent), int(leaf.sum()))
    k0, d0, k1, d1, sf = _two_frames(640, 480, 1000)
    # descriptors near vocabulary nodes + exact ties between sibling nodes (first child must win)
    d = d0.copy()
    d[:50] = vdesc[np.random.default_rng(1).integers(1, len(parent), 50)]
    for levelsup in (4, 2, 0, L + 3):
        bow, fv, (w, nd, wt) = voc.transform(d, levelsup)
        obow, ofv, (ow, ond, owt) = ov.transform(d, levelsup)
        np.testing.assert_array_equal(w, ow); np.testing.assert_array_equal(nd, ond); np.testing.assert_array_equal(wt, owt)
        assert bow == obow and fv == ofv  # doubles compared exactly
        assert sum(len(v) for v in fv.values()) == int((owt > 0).sum())
    if scoring == 0 and weighting == 0:
        assert abs(sum(bow.values()) - 1.0) < 1e-12  # L1-normalised
    # the FeatureVectors feed SearchByBoW end to end
    bow1, fv1, _ = voc.transform(d1, 4 if L > 4 else L - 1)
    bow0, fv0, _ = voc.transform(d0, 4 if L > 4 else L - 1)
    valid = np.ones(len(d0), np.uint8)
    nm, mB = ORBmatcher(0.9, True).SearchByBoW(d0, k0["angle"], valid, fv0, d1, k1["angle"], fv1)
    onm, omB = ol.search_by_bow(d0, k0["angle"], valid, fv0, d1, k1["angle"], fv1, np.float32(0.9), True)
    assert nm == onm and np.array_equal(mB, omB)
    assert not ORBVocabulary().loadFromTextFile(str(tmp_path / "missing.txt"))
    voc.close()
    # ---- loadFromBinaryFile: float weights and the duplicated last node of the reference's eof loop
    bpath = str(tmp_path / "voc.bin")
    ol.write_vocabulary_binary(bpath, k, L, parent, leaf, vdesc, weight, scoring, weighting)
    vb = ORBVocabulary()
    assert vb.loadFromBinaryFile(bpath)
    ob = ol.OracleVocabulary.load_binary(bpath)
    assert vb.info()[2:] == ob.info() == (len(parent) + 1, int(leaf.sum()) + int(leaf[-1]))
    bow, fv, (w, nd, wt) = vb.transform(d, 4)
    obow, ofv, (ow, ond, owt) = ob.transform(d, 4)
    np.testing.assert_array_equal(w, ow); np.testing.assert_array_equal(nd, ond); np.testing.assert_array_equal(wt, owt)
    assert bow == obow and fv == ofv
    assert not ORBVocabulary().loadFromBinaryFile(str(tmp_path / "missing.bin"))
    vb.close()


def _write_png_gray(path, img):
    """minimal 8-bit greyscale PNG writer for the driver test (zlib only): filter types cycle through 0..4 so that the
    library's reader is exercised on all of them"""
    import struct, zlib
    h, w = img.shape
    rows = bytearray()
    prev = np.zeros(w, np.int16)
    for y in range(h):
        cur = img[y].astype(np.int16)
        ft = y % 5
        left = np.concatenate([[0], cur[:-1]]); ul = np.concatenate([[0], prev[:-1]])
        if ft == 0: f = cur
        elif ft == 1: f = cur - left
        elif ft == 2: f = cur - prev
        elif ft == 3: f = cur - ((left + prev) >> 1)
        else:
            p = left + prev - ul
            pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
            f = cur - pred
        rows += bytes([ft]) + (f & 0xff).astype(np.uint8).tobytes()
        prev = cur
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    with open(path, "wb") as fh:
        fh.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) +
                 chunk(b"IDAT", zlib.compress(bytes(rows), 6)) + chunk(b"IEND", b""))


def test_sequence_driver_on_kitti_layout(tmp_path):
    """examples/stereo_kitti.py on a synthetic sequence written in the KITTI directory layout (PNG + times.txt), read with the
    library's zlib PNG reader: every frame's keypoints, descriptors, mvuRight / mvDepth and the SearchByProjection(cur, last)
    assignment against the previous frame == the oracle; then the sharded mode (two ranks, contiguous chunks, one gather per
    batch) delivers the same per-frame records."""
    import subprocess, sys, os
    from refactored_orb_slam2_amd._lib import TRACK_POSE_DTYPE, UNPROJECT_CAM_DTYPE
    seq = tmp_path / "00"
    (seq / "image_0").mkdir(parents=True); (seq / "image_1").mkdir()
    W, H, NF, N = 1241, 376, 2000, 5
    pairs = synth.sequence(W, H, N, seq=20, stereo=True)
    with open(seq / "times.txt", "w") as f:
        for i, (L, R) in enumerate(pairs):
            _write_png_gray(seq / "image_0" / f"{i:06d}.png", L)
            _write_png_gray(seq / "image_1" / f"{i:06d}.png", R)
            f.write(f"{i * 0.1:e}\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    drv = os.path.join(root, "examples", "stereo_kitti.py")
    # ---- oracle, frame by frame
    oL, oR = ol.OracleExtractor(NF), ol.OracleExtractor(NF)
    sf, isf = oL.scale_factors, oL.inv_scale_factors
    bf, fx, fy, cx, cy = 386.1448, 718.856, 718.856, 607.1928, 185.2157
    cam = np.zeros(1, UNPROJECT_CAM_DTYPE); pose = np.zeros(1, TRACK_POSE_DTYPE)
    eye = np.eye(3, dtype=np.float32).reshape(9)
    cam["Rwc"] = eye; cam["cx"] = cx; cam["cy"] = cy; cam["invfx"] = np.float32(1) / np.float32(fx); cam["invfy"] = np.float32(1) / np.float32(fy)
    pose["Rcw"] = eye; pose["fx"] = fx; pose["fy"] = fy; pose["cx"] = cx; pose["cy"] = cy; pose["mbf"] = bf
    pose["max_x"] = W; pose["max_y"] = H; pose["th"] = 7.0; pose["scale_factors"][0, :8] = sf
    exp = []
    prev = None
    for (L, R) in pairs:
        kL, dL = oL(L); kR, dR = oR(R)
        _, ur, depth = ol.compute_stereo_matches(kL, dL, kR, dR, [oL.level_pixels(l) for l in range(8)], [oR.level_pixels(l) for l in range(8)],
                                                 sf, isf, bf, bf / fx)
        nm, assigned = 0, np.full(len(kL), -1, np.int32)
        if prev is not None:
            nm, assigned, _ = ol.OracleFrame(kL, dL, sf, 0, W, 0, H, ur).search_by_projection_frame(ol.track_queries(pose, prev), True)
        exp.append((kL, dL, ur, depth, nm, assigned))
        prev = ol.unproject_stereo(cam, kL, dL, depth)
    for extra in ([], ["--batch", "4"], ["--batch", "2"]):
        dump = str(tmp_path / "dump.npz")
        r = subprocess.run([sys.executable, drv, str(seq), "--dump", dump] + extra, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        assert "median tracking time" in r.stdout and "Images in the sequence: 5" in r.stdout
        g = np.load(dump)
        for i, (kL, dL, ur, depth, nm, assigned) in enumerate(exp):
            np.testing.assert_array_equal(g[f"kp_{i}"], kL, err_msg=f"keypoints of frame {i} ({extra})")
            np.testing.assert_array_equal(g[f"desc_{i}"], dL)
            np.testing.assert_array_equal(g[f"ur_{i}"], ur); np.testing.assert_array_equal(g[f"depth_{i}"], depth)
            assert int(g[f"ntrack_{i}"]) == nm, (i, extra)
            np.testing.assert_array_equal(g[f"assigned_{i}"], assigned)
        assert exp[2][4] > 500
    # ---- config C5 in small: two ranks, contiguous frame shards, gather of the padded per-frame records
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dump = str(tmp_path / "shard")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), drv, str(seq), "--shard", "--batch", "2", "--dump", dump], capture_output=True, text=True,
                       env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert "sharded over 2 ranks, 3 frames per rank" in r.stdout
    for rank in (0, 1):
        g = np.load(f"{dump}.rank{rank}.npz")
        for i, (kL, dL, *_rest) in enumerate(exp):   # every rank holds every frame's gathered record
            np.testing.assert_array_equal(g[f"g_kp_{i}"], kL, err_msg=f"gathered keypoints of frame {i} on rank {rank}")
            np.testing.assert_array_equal(g[f"g_desc_{i}"], dL)


# ------------------------------------------------------------------------------------------------ whole-function A15
def _kf_scene(seed, nlevels=8, w=640, h=480, nfeat=1000):
    """A keyframe (extracted synthetic frame with a partial mvuRight), a camera looking at it and candidate map points that
    project onto its keypoints (plus points that fail each gate: behind the camera, outside the image, out of the distance
    range, oblique normals)."""
    from refactored_orb_slam2_amd import _lib
    from refactored_orb_slam2_amd.matcher import make_frustum
    ex = ORBextractor(nfeat, 1.2, nlevels, 20, 7)
    k, d = ex(synth.frame(w, h, seq=seed, f=0))
    sf = ex.GetScaleFactors(); inv_s2 = ex.GetInverseScaleSigmaSquares()
    ex.close()
    rng = np.random.default_rng(seed)
    R, t = synth.camera_pose(seed)
    fr = make_frustum(R, t, 517.3, 516.5, 318.6, 255.3, 40.0, (0, w, 0, h), 1.2, nlevels)
    mp = synth.local_map(k, d, fr, seed + 1, n_extra=300)
    cam = np.zeros(1, _lib.KF_CAMERA_DTYPE)
    for f in ("fx", "fy", "cx", "cy", "mbf", "min_x", "max_x", "min_y", "max_y", "log_scale_factor", "n_levels"):
        cam[f] = fr[f]
    cam["R"] = fr["Rcw"]; cam["t"] = fr["tcw"]; cam["Ow"] = fr["Ow"]; cam["scale_factors"] = fr["scale_factors"]
    pts = np.zeros(len(mp), _lib.KF_POINT_DTYPE)
    for f in ("pos", "normal", "min_distance", "max_distance", "skip", "desc"):
        pts[f] = mp[f]
    pts["angle"] = rng.uniform(0, 360, len(pts)).astype(np.float32)
    ur = np.where(rng.random(len(k)) < 0.5, k["x"] - np.float32(20) + rng.normal(0, 1.5, len(k)).astype(np.float32), -1).astype(np.float32)
    return k, d, sf, inv_s2, ur, cam, pts, (w, h)


def _check_kf(res, ores, fields=("best_idx", "best_dist", "level", "u", "v", "u_r")):
    for f in fields:
        np.testing.assert_array_equal(res[f], ores[f], err_msg=f)


@pytest.mark.parametrize("nlevels", [8, 12])
def test_whole_function_fuse_sim3_loop_reloc(nlevels):
    """ORBmatcher::Fuse, Fuse(Sim3), SearchBySim3 (one direction), SearchByProjection(KF,Scw) and SearchByProjection(Frame,KF,...)
    from the projection on: prologue kernel + window search on the device == the oracle's restatement of the whole loops
    (L/src/ORBmatcher.cc:766-1245, 275-386, 1385-1504).  12 levels: no octave is aliased onto the first eight."""
    from refactored_orb_slam2_amd import _lib
    k, d, sf, inv_s2, ur, cam, pts, (w, h) = _kf_scene(41 + nlevels, nlevels)
    kf = FrameView(k, d, 0, w, 0, h, ur); okf = ol.OracleFrame(k, d, sf, 0, w, 0, h, ur)
    kf_mono = FrameView(k, d, 0, w, 0, h); okf_mono = ol.OracleFrame(k, d, sf, 0, w, 0, h)
    m = ORBmatcher(0.8, True)
    # Fuse: th = 3 (LocalMapping) with the chi-square gate, stereo and monocular keyframe
    cam["th"] = 3.0
    for view, oview in ((kf, okf), (kf_mono, okf_mono)):
        n, res, _ = m.KeyFrameSearch(view, cam, pts, _lib.KF_FUSE, inv_level_sigma2=inv_s2)
        on, ores, _ = ol.kf_search(oview, cam, pts, 1, inv_level_sigma2=inv_s2)
        _check_kf(res, ores); assert n == on and n > 300
    # Fuse(Sim3): th = 4 (LoopClosing::SearchAndFuse)
    cam["th"] = 4.0
    n, res, _ = m.KeyFrameSearch(kf, cam, pts, _lib.KF_FUSE_SIM3)
    on, ores, _ = ol.kf_search(okf, cam, pts, 2)
    _check_kf(res, ores); assert n == on and n > 300
    # SearchBySim3 direction: second transform = a small similarity, th = 7.5
    cam3 = cam.copy(); cam3["th"] = 7.5
    s12 = np.float32(1.03)
    a = 0.01
    R12 = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], np.float32)
    cam3["R2"] = ((np.float32(1.0 / float(s12))) * R12.T).astype(np.float32).reshape(9)
    cam3["t2"] = np.array([0.02, -0.01, 0.03], np.float32)
    n, res, _ = m.KeyFrameSearch(kf, cam3, pts, _lib.KF_SIM3)
    on, ores, _ = ol.kf_search(okf, cam3, pts, 3)
    _check_kf(res, ores); assert n == on and n > 200
    # SearchByProjection(KF, Scw, ...): th = 10, some keypoints already matched
    cam["th"] = 10.0
    rng = np.random.default_rng(5)
    matched0 = (rng.random(len(k)) < 0.15).astype(np.uint8)
    n, res, blk = m.KeyFrameSearch(kf, cam, pts, _lib.KF_LOOP, blocked=matched0, max_dist=50)
    on, ores, oblk = ol.kf_search(okf, cam, pts, 4, matched=matched0, th_low=50)
    _check_kf(res, ores, ("best_idx", "level", "u", "v")); assert n == on and n > 200
    np.testing.assert_array_equal(blk, oblk)
    # SearchByProjection(Frame, KF, sAlreadyFound, th, ORBdist): relocalisation, th = 10, ORBdist = 100, rotation histogram
    for check in (True, False):
        mm = ORBmatcher(0.9, check)
        n, res, blk = mm.KeyFrameSearch(kf_mono, cam, pts, _lib.KF_RELOC, blocked=matched0, max_dist=100)
        on, ores, _ = ol.kf_search(okf_mono, cam, pts, 5, matched=matched0, th_low=100, check_orientation=check)
        _check_kf(res, ores, ("best_idx",)); assert n == on and n > 100


def test_twelve_level_motion_model_and_triangulation():
    """SearchByProjection(cur, last) through the device query preparation and SearchForTriangulation with a 12-level pyramid:
    mvScaleFactors[octave] / mvLevelSigma2[octave] are indexed unmasked (L/src/ORBmatcher.cc:1297, 700-745)."""
    import torch
    from refactored_orb_slam2_amd import _lib
    from refactored_orb_slam2_amd.matcher import search_for_triangulation, track_queries_batch, unproject_stereo_batch
    w, h, nl = 752, 480, 12
    ex = ORBextractor(1500, 1.2, nl, 20, 7)
    a, b = synth.sequence(w, h, 2, seq=33)
    (k0, d0), (k1, d1) = ex.extract_batch([a, b])
    sf = ex.GetScaleFactors(); s2 = ex.GetScaleSigmaSquares()
    ex.close()
    assert k0["octave"].max() >= 9
    rng = np.random.default_rng(3)
    # motion model: oracle UnprojectStereo + track queries vs the device kernels
    cam = np.zeros(1, _lib.UNPROJECT_CAM_DTYPE); pose = np.zeros(1, _lib.TRACK_POSE_DTYPE)
    cam["Rwc"][0] = np.eye(3, dtype=np.float32).reshape(9); cam["cx"] = 367.4; cam["cy"] = 252.2
    cam["invfx"] = np.float32(1) / np.float32(435.2); cam["invfy"] = np.float32(1) / np.float32(435.2)
    pose["Rcw"][0] = np.eye(3, dtype=np.float32).reshape(9); pose["fx"] = 435.2; pose["fy"] = 435.2; pose["cx"] = 365.4; pose["cy"] = 252.2
    pose["mbf"] = 47.9; pose["max_x"] = w; pose["max_y"] = h; pose["th"] = 7.0; pose["scale_factors"][0, :nl] = sf
    depth = np.where(rng.random(len(k0)) < 0.8, rng.uniform(2, 30, len(k0)), -1).astype(np.float32)
    opts = ol.unproject_stereo(cam, k0, d0, depth)
    oq = ol.track_queries(pose, opts)
    dev = "cuda"
    capn = len(k0)
    tk = torch.from_numpy(k0.view(np.uint8).reshape(1, capn, 28)).to(dev); td = torch.from_numpy(d0.reshape(1, capn, 32)).to(dev)
    tn = torch.tensor([capn], dtype=torch.int32, device=dev); tdep = torch.from_numpy(depth.reshape(1, capn)).to(dev)
    tc = torch.from_numpy(cam.view(np.uint8).reshape(1, -1)).to(dev); tp = torch.from_numpy(pose.view(np.uint8).reshape(1, -1)).to(dev)
    pts = torch.zeros((1, capn, 60), dtype=torch.uint8, device=dev); q = torch.zeros((1, capn, 68), dtype=torch.uint8, device=dev)
    nq = torch.zeros(1, dtype=torch.int32, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        unproject_stereo_batch(tk, td, tn, tdep, tc, 1, pts, s)
        track_queries_batch(tp, pts, tn, 0, q, nq, s)
    torch.cuda.synchronize()
    assert q.cpu().numpy().reshape(-1)[: capn * 68].tobytes() == oq.tobytes()
    nm, assigned, _ = ORBmatcher(0.9, True).SearchByProjectionFrame(FrameView(k1, d1, 0, w, 0, h), oq)
    onm, oassigned, _ = ol.OracleFrame(k1, d1, sf, 0, w, 0, h).search_by_projection_frame(oq, True)
    assert nm == onm and nm > 300
    np.testing.assert_array_equal(assigned, oassigned)
    # triangulation: 40 synthetic vocabulary buckets, epipolar gate with 12-level sigma tables
    ga = {}; gb = {}
    for i, dd in enumerate(d0):
        ga.setdefault(int(dd[0]) % 40, []).append(i)
    for i, dd in enumerate(d1):
        gb.setdefault(int(dd[0]) % 40, []).append(i)
    ep = np.zeros(1, _lib.EPIPOLAR_DTYPE)
    ep["F12"][0] = np.array([0, 0, 0, 0, 0, -1e-2, 0, 1e-2, 0], np.float32)   # horizontal epipolar lines
    ep["ex"] = -1e6; ep["ey"] = 240.0
    ep["scale_factors"][0, :nl] = sf; ep["level_sigma2"][0, :nl] = s2
    has0 = (rng.random(len(k0)) < 0.3).astype(np.uint8); has1 = (rng.random(len(k1)) < 0.3).astype(np.uint8)
    nmt, mA = search_for_triangulation(k0, d0, None, has0, ga, k1, d1, None, has1, gb, ep, False, True)
    onmt, omA = ol.search_for_triangulation(k0, d0, None, has0, ga, k1, d1, None, has1, gb, ep.astype(ol.EPIPOLAR_DTYPE), False, True)
    assert nmt == onmt and nmt > 20
    np.testing.assert_array_equal(mA, omA)


def test_mono_sequence_driver_on_kitti_layout(tmp_path):
    """examples/mono_kitti.py (config C1 in small): image_0 + times.txt, mpIniORBextractor's 2 x nFeatures, the front-end half
    of Tracking::MonocularInitialization (reference frame, SearchForInitialization with window 100, reset rules) == the oracle
    frame by frame, including a frame that drops the reference (a blank image: no keypoints)."""
    import subprocess, sys, os
    seq = tmp_path / "00"
    (seq / "image_0").mkdir(parents=True)
    W, H, NF, N = 1241, 376, 1000, 6
    frames = [f for f in synth.sequence(W, H, N, seq=31)]
    frames[3] = np.full((H, W), 90, np.uint8)          # no corners: <= 100 keypoints -> the initializer is deleted (:527-532)
    with open(seq / "times.txt", "w") as f:
        for i, im in enumerate(frames):
            _write_png_gray(seq / "image_0" / f"{i:06d}.png", im)
            f.write(f"{i * 0.1:e}\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dump = str(tmp_path / "mono.npz")
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "mono_kitti.py"), str(seq), "--features", str(NF), "--dump", dump],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
    assert "median tracking time" in r.stdout and f"Images in the sequence: {N}" in r.stdout
    g = np.load(dump)
    orc = ol.OracleExtractor(2 * NF)
    sf = orc.scale_factors
    ini, states = None, []
    for i, im in enumerate(frames):
        k, d = orc(im)
        np.testing.assert_array_equal(g[f"kp_{i}"], k, err_msg=f"keypoints of frame {i}")
        np.testing.assert_array_equal(g[f"desc_{i}"], d)
        nm, m12, state = 0, np.zeros(0, np.int32), "idle"
        if ini is None:
            if len(k) > 100:
                ini = (k, d, np.stack([k["x"], k["y"]], 1).astype(np.float32)); state = "reference"
        elif len(k) <= 100:
            ini, state = None, "reset"
        else:
            nm, m12, prev = ol.search_for_initialization(ini[0], ini[1], ol.OracleFrame(k, d, sf, 0, W, 0, H), ini[2], 100, np.float32(0.9), True)
            if nm < 100:
                ini, state = None, "reset"
            else:
                ini = (ini[0], ini[1], prev); state = "matched"
        states.append(state)
        assert str(g[f"state_{i}"]) == state, (i, str(g[f"state_{i}"]), state)
        assert int(g[f"nm_{i}"]) == nm, (i, int(g[f"nm_{i}"]), nm)
        np.testing.assert_array_equal(g[f"m12_{i}"], m12, err_msg=f"vnMatches12 of frame {i}")
    assert states[:5] == ["reference", "matched", "matched", "reset", "reference"] and states[5] == "matched", states


@pytest.mark.parametrize("geom", [(1241, 376, 2000), (752, 480, 1200)])
def test_stereo_matches_one_pair_host_api(geom):
    """orbfe_stereo_match: the per-frame call of Frame::Frame (L/src/Frame.cc:91-99) -- two host-API extractions, then the stereo
    association on the pyramids still in HBM, host keypoints in, mvuRight / mvDepth out -- against oo_compute_stereo_matches and
    against the independent Python reading; degenerate inputs (no right keypoints, mb <= 0) behave as documented."""
    from refactored_orb_slam2_amd.matcher import compute_stereo_matches
    from refactored_orb_slam2_amd import _lib
    from tests import np_restatement as nr
    w, h, nf = geom
    L, R = synth.stereo_pair(w, h, seq=14, f=3)
    exL, exR = ORBextractor(nf), ORBextractor(nf)
    kL, dL = exL(L); kR, dR = exR(R)
    mbf, mb = np.float32(386.1448), np.float32(386.1448 / 718.856)
    nm, ur, depth = compute_stereo_matches(exL, exR, kL, dL, kR, dR, mbf, mb)
    oL, oR = ol.OracleExtractor(nf), ol.OracleExtractor(nf)
    okL, odL = oL(L); okR, odR = oR(R)
    pL = [oL.level_pixels(l).copy() for l in range(8)]; pR = [oR.level_pixels(l).copy() for l in range(8)]
    _, our, odepth = ol.compute_stereo_matches(okL, odL, okR, odR, pL, pR, oL.scale_factors, oL.inv_scale_factors, float(mbf), float(mb))
    np.testing.assert_array_equal(ur, our); np.testing.assert_array_equal(depth, odepth)
    assert nm == int((our >= 0).sum()) and nm > len(kL) // 3
    if w < 1000:   # the second reading is pure Python: the smaller geometry only
        rur, rdepth = nr.ref_compute_stereo_matches(okL, odL, okR, odR, pL, pR, oL.scale_factors, oL.inv_scale_factors, mbf, mb)
        np.testing.assert_array_equal(ur, rur); np.testing.assert_array_equal(depth, rdepth)
    nm0, ur0, depth0 = compute_stereo_matches(exL, exR, kL, dL, kR[:0], dR[:0], mbf, mb)
    assert nm0 == 0 and np.all(ur0 == -1) and np.all(depth0 == -1)
    with pytest.raises(_lib.OrbfeError):
        compute_stereo_matches(exL, exR, kL, dL, kR, dR, mbf, 0.0)
    exL.close(); exR.close()


@pytest.mark.gpu
def test_pipeline_handle_equals_the_oracle_across_chunks():
    """refactored_orb_slam2_amd.pipeline.StereoPipeline (ctypes mirror of orbfe_pipeline_*): seven EuRoC-size stereo frames in chunks of
    3 + 3 + 1 through two buffer sets -- every frame's keypoints, descriptors, mvuRight / mvDepth and its SearchByProjection(cur, last)
    assignment against the points of the frame before it (the last frame of the previous chunk for a chunk's first frame) equal the
    oracle's; a chunk with no frames and a re-submitted slot are harmless; per-frame poses written into the slot are honoured."""
    from refactored_orb_slam2_amd._lib import TRACK_POSE_DTYPE, UNPROJECT_CAM_DTYPE
    from refactored_orb_slam2_amd.pipeline import StereoPipeline
    W, H, NF, N, F = 752, 480, 1200, 7, 3
    fx, fy, cx, cy, bf = (np.float32(v) for v in (435.2047, 435.2047, 367.4517, 252.2008, 47.9064))
    pairs = synth.sequence(W, H, N, seq=31, stereo=True)
    oL, oR = ol.OracleExtractor(NF), ol.OracleExtractor(NF)
    sf, isf = oL.scale_factors, oL.inv_scale_factors
    cam = np.zeros(1, UNPROJECT_CAM_DTYPE); pose = np.zeros(1, TRACK_POSE_DTYPE)
    eye = np.eye(3, dtype=np.float32).reshape(9)
    cam["Rwc"] = eye; cam["cx"] = cx; cam["cy"] = cy; cam["invfx"] = np.float32(1) / fx; cam["invfy"] = np.float32(1) / fy
    pose["Rcw"] = eye; pose["fx"] = fx; pose["fy"] = fy; pose["cx"] = cx + np.float32(-2.0); pose["cy"] = cy; pose["mbf"] = bf
    pose["max_x"] = W; pose["max_y"] = H; pose["th"] = 7.0; pose["scale_factors"][0, :8] = sf
    exp, prev = [], None
    for (Li, Ri) in pairs:
        kL, dL = oL(Li); kR, dR = oR(Ri)
        _, ur, depth = ol.compute_stereo_matches(kL, dL, kR, dR, [oL.level_pixels(l) for l in range(8)], [oR.level_pixels(l) for l in range(8)],
                                                 sf, isf, float(bf), float(bf / fx))
        nm, assigned = 0, np.full(len(kL), -1, np.int32)
        if prev is not None:
            nm, assigned, _ = ol.OracleFrame(kL, dL, sf, 0, W, 0, H, ur).search_by_projection_frame(ol.track_queries(pose, prev), True)
        exp.append((kL, dL, ur, depth, nm, assigned))
        prev = ol.unproject_stereo(cam, kL, dL, depth)
    with StereoPipeline(W, H, F, float(fx), float(fy), float(cx), float(cy), float(bf), 7.0, n_features=NF, slots=2) as p:
        for s in range(2):
            p.poses(s)["cx"] = cx + np.float32(-2.0)     # the synthetic sequence moves 2 px per frame: the prediction follows it
        got = []
        chunks = [list(range(0, 3)), list(range(3, 6)), [], [6]]       # an empty chunk in between changes nothing
        for k, idx in enumerate(chunks):
            s = k % 2
            p.wait(s)                                                  # the slot's previous results have been consumed below
            for j, i in enumerate(idx):
                p.left(s)[j, :, :W] = pairs[i][0]; p.right(s)[j, :, :W] = pairs[i][1]
            p.submit(s, len(idx), has_predecessor=bool(got) or k > 0)
            p.wait(s)
            out = p.output(s)
            assert int(out["n_left"][len(idx):].sum()) == 0           # rows behind the chunk carry no keypoints
            for j, i in enumerate(idx):
                n = int(out["n_left"][j])
                got.append((out["kps_left"][j, :n].copy(), out["desc_left"][j, :n].copy(), out["u_right"][j, :n].copy(),
                            out["depth"][j, :n].copy(), int(out["n_tracked"][j]), out["assigned"][j, :n].copy()))
        assert len(got) == N
        for i, (g, e) in enumerate(zip(got, exp)):
            np.testing.assert_array_equal(g[0], e[0], err_msg=f"keypoints of frame {i}")
            np.testing.assert_array_equal(g[1], e[1]); np.testing.assert_array_equal(g[2], e[2], err_msg=f"mvuRight of frame {i}")
            np.testing.assert_array_equal(g[3], e[3])
            assert g[4] == e[4], (i, g[4], e[4])
            np.testing.assert_array_equal(g[5], e[5], err_msg=f"tracked assignment of frame {i}")
        assert exp[4][4] > 300
        # without waiting in between: chunk (3, 4, 5) in slot 1 while chunk (0, 1, 2) still runs in slot 0 -- the extraction of the
        # second beside the matching half of the first, each slot on its own handles -- and then both again with the images left where
        # the uploads put them (orbfe_pipeline_submit_resident), a host frame overwritten to prove nothing is uploaded
        for s, idx in ((0, [0, 1, 2]), (1, [3, 4, 5])):
            for j, i in enumerate(idx):
                p.left(s)[j, :, :W] = pairs[i][0]; p.right(s)[j, :, :W] = pairs[i][1]
        p.submit(0, 3, has_predecessor=False); p.submit(1, 3, has_predecessor=True)
        p.wait(0); p.wait(1)
        first = [{k: np.array(v, copy=True) for k, v in p.output(s).items()} for s in (0, 1)]
        p.left(0)[1, :, :W] = 0; p.right(1)[2, :, :W] = 255
        p.submit_resident(0, 3, has_predecessor=False); p.submit_resident(1, 3, has_predecessor=True)
        p.wait(0); p.wait(1)
        for s, idx in ((0, [0, 1, 2]), (1, [3, 4, 5])):
            out = p.output(s)
            for key in ("n_left", "kps_left", "desc_left", "u_right", "depth", "n_tracked", "assigned"):
                np.testing.assert_array_equal(np.asarray(out[key]), first[s][key], err_msg=f"resident resubmit, slot {s}, {key}")
            for j, i in enumerate(idx):
                n = int(out["n_left"][j])
                np.testing.assert_array_equal(out["kps_left"][j, :n], exp[i][0]); np.testing.assert_array_equal(out["depth"][j, :n], exp[i][3])
                if i > 0:
                    assert int(out["n_tracked"][j]) == exp[i][4]
        dl, dr, pitch, ib = p.device_input(1)
        assert dl and dr and dr > dl and pitch >= W and ib == pitch * H
