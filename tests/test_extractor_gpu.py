"""GPU parity: liborbfe (HIP, through the C ABI) against the CPU oracle, stage by stage and end to end.

Bar: bit-exact pyramid / blur planes, candidate lists, octree selections (indices and order), keypoint
fields incl. angle (tolerance 1e-4 deg stated by north_star; we assert exact equality and report) and
descriptor bytes.
"""
import numpy as np
import pytest

from refactored_orb_slam2_amd import ORBextractor, synth
from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu

GEOMS = {
    "kitti": (1241, 376, 2000),
    "tum": (640, 480, 1000),
    "euroc": (752, 480, 1200),
}


def _compare_stages(img, nfeat, ini=20, mn=7, nlevels=8, sf=1.2):
    ex = ORBextractor(nfeat, sf, nlevels, ini, mn)
    orc = ol.OracleExtractor(nfeat, sf, nlevels, ini, mn)
    kps, desc = ex(img)
    okps, odesc = orc(img)
    for l in range(nlevels):
        assert ex.level_size(l) == orc.level_size(l)
        np.testing.assert_array_equal(ex.debug_pyramid(0, l), orc.level_pixels(l), err_msg=f"pyramid level {l}")
        x, y, s = ex.debug_candidates(0, l)
        ox, oy, os_ = orc.level_candidates(l)
        np.testing.assert_array_equal(x, ox, err_msg=f"candidate x level {l}")
        np.testing.assert_array_equal(y, oy, err_msg=f"candidate y level {l}")
        np.testing.assert_array_equal(s, os_, err_msg=f"candidate score level {l}")
        kx, ky, ks = ex.debug_level_keypoints(0, l)
        ok = orc.level_keypoints(l)
        np.testing.assert_array_equal(kx, ok["x"].astype(np.int32), err_msg=f"octree x level {l}")
        np.testing.assert_array_equal(ky, ok["y"].astype(np.int32), err_msg=f"octree y level {l}")
        np.testing.assert_array_equal(ks, ok["response"].astype(np.int32), err_msg=f"octree score level {l}")
        ob = orc.level_blurred(l)
        if ob is not None:
            np.testing.assert_array_equal(ex.debug_blurred(0, l), ob, err_msg=f"blur level {l}")
    assert len(kps) == len(okps)
    for f in ("x", "y", "size", "response", "octave", "class_id"):
        np.testing.assert_array_equal(kps[f], okps[f], err_msg=f"keypoint field {f}")
    assert np.max(np.abs(kps["angle"] - okps["angle"])) <= 1e-4  # north_star tolerance
    np.testing.assert_array_equal(kps["angle"], okps["angle"])    # and in fact bit-exact
    np.testing.assert_array_equal(desc, odesc)
    ex.close()
    return len(kps)


@pytest.mark.parametrize("name", list(GEOMS))
def test_stage_parity_synthetic(name):
    w, h, nf = GEOMS[name]
    img = synth.frame(w, h, seq=3, f=1)
    n = _compare_stages(img, nf)
    assert n >= nf * 0.9


# ------------------------------------------------------------------------------------------ golden fixtures
import glob
import os

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "extract_*.npz"))))
def test_hip_reproduces_golden_extraction(path):
    g = np.load(path)
    ex = ORBextractor(int(g["nfeatures"]))
    k, d = ex(g["image"])
    np.testing.assert_array_equal(k, g["keypoints"])
    np.testing.assert_array_equal(d, g["descriptors"])
    for l in (0, 7):
        x, y, s = ex.debug_candidates(0, l)
        np.testing.assert_array_equal(np.stack([x, y, s]).astype(np.int16), g[f"cand_{l}"])
    # mvImagePyramid semantics: level 0 is the input itself
    np.testing.assert_array_equal(ex.pyramid_level(0), g["image"])
    assert [p.shape for p in ex.mvImagePyramid] == [(ex.level_size(l)[1], ex.level_size(l)[0]) for l in range(8)]
    ex.close()


# ------------------------------------------------------------------------------------------ batch / device API
def test_batch_equals_single_and_is_idempotent():
    import torch
    w, h, nf, B = 1241, 376, 2000, 6
    imgs = synth.sequence(w, h, B, seq=6)
    ex = ORBextractor(nf)
    single = [ex(i) for i in imgs]
    batch = ex.extract_batch(imgs)
    for (k1, d1), (k2, d2) in zip(single, batch):
        np.testing.assert_array_equal(k1, k2); np.testing.assert_array_equal(d1, d2)
    # the three host paths: one or two images (zero-copy through pinned memory, one FAST cell per wave), up to eight (DMA copies,
    # one cell per wave), more (runs of four cells) -- B = 6 above is the middle one, a pair and a batch of 9 the other two
    pair = ex.extract_batch(imgs[1:3])
    for (k1, d1), (k2, d2) in zip(single[1:3], pair):
        np.testing.assert_array_equal(k1, k2); np.testing.assert_array_equal(d1, d2)
    nine = ex.extract_batch(imgs + imgs[:3])
    for (k1, d1), (k2, d2) in zip(single + single[:3], nine):
        np.testing.assert_array_equal(k1, k2); np.testing.assert_array_equal(d1, d2)
    # device-resident path, non-contiguous source rows (stride > width), twice (idempotence)
    pad = np.zeros((B, h, w + 37), np.uint8)
    pad[:, :, :w] = np.stack(imgs)
    dimg = torch.from_numpy(pad).cuda()[:, :, :w]
    cap = ex.max_keypoints(w, h)
    kps = torch.zeros((B, cap, 28), dtype=torch.uint8, device="cuda")
    desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    n = torch.zeros(B, dtype=torch.int32, device="cuda")
    from refactored_orb_slam2_amd._lib import KP_DTYPE
    for _ in range(2):
        kps.zero_(); desc.zero_()
        torch.cuda.synchronize()  # torch's default stream does not order against the handle's own stream
        ex.extract_batch_device(dimg, kps, desc, n)
        ex.sync(); ex.device_status()
        for i in range(B):
            c = int(n[i])
            np.testing.assert_array_equal(kps[i, :c].cpu().numpy().view(KP_DTYPE).reshape(-1), single[i][0])
            np.testing.assert_array_equal(desc[i, :c].cpu().numpy(), single[i][1])
    ex.close()


def test_stage_intervals_share_one_clock():
    """orbfe_stage_intervals: the timed stage launches of two handles that run side by side on two streams, as intervals on the clock of
    one caller-recorded event -- ordered inside a handle, overlapping across the handles, and adding up to orbfe_stage_times' totals"""
    import torch
    w, h, nf, B = 1241, 376, 2000, 32
    imgs = torch.from_numpy(np.stack(synth.sequence(w, h, B, seq=9))).cuda()
    exs = [ORBextractor(nf), ORBextractor(nf)]
    cap = exs[0].max_keypoints(w, h)
    outs = [(torch.zeros((B, cap, 28), dtype=torch.uint8, device="cuda"), torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda"),
             torch.zeros(B, dtype=torch.int32, device="cuda")) for _ in exs]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for e, o, s in zip(exs, outs, streams):
        e.extract_batch_device(imgs, *o, stream=s)       # warm-up (plan, buffers)
    torch.cuda.synchronize()
    ref = torch.cuda.Event(enable_timing=True)
    ref.record(streams[0])
    torch.cuda.synchronize()
    for e in exs:
        e.profile(True); e.stage_times(reset=True)
    for _ in range(3):
        for e, o, s in zip(exs, outs, streams):
            e.extract_batch_device(imgs, *o, stream=s)
    torch.cuda.synchronize()
    ivs = [e.stage_intervals(ref) for e in exs]
    for e, iv in zip(exs, ivs):
        assert len(iv) >= 3 * 5 and all(b > a >= 0 for _, a, b in iv)
        assert all(iv[i][1] >= iv[i - 1][2] - 1e-3 for i in range(1, len(iv)))          # one stream: launch after launch
        tot = e.stage_times()
        for st in ("fast", "describe"):
            assert abs(sum(b - a for s_, a, b in iv if s_ == st) - tot[st][0]) < 1e-2 and tot[st][1] == 3
        assert e.stage_intervals(ref) == []                                              # drained
    lo = max(iv[0][1] for iv in ivs); hi = min(iv[-1][2] for iv in ivs)
    assert lo < hi                                                                       # the two handles' spans overlap
    np.testing.assert_array_equal(outs[0][0].cpu().numpy(), outs[1][0].cpu().numpy())
    for e in exs:
        e.profile(False); e.close()


def test_aligned_device_images_are_level0_in_place():
    """Device images whose rows start on 16-byte boundaries (pitch 1280 for a 1241-pixel row) are used as pyramid level 0 without
    the pitched copy: results, the level-0 plane seen through orbfe_device_pyramid / debug_pyramid, and a stereo match on the
    in-place pyramids must equal the copied path (tightly packed rows) and the oracle.  The caller's pad columns hold random
    bytes (the copy path replicates the last pixel there, the in-place kernels see whatever the caller left: no result may
    depend on them), and the tightest legal layout -- pitch = ceil16(w) = 1248, the buffer ending with the last row of the last
    image -- is run as well."""
    import torch
    from refactored_orb_slam2_amd._lib import KP_DTYPE
    from refactored_orb_slam2_amd.matcher import Matcher
    w, h, nf, B = 1241, 376, 2000, 3
    pairs = synth.sequence(w, h, B, seq=31, stereo=True)
    cap = None
    outs = {}
    pad_rng = np.random.default_rng(77)
    for name, pitch in (("packed", w), ("in_place", 1280), ("tight", 1248)):
        exL, exR, mt = ORBextractor(nf), ORBextractor(nf), Matcher(0)
        cap = exL.max_keypoints(w, h)
        bufs = []
        for side in (0, 1):
            host = pad_rng.integers(0, 256, (B, h, pitch), dtype=np.uint8)
            host[:, :, :w] = np.stack([p[side] for p in pairs])
            full = torch.from_numpy(host).cuda()
            bufs.append(full)
        z = lambda *s, dt=torch.uint8: torch.zeros(s, dtype=dt, device="cuda")
        kl, dl, nl, kr, dr, nr = z(B, cap, 28), z(B, cap, 32), z(B, dt=torch.int32), z(B, cap, 28), z(B, cap, 32), z(B, dt=torch.int32)
        ur, dp, ns = z(B, cap, dt=torch.float32), z(B, cap, dt=torch.float32), z(B, dt=torch.int32)
        st = torch.cuda.Stream()
        torch.cuda.synchronize()
        exL.extract_batch_device(bufs[0][:, :, :w], kl, dl, nl, stream=st)
        exR.extract_batch_device(bufs[1][:, :, :w], kr, dr, nr, stream=st)
        mt.stereo_match(exL, exR, kl, dl, nl, kr, dr, nr, 386.1448, 386.1448 / 718.856, ur, dp, ns, stream=st)
        torch.cuda.synchronize()
        exL.device_status(); exR.device_status()
        lvl0 = exL.debug_pyramid(1, 0)
        np.testing.assert_array_equal(lvl0, pairs[1][0])
        outs[name] = [t.cpu().numpy() for t in (kl, dl, nl, ur, dp, ns)]
        for hnd in (exL, exR, mt):
            hnd.close()
    for other in ("in_place", "tight"):
        for a, b in zip(outs["packed"], outs[other]):
            np.testing.assert_array_equal(a, b)
    orc = ol.OracleExtractor(nf)
    ok, od = orc(pairs[0][0])
    n0 = int(outs["in_place"][2][0])
    np.testing.assert_array_equal(outs["in_place"][0][0, :n0].view(KP_DTYPE).reshape(-1), ok)
    np.testing.assert_array_equal(outs["in_place"][1][0, :n0], od)
    assert int(outs["in_place"][5].min()) > 800


def test_batch_of_19_images_covers_both_workgroup_orders():
    """orient_describe hands whole images to XCDs in sets of eight and keeps the dispatch order for the remainder: 19 images
    exercise both in one launch; every image must equal its single-image extraction"""
    w, h, nf, B = 640, 480, 1000, 19
    imgs = synth.sequence(w, h, B, seq=9)
    ex = ORBextractor(nf)
    batch = ex.extract_batch(imgs)
    for i in (0, 7, 8, 15, 16, 18):
        k, d = ex(imgs[i])
        np.testing.assert_array_equal(batch[i][0], k); np.testing.assert_array_equal(batch[i][1], d)
    ok, od = ol.OracleExtractor(nf)(imgs[17])
    np.testing.assert_array_equal(batch[17][0], ok); np.testing.assert_array_equal(batch[17][1], od)
    ex.close()


def test_keypoint_invariants_at_full_size():
    """size-independent properties on the benchmark geometry (no oracle needed)."""
    w, h, nf = 1241, 376, 2000
    ex = ORBextractor(nf)
    k, d = ex(synth.frame(w, h, seq=9, f=4))
    sf = ex.GetScaleFactors()
    fpl = ex.features_per_level()
    assert np.all(np.diff(k["octave"]) >= 0)                       # level order
    for l in range(8):
        m = k["octave"] == l
        assert fpl[l] <= m.sum() <= fpl[l] + 3 or m.sum() < fpl[l]  # N..N+3 per level (SURVEY §7 hard part 2)
        lw, lh = ex.level_size(l)
        x, y = k["x"][m] / sf[l], k["y"][m] / sf[l]
        assert x.min() >= 19 - 1e-3 and x.max() <= lw - 19 + 1e-3 and y.min() >= 19 - 1e-3 and y.max() <= lh - 19 + 1e-3
        assert np.all(k["size"][m] == float(int(31 * sf[l])))
    assert np.all((k["angle"] >= 0) & (k["angle"] < 360)) and np.all(k["class_id"] == -1)
    assert np.all(k["response"] >= 7) and len(np.unique(d, axis=0)) > 0.98 * len(d)
    ex.close()


def test_edge_cases():
    from refactored_orb_slam2_amd import _lib
    ex = ORBextractor(500)
    # flat image: no corners at all -> zero keypoints (reference releases the descriptor matrix)
    k, d = ex(np.full((120, 160), 77, np.uint8))
    assert len(k) == 0 and d.shape == (0, 32)
    # small image whose upper levels have no FAST cell (nCols/nRows < 1, guarded divide-by-zero of the reference)
    img = synth.frame(100, 90, seq=1)
    k, d = ex(img)
    ok, od = ol.OracleExtractor(500)(img)
    np.testing.assert_array_equal(k, ok); np.testing.assert_array_equal(d, od)
    # empty image: silent no-op in the reference, ORBFE_ERR_EMPTY at the C ABI, None in the mirror
    assert ex(np.zeros((0, 0), np.uint8)) is None
    # capacity too small is reported, never silently truncated
    import ctypes as C
    img = synth.frame(640, 480, seq=2)
    kps = np.zeros(10, _lib.KP_DTYPE); desc = np.zeros((10, 32), np.uint8); n = C.c_int(0)
    rc = ex._L.orbfe_extract(ex._h, _lib.ptr(img), 640, 480, 640, _lib.ptr(kps), _lib.ptr(desc), 10, C.byref(n))
    assert rc == _lib.ERR_CAPACITY and n.value > 10
    # geometry change on the same handle re-plans
    k2, d2 = ex(synth.frame(320, 240, seq=2))
    ok2, od2 = ol.OracleExtractor(500)(synth.frame(320, 240, seq=2))
    np.testing.assert_array_equal(k2, ok2); np.testing.assert_array_equal(d2, od2)
    ex.close()


@pytest.mark.parametrize("params", [(1500, 1.2, 8, 20, 7), (800, 1.1, 12, 12, 5), (3000, 1.5, 4, 30, 10), (200, 1.2, 1, 20, 7)])
def test_other_extractor_parameters(params):
    nf, sf, nl, ini, mn = params
    img = synth.frame(752, 480, seq=10, f=2)
    ex = ORBextractor(nf, sf, nl, ini, mn)
    k, d = ex(img)
    ok, od = ol.OracleExtractor(nf, sf, nl, ini, mn)(img)
    np.testing.assert_array_equal(k, ok); np.testing.assert_array_equal(d, od)
    np.testing.assert_array_equal(ex.GetScaleFactors(), ol.OracleExtractor(nf, sf, nl, ini, mn).scale_factors)
    np.testing.assert_array_equal(ex.GetInverseScaleSigmaSquares(), ol.OracleExtractor(nf, sf, nl, ini, mn).inv_sigma2)
    ex.close()


@pytest.mark.parametrize("ini,mn", [(20, 7), (7, 7), (5, 12), (60, 2), (250, 1), (1, 1)])
def test_threshold_order_and_fallback(ini, mn):
    """FAST(iniThFAST) first, FAST(minThFAST) only for cells that came back empty (L/src/ORBextractor.cc:773-780), on an image
    whose left half is soft (every cell falls back), whose right half is textured (no cell does) and whose middle band is mixed;
    thresholds in the usual order, equal, inverted, and at the extremes.  Candidate lists per level and the final result vs the
    oracle."""
    from scipy.ndimage import gaussian_filter
    w, h, nf = 752, 480, 1200
    base = synth.frame(w, h, seq=12, f=3)
    soft = np.clip(gaussian_filter(base.astype(np.float32), 2.5) * 0.6 + 50, 0, 255).astype(np.uint8)
    img = base.copy()
    img[:, : w // 2] = soft[:, : w // 2]
    img[200:280, 300:450] = soft[200:280, 300:450]
    ex = ORBextractor(nf, 1.2, 8, ini, mn)
    orc = ol.OracleExtractor(nf, 1.2, 8, ini, mn)
    k, d = ex(img)
    ok, od = orc(img)
    for l in range(8):
        x, y, s = ex.debug_candidates(0, l)
        ox, oy, os_ = orc.level_candidates(l)
        np.testing.assert_array_equal(np.stack([x, y, s]), np.stack([ox, oy, os_]), err_msg=f"candidates of level {l} at ({ini}, {mn})")
    np.testing.assert_array_equal(k, ok); np.testing.assert_array_equal(d, od)
    ex.close()


@pytest.mark.parametrize("w,h,nf", [(91, 91, 300), (123, 95, 300), (1920, 1080, 4000), (2047, 211, 1500), (64, 64, 100)])
def test_unusual_geometries(w, h, nf):
    """cell sizes at the extremes (one 59 x 59 cell per level; 64 x 34 cell rows of a full-HD frame; a strip), odd widths,
    an image too small for any upper level: whole-pipeline equality with the oracle"""
    img = synth.frame(w, h, seq=7, f=1) if w <= 1600 else np.tile(synth.frame(960, 540, seq=7, f=1), (2, 2))[:h, :w].copy()
    ex = ORBextractor(nf)
    r = ex(img)
    ok, od = ol.OracleExtractor(nf)(img)
    assert r is not None
    np.testing.assert_array_equal(r[0], ok); np.testing.assert_array_equal(r[1], od)
    if w >= 91:
        assert len(ok) > 20
    ex.close()


@pytest.mark.parametrize("kind", ["white_noise", "noise_block", "noisy_frame", "two_blocks"])
def test_dense_texture_spills_keys_to_hbm(kind):
    """More FAST candidates per level than the LDS key budget -> the keys live in HBM and the quadtree takes its first subdivisions
    from count tables (octree_select_kernel's fast-forward).  white noise: tens of thousands of candidates, evenly spread -- the tree
    is finished inside the tables; a block of noise in a flat frame / two blocks: the tree goes deeper than the tables inside the
    block, so labels are written mid-way and the explicit rounds take over; synthetic frame + sensor noise: level 0 only"""
    rng = np.random.default_rng(0)
    h, w = 376, 1241
    if kind == "white_noise":
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    elif kind == "noise_block":
        img = np.full((h, w), 90, np.uint8)
        img[60:300, 700:1100] = rng.integers(0, 256, (240, 400), dtype=np.uint8)
    elif kind == "two_blocks":
        img = synth.frame(w, h, seq=3, f=0).copy()
        img[20:200, 30:330] = rng.integers(0, 256, (180, 300), dtype=np.uint8)
        img[150:370, 800:1230] = rng.integers(0, 256, (220, 430), dtype=np.uint8)
    else:
        img = np.clip(synth.frame(w, h, seq=5, f=0).astype(np.int16) + rng.integers(-8, 9, (h, w)), 0, 255).astype(np.uint8)
    ex = ORBextractor(2000)
    k, d = ex(img)
    assert len(ex.debug_candidates(0, 0)[0]) > (8000 if kind == "white_noise" else 1700)
    ok, od = ol.OracleExtractor(2000)(img)
    np.testing.assert_array_equal(k, ok); np.testing.assert_array_equal(d, od)
    ex.close()


def test_rccl_gather_self_check_world1():
    """with one visible GPU the exchange degrades to a self-check -- pack the per-frame records, gather them over
    RCCL (backend "nccl") at world size 1, unpack, compare with the unsharded extraction"""
    import socket
    import torch
    import torch.distributed as dist
    from refactored_orb_slam2_amd import sharding
    w, h, nf, F = 640, 480, 1000, 5
    imgs = synth.sequence(w, h, F, seq=4)
    ex = ORBextractor(nf)
    single = ex.extract_batch(imgs)
    cap = ex.max_keypoints(w, h)
    dimg = torch.from_numpy(np.stack(imgs)).cuda()
    kps = torch.zeros((F, cap, 28), dtype=torch.uint8, device="cuda")
    desc = torch.zeros((F, cap, 32), dtype=torch.uint8, device="cuda")
    n = torch.zeros(F, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ex.extract_batch_device(dimg, kps, desc, n)
    ex.sync()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        for mode in ("all", "root"):                        # all_gather_into_tensor, and gather to rank 0 (ncclSend / Recv group)
            ag = sharding.AsyncGather(n, kps, desc, mode=mode)   # the overlapped variant bench.py uses: really runs the collective
            ag.launch(n, kps, desc)
            n_all, k_all, d_all = ag.result()
            torch.cuda.synchronize()
            frames = sharding.unpack_records(n_all, k_all, d_all, F, 1)
            assert len(frames) == F
            for (k, d), (k0, d0) in zip(frames, single):
                np.testing.assert_array_equal(k, k0)
                np.testing.assert_array_equal(d, d0)
    finally:
        dist.destroy_process_group()
    ex.close()


def test_c_abi_record_gather_world1_and_comm_init_all():
    """The batched-sequence exchange behind the C ABI (include/orbfe.h: orbfe_gather_*; RCCL called by liborbfe, no
    torch.distributed in the data path): pack the per-frame records of a device batch, gather them at world size 1 --
    ncclAllGather, then the rank-0 form (grouped send / recv; rank 0's own records are a stream-ordered copy) --, unpack,
    compare with the unsharded extraction.  Both ways of making the communicator: unique id + orbfe_gather_create (one process
    per GPU) and orbfe_gather_create_all (one process, one thread per GPU: ncclCommInitAll)."""
    import ctypes as C
    import torch
    from refactored_orb_slam2_amd import _lib, sharding
    w, h, nf, F = 640, 480, 1000, 5
    imgs = synth.sequence(w, h, F, seq=4)
    ex = ORBextractor(nf)
    single = ex.extract_batch(imgs)
    cap = ex.max_keypoints(w, h)
    dimg = torch.from_numpy(np.stack(imgs)).cuda()
    kps = torch.zeros((F, cap, 28), dtype=torch.uint8, device="cuda")
    desc = torch.zeros((F, cap, 32), dtype=torch.uint8, device="cuda")
    n = torch.zeros(F, dtype=torch.int32, device="cuda")
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    ex.extract_batch_device(dimg, kps, desc, n, stream=st)

    def check(res):
        torch.cuda.synchronize()
        frames = sharding.unpack_records(*res, F, 1)
        for (k, d), (k0, d0) in zip(frames, single):
            np.testing.assert_array_equal(k, k0)
            np.testing.assert_array_equal(d, d0)

    g = sharding.RcclGather(sharding.RcclGather.unique_id(), 0, 1, 0)
    for mode in ("all", "root"):
        check(g.gather(n, kps, desc, mode=mode, stream=st))      # same stream as the extraction: ordered behind it
    st.synchronize()
    check(g.gather(n, kps, desc, mode="all"))                    # the handle's own stream + orbfe_gather_sync
    g.sync()
    g.close()
    L = _lib.lib()
    hs = (C.c_void_p * 1)()
    assert L.orbfe_gather_create_all(1, None, hs) == 0, L.orbfe_last_error()
    rank, world = C.c_int(-1), C.c_int(-1)
    assert L.orbfe_gather_rank(C.c_void_p(hs[0]), C.byref(rank), C.byref(world)) == 0 and (rank.value, world.value) == (0, 1)
    out = [torch.empty_like(t) for t in (n, kps, desc)]
    assert L.orbfe_gather_records(C.c_void_p(hs[0]), _lib.ptr(n), _lib.ptr(kps), _lib.ptr(desc), F, cap, 1, _lib.ptr(out[0]),
                                  _lib.ptr(out[1]), _lib.ptr(out[2]), None) == 0, L.orbfe_last_error()
    assert L.orbfe_gather_sync(C.c_void_p(hs[0])) == 0
    check(out)
    # argument errors are codes
    assert L.orbfe_gather_records(C.c_void_p(hs[0]), _lib.ptr(n), _lib.ptr(kps), _lib.ptr(desc), F, cap, 7, None, None, None, None) == -1
    assert L.orbfe_gather_records(C.c_void_p(hs[0]), _lib.ptr(n), _lib.ptr(kps), _lib.ptr(desc), F, cap, 0, None, None, None, None) == -1
    assert L.orbfe_gather_destroy(C.c_void_p(hs[0])) == 0
    ex.close()


def test_prepare_moves_the_first_call_cost_out_of_the_first_frame():
    """orbfe_extractor_prepare / orbfe_frontend_prepare (include/orbfe.h, warm-up): the plan, the work space, the code objects and
    the launch graph exist before the first frame.  Results after a prepare are the results without one (the synthetic warm-up
    frames leave no state behind), and the first real call costs what a later one costs -- not the ~30 ms of a cold handle."""
    import ctypes as C
    import time
    from refactored_orb_slam2_amd import _lib
    w, h, nf = 1241, 376, 2000
    img = synth.sequence(w, h, 2, seq=12)
    cold = ORBextractor(nf)
    k0, d0 = cold(img[0])
    warm = ORBextractor(nf)
    warm.prepare(w, h, 1)
    t = []
    for i in range(6):
        t0 = time.perf_counter()
        k1, d1 = warm(img[0])
        t.append(time.perf_counter() - t0)
    np.testing.assert_array_equal(k1, k0)
    np.testing.assert_array_equal(d1, d0)
    assert t[0] < 3 * sorted(t)[len(t) // 2] + 1e-3, t          # no first-call spike left
    # both eyes + the calling thread's matcher handle; then a stereo match and a search on real frames equal the oracle's
    L = _lib.lib()
    exL, exR = ORBextractor(nf), ORBextractor(nf)
    assert L.orbfe_frontend_prepare(exL._h, exR._h, w, h, 0) == 0, L.orbfe_last_error()
    assert L.orbfe_frontend_prepare(exL._h, None, w, h, 500) == 0, L.orbfe_last_error()      # monocular form
    pair = synth.sequence(w, h, 1, seq=13, stereo=True)[0]
    kl, dl = exL(pair[0]); kr, dr = exR(pair[1])
    orc = ol.OracleExtractor(nf)
    okl, odl = orc(pair[0])
    np.testing.assert_array_equal(kl, okl); np.testing.assert_array_equal(dl, odl)
    for e in (cold, warm, exL, exR):
        e.close()
    assert L.orbfe_extractor_prepare(None, w, h, 1) == -1 and L.orbfe_extractor_prepare(cold._h, 0, h, 1) == -1


@pytest.mark.parametrize("geom", ["kitti", "tum", "odd"])
def test_blur_on_the_matrix_cores_is_byte_equal(geom):
    """gauss_blur7_mfma_kernel (the 7-tap passes as v_mfma_i32_16x16x64_i8 band-matrix products, REFLECT_101 folded into the edge
    matrices / the row fetch; DESIGN lesson 31) against the LDS kernel and the oracle: every blurred plane, every keypoint and
    descriptor.  `odd`: a size whose levels end inside a strip / a window and whose smallest level is 8 pixels high."""
    w, h, nf, nl = {"kitti": (1241, 376, 2000, 8), "tum": (640, 480, 1000, 8), "odd": (333, 67, 300, 12)}[geom]
    imgs = synth.sequence(w, h, 2, seq=23)
    a = ORBextractor(nf, 1.2, nl, 20, 7)
    b = ORBextractor(nf, 1.2, nl, 20, 7)
    assert b._L.orbfe_debug_blur_kernel(b._h, 1) == 0
    ra, rb = a.extract_batch(imgs), b.extract_batch(imgs)
    for (ka, da), (kb, db) in zip(ra, rb):
        np.testing.assert_array_equal(ka, kb)
        np.testing.assert_array_equal(da, db)
    orc = ol.OracleExtractor(nf, 1.2, nl, 20, 7)
    ok, od = orc(imgs[1])
    np.testing.assert_array_equal(rb[1][0], ok); np.testing.assert_array_equal(rb[1][1], od)
    for lvl in range(nl):
        lw, lh = b.level_size(lvl, w, h)
        if lw < 8 or lh < 8:
            continue
        np.testing.assert_array_equal(b.debug_blurred(1, lvl), a.debug_blurred(1, lvl), err_msg=f"blurred level {lvl}")
        ob = orc.level_blurred(lvl)          # None: the reference skips the blur of a level without keypoints
        if ob is not None:
            np.testing.assert_array_equal(b.debug_blurred(1, lvl), ob, err_msg=f"blurred level {lvl} vs oracle")
    # the one-image latency path (captured launch graph) with the other kernel, and back
    for _ in range(4):
        k1, d1 = b(imgs[0])
    np.testing.assert_array_equal(k1, ra[0][0]); np.testing.assert_array_equal(d1, ra[0][1])
    assert b._L.orbfe_debug_blur_kernel(b._h, 0) == 0
    k2, d2 = b(imgs[0])
    np.testing.assert_array_equal(k2, ra[0][0]); np.testing.assert_array_equal(d2, ra[0][1])
    assert b._L.orbfe_debug_blur_kernel(b._h, 3) == -1
    a.close(); b.close()


@pytest.mark.parametrize("geom", ["kitti", "tum", "euroc", "odd", "sf15", "sf11", "tiny"])
def test_fused_level_chain_equals_separate_launches(geom):
    """blur_level_kernel<true> -- launch l blurs level l and writes level l + 1 from the same staged windows (cv::resize
    INTER_LINEAR, L/src/ORBextractor.cc:1041-1065, and GaussianBlur, :1017-1019, reading each level once) -- against the
    pipeline of rounds 1-4 (resize chain, then one blur launch: orbfe_debug_blur_kernel(e, 2)) and the oracle: every pyramid
    plane, every blurred plane, keypoints and descriptors.  `sf11`: a scale factor whose tiles would own more destination rows
    than the kernel's thread layout holds, so every step falls back to a resize and a blur launch of its own; `sf15`: a coarser
    chain; `odd` / `tiny`: levels that end inside a tile, a smallest level of a few pixels."""
    w, h, nf, sf, nl = {"kitti": (1241, 376, 2000, 1.2, 8), "tum": (640, 480, 1000, 1.2, 8), "euroc": (752, 480, 1200, 1.2, 8),
                        "odd": (333, 67, 300, 1.2, 12), "sf15": (1241, 376, 1500, 1.5, 5), "sf11": (641, 377, 800, 1.1, 12),
                        "tiny": (70, 61, 50, 1.2, 8)}[geom]
    imgs = synth.sequence(w, h, 3, seq=29)
    a = ORBextractor(nf, sf, nl, 20, 7)
    b = ORBextractor(nf, sf, nl, 20, 7)
    assert b._L.orbfe_debug_blur_kernel(b._h, 2) == 0
    ra, rb = a.extract_batch(imgs), b.extract_batch(imgs)
    for (ka, da), (kb, db) in zip(ra, rb):
        np.testing.assert_array_equal(ka, kb)
        np.testing.assert_array_equal(da, db)
    orc = ol.OracleExtractor(nf, sf, nl, 20, 7)
    ok, od = orc(imgs[2])
    np.testing.assert_array_equal(ra[2][0], ok); np.testing.assert_array_equal(ra[2][1], od)
    for lvl in range(nl):
        np.testing.assert_array_equal(a.debug_pyramid(2, lvl), b.debug_pyramid(2, lvl), err_msg=f"pyramid level {lvl}")
        np.testing.assert_array_equal(a.debug_pyramid(2, lvl), orc.level_pixels(lvl), err_msg=f"pyramid level {lvl} vs oracle")
        np.testing.assert_array_equal(a.debug_blurred(2, lvl), b.debug_blurred(2, lvl), err_msg=f"blurred level {lvl}")
        ob = orc.level_blurred(lvl)          # None: the reference skips the blur of a level without keypoints
        if ob is not None:
            np.testing.assert_array_equal(a.debug_blurred(2, lvl), ob, err_msg=f"blurred level {lvl} vs oracle")
    # the one-image latency path (captured launch graph) runs the same chain
    for _ in range(4):
        k1, d1 = a(imgs[0])
    np.testing.assert_array_equal(k1, rb[0][0]); np.testing.assert_array_equal(d1, rb[0][1])
    a.close(); b.close()


def test_replayed_launch_graph_keeps_the_host_state_current():
    """A one-image host call is replayed from a captured hipGraph from its third use on (extractor.cpp): the launches' host-side
    bookkeeping does not run then.  The number of images of the last call guards orbfe_debug_* / orbfe_pyramid_level / the stereo
    entry points: after a five-image device batch a REPLAYED one-image call must leave it at 1 (image 1 is refused, image 0 is
    the new image), and the results of the replayed call are those of a fresh handle."""
    import torch
    w, h, nf = 640, 480, 1000
    imgs = synth.sequence(w, h, 6, seq=17)
    ex = ORBextractor(nf)
    for _ in range(4):                       # direct, direct, capture + replay, replay
        k_warm, d_warm = ex(imgs[5])
    cap = ex.max_keypoints(w, h)
    dimg = torch.from_numpy(np.stack(imgs[:5])).cuda()
    kps = torch.zeros((5, cap, 28), dtype=torch.uint8, device="cuda")
    desc = torch.zeros((5, cap, 32), dtype=torch.uint8, device="cuda")
    n = torch.zeros(5, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ex.extract_batch_device(dimg, kps, desc, n)
    ex.sync()
    np.testing.assert_array_equal(ex.debug_pyramid(4, 0), imgs[4])       # five images resident
    k, d = ex(imgs[5])                                                   # replayed graph
    np.testing.assert_array_equal(k, k_warm); np.testing.assert_array_equal(d, d_warm)
    np.testing.assert_array_equal(ex.debug_pyramid(0, 0), imgs[5])
    with pytest.raises(Exception):
        ex.debug_pyramid(1, 0)                                           # slot 1 belongs to the earlier batch: refused
    fresh = ORBextractor(nf)
    k0, d0 = fresh(imgs[5])
    np.testing.assert_array_equal(k, k0); np.testing.assert_array_equal(d, d0)
    ex.close(); fresh.close()


def test_bench_spawns_its_ranks_and_checks_the_gather():
    """`bench.py --gpus 2` started without a launcher must produce two ranks by itself (one process each; on this one-GPU box
    both share the device and the gather runs over gloo, because RCCL refuses two ranks per device) and print n_gpus = 2.  Every
    rank compares its slot of the gathered records with its own records before the line is printed (bench.py)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ORBFE_BENCH_SHARE_DEVICE="1")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    for mode, name in (("all", "all_gather"), ("root", "gather to rank 0")):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--frames", "8", "--cpu-sample", "0",
                            "--e2e-steps", "0", "--gather", mode], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        out = json.loads(line)
        assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "frame-shard x2" and out["value"] > 0
        assert name in out["config"]["collective"]
        g = out["config"]["gather"]
        assert g["mode"] == mode and g["record_bytes_per_rank"] > 8 * 2000 * 60 and g["standalone_ms_rank0"] > 0
        assert len(out["config"]["frames_per_s_by_rank"]) == 2 and min(out["config"]["frames_per_s_by_rank"]) > 0


def test_bench_times_the_c_abi_gather_when_asked():
    """`bench.py --gather-impl cabi`: the record gather of every step goes through the C ABI (orbfe_gather_records: RCCL called by
    liborbfe, the collective a C++ host uses) instead of torch.distributed; on one GPU a one-rank communicator -- the code path,
    the bench line's `config.gather` block and the check that the gathered records equal the step's own."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for mode in ("all", "root"):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--frames", "8", "--cpu-sample", "0", "--e2e-steps", "0",
                            "--per-frame", "0", "--content-steps", "0", "--gather-impl", "cabi", "--gather", mode], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        g = out["config"]["gather"]
        assert g["impl"] == "cabi" and g["mode"] == mode and g["own_slot_equal_rank0"] is True, g
        assert "orbfe_gather_records" in out["config"]["collective"] and out["value"] > 0
