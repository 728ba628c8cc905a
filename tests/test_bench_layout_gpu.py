"""Parity at the configuration bench.py TIMES (round-5 review, item 1): 256 KITTI stereo frames per step, three sets of handles and
buffers taking the steps in turn, left | right extractor on two HIP streams, the matching half on a third -- and the C-ABI pipeline
handle (orbfe_pipeline_*) at batch 256 with three slots.  The test drives bench.StepRig, the object bench.py's timed loop calls, so
what is timed and what is compared with the oracle are one function.

Reference contract: two extractors at once (L/src/Frame.cc:87-90), Frame::ComputeStereoMatches (L/src/Frame.cc:477-646),
SearchByProjection(cur, last) (L/src/ORBmatcher.cc:1247-1383)."""
import numpy as np
import pytest

from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu

F_FULL = 256
SAMPLED = (0, 1, 7, 8, 31, 32, 63, 64, 127, 128, 200, 255)   # >= 12 frames per set; XCD-group edges (x8), tile-batch edges (x64), both ends


def _host_outputs(rig, B):
    """Every field a step leaves in one set's buffers, as numpy arrays (rows cut to the frame's count by the caller)."""
    import torch
    torch.cuda.synchronize()
    g = lambda t: t.cpu().numpy()
    return dict(nl=g(B.nl), kl=g(B.kl), dl=g(B.dl), nr=g(B.nr), kr=g(B.kr), dr=g(B.dr), ur=g(B.ur), depth=g(B.depth),
                n_stereo=g(B.n_stereo), n_track=g(B.n_track), assigned=g(B.assigned), blocked=g(B.blocked), nq=g(B.nq))


def _assert_sets_equal(a, b, what):
    """Byte-compare ALL frames of two output sets (the first n entries of every row; the rest is scratch)."""
    np.testing.assert_array_equal(a["nl"], b["nl"], err_msg=f"{what}: left counts")
    np.testing.assert_array_equal(a["nr"], b["nr"], err_msg=f"{what}: right counts")
    np.testing.assert_array_equal(a["n_stereo"], b["n_stereo"], err_msg=f"{what}: stereo counts")
    np.testing.assert_array_equal(a["n_track"], b["n_track"], err_msg=f"{what}: tracked counts")
    ml = np.arange(a["kl"].shape[1])[None, :] < a["nl"][:, None]
    mr = np.arange(a["kr"].shape[1])[None, :] < a["nr"][:, None]
    for key, m in (("kl", ml), ("dl", ml), ("ur", ml), ("depth", ml), ("assigned", ml), ("blocked", ml), ("kr", mr), ("dr", mr)):
        x, y = a[key][m], b[key][m]
        if not np.array_equal(x.view(np.uint8), y.view(np.uint8)):   # bytes: NaN-safe, -0.0-safe
            rows = np.nonzero((a[key] != b[key]).reshape(a[key].shape[0], -1).any(1) if a[key].dtype.kind != "f"
                              else (a[key].view(np.uint32) != b[key].view(np.uint32)).reshape(a[key].shape[0], -1).any(1))[0]
            raise AssertionError(f"{what}: field {key} differs (frames {rows[:8].tolist()} ...)")


def _oracle_frames(rig, frames):
    """The oracle's whole per-frame path for `frames` and their predecessors ((f - 1) % F): keypoints, descriptors, mvuRight / mvDepth,
    and the SearchByProjection(cur, last) assignment against the predecessor's un-projected stereo points."""
    from refactored_orb_slam2_amd._lib import KP_DTYPE  # noqa: F401  (dtype of the oracle's keypoint records)
    cfg, F, W, H = rig.cfg, rig.F, rig.W, rig.H
    oL, oR = ol.OracleExtractor(rig.NFEAT), ol.OracleExtractor(rig.NFEAT)
    sf, isf = oL.scale_factors, oL.inv_scale_factors
    need = sorted(set(frames) | {(f - 1) % F for f in frames})
    per = {}
    for f in need:
        Li, Ri = rig.data[f]
        kL, dL = oL(Li)
        lpx = [oL.level_pixels(l) for l in range(8)]
        kR, dR = oR(Ri)
        rpx = [oR.level_pixels(l) for l in range(8)]
        n_st, ur, depth = ol.compute_stereo_matches(kL, dL, kR, dR, lpx, rpx, sf, isf, float(cfg["bf"]), float(rig.mb))
        pts = ol.unproject_stereo(rig.cams_np[f:f + 1], kL, dL, depth)
        per[f] = dict(kL=kL, dL=dL, kR=kR, dR=dR, ur=ur, depth=depth, pts=pts)
    for f in frames:
        p, c = per[(f - 1) % F], per[f]
        oq = ol.track_queries(rig.poses_np[f:f + 1], p["pts"])
        of = ol.OracleFrame(c["kL"], c["dL"], sf, 0, W, 0, H, c["ur"])
        c["nm"], c["assigned"], c["blocked"] = of.search_by_projection_frame(oq, True)
        c["nq"] = len(p["kL"])
    return per


def _assert_frame_equals_oracle(out, f, o, what):
    from refactored_orb_slam2_amd._lib import KP_DTYPE
    n, nr = int(out["nl"][f]), int(out["nr"][f])
    assert n == len(o["kL"]) and nr == len(o["kR"]), (what, f, n, len(o["kL"]), nr, len(o["kR"]))
    k = out["kl"][f].reshape(-1).view(KP_DTYPE)[:n]
    np.testing.assert_array_equal(k, o["kL"], err_msg=f"{what}: left keypoints of frame {f}")
    np.testing.assert_array_equal(out["dl"][f][:n], o["dL"], err_msg=f"{what}: left descriptors of frame {f}")
    kr = out["kr"][f].reshape(-1).view(KP_DTYPE)[:nr]
    np.testing.assert_array_equal(kr, o["kR"], err_msg=f"{what}: right keypoints of frame {f}")
    np.testing.assert_array_equal(out["dr"][f][:nr], o["dR"], err_msg=f"{what}: right descriptors of frame {f}")
    assert out["ur"][f][:n].tobytes() == np.asarray(o["ur"], np.float32).tobytes(), f"{what}: mvuRight of frame {f}"
    assert out["depth"][f][:n].tobytes() == np.asarray(o["depth"], np.float32).tobytes(), f"{what}: mvDepth of frame {f}"
    if "nm" in o:
        assert int(out["nq"][f]) == o["nq"], (what, f)
        assert int(out["n_track"][f]) == o["nm"], (what, f, int(out["n_track"][f]), o["nm"])
        np.testing.assert_array_equal(out["assigned"][f][:n], o["assigned"], err_msg=f"{what}: tracked assignment of frame {f}")
        np.testing.assert_array_equal(out["blocked"][f][:n], o["blocked"], err_msg=f"{what}: blocked flags of frame {f}")


@pytest.fixture(scope="module")
def rig():
    import bench
    r = bench.StepRig(bench.CONFIGS["kitti_stereo"], F_FULL, n_sets=3, lr_streams=2)
    yield r
    r.close()


@pytest.fixture(scope="module")
def oracle(rig):
    return _oracle_frames(rig, SAMPLED)


def test_bench_step_at_full_batch_matches_oracle(rig, oracle):
    """bench.py's own step at its own batch size and stream layout: eight consecutive steps without a host synchronisation in
    between (sets 0, 1, 2, 0, 1, 2, 0, 1 -- every set is written while the matching half of the step before it still runs), then
    (a) twelve sampled frames of EVERY set equal the oracle in every field the step produces, (b) ALL 256 frames of the three sets
    are byte-equal to each other, (c) and to what a one-stream, one-set step leaves (every kernel alone on the chip)."""
    assert rig.two_sets and rig.n_sets == 3 and rig.lr["n"] == 2 and rig.F == 256
    for _ in range(8):
        rig.step()
    rig.barrier()
    for e in rig.extractors:
        e.device_status()
    outs = [_host_outputs(rig, B) for B in rig.all_buffers()]
    for s, out in enumerate(outs):
        for f in SAMPLED:
            _assert_frame_equals_oracle(out, f, oracle[f], f"set {s}")
    # (frame 0 is searched with frame 255's points, as the bench's cyclic batch does: the sequence does not wrap, few matches there)
    assert min(oracle[f]["nm"] for f in SAMPLED if f) > 500 and min(len(oracle[f]["kL"]) for f in SAMPLED) >= 1990
    _assert_sets_equal(outs[0], outs[1], "set 0 vs set 1")
    _assert_sets_equal(outs[0], outs[2], "set 0 vs set 2")
    # the one-stream / one-set path (the layout of rounds 1-4 and of the stage parity tests)
    rig.lr["n"] = 1
    rig.step(); rig.barrier()
    one = _host_outputs(rig, rig.B0)
    rig.lr["n"] = 2
    _assert_sets_equal(one, outs[1], "one-stream step vs set 1 of the timed layout")
    # and back: the next pipelined steps start from the one-stream step's state
    for _ in range(4):
        rig.step()
    rig.barrier()
    for s, B in enumerate(rig.all_buffers()):
        _assert_sets_equal(_host_outputs(rig, B), one, f"set {s} after the one-stream step")
    chk = rig.self_check()   # what bench.py puts into config.self_check
    assert chk["sets_equal"] and chk["equals_one_stream"] and chk["sets"] == 3, chk


def test_self_check_sees_a_difference(rig):
    """The digest bench.py compares is sensitive: one descriptor byte of one frame of one set changes it."""
    for _ in range(3):
        rig.step()
    rig.barrier()
    d0 = rig.output_digest(rig.more_B[0])
    assert d0 == rig.output_digest(rig.B0)
    rig.more_B[0].dl[200, 5, 3] ^= 1
    assert rig.output_digest(rig.more_B[0]) != d0
    rig.more_B[0].dl[200, 5, 3] ^= 1
    assert rig.output_digest(rig.more_B[0]) == d0


def test_pipeline_handle_at_batch_256_three_slots_equals_the_step(rig, oracle):
    """orbfe_pipeline_* (what a C++ host drives; examples/stereo_kitti.cc --batch 256) with the bench's 256 frames resident in three
    slots: six chunks submitted back to back (every slot twice, no wait in between -- chunk k + 1's extraction beside chunk k's
    matching half), each chunk's results equal the torch-driven step's (hence, frame by frame, the oracle's: sampled directly too)."""
    from refactored_orb_slam2_amd._lib import KP_DTYPE
    from refactored_orb_slam2_amd.pipeline import StereoPipeline
    import bench
    cfg, W, H, F = rig.cfg, rig.W, rig.H, rig.F
    for _ in range(3):
        rig.step()
    rig.barrier()
    ref = _host_outputs(rig, rig.B0)
    with StereoPipeline(W, H, F, cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], cfg["bf"], cfg["th"], n_features=rig.NFEAT, slots=3) as p:
        for s in range(3):
            p.poses(s)["cx"] = np.float32(cfg["cx"]) + np.float32(bench.SHIFT_X)
            L, R = p.left(s), p.right(s)
            pitch = L.shape[2]
            assert pitch == rig.PITCH
            L[:] = rig.hL.numpy(); R[:] = rig.hR.numpy()
        # first round uploads (and seeds the carried last frame), second round runs from the device-resident images
        for s in range(3):
            p.submit(s, F, has_predecessor=s > 0)
        for s in range(3):
            p.wait(s)
        p.left(0)[3, :, :W] = 0   # proves the resident round reads HBM, not the host slot
        got = []
        for rnd in range(2):
            for s in range(3):
                p.submit_resident(s, F, has_predecessor=True)
            for s in range(3):
                p.wait(s)
                o = p.output(s)
                got.append({k: np.array(v, copy=True) for k, v in o.items() if k != "cap"})
        cap = min(got[0]["kps_left"].shape[1], ref["kl"].shape[1])
        for i, g in enumerate(got):
            what = f"pipeline chunk {i}"
            np.testing.assert_array_equal(g["n_left"], ref["nl"], err_msg=what)
            np.testing.assert_array_equal(g["n_right"], ref["nr"], err_msg=what)
            np.testing.assert_array_equal(g["n_stereo"], ref["n_stereo"], err_msg=what)
            np.testing.assert_array_equal(g["n_tracked"], ref["n_track"], err_msg=what)
            m = np.arange(cap)[None, :] < ref["nl"][:, None]
            rk = ref["kl"].reshape(F, -1).view(KP_DTYPE).reshape(F, -1)
            assert g["kps_left"][:, :cap][m].tobytes() == rk[:, :cap][m].tobytes(), what
            assert g["desc_left"][:, :cap][m].tobytes() == ref["dl"][:, :cap][m].tobytes(), what
            assert g["u_right"][:, :cap][m].tobytes() == ref["ur"][:, :cap][m].tobytes(), what
            assert g["depth"][:, :cap][m].tobytes() == ref["depth"][:, :cap][m].tobytes(), what
            assert g["assigned"][:, :cap][m].tobytes() == ref["assigned"][:, :cap][m].tobytes(), what
        # directly against the oracle as well (frames 0, 8, 64, 255 of the last chunk)
        g = got[-1]
        for f in (0, 8, 64, 255):
            o = oracle[f]
            n = int(g["n_left"][f])
            np.testing.assert_array_equal(g["kps_left"][f, :n], o["kL"]); np.testing.assert_array_equal(g["desc_left"][f, :n], o["dL"])
            assert g["depth"][f, :n].tobytes() == np.asarray(o["depth"], np.float32).tobytes()
            assert int(g["n_tracked"][f]) == o["nm"]
            np.testing.assert_array_equal(g["assigned"][f, :n], o["assigned"])


def test_pipeline_output_mask_copies_what_was_asked_for(rig):
    """orbfe_pipeline_config.output_mask: a handle that copies only the tracked assignments (+ the counts, always) delivers those
    byte-equal to the full handle's, leaves the other host blocks untouched, and keeps everything readable in HBM."""
    from refactored_orb_slam2_amd.pipeline import OUT_ASSIGNED, OUT_COUNTS, StereoPipeline
    import bench
    cfg, W, H = rig.cfg, rig.W, rig.H
    F = 16
    outs = {}
    for mask in (0, OUT_ASSIGNED, OUT_COUNTS):
        with StereoPipeline(W, H, F, cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], cfg["bf"], cfg["th"], n_features=rig.NFEAT, slots=2,
                            output_mask=mask) as p:
            for s in range(2):
                p.poses(s)["cx"] = np.float32(cfg["cx"]) + np.float32(bench.SHIFT_X)
                p.left(s)[:] = rig.hL.numpy()[s * F:(s + 1) * F]; p.right(s)[:] = rig.hR.numpy()[s * F:(s + 1) * F]
            p.submit(0, F, has_predecessor=False); p.submit(1, F, has_predecessor=True)
            p.wait(0); p.wait(1)
            outs[mask] = [{k: np.array(v, copy=True) for k, v in p.output(s).items() if k != "cap"} for s in range(2)]
    for s in range(2):
        full, am, cm = outs[0][s], outs[OUT_ASSIGNED][s], outs[OUT_COUNTS][s]
        assert int(full["n_tracked"][1:].min()) > 500
        for key in ("n_left", "n_right", "n_stereo", "n_tracked"):
            np.testing.assert_array_equal(am[key], full[key]); np.testing.assert_array_equal(cm[key], full[key])
        n = full["n_left"]
        m = np.arange(full["assigned"].shape[1])[None, :] < n[:, None]
        np.testing.assert_array_equal(am["assigned"][m], full["assigned"][m])
        # blocks that were not asked for stay as the handle's warm-up left them (zeroed at creation): nothing of this chunk arrived
        assert not am["desc_left"][m].any() and not am["u_right"][m].any() and not cm["assigned"][m].any() and not cm["depth"][m].any()
