"""The C++ drop-in classes (csrc/host/): build everywhere, behave without a GPU, parity on the GPU."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "_build", "test_dropin")


def _build():
    from refactored_orb_slam2_amd import _lib
    from tests import oracle_lib as ol
    _lib.build()
    ol.build()
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp")], check=True, capture_output=True)


def test_dropin_builds_and_fails_loudly_without_device():
    _build()
    import ctypes as C
    from refactored_orb_slam2_amd import _lib
    n = C.c_int(0)
    if _lib.lib().orbfe_device_count(C.byref(n)) == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([EXE, "--nodevice"], capture_output=True, text=True)
    assert r.returncode == 0 and "nodevice ok" in r.stdout
    assert "no CPU fallback" in r.stderr  # logged, never thrown (SURVEY.md §8(b) Errors)


@pytest.mark.gpu
@pytest.mark.parametrize("geom", [(1241, 376, 2000), (640, 480, 1000)])
def test_dropin_parity_on_gpu(tmp_path, geom):
    from refactored_orb_slam2_amd import synth
    _build()
    w, h, nf = geom
    img = synth.frame(w, h, seq=13, f=2)
    p = str(tmp_path / "img.raw")
    img.tofile(p)
    r = subprocess.run([EXE, p, str(w), str(h), str(nf)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "dropin ok" in r.stdout and "SearchLocalPoints ok" in r.stdout and "SearchByBoW ok" in r.stdout and "SearchForTriangulation ok" in r.stdout and "ComputeBoW ok" in r.stdout
    assert "ComputeStereoMatches ok" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("geom", [(640, 480, 1000), (1241, 376, 2000)])
def test_matcher_dropin_all_eleven_methods_on_gpu(tmp_path, geom):
    """csrc/host/ORBmatcher.cc -- ORB_SLAM2::ORBmatcher with the reference's declarations -- linked against mock Frame /
    KeyFrame / MapPoint headers: the eleven methods, DescriptorDistance and the protected helpers vs the oracle, Fuse x2 and
    SearchBySim3 vs a literal sequential replay of the reference loops on a twin world."""
    from refactored_orb_slam2_amd import synth
    _build()
    w, h, nf = geom
    a, b = synth.sequence(w, h, 2, seq=17)
    pa, pb = str(tmp_path / "a.raw"), str(tmp_path / "b.raw")
    a.tofile(pa); b.tofile(pb)
    exe = os.path.join(ROOT, "tests", "cpp", "_build", "test_matcher_dropin")
    r = subprocess.run([exe, pa, pb, str(w), str(h), str(nf)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    for tag in ("SearchByProjection(F, points) ok", "SearchByProjection(cur, last) ok", "SearchByProjection(F, KF, found) ok",
                "SearchByProjection(KF, Scw) ok", "SearchByBoW ok", "SearchForInitialization ok", "SearchForTriangulation ok",
                "SearchBySim3 ok", "Fuse ok", "Fuse(Sim3) ok", "matcher dropin ok"):
        assert tag in r.stdout, r.stdout


# ---- the C ABI from plain C99 (examples/extract_c.c): what a cgo / JNI / FFI binding would do
def _build_c_example(tmp_path):
    from refactored_orb_slam2_amd import _lib
    _lib.build()
    exe = str(tmp_path / "extract_c")
    csrc = os.path.join(ROOT, "refactored_orb_slam2_amd", "csrc")
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-O2", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", "extract_c.c"), "-L" + csrc, "-lorbfe", "-Wl,-rpath," + csrc, "-Wl,-rpath,/opt/rocm/lib",
                    "-o", exe], check=True, capture_output=True)
    return exe


def _png(path, img):
    import struct, zlib
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    h, w = img.shape
    rows = b"".join(b"\x00" + img[y].tobytes() for y in range(h))
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(rows, 6)) +
                chunk(b"IEND", b""))


def test_c99_example_builds_and_fails_loudly_without_device(tmp_path):
    """include/orbfe.h is valid C99 (-pedantic -Werror) and the plain-C caller gets ORBFE_ERR_NO_DEVICE, not a CPU result."""
    import ctypes as C
    from refactored_orb_slam2_amd import _lib, synth
    exe = _build_c_example(tmp_path)
    n = C.c_int(0)
    if _lib.lib().orbfe_device_count(C.byref(n)) == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    _png(tmp_path / "a.png", synth.sequence(320, 240, 1, seq=4)[0])
    r = subprocess.run([exe, str(tmp_path / "a.png")], capture_output=True, text=True)
    assert r.returncode == 3 and "no CPU fallback" in r.stderr, (r.returncode, r.stderr)


@pytest.mark.gpu
def test_c99_example_matches_the_oracle(tmp_path):
    from refactored_orb_slam2_amd import synth
    from tests import oracle_lib as ol
    exe = _build_c_example(tmp_path)
    img = synth.sequence(640, 480, 1, seq=4)[0]
    _png(tmp_path / "a.png", img)
    r = subprocess.run([exe, str(tmp_path / "a.png"), "1000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    k, d = ol.OracleExtractor(1000)(img)
    s = 0
    for b in d.reshape(-1).tolist():
        s = ((s * 16777619) & 0xffffffff) ^ b
    assert f"640x480: {len(k)} keypoints, descriptor checksum {s:08x}" in r.stdout, r.stdout
    assert f"kp 0: ({k['x'][0]:.1f}, {k['y'][0]:.1f}) octave {k['octave'][0]}" in r.stdout


# ---- the C++ sequence driver on the drop-in classes (examples/stereo_kitti.cc): Frame.cc:91-99 + Tracking.cc:857-884 per pair
@pytest.mark.gpu
def test_cpp_sequence_driver_on_kitti_layout(tmp_path):
    """examples/stereo_kitti.cc -- ORBextractor::operator() on two threads, orbfe_host::ComputeStereoMatches on the pyramids in
    HBM, ORBmatcher::SearchByProjection(cur, last) -- over a synthetic sequence in the KITTI directory layout: every frame's
    keypoints, descriptors, mvuRight / mvDepth and tracked assignment == the oracle's (oo_compute_stereo_matches etc.)."""
    from refactored_orb_slam2_amd import synth
    from refactored_orb_slam2_amd._lib import KP_DTYPE, TRACK_POSE_DTYPE, UNPROJECT_CAM_DTYPE
    from tests import oracle_lib as ol
    from tests.test_matcher_gpu import _write_png_gray
    _build()
    seq = tmp_path / "00"
    (seq / "image_0").mkdir(parents=True); (seq / "image_1").mkdir()
    W, H, NF, N = 1241, 376, 2000, 4
    pairs = synth.sequence(W, H, N, seq=21, stereo=True)
    with open(seq / "times.txt", "w") as f:
        for i, (L, R) in enumerate(pairs):
            _write_png_gray(seq / "image_0" / f"{i:06d}.png", L)
            _write_png_gray(seq / "image_1" / f"{i:06d}.png", R)
            f.write(f"{i * 0.1:e}\n")
    oL, oR = ol.OracleExtractor(NF), ol.OracleExtractor(NF)
    sf, isf = oL.scale_factors, oL.inv_scale_factors
    bf, fx, fy, cx, cy = np.float32(386.1448), np.float32(718.856), np.float32(718.856), np.float32(607.1928), np.float32(185.2157)
    cam = np.zeros(1, UNPROJECT_CAM_DTYPE); pose = np.zeros(1, TRACK_POSE_DTYPE)
    eye = np.eye(3, dtype=np.float32).reshape(9)
    cam["Rwc"] = eye; cam["cx"] = cx; cam["cy"] = cy; cam["invfx"] = np.float32(1) / fx; cam["invfy"] = np.float32(1) / fy
    pose["Rcw"] = eye; pose["fx"] = fx; pose["fy"] = fy; pose["cx"] = cx; pose["cy"] = cy; pose["mbf"] = bf
    pose["max_x"] = W; pose["max_y"] = H; pose["th"] = 7.0; pose["scale_factors"][0, :8] = sf
    exp, prev = [], None
    for (L, R) in pairs:
        kL, dL = oL(L); kR, dR = oR(R)
        _, ur, depth = ol.compute_stereo_matches(kL, dL, kR, dR, [oL.level_pixels(l) for l in range(8)], [oR.level_pixels(l) for l in range(8)],
                                                 sf, isf, float(bf), float(bf / fx))
        nm, assigned = 0, np.full(len(kL), -1, np.int32)
        if prev is not None:
            nm, assigned, _ = ol.OracleFrame(kL, dL, sf, 0, W, 0, H, ur).search_by_projection_frame(ol.track_queries(pose, prev), True)
        exp.append((kL, dL, ur, depth, nm, assigned))
        prev = ol.unproject_stereo(cam, kL, dL, depth)
    exe = os.path.join(ROOT, "tests", "cpp", "_build", "stereo_kitti")

    def check_dump(path, expected):
        raw = open(path, "rb").read()
        off = 0
        for i, (kL, dL, ur, depth, nm, assigned) in enumerate(expected):
            n, nmatches = np.frombuffer(raw, np.int32, 2, off); off += 8
            assert n == len(kL), (i, n, len(kL))
            k = np.frombuffer(raw, KP_DTYPE, n, off); off += 28 * n
            d = np.frombuffer(raw, np.uint8, 32 * n, off).reshape(n, 32); off += 32 * n
            u = np.frombuffer(raw, np.float32, n, off); off += 4 * n
            z = np.frombuffer(raw, np.float32, n, off); off += 4 * n
            a = np.frombuffer(raw, np.int32, n, off); off += 4 * n
            np.testing.assert_array_equal(k, kL, err_msg=f"keypoints of frame {i}")
            np.testing.assert_array_equal(d, dL)
            np.testing.assert_array_equal(u, ur, err_msg=f"mvuRight of frame {i}"); np.testing.assert_array_equal(z, depth)
            assert nmatches == nm, (i, nmatches, nm)
            np.testing.assert_array_equal(a, assigned, err_msg=f"tracked assignment of frame {i}")
        assert off == len(raw)

    dump = str(tmp_path / "dump.bin")
    r = subprocess.run([exe, str(seq), "--dump", dump], capture_output=True, text=True, timeout=150)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "median tracking time" in r.stdout and f"Images in the sequence: {N}" in r.stdout
    check_dump(dump, exp)
    assert exp[2][4] > 500 and (exp[1][3] > 0).sum() > 800
    # ---- the batched pipeline from the C++ host (orbfe_pipeline_*, --batch F): chunks of 3 + 1 frames through two buffer sets, the
    #      frame that opens the second chunk tracked against the last frame of the first; then the sequence walked twice from
    #      pre-decoded frames (frame 4 = image 0 again, tracked against image 3's stereo points)
    dumpb = str(tmp_path / "dump_batch.bin")
    rb = subprocess.run([exe, str(seq), "--dump", dumpb, "--batch", "3", "--slots", "2"], capture_output=True, text=True, timeout=150)
    assert rb.returncode == 0, rb.stdout + rb.stderr
    assert "batched pipeline: chunks of 3 pairs, 2 buffer sets" in rb.stdout and "frames/s end to end" in rb.stdout, rb.stdout
    check_dump(dumpb, exp)
    kL0, dL0, ur0, depth0 = exp[0][:4]
    nm4, as4, _ = ol.OracleFrame(kL0, dL0, sf, 0, W, 0, H, ur0).search_by_projection_frame(
        ol.track_queries(pose, ol.unproject_stereo(cam, exp[3][0], exp[3][1], exp[3][3])), True)
    exp2 = exp + [(kL0, dL0, ur0, depth0, nm4, as4)] + exp[1:]
    dumpr = str(tmp_path / "dump_repeat.bin")
    rr = subprocess.run([exe, str(seq), "--dump", dumpr, "--batch", "4", "--preload", "1", "--repeat", "2"], capture_output=True, text=True, timeout=150)
    assert rr.returncode == 0, rr.stdout + rr.stderr
    assert "Images in the sequence: 8" in rr.stdout and "frames decoded before the clock started" in rr.stdout, rr.stdout
    check_dump(dumpr, exp2)
    # frames resident in the pinned slots (--preload 2) / in the slots' device blocks (--preload 3: orbfe_pipeline_submit_resident) after
    # the first chunks: no dump (the results are those of the resident frames); the same keypoint and stereo statistics as the plain run
    stats = re.search(r"keypoints/left image: ([0-9.]+), stereo matches/frame: ([0-9.]+)", rr.stdout)
    for pre, what in (("2", "resident in the pinned slots"), ("3", "resident in HBM")):
        rq = subprocess.run([exe, str(seq), "--batch", "4", "--slots", "2", "--preload", pre, "--repeat", "4"], capture_output=True, text=True, timeout=150)
        assert rq.returncode == 0, rq.stdout + rq.stderr
        assert what in rq.stdout and "Images in the sequence: 16" in rq.stdout, rq.stdout
        sq = re.search(r"keypoints/left image: ([0-9.]+), stereo matches/frame: ([0-9.]+)", rq.stdout)
        assert stats and sq and sq.group(1) == stats.group(1) and sq.group(2) == stats.group(2), (rq.stdout, rr.stdout)
    # the report: every frame counts (the front end was prepared before frame 0), tail latencies, the end-to-end rate
    for key in ("mean tracking time", "p95 tracking time", "p99 tracking time", "max tracking time", "frames/s end to end", "front end prepared in"):
        assert key in r.stdout, key
    # the batched mode's exchange through the C ABI (orbfe_gather_create_all + orbfe_gather_records; one GPU here: world 1), both
    # forms, and the reference's load-then-track loop (no decode pool, no warm-up): the same records
    for extra in (["--gather", "all"], ["--gather", "root", "--decode-threads", "0", "--prepare", "0"],
                  ["--gather", "all", "--batch", "2"], ["--gather", "root", "--batch", "3", "--slots", "1"]):   # the last two: per chunk, from HBM
        gd = str(tmp_path / "gather.bin")
        r2 = subprocess.run([exe, str(seq), "--gather-dump", gd] + extra, capture_output=True, text=True, timeout=150)
        assert r2.returncode == 0, r2.stdout + r2.stderr
        assert "RCCL through the C ABI" in r2.stdout and "0 frame(s) differ" in r2.stdout, r2.stdout
        graw = open(gd, "rb").read()
        goff = 0
        for i, (kL, dL, *_rest) in enumerate(exp):
            n = int(np.frombuffer(graw, np.int32, 1, goff)[0]); goff += 4
            assert n == len(kL)
            np.testing.assert_array_equal(np.frombuffer(graw, KP_DTYPE, n, goff), kL, err_msg=f"gathered keypoints of frame {i}"); goff += 28 * n
            np.testing.assert_array_equal(np.frombuffer(graw, np.uint8, 32 * n, goff).reshape(n, 32), dL); goff += 32 * n
        assert goff == len(graw)
