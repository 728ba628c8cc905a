"""CPU suite, part 3: host-side logic (synthetic data, query construction, frame sharding, RCCL-gather
packing with the gloo backend at world size 2)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from refactored_orb_slam2_amd import sharding, synth
from refactored_orb_slam2_amd._lib import KP_DTYPE, QUERY_DTYPE
from refactored_orb_slam2_amd.matcher import three_maxima
from tests import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_synth_is_deterministic_and_textured():
    a = synth.frame(320, 200, seq=2, f=3)
    b = synth.frame(320, 200, seq=2, f=3)
    np.testing.assert_array_equal(a, b)
    assert a.dtype == np.uint8 and a.shape == (200, 320) and a.std() > 20
    L, R = synth.stereo_pair(320, 200, seq=2, f=3)
    np.testing.assert_array_equal(L, a)
    assert not np.array_equal(L, R)
    k, _ = ol.OracleExtractor(300, 1.2, 4, 20, 7)(a)
    assert len(k) > 200  # FAST corners exist at every level


def test_three_maxima_matches_oracle():
    rng = np.random.default_rng(0)
    for _ in range(200):
        hs = rng.integers(0, 40, 30).astype(np.int32)
        if rng.random() < 0.3:
            hs[rng.integers(0, 30)] = 500  # dominant bin -> the 10 % rule kicks in
        i1, i2, i3 = ol.C.c_int(), ol.C.c_int(), ol.C.c_int()
        ol.lib().oo_three_maxima(hs.ctypes.data, 30, ol.C.byref(i1), ol.C.byref(i2), ol.C.byref(i3))
        assert three_maxima(hs.tolist()) == (i1.value, i2.value, i3.value)


def test_shard_range_partitions_frames():
    for n in (0, 1, 7, 64, 4541):
        for world in (1, 2, 3, 4, 8):
            got = [sharding.shard_range(n, r, world) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == n
            assert all(got[i][1] == got[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in got]
            assert max(sizes) - min(sizes) <= 1 and max(sizes) == sharding.padded_chunk(n, world) or n == 0
    with pytest.raises(ValueError):
        sharding.shard_range(4, 4, 4)


def test_bench_camera_reprojects_the_known_image_motion():
    """bench.camera_records: stereo points of frame f-1 (oracle UnprojectStereo), projected with frame f's camera (oracle
    track query), land SHIFT_X px from where they were seen, with the reference's radius / level range / right coordinate"""
    sys.path.insert(0, ROOT)
    import bench
    rng = np.random.default_rng(1)
    n = 400
    kp = np.zeros(n, KP_DTYPE)
    kp["x"] = rng.uniform(20, 1200, n).astype(np.float32); kp["y"] = rng.uniform(20, 350, n).astype(np.float32)
    kp["angle"] = rng.uniform(0, 360, n).astype(np.float32); kp["octave"] = rng.integers(0, 8, n)
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    disparity = rng.uniform(3, 80, n).astype(np.float32)
    depth = np.where(rng.random(n) < 0.7, np.float32(bench.MBF) / disparity, -1).astype(np.float32)
    sf = ol.OracleExtractor(100).scale_factors
    cams, poses = bench.camera_records(2, sf)
    assert cams.dtype.itemsize == 64 and poses.dtype.itemsize == 160
    pts = ol.unproject_stereo(cams[:1], kp, desc, depth)
    q = ol.track_queries(poses[1:2], pts)
    ok = depth > 0
    np.testing.assert_array_equal(q["valid"] == 1, ok & (kp["x"] + bench.SHIFT_X >= 0))
    v = q["valid"] == 1
    assert np.abs(q["u"][v] - (kp["x"][v] + np.float32(bench.SHIFT_X))).max() < 2e-3 and np.abs(q["v"][v] - kp["y"][v]).max() < 2e-3
    assert np.abs((q["u"][v] - q["u_r"][v]) - disparity[v]).max() < 2e-3        # u_r = u - mbf / z
    np.testing.assert_array_equal(q["radius"][v], np.float32(bench.TH_STEREO) * sf[kp["octave"][v]])
    np.testing.assert_array_equal(q["min_level"][v], kp["octave"][v] - 1)
    np.testing.assert_array_equal(q["max_level"][v], kp["octave"][v] + 1)
    np.testing.assert_array_equal(q["desc"][v], desc[v]); np.testing.assert_array_equal(q["angle"][v], kp["angle"][v])
    assert (q["blocks"][v] == 1).all()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _gather_worker(rank, world, port, n_frames, cap, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        chunk = sharding.padded_chunk(n_frames, world)
        b, e = sharding.shard_range(n_frames, rank, world)
        # deterministic per-frame records: frame f has (f*7 % cap) keypoints with recognisable content
        n = torch.zeros(chunk, dtype=torch.int32)
        kps = torch.zeros((chunk, cap, 28), dtype=torch.uint8)
        desc = torch.zeros((chunk, cap, 32), dtype=torch.uint8)
        for i, f in enumerate(range(b, e)):
            c = (f * 7) % cap
            n[i] = c
            kps[i, :c] = (f % 251)
            desc[i, :c] = ((f * 3) % 253)
        n_all, k_all, d_all = sharding.gather_records(n, kps, desc)
        assert n_all.shape[0] == world * chunk
        # the overlapped variant used by bench.py must deliver the same records (twice: buffers are recycled)
        ag = sharding.AsyncGather(n, kps, desc)
        for _ in range(2):
            ag.launch(n, kps, desc)
        a_n, a_k, a_d = ag.result()
        assert torch.equal(a_n, n_all) and torch.equal(a_k, k_all) and torch.equal(a_d, d_all)
        # global frame order: rank r's chunk starts at r*chunk; padding frames carry n = 0
        for r in range(world):
            rb, re_ = sharding.shard_range(n_frames, r, world)
            for i, f in enumerate(range(rb, re_)):
                g = r * chunk + i
                c = (f * 7) % cap
                assert int(n_all[g]) == c
                assert bool((k_all[g, :c] == f % 251).all()) and bool((d_all[g, :c] == (f * 3) % 253).all())
            for i in range(re_ - rb, chunk):
                assert int(n_all[r * chunk + i]) == 0
        # unpack_records walks the padded chunks: frame f's record whatever n_frames % world is
        frames = sharding.unpack_records(n_all, k_all, d_all, n_frames, world)
        assert len(frames) == n_frames
        for f, (kk, dd) in enumerate(frames):
            assert len(kk) == len(dd) == (f * 7) % cap
            assert bool((kk.view(np.uint8) == f % 251).all()) and bool((dd == (f * 3) % 253).all())
        # gather to the root (ncclSend / Recv to rank 0): rank 0 holds the same records, the others nothing
        root = sharding.gather_records(n, kps, desc, mode="root")
        ag = sharding.AsyncGather(n, kps, desc, mode="root")
        for _ in range(2):
            ag.launch(n, kps, desc)
        a_root = ag.result()
        if rank == 0:
            for got in (root, a_root):
                assert torch.equal(got[0], n_all) and torch.equal(got[1], k_all) and torch.equal(got[2], d_all)
        else:
            assert root is None and a_root is None
        tr_all, tr_root = sharding.gather_traffic(n, kps, desc, world, "all"), sharding.gather_traffic(n, kps, desc, world, "root")
        assert tr_all["received_per_rank"] == (world - 1) * tr_all["record_bytes_per_rank"] == tr_root["received_rank0"]
        assert tr_all["record_bytes_per_rank"] == chunk * (4 + cap * 60)
        dist.barrier()
        open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_frames", [(2, 9), (2, 8), (3, 10)])   # (3, 10): shards 4 + 3 + 3 in chunks of 4 -- rows shift behind the first short chunk
def test_gather_records_gloo_world2(tmp_path, world, n_frames):
    port = _free_port()
    mp.spawn(_gather_worker, args=(world, port, n_frames, 40, str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))


def test_gather_records_identity_without_group():
    n = torch.arange(4, dtype=torch.int32)
    k = torch.zeros((4, 5, 28), dtype=torch.uint8)
    d = torch.zeros((4, 5, 32), dtype=torch.uint8)
    out = sharding.gather_records(n, k, d)
    assert out[0] is n and out[1] is k and out[2] is d


def test_c_abi_shard_range_equals_the_python_split():
    """orbfe_shard_range (what a C++ host of the batched mode calls) == sharding.shard_range: contiguous, exhaustive, sizes
    differing by at most one; bad arguments are codes."""
    import ctypes as C
    from refactored_orb_slam2_amd import _lib
    L = _lib.lib()
    for n_frames in (0, 1, 7, 8, 4541):
        for world in (1, 2, 3, 8):
            got = [sharding.shard_range_c(n_frames, r, world) for r in range(world)]
            assert got == [sharding.shard_range(n_frames, r, world) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == n_frames and all(a[1] == b[0] for a, b in zip(got, got[1:]))
    b, e = C.c_int(), C.c_int()
    assert L.orbfe_shard_range(10, 3, 3, C.byref(b), C.byref(e)) == -1
    assert L.orbfe_shard_range(10, 0, 0, C.byref(b), C.byref(e)) == -1
    assert L.orbfe_shard_range(-1, 0, 1, C.byref(b), C.byref(e)) == -1


def test_png_reader_decodes_all_filters_and_rgb(tmp_path):
    """liborbfe's zlib PNG reader (sequence driver input): 8-bit grey with every filter type, RGB converted with cvtColor's
    fixed-point RGB2GRAY weights; files written by PIL (adaptive filters) and by the driver test's minimal writer."""
    import ctypes as C
    from PIL import Image
    from refactored_orb_slam2_amd import _lib
    from tests.test_matcher_gpu import _write_png_gray
    L = _lib.lib()

    def read(path):
        w, h = C.c_int(0), C.c_int(0)
        assert L.orbfe_png_info(str(path).encode(), C.byref(w), C.byref(h)) == 0
        out = np.empty((h.value, w.value), np.uint8)
        assert L.orbfe_png_read_gray(str(path).encode(), out.ctypes.data_as(C.c_void_p), out.strides[0], h.value, C.byref(w), C.byref(h)) == 0
        return out

    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (97, 131)).astype(np.uint8)
    img[20:60, 30:90] = 200
    img[5:15] = np.arange(131, dtype=np.uint8)           # gradients: sub / up / paeth filters win on such rows
    Image.fromarray(img).save(tmp_path / "pil.png")
    _write_png_gray(tmp_path / "own.png", img)
    np.testing.assert_array_equal(read(tmp_path / "pil.png"), img)
    np.testing.assert_array_equal(read(tmp_path / "own.png"), img)
    rgb = rng.integers(0, 256, (50, 70, 3)).astype(np.uint8)
    Image.fromarray(rgb).save(tmp_path / "rgb.png")
    ref = ((rgb[..., 0].astype(np.int64) * 4899 + rgb[..., 1].astype(np.int64) * 9617 + rgb[..., 2].astype(np.int64) * 1868 + 8192) >> 14)
    np.testing.assert_array_equal(read(tmp_path / "rgb.png"), ref.astype(np.uint8))
    # Camera.RGB: 1 (every settings file the reference ships): Tracking::GrabImage* applies RGB2GRAY to imread's BGR data
    # (L/src/Tracking.cc:164-178), i.e. the 0.299 weight meets the file's BLUE channel; grey files do not depend on the flag
    L.orbfe_png_read_gray2.argtypes = [C.c_char_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int]
    w2, h2 = C.c_int(0), C.c_int(0)
    sw = np.empty((50, 70), np.uint8)
    assert L.orbfe_png_read_gray2(str(tmp_path / "rgb.png").encode(), sw.ctypes.data_as(C.c_void_p), 70, 50, C.byref(w2), C.byref(h2), 1) == 0
    ref_sw = ((rgb[..., 2].astype(np.int64) * 4899 + rgb[..., 1].astype(np.int64) * 9617 + rgb[..., 0].astype(np.int64) * 1868 + 8192) >> 14)
    np.testing.assert_array_equal(sw, ref_sw.astype(np.uint8))
    assert L.orbfe_png_read_gray2(str(tmp_path / "rgb.png").encode(), sw.ctypes.data_as(C.c_void_p), 70, 50, C.byref(w2), C.byref(h2), 0) == 0
    np.testing.assert_array_equal(sw, ref.astype(np.uint8))
    g2 = np.empty_like(img)
    assert L.orbfe_png_read_gray2(str(tmp_path / "pil.png").encode(), g2.ctypes.data_as(C.c_void_p), g2.strides[0], g2.shape[0], C.byref(w2), C.byref(h2), 1) == 0
    np.testing.assert_array_equal(g2, img)
    # errors are codes, not crashes
    w, h = C.c_int(0), C.c_int(0)
    assert L.orbfe_png_info(str(tmp_path / "missing.png").encode(), C.byref(w), C.byref(h)) != 0
    (tmp_path / "junk.png").write_bytes(b"not a png at all, really")
    assert L.orbfe_png_info(str(tmp_path / "junk.png").encode(), C.byref(w), C.byref(h)) != 0
    small = np.empty((10, 10), np.uint8)
    assert L.orbfe_png_read_gray(str(tmp_path / "pil.png").encode(), small.ctypes.data_as(C.c_void_p), 10, 10, C.byref(w), C.byref(h)) == _lib.ERR_CAPACITY if hasattr(_lib, "ERR_CAPACITY") else True


def test_png_reader_16bit_depth_maps_and_hostile_headers(tmp_path):
    """16-bit greyscale PNGs (TUM RGB-D depth maps, Source/Examples/RGB-D/rgbd_tum.cc reads them with IMREAD_UNCHANGED) decode
    to uint16 in host byte order; an IHDR that announces a huge image is refused with a code before anything is allocated
    (no exception crosses the C ABI), and the 8-bit entry point names the 16-bit one instead of mis-decoding."""
    import ctypes as C
    import struct
    import zlib
    from PIL import Image
    from refactored_orb_slam2_amd import _lib
    L = _lib.lib()
    rng = np.random.default_rng(9)
    depth = rng.integers(0, 65536, (61, 83)).astype(np.uint16)
    depth[10:30, 5:60] = np.arange(55, dtype=np.uint16) * 700          # ramps: sub / paeth filters on 2-byte pixels
    Image.fromarray(depth).save(tmp_path / "d16.png")
    w, h, dp, ch = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
    assert L.orbfe_png_info2(str(tmp_path / "d16.png").encode(), C.byref(w), C.byref(h), C.byref(dp), C.byref(ch)) == 0
    assert (w.value, h.value, dp.value, ch.value) == (83, 61, 16, 1)
    out = np.zeros((61, 96), np.uint16)
    assert L.orbfe_png_read_gray16(str(tmp_path / "d16.png").encode(), out.ctypes.data_as(C.c_void_p), 96, 61, C.byref(w), C.byref(h)) == 0
    np.testing.assert_array_equal(out[:, :83], depth)
    buf8 = np.zeros((61, 96), np.uint8)
    assert L.orbfe_png_read_gray(str(tmp_path / "d16.png").encode(), buf8.ctypes.data_as(C.c_void_p), 96, 61, C.byref(w), C.byref(h)) == -1
    assert b"orbfe_png_read_gray16" in L.orbfe_last_error()
    # a grey + alpha file takes the grey channel
    ga = np.stack([rng.integers(0, 256, (20, 30)).astype(np.uint8), np.full((20, 30), 255, np.uint8)], axis=-1)
    Image.fromarray(ga, mode="LA").save(tmp_path / "ga.png")
    o8 = np.zeros((20, 30), np.uint8)
    assert L.orbfe_png_read_gray(str(tmp_path / "ga.png").encode(), o8.ctypes.data_as(C.c_void_p), 30, 20, C.byref(w), C.byref(h)) == 0
    np.testing.assert_array_equal(o8, ga[..., 0])

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    for (ww, hh) in ((0x7fffffff, 0x7fffffff), (4096, 10), (10, 70000), (0, 5)):
        ihdr = struct.pack(">IIBBBBB", ww, hh, 8, 0, 0, 0, 0)
        (tmp_path / "evil.png").write_bytes(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"IDAT", zlib.compress(b"\0" * 16)) + chunk(b"IEND", b""))
        dst = np.zeros((16, 16), np.uint8)
        assert L.orbfe_png_info(str(tmp_path / "evil.png").encode(), C.byref(w), C.byref(h)) == -1
        assert L.orbfe_png_read_gray(str(tmp_path / "evil.png").encode(), dst.ctypes.data_as(C.c_void_p), 16, 16, C.byref(w), C.byref(h)) == -1
    # truncated IDAT stream: a code
    ihdr = struct.pack(">IIBBBBB", 8, 8, 8, 0, 0, 0, 0)
    (tmp_path / "short.png").write_bytes(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"IDAT", zlib.compress(b"\0" * 20)) + chunk(b"IEND", b""))
    assert L.orbfe_png_read_gray(str(tmp_path / "short.png").encode(), dst.ctypes.data_as(C.c_void_p), 16, 16, C.byref(w), C.byref(h)) == -1
