#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: frames/s of the ORB front end (extract + match).

One "step" = one batch of F synthetic frames resident in HBM, pushed through the whole hot path on one GPU.  Workloads
(`--config`, the headline is the default):
  kitti_stereo  1241x376, 2000 features, 8 levels (BASELINE.json's metric):
                ORBextractor left + right -> Frame::ComputeStereoMatches -> Frame::UnprojectStereo of every stereo point
                -> its projection into the next frame (ORBmatcher.cc:1270-1308) -> SearchByProjection(cur, last)
  euroc_stereo  752x480, 2 x 1200 features: the same pipeline with the EuRoC camera
  kitti_mono    1241x376, 1000 features, one extractor: extraction + SearchByProjection(cur, last, th = 15) against the
                previous frame's keypoints un-projected at a constant depth
  tum_bow       640x480, 1000 features: extraction + the Hamming brute force of SearchByBoW between consecutive frames
                (descriptors grouped by a synthetic 100-bucket node id standing in for the missing vocabulary)
`--gpus N` shards independent frames over N ranks (one process per GPU, weak scaling: F frames per rank per step) and
gathers the per-frame keypoint/descriptor records with one RCCL all_gather per step -- the only exchange the path has.
Started without a launcher (`WORLD_SIZE` unset) it spawns the N ranks itself before touching the GPU; under
`python -m torch.distributed.run` it uses the ranks it is given.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel, HIP events on the
launching stream), at N=1 `cpu_baseline` (the CPU oracle on a bounded sample) and `e2e_frames_per_s` (host images in,
host keypoints / descriptors / matches out, double-buffered over PCIe) -- `value` itself is HBM-resident.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: 8 TB/s spec
NLEVELS = 8
# camera rows: Source/Examples/Stereo/KITTI00-02.yaml, EuRoC.yaml; Monocular/KITTI00-02.yaml; RGB-D/TUM1.yaml
CONFIGS = {
    "kitti_stereo": dict(w=1241, h=376, nfeat=2000, stereo=True, fx=718.856, fy=718.856, cx=607.1928, cy=185.2157,
                         bf=386.1448, th=7.0, match="projection",
                         label="kitti_stereo_1241x376_2000feat_8lvl: 2x ORBextractor + ComputeStereoMatches + "
                               "SearchByProjection(cur,last)"),
    "euroc_stereo": dict(w=752, h=480, nfeat=1200, stereo=True, fx=435.2046959714599, fy=435.2046959714599,
                         cx=367.4517211914062, cy=252.2008514404297, bf=47.90639384423901, th=7.0, match="projection",
                         label="euroc_stereo_752x480_2x1200feat_8lvl: 2x ORBextractor + ComputeStereoMatches + "
                               "SearchByProjection(cur,last)"),
    "kitti_mono": dict(w=1241, h=376, nfeat=1000, stereo=False, fx=718.856, fy=718.856, cx=607.1928, cy=185.2157,
                       bf=0.0, th=15.0, match="projection",
                       label="kitti_mono_1241x376_1000feat_8lvl: ORBextractor + SearchByProjection(cur,last,th=15)"),
    "tum_bow": dict(w=640, h=480, nfeat=1000, stereo=False, fx=517.306408, fy=516.469215, cx=318.643040, cy=255.313989,
                    bf=40.0, th=15.0, match="bow",
                    label="tum_640x480_1000feat_8lvl: ORBextractor + SearchByBoW Hamming brute force (100 node groups) "
                          "between consecutive frames"),
}
W, H, NFEAT = CONFIGS["kitti_stereo"]["w"], CONFIGS["kitti_stereo"]["h"], CONFIGS["kitti_stereo"]["nfeat"]   # the headline workload
MBF, FX, TH_STEREO = CONFIGS["kitti_stereo"]["bf"], CONFIGS["kitti_stereo"]["fx"], CONFIGS["kitti_stereo"]["th"]   # th = 7: Tracking.cc:793-798
SHIFT_X = -2.0                             # synth.sequence: the image content moves 2 px per frame
MONO_DEPTH = 10.0                          # kitti_mono: depth at which the previous frame's keypoints are un-projected


def camera_records(n, sf, cfg=None):
    """Synthetic camera for the motion-model search: identity pose, and a principal point that moves with the image
    content (SHIFT_X px per frame), so that re-projecting the last frame's points into the current frame
    (UnprojectStereo -> Rcw*x3Dw+tcw -> pinhole) lands on the known image motion for every depth."""
    from refactored_orb_slam2_amd._lib import TRACK_POSE_DTYPE, UNPROJECT_CAM_DTYPE
    cfg = cfg or CONFIGS["kitti_stereo"]
    cams = np.zeros(n, UNPROJECT_CAM_DTYPE); poses = np.zeros(n, TRACK_POSE_DTYPE)
    eye = np.eye(3, dtype=np.float32).reshape(9)
    cams["Rwc"] = eye; cams["cx"] = cfg["cx"]; cams["cy"] = cfg["cy"]
    cams["invfx"] = np.float32(1) / np.float32(cfg["fx"]); cams["invfy"] = np.float32(1) / np.float32(cfg["fy"])
    poses["Rcw"] = eye; poses["fx"] = cfg["fx"]; poses["fy"] = cfg["fy"]
    poses["cx"] = np.float32(cfg["cx"]) + np.float32(SHIFT_X); poses["cy"] = cfg["cy"]
    poses["mbf"] = cfg["bf"]; poses["max_x"] = cfg["w"]; poses["max_y"] = cfg["h"]; poses["th"] = cfg["th"]
    poses["scale_factors"][:, :len(sf)] = np.asarray(sf, np.float32)
    return cams, poses


def node_ids(desc):
    """Stand-in for the DBoW2 node id of a descriptor (the vocabulary file is missing from the reference checkout):
    bucket = (desc[0] + 256 * desc[1]) % 100.  Works on numpy arrays and torch tensors alike."""
    return (desc[..., 0].astype("int32") + 256 * desc[..., 1].astype("int32")) % 100 if isinstance(desc, np.ndarray) \
        else (desc[..., 0].to(dtype=__import__("torch").int32) + 256 * desc[..., 1].to(dtype=__import__("torch").int32)) % 100


def _cpu_frames(cfg, frames, seq):
    """One worker's share of the CPU baseline: the whole per-frame path of `cfg` on the oracle.  Returns seconds."""
    from refactored_orb_slam2_amd import synth
    from tests import oracle_lib as ol
    W, H, NF = cfg["w"], cfg["h"], cfg["nfeat"]
    data = synth.sequence(W, H, frames, seq=seq, stereo=cfg["stereo"])
    oL = ol.OracleExtractor(NF, 1.2, NLEVELS, 20, 7)
    oR = ol.OracleExtractor(NF, 1.2, NLEVELS, 20, 7) if cfg["stereo"] else None
    sf, isf = oL.scale_factors, oL.inv_scale_factors
    cams, poses = camera_records(1, sf, cfg)
    prev = None
    pool = None
    if cfg["stereo"]:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(1)   # right extractor on a second thread, as Frame::Frame does (Frame.cc:87-90)
    t0 = time.perf_counter()
    for item in data:
        if cfg["stereo"]:
            L, R = item
            fut = pool.submit(oR, R)   # ctypes releases the GIL inside the C oracle
            kL, dL = oL(L)
            kR, dR = fut.result()
            planesL = [oL.level_pixels(l) for l in range(NLEVELS)]
            planesR = [oR.level_pixels(l) for l in range(NLEVELS)]
            _, ur, depth = ol.compute_stereo_matches(kL, dL, kR, dR, planesL, planesR, sf, isf, cfg["bf"], cfg["bf"] / cfg["fx"])
        else:
            kL, dL = oL(item)
            ur, depth = None, np.full(len(kL), MONO_DEPTH, np.float32)
        if cfg["match"] == "projection":
            if prev is not None:
                q = ol.track_queries(poses[:1], prev)     # projection of the last frame's map points (ORBmatcher.cc:1270-1308)
                ol.OracleFrame(kL, dL, sf, 0, W, 0, H, ur).search_by_projection_frame(q, True)
            prev = ol.unproject_stereo(cams[:1], kL, dL, depth)   # Frame::UnprojectStereo for every point with depth
        else:
            if prev is not None:
                ol.hamming_bf(prev[1], dL, node_ids(prev[1]), node_ids(dL))
            prev = (kL, dL)
    dt = time.perf_counter() - t0
    if pool:
        pool.shutdown()
    return dt


def _cpu_worker(args):
    name, frames, seq = args
    return _cpu_frames(CONFIGS[name], frames, seq)


def usable_cpus() -> int:
    """CPUs this process may actually use: the smaller of the online count, the affinity mask and the cgroup CPU quota
    (a GPU box hands a job a share of its host, e.g. 16 of 256 hardware threads)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1") and period > 0:
                n = min(n, max(1, int(float(quota) / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(n, 1)


def cpu_baseline(name: str, sample_frames: int):
    """The CPU oracle (scalar C restatement of the reference path) on a bounded sample, with the reference's own threading
    (two extractor threads per stereo frame), plus a frame-parallel run on all host cores for context."""
    import multiprocessing as mp
    import platform
    cfg = CONFIGS[name]
    dt = _cpu_frames(cfg, sample_frames, 0)
    threads = 2 if cfg["stereo"] else 1
    nproc = os.cpu_count() or 1
    usable = usable_cpus()
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    flags = ""
    try:
        for line in open(os.path.join(ROOT, "oracle", "Makefile")):
            if line.startswith("CFLAGS"):
                flags = line.split("=", 1)[1].strip()
    except OSError:
        pass
    out = {"value": round(sample_frames / dt, 3), "unit": "frames/s", "cores": threads, "kind": "port",
           "sample": f"{sample_frames} synthetic frames of {cfg['label'].split(':')[0]} through oracle/orb_oracle.c "
                     f"({'left || right extraction on two threads as in Frame.cc:87-90, the rest on one' if cfg['stereo'] else 'one thread'}), {dt:.1f} s",
           "nproc": nproc, "usable_cpus": usable, "cpu_model": model or platform.processor(), "compiler": "gcc " + flags,
           "note": "scalar restatement; OpenCV's SIMD FAST / resize / GaussianBlur would make the real reference faster"}
    # all cores, frame-parallel: every worker process runs whole frames (for stereo with its two extractor threads)
    workers = max(1, usable // threads)
    per = max(2, min(sample_frames, 24))
    try:
        ctx = mp.get_context("fork")
        t0 = time.perf_counter()
        with ctx.Pool(workers) as pool:
            pool.map(_cpu_worker, [(name, per, 100 + i) for i in range(workers)])
        wall = time.perf_counter() - t0
        out["all_cores"] = {"value": round(workers * per / wall, 3), "unit": "frames/s", "cores": workers * threads,
                            "sample": f"{workers} worker processes x {per} frames, frame-parallel, {wall:.1f} s wall (incl. input synthesis)"}
    except Exception as exc:   # the headline row above stands on its own
        out["all_cores"] = {"error": str(exc)}
    return out


def _write_png_gray(path, img, level=6):
    """8-bit greyscale PNG (filter 0, zlib level 6 -- what libpng writes by default and KITTI ships): input for the C++ driver."""
    import struct, zlib
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    h, w = img.shape
    rows = np.concatenate([np.zeros((h, 1), np.uint8), img], axis=1).tobytes()
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(rows, level)) +
                chunk(b"IEND", b""))


def per_frame_latency(cfg, n_frames: int):
    """Latency of ONE stereo pair through the calling pattern the reference uses (Frame.cc:91-99, Tracking.cc:857-884): the
    C++ drop-in classes -- ORBextractor::operator() for the two eyes on two std::threads, Frame::ComputeStereoMatches on the
    pyramids in HBM, ORBmatcher::SearchByProjection(cur, last) -- driven by examples/stereo_kitti.cc over a synthetic sequence
    written as PNGs (decoding is outside the driver's per-frame clock, as imread is outside the reference's).  Separate process,
    outside the timed region of the batched metric."""
    import re, tempfile
    from refactored_orb_slam2_amd import synth
    exe = os.path.join(ROOT, "tests", "cpp", "_build", "stereo_kitti")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp"), "_build/stereo_kitti"], capture_output=True, text=True)   # (a no-op when up to date)
    if r.returncode != 0 and not os.path.exists(exe):
        return {"error": "examples/stereo_kitti.cc not built: " + r.stderr[-300:]}
    with tempfile.TemporaryDirectory() as tmp:
        seq = os.path.join(tmp, "00")
        os.makedirs(os.path.join(seq, "image_0")); os.makedirs(os.path.join(seq, "image_1"))
        pairs = synth.sequence(cfg["w"], cfg["h"], n_frames, seq=7, stereo=True)
        with open(os.path.join(seq, "times.txt"), "w") as f:
            for i, (L, R) in enumerate(pairs):
                _write_png_gray(os.path.join(seq, "image_0", f"{i:06d}.png"), L)
                _write_png_gray(os.path.join(seq, "image_1", f"{i:06d}.png"), R)
                f.write(f"{i * 0.1:e}\n")
        base = [exe, seq, "--features", str(cfg["nfeat"]), "--bf", str(cfg["bf"]), "--fx", str(cfg["fx"]), "--fy", str(cfg["fy"]),
                "--cx", str(cfg["cx"]), "--cy", str(cfg["cy"]), "--th", str(cfg["th"])]
        usable = usable_cpus()   # affinity AND the cgroup quota: a GPU box shows 256 hardware threads and grants 16
        threads = max(1, min(16, usable - 3))   # the tracking thread and the two extractor threads keep a core each
        bthreads = max(1, min(16, usable - 1))  # the batched driver: one submitting thread beside the decode pool
        r = subprocess.run(base + ["--decode-threads", str(threads), "--prefetch", "32"], capture_output=True, text=True, timeout=600)
        r0 = subprocess.run(base + ["--decode-threads", "0"], capture_output=True, text=True, timeout=600)   # load, then track
        # the batched pipeline from the same C++ host (orbfe_pipeline_*): chunks of 256 pairs, PNGs decoded inside the clock
        # (decode-bound), and the sequence walked eight times from frames decoded before the clock (the pipeline itself, PCIe included)
        rb = subprocess.run(base + ["--decode-threads", str(bthreads), "--batch", "256", "--repeat", "4"], capture_output=True, text=True, timeout=600)
        rp = subprocess.run(base + ["--decode-threads", str(bthreads), "--batch", "256", "--preload", "1", "--repeat", "16"], capture_output=True, text=True, timeout=600)
        rq = subprocess.run(base + ["--decode-threads", str(threads), "--batch", "256", "--preload", "2", "--repeat", "24"], capture_output=True, text=True, timeout=600)
        rd = subprocess.run(base + ["--decode-threads", str(threads), "--batch", "256", "--preload", "3", "--repeat", "288"], capture_output=True, text=True, timeout=600)
        # the same with orbfe_pipeline_config.output_mask: only the tracked assignments (+ the counts), only the counts copied to the host
        rdm = subprocess.run(base + ["--decode-threads", str(threads), "--batch", "256", "--preload", "3", "--repeat", "288", "--outputs", "matches"], capture_output=True, text=True, timeout=600)
        rdc = subprocess.run(base + ["--decode-threads", str(threads), "--batch", "256", "--preload", "3", "--repeat", "288", "--outputs", "counts"], capture_output=True, text=True, timeout=600)
    if r.returncode != 0:
        return {"error": f"stereo_kitti exited with {r.returncode}: " + (r.stderr or r.stdout)[-300:]}
    med = re.search(r"median tracking time: ([0-9.eE+-]+)", r.stdout)
    mean = re.search(r"mean tracking time: ([0-9.eE+-]+)", r.stdout)
    tail = {k: re.search(k + r" tracking time: ([0-9.eE+-]+)", r.stdout) for k in ("p95", "p99", "max")}
    stats = re.search(r"keypoints/left image: ([0-9.]+), stereo matches/frame: ([0-9.]+), tracked/frame: ([0-9.]+)", r.stdout)
    ph = re.search(r"two threads\) ([0-9.]+), ComputeStereoMatches ([0-9.]+), SearchByProjection\(cur,last\) ([0-9.]+)", r.stdout)
    seq_re = r"sequence: (\d+) frames in ([0-9.]+) s = ([0-9.]+) frames/s end to end \(decode threads (\d+), prefetch (\d+); decode ([0-9.]+) s of CPU time = ([0-9.]+) ms per pair; tracking thread waited ([0-9.]+) s"
    sq, sq0 = re.search(seq_re, r.stdout), (re.search(seq_re, r0.stdout) if r0.returncode == 0 else None)
    sqb, sqp = (re.search(seq_re, rb.stdout) if rb.returncode == 0 else None), (re.search(seq_re, rp.stdout) if rp.returncode == 0 else None)
    sqq = re.search(seq_re, rq.stdout) if rq.returncode == 0 else None
    sqd = re.search(seq_re, rd.stdout) if rd.returncode == 0 else None
    sqdm = re.search(seq_re, rdm.stdout) if rdm.returncode == 0 else None
    sqdc = re.search(seq_re, rdc.stdout) if rdc.returncode == 0 else None
    prep = re.search(r"front end prepared in ([0-9.]+) ms", r.stdout)
    sequence = None
    if sq:
        sequence = {"frames_per_s": float(sq.group(3)), "decode_threads": int(sq.group(4)), "prefetch_pairs": int(sq.group(5)),
                    "decode_ms_per_pair_cpu": float(sq.group(7)), "tracking_thread_waited_s": float(sq.group(8)),
                    "frames_per_s_load_then_track": float(sq0.group(3)) if sq0 else None,
                    "frames_per_s_batched": float(sqb.group(3)) if sqb else None,
                    "batched": ({"frames": int(sqb.group(1)), "decode_ms_per_pair_cpu": float(sqb.group(7)), "decode_threads": int(sqb.group(4)),
                                 "pipeline_waited_for_images_s": float(sqb.group(8)),
                                 "path": "examples/stereo_kitti.cc --batch 256: decode pool -> pinned pitched slots -> orbfe_pipeline_submit / wait "
                                         "(H2D, 2x extract on two streams, stereo, unproject, track queries, projection search on a third, D2H), three slots with handles of their own"} if sqb else
                                {"error": (rb.stderr or rb.stdout)[-300:]}),
                    "frames_per_s_batched_predecoded": float(sqp.group(3)) if sqp else None,
                    "batched_predecoded_frames": int(sqp.group(1)) if sqp else None,
                    "batched_predecoded_note": "frames decoded before the clock; per frame one host copy (0.96 MB) from pageable memory into the pinned slot by the pool",
                    "frames_per_s_batched_pinned_resident": float(sqq.group(3)) if sqq else None,
                    "batched_pinned_resident_note": "--preload 2: the slots keep their frames after the first chunks, no host work per frame: the C++ pipeline's own rate with PCIe both ways (what e2e_frames_per_s measures from Python)",
                    "frames_per_s_batched_device_resident": float(sqd.group(3)) if sqd else None,
                    "batched_device_resident_note": "--preload 3 --repeat 288 (864 chunks; the first three are filled by the host pool and uploaded inside the clock: ~50 ms): after the first chunks the frames stay in the slots' DEVICE input blocks (orbfe_pipeline_submit_resident): the C++ "
                                                    "pipeline handle at its kernels' rate, every result block still copied to the host (37 MB per chunk) -- what a device-side producer of frames gets",
                    "frames_per_s_batched_device_resident_matches": float(sqdm.group(3)) if sqdm else None,
                    "frames_per_s_batched_device_resident_counts": float(sqdc.group(3)) if sqdc else None,
                    "batched_device_resident_outputs_note": "--outputs matches / counts (orbfe_pipeline_config.output_mask): only the tracked assignments + counts (2.3 MB per "
                                                            "chunk) / only the per-frame counts go to the host; everything stays readable in HBM (orbfe_pipeline_device_records)",
                    "input": "synthetic KITTI-layout sequence written as 8-bit grey PNGs (zlib level 6, filter 0), decoded by orbfe_png_read_gray"}
    return {"median_ms": round(float(med.group(1)) * 1e3, 4), "mean_ms": round(float(mean.group(1)) * 1e3, 4),
            **{k + "_ms": (round(float(m.group(1)) * 1e3, 4) if m else None) for k, m in tail.items()},
            "prepare_ms": float(prep.group(1)) if prep else None, "sequence": sequence, "frames": n_frames,
            "keypoints_per_left_image": float(stats.group(1)), "stereo_matches_per_frame": float(stats.group(2)),
            "tracked_per_frame": float(stats.group(3)),
            "median_ms_by_phase": ({"extract_x2": float(ph.group(1)), "stereo": float(ph.group(2)), "search_by_projection": float(ph.group(3))} if ph else None),
            "path": "examples/stereo_kitti.cc: C++ ORBextractor x2 on two threads + orbfe_host::ComputeStereoMatches + "
                    "ORBmatcher::SearchByProjection(cur,last), host images in, host keypoints / matches out, one pair at a time"}


def cabi_driver(rig, steps: int, output_mask: int, slots: int = 3):
    """The step driven through the C ABI's pipeline handle (orbfe_pipeline_*: what a C / C++ host calls; no torch in the loop): the rig's
    frames uploaded once into the slots' device blocks, then `steps` chunks by orbfe_pipeline_submit_resident in a ring of `slots`
    slots.  Returns the rate and a self-check: a second handle that copies every block out must reproduce, byte for byte, what the
    torch-driven step left in the rig's buffers (which tests/test_bench_layout_gpu.py compares with the oracle)."""
    import torch
    from refactored_orb_slam2_amd._lib import KP_DTYPE
    from refactored_orb_slam2_amd.pipeline import StereoPipeline
    cfg, F, W, H = rig.cfg, rig.F, rig.W, rig.H

    def make(mask):
        p = StereoPipeline(W, H, F, cfg["fx"], cfg["fy"], cfg["cx"], cfg["cy"], cfg["bf"], cfg["th"], n_features=rig.NFEAT, slots=slots,
                           output_mask=mask)
        for s in range(slots):
            p.poses(s)["cx"] = np.float32(cfg["cx"]) + np.float32(SHIFT_X)
            p.left(s)[:] = rig.hL.numpy(); p.right(s)[:] = rig.hR.numpy()
        for s in range(slots):
            p.submit(s, F, has_predecessor=s > 0)
        for s in range(slots):
            p.wait(s)
        return p

    torch.cuda.synchronize()
    with make(output_mask) as p:
        for k in range(slots):   # warm: every slot once from its resident images
            p.wait(k); p.submit_resident(k, F, has_predecessor=True)
        for s in range(slots):
            p.wait(s)
        t0 = time.perf_counter()
        for k in range(steps):
            s = k % slots
            p.wait(s)
            p.submit_resident(s, F, has_predecessor=True)
        for s in range(slots):
            p.wait(s)
        dt = time.perf_counter() - t0
        n_tr = int(np.asarray(p.output(0)["n_tracked"]).sum())
    # ---- self-check against the torch-driven step (one chunk per slot through a handle that copies everything out)
    rig.step(); rig.barrier()
    B = rig.B0
    ref = {k: getattr(B, k).cpu().numpy() for k in ("nl", "nr", "n_stereo", "n_track", "kl", "dl", "ur", "depth", "assigned")}
    ok = True
    with make(0) as p:
        for s in range(slots):
            p.wait(s); p.submit_resident(s, F, has_predecessor=True)
        for s in range(slots):
            p.wait(s)
            o = p.output(s)
            cap = min(o["kps_left"].shape[1], ref["kl"].shape[1])
            m = np.arange(cap)[None, :] < ref["nl"][:, None]
            rk = ref["kl"].reshape(F, -1).view(KP_DTYPE).reshape(F, -1)
            ok = ok and all(np.array_equal(np.asarray(o[a]), ref[b]) for a, b in (("n_left", "nl"), ("n_right", "nr"), ("n_stereo", "n_stereo"), ("n_tracked", "n_track")))
            ok = ok and o["kps_left"][:, :cap][m].tobytes() == rk[:, :cap][m].tobytes() and o["desc_left"][:, :cap][m].tobytes() == ref["dl"][:, :cap][m].tobytes()
            ok = ok and o["u_right"][:, :cap][m].tobytes() == ref["ur"][:, :cap][m].tobytes() and o["depth"][:, :cap][m].tobytes() == ref["depth"][:, :cap][m].tobytes()
            ok = ok and o["assigned"][:, :cap][m].tobytes() == ref["assigned"][:, :cap][m].tobytes()
    return {"frames_per_s": round(F * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps, "slots": slots,
            "output_mask": output_mask, "tracked_per_frame": round(n_tr / F, 1), "equals_torch_driven_step": bool(ok),
            "path": "orbfe_pipeline_submit_resident / orbfe_pipeline_wait in a ring of slots (ctypes; frames resident in the slots' device blocks); "
                    "output_mask 8 = tracked assignments + counts copied to the host per chunk, 16 = counts only, 0 = every block (37 MB)"}


def spawn_ranks(n: int) -> int:
    """Start n ranks of this script (one process per GPU) before anything touched the GPU in this process."""
    import torch
    have = torch.cuda.device_count()   # does not initialise the GPU on this image
    share = os.environ.get("ORBFE_BENCH_SHARE_DEVICE") == "1"
    if have < n and not share:
        print(f"bench.py: --gpus {n} but only {have} device(s) visible (ORBFE_BENCH_SHARE_DEVICE=1 rehearses the ranks on "
              f"fewer devices)", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    return rc


class StepRig:
    """Everything one rank's step needs -- synthetic frames resident in HBM, `n_sets` sets of {two extractors, matcher, buffers}, the
    HIP streams -- and the step itself.  `bench.py` times `step()`; `tests/test_bench_layout_gpu.py` runs the SAME object at the
    same batch size and stream layout and compares what it leaves in the buffers with the oracle (round-5 review: the layout the
    bench times had no parity test).  Layout of a step (lr_streams = 2, n_sets > 1): left | right extractor on two streams as the
    reference runs them on two threads (L/src/Frame.cc:87-90), the matching half (Frame::ComputeStereoMatches L/src/Frame.cc:477-646,
    UnprojectStereo, the projection of L/src/ORBmatcher.cc:1270-1308, SearchByProjection(cur, last) :1247-1383) on a third; the sets
    take the steps in turn."""

    def __init__(self, cfg, F, local=0, rank=0, n_sets=3, lr_streams=2, blur_kind=0, fused_queries=True):
        import torch
        from refactored_orb_slam2_amd import ORBextractor, synth
        from refactored_orb_slam2_amd.matcher import Matcher
        self.cfg, self.F, self.local = cfg, F, local
        self.fused_queries = bool(fused_queries)
        W, H, NFEAT, STEREO = cfg["w"], cfg["h"], cfg["nfeat"], cfg["stereo"]
        self.W, self.H, self.NFEAT, self.STEREO = W, H, NFEAT, STEREO
        self.dev = dev = torch.device("cuda", local)
        # ---- synthetic input (each rank its own sequence), resident in HBM before the timed region; the pinned host copy
        #      feeds the PCIe-inclusive measurement
        #      Images live in buffers whose rows are PITCH = ceil64(W) bytes apart (what hipMemcpy2D / a decoder delivers): rows that
        #      start on 16-byte boundaries are used as pyramid level 0 in place (include/orbfe.h); tightly packed odd-width rows would
        #      cost one pitched copy per image first
        self.data = data = synth.sequence(W, H, F, seq=rank, stereo=STEREO)
        self.PITCH = PITCH = (W + 63) // 64 * 64

        def pitched_host(imgs):
            t = torch.zeros((F, H, PITCH), dtype=torch.uint8).pin_memory()
            t[:, :, :W] = torch.from_numpy(np.stack(imgs))
            return t

        self.hL = pitched_host([p[0] for p in data] if STEREO else data)
        self.hR = pitched_host([p[1] for p in data]) if STEREO else None
        mkex = lambda: ORBextractor(NFEAT, 1.2, NLEVELS, 20, 7, device=local)
        self.exL = mkex()
        self.exR = mkex() if STEREO else None
        self.extractors = [e for e in (self.exL, self.exR) if e is not None]
        self.mt = Matcher(local)
        # more sets (--sets S): their own extractors (the matching half of a step reads their pyramids: the stereo SAD windows) and matcher
        self.two_sets = n_sets >= 2 and lr_streams == 2 and STEREO and cfg["match"] == "projection"
        self.n_sets = n_sets if self.two_sets else 1
        self.more_ex, self.more_mt = [], []
        for _ in range(self.n_sets - 1):
            self.more_ex.append((mkex(), mkex()))
            self.more_mt.append(Matcher(local))
            self.extractors += list(self.more_ex[-1])
        if blur_kind:
            for e in self.extractors:
                if e._L.orbfe_debug_blur_kernel(e._h, blur_kind) != 0:
                    raise SystemExit("orbfe_debug_blur_kernel refused the kind")
        self.cap = self.exL.max_keypoints(W, H)
        self.cams_np, self.poses_np = camera_records(F, self.exL.GetScaleFactors(), cfg)
        self.t_cams = torch.from_numpy(self.cams_np.view(np.uint8).reshape(F, -1)).to(dev)
        self.t_poses = torch.from_numpy(self.poses_np.view(np.uint8).reshape(F, -1)).to(dev)
        self.mb = cfg["bf"] / cfg["fx"]
        self.B0 = self.Buffers()
        # explicit HIP streams: torch's default stream is the NULL stream, which the C ABI reads as "use the handle's own
        # stream"; the whole step therefore runs on named streams ordered by events
        self.sM, self.sL, self.sR = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        self.evL, self.evR = torch.cuda.Event(), torch.cuda.Event()
        self.lr = {"n": lr_streams}   # the stage-time passes switch to one stream: a launch's event time is then the kernel's own
        # ---- the sets in turn: extraction of set k % S on sL / sR as soon as the matching half that last read that set (S
        #      steps back) has ended; the matching half on sM behind the two extractions
        self.more_B = [self.Buffers() for _ in range(self.n_sets - 1)]
        ev = torch.cuda.Event
        self.pipe_sets = ([(self.exL, self.exR, self.mt, self.B0, ev(), ev(), ev())] +
                          [(self.more_ex[i][0], self.more_ex[i][1], self.more_mt[i], self.more_B[i], ev(), ev(), ev())
                           for i in range(self.n_sets - 1)]) if self.two_sets else None
        self.pipe_k = 0
        if self.two_sets:
            for ps in self.pipe_sets:
                ps[6].record(self.sM)
        self.gatherer = None
        self.world = 1
        self._dist = None

    def Buffers(self):
        return _Buffers(self)

    def all_buffers(self):
        return [self.B0] + self.more_B

    def close(self):   # release the library handles while the HIP runtime is still alive, also after an exception
        import torch
        try:
            torch.cuda.synchronize()
        except Exception:
            pass
        for hnd in self.extractors + [self.mt] + self.more_mt:
            try:
                hnd.close()
            except Exception:
                pass

    def _queries(self, B, s):
        """Frame::UnprojectStereo per keypoint with depth (L/src/Frame.cc:668-679), projected into the next frame (L/src/ORBmatcher.cc:1270-1308):
        one pass (default) or the two calls with the 60-byte point records in between -- byte-equal queries either way
        (tests/test_matcher_gpu.py)."""
        from refactored_orb_slam2_amd.matcher import track_queries_batch, track_queries_stereo_batch, unproject_stereo_batch
        if self.fused_queries:
            track_queries_stereo_batch(B.kl, B.dl, B.nl, B.depth, self.t_cams, 1, self.t_poses, 1, B.q, B.nq, s)
        else:
            unproject_stereo_batch(B.kl, B.dl, B.nl, B.depth, self.t_cams, 1, B.pts, s)
            track_queries_batch(self.t_poses, B.pts, B.nl, 1, B.q, B.nq, s)

    def _step(self, cur, B):
        cfg, W, H, STEREO = self.cfg, self.W, self.H, self.STEREO
        exL, exR, mt, sL, sR = self.exL, self.exR, self.mt, self.sL, self.sR
        if STEREO and self.lr["n"] == 2:
            sL.wait_stream(cur); sR.wait_stream(cur)
            exL.extract_batch_device(B.dL, B.kl, B.dl, B.nl, stream=sL)   # ORBextractor left  (Frame.cc:87-90: two threads)
            exR.extract_batch_device(B.dR, B.kr, B.dr, B.nr, stream=sR)   # ORBextractor right
            self.evL.record(sL); self.evR.record(sR)
            cur.wait_event(self.evL); cur.wait_event(self.evR)
        else:
            exL.extract_batch_device(B.dL, B.kl, B.dl, B.nl, stream=cur)
            if STEREO:
                exR.extract_batch_device(B.dR, B.kr, B.dr, B.nr, stream=cur)
        if STEREO:
            mt.stereo_match(exL, exR, B.kl, B.dl, B.nl, B.kr, B.dr, B.nr, cfg["bf"], self.mb, B.ur, B.depth, B.n_stereo, stream=cur)
        if cfg["match"] == "projection":
            self._queries(B, cur)
            B.blocked.zero_(); B.assigned.fill_(-1)
            mt.proj_match_batch(B.kl, B.dl, B.nl, B.ur, (0.0, float(W), 0.0, float(H)), B.q, B.nq, 1, 0.9, True, B.blocked,
                                B.assigned, B.n_track, stream=cur)                    # SearchByProjection(cur, last, th)
        else:
            # SearchByBoW's brute force: frame f against frame f-1 (slices of the same buffers: no copies), grouped by node id
            B.grp.copy_(node_ids(B.dl))
            mt.hamming_bf_batch(B.dl[1:], B.nl[1:], B.dl[:-1], B.nl[:-1], B.grp[1:], B.grp[:-1], B.bf[1:], stream=cur)
        if self.gatherer is not None:  # the path's only exchange: gather of the per-frame keypoint records (overlaps the next step)
            self.gatherer.launch(B.nl, B.kl, B.dl)

    def _step_piped(self):
        cfg, W, H, sM, sL, sR = self.cfg, self.W, self.H, self.sM, self.sL, self.sR
        xl, xr, m, B, eL, eR, eT = self.pipe_sets[self.pipe_k % self.n_sets]
        self.pipe_k += 1
        sL.wait_event(eT); sR.wait_event(eT)
        xl.extract_batch_device(B.dL, B.kl, B.dl, B.nl, stream=sL)
        xr.extract_batch_device(B.dR, B.kr, B.dr, B.nr, stream=sR)
        eL.record(sL); eR.record(sR)
        sM.wait_event(eL); sM.wait_event(eR)
        m.stereo_match(xl, xr, B.kl, B.dl, B.nl, B.kr, B.dr, B.nr, cfg["bf"], self.mb, B.ur, B.depth, B.n_stereo, stream=sM)
        self._queries(B, sM)
        B.blocked.zero_(); B.assigned.fill_(-1)
        m.proj_match_batch(B.kl, B.dl, B.nl, B.ur, (0.0, float(W), 0.0, float(H)), B.q, B.nq, 1, 0.9, True, B.blocked, B.assigned,
                           B.n_track, stream=sM)
        if self.gatherer is not None:
            self.gatherer.launch(B.nl, B.kl, B.dl)
        eT.record(sM)

    def step(self):
        import torch
        sM = self.sM
        with torch.cuda.stream(sM):
            if self.two_sets and self.lr["n"] == 2:
                self._step_piped()
            else:
                if self.two_sets:
                    sM.wait_stream(self.sL); sM.wait_stream(self.sR)   # (a one-stream pass behind pipelined steps ...
                self._step(sM, self.B0)
                if self.two_sets:
                    self.pipe_sets[0][6].record(sM)               #  ... and in front of the next ones: set 0 is free when this step has ended)

    def barrier(self):
        import torch
        if self.gatherer is not None:
            with torch.cuda.stream(self.sM):
                self.gatherer.wait()
        torch.cuda.synchronize()
        if self.world > 1:
            self._dist.barrier()
        torch.cuda.synchronize()

    # ---- self-check of the timed layout (config.self_check): every set's outputs hashed on the device side's host copy
    OUT_FIELDS = ("nl", "kl", "dl", "nr", "kr", "dr", "ur", "depth", "n_stereo", "n_track", "assigned")

    def output_digest(self, B):
        """SHA-256 over every field a step leaves in `B` (counts, keypoints, descriptors of both eyes, mvuRight / mvDepth, stereo and
        tracked counts, the tracked assignment).  Rows behind a frame's count are whatever earlier steps left: only the first n
        entries of a row are hashed."""
        import hashlib
        import torch
        h = hashlib.sha256()
        torch.cuda.synchronize()
        nl = B.nl.cpu().numpy()
        nr = B.nr.cpu().numpy() if self.STEREO else None
        h.update(nl.tobytes())

        def rows(t, n):
            a = t.cpu().numpy()
            m = np.arange(a.shape[1])[None, :] < n[:, None]
            h.update(np.ascontiguousarray(a[m]).tobytes())

        rows(B.kl, nl); rows(B.dl, nl)
        if self.STEREO:
            h.update(nr.tobytes()); rows(B.kr, nr); rows(B.dr, nr); rows(B.ur, nl); rows(B.depth, nl)
            h.update(B.n_stereo.cpu().numpy().tobytes())
        if self.cfg["match"] == "projection":
            h.update(B.n_track.cpu().numpy().tobytes()); rows(B.assigned, nl)
        return h.hexdigest()

    def self_check(self):
        """After steps on the timed layout: every set's outputs equal each other and equal what a ONE-stream, one-set step leaves
        (every kernel alone on the chip, the layout the stage parity tests run)."""
        digs = [self.output_digest(B) for B in self.all_buffers()]
        keep = self.lr["n"]
        self.lr["n"] = 1
        self.step(); self.barrier()
        one = self.output_digest(self.B0)
        self.lr["n"] = keep
        for _ in range(self.n_sets):   # back on the timed layout, every set written once more
            self.step()
        self.barrier()
        again = [self.output_digest(B) for B in self.all_buffers()]
        return {"sets_equal": len(set(digs)) == 1 and len(set(again)) == 1, "equals_one_stream": all(d == one for d in digs + again),
                "sets": len(digs), "sha256_16": one[:16]}


class _Buffers:
    """One set of device inputs and outputs of a step (two sets double-buffer the PCIe-inclusive run)."""

    def __init__(self, rig):
        import torch
        F, cap, dev, STEREO, W = rig.F, rig.cap, rig.dev, rig.STEREO, rig.W
        z = lambda *s, dt=torch.uint8: torch.zeros(s, dtype=dt, device=dev)
        self.dL_full = rig.hL.to(dev)                       # (F, H, PITCH); the extractor sees the (F, H, W) view
        self.dR_full = rig.hR.to(dev) if STEREO else None
        self.dL = self.dL_full[:, :, :W]
        self.dR = self.dR_full[:, :, :W] if STEREO else None
        self.kl, self.dl, self.nl = z(F, cap, 28), z(F, cap, 32), z(F, dt=torch.int32)
        if STEREO:
            self.kr, self.dr, self.nr = z(F, cap, 28), z(F, cap, 32), z(F, dt=torch.int32)
        self.ur = z(F, cap, dt=torch.float32) if STEREO else None
        self.depth = z(F, cap, dt=torch.float32) if STEREO else torch.full((F, cap), MONO_DEPTH, dtype=torch.float32, device=dev)
        self.n_stereo = z(F, dt=torch.int32)
        self.blocked, self.assigned, self.n_track = z(F, cap), z(F, cap, dt=torch.int32), z(F, dt=torch.int32)
        self.pts, self.q, self.nq = z(F, cap, 60), z(F, cap, 68), z(F, dt=torch.int32)   # orbfe_last_point / orbfe_query
        self.bf = z(F, cap, 12)                                                          # orbfe_bf_match records
        self.grp = z(F, cap, dt=torch.int32)



def main():
    if os.environ.get("ORBFE_BENCH_WATCHDOG"):  # debugging aid: dump all stacks and exit if the run exceeds N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["ORBFE_BENCH_WATCHDOG"]), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300, help="timed steps (default: ~1.2 s of timed region)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=256, help="frames per GPU per step")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="kitti_stereo")
    ap.add_argument("--cpu-sample", type=int, default=160, help="frames timed on the CPU oracle (0 = skip)")
    ap.add_argument("--e2e-steps", type=int, default=12, help="steps of the PCIe-inclusive measurement (0 = skip)")
    ap.add_argument("--per-frame", type=int, default=192, help="stereo pairs pushed one at a time through the C++ drop-in classes for per_frame_ms (0 = skip)")
    ap.add_argument("--c-abi-gather", type=int, default=0, help="N > 1: after the line, run the record gather through the C ABI (orbfe_gather_*, RCCL inside "
                                                                "liborbfe) in both modes, compare with the local records, report on stderr (default off: "
                                                                "the path has never run with peers and must not be able to cost the line)")
    ap.add_argument("--content-steps", type=int, default=30, help="steps per image content of the content-sensitivity key (0 = skip)")
    ap.add_argument("--gather-impl", choices=("auto", "torch", "cabi"), default="auto",
                    help="which implementation of the record gather is the timed collective: `cabi` = the C ABI's (orbfe_gather_*: RCCL called by "
                         "liborbfe, what a C++ host of the batched mode uses; with it the gather also runs at N = 1, a one-rank communicator), "
                         "`torch` = torch.distributed (sharding.AsyncGather).  `auto` (default): at N > 1 on RCCL the C ABI's -- the product's own "
                         "collective --, falling back to torch.distributed if its communicator cannot be created (the path has not run with "
                         "peers on this pool: DESIGN 6; the line says which one ran); nothing at N = 1")
    ap.add_argument("--gather", choices=("all", "root"), default="all",
                    help="N > 1: all_gather of the per-frame records on every rank, or gather to rank 0 only (SURVEY.md 8(e))")
    ap.add_argument("--blur-kind", type=int, default=0, choices=(0, 1, 2),
                    help="A/B of the level chain (orbfe_debug_blur_kernel): 0 fused level kernels (default), 1 resize chain + matrix-core blur, "
                         "2 resize chain + one LDS blur launch (rounds 1-4)")
    ap.add_argument("--driver", choices=("torch", "cabi"), default="torch",
                    help="what drives the timed step: torch streams / events around the C ABI's device entry points (default), or the C ABI's "
                         "own pipeline handle (orbfe_pipeline_submit_resident: no torch in the loop, what a C++ host runs).  At N = 1 the line "
                         "carries both (config.cabi_driver); with `cabi`, `value` is the handle's rate")
    ap.add_argument("--cabi-outputs", type=int, default=8, help="orbfe_pipeline_config.output_mask of the C-ABI driver (8: tracked assignments + counts, "
                                                                 "16: counts only, 0: every block)")
    ap.add_argument("--cabi-steps", type=int, default=120, help="chunks timed through the C-ABI driver (0 = skip)")
    ap.add_argument("--queries", choices=("fused", "two_pass"), default="fused",
                    help="fused (default): UnprojectStereo + the projection of SearchByProjection(cur, last) in one pass "
                         "(orbfe_track_queries_stereo_device); two_pass: orbfe_unproject_stereo_device -> 60-byte point records -> "
                         "orbfe_track_queries_device (rounds 3-5).  Byte-equal queries")
    ap.add_argument("--lr-streams", type=int, default=2, choices=(1, 2),
                    help="2 (default): left / right extractor on two HIP streams, as the reference runs them on two threads (Frame.cc:87-90): "
                         "their launches overlap and fill each other's tails; 1: one stream, every kernel alone on the chip (the per-stage times "
                         "of `roofline.stage_ms_per_batch` are always measured that way, in an untimed pass)")
    ap.add_argument("--sets", type=int, default=3, choices=(1, 2, 3, 4),
                    help="S > 1 (default 3, with --lr-streams 2): S sets of handles and buffers take the steps in turn -- the extraction of step "
                         "k + 1 runs beside the matching half of step k (stereo match, unproject, track queries, projection search), which has its "
                         "own stream, and waits only for the matching half that last read ITS set, S steps back; every step still does all of its "
                         "work inside the timed region.  1: a step starts when the one before it has ended")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))   # the parent never initialises the GPU; children report through their exit codes

    import torch
    import torch.distributed as dist
    from refactored_orb_slam2_amd import ORBextractor, synth
    from refactored_orb_slam2_amd.matcher import Matcher
    from refactored_orb_slam2_amd.sharding import AsyncGather, CabiAsyncGather, gather_traffic

    cfg = CONFIGS[args.config]
    W, H, NFEAT, STEREO = cfg["w"], cfg["h"], cfg["nfeat"], cfg["stereo"]
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start bench.py with --gpus equal to the number of ranks")
    # rehearsal switches for boxes with fewer GPUs than ranks (not used by the driver): ORBFE_BENCH_SHARE_DEVICE=1 maps
    # every rank to the visible devices round-robin (the gather then runs over gloo through host memory)
    share = os.environ.get("ORBFE_BENCH_SHARE_DEVICE") == "1" and torch.cuda.device_count() < world
    backend = os.environ.get("ORBFE_BENCH_BACKEND", "gloo" if share else "nccl")   # RCCL refuses two ranks on one device
    if share:
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    F = args.frames
    rig = StepRig(cfg, F, local=local, rank=rank, n_sets=args.sets, lr_streams=args.lr_streams, blur_kind=args.blur_kind,
                  fused_queries=args.queries == "fused")
    import atexit
    atexit.register(rig.close)
    rig.world, rig._dist = world, dist
    data, PITCH, hL, hR = rig.data, rig.PITCH, rig.hL, rig.hR
    exL, exR, extractors, mt = rig.exL, rig.exR, rig.extractors, rig.mt
    two_sets, n_sets, more_B, B0, Buffers = rig.two_sets, rig.n_sets, rig.more_B, rig.B0, rig.Buffers
    sM, sL, sR, lr = rig.sM, rig.sL, rig.sR, rig.lr
    _step, step, barrier = rig._step, rig.step, rig.barrier
    gatherer, gather_impl, gather_fallback = None, None, None
    want_cabi = backend == "nccl" and (args.gather_impl == "cabi" or (args.gather_impl == "auto" and world > 1))
    if want_cabi:
        try:
            rig.gatherer = gatherer = CabiAsyncGather(B0.nl, B0.kl, B0.dl, rank, world, local, mode=args.gather)
            gather_impl = "cabi"
        except Exception as ex:   # every rank takes part in the unique-id broadcast first, so a refusal is seen by all of them alike
            if args.gather_impl == "cabi":
                raise
            gather_fallback = str(ex)[:200]
        if world > 1 and args.gather_impl == "auto":
            # all ranks agree on the implementation: one rank without a communicator sends every rank to torch.distributed
            flag = torch.tensor([1 if gatherer is not None else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0 and gatherer is not None:
                gatherer.close(); rig.gatherer = gatherer = None; gather_impl = None
                gather_fallback = gather_fallback or "another rank could not create its communicator"
    if gatherer is None and world > 1:
        rig.gatherer = gatherer = AsyncGather(B0.nl, B0.kl, B0.dl, mode=args.gather)
        gather_impl = "torch"

    for _ in range(max(args.warmup, 1) if args.warmup >= 0 else 0):
        step()
    barrier()
    for e in extractors:
        e.device_status()
    n_img = (2 if STEREO else 1) * F   # images per step (a second set of handles takes every other step: not more images)
    n_kp = int(B0.nl.sum().item()) + (int(B0.nr.sum().item()) if STEREO else 0)
    n_st = int(B0.n_stereo.sum().item())
    n_tr = int(B0.n_track.sum().item()) if cfg["match"] == "projection" else int((B0.bf.view(torch.int32)[..., 0] >= 0).sum().item())
    # measured FAST candidates per image (for the algorithmic byte count of the FAST kernel)
    cand_per_img = float(np.mean([sum(len(exL.debug_candidates(i, l)[0]) for l in range(NLEVELS)) for i in range(min(F, 4))]))

    # all-stage event timing costs ~2 % of the step: a short untimed pass yields the stage breakdown (and names the
    # dominant kernel); inside the timed region only that kernel is bracketed by HIP events on its launch stream
    def stage_sums():
        tot = {}
        for e in extractors:
            for k, v in e.stage_times().items():
                a = tot.get(k, (0.0, 0))
                tot[k] = (a[0] + v[0], a[1] + v[1])
        return tot

    lr["n"] = 1
    for e in extractors:
        e.profile(True); e.stage_times(reset=True)
    for _ in range(3):
        step()
    barrier()
    lr["n"] = args.lr_streams
    stage_ms_all = {k: (v[0] / max(v[1] // 2 if k == "pyramid" else v[1], 1)) for k, v in stage_sums().items()}
    # the dominant KERNEL: the level chain ("pyramid") is eight launches, none of them as long as FAST's one
    dom = max((k for k in stage_ms_all if k != "pyramid"), key=lambda k: stage_ms_all[k])
    # the dominant kernel is bracketed by HIP events on every DOM_EVERY-th step of the timed region: an event record in front of and
    # behind a launch leaves ~6 us of idle GPU each (kernel trace), 24 us per stereo step if every launch were timed (0.8 %)
    DOM_EVERY = 8
    # the dominant kernel alone on the chip, timed without event records around the other stages (those leave idle gaps in front of
    # it: FAST then finds less of the pyramid in the Infinity Cache): sixteen one-stream steps -- the `roofline` figure
    lr["n"] = 1
    for e in extractors:
        e.profile(True, [dom]); e.stage_times(reset=True)
    for _ in range(16):
        step()
    barrier()
    lr["n"] = args.lr_streams
    _sd = stage_sums()[dom]
    dom_alone_launches = max(_sd[1] // 2 if dom == "pyramid" else _sd[1], 1)
    stage_ms_all[dom] = _sd[0] / dom_alone_launches
    for e in extractors:
        e.profile(False); e.stage_times(reset=True)
    for _ in range(max(n_sets, 2)):
        step()           # back on the timed region's stream layout
    barrier()
    ref_ev = torch.cuda.Event(enable_timing=True)   # the clock of the dominant kernel's launch intervals (orbfe_stage_intervals)
    ref_ev.record(sM)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i % DOM_EVERY == 0:
            for e in extractors:
                e.profile(True, [dom])
        elif i % DOM_EVERY == 1:
            for e in extractors:
                e.profile(False)
        step()
    barrier()
    dt = time.perf_counter() - t0
    rank_rates, coll_ms = None, None
    gather_check = None
    if gatherer is not None:
        # the gathered records of the last timed step == this rank's own records at its slot (the collective moved the right bytes)
        res = gatherer.result()
        if res is not None:
            mine = slice(rank * F, (rank + 1) * F)
            gather_check = bool((res[0][mine].to(B0.nl.device) == B0.nl).all()) and bool((res[2][mine].to(B0.dl.device) == B0.dl).all())
    if world > 1:
        # every rank's own clock (the straggler shows), the MAX over ranks is the job's time
        mine = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_rates = [round(F * args.steps / float(x.item()), 1) for x in every]
        dt = max(float(x.item()) for x in every)
        # the collective on its own (not overlapped with a step): HIP events around three blocking gathers
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(sM):
            gatherer.launch(B0.nl, B0.kl, B0.dl); gatherer.wait()
            torch.cuda.synchronize(); dist.barrier()
            ev0.record(sM)
            for _ in range(3):
                gatherer.launch(B0.nl, B0.kl, B0.dl); gatherer.wait()
            ev1.record(sM)
        torch.cuda.synchronize()
        coll_ms = ev0.elapsed_time(ev1) / 3
    # the dominant kernel's timed launches as intervals on one clock: with the two extractors on two streams a launch's own
    # first-to-last-event time contains the other launch's share of the chip; the UNION of the intervals is the time the chip spent on
    # the kernel, and the bytes of all those launches over that time its achieved rate
    dom_iv = []
    for e in extractors:
        dom_iv += [(a, b) for (st, a, b) in e.stage_intervals(ref_ev) if st == dom]
    dom_iv.sort()
    dom_union, cur_a, cur_b = 0.0, None, None
    for a, b in dom_iv:
        if cur_b is None or a > cur_b:
            if cur_b is not None:
                dom_union += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    if cur_b is not None:
        dom_union += cur_b - cur_a
    dom_own = sum(b - a for a, b in dom_iv)
    st_dom = stage_sums()[dom]
    for e in extractors:
        e.profile(False)
    # ---- self-check of the layout that was just timed (after the clock stopped): what the last steps left in every set's buffers
    #      is identical across the sets and identical to a one-stream, one-set step (the layout the stage parity tests cover);
    #      tests/test_bench_layout_gpu.py compares the same object's outputs with the oracle
    self_check = rig.self_check()
    # ---- the same step driven through the C ABI's pipeline handle (N = 1, STEREO + projection only: what the handle implements)
    cabi = None
    if world == 1 and STEREO and cfg["match"] == "projection" and (args.cabi_steps > 0 or args.driver == "cabi"):
        try:
            cabi = cabi_driver(rig, args.steps if args.driver == "cabi" else args.cabi_steps, args.cabi_outputs, slots=max(n_sets, 2))
        except Exception as ex:
            if args.driver == "cabi":
                raise
            cabi = {"error": str(ex)[:300]}

    # ---- PCIe-inclusive rate (N = 1): pinned host images in, host keypoints / descriptors / match results out, two buffer
    #      sets: the H2D copy of step i+1 and the D2H copy of step i-1 run beside the kernels of step i
    e2e = None
    if world == 1 and args.e2e_steps > 0:
        Bs = [B0, Buffers()]
        sH, sD = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        host_out = [[torch.empty_like(t, device="cpu").pin_memory() for t in
                     ([b.kl, b.dl, b.nl, b.assigned] + ([b.ur, b.depth] if STEREO else []))] for b in Bs]
        ev_in = [torch.cuda.Event() for _ in Bs]; ev_done = [torch.cuda.Event() for _ in Bs]; ev_out = [torch.cuda.Event() for _ in Bs]

        def e2e_run(n):
            for i in range(n):
                j = i & 1
                b = Bs[j]
                with torch.cuda.stream(sH):
                    sH.wait_event(ev_done[j])                 # the kernels that read this input set have finished
                    b.dL_full.copy_(hL, non_blocking=True)
                    if STEREO:
                        b.dR_full.copy_(hR, non_blocking=True)
                    ev_in[j].record(sH)
                with torch.cuda.stream(sM):
                    sM.wait_event(ev_in[j]); sM.wait_event(ev_out[j])   # inputs here, previous results of this set copied out
                    _step(sM, b)
                    ev_done[j].record(sM)
                with torch.cuda.stream(sD):
                    sD.wait_event(ev_done[j])
                    srcs = [b.kl, b.dl, b.nl, b.assigned] + ([b.ur, b.depth] if STEREO else [])
                    for dst, src in zip(host_out[j], srcs):
                        dst.copy_(src, non_blocking=True)
                    ev_out[j].record(sD)
            torch.cuda.synchronize()

        for ev in ev_done + ev_out:
            ev.record(sM)
        e2e_run(2)
        t1 = time.perf_counter()
        e2e_run(args.e2e_steps)
        e2e = F * args.e2e_steps / (time.perf_counter() - t1)

    # ---- content sensitivity (N = 1, outside the timed region): the same step on other image content.  The headline runs on the
    #      synthetic generator of SURVEY.md 8(d); FAST's cost drivers (fallback cells, quick-test early-outs) differ 2-3 x between
    #      it and real texture (DESIGN lesson 18), dense noise sends the quadtree's keys to HBM, a flat frame has no corner at all.
    content = None
    if world == 1 and args.content_steps > 0 and STEREO and cfg["match"] == "projection":
        def content_images(kind):
            rng = np.random.default_rng(1234)
            L = np.zeros((F, H, PITCH), np.uint8)
            if kind == "real_texture":      # the four DBoW2 demo images of the reference checkout (tests/golden/real_demo.npz), tiled
                demo = np.load(os.path.join(ROOT, "tests", "golden", "real_demo.npz"))["images"][:4]
                reps = (H + demo.shape[1] - 1) // demo.shape[1] + 1, (W + 64 + demo.shape[2] - 1) // demo.shape[2] + 1
                big = [np.tile(d, reps) for d in demo]
                for f in range(F):
                    oy, ox = (37 * f) % demo.shape[1], (53 * f) % demo.shape[2]
                    L[f, :, :W] = big[f % 4][oy:oy + H, ox:ox + W]
            elif kind == "dense_texture":   # a stand-in for foliage: the demo images at twice the contrast plus +-6 grey levels of pixel noise
                demo = np.load(os.path.join(ROOT, "tests", "golden", "real_demo.npz"))["images"][:4]
                reps = (H + demo.shape[1] - 1) // demo.shape[1] + 1, (W + 64 + demo.shape[2] - 1) // demo.shape[2] + 1
                big = [np.tile(d, reps) for d in demo]
                for f in range(F):
                    oy, ox = (37 * f) % demo.shape[1], (53 * f) % demo.shape[2]
                    t = big[f % 4][oy:oy + H, ox:ox + W].astype(np.int16)
                    L[f, :, :W] = np.clip(128 + 2 * (t - 128) + rng.integers(-6, 7, (H, W)), 0, 255).astype(np.uint8)
            elif kind == "uniform_noise":
                base = rng.integers(0, 256, (8, H, W + 64), dtype=np.uint8)
                for f in range(F):
                    L[f, :, :W] = base[f % 8][:, (f // 8) % 32:(f // 8) % 32 + W]
            elif kind == "sensor_noise":    # the synthetic sequence with +-8 grey levels of per-pixel noise: 4.5 k FAST candidates at level 0
                for f in range(F):          # instead of 1.4 k -- more than the quadtree's LDS holds (1 750 keys per level)
                    L[f, :, :W] = np.clip(hL[f, :, :W].numpy().astype(np.int16) + rng.integers(-8, 9, (H, W)), 0, 255).astype(np.uint8)
            elif kind == "flat":
                L[:, :, :W] = 128
            R = np.zeros_like(L)
            R[:, :, :W - 10] = L[:, :, 10:W]     # the right eye sees the scene 10 pixels further left
            R[:, :, W - 10:W] = L[:, :, W - 10:W]
            return torch.from_numpy(L), torch.from_numpy(R)

        content = {}
        for kind in ("real_texture", "dense_texture", "sensor_noise", "uniform_noise", "flat"):
            try:
                tl, tr = content_images(kind)
                B0.dL_full.copy_(tl); B0.dR_full.copy_(tr)
                for Bx in more_B:
                    Bx.dL_full.copy_(tl); Bx.dR_full.copy_(tr)
                lr["n"] = 1
                for e in extractors:
                    e.profile(True); e.stage_times(reset=True)
                for _ in range(3):
                    step()
                barrier()
                lr["n"] = args.lr_streams
                for e in extractors:
                    e.device_status()
                st = {k: round(v[0] / max(v[1] // 2 if k == "pyramid" else v[1], 1), 4) for k, v in stage_sums().items()}
                for e in extractors:
                    e.profile(False)
                barrier()
                tc = time.perf_counter()
                for _ in range(args.content_steps):
                    step()
                barrier()
                dtc = time.perf_counter() - tc
                cand_lv = [round(float(np.mean([len(exL.debug_candidates(i, l)[0]) for i in range(min(F, 4))])), 1) for l in range(NLEVELS)]
                content[kind] = {"frames_per_s": round(F * args.content_steps / dtc, 1), "ms_per_step": round(dtc / args.content_steps * 1e3, 4),
                                 "fast_candidates_per_level": cand_lv,
                                 "stage_ms_per_batch": st, "keypoints_per_image": round((int(B0.nl.sum().item()) + int(B0.nr.sum().item())) / (2 * F), 1),
                                 "stereo_matches_per_frame": round(int(B0.n_stereo.sum().item()) / F, 1),
                                 "tracked_per_frame": round(int(B0.n_track.sum().item()) / F, 1)}
            except Exception as ex:   # a content the library refuses must not cost the headline line
                content[kind] = {"error": str(ex)[:200]}
        # packed input for comparison with earlier rounds: tightly packed rows (W bytes apart) cost one pitched copy per image first
        try:
            pk_l = torch.from_numpy(np.stack([p[0] for p in data])).to(dev); pk_r = torch.from_numpy(np.stack([p[1] for p in data])).to(dev)
            keepL, keepR = B0.dL, B0.dR
            B0.dL, B0.dR = pk_l, pk_r
            for Bx in more_B:
                Bx.dL, Bx.dR = pk_l, pk_r
            for _ in range(3):
                step()
            barrier()
            tc = time.perf_counter()
            for _ in range(args.content_steps):
                step()
            barrier()
            content["synthetic_packed_rows"] = {"frames_per_s": round(F * args.content_steps / (time.perf_counter() - tc), 1),
                                                "note": f"the headline's images in tightly packed rows ({W} bytes apart): level 0 is a pitched copy"}
            B0.dL, B0.dR = keepL, keepR
            for Bx in more_B:
                Bx.dL, Bx.dR = Bx.dL_full[:, :, :W], Bx.dR_full[:, :, :W]
        except Exception as ex:
            content["synthetic_packed_rows"] = {"error": str(ex)[:200]}
        B0.dL_full.copy_(hL); B0.dR_full.copy_(hR)
        for Bx in more_B:
            Bx.dL_full.copy_(hL); Bx.dR_full.copy_(hR)
        for _ in range(n_sets):
            step()
        barrier()

    if rank == 0:
        px = [exL.level_size(l, W, H) for l in range(NLEVELS)]
        sumP = sum(w * h for w, h in px)
        P0, P7 = px[0][0] * px[0][1], px[-1][0] * px[-1][1]
        # algorithmic bytes per image and per kernel; one launch processes F images
        alg = {
            # level chain: fused (launch l reads level l once, writes level l + 1 and the blurred level l), or the resize chain alone
            "pyramid": (3 * sumP - P0) if args.blur_kind == 0 else (sumP - P7) + (sumP - P0),
            "fast": sumP + 8 * cand_per_img,
            "octree": 8 * cand_per_img + 4 * NFEAT,
            "blur": 0 if args.blur_kind == 0 else 2 * sumP,
            "describe": NFEAT * (749 + 512 + 60),
        }
        per_launch_ms = dict(stage_ms_all)   # untimed 3-step pass on ONE stream (every stage, every kernel alone on the chip)
        cnt = st_dom[1] // 2 if dom == "pyramid" else st_dom[1]   # pyramid: two timed groups (level-0 copy, resize chain) per batch
        own_ms = st_dom[0] / max(cnt, 1)                          # the dominant kernel: live, over the timed region, first to last event of a launch
        n_iv = max(len(dom_iv), 1)
        chip_ms = (dom_union / n_iv) if dom_iv else own_ms        # the chip's time per launch: union of the launches' intervals / launches
        achieved = alg[dom] * F / (max(chip_ms, 1e-9) * 1e-3) / 1e9
        achieved_alone = alg[dom] * F / (max(per_launch_ms[dom], 1e-9) * 1e-3) / 1e9
        # HBM-side traffic of the dominant kernel: FETCH_SIZE (x2 for 16-byte-per-lane streams on gfx950) + WRITE_SIZE from the
        # committed PMC pass of THIS kernel build (profiles/rNN_pmc_dominant.json, latest round), scaled to this launch's image count;
        # null when the pass covers another kernel or configuration
        traffic, traffic_src, issue = None, None, None
        try:
            import glob
            pmc_path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_dominant.json")))[-1]   # the latest round's pass
            pmc = json.load(open(pmc_path))
            if pmc["stage"] == dom and pmc["config"] == args.config:
                if pmc.get("read_kb_by_request_size") is not None:   # fabric requests split by size (TCC_EA0_RDREQ / WRREQ passes): no correction factor
                    kb = pmc["read_kb_by_request_size"] + pmc["write_kb_by_request_size"]
                else:
                    kb = pmc["fetch_kb"] * pmc.get("fetch_correction", 1.0) + pmc["write_kb"]
                traffic = int(kb * 1024 * F / pmc["images_per_launch"])
                traffic_src = f"committed rocprofv3 --pmc pass ({pmc['kernel']}, {os.path.basename(pmc_path)}), not measured in this run"
                if pmc.get("issue"):
                    # what the kernel is bound by when it is not HBM (profiles/r06_fast.md): the instruction stream of its waves.  Lane-
                    # instructions per pixel = 64 lanes x vector instructions per wave x waves / the pixels one launch tests
                    issue = dict(pmc["issue"])
                    if dom == "fast":
                        issue["lane_instructions_per_pixel"] = round(64.0 * issue["valu_per_wave"] * issue["waves_per_launch"] /
                                                                     max(sumP * pmc["images_per_launch"], 1), 2)
                    issue["source"] = traffic_src
        except Exception:
            pass
        value = world * F * args.steps / dt
        ms_step = dt / args.steps * 1e3
        if args.driver == "cabi":   # the C ABI's handle is the measured step; the torch-driven figure stays in config.torch_driver
            torch_driver = {"frames_per_s": round(value, 2), "ms_per_step": round(ms_step, 4)}
            value, ms_step = cabi["frames_per_s"], cabi["ms_per_step"]
        out = {
            "metric": "frames/s (extract+match) at KITTI 1241×376, 2000 feat; 1/2/4/8 GPU + CPU ref",
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "input_layout": f"u8 images resident in HBM, rows {PITCH} bytes apart (16-byte aligned: level 0 of the pyramid in place)",
            "config": {"workload": cfg["label"], "name": args.config, "frames_per_gpu_per_step": F,
                       "step_layout": (f"left and right extractor on two HIP streams; {n_sets} sets of handles and buffers take the steps in turn: the "
                                       "extraction of step k + 1 runs beside the matching half of step k (third stream)") if two_sets else
                                      (f"one set of handles; extractors on {args.lr_streams if STEREO else 1} stream(s), a step starts when the one before it has ended"),
                       "images_per_step": n_img * world, "parallelism": f"frame-shard x{world}",
                       "collective": ((f"{'RCCL through the C ABI (orbfe_gather_records)' if gather_impl == 'cabi' else backend} "
                                       f"{'all_gather' if args.gather == 'all' else 'gather to rank 0'} of padded per-frame records, world size {world}") if gatherer is not None else "none"),
                       "keypoints_per_image": round(n_kp / n_img, 1), "stereo_matches_per_frame": round(n_st / F, 1),
                       "matches_per_frame": round(n_tr / F, 1), "timed_region_s": round(dt, 3), "self_check": self_check},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved_alone, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved_alone / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src, "issue": issue,
                         "algorithmic_bytes_per_launch": int(alg[dom] * F), "avg_launch_ms": round(per_launch_ms[dom], 4),
                         "measured": (f"HIP events on the launch stream around {dom_alone_launches} launches of the kernel in a one-stream pass of this run (every "
                                      "kernel alone on the chip: the figure rocprofv3 --stats and the serialised --pmc passes can be compared with -- "
                                      "profiles/rNN_kernel_stats.md, one-stream table).  In the TIMED region launches overlap by design (left | right "
                                      "extractor on two streams, the matching half of the step before on a third), a launch's own duration there contains "
                                      "the others' share of the chip: see timed_region"),
                         "timed_region": {"avg_launch_ms": round(own_ms, 4), "chip_ms_per_launch": round(chip_ms, 4),
                                          "launches_overlapping": round(dom_own / max(dom_union, 1e-9), 2) if dom_iv else 1.0,
                                          "timed_launches": len(dom_iv), "achieved": round(achieved, 2), "frac": round(achieved / HBM_PEAK_GBS, 5),
                                          "lr_streams": lr["n"] if STEREO else 1, "handle_sets": n_sets,
                                          "note": "HIP events around the kernel's launches on every 8th step of the timed region (orbfe_stage_intervals): "
                                                  "avg_launch_ms = a launch's own first-to-last-event time; chip_ms_per_launch = union of the launches' "
                                                  "intervals / launches; achieved = their algorithmic bytes / that union"},
                         "stage_ms_per_batch": {k: round(v, 4) for k, v in per_launch_ms.items()}},
        }
        out["config"]["driver"] = args.driver
        if cfg["match"] == "projection":
            out["config"]["queries"] = ("one pass: orbfe_track_queries_stereo_device" if rig.fused_queries
                                        else "two passes: orbfe_unproject_stereo_device + orbfe_track_queries_device")
        if cabi is not None:
            out["config"]["cabi_driver"] = cabi
        if args.driver == "cabi":
            out["config"]["torch_driver"] = torch_driver
        if gatherer is not None:
            out["config"]["gather"] = dict(gather_traffic(B0.nl, B0.kl, B0.dl, world, args.gather), impl=gather_impl, impl_fallback_reason=gather_fallback,
                                           own_slot_equal_rank0=gather_check,
                                           standalone_ms_rank0=(round(coll_ms, 4) if coll_ms is not None else None),
                                           note="per step, overlapped with the next step's kernels; standalone_ms = the same collective alone, HIP events on rank 0"
                                                + ("; world 1: a one-rank communicator, the collective degenerates to a copy" if world == 1 else ""))
        if world > 1:
            out["config"]["frames_per_s_by_rank"] = rank_rates
        if e2e is not None:
            out["e2e_frames_per_s"] = round(e2e, 1)
            out["e2e_note"] = "pinned host images -> H2D -> step -> D2H of keypoints, descriptors, counts, matches (and mvuRight / mvDepth), double-buffered"
        if content is not None:
            out["content"] = content
            out["content_note"] = (f"the same step ({args.content_steps} steps each, outside the timed region) on other content: the DBoW2 demo images tiled "
                                   f"to {W}x{H}, the same at twice the contrast with +-6 grey levels of noise (a stand-in for foliage), the synthetic sequence with +-8 grey levels "
                                   f"of pixel noise (the quadtree's keys of level 0 leave LDS: 1 664 per level), "
                                   f"uniform noise (every level's keys in HBM), a flat frame; `value` is the synthetic sequence of SURVEY.md 8(d)")
        if world == 1 and args.per_frame > 0 and STEREO:
            pf = per_frame_latency(cfg, args.per_frame)
            out["per_frame_ms"] = pf.get("median_ms")
            out["per_frame"] = pf
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(args.config, args.cpu_sample)
        print(json.dumps(out), flush=True)
    if cabi is not None and cabi.get("equals_torch_driven_step") is False:
        raise SystemExit(f"rank {rank}: the C-ABI pipeline handle's results differ from the torch-driven step's")
    if not (self_check["sets_equal"] and self_check["equals_one_stream"]):
        raise SystemExit(f"rank {rank}: the timed layout's outputs differ between the handle sets or from the one-stream step: {self_check}")
    if world > 1:
        # the gathered records of the last step == every rank's own records at its slot (the collective moved the right bytes)
        res = gatherer.result()
        if res is not None:   # mode "root": rank 0 alone holds the records; it checks its own slot
            n_all, k_all, d_all = res
            mine = slice(rank * F, (rank + 1) * F)
            ok = bool((n_all[mine].to(B0.nl.device) == B0.nl).all()) and bool((d_all[mine].to(B0.dl.device) == B0.dl).all())
            if not ok:
                raise SystemExit(f"rank {rank}: gathered records differ from the local ones")
        if args.c_abi_gather and backend == "nccl":
            # the same exchange through the C ABI: unique id from rank 0 over the control-plane group, one communicator per rank
            from refactored_orb_slam2_amd.sharding import RcclGather
            uid = [RcclGather.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            cg = RcclGather(uid[0], rank, world, local)
            report = {}
            for mode in ("all", "root"):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                with torch.cuda.stream(sM):
                    e0.record(sM)
                    got = cg.gather(B0.nl, B0.kl, B0.dl, mode=mode, stream=sM)
                    e1.record(sM)
                torch.cuda.synchronize()
                good = True
                if got is not None:
                    mine = slice(rank * F, (rank + 1) * F)
                    good = bool((got[0][mine] == B0.nl).all()) and bool((got[2][mine] == B0.dl).all())
                report[mode] = {"ms": round(e0.elapsed_time(e1), 4), "own_slot_equal": good}
            cg.close()
            print(json.dumps({"c_abi_gather": report, "rank": rank, "world": world}), file=sys.stderr, flush=True)
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
