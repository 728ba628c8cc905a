#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: stereo frames/s of the ORB front end (extract + match).

One "step" = one batch of F synthetic KITTI-geometry stereo frames (1241x376, 2000 features, 8 levels)
resident in HBM, pushed through the whole hot path on one GPU:
    ORBextractor left + right  ->  Frame::ComputeStereoMatches  ->  Frame::UnprojectStereo of every stereo point
    ->  its projection into the next frame (ORBmatcher.cc:1270-1308)  ->  SearchByProjection(cur, last)
N > 1 shards independent frames over ranks (one process per GPU, weak scaling: F frames per rank per step)
and gathers the per-frame keypoint/descriptor records with one RCCL all_gather per step -- the only
exchange the path has (BASELINE.json north_star).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel, HIP
events on the launching stream) and, at N=1, `cpu_baseline` (the CPU oracle on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H, NFEAT, NLEVELS = 1241, 376, 2000, 8
MBF, FX = 386.1448, 718.856           # Source/Examples/Stereo/KITTI00-02.yaml: Camera.bf, Camera.fx
TH_STEREO = 7.0                       # Tracking::TrackWithMotionModel: th = 7 for stereo (Tracking.cc:793-798)
HBM_PEAK_GBS = 8000.0                 # MI355X_MICROARCH.md: 8 TB/s spec


def level_pixels(ex):
    return [ex.level_size(l, W, H) for l in range(NLEVELS)]


FY, CX, CY = 718.856, 607.1928, 185.2157   # KITTI00-02.yaml
SHIFT_X = -2.0                             # synth.sequence: the image content moves 2 px per frame


def camera_records(n, sf):
    """Synthetic camera for the motion-model search: identity pose, and a principal point that moves with the image
    content (SHIFT_X px per frame), so that re-projecting the last frame's stereo points into the current frame
    (UnprojectStereo -> Rcw*x3Dw+tcw -> pinhole) lands on the known image motion for every depth."""
    from refactored_orb_slam2_amd._lib import TRACK_POSE_DTYPE, UNPROJECT_CAM_DTYPE
    cams = np.zeros(n, UNPROJECT_CAM_DTYPE); poses = np.zeros(n, TRACK_POSE_DTYPE)
    eye = np.eye(3, dtype=np.float32).reshape(9)
    cams["Rwc"] = eye; cams["cx"] = CX; cams["cy"] = CY
    cams["invfx"] = np.float32(1) / np.float32(FX); cams["invfy"] = np.float32(1) / np.float32(FY)
    poses["Rcw"] = eye; poses["fx"] = FX; poses["fy"] = FY; poses["cx"] = np.float32(CX) + np.float32(SHIFT_X); poses["cy"] = CY
    poses["mbf"] = MBF; poses["max_x"] = W; poses["max_y"] = H; poses["th"] = TH_STEREO
    poses["scale_factors"] = np.asarray(sf, np.float32)[:8]
    return cams, poses


def cpu_baseline(sample_frames: int):
    """The CPU oracle (scalar C restatement of the reference path, 1 thread) on a bounded sample."""
    from refactored_orb_slam2_amd import synth
    from tests import oracle_lib as ol
    pairs = synth.sequence(W, H, sample_frames, seq=0, stereo=True)
    oL, oR = ol.OracleExtractor(NFEAT, 1.2, NLEVELS, 20, 7), ol.OracleExtractor(NFEAT, 1.2, NLEVELS, 20, 7)
    sf, isf = oL.scale_factors, oL.inv_scale_factors
    mb = MBF / FX
    cams, poses = camera_records(1, sf)
    prev = None
    t0 = time.perf_counter()
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(2)   # left and right extractor on two threads, as Frame::Frame does (Frame.cc:87-90)
    for (L, R) in pairs:
        fut = pool.submit(oR, R)   # ctypes releases the GIL inside the C oracle
        kL, dL = oL(L)
        kR, dR = fut.result()
        planesL = [oL.level_pixels(l) for l in range(NLEVELS)]
        planesR = [oR.level_pixels(l) for l in range(NLEVELS)]
        _, ur, depth = ol.compute_stereo_matches(kL, dL, kR, dR, planesL, planesR, sf, isf, MBF, mb)
        if prev is not None:
            q = ol.track_queries(poses[:1], prev)     # projection of the last frame's map points (ORBmatcher.cc:1270-1308)
            ol.OracleFrame(kL, dL, sf, 0, W, 0, H, ur).search_by_projection_frame(q, True)
        prev = ol.unproject_stereo(cams[:1], kL, dL, depth)   # Frame::UnprojectStereo for every stereo point
    dt = time.perf_counter() - t0
    pool.shutdown()
    return {"value": round(sample_frames / dt, 3), "unit": "frames/s", "cores": 2, "kind": "port",
            "sample": f"{sample_frames} synthetic KITTI-geometry stereo frames (left || right extraction on two threads as in "
                      f"Frame.cc:87-90, then stereo match + UnprojectStereo + SearchByProjection vs previous frame on one), "
                      f"oracle/orb_oracle.c -O2 scalar, {dt:.1f} s"}


def main():
    if os.environ.get("ORBFE_BENCH_WATCHDOG"):  # debugging aid: dump all stacks and exit if the run exceeds N seconds
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["ORBFE_BENCH_WATCHDOG"]), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=256, help="stereo frames per GPU per step")
    ap.add_argument("--cpu-sample", type=int, default=240, help="stereo frames timed on the CPU oracle (0 = skip)")
    ap.add_argument("--lr-streams", type=int, default=1, choices=(1, 2),
                    help="2: left/right extractors on two HIP streams (the reference uses two threads); 1: one stream")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from refactored_orb_slam2_amd import ORBextractor, synth
    from refactored_orb_slam2_amd.matcher import Matcher, track_queries_batch, unproject_stereo_batch
    from refactored_orb_slam2_amd.sharding import AsyncGather

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # rehearsal switches for boxes with fewer GPUs than ranks (not used by the driver): ORBFE_BENCH_SHARE_DEVICE=1 maps
    # every rank to the visible devices round-robin, ORBFE_BENCH_BACKEND=gloo replaces RCCL for the gather
    backend = os.environ.get("ORBFE_BENCH_BACKEND", "nccl")
    if os.environ.get("ORBFE_BENCH_SHARE_DEVICE") == "1":
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
    # ---- synthetic input, resident in HBM before the timed region (each rank its own sequence)
    pairs = synth.sequence(W, H, F, seq=rank, stereo=True)
    dL = torch.from_numpy(np.stack([p[0] for p in pairs])).to(dev)
    dR = torch.from_numpy(np.stack([p[1] for p in pairs])).to(dev)

    exL, exR = ORBextractor(NFEAT, 1.2, NLEVELS, 20, 7, device=local), ORBextractor(NFEAT, 1.2, NLEVELS, 20, 7, device=local)
    mt = Matcher(local)
    import atexit

    def _close_handles():  # release the library handles while the HIP runtime is still alive, also after an exception
        try:
            torch.cuda.synchronize()
        except Exception:
            pass
        for hnd in (exL, exR, mt):
            try:
                hnd.close()
            except Exception:
                pass

    atexit.register(_close_handles)
    cap = exL.max_keypoints(W, H)
    mk = lambda: (torch.zeros((F, cap, 28), dtype=torch.uint8, device=dev),
                  torch.zeros((F, cap, 32), dtype=torch.uint8, device=dev), torch.zeros(F, dtype=torch.int32, device=dev))
    kl, dl, nl = mk()
    kr, dr, nr = mk()
    ur = torch.zeros((F, cap), dtype=torch.float32, device=dev)
    depth = torch.zeros((F, cap), dtype=torch.float32, device=dev)
    n_stereo = torch.zeros(F, dtype=torch.int32, device=dev)
    blocked = torch.zeros((F, cap), dtype=torch.uint8, device=dev)
    assigned = torch.zeros((F, cap), dtype=torch.int32, device=dev)
    n_track = torch.zeros(F, dtype=torch.int32, device=dev)
    cams_np, poses_np = camera_records(F, exL.GetScaleFactors())
    t_cams = torch.from_numpy(cams_np.view(np.uint8).reshape(F, -1)).to(dev)
    t_poses = torch.from_numpy(poses_np.view(np.uint8).reshape(F, -1)).to(dev)
    pts = torch.zeros((F, cap, 60), dtype=torch.uint8, device=dev)      # orbfe_last_point records
    q = torch.zeros((F, cap, 68), dtype=torch.uint8, device=dev)        # orbfe_query records
    nq = torch.zeros(F, dtype=torch.int32, device=dev)
    mb = MBF / FX
    # three explicit HIP streams: torch's default stream is the NULL stream, which the C ABI reads as "use the
    # handle's own stream"; the whole step therefore runs on named streams ordered by events
    sM, sL, sR = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    evL, evR = torch.cuda.Event(), torch.cuda.Event()

    def step():
        with torch.cuda.stream(sM):
            _step(sM)

    def _step(cur):
        if args.lr_streams == 2:
            sL.wait_stream(cur); sR.wait_stream(cur)
            exL.extract_batch_device(dL, kl, dl, nl, stream=sL)   # ORBextractor left  (Frame.cc:87-90: two threads)
            exR.extract_batch_device(dR, kr, dr, nr, stream=sR)   # ORBextractor right
            evL.record(sL); evR.record(sR)
            cur.wait_event(evL); cur.wait_event(evR)
        else:
            exL.extract_batch_device(dL, kl, dl, nl, stream=cur)
            exR.extract_batch_device(dR, kr, dr, nr, stream=cur)
        mt.stereo_match(exL, exR, kl, dl, nl, kr, dr, nr, MBF, mb, ur, depth, n_stereo, stream=cur)  # ComputeStereoMatches
        unproject_stereo_batch(kl, dl, nl, depth, t_cams, 1, pts, cur)   # Frame::UnprojectStereo: the stereo points of every frame
        track_queries_batch(t_poses, pts, nl, 1, q, nq, cur)             # projected into the next frame (ORBmatcher.cc:1270-1308)
        blocked.zero_(); assigned.fill_(-1)
        mt.proj_match_batch(kl, dl, nl, ur, (0.0, float(W), 0.0, float(H)), q, nq, 1, 0.9, True, blocked, assigned,
                            n_track, stream=cur)              # SearchByProjection(cur, last, th=7)
        if world > 1:  # the path's only exchange: gather of the per-frame keypoint records (overlaps the next step)
            gatherer.launch(nl, kl, dl)

    gatherer = AsyncGather(nl, kl, dl) if world > 1 else None

    def barrier():
        if gatherer is not None:
            with torch.cuda.stream(sM):
                gatherer.wait()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1) if args.warmup >= 0 else 0):
        step()
    barrier()
    exL.device_status(); exR.device_status()
    n_kp = int(nl.sum().item()) + int(nr.sum().item())
    n_st = int(n_stereo.sum().item())
    n_tr = int(n_track.sum().item())
    # measured FAST candidates per image (for the algorithmic byte count of the FAST kernel)
    cand_per_img = float(np.mean([sum(len(exL.debug_candidates(i, l)[0]) for l in range(NLEVELS)) for i in range(min(F, 4))]))

    # all-stage event timing costs ~2 % of the step: a short untimed pass yields the stage breakdown (and names the
    # dominant kernel); inside the timed region only that kernel is bracketed by HIP events on its launch stream
    exL.profile(True); exR.profile(True)
    exL.stage_times(reset=True); exR.stage_times(reset=True)
    for _ in range(3):
        step()
    barrier()
    stage_all = {k: (a[0] + b[0], a[1] + b[1]) for (k, a), b in zip(exL.stage_times().items(), exR.stage_times().values())}
    stage_ms_all = {k: (v[0] / max(v[1] // 2 if k == "pyramid" else v[1], 1)) for k, v in stage_all.items()}
    dom = max(stage_ms_all, key=lambda k: stage_ms_all[k])
    exL.profile(True, [dom]); exR.profile(True, [dom])
    exL.stage_times(reset=True); exR.stage_times(reset=True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    stL, stR = exL.stage_times(), exR.stage_times()
    exL.profile(False); exR.profile(False)

    if rank == 0:
        px = level_pixels(exL)
        sumP = sum(w * h for w, h in px)
        P0, P7 = px[0][0] * px[0][1], px[-1][0] * px[-1][1]
        # algorithmic bytes per image and per kernel; one launch processes F images
        alg = {
            "pyramid": (sumP - P7) + (sumP - P0) + 2 * P0,   # level chain + the level-0 copy (read + write)
            "fast": sumP + 8 * cand_per_img,
            "octree": 8 * cand_per_img + 4 * NFEAT,
            "blur": 2 * sumP,
            "describe": NFEAT * (749 + 512 + 60),
        }
        per_launch_ms = dict(stage_ms_all)   # untimed 3-step pass (every stage)
        cnt = stL[dom][1] + stR[dom][1]
        if dom == "pyramid":
            cnt //= 2  # two timed groups (level-0 copy, resize chain) per batch
        per_launch_ms[dom] = (stL[dom][0] + stR[dom][0]) / max(cnt, 1)   # the dominant kernel: live, over the timed region
        achieved = alg[dom] * F / (max(per_launch_ms[dom], 1e-9) * 1e-3) / 1e9
        # HBM traffic of the dominant kernel from the committed PMC pass (FETCH_SIZE + WRITE_SIZE, separate rocprofv3
        # runs: profiles/r01_pmc_counters.md), scaled to this launch's image count; null if no pass covers the kernel
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_fast.json")))
            if pmc["kernel"].startswith(dom):
                traffic = int((pmc["fetch_kb"] * pmc.get("fetch_correction", 1.0) + pmc["write_kb"]) * 1024 * F / pmc["images_per_launch"])
        except Exception:
            traffic = None
        value = world * F * args.steps / dt
        out = {
            "metric": "frames/s (extract+match) at KITTI 1241×376, 2000 feat; 1/2/4/8 GPU + CPU ref",
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "kitti_stereo_1241x376_2000feat_8lvl: 2x ORBextractor + ComputeStereoMatches + "
                                   "SearchByProjection(cur,last)", "stereo_frames_per_gpu_per_step": F,
                       "images_per_step": 2 * F * world, "parallelism": f"frame-shard x{world}",
                       "keypoints_per_image": round(n_kp / (2 * F), 1), "stereo_matches_per_frame": round(n_st / F, 1),
                       "tracked_per_frame": round(n_tr / F, 1)},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_launch": int(alg[dom] * F), "avg_launch_ms": round(per_launch_ms[dom], 4),
                         "stage_ms_per_batch": {k: round(v, 4) for k, v in per_launch_ms.items()}},
        }
        if world == 1 and args.cpu_sample > 0:
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
