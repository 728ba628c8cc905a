#!/usr/bin/env python3
"""Sequence driver for the ORB front end on a KITTI-style stereo directory.

Mirrors what Source/Examples/Stereo/stereo_kitti.cc:36-210 does around the front end: reads `times.txt` and
`image_0/%06d.png`, `image_1/%06d.png` (LoadImages, :152-210) -- with the library's own zlib PNG reader, no PIL / OpenCV --
and pushes every stereo pair through the per-frame hot path
    ORBextractor left + right -> Frame::ComputeStereoMatches -> Frame::UnprojectStereo of the stereo points
    -> SearchByProjection(cur, last, th = 7) against the previous frame with the constant-velocity prediction Tcw = Tlw
       (what Tracking::TrackWithMotionModel searches with before the pose optimisation, L/src/Tracking.cc:780-805)
then prints the examples' timing report ("median tracking time" / "mean tracking time", :137-144) for the front end; pose
optimisation, local mapping and loop closing are out of scope.

  per-frame  (default)  one pair at a time, like the reference's loop
  --batch F             F pairs per launch through the device-resident batch API
  --shard               batched-sequence mode (C5): the frame range is cut into contiguous chunks, one per rank
                        (`python -m torch.distributed.run --nproc-per-node N examples/stereo_kitti.py ... --shard`), every rank
                        processes its chunk and the per-frame keypoint records are gathered with one RCCL all_gather per batch
  --dump FILE.npz       per-frame outputs (keypoints, descriptors, mvuRight, mvDepth, tracked assignments) for parity checks

usage: stereo_kitti.py <sequence_dir> [--features 2000] [--batch 64] [--max-frames N] [--bf 386.1448 --fx 718.856 ...]
"""
from __future__ import annotations

import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load_images(seq_dir: str):
    """LoadImages of stereo_kitti.cc:152-210."""
    times = [float(l) for l in open(os.path.join(seq_dir, "times.txt")) if l.strip()]
    left = [os.path.join(seq_dir, "image_0", f"{i:06d}.png") for i in range(len(times))]
    right = [os.path.join(seq_dir, "image_1", f"{i:06d}.png") for i in range(len(times))]
    return left, right, times


def read_gray(path: str, out: np.ndarray | None = None) -> np.ndarray:
    """8-bit grey PNG (or RGB, converted as cvtColor does) through liborbfe's zlib reader."""
    from refactored_orb_slam2_amd import _lib
    L = _lib.lib()
    w, h = C.c_int(0), C.c_int(0)
    if out is None:
        _lib.check(L.orbfe_png_info(path.encode(), C.byref(w), C.byref(h)), "orbfe_png_info")
        out = np.empty((h.value, w.value), np.uint8)
    _lib.check(L.orbfe_png_read_gray(path.encode(), out.ctypes.data_as(C.c_void_p), out.strides[0], out.shape[0], C.byref(w), C.byref(h)),
               "orbfe_png_read_gray")
    if (h.value, w.value) != out.shape:
        raise ValueError(f"{path}: {w.value}x{h.value}, expected {out.shape[1]}x{out.shape[0]}")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("sequence_dir")
    ap.add_argument("--features", type=int, default=2000)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--max-frames", type=int, default=0)
    ap.add_argument("--bf", type=float, default=386.1448)    # Source/Examples/Stereo/KITTI00-02.yaml
    ap.add_argument("--fx", type=float, default=718.856)
    ap.add_argument("--fy", type=float, default=718.856)
    ap.add_argument("--cx", type=float, default=607.1928)
    ap.add_argument("--cy", type=float, default=185.2157)
    ap.add_argument("--th", type=float, default=7.0)
    ap.add_argument("--shard", action="store_true")
    ap.add_argument("--dump", default="")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from refactored_orb_slam2_amd import ORBextractor, sharding
    from refactored_orb_slam2_amd._lib import KP_DTYPE, TRACK_POSE_DTYPE, UNPROJECT_CAM_DTYPE
    from refactored_orb_slam2_amd.matcher import Matcher, track_queries_batch, unproject_stereo_batch

    rank, world, local = 0, 1, 0
    if args.shard and "WORLD_SIZE" in os.environ:
        rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        ndev = max(torch.cuda.device_count(), 1)
        backend = os.environ.get("ORBFE_BACKEND", "nccl" if ndev >= world else "gloo")   # RCCL refuses two ranks on one device
        local %= ndev
        torch.cuda.set_device(local)
        dist.init_process_group(backend, **({"device_id": torch.device("cuda", local)} if backend == "nccl" else {}))
    dev = torch.device("cuda", local)

    left, right, times = load_images(args.sequence_dir)
    n_all = len(times) if args.max_frames <= 0 else min(len(times), args.max_frames)
    begin, end = sharding.shard_range(n_all, rank, world)
    chunk = sharding.padded_chunk(n_all, world)           # every rank contributes equally sized records to the gather
    if rank == 0:
        print(f"\n-------\nStart processing sequence ...\nImages in the sequence: {n_all}\n" +
              (f"sharded over {world} ranks, {chunk} frames per rank\n" if world > 1 else ""))
    exL, exR, mt = ORBextractor(args.features, device=local), ORBextractor(args.features, device=local), Matcher(local)
    first = read_gray(left[0])
    h, w = first.shape
    cap = exL.max_keypoints(w, h)
    sf = exL.GetScaleFactors()
    F = max(args.batch, 1)
    z = lambda *s, dt=torch.uint8: torch.zeros(s, dtype=dt, device=dev)
    kl, dl, nl = z(F, cap, 28), z(F, cap, 32), z(F, dt=torch.int32)
    kr, dr, nr = z(F, cap, 28), z(F, cap, 32), z(F, dt=torch.int32)
    ur, depth, nst = z(F, cap, dt=torch.float32), z(F, cap, dt=torch.float32), z(F, dt=torch.int32)
    pts, q, nq = z(F + 1, cap, 60), z(F, cap, 68), z(F, dt=torch.int32)       # slot 0 of pts = last frame of the previous batch
    npts = z(F + 1, dt=torch.int32)
    blocked, assigned, ntr = z(F, cap), z(F, cap, dt=torch.int32), z(F, dt=torch.int32)
    # constant-velocity prediction with zero velocity: the current pose equals the last one (identity in the last camera's frame)
    cams = np.zeros(F, UNPROJECT_CAM_DTYPE); poses = np.zeros(F, TRACK_POSE_DTYPE)
    eye = np.eye(3, dtype=np.float32).reshape(9)
    cams["Rwc"] = eye; cams["cx"] = args.cx; cams["cy"] = args.cy
    cams["invfx"] = np.float32(1) / np.float32(args.fx); cams["invfy"] = np.float32(1) / np.float32(args.fy)
    poses["Rcw"] = eye; poses["fx"] = args.fx; poses["fy"] = args.fy; poses["cx"] = args.cx; poses["cy"] = args.cy
    poses["mbf"] = args.bf; poses["max_x"] = w; poses["max_y"] = h; poses["th"] = args.th
    poses["scale_factors"][:, :len(sf)] = sf
    t_cams = torch.from_numpy(cams.view(np.uint8).reshape(F, -1)).to(dev)
    t_poses = torch.from_numpy(poses.view(np.uint8).reshape(F, -1)).to(dev)
    stream = torch.cuda.Stream(dev)
    track_times, n_kp, n_st, n_tr = [], 0, 0, 0
    dump = {}
    gathered = []
    imgsL = np.empty((F, h, w), np.uint8); imgsR = np.empty((F, h, w), np.uint8)
    have_prev = False
    for b in range(begin, begin + chunk, F):
        idx = list(range(b, min(b + F, end)))
        B = len(idx)
        for j, i in enumerate(idx):
            read_gray(left[i], imgsL[j]); read_gray(right[i], imgsR[j])
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            if B:
                dL = torch.from_numpy(imgsL[:B]).to(dev, non_blocking=True)
                dR = torch.from_numpy(imgsR[:B]).to(dev, non_blocking=True)
                exL.extract_batch_device(dL, kl[:B], dl[:B], nl[:B], stream=stream)
                exR.extract_batch_device(dR, kr[:B], dr[:B], nr[:B], stream=stream)
                mt.stereo_match(exL, exR, kl[:B], dl[:B], nl[:B], kr[:B], dr[:B], nr[:B], args.bf, args.bf / args.fx, ur[:B], depth[:B],
                                nst[:B], stream=stream)
                # the stereo points of every frame of the batch (slot j + 1), then frame j is searched with the points of slot j
                unproject_stereo_batch(kl[:B], dl[:B], nl[:B], depth[:B], t_cams[:B], 1, pts[1:B + 1], stream)
                npts[1:B + 1].copy_(nl[:B])
                track_queries_batch(t_poses[:B], pts[:B], npts[:B], 0, q[:B], nq[:B], stream)
                blocked[:B].zero_(); assigned[:B].fill_(-1)
                mt.proj_match_batch(kl[:B], dl[:B], nl[:B], ur[:B], (0.0, float(w), 0.0, float(h)), q[:B], nq[:B], 1, 0.9, True,
                                    blocked[:B], assigned[:B], ntr[:B], stream=stream)
                if not have_prev:   # the first frame of this rank's chunk has no predecessor
                    ntr[0] = 0; assigned[0].fill_(-1)
            if world > 1:
                if B < F:
                    nl[B:].zero_()   # padding frames of the last chunk carry no keypoints
                gathered.append(tuple(t.cpu() for t in sharding.gather_records(nl, kl, dl)))
        stream.synchronize()
        dt = time.perf_counter() - t0
        if B:
            track_times += [dt / B] * B
            n_kp += int(nl[:B].sum()); n_st += int(nst[:B].sum()); n_tr += int(ntr[:B].sum())
            if args.dump:
                for j, i in enumerate(idx):
                    n = int(nl[j])
                    dump[f"kp_{i}"] = kl[j, :n].cpu().numpy().reshape(-1).view(KP_DTYPE)
                    dump[f"desc_{i}"] = dl[j, :n].cpu().numpy()
                    dump[f"ur_{i}"] = ur[j, :n].cpu().numpy(); dump[f"depth_{i}"] = depth[j, :n].cpu().numpy()
                    dump[f"assigned_{i}"] = assigned[j, :n].cpu().numpy(); dump[f"ntrack_{i}"] = np.int32(int(ntr[j]))
            with torch.cuda.stream(stream):   # the last frame of this batch becomes slot 0 for the next one
                pts[0].copy_(pts[B]); npts[0:1].copy_(npts[B:B + 1])
            have_prev = True
    if world > 1:
        stats = torch.tensor([n_kp, n_st, n_tr, len(track_times), sum(track_times)], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(stats)
        n_kp, n_st, n_tr = (int(stats[i]) for i in range(3))
    if args.dump:
        if world > 1:   # the gathered records: frame order = rank order x batch order (contiguous shards)
            F_ = F
            for r in range(world):
                rb, re_ = sharding.shard_range(n_all, r, world)
                for bi, (gn, gk, gd) in enumerate(gathered):
                    for j in range(F_):
                        i = rb + bi * F_ + j
                        if i < re_:
                            n = int(gn[r * F_ + j])
                            dump[f"g_kp_{i}"] = gk[r * F_ + j, :n].numpy().reshape(-1).view(KP_DTYPE)
                            dump[f"g_desc_{i}"] = gd[r * F_ + j, :n].numpy()
        np.savez_compressed(args.dump if world == 1 else f"{args.dump}.rank{rank}.npz", **dump)
    if rank == 0:
        track_times.sort()
        n = n_all
        print("-------\n")
        print(f"median tracking time: {track_times[len(track_times) // 2]}")
        print(f"mean tracking time: {sum(track_times) / len(track_times)}")
        print(f"frames: {n}, keypoints/left image: {n_kp / n:.1f}, stereo matches/frame: {n_st / n:.1f}, tracked/frame: {n_tr / max(n - world, 1):.1f}, "
              f"front-end frames/s of rank 0 (incl. H2D, excl. PNG decoding): {len(track_times) / sum(track_times):.1f}")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for hnd in (exL, exR, mt):
        hnd.close()


if __name__ == "__main__":
    main()
