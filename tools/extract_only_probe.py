#!/usr/bin/env python3
"""What the matching half costs inside the overlapped step: bench.StepRig's step with and without it (extraction of both eyes on
two streams, three sets in turn either way).  Prints ms per 256-frame step for: full step, extraction only, matching half only."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from refactored_orb_slam2_amd.matcher import track_queries_batch, unproject_stereo_batch

rig = bench.StepRig(bench.CONFIGS["kitti_stereo"], 256, n_sets=3, lr_streams=2)
W, H = rig.W, rig.H


def extract_only():
    xl, xr, m, B, eL, eR, eT = rig.pipe_sets[rig.pipe_k % rig.n_sets]
    rig.pipe_k += 1
    rig.sL.wait_event(eT); rig.sR.wait_event(eT)
    xl.extract_batch_device(B.dL, B.kl, B.dl, B.nl, stream=rig.sL)
    xr.extract_batch_device(B.dR, B.kr, B.dr, B.nr, stream=rig.sR)
    eL.record(rig.sL); eR.record(rig.sR)
    rig.sM.wait_event(eL); rig.sM.wait_event(eR)
    eT.record(rig.sM)


def match_only():
    xl, xr, m, B, eL, eR, eT = rig.pipe_sets[rig.pipe_k % rig.n_sets]
    rig.pipe_k += 1
    sM, cfg = rig.sM, rig.cfg
    m.stereo_match(xl, xr, B.kl, B.dl, B.nl, B.kr, B.dr, B.nr, cfg["bf"], rig.mb, B.ur, B.depth, B.n_stereo, stream=sM)
    unproject_stereo_batch(B.kl, B.dl, B.nl, B.depth, rig.t_cams, 1, B.pts, sM)
    track_queries_batch(rig.t_poses, B.pts, B.nl, 1, B.q, B.nq, sM)
    B.blocked.zero_(); B.assigned.fill_(-1)
    m.proj_match_batch(B.kl, B.dl, B.nl, B.ur, (0.0, float(W), 0.0, float(H)), B.q, B.nq, 1, 0.9, True, B.blocked, B.assigned, B.n_track, stream=sM)


def timed(fn, n=150):
    with torch.cuda.stream(rig.sM):
        for _ in range(6):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(6):
    rig.step()
rig.barrier()
for name, fn in (("full step", rig._step_piped), ("extraction only", extract_only), ("matching half only", match_only), ("full step", rig._step_piped)):
    print(f"{name:20s} {timed(fn):.4f} ms per step")
rig.close()
