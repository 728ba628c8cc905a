#!/usr/bin/env python3
"""Sequence driver for the ORB front end on a KITTI-style stereo directory.

Mirrors what Source/Examples/Stereo/stereo_kitti.cc:36-210 does around the front end: reads `times.txt` and
`image_0/%06d.png`, `image_1/%06d.png` (LoadImages, :152-210), pushes every stereo pair through
ORBextractor (left/right) + Frame::ComputeStereoMatches, and prints the examples' timing report
("median tracking time" / "mean tracking time", :137-144) -- here for the front end only, since the rest of
Tracking is out of scope.  Two modes:

  per-frame  (default)  one pair at a time through the synchronous host API, like the reference's loop
  --batch F             F pairs per launch through the device-resident batch API (frames/s of the hot path)

usage: stereo_kitti.py <sequence_dir> [--features 2000] [--batch 64] [--max-frames N] [--bf 386.1448 --fx 718.856]
"""
from __future__ import annotations

import argparse
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


def read_gray(path: str) -> np.ndarray:
    from PIL import Image
    im = Image.open(path)
    if im.mode != "L":
        im = im.convert("L")  # cvtColor(..., CV_RGB2GRAY) of Tracking::GrabImageStereo for colour input
    return np.ascontiguousarray(np.asarray(im, dtype=np.uint8))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("sequence_dir")
    ap.add_argument("--features", type=int, default=2000)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--max-frames", type=int, default=0)
    ap.add_argument("--bf", type=float, default=386.1448)
    ap.add_argument("--fx", type=float, default=718.856)
    args = ap.parse_args()

    import torch
    from refactored_orb_slam2_amd import ORBextractor
    from refactored_orb_slam2_amd.matcher import Matcher

    left, right, times = load_images(args.sequence_dir)
    n = len(times) if args.max_frames <= 0 else min(len(times), args.max_frames)
    print(f"\n-------\nStart processing sequence ...\nImages in the sequence: {n}\n")
    exL, exR, mt = ORBextractor(args.features), ORBextractor(args.features), Matcher()
    first = read_gray(left[0])
    h, w = first.shape
    cap = exL.max_keypoints(w, h)
    F = max(args.batch, 1)
    dev = "cuda"
    mk = lambda: (torch.zeros((F, cap, 28), dtype=torch.uint8, device=dev), torch.zeros((F, cap, 32), dtype=torch.uint8, device=dev),
                  torch.zeros(F, dtype=torch.int32, device=dev))
    kl, dl, nl = mk(); kr, dr, nr = mk()
    ur = torch.zeros((F, cap), dtype=torch.float32, device=dev); depth = torch.zeros_like(ur)
    nst = torch.zeros(F, dtype=torch.int32, device=dev)
    stream = torch.cuda.Stream()
    track_times, n_kp, n_st = [], 0, 0
    for b in range(0, n, F):
        idx = list(range(b, min(b + F, n)))
        imgsL = np.stack([read_gray(left[i]) for i in idx])
        imgsR = np.stack([read_gray(right[i]) for i in idx])
        t0 = time.perf_counter()
        with torch.cuda.stream(stream):
            dL = torch.from_numpy(imgsL).to(dev, non_blocking=True)
            dR = torch.from_numpy(imgsR).to(dev, non_blocking=True)
            B = len(idx)
            exL.extract_batch_device(dL, kl[:B], dl[:B], nl[:B], stream=stream)
            exR.extract_batch_device(dR, kr[:B], dr[:B], nr[:B], stream=stream)
            mt.stereo_match(exL, exR, kl[:B], dl[:B], nl[:B], kr[:B], dr[:B], nr[:B], args.bf, args.bf / args.fx, ur[:B], depth[:B],
                            nst[:B], stream=stream)
        stream.synchronize()
        dt = time.perf_counter() - t0
        track_times += [dt / len(idx)] * len(idx)
        n_kp += int(nl[:B].sum()); n_st += int(nst[:B].sum())
    track_times.sort()
    print("-------\n")
    print(f"median tracking time: {track_times[len(track_times) // 2]}")
    print(f"mean tracking time: {sum(track_times) / len(track_times)}")
    print(f"frames: {n}, keypoints/left image: {n_kp / n:.1f}, stereo matches/frame: {n_st / n:.1f}, "
          f"front-end frames/s (incl. H2D): {n / sum(track_times):.1f}")


if __name__ == "__main__":
    main()
