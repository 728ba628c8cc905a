"""Times Tracking::SearchLocalPoints on the device (orbfe_search_local_points_batch_device): F KITTI-sized frames, each
with its own pose and local map (~2900 map points for 2000 features), everything resident in HBM.  Prints one JSON line.
Not the headline bench (that is bench.py); SURVEY 8(f) row 3 measurement."""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from refactored_orb_slam2_amd import synth  # noqa: E402
from refactored_orb_slam2_amd._lib import KP_DTYPE  # noqa: E402
from refactored_orb_slam2_amd.extractor import ORBextractor  # noqa: E402
from refactored_orb_slam2_amd.matcher import Matcher, make_frustum  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=128)
    ap.add_argument("--distinct", type=int, default=8)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    w, h, nf = 1241, 376, 2000
    ex = ORBextractor(nf)
    res = ex.extract_batch(synth.sequence(w, h, a.distinct, seq=3))
    ex.close()
    frs, maps = [], []
    for i, (k, d) in enumerate(res):
        R, t = synth.camera_pose(50 + i)
        fr = make_frustum(R, t, 718.856, 718.856, 607.1928, 185.2157, 386.1448, (0, w, 0, h), 1.2, 8)
        frs.append(fr); maps.append(synth.local_map(k, d, fr, 60 + i, n_extra=500))
    F = a.frames
    cap = max(len(k) for k, _ in res) + 8
    pcap = max(len(m) for m in maps) + 8
    kps = np.zeros((F, cap), KP_DTYPE); desc = np.zeros((F, cap, 32), np.uint8); n = np.zeros(F, np.int32)
    pts = np.zeros((F, pcap), maps[0].dtype); npts = np.zeros(F, np.int32); fru = np.zeros(F, frs[0].dtype)
    for f in range(F):
        k, d = res[f % a.distinct]
        kps[f, :len(k)] = k; desc[f, :len(k)] = d; n[f] = len(k)
        m = maps[f % a.distinct]
        pts[f, :len(m)] = m; npts[f] = len(m); fru[f] = frs[f % a.distinct][0]
    dev = lambda x: torch.from_numpy(x.view(np.uint8).reshape(x.shape + (-1,)) if x.dtype.names else x).cuda()
    t_kps, t_desc, t_n, t_pts, t_np, t_fr = dev(kps), dev(desc), dev(n), dev(pts), dev(npts), dev(fru)
    t_track = torch.zeros((F, pcap, 24), dtype=torch.uint8, device="cuda")
    t_blocked = torch.zeros((F, cap), dtype=torch.uint8, device="cuda")
    t_assigned = torch.full((F, cap), -1, dtype=torch.int32, device="cuda")
    t_ntm = torch.zeros(F, dtype=torch.int32, device="cuda"); t_nm = torch.zeros(F, dtype=torch.int32, device="cuda")
    s = torch.cuda.Stream()
    m = Matcher()

    def step():
        t_blocked.zero_(); t_assigned.fill_(-1)
        m.search_local_points_batch(t_kps, t_desc, t_n, None, (0, w, 0, h), t_fr, t_pts, t_np, 1.0, 0.8, t_track, t_blocked,
                                    t_assigned, t_ntm, t_nm, stream=s)

    with torch.cuda.stream(s):
        for _ in range(a.warmup):
            step()
        s.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(a.steps):
            step()
        e1.record(s)
        s.synchronize()
    ms = e0.elapsed_time(e1) / a.steps
    print(json.dumps({"metric": "SearchLocalPoints frames/s (isInFrustum + SearchByProjection, device-resident)",
                      "value": F / ms * 1e3, "unit": "frames/s", "ms_per_step": ms, "frames": F,
                      "points_per_frame": float(npts.mean()), "to_match_per_frame": float(t_ntm.float().mean()),
                      "matches_per_frame": float(t_nm.float().mean()), "points_per_s": float(npts.sum()) / ms * 1e3}))
    m.close()


if __name__ == "__main__":
    main()
