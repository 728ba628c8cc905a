#!/usr/bin/env python3
"""The whole bench step with TWO handle / buffer sets: extraction of step k + 1 (left and right extractor on two streams) beside the
matching half of step k (stereo match, unproject, track queries, projection search) on a third stream.  Prints frames/s of the
plain step (one set, bench.py's layout) and of the pipelined one."""
import json, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from refactored_orb_slam2_amd import ORBextractor, synth
from refactored_orb_slam2_amd.matcher import Matcher, track_queries_batch, unproject_stereo_batch

cfg = bench.CONFIGS["kitti_stereo"]
W, H, NF, F = cfg["w"], cfg["h"], cfg["nfeat"], int(os.environ.get("F", "256"))
dev = torch.device("cuda", 0)
data = synth.sequence(W, H, F, seq=0, stereo=True)
PITCH = (W + 63) // 64 * 64
def pitched(imgs):
    t = torch.zeros((F, H, PITCH), dtype=torch.uint8)
    t[:, :, :W] = torch.from_numpy(np.stack(imgs))
    return t.to(dev)
dLf, dRf = pitched([p[0] for p in data]), pitched([p[1] for p in data])
z = lambda *s, dt=torch.uint8: torch.zeros(s, dtype=dt, device=dev)

class Set:
    def __init__(self):
        self.exL, self.exR, self.mt = ORBextractor(NF, device=0), ORBextractor(NF, device=0), Matcher(0)
        cap = self.exL.max_keypoints(W, H)
        self.kl, self.dl, self.nl = z(F, cap, 28), z(F, cap, 32), z(F, dt=torch.int32)
        self.kr, self.dr, self.nr = z(F, cap, 28), z(F, cap, 32), z(F, dt=torch.int32)
        self.ur, self.depth, self.n_stereo = z(F, cap, dt=torch.float32), z(F, cap, dt=torch.float32), z(F, dt=torch.int32)
        self.blocked, self.assigned, self.n_track = z(F, cap), z(F, cap, dt=torch.int32), z(F, dt=torch.int32)
        self.pts, self.q, self.nq = z(F, cap, 60), z(F, cap, 68), z(F, dt=torch.int32)
        self.evL, self.evR, self.evT = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()

NSETS = int(os.environ.get("NSETS", "2"))
sets = [Set() for _ in range(NSETS)]
cams_np, poses_np = bench.camera_records(F, sets[0].exL.GetScaleFactors(), cfg)
t_cams = torch.from_numpy(cams_np.view(np.uint8).reshape(F, -1)).to(dev)
t_poses = torch.from_numpy(poses_np.view(np.uint8).reshape(F, -1)).to(dev)
mb = cfg["bf"] / cfg["fx"]
sM, sL, sR = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)
dL, dR = dLf[:, :, :W], dRf[:, :, :W]

def extract(S):
    S.exL.extract_batch_device(dL, S.kl, S.dl, S.nl, stream=sL)
    S.exR.extract_batch_device(dR, S.kr, S.dr, S.nr, stream=sR)
    S.evL.record(sL); S.evR.record(sR)

def tail(S):
    with torch.cuda.stream(sM):
        sM.wait_event(S.evL); sM.wait_event(S.evR)
        S.mt.stereo_match(S.exL, S.exR, S.kl, S.dl, S.nl, S.kr, S.dr, S.nr, cfg["bf"], mb, S.ur, S.depth, S.n_stereo, stream=sM)
        unproject_stereo_batch(S.kl, S.dl, S.nl, S.depth, t_cams, 1, S.pts, sM)
        track_queries_batch(t_poses, S.pts, S.nl, 1, S.q, S.nq, sM)
        S.blocked.zero_(); S.assigned.fill_(-1)
        S.mt.proj_match_batch(S.kl, S.dl, S.nl, S.ur, (0.0, float(W), 0.0, float(H)), S.q, S.nq, 1, 0.9, True, S.blocked, S.assigned,
                              S.n_track, stream=sM)
        S.evT.record(sM)

def plain(k):      # bench.py: the next extraction waits for the whole step before it
    S = sets[0]
    sL.wait_stream(sM); sR.wait_stream(sM)
    extract(S); tail(S)

def piped(k):      # the next extraction only waits for the matching half that last read ITS set (two steps back)
    S = sets[k % NSETS]
    sL.wait_event(S.evT); sR.wait_event(S.evT)
    extract(S); tail(S)

# fully independent sets: each has its own three streams and runs its steps back to back; the chip sees NSETS steps in flight
own = [(torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)) for _ in range(NSETS)]

def indep(k):
    S = sets[k % NSETS]
    m, l, r = own[k % NSETS]
    l.wait_event(S.evT); r.wait_event(S.evT)
    S.exL.extract_batch_device(dL, S.kl, S.dl, S.nl, stream=l)
    S.exR.extract_batch_device(dR, S.kr, S.dr, S.nr, stream=r)
    S.evL.record(l); S.evR.record(r)
    with torch.cuda.stream(m):
        m.wait_event(S.evL); m.wait_event(S.evR)
        S.mt.stereo_match(S.exL, S.exR, S.kl, S.dl, S.nl, S.kr, S.dr, S.nr, cfg["bf"], mb, S.ur, S.depth, S.n_stereo, stream=m)
        unproject_stereo_batch(S.kl, S.dl, S.nl, S.depth, t_cams, 1, S.pts, m)
        track_queries_batch(t_poses, S.pts, S.nl, 1, S.q, S.nq, m)
        S.blocked.zero_(); S.assigned.fill_(-1)
        S.mt.proj_match_batch(S.kl, S.dl, S.nl, S.ur, (0.0, float(W), 0.0, float(H)), S.q, S.nq, 1, 0.9, True, S.blocked, S.assigned,
                              S.n_track, stream=m)
        S.evT.record(m)

def run(fn, K):
    for k in range(4): fn(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K): fn(k)
    torch.cuda.synchronize()
    return F * K / (time.perf_counter() - t0)

K = int(os.environ.get("K", "100"))
for S in sets: S.evT.record(sM)
res = {}
for name, fn in (("plain", plain), ("piped", piped), ("indep", indep), ("piped2", piped), ("indep2", indep)):
    res[name] = round(run(fn, K), 1)
res["n_track"] = [int(S.n_track.sum().item()) for S in sets]
print(json.dumps(res))
