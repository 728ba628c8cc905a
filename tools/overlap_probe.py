#!/usr/bin/env python3
"""Do the latency-bound matcher kernels hide under the extractor when they run on a second stream?  Extraction of 256 KITTI
images (stream A) and the projection search of 256 frames against themselves (grid_build + proj_candidates + proj_resolve,
stream B): each alone, back to back on one stream, and concurrently on two."""
import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from refactored_orb_slam2_amd import _lib
if os.environ.get("ORBFE_AB_LIB"): _lib.LIB_PATH = os.path.join(_lib.CSRC, "_ab", "liborbfe_%s.so" % os.environ["ORBFE_AB_LIB"])
from refactored_orb_slam2_amd import ORBextractor, synth
from refactored_orb_slam2_amd.matcher import Matcher

W, H, NF, B = 1241, 376, 2000, 256
imgs = synth.sequence(W, H, 8, seq=5)
It = torch.from_numpy(np.stack([imgs[i % 8] for i in range(B)])).cuda()
ex = ORBextractor(NF, device=0); mt = Matcher(0)
cap = ex.max_keypoints(W, H)
z = lambda *s, dt=torch.uint8: torch.zeros(s, dtype=dt, device="cuda")
k, de, n = z(B, cap, 28), z(B, cap, 32), z(B, dt=torch.int32)
k2, de2, n2 = z(B, cap, 28), z(B, cap, 32), z(B, dt=torch.int32)   # the search's own copy: the extractor rewrites k / de / n
sA, sB = torch.cuda.Stream(), torch.cuda.Stream(priority=int(os.environ.get("PROBE_PRIORITY", "0")))   # -1: high priority
ex.extract_batch_device(It, k, de, n, stream=sA); torch.cuda.synchronize()
k2.copy_(k); de2.copy_(de); n2.copy_(n)
kn = k.cpu().numpy().view(_lib.KP_DTYPE).reshape(B, cap)
sf = ex.GetScaleFactors()
q = np.zeros((B, cap), _lib.QUERY_DTYPE)
q["u"], q["v"] = kn["x"] + 2.0, kn["y"]; q["u_r"] = q["u"] - 10
q["radius"] = np.float32(7.0) * sf[np.clip(kn["octave"], 0, 7)]
q["min_level"], q["max_level"] = kn["octave"] - 1, kn["octave"] + 1
q["valid"] = 1; q["blocks"] = 1; q["angle"] = kn["angle"]; q["desc"] = de.cpu().numpy()
tq = torch.from_numpy(q.view(np.uint8).reshape(B, cap, -1)).cuda()
blocked, assigned, nm = z(B, cap), z(B, cap, dt=torch.int32), z(B, dt=torch.int32)

def extract(s): ex.extract_batch_device(It, k, de, n, stream=s)
def search(s):
    with torch.cuda.stream(s):
        blocked.zero_(); assigned.fill_(-1)
    mt.proj_match_batch(k2, de2, n2, None, (0.0, float(W), 0.0, float(H)), tq, n2, 1, 0.9, True, blocked, assigned, nm, stream=s)

def timed(fn, R=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(sA)
    for _ in range(R): fn()
    sA.wait_stream(sB); e1.record(sA)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / R

def both():
    sB.wait_stream(sA)      # the round starts together
    extract(sA); search(sB)
    sA.wait_stream(sB)

res = {"extract_ms": round(timed(lambda: extract(sA)), 4), "search_ms": round(timed(lambda: search(sA)), 4),
       "one_stream_ms": round(timed(lambda: (extract(sA), search(sA))), 4), "two_streams_ms": round(timed(both), 4),
       "matches_per_frame": float(nm.float().mean()), "search_stream_priority": int(os.environ.get("PROBE_PRIORITY", "0"))}
print(json.dumps(res))
