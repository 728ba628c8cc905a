#!/usr/bin/env python3
"""The C ABI's pipeline handle (orbfe_pipeline_*) at its kernels' rate: frames uploaded once, then orbfe_pipeline_submit_resident in a
ring of slots.  SLOTS (default 3), F (256), K chunks (60), MASK (orbfe_pipeline_config.output_mask: 0 = every block copied out,
16 = counts only), LAYOUT (orbfe_debug_pipeline_streams: the stream -> priority / creation-order layouts of csrc/pipeline.cpp)."""
import json, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from refactored_orb_slam2_amd import _lib
if os.environ.get("ORBFE_AB_LIB"): _lib.LIB_PATH = os.path.join(_lib.CSRC, "_ab", "liborbfe_%s.so" % os.environ["ORBFE_AB_LIB"])
from refactored_orb_slam2_amd import synth
from refactored_orb_slam2_amd.pipeline import StereoPipeline
W, H, NF = 1241, 376, 2000
F, S, K = int(os.environ.get("F", "256")), int(os.environ.get("SLOTS", "3")), int(os.environ.get("K", "60"))
MASK, LAYOUT = int(os.environ.get("MASK", "0")), int(os.environ.get("LAYOUT", "-1"))
pairs = synth.sequence(W, H, 16, seq=0, stereo=True)
if LAYOUT >= 0:
    assert _lib.lib().orbfe_debug_pipeline_streams(LAYOUT) == 0
with StereoPipeline(W, H, F, 718.856, 718.856, 607.1928, 185.2157, 386.1448, 7.0, n_features=NF, slots=S, output_mask=MASK) as p:
    for s in range(S):
        for j in range(F):
            p.left(s)[j, :, :W] = pairs[j % 16][0]; p.right(s)[j, :, :W] = pairs[j % 16][1]
        p.submit(s, F, has_predecessor=s > 0)
    for s in range(S):
        p.wait(s)
    res = {}
    for name, fn in (("upload", p.submit), ("resident", p.submit_resident)):
        t0 = time.perf_counter()
        for k in range(K):
            s = k % S
            p.wait(s)
            fn(s, F, has_predecessor=True)
        for s in range(S):
            p.wait(s)
        res[name + "_frames_per_s"] = round(F * K / (time.perf_counter() - t0), 1)
    res["keypoints_frame0"] = int(p.output(0)["n_left"][0]); res["tracked_frame1"] = int(p.output(0)["n_tracked"][1])
    res["slots"] = S; res["batch"] = F; res["mask"] = MASK; res["layout"] = LAYOUT
print(json.dumps(res))
