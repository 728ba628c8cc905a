#!/usr/bin/env python3
"""Per-stage HIP-event times of the extractor alone on a batch of synthetic KITTI images (kernel A/B runs:
`ORBFE_FAST_VARIANT=1 python tools/stage_times.py`).  Prints one JSON line: ms per launch of each stage."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from refactored_orb_slam2_amd import _lib
if os.environ.get("ORBFE_AB_LIB"):   # an experiment build of tools/ab_build.sh
    _lib.LIB_PATH = os.path.join(_lib.CSRC, "_ab", "liborbfe_%s.so" % os.environ["ORBFE_AB_LIB"])
from refactored_orb_slam2_amd import ORBextractor, synth

F = int(os.environ.get("F", "256")); W, H = 1241, 376
dev = torch.device("cuda", 0)
imgs = torch.from_numpy(np.stack([p for p in synth.sequence(W, H, F, seq=0)])).to(dev)
ex = ORBextractor(2000, 1.2, 8, 20, 7, device=0)
if os.environ.get("BLUR_KIND"):   # 0 fused level chain (default), 1 resize chain + matrix-core blur, 2 resize chain + one LDS blur launch
    assert ex._L.orbfe_debug_blur_kernel(ex._h, int(os.environ["BLUR_KIND"])) == 0
cap = ex.max_keypoints(W, H)
k = torch.zeros((F, cap, 28), dtype=torch.uint8, device=dev); d = torch.zeros((F, cap, 32), dtype=torch.uint8, device=dev)
n = torch.zeros(F, dtype=torch.int32, device=dev)
s = torch.cuda.Stream(dev)
for _ in range(3):
    ex.extract_batch_device(imgs, k, d, n, stream=s)
torch.cuda.synchronize()
ex.profile(True); ex.stage_times(reset=True)
R = int(os.environ.get("R", "10"))
for _ in range(R):
    ex.extract_batch_device(imgs, k, d, n, stream=s)
torch.cuda.synchronize()
st = ex.stage_times()
out = {kk: round(v[0] / max(v[1] // 2 if kk == "pyramid" else v[1], 1), 4) for kk, v in st.items()}
out["sum"] = round(sum(out.values()), 4); out["kp"] = float(n.float().mean().item()); out["tag"] = os.environ.get("TAG", "")
print(json.dumps(out))
ex.close()
