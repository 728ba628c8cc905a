"""Per-phase cycle breakdown of octree_select_kernel, per level, on one of bench.py's content kinds (default uniform_noise).
Needs a library built with -DFC_TIMING=1 (tools/ab_build.sh timing "-DFC_TIMING=1" extract_kernels.hip; ORBFE_AB_LIB=timing)."""
import ctypes as C, numpy as np, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from refactored_orb_slam2_amd import _lib
if os.environ.get("ORBFE_AB_LIB"): _lib.LIB_PATH = os.path.join(_lib.CSRC, "_ab", "liborbfe_%s.so" % os.environ["ORBFE_AB_LIB"])
from refactored_orb_slam2_amd import ORBextractor, synth
L = _lib.lib()
kind = sys.argv[1] if len(sys.argv) > 1 else "uniform_noise"
W, H, NF, B = 1241, 376, 2000, int(os.environ.get("B", "128"))
rng = np.random.default_rng(7)
if kind == "uniform_noise":
    imgs = [rng.integers(0, 256, (H, W), dtype=np.uint8) for _ in range(8)]
elif kind == "sensor_noise":
    imgs = [np.clip(synth.sequence(W, H, 1, seq=100 + i)[0].astype(np.int16) + rng.integers(-8, 9, (H, W)), 0, 255).astype(np.uint8) for i in range(8)]
else:
    imgs = [synth.sequence(W, H, 1, seq=100 + i)[0] for i in range(8)]
d = torch.from_numpy(np.stack([imgs[i % 8] for i in range(B)])).cuda()
ex = ORBextractor(NF, device=0)
cap = ex.max_keypoints(W, H)
k = torch.zeros(B, cap, 28, dtype=torch.uint8, device='cuda'); de = torch.zeros(B, cap, 32, dtype=torch.uint8, device='cuda'); n = torch.zeros(B, dtype=torch.int32, device='cuda')
for _ in range(2): ex.extract_batch_device(d, k, de, n)
ex.sync()
out = (C.c_ulonglong * 256)()
L.orbfe_debug_oct_profile(out, 1)
R = 5
for _ in range(R): ex.extract_batch_device(d, k, de, n)
ex.sync()
L.orbfe_debug_oct_profile(out, 0)
names = ["offsets", "gather", "first relabel", "node work", "round sweeps", "best+out"]
print(kind, "octree: cycles per workgroup by level and phase (thread 0's clock)")
print("level   total  rounds " + " ".join(f"{x:>13s}" for x in names))
for l in range(8):
    print(f"{l:5d} {out[l]/(R*B):8.0f} {out[32+l]/(R*B):6.1f}  " + " ".join(f"{out[64+8*l+i]/(R*B):13.0f}" for i in range(6)))
