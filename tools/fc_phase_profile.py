"""Per-phase wave-cycle breakdown of fast_cells_kernel.  Needs a library built with -DFC_TIMING=1:
   make -C refactored_orb_slam2_amd/csrc clean && make -C refactored_orb_slam2_amd/csrc EXTRA=-DFC_TIMING=1   (then rebuild without it)."""
import ctypes as C, numpy as np, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from refactored_orb_slam2_amd import _lib
if os.environ.get("ORBFE_AB_LIB"): _lib.LIB_PATH = os.path.join(_lib.CSRC, "_ab", "liborbfe_%s.so" % os.environ["ORBFE_AB_LIB"])
from refactored_orb_slam2_amd import ORBextractor, synth
L=_lib.lib()
W,H,NF,B=1241,376,2000,256
imgs=[synth.sequence(W,H,1,seq=100+i)[0] for i in range(8)]
d=torch.from_numpy(np.stack([imgs[i%8] for i in range(B)])).cuda()
ex=ORBextractor(NF,device=0)
cap=ex.max_keypoints(W,H)
k=torch.zeros(B,cap,28,dtype=torch.uint8,device='cuda'); de=torch.zeros(B,cap,32,dtype=torch.uint8,device='cuda'); n=torch.zeros(B,dtype=torch.int32,device='cuda')
for _ in range(2): ex.extract_batch_device(d,k,de,n)
ex.sync()
out=(C.c_ulonglong*8)()
L.orbfe_debug_fc_profile(out,1)
R=5
for _ in range(R): ex.extract_batch_device(d,k,de,n)
ex.sync()
L.orbfe_debug_fc_profile(out,0)
v=list(out)
waves=v[6]; cells=v[7]
names=["wait pixels","stage+clear","A1 quick test","A2 score","B NMS","C scan+emit"]
tot=sum(v[:6])
print("waves",waves/R,"cells",cells/R,"cycles/wave",tot/waves)
for nme,x in zip(names,v[:6]): print(f"{nme:14s} {x/cells:9.0f} cycles/cell  {100*x/tot:5.1f}%")

out=(C.c_ulonglong*8)()
L.orbfe_debug_od_profile(out,1)
for _ in range(R): ex.extract_batch_device(d,k,de,n)
ex.sync()
L.orbfe_debug_od_profile(out,0)
v=list(out); waves=v[6]; kps=v[7]; tot=sum(v[:6])
print("describe8: waves",waves/R,"keypoints",kps/R,"cycles/wave",tot/max(waves,1))
for nme,x in zip(["tables+bookkeeping","wait raw patch","moments","angle sincos","wait blurred patch","BRIEF+stores"],v[:6]): print(f"{nme:20s} {x/max(kps,1):9.0f} cycles/keypoint  {100*x/max(tot,1):5.1f}%")

# ---- octree_select_kernel: cycles and subdivision iterations per level
out64=(C.c_ulonglong*256)()
L.orbfe_debug_oct_profile(out64,1)
for _ in range(R): ex.extract_batch_device(d,k,de,n)
ex.sync()
L.orbfe_debug_oct_profile(out64,0)
print("octree: level, cycles per workgroup, loop iterations per workgroup")
for l in range(8): print(f"  level {l}: {out64[l]/(R*B):9.0f} cycles  {out64[32+l]/(R*B):5.1f} iterations")
