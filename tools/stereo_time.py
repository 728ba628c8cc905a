import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from refactored_orb_slam2_amd import _lib
if os.environ.get("ORBFE_AB_LIB"): _lib.LIB_PATH = os.path.join(_lib.CSRC, "_ab", "liborbfe_%s.so" % os.environ["ORBFE_AB_LIB"])
from refactored_orb_slam2_amd import ORBextractor, synth
from refactored_orb_slam2_amd.matcher import Matcher
W,H,NF,B=1241,376,2000,256
pairs=synth.sequence(W,H,8,seq=5,stereo=True)
Lt=torch.from_numpy(np.stack([pairs[i%8][0] for i in range(B)])).cuda(); Rt=torch.from_numpy(np.stack([pairs[i%8][1] for i in range(B)])).cuda()
ex=ORBextractor(NF,device=0); exR=ORBextractor(NF,device=0); mt=Matcher(0)
cap=ex.max_keypoints(W,H)
z=lambda *s,dt=torch.uint8: torch.zeros(s,dtype=dt,device='cuda')
k,de,n=z(B,cap,28),z(B,cap,32),z(B,dt=torch.int32); kr,dr,nr=z(B,cap,28),z(B,cap,32),z(B,dt=torch.int32)
ur,dp,ns=z(B,cap,dt=torch.float32),z(B,cap,dt=torch.float32),z(B,dt=torch.int32)
s=torch.cuda.Stream()
ex.extract_batch_device(Lt,k,de,n,stream=s); exR.extract_batch_device(Rt,kr,dr,nr,stream=s); torch.cuda.synchronize()
for _ in range(3): mt.stereo_match(ex,exR,k,de,n,kr,dr,nr,386.1448,386.1448/718.856,ur,dp,ns,stream=s)
torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
R=20
with torch.cuda.stream(s):
    e0.record(s)
    for _ in range(R): mt.stereo_match(ex,exR,k,de,n,kr,dr,nr,386.1448,386.1448/718.856,ur,dp,ns,stream=s)
    e1.record(s)
torch.cuda.synchronize()
print(json.dumps({"tag":os.environ.get("TAG",""),"stereo_ms":round(e0.elapsed_time(e1)/R,4),"matched":float(ns.float().mean()), "chk": float(ur.double().sum())}))
