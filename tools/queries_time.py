"""HIP-event time of the search-query preparation of the bench step (256 KITTI stereo frames): one pass
(orbfe_track_queries_stereo_device) and two passes (orbfe_unproject_stereo_device + orbfe_track_queries_device), each alone on the chip.
A/B builds: ORBFE_AB_LIB=<name> (tools/ab_build.sh <name> "<flags>" frustum_kernels.hip)."""
import json, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch
from refactored_orb_slam2_amd import _lib
if os.environ.get("ORBFE_AB_LIB"): _lib.LIB_PATH = os.path.join(_lib.CSRC, "_ab", "liborbfe_%s.so" % os.environ["ORBFE_AB_LIB"])
import bench
rig = bench.StepRig(bench.CONFIGS["kitti_stereo"], 256, n_sets=1, lr_streams=1)
for _ in range(3): rig.step()
rig.barrier()
B, s, R, out = rig.B0, rig.sM, 50, {"tag": os.environ.get("ORBFE_AB_LIB", "")}
for name, fused in (("one_pass_ms", True), ("two_pass_ms", False)):
    rig.fused_queries = fused
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s):
        for _ in range(5): rig._queries(B, s)
        e0.record(s)
        for _ in range(R): rig._queries(B, s)
        e1.record(s)
    torch.cuda.synchronize()
    out[name] = round(e0.elapsed_time(e1) / R, 4)
    out[name.replace("_ms", "_sum")] = int(B.q.to(torch.int64).sum())
print(json.dumps(out))
rig.close()
