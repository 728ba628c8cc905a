"""Per-phase wave-cycle breakdown of fast_cells_kernel and orient_describe8_kernel INSIDE the overlapped bench step (bench.py's
StepRig: 256 stereo frames, three handle sets, left | right extractor on two streams, the matching half beside them) and, for
comparison, with every kernel alone on the chip (one stream, one set).  Needs a -DFC_TIMING=1 build:
  tools/ab_build.sh timing "-DFC_TIMING=1" extract_kernels.hip;  ORBFE_AB_LIB=timing python tools/step_phase_profile.py"""
import ctypes as C, os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch
from refactored_orb_slam2_amd import _lib
if os.environ.get("ORBFE_AB_LIB"): _lib.LIB_PATH = os.path.join(_lib.CSRC, "_ab", "liborbfe_%s.so" % os.environ["ORBFE_AB_LIB"])
import bench
L = _lib.lib()
F = int(os.environ.get("F", "256")); STEPS = int(os.environ.get("STEPS", "30"))
FC = ["wait pixels", "stage+clear", "A1 quick test", "A2 score", "B NMS", "C scan+emit"]
OD = ["tables+bookkeeping", "wait raw patch", "moments", "angle sincos", "wait blurred patch", "BRIEF+stores"]


def run(n_sets, lr):
    rig = bench.StepRig(bench.CONFIGS["kitti_stereo"], F, n_sets=n_sets, lr_streams=lr)
    for _ in range(6): rig.step()
    rig.barrier()
    out = (C.c_ulonglong * 8)()
    L.orbfe_debug_fc_profile(out, 1); L.orbfe_debug_od_profile(out, 1)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    rig.barrier(); t0.record(rig.sM)
    for _ in range(STEPS): rig.step()
    rig.barrier(); t1.record(rig.sM); torch.cuda.synchronize()
    print(f"sets {n_sets}, L|R streams {lr}: {t0.elapsed_time(t1) / STEPS:.4f} ms per step (timing build: slower than the product build)")
    for name, fn, names, unit in (("fast_cells", L.orbfe_debug_fc_profile, FC, "cell"), ("orient_describe8", L.orbfe_debug_od_profile, OD, "keypoint")):
        fn(out, 0)
        v = list(out); tot = sum(v[:6]); waves = max(v[6], 1); units = max(v[7], 1)
        print(f"  {name}: {tot / waves:9.0f} cycles per wave, {tot / units:8.0f} per {unit}")
        for nme, x in zip(names, v[:6]): print(f"    {nme:20s} {x / units:9.0f} cycles/{unit}  {100 * x / max(tot, 1):5.1f} %")
    rig.close()


run(1, 1)
run(3, 2)
