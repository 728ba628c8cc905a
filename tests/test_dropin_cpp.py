"""The C++ drop-in classes (csrc/host/): build everywhere, behave without a GPU, parity on the GPU."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "_build", "test_dropin")


def _build():
    from refactored_orb_slam2_amd import _lib
    from tests import oracle_lib as ol
    _lib.build()
    ol.build()
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "cpp")], check=True, capture_output=True)


def test_dropin_builds_and_fails_loudly_without_device():
    _build()
    import ctypes as C
    from refactored_orb_slam2_amd import _lib
    n = C.c_int(0)
    if _lib.lib().orbfe_device_count(C.byref(n)) == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    r = subprocess.run([EXE, "--nodevice"], capture_output=True, text=True)
    assert r.returncode == 0 and "nodevice ok" in r.stdout
    assert "no CPU fallback" in r.stderr  # logged, never thrown (SURVEY.md §8(b) Errors)


@pytest.mark.gpu
@pytest.mark.parametrize("geom", [(1241, 376, 2000), (640, 480, 1000)])
def test_dropin_parity_on_gpu(tmp_path, geom):
    from refactored_orb_slam2_amd import synth
    _build()
    w, h, nf = geom
    img = synth.frame(w, h, seq=13, f=2)
    p = str(tmp_path / "img.raw")
    img.tofile(p)
    r = subprocess.run([EXE, p, str(w), str(h), str(nf)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "dropin ok" in r.stdout and "SearchLocalPoints ok" in r.stdout and "SearchByBoW ok" in r.stdout and "SearchForTriangulation ok" in r.stdout and "ComputeBoW ok" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("geom", [(640, 480, 1000), (1241, 376, 2000)])
def test_matcher_dropin_all_eleven_methods_on_gpu(tmp_path, geom):
    """csrc/host/ORBmatcher.cc -- ORB_SLAM2::ORBmatcher with the reference's declarations -- linked against mock Frame /
    KeyFrame / MapPoint headers: the eleven methods, DescriptorDistance and the protected helpers vs the oracle, Fuse x2 and
    SearchBySim3 vs a literal sequential replay of the reference loops on a twin world."""
    from refactored_orb_slam2_amd import synth
    _build()
    w, h, nf = geom
    a, b = synth.sequence(w, h, 2, seq=17)
    pa, pb = str(tmp_path / "a.raw"), str(tmp_path / "b.raw")
    a.tofile(pa); b.tofile(pb)
    exe = os.path.join(ROOT, "tests", "cpp", "_build", "test_matcher_dropin")
    r = subprocess.run([exe, pa, pb, str(w), str(h), str(nf)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    for tag in ("SearchByProjection(F, points) ok", "SearchByProjection(cur, last) ok", "SearchByProjection(F, KF, found) ok",
                "SearchByProjection(KF, Scw) ok", "SearchByBoW ok", "SearchForInitialization ok", "SearchForTriangulation ok",
                "SearchBySim3 ok", "Fuse ok", "Fuse(Sim3) ok", "matcher dropin ok"):
        assert tag in r.stdout, r.stdout
