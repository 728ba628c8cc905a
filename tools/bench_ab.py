import os, sys, runpy
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from refactored_orb_slam2_amd import _lib
if os.environ.get("ORBFE_AB_LIB"): _lib.LIB_PATH = os.path.join(_lib.CSRC, "_ab", "liborbfe_%s.so" % os.environ["ORBFE_AB_LIB"])
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "bench.py"), run_name="__main__")
