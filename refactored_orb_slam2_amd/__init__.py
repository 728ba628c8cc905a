"""refactored_orb_slam2_amd -- MI355X-native ORB-SLAM2 front end (ORBextractor + ORBmatcher hot path).

The product is the in-tree HIP library csrc/liborbfe.so behind the C ABI of include/orbfe.h; the Python
modules here are the host-side mirror of the reference classes used by tests/ and bench.py.
"""
from .extractor import ORBextractor  # noqa: F401
