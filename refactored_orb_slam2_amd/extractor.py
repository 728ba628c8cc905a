"""Host-side mirror of ORB_SLAM2::ORBextractor over the C ABI of liborbfe.so.

Same constructor arguments, getters and call semantics as the reference class
(Source/Libraries/ORB_SLAM2/include/ORBextractor.h:43-104): ``extractor(image, mask)`` returns the
keypoints (cv::KeyPoint-layout records) and the N x 32 descriptor matrix; ``mvImagePyramid`` exposes the
8-bit pyramid of the last call.  This module is plumbing for tests and the benchmark; the C++ drop-in
classes live in csrc/host/.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import KP_DTYPE, STAGES


class ORBextractor:
    HARRIS_SCORE = 0
    FAST_SCORE = 1

    def __init__(self, nfeatures: int = 2000, scaleFactor: float = 1.2, nlevels: int = 8, iniThFAST: int = 20,
                 minThFAST: int = 7, device: int = -1):
        self._L = _lib.lib()
        self._h = C.c_void_p(None)
        prm = _lib.Params(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)
        _lib.check(self._L.orbfe_extractor_create(C.byref(prm), device, C.byref(self._h)), "orbfe_extractor_create")
        self.nfeatures, self.nlevels = nfeatures, nlevels
        self.scaleFactor = float(np.float32(scaleFactor))
        self._last_shape = None

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.orbfe_extractor_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- getters of ORBextractor.h:60-74
    def GetLevels(self) -> int:
        return self.nlevels

    def GetScaleFactor(self) -> float:
        return self.scaleFactor

    def _vec(self, fn, dtype=np.float32):
        out = np.zeros(self.nlevels, dtype)
        _lib.check(fn(self._h, _lib.ptr(out)), fn.__name__)
        return out

    def GetScaleFactors(self):
        return self._vec(self._L.orbfe_extractor_scale_factors)

    def GetInverseScaleFactors(self):
        return self._vec(self._L.orbfe_extractor_inv_scale_factors)

    def GetScaleSigmaSquares(self):
        return self._vec(self._L.orbfe_extractor_sigma2)

    def GetInverseScaleSigmaSquares(self):
        return self._vec(self._L.orbfe_extractor_inv_sigma2)

    def features_per_level(self):
        return self._vec(self._L.orbfe_extractor_features_per_level, np.int32)

    def max_keypoints(self, w: int, h: int) -> int:
        cap = C.c_int(0)
        _lib.check(self._L.orbfe_extractor_max_keypoints(self._h, w, h, C.byref(cap)), "orbfe_extractor_max_keypoints")
        return cap.value

    def prepare(self, w: int, h: int, n_images: int = 1):
        """orbfe_extractor_prepare: plan, work space, code objects and launch graph for w x h images, before the first frame."""
        _lib.check(self._L.orbfe_extractor_prepare(self._h, w, h, n_images), "orbfe_extractor_prepare")

    # ---- operator()
    def __call__(self, image: np.ndarray, mask=None):
        """ORBextractor::operator() on one host image (mask ignored, as in the reference)."""
        if image is None or image.size == 0:
            return None  # reference: silent return, outputs untouched
        if image.dtype != np.uint8 or image.ndim != 2:
            raise TypeError("image must be CV_8UC1 (2-D uint8)")
        if image.strides[1] != 1:
            image = np.ascontiguousarray(image)
        h, w = image.shape
        cap = self.max_keypoints(w, h)
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        _lib.check(self._L.orbfe_extract(self._h, _lib.ptr(image), w, h, image.strides[0], _lib.ptr(kps),
                                         _lib.ptr(desc), cap, C.byref(n)), "orbfe_extract")
        self._last_shape = (h, w)
        return kps[: n.value].copy(), desc[: n.value].copy()

    def extract_batch(self, images):
        """Batched operator() on a list of equally sized host images."""
        imgs = [np.ascontiguousarray(i, np.uint8) for i in images]
        h, w = imgs[0].shape
        B = len(imgs)
        cap = self.max_keypoints(w, h)
        kps = np.zeros((B, cap), KP_DTYPE)
        desc = np.zeros((B, cap, 32), np.uint8)
        n = np.zeros(B, np.int32)
        ptrs = (C.c_void_p * B)(*[i.ctypes.data for i in imgs])
        _lib.check(self._L.orbfe_extract_batch(self._h, C.cast(ptrs, C.c_void_p), B, w, h, imgs[0].strides[0],
                                               _lib.ptr(kps), _lib.ptr(desc), cap, _lib.ptr(n)), "orbfe_extract_batch")
        self._last_shape = (h, w)
        return [(kps[i, : n[i]].copy(), desc[i, : n[i]].copy()) for i in range(B)]

    def extract_batch_device(self, d_imgs, kps, desc, n_out, stream=None):
        """Device-resident batch.  d_imgs: torch uint8 (B, h, w) on the GPU; kps: uint8 (B, cap, 28);
        desc: uint8 (B, cap, 32); n_out: int32 (B).  Asynchronous on `stream` (torch stream or None)."""
        B, h, w = d_imgs.shape
        cap = desc.shape[1]
        s = _lib.stream_handle(stream)
        _lib.check(self._L.orbfe_extract_batch_device(
            self._h, _lib.ptr(d_imgs), B, w, h, d_imgs.stride(1), d_imgs.stride(0), _lib.ptr(kps), _lib.ptr(desc), cap,
            _lib.ptr(n_out), s), "orbfe_extract_batch_device")
        self._last_shape = (h, w)

    def sync(self):
        _lib.check(self._L.orbfe_sync(self._h), "orbfe_sync")

    def device_status(self):
        _lib.check(self._L.orbfe_device_status(self._h), "orbfe_device_status")

    # ---- mvImagePyramid
    def level_size(self, level: int, w: int | None = None, h: int | None = None):
        if w is None:
            h, w = self._last_shape
        lw, lh = C.c_int(), C.c_int()
        _lib.check(self._L.orbfe_pyramid_level_size(self._h, w, h, level, C.byref(lw), C.byref(lh)), "level_size")
        return lw.value, lh.value

    def pyramid_level(self, level: int) -> np.ndarray:
        lw, lh = self.level_size(level)
        out = np.zeros((lh, lw), np.uint8)
        w, h = C.c_int(), C.c_int()
        _lib.check(self._L.orbfe_pyramid_level(self._h, level, _lib.ptr(out), lw, C.byref(w), C.byref(h)), "orbfe_pyramid_level")
        return out

    @property
    def mvImagePyramid(self):
        """All levels with one device synchronisation (orbfe_pyramid_levels)."""
        outs = [np.zeros(self.level_size(l)[::-1], np.uint8) for l in range(self.nlevels)]
        ptrs = (C.c_void_p * self.nlevels)(*[o.ctypes.data for o in outs])
        strides = (C.c_int * self.nlevels)(*[o.strides[0] for o in outs])
        _lib.check(self._L.orbfe_pyramid_levels(self._h, C.cast(ptrs, C.c_void_p), C.cast(strides, C.c_void_p)), "orbfe_pyramid_levels")
        return outs

    # ---- stage-level outputs (parity tests)
    def debug_pyramid(self, image: int, level: int) -> np.ndarray:
        lw, lh = self.level_size(level)
        out = np.zeros((lh, lw), np.uint8)
        _lib.check(self._L.orbfe_debug_pyramid(self._h, image, level, _lib.ptr(out), lw), "orbfe_debug_pyramid")
        return out

    def debug_blurred(self, image: int, level: int) -> np.ndarray:
        lw, lh = self.level_size(level)
        out = np.zeros((lh, lw), np.uint8)
        _lib.check(self._L.orbfe_debug_blurred(self._h, image, level, _lib.ptr(out), lw), "orbfe_debug_blurred")
        return out

    def _xys(self, fn, image, level, cap):
        x = np.zeros(cap, np.int32); y = np.zeros(cap, np.int32); s = np.zeros(cap, np.int32)
        n = C.c_int(0)
        _lib.check(fn(self._h, image, level, _lib.ptr(x), _lib.ptr(y), _lib.ptr(s), cap, C.byref(n)), fn.__name__)
        return x[: n.value], y[: n.value], s[: n.value]

    def debug_candidates(self, image: int, level: int, cap: int = 1 << 18):
        return self._xys(self._L.orbfe_debug_candidates, image, level, cap)

    def debug_level_keypoints(self, image: int, level: int, cap: int = 1 << 14):
        return self._xys(self._L.orbfe_debug_level_keypoints, image, level, cap)

    # ---- HIP-event stage timing
    def profile(self, enable=True, stages=None):
        """Per-stage HIP-event timing on the launch stream.  stages: optional list of stage names to time (default all)."""
        code = int(bool(enable))
        if enable and stages is not None:
            code = sum(2 << STAGES.index(s) for s in stages)
        _lib.check(self._L.orbfe_profile_enable(self._h, code), "orbfe_profile_enable")

    def stage_intervals(self, ref_event, cap: int = 4096):
        """(stage name, start ms, end ms) of every stage launch timed since the last drain, on the clock of `ref_event` (a
        torch.cuda.Event(enable_timing=True) recorded earlier): orbfe_stage_intervals."""
        st = np.zeros(cap, np.int32); a = np.zeros(cap, np.float32); b = np.zeros(cap, np.float32); n = np.zeros(1, np.int32)
        _lib.check(self._L.orbfe_stage_intervals(self._h, C.c_void_p(int(ref_event.cuda_event)), _lib.ptr(st), _lib.ptr(a), _lib.ptr(b),
                                                 cap, _lib.ptr(n)), "orbfe_stage_intervals")
        return [(STAGES[int(st[i])], float(a[i]), float(b[i])) for i in range(int(n[0]))]

    def stage_times(self, reset: bool = True):
        ms = np.zeros(len(STAGES), np.float32)
        launches = np.zeros(len(STAGES), np.int32)
        _lib.check(self._L.orbfe_stage_times(self._h, _lib.ptr(ms), _lib.ptr(launches), int(reset)), "orbfe_stage_times")
        return {STAGES[i]: (float(ms[i]), int(launches[i])) for i in range(len(STAGES))}
