"""ctypes mirror of the batched stereo pipeline handle of the C ABI (include/orbfe.h: orbfe_pipeline_*): what a C / C++ host of
the batched-sequence mode drives -- pinned pitched input slots, H2D, ORBextractor x 2, Frame::ComputeStereoMatches,
Frame::UnprojectStereo, SearchByProjection(cur, last), D2H -- without torch anywhere.  `examples/stereo_kitti.cc --batch F` is
the C++ caller; this class serves the tests and Python hosts.

The reference loop it batches: Source/Examples/Stereo/stereo_kitti.cc:88-106 (imread left / right, TrackStereo), i.e.
L/src/Frame.cc:66-127 + L/src/Tracking.cc:857-884 per pair."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


class PipelineConfig(C.Structure):
    _fields_ = [("extractor", _lib.Params), ("width", C.c_int32), ("height", C.c_int32), ("batch", C.c_int32), ("slots", C.c_int32),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("bf", C.c_float), ("th", C.c_float),
                ("check_orientation", C.c_int32), ("output_mask", C.c_int32)]


OUT_KEYPOINTS, OUT_DESCRIPTORS, OUT_STEREO, OUT_ASSIGNED, OUT_COUNTS = 1, 2, 4, 8, 16   # ORBFE_PIPE_OUT_*


class _InputView(C.Structure):
    _fields_ = [("left", C.c_void_p), ("right", C.c_void_p), ("pitch", C.c_int32), ("image_bytes", C.c_size_t),
                ("cams", C.c_void_p), ("poses", C.c_void_p)]


class _OutputView(C.Structure):
    _fields_ = [("cap", C.c_int32), ("n_left", C.c_void_p), ("kps_left", C.c_void_p), ("desc_left", C.c_void_p), ("n_right", C.c_void_p),
                ("u_right", C.c_void_p), ("depth", C.c_void_p), ("n_stereo", C.c_void_p), ("assigned", C.c_void_p), ("n_tracked", C.c_void_p)]


def _view(ptr, shape, dtype):
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    return np.frombuffer((C.c_uint8 * n).from_address(ptr), dtype=dtype).reshape(shape)


class StereoPipeline:
    """with StereoPipeline(w, h, batch, camera...) as p:  p.left(slot)[j, :h, :w] = image; p.submit(slot, n, has_predecessor);
    p.wait(slot); out = p.output(slot)  (numpy views of the slot's pinned blocks: valid until the slot is submitted again)."""

    def __init__(self, width: int, height: int, batch: int, fx: float, fy: float, cx: float, cy: float, bf: float, th: float = 7.0,
                 n_features: int = 2000, scale_factor: float = 1.2, n_levels: int = 8, ini_th: int = 20, min_th: int = 7, slots: int = 2,
                 check_orientation: bool = True, device: int = -1, output_mask: int = 0):
        self._L = _lib.lib()
        self._L.orbfe_pipeline_create.argtypes = [C.POINTER(PipelineConfig), C.c_int, C.POINTER(C.c_void_p)]
        self._L.orbfe_pipeline_destroy.argtypes = [C.c_void_p]
        self._L.orbfe_pipeline_input.argtypes = [C.c_void_p, C.c_int, C.POINTER(_InputView)]
        self._L.orbfe_pipeline_output.argtypes = [C.c_void_p, C.c_int, C.POINTER(_OutputView)]
        self._L.orbfe_pipeline_submit.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        self._L.orbfe_pipeline_wait.argtypes = [C.c_void_p, C.c_int]
        self._L.orbfe_pipeline_submit_resident.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        self._L.orbfe_pipeline_device_input.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_size_t)]
        self.cfg = PipelineConfig(_lib.Params(n_features, scale_factor, n_levels, ini_th, min_th), width, height, batch, slots,
                                  fx, fy, cx, cy, bf, th, int(check_orientation), int(output_mask))
        self._h = C.c_void_p(None)
        _lib.check(self._L.orbfe_pipeline_create(C.byref(self.cfg), device, C.byref(self._h)), "orbfe_pipeline_create")
        self.batch, self.slots, self.width, self.height = batch, slots, width, height

    def _input(self, slot: int) -> _InputView:
        v = _InputView()
        _lib.check(self._L.orbfe_pipeline_input(self._h, slot, C.byref(v)), "orbfe_pipeline_input")
        return v

    def left(self, slot: int) -> np.ndarray:
        """(batch, height, pitch) uint8 view of the slot's pinned left images; columns beyond `width` are padding."""
        v = self._input(slot)
        return _view(v.left, (self.batch, self.height, v.pitch), np.uint8)

    def right(self, slot: int) -> np.ndarray:
        v = self._input(slot)
        return _view(v.right, (self.batch, self.height, v.pitch), np.uint8)

    def cams(self, slot: int) -> np.ndarray:
        return _view(self._input(slot).cams, (self.batch,), _lib.UNPROJECT_CAM_DTYPE)

    def poses(self, slot: int) -> np.ndarray:
        return _view(self._input(slot).poses, (self.batch,), _lib.TRACK_POSE_DTYPE)

    def submit(self, slot: int, n_frames: int, has_predecessor: bool):
        _lib.check(self._L.orbfe_pipeline_submit(self._h, slot, n_frames, int(has_predecessor)), "orbfe_pipeline_submit")

    def submit_resident(self, slot: int, n_frames: int, has_predecessor: bool):
        """The chunk whose images already are in the slot's DEVICE input blocks (an earlier submit's, or a device-side producer's)."""
        _lib.check(self._L.orbfe_pipeline_submit_resident(self._h, slot, n_frames, int(has_predecessor)), "orbfe_pipeline_submit_resident")

    def device_input(self, slot: int):
        """(device pointer of the left images, of the right images, row pitch, bytes per image) of slot `slot`."""
        dl, dr, pitch, ib = C.c_void_p(), C.c_void_p(), C.c_int(), C.c_size_t()
        _lib.check(self._L.orbfe_pipeline_device_input(self._h, slot, C.byref(dl), C.byref(dr), C.byref(pitch), C.byref(ib)), "orbfe_pipeline_device_input")
        return dl.value, dr.value, pitch.value, ib.value

    def wait(self, slot: int):
        _lib.check(self._L.orbfe_pipeline_wait(self._h, slot), "orbfe_pipeline_wait")

    def output(self, slot: int) -> dict:
        v = _OutputView()
        _lib.check(self._L.orbfe_pipeline_output(self._h, slot, C.byref(v)), "orbfe_pipeline_output")
        F, cap = self.batch, v.cap
        return {"cap": cap, "n_left": _view(v.n_left, (F,), np.int32), "kps_left": _view(v.kps_left, (F, cap), _lib.KP_DTYPE),
                "desc_left": _view(v.desc_left, (F, cap, 32), np.uint8), "n_right": _view(v.n_right, (F,), np.int32),
                "u_right": _view(v.u_right, (F, cap), np.float32), "depth": _view(v.depth, (F, cap), np.float32),
                "n_stereo": _view(v.n_stereo, (F,), np.int32), "assigned": _view(v.assigned, (F, cap), np.int32),
                "n_tracked": _view(v.n_tracked, (F,), np.int32)}

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._L.orbfe_pipeline_destroy(self._h)
            self._h = C.c_void_p(None)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
