"""Host-side mirror of ORB_SLAM2::ORBVocabulary (Source/Libraries/ORB_SLAM2/include/ORBVocabulary.h) for the part
the front end needs: loading the text format and ``transform(descriptors, BowVector, FeatureVector, levelsup)``
(Frame::ComputeBoW, Source/Libraries/ORB_SLAM2/src/Frame.cc:412-417).  The tree descent runs on the GPU."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


class ORBVocabulary:
    def __init__(self):
        self._L = _lib.lib()
        self._h = C.c_void_p(None)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.orbfe_vocabulary_destroy(self._h)
            self._h = C.c_void_p(None)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def loadFromTextFile(self, filename: str, device: int = -1) -> bool:
        self.close()
        rc = self._L.orbfe_vocabulary_load_text(filename.encode(), device, C.byref(self._h))
        if rc == _lib.ERR_INVALID:
            return False  # the reference returns false on a malformed file
        _lib.check(rc, "orbfe_vocabulary_load_text")
        return True

    def loadFromBinaryFile(self, filename: str, device: int = -1) -> bool:
        """ORBVocabulary::loadFromBinaryFile (ORBVocabulary.cc:152-213)"""
        self.close()
        rc = self._L.orbfe_vocabulary_load_binary(filename.encode(), device, C.byref(self._h))
        if rc == _lib.ERR_INVALID:
            return False
        _lib.check(rc, "orbfe_vocabulary_load_binary")
        return True

    @classmethod
    def from_arrays(cls, k, L, parent, is_leaf, desc, weight, scoring=0, weighting=0, device=-1):
        v = cls()
        parent = np.ascontiguousarray(parent, np.int32); is_leaf = np.ascontiguousarray(is_leaf, np.uint8)
        desc = np.ascontiguousarray(desc, np.uint8); weight = np.ascontiguousarray(weight, np.float64)
        _lib.check(v._L.orbfe_vocabulary_create(k, L, scoring, weighting, len(parent), _lib.ptr(parent), _lib.ptr(is_leaf),
                                                _lib.ptr(desc), _lib.ptr(weight), device, C.byref(v._h)), "orbfe_vocabulary_create")
        return v

    def info(self):
        k, L, n, w = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        _lib.check(self._L.orbfe_vocabulary_info(self._h, C.byref(k), C.byref(L), C.byref(n), C.byref(w)), "orbfe_vocabulary_info")
        return k.value, L.value, n.value, w.value

    def transform(self, descriptors, levelsup: int = 4):
        """Returns (BowVector {word: value}, FeatureVector {node: [indices]}, per-feature (word, node, weight))."""
        d = np.ascontiguousarray(descriptors, np.uint8).reshape(-1, 32)
        n = len(d)
        word = np.zeros(n, np.int32); node = np.zeros(n, np.int32); weight = np.zeros(n, np.float64)
        bow_ids = np.zeros(max(n, 1), np.int32); bow_vals = np.zeros(max(n, 1), np.float64)
        fv_nodes = (_lib.FeatVecNode * max(n, 1))()
        fv_idx = np.zeros(max(n, 1), np.int32)
        nb, nf = C.c_int(0), C.c_int(0)
        _lib.check(self._L.orbfe_compute_bow(self._h, _lib.ptr(d), n, levelsup, _lib.ptr(word), _lib.ptr(node), _lib.ptr(weight),
                                             _lib.ptr(bow_ids), _lib.ptr(bow_vals), C.byref(nb), C.cast(fv_nodes, C.c_void_p),
                                             _lib.ptr(fv_idx), C.byref(nf)), "orbfe_compute_bow")
        bow = {int(bow_ids[i]): float(bow_vals[i]) for i in range(nb.value)}
        fv = {int(fv_nodes[i].node_id): fv_idx[fv_nodes[i].start: fv_nodes[i].start + fv_nodes[i].count].tolist()
              for i in range(nf.value)}
        return bow, fv, (word, node, weight)
