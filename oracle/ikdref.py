"""ctypes binding of oracle/_ref/libikdtree_ref.so: the REFERENCE's own ikd-Tree.  TEST INFRASTRUCTURE ONLY.

The library is compiled by oracle/ref_ikdtree/Makefile from /root/reference/include/ikd-Tree/ikd_Tree.cpp where it lies (that
tree exists only in the build container; the GPU box uses the prebuilt .so that travels with the snapshot).  Only tests/ may
import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_DIR, "_ref", "libikdtree_ref.so")
REF_SRC = "/root/reference/include/ikd-Tree/ikd_Tree.cpp"
_lib = None


def build(force=False):
    """Build when the reference tree is present; otherwise the prebuilt library must already be there."""
    mk = os.path.join(_DIR, "ref_ikdtree")
    if os.path.exists(REF_SRC):
        if force and os.path.exists(LIB_PATH):
            os.remove(LIB_PATH)
        subprocess.check_call(["make", "-C", mk, "-s"])
    return LIB_PATH if os.path.exists(LIB_PATH) else None


def available():
    return build() is not None


def lib():
    global _lib
    if _lib is None:
        if build() is None:
            raise RuntimeError("oracle/_ref/libikdtree_ref.so missing and /root/reference not present to build it")
        L = C.CDLL(LIB_PATH)
        L.ikdref_create.restype = C.c_void_p
        L.ikdref_create.argtypes = [C.c_float, C.c_float, C.c_float]
        L.ikdref_destroy.argtypes = [C.c_void_p]
        L.ikdref_set_downsample.argtypes = [C.c_void_p, C.c_float]
        L.ikdref_size.argtypes = [C.c_void_p]
        L.ikdref_validnum.argtypes = [C.c_void_p]
        L.ikdref_wait_rebuild.argtypes = [C.c_void_p, C.c_int]
        L.ikdref_build.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.ikdref_nearest.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.ikdref_nearest_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.ikdref_add_points.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.ikdref_delete_boxes.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.ikdref_flatten.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        _lib = L
    return _lib


def _f32(a, cols):
    a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1, cols)
    return a


class IkdTree:
    """KD_TREE with the parameters the reference constructs it with (defaults of ikd_Tree.h:165: 0.5, 0.6, 0.2;
    laserMapping.cpp then calls set_downsample_param(filter_size_map_min) before Build)."""

    def __init__(self, downsample=0.2, delete_param=0.5, balance_param=0.6):
        self.L = lib()
        self.h = self.L.ikdref_create(delete_param, balance_param, downsample)

    def close(self):
        if self.h:
            self.L.ikdref_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_downsample(self, ds):
        self.L.ikdref_set_downsample(self.h, ds)

    def build(self, xyz):
        xyz = _f32(xyz, 3)
        self.L.ikdref_build(self.h, xyz.ctypes.data, len(xyz))

    def size(self):
        return self.L.ikdref_size(self.h)

    def validnum(self):
        return self.L.ikdref_validnum(self.h)

    def wait_rebuild(self, timeout_ms=20000):
        return bool(self.L.ikdref_wait_rebuild(self.h, timeout_ms))

    def nearest(self, query, k=5, nthreads=1):
        """nthreads > 1: under `#pragma omp parallel for` as the reference calls it (laserMapping.cpp:1516-1519)."""
        q = _f32(query, 3)
        n = len(q)
        xyz = np.zeros((n, k, 3), np.float32)
        sq = np.zeros((n, k), np.float32)
        found = np.zeros(n, np.int32)
        if nthreads > 1:
            self.L.ikdref_nearest_mt(self.h, q.ctypes.data, n, k, xyz.ctypes.data, sq.ctypes.data, found.ctypes.data, int(nthreads))
        else:
            self.L.ikdref_nearest(self.h, q.ctypes.data, n, k, xyz.ctypes.data, sq.ctypes.data, found.ctypes.data)
        return xyz, sq, found

    def add_points(self, xyz, downsample=True):
        xyz = _f32(xyz, 3)
        return self.L.ikdref_add_points(self.h, xyz.ctypes.data, len(xyz), 1 if downsample else 0)

    def delete_boxes(self, boxes):
        b = _f32(boxes, 6)
        return self.L.ikdref_delete_boxes(self.h, b.ctypes.data, len(b))

    def flatten(self):
        cap = max(self.size(), 1) + 16
        out = np.zeros((cap, 3), np.float32)
        n = self.L.ikdref_flatten(self.h, out.ctypes.data, cap)
        if n > cap:
            out = np.zeros((n, 3), np.float32)
            n = self.L.ikdref_flatten(self.h, out.ctypes.data, n)
        return out[:n].copy()
