"""ctypes binding of oracle/_ref/libeigen_ref.so: the REFERENCE's own Eigen-dependent headers (common_lib.h, so3_math.h and, with
Boost, use-ikfom.hpp + the IKFoM toolkit) behind a C driver.  TEST INFRASTRUCTURE ONLY; only tests/ may import this module.

Recipe: oracle/ref_eigen/ (Makefile, stubs/, eigen_driver.cpp).  It needs Eigen 3, which this image does not have: `build()` runs the
recipe when /root/reference is present, the recipe skips itself (successfully, saying why) when it finds no Eigen, and `why_not()`
returns the reason the tests print when they skip.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_DIR, "_ref", "libeigen_ref.so")
REF_HDR = "/root/reference/include/common_lib.h"
_lib = None
_why = None

H_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                   C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.POINTER(C.c_double)))


def build():
    global _why
    mk = os.path.join(_DIR, "ref_eigen")
    if os.path.exists(REF_HDR):
        r = subprocess.run(["make", "-C", mk, "-s"], capture_output=True, text=True)
        if r.returncode != 0:
            _why = "oracle/ref_eigen failed to build: " + (r.stderr or r.stdout)[-400:]
            return None
        if not os.path.exists(LIB_PATH):
            _why = (r.stdout.strip().splitlines() or ["oracle/ref_eigen produced no library"])[-1]
    elif not os.path.exists(LIB_PATH):
        _why = "oracle/_ref/libeigen_ref.so not prebuilt and /root/reference not present to build it"
    return LIB_PATH if os.path.exists(LIB_PATH) else None


def available():
    return build() is not None


def why_not():
    if _why is None:
        build()
    return _why or "available"


def lib():
    global _lib
    if _lib is None:
        if build() is None:
            raise RuntimeError(why_not())
        L = C.CDLL(LIB_PATH)
        L.ref_eigen_version.restype = C.c_char_p
        dp, fp = C.POINTER(C.c_double), C.POINTER(C.c_float)
        L.ref_esti_plane.argtypes = [fp, C.c_float, fp]
        L.ref_state18_plus.argtypes = [dp, dp, dp]
        L.ref_state18_minus.argtypes = [dp, dp, dp, dp, dp]
        L.ref_so3_exp.argtypes = [dp, dp]
        L.ref_so3_exp_dt.argtypes = [dp, C.c_double, dp]
        L.ref_so3_log.argtypes = [dp, dp]
        if L.ref_have_mtk():
            L.ref_state23_boxplus.argtypes = [dp, dp]
            L.ref_state23_boxminus.argtypes = [dp, dp, dp]
            L.ref_A_matrix.argtypes = [dp, dp]
            L.ref_S2_Bx.argtypes = [dp, dp]
            L.ref_S2_Nx_yy.argtypes = [dp, dp]
            L.ref_S2_Mx.argtypes = [dp, dp, dp]
            L.ref_ikfom_update_dyn_share.argtypes = [dp, dp, C.c_double, C.c_int, dp, H_FN, C.c_void_p]
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def esti_plane(near, threshold=0.1):
    near = np.ascontiguousarray(near, dtype=np.float32).reshape(5, 3)
    out = np.zeros(4, dtype=np.float32)
    ok = lib().ref_esti_plane(near.ctypes.data_as(C.POINTER(C.c_float)), C.c_float(threshold), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out, bool(ok)


def state18_plus(rot, v15, d18):
    rot = np.array(rot, dtype=np.float64).reshape(9).copy()
    v = np.array(v15, dtype=np.float64).copy()
    d = np.ascontiguousarray(d18, dtype=np.float64)
    lib().ref_state18_plus(_d(rot), _d(v), _d(d))
    return rot.reshape(3, 3), v


def state18_minus(rot_a, v_a, rot_b, v_b):
    a = np.ascontiguousarray(rot_a, dtype=np.float64).reshape(9)
    b = np.ascontiguousarray(rot_b, dtype=np.float64).reshape(9)
    va = np.ascontiguousarray(v_a, dtype=np.float64)
    vb = np.ascontiguousarray(v_b, dtype=np.float64)
    out = np.zeros(18)
    lib().ref_state18_minus(_d(a), _d(va), _d(b), _d(vb), _d(out))
    return out


def so3_exp(v):
    v = np.ascontiguousarray(v, dtype=np.float64)
    R = np.zeros(9)
    lib().ref_so3_exp(_d(v), _d(R))
    return R.reshape(3, 3)


def so3_log(R):
    R = np.ascontiguousarray(R, dtype=np.float64).reshape(9)
    out = np.zeros(3)
    lib().ref_so3_log(_d(R), _d(out))
    return out


def have_mtk():
    return bool(lib().ref_have_mtk())


def state23_boxplus(s26, d23):
    s = np.array(s26, dtype=np.float64).copy()
    d = np.ascontiguousarray(d23, dtype=np.float64)
    lib().ref_state23_boxplus(_d(s), _d(d))
    return s


def state23_boxminus(s26, o26):
    s = np.ascontiguousarray(s26, dtype=np.float64)
    o = np.ascontiguousarray(o26, dtype=np.float64)
    out = np.zeros(23)
    lib().ref_state23_boxminus(_d(s), _d(o), _d(out))
    return out


def ikfom_update_dyn_share(s26, P, R, max_iter, h_dyn_share, limit=None, state_cls=None):
    """The reference's unmodified update_iterated_dyn_share_modified around a Python callback of the oracle's shape:
    h_dyn_share(state (oracle.State23), valid, converge) -> (valid, h_x (rows, 12), h (rows,)).  Returns (state26, P, calls)."""
    from . import oracle as orc
    limit = np.full(23, 0.001) if limit is None else np.ascontiguousarray(limit, dtype=np.float64)
    s = np.array(s26, dtype=np.float64).copy()
    P = np.array(P, dtype=np.float64).copy()
    keep = []

    def cb(ctx, st, valid, converge, rows, hx_out, h_out):
        xs = orc.State23.from_buffer_copy(np.ctypeslib.as_array(st, shape=(26,)).tobytes())
        v, hx, hv = h_dyn_share(xs, bool(valid[0]), bool(converge[0]))
        hx = np.ascontiguousarray(hx, dtype=np.float64).reshape(-1, 12)
        hv = np.ascontiguousarray(hv, dtype=np.float64).reshape(-1)
        keep[:] = [hx, hv]
        valid[0] = 1 if v else 0
        rows[0] = hx.shape[0]
        hx_out[0] = _d(hx) if hx.size else None
        h_out[0] = _d(hv) if hv.size else None
    calls = lib().ref_ikfom_update_dyn_share(_d(s), _d(P), R, max_iter, _d(limit), H_FN(cb), None)
    return s, P, calls
