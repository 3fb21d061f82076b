"""Harness of tests/test_ikfom_boundary_ref_{cpu,gpu}.py (TEST INFRASTRUCTURE): the PRODUCT's registered 2-argument h_share_model
(fast-livo_amd/host/fastlivo_shim.hpp, sum-compat surrogate) as the callback of the REFERENCE's own updater text
(`ref_ikfom_update_text` of oracle/_ref/libeigen_ref.so = esekf::update_iterated_dyn_share_modified, esekfom.hpp:1619-1928, compiled from
the reference's source) -- and, for comparison, the reference's own h_share_model text (laserMapping.cpp:961-1093) under the same
updater.  Both halves search the reference's own ikd-Tree."""
import ctypes as C
import os
import subprocess

import numpy as np

from oracle import eigenref, ikdref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
D = os.path.join(ROOT, "tests", "host_emul")
KNN_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_float), C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_uint8))


def _stale(out, srcs):
    return not os.path.exists(out) or any(os.path.getmtime(out) < os.path.getmtime(s) for s in srcs if os.path.exists(s))


def build(emul):
    """libhshare_product.so (against libfastlivo_hip.so) or libhshare_product_emul.so (against the host build of the four entry points)."""
    shim = [os.path.join(ROOT, "fast-livo_amd", "host", f) for f in ("fastlivo_shim.hpp", "fastlivo_types.hpp")]
    src = os.path.join(D, "hshare_product.cpp")
    if emul:
        abi = os.path.join(D, "libflabi_emul.so")
        csrc = [os.path.join(ROOT, "fast-livo_amd", "csrc", f) for f in ("fl_math.h", "fl_ikfom_math.h")]
        if _stale(abi, [os.path.join(D, "flabi_emul.cpp")] + csrc):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas", "-o", abi,
                                   os.path.join(D, "flabi_emul.cpp")])
        so = os.path.join(D, "libhshare_product_emul.so")
        if _stale(so, [src, abi] + shim):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", so, src, "-L" + D, "-lflabi_emul",
                                   "-Wl,-rpath," + D])
        return C.CDLL(so), C.CDLL(abi)
    so = os.path.join(D, "libhshare_product.so")
    libdir = os.path.join(ROOT, "fast-livo_amd")
    if _stale(so, [src, os.path.join(libdir, "libfastlivo_hip.so")] + shim):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", so, src, "-L" + libdir, "-lfastlivo_hip",
                               "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return C.CDLL(so), None


class RefTreeKnn:
    """fl_knn_fn over the reference's ikd-Tree built as h_share_model's harness builds it (oracle/ref_eigen/text/lio_4.inc ref_hsm_begin):
    5 neighbours ascending; valid = 5 found && sq[4] <= 5 (laserMapping.cpp:1011-1012)."""

    def __init__(self, map_xyz):
        self.tree = ikdref.IkdTree(0.2, 0.5, 0.6)
        self.tree.build(map_xyz)
        self.calls = 0

        def cb(ctx, world, n, nbr, valid):
            w = np.ctypeslib.as_array(world, shape=(n, 3))
            xyz, sq, found = self.tree.nearest(w, 5)
            np.ctypeslib.as_array(nbr, shape=(n, 5, 3))[:] = xyz
            np.ctypeslib.as_array(valid, shape=(n,))[:] = ((found >= 5) & (sq[:, 4] <= 5.0)).astype(np.uint8)
            self.calls += 1
        self.fn = KNN_FN(cb)

    def close(self):
        self.tree.close()


def run_product(lib, handle_ptr, set_points, body, map_xyz, s26, P, R, max_iter):
    """The reference's updater text around the product's callback.  `set_points(body)` stages the scan through the ABI in use.
    Returns dict(state, P, calls, searches, invalid, neff, status, knn_calls)."""
    body = np.ascontiguousarray(body, np.float32)
    knn = RefTreeKnn(map_xyz)
    try:
        set_points(body)
        lib.product_hsm_setup.argtypes = [C.c_void_p, KNN_FN, C.c_void_p, C.c_int]
        lib.product_hsm_setup(handle_ptr, knn.fn, None, len(body))
        cb = C.cast(lib.product_hsm_callback, eigenref.H_FN)
        s, Pn, calls = eigenref.ikfom_update_text_c(s26, P, R, max_iter, cb)
        st = (C.c_int * 5)()
        tr = C.c_double()
        lib.product_hsm_stats(st, C.byref(tr))
        assert st[0] == calls
        return dict(state=s, P=Pn, calls=calls, searches=st[1], invalid=st[2], neff=st[3], status=st[4], knn_calls=knn.calls,
                    total_residual=tr.value)
    finally:
        knn.close()


def run_reference(body, map_xyz, s26, P, R, max_iter):
    """Both halves the reference's text: its updater around its own h_share_model over its own tree."""
    hm = eigenref.HShareModel(body, map_xyz)
    try:
        s, Pn, calls = eigenref.ikfom_update_text_c(s26, P, R, max_iter, hm.callback)
        last = hm.last()
    finally:
        hm.close()
    return dict(state=s, P=Pn, calls=calls, neff=last["effct_feat_num"], total_residual=last["total_residual"])


def tiny_scan(fr, keep):
    """`keep` points of the frame's scan spread over it (rows < 23: the updater's N x N branch, esekfom.hpp:1712-1741, on the reference side)"""
    idx = np.linspace(0, fr.n - 1, keep).astype(int)
    return np.ascontiguousarray(fr.body_xyz[idx])


def far_scan(fr, n=200):
    """a scan nowhere near the map: no point has 5 neighbours within sqrt(5) m -> effct_feat_num = 0 -> valid = false every pass"""
    return np.ascontiguousarray(fr.body_xyz[:n] + np.array([500.0, 500.0, 300.0], np.float32))
