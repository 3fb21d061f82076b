"""SURVEY 8b, IKFoM callback boundary, closed on the reference's OWN updater -- CPU form (no GPU): the product's 2-argument
`h_share_model` (fast-livo_amd/host/fastlivo_shim.hpp; it returns the 23x12 sum-compat surrogate of the reduced h_x^T h_x / h_x^T h)
is the callback of `esekf::update_iterated_dyn_share_modified` compiled from the reference's source text (esekfom.hpp:1619-1928;
registration laserMapping.cpp:1233-1235, call :1484, invocation esekfom.hpp:1636), and must drive it to the state and covariance the
reference's own h_share_model text (laserMapping.cpp:961-1093) drives it to.  The four ABI entry points the callback uses run over the
host build of the product's per-point arithmetic here (tests/host_emul/flabi_emul.cpp); tests/test_ikfom_boundary_ref_gpu.py is the
same test over libfastlivo_hip.so.  Tolerances: state 1e-9, covariance 1e-10 relative."""
import ctypes as C

import numpy as np
import pytest

import boundary_ref as br
from oracle import eigenref, ikdref

pytestmark = pytest.mark.skipif(not (eigenref.available() and ikdref.available()),
                                reason="oracle/_ref not available: " + (eigenref.why_not() or "no libikdtree_ref.so"))
R = 0.001


def _emul():
    lib, abi = br.build(emul=True)
    abi.flabi_emul_create.argtypes = [C.POINTER(C.c_void_p)]
    abi.fl_lio_set_points.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int32]
    abi.flabi_emul_destroy.argtypes = [C.c_void_p]
    h = C.c_void_p()
    assert abi.flabi_emul_create(C.byref(h)) == 0

    def set_points(body):
        assert abi.fl_lio_set_points(h, body.ctypes.data_as(C.POINTER(C.c_float)), len(body)) == 0
    return lib, abi, h, set_points


def _frame(oracle_lib, scene, n):
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(n, scene=scene)
    return fr, oracle_lib.state23_from_frame(fr, synth.quat_from_R).vec(), fr.cov23.copy()


def _compare(a, b, s0):
    assert a["calls"] == b["calls"]
    assert np.abs(a["state"] - b["state"]).max() <= 1e-9
    assert np.abs(a["P"] - b["P"]).max() <= 1e-10 * max(1.0, np.abs(b["P"]).max())
    return np.abs(b["state"] - s0).max()


@pytest.mark.parametrize("n,max_iter", [(6000, 4), (3000, 10)])
def test_product_callback_drives_the_reference_updater(oracle_lib, scene, n, max_iter):
    lib, abi, h, set_points = _emul()
    try:
        fr, s0, P0 = _frame(oracle_lib, scene, n)
        a = br.run_product(lib, h, set_points, fr.body_xyz, scene.map_xyz, s0, P0, R, max_iter)
        b = br.run_reference(fr.body_xyz, scene.map_xyz, s0, P0, R, max_iter)
        assert a["status"] == 0 and a["invalid"] == 0 and a["neff"] == b["neff"] and a["neff"] > n // 4
        assert a["searches"] == a["knn_calls"] >= 2                   # the `converge` rematch restaged the neighbours (laserMapping.cpp:994)
        # (the two runs' states differ by ~1e-13: a point whose FLOAT world coordinate sits on a rounding boundary moves its pd2 by one float
        # ulp, ~1e-6 m -- a handful of those is all the slack this diagnostic sum needs)
        assert abs(a["total_residual"] - b["total_residual"]) <= 1e-5
        assert _compare(a, b, s0) > 1e-4                              # the update did something
    finally:
        abi.flabi_emul_destroy(h)


def test_fewer_than_23_rows(oracle_lib, scene):
    """the reference side takes the N x N branch (rows < 23, esekfom.hpp:1712-1741); the surrogate always has 23 rows"""
    lib, abi, h, set_points = _emul()
    try:
        fr, s0, P0 = _frame(oracle_lib, scene, 2000)
        body = br.tiny_scan(fr, 18)
        a = br.run_product(lib, h, set_points, body, scene.map_xyz, s0, P0, R, 4)
        b = br.run_reference(body, scene.map_xyz, s0, P0, R, 4)
        assert 1 <= b["neff"] < 23 and a["neff"] == b["neff"] and a["invalid"] == 0
        assert _compare(a, b, s0) > 1e-6
    finally:
        abi.flabi_emul_destroy(h)


def test_no_effective_point(oracle_lib, scene):
    """effct_feat_num = 0: FAST-LIVO's h_share_model has no `valid = false` early-out (laserMapping.cpp:1040-1060 -- FAST-LIO's has): it
    returns 0 rows, the updater's N x N branch runs over empty matrices (K_h = K_x = 0) and stops after two `converge` passes.  The
    product's callback returns the all-zero surrogate: same calls, same state, same covariance."""
    lib, abi, h, set_points = _emul()
    try:
        fr, s0, P0 = _frame(oracle_lib, scene, 2000)
        body = br.far_scan(fr)
        a = br.run_product(lib, h, set_points, body, scene.map_xyz, s0, P0, R, 5)
        b = br.run_reference(body, scene.map_xyz, s0, P0, R, 5)
        assert a["calls"] == b["calls"] == 2 and a["invalid"] == 0 and a["neff"] == b["neff"] == 0
        assert np.array_equal(a["state"], b["state"]) and np.abs(b["state"] - s0).max() <= 1e-15
        assert np.array_equal(a["P"], b["P"])
    finally:
        abi.flabi_emul_destroy(h)
