"""SURVEY 8b, IKFoM callback boundary, closed on the reference's OWN updater, on the MI355X: the product's registered 2-argument
`h_share_model` (fast-livo_amd/host/fastlivo_shim.hpp over libfastlivo_hip.so: fl_ikfom_world_points / fl_lio_set_neighbours /
fl_h_share_model_sums, the device-reduced sums returned as the 23x12 sum-compat surrogate) is the callback of
`esekf::update_iterated_dyn_share_modified` compiled from the reference's source text (esekfom.hpp:1619-1928 in
oracle/_ref/libeigen_ref.so; registration laserMapping.cpp:1233-1235, call :1484, invocation esekfom.hpp:1636).  The outcome must equal
  (b) the reference's updater text around the reference's own h_share_model text (laserMapping.cpp:961-1093) over the reference's ikd-Tree,
  (c) fl_ikfom_update_iterated_dev, the product's whole update on the device (its own k-NN),
at the tolerances of SURVEY 8c: state 1e-9, covariance 1e-10 relative; same number of callback invocations and `converge` rematches.
tests/test_ikfom_boundary_ref_cpu.py is the same harness over the host build of the product's arithmetic."""
import ctypes as C

import numpy as np
import pytest

import boundary_ref as br
from oracle import eigenref, ikdref

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not (eigenref.available() and ikdref.available()),
                                 reason="oracle/_ref not available: " + (eigenref.why_not() or "no libikdtree_ref.so"))]
R = 0.001


def _run(gpu_lib, oracle_lib, scene, fr, body, max_iter):
    capi = gpu_lib
    from fast_livo_amd import synth
    lib, _ = br.build(emul=False)
    s0 = oracle_lib.state23_from_frame(fr, synth.quat_from_R).vec()
    P0 = fr.cov23.copy()
    h = capi.Handle(capi.config_from_frames(fr, max_iterations=max_iter))
    try:
        a = br.run_product(lib, h.h, lambda b: h.lio_set_points(b), body,
                           scene.map_xyz, s0, P0, R, max_iter)
        b = br.run_reference(body, scene.map_xyz, s0, P0, R, max_iter)
        h.map_set_points(scene.map_xyz, 0.5)
        xg = capi.state23_from_frame(fr)
        Pg = fr.cov23.copy()
        info = h.ikfom_update_iterated_dev(xg, Pg, body, R)
        c = dict(state=xg.vec(), P=Pg, calls=info.iterations, neff=info.effct_feat_num)
    finally:
        h.close()
    return s0, P0, a, b, c


def _close(x, y):
    assert x["calls"] == y["calls"], (x["calls"], y["calls"])
    assert np.abs(x["state"] - y["state"]).max() <= 1e-9
    assert np.abs(x["P"] - y["P"]).max() <= 1e-10 * max(1.0, np.abs(y["P"]).max())


@pytest.mark.parametrize("n,max_iter", [(8000, 4), (50000, 10)])
def test_product_callback_drives_the_reference_updater(gpu_lib, oracle_lib, scene, n, max_iter):
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(n, scene=scene)
    s0, P0, a, b, c = _run(gpu_lib, oracle_lib, scene, fr, fr.body_xyz, max_iter)
    assert a["status"] == 0 and a["invalid"] == 0 and a["neff"] == b["neff"] == c["neff"] and a["neff"] > n // 4
    assert a["searches"] == a["knn_calls"] >= 2                   # the `converge` rematch restaged the neighbours (laserMapping.cpp:994)
    # (the two runs' states differ by ~1e-13: a point whose FLOAT world coordinate sits on a rounding boundary moves its pd2 by one float
    # ulp, ~1e-6 m -- a handful of those is all the slack this diagnostic sum needs)
    assert abs(a["total_residual"] - b["total_residual"]) <= 1e-5
    _close(a, b)
    _close(a, c)
    assert np.abs(b["state"] - s0).max() > 1e-4                   # the update did something


def test_fewer_than_23_rows(gpu_lib, oracle_lib, scene):
    """the reference side takes the N x N branch (rows < 23, esekfom.hpp:1712-1741); the surrogate always has 23 rows"""
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(2000, scene=scene)
    s0, P0, a, b, c = _run(gpu_lib, oracle_lib, scene, fr, br.tiny_scan(fr, 18), 4)
    assert 1 <= b["neff"] < 23 and a["neff"] == b["neff"] == c["neff"] and a["invalid"] == 0
    _close(a, b)
    _close(a, c)
    assert np.abs(b["state"] - s0).max() > 1e-6


def test_no_effective_point(gpu_lib, oracle_lib, scene):
    """effct_feat_num = 0: FAST-LIVO's callback returns 0 rows with `valid` still true (laserMapping.cpp:1040-1060), the product's the all-zero
    surrogate: two `converge` passes, state and covariance unchanged on every side"""
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(2000, scene=scene)
    s0, P0, a, b, c = _run(gpu_lib, oracle_lib, scene, fr, br.far_scan(fr), 5)
    assert a["calls"] == b["calls"] == c["calls"] == 2 and a["invalid"] == 0 and a["neff"] == b["neff"] == c["neff"] == 0
    assert np.array_equal(a["state"], b["state"]) and np.array_equal(a["P"], b["P"])
    _close(a, c)
