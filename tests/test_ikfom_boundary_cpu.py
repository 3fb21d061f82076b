"""SURVEY 8b, IKFoM callback boundary, the arithmetic claim behind the sum-compat body -- on the CPU, no GPU involved:
the unmodified updater (oracle restatement of esekfom.hpp:1619-1928, orc_ikfom_update_dyn_share) consumes h_x only through
h_x^T h_x and h_x^T h once rows >= 23 (:1781,:1801,:1806), so a callback that returns the 23x12 surrogate S (S^T S = H^T H,
S^T h = H^T z) drives it to the same state and covariance as the reference's own N_eff x 12 rows."""
import ctypes as C

import numpy as np

from helpers import p


def _rows_callback(orc, fr, scene_map, counter):
    """the reference's h_share_model (oracle restatement): kNN on converge, then the N_eff x 12 rows"""
    n = fr.n
    st = dict(nbr=np.zeros((n, 5, 3), np.float32), sel=np.zeros(n, np.uint8))
    normvec = np.zeros((n, 4), np.float32)
    res = np.zeros(n)
    world = np.zeros((n, 3), np.float32)

    def cb(xs, valid, converge):
        hx = np.zeros((n, 12)); hv = np.zeros(n); tr = C.c_double()
        if converge:
            # world points at the current state (:980-984): a pass with an all-zero selection computes nothing else
            scratch = np.zeros(n, np.uint8)
            orc.lib().orc_h_share_model(C.byref(xs), p(fr.body_xyz, C.c_float), p(st["nbr"], C.c_float), p(scratch, C.c_uint8), n, 1,
                                        p(world, C.c_float), p(normvec, C.c_float), p(res, C.c_double), p(hx, C.c_double), p(hv, C.c_double),
                                        C.byref(tr))
            nb, _, va, _ = orc.knn5_bruteforce(scene_map, world)
            st["nbr"][:] = nb
            st["sel"][:] = va
            counter["searches"] += 1
        neff = orc.lib().orc_h_share_model(C.byref(xs), p(fr.body_xyz, C.c_float), p(st["nbr"], C.c_float), p(st["sel"], C.c_uint8), n, 4,
                                           None, p(normvec, C.c_float), p(res, C.c_double), p(hx, C.c_double), p(hv, C.c_double), C.byref(tr))
        counter["calls"] += 1
        return True, hx[:neff].copy(), hv[:neff].copy()
    return cb


def _surrogate_of(cb_rows):
    """sum-compat: reduce the rows to H^T H / H^T z (what the device returns) and hand back the 23x12 surrogate"""
    def cb(xs, valid, converge):
        v, hx, hv = cb_rows(xs, valid, converge)
        if hx.shape[0] < 1:
            return False, np.zeros((0, 12)), np.zeros(0)
        HTH, HTh = hx.T @ hx, hx.T @ hv
        w, V = np.linalg.eigh(HTH)
        sq = np.sqrt(np.maximum(w, 0.0))
        S = np.zeros((23, 12)); h = np.zeros(23)
        S[:12] = sq[:, None] * V.T
        h[:12] = (V.T @ HTh) / np.maximum(sq, 1e-150)
        return True, S, h
    return cb


def test_surrogate_rows_drive_the_unmodified_update_to_the_same_state(oracle_lib, scene):
    orc = oracle_lib
    from fast_livo_amd import synth
    n, max_iter, R = 6000, 4, 0.001
    fr = synth.make_lio_frame(n, scene=scene)
    outs = []
    for make in (lambda c: _rows_callback(orc, fr, scene.map_xyz, c), lambda c: _surrogate_of(_rows_callback(orc, fr, scene.map_xyz, c))):
        cnt = dict(searches=0, calls=0)
        x = orc.state23_from_frame(fr, synth.quat_from_R)
        P = fr.cov23.copy()
        r = orc.ikfom_update_dyn_share(x, P, R, max_iter, make(cnt))
        outs.append((x.vec(), P.copy(), r["out"].iterations, cnt))
    (xa, Pa, ia, ca), (xb, Pb, ib, cb) = outs
    assert ia == ib and ca == cb and ia >= 2
    assert np.abs(xa - xb).max() <= 1e-9
    assert np.abs(Pa - Pb).max() <= 1e-9 * max(1.0, np.abs(Pa).max())
    # and the callback form of the update is the same function the whole-update oracle entry point runs
    x = orc.state23_from_frame(fr, synth.quat_from_R)
    P = fr.cov23.copy()

    def knn(w):
        nb, _, va, _ = orc.knn5_bruteforce(scene.map_xyz, w)
        return nb, va
    ro = orc.ikfom_update(x, P, fr.body_xyz, R, max_iter, knn)
    assert ro["out"].iterations == ia
    assert np.array_equal(x.vec(), xa) and np.array_equal(P, Pa)


def test_invalid_measurement_skips_the_iteration(oracle_lib, scene):
    """valid = false -> `continue` (esekfom.hpp:1649-1652): the state and covariance stay what they were."""
    orc = oracle_lib
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(500, scene=scene)
    x = orc.state23_from_frame(fr, synth.quat_from_R)
    x0 = x.vec().copy()
    P = fr.cov23.copy()
    P0 = P.copy()
    calls = []

    def cb(xs, valid, converge):
        calls.append((valid, converge))
        return False, np.zeros((0, 12)), np.zeros(0)
    r = orc.ikfom_update_dyn_share(x, P, 0.001, 3, cb)
    assert r["out"].iterations == 4 and all(v for v, _ in calls)      # valid is reset to true before every call (:1635)
    assert np.array_equal(x.vec(), x0) and np.array_equal(P, P0)
