"""CPU check of the claim the one-patch-per-lane VIO producers rest on (csrc/vio_kernels.h vio_produce_wide): the reference
forms, for every pixel of a patch, five bilinear values of its own (centre, left, right, up, down: lidar_selection.cpp:826-829,837), 320 per
patch; the producers form each DISTINCT value once -- 96 per patch, a 10 x 10 array without its corners -- because the value right of pixel
(x, y) is the centre value of pixel (x, y + 1): the same expression over the same four taps in the same order, hence the same float32 bits.
Both formulations are written out here in numpy float32 (IEEE, no contraction) over random taps and sub-pixel weights and compared bit for
bit: du, dv, the residual, and the reference's running float patch_error (:849, a double product added into a float)."""
import numpy as np


def _per_pixel(T, w, ref):
    """The reference's loop: T = 11 x 11 taps (row a, column b = image row v_i + (a - 5) s, column u_i + (b - 5) s), pixel (x, y) has its
    centre tap at T[x + 1, y + 1]."""
    wtl, wtr, wbl, wbr = w
    du = np.zeros((8, 8), np.float32); dv = np.zeros((8, 8), np.float32); res = np.zeros((8, 8), np.float32)
    half = np.float32(0.5)
    for x in range(8):
        for y in range(8):
            p = lambda dr, dc: T[x + 1 + dr, y + 1 + dc]          # img_ptr[dr * s * W + dc * s]
            du[x, y] = half * ((wtl * p(0, 1) + wtr * p(0, 2) + wbl * p(1, 1) + wbr * p(1, 2))
                               - (wtl * p(0, -1) + wtr * p(0, 0) + wbl * p(1, -1) + wbr * p(1, 0)))
            dv[x, y] = half * ((wtl * p(1, 0) + wtr * p(1, 1) + wbl * p(2, 0) + wbr * p(2, 1))
                               - (wtl * p(-1, 0) + wtr * p(-1, 1) + wbl * p(0, 0) + wbr * p(0, 1)))
            res[x, y] = wtl * p(0, 0) + wtr * p(0, 1) + wbl * p(1, 0) + wbr * p(1, 1) - ref[x * 8 + y]
    return du, dv, res


def _shared(T, w, ref):
    """The producers' walk: I[r, b] = the bilinear value of tap rows r, r + 1 at columns b, b + 1, formed once; pixel (x, y) reads
    I[x + 1, y + 1] (centre), I[x + 1, y] / I[x + 1, y + 2] (left / right), I[x, y + 1] / I[x + 2, y + 1] (up / down)."""
    wtl, wtr, wbl, wbr = w
    I = (wtl * T[:10, :10] + wtr * T[:10, 1:11] + wbl * T[1:11, :10] + wbr * T[1:11, 1:11]).astype(np.float32)
    half = np.float32(0.5)
    du = half * (I[1:9, 2:10] - I[1:9, 0:8])
    dv = half * (I[2:10, 1:9] - I[0:8, 1:9])
    res = I[1:9, 1:9] - ref.reshape(8, 8)
    return du.astype(np.float32), dv.astype(np.float32), res.astype(np.float32)


def _patch_error(res):
    pe = np.float32(0.0)
    for r in res.reshape(-1):
        rd = np.float64(r)
        pe = np.float32(rd * rd + np.float64(pe))              # float patch_error += double res * res
    return pe


def test_shared_bilinear_values_give_the_reference_bits():
    rng = np.random.default_rng(20260925)
    for trial in range(300):
        T = rng.integers(0, 256, (11, 11)).astype(np.float32)
        su, sv = np.float32(rng.uniform(0, 1)), np.float32(rng.uniform(0, 1))
        one = np.float64(1.0)
        # the weights as fl_patch_geom rounds them (lidar_selection.cpp:813-816: double products of float sub-pixel offsets)
        w = (np.float32((one - su) * (one - sv)), np.float32(su * (one - sv)), np.float32((one - su) * sv), np.float32(su * sv))
        ref = (rng.uniform(0, 255, 64) + rng.normal(0, 2, 64)).astype(np.float32)
        a = _per_pixel(T, w, ref)
        b = _shared(T, w, ref)
        for u, v in zip(a, b):
            assert np.array_equal(u.view(np.uint32), v.view(np.uint32)), trial
        assert _patch_error(a[2]).view(np.uint32) == _patch_error(b[2]).view(np.uint32)


def test_the_four_corner_values_are_never_read():
    """The 10 x 10 array of bilinear values is used without its corners (96 values): poison them and nothing changes."""
    rng = np.random.default_rng(7)
    T = rng.integers(0, 256, (11, 11)).astype(np.float32)
    w = tuple(np.float32(x) for x in (0.4, 0.3, 0.2, 0.1))
    ref = rng.uniform(0, 255, 64).astype(np.float32)
    base = _shared(T, w, ref)
    T2 = T.copy()
    T2[0, 0] = T2[0, 10] = T2[10, 0] = T2[10, 10] = np.float32(1e30)      # taps that only the corner values see
    other = _shared(T2, w, ref)
    for u, v in zip(base, other):
        assert np.array_equal(u.view(np.uint32), v.view(np.uint32))
