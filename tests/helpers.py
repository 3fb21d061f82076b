"""Shared helpers for the parity tests."""
import ctypes as C

import numpy as np


def p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


def sums_to_HTH(sums):
    S = np.zeros((6, 6))
    k = 0
    for i in range(6):
        for j in range(i, 6):
            S[i, j] = S[j, i] = sums[k]
            k += 1
    return S, np.array(sums[21:27])


def copy_state(dst_cls, src):
    """Copy an oracle State18 into a capi State18 (identical layout) or vice versa."""
    d = dst_cls()
    C.memmove(C.byref(d), C.byref(src), C.sizeof(dst_cls))
    return d


# Stated tolerances (BASELINE.md section 2 / SURVEY.md 8c)
TOL_DELTA_ABS = 1e-9      # |delta_gpu - delta_cpu|_inf <= 1e-9 * max(1, |delta_cpu|_inf)
TOL_SUMS_REL = 1e-12      # reduced sums: fp64 re-ordering error


def assert_delta_close(d_gpu, d_cpu, tol=TOL_DELTA_ABS):
    d_gpu = np.asarray(d_gpu, dtype=np.float64)
    d_cpu = np.asarray(d_cpu, dtype=np.float64)
    bound = tol * max(1.0, np.abs(d_cpu).max())
    err = np.abs(d_gpu - d_cpu).max()
    assert err <= bound, f"state delta differs: {err:.3e} > {bound:.3e}"
    return err
