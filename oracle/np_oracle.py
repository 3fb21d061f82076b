"""Independent numpy/scipy restatement of the ESKF hot path.  TEST INFRASTRUCTURE ONLY.

Second opinion on oracle/*.c: written against the same
reference lines but with library linear algebra (scipy pivoted QR in float32, numpy.linalg.inv in
float64, vectorised numpy) instead of hand-restated loops, so a transcription slip in either shows
up as a disagreement far above rounding.  Rounding itself differs (LAPACK vs the Eigen-order
restatement), hence the cross-check tolerances in tests/test_cross_oracle_cpu.py are ~1e-5 relative.

Reference lines: src/laserMapping.cpp:1506-1695, include/common_lib.h:343-365,448-493,
include/so3_math.h:54-81, src/lidar_selection.cpp:743-902.
"""
from __future__ import annotations

import numpy as np
import scipy.linalg


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]], dtype=np.float64)


def Exp(v):
    n = np.linalg.norm(v)
    if n <= 1e-5:
        return np.eye(3)
    K = skew(v / n)
    return np.eye(3) + np.sin(n) * K + (1 - np.cos(n)) * K @ K


def Log(R):
    tr = np.trace(R)
    theta = 0.0 if tr > 3.0 - 1e-6 else np.arccos(0.5 * (tr - 1))
    K = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    return 0.5 * K if abs(theta) < 0.001 else 0.5 * theta / np.sin(theta) * K


def esti_plane(near):
    """common_lib.h:448-493 with scipy's column-pivoted QR in float32."""
    A = near.astype(np.float32)
    b = -np.ones(5, dtype=np.float32)
    Q, R, P = scipy.linalg.qr(A, mode="economic", pivoting=True)
    y = scipy.linalg.solve_triangular(R, Q.T @ b).astype(np.float32)
    x = np.zeros(3, dtype=np.float32)
    x[P] = y
    n = np.float32(np.linalg.norm(x))
    pabcd = np.concatenate([x / n, [np.float32(1.0) / n]]).astype(np.float32)
    ok = np.all(np.abs(A @ pabcd[:3] + pabcd[3]) <= np.float32(0.1))
    return pabcd, bool(ok)


def state_minus(Ra, xa, Rb, xb):
    """StatesGroup a - b (common_lib.h:354-365); x = concat(pos, vel, bg, ba, grav)."""
    return np.concatenate([Log(Rb.T @ Ra), xa - xb])


def solve18(R, x15, Rp, xp15, P, HTH, HTz, cov, sign):
    """laserMapping.cpp:1664-1683 with dense inverses."""
    H = np.zeros((18, 18))
    H[:6, :6] = HTH
    K1 = np.linalg.inv(H + np.linalg.inv(P / cov))
    G = np.zeros((18, 18))
    G[:, :6] = K1[:, :6] @ HTH
    vec = state_minus(Rp, xp15, R, x15)
    sol = sign * K1[:, :6] @ HTz + vec - G[:, :6] @ vec[:6]
    return sol, G


def lio18_iterate(R, p, rest12, Rp, pp, restp12, P, body, nbr, sel, R_LI, t_LI, cov):
    """One Mode-18 pass. Returns (solution, HTH, HTz, eff_mask, normvec)."""
    n = body.shape[0]
    pb = body.astype(np.float64)
    p_i = pb @ R_LI.T + t_LI
    pw = (p_i @ R.T + p).astype(np.float32)
    normvec = np.zeros((n, 4), dtype=np.float32)
    new_sel = np.zeros(n, dtype=bool)
    for i in np.nonzero(sel)[0]:
        pabcd, ok = esti_plane(nbr[i])
        if not ok:
            continue
        pd2 = np.float32(pabcd[:3] @ pw[i] + pabcd[3])
        s = np.float32(1 - 0.9 * abs(float(pd2)) / np.sqrt(np.linalg.norm(pb[i])))
        if float(s) > 0.9:
            new_sel[i] = True
            normvec[i] = [pabcd[0], pabcd[1], pabcd[2], pd2]
    eff = new_sel & (np.abs(normvec[:, 3].astype(np.float64)) <= 2.0)
    nv = normvec[eff, :3].astype(np.float64)
    C = nv @ R                                    # rows: (R^T n)^T
    A = np.cross(p_i[eff], C)
    H = np.concatenate([A, nv], axis=1)
    z = -normvec[eff, 3].astype(np.float64)
    HTH = H.T @ H
    HTz = H.T @ z
    x15 = np.concatenate([p, rest12])
    xp15 = np.concatenate([pp, restp12])
    sol, G = solve18(R, x15, Rp, xp15, P, HTH, HTz, cov, 1.0)
    return sol, HTH, HTz, eff, normvec, new_sel


def vio_iteration(vf, R, p, rest12, Rp, pp, restp12, P, level):
    """One accepted UpdateState iteration (lidar_selection.cpp:772-879). Returns (solution, error, HTH, HTz)."""
    Rli = vf.R_LI.T
    Pli = -vf.R_LI.T @ vf.t_LI
    Rci = vf.Rcl @ Rli
    Pci = vf.Rcl @ Pli + vf.Pcl
    Jdphi_dR = Rci
    Pic = -Rci.T @ Pci
    Jdp_dR = -Rci @ skew(Pic)
    fx, fy = abs(vf.cam["fx"]), abs(vf.cam["fy"])
    Rcw = Rci @ R.T
    Pcw = -Rci @ R.T @ p + Pci
    Jdp_dt = Rcw
    img = vf.img.astype(np.float32)
    Wd = vf.cam["width"]
    rows, zs = [], []
    from fast_livo_amd import synth
    for i in range(vf.m):
        scale = 1 << (level + int(vf.search_level[i]))
        pf = Rcw @ vf.pos[i] + Pcw
        pc = synth.world2cam(vf.cam, pf)
        zi = 1.0 / pf[2]
        Jdpi = np.array([[fx * zi, 0, -fx * pf[0] * zi * zi], [0, fy * zi, -fy * pf[1] * zi * zi]])
        p_hat = skew(pf)
        u_i = int(np.floor(np.float32(pc[0] / scale)) * scale)
        v_i = int(np.floor(np.float32(pc[1] / scale)) * scale)
        su = np.float32((np.float32(pc[0]) - u_i) / scale)
        sv = np.float32((np.float32(pc[1]) - v_i) / scale)
        wtl = np.float32((1.0 - su) * (1.0 - sv)); wtr = np.float32(su * (1.0 - sv))
        wbl = np.float32((1.0 - su) * sv); wbr = np.float32(su * sv)

        def I(r, c):
            return wtl * img[r, c] + wtr * img[r, c + scale] + wbl * img[r + scale, c] + wbr * img[r + scale, c + scale]
        for xr in range(8):
            for y in range(8):
                r0 = v_i + (xr - 4) * scale
                c0 = u_i + (y - 4) * scale
                du = np.float32(0.5) * (I(r0, c0 + scale) - I(r0, c0 - scale))
                dv = np.float32(0.5) * (I(r0 + scale, c0) - I(r0 - scale, c0))
                Jimg = np.array([float(du), float(dv)]) / scale
                Jdphi = Jimg @ Jdpi @ p_hat
                Jdp = -Jimg @ Jdpi
                JdR = Jdphi @ Jdphi_dR + Jdp @ Jdp_dR
                Jdt = Jdp @ Jdp_dt
                res = float(I(r0, c0) - vf.ref_patch[i, level, 8 * xr + y])
                rows.append(np.concatenate([JdR, Jdt]))
                zs.append(res)
    H = np.array(rows)
    z = np.array(zs)
    error = float(np.sum(z * z) / len(z))
    HTH = H.T @ H
    HTz = H.T @ z
    x15 = np.concatenate([p, rest12])
    xp15 = np.concatenate([pp, restp12])
    sol, G = solve18(R, x15, Rp, xp15, P, HTH, HTz, vf.img_point_cov, -1.0)
    return sol, error, HTH, HTz
