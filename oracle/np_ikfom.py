"""Independent numpy restatement of the Mode-23 (IKFoM) update.  TEST INFRASTRUCTURE ONLY.

Second opinion on oracle/orc_ikfom.c, written from the reference headers alone (not from the C restatement) with library linear
algebra (numpy.linalg.inv, matrix products on whole blocks) where the C file has hand-written loops, so that a transcription slip
in either shows up far above rounding (tests/test_cross_oracle_cpu.py: state <= 1e-9, covariance <= 1e-10).

Reference lines followed:
  update_iterated_dyn_share_modified   include/IKFoM_toolkit/esekfom/esekfom.hpp:1619-1928
  state_ikfom layout                   include/use-ikfom.hpp:6-21   (pos 0, rot 3, offset_R_L_I 6, offset_T_L_I 9, vel 12, bg 15, ba 18, grav 21)
  compound boxplus / boxminus          include/IKFoM_toolkit/mtk/build_manifold.hpp:101-103,192-200
  SO3 boxplus / boxminus / exp / log   include/IKFoM_toolkit/mtk/types/SOn.hpp:233-239,284-297
  S2 boxplus / boxminus / Bx / Nx_yy / Mx   include/IKFoM_toolkit/mtk/types/S2.hpp:136-167,215-231,259-280
  cos_sinc_sqrt, exp, log, A_matrix    include/IKFoM_toolkit/mtk/src/mtkmath.hpp:142-174,236-288
  h_share_model                        src/laserMapping.cpp:961-1093
"""
from __future__ import annotations

import math

import numpy as np

TOL = 1e-11                      # MTK::tolerance<double>(), mtkmath.hpp:122
S2_LENGTH = 98090 / 10000        # S2<double, 98090, 10000, 1>, use-ikfom.hpp:8 ; S2.hpp: length = den / num
N = 23
SO3_IDX = (3, 6)                 # rot, offset_R_L_I
S2_IDX = 21                      # grav

_EPS = np.finfo(np.float64).eps
_TAYLOR_N_BOUND = math.sqrt(math.sqrt(_EPS))     # mtkmath.hpp:147-149


def hat(v):
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


# ------------------------------------------------------------------ quaternions, stored Eigen-style (x, y, z, w)
def qmul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw,
                     aw * bw - ax * bx - ay * by - az * bz])


def qconj(q):
    return np.array([-q[0], -q[1], -q[2], q[3]])


def qmat(q):
    """Eigen::QuaternionBase::toRotationMatrix (no normalisation)."""
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


# ------------------------------------------------------------------ mtkmath.hpp
def cos_sinc_sqrt(x2):
    if x2 >= _TAYLOR_N_BOUND:
        x = math.sqrt(x2)
        return math.cos(x), math.sin(x) / x
    inv = [1 / 3., 1 / 4., 1 / 5., 1 / 6., 1 / 7., 1 / 8., 1 / 9.]
    cosi, sinc = 1.0, 1.0
    term = -1 / 2. * x2
    for i in range(3):
        cosi += term
        term *= inv[2 * i]
        sinc += term
        term *= -inv[2 * i + 1] * x2
    return cosi, sinc


def mtk_exp(vec, scale):
    """MTK::exp (mtkmath.hpp:249-256): returns the quaternion (x, y, z, w) = (sinc * scale * vec, cos)."""
    c, s = cos_sinc_sqrt(scale * scale * float(vec @ vec))
    return np.array([*(s * scale * np.asarray(vec, dtype=np.float64)), c])


def mtk_log(q, scale, plus_minus_periodicity):
    """MTK::log (mtkmath.hpp:268-288)."""
    vec, w = q[:3], q[3]
    nv = float(np.linalg.norm(vec))
    if nv < TOL:
        if (not plus_minus_periodicity) and w < 0:
            i = int(np.argmax(np.abs(vec)))
            out = np.zeros(3)
            out[i] = scale * math.atan2(abs(vec[i]), w)
            return out
        nv = TOL
    s = scale / nv * (math.atan(nv / w) if plus_minus_periodicity else math.atan2(nv, w))
    return s * vec


def A_matrix(v):
    sq = v[0] * v[0] + v[1] * v[1] + v[2] * v[2]
    nrm = math.sqrt(sq)
    if nrm < TOL:
        return np.eye(3)
    K = hat(v)
    return np.eye(3) + (1 - math.cos(nrm)) / sq * K + (1 - math.sin(nrm) / nrm) / sq * (K @ K)


# ------------------------------------------------------------------ SO3 (SOn.hpp)
def so3_boxplus(q, v, scale=1.0):
    return qmul(q, mtk_exp(v, scale / 2))


def so3_boxminus(q, other):
    return mtk_log(qmul(qconj(other), q), 2.0, True)


# ------------------------------------------------------------------ S2 (S2.hpp), S2_typ = 1
def s2_Bx(vec):
    L = S2_LENGTH
    if vec[0] + L > TOL:
        d = L + vec[0]
        res = np.array([[-vec[1], -vec[2]],
                        [L - vec[1] * vec[1] / d, -vec[2] * vec[1] / d],
                        [-vec[2] * vec[1] / d, L - vec[2] * vec[2] / d]])
        return res / L
    res = np.zeros((3, 2))
    res[1, 1] = -1
    res[2, 0] = 1
    return res


def s2_boxplus(vec, delta, scale=1.0):
    Bu = s2_Bx(vec) @ delta
    return qmat(mtk_exp(Bu, scale / 2)) @ vec


def s2_boxminus(vec, other):
    v_sin = float(np.linalg.norm(hat(vec) @ other))
    v_cos = float(vec @ other)
    theta = math.atan2(v_sin, v_cos)
    if v_sin < TOL:
        return np.array([3.1415926, 0.0]) if abs(theta) > TOL else np.zeros(2)
    return theta / v_sin * (s2_Bx(other).T @ (hat(other) @ vec))


def s2_Nx_yy(vec):
    return (1 / S2_LENGTH / S2_LENGTH) * (s2_Bx(vec).T @ hat(vec))


def s2_Mx(vec, delta):
    Bx = s2_Bx(vec)
    if float(np.linalg.norm(delta)) < TOL:
        return -hat(vec) @ Bx
    Bu = Bx @ delta
    # S2.hpp:277: `scalar(1/2)` is an integer division -> scale 0 -> exp_delta is the identity rotation.  Kept, not repaired.
    exp_delta = mtk_exp(Bu, float(1 // 2))
    return -qmat(exp_delta) @ hat(vec) @ A_matrix(Bu).T @ Bx


# ------------------------------------------------------------------ state_ikfom
class State:
    __slots__ = ("pos", "rot", "offset_R_L_I", "offset_T_L_I", "vel", "bg", "ba", "grav")

    def __init__(self, pos, rot, offset_R_L_I, offset_T_L_I, vel, bg, ba, grav):
        f = lambda a: np.array(a, dtype=np.float64)
        self.pos, self.rot, self.offset_R_L_I, self.offset_T_L_I = f(pos), f(rot), f(offset_R_L_I), f(offset_T_L_I)
        self.vel, self.bg, self.ba, self.grav = f(vel), f(bg), f(ba), f(grav)

    @classmethod
    def from_c(cls, s):
        """from oracle.State23 (same field names, quaternions x, y, z, w)"""
        return cls(*(list(getattr(s, k)) for k in cls.__slots__))

    def to_c(self, s):
        for k in self.__slots__:
            v = getattr(self, k)
            for i in range(len(v)):
                getattr(s, k)[i] = float(v[i])

    def copy(self):
        return State(*(getattr(self, k) for k in self.__slots__))

    def vec(self):
        return np.concatenate([getattr(self, k) for k in self.__slots__])

    def boxplus(self, d):
        self.pos = self.pos + d[0:3]
        self.rot = so3_boxplus(self.rot, d[3:6])
        self.offset_R_L_I = so3_boxplus(self.offset_R_L_I, d[6:9])
        self.offset_T_L_I = self.offset_T_L_I + d[9:12]
        self.vel = self.vel + d[12:15]
        self.bg = self.bg + d[15:18]
        self.ba = self.ba + d[18:21]
        self.grav = s2_boxplus(self.grav, d[21:23])

    def boxminus(self, o):
        return np.concatenate([self.pos - o.pos, so3_boxminus(self.rot, o.rot), so3_boxminus(self.offset_R_L_I, o.offset_R_L_I),
                               self.offset_T_L_I - o.offset_T_L_I, self.vel - o.vel, self.bg - o.bg, self.ba - o.ba,
                               s2_boxminus(self.grav, o.grav)])


def _left_right(P, J, idx):
    """rows idx.. <- J rows ; then columns idx.. <- columns J^T   (esekfom.hpp:1666-1671, 1689-1694)"""
    k = J.shape[0]
    P[idx:idx + k, :] = J @ P[idx:idx + k, :]
    P[:, idx:idx + k] = P[:, idx:idx + k] @ J.T


def update_iterated_dyn_share_modified(x, P, R, maximum_iter, limit, h_dyn_share):
    """esekfom.hpp:1619-1928.  x: State (updated in place), P: (23, 23) (updated in place).
    h_dyn_share(x, valid, converge) -> (valid, h_x (rows, 12), h (rows,)).  Returns a dict of counters and the last dx_."""
    x_prop = x.copy()
    P_prop = P.copy()
    t = 0
    converge = True
    calls = searches = 0
    dx_ = np.zeros(N)
    I = np.eye(N)
    K_x = np.zeros((N, N))
    for i in range(-1, maximum_iter):
        calls += 1
        searches += int(converge)
        valid, h_x, h = h_dyn_share(x, True, converge)
        h_x = np.asarray(h_x, dtype=np.float64).reshape(-1, 12)
        h = np.asarray(h, dtype=np.float64).reshape(-1)
        dof = h_x.shape[0]
        dx = x.boxminus(x_prop)
        dx_new = dx.copy()
        if not valid:
            continue
        Pw = P_prop.copy()
        for idx in SO3_IDX:
            J = A_matrix(dx[idx:idx + 3]).T
            dx_new[idx:idx + 3] = J @ dx_new[idx:idx + 3]
            _left_right(Pw, J, idx)
        J2 = s2_Nx_yy(x.grav) @ s2_Mx(x_prop.grav, dx[S2_IDX:S2_IDX + 2])
        dx_new[S2_IDX:S2_IDX + 2] = J2 @ dx_new[S2_IDX:S2_IDX + 2]
        _left_right(Pw, J2, S2_IDX)

        if N > dof:                                                                  # :1712-1741
            Hc = np.zeros((dof, N))
            Hc[:, :12] = h_x
            K = Pw @ Hc.T @ np.linalg.inv(Hc @ Pw @ Hc.T / R + np.eye(dof)) / R
            K_h = K @ h
            K_x = K @ Hc
        else:                                                                        # :1779-1806
            Pt = np.linalg.inv(Pw / R)
            HTH = h_x.T @ h_x
            Pt[:12, :12] += HTH
            Pi = np.linalg.inv(Pt)
            K_h = Pi[:, :12] @ (h_x.T @ h)
            K_x = np.zeros((N, N))
            K_x[:, :12] = Pi[:, :12] @ HTH

        dx_ = K_h + (K_x - I) @ dx_new
        x.boxplus(dx_)
        converge = bool(np.all(np.abs(dx_) <= limit))
        if converge:
            t += 1
        if t == 0 and i == maximum_iter - 2:
            converge = True
        if t > 1 or i == maximum_iter - 1:                                          # :1831-1921
            L = Pw.copy()
            for idx in SO3_IDX:
                J = A_matrix(dx_[idx:idx + 3]).T
                L[idx:idx + 3, :] = J @ Pw[idx:idx + 3, :]
                K_x[idx:idx + 3, :12] = J @ K_x[idx:idx + 3, :12]
                L[:, idx:idx + 3] = L[:, idx:idx + 3] @ J.T
                Pw[:, idx:idx + 3] = Pw[:, idx:idx + 3] @ J.T
            J2 = s2_Nx_yy(x.grav) @ s2_Mx(x_prop.grav, dx_[S2_IDX:S2_IDX + 2])
            L[S2_IDX:S2_IDX + 2, :] = J2 @ Pw[S2_IDX:S2_IDX + 2, :]
            K_x[S2_IDX:S2_IDX + 2, :12] = J2 @ K_x[S2_IDX:S2_IDX + 2, :12]
            L[:, S2_IDX:S2_IDX + 2] = L[:, S2_IDX:S2_IDX + 2] @ J2.T
            Pw[:, S2_IDX:S2_IDX + 2] = Pw[:, S2_IDX:S2_IDX + 2] @ J2.T
            P[:, :] = L - K_x[:, :12] @ Pw[:12, :]
            return dict(iterations=calls, searches=searches, dx=dx_, finished=True)
        P[:, :] = Pw          # P_ of the filter object is the working copy between iterations (esekfom.hpp:1657)
    return dict(iterations=calls, searches=searches, dx=dx_, finished=False)


# ------------------------------------------------------------------ h_share_model rows (laserMapping.cpp:961-1093)
def h_share_model_rows(x, body_xyz, nbr, sel, esti_plane):
    """Rows for the points with sel[i] set (neighbours nbr[i] (5, 3) float32 already found).  `esti_plane(near) -> (pabcd f32[4], ok)`
    is passed in (np_oracle.esti_plane, or the C oracle's, when the caller wants to separate the plane fit's rounding from the rest).
    Returns (h_x (n_eff, 12), h (n_eff,), new_sel, eff_mask)."""
    n = body_xyz.shape[0]
    R, Roff = qmat(x.rot), qmat(x.offset_R_L_I)
    pb = body_xyz.astype(np.float64)
    p_i = pb @ Roff.T + x.offset_T_L_I
    pw = (p_i @ R.T + x.pos).astype(np.float32)
    new_sel = np.zeros(n, dtype=bool)
    res_last = np.zeros(n)
    nv = np.zeros((n, 4), dtype=np.float32)
    for i in np.nonzero(sel)[0]:
        pabcd, ok = esti_plane(nbr[i])
        if not ok:
            continue
        pd2 = np.float32(np.float32(np.float32(pabcd[0] * pw[i, 0]) + np.float32(pabcd[1] * pw[i, 1])) + np.float32(pabcd[2] * pw[i, 2])) + pabcd[3]
        pd2 = np.float32(pd2)
        s = np.float32(1 - 0.9 * abs(float(pd2)) / math.sqrt(float(np.linalg.norm(pb[i]))))
        if s > 0.9:
            new_sel[i] = True
            nv[i, :3] = pabcd[:3]
            nv[i, 3] = pd2
            res_last[i] = abs(float(pd2))
    eff = new_sel & (res_last <= 2.0)
    rows = []
    hv = []
    for i in np.nonzero(eff)[0]:
        n_vec = nv[i, :3].astype(np.float64)
        C = R.T @ n_vec
        A = hat(p_i[i]) @ C
        B = hat(pb[i]) @ (Roff.T @ C)
        rows.append(np.concatenate([n_vec, A, B, C]))
        hv.append(-float(nv[i, 3]))
    return np.array(rows).reshape(-1, 12), np.array(hv), new_sel, eff
