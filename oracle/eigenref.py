"""ctypes binding of oracle/_ref/libeigen_ref.so: the REFERENCE's own sources behind a C driver -- include/common_lib.h and
include/so3_math.h, the text of the Mode-18 loop (laserMapping.cpp:1506-1732), of LidarSelector::UpdateState / ComputeJ and of
ImuProcess::UndistortPcl, all over the reference's own ikd-Tree; with Boost + a real Eigen also use-ikfom.hpp and the IKFoM toolkit.
TEST INFRASTRUCTURE ONLY; only tests/ may import this module.

Recipe: oracle/ref_eigen/ (Makefile, ref_text.sh, text/, stubs/, eigen_driver.cpp, shim/).  `build()` runs it when /root/reference is
present; elsewhere the prebuilt library that travelled in oracle/_ref/ is used.  No Eigen is installed here: the recipe then compiles
against oracle/ref_eigen/shim (NOT Eigen; `linalg_kind()` returns "shim"), which pins the reference's logic and leaves Eigen's own
arithmetic unpinned.  `why_not()` returns the reason the tests print when they have to skip.
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
        L.ref_linalg_kind.restype = C.c_int
        vp = C.c_void_p
        L.ref_lio18_frame.argtypes = [vp, vp, C.c_int, vp, C.c_int, dp, dp, C.c_double, C.c_int, vp, vp, vp, vp]
        L.ref_vio_update_state.restype = C.c_float
        L.ref_vio_update_state.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_float, C.c_int, vp, vp, vp]
        L.ref_vio_compute_j.argtypes = [vp, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp]
        L.ref_imu_undistort.argtypes = [vp, vp, vp, C.c_int, C.c_double, vp, C.c_int, vp, vp, vp, vp]
        L.ref_vio_select.argtypes = [vp, dp, dp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_double, C.c_double, vp, vp, vp, vp, vp]
        L.ref_hsm_begin.argtypes = [vp, C.c_int, vp, C.c_int]
        L.ref_hsm_get.argtypes = [vp, vp, vp, vp]
        L.ref_localmap_create.argtypes = [vp, C.c_int, C.c_float, C.c_double, C.c_float]
        L.ref_localmap_fov_segment.argtypes = [dp, vp, vp, vp]
        L.ref_localmap_incremental.argtypes = [vp, dp, dp, vp, C.c_int, vp]
        L.ref_localmap_flatten.argtypes = [vp, C.c_int]
        L.ref_vmap_create.restype = C.c_void_p
        L.ref_vmap_create.argtypes = [vp, C.c_int]
        L.ref_vmap_destroy.argtypes = [vp]
        L.ref_vmap_size.argtypes = [vp]
        L.ref_vmap_get_point.argtypes = [vp, C.c_int, vp, vp, vp, vp]
        L.ref_vmap_get_grid.argtypes = [vp, vp, vp]
        L.ref_vmap_add_sparse.argtypes = [vp, dp, dp, vp, vp, C.c_int, C.c_int]
        L.ref_vmap_select.argtypes = [vp, dp, dp, vp, vp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, vp, vp, vp, vp, vp]
        L.ref_vmap_add_observation.argtypes = [vp, dp, dp, vp, C.c_int]
        L.ref_mtk_state_boxplus.argtypes = [dp, dp]
        L.ref_mtk_state_boxminus.argtypes = [dp, dp, dp]
        L.ref_mtk_A_matrix.argtypes = [dp, dp]
        L.ref_mtk_S2_Bx.argtypes = [dp, dp]
        L.ref_mtk_S2_Nx_yy.argtypes = [dp, dp]
        L.ref_mtk_S2_Mx.argtypes = [dp, dp, dp]
        L.ref_ikfom_update_text.argtypes = [dp, dp, C.c_double, C.c_int, dp, H_FN, C.c_void_p]
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


def linalg_kind():
    """'eigen' = the library was compiled against a real Eigen; 'shim' = against oracle/ref_eigen/shim (NOT Eigen: the reference's
    text runs, Eigen's own arithmetic does not)."""
    return "shim" if lib().ref_linalg_kind() else "eigen"


def lio18_frame(x, body, map_xyz, R_LI, t_LI, cov, max_iter):
    """The reference's Mode-18 loop text (laserMapping.cpp:1506-1732) over its own ikd-Tree built from map_xyz.  Mutates x
    (oracle.State18); returns the oracle's frame outputs."""
    from . import oracle as orc
    body = np.ascontiguousarray(body, dtype=np.float32)
    map_xyz = np.ascontiguousarray(map_xyz, dtype=np.float32)
    n = body.shape[0]
    out = orc.LioFrameOut()
    sel = np.zeros(n, dtype=np.uint8)
    normvec = np.zeros((n, 4), dtype=np.float32)
    world = np.zeros((n, 3), dtype=np.float32)
    R_LI = np.ascontiguousarray(R_LI, dtype=np.float64)
    t_LI = np.ascontiguousarray(t_LI, dtype=np.float64)
    st = lib().ref_lio18_frame(C.addressof(x), body.ctypes.data, n, map_xyz.ctypes.data, len(map_xyz), _d(R_LI), _d(t_LI), cov, max_iter,
                               sel.ctypes.data, normvec.ctypes.data, world.ctypes.data, C.addressof(out))
    return dict(status=st, out=out, sel=sel, normvec=normvec, world=world)


MP4_LIB_PATH = os.path.join(_DIR, "_ref", "libeigen_ref_mp4.so")
_lib_mp4 = None


def lib_mp4():
    """The Mode-18 loop text compiled as CMakeLists.txt:19-37 compiles it on a host with more than 5 cores (-DMP_EN -DMP_PROC_NUM=4): a
    library of its own (oracle/ref_eigen/Makefile step 4), or None."""
    global _lib_mp4
    if _lib_mp4 is None and build() is not None and os.path.exists(MP4_LIB_PATH):
        L = C.CDLL(MP4_LIB_PATH)
        vp, dp = C.c_void_p, C.POINTER(C.c_double)
        L.ref_lio18_frame.argtypes = [vp, vp, C.c_int, vp, C.c_int, dp, dp, C.c_double, C.c_int, vp, vp, vp, vp]
        L.ref_lio18_timers.argtypes = [dp]
        _lib_mp4 = L
    return _lib_mp4


def lio18_frame_timed(x, body, map_xyz, R_LI, t_LI, cov, max_iter, mp4=False):
    """lio18_frame + the reference's OWN timers of that frame (laserMapping.cpp:1508,1535-1554,1604-1605,1729): returns
    dict(iterations, searches, match_s, solve_s, kdtree_search_s (summed over the loop's threads), threads)."""
    from . import oracle as orc
    L = lib_mp4() if mp4 else lib()
    if L is None:
        return None
    if not mp4:
        L.ref_lio18_timers.argtypes = [C.POINTER(C.c_double)]
    body = np.ascontiguousarray(body, dtype=np.float32)
    map_xyz = np.ascontiguousarray(map_xyz, dtype=np.float32)
    out = orc.LioFrameOut()
    R_LI = np.ascontiguousarray(R_LI, dtype=np.float64)
    t_LI = np.ascontiguousarray(t_LI, dtype=np.float64)
    L.ref_lio18_frame(C.addressof(x), body.ctypes.data, body.shape[0], map_xyz.ctypes.data, len(map_xyz), _d(R_LI), _d(t_LI), cov, max_iter,
                      None, None, None, C.addressof(out))
    t = np.zeros(4)
    L.ref_lio18_timers(_d(t))
    return dict(iterations=int(out.iterations), searches=int(out.searches), match_s=t[0], solve_s=t[1], kdtree_search_s=t[2], threads=int(t[3]),
                effct_feat_num=int(out.effct_feat_num))


def vio_update_state(vf, x, x_prop, total_residual, level, G=None):
    """LidarSelector::UpdateState, the reference's text (lidar_selection.cpp:743-902).  Mutates x and G."""
    from . import oracle as orc
    cfg = orc.vio_config(vf)
    G = np.zeros((18, 18)) if G is None else G
    errors = np.zeros(vf.m, dtype=np.float32)
    HTH = np.zeros((6, 6))
    err = lib().ref_vio_update_state(C.addressof(cfg), C.addressof(x), C.addressof(x_prop), vf.img.ctypes.data, vf.ref_patch.ctypes.data,
                                     vf.pos.ctypes.data, vf.search_level.ctypes.data, vf.m, total_residual, level, errors.ctypes.data,
                                     G.ctypes.data, HTH.ctypes.data)
    return dict(error=err, errors=errors, G=G, HTH=HTH)


def vio_compute_j(vf, x, x_prop):
    """LidarSelector::ComputeJ, the reference's text (lidar_selection.cpp:967-983 over :743-911).  Mutates x."""
    from . import oracle as orc
    cfg = orc.vio_config(vf)
    errors = np.zeros(vf.m, dtype=np.float32)
    Tcw = np.zeros(12)
    st = lib().ref_vio_compute_j(C.addressof(cfg), C.addressof(x), C.addressof(x_prop), vf.img.ctypes.data, vf.ref_patch.ctypes.data,
                                 vf.pos.ctypes.data, vf.search_level.ctypes.data, vf.m, errors.ctypes.data, Tcw.ctypes.data)
    return dict(status=st, errors=errors, Tcw=Tcw)


def imu_undistort(proc, state, imu, pcl_beg_time, pts_xyzt):
    """ImuProcess::UndistortPcl, the reference's text (IMU_Processing.cpp:611-809), on one whole scan (is_lidar_end).  Mutates proc
    (oracle.ImuProc) and state (oracle.State18).  Returns (compensated cloud of the KEPT points (k, 4), poses, pcl_end_time): the
    reference derives pcl_end_time from the last point and may leave the last point(s) out (see text/imu_2.inc)."""
    from . import oracle as orc
    samples = imu if not isinstance(imu, np.ndarray) else orc.imu_samples(imu)
    k = len(samples)
    pts = np.array(pts_xyzt, dtype=np.float32, order="C", copy=True)
    poses = (orc.Pose6d * (k + 1))()
    npz, kept, t_end = C.c_int32(0), C.c_int32(0), C.c_double(0.0)
    rc = lib().ref_imu_undistort(C.addressof(proc), C.addressof(state), C.addressof(samples), k, pcl_beg_time, pts.ctypes.data, pts.shape[0],
                                 C.addressof(poses), C.addressof(npz), C.addressof(kept), C.addressof(t_end))
    if rc != 0:
        raise RuntimeError("ref_imu_undistort failed")
    return pts[:kept.value], [poses[i] for i in range(npz.value)], t_end.value


def vio_select(cfg, Rcw, Pcw, cur_img, keyframes, depth, cand, ncc_en=False, ncc_thre=0.0, outlier_threshold=300.0):
    """The pixel-level part of LidarSelector::addFromSparseMap, the reference's text (lidar_selection.cpp:476-582 over getpatch,
    getWarpMatrixAffine, warpAffine, NCC, getBestSearchLevel).  Same inputs and outputs as oracle.vio_select (without `reason`)."""
    Rcw = np.ascontiguousarray(Rcw, np.float64).reshape(9)
    Pcw = np.ascontiguousarray(Pcw, np.float64)
    cur = np.ascontiguousarray(cur_img, np.uint8)
    kfs = [np.ascontiguousarray(k, np.uint8) for k in keyframes]
    ptrs = (C.c_void_p * len(kfs))(*[k.ctypes.data for k in kfs])
    depth = np.ascontiguousarray(depth, np.float32)
    m = len(cand)
    idx = np.zeros(max(m, 1), np.int32); err = np.zeros(max(m, 1), np.float32); lvl = np.zeros(max(m, 1), np.int32)
    patches = np.zeros((max(m, 1), 192), np.float32)
    na = C.c_int32(0)
    rc = lib().ref_vio_select(C.addressof(cfg), _d(Rcw), _d(Pcw), cur.ctypes.data, C.addressof(ptrs), depth.ctypes.data, C.addressof(cand), m,
                              1 if ncc_en else 0, ncc_thre, outlier_threshold, idx.ctypes.data, patches.ctypes.data, err.ctypes.data,
                              lvl.ctypes.data, C.addressof(na))
    if rc != 0:
        raise RuntimeError("ref_vio_select failed: %d" % rc)
    k = na.value
    return dict(idx=idx[:k].copy(), errors=err[:k].copy(), levels=lvl[:k].copy(), patches=patches[:k].copy())


class HShareModel:
    """h_share_model, the reference's text (laserMapping.cpp:961-1093) over the reference's own ikd-Tree, for one scan and one map: ONE per
    process at a time.  `callback` is a C function pointer of the updaters' callback shape -- hand it to `ikfom_update_text_c` (the
    reference's updater text) or call it through `rows()`."""

    def __init__(self, body_xyz, map_xyz):
        self.L = lib()
        self.body = np.ascontiguousarray(body_xyz, np.float32).reshape(-1, 3)
        m = np.ascontiguousarray(map_xyz, np.float32).reshape(-1, 3)
        if self.L.ref_hsm_begin(self.body.ctypes.data, len(self.body), m.ctypes.data, len(m)) != 0:
            raise RuntimeError("another HShareModel is alive in this process")
        self.open = True
        self.callback = C.cast(self.L.ref_hsm_callback, H_FN)

    def close(self):
        if self.open:
            self.L.ref_hsm_end()
            self.open = False

    def rows(self, s26, converge):
        """One call.  Returns (valid, h_x (rows, 12), h (rows,))."""
        st = np.array(s26, dtype=np.float64).copy()
        valid, conv, rows = C.c_int(1), C.c_int(1 if converge else 0), C.c_int(0)
        hx, h = C.POINTER(C.c_double)(), C.POINTER(C.c_double)()
        self.callback(None, _d(st), C.byref(valid), C.byref(conv), C.byref(rows), C.byref(hx), C.byref(h))
        r = rows.value
        return bool(valid.value), np.ctypeslib.as_array(hx, shape=(r, 12)).copy() if r else np.zeros((0, 12)), \
            np.ctypeslib.as_array(h, shape=(r,)).copy() if r else np.zeros(0)

    def last(self):
        n = len(self.body)
        sel = np.zeros(n, np.uint8); nv = np.zeros((n, 4), np.float32); world = np.zeros((n, 3), np.float32); tr = C.c_double(0)
        eff = self.L.ref_hsm_get(sel.ctypes.data, nv.ctypes.data, world.ctypes.data, C.addressof(tr))
        return dict(sel=sel, normvec=nv, world=world, effct_feat_num=eff, total_residual=tr.value)


def ikfom_update_text_c(s26, P, R, max_iter, c_callback, limit=None):
    """The reference's updater text around a C callback (e.g. HShareModel.callback: then BOTH halves of the Mode-23 update are the
    reference's text).  Returns (state26, P, calls)."""
    limit = np.full(23, 0.001) if limit is None else np.ascontiguousarray(limit, dtype=np.float64)
    s = np.array(s26, dtype=np.float64).copy()
    P = np.array(P, dtype=np.float64).copy()
    calls = lib().ref_ikfom_update_text(_d(s), _d(P), R, max_iter, _d(limit), c_callback, None)
    return s, P, calls


class LocalMap:
    """lasermap_fov_segment + map_incremental, the reference's text (laserMapping.cpp:361-421, 692-706) over a persistent ikd-Tree of the
    reference.  The window is a file-scope variable of the reference: ONE LocalMap per process at a time."""

    def __init__(self, map_xyz, downsample, cube_len, det_range):
        self.L = lib()
        m = np.ascontiguousarray(map_xyz, np.float32).reshape(-1, 3)
        if self.L.ref_localmap_create(m.ctypes.data, len(m), downsample, cube_len, det_range) != 0:
            raise RuntimeError("another LocalMap is alive in this process")
        self.open = True

    def close(self):
        if self.open:
            self.L.ref_localmap_destroy()
            self.open = False

    def fov_segment(self, pos):
        """Returns (window float32[6], boxes (nb, 6), points deleted)."""
        pos = np.ascontiguousarray(pos, np.float64)
        win = np.zeros(6, np.float32); boxes = np.zeros((3, 6), np.float32); deleted = C.c_int32(0)
        nb = self.L.ref_localmap_fov_segment(_d(pos), win.ctypes.data, boxes.ctypes.data, C.addressof(deleted))
        return win, boxes[:nb].copy(), deleted.value

    def incremental(self, x, R_LI, t_LI, body):
        """Returns (points the tree grew by, feats_down_world (n, 3))."""
        R_LI = np.ascontiguousarray(R_LI, np.float64); t_LI = np.ascontiguousarray(t_LI, np.float64)
        body = np.ascontiguousarray(body, np.float32).reshape(-1, 3)
        world = np.zeros_like(body)
        grew = self.L.ref_localmap_incremental(C.addressof(x), _d(R_LI), _d(t_LI), body.ctypes.data, len(body), world.ctypes.data)
        return grew, world

    def flatten(self):
        cap = 1 << 20
        out = np.zeros((cap, 3), np.float32)
        n = self.L.ref_localmap_flatten(out.ctypes.data, cap)
        return out[:n].copy()


class VMap:
    """The visual map of LidarSelector, the reference's text (addSparseMap / AddPoint / addFromSparseMap / addObservation over the
    reference's Feature and Point), driven as detect() drives it.  Same methods as oracle.VMap; images handed in must stay alive
    as long as the map (the reference's Features keep a header over their frame's pixels)."""

    def __init__(self, cfg, grid_size):
        self.L, self.cfg = lib(), cfg
        self.cells = (cfg.width // grid_size) * (cfg.height // grid_size)
        self.m = self.L.ref_vmap_create(C.addressof(cfg), grid_size)
        self._keep = []

    def close(self):
        if self.m:
            self.L.ref_vmap_destroy(self.m)
            self.m = None

    def size(self):
        return self.L.ref_vmap_size(self.m)

    def get_point(self, i):
        from . import oracle as orc
        pos = np.zeros(3, np.float64); val = C.c_float(0); nobs = C.c_int32(0)
        obs = (orc.VmapObs * 20)()
        assert self.L.ref_vmap_get_point(self.m, i, pos.ctypes.data, C.addressof(val), C.addressof(nobs), C.addressof(obs)) == 0
        return pos, val.value, [obs[k] for k in range(nobs.value)]

    def get_grid(self):
        mv = np.zeros(self.cells, np.float32); gn = np.zeros(self.cells, np.int32)
        self.L.ref_vmap_get_grid(self.m, mv.ctypes.data, gn.ctypes.data)
        return mv, gn

    def _pose(self, Rcw, Pcw, img):
        Rcw = np.ascontiguousarray(Rcw, np.float64).reshape(9); Pcw = np.ascontiguousarray(Pcw, np.float64)
        img = np.ascontiguousarray(img, np.uint8)
        self._keep.append(img)
        return Rcw, Pcw, img

    def add_sparse(self, Rcw, Pcw, img, scan_world, kf_id, frame_id):
        Rcw, Pcw, img = self._pose(Rcw, Pcw, img)
        scan = np.ascontiguousarray(scan_world, np.float32).reshape(-1, 3)
        return self.L.ref_vmap_add_sparse(self.m, _d(Rcw), _d(Pcw), img.ctypes.data, scan.ctypes.data, len(scan), frame_id)

    def select(self, Rcw, Pcw, cur_img, keyframes, scan_down_world, ncc_en=False, ncc_thre=0.0, outlier_threshold=300.0, frame_id=-1):
        Rcw, Pcw, cur = self._pose(Rcw, Pcw, cur_img)
        scan = np.ascontiguousarray(scan_down_world, np.float32).reshape(-1, 3)
        m = self.cells
        sel = np.zeros(m, np.int32); err = np.zeros(m, np.float32); lvl = np.zeros(m, np.int32); patches = np.zeros((m, 192), np.float32)
        ns = C.c_int32(0)
        rc = self.L.ref_vmap_select(self.m, _d(Rcw), _d(Pcw), cur.ctypes.data, scan.ctypes.data, len(scan), 1 if ncc_en else 0, ncc_thre,
                                    outlier_threshold, frame_id, sel.ctypes.data, err.ctypes.data, lvl.ctypes.data, patches.ctypes.data,
                                    C.addressof(ns))
        if rc != 0:
            raise RuntimeError("ref_vmap_select failed: %d" % rc)
        k = ns.value
        return dict(points=sel[:k].copy(), errors=err[:k].copy(), levels=lvl[:k].copy(), patches=patches[:k].copy())

    def add_observation(self, Rcw, Pcw, img, sel_points, levels, kf_id, frame_id):
        """sel_points / levels are what the last select() left in the reference's sub_sparse_map (arguments kept for oracle.VMap's shape)."""
        Rcw, Pcw, img = self._pose(Rcw, Pcw, img)
        return self.L.ref_vmap_add_observation(self.m, _d(Rcw), _d(Pcw), img.ctypes.data, frame_id)


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


def mtk_boxplus(s26, d23):
    """state_ikfom::boxplus through the toolkit's own text (vect / SO3 / S2 boxplus, MTK::exp, cos_sinc_sqrt)."""
    s = np.array(s26, dtype=np.float64).copy()
    d = np.ascontiguousarray(d23, dtype=np.float64)
    lib().ref_mtk_state_boxplus(_d(s), _d(d))
    return s


def mtk_boxminus(s26, o26):
    s = np.ascontiguousarray(s26, dtype=np.float64); o = np.ascontiguousarray(o26, dtype=np.float64)
    out = np.zeros(23)
    lib().ref_mtk_state_boxminus(_d(s), _d(o), _d(out))
    return out


def mtk_A_matrix(v):
    v = np.ascontiguousarray(v, dtype=np.float64); out = np.zeros(9)
    lib().ref_mtk_A_matrix(_d(v), _d(out))
    return out.reshape(3, 3)


def mtk_S2(vec, delta=None):
    """(Bx 3x2, Nx_yy 2x3, Mx 3x2 or None) of an S2 element with the given vector (not renormalised)."""
    vec = np.ascontiguousarray(vec, dtype=np.float64)
    bx, nx, mx = np.zeros(6), np.zeros(6), np.zeros(6)
    lib().ref_mtk_S2_Bx(_d(vec), _d(bx))
    lib().ref_mtk_S2_Nx_yy(_d(vec), _d(nx))
    if delta is not None:
        delta = np.ascontiguousarray(delta, dtype=np.float64)
        lib().ref_mtk_S2_Mx(_d(vec), _d(delta), _d(mx))
    return bx.reshape(3, 2), nx.reshape(2, 3), (mx.reshape(3, 2) if delta is not None else None)


def ikfom_update_text(s26, P, R, max_iter, h_dyn_share, limit=None):
    """The TEXT of esekf::update_iterated_dyn_share_modified (esekfom.hpp:1619-1928) around a Python callback, over the toolkit's own
    SO3 / S2 / vect / mtkmath text; only the Boost-generated compound state and vectview are stand-ins (oracle/ref_eigen/text/ikf_1.inc,
    ikf_1c.inc).  Same contract as ikfom_update_dyn_share below; needs neither Boost nor a real Eigen."""
    return _ikfom_update(lib().ref_ikfom_update_text, s26, P, R, max_iter, h_dyn_share, limit)


def ikfom_update_dyn_share(s26, P, R, max_iter, h_dyn_share, limit=None, state_cls=None):
    """The reference's unmodified update_iterated_dyn_share_modified (its own MTK state type; needs Boost + Eigen) around a Python callback
    of the oracle's shape: h_dyn_share(state (oracle.State23), valid, converge) -> (valid, h_x (rows, 12), h (rows,)).
    Returns (state26, P, calls)."""
    return _ikfom_update(lib().ref_ikfom_update_dyn_share, s26, P, R, max_iter, h_dyn_share, limit)


def _ikfom_update(entry, s26, P, R, max_iter, h_dyn_share, limit):
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
    calls = entry(_d(s), _d(P), R, max_iter, _d(limit), H_FN(cb), None)
    return s, P, calls
