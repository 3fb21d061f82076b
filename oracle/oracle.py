"""ctypes binding of the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Parity status (what is pinned to the reference's ikd-Tree, what to the reference's own text, what stays unpinned): oracle/fastlivo_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_DIR, "liboracle.so")


def build(force=False):
    srcs = [os.path.join(_DIR, f) for f in os.listdir(_DIR) if f.endswith((".c", ".h")) or f == "Makefile"]
    if (not force and os.path.exists(LIB_PATH)
            and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return LIB_PATH
    subprocess.check_call(["make", "-C", _DIR, "-s", "liboracle.so"])
    return LIB_PATH


class State18(C.Structure):
    _fields_ = [("rot", C.c_double * 9), ("pos", C.c_double * 3), ("vel", C.c_double * 3),
                ("bg", C.c_double * 3), ("ba", C.c_double * 3), ("grav", C.c_double * 3),
                ("cov", C.c_double * 324)]

    @staticmethod
    def make(R, p, vel, bg, ba, grav, cov):
        s = State18()
        s.rot[:] = np.asarray(R, dtype=np.float64).reshape(9)
        s.pos[:] = p
        s.vel[:] = vel
        s.bg[:] = bg
        s.ba[:] = ba
        s.grav[:] = grav
        s.cov[:] = np.asarray(cov, dtype=np.float64).reshape(324)
        return s

    def copy(self):
        o = State18()
        C.memmove(C.byref(o), C.byref(self), C.sizeof(State18))
        return o

    def vec(self):
        """rot(9), pos, vel, bg, ba, grav as one flat array (24)."""
        return np.concatenate([np.array(self.rot), np.array(self.pos), np.array(self.vel),
                               np.array(self.bg), np.array(self.ba), np.array(self.grav)])

    def cov_np(self):
        return np.array(self.cov).reshape(18, 18)


class LioIterOut(C.Structure):
    _fields_ = [("HTH", C.c_double * 36), ("HTz", C.c_double * 6), ("solution", C.c_double * 18),
                ("total_residual", C.c_double), ("effct_feat_num", C.c_int32), ("converged", C.c_int32),
                ("status", C.c_int32), ("pad", C.c_int32)]


class LioFrameOut(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("searches", C.c_int32), ("effct_feat_num", C.c_int32),
                ("converged_last", C.c_int32), ("total_residual", C.c_double)]


class VioConfig(C.Structure):
    _fields_ = [("Rcl", C.c_double * 9), ("Pcl", C.c_double * 3), ("R_LI", C.c_double * 9),
                ("t_LI", C.c_double * 3), ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double),
                ("cy", C.c_double), ("d", C.c_double * 5), ("width", C.c_int32), ("height", C.c_int32),
                ("max_iterations", C.c_int32), ("patch_size", C.c_int32), ("img_point_cov", C.c_double)]


class VioLevelOut(C.Structure):
    _fields_ = [("HTH", C.c_double * 36), ("HTz", C.c_double * 6), ("solution", C.c_double * 18),
                ("error", C.c_float), ("iterations", C.c_int32), ("n_meas", C.c_int32),
                ("accepted", C.c_int32), ("fragile", C.c_int32)]


class State23(C.Structure):
    _fields_ = [("pos", C.c_double * 3), ("rot", C.c_double * 4), ("offset_R_L_I", C.c_double * 4),
                ("offset_T_L_I", C.c_double * 3), ("vel", C.c_double * 3), ("bg", C.c_double * 3),
                ("ba", C.c_double * 3), ("grav", C.c_double * 3)]

    def copy(self):
        o = State23()
        C.memmove(C.byref(o), C.byref(self), C.sizeof(State23))
        return o

    def vec(self):
        return np.concatenate([np.array(getattr(self, f)) for f, _ in self._fields_])


class IkfomOut(C.Structure):
    _fields_ = [("HTH", C.c_double * 144), ("HTh", C.c_double * 12), ("dx", C.c_double * 23),
                ("iterations", C.c_int32), ("searches", C.c_int32), ("effct_feat_num", C.c_int32),
                ("status", C.c_int32)]


KNN_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float),
                     C.POINTER(C.c_uint8))

_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        dp, fp, u8p, i32p = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
        L.orc_lio18_iterate.restype = C.c_int
        L.orc_lio18_iterate.argtypes = [C.POINTER(State18), C.POINTER(State18), fp, fp, u8p, C.c_int, dp, dp,
                                        C.c_double, C.c_int, fp, fp, dp, dp, C.POINTER(LioIterOut)]
        L.orc_lio18_frame.restype = C.c_int
        L.orc_lio18_frame.argtypes = [C.POINTER(State18), fp, C.c_int, dp, dp, C.c_double, C.c_int, KNN_FN,
                                      C.c_void_p, C.c_int, u8p, fp, C.POINTER(LioFrameOut)]
        L.orc_vio_update_state.restype = C.c_float
        L.orc_vio_update_state.argtypes = [C.POINTER(VioConfig), C.POINTER(State18), C.POINTER(State18), u8p, fp,
                                           dp, i32p, C.c_int, C.c_float, C.c_int, fp, dp, C.POINTER(VioLevelOut)]
        L.orc_vio_compute_j.restype = C.c_int
        L.orc_vio_compute_j.argtypes = [C.POINTER(VioConfig), C.POINTER(State18), C.POINTER(State18), u8p, fp, dp,
                                        i32p, C.c_int, fp, C.POINTER(VioLevelOut)]
        L.orc_world2cam.restype = None
        L.orc_world2cam.argtypes = [C.POINTER(VioConfig), dp, dp]
        L.orc_h_share_model.restype = C.c_int
        L.orc_h_share_model.argtypes = [C.POINTER(State23), fp, fp, u8p, C.c_int, C.c_int, fp, fp, dp, dp, dp, dp]
        L.orc_ikfom_update_iterated.restype = C.c_int
        L.orc_ikfom_update_iterated.argtypes = [C.POINTER(State23), dp, fp, C.c_int, C.c_double, C.c_int, dp,
                                                KNN_FN, C.c_void_p, C.c_int, u8p, fp, C.POINTER(IkfomOut)]
        L.orc_knn5.restype = C.c_int
        L.orc_knn5.argtypes = [fp, C.c_int, fp, C.c_int, fp, fp, u8p, i32p, C.c_int]
        L.orc_state23_boxplus.restype = None
        L.orc_state23_boxplus.argtypes = [C.POINTER(State23), dp]
        L.orc_state23_boxminus.restype = None
        L.orc_state23_boxminus.argtypes = [C.POINTER(State23), C.POINTER(State23), dp]
        _lib = L
    return _lib


def _p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


def knn_callback(scene_knn):
    """Wrap `scene_knn(world (n,3) float32) -> (nbr (n,5,3) f32, valid (n,) u8)` as an orc_knn_fn."""
    def cb(ctx, world, n, nbr, valid):
        w = np.ctypeslib.as_array(world, shape=(n, 3))
        nb, va = scene_knn(w)
        np.ctypeslib.as_array(nbr, shape=(n, 5, 3))[:] = nb
        np.ctypeslib.as_array(valid, shape=(n,))[:] = va
    return KNN_FN(cb)


class ImuSample(C.Structure):
    _fields_ = [("t", C.c_double), ("gyr", C.c_double * 3), ("acc", C.c_double * 3)]


class Pose6d(C.Structure):
    _fields_ = [("offset_time", C.c_double), ("acc", C.c_double * 3), ("gyr", C.c_double * 3), ("vel", C.c_double * 3),
                ("pos", C.c_double * 3), ("rot", C.c_double * 9)]


class ImuProc(C.Structure):
    """ImuProcess members used by UndistortPcl (IMU_Processing.cpp:611-809)."""
    _fields_ = [("cov_gyr", C.c_double * 3), ("cov_acc", C.c_double * 3), ("cov_bias_gyr", C.c_double * 3),
                ("cov_bias_acc", C.c_double * 3), ("mean_acc", C.c_double * 3), ("Lid_rot_to_IMU", C.c_double * 9),
                ("Lid_offset_to_IMU", C.c_double * 3), ("acc_s_last", C.c_double * 3), ("angvel_last", C.c_double * 3),
                ("last_imu", ImuSample), ("last_lidar_end_time", C.c_double)]


def imu_proc_from_frame(f):
    """f: synth.ImuFrame"""
    p = ImuProc()
    p.cov_gyr[:] = f.cov_gyr; p.cov_acc[:] = f.cov_acc; p.cov_bias_gyr[:] = f.cov_bias_gyr; p.cov_bias_acc[:] = f.cov_bias_acc
    p.mean_acc[:] = f.mean_acc
    p.Lid_rot_to_IMU[:] = np.asarray(f.R_LI, np.float64).reshape(9); p.Lid_offset_to_IMU[:] = f.t_LI
    p.acc_s_last[:] = f.acc_s_last; p.angvel_last[:] = f.angvel_last
    p.last_imu.t = f.last_imu[0]; p.last_imu.gyr[:] = f.last_imu[1:4]; p.last_imu.acc[:] = f.last_imu[4:7]
    p.last_lidar_end_time = f.last_lidar_end_time
    return p


def imu_samples(arr):
    """(k,7) array [t, gyr xyz, acc xyz] -> ctypes array of ImuSample"""
    arr = np.ascontiguousarray(arr, np.float64)
    out = (ImuSample * arr.shape[0])()
    C.memmove(out, arr.ctypes.data, arr.nbytes)
    return out


class PatchCandidate(C.Structure):
    _fields_ = [("pos", C.c_double * 3), ("px_ref", C.c_double * 2), ("f_ref", C.c_double * 3), ("R_ref", C.c_double * 9),
                ("t_ref", C.c_double * 3), ("keyframe_id", C.c_int32), ("level_ref", C.c_int32), ("grid_index", C.c_int32),
                ("reserved", C.c_int32)]


def patch_candidates(sf, kf_ids=None):
    """ctypes array of candidates from a synth.SelectFrame; kf_ids maps keyframe index -> registered id."""
    m = sf.cand_pos.shape[0]
    arr = (PatchCandidate * m)()
    for i in range(m):
        c = arr[i]
        k = int(sf.cand_kf[i])
        c.pos[:] = sf.cand_pos[i]; c.px_ref[:] = sf.cand_px[i]; c.f_ref[:] = sf.cand_f[i]
        c.R_ref[:] = sf.kf_R[k].reshape(9); c.t_ref[:] = sf.kf_t[k]
        c.keyframe_id = int(kf_ids[k]) if kf_ids is not None else k
        c.level_ref = 0; c.grid_index = i
    return arr


def state18_from_frame(fr, R=None, p=None):
    return State18.make(fr.R_prior if R is None else R, fr.p_prior if p is None else p,
                        fr.vel, fr.bg, fr.ba, fr.grav, fr.cov18)


def state23_from_frame(fr, quat_from_R):
    s = State23()
    s.pos[:] = fr.p_prior
    s.rot[:] = quat_from_R(fr.R_prior)
    s.offset_R_L_I[:] = quat_from_R(fr.R_LI)
    s.offset_T_L_I[:] = fr.t_LI
    s.vel[:] = fr.vel
    s.bg[:] = fr.bg
    s.ba[:] = fr.ba
    g = np.asarray(fr.grav, dtype=np.float64)
    s.grav[:] = g / np.linalg.norm(g) * 9.809
    return s


def lio18_iterate(x, x_prop, body, nbr, sel, R_LI, t_LI, cov, nthreads=4, G=None,
                  normvec=None, res_last=None, want_world=False):
    """One Mode-18 iteration.  Mutates x, sel (and G / normvec / res_last when given)."""
    n = body.shape[0]
    out = LioIterOut()
    G = np.zeros((18, 18)) if G is None else G
    normvec = np.zeros((n, 4), dtype=np.float32) if normvec is None else normvec
    res_last = np.zeros(n) if res_last is None else res_last
    world = np.zeros((n, 3), dtype=np.float32) if want_world else None
    R_LI = np.ascontiguousarray(R_LI, dtype=np.float64)
    t_LI = np.ascontiguousarray(t_LI, dtype=np.float64)
    st = lib().orc_lio18_iterate(C.byref(x), C.byref(x_prop), _p(body, C.c_float), _p(nbr, C.c_float),
                                 _p(sel, C.c_uint8), n, _p(R_LI, C.c_double), _p(t_LI, C.c_double), cov,
                                 nthreads, _p(world, C.c_float) if want_world else None,
                                 _p(normvec, C.c_float), _p(res_last, C.c_double), _p(G, C.c_double),
                                 C.byref(out))
    return dict(status=st, out=out, G=G, normvec=normvec, res_last=res_last, world=world)


def lio18_frame(x, body, R_LI, t_LI, cov, max_iter, scene_knn, nthreads=4):
    n = body.shape[0]
    out = LioFrameOut()
    sel = np.zeros(n, dtype=np.uint8)
    normvec = np.zeros((n, 4), dtype=np.float32)
    cb = knn_callback(scene_knn)
    R_LI = np.ascontiguousarray(R_LI, dtype=np.float64)
    t_LI = np.ascontiguousarray(t_LI, dtype=np.float64)
    st = lib().orc_lio18_frame(C.byref(x), _p(body, C.c_float), n, _p(R_LI, C.c_double), _p(t_LI, C.c_double),
                               cov, max_iter, cb, None, nthreads, _p(sel, C.c_uint8), _p(normvec, C.c_float),
                               C.byref(out))
    return dict(status=st, out=out, sel=sel, normvec=normvec)


def vio_config(vf):
    c = VioConfig()
    c.Rcl[:] = np.asarray(vf.Rcl).reshape(9)
    c.Pcl[:] = vf.Pcl
    c.R_LI[:] = np.asarray(vf.R_LI).reshape(9)
    c.t_LI[:] = vf.t_LI
    c.fx, c.fy, c.cx, c.cy = vf.cam["fx"], vf.cam["fy"], vf.cam["cx"], vf.cam["cy"]
    c.d[:] = vf.cam["d"]
    c.width, c.height = vf.cam["width"], vf.cam["height"]
    c.max_iterations = vf.max_iterations
    c.patch_size = vf.patch_size
    c.img_point_cov = vf.img_point_cov
    return c


def vio_update_state(vf, x, x_prop, total_residual, level, G=None):
    cfg = vio_config(vf)
    G = np.zeros((18, 18)) if G is None else G
    errors = np.zeros(vf.m, dtype=np.float32)
    out = VioLevelOut()
    err = lib().orc_vio_update_state(C.byref(cfg), C.byref(x), C.byref(x_prop), _p(vf.img, C.c_uint8),
                                     _p(vf.ref_patch, C.c_float), _p(vf.pos, C.c_double),
                                     _p(vf.search_level, C.c_int32), vf.m, total_residual, level,
                                     _p(errors, C.c_float), _p(G, C.c_double), C.byref(out))
    return dict(error=err, out=out, errors=errors, G=G)


def vio_compute_j(vf, x, x_prop):
    cfg = vio_config(vf)
    errors = np.zeros(vf.m, dtype=np.float32)
    outs = (VioLevelOut * 3)()
    st = lib().orc_vio_compute_j(C.byref(cfg), C.byref(x), C.byref(x_prop), _p(vf.img, C.c_uint8),
                                 _p(vf.ref_patch, C.c_float), _p(vf.pos, C.c_double),
                                 _p(vf.search_level, C.c_int32), vf.m, _p(errors, C.c_float), outs)
    return dict(status=st, outs=outs, errors=errors)


def ikfom_update(x, P, body, R, max_iter, scene_knn, limit=None, nthreads=4):
    n = body.shape[0]
    out = IkfomOut()
    sel = np.zeros(n, dtype=np.uint8)
    normvec = np.zeros((n, 4), dtype=np.float32)
    limit = np.full(23, 0.001) if limit is None else np.ascontiguousarray(limit, dtype=np.float64)
    cb = knn_callback(scene_knn)
    st = lib().orc_ikfom_update_iterated(C.byref(x), _p(P, C.c_double), _p(body, C.c_float), n, R, max_iter,
                                         _p(limit, C.c_double), cb, None, nthreads, _p(sel, C.c_uint8),
                                         _p(normvec, C.c_float), C.byref(out))
    return dict(status=st, out=out, sel=sel, normvec=normvec)


H_DYN_SHARE_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(State23), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                             C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.POINTER(C.c_double)))


def ikfom_update_dyn_share(x, P, R, max_iter, h_dyn_share, limit=None):
    """update_iterated_dyn_share_modified (esekfom.hpp:1619-1928) around a Python measurement callback of the reference's shape:
    h_dyn_share(state: State23, valid: bool, converge: bool) -> (valid, h_x (rows,12) float64, h (rows,) float64)."""
    limit = np.full(23, 0.001) if limit is None else np.ascontiguousarray(limit, dtype=np.float64)
    out = IkfomOut()
    keep = []

    def cb(ctx, xs, valid, converge, rows, hx_out, h_out):
        v, hx, hv = h_dyn_share(xs.contents, bool(valid[0]), bool(converge[0]))
        hx = np.ascontiguousarray(hx, dtype=np.float64).reshape(-1, 12)
        hv = np.ascontiguousarray(hv, dtype=np.float64).reshape(-1)
        keep[:] = [hx, hv]
        valid[0] = 1 if v else 0
        rows[0] = hx.shape[0]
        hx_out[0] = _p(hx, C.c_double) if hx.size else None
        h_out[0] = _p(hv, C.c_double) if hv.size else None
    L = lib()
    L.orc_ikfom_update_dyn_share.restype = C.c_int
    L.orc_ikfom_update_dyn_share.argtypes = [C.POINTER(State23), C.POINTER(C.c_double), C.c_double, C.c_int, C.POINTER(C.c_double),
                                             H_DYN_SHARE_FN, C.c_void_p, C.POINTER(IkfomOut)]
    st = L.orc_ikfom_update_dyn_share(C.byref(x), _p(P, C.c_double), R, max_iter, _p(limit, C.c_double), H_DYN_SHARE_FN(cb), None,
                                      C.byref(out))
    return dict(status=st, out=out)


def knn5_bruteforce(map_xyz, query_xyz, nthreads=8):
    """Exact float-distance 5-NN (oracle/orc_knn.c): nbr (n,5,3), sqdist (n,5), valid (n,), idx (n,5)."""
    map_xyz = np.ascontiguousarray(map_xyz, dtype=np.float32)
    q = np.ascontiguousarray(query_xyz, dtype=np.float32)
    n = q.shape[0]
    nbr = np.zeros((n, 5, 3), dtype=np.float32)
    sq = np.zeros((n, 5), dtype=np.float32)
    valid = np.zeros(n, dtype=np.uint8)
    idx = np.zeros((n, 5), dtype=np.int32)
    lib().orc_knn5(_p(map_xyz, C.c_float), map_xyz.shape[0], _p(q, C.c_float), n, _p(nbr, C.c_float), _p(sq, C.c_float),
                   _p(valid, C.c_uint8), _p(idx, C.c_int32), nthreads)
    return nbr, sq, valid, idx


class MapInfo(C.Structure):
    _fields_ = [("n_before", C.c_int32), ("n_after", C.c_int32), ("n_added", C.c_int32), ("n_removed", C.c_int32),
                ("n_ambiguous", C.c_int32)]


def map_add_points(map_xyz, new_xyz, downsample_size):
    """KD_TREE::Add_Points(new, downsample_on) on a flat array, sequentially (oracle/orc_map.c). Returns (map' (k,3), MapInfo)."""
    map_xyz = np.ascontiguousarray(map_xyz, dtype=np.float32).reshape(-1, 3)
    new_xyz = np.ascontiguousarray(new_xyz, dtype=np.float32).reshape(-1, 3)
    out = np.zeros((max(len(map_xyz) + len(new_xyz), 1), 3), dtype=np.float32)
    info = MapInfo()
    L = lib()
    fp = C.POINTER(C.c_float)
    L.orc_map_add_points.argtypes = [fp, C.c_int, fp, C.c_int, C.c_float, fp, C.POINTER(MapInfo)]
    L.orc_map_add_points.restype = C.c_int
    if L.orc_map_add_points(_p(map_xyz, C.c_float), len(map_xyz), _p(new_xyz, C.c_float), len(new_xyz), float(downsample_size),
                            _p(out, C.c_float), C.byref(info)) != 0:
        raise RuntimeError("orc_map_add_points failed")
    return out[:info.n_after].copy(), info


def map_delete_boxes(map_xyz, boxes):
    """KD_TREE::Delete_Point_Boxes on a flat array (oracle/orc_map.c). boxes (nb,6) = min xyz, max xyz."""
    map_xyz = np.ascontiguousarray(map_xyz, dtype=np.float32).reshape(-1, 3)
    boxes = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 6)
    out = np.zeros((max(len(map_xyz), 1), 3), dtype=np.float32)
    info = MapInfo()
    L = lib()
    fp = C.POINTER(C.c_float)
    L.orc_map_delete_boxes.argtypes = [fp, C.c_int, fp, C.c_int, fp, C.POINTER(MapInfo)]
    L.orc_map_delete_boxes.restype = C.c_int
    L.orc_map_delete_boxes(_p(map_xyz, C.c_float), len(map_xyz), _p(boxes, C.c_float), len(boxes), _p(out, C.c_float), C.byref(info))
    return out[:info.n_after].copy(), info


def fov_segment(win, initialized, pos_lid, cube_len, det_range=300.0, mov_threshold=1.5):
    """lasermap_fov_segment's window logic (oracle/orc_map.c). win: float32[6] in/out. Returns (boxes (nb,6), initialized)."""
    L = lib()
    fp = C.POINTER(C.c_float)
    L.orc_fov_segment.argtypes = [fp, C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_double, C.c_float, C.c_float, fp]
    L.orc_fov_segment.restype = C.c_int
    init = C.c_int(int(initialized))
    pos = np.ascontiguousarray(pos_lid, dtype=np.float64)
    boxes = np.zeros((3, 6), dtype=np.float32)
    nb = L.orc_fov_segment(_p(win, C.c_float), C.byref(init), _p(pos, C.c_double), float(cube_len), float(det_range), float(mov_threshold),
                           _p(boxes, C.c_float))
    return boxes[:nb].copy(), bool(init.value)


def voxel_grid(xyzi, leaf):
    """pcl::VoxelGrid restatement (oracle/orc_voxel.c): returns (centroids (m,4) float32, leaf_too_small)."""
    xyzi = np.ascontiguousarray(xyzi, dtype=np.float32)
    n = xyzi.shape[0]
    leaf = (leaf, leaf, leaf) if np.isscalar(leaf) else tuple(leaf)
    out = np.zeros((max(n, 1), 4), dtype=np.float32)
    m = C.c_int32(0)
    small = C.c_int32(0)
    L = lib()
    L.orc_voxel_grid.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float),
                                 C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.orc_voxel_grid.restype = C.c_int
    rc = L.orc_voxel_grid(_p(xyzi, C.c_float), n, leaf[0], leaf[1], leaf[2], _p(out, C.c_float), C.byref(m), C.byref(small))
    if rc != 0:
        raise RuntimeError("orc_voxel_grid failed")
    return out[:m.value].copy(), bool(small.value)


def imu_undistort(proc, state, imu, pcl_beg_time, pcl_end_time, pts_xyzt):
    """ImuProcess::UndistortPcl restatement (oracle/orc_imu.c). Returns (compensated cloud (n,4), poses list)."""
    samples = imu if not isinstance(imu, np.ndarray) else imu_samples(imu)
    k = len(samples)
    pts = np.array(pts_xyzt, dtype=np.float32, order="C", copy=True)
    n = pts.shape[0]
    poses = (Pose6d * (k + 1))()
    npz = C.c_int32(0)
    L = lib()
    L.orc_imu_undistort.argtypes = [C.POINTER(ImuProc), C.POINTER(State18), C.POINTER(ImuSample), C.c_int, C.c_double, C.c_double,
                                    C.POINTER(C.c_float), C.c_int, C.POINTER(Pose6d), C.POINTER(C.c_int32)]
    L.orc_imu_undistort.restype = C.c_int
    rc = L.orc_imu_undistort(C.byref(proc), C.byref(state), samples, k, pcl_beg_time, pcl_end_time, _p(pts, C.c_float), n, poses, C.byref(npz))
    if rc != 0:
        raise RuntimeError("orc_imu_undistort failed")
    return pts, [poses[i] for i in range(npz.value)]


def vio_depth_image(cfg, Rcw, Pcw, scan_world):
    Rcw = np.ascontiguousarray(Rcw, np.float64).reshape(9); Pcw = np.ascontiguousarray(Pcw, np.float64)
    scan = np.ascontiguousarray(scan_world, np.float32)
    depth = np.zeros((cfg.height, cfg.width), np.float32)
    L = lib()
    L.orc_vio_depth_image.argtypes = [C.POINTER(VioConfig), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_float), C.c_int,
                                      C.POINTER(C.c_float)]
    L.orc_vio_depth_image.restype = None
    L.orc_vio_depth_image(C.byref(cfg), _p(Rcw, C.c_double), _p(Pcw, C.c_double), _p(scan, C.c_float), scan.shape[0], _p(depth, C.c_float))
    return depth


def vio_select(cfg, Rcw, Pcw, cur_img, keyframes, depth, cand, ncc_en=False, ncc_thre=0.0, outlier_threshold=300.0):
    """orc_vio_select: returns dict(idx, errors, levels, reason, patches)."""
    Rcw = np.ascontiguousarray(Rcw, np.float64).reshape(9); Pcw = np.ascontiguousarray(Pcw, np.float64)
    cur = np.ascontiguousarray(cur_img, np.uint8)
    kfs = [np.ascontiguousarray(k, np.uint8) for k in keyframes]
    ptrs = (C.POINTER(C.c_uint8) * len(kfs))(*[k.ctypes.data_as(C.POINTER(C.c_uint8)) for k in kfs])
    m = len(cand)
    idx = np.zeros(max(m, 1), np.int32); err = np.zeros(max(m, 1), np.float32); lvl = np.zeros(max(m, 1), np.int32)
    reason = np.zeros(max(m, 1), np.int32); patches = np.zeros((max(m, 1), 192), np.float32)
    na = C.c_int32(0)
    L = lib()
    L.orc_vio_select.argtypes = [C.POINTER(VioConfig), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint8),
                                 C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_float), C.POINTER(PatchCandidate), C.c_int, C.c_int,
                                 C.c_double, C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                 C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.orc_vio_select.restype = C.c_int
    rc = L.orc_vio_select(C.byref(cfg), _p(Rcw, C.c_double), _p(Pcw, C.c_double), _p(cur, C.c_uint8), ptrs, _p(depth, C.c_float), cand, m,
                          1 if ncc_en else 0, ncc_thre, outlier_threshold, _p(idx, C.c_int32), _p(patches, C.c_float), _p(err, C.c_float),
                          _p(lvl, C.c_int32), C.byref(na), _p(reason, C.c_int32))
    if rc != 0:
        raise RuntimeError("orc_vio_select failed: %d" % rc)
    k = na.value
    return dict(idx=idx[:k].copy(), errors=err[:k].copy(), levels=lvl[:k].copy(), reason=reason[:m].copy(), patches=patches[:k].copy())


class VmapObs(C.Structure):
    _fields_ = [("px", C.c_double * 2), ("f", C.c_double * 3), ("R", C.c_double * 9), ("t", C.c_double * 3), ("score", C.c_float),
                ("level", C.c_int32), ("kf_id", C.c_int32), ("frame_id", C.c_int32)]


class VMap:
    """The visual map of LidarSelector on flat arrays (oracle/orc_vmap.c)."""

    def __init__(self, cfg, grid_size):
        L = lib()
        L.orc_vmap_create.restype = C.c_void_p
        L.orc_vmap_create.argtypes = [C.POINTER(VioConfig), C.c_int]
        L.orc_vmap_destroy.argtypes = [C.c_void_p]
        L.orc_vmap_size.argtypes = [C.c_void_p]
        L.orc_vmap_get_point.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_void_p]
        dp, fp, u8, i32 = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
        L.orc_vmap_add_sparse.argtypes = [C.c_void_p, dp, dp, u8, fp, C.c_int, C.c_int, C.c_int]
        L.orc_vmap_select.argtypes = [C.c_void_p, dp, dp, u8, C.POINTER(u8), fp, C.c_int, C.c_int, C.c_double, C.c_double, i32, fp, i32, fp, i32]
        L.orc_vmap_add_observation.argtypes = [C.c_void_p, dp, dp, u8, i32, i32, C.c_int, C.c_int, C.c_int]
        self.L, self.cfg = L, cfg
        self.cells = (cfg.width // grid_size) * (cfg.height // grid_size)
        self.m = L.orc_vmap_create(C.byref(cfg), grid_size)

    def close(self):
        if self.m:
            self.L.orc_vmap_destroy(self.m)
            self.m = None

    def size(self):
        return self.L.orc_vmap_size(self.m)

    def get_point(self, i):
        pos = np.zeros(3, np.float64); val = C.c_float(0); nobs = C.c_int32(0)
        obs = (VmapObs * 20)()
        assert self.L.orc_vmap_get_point(self.m, i, _p(pos, C.c_double), C.byref(val), C.byref(nobs), obs) == 0
        return pos, val.value, [obs[k] for k in range(nobs.value)]

    def add_sparse(self, Rcw, Pcw, img, scan_world, kf_id, frame_id):
        Rcw = np.ascontiguousarray(Rcw, np.float64).reshape(9); Pcw = np.ascontiguousarray(Pcw, np.float64)
        img = np.ascontiguousarray(img, np.uint8); scan = np.ascontiguousarray(scan_world, np.float32).reshape(-1, 3)
        return self.L.orc_vmap_add_sparse(self.m, _p(Rcw, C.c_double), _p(Pcw, C.c_double), _p(img, C.c_uint8), _p(scan, C.c_float), len(scan),
                                          kf_id, frame_id)

    def select(self, Rcw, Pcw, cur_img, keyframes, scan_down_world, ncc_en=False, ncc_thre=0.0, outlier_threshold=300.0):
        Rcw = np.ascontiguousarray(Rcw, np.float64).reshape(9); Pcw = np.ascontiguousarray(Pcw, np.float64)
        cur = np.ascontiguousarray(cur_img, np.uint8)
        kfs = [np.ascontiguousarray(k, np.uint8) for k in keyframes]
        ptrs = (C.POINTER(C.c_uint8) * max(len(kfs), 1))(*[k.ctypes.data_as(C.POINTER(C.c_uint8)) for k in kfs])
        scan = np.ascontiguousarray(scan_down_world, np.float32).reshape(-1, 3)
        m = self.cells
        sel = np.zeros(m, np.int32); err = np.zeros(m, np.float32); lvl = np.zeros(m, np.int32); patches = np.zeros((m, 192), np.float32)
        ns = C.c_int32(0)
        rc = self.L.orc_vmap_select(self.m, _p(Rcw, C.c_double), _p(Pcw, C.c_double), _p(cur, C.c_uint8), ptrs, _p(scan, C.c_float), len(scan),
                                    1 if ncc_en else 0, ncc_thre, outlier_threshold, _p(sel, C.c_int32), _p(err, C.c_float), _p(lvl, C.c_int32),
                                    _p(patches, C.c_float), C.byref(ns))
        if rc != 0:
            raise RuntimeError("orc_vmap_select failed: %d" % rc)
        k = ns.value
        return dict(points=sel[:k].copy(), errors=err[:k].copy(), levels=lvl[:k].copy(), patches=patches[:k].copy())

    def add_observation(self, Rcw, Pcw, img, sel_points, levels, kf_id, frame_id):
        Rcw = np.ascontiguousarray(Rcw, np.float64).reshape(9); Pcw = np.ascontiguousarray(Pcw, np.float64)
        img = np.ascontiguousarray(img, np.uint8)
        sp = np.ascontiguousarray(sel_points, np.int32); lv = np.ascontiguousarray(levels, np.int32)
        return self.L.orc_vmap_add_observation(self.m, _p(Rcw, C.c_double), _p(Pcw, C.c_double), _p(img, C.c_uint8), _p(sp, C.c_int32),
                                               _p(lv, C.c_int32), len(sp), kf_id, frame_id)


def vio_grid_select(cfg, Rcw, Pcw, pos, value, grid_size):
    Rcw = np.ascontiguousarray(Rcw, np.float64).reshape(9); Pcw = np.ascontiguousarray(Pcw, np.float64)
    pos = np.ascontiguousarray(pos, np.float64); value = np.ascontiguousarray(value, np.float32)
    length = (cfg.width // grid_size) * (cfg.height // grid_size)
    win = np.zeros(length, np.int32); md = np.zeros(length, np.float32); mv = np.zeros(length, np.float32); gn = np.zeros(length, np.int32)
    L = lib()
    L.orc_vio_grid_select.argtypes = [C.POINTER(VioConfig), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                      C.POINTER(C.c_float), C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                      C.POINTER(C.c_int32)]
    L.orc_vio_grid_select.restype = C.c_int
    L.orc_vio_grid_select(C.byref(cfg), _p(Rcw, C.c_double), _p(Pcw, C.c_double), _p(pos, C.c_double), _p(value, C.c_float), pos.shape[0],
                          grid_size, _p(win, C.c_int32), _p(md, C.c_float), _p(mv, C.c_float), _p(gn, C.c_int32))
    return dict(winner=win, map_dist=md, map_value=mv, grid_num=gn)
