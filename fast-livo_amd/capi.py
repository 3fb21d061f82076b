"""ctypes binding of libfastlivo_hip.so (include/fastlivo_hip.h).

This is the host-side mirror used by tests and bench.py.  It raises if the HIP extension is missing
or no GPU is visible -- there is no CPU fallback anywhere in the product path.
"""
from __future__ import annotations

import ctypes as C
import weakref
import os
import subprocess

import numpy as np

from . import LIB_PATH, PKG_DIR

FL_ITER_FORCE = 1
FL_ITER_KEEP_NORMVEC = 2
FL_ITER_STAMP = 4
FL_SUMS18 = 32
FL_SUMS23 = 96


class Config(C.Structure):
    _fields_ = [("device", C.c_int32), ("max_iterations", C.c_int32), ("img_width", C.c_int32),
                ("img_height", C.c_int32), ("patch_size", C.c_int32), ("reserved0", C.c_int32),
                ("R_LI", C.c_double * 9), ("t_LI", C.c_double * 3), ("Rcl", C.c_double * 9),
                ("Pcl", C.c_double * 3), ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double),
                ("cy", C.c_double), ("d", C.c_double * 5), ("laser_point_cov", C.c_double),
                ("img_point_cov", C.c_double)]


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
        return np.concatenate([np.array(self.rot), np.array(self.pos), np.array(self.vel),
                               np.array(self.bg), np.array(self.ba), np.array(self.grav)])

    def cov_np(self):
        return np.array(self.cov).reshape(18, 18)


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


class IterInfo(C.Structure):
    _fields_ = [("solution", C.c_double * 23), ("total_residual", C.c_double), ("effct_feat_num", C.c_int32),
                ("converged", C.c_int32), ("status", C.c_int32), ("iterations", C.c_int32),
                ("need_search", C.c_int32), ("stop", C.c_int32), ("accepted", C.c_int32), ("reserved", C.c_int32)]


class Diagnostics(C.Structure):
    """fl_diagnostics"""
    _fields_ = [("multipass_fallbacks", C.c_int32), ("frames_resumed", C.c_int32), ("multipass_capacity", C.c_int32),
                ("compute_units", C.c_int32), ("demotions", C.c_int32), ("demoted_calls_left", C.c_int32)]


class FrameTiming(C.Structure):
    """fl_frame_timing"""
    _fields_ = [("match_ms", C.c_float), ("solve_ms", C.c_float), ("total_ms", C.c_float), ("searches", C.c_int32)]


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


class VmapObs(C.Structure):
    """fl_vmap_obs"""
    _fields_ = [("px", C.c_double * 2), ("f", C.c_double * 3), ("R", C.c_double * 9), ("t", C.c_double * 3), ("score", C.c_float),
                ("level", C.c_int32), ("kf_id", C.c_int32), ("frame_id", C.c_int32)]


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


KNN_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_float), C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_uint8))

_dp, _fp, _u8p, _i32p = C.POINTER(C.c_double), C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
_H = C.c_void_p

# name -> (restype, argtypes): every symbol include/fastlivo_hip.h declares
SYMBOLS = {
    "fl_create": (C.c_int32, [C.POINTER(Config), C.POINTER(_H)]),
    "fl_destroy": (C.c_int32, [_H]),
    "fl_last_error_string": (C.c_char_p, [_H]),
    "fl_set_stream": (C.c_int32, [_H, C.c_void_p]),
    "fl_sync": (C.c_int32, [_H]),
    "fl_scan_voxel_filter": (C.c_int32, [_H, _fp, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_int32, _fp, C.POINTER(C.c_int32),
                                         C.POINTER(C.c_int32)]),
    "fl_imu_undistort": (C.c_int32, [_H, C.POINTER(ImuProc), C.POINTER(State18), C.POINTER(ImuSample), C.c_int32, C.c_double, C.c_double,
                                     _fp, C.c_int32, _fp, C.POINTER(Pose6d), C.POINTER(C.c_int32)]),
    "fl_lidar_front": (C.c_int32, [_H, C.POINTER(ImuProc), C.POINTER(State18), C.POINTER(ImuSample), C.c_int32, C.c_double, C.c_double, _fp, C.c_int32,
                                   C.c_float, C.c_int32, C.POINTER(IterInfo), C.POINTER(C.c_int32)]),
    "fl_vio_detect": (C.c_int32, [_H, _u8p, C.c_int32, C.c_int32, C.c_int32, _fp, C.c_int32, _fp, C.c_int32, _dp, _dp, C.POINTER(State18), C.c_int32,
                                  C.c_int32, C.c_double, C.c_double, _i32p, _i32p, _i32p]),
    "fl_vio_grid_select": (C.c_int32, [_H, _dp, _dp, _dp, _fp, C.c_int32, C.c_int32, _i32p, _fp, _fp, _i32p, _i32p]),
    "fl_vio_add_keyframe": (C.c_int32, [_H, _u8p, C.c_int32, C.c_int32, C.c_int32, _i32p]),
    "fl_vio_drop_keyframe": (C.c_int32, [_H, C.c_int32]),
    "fl_vio_select_patches": (C.c_int32, [_H, _dp, _dp, _fp, C.c_int32, C.POINTER(PatchCandidate), C.c_int32, C.c_int32, C.c_double, C.c_double,
                                          _i32p, _fp, _i32p, _i32p, _i32p, _fp, _fp]),
    "fl_comm_unique_id": (C.c_int32, [_H, C.c_void_p]),
    "fl_comm_init": (C.c_int32, [_H, C.c_void_p, C.c_int32, C.c_int32]),
    "fl_comm_destroy": (C.c_int32, [_H]),
    "fl_lio_iterate18_sharded": (C.c_int32, [_H, C.c_int32, C.c_int32, C.POINTER(IterInfo)]),
    "fl_vio_iterate_sharded": (C.c_int32, [_H, C.c_int32, C.c_int32, C.c_int32, C.POINTER(IterInfo)]),
    "fl_ikfom_iterate_sharded": (C.c_int32, [_H, C.c_int32, C.c_int32, C.POINTER(IterInfo)]),
    "fl_host_alloc": (C.c_int32, [_H, C.c_size_t, C.POINTER(C.c_void_p)]),
    "fl_host_free": (C.c_int32, [_H, C.c_void_p]),
    "fl_set_timing": (C.c_int32, [_H, C.c_int32]),
    "fl_get_last_kernel_ms": (C.c_int32, [_H, _fp]),
    "fl_get_frame_timing": (C.c_int32, [_H, C.POINTER(FrameTiming)]),
    "fl_set_option": (C.c_int32, [_H, C.c_int32, C.c_int32]),
    "fl_abi_revision": (C.c_int32, []),
    "fl_get_diagnostics": (C.c_int32, [_H, C.POINTER(Diagnostics)]),
    "fl_lio_set_points": (C.c_int32, [_H, _fp, C.c_int32]),
    "fl_lio_set_neighbours": (C.c_int32, [_H, _fp, _u8p, C.c_int32]),
    "fl_lio_get_selection": (C.c_int32, [_H, _u8p, _fp]),
    "fl_lio_get_world_points": (C.c_int32, [_H, _fp]),
    "fl_lio_begin18": (C.c_int32, [_H, C.POINTER(State18), C.POINTER(State18)]),
    "fl_lio_iterate18": (C.c_int32, [_H, C.c_int32, C.c_int32, C.POINTER(IterInfo)]),
    "fl_lio_finish18": (C.c_int32, [_H, C.POINTER(State18)]),
    "fl_lio_get_state18": (C.c_int32, [_H, C.POINTER(State18)]),
    "fl_lio_frame18": (C.c_int32, [_H, C.POINTER(State18), _fp, C.c_int32, KNN_FN, C.c_void_p, C.POINTER(IterInfo)]),
    "fl_lio_accumulate18": (C.c_int32, [_H, C.c_void_p, C.c_int32]),
    "fl_lio_solve18": (C.c_int32, [_H, C.c_void_p, C.c_int32, C.POINTER(IterInfo)]),
    "fl_vio_set_frame": (C.c_int32, [_H, _u8p, C.c_int32, C.c_int32, C.c_int32]),
    "fl_vio_set_patches": (C.c_int32, [_H, _fp, _dp, _i32p, C.c_int32]),
    "fl_vio_begin": (C.c_int32, [_H, C.POINTER(State18), C.POINTER(State18)]),
    "fl_vio_update_state": (C.c_int32, [_H, C.c_float, C.c_int32, _fp, C.POINTER(IterInfo)]),
    "fl_vio_compute_j": (C.c_int32, [_H, C.POINTER(State18), C.POINTER(State18), C.POINTER(IterInfo)]),
    "fl_vio_get_errors": (C.c_int32, [_H, _fp]),
    "fl_vio_iterate": (C.c_int32, [_H, C.c_int32, C.c_int32, C.c_int32, C.POINTER(IterInfo)]),
    "fl_vio_accumulate": (C.c_int32, [_H, C.c_int32, C.c_void_p]),
    "fl_vio_solve": (C.c_int32, [_H, C.c_void_p, C.c_int32, C.POINTER(IterInfo)]),
    "fl_vio_errors_chunk": (C.c_int32, [_H, C.c_void_p, C.c_int32]),
    "fl_vio_solve_exact": (C.c_int32, [_H, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(IterInfo)]),
    "fl_vio_get_state18": (C.c_int32, [_H, C.POINTER(State18)]),
    "fl_ikfom_begin": (C.c_int32, [_H, C.POINTER(State23), _dp, _dp]),
    "fl_h_share_model_sums": (C.c_int32, [_H, C.POINTER(State23), _dp, _dp, _i32p, _dp]),
    "fl_h_share_model_rows": (C.c_int32, [_H, C.POINTER(State23), _dp, _dp, _i32p]),
    "fl_ikfom_world_points": (C.c_int32, [_H, C.POINTER(State23), _fp]),
    "fl_ikfom_iterate": (C.c_int32, [_H, C.c_int32, C.c_int32, C.POINTER(IterInfo)]),
    "fl_ikfom_get": (C.c_int32, [_H, C.POINTER(State23), _dp]),
    "fl_ikfom_update_iterated": (C.c_int32, [_H, C.POINTER(State23), _dp, _fp, C.c_int32, C.c_double, _dp, KNN_FN,
                                             C.c_void_p, C.POINTER(IterInfo)]),
    "fl_ikfom_accumulate": (C.c_int32, [_H, C.c_void_p, C.c_int32]),
    "fl_ikfom_solve": (C.c_int32, [_H, C.c_void_p, C.c_int32, C.POINTER(IterInfo)]),
    "fl_map_set_points": (C.c_int32, [_H, _fp, C.c_int32, C.c_float]),
    "fl_vmap_clear": (C.c_int32, [_H, C.c_int32]),
    "fl_vmap_size": (C.c_int32, [_H, C.POINTER(C.c_int32)]),
    "fl_vmap_get_point": (C.c_int32, [_H, C.c_int32, _dp, _fp, C.POINTER(C.c_int32), C.c_void_p]),
    "fl_vmap_select": (C.c_int32, [_H, _dp, _dp, _fp, C.c_int32, C.c_int32, C.c_double, C.c_double, C.POINTER(C.c_int32), _i32p, _fp, _i32p, _fp]),
    "fl_vmap_add_sparse": (C.c_int32, [_H, _dp, _dp, _fp, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]),
    "fl_vmap_add_observation": (C.c_int32, [_H, _dp, _dp, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]),
    "fl_vmap_release_keyframes": (C.c_int32, [_H, C.POINTER(C.c_int32)]),
    "fl_p2p_export": (C.c_int32, [_H, C.c_int32, C.c_void_p]),
    "fl_p2p_connect": (C.c_int32, [_H, C.c_int32, C.c_int32, C.c_void_p]),
    "fl_p2p_connect_local": (C.c_int32, [_H, C.c_int32, C.c_int32, C.c_void_p]),
    "fl_p2p_disconnect": (C.c_int32, [_H]),
    "fl_map_clear": (C.c_int32, [_H, C.c_float]),
    "fl_map_add_points": (C.c_int32, [_H, _fp, C.c_int32, C.c_float, C.c_void_p]),
    "fl_map_delete_boxes": (C.c_int32, [_H, _fp, C.c_int32, C.c_void_p]),
    "fl_map_get_points": (C.c_int32, [_H, _fp, C.c_int32, C.POINTER(C.c_int32)]),
    "fl_map_compact": (C.c_int32, [_H]),
    "fl_lio_search18": (C.c_int32, [_H, _fp, _u8p]),
    "fl_ikfom_search": (C.c_int32, [_H, _fp, _u8p]),
    "fl_lio_frame18_dev": (C.c_int32, [_H, C.POINTER(State18), _fp, C.c_int32, C.POINTER(IterInfo)]),
    "fl_ikfom_update_iterated_dev": (C.c_int32, [_H, C.POINTER(State23), _dp, _fp, C.c_int32, C.c_double, _dp, C.POINTER(IterInfo)]),
}

# include/fastlivo_hip_debug.h: exported by the instrumented build (libfastlivo_hip_debug.so) only
DEBUG_SYMBOLS = {
    "fl_debug_get_wall": (C.c_int32, [_H, C.POINTER(C.c_longlong)]),
    "fl_debug_get_stamps": (C.c_int32, [_H, C.POINTER(C.c_longlong)]),
    "fl_debug_knn_stamp": (C.c_int32, [_H, C.c_int32]),
    "fl_debug_hog": (C.c_int32, [_H, C.c_int32, C.c_int32, C.c_int32]),
    "fl_debug_chain": (C.c_int32, [_H, C.POINTER(C.c_float), C.c_int32, C.c_float, C.POINTER(C.c_float)]),
    "fl_debug_drop_record": (C.c_int32, [_H, C.c_int32]),
    "fl_debug_map_pool_limit": (C.c_int32, [_H, C.c_int32]),
    "fl_debug_mp_refuse": (C.c_int32, [_H, C.c_int32, C.c_int32]),
}
FL_OPT_MULTIPASS, FL_OPT_MAX_PRODUCERS, FL_OPT_IK_PRODUCERS, FL_OPT_MP_CAPACITY, FL_OPT_VIO_WHOLE_CU, FL_OPT_MAILBOX, FL_OPT_SCAN_PULL, FL_OPT_INCR_SEARCH = 1, 2, 3, 4, 5, 6, 7, 8
FL_OPT_DEMOTE_AFTER, FL_OPT_DEMOTE_CALLS, FL_OPT_VOXEL_SORT, FL_OPT_MAP_INCREMENTAL, FL_OPT_VIO_SPECULATE, FL_OPT_VIO_WIDE = 9, 10, 11, 12, 13, 14
FL_OPT_DETECT_FUSED = 15
DEBUG_LIB_PATH = os.path.join(PKG_DIR, "libfastlivo_hip_debug.so")

_lib = None
_lib_debug = None


def _sources():
    csrc = os.path.join(PKG_DIR, "csrc")
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc)]
    srcs.append(os.path.join(os.path.dirname(PKG_DIR), "include", "fastlivo_hip.h"))
    srcs.append(os.path.join(os.path.dirname(PKG_DIR), "include", "fastlivo_hip_debug.h"))
    srcs.append(os.path.join(PKG_DIR, "build.sh"))
    return srcs


def _fresh(path):
    return os.path.exists(path) and all(os.path.getmtime(path) >= os.path.getmtime(s) for s in _sources())


def build(force=False, debug=True):
    """Compile libfastlivo_hip.so (and, debug=True, the instrumented libfastlivo_hip_debug.so) for gfx950; hipcc cross-compiles
    without a GPU. The two compilations run side by side."""
    procs = []
    if os.environ.get("FL_LIB_PATH"):          # an A/B build selected by the caller (tools/): nothing to compile here
        return LIB_PATH
    if force or not _fresh(LIB_PATH):
        procs.append(subprocess.Popen(["bash", os.path.join(PKG_DIR, "build.sh")]))
    if debug and (force or not _fresh(DEBUG_LIB_PATH)):
        procs.append(subprocess.Popen(["bash", os.path.join(PKG_DIR, "build.sh"), "debug"]))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed (fast-livo_amd/build.sh)")
    return LIB_PATH


def _load(path, symbols):
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950); "
                           "there is no CPU fallback for the ESKF hot path")
    # One process must not mix two HIP/HSA runtimes: PyTorch bundles its own libamdhip64.so.7 and
    # refuses to see the GPU if the system copy (our RUNPATH) was loaded first.  Importing torch
    # first makes both share torch's copy (same soname); torch is only plumbing here.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    for name, (res, args) in symbols.items():
        if os.environ.get("FL_LIB_PATH") and not hasattr(L, name):
            continue               # an A/B build of an older revision (tools/) may lack newer entry points
        fn = getattr(L, name)      # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return L


def lib(debug=False):
    """Load the HIP library (debug=True: the instrumented build); raises (never falls back) when it is missing."""
    global _lib, _lib_debug
    if debug:
        if _lib_debug is None:
            path = DEBUG_LIB_PATH
            if os.environ.get("FL_LIB_PATH"):      # tools/: an A/B variant built with -DFL_INSTRUMENT
                path = LIB_PATH
            _lib_debug = _load(path, {**SYMBOLS, **DEBUG_SYMBOLS})
        return _lib_debug
    if _lib is None:
        _lib = _load(LIB_PATH, SYMBOLS)
    return _lib


class FlError(RuntimeError):
    pass


def _p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


_hot_ptr = {}
_DEFAULT_LIMIT23 = np.full(23, 0.001)      # esekfom's epsi (laserMapping.cpp:1140)


def _p_hot(a, ty, cache=True):
    """_p for the arrays a frame driver gets again and again (the caller's scan buffer): `a.ctypes.data_as` costs 4 us per call --
    of a 0.11 ms frame --, a look-up 0.2 us. Keyed by the array object through a weak reference (the cache keeps no scan alive, and a
    recycled id() cannot alias: the entry's referent must be this very object); entries are dropped when their array dies and by
    Handle.host_free for page-locked buffers. cache=False for temporaries the wrapper made itself (ascontiguousarray copies)."""
    if not cache:
        return a.ctypes.data_as(C.POINTER(ty))
    k = id(a)
    e = _hot_ptr.get(k)
    if e is None or e[0]() is not a or e[2] is not ty:
        ptr = a.ctypes.data_as(C.POINTER(ty))
        try:
            ref = weakref.ref(a, lambda _r, k=k, d=_hot_ptr: d.pop(k, None))      # (d bound now: module globals are gone at interpreter exit)
        except TypeError:                      # not weak-referenceable: do not cache
            return ptr
        if len(_hot_ptr) > 16:
            _hot_ptr.clear()
        e = (ref, ptr, ty, a.ctypes.data)
        _hot_ptr[k] = e
    return e[1]


def _hot_evict(address):
    for k in [k for k, e in _hot_ptr.items() if e[3] == address]:
        _hot_ptr.pop(k, None)


def make_config(R_LI, t_LI, Rcl, Pcl, cam, max_iterations=10, laser_point_cov=0.001, img_point_cov=100.0, device=0):
    c = Config()
    c.device = device
    c.max_iterations = max_iterations
    c.img_width, c.img_height = cam["width"], cam["height"]
    c.patch_size = 8
    c.R_LI[:] = np.asarray(R_LI, dtype=np.float64).reshape(9)
    c.t_LI[:] = np.asarray(t_LI, dtype=np.float64)
    c.Rcl[:] = np.asarray(Rcl, dtype=np.float64).reshape(9)
    c.Pcl[:] = np.asarray(Pcl, dtype=np.float64)
    c.fx, c.fy, c.cx, c.cy = cam["fx"], cam["fy"], cam["cx"], cam["cy"]
    c.d[:] = cam["d"]
    c.laser_point_cov = laser_point_cov
    c.img_point_cov = img_point_cov
    return c


class Handle:
    """RAII wrapper of fl_handle with numpy-friendly methods (names follow the C ABI)."""

    def __init__(self, cfg: Config, debug=False):
        """debug=True: a handle of the instrumented build (include/fastlivo_hip_debug.h)."""
        self.L = lib(debug)
        self.h = _H()
        self.cfg = cfg
        st = self.L.fl_create(C.byref(cfg), C.byref(self.h))
        if st != 0:
            msg = self.L.fl_last_error_string(None)
            raise FlError(f"fl_create failed ({st}): {msg.decode() if msg else ''}")
        self._keep = []
        # measurement scripts (tools/) select A/B behaviour through the environment of the PYTHON process; the library itself reads none
        if not hasattr(self.L, "fl_set_option"):      # an older A/B build: it reads these variables itself
            return
        if os.environ.get("FL_NO_MULTIPASS"):
            self.set_option(FL_OPT_MULTIPASS, 0)
        if os.environ.get("FL_MAX_PRODUCERS"):
            self.set_option(FL_OPT_MAX_PRODUCERS, int(os.environ["FL_MAX_PRODUCERS"]))
        if os.environ.get("FL_IK_PRODUCERS"):
            self.set_option(FL_OPT_IK_PRODUCERS, int(os.environ["FL_IK_PRODUCERS"]))
        if os.environ.get("FL_VIO_OCC1") == "0":
            self.set_option(FL_OPT_VIO_WHOLE_CU, 0)
        if os.environ.get("FL_NO_VIO_SPEC"):
            self.set_option(FL_OPT_VIO_SPECULATE, 0)
        elif os.environ.get("FL_VIO_SPEC"):           # (A/B tools: 1 = speculating accept with one launch per pyramid level, 2 = one launch)
            self.set_option(FL_OPT_VIO_SPECULATE, int(os.environ["FL_VIO_SPEC"]))

    def close(self):
        if self.h:
            self.L.fl_destroy(self.h)
            self.h = _H()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, st, what):
        if st < 0:
            msg = self.L.fl_last_error_string(self.h)
            raise FlError(f"{what} failed ({st}): {msg.decode() if msg else ''}")
        return st

    # ---- common
    def set_stream(self, stream_ptr):
        self._chk(self.L.fl_set_stream(self.h, C.c_void_p(stream_ptr)), "fl_set_stream")

    def sync(self):
        self._chk(self.L.fl_sync(self.h), "fl_sync")

    def scan_voxel_filter(self, xyzi, leaf, stage_as_scan=False, want=True):
        """pcl::VoxelGrid on the device. Returns (centroids (m,4) or None, m, leaf_too_small)."""
        xyzi = np.ascontiguousarray(xyzi, np.float32)
        n = xyzi.shape[0]
        leaf = (leaf, leaf, leaf) if np.isscalar(leaf) else tuple(leaf)
        out = np.empty((n, 4), np.float32) if want else None
        m = C.c_int32(0); small = C.c_int32(0)
        self._chk(self.L.fl_scan_voxel_filter(self.h, xyzi.ctypes.data_as(_fp), n, leaf[0], leaf[1], leaf[2], 1 if stage_as_scan else 0,
                                              out.ctypes.data_as(_fp) if want else None, C.byref(m), C.byref(small)), "fl_scan_voxel_filter")
        return (out[:m.value].copy() if want else None), m.value, bool(small.value)

    def imu_undistort(self, proc, state, imu, pcl_beg_time, pcl_end_time, pts_xyzt, want=True):
        """ImuProcess::UndistortPcl on the device. imu: (k,7) array or ImuSample array. Returns (cloud (n,4) or None, poses list)."""
        samples = imu if not isinstance(imu, np.ndarray) else imu_samples(imu)
        k = len(samples)
        pts = np.ascontiguousarray(pts_xyzt, np.float32)
        n = pts.shape[0]
        out = np.empty((n, 4), np.float32) if want else None
        poses = (Pose6d * (k + 1))()
        npz = C.c_int32(0)
        self._chk(self.L.fl_imu_undistort(self.h, C.byref(proc), C.byref(state), samples, k, pcl_beg_time, pcl_end_time,
                                          pts.ctypes.data_as(_fp) if n else None, n, out.ctypes.data_as(_fp) if (want and n) else None,
                                          poses, C.byref(npz)), "fl_imu_undistort")
        return out, [poses[i] for i in range(npz.value)]

    def lidar_front(self, proc, state, imu, pcl_beg_time, pcl_end_time, pts_xyzt, leaf, staged=False):
        """fl_lidar_front: undistortion -> voxel filter -> Mode-18 update in one enqueue. Mutates proc and state; returns (info, scan points)."""
        samples = imu if not isinstance(imu, np.ndarray) else imu_samples(imu)
        pts = pts_xyzt if (isinstance(pts_xyzt, np.ndarray) and pts_xyzt.dtype == np.float32 and pts_xyzt.flags["C_CONTIGUOUS"]) \
            else np.ascontiguousarray(pts_xyzt, np.float32)
        info = IterInfo(); m = C.c_int32(0)
        self._chk(self.L.fl_lidar_front(self.h, C.byref(proc), C.byref(state), samples, len(samples), pcl_beg_time, pcl_end_time,
                                        pts.ctypes.data_as(_fp), pts.shape[0], leaf, 1 if staged else 0, C.byref(info), C.byref(m)), "fl_lidar_front")
        return info, m.value

    def vio_detect(self, img, pg, pg_down, Rci, Pci, state, frame_id, ncc_en=False, ncc_thre=0.0, outlier_threshold=300.0):
        """fl_vio_detect: LidarSelector::detect in one call. Mutates state; returns (selected, founded, observed)."""
        img = np.ascontiguousarray(img, np.uint8)
        Rci = np.ascontiguousarray(Rci, np.float64).reshape(9); Pci = np.ascontiguousarray(Pci, np.float64)
        a, b, c = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        if pg is None:              # FL_DETECT_SCAN_ON_DEVICE: the handle's staged scan under `state`, down-sampled on the device
            self._chk(self.L.fl_vio_detect(self.h, img.ctypes.data_as(_u8p), img.shape[1], img.shape[0], img.shape[1], None, -1, None, 0,
                                           Rci.ctypes.data_as(_dp), Pci.ctypes.data_as(_dp), C.byref(state), frame_id,
                                           1 if ncc_en else 0, ncc_thre, outlier_threshold, C.byref(a), C.byref(b), C.byref(c)), "fl_vio_detect")
            return a.value, b.value, c.value
        pg = np.ascontiguousarray(pg, np.float32).reshape(-1, 3); pd = np.ascontiguousarray(pg_down, np.float32).reshape(-1, 3)
        self._chk(self.L.fl_vio_detect(self.h, img.ctypes.data_as(_u8p), img.shape[1], img.shape[0], img.shape[1], pg.ctypes.data_as(_fp), len(pg),
                                       pd.ctypes.data_as(_fp), len(pd), Rci.ctypes.data_as(_dp), Pci.ctypes.data_as(_dp), C.byref(state), frame_id,
                                       1 if ncc_en else 0, ncc_thre, outlier_threshold, C.byref(a), C.byref(b), C.byref(c)), "fl_vio_detect")
        return a.value, b.value, c.value

    def scan_voxel_filter_resident(self, n, leaf, stage_as_scan=True):
        """VoxelGrid of the n-point cloud fl_imu_undistort left on the device; result staged as the scan."""
        leaf = (leaf, leaf, leaf) if np.isscalar(leaf) else tuple(leaf)
        m = C.c_int32(0); small = C.c_int32(0)
        self._chk(self.L.fl_scan_voxel_filter(self.h, None, n, leaf[0], leaf[1], leaf[2], 1 if stage_as_scan else 0, None, C.byref(m),
                                              C.byref(small)), "fl_scan_voxel_filter")
        return None, m.value, bool(small.value)

    def vio_grid_select(self, Rcw, Pcw, pos, value, grid_size):
        Rcw = np.ascontiguousarray(Rcw, np.float64).reshape(9); Pcw = np.ascontiguousarray(Pcw, np.float64)
        pos = np.ascontiguousarray(pos, np.float64); value = np.ascontiguousarray(value, np.float32)
        length = (self.cfg.img_width // grid_size) * (self.cfg.img_height // grid_size)
        win = np.zeros(length, np.int32); md = np.zeros(length, np.float32); mv = np.zeros(length, np.float32); gn = np.zeros(length, np.int32)
        ln = C.c_int32(0)
        self._chk(self.L.fl_vio_grid_select(self.h, Rcw.ctypes.data_as(_dp), Pcw.ctypes.data_as(_dp), pos.ctypes.data_as(_dp),
                                            value.ctypes.data_as(_fp), pos.shape[0], grid_size, win.ctypes.data_as(_i32p), md.ctypes.data_as(_fp),
                                            mv.ctypes.data_as(_fp), gn.ctypes.data_as(_i32p), C.byref(ln)), "fl_vio_grid_select")
        return dict(winner=win, map_dist=md, map_value=mv, grid_num=gn)

    def vio_add_keyframe(self, img=None):
        """img None: the image staged by vio_set_frame becomes the keyframe (no second upload)"""
        kid = C.c_int32(-1)
        if img is None:
            self._chk(self.L.fl_vio_add_keyframe(self.h, None, self.cfg.img_width, self.cfg.img_height, self.cfg.img_width, C.byref(kid)),
                      "fl_vio_add_keyframe")
            return kid.value
        img = np.ascontiguousarray(img, np.uint8)
        self._chk(self.L.fl_vio_add_keyframe(self.h, img.ctypes.data_as(_u8p), img.shape[1], img.shape[0], img.shape[1], C.byref(kid)),
                  "fl_vio_add_keyframe")
        return kid.value

    def vio_drop_keyframe(self, kid):
        self._chk(self.L.fl_vio_drop_keyframe(self.h, kid), "fl_vio_drop_keyframe")

    def vio_select_patches(self, Rcw, Pcw, scan_world, cand, ncc_en=False, ncc_thre=0.0, outlier_threshold=300.0, want_patches=True,
                           want_depth=False):
        """Returns dict(idx, errors, levels, reason, patches, depth); the accepted patches stay staged for vio_compute_j."""
        Rcw = np.ascontiguousarray(Rcw, np.float64).reshape(9); Pcw = np.ascontiguousarray(Pcw, np.float64)
        scan = np.ascontiguousarray(scan_world, np.float32)
        m = len(cand)
        idx = np.zeros(max(m, 1), np.int32); err = np.zeros(max(m, 1), np.float32); lvl = np.zeros(max(m, 1), np.int32)
        reason = np.zeros(max(m, 1), np.int32)
        patches = np.zeros((max(m, 1), 192), np.float32) if want_patches else None
        depth = np.zeros((self.cfg.img_height, self.cfg.img_width), np.float32) if want_depth else None
        na = C.c_int32(0)
        self._chk(self.L.fl_vio_select_patches(self.h, Rcw.ctypes.data_as(_dp), Pcw.ctypes.data_as(_dp), scan.ctypes.data_as(_fp) if len(scan) else None,
                                               len(scan), cand, m, 1 if ncc_en else 0, ncc_thre, outlier_threshold, idx.ctypes.data_as(_i32p),
                                               err.ctypes.data_as(_fp), lvl.ctypes.data_as(_i32p), C.byref(na), reason.ctypes.data_as(_i32p),
                                               patches.ctypes.data_as(_fp) if want_patches else None,
                                               depth.ctypes.data_as(_fp) if want_depth else None), "fl_vio_select_patches")
        k = na.value
        return dict(idx=idx[:k].copy(), errors=err[:k].copy(), levels=lvl[:k].copy(), reason=reason[:m].copy(),
                    patches=patches[:k].copy() if want_patches else None, depth=depth)

    # ---- visual map on the device (api_vmap.inc)
    def vmap_clear(self, grid_size):
        self._chk(self.L.fl_vmap_clear(self.h, grid_size), "fl_vmap_clear")
        self._vm_cells = (self.cfg.img_width // grid_size) * (self.cfg.img_height // grid_size)

    def vmap_size(self):
        n = C.c_int32(0)
        self._chk(self.L.fl_vmap_size(self.h, C.byref(n)), "fl_vmap_size")
        return n.value

    def vmap_get_point(self, i):
        pos = np.zeros(3, np.float64); val = C.c_float(0); nobs = C.c_int32(0)
        obs = (VmapObs * 20)()
        self._chk(self.L.fl_vmap_get_point(self.h, i, pos.ctypes.data_as(_dp), C.byref(val), C.byref(nobs), obs), "fl_vmap_get_point")
        return pos, val.value, [obs[k] for k in range(nobs.value)]

    def vmap_select(self, Rcw, Pcw, scan_down_world, ncc_en=False, ncc_thre=0.0, outlier_threshold=300.0, want_patches=True):
        Rcw = np.ascontiguousarray(Rcw, np.float64).reshape(9); Pcw = np.ascontiguousarray(Pcw, np.float64)
        scan = np.ascontiguousarray(scan_down_world, np.float32).reshape(-1, 3)
        m = self._vm_cells
        sel = np.zeros(m, np.int32); err = np.zeros(m, np.float32); lvl = np.zeros(m, np.int32)
        patches = np.zeros((m, 192), np.float32) if want_patches else None
        ns = C.c_int32(0)
        self._chk(self.L.fl_vmap_select(self.h, Rcw.ctypes.data_as(_dp), Pcw.ctypes.data_as(_dp), scan.ctypes.data_as(_fp) if len(scan) else None,
                                        len(scan), 1 if ncc_en else 0, ncc_thre, outlier_threshold, C.byref(ns), sel.ctypes.data_as(_i32p),
                                        err.ctypes.data_as(_fp), lvl.ctypes.data_as(_i32p),
                                        patches.ctypes.data_as(_fp) if want_patches else None), "fl_vmap_select")
        k = ns.value
        return dict(points=sel[:k].copy(), errors=err[:k].copy(), levels=lvl[:k].copy(), patches=patches[:k].copy() if want_patches else None)

    def vmap_add_sparse(self, Rcw, Pcw, scan_world, keyframe_id, frame_id):
        Rcw = np.ascontiguousarray(Rcw, np.float64).reshape(9); Pcw = np.ascontiguousarray(Pcw, np.float64)
        scan = np.ascontiguousarray(scan_world, np.float32).reshape(-1, 3)
        na = C.c_int32(0)
        self._chk(self.L.fl_vmap_add_sparse(self.h, Rcw.ctypes.data_as(_dp), Pcw.ctypes.data_as(_dp), scan.ctypes.data_as(_fp), len(scan),
                                            keyframe_id, frame_id, C.byref(na)), "fl_vmap_add_sparse")
        return na.value

    def vmap_add_observation(self, Rcw, Pcw, keyframe_id, frame_id):
        Rcw = np.ascontiguousarray(Rcw, np.float64).reshape(9); Pcw = np.ascontiguousarray(Pcw, np.float64)
        na = C.c_int32(0)
        self._chk(self.L.fl_vmap_add_observation(self.h, Rcw.ctypes.data_as(_dp), Pcw.ctypes.data_as(_dp), keyframe_id, frame_id, C.byref(na)),
                  "fl_vmap_add_observation")
        return na.value

    def vmap_release_keyframes(self):
        n = C.c_int32(0)
        self._chk(self.L.fl_vmap_release_keyframes(self.h, C.byref(n)), "fl_vmap_release_keyframes")
        return n.value

    def comm_unique_id(self):
        buf = (C.c_char * 128)()
        self._chk(self.L.fl_comm_unique_id(self.h, buf), "fl_comm_unique_id")
        return bytes(buf)

    def comm_init(self, uid, rank, world):
        buf = (C.c_char * 128).from_buffer_copy(uid)
        self._chk(self.L.fl_comm_init(self.h, buf, rank, world), "fl_comm_init")

    def comm_destroy(self):
        self._chk(self.L.fl_comm_destroy(self.h), "fl_comm_destroy")

    # ---- peer exchange inside the pass kernels (api_p2p.inc)
    def p2p_export(self, world):
        buf = (C.c_char * 64)()
        self._chk(self.L.fl_p2p_export(self.h, world, buf), "fl_p2p_export")
        return bytes(buf)

    def p2p_connect(self, rank, world, handles):
        """handles: list of `world` 64-byte strings (fl_p2p_export of every rank)"""
        blob = (C.c_char * (64 * world)).from_buffer_copy(b"".join(handles))
        self._chk(self.L.fl_p2p_connect(self.h, rank, world, blob), "fl_p2p_connect")

    def p2p_disconnect(self):
        self._chk(self.L.fl_p2p_disconnect(self.h), "fl_p2p_disconnect")

    def lio_iterate18_sharded(self, count=1, flags=0, want_info=True):
        info = IterInfo()
        self._chk(self.L.fl_lio_iterate18_sharded(self.h, count, flags, C.byref(info) if want_info else None), "fl_lio_iterate18_sharded")
        return info if want_info else None

    def vio_iterate_sharded(self, level, count=1, flags=0, want_info=True):
        info = IterInfo()
        self._chk(self.L.fl_vio_iterate_sharded(self.h, level, count, flags, C.byref(info) if want_info else None), "fl_vio_iterate_sharded")
        return info if want_info else None

    def ikfom_iterate_sharded(self, count=1, flags=0, want_info=True):
        info = IterInfo()
        self._chk(self.L.fl_ikfom_iterate_sharded(self.h, count, flags, C.byref(info) if want_info else None), "fl_ikfom_iterate_sharded")
        return info if want_info else None

    def host_alloc(self, shape, dtype=np.float32):
        """numpy view of page-locked host memory owned by the library (free with host_free)."""
        dt = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dt.itemsize
        p = C.c_void_p()
        self._chk(self.L.fl_host_alloc(self.h, nbytes, C.byref(p)), "fl_host_alloc")
        buf = (C.c_char * nbytes).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dt).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p.value
        return arr

    def host_free(self, arr):
        _hot_evict(arr.ctypes.data)            # a cached pointer into this buffer would dangle
        p = self._pinned.pop(arr.ctypes.data)
        self._chk(self.L.fl_host_free(self.h, C.c_void_p(p)), "fl_host_free")

    def set_timing(self, on):
        self._chk(self.L.fl_set_timing(self.h, 1 if on else 0), "fl_set_timing")

    def frame_timing(self):
        t = FrameTiming()
        self._chk(self.L.fl_get_frame_timing(self.h, C.byref(t)), "fl_get_frame_timing")
        return dict(match_ms=t.match_ms, solve_ms=t.solve_ms, total_ms=t.total_ms, searches=t.searches)

    def last_kernel_ms(self):
        ms = C.c_float()
        self._chk(self.L.fl_get_last_kernel_ms(self.h, C.byref(ms)), "fl_get_last_kernel_ms")
        return ms.value

    def debug_stamps(self):
        a = (C.c_longlong * 64)()
        self._chk(self.L.fl_debug_get_stamps(self.h, a), "fl_debug_get_stamps")
        return np.array(a[:], dtype=np.int64)

    def debug_wall(self):
        a = (C.c_longlong * 2048)()
        self._chk(self.L.fl_debug_get_wall(self.h, a), "fl_debug_get_wall")
        return np.array(a[:], dtype=np.int64)

    # ---- LIO
    def lio_set_points(self, body):
        body = np.ascontiguousarray(body, dtype=np.float32)
        self._chk(self.L.fl_lio_set_points(self.h, _p(body, C.c_float), body.shape[0]), "fl_lio_set_points")

    def lio_set_neighbours(self, nbr, valid):
        nbr = np.ascontiguousarray(nbr, dtype=np.float32)
        valid = np.ascontiguousarray(valid, dtype=np.uint8)
        self._chk(self.L.fl_lio_set_neighbours(self.h, _p(nbr, C.c_float), _p(valid, C.c_uint8), valid.shape[0]),
                  "fl_lio_set_neighbours")

    def lio_get_selection(self, n):
        mask = np.zeros(n, dtype=np.uint8)
        nv = np.zeros((n, 4), dtype=np.float32)
        self._chk(self.L.fl_lio_get_selection(self.h, _p(mask, C.c_uint8), _p(nv, C.c_float)), "fl_lio_get_selection")
        return mask, nv

    def lio_get_world_points(self, n):
        w = np.zeros((n, 3), dtype=np.float32)
        self._chk(self.L.fl_lio_get_world_points(self.h, _p(w, C.c_float)), "fl_lio_get_world_points")
        return w

    def lio_begin18(self, state, prop):
        self._chk(self.L.fl_lio_begin18(self.h, C.byref(state), C.byref(prop)), "fl_lio_begin18")

    def lio_iterate18(self, count=1, flags=0, want_info=True):
        info = IterInfo()
        st = self.L.fl_lio_iterate18(self.h, count, flags, C.byref(info) if want_info else None)
        self._chk(st, "fl_lio_iterate18")
        return info

    def lio_finish18(self):
        out = State18()
        self._chk(self.L.fl_lio_finish18(self.h, C.byref(out)), "fl_lio_finish18")
        return out

    def lio_get_state18(self):
        out = State18()
        self._chk(self.L.fl_lio_get_state18(self.h, C.byref(out)), "fl_lio_get_state18")
        return out

    def lio_frame18(self, state, body, scene_knn):
        body = np.ascontiguousarray(body, dtype=np.float32)
        n = body.shape[0]

        def cb(ctx, world, nn, nbr, valid):
            w = np.ctypeslib.as_array(world, shape=(nn, 3))
            nb, va = scene_knn(w)
            np.ctypeslib.as_array(nbr, shape=(nn, 5, 3))[:] = nb
            np.ctypeslib.as_array(valid, shape=(nn,))[:] = va
        cbf = KNN_FN(cb)
        info = IterInfo()
        self._chk(self.L.fl_lio_frame18(self.h, C.byref(state), _p(body, C.c_float), n, cbf, None, C.byref(info)),
                  "fl_lio_frame18")
        return info

    def lio_accumulate18(self, d_sums_ptr, flags=0):
        self._chk(self.L.fl_lio_accumulate18(self.h, C.c_void_p(d_sums_ptr), flags), "fl_lio_accumulate18")

    def lio_solve18(self, d_sums_ptr, flags=0, want_info=False):
        info = IterInfo()
        self._chk(self.L.fl_lio_solve18(self.h, C.c_void_p(d_sums_ptr), flags, C.byref(info) if want_info else None),
                  "fl_lio_solve18")
        return info

    # ---- VIO
    def vio_set_frame(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        self._chk(self.L.fl_vio_set_frame(self.h, _p(img, C.c_uint8), img.shape[1], img.shape[0], img.shape[1]),
                  "fl_vio_set_frame")

    def vio_set_patches(self, ref, pos, slevel):
        ref = np.ascontiguousarray(ref, dtype=np.float32)
        pos = np.ascontiguousarray(pos, dtype=np.float64)
        slevel = np.ascontiguousarray(slevel, dtype=np.int32)
        self._chk(self.L.fl_vio_set_patches(self.h, _p(ref, C.c_float), _p(pos, C.c_double), _p(slevel, C.c_int32),
                                            slevel.shape[0]), "fl_vio_set_patches")

    def vio_begin(self, state, prop):
        self._chk(self.L.fl_vio_begin(self.h, C.byref(state), C.byref(prop)), "fl_vio_begin")

    def vio_update_state(self, total_residual, level):
        err = C.c_float()
        info = IterInfo()
        self._chk(self.L.fl_vio_update_state(self.h, total_residual, level, C.byref(err), C.byref(info)),
                  "fl_vio_update_state")
        return err.value, info

    def vio_compute_j(self, state, prop):
        infos = (IterInfo * 3)()
        self._chk(self.L.fl_vio_compute_j(self.h, C.byref(state), C.byref(prop), infos), "fl_vio_compute_j")
        return infos

    def vio_get_errors(self, m):
        e = np.zeros(m, dtype=np.float32)
        self._chk(self.L.fl_vio_get_errors(self.h, _p(e, C.c_float)), "fl_vio_get_errors")
        return e

    def vio_iterate(self, level, count=1, flags=0, want_info=True):
        info = IterInfo()
        self._chk(self.L.fl_vio_iterate(self.h, level, count, flags, C.byref(info) if want_info else None),
                  "fl_vio_iterate")
        return info

    def vio_accumulate(self, level, d_sums_ptr):
        self._chk(self.L.fl_vio_accumulate(self.h, level, C.c_void_p(d_sums_ptr)), "fl_vio_accumulate")

    def vio_solve(self, d_sums_ptr, flags=0, want_info=False):
        info = IterInfo()
        self._chk(self.L.fl_vio_solve(self.h, C.c_void_p(d_sums_ptr), flags, C.byref(info) if want_info else None),
                  "fl_vio_solve")
        return info

    def vio_errors_chunk(self, d_chunk_ptr, stride):
        """this rank's per-patch floats as one chunk of the all-gather behind the exact accept test (device pointer, stride floats)"""
        self._chk(self.L.fl_vio_errors_chunk(self.h, C.c_void_p(d_chunk_ptr), int(stride)), "fl_vio_errors_chunk")

    def vio_solve_exact(self, d_sums_ptr, flags, d_all_chunks_ptr, stride, world, want_info=False):
        info = IterInfo()
        self._chk(self.L.fl_vio_solve_exact(self.h, C.c_void_p(d_sums_ptr), flags, C.c_void_p(d_all_chunks_ptr), int(stride), int(world),
                                            C.byref(info) if want_info else None), "fl_vio_solve_exact")
        return info

    def vio_get_state18(self):
        out = State18()
        self._chk(self.L.fl_vio_get_state18(self.h, C.byref(out)), "fl_vio_get_state18")
        return out


def config_from_frames(lio, vio=None, max_iterations=10, device=0):
    """fl_config for a synthetic frame pair (fast_livo_amd.synth)."""
    from . import synth
    cam = vio.cam if vio is not None else dict(synth.PINHOLE, d=(0.0,) * 5)
    Rcl = vio.Rcl if vio is not None else synth.AVIA_RCL
    Pcl = vio.Pcl if vio is not None else synth.AVIA_PCL
    ipc = vio.img_point_cov if vio is not None else synth.IMG_POINT_COV
    return make_config(lio.R_LI, lio.t_LI, Rcl, Pcl, cam, max_iterations=max_iterations,
                       laser_point_cov=lio.laser_point_cov, img_point_cov=ipc, device=device)


def state18_from_frame(fr, R=None, p=None):
    return State18.make(fr.R_prior if R is None else R, fr.p_prior if p is None else p,
                        fr.vel, fr.bg, fr.ba, fr.grav, fr.cov18)


# ---------------------------------------------------------------------------------------- Mode-23
def _ikfom_methods():
    def ikfom_begin(self, x23, P, limit=None):
        P = np.ascontiguousarray(P, dtype=np.float64)
        limit = np.full(23, 0.001) if limit is None else np.ascontiguousarray(limit, dtype=np.float64)
        self._chk(self.L.fl_ikfom_begin(self.h, C.byref(x23), _p(P, C.c_double), _p(limit, C.c_double)), "fl_ikfom_begin")

    def ikfom_iterate(self, count=1, flags=0, want_info=True):
        info = IterInfo()
        self._chk(self.L.fl_ikfom_iterate(self.h, count, flags, C.byref(info) if want_info else None), "fl_ikfom_iterate")
        return info

    def ikfom_get(self):
        x = State23()
        P = np.zeros((23, 23))
        self._chk(self.L.fl_ikfom_get(self.h, C.byref(x), _p(P, C.c_double)), "fl_ikfom_get")
        return x, P

    def h_share_model_sums(self, s23):
        HTH = np.zeros((12, 12))
        HTh = np.zeros(12)
        neff = C.c_int32()
        res = C.c_double()
        self._chk(self.L.fl_h_share_model_sums(self.h, C.byref(s23), _p(HTH, C.c_double), _p(HTh, C.c_double), C.byref(neff),
                                               C.byref(res)), "fl_h_share_model_sums")
        return HTH, HTh, neff.value, res.value

    def h_share_model_rows(self, s23, n):
        h_x = np.zeros((n, 12))
        hv = np.zeros(n)
        neff = C.c_int32()
        self._chk(self.L.fl_h_share_model_rows(self.h, C.byref(s23), _p(h_x, C.c_double), _p(hv, C.c_double), C.byref(neff)),
                  "fl_h_share_model_rows")
        return h_x[:neff.value], hv[:neff.value]

    def ikfom_update_iterated(self, x23, P, body, R, scene_knn, limit=None):
        body = np.ascontiguousarray(body, dtype=np.float32)
        n = body.shape[0]
        limit = np.full(23, 0.001) if limit is None else np.ascontiguousarray(limit, dtype=np.float64)

        def cb(ctx, world, nn, nbr, valid):
            w = np.ctypeslib.as_array(world, shape=(nn, 3))
            nb, va = scene_knn(w)
            np.ctypeslib.as_array(nbr, shape=(nn, 5, 3))[:] = nb
            np.ctypeslib.as_array(valid, shape=(nn,))[:] = va
        cbf = KNN_FN(cb)
        info = IterInfo()
        self._chk(self.L.fl_ikfom_update_iterated(self.h, C.byref(x23), _p(P, C.c_double), _p(body, C.c_float), n, R,
                                                  _p(limit, C.c_double), cbf, None, C.byref(info)), "fl_ikfom_update_iterated")
        return info

    def ikfom_world_points(self, s23, n):
        w = np.zeros((n, 3), dtype=np.float32)
        self._chk(self.L.fl_ikfom_world_points(self.h, C.byref(s23), _p(w, C.c_float)), "fl_ikfom_world_points")
        return w

    def ikfom_accumulate(self, d_sums_ptr, flags=0):
        self._chk(self.L.fl_ikfom_accumulate(self.h, C.c_void_p(d_sums_ptr), flags), "fl_ikfom_accumulate")

    def ikfom_solve(self, d_sums_ptr, flags=0, want_info=False):
        info = IterInfo()
        self._chk(self.L.fl_ikfom_solve(self.h, C.c_void_p(d_sums_ptr), flags, C.byref(info) if want_info else None),
                  "fl_ikfom_solve")
        return info

    for f in (ikfom_world_points, ikfom_begin, ikfom_iterate, ikfom_get, h_share_model_sums, h_share_model_rows, ikfom_update_iterated,
              ikfom_accumulate, ikfom_solve):
        setattr(Handle, f.__name__, f)


_ikfom_methods()


def state23_from_frame(fr):
    from . import synth
    s = State23()
    s.pos[:] = fr.p_prior
    s.rot[:] = synth.quat_from_R(fr.R_prior)
    s.offset_R_L_I[:] = synth.quat_from_R(fr.R_LI)
    s.offset_T_L_I[:] = fr.t_LI
    s.vel[:] = fr.vel
    s.bg[:] = fr.bg
    s.ba[:] = fr.ba
    g = np.asarray(fr.grav, dtype=np.float64)
    s.grav[:] = g / np.linalg.norm(g) * 9.809
    return s


def p2p_connect_local(handles):
    """Connect Handle objects living in this process (rank = position in the list)."""
    world = len(handles)
    arr = (C.c_void_p * world)(*[C.cast(hh.h, C.c_void_p).value for hh in handles])
    for r, hh in enumerate(handles):
        hh._chk(hh.L.fl_p2p_connect_local(hh.h, r, world, arr), "fl_p2p_connect_local")


# ------------------------------------------------------------------------------------ device k-NN
class MapInfo(C.Structure):
    """fl_map_info"""
    _fields_ = [("n_before", C.c_int32), ("n_after", C.c_int32), ("n_added", C.c_int32), ("n_removed", C.c_int32),
                ("n_ambiguous", C.c_int32), ("status", C.c_int32), ("cell_size", C.c_float)]


def _knn_methods():
    def map_set_points(self, map_xyz, cell_size=0.5):
        m = np.ascontiguousarray(map_xyz, dtype=np.float32)
        self._chk(self.L.fl_map_set_points(self.h, _p(m, C.c_float), m.shape[0], cell_size), "fl_map_set_points")

    def map_clear(self, cell_size=0.5):
        self._chk(self.L.fl_map_clear(self.h, cell_size), "fl_map_clear")

    def map_add_points(self, world_xyz, downsample_size, want_info=True):
        """map_incremental on the device map; world_xyz None = the staged scan under the device's 18-state. Returns MapInfo
        (want_info=False: no counters asked for -- the in-place form then returns without waiting for the device)."""
        info = MapInfo()
        ip = C.addressof(info) if want_info else None
        if world_xyz is None:
            self._chk(self.L.fl_map_add_points(self.h, None, 0, downsample_size, ip), "fl_map_add_points")
        else:
            w = np.ascontiguousarray(world_xyz, dtype=np.float32).reshape(-1, 3)
            self._chk(self.L.fl_map_add_points(self.h, _p(w, C.c_float), w.shape[0], downsample_size, ip), "fl_map_add_points")
        return info if want_info else None

    def map_delete_boxes(self, boxes, want_info=True):
        b = np.ascontiguousarray(boxes, dtype=np.float32).reshape(-1, 6)
        info = MapInfo()
        self._chk(self.L.fl_map_delete_boxes(self.h, _p(b, C.c_float), b.shape[0], C.addressof(info) if want_info else None), "fl_map_delete_boxes")
        return info if want_info else None

    def map_get_points(self):
        n = C.c_int32(0)
        self._chk(self.L.fl_map_get_points(self.h, None, 0, C.byref(n)), "fl_map_get_points")
        out = np.zeros((max(n.value, 1), 3), dtype=np.float32)
        if n.value:
            self._chk(self.L.fl_map_get_points(self.h, _p(out, C.c_float), n.value, C.byref(n)), "fl_map_get_points")
        return out[:n.value]

    def lio_search18(self, n, want=True):
        nbr = np.zeros((n, 5, 3), dtype=np.float32)
        valid = np.zeros(n, dtype=np.uint8)
        self._chk(self.L.fl_lio_search18(self.h, _p(nbr, C.c_float) if want else None, _p(valid, C.c_uint8) if want else None),
                  "fl_lio_search18")
        return nbr, valid

    def ikfom_search(self, n, want=True):
        nbr = np.zeros((n, 5, 3), dtype=np.float32)
        valid = np.zeros(n, dtype=np.uint8)
        self._chk(self.L.fl_ikfom_search(self.h, _p(nbr, C.c_float) if want else None, _p(valid, C.c_uint8) if want else None),
                  "fl_ikfom_search")
        return nbr, valid

    def debug_hog(self, blocks, lds_bytes, usec):
        self._chk(self.L.fl_debug_hog(self.h, int(blocks), int(lds_bytes), int(usec)), "fl_debug_hog")

    def debug_chain(self, e, init=0.0, full=False):
        """(workgroup form, one-lane) float running sums of init + e[0] + e[1] + ... on the device (csrc/exact_chain.h);
        full: (workgroup form, one-lane, wavefront form, chunks in which the workgroup form fell back)"""
        e = np.ascontiguousarray(e, dtype=np.float32)
        out = np.zeros(16, dtype=np.float32)
        self._chk(self.L.fl_debug_chain(self.h, _p(e, C.c_float), len(e), C.c_float(init), _p(out, C.c_float)), "fl_debug_chain")
        self.chain_profile = out[4:].copy()     # shader-clock offsets of the last chunk's phases (tools/chain_profile.py)
        return (out[0], out[1], out[2], int(out[3])) if full else (out[0], out[1])

    def debug_drop_record(self, passes_ahead=0):
        self._chk(self.L.fl_debug_drop_record(self.h, int(passes_ahead)), "fl_debug_drop_record")

    def map_compact(self):
        self._chk(self.L.fl_map_compact(self.h), "fl_map_compact")

    def debug_mp_refuse(self, nth, count=1):
        self._chk(self.L.fl_debug_mp_refuse(self.h, int(nth), int(count)), "fl_debug_mp_refuse")

    def debug_map_pool_limit(self, spare_entries):
        self._chk(self.L.fl_debug_map_pool_limit(self.h, int(spare_entries)), "fl_debug_map_pool_limit")

    def debug_knn_stamp(self, on=True):
        self._chk(self.L.fl_debug_knn_stamp(self.h, 1 if on else 0), "fl_debug_knn_stamp")

    def set_option(self, option, value):
        self._chk(self.L.fl_set_option(self.h, int(option), int(value)), "fl_set_option")

    def diagnostics(self):
        """dict(fallbacks, resumes, capacity, cus): multi-pass launches refused by the admission check, frames resumed after an
        abandoned pass, workgroups of a multi-pass kernel the device holds at once, compute units (fl_get_diagnostics)."""
        d = Diagnostics()
        self._chk(self.L.fl_get_diagnostics(self.h, C.byref(d)), "fl_get_diagnostics")
        return dict(fallbacks=d.multipass_fallbacks, resumes=d.frames_resumed, capacity=d.multipass_capacity, cus=d.compute_units,
                    demotions=d.demotions, demoted_calls_left=d.demoted_calls_left)

    def lio_frame18_dev(self, state, body):
        """body None: use the scan already staged on the device (lio_set_points / scan_voxel_filter)."""
        info = IterInfo()
        if body is None:
            self._chk(self.L.fl_lio_frame18_dev(self.h, C.byref(state), None, 0, C.byref(info)), "fl_lio_frame18_dev")
            return info
        own = isinstance(body, np.ndarray) and body.dtype == np.float32 and body.flags["C_CONTIGUOUS"]
        if not own:
            body = np.ascontiguousarray(body, dtype=np.float32)
        self._chk(self.L.fl_lio_frame18_dev(self.h, C.byref(state), _p_hot(body, C.c_float, own), body.shape[0], C.byref(info)),
                  "fl_lio_frame18_dev")
        return info

    def ikfom_update_iterated_dev(self, x23, P, body, R, limit=None):
        own = isinstance(body, np.ndarray) and body.dtype == np.float32 and body.flags["C_CONTIGUOUS"]
        if not own:
            body = np.ascontiguousarray(body, dtype=np.float32)
        own_l = limit is None
        limit = _DEFAULT_LIMIT23 if limit is None else np.ascontiguousarray(limit, dtype=np.float64)
        info = IterInfo()
        self._chk(self.L.fl_ikfom_update_iterated_dev(self.h, C.byref(x23), _p(P, C.c_double), _p_hot(body, C.c_float, own), body.shape[0], R,
                                                      _p_hot(limit, C.c_double, own_l), C.byref(info)), "fl_ikfom_update_iterated_dev")
        return info

    for f in (map_set_points, map_clear, map_add_points, map_delete_boxes, map_get_points, map_compact, lio_search18, ikfom_search, lio_frame18_dev, ikfom_update_iterated_dev,
              debug_hog, debug_chain, debug_drop_record, debug_knn_stamp, debug_map_pool_limit, debug_mp_refuse, set_option, diagnostics):
        setattr(Handle, f.__name__, f)


_knn_methods()
