"""Misuse of the C ABI must come back as a negative status with a message, never as a fault or a silent no-op
(raw ctypes calls: the Python wrapper would refuse some of these before the library sees them)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _msg(L, h):
    s = L.fl_last_error_string(h)
    return s.decode() if s else ""


def test_bad_arguments_and_wrong_call_order(gpu_lib, scene):
    capi = gpu_lib
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(500, scene=scene)
    h = capi.Handle(capi.config_from_frames(fr))
    L, H = h.L, h.h
    fp = C.POINTER(C.c_float)
    x = capi.state18_from_frame(fr)
    body = np.ascontiguousarray(fr.body_xyz, dtype=np.float32)

    assert L.fl_lio_set_points(H, None, 10) < 0 and "fl_lio_set_points" in _msg(L, H)
    assert L.fl_lio_set_points(H, body.ctypes.data_as(fp), 0) < 0
    # no scan staged yet: passes and searches refuse
    assert L.fl_lio_iterate18(H, 1, 0, None) < 0
    assert L.fl_lio_search18(H, None, None) < 0 and "map" in _msg(L, H)
    assert L.fl_lio_frame18_dev(H, C.byref(x), body.ctypes.data_as(fp), fr.n, None) < 0 and "map" in _msg(L, H)
    # map calls before a map exists
    info = capi.MapInfo()
    assert L.fl_map_add_points(H, body.ctypes.data_as(fp), 10, C.c_float(0.5), C.addressof(info)) < 0 and "fl_map_clear" in _msg(L, H)
    assert L.fl_map_delete_boxes(H, body.ctypes.data_as(fp), 1, None) < 0
    h.map_set_points(scene.map_xyz, 0.5)
    assert L.fl_map_add_points(H, None, 0, C.c_float(0.5), None) < 0          # NULL = staged scan, but none is staged
    assert L.fl_map_delete_boxes(H, body.ctypes.data_as(fp), 65, None) < 0    # > 64 boxes
    assert L.fl_map_delete_boxes(H, None, 1, None) < 0
    n = C.c_int32(0)
    small = np.zeros((10, 3), dtype=np.float32)
    assert L.fl_map_get_points(H, small.ctypes.data_as(fp), 10, C.byref(n)) < 0 and n.value == len(scene.map_xyz)
    assert L.fl_map_get_points(H, None, 0, C.byref(n)) == 0 and n.value == len(scene.map_xyz)
    # VIO without an image / patches
    assert L.fl_vio_iterate(H, 0, 1, 0, None) < 0
    assert L.fl_vio_compute_j(H, C.byref(x), C.byref(x), None) < 0
    # voxel filter: resident cloud requested but none there; non-positive leaf
    m = C.c_int32(0)
    assert L.fl_scan_voxel_filter(H, None, 100, C.c_float(0.2), C.c_float(0.2), C.c_float(0.2), 1, None, C.byref(m), None) < 0
    pts4 = np.zeros((100, 4), dtype=np.float32)
    assert L.fl_scan_voxel_filter(H, pts4.ctypes.data_as(fp), 100, C.c_float(0.0), C.c_float(0.2), C.c_float(0.2), 0, None, C.byref(m), None) < 0
    # null handle: every entry point returns an error (spot check), fl_destroy(NULL) is a no-op
    assert L.fl_lio_set_points(None, body.ctypes.data_as(fp), 10) < 0
    assert L.fl_map_clear(None, C.c_float(0.5)) < 0
    assert L.fl_destroy(None) == 0
    # and the handle still works after all of that
    h.lio_set_points(fr.body_xyz)
    xi = capi.state18_from_frame(fr)
    info = h.lio_frame18_dev(xi, fr.body_xyz)
    assert info.status == 0 and info.effct_feat_num > 0
    h.close()
