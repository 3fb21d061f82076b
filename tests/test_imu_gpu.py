"""ImuProcess::UndistortPcl on the device vs the CPU restatement (oracle/orc_imu.c), through the C ABI.

fp64 propagation: the device sums the 18-term covariance products in the oracle's order without contraction; the only
systematic difference is sin/cos (device libm vs glibc, <= 2 ulp), so state/covariance/poses agree to 1e-12 relative and the
float cloud to 1 ulp of float (tolerance 2e-6 m absolute at 30 m range, stated here)."""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
TOL64 = 1e-12
TOL_PT = 4e-6


def _run(h, f, capi):
    xo = orc.state18_from_frame(f.lio); po = orc.imu_proc_from_frame(f)
    ref_pts, ref_poses = orc.imu_undistort(po, xo, f.imu, f.pcl_beg_time, f.pcl_end_time, f.pts_xyzt)
    xg = capi.state18_from_frame(f.lio); pg = capi.imu_proc_from_frame(f)
    out, poses = h.imu_undistort(pg, xg, f.imu, f.pcl_beg_time, f.pcl_end_time, f.pts_xyzt)
    assert len(poses) == len(ref_poses)
    for a, b in zip(poses, ref_poses):
        va = np.frombuffer(bytes(a), np.float64); vb = np.frombuffer(bytes(b), np.float64)
        assert np.allclose(va, vb, rtol=TOL64, atol=TOL64)
    sa = np.frombuffer(bytes(xg), np.float64); sb = np.frombuffer(bytes(xo), np.float64)
    assert np.allclose(sa, sb, rtol=1e-11, atol=1e-14), np.abs(sa - sb).max()
    qa = np.frombuffer(bytes(pg), np.float64); qb = np.frombuffer(bytes(po), np.float64)
    assert np.allclose(qa, qb, rtol=TOL64, atol=TOL64)
    assert np.array_equal(out[:, 3], f.pts_xyzt[:, 3])
    d = np.abs(out[:, :3].astype(np.float64) - ref_pts[:, :3].astype(np.float64))
    assert d.max() <= TOL_PT, d.max()
    return out, ref_pts, poses


@pytest.mark.parametrize("n,n_imu", [(1, 5), (300, 1), (24000, 20), (200000, 100)])
def test_sorted_cloud(gpu_lib, n, n_imu):
    from fast_livo_amd import capi, synth
    f = synth.make_imu_frame(n, n_imu=n_imu)
    h = capi.Handle(capi.config_from_frames(f.lio))
    out, ref, poses = _run(h, f, capi)
    if n >= 300:
        moved = np.abs(out[:, :3] - f.pts_xyzt[:, :3]).max(1) > 0
        assert moved.mean() > 0.9
        assert not moved[0]            # offset 0: not later than IMUpose[0], left alone by the reference's loop


def test_unsorted_cloud_follows_the_sequential_loop(gpu_lib):
    from fast_livo_amd import capi, synth
    f = synth.make_imu_frame(50000, n_imu=20, time_sorted=False, seed=5)
    h = capi.Handle(capi.config_from_frames(f.lio))
    out, ref, poses = _run(h, f, capi)
    # the loop stops for good at the last point (from the back) with offset 0 once it is at head 0: a whole prefix is untouched
    same = (out[:, :3] == f.pts_xyzt[:, :3]).all(1)
    assert same[:2].all()


def test_first_point_recompensated_by_earlier_intervals(gpu_lib):
    from fast_livo_amd import capi, synth
    f = synth.make_imu_frame(5000, n_imu=20, first_point_late=True, seed=9)
    h = capi.Handle(capi.config_from_frames(f.lio))
    out, ref, poses = _run(h, f, capi)
    # point 0 went through several intervals: it differs from a single compensation far more than its neighbours do
    d0 = np.linalg.norm(out[0, :3] - f.pts_xyzt[0, :3]); d1 = np.linalg.norm(out[1, :3] - f.pts_xyzt[1, :3])
    assert d0 > 0 and d1 > 0


def test_imu_not_straddling_and_state_chain(gpu_lib):
    from fast_livo_amd import capi, synth
    f = synth.make_imu_frame(8000, n_imu=12, imu_before_frame=False, seed=11)
    h = capi.Handle(capi.config_from_frames(f.lio))
    _run(h, f, capi)


def test_undistorted_cloud_stays_on_device_for_the_voxel_filter(gpu_lib):
    from fast_livo_amd import capi, synth
    f = synth.make_imu_frame(30000, n_imu=20, seed=3)
    h = capi.Handle(capi.config_from_frames(f.lio))
    xg = capi.state18_from_frame(f.lio); pg = capi.imu_proc_from_frame(f)
    out, _ = h.imu_undistort(pg, xg, f.imu, f.pcl_beg_time, f.pcl_end_time, f.pts_xyzt)
    ref, _ = orc.voxel_grid(out, 0.15)
    m = C.c_int32(0)
    dev = np.empty((out.shape[0], 4), np.float32)
    rc = h.L.fl_scan_voxel_filter(h.h, None, out.shape[0], 0.15, 0.15, 0.15, 0, dev.ctypes.data_as(C.POINTER(C.c_float)), C.byref(m), None)
    assert rc == 0 and m.value == ref.shape[0]
    assert np.array_equal(dev[:m.value].view(np.uint32), ref.view(np.uint32))
