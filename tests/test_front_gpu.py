"""fl_lidar_front (csrc/api_front.inc): the LiDAR half of a frame in one enqueue -- UndistortPcl (IMU_Processing.cpp:611-809) ->
downSizeFilterSurf (laserMapping.cpp:1398-1399) -> the Mode-18 iterated update over the device map (:1504-1733) -- against the three staged
calls it replaces (fl_imu_undistort, fl_scan_voxel_filter, fl_lio_frame18_dev; each held to the oracle and to the reference's text by its own
tests): EVERY BIT of the state, the covariance, the ImuProcess members carried to the next frame, the scan size, the counters and the
per-point selection must be the same -- the fused form only removes the host round trips between the stages.  And once directly against the
CPU oracle's pipeline at the tolerance of tools/pipeline_bench.py."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _frame(synth, raw, seed=0):
    lio = synth.make_lio_frame(raw)
    f = synth.make_imu_frame(raw, n_imu=20, lio=lio, quiet=True)
    f.pts_xyzt[:, :3] = lio.body_xyz                   # the raw scan = every synthetic return, in time order
    return lio, f


def _run(capi, h, lio, f, leaf, staged, frames=1):
    x = capi.state18_from_frame(lio); pr = capi.imu_proc_from_frame(f)
    outs = []
    for _ in range(frames):                            # the same measurement again from the carried state: proc and state both move on
        info, m = h.lidar_front(pr, x, f.imu, f.pcl_beg_time, f.pcl_end_time, f.pts_xyzt, leaf, staged=staged)
        mask, normvec = h.lio_get_selection(m)
        outs.append((bytes(x), bytes(pr), m, info.iterations, info.effct_feat_num, info.status, info.stop, mask.copy(), normvec.copy()))
    return outs


@pytest.mark.parametrize("raw,leaf,max_iter", [(24000, 0.15, 10), (100000, 0.15, 10), (60000, 0.3, 4), (3000, 0.5, 3)])
def test_fused_front_equals_the_staged_calls(gpu_lib, raw, leaf, max_iter):
    capi = gpu_lib
    from fast_livo_amd import synth
    lio, f = _frame(synth, raw)
    res = []
    for staged in (True, False):
        h = capi.Handle(capi.config_from_frames(lio, max_iterations=max_iter))
        h.map_set_points(lio.scene.map_xyz, 0.5)
        res.append(_run(capi, h, lio, f, leaf, staged, frames=3))
        h.close()
    for a, b in zip(*res):
        assert a[2] == b[2] and a[3:7] == b[3:7], (a[2:7], b[2:7])
        assert a[0] == b[0]                            # fl_state18: rot, pos, vel, biases, gravity, cov -- bit for bit
        assert a[1] == b[1]                            # fl_imu_proc
        assert np.array_equal(a[7], b[7]) and np.array_equal(a[8].view(np.uint32), b[8].view(np.uint32))
        assert a[5] == 0 and a[6] == 1 and a[2] > 0 and a[4] > a[2] // 4


def test_fused_front_alternating_with_other_calls_on_the_handle(gpu_lib):
    """the fused frame leaves the handle as the staged calls do: a staged frame, a plain fl_lio_frame18_dev with a host scan and a voxel
    filter call in between change nothing of the next fused frame's result"""
    capi = gpu_lib
    from fast_livo_amd import synth
    lio, f = _frame(synth, 40000)
    h = capi.Handle(capi.config_from_frames(lio, max_iterations=10))
    h.map_set_points(lio.scene.map_xyz, 0.5)
    first = _run(capi, h, lio, f, 0.2, False)[0]
    _run(capi, h, lio, f, 0.2, True)
    small = synth.make_lio_frame(5000)
    xs = capi.state18_from_frame(small)
    h.lio_frame18_dev(xs, small.body_xyz)
    p = np.concatenate([small.body_xyz, np.zeros((small.n, 1), np.float32)], 1).astype(np.float32)
    h.scan_voxel_filter(p, 0.4)
    again = _run(capi, h, lio, f, 0.2, False)[0]
    assert first[0] == again[0] and first[1] == again[1] and first[2:7] == again[2:7]
    h.close()


def test_fused_front_against_the_cpu_oracle(gpu_lib):
    capi = gpu_lib
    from fast_livo_amd import synth
    lio, f = _frame(synth, 50000)
    h = capi.Handle(capi.config_from_frames(lio, max_iterations=10))
    h.map_set_points(lio.scene.map_xyz, 0.5)
    x = capi.state18_from_frame(lio); pr = capi.imu_proc_from_frame(f)
    info, m = h.lidar_front(pr, x, f.imu, f.pcl_beg_time, f.pcl_end_time, f.pts_xyzt, 0.15)
    xo = orc.state18_from_frame(lio); po = orc.imu_proc_from_frame(f)
    pts, _ = orc.imu_undistort(po, xo, f.imu, f.pcl_beg_time, f.pcl_end_time, f.pts_xyzt)
    vox, _ = orc.voxel_grid(pts, 0.15)
    out = orc.lio18_frame(xo, np.ascontiguousarray(vox[:, :3]), lio.R_LI, lio.t_LI, lio.laser_point_cov, 10, lambda w: synth.knn5(lio.scene, w), nthreads=4)
    assert m == vox.shape[0] and info.iterations == out["out"].iterations
    sg = np.frombuffer(bytes(x), np.float64); sc = np.frombuffer(bytes(xo), np.float64)
    assert np.abs(sg - sc).max() <= 1e-9
    h.close()


@pytest.mark.parametrize("kind", ["shuffled", "one_swap", "nan_time", "first_point_late"])
def test_fused_front_with_a_cloud_out_of_time_order(gpu_lib, kind):
    """The fused frame undistorts with ONE kernel that assumes the cloud's times never decrease and checks it (imu_kernels.h
    undistort_sorted_kernel); a cloud that fails the check is run again with the general kernels (the reference's loops handle any order,
    IMU_Processing.cpp:778-808, its sort is commented out, :647). Either way every bit equals the staged calls' -- also for the frames that
    follow (the handle keeps to the general kernels for a while, then tries the fast one again)."""
    capi = gpu_lib
    from fast_livo_amd import synth
    lio = synth.make_lio_frame(30000)
    if kind == "first_point_late":
        f = synth.make_imu_frame(30000, n_imu=20, lio=lio, quiet=True, first_point_late=True)
    else:
        f = synth.make_imu_frame(30000, n_imu=20, lio=lio, quiet=True)
    f.pts_xyzt[:, :3] = lio.body_xyz
    rng = np.random.default_rng(3)
    if kind == "shuffled":
        f.pts_xyzt = np.ascontiguousarray(f.pts_xyzt[rng.permutation(len(f.pts_xyzt))])
    elif kind == "one_swap":
        f.pts_xyzt[[20000, 20001]] = f.pts_xyzt[[20001, 20000]]
        assert f.pts_xyzt[20000, 3] > f.pts_xyzt[20001, 3]
    elif kind == "nan_time":
        f.pts_xyzt[12345, 3] = np.nan
    res = []
    for staged in (True, False):
        h = capi.Handle(capi.config_from_frames(lio, max_iterations=6))
        h.map_set_points(lio.scene.map_xyz, 0.5)
        res.append(_run(capi, h, lio, f, 0.2, staged, frames=3))
        h.close()
    for a, b in zip(*res):
        assert a[2] == b[2] and a[3:7] == b[3:7], (a[2:7], b[2:7])
        assert a[0] == b[0] and a[1] == b[1]
        assert np.array_equal(a[7], b[7]) and np.array_equal(a[8].view(np.uint32), b[8].view(np.uint32))


@pytest.mark.parametrize("nth", [1, 2])
def test_fused_front_losing_the_multipass_admission(gpu_lib, nth):
    """fl_lidar_front keeps the scan's size on the device, which only the multi-pass LIO kernel can read. If its reservation is refused (another
    handle of the process launched in between; here: fl_debug_mp_refuse of the instrumented build, first or second pass launch of the frame),
    nothing more is enqueued, the voxel count is read back and the frame is finished with launches that know it. Same bits as the staged calls."""
    capi = gpu_lib
    from fast_livo_amd import synth
    lio, f = _frame(synth, 30000)
    hs = capi.Handle(capi.config_from_frames(lio, max_iterations=6))
    hs.map_set_points(lio.scene.map_xyz, 0.5)
    want = _run(capi, hs, lio, f, 0.2, True, frames=2)
    hs.close()
    h = capi.Handle(capi.config_from_frames(lio, max_iterations=6), debug=True)
    h.map_set_points(lio.scene.map_xyz, 0.5)
    x = capi.state18_from_frame(lio); pr = capi.imu_proc_from_frame(f)
    got = []
    for k in range(2):
        f0 = h.diagnostics()["fallbacks"]
        if k == 1:
            h.debug_mp_refuse(nth, 1)
        info, m = h.lidar_front(pr, x, f.imu, f.pcl_beg_time, f.pcl_end_time, f.pts_xyzt, 0.2)
        if k == 1:
            assert h.diagnostics()["fallbacks"] - f0 >= 1
            h.debug_mp_refuse(0, 0)
        mask, normvec = h.lio_get_selection(m)
        got.append((bytes(x), bytes(pr), m, info.iterations, info.effct_feat_num, info.status, info.stop, mask.copy(), normvec.copy()))
    h.close()
    for a, b in zip(want, got):
        assert a[2] == b[2] and a[3:7] == b[3:7], (a[2:7], b[2:7])
        assert a[0] == b[0] and a[1] == b[1]
        assert np.array_equal(a[7], b[7]) and np.array_equal(a[8].view(np.uint32), b[8].view(np.uint32))
