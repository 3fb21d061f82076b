"""The C++ host mirror (fast-livo_amd/host: StatesGroup / state_ikfom / dyn_share_datastruct stand-ins and
the shim bodies of INTEGRATION.md) driven from a plain C++ program over the C ABI, compared with the
ctypes path on the same frame and the same (replayed) kNN results."""
import os
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_host_mirror_matches_ctypes_path(gpu_lib, scene, tmp_path):
    capi = gpu_lib
    from fast_livo_amd import synth
    demo = os.path.join(ROOT, "fast-livo_amd", "host", "demo_host")
    if not os.path.exists(demo):
        subprocess.check_call(["make", "-C", os.path.dirname(demo), "-s"])
    n, max_iter = 8000, 5
    fr = synth.make_lio_frame(n, scene=scene)
    recorded = []

    def knn(w):
        nb, va = synth.knn5(scene, w)
        recorded.append((nb.copy(), va.copy()))
        return nb, va
    h = capi.Handle(capi.config_from_frames(fr, max_iterations=max_iter))
    x = capi.state18_from_frame(fr)
    info = h.lio_frame18(x, fr.body_xyz, knn)
    h.close()
    f = tmp_path / "frame.bin"
    with open(f, "wb") as fh:
        fh.write(struct.pack("<iii", n, len(recorded), max_iter))
        fh.write(np.asarray(fr.R_LI, dtype="<f8").tobytes())
        fh.write(np.asarray(fr.t_LI, dtype="<f8").tobytes())
        x0 = capi.state18_from_frame(fr)
        fh.write(x0.vec().astype("<f8").tobytes())
        fh.write(np.asarray(x0.cov_np(), dtype="<f8").tobytes())
        fh.write(fr.body_xyz.astype("<f4").tobytes())
        for nb, va in recorded:
            fh.write(nb.astype("<f4").tobytes())
            fh.write(va.astype("u1").tobytes())
    out = subprocess.run([demo, str(f)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    head = lines[0].split()
    assert int(head[1]) == 0                                  # status
    assert int(head[5]) == info.effct_feat_num
    vals = np.array(lines[1].split(), dtype=np.float64)
    assert np.abs(vals - x.vec()[:12]).max() <= 1e-13          # rot + pos after the frame
    diag = np.array(lines[2].split(), dtype=np.float64)
    assert np.abs(diag - np.diag(x.cov_np())).max() <= 1e-15
    # h_share_model surrogate: 23 x 12 with S^T S = H^T H and S^T h = H^T z
    sur = lines[3].split()
    assert int(sur[2]) == 23 and int(sur[4]) == 1
    assert float(sur[8]) <= 1e-12 and float(sur[10]) <= 1e-9 * max(1.0, float(sur[12]))


def test_cpp_device_pipeline_matches_ctypes_path(gpu_lib, oracle_lib, tmp_path):
    """ImuProcessDev::UndistortPcl -> VoxelGridDev::filter_to_scan -> LioMode18Dev::update from plain C++ (demo_pipeline)
    vs the same three calls through ctypes: identical state."""
    capi = gpu_lib
    import ctypes as C
    from fast_livo_amd import synth
    hostdir = os.path.join(ROOT, "fast-livo_amd", "host")
    demo = os.path.join(hostdir, "demo_pipeline")
    if not os.path.exists(demo):
        subprocess.check_call(["make", "-C", hostdir, "-s", "demo_pipeline"])
    n, max_iter, leaf, cell = 30000, 6, 0.2, 0.5
    lio = synth.make_lio_frame(n)
    f = synth.make_imu_frame(n, n_imu=20, lio=lio, quiet=True)
    f.pts_xyzt[:, :3] = lio.body_xyz
    # the configuration demo_pipeline builds: identity camera extrinsic, fx = fy = 400, principal point (320, 256)
    cam = dict(width=640, height=512, fx=400.0, fy=400.0, cx=320.0, cy=256.0, d=(0.0,) * 5)
    h = capi.Handle(capi.make_config(lio.R_LI, lio.t_LI, np.eye(3), np.zeros(3), cam, max_iterations=max_iter))
    h.map_set_points(lio.scene.map_xyz, cell)
    x = capi.state18_from_frame(lio); pr = capi.imu_proc_from_frame(f)
    pr0 = capi.imu_proc_from_frame(f)
    h.imu_undistort(pr, x, f.imu, f.pcl_beg_time, f.pcl_end_time, f.pts_xyzt, want=False)
    _, m, _ = h.scan_voxel_filter_resident(n, leaf)
    info = h.lio_frame18_dev(x, None)
    # the frame's tail as demo_pipeline runs it: window check (oracle's lasermap_fov_segment), one slab cut off, map_incremental
    orc = oracle_lib
    win = np.zeros(6, dtype=np.float32)
    pos = np.array(x.vec()[9:12])
    _, init = orc.fov_segment(win, False, pos, cube_len=24.0, det_range=4.0, mov_threshold=1.5)
    pos[0] += 8.0
    boxes, init = orc.fov_segment(win, init, pos, 24.0, 4.0, 1.5)
    assert len(boxes) == 1
    di = h.map_delete_boxes(boxes)
    ai = h.map_add_points(None, leaf)
    map_after = h.map_get_points()
    x_lio = x.copy()
    # the camera half as LidarSelectorDev::detect runs it, twice on the same image
    img = synth.make_image(640, 512, seed=3)
    Rci = np.eye(3) @ lio.R_LI.T
    Pci = np.eye(3) @ (-lio.R_LI.T @ lio.t_LI)
    world = h.lio_get_world_points(m)
    down, nd, _ = h.scan_voxel_filter(np.ascontiguousarray(np.concatenate([world, np.zeros((m, 1), np.float32)], axis=1)), 0.2)
    down = np.ascontiguousarray(down[:nd, :3])
    h.vmap_clear(40)
    cam_counts = []
    for f2 in range(2):
        h.vio_set_frame(img)
        kf = h.vio_add_keyframe()

        def pose(st):
            R = np.array(st.rot).reshape(3, 3)
            Rcw = Rci @ R.T
            return Rcw, -(Rcw @ np.array(st.pos[:])) + Pci
        Rcw, Pcw = pose(x)
        g = h.vmap_select(Rcw, Pcw, down, outlier_threshold=1e12, want_patches=False)
        na = h.vmap_add_sparse(Rcw, Pcw, world, kf, f2)
        if len(g["points"]):
            xp = x.copy()
            h.vio_compute_j(x, xp)
            Rcw, Pcw = pose(x)
        no = h.vmap_add_observation(Rcw, Pcw, kf, f2)
        cam_counts.append((len(g["points"]), na, no))
    assert cam_counts[0][1] > 20 and cam_counts[1][0] > 10
    h.close()
    fn = tmp_path / "pipe.bin"
    with open(fn, "wb") as fh:
        fh.write(struct.pack("<iiiiffdd", n, f.imu.shape[0], lio.scene.map_xyz.shape[0], max_iter, leaf, cell, f.pcl_beg_time, f.pcl_end_time))
        fh.write(np.asarray(lio.R_LI, dtype="<f8").tobytes()); fh.write(np.asarray(lio.t_LI, dtype="<f8").tobytes())
        x0 = capi.state18_from_frame(lio)
        fh.write(x0.vec().astype("<f8").tobytes()); fh.write(np.asarray(x0.cov_np(), dtype="<f8").tobytes())
        fh.write(bytes(pr0))
        fh.write(np.ascontiguousarray(f.imu, dtype="<f8").tobytes())
        fh.write(f.pts_xyzt.astype("<f4").tobytes()); fh.write(lio.scene.map_xyz.astype("<f4").tobytes())
        fh.write(np.asarray(Rci, dtype="<f8").tobytes()); fh.write(np.asarray(Pci, dtype="<f8").tobytes()); fh.write(img.tobytes())
    out = subprocess.run([demo, str(fn)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    head = lines[0].split()
    assert int(head[1]) == 0 and int(head[5]) == info.effct_feat_num and int(head[7]) == m
    vals = np.array(lines[1].split(), dtype=np.float64)
    assert np.array_equal(vals, x_lio.vec()[:15])             # rot, pos, vel: bit-identical (%.17g round-trips)
    diag = np.array(lines[2].split(), dtype=np.float64)
    assert np.array_equal(diag, np.diag(x_lio.cov_np()))
    tail = np.array(lines[3].split(), dtype=np.float64)
    assert tail[0] == pr.last_lidar_end_time and tail[1] == pr.acc_s_last[2]
    mp = lines[4].split()
    assert [int(mp[k]) for k in (1, 3, 5, 7, 9, 11)] == [0, 1, di.n_removed, ai.n_before, ai.n_after, ai.n_added]
    assert di.n_removed > 0 and ai.n_added > 0 and ai.n_after == len(map_after)
    for f2 in range(2):
        cl = lines[5 + f2].split()
        assert (int(cl[3]), int(cl[5]), int(cl[7])) == cam_counts[f2], (cl, cam_counts)
    cam_state = np.array(lines[7].split(), dtype=np.float64)
    assert np.array_equal(cam_state, x.vec()[:12])               # state after the second frame's ComputeJ: bit-identical
