"""The HIP path held DIRECTLY to the reference's own text (no oracle in between).

oracle/_ref/libeigen_ref.so (recipe oracle/ref_eigen/, prebuilt where /root/reference exists and shipped with the snapshot) holds the
reference's Mode-18 loop (laserMapping.cpp:1506-1732, over its own ikd-Tree), LidarSelector::UpdateState / ComputeJ and the
patch-selection loop with warpAffine / getpatch / NCC (lidar_selection.cpp), ImuProcess::UndistortPcl (IMU_Processing.cpp:611-809)
and esekf::update_iterated_dyn_share_modified (esekfom.hpp:1619-1928) compiled from the reference's source text -- over a real Eigen
where one exists, otherwise over the stand-in oracle/ref_eigen/shim (`eigenref.linalg_kind()`).  tests/test_ref_eigen_cpu.py shows the
CPU oracle equal to it bit for bit; here the device results are compared with it at the tolerances the oracle tests use:
integer / float-pixel work bit-exact, fp64 state 1e-9, covariance 1e-12 (north_star).
"""
import numpy as np
import pytest

from oracle import eigenref

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not eigenref.available(), reason="oracle/_ref/libeigen_ref.so not available: " + eigenref.why_not())]


@pytest.mark.parametrize("n,max_iter", [(5000, 3), (50000, 4), (20000, 10)])
def test_lio_frame_equals_the_reference_loop(gpu_lib, oracle_lib, scene, n, max_iter):
    """fl_lio_frame18_dev (device k-NN, multi-pass kernels, covariance update) vs the reference's loop over the reference's tree."""
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(n, scene=scene)
    xr = orc.state18_from_frame(fr)
    rr = eigenref.lio18_frame(xr, fr.body_xyz, scene.map_xyz, fr.R_LI, fr.t_LI, fr.laser_point_cov, max_iter)
    h = capi.Handle(capi.config_from_frames(fr, max_iterations=max_iter))
    h.map_set_points(scene.map_xyz, 0.5)
    xg = capi.state18_from_frame(fr)
    info = h.lio_frame18_dev(xg, fr.body_xyz)
    assert info.status == 0 and info.stop == 1
    assert info.iterations == rr["out"].iterations
    assert info.effct_feat_num == rr["out"].effct_feat_num
    assert np.abs(xg.vec() - xr.vec()).max() <= 1e-9
    assert np.abs(xg.cov_np() - xr.cov_np()).max() <= 1e-12
    mask, normvec = h.lio_get_selection(n)
    keep = rr["sel"] != 0                                                   # the last pass's point_selected_surf
    assert np.array_equal(mask != 0, keep & (np.abs(rr["normvec"][:, 3]) <= 2.0))   # ... && res_last <= 2.0 (:1593): the effective points
    assert np.array_equal(normvec[keep].view(np.uint32), rr["normvec"][keep].view(np.uint32))   # float plane + residual: bit for bit
    h.close()


@pytest.mark.parametrize("m,distortion", [(500, False), (2000, False), (500, True)])
def test_compute_j_equals_the_reference_text(gpu_lib, oracle_lib, scene, m, distortion):
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(2000, scene=scene)
    vf = synth.make_vio_frame(m, fr, distortion=distortion)
    xr = orc.state18_from_frame(fr)
    rr = eigenref.vio_compute_j(vf, xr, orc.state18_from_frame(fr))
    h = capi.Handle(capi.config_from_frames(fr, vf, max_iterations=vf.max_iterations))
    h.vio_set_frame(vf.img)
    h.vio_set_patches(vf.ref_patch, vf.pos, vf.search_level)
    xg = capi.state18_from_frame(fr)
    h.vio_compute_j(xg, xg.copy())
    assert np.abs(xg.vec() - xr.vec()).max() <= 1e-9
    assert np.abs(xg.cov_np() - xr.cov_np()).max() <= 1e-12
    e = h.vio_get_errors(vf.m)
    assert np.array_equal(np.asarray(e, np.float32).view(np.uint32), np.asarray(rr["errors"], np.float32).view(np.uint32))   # per-patch float errors: bit for bit
    h.close()


def test_config5_full_frame_equals_the_reference_text(gpu_lib, oracle_lib, scene):
    """BASELINE config 5 (NTU_VIRAL: 200 000 points, max_iteration 10, 752x480 radtan camera, img_point_cov 1000, 2 000 patches): the
    all-device LIO frame and ComputeJ against the reference's loop over its own tree and its ComputeJ."""
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    from helpers import copy_state
    fr = synth.make_lio_frame(200000, scene=scene, t_LI=synth.NTU_T_LI)
    vf = synth.make_vio_frame(2000, fr, cam=synth.NTU_CAM, Rcl=synth.NTU_RCL, Pcl=synth.NTU_PCL, distortion=True, img_point_cov=1000.0,
                              max_iterations=10)
    xr = orc.state18_from_frame(fr)
    rr = eigenref.lio18_frame(xr, fr.body_xyz, scene.map_xyz, fr.R_LI, fr.t_LI, fr.laser_point_cov, 10)
    h = capi.Handle(capi.config_from_frames(fr, vf, max_iterations=10))
    h.map_set_points(scene.map_xyz, 0.5)
    xg = capi.state18_from_frame(fr)
    info = h.lio_frame18_dev(xg, fr.body_xyz)
    assert info.status == 0 and info.iterations == rr["out"].iterations and info.effct_feat_num == rr["out"].effct_feat_num
    assert np.abs(xg.vec() - xr.vec()).max() <= 1e-9
    assert np.abs(xg.cov_np() - xr.cov_np()).max() <= 1e-11
    # ComputeJ from the reference's LIO posterior on both sides: per-patch errors bit for bit
    xvg, prop = copy_state(capi.State18, xr), copy_state(capi.State18, xr)
    vr = eigenref.vio_compute_j(vf, xr, xr.copy())
    h.vio_set_frame(vf.img)
    h.vio_set_patches(vf.ref_patch, vf.pos, vf.search_level)
    h.vio_compute_j(xvg, prop)
    assert np.abs(xvg.vec() - xr.vec()).max() <= 1e-9
    assert np.abs(xvg.cov_np() - xr.cov_np()).max() <= 1e-11
    assert np.array_equal(h.vio_get_errors(vf.m), vr["errors"])
    h.close()


@pytest.mark.parametrize("m,opt", [(600, {}), (600, dict(ncc_en=True, ncc_thre=0.5))])
def test_patch_selection_equals_the_reference_text(gpu_lib, oracle_lib, m, opt):
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    sf = synth.make_select_frame(m)
    h = capi.Handle(capi.config_from_frames(sf.lio, sf.vio))
    ids = [h.vio_add_keyframe(k) for k in sf.keyframes]
    h.vio_set_frame(sf.vio.img)
    dev = h.vio_select_patches(sf.Rcw, sf.Pcw, sf.scan_world, capi.patch_candidates(sf, ids), outlier_threshold=sf.outlier_threshold,
                               want_depth=True, **opt)
    cfg = orc.vio_config(sf.vio)
    ref = eigenref.vio_select(cfg, sf.Rcw, sf.Pcw, sf.vio.img, sf.keyframes, dev["depth"], orc.patch_candidates(sf),
                              outlier_threshold=sf.outlier_threshold, **opt)
    assert 0 < len(ref["idx"]) < m
    assert np.array_equal(dev["idx"], ref["idx"]) and np.array_equal(dev["levels"], ref["levels"])
    if eigenref.linalg_kind() == "shim":
        assert np.array_equal(dev["errors"].view(np.uint32), ref["errors"].view(np.uint32))
        assert np.array_equal(dev["patches"].view(np.uint32), ref["patches"].view(np.uint32))
    else:
        assert np.abs(dev["patches"] - ref["patches"]).max() <= 1e-4
    h.close()


def test_undistortion_equals_the_reference_text(gpu_lib, oracle_lib):
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    f = synth.make_imu_frame(24000, n_imu=20)
    sr, pr = orc.state18_from_frame(f.lio), orc.imu_proc_from_frame(f)
    pts_r, poses_r, t_end = eigenref.imu_undistort(pr, sr, f.imu, f.pcl_beg_time, f.pts_xyzt)
    kept = len(pts_r)
    assert kept >= 24000 - 2 and t_end == f.pcl_end_time
    h = capi.Handle(capi.config_from_frames(f.lio))
    xg, pg = capi.state18_from_frame(f.lio), capi.imu_proc_from_frame(f)
    out, poses = h.imu_undistort(pg, xg, f.imu, f.pcl_beg_time, t_end, f.pts_xyzt[:kept])
    assert len(poses) == len(poses_r)
    for a, b in zip(poses, poses_r):
        assert np.allclose(np.frombuffer(bytes(a), np.float64), np.frombuffer(bytes(b), np.float64), rtol=1e-12, atol=1e-12)
    assert np.allclose(np.frombuffer(bytes(xg), np.float64), np.frombuffer(bytes(sr), np.float64), rtol=1e-11, atol=1e-14)
    assert np.abs(out[:, :3].astype(np.float64) - pts_r[:, :3].astype(np.float64)).max() <= 4e-6
    h.close()


@pytest.mark.parametrize("n,max_iter", [(20000, 4), (3000, 10)])
def test_mode23_update_equals_the_reference_updater_text(gpu_lib, oracle_lib, scene, n, max_iter):
    """fl_ikfom_update_iterated_dev vs the reference's updater text driven by the C oracle's h_share_model over an exact k-NN."""
    import test_cross_oracle_cpu as xo
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(n, scene=scene)
    cb, _ = xo._c_rows_callback(orc, fr, scene.map_xyz)
    cnt = dict(calls=0, searches=0)
    s_r, P_r, calls = eigenref.ikfom_update_text(orc.state23_from_frame(fr, synth.quat_from_R).vec(), fr.cov23.copy(), 0.001, max_iter,
                                                 xo._counting(cb, cnt))
    h = capi.Handle(capi.config_from_frames(fr, max_iterations=max_iter))
    h.map_set_points(scene.map_xyz, 0.5)
    xg = capi.state23_from_frame(fr)
    Pg = fr.cov23.copy()
    info = h.ikfom_update_iterated_dev(xg, Pg, fr.body_xyz, 0.001)
    assert info.iterations == calls
    assert np.abs(xg.vec() - s_r).max() <= 1e-9
    assert np.abs(Pg - P_r).max() <= 1e-10 * max(1.0, np.abs(P_r).max())
    h.close()


def test_visual_map_sequence_equals_the_reference_text(gpu_lib, oracle_lib, scene, capfd):
    """The device's visual map (api_vmap.inc) over 8 frames against the reference's addFromSparseMap / addSparseMap / addObservation
    text on a persistent LidarSelector: same harness as tests/test_vmap_gpu.py with the reference in the oracle's place."""
    import types

    import test_vmap_gpu as tv
    from fast_livo_amd import synth
    ref = types.SimpleNamespace(vio_config=oracle_lib.vio_config, voxel_grid=oracle_lib.voxel_grid, VMap=eigenref.VMap)
    st = tv._run(gpu_lib, ref, synth, scene, frames=8, step=np.array([0.04, 0.02, 0.0]), thr=300.0)
    capfd.readouterr()
    assert st["added"] > 100 and st["selected"] > 0


@pytest.mark.parametrize("n,max_iter", [(50000, 4), (5000, 10)])
def test_mode23_update_equals_the_reference_text_in_both_halves(gpu_lib, oracle_lib, scene, n, max_iter):
    """fl_ikfom_update_iterated_dev (BASELINE config 2 at n = 50 000) vs the reference's updater text calling the reference's
    h_share_model text over the reference's ikd-Tree."""
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(n, scene=scene)
    hm = eigenref.HShareModel(fr.body_xyz, scene.map_xyz)
    try:
        s_r, P_r, calls = eigenref.ikfom_update_text_c(orc.state23_from_frame(fr, synth.quat_from_R).vec(), fr.cov23.copy(), 0.001, max_iter,
                                                       hm.callback)
        eff = hm.last()["effct_feat_num"]
    finally:
        hm.close()
    h = capi.Handle(capi.config_from_frames(fr, max_iterations=max_iter))
    h.map_set_points(scene.map_xyz, 0.5)
    xg = capi.state23_from_frame(fr)
    Pg = fr.cov23.copy()
    info = h.ikfom_update_iterated_dev(xg, Pg, fr.body_xyz, 0.001)
    assert info.iterations == calls and info.effct_feat_num == eff
    assert np.abs(xg.vec() - s_r).max() <= 1e-9
    assert np.abs(Pg - P_r).max() <= 1e-10 * max(1.0, np.abs(P_r).max())
    h.close()
