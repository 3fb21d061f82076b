"""The other two shipped configurations as parity cases (avia is every other test's default, NTU_VIRAL is test_config5_gpu.py):
  MARS_LVIG  config/MARS_LVIG.yaml + camera_MARS_LVIG.yaml: 1224 x 1024 radtan camera, its camera-LiDAR extrinsic, avia LiDAR-IMU extrinsic
  mid360     config/mid360.yaml: max_iteration 5, its LiDAR-IMU translation and camera-LiDAR extrinsic (a camera pitched 9 degrees), pinhole intrinsics
Each: the all-device LIO frame (searches, plane fits, passes, covariance) and ComputeJ (3 pyramid levels, lens distortion on) against
the CPU oracle -- iteration and accept counts, per-patch errors bit for bit, states 1e-9, covariance 1e-11."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _case(synth, scene, name):
    if name == "MARS_LVIG":
        fr = synth.make_lio_frame(30000, scene=scene)
        vf = synth.make_vio_frame(800, fr, cam=synth.MARS_CAM, Rcl=synth.MARS_RCL, Pcl=synth.MARS_PCL, distortion=True, img_point_cov=100.0,
                                  max_iterations=10)
        return fr, vf, 10
    fr = synth.make_lio_frame(30000, scene=scene, t_LI=synth.MID360_T_LI)
    vf = synth.make_vio_frame(800, fr, Rcl=synth.MID360_RCL, Pcl=synth.MID360_PCL, distortion=True, img_point_cov=100.0, max_iterations=5)
    return fr, vf, 5


@pytest.mark.parametrize("name", ["MARS_LVIG", "mid360"])
def test_shipped_config_full_frame(gpu_lib, oracle_lib, scene, name):
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    from helpers import copy_state
    fr, vf, max_iter = _case(synth, scene, name)

    def knn(w):
        nb, _, va, _ = orc.knn5_bruteforce(scene.map_xyz, w)
        return nb, va
    xo = orc.state18_from_frame(fr)
    ro = orc.lio18_frame(xo, fr.body_xyz, fr.R_LI, fr.t_LI, fr.laser_point_cov, max_iter, knn, nthreads=8)
    xvo = xo.copy()
    rv = orc.vio_compute_j(vf, xvo, xo.copy())

    h = capi.Handle(capi.config_from_frames(fr, vf, max_iterations=max_iter))
    h.map_set_points(scene.map_xyz, 0.5)
    xg = capi.state18_from_frame(fr)
    info = h.lio_frame18_dev(xg, fr.body_xyz)
    assert info.status == 0 and info.iterations == ro["out"].iterations and info.effct_feat_num == ro["out"].effct_feat_num
    assert np.abs(xg.vec() - xo.vec()).max() <= 1e-9
    assert np.abs(xg.cov_np() - xo.cov_np()).max() <= 1e-11
    h.vio_set_frame(vf.img); h.vio_set_patches(vf.ref_patch, vf.pos, vf.search_level)
    xvg = copy_state(capi.State18, xo)          # ComputeJ from the oracle's LIO posterior on both sides: per-patch errors comparable bit for bit
    infos = h.vio_compute_j(xvg, copy_state(capi.State18, xo))
    for lv in (2, 1, 0):
        assert infos[lv].iterations == rv["outs"][lv].iterations and infos[lv].accepted == rv["outs"][lv].accepted, (name, lv)
        assert infos[lv].iterations <= max_iter
    assert np.abs(xvg.vec() - xvo.vec()).max() <= 1e-9
    assert np.abs(xvg.cov_np() - xvo.cov_np()).max() <= 1e-11
    assert np.array_equal(h.vio_get_errors(vf.m), rv["errors"])
    h.close()
