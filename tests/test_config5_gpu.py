"""BASELINE config 5 as ONE case: NTU_VIRAL (config/NTU_VIRAL.yaml:3,5,15,32-35,43-46 and camera_NTU_VIRAL.yaml: 752x480 radtan camera,
img_point_cov 1000, zero LiDAR->IMU translation, max_iteration 10), an outdoor-scale scan of 200 000 points + 2 000 patches, the FULL
frame: all-device LIO block (k-NN searches, plane fits, passes, covariance) followed by ComputeJ (3 pyramid levels) -- on one GPU and
sharded over 2 ranks (point/patch ranges, in-kernel peer exchange) -- against the CPU oracle's frame loop."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _frames(synth, scene):
    fr = synth.make_lio_frame(200000, scene=scene, t_LI=synth.NTU_T_LI)
    vf = synth.make_vio_frame(2000, fr, cam=synth.NTU_CAM, Rcl=synth.NTU_RCL, Pcl=synth.NTU_PCL, distortion=True, img_point_cov=1000.0,
                              max_iterations=10)
    return fr, vf


def test_ntu_viral_full_frame_one_gpu_and_two_ranks(gpu_lib, oracle_lib, scene):
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    fr, vf = _frames(synth, scene)
    n, m, max_iter = fr.n, vf.m, 10

    # ---- oracle: LIO frame (k-NN by the cKDTree stand-in, validated against the reference's ikd-Tree elsewhere), then ComputeJ
    def knn(w):
        return synth.knn5(scene, w)
    xo = orc.state18_from_frame(fr)
    ro = orc.lio18_frame(xo, fr.body_xyz, fr.R_LI, fr.t_LI, fr.laser_point_cov, max_iter, knn, nthreads=8)
    xvo = xo.copy()
    rv = orc.vio_compute_j(vf, xvo, xo.copy())

    # ---- one GPU
    cfg = capi.config_from_frames(fr, vf, max_iterations=max_iter)
    h = capi.Handle(cfg)
    h.map_set_points(scene.map_xyz, 0.5)
    xg = capi.state18_from_frame(fr)
    info = h.lio_frame18_dev(xg, fr.body_xyz)
    assert info.status == 0 and info.iterations == ro["out"].iterations and info.effct_feat_num == ro["out"].effct_feat_num
    assert np.abs(xg.vec() - xo.vec()).max() <= 1e-9
    assert np.abs(xg.cov_np() - xo.cov_np()).max() <= 1e-11
    h.vio_set_frame(vf.img); h.vio_set_patches(vf.ref_patch, vf.pos, vf.search_level)
    # ComputeJ starts from the ORACLE's LIO posterior on both sides, so that per-patch errors can be compared bit for bit
    from helpers import copy_state
    xvg = copy_state(capi.State18, xo)
    prop = copy_state(capi.State18, xo)
    infos = h.vio_compute_j(xvg, prop)
    for lv in (2, 1, 0):
        assert infos[lv].iterations == rv["outs"][lv].iterations and infos[lv].accepted == rv["outs"][lv].accepted, lv
    assert np.abs(xvg.vec() - xvo.vec()).max() <= 1e-9
    assert np.abs(xvg.cov_np() - xvo.cov_np()).max() <= 1e-11
    assert np.array_equal(h.vio_get_errors(m), rv["errors"])           # per-patch errors as the reference rounds them
    h.close()

    # ---- two ranks (both on this device), contiguous halves of the points and of the patches, map and image replicated
    hs = [capi.Handle(cfg) for _ in range(2)]
    capi.p2p_connect_local(hs)
    out, err = [None, None], [None, None]

    def rank(r):
        try:
            hh = hs[r]
            hh.map_set_points(scene.map_xyz, 0.5)
            x = capi.state18_from_frame(fr)
            i_ = hh.lio_frame18_dev(x, fr.body_xyz[r * n // 2:(r + 1) * n // 2])
            hh.vio_set_frame(vf.img)
            sl = slice(r * m // 2, (r + 1) * m // 2)
            hh.vio_set_patches(vf.ref_patch[sl], vf.pos[sl], vf.search_level[sl])
            xv = capi.State18.make(np.array(x.rot).reshape(3, 3), x.pos[:], x.vel[:], x.bg[:], x.ba[:], x.grav[:], x.cov_np())
            pr = capi.State18.make(np.array(x.rot).reshape(3, 3), x.pos[:], x.vel[:], x.bg[:], x.ba[:], x.grav[:], x.cov_np())
            hh.vio_compute_j(xv, pr)
            out[r] = (i_, x, xv)
        except Exception as e:      # noqa: BLE001
            err[r] = e
    th = [threading.Thread(target=rank, args=(r,)) for r in range(2)]
    [t.start() for t in th]
    [t.join(300) for t in th]
    assert not any(t.is_alive() for t in th), "a rank hangs"
    assert err == [None, None], err
    (i0, xa, va), (i1, xb, vb) = out
    assert i0.status == 0 and i1.status == 0 and i0.iterations == i1.iterations == ro["out"].iterations
    assert np.array_equal(xa.vec(), xb.vec()) and np.array_equal(xa.cov_np(), xb.cov_np())        # ranks bitwise equal
    assert np.abs(xa.vec() - xo.vec()).max() <= 1e-9 and np.abs(xa.cov_np() - xo.cov_np()).max() <= 1e-11
    assert np.array_equal(va.vec(), vb.vec())
    assert np.abs(va.vec() - xvo.vec()).max() <= 1e-8      # (VIO starts from each side's own LIO posterior, equal to 1e-9)
    for hh in hs:
        hh.close()
