"""The restatements of oracle/orc_*.c held to the REFERENCE'S OWN SOURCE, compiled where it lies (recipe: oracle/ref_eigen/).

oracle/_ref/libeigen_ref.so holds
  * the reference's include/common_lib.h + include/so3_math.h behind a C driver (and, with Boost + a real Eigen, include/use-ikfom.hpp
    + IKFoM_toolkit),
  * the TEXT of what the reference only has inside main() or inside LidarSelector, read from /root/reference at build time and piped
    into the compiler between hand-written declarations (oracle/ref_eigen/ref_text.sh): the whole Mode-18 loop
    laserMapping.cpp:1506-1732 with pointBodyToWorld :272-286, over the reference's own ikd-Tree; and LidarSelector::set_extrinsic /
    init (Jacobian part) / dpi / UpdateState / updateFrameState / ComputeJ, lidar_selection.cpp:35-59, 92-103, 743-911, 967-983.
This image and the GPU box have no Eigen (NOTES.md).  The recipe then compiles against oracle/ref_eigen/shim -- a small dense-matrix
library of THIS repository behind the part of Eigen's API those sources use; NOT Eigen -- and `eigenref.linalg_kind()` says "shim".
What such a build pins is the reference's own logic (loop order, indices, gates, float / double promotions, operand grouping), not
Eigen's arithmetic (the shim evaluates in the same documented order the oracle does).  Against the shim every row below must hold BIT
FOR BIT; against a real Eigen the float plane fit and the scalar SO(3) functions still must, and the dense algebra gets the tolerance
written next to it:
    esti_plane<float>                       common_lib.h:448-493      bit for bit (float)
    StatesGroup += / -                      common_lib.h:343-365      bit for bit (double)
    Exp / Log                               so3_math.h:54-81          bit for bit
    the Mode-18 loop (a whole frame)        laserMapping.cpp:1506-1732  selection, plane rows, pass / search counts identical;
                                                                      state <= 1e-11, covariance <= 1e-12 relative
    UpdateState / ComputeJ                  lidar_selection.cpp       per-patch errors (float), level errors, state, G, cov likewise
    addFromSparseMap, pixel-level part      lidar_selection.cpp:476-582 over getpatch / getWarpMatrixAffine / warpAffine / NCC /
                                            getBestSearchLevel: accepted set, search levels identical; patches, errors (float) bit for
                                            bit against the shim, <= 1e-4 grey levels against a real Eigen (2x2 inverse, float warp)
    the visual map over several frames      addFromSparseMap :346-587 (whole), addSparseMap :142-202, AddPoint :204-230, addObservation
                                            :913-965 over the reference's own Feature / Point (feature.h, point.h, point.cpp): selected
                                            points, levels, founded points, observation lists identical frame by frame
    lasermap_fov_segment, map_incremental   laserMapping.cpp:361-421, 692-706 over a persistent tree: window, slabs, deleted counts, world
                                            points (float) bit for bit; the tree's content as a set
    ImuProcess::UndistortPcl                IMU_Processing.cpp:611-809  compensated points (float) and kept count identical; poses,
                                                                      state <= 1e-12, covariance <= 1e-12 relative
    h_share_model                           laserMapping.cpp:961-1093  the Mode-23 measurement model's TEXT over the reference's ikd-Tree (the
                                                                      IKFoM state type stood in for by a struct with Eigen::Quaternion
                                                                      members, text/lio_1b.inc): rows h_x, h, selection bit for bit; and
                                                                      the WHOLE update with both halves from the reference's text
    update_iterated_dyn_share_modified      esekfom.hpp:1619-1928     the updater's TEXT over the toolkit's own SO3 / S2 / vect / mtkmath
    vect / SO3 / S2 boxplus, boxminus,      mtk/types/*.hpp,          TEXT (only vectview and the Boost-generated compound state are
    MTK::exp / log / A_matrix, S2_Bx /      mtk/src/mtkmath.hpp       stand-ins, text/ikf_1.inc): callback count identical, state and
    S2_Nx_yy / S2_Mx                                                  covariance <= 1e-12; the single functions bit for bit on 3 000
                                                                      random + edge inputs each
    state_ikfom boxplus / boxminus          use-ikfom.hpp, MTK        <= 1e-15 / 1e-13                    (needs Boost + Eigen)
    update_iterated_dyn_share_modified      esekfom.hpp:1619-1928     state <= 1e-12, covariance <= 1e-12  (needs Boost + Eigen)
Round 4: running this file against the shim found that the oracle's Exp multiplied (1 - cos) into K*K instead of into the left K as the
reference's expression does (one unit in the last place in 14 % of random rotations; oracle/orc_math.h fixed); likewise the few-rows
gain of the updater ((H P) H^T) and MTK::A_matrix ((b K) K) in oracle/orc_ikfom.c.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import eigenref

pytestmark = pytest.mark.skipif(not eigenref.available(), reason="the reference's sources could not be compiled here: " + eigenref.why_not())


def _exact():
    return eigenref.linalg_kind() == "shim"


def _close(a, b, tol):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    if _exact():
        return np.array_equal(a, b)
    return bool(np.abs(a - b).max() <= tol * max(1.0, float(np.abs(b).max())))


def _neighbour_sets(orc, scene, n):
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(n, scene=scene)
    world = fr.world_at(fr.R_prior, fr.p_prior).astype(np.float32)
    nbr, _, valid, _ = orc.knn5_bruteforce(scene.map_xyz, world)
    return nbr[valid != 0]


def test_esti_plane_bit_for_bit(oracle_lib, scene):
    orc = oracle_lib
    sets = _neighbour_sets(orc, scene, 20000)
    # plus degenerate inputs: collinear, coincident, far from the origin
    rng = np.random.default_rng(3)
    extra = [np.repeat(rng.standard_normal((1, 3)).astype(np.float32), 5, 0),
             (np.arange(5, dtype=np.float32)[:, None] * np.array([[1.0, 2.0, 3.0]], np.float32)),
             (rng.standard_normal((5, 3)) * 1e-3 + 500.0).astype(np.float32)]
    sets = np.concatenate([sets, np.stack(extra)])
    differ = 0
    worst = 0.0
    for near in sets:
        near = np.ascontiguousarray(near, np.float32)
        a = np.zeros(4, np.float32)
        ok_a = orc.lib().orc_unit_esti_plane(near.ctypes.data_as(C.POINTER(C.c_float)), C.c_float(0.1), a.ctypes.data_as(C.POINTER(C.c_float)))
        b, ok_b = eigenref.esti_plane(near, 0.1)
        if bool(ok_a) != ok_b or not np.array_equal(a, b, equal_nan=True):
            differ += 1
            if np.isfinite(a).all() and np.isfinite(b).all():
                worst = max(worst, float(np.abs(a - b).max()))
    assert differ == 0, f"{differ} of {len(sets)} planes differ from Eigen {eigenref.lib().ref_eigen_version().decode()} (max |d| {worst:g})"


def test_state18_ops_and_so3_bit_for_bit(oracle_lib):
    orc = oracle_lib
    rng = np.random.default_rng(5)
    for k in range(2000):
        mag = [1.0, 1e-2, 1e-5, 1.1e-5, 1e-9, 0.0][k % 6]
        v = rng.standard_normal(3) * mag
        R_o = np.zeros(9)
        orc.lib().orc_unit_so3_exp(v.ctypes.data_as(C.POINTER(C.c_double)), R_o.ctypes.data_as(C.POINTER(C.c_double)))
        R_r = eigenref.so3_exp(v)
        assert np.array_equal(R_o.reshape(3, 3), R_r)
        l_o = np.zeros(3)
        orc.lib().orc_unit_so3_log(R_o.ctypes.data_as(C.POINTER(C.c_double)), l_o.ctypes.data_as(C.POINTER(C.c_double)))
        assert np.array_equal(l_o, eigenref.so3_log(R_r))
        # StatesGroup += and -
        a = orc.State18.make(R_r, *(rng.standard_normal(3) for _ in range(5)), np.eye(18))
        d = rng.standard_normal(18) * mag
        b = a.copy()
        orc.lib().orc_unit_state18_plus(C.byref(b), d.ctypes.data_as(C.POINTER(C.c_double)))
        v15 = np.concatenate([np.array(a.pos), np.array(a.vel), np.array(a.bg), np.array(a.ba), np.array(a.grav)])
        rot_r, v_r = eigenref.state18_plus(np.array(a.rot), v15, d)
        assert np.array_equal(np.array(b.rot).reshape(3, 3), rot_r)
        assert np.array_equal(np.concatenate([np.array(b.pos), np.array(b.vel), np.array(b.bg), np.array(b.ba), np.array(b.grav)]), v_r)
        out = np.zeros(18)
        orc.lib().orc_unit_state18_minus(C.byref(b), C.byref(a), out.ctypes.data_as(C.POINTER(C.c_double)))
        assert np.array_equal(out, eigenref.state18_minus(rot_r, v_r, np.array(a.rot), v15))


def _tree_knn(scene_map):
    from oracle import ikdref
    tree = ikdref.IkdTree()
    tree.build(scene_map)

    def knn(w):
        xyz, sq, found = tree.nearest(w, 5)
        return xyz, ((found == 5) & (sq[:, 4] <= 5.0)).astype(np.uint8)
    return tree, knn


@pytest.mark.parametrize("n,max_iter,seed,far", [(5000, 3, 1, False),      # N = 5 000, max_iteration 3
                                                  (20000, 4, 2, False),     # avia.yaml max_iteration
                                                  (20000, 10, 3, False),    # NTU_VIRAL.yaml max_iteration
                                                  (300, 4, 4, False),       # a handful of points
                                                  (2000, 4, 5, True)])      # a bad prior: most planes fail their gates, several passes
def test_mode18_loop_text_equals_the_oracle(oracle_lib, scene, n, max_iter, seed, far):
    """The reference's loop, its text compiled as it stands, against orc_lio18_frame -- both over the reference's own ikd-Tree."""
    from fast_livo_amd import synth
    orc = oracle_lib
    fr = synth.make_lio_frame(n, seed=seed, scene=scene)
    if far:
        fr.p_prior = fr.p_prior + np.array([0.25, -0.2, 0.1])
    tree, knn = _tree_knn(scene.map_xyz)
    try:
        xo, xr = orc.state18_from_frame(fr), orc.state18_from_frame(fr)
        ro = orc.lio18_frame(xo, fr.body_xyz, fr.R_LI, fr.t_LI, fr.laser_point_cov, max_iter, knn, nthreads=1)
        rr = eigenref.lio18_frame(xr, fr.body_xyz, scene.map_xyz, fr.R_LI, fr.t_LI, fr.laser_point_cov, max_iter)
    finally:
        tree.close()
    assert ro["status"] == 0 and rr["status"] == 0
    for f in ("iterations", "searches", "effct_feat_num", "converged_last"):
        assert getattr(ro["out"], f) == getattr(rr["out"], f), f
    assert ro["out"].iterations >= 2 and ro["out"].effct_feat_num > 0
    assert np.array_equal(ro["sel"], rr["sel"])
    keep = ro["sel"] != 0
    assert np.array_equal(ro["normvec"][keep], rr["normvec"][keep])           # float plane + float residual: exact either way
    assert _close(ro["out"].total_residual, rr["out"].total_residual, 1e-12)
    assert _close(xo.vec(), xr.vec(), 1e-11)
    assert _close(xo.cov_np(), xr.cov_np(), 1e-12)


@pytest.mark.parametrize("m,seed,distortion", [(200, 1, False), (2000, 2, False), (500, 3, True)])
def test_vio_text_equals_the_oracle(oracle_lib, scene, m, seed, distortion):
    """LidarSelector::UpdateState per level and the whole ComputeJ, the reference's text, against orc_vio_update_state / _compute_j."""
    from fast_livo_amd import synth
    orc = oracle_lib
    lf = synth.make_lio_frame(5000, seed=seed, scene=scene)
    vf = synth.make_vio_frame(m, lf, seed=seed, distortion=distortion)
    xp = orc.state18_from_frame(lf)
    for level in (2, 1, 0):
        xo, xr = orc.state18_from_frame(lf), orc.state18_from_frame(lf)
        G0 = np.random.default_rng(level).standard_normal((18, 18)) * 1e-3     # the member G persists between calls: columns 6.. are kept
        a = orc.vio_update_state(vf, xo, xp, 1e10, level, G=G0.copy())
        b = eigenref.vio_update_state(vf, xr, xp, 1e10, level, G=G0.copy())
        assert a["error"] == b["error"] or (not _exact() and abs(a["error"] - b["error"]) <= 1e-6 * abs(b["error"]))
        assert np.array_equal(a["errors"], b["errors"])                        # per-patch float sums: no Eigen in them
        assert a["out"].iterations >= 2
        assert _close(np.array(a["out"].HTH).reshape(6, 6), b["HTH"], 1e-12)
        assert _close(a["G"], b["G"], 1e-11)
        assert np.array_equal(a["G"][:, 6:], G0[:, 6:])
        assert _close(xo.vec(), xr.vec(), 1e-11)
    xo, xr = orc.state18_from_frame(lf), orc.state18_from_frame(lf)
    ro = orc.vio_compute_j(vf, xo, xp)
    rr = eigenref.vio_compute_j(vf, xr, xp)
    assert ro["status"] == 0 and rr["status"] == 0
    assert np.array_equal(ro["errors"], rr["errors"])
    assert _close(xo.vec(), xr.vec(), 1e-11)
    assert _close(xo.cov_np(), xr.cov_np(), 1e-12)
    assert not np.array_equal(xo.cov_np(), xp.cov_np())                        # cov -= G * cov happened (:978-981)
    # updateFrameState (:904-911): the camera pose of the final state
    Rcw, Pcw = synth.cam_pose(vf.Rcl, vf.Pcl, vf.R_LI, vf.t_LI, np.array(xr.rot).reshape(3, 3), np.array(xr.pos))
    assert np.abs(rr["Tcw"][:9].reshape(3, 3) - Rcw).max() <= 1e-14 and np.abs(rr["Tcw"][9:] - Pcw).max() <= 1e-13


@pytest.mark.parametrize("kw,opt", [(dict(m=60, seed=41), {}),
                                    (dict(m=200, seed=13, n_keyframes=1, discont_frac=0.0), {}),
                                    (dict(m=300, seed=5, n_keyframes=4), dict(ncc_en=True, ncc_thre=0.5)),
                                    (dict(m=300, seed=6, distortion=True), {}),
                                    (dict(m=150, seed=7), dict(outlier_threshold=30.0))])
def test_patch_selection_text_equals_the_oracle(oracle_lib, kw, opt):
    """The loop of addFromSparseMap over the grid winners (depth-continuity test, Warp_map reuse, affine warp of the reference patch on
    three pyramid levels, current patch, NCC gate, outlier gate), the reference's text, against orc_vio_select."""
    from fast_livo_amd import synth
    orc = oracle_lib
    sf = synth.make_select_frame(**kw)
    cfg = orc.vio_config(sf.vio)
    depth = orc.vio_depth_image(cfg, sf.Rcw, sf.Pcw, sf.scan_world)
    a = orc.vio_select(cfg, sf.Rcw, sf.Pcw, sf.vio.img, sf.keyframes, depth, orc.patch_candidates(sf), **opt)
    b = eigenref.vio_select(cfg, sf.Rcw, sf.Pcw, sf.vio.img, sf.keyframes, depth, orc.patch_candidates(sf), **opt)
    assert 0 < len(a["idx"]) < kw["m"]
    assert np.array_equal(a["idx"], b["idx"]) and np.array_equal(a["levels"], b["levels"])
    if _exact():
        assert np.array_equal(a["patches"], b["patches"]) and np.array_equal(a["errors"], b["errors"])
    else:
        assert np.abs(a["patches"] - b["patches"]).max() <= 1e-4 and np.abs(a["errors"] - b["errors"]).max() <= 1e-3 * a["errors"].max()
    rs = np.bincount(a["reason"], minlength=5)
    assert rs[1] > 0 or kw.get("discont_frac") == 0.0            # the depth-discontinuity exit was taken
    if opt.get("ncc_en"):
        assert rs[3] > 0                                         # and the NCC gate


@pytest.mark.parametrize("frames,step,rot", [(4, (0.2, 0.1, 0.0), (0.0, 0.0, 0.005)),          # the sequence of tests/golden/vmap_sequence.json
                                             (36, (0.09, 0.04, 0.0), (0.0, 0.0005, 0.004))])    # long enough for points to reach 20 observations
def test_visual_map_text_equals_the_oracle(oracle_lib, scene, frames, step, rot, capfd):
    """detect()'s per-frame sequence -- addFromSparseMap, addSparseMap, addObservation -- on a persistent LidarSelector, the reference's
    text over its own Feature and Point classes (getCloseViewObs, getFurthestViewObs, deleteFeatureRef, addFrameRef ...), against
    oracle/orc_vmap.c: which map points are selected, their patches, which scan points found new map points, which observations are
    added and which are dropped at the cap of 20."""
    import make_golden_lib as mg
    from fast_livo_amd import synth
    orc = oracle_lib
    lio, vf, scan, _, _ = mg.vmap_case(scene)
    R_wi, p_wi = lio.R_true.copy(), lio.p_true.copy()
    poses, imgs = [], []
    for k in range(frames):
        R_wi = R_wi @ synth.exp_so3(np.array(rot))
        p_wi = p_wi + np.array(step)
        poses.append(synth.cam_pose(vf.Rcl, vf.Pcl, lio.R_LI, lio.t_LI, R_wi, p_wi))
        imgs.append(np.ascontiguousarray(np.roll(vf.img, (k, -2 * k), axis=(0, 1))))
    cfg = orc.vio_config(vf)
    a, b = orc.VMap(cfg, 40), eigenref.VMap(cfg, 40)
    try:
        for k, ((Rcw, Pcw), img) in enumerate(zip(poses, imgs)):
            sa = a.select(Rcw, Pcw, img, imgs[:k + 1], scan, outlier_threshold=1e12)
            sb = b.select(Rcw, Pcw, img, imgs[:k + 1], scan, outlier_threshold=1e12, frame_id=k)
            assert np.array_equal(sa["points"], sb["points"]) and np.array_equal(sa["levels"], sb["levels"]), k
            if _exact():
                assert np.array_equal(sa["errors"], sb["errors"]) and np.array_equal(sa["patches"], sb["patches"]), k
            else:
                assert np.abs(sa["patches"] - sb["patches"]).max() <= 1e-4, k
            assert a.add_sparse(Rcw, Pcw, img, scan, k, k) == b.add_sparse(Rcw, Pcw, img, scan, k, k) > 0
            assert a.add_observation(Rcw, Pcw, img, sa["points"], sa["levels"], k, k) == b.add_observation(Rcw, Pcw, img, sb["points"], sb["levels"], k, k)
            assert a.size() == b.size()
        most = 0
        for i in range(a.size()):
            (pa, va, oa), (pb, vb, ob) = a.get_point(i), b.get_point(i)
            assert np.array_equal(pa, pb) and va == vb and len(oa) == len(ob), i
            most = max(most, len(oa))
            for x, y in zip(oa, ob):
                assert x.frame_id == y.frame_id and x.level == y.level and x.score == y.score
                assert np.array_equal(np.array(x.px), np.array(y.px)) and np.array_equal(np.array(x.R), np.array(y.R))
                assert np.array_equal(np.array(x.t), np.array(y.t))
                assert _close(np.array(x.f), np.array(y.f), 1e-15)
        assert most == (20 if frames > 30 else 2)           # 36 frames: the cap (getFurthestViewObs + deleteFeatureRef) was reached
    finally:
        a.close(); b.close()
        capfd.readouterr()          # the reference's own printf lines ("[ VIO ]: Add %d 3D points." ...)


def test_ntu_viral_configuration_text_equals_the_oracle(oracle_lib, scene):
    """BASELINE config 5 at a CPU-sized scan: NTU_VIRAL extrinsics (zero LiDAR offset), max_iteration 10, the 752x480 radtan camera with
    img_point_cov 1000 -- the LIO frame, then ComputeJ from its posterior."""
    from fast_livo_amd import synth
    orc = oracle_lib
    fr = synth.make_lio_frame(30000, scene=scene, t_LI=synth.NTU_T_LI)
    vf = synth.make_vio_frame(2000, fr, cam=synth.NTU_CAM, Rcl=synth.NTU_RCL, Pcl=synth.NTU_PCL, distortion=True, img_point_cov=1000.0,
                              max_iterations=10)
    tree, knn = _tree_knn(scene.map_xyz)
    try:
        xo, xr = orc.state18_from_frame(fr), orc.state18_from_frame(fr)
        ro = orc.lio18_frame(xo, fr.body_xyz, fr.R_LI, fr.t_LI, fr.laser_point_cov, 10, knn, nthreads=1)
        rr = eigenref.lio18_frame(xr, fr.body_xyz, scene.map_xyz, fr.R_LI, fr.t_LI, fr.laser_point_cov, 10)
    finally:
        tree.close()
    assert ro["out"].iterations == rr["out"].iterations and ro["out"].effct_feat_num == rr["out"].effct_feat_num
    assert np.array_equal(ro["sel"], rr["sel"])
    assert _close(xo.vec(), xr.vec(), 1e-11) and _close(xo.cov_np(), xr.cov_np(), 1e-12)
    po, pr = xo.copy(), xr.copy()
    vo = orc.vio_compute_j(vf, xo, po)
    vr = eigenref.vio_compute_j(vf, xr, pr)
    assert np.array_equal(vo["errors"], vr["errors"])
    assert _close(xo.vec(), xr.vec(), 1e-11) and _close(xo.cov_np(), xr.cov_np(), 1e-12)
    assert sum(vo["outs"][lv].iterations for lv in (2, 1, 0)) >= 6


def test_vio_text_without_patches(oracle_lib, scene):
    """total_points == 0: UpdateState returns 0 and ComputeJ leaves the state alone (:745-746, :969-970)."""
    from fast_livo_amd import synth
    orc = oracle_lib
    lf = synth.make_lio_frame(2000, scene=scene)
    vf = synth.make_vio_frame(8, lf)
    vf.m = 0
    xr, xp = orc.state18_from_frame(lf), orc.state18_from_frame(lf)
    assert eigenref.vio_update_state(vf, xr, xp, 1e10, 0)["error"] == 0.0
    eigenref.vio_compute_j(vf, xr, xp)
    assert np.array_equal(xr.vec(), xp.vec()) and np.array_equal(xr.cov_np(), xp.cov_np())


@pytest.mark.parametrize("kw", [dict(n=5000), dict(n=20000, n_imu=40, seed=5), dict(n=3000, imu_before_frame=False),
                                dict(n=3000, first_point_late=True), dict(n=2000, quiet=True), dict(n=1), dict(n=4000, seed=9, n_imu=7)])
def test_undistort_text_equals_the_oracle(oracle_lib, kw):
    """ImuProcess::UndistortPcl, the reference's text, against orc_imu_undistort.  The reference derives pcl_end_time from the scan's last
    point and keeps the points up to it -- through a product that can round below the float the last point carries, which then stays
    out (always so for a one-point scan); the oracle takes the kept points and the end time as arguments, so it is run on what the
    reference kept."""
    from fast_livo_amd import synth
    orc = oracle_lib
    f = synth.make_imu_frame(**kw)
    so, sr = orc.state18_from_frame(f.lio), orc.state18_from_frame(f.lio)
    po, pr = orc.imu_proc_from_frame(f), orc.imu_proc_from_frame(f)
    pts_r, poses_r, t_end = eigenref.imu_undistort(pr, sr, f.imu, f.pcl_beg_time, f.pts_xyzt)
    kept = len(pts_r)
    assert t_end == f.pcl_end_time
    assert kept >= kw["n"] - 2 and (kept > 0 or kw["n"] == 1)
    pts_o, poses_o = orc.imu_undistort(po, so, f.imu, f.pcl_beg_time, t_end, f.pts_xyzt[:kept])

    def flat(P):
        return np.array([[q.offset_time, *q.acc, *q.gyr, *q.vel, *q.pos, *q.rot] for q in P])
    assert len(poses_o) == len(poses_r) >= 2
    assert _close(flat(poses_o), flat(poses_r), 1e-12)
    assert np.array_equal(pts_o, pts_r) if _exact() else np.abs(pts_o - pts_r).max() <= 1e-6
    assert _close(so.vec(), sr.vec(), 1e-12)
    assert _close(so.cov_np(), sr.cov_np(), 1e-12)
    for fld in ("acc_s_last", "angvel_last"):
        assert _close(np.array(getattr(po, fld)), np.array(getattr(pr, fld)), 1e-12)
    assert po.last_lidar_end_time == pr.last_lidar_end_time and po.last_imu.t == pr.last_imu.t


def test_local_map_text_equals_the_oracle(oracle_lib, scene):
    """The sensor walks until the window has moved three times: lasermap_fov_segment's window and slabs against orc_fov_segment, the
    tree behind it against orc_map_delete_boxes / orc_map_add_points on a flat array, map_incremental's world points against
    fr.world_at rounded as the reference stores them."""
    from test_ref_ikdtree_cpu import sorted_rows
    from fast_livo_amd import synth
    orc = oracle_lib
    cube, det, ds = 24.0, 4.0, 0.5
    lm = eigenref.LocalMap(scene.map_xyz, ds, cube, det)
    try:
        flat = scene.map_xyz.copy()
        win = np.zeros(6, np.float32)
        init = False
        pos = np.array([0.5, -0.3, 1.0])
        moves = 0
        for k in range(40):
            w_r, boxes_r, deleted = lm.fov_segment(pos)
            boxes_o, init = orc.fov_segment(win, init, pos, cube, det, 1.5)
            assert np.array_equal(w_r, win), k
            assert np.array_equal(boxes_r, boxes_o), k
            if len(boxes_o):
                moves += 1
                flat, info = orc.map_delete_boxes(flat, boxes_o)
                assert info.n_removed == deleted
            if k % 4 == 3:                                     # every fourth frame a registered scan goes in
                fr = synth.make_lio_frame(1500, seed=100 + k, scene=scene)
                x = orc.state18_from_frame(fr, R=fr.R_true, p=pos)
                grew, world = lm.incremental(x, fr.R_LI, fr.t_LI, fr.body_xyz)
                assert np.array_equal(world, fr.world_at(fr.R_true, pos))
                flat, info = orc.map_add_points(flat, world, ds)
                assert info.n_after - info.n_before == grew
            assert np.array_equal(sorted_rows(lm.flatten()), sorted_rows(flat)), k
            pos = pos + np.array([0.9, 0.35, 0.0])
        assert moves >= 3
    finally:
        lm.close()


def test_mtk_manifold_text_equals_the_oracle(oracle_lib):
    """The toolkit's own text -- vect / SO3 / S2 boxplus and boxminus over MTK::exp / log / cos_sinc_sqrt, A_matrix, S2_Bx / S2_Nx_yy /
    S2_Mx (with the `scalar(1/2) == 0` quirk of S2.hpp:277) -- against orc_state23_boxplus / boxminus and the oracle's unit functions:
    random states and steps from |d| = 3 down to 0 (the Taylor range of cos_sinc_sqrt, the log's tolerance clamp), S2's fallback chart
    (vec[2] + length <= tolerance) and its opposed-vector branch."""
    orc = oracle_lib
    L = orc.lib()
    rng = np.random.default_rng(7)
    dp = C.POINTER(C.c_double)

    def rand_state():
        s = orc.State23()
        for f, _ in s._fields_:
            a = getattr(s, f)
            a[:] = rng.standard_normal(len(a))
        for f in ("rot", "offset_R_L_I"):
            q = np.array(getattr(s, f))
            getattr(s, f)[:] = q / np.linalg.norm(q)
        g = np.array(s.grav)
        s.grav[:] = g / np.linalg.norm(g) * 9.809
        return s
    tol_p, tol_m = 1e-15 * 10, 1e-13
    for k in range(1500):
        a = rand_state()
        if k % 50 == 0:
            a.grav[:] = [0, 0, -9.809]
        if k % 50 == 1:
            a.grav[:] = [1e-13, 0, -9.809]
        d = rng.standard_normal(23) * [1.0, 1e-2, 1e-3, 1e-6, 1e-13, 0.0, 3.0][k % 7]
        b = a.copy()
        L.orc_state23_boxplus(C.byref(b), d.ctypes.data_as(dp))
        assert _close(b.vec(), eigenref.mtk_boxplus(a.vec(), d), tol_p), k
        out = np.zeros(23)
        L.orc_state23_boxminus(C.byref(b), C.byref(a), out.ctypes.data_as(dp))
        assert _close(out, eigenref.mtk_boxminus(b.vec(), a.vec()), tol_m), k
        if k % 300 == 0:                                       # opposed gravity vectors: `res[0] = 3.1415926` (S2.hpp:155)
            c = a.copy()
            c.grav[:] = [-x for x in a.grav]
            L.orc_state23_boxminus(C.byref(c), C.byref(a), out.ctypes.data_as(dp))
            r = eigenref.mtk_boxminus(c.vec(), a.vec())
            assert _close(out, r, tol_m) and r[21] == 3.1415926
    for k in range(1500):
        v = rng.standard_normal(3) * [1.0, 1e-2, 1e-6, 1e-11, 1e-12, 0.0, 3.0][k % 7]
        o = np.zeros(9)
        L.orc_unit_A_matrix(v.ctypes.data_as(dp), o.ctypes.data_as(dp))
        assert _close(o.reshape(3, 3), eigenref.mtk_A_matrix(v), 1e-15), k
        g = rng.standard_normal(3)
        g = g / np.linalg.norm(g) * 9.809
        if k % 40 == 0:
            g = np.array([0, 0, -9.809])
        dl = rng.standard_normal(2) * [1.0, 1e-3, 1e-12, 0.0][k % 4]
        N, M = np.zeros(6), np.zeros(6)
        L.orc_unit_s2_Nx_yy(g.ctypes.data_as(dp), N.ctypes.data_as(dp))
        L.orc_unit_s2_Mx(g.ctypes.data_as(dp), dl.ctypes.data_as(dp), M.ctypes.data_as(dp))
        _, nx, mx = eigenref.mtk_S2(g, dl)
        assert _close(N.reshape(2, 3), nx, 1e-15) and _close(M.reshape(3, 2), mx, 1e-14), k


def _fixture(name):
    import json
    import os
    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".json")))


def _same(got, want, keys):
    for k in keys:
        a, b = np.asarray(got[k], dtype=np.float64), np.asarray(want[k], dtype=np.float64)
        assert a.shape == b.shape, k
        assert np.abs(a - b).max() <= 1e-12 * max(1.0, np.abs(b).max()), k


def test_committed_fixtures_are_outputs_of_the_reference_text(oracle_lib, scene, capfd):
    """tests/golden/*.json were generated from the CPU oracle (tests/golden/make_golden.py).  Here the same cases are run through the
    reference's own text and compared with the COMMITTED numbers, at the tolerance tests/test_oracle_cpu.py holds the oracle to them:
    the fixtures for the VIO level, the Mode-23 update, the undistortion, the patch selection and the visual-map sequence are outputs
    of the reference's text (the k-NN, map and VoxelGrid fixtures belong to the ikd-Tree / PCL and are pinned or unpinned elsewhere)."""
    import make_golden_lib as mg
    from fast_livo_amd import synth
    orc = oracle_lib
    # vio_level (make_golden_lib.run_vio_level)
    fr = synth.make_lio_frame(256, scene=scene)
    vf = synth.make_vio_frame(40, fr)
    x = orc.state18_from_frame(fr)
    r = eigenref.vio_update_state(vf, x, x.copy(), 1e10, 1)
    _same({"error": [r["error"]], "errors_head": r["errors"][:8].tolist(), "state_after": x.vec().tolist(), "HTH": r["HTH"].reshape(-1).tolist()},
          _fixture("vio_level"), ("error", "errors_head", "state_after", "HTH"))
    # ikfom_update: the updater's text calling h_share_model's text over the reference's tree
    fr = synth.make_lio_frame(1200, scene=scene)
    hm = eigenref.HShareModel(fr.body_xyz, scene.map_xyz)
    try:
        s_r, P_r, calls = eigenref.ikfom_update_text_c(orc.state23_from_frame(fr, synth.quat_from_R).vec(), fr.cov23.copy(), 0.001, 4, hm.callback)
        neff = hm.last()["effct_feat_num"]
    finally:
        hm.close()
    _same({"iterations": [calls], "neff": [neff], "state_after": s_r.tolist(), "P_diag": np.diag(P_r).tolist(), "P_row0": P_r[0].tolist()},
          _fixture("ikfom_update"), ("iterations", "neff", "state_after", "P_diag", "P_row0"))
    # imu_undistort
    f = synth.make_imu_frame(500, n_imu=12, seed=31)
    x, pr = orc.state18_from_frame(f.lio), orc.imu_proc_from_frame(f)
    pts, poses, t_end = eigenref.imu_undistort(pr, x, f.imu, f.pcl_beg_time, f.pts_xyzt)
    assert len(pts) >= 499 and t_end == f.pcl_end_time       # (this scan's last point is one the reference's own selection drops, see
    #                                                           test_undistort_text_equals_the_oracle: `pts_mean` of the fixture counts it)
    _same({"n_poses": [len(poses)], "pose_last_pos": list(poses[-1].pos), "pose_last_rot": list(poses[-1].rot), "state_after": x.vec().tolist(),
           "cov_diag": np.diag(np.array(x.cov).reshape(18, 18)).tolist(), "pts_head": pts[:4].reshape(-1).tolist(),
           "acc_s_last": list(pr.acc_s_last)},
          _fixture("imu_undistort"), ("n_poses", "pose_last_pos", "pose_last_rot", "state_after", "cov_diag", "pts_head", "acc_s_last"))
    # vio_select (the depth image is the oracle's: the reference builds it inside addFromSparseMap, covered by the visual-map sequence)
    sf = synth.make_select_frame(60, seed=41)
    cfg = orc.vio_config(sf.vio)
    depth = orc.vio_depth_image(cfg, sf.Rcw, sf.Pcw, sf.scan_world)
    r = eigenref.vio_select(cfg, sf.Rcw, sf.Pcw, sf.vio.img, sf.keyframes, depth, orc.patch_candidates(sf), outlier_threshold=300.0)
    _same({"idx": r["idx"].tolist(), "levels": r["levels"].tolist(), "errors": r["errors"].tolist(), "patch0": r["patches"][0].tolist(),
           "patches_checksum": mg._ck(r["patches"])}, _fixture("vio_select"), ("idx", "levels", "errors", "patch0") + (("patches_checksum",) if _exact() else ()))
    # vmap_sequence: the whole summary (bitwise checksums over positions, values, observation lists)
    import types
    ref = types.SimpleNamespace(VMap=eigenref.VMap, vio_config=orc.vio_config)
    got = mg.run_vmap_sequence(ref, scene)
    capfd.readouterr()
    want = _fixture("vmap_sequence")
    _same(got, want, ("selected", "added", "obs_added", "n_points") + (("pos_checksum", "value_checksum", "n_obs_checksum", "px_checksum", "frame_checksum") if _exact() else ()))


def _few(cb):
    def w(xs, valid, converge):
        v, hx, hv = cb(xs, valid, converge)
        return v, hx[:15].copy(), hv[:15].copy()
    return w


def _flaky(cb):
    k = dict(i=0)

    def w(xs, valid, converge):
        k["i"] += 1
        v, hx, hv = cb(xs, valid, converge)
        return (k["i"] != 2), hx, hv
    return w


@pytest.mark.parametrize("n,max_iter,pseed,wrap,limit", [
    (3000, 4, None, None, None),            # avia.yaml: rows >= 23 branch (:1779-1806)
    (3000, 10, 1, None, None),              # NTU_VIRAL.yaml max_iteration, a dense prior covariance
    (400, 3, 2, None, None),
    (3000, 4, 3, _few, None),               # fewer rows than states: K = P H^T (H P H^T / R + I)^-1 / R (:1712-1741)
    (2000, 5, 4, _flaky, 1e-30),            # an invalid pass (:1651-1654) + limits nothing meets: forced rematch (:1826-1829), exit at the last pass
])
def test_mode23_updater_text_equals_the_oracle(oracle_lib, scene, n, max_iter, pseed, wrap, limit):
    """esekf::update_iterated_dyn_share_modified, the reference's text compiled as it stands (over the oracle's manifold operations),
    against orc_ikfom_update_dyn_share -- both around the C oracle's h_share_model and an exact brute-force k-NN."""
    import test_cross_oracle_cpu as xo
    from fast_livo_amd import synth
    orc = oracle_lib
    fr = synth.make_lio_frame(n, scene=scene)
    P0 = fr.cov23.copy() if pseed is None else xo._spd23(pseed, 1e-3)
    lim = np.full(23, 0.001 if limit is None else limit)
    wrap = wrap or (lambda cb: cb)
    x_c = orc.state23_from_frame(fr, synth.quat_from_R)
    P_c = P0.copy()
    cb_c, _ = xo._c_rows_callback(orc, fr, scene.map_xyz)
    cnt_c = dict(calls=0, searches=0)
    r_c = orc.ikfom_update_dyn_share(x_c, P_c, 0.001, max_iter, xo._counting(wrap(cb_c), cnt_c), limit=lim)
    cb_r, _ = xo._c_rows_callback(orc, fr, scene.map_xyz)
    cnt_r = dict(calls=0, searches=0)
    s_r, P_r, calls = eigenref.ikfom_update_text(orc.state23_from_frame(fr, synth.quat_from_R).vec(), P0, 0.001, max_iter,
                                                 xo._counting(wrap(cb_r), cnt_r), limit=lim)
    assert calls == cnt_r["calls"] == cnt_c["calls"] == r_c["out"].iterations
    assert cnt_r["searches"] == cnt_c["searches"] >= 2
    assert _close(x_c.vec(), s_r, 1e-12)
    assert _close(P_c, P_r, 1e-12)
    assert not np.array_equal(P_r, P0)


@pytest.mark.parametrize("n,max_iter", [(3000, 4), (20000, 4), (5000, 10)])
def test_mode23_measurement_model_and_whole_update_from_the_reference_text(oracle_lib, scene, n, max_iter):
    """h_share_model's text against orc_h_share_model (a call that searches, a call that does not, at a moved state), then the whole
    Mode-23 update with the reference's updater text calling the reference's h_share_model text -- against orc_ikfom_update_iterated
    over an exact brute-force k-NN."""
    import test_cross_oracle_cpu as xo
    from fast_livo_amd import synth
    orc = oracle_lib
    fr = synth.make_lio_frame(n, scene=scene)
    x0 = orc.state23_from_frame(fr, synth.quat_from_R)
    hm = eigenref.HShareModel(fr.body_xyz, scene.map_xyz)
    try:
        cb_c, st = xo._c_rows_callback(orc, fr, scene.map_xyz)
        d = np.zeros(23)
        d[:6] = [0.01, -0.02, 0.005, 1e-3, -2e-3, 5e-4]
        x1 = x0.copy()
        orc.lib().orc_state23_boxplus(C.byref(x1), d.ctypes.data_as(C.POINTER(C.c_double)))
        for x, search in ((x0, True), (x1, False)):
            _, hx_o, h_o = cb_c(x.copy(), True, search)
            valid, hx_r, h_r = hm.rows(x.vec(), search)
            assert valid and len(h_r) == len(h_o) > 0.9 * n
            assert _close(hx_o, hx_r, 1e-13) and _close(h_o, h_r, 1e-13)
            last = hm.last()
            assert last["effct_feat_num"] == len(h_r)
            assert np.array_equal(last["sel"] != 0, st["sel"] != 0)
        s_r, P_r, calls = eigenref.ikfom_update_text_c(x0.vec(), fr.cov23.copy(), 0.001, max_iter, hm.callback)
    finally:
        hm.close()

    def knn(w):
        nb, _, va, _ = orc.knn5_bruteforce(scene.map_xyz, w)
        return nb, va
    x_o, P_o = orc.state23_from_frame(fr, synth.quat_from_R), fr.cov23.copy()
    ro = orc.ikfom_update(x_o, P_o, fr.body_xyz, 0.001, max_iter, knn, nthreads=1)
    assert calls == ro["out"].iterations
    assert _close(x_o.vec(), s_r, 1e-12) and _close(P_o, P_r, 1e-12)


def test_mode23_against_the_reference_toolkit(oracle_lib, scene):
    if not eigenref.have_mtk():
        pytest.skip("libeigen_ref.so was built without Boost: the MTK / esekf part of the reference is not in it")
    import test_cross_oracle_cpu as xo
    from fast_livo_amd import synth
    orc = oracle_lib
    rng = np.random.default_rng(11)
    # box operators
    for k in range(200):
        mag = [1.0, 1e-2, 1e-3, 1e-6, 1e-13, 0.0][k % 6]
        s = orc.State23()
        for f, _ in s._fields_:
            getattr(s, f)[:] = rng.standard_normal(len(getattr(s, f)))
        for f in ("rot", "offset_R_L_I"):
            q = np.array(getattr(s, f)); getattr(s, f)[:] = q / np.linalg.norm(q)
        g = np.array(s.grav); s.grav[:] = g / np.linalg.norm(g) * 9.809
        d = rng.standard_normal(23) * mag
        b = s.copy()
        orc.lib().orc_state23_boxplus(C.byref(b), d.ctypes.data_as(C.POINTER(C.c_double)))
        assert np.abs(b.vec() - eigenref.state23_boxplus(s.vec(), d)).max() <= 1e-15 * 10
        out = np.zeros(23)
        orc.lib().orc_state23_boxminus(C.byref(b), C.byref(s), out.ctypes.data_as(C.POINTER(C.c_double)))
        assert np.abs(out - eigenref.state23_boxminus(b.vec(), s.vec())).max() <= 1e-13
    # the whole update, the reference's own esekf around the C oracle's h_share_model
    for n, max_iter, P0 in ((3000, 4, None), (3000, 10, xo._spd23(1, 1e-3)), (15, 4, xo._spd23(3, 1e-3))):
        fr = synth.make_lio_frame(n, scene=scene)
        P0 = fr.cov23.copy() if P0 is None else P0
        x_c = orc.state23_from_frame(fr, synth.quat_from_R)
        P_c = P0.copy()
        cb_c, _ = xo._c_rows_callback(orc, fr, scene.map_xyz)
        r_c = orc.ikfom_update_dyn_share(x_c, P_c, 0.001, max_iter, cb_c)
        cb_r, _ = xo._c_rows_callback(orc, fr, scene.map_xyz)
        s_r, P_r, calls = eigenref.ikfom_update_dyn_share(orc.state23_from_frame(fr, synth.quat_from_R).vec(), P0, 0.001, max_iter, cb_r)
        assert calls == r_c["out"].iterations
        assert np.abs(x_c.vec() - s_r).max() <= 1e-12
        assert np.abs(P_c - P_r).max() <= 1e-12 * max(1.0, np.abs(P_r).max())
