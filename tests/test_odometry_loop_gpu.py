"""The LiDAR loop of the reference as a loop: per frame the scan is registered against the map (k-NN + ESKF passes, all on the
device), the window follows the sensor (lasermap_fov_segment) and the registered scan goes into the map (map_incremental) -- the
map never leaves the device. The same loop on the CPU oracle (brute-force 5-NN on the oracle's own map, sequential Add_Points):
states equal to 1e-9 and map arrays identical frame after frame, so an error anywhere (search, filter, map update, window)
would compound and show."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_trajectory_with_the_map_on_the_device(gpu_lib, oracle_lib, scene):
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    frames, n_scan, max_iter, ds = 10, 4000, 4, 0.25
    fr0 = synth.make_lio_frame(n_scan, scene=scene)
    h = capi.Handle(capi.config_from_frames(fr0, max_iterations=max_iter))
    rng = np.random.default_rng(77)
    # truth: a gentle arc; the filter is seeded with the truth of frame 0 and carries its own estimate from then on
    R_t, p_t = fr0.R_true.copy(), fr0.p_true.copy()
    step_rot, step_pos = np.array([0.0, 0.0, 0.012]), np.array([0.06, 0.03, 0.0])
    xg = capi.State18.make(R_t, p_t, fr0.vel, fr0.bg, fr0.ba, fr0.grav, fr0.cov18)
    xo = orc.State18.make(R_t, p_t, fr0.vel, fr0.bg, fr0.ba, fr0.grav, fr0.cov18)
    # first frame: ikdtree.Build(feats_down_world) -- the map is the first scan under the initial pose, no down-sampling
    body0 = synth.scan_from_pose(scene, R_t, p_t, 3 * n_scan, seed=1000)
    h.map_clear(0.0)
    h.lio_set_points(body0); h.lio_begin18(xg, xg)
    h.map_add_points(None, 0.0)
    map_o = h.map_get_points().copy()                    # world points of frame 0 (lio_world_points parity is tested elsewhere)
    assert len(map_o) == 3 * n_scan
    win = np.zeros(6, dtype=np.float32)
    init = False
    Q = np.diag([1e-5] * 3 + [1e-4] * 3 + [1e-3] * 3 + [1e-8] * 9)
    err_pos = []
    deleted = 0
    for k in range(1, frames + 1):
        R_t = R_t @ synth.exp_so3(step_rot)
        p_t = p_t + step_pos
        body = synth.scan_from_pose(scene, R_t, p_t, n_scan, seed=2000 + k)
        # prediction = last estimate, covariance inflated (the IMU propagation has its own tests)
        xg = capi.State18.make(np.array(xg.rot).reshape(3, 3), xg.pos[:], xg.vel[:], xg.bg[:], xg.ba[:], xg.grav[:], xg.cov_np() + Q)
        xo = orc.State18.make(np.array(xo.rot).reshape(3, 3), xo.pos[:], xo.vel[:], xo.bg[:], xo.ba[:], xo.grav[:], xo.cov_np() + Q)
        # --- window (oracle arithmetic drives both sides; a small cube so that it moves within the test)
        boxes, init = orc.fov_segment(win, init, np.array(xo.pos[:]), cube_len=30.0, det_range=8.0, mov_threshold=1.5)
        if len(boxes):
            di = h.map_delete_boxes(boxes)
            map_o, oi = orc.map_delete_boxes(map_o, boxes)
            assert di.n_removed == oi.n_removed
            deleted += oi.n_removed
        # --- registration
        info = h.lio_frame18_dev(xg, body)

        def knn(w, m=map_o):
            nb, _, va, _ = orc.knn5_bruteforce(m, w)
            return nb, va
        ro = orc.lio18_frame(xo, body, fr0.R_LI, fr0.t_LI, fr0.laser_point_cov, max_iter, knn)
        assert info.iterations == ro["out"].iterations and info.effct_feat_num == ro["out"].effct_feat_num, f"frame {k}"
        assert np.abs(xg.vec() - xo.vec()).max() <= 1e-9, f"frame {k}"
        assert np.abs(xg.cov_np() - xo.cov_np()).max() <= 1e-11, f"frame {k}"
        # --- map_incremental under the updated state
        mi = h.map_add_points(None, ds)
        world = h.lio_get_world_points(n_scan)
        map_o, oi = orc.map_add_points(map_o, world, ds)
        assert mi.n_ambiguous == oi.n_ambiguous == 0
        assert np.array_equal(h.map_get_points(), map_o), f"frame {k}"
        err_pos.append(float(np.linalg.norm(np.array(xg.pos[:]) - p_t)))
    # it is an odometry: the estimate follows the truth (scan noise 1 cm), and the loop exercised every part
    assert max(err_pos) < 0.03, err_pos
    assert mi.cell_size > 0 and len(map_o) > n_scan
    h.close()
