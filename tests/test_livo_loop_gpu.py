"""LiDAR and camera of the same frames on ONE handle, everything map-sized on the device, against the same loop on the oracle:
per frame  LIO (k-NN + ESKF passes) -> map_incremental -> [image rendered from the scene at the true pose] addFromSparseMap ->
addSparseMap -> ComputeJ (photometric update of the LIO posterior) -> addObservation.
The images come from a texture attached to the world (synth.render_image), so the photometric update is a real alignment and
the two filters share state and covariance from frame to frame. Required per frame: the same points tracked, states equal to
1e-9 after the LIO block and after ComputeJ, LiDAR map arrays and visual map identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _VF:                       # what oracle.vio_compute_j / vio_config read
    pass


def test_livo_frames_on_one_handle(gpu_lib, oracle_lib, scene):
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    frames, n_scan, max_iter, grid = 6, 3000, 4, 40
    fr0 = synth.make_lio_frame(n_scan, scene=scene)
    vf0 = synth.make_vio_frame(8, fr0, max_iterations=max_iter)
    h = capi.Handle(capi.config_from_frames(fr0, vf0, max_iterations=max_iter))
    ocfg = orc.vio_config(vf0)
    vm = orc.VMap(ocfg, grid)
    h.vmap_clear(grid)
    R_t, p_t = fr0.R_true.copy(), fr0.p_true.copy()
    xg = capi.State18.make(R_t, p_t, fr0.vel, fr0.bg, fr0.ba, fr0.grav, fr0.cov18)
    xo = orc.State18.make(R_t, p_t, fr0.vel, fr0.bg, fr0.ba, fr0.grav, fr0.cov18)
    body0 = synth.scan_from_pose(scene, R_t, p_t, 3 * n_scan, seed=500)
    h.map_clear(0.0)
    h.lio_set_points(body0); h.lio_begin18(xg, xg)
    h.map_add_points(None, 0.0)
    map_o = h.map_get_points().copy()
    Q = np.diag([1e-5] * 3 + [1e-4] * 3 + [1e-3] * 3 + [1e-8] * 9)
    kf_imgs = []
    tracked = []
    for k in range(frames):
        R_t = R_t @ synth.exp_so3(np.array([0.0, 0.0, 0.01]))
        p_t = p_t + np.array([0.05, 0.03, 0.0])
        body = synth.scan_from_pose(scene, R_t, p_t, n_scan, seed=600 + k)
        Rc_t, Pc_t = synth.cam_pose(vf0.Rcl, vf0.Pcl, fr0.R_LI, fr0.t_LI, R_t, p_t)
        img = synth.render_image(scene, vf0.cam, Rc_t, Pc_t, seed=k)
        xg = capi.State18.make(np.array(xg.rot).reshape(3, 3), xg.pos[:], xg.vel[:], xg.bg[:], xg.ba[:], xg.grav[:], xg.cov_np() + Q)
        xo = orc.State18.make(np.array(xo.rot).reshape(3, 3), xo.pos[:], xo.vel[:], xo.bg[:], xo.ba[:], xo.grav[:], xo.cov_np() + Q)
        # ---- LiDAR
        info = h.lio_frame18_dev(xg, body)

        def knn(w, m=map_o):
            nb, _, va, _ = orc.knn5_bruteforce(m, w)
            return nb, va
        ro = orc.lio18_frame(xo, body, fr0.R_LI, fr0.t_LI, fr0.laser_point_cov, max_iter, knn)
        assert info.iterations == ro["out"].iterations and info.effct_feat_num == ro["out"].effct_feat_num, f"frame {k}"
        assert np.abs(xg.vec() - xo.vec()).max() <= 1e-9, f"frame {k} LIO"
        mi = h.map_add_points(None, 0.25)
        world = h.lio_get_world_points(n_scan)                       # pg in the world frame, as the camera half receives it
        map_o, oi = orc.map_add_points(map_o, world, 0.25)
        assert mi.n_ambiguous == oi.n_ambiguous == 0 and np.array_equal(h.map_get_points(), map_o), f"frame {k} map"
        # ---- camera (LidarSelector::detect)
        down, _ = orc.voxel_grid(np.concatenate([world, np.zeros((n_scan, 1), np.float32)], axis=1), 0.2)
        down = np.ascontiguousarray(down[:, :3])
        Rcw, Pcw = synth.cam_pose(vf0.Rcl, vf0.Pcl, fr0.R_LI, fr0.t_LI, np.array(xo.rot).reshape(3, 3), np.array(xo.pos[:]))
        h.vio_set_frame(img)
        kf = h.vio_add_keyframe()             # the staged image becomes the keyframe
        kf_imgs.append(img)
        g = h.vmap_select(Rcw, Pcw, down, outlier_threshold=3000.0)
        o = vm.select(Rcw, Pcw, img, kf_imgs, down, outlier_threshold=3000.0)
        assert np.array_equal(g["points"], o["points"]) and np.array_equal(g["patches"], o["patches"]), f"frame {k} select"
        assert h.vmap_add_sparse(Rcw, Pcw, world, kf, k) == vm.add_sparse(Rcw, Pcw, img, world, kf, k)
        m = len(o["points"])
        tracked.append(m)
        if m > 0:
            vf = _VF()
            for a in ("Rcl", "Pcl", "R_LI", "t_LI", "cam", "img_point_cov", "max_iterations", "patch_size"):
                setattr(vf, a, getattr(vf0, a))
            vf.m = m; vf.img = img
            vf.ref_patch = np.ascontiguousarray(o["patches"].reshape(m, 3, 64))
            vf.pos = np.ascontiguousarray(np.stack([vm.get_point(int(i))[0] for i in o["points"]]))
            vf.search_level = np.ascontiguousarray(o["levels"].astype(np.int32))
            xpg, xpo = xg.copy(), xo.copy()
            h.vio_compute_j(xg, xpg)
            orc.vio_compute_j(vf, xo, xpo)
            assert np.abs(xg.vec() - xo.vec()).max() <= 1e-9, f"frame {k} VIO"
            assert np.abs(xg.cov_np() - xo.cov_np()).max() <= 1e-11, f"frame {k} VIO cov"
        Rc2, Pc2 = synth.cam_pose(vf0.Rcl, vf0.Pcl, fr0.R_LI, fr0.t_LI, np.array(xo.rot).reshape(3, 3), np.array(xo.pos[:]))
        assert h.vmap_add_observation(Rc2, Pc2, kf, k) == vm.add_observation(Rc2, Pc2, img, o["points"], o["levels"], kf, k)
        assert np.linalg.norm(np.array(xg.pos[:]) - p_t) < 0.05, f"frame {k}: lost track"
    assert h.vmap_size() == vm.size() and sum(tracked[1:]) > 60, tracked
    for i in range(0, vm.size(), 7):
        pg, vg, obg = h.vmap_get_point(i)
        po, vo, obo = vm.get_point(i)
        assert np.array_equal(pg, po) and vg == vo and len(obg) == len(obo)
    vm.close(); h.close()
