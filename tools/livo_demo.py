"""A LiDAR-inertial-visual odometry loop with everything map-sized on the device, over a synthetic trajectory: per frame
scan -> fl_scan_voxel_filter -> fl_lio_frame18_dev -> fl_map_add_points  |  image -> fl_vmap_select -> fl_vmap_add_sparse ->
fl_vio_compute_j -> fl_vmap_add_observation. Reports the wall time per frame of the two halves and the tracking error against
the synthetic truth (scan noise 1 cm, image noise 1 grey level, images rendered from a texture attached to the world)."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, "/root/repo")
import fastlivo  # noqa: F401,E402
from fast_livo_amd import capi, synth  # noqa: E402

frames, n_raw, leaf, map_ds, grid = 20, 24000, 0.15, 0.25, 40
scene = synth.make_scene()
fr0 = synth.make_lio_frame(1000, scene=scene)
vf0 = synth.make_vio_frame(8, fr0, max_iterations=4)
h = capi.Handle(capi.config_from_frames(fr0, vf0, max_iterations=4))
h.vmap_clear(grid)
R_t, p_t = fr0.R_true.copy(), fr0.p_true.copy()
x = capi.State18.make(R_t, p_t, fr0.vel, fr0.bg, fr0.ba, fr0.grav, fr0.cov18)
h.map_clear(0.0)
h.lio_set_points(synth.scan_from_pose(scene, R_t, p_t, 60000, seed=1)); h.lio_begin18(x, x)
h.map_add_points(None, 0.0)
Q = np.diag([1e-5] * 3 + [1e-4] * 3 + [1e-3] * 3 + [1e-8] * 9)
t_lidar, t_cam, err_lio, err_livo, tracked = [], [], [], [], []
for k in range(frames):
    R_t = R_t @ synth.exp_so3(np.array([0.0, 0.0, 0.01]))
    p_t = p_t + np.array([0.05, 0.03, 0.0])
    body = synth.scan_from_pose(scene, R_t, p_t, n_raw, seed=100 + k)
    raw = np.ascontiguousarray(np.concatenate([body, np.zeros((n_raw, 1), np.float32)], axis=1))
    Rc_t, Pc_t = synth.cam_pose(vf0.Rcl, vf0.Pcl, fr0.R_LI, fr0.t_LI, R_t, p_t)
    img = synth.render_image(scene, vf0.cam, Rc_t, Pc_t, seed=k)
    x = capi.State18.make(np.array(x.rot).reshape(3, 3), x.pos[:], x.vel[:], x.bg[:], x.ba[:], x.grav[:], x.cov_np() + Q)
    h.sync()
    t0 = time.perf_counter()
    _, m, _ = h.scan_voxel_filter(raw, leaf, stage_as_scan=True, want=False)
    h.lio_frame18_dev(x, None)
    h.map_add_points(None, map_ds)
    t1 = time.perf_counter()
    err_lio.append(float(np.linalg.norm(np.array(x.pos[:]) - p_t)))
    world = h.lio_get_world_points(m)                      # (outside the timing: the camera half takes the registered scan)
    down, _, _ = h.scan_voxel_filter(np.ascontiguousarray(np.concatenate([world, np.zeros((m, 1), np.float32)], axis=1)), 0.2,
                                     stage_as_scan=False)
    down = np.ascontiguousarray(down[:, :3])
    h.sync()
    t2 = time.perf_counter()
    Rcw, Pcw = synth.cam_pose(vf0.Rcl, vf0.Pcl, fr0.R_LI, fr0.t_LI, np.array(x.rot).reshape(3, 3), np.array(x.pos[:]))
    h.vio_set_frame(img)
    kf = h.vio_add_keyframe()
    g = h.vmap_select(Rcw, Pcw, down, outlier_threshold=3000.0, want_patches=False)
    h.vmap_add_sparse(Rcw, Pcw, world, kf, k)
    if len(g["points"]):
        xp = x.copy()
        h.vio_compute_j(x, xp)
    Rcw, Pcw = synth.cam_pose(vf0.Rcl, vf0.Pcl, fr0.R_LI, fr0.t_LI, np.array(x.rot).reshape(3, 3), np.array(x.pos[:]))
    h.vmap_add_observation(Rcw, Pcw, kf, k)
    t3 = time.perf_counter()
    err_livo.append(float(np.linalg.norm(np.array(x.pos[:]) - p_t)))
    tracked.append(len(g["points"]))
    if k >= 3:
        t_lidar.append((t1 - t0) * 1e3); t_cam.append((t3 - t2) * 1e3)
print(json.dumps({"frames": frames, "raw_points": n_raw, "lidar_half_ms": round(float(np.median(t_lidar)), 3),
                  "camera_half_ms": round(float(np.median(t_cam)), 3), "tracked_patches_median": float(np.median(tracked[1:])),
                  "visual_map_points": h.vmap_size(), "lidar_map_points": int(len(h.map_get_points())),
                  "pos_err_after_lio_cm(mean,max)": [round(100 * float(np.mean(err_lio)), 2), round(100 * float(np.max(err_lio)), 2)],
                  "pos_err_after_vio_cm(mean,max)": [round(100 * float(np.mean(err_livo)), 2), round(100 * float(np.max(err_livo)), 2)]}))
h.close()
