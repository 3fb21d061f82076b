"""The camera half of a frame with the visual map on the device (api_vmap.inc), in the order of LidarSelector::detect:
fl_vmap_select (addFromSparseMap, whole) -> fl_vmap_add_sparse (addSparseMap) -> fl_vio_compute_j -> fl_vmap_add_observation,
against the same steps of the CPU oracle. Steady state of a moving camera: the map is grown for a number of frames first."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, "/root/repo")
import fastlivo  # noqa: F401,E402
from fast_livo_amd import capi, synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

frames, warm, n_scan, grid = 40, 15, 20000, 40
rng = np.random.default_rng(5)
scene = synth.make_scene()
lio = synth.make_lio_frame(n_scan, scene=scene)
vf = synth.make_vio_frame(16, lio)
h = capi.Handle(capi.config_from_frames(lio, vf, max_iterations=4))
ocfg = orc.vio_config(vf)
vm = orc.VMap(ocfg, grid)
h.vmap_clear(grid)
scan_all = lio.world_at(lio.R_true, lio.p_true).astype(np.float32)
down, _ = orc.voxel_grid(np.concatenate([scan_all, np.zeros((len(scan_all), 1), np.float32)], axis=1), 0.2)
down = np.ascontiguousarray(down[:, :3])
x0 = capi.state18_from_frame(lio, R=lio.R_true, p=lio.p_true)
kfs = []
tg, tc, nsel, parts = [], [], [], []
R_wi, p_wi = lio.R_true.copy(), lio.p_true.copy()
for k in range(frames):
    R_wi = R_wi @ synth.exp_so3(np.array([0.0, 0.0, 0.003]))
    p_wi = p_wi + np.array([0.03, 0.02, 0.0])
    Rcw, Pcw = synth.cam_pose(vf.Rcl, vf.Pcl, lio.R_LI, lio.t_LI, R_wi, p_wi)
    img = np.ascontiguousarray(np.roll(vf.img, (k % 3, -(k % 4)), axis=(0, 1)))
    h.vio_set_frame(img)
    kf = h.vio_add_keyframe(img)
    kfs.append(img)
    xs = capi.State18.make(R_wi, p_wi, lio.vel, lio.bg, lio.ba, lio.grav, lio.cov18)
    t0 = time.perf_counter()
    g = h.vmap_select(Rcw, Pcw, down, outlier_threshold=1e12, want_patches=False)
    t1 = time.perf_counter()
    h.vmap_add_sparse(Rcw, Pcw, scan_all, kf, k)
    t2 = time.perf_counter()
    if len(g["points"]):
        xg = xs.copy()
        h.vio_compute_j(xg, xs)
    t3 = time.perf_counter()
    h.vmap_add_observation(Rcw, Pcw, kf, k)
    t4 = time.perf_counter()
    c0 = time.perf_counter()
    o = vm.select(Rcw, Pcw, img, kfs, down, outlier_threshold=1e12)
    c1 = time.perf_counter()
    vm.add_sparse(Rcw, Pcw, img, scan_all, kf, k)
    c2 = time.perf_counter()
    vm.add_observation(Rcw, Pcw, img, o["points"], o["levels"], kf, k)
    c3 = time.perf_counter()
    assert np.array_equal(g["points"], o["points"])
    if k >= warm:
        tg.append(((t1 - t0) + (t2 - t1) + (t4 - t3)) * 1e3)
        parts.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
        tc.append((c3 - c0) * 1e3)
        nsel.append(len(o["points"]))
pm = np.median(np.array(parts), axis=0)
print(json.dumps({"scan_points": len(scan_all), "scan_points_down": len(down), "map_points": vm.size(), "selected_per_frame": float(np.median(nsel)),
                  "gpu_map_steps_ms": round(float(np.median(tg)), 3),
                  "gpu_parts_ms(select,add_sparse,compute_j,add_observation)": [round(float(v), 3) for v in pm],
                  "cpu_map_steps_ms (oracle; its voxel matching is brute force)": round(float(np.median(tc)), 1)}))
h.close()
