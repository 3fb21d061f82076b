"""Whole LiDAR front of a frame on the device (SURVEY 8f N4 -> N3 -> N1 -> a): raw scan + IMU samples in, updated state out.

GPU:  fl_imu_undistort (cloud stays on the device) -> fl_scan_voxel_filter(NULL, stage_as_scan) -> fl_lio_frame18_dev(NULL)
      [-> fl_map_add_points(NULL): map_incremental on the device map, timed as a second figure]
CPU:  oracle UndistortPcl -> oracle VoxelGrid -> oracle Mode-18 frame with a cKDTree 5-NN (4 threads)"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlivo  # noqa
from fast_livo_amd import capi, synth
from oracle import oracle as orc

ap = argparse.ArgumentParser()
ap.add_argument("--raw", type=int, default=100000)
ap.add_argument("--leaf", type=float, default=0.15)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--map-ds", type=float, default=0.15, help="filter_size_map_min of the map update stage")
ap.add_argument("--cell", type=float, default=0.5, help="k-NN cell edge (2-3x the map's point spacing); 0 = automatic")
a = ap.parse_args()
lio = synth.make_lio_frame(a.raw)
f = synth.make_imu_frame(a.raw, n_imu=20, lio=lio, quiet=True)
f.pts_xyzt[:, :3] = lio.body_xyz                   # the raw scan = every synthetic return, in time order
scene = lio.scene
h = capi.Handle(capi.config_from_frames(lio, max_iterations=10))
h.map_set_points(scene.map_xyz, a.cell)
knn = lambda w: synth.knn5(scene, w)

def gpu(map_update=False):
    x = capi.state18_from_frame(lio); pr = capi.imu_proc_from_frame(f)
    h.imu_undistort(pr, x, f.imu, f.pcl_beg_time, f.pcl_end_time, f.pts_xyzt, want=False)
    _, m, _ = h.scan_voxel_filter_resident(a.raw, a.leaf)
    info = h.lio_frame18_dev(x, None)
    if map_update:
        h.map_add_points(None, a.map_ds)
    return x, m, info

def cpu():
    x = orc.state18_from_frame(lio); pr = orc.imu_proc_from_frame(f)
    t0 = time.perf_counter(); pts, _ = orc.imu_undistort(pr, x, f.imu, f.pcl_beg_time, f.pcl_end_time, f.pts_xyzt); t1 = time.perf_counter()
    vox, _ = orc.voxel_grid(pts, a.leaf); t2 = time.perf_counter()
    out = orc.lio18_frame(x, np.ascontiguousarray(vox[:, :3]), lio.R_LI, lio.t_LI, lio.laser_point_cov, 10, knn, nthreads=4); t3 = time.perf_counter()
    return x, vox.shape[0], out, (t1 - t0, t2 - t1, t3 - t2)

xg, mg, ig = gpu()
xc, mc, oc, _ = cpu()
sg = np.frombuffer(bytes(xg), np.float64); sc = np.frombuffer(bytes(xc), np.float64)
res = {"raw_points": a.raw, "leaf": a.leaf, "map_ds": a.map_ds, "cell": a.cell, "scan_points_gpu": mg, "scan_points_cpu": mc, "iterations_gpu": ig.iterations,
       "effective_points_gpu": ig.effct_feat_num, "state_max_abs_diff_gpu_vs_cpu": float(np.abs(sg - sc).max())}
ts = []
for _ in range(a.reps):
    t0 = time.perf_counter(); gpu(); ts.append(time.perf_counter() - t0)
res["gpu_pipeline_ms"] = round(float(np.median(ts)) * 1e3, 3)
ts = []
for _ in range(a.reps):       # the map now changes from frame to frame, as in a running system
    t0 = time.perf_counter(); gpu(True); ts.append(time.perf_counter() - t0)
res["gpu_pipeline_with_map_update_ms"] = round(float(np.median(ts)) * 1e3, 3)
res["map_points_after"] = int(len(h.map_get_points()))
cs = []; parts = []
for _ in range(3):
    t0 = time.perf_counter(); r = cpu(); cs.append(time.perf_counter() - t0); parts.append(r[3])
res["cpu_pipeline_ms"] = round(float(np.median(cs)) * 1e3, 1)
res["cpu_parts_ms(undistort,voxel,lio_frame)"] = [round(float(np.median([p[i] for p in parts])) * 1e3, 2) for i in range(3)]
res["speedup"] = round(res["cpu_pipeline_ms"] / res["gpu_pipeline_ms"], 1)
print(json.dumps(res))
