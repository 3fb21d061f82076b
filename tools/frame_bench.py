#!/usr/bin/env python3
"""Whole-frame LIO update (Mode-18, laserMapping.cpp:1504-1733 incl. the 2 search passes):
CPU oracle + cKDTree  vs  GPU passes + host kNN callback  vs  GPU passes + device k-NN (all on device)."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fastlivo  # noqa
from fast_livo_amd import capi, synth
from oracle import oracle as orc

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=50000)
ap.add_argument("--max-iter", type=int, default=10)
ap.add_argument("--reps", type=int, default=20)
a = ap.parse_args()
scene = synth.make_scene()
fr = synth.make_lio_frame(a.points, scene=scene)
knn = lambda w: synth.knn5(scene, w)
h = capi.Handle(capi.config_from_frames(fr, max_iterations=a.max_iter))
res = {"points": a.points, "map_points": int(len(scene.map_xyz)), "max_iterations": a.max_iter}

t0 = time.perf_counter(); h.map_set_points(scene.map_xyz, 0.5); h.sync(); res["map_build_first_ms"] = (time.perf_counter() - t0) * 1e3
ts = []
for _ in range(10):
    t0 = time.perf_counter(); h.map_set_points(scene.map_xyz, 0.5); h.sync(); ts.append(time.perf_counter() - t0)
res["map_build_ms"] = float(np.median(ts) * 1e3)

# search alone
x = capi.state18_from_frame(fr); h.lio_set_points(fr.body_xyz); h.lio_begin18(x, x)
h.set_timing(True)
ks = []
for _ in range(20):
    h.lio_search18(fr.n, want=False); ks.append(h.last_kernel_ms() * 1e3)
res["search_fit_us"] = float(np.median(ks))
h.set_timing(False)

def timed(fn, reps):
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); out = fn(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts) * 1e3), out

def gpu_dev():
    xg = capi.state18_from_frame(fr); return h.lio_frame18_dev(xg, fr.body_xyz)
def gpu_host():
    xg = capi.state18_from_frame(fr); return h.lio_frame18(xg, fr.body_xyz, knn)
def cpu():
    xo = orc.state18_from_frame(fr); return orc.lio18_frame(xo, fr.body_xyz, fr.R_LI, fr.t_LI, fr.laser_point_cov, a.max_iter, knn, nthreads=4)
res["gpu_all_device_frame_ms"], info = timed(gpu_dev, a.reps)
res["iterations"] = info.iterations
res["gpu_host_knn_frame_ms"], _ = timed(gpu_host, max(3, a.reps // 4))
res["cpu_oracle_frame_ms"], _ = timed(cpu, 3)
# host kNN alone (cKDTree, all cores) for reference
w = fr.world_at(fr.R_prior, fr.p_prior)
t0 = time.perf_counter(); knn(w); res["host_ckdtree_search_ms"] = (time.perf_counter() - t0) * 1e3
res["frames_per_s_all_device"] = 1e3 / res["gpu_all_device_frame_ms"]
print(json.dumps(res))
