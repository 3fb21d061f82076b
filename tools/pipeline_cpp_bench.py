#!/usr/bin/env python3
"""The LiDAR front of a frame (undistortion -> voxel filter -> Mode-18 update) timed from PLAIN C++ over the C ABI: writes the frame file
fast-livo_amd/host/demo_pipeline reads (same layout as tests/test_host_mirror_gpu.py) and runs it with FL_DEMO_TIME_REPS.  The figure beside
tools/pipeline_bench.py's `gpu_pipeline_ms`, which drives the same three calls through python + ctypes.

    python tools/pipeline_cpp_bench.py [--raw 100000] [--leaf 0.15] [--cell 0.5] [--reps 30]
"""
import argparse
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fastlivo  # noqa: E402,F401
from fast_livo_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--raw", type=int, default=100000)
ap.add_argument("--leaf", type=float, default=0.15)
ap.add_argument("--cell", type=float, default=0.5)
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--camera", action="store_true", help="also time the camera half (LidarSelectorDev::detect)")
a = ap.parse_args()
hostdir = os.path.join(ROOT, "fast-livo_amd", "host")
demo = os.path.join(hostdir, "demo_pipeline")
subprocess.check_call(["make", "-C", hostdir, "-s", "demo_pipeline"])
lio = synth.make_lio_frame(a.raw)
f = synth.make_imu_frame(a.raw, n_imu=20, lio=lio, quiet=True)
f.pts_xyzt[:, :3] = lio.body_xyz
pr0 = capi.imu_proc_from_frame(f)
x0 = capi.state18_from_frame(lio)
with tempfile.TemporaryDirectory() as d:
    fn = os.path.join(d, "pipe.bin")
    with open(fn, "wb") as fh:
        fh.write(struct.pack("<iiiiffdd", a.raw, f.imu.shape[0], lio.scene.map_xyz.shape[0], 10, a.leaf, a.cell, f.pcl_beg_time, f.pcl_end_time))
        fh.write(np.asarray(lio.R_LI, dtype="<f8").tobytes()); fh.write(np.asarray(lio.t_LI, dtype="<f8").tobytes())
        fh.write(x0.vec().astype("<f8").tobytes()); fh.write(np.asarray(x0.cov_np(), dtype="<f8").tobytes())
        fh.write(bytes(pr0))
        fh.write(np.ascontiguousarray(f.imu, dtype="<f8").tobytes())
        fh.write(f.pts_xyzt.astype("<f4").tobytes()); fh.write(lio.scene.map_xyz.astype("<f4").tobytes())
        if a.camera:      # the camera half: extrinsics of the state frame + a 640 x 512 grey image (tests/test_host_mirror_gpu.py)
            Rci = np.eye(3) @ lio.R_LI.T
            Pci = np.eye(3) @ (-lio.R_LI.T @ lio.t_LI)
            fh.write(np.asarray(Rci, dtype="<f8").tobytes()); fh.write(np.asarray(Pci, dtype="<f8").tobytes())
            fh.write(synth.make_image(640, 512, seed=3).tobytes())
    out = subprocess.run([demo, fn], capture_output=True, text=True, timeout=600, env=dict(os.environ, FL_DEMO_TIME_REPS=str(a.reps)))
print(out.stdout.strip().splitlines()[0])
for line in out.stderr.strip().splitlines():
    if line.startswith("lidar_front") or line.startswith("camera_half"):
        print(line)
sys.exit(out.returncode)
