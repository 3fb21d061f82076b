"""Device map maintenance (fl_map_add_points = map_incremental, fl_map_delete_boxes = lasermap_fov_segment's deletion):
wall time per call (host clock around the call, which ends with the one host sync it needs) and the device time between its
events, for a local map of K points and a down-sampled scan of N points. Run under rocprofv3 --kernel-trace --stats for the
per-kernel split."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, "/root/repo")
import fastlivo  # noqa: F401,E402
from fast_livo_amd import capi, synth  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
rng = np.random.default_rng(0)
scene = synth.make_scene()
fr = synth.make_lio_frame(N, scene=scene)
base = scene.map_xyz
reps = (K + len(base) - 1) // len(base)
m0 = (np.tile(base, (reps, 1))[:K] + rng.normal(0, 0.05, (K, 3))).astype(np.float32)
h = capi.Handle(capi.config_from_frames(fr, max_iterations=4))
h.set_timing(True)
out = {"map_points": K, "scan_points": N}
for name, ds in (("add_downsample_0.5", 0.5), ("add_downsample_0.2", 0.2), ("append_only", 0.0)):
    wall, dev, after = [], [], 0
    for r in range(12):
        h.map_set_points(m0, 0.5)
        new = (base[rng.integers(0, len(base), N)] + rng.normal(0, 0.03, (N, 3))).astype(np.float32)
        h.sync()
        t0 = time.perf_counter()
        info = h.map_add_points(new, ds)
        wall.append(time.perf_counter() - t0)
        dev.append(h.last_kernel_ms())
        after = info.n_after
    out[name] = {"wall_ms_median": round(float(np.median(wall[2:])) * 1e3, 3), "device_ms_median": round(float(np.median(dev[2:])), 3),
                 "n_after": int(after), "n_added": int(info.n_added), "n_removed": int(info.n_removed)}
# resident scan (no upload): after a frame
h.map_set_points(m0, 0.5)
x = capi.state18_from_frame(fr)
h.lio_frame18_dev(x, fr.body_xyz)
wall = []
for r in range(10):
    h.map_set_points(m0, 0.5); h.sync()
    t0 = time.perf_counter(); info = h.map_add_points(None, 0.5); wall.append(time.perf_counter() - t0)
out["add_resident_scan_0.5"] = {"wall_ms_median": round(float(np.median(wall[2:])) * 1e3, 3), "n_added": int(info.n_added)}
lo, hi = m0.min(0), m0.max(0)
box = np.array([[lo[0], lo[1], lo[2], lo[0] + 0.3 * (hi[0] - lo[0]), hi[1] + 1, hi[2] + 1]], dtype=np.float32)
wall = []
for r in range(10):
    h.map_set_points(m0, 0.5); h.sync()
    t0 = time.perf_counter(); info = h.map_delete_boxes(box); wall.append(time.perf_counter() - t0)
out["delete_one_slab"] = {"wall_ms_median": round(float(np.median(wall[2:])) * 1e3, 3), "n_removed": int(info.n_removed)}
wall = []
for r in range(10):
    h.sync(); t0 = time.perf_counter(); h.map_set_points(m0, 0.5); h.sync(); wall.append(time.perf_counter() - t0)
out["restage_from_host (fl_map_set_points)"] = {"wall_ms_median": round(float(np.median(wall[2:])) * 1e3, 3)}
h.close()
print(json.dumps(out))
