"""Run the device k-NN search kernel a few times (for rocprofv3 counter collection)."""
import argparse, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlivo  # noqa: F401
from fast_livo_amd import capi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=50000)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--cell", type=float, default=0.5)
a = ap.parse_args()
fr = synth.make_lio_frame(a.points)
h = capi.Handle(capi.config_from_frames(fr, max_iterations=10))
h.map_set_points(fr.scene.map_xyz, a.cell)
x = capi.state18_from_frame(fr); h.lio_set_points(fr.body_xyz); h.lio_begin18(x, x)
h.set_timing(True)
ks = []
for _ in range(a.reps):
    h.lio_search18(fr.n, want=False); ks.append(h.last_kernel_ms() * 1e3)
print("search_fit_us", ks)
