"""Per-pass time of the LIO pass kernel: one launch per pass vs the multi-pass kernel (run twice: FL_NO_MULTIPASS=1 / unset)."""
import os, sys, json, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlivo  # noqa
from fast_livo_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
C = int(sys.argv[2]) if len(sys.argv) > 2 else 11
scene = synth.make_scene()
fr = synth.make_lio_frame(n, scene=scene)
nbr, valid = synth.knn5(scene, fr.world_at(fr.R_prior, fr.p_prior))
h = capi.Handle(capi.config_from_frames(fr))
x0 = capi.state18_from_frame(fr)
h.lio_set_points(fr.body_xyz); h.lio_begin18(x0, x0); h.lio_set_neighbours(nbr, valid)
F = capi.FL_ITER_FORCE
for _ in range(20):
    h.lio_iterate18(C, F, want_info=False)
h.sync(); h.set_timing(True)
ks = []
for _ in range(200):
    h.lio_iterate18(C, F, want_info=False); h.sync(); ks.append(h.last_kernel_ms() * 1e3 / C)
t0 = time.perf_counter()
for _ in range(300):
    h.lio_iterate18(C, F, want_info=False)
h.sync()
wall = (time.perf_counter() - t0) / 300 / C * 1e6
print(json.dumps({"points": n, "passes_per_call": C, "multipass": os.environ.get("FL_NO_MULTIPASS") is None,
                  "us_per_pass_events": round(float(np.median(ks)), 2), "us_per_pass_back_to_back_wall": round(wall, 2)}))
