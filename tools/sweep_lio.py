"""Kernel-only size sweep of the forced LIO pass (one launch per pass): us per pass and algorithmic GB/s (29 B per point)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import fastlivo  # noqa: E402,F401
from fast_livo_amd import capi, synth  # noqa: E402
sizes = [int(a) for a in sys.argv[1:]] or [50000, 200000, 1000000, 4000000, 8000000]
scene = synth.make_scene()
fr = synth.make_lio_frame(1000, scene=scene)
cfg = capi.config_from_frames(fr)
x0 = capi.state18_from_frame(fr)
for n in sizes:
    us, gbs = bench.lio_pass_at(capi, synth, scene, cfg, x0, n)
    print(json.dumps({"points": n, "lio_pass_us": round(us, 2), "algorithmic_GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / 8000.0, 4)}))
