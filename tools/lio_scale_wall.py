"""Start / end of every producer workgroup of ONE forced at-scale LIO pass (100 MHz wall clock, debug library): is the launch as long as
its slowest producers, and which are they?   python tools/lio_scale_wall.py [points]"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlivo  # noqa
from fast_livo_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000000
scene = synth.make_scene()
fr0 = synth.make_lio_frame(200000, scene=scene)
vf = synth.make_vio_frame(16, fr0)
cfg = capi.config_from_frames(fr0, vf, max_iterations=1)
x0 = capi.state18_from_frame(fr0)
nbr0, valid0 = synth.knn5(scene, fr0.world_at(fr0.R_prior, fr0.p_prior))
reps = (n + fr0.n - 1) // fr0.n
body = np.tile(fr0.body_xyz, (reps, 1))[:n]; nbr = np.tile(nbr0, (reps, 1, 1))[:n]; valid = np.tile(valid0, reps)[:n]
h = capi.Handle(cfg, debug=True)
h.lio_set_points(body); h.lio_begin18(x0, x0); h.lio_set_neighbours(nbr, valid)
del nbr, body
for rep in range(3):
    for _ in range(3): h.lio_iterate18(1, capi.FL_ITER_FORCE, want_info=False)
    h.lio_iterate18(1, capi.FL_ITER_FORCE | capi.FL_ITER_STAMP, want_info=False); h.sync()
    w = h.debug_wall()
    nb = 1023
    s = (w[:nb] - w[:nb].min()) / 100.0; e = (w[1024:1024 + nb] - w[:nb].min()) / 100.0
    grp = [slice(0, 256), slice(256, 512), slice(512, 768), slice(768, 1023)]
    print(json.dumps({"points": n, "starts us (mean by blocks 0-255 / 256-511 / 512-767 / 768-1022)": [round(float(s[g].mean()), 1) for g in grp],
                      "ends us": [round(float(e[g].mean()), 1) for g in grp], "end min / median / max": [round(float(x), 1) for x in (e.min(), np.median(e), e.max())]}))
