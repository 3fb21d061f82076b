"""us per Mode-23 pass: `first3` = the metric of bench.py (the first 3 passes after a begin in one multi-pass launch, launch included),
`steady10` = forced 10-pass launches long after convergence (every pass also runs the final covariance block). FL_IK_PRODUCERS caps the grid."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlivo  # noqa
from fast_livo_amd import capi, synth
scene = synth.make_scene()
fr = synth.make_lio_frame(50000, scene=scene)
nbr, valid = synth.knn5(scene, fr.world_at(fr.R_prior, fr.p_prior))
h = capi.Handle(capi.config_from_frames(fr))
x23 = capi.state23_from_frame(fr)
h.lio_set_points(fr.body_xyz); h.ikfom_begin(x23, fr.cov23.copy()); h.lio_set_neighbours(nbr, valid)
F = capi.FL_ITER_FORCE
h.set_timing(True)
res = {}
for C in (3, 6):
    ts = []
    for rep in range(45):
        h.ikfom_begin(x23, fr.cov23.copy()); h.lio_set_neighbours(nbr, valid); h.sync()
        h.ikfom_iterate(C, F, want_info=False); h.sync()
        ts.append(h.last_kernel_ms() * 1e3 / C)
    res[f"first{C}_us"] = round(float(np.median(ts[5:])), 2)
res["marginal_pass_us"] = round((res["first6_us"] * 6 - res["first3_us"] * 3) / 3, 2)
print(json.dumps(res))
