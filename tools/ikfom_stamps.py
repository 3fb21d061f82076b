"""Phase stamps (ns) of the solver workgroup of ONE single-pass Mode-23 launch (library built with -DFL_IK_STAMPS)."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlivo  # noqa
from fast_livo_amd import capi, synth
scene = synth.make_scene()
fr = synth.make_lio_frame(50000, scene=scene)
nbr, valid = synth.knn5(scene, fr.world_at(fr.R_prior, fr.p_prior))
h = capi.Handle(capi.config_from_frames(fr), debug=True)
x23 = capi.state23_from_frame(fr)
h.lio_set_points(fr.body_xyz); h.ikfom_begin(x23, fr.cov23.copy()); h.lio_set_neighbours(nbr, valid)
names = {32: "solver start", 33: "staged (x, Pprop)", 44: "pre: dx, J (three lanes) done", 45: "pre: P done", 34: "pre done (dx, J, P, A12)", 35: "gather done", 36: "S, SA, M, rhs, y0", 37: "LDL^T + dx_", 38: "boxplus + judge",
         40: "post returned", 41: "producer0 loop start", 42: "producer0 loop end", 43: "producer0 published"}
for _ in range(5): h.ikfom_iterate(1, capi.FL_ITER_FORCE, want_info=False)
for _ in range(3):
    h.ikfom_iterate(1, capi.FL_ITER_FORCE, want_info=False); h.sync()
    st = np.array(h.debug_stamps(), dtype=np.int64)
    print(json.dumps({names[k]: int(st[k] - st[32]) * 10 for k in sorted(names, key=lambda k: st[k])}))
    print("  gather sweeps (ns after solver start, slots still missing):", [(int(st[48 + i] - st[32]) * 10, int(st[56 + i])) for i in range(6) if st[48 + i] > st[32]])
