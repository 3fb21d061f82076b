"""Phase stamps (ns) of pass 3 of a multi-pass Mode-23 launch (library built with -DFL_INSTRUMENT -DFL_IK_STAMPS): solver workgroup and
producer workgroup 0, relative to the solver's loop top of that pass."""
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
names = {20: "solver: loop top (pass 3)", 44: "solver: pre dx/J done", 45: "solver: pre P done", 21: "solver: pre done", 22: "solver: gather done", 36: "solver: S, SA, M", 37: "solver: LDL + dx_", 38: "solver: boxplus + judge",
         23: "solver: post returned (state broadcast)", 24: "producer0: waiting for pass 3's state", 25: "producer0: got it", 41: "producer0: loop start", 42: "producer0: loop end", 43: "producer0: published",
         26: "producer0: produce returned", 28: "producer0: waiting for pass 4's state", 29: "producer0: got pass 4's state", 30: "producer0: pass 4 produce returned"}
F = capi.FL_ITER_FORCE
fresh = len(sys.argv) > 1 and sys.argv[1] == "fresh"      # "fresh": begin before every launch -> pass 3 is the third pass after a begin (not finishing)
for _ in range(3): h.ikfom_iterate(6, F, want_info=False)
for _ in range(3):
    if fresh:
        h.ikfom_begin(x23, fr.cov23.copy()); h.lio_set_neighbours(nbr, valid)
    h.ikfom_iterate(4 if fresh else 6, F, want_info=False); h.sync()
    st = np.array(h.debug_stamps(), dtype=np.int64)
    t0 = st[20]
    print(json.dumps({names[k]: int(st[k] - t0) * 10 for k in sorted(names, key=lambda k: st[k])}))
