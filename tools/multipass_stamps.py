"""Phase stamps (100 MHz wall clock) of passes 5/6 inside one multi-pass LIO launch."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlivo  # noqa
from fast_livo_amd import capi, synth
scene = synth.make_scene()
fr = synth.make_lio_frame(50000, scene=scene)
nbr, valid = synth.knn5(scene, fr.world_at(fr.R_prior, fr.p_prior))
h = capi.Handle(capi.config_from_frames(fr))
x0 = capi.state18_from_frame(fr)
h.lio_set_points(fr.body_xyz); h.lio_begin18(x0, x0); h.lio_set_neighbours(nbr, valid)
F = capi.FL_ITER_FORCE
for _ in range(5):
    h.lio_iterate18(11, F, want_info=False)
for rep in range(4):
    h.lio_iterate18(11, F | capi.FL_ITER_STAMP, want_info=False); h.sync()
    st = np.array(h.debug_stamps(), dtype=np.int64)
    t0 = st[20]
    names = {20: "prod0 p5 wait_start", 21: "prod0 p5 pose_seen", 22: "prod0 p5 compute_done", 23: "prod0 p5 published", 16: "solver p5 gather_start",
             17: "solver p5 gather_done", 18: "solver p5 solve_done+sync", 24: "prod0 p6 wait_start", 25: "prod0 p6 pose_seen", 26: "prod0 p6 compute_done",
             27: "prod0 p6 published", 30: "solve: start", 31: "solve: C,rhs formed", 32: "solve: eliminated", 33: "solve: delta", 34: "solve: state", 35: "solve: returned", 36: "solve: stores issued", 37: "solve: judged"}
    print(json.dumps({names[k]: int(st[k] - t0) * 10 for k in sorted(names, key=lambda k: st[k])}))
