import os, sys, json
import numpy as np
sys.path.insert(0, "/root/repo")
import fastlivo
from fast_livo_amd import capi, synth
scene = synth.make_scene()
fr = synth.make_lio_frame(50000, scene=scene)
nbr, valid = synth.knn5(scene, fr.world_at(fr.R_prior, fr.p_prior))
h = capi.Handle(capi.config_from_frames(fr))
x23 = capi.state23_from_frame(fr)
h.lio_set_points(fr.body_xyz); h.ikfom_begin(x23, fr.cov23.copy()); h.lio_set_neighbours(nbr, valid)
for _ in range(5): h.ikfom_iterate(1, capi.FL_ITER_FORCE, want_info=False)
for _ in range(3):
    h.ikfom_iterate(1, capi.FL_ITER_FORCE, want_info=False); h.sync()
    st = np.array(h.debug_stamps(), dtype=np.int64)
    print([int(st[k] - st[32]) * 10 for k in range(32, 41)])
