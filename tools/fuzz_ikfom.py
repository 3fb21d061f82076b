"""Randomised cross-check of the Mode-23 (IKFoM) update (run on the GPU box): all-device update and host-kNN update vs the oracle."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlivo  # noqa
from fast_livo_amd import capi, synth
from oracle import oracle as orc
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
T = int(sys.argv[2]) if len(sys.argv) > 2 else 20
scene = synth.make_scene()
bad = 0
for trial in range(T):
    n = int(rng.choice([7, 64, 257, 1000, 5000, 20000, 50000, 66000]))
    max_iter = int(rng.integers(1, 8))
    fr = synth.make_lio_frame(n, scene=scene, point_seed=int(rng.integers(1 << 30)))
    def knn(w):
        nb, _, va, _ = orc.knn5_bruteforce(scene.map_xyz, w)
        return nb, va
    xo = orc.state23_from_frame(fr, synth.quat_from_R); Po = fr.cov23.copy()
    ro = orc.ikfom_update(xo, Po, fr.body_xyz, 0.001, max_iter, knn)
    h = capi.Handle(capi.config_from_frames(fr, max_iterations=max_iter))
    h.map_set_points(scene.map_xyz, 0.5)
    xg = capi.state23_from_frame(fr); Pg = fr.cov23.copy()
    ig = h.ikfom_update_iterated_dev(xg, Pg, fr.body_xyz, 0.001)
    xh = capi.state23_from_frame(fr); Ph = fr.cov23.copy()
    ih = h.ikfom_update_iterated(xh, Ph, fr.body_xyz, 0.001, knn)
    e1 = np.abs(xg.vec() - xo.vec()).max(); e2 = np.abs(Pg - Po).max(); e3 = np.abs(xh.vec() - xg.vec()).max(); e4 = np.abs(Ph - Pg).max()
    ok = e1 <= 1e-9 and e2 <= 1e-10 and e3 <= 1e-12 and e4 <= 1e-13 and ig.iterations == ro["out"].iterations and ig.effct_feat_num == ro["out"].effct_feat_num
    if not ok:
        bad += 1
        print("MISMATCH", dict(n=n, max_iter=max_iter, e1=e1, e2=e2, e3=e3, e4=e4, it=(ig.iterations, ih.iterations, ro["out"].iterations),
                               neff=(ig.effct_feat_num, ro["out"].effct_feat_num)))
    h.close()
print(json.dumps({"trials": T, "mismatches": bad}))
