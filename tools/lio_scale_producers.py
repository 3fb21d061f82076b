"""The at-scale LIO pass (one launch per pass, forced) by number of producer workgroups (FL_OPT_MAX_PRODUCERS): us per pass and algorithmic
GB/s at 8 M and 32 M points. (The size policy -- fl_lio_producers, lio_kernels.h -- takes 1023 beyond 2 M points; 3 workgroups per CU are
resident, i.e. 768.)   python tools/lio_scale_producers.py [counts ...]"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlivo  # noqa
import torch
from fast_livo_amd import capi, synth
counts = [int(a) for a in sys.argv[1:]] or [1023, 767, 511]
scene = synth.make_scene()
fr0 = synth.make_lio_frame(200000, scene=scene)
vf = synth.make_vio_frame(16, fr0)
cfg = capi.config_from_frames(fr0, vf, max_iterations=1)
x0 = capi.state18_from_frame(fr0)
w = fr0.world_at(fr0.R_prior, fr0.p_prior)
nbr0, valid0 = synth.knn5(scene, w)
for n in (8000000, 32000000):
    reps = (n + fr0.n - 1) // fr0.n
    body = np.tile(fr0.body_xyz, (reps, 1))[:n]; nbr = np.tile(nbr0, (reps, 1, 1))[:n]; valid = np.tile(valid0, reps)[:n]
    h = capi.Handle(cfg)
    h.set_stream(torch.cuda.current_stream().cuda_stream)
    h.lio_set_points(body); h.lio_begin18(x0, x0); h.lio_set_neighbours(nbr, valid)
    del nbr, body
    row = {}
    for rnd in range(2):
        for c in counts:
            h.set_option(capi.FL_OPT_MAX_PRODUCERS, c)
            for _ in range(4): h.lio_iterate18(1, capi.FL_ITER_FORCE, want_info=False)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            K = 20
            e0.record()
            for _ in range(K): h.lio_iterate18(1, capi.FL_ITER_FORCE, want_info=False)
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / K
            row.setdefault(str(c), []).append(round(us, 1))
    li = h.lio_iterate18(0, capi.FL_ITER_FORCE)
    print(json.dumps({"points": n, "pass_us_by_producers": row, "effective": int(li.effct_feat_num), "status": int(li.status)}), flush=True)
    h.close()
