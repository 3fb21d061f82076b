"""Several independent 50 k-point LIO filters on one MI355X, one HIP stream each (multi-sensor serving): aggregate passes/s.
Each multi-pass launch needs its <= 256 workgroups resident (4 such grids fit the 1024 workgroup slots of the device)."""
import os, sys, json, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastlivo  # noqa
from fast_livo_amd import capi, synth
scene = synth.make_scene()
res = []
for K in (1, 2, 3, 4):
    hs, streams = [], []
    for k in range(K):
        fr = synth.make_lio_frame(50000, scene=scene, point_seed=1000 + k)
        nbr, valid = synth.knn5(scene, fr.world_at(fr.R_prior, fr.p_prior))
        h = capi.Handle(capi.config_from_frames(fr))
        s = torch.cuda.Stream(); h.set_stream(s.cuda_stream)
        x0 = capi.state18_from_frame(fr)
        h.lio_set_points(fr.body_xyz); h.lio_begin18(x0, x0); h.lio_set_neighbours(nbr, valid)
        hs.append(h); streams.append(s)
    F = capi.FL_ITER_FORCE
    for _ in range(10):
        for h in hs: h.lio_iterate18(10, F, want_info=False)
    torch.cuda.synchronize()
    R = 200
    t0 = time.perf_counter()
    for _ in range(R):
        for h in hs: h.lio_iterate18(10, F, want_info=False)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = [h.lio_iterate18(1, F).status for h in hs]
    res.append({"filters": K, "aggregate_passes_per_s": round(K * R * 10 / dt), "us_per_pass_per_filter": round(dt / (R * 10) * 1e6, 2), "status": st})
    for h in hs: h.close()
print(json.dumps(res))
