"""Mode-23 (IKFoM) pass timing: fused pass (accumulate + cooperative solver) and accumulate-only, 50 k points."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastlivo  # noqa
from fast_livo_amd import capi, synth
scene = synth.make_scene()
fr = synth.make_lio_frame(50000, scene=scene)
nbr, valid = synth.knn5(scene, fr.world_at(fr.R_prior, fr.p_prior))
h = capi.Handle(capi.config_from_frames(fr))
x23 = capi.state23_from_frame(fr)
h.lio_set_points(fr.body_xyz); h.ikfom_begin(x23, fr.cov23.copy()); h.lio_set_neighbours(nbr, valid)
F = capi.FL_ITER_FORCE
def timeit(fn, reps=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    h.set_stream(torch.cuda.current_stream().cuda_stream)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
res = {"ikfom_fused_pass_us": round(timeit(lambda: h.ikfom_iterate(1, F, want_info=False)), 2)}
t96 = torch.zeros(96, dtype=torch.float64, device="cuda")
res["ikfom_accumulate_us"] = round(timeit(lambda: h.ikfom_accumulate(t96.data_ptr(), F)), 2)
print(json.dumps(res))
