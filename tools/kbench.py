#!/usr/bin/env python3
"""Kernel micro-benchmark: per-launch time of each ESKF kernel (HIP events, back-to-back launches)."""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fastlivo  # noqa
from fast_livo_amd import capi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=50000)
ap.add_argument("--patches", type=int, default=2000)
ap.add_argument("--reps", type=int, default=500)
ap.add_argument("--level", type=int, default=0)
a = ap.parse_args()
scene = synth.make_scene()
fr = synth.make_lio_frame(min(a.points, 200000), scene=scene)
vf = synth.make_vio_frame(a.patches, fr)
nbr, valid = synth.knn5(scene, fr.world_at(fr.R_prior, fr.p_prior))
body = fr.body_xyz
if a.points > fr.n:
    reps = (a.points + fr.n - 1) // fr.n
    body = np.tile(body, (reps, 1))[:a.points]; nbr = np.tile(nbr, (reps, 1, 1))[:a.points]; valid = np.tile(valid, reps)[:a.points]
cfg = capi.config_from_frames(fr, vf)
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
hl = capi.Handle(cfg); hv = capi.Handle(cfg)
for h in (hl, hv): h.set_stream(s.cuda_stream)
x0 = capi.state18_from_frame(fr)
hl.lio_set_points(body); hl.lio_begin18(x0, x0); hl.lio_set_neighbours(nbr, valid)
hv.vio_set_frame(vf.img); hv.vio_set_patches(vf.ref_patch, vf.pos, vf.search_level); hv.vio_begin(x0, x0)
F = capi.FL_ITER_FORCE
t = torch.zeros(32, dtype=torch.float64, device="cuda")
def timeit(fn, reps=a.reps):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
res = {
  "lio_fused_us": timeit(lambda: hl.lio_iterate18(1, F, want_info=False)),
  "lio_accumulate_us": timeit(lambda: hl.lio_accumulate18(t.data_ptr(), F)),
  "lio_solve_us": timeit(lambda: hl.lio_solve18(t.data_ptr(), F)),
  "vio_fused_us": timeit(lambda: hv.vio_iterate(a.level, 1, F, want_info=False)),
  "vio_accumulate_us": timeit(lambda: hv.vio_accumulate(a.level, t.data_ptr())),
  "vio_solve_us": timeit(lambda: hv.vio_solve(t.data_ptr(), F)),
  "fit_planes_us": timeit(lambda: hl.lio_set_neighbours(nbr, valid), reps=20) if a.points <= 200000 else None,
  "points": a.points, "patches": a.patches,
}
# Mode-23 pass
hi = capi.Handle(cfg); hi.set_stream(s.cuda_stream)
hi.lio_set_points(body)
x23 = capi.state23_from_frame(fr)
hi.ikfom_begin(x23, fr.cov23.copy())
hi.lio_set_neighbours(nbr, valid)
res["ikfom_fused_us"] = timeit(lambda: hi.ikfom_iterate(1, F, want_info=False), reps=100)
t96 = torch.zeros(96, dtype=torch.float64, device="cuda")
res["ikfom_accumulate_us"] = timeit(lambda: hi.ikfom_accumulate(t96.data_ptr(), F), reps=100)
print(json.dumps(res))
