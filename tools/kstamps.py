#!/usr/bin/env python3
"""Phase breakdown of one fused pass from in-kernel shader-clock stamps (FL_ITER_STAMP).
slots: 0 block0 start-of-loop, 1 block0 end-of-loop, 2 block0 after publish+ticket,
       8 last block after ticket, 9 after final reduce, 10 after solve."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fastlivo  # noqa
from fast_livo_amd import capi, synth
scene = synth.make_scene()
fr = synth.make_lio_frame(50000, scene=scene)
vf = synth.make_vio_frame(2000, fr)
nbr, valid = synth.knn5(scene, fr.world_at(fr.R_prior, fr.p_prior))
cfg = capi.config_from_frames(fr, vf)
hl = capi.Handle(cfg, debug=True); hv = capi.Handle(cfg, debug=True)
x0 = capi.state18_from_frame(fr)
hl.lio_set_points(fr.body_xyz); hl.lio_begin18(x0, x0); hl.lio_set_neighbours(nbr, valid)
hv.vio_set_frame(vf.img); hv.vio_set_patches(vf.ref_patch, vf.pos, vf.search_level); hv.vio_begin(x0, x0)
F = capi.FL_ITER_FORCE | capi.FL_ITER_STAMP
def show(name, st):
    print(name, json.dumps({"producer0_loop": int(st[1]-st[0]), "producer0_reduce_publish": int(st[2]-st[1]),
                            "solver_prefetch": int(st[9]-st[8]), "solver_gather_wait": int(st[10]-st[9]), "solver_solve": int(st[11]-st[10]),
                            "solver_total": int(st[11]-st[8]), "spins": int(st[39]),
                            "sweep_ends_rel_prefetch_end": [int(st[40+i]-st[9]) for i in range(min(int(st[39])+1, 8))]}))
for rep in range(3):
    for _ in range(5): hl.lio_iterate18(1, capi.FL_ITER_FORCE, want_info=False)
    hl.lio_iterate18(1, F, want_info=False); show("lio", hl.debug_stamps())
    for _ in range(5): hv.vio_iterate(0, 1, capi.FL_ITER_FORCE, want_info=False)
    hv.vio_iterate(0, 1, F, want_info=False); show("vio", hv.debug_stamps())
