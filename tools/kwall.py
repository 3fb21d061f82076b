#!/usr/bin/env python3
"""Cross-workgroup timeline of one LIO pass from the 100 MHz wall clock (debug stamps)."""
import os, sys, json, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import fastlivo  # noqa
from fast_livo_amd import capi, synth
scene = synth.make_scene()
fr = synth.make_lio_frame(50000, scene=scene)
nbr, valid = synth.knn5(scene, fr.world_at(fr.R_prior, fr.p_prior))
cfg = capi.config_from_frames(fr)
hl = capi.Handle(cfg, debug=True)
L = capi.lib(); L.fl_debug_get_wall.restype = C.c_int32; L.fl_debug_get_wall.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
x0 = capi.state18_from_frame(fr)
hl.lio_set_points(fr.body_xyz); hl.lio_begin18(x0, x0); hl.lio_set_neighbours(nbr, valid)
F = capi.FL_ITER_FORCE | capi.FL_ITER_STAMP
vf = synth.make_vio_frame(2000, fr)
hv = capi.Handle(capi.config_from_frames(fr, vf), debug=True)
hv.vio_set_frame(vf.img); hv.vio_set_patches(vf.ref_patch, vf.pos, vf.search_level); hv.vio_begin(x0, x0)
for rep in range(6):
    vio = rep >= 3
    if not vio:
        for _ in range(5): hl.lio_iterate18(1, capi.FL_ITER_FORCE, want_info=False)
        hl.lio_iterate18(1, F, want_info=False)
        st = hl.debug_stamps()
        nb = 197
    else:
        for _ in range(5): hv.vio_iterate(0, 1, capi.FL_ITER_FORCE, want_info=False)
        hv.vio_iterate(0, 1, F, want_info=False)
        st = hv.debug_stamps()
        nb = 251
    w = (C.c_longlong * 2048)(); L.fl_debug_get_wall(hl.h, w); w = np.array(w[:], dtype=np.int64)
    starts = w[:nb]; ends = w[1024:1024 + nb - 1]
    t0 = starts.min()
    print("VIO" if vio else "LIO", json.dumps({"unit": "10ns ticks from first workgroup start",
        "producer_start_min_med_max": [int(starts[:nb-1].min()-t0), int(np.median(starts[:nb-1])-t0), int(starts[:nb-1].max()-t0)],
        "producer_end_min_med_max": [int(ends.min()-t0), int(np.median(ends)-t0), int(ends.max()-t0)],
        "solver_start": int(starts[nb-1]-t0), "solver_after_prologue": int(st[8]-t0), "solver_prefetch_end": int(st[9]-t0),
        "gather_done": int(st[10]-t0), "solve_done": int(st[11]-t0),
        "blk0": {"start": int(w[0]-t0), "loop_start": int(st[0]-t0), "iter_start": int(st[3]-t0), "geom_done": int(st[4]-t0), "M_done_taps_issued": int(st[5]-t0),
                 "grad_done(taps arrived)": int(st[6]-t0), "wave_sum_done": int(st[7]-t0), "loop_end": int(st[1]-t0), "published": int(st[2]-t0)} if vio else None}))
