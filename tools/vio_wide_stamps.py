"""Where a sweep of the one-patch-per-lane VIO producers goes (100 MHz wall clock, ns): phases of every sweep of producer wavefront 0 in
ONE forced pass over 1 M patches. Needs a build with -DFL_INSTRUMENT -DFL_WIDE_STAMPS:
    FL_OUT=build_ab/lib_ws.so FL_EXTRA_FLAGS="-DFL_INSTRUMENT -DFL_WIDE_STAMPS" bash fast-livo_amd/build.sh
    FL_LIB_PATH=build_ab/lib_ws.so python tools/vio_wide_stamps.py [patches]"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlivo  # noqa
from fast_livo_amd import capi, synth
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
fr = synth.make_lio_frame(2000)
vf = synth.make_vio_frame(2000, fr)
reps = (m + vf.m - 1) // vf.m
ref = np.tile(vf.ref_patch, (reps, 1, 1))[:m]; pos = np.tile(vf.pos, (reps, 1))[:m]; sl = np.tile(vf.search_level, reps)[:m]
h = capi.Handle(capi.config_from_frames(fr, vf, max_iterations=1), debug=True)
x0 = capi.state18_from_frame(fr)
h.vio_set_frame(vf.img); h.vio_set_patches(ref, pos, sl); h.vio_begin(x0, x0)
names = ["requests issued", "everything arrived", "pixel rows done", "outputs reduced"]
# finer: slot 5 = reference-patch requests issued, slot 6 = positions arrived + projection done (the rest of "requests issued" = tap rows requested)
for rep in range(3):
    for _ in range(3): h.vio_iterate(0, 1, capi.FL_ITER_FORCE, want_info=False)
    h.sync()
    wall = h.debug_wall()
    ends = (wall[1024:1024 + 512] - wall[0]) * 10
    ends = ends[ends > 0]
    print(json.dumps({"producer workgroups' ends after workgroup 0's first sweep began, ns": {"min": int(ends.min()), "p10": int(np.percentile(ends, 10)),
                      "median": int(np.median(ends)), "p90": int(np.percentile(ends, 90)), "max": int(ends.max())},
                      "solver": {"entered": int(wall[2040] - wall[0]) * 10, "gather done": int(wall[2041] - wall[0]) * 10, "returned": int(wall[2042] - wall[0]) * 10}}))
    e = (wall[1024:1024 + 512] - wall[0]) * 10
    print(json.dumps({"mean end by blockIdx % 8 (XCD), us": [round(float(e[k::8].mean()) / 1e3, 1) for k in range(8)],
                      "mean end of blocks 0..255 / 256..511, us": [round(float(e[:256].mean()) / 1e3, 1), round(float(e[256:].mean()) / 1e3, 1)],
                      "slowest 12 blocks": [int(b) for b in np.argsort(e)[-12:]], "fastest 12": [int(b) for b in np.argsort(e)[:12]]}))
    w = wall[:2000].reshape(-1, 8)
    rows = [r for r in w if r[0] > 0 and r[4] > r[0]]
    t0 = rows[0][0]
    out = []
    for r in rows:
        out.append({"top_ns": int(r[0] - t0) * 10, "ref requests": int(r[5] - r[0]) * 10, "positions + projection": int(r[6] - r[5]) * 10,
                    "tap requests": int(r[1] - r[6]) * 10, **{names[j]: int(r[j + 1] - r[j]) * 10 for j in range(1, 4)}})
    print(json.dumps(out))
