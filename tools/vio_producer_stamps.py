"""Phase stamps (100 MHz wall clock, ns) of producer workgroup 0 of ONE single-pass VIO launch (2000 patches, level 0)."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlivo  # noqa
from fast_livo_amd import capi, synth
lio = synth.make_lio_frame(2000)
vf = synth.make_vio_frame(2000, lio)
h = capi.Handle(capi.config_from_frames(lio, vf), debug=True)
x0 = capi.state18_from_frame(lio)
h.vio_set_frame(vf.img); h.vio_set_patches(vf.ref_patch, vf.pos, vf.search_level); h.vio_begin(x0, x0)
F = capi.FL_ITER_FORCE
names = {0: "loop start", 40: "geometry", 41: "taps + ref arrived", 42: "patch M", 43: "pixel math", 44: "half-wave sums", 45: "6x6 accumulated",
         1: "loop end", 3: "record published", 2: "patch_error chain done"}
for rep in range(4):
    for _ in range(5): h.vio_iterate(0, 1, F, want_info=False)
    h.vio_iterate(0, 1, F | capi.FL_ITER_STAMP, want_info=False); h.sync()
    st = np.array(h.debug_stamps(), dtype=np.int64)
    print(json.dumps({names[k]: int(st[k] - st[0]) * 10 for k in sorted(names, key=lambda k: st[k])}))
