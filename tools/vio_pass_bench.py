"""us per forced VIO pass (A/B of library builds via FL_LIB_PATH): `fresh` = the 10 passes right after fl_vio_begin (every pass is
accepted and runs the full gain solve: what a 20-step bench sample measures), `steady` = forced passes long after convergence (many
are rejected -- error went up -- and skip the solve, as in the reference)."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlivo  # noqa
from fast_livo_amd import capi, synth
lio = synth.make_lio_frame(2000)
vf = synth.make_vio_frame(2000, lio)
h = capi.Handle(capi.config_from_frames(lio, vf))
x0 = capi.state18_from_frame(lio)
h.vio_set_frame(vf.img); h.vio_set_patches(vf.ref_patch, vf.pos, vf.search_level); h.vio_begin(x0, x0)
F = capi.FL_ITER_FORCE
LV = int(os.environ.get("FL_LEVEL", "0"))
h.set_timing(True)
fresh, acc = [], 0
for rep in range(60):
    h.vio_begin(x0, x0); h.sync()
    i0 = h.vio_iterate(LV, 0, F)
    h.vio_iterate(LV, 10, F, want_info=False); h.sync()
    fresh.append(h.last_kernel_ms() * 100)
    acc = h.vio_iterate(LV, 0, F).accepted - i0.accepted
for _ in range(20): h.vio_iterate(LV, 10, F, want_info=False)
steady = []
for _ in range(100):
    h.vio_iterate(LV, 10, F, want_info=False); h.sync(); steady.append(h.last_kernel_ms() * 100)
print(json.dumps({"vio_pass_us_fresh": round(float(np.median(fresh[10:])), 2), "accepted_of_10_fresh": int(acc), "vio_pass_us_steady": round(float(np.median(steady)), 2)}))
