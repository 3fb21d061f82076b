import os, sys, json, time
import numpy as np
sys.path.insert(0, "/root/repo")
import fastlivo
from fast_livo_amd import capi, synth
lio = synth.make_lio_frame(2000)
vf = synth.make_vio_frame(2000, lio)
h = capi.Handle(capi.config_from_frames(lio, vf))
x0 = capi.state18_from_frame(lio)
h.vio_set_frame(vf.img); h.vio_set_patches(vf.ref_patch, vf.pos, vf.search_level); h.vio_begin(x0, x0)
F = capi.FL_ITER_FORCE
LV = int(os.environ.get("FL_LEVEL", "0"))
for _ in range(20): h.vio_iterate(LV, 10, F, want_info=False)
h.sync(); h.set_timing(True); ks=[]
for _ in range(100):
    h.vio_iterate(LV, 10, F, want_info=False); h.sync(); ks.append(h.last_kernel_ms()*100)
print(round(float(np.median(ks)),2), "us per VIO pass")
