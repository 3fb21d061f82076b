"""Soak: the same all-device LIO frame and VIO ComputeJ repeated many times -- results must be bit-identical every time and no
hand-off may time out (status bit 8)."""
import os, sys, json, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlivo  # noqa
from fast_livo_amd import capi, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
fr = synth.make_lio_frame(50000)
h = capi.Handle(capi.config_from_frames(fr, max_iterations=10))
h.map_set_points(fr.scene.map_xyz, 0.5)
vf = synth.make_vio_frame(2000, fr)
hv = capi.Handle(capi.config_from_frames(fr, vf, max_iterations=10))
hv.vio_set_frame(vf.img); hv.vio_set_patches(vf.ref_patch, vf.pos, vf.search_level)
first = None; firstv = None; bad = 0; badstatus = 0
t0 = time.perf_counter()
for i in range(N):
    x = capi.state18_from_frame(fr)
    info = h.lio_frame18_dev(x, fr.body_xyz)
    b = bytes(x)
    if first is None: first = b
    bad += (b != first); badstatus += int((info.status & ~16) != 0)
    xv = capi.state18_from_frame(fr); xp = capi.state18_from_frame(fr)
    infos = hv.vio_compute_j(xv, xp)
    bv = bytes(xv)
    if firstv is None: firstv = bv
    bad += (bv != firstv); badstatus += sum(int((i_.status & ~16) != 0) for i_ in infos)      # bit 16 = the exact accept test ran: not an error
print(json.dumps({"frames": N, "different_results": bad, "status_other_than_fragile": badstatus, "seconds": round(time.perf_counter() - t0, 1)}))
