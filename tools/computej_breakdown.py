"""Where the time of fl_vio_compute_j goes: per-level passes, status bits (16 = an accept test needed the exact replay) and the host
wall time of the call. Run it with FL_LIB_PATH=build_ab/lib_noexact.so (built with FL_EXTRA_FLAGS=-DFL_AB_NO_EXACT) beside the
current library to see what the replays cost.  usage: python tools/computej_breakdown.py [patches]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi = importlib.import_module("fast-livo_amd.capi")
synth = importlib.import_module("fast-livo_amd.synth")

m = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
scene = synth.make_scene()
fr = synth.make_lio_frame(50000, scene=scene, point_seed=synth.SEED + 101)
vf = synth.make_vio_frame(m, fr, patch_seed=synth.SEED + 103)
h = capi.Handle(capi.config_from_frames(fr, vf, max_iterations=10, device=0))
h.vio_set_frame(vf.img)
h.vio_set_patches(vf.ref_patch, vf.pos, vf.search_level)
ts = []
for rep in range(60):
    xv = capi.state18_from_frame(fr)
    t0 = time.perf_counter()
    infos = h.vio_compute_j(xv, capi.state18_from_frame(fr))
    ts.append(time.perf_counter() - t0)
print("levels (iterations, accepted, status):", [(int(i.iterations), int(i.accepted), int(i.status)) for i in infos])
print("compute_j median %.1f us  min %.1f us" % (np.median(ts[5:]) * 1e6, np.min(ts[5:]) * 1e6))
