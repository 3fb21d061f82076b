"""Where the time of fl_lio_frame18_dev goes (host wall time of the call; run under tools/ktrace.sh for the kernel time line).
usage: [FL_PAGEABLE=1] python tools/lioframe_breakdown.py [points]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi = importlib.import_module("fast-livo_amd.capi")
synth = importlib.import_module("fast-livo_amd.synth")

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
scene = synth.make_scene()
fr = synth.make_lio_frame(n, scene=scene, point_seed=synth.SEED + 101)
vf = synth.make_vio_frame(2000, fr, patch_seed=synth.SEED + 103)
h = capi.Handle(capi.config_from_frames(fr, vf, max_iterations=10, device=0))
h.map_set_points(scene.map_xyz, float(os.environ.get("FL_CELL", "0.5")))
if os.environ.get("FL_INCR") is not None:
    h.set_option(capi.FL_OPT_INCR_SEARCH, int(os.environ["FL_INCR"]))
if os.environ.get("FL_ORDER") == "voxel":
    fr = synth.in_voxel_order(fr, 0.15)
scan = fr.body_xyz
if os.environ.get("FL_PAGEABLE", "0") != "1":      # the scan in page-locked memory of the library (the frame's first search kernel fetches it)
    scan = h.host_alloc(fr.body_xyz.shape, np.float32)
    scan[:] = fr.body_xyz
ts = []
for rep in range(60):
    x = capi.state18_from_frame(fr)
    t0 = time.perf_counter()
    info = h.lio_frame18_dev(x, scan)
    ts.append(time.perf_counter() - t0)
print("iterations", int(info.iterations), "status", int(info.status), "effective", int(info.effct_feat_num))
print("lio_frame18_dev median %.1f us  min %.1f us" % (np.median(ts[5:]) * 1e6, np.min(ts[5:]) * 1e6))
