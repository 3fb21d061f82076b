"""fl_ikfom_update_iterated_dev (the whole Mode-23 update on the device) repeated; host wall time. Run under tools/ktrace.sh for the time line."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlivo  # noqa
from fast_livo_amd import capi, synth
scene = synth.make_scene()
fr = synth.in_voxel_order(synth.make_lio_frame(50000, scene=scene), 0.15)
h = capi.Handle(capi.config_from_frames(fr, max_iterations=10))
h.map_set_points(scene.map_xyz, 0.5)
if os.environ.get("FL_MAILBOX") is not None: h.set_option(capi.FL_OPT_MAILBOX, int(os.environ["FL_MAILBOX"]))
if os.environ.get("FL_PULL") is not None: h.set_option(capi.FL_OPT_SCAN_PULL, int(os.environ["FL_PULL"]))
scan = h.host_alloc(fr.body_xyz.shape, np.float32); scan[...] = fr.body_xyz
ts = []
for rep in range(40):
    x23 = capi.state23_from_frame(fr); P = fr.cov23.copy()
    t0 = time.perf_counter()
    info = h.ikfom_update_iterated_dev(x23, P, scan, 0.001)
    ts.append(time.perf_counter() - t0)
print("passes", int(info.iterations), "status", int(info.status), "update median %.1f us min %.1f us" % (np.median(ts[5:]) * 1e6, np.min(ts[5:]) * 1e6))
