"""Host-side cost of the individual boundary calls of an all-device frame (wall clock, median of 200)."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlivo  # noqa
from fast_livo_amd import capi, synth
fr = synth.make_lio_frame(50000)
h = capi.Handle(capi.config_from_frames(fr, max_iterations=10))
h.map_set_points(fr.scene.map_xyz, 0.5)
x = capi.state18_from_frame(fr)
def med(fn, n=200):
    fn(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return round(float(np.median(ts)) * 1e6, 1)
res = {}
res["python_state18_from_frame_us"] = med(lambda: capi.state18_from_frame(fr))
res["set_points+sync_us"] = med(lambda: (h.lio_set_points(fr.body_xyz), h.sync()))
res["sync_only_us"] = med(lambda: h.sync())
h.lio_set_points(fr.body_xyz)
res["begin18+sync_us"] = med(lambda: (h.lio_begin18(x, x), h.sync()))
def frame():
    xg = capi.state18_from_frame(fr); return h.lio_frame18_dev(xg, fr.body_xyz)
res["frame18_dev_incl_python_us"] = med(frame)
xg = capi.state18_from_frame(fr)
import ctypes as C
keep = capi.state18_from_frame(fr)
def frame2():
    C.memmove(C.byref(xg), C.byref(keep), C.sizeof(xg)); return h.lio_frame18_dev(xg, fr.body_xyz)
res["frame18_dev_ctypes_only_us"] = med(frame2)
pin = h.host_alloc(fr.body_xyz.shape, np.float32); pin[:] = fr.body_xyz
res["set_points_pinned+sync_us"] = med(lambda: (h.lio_set_points(pin), h.sync()))
def frame3():
    C.memmove(C.byref(xg), C.byref(keep), C.sizeof(xg)); return h.lio_frame18_dev(xg, pin)
res["frame18_dev_pinned_us"] = med(frame3)
print(json.dumps(res))
