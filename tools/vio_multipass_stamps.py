"""Phase stamps (100 MHz wall clock) of passes 5/6 inside one multi-pass VIO launch (2000 patches, level 0)."""
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
for _ in range(5):
    h.vio_iterate(0, 10, F, want_info=False)
names = {20: "prod0 p5 wait_start", 21: "prod0 p5 pose_seen", 23: "prod0 p5 published", 16: "solver p5 gather_start", 17: "solver p5 gather_done",
         30: "solve: start", 31: "solve: C,rhs formed", 32: "solve: eliminated", 33: "solve: delta", 34: "solve: state", 36: "solve: stores issued",
         35: "solve: returned", 18: "solver p5 solve_done+sync", 24: "prod0 p6 wait_start", 25: "prod0 p6 pose_seen", 27: "prod0 p6 published"}
for rep in range(4):
    h.vio_iterate(0, 10, F | capi.FL_ITER_STAMP, want_info=False); h.sync()
    st = np.array(h.debug_stamps(), dtype=np.int64)
    t0 = st[20]
    print(json.dumps({names[k]: int(st[k] - t0) * 10 for k in sorted(names, key=lambda k: st[k]) if st[k] > 0}))
    w = np.array(h.debug_wall(), dtype=np.int64)[:125]
    pub = (w - t0) * 10
    print(json.dumps({"pass 5: record published, ns after prod0 wait_start": {"prod0": int(st[3] - t0) * 10, "min": int(pub.min()), "p50": int(np.median(pub)), "p90": int(np.percentile(pub, 90)),
                      "max": int(pub.max()), "argmax_block": int(pub.argmax())}, "gather_done": int(st[17] - t0) * 10}))
