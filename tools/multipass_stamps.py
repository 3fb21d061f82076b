"""Phase stamps (100 MHz wall clock) of passes 5/6 inside one multi-pass LIO launch."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlivo  # noqa
from fast_livo_amd import capi, synth
scene = synth.make_scene()
fr = synth.make_lio_frame(50000, scene=scene)
nbr, valid = synth.knn5(scene, fr.world_at(fr.R_prior, fr.p_prior))
h = capi.Handle(capi.config_from_frames(fr), debug=True)
x0 = capi.state18_from_frame(fr)
h.lio_set_points(fr.body_xyz); h.lio_begin18(x0, x0); h.lio_set_neighbours(nbr, valid)
F = capi.FL_ITER_FORCE
for _ in range(5):
    h.lio_iterate18(11, F, want_info=False)
for rep in range(4):
    h.lio_iterate18(11, F | capi.FL_ITER_STAMP, want_info=False); h.sync()
    st = np.array(h.debug_stamps(), dtype=np.int64)
    t0 = st[20]
    names = {20: "prod0 p5 wait_start", 21: "prod0 p5 pose_seen", 22: "prod0 p5 compute_done", 23: "prod0 p5 published", 16: "solver p5 gather_start",
             17: "solver p5 gather_done", 18: "solver p5 solve_done+sync", 24: "prod0 p6 wait_start", 25: "prod0 p6 pose_seen", 26: "prod0 p6 compute_done",
             27: "prod0 p6 published", 30: "solve: start", 31: "solve: C,rhs formed", 32: "solve: eliminated", 33: "solve: delta", 34: "solve: state", 35: "solve: returned", 36: "solve: stores issued", 37: "solve: judged"}
    print(json.dumps({names[k]: int(st[k] - t0) * 10 for k in sorted(names, key=lambda k: st[k])}))

# spread of the producers' publish times of pass 5 (g_fl_wall), relative to producer 0's "published" stamp, and the solver's gather_done
w = np.array(h.debug_wall(), dtype=np.int64) if hasattr(h, "debug_wall") else None
if w is not None:
    nprod = (50000 + 255) // 256
    pub = (w[:nprod] - st[23]) * 10
    print(json.dumps({"producers": int(nprod), "publish_ns_rel_prod0": {"min": int(pub.min()), "p50": int(np.median(pub)), "p90": int(np.percentile(pub, 90)), "max": int(pub.max())},
                      "gather_done_ns_rel_prod0": int(st[17] - st[23]) * 10}))

if "gst" in os.environ.get("FL_LIB_PATH", ""):       # library built with -DFL_GATHER_STAMPS: wave 0's sweeps of the LAST gather of the launch
    k = int(st[47])
    # a 6-pass launch: the LAST gather is the one of pass 5, whose producer stamps are st[20..23]
    h.lio_iterate18(6, F | capi.FL_ITER_STAMP, want_info=False); h.sync()
    st = np.array(h.debug_stamps(), dtype=np.int64); w = np.array(h.debug_wall(), dtype=np.int64)
    k = int(st[47]); t0 = st[23]
    print(json.dumps({"rel": "prod0 published (pass 5)", "gather_start": int(st[16] - t0) * 10, "sentinel_done": int(st[45] - t0) * 10,
                      "sweeps_of_wave0": k, "sweep_start": [int(st[56 + i] - t0) * 10 for i in range(min(k, 8))],
                      "sweep_end": [int(st[48 + i] - t0) * 10 for i in range(min(k, 8))], "batches_needed_at_start": [int(w[2040 + i]) for i in range(min(k, 8))],
                      "waves_polling_done": [int(w[2024 + i] - t0) * 10 for i in range(4)], "reduced": int(w[2036] - t0) * 10, "gather_done": int(st[17] - t0) * 10, "last_publish": int((w[:160] - t0).max()) * 10}))
