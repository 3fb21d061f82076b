"""Per-workgroup timeline of the device k-NN search kernel (instrumented build, fl_debug_knn_stamp, 100 MHz wall clock)."""
import os, sys, json, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlivo  # noqa
from fast_livo_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
fr = synth.make_lio_frame(n)
if os.environ.get("FL_ORDER", "voxel") == "voxel":      # as pcl::VoxelGrid emits feats_down_body (bench.py SCAN_ORDER_NOTE)
    fr = synth.in_voxel_order(fr, 0.15)
h = capi.Handle(capi.config_from_frames(fr, max_iterations=10), debug=True)
L = capi.lib(debug=True); h.debug_knn_stamp(True)
h.map_set_points(fr.scene.map_xyz, 0.5)
x = capi.state18_from_frame(fr); h.lio_set_points(fr.body_xyz); h.lio_begin18(x, x)
h.set_timing(True)
for _ in range(4):
    h.lio_search18(fr.n, want=False); h.sync()
    us = h.last_kernel_ms() * 1e3
    w = (C.c_longlong * 2048)(); L.fl_debug_get_wall(h.h, w); w = np.array(w[:], dtype=np.int64).reshape(4, 512)
    nb = min(512, (n + 63) // 64)
    w = w[:, :nb]; t0 = w[0].min()
    q = lambda a: [int(np.min(a) - t0), int(np.median(a) - t0), int(np.max(a) - t0)]
    print(json.dumps({"kernel_us": us, "blocks_stamped": nb, "unit": "10 ns", "start": q(w[0]), "lookups_done": q(w[1]), "phase1_done": q(w[2]), "end": q(w[3]),
                      "dur_lookup_med": float(np.median(w[1] - w[0])), "dur_walk_med": float(np.median(w[2] - w[1])), "dur_fit_med": float(np.median(w[3] - w[2]))}))
