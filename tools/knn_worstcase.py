import os, sys, json
import numpy as np
sys.path.insert(0, "/root/repo")
import fastlivo
from fast_livo_amd import capi, synth
fr = synth.make_lio_frame(50000)
h = capi.Handle(capi.config_from_frames(fr, max_iterations=10))
x = capi.state18_from_frame(fr); h.lio_set_points(fr.body_xyz); h.lio_begin18(x, x)
h.set_timing(True)
out = {}
for name, m in (("near", fr.scene.map_xyz), ("far_all_invalid", fr.scene.map_xyz + np.float32(50.0)),
                ("half_far", np.where((np.arange(len(fr.scene.map_xyz)) % 2 == 0)[:, None], fr.scene.map_xyz, fr.scene.map_xyz + np.float32(50.0)).astype(np.float32)),
                ("sparse_1pct", fr.scene.map_xyz[::100])):
    h.map_set_points(m, 0.5)
    ks = []
    for _ in range(5):
        h.lio_search18(fr.n, want=False); ks.append(h.last_kernel_ms() * 1e3)
    out[name] = round(float(np.median(ks)), 1)
print(json.dumps(out))
