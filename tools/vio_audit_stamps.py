"""Time line of the auditor workgroup against the solver in the LAST level launch of fl_vio_compute_j (needs a library built with
FL_EXTRA_FLAGS=-DFL_AUDIT_STAMPS):  FL_LIB_PATH=build_ab/lib_audst.so python tools/vio_audit_stamps.py"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi = importlib.import_module("fast-livo_amd.capi")
synth = importlib.import_module("fast-livo_amd.synth")
m = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
scene = synth.make_scene()
fr = synth.make_lio_frame(50000, scene=scene, point_seed=synth.SEED + 101)
vf = synth.make_vio_frame(m, fr, patch_seed=synth.SEED + 103)
h = capi.Handle(capi.config_from_frames(fr, vf, max_iterations=10, device=0), debug=True)
h.vio_set_frame(vf.img)
h.vio_set_patches(vf.ref_patch, vf.pos, vf.search_level)
for rep in range(5):
    infos = h.vio_compute_j(capi.state18_from_frame(fr), capi.state18_from_frame(fr))
w = np.array(h.debug_wall(), dtype=np.int64)
t0 = w[0]
print("levels:", [(int(i.iterations), int(i.accepted), int(i.status)) for i in infos])
for p in range(int(infos[0].iterations) if False else 6):
    a = w[16 * p: 16 * p + 16]
    us = lambda v: (v - t0) / 100.0
    print(f"pass {p}: auditor loop {us(a[0]):8.2f} bcast {us(a[1]):8.2f} chain done {us(a[2]):8.2f} to={a[3]} | solver gathered {us(a[8]):8.2f} "
          f"pass done {us(a[9]):8.2f} flags(fragile16,audited2,timeout4)={a[10]}")

c = w[1024:1024 + 256].reshape(64, 4)
rows = sorted((r for r in c if r[0] > 0), key=lambda r: r[0])[-6:]
for r in rows:
    print("chain: entry %8.2f staged %8.2f (+%.2f) added %8.2f (+%.2f)" % ((r[0] - t0) / 100.0, (r[1] - t0) / 100.0, (r[1] - r[0]) / 100.0,
                                                                          (r[2] - t0) / 100.0, (r[2] - r[1]) / 100.0))
