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
print("levels:", [(int(i.iterations), int(i.accepted), int(i.status)) for i in infos])
rows = []
for k in range(16):
    a = w[16 * k: 16 * k + 16]
    if a[8] > 0:
        rows.append(a)
rows.sort(key=lambda a: a[11])
rows = rows[-10:]                      # the passes of the last ComputeJ call, by epoch
t0 = rows[0][8]
us = lambda v: round((int(v) - int(t0)) / 100.0, 2)
for a in rows:
    print(f"epoch {int(a[11])} level {int(a[12])}: solver gathered {us(a[8]):8.2f}  pass done {us(a[9]):8.2f} (+{(int(a[9]) - int(a[8])) / 100.0:5.2f})  "
          f"flags(fragile 16, audited 2, timeout 4, accept 8) = {int(a[10]):2d} | auditor: loop top {us(a[0]):8.2f} pose seen {us(a[1]):8.2f} chain done {us(a[2]):8.2f}")
    print(f"          solver: loop top {us(a[13]):8.2f} gather entered {us(a[14]):8.2f}")
    print(f"          producer wg 0: loop top {us(a[4]):8.2f} pose seen {us(a[5]):8.2f} produce returned {us(a[6]):8.2f}")
    c = w[1024 + 4 * (int(a[11]) & 63): 1024 + 4 * (int(a[11]) & 63) + 3]
    print(f"          auditor chain stamps: entry {us(c[0]):8.2f} staged {us(c[1]):8.2f} (+{(int(c[1]) - int(c[0])) / 100.0:5.2f}) added {us(c[2]):8.2f} (+{(int(c[2]) - int(c[1])) / 100.0:5.2f})")

# every producer workgroup's last pass (= the call's last pass): pose seen / produce returned, against that pass's gather
last = rows[-1]
ps = w[256:256 + 254].reshape(127, 2)
ps = ps[ps[:, 0] > 0]
seen = (ps[:, 0] - int(last[8])) / 100.0
ret = (ps[:, 1] - int(last[8])) / 100.0
print(f"last pass, {len(ps)} producer workgroups, us before its gather completed: pose seen min {seen.min():.2f} med {np.median(seen):.2f} max {seen.max():.2f} | "
      f"produce returned min {ret.min():.2f} med {np.median(ret):.2f} max {ret.max():.2f}; slowest workgroups {np.argsort(ret)[-5:].tolist()}")
