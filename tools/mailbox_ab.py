"""Frame drivers with and without the result mailbox (FL_OPT_MAILBOX): host wall time of fl_lio_frame18_dev and fl_vio_compute_j."""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi = importlib.import_module("fast-livo_amd.capi")
synth = importlib.import_module("fast-livo_amd.synth")
scene = synth.make_scene()
fr = synth.make_lio_frame(50000, scene=scene, point_seed=synth.SEED + 101)
vf = synth.make_vio_frame(2000, fr, patch_seed=synth.SEED + 103)
h = capi.Handle(capi.config_from_frames(fr, vf, max_iterations=10, device=0))
h.map_set_points(scene.map_xyz, 0.5)
h.vio_set_frame(vf.img)
h.vio_set_patches(vf.ref_patch, vf.pos, vf.search_level)
scan = h.host_alloc(fr.body_xyz.shape, np.float32)
scan[:] = fr.body_xyz
for mailbox in (3, 0, 1, 2, 3, 0, 1, 2):      # bit 0: fl_vio_compute_j, bit 1: fl_lio_frame18_dev
    h.set_option(capi.FL_OPT_MAILBOX, mailbox)
    tl, tv = [], []
    for rep in range(300):
        x = capi.state18_from_frame(fr)
        t0 = time.perf_counter(); h.lio_frame18_dev(x, scan); t1 = time.perf_counter()
        xv = capi.state18_from_frame(fr); xp = capi.state18_from_frame(fr)
        t2 = time.perf_counter(); h.vio_compute_j(xv, xp); t3 = time.perf_counter()
        tl.append(t1 - t0); tv.append(t3 - t2)
    print(f"mailbox {mailbox}: lio_frame median {np.median(tl[20:]) * 1e6:7.1f} us, compute_j median {np.median(tv[20:]) * 1e6:7.1f} us")
