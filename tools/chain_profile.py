"""Phases of exact_chain.h's workgroup chain (fl_chain_f32_block) on 2 k patch-error-like floats, in shader-clock cycles, beside the
wavefront form (alone on a CU: the best case).  python tools/chain_profile.py [m]"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
capi = importlib.import_module("fast-livo_amd.capi")
synth = importlib.import_module("fast-livo_amd.synth")
m = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
scene = synth.make_scene()
fr = synth.make_lio_frame(2000, scene=scene)
vf = synth.make_vio_frame(64, fr)
h = capi.Handle(capi.config_from_frames(fr, vf), debug=True)
rng = np.random.default_rng(3)
names = ["entry", "scan1 done", "barrier1", "scan2 done", "barrier2", "events written", "barrier3", "walk done", "barrier4", "exit", "(wave form start)", "(wave form end)"]
acc = np.zeros(12)
reps = 20
for r in range(reps):
    e = ((rng.standard_normal((m, 64)).astype(np.float32) * 10) ** 2).sum(axis=1, dtype=np.float32)
    a, b, w, fell = h.debug_chain(e, 0.0, full=True)
    assert a.tobytes() == b.tobytes() == w.tobytes()
    acc += h.chain_profile
acc /= reps
for n, v in zip(names, acc):
    print(f"{n:20s} {v:9.0f} cycles")
print(f"workgroup form {acc[9]:.0f} cycles, wavefront form {acc[11] - acc[10]:.0f} cycles (100 MHz counter x ?: see NOTES -- clock64 ticks)")
