"""us per forced VIO pass at scale for several library builds (A/B of the one-patch-per-lane producers), interleaved in ONE gpurun call:
    python tools/vio_wide_ab.py build_ab/lib_a.so build_ab/lib_b.so ...     ("-" = the in-tree library)
Each library runs in a process of its own (FL_LIB_PATH); patches = the frame's 2 000 tiled to 1 M and 200 k, as bench.py's vio_sweep."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, %r)
import fastlivo  # noqa
import torch
from fast_livo_amd import capi, synth
fr = synth.make_lio_frame(2000)
vf = synth.make_vio_frame(2000, fr)
cfg = capi.config_from_frames(fr, vf, max_iterations=1)
x0 = capi.state18_from_frame(fr)
out = {}
for m in (1000000, 200000):
    reps = (m + vf.m - 1) // vf.m
    ref = np.tile(vf.ref_patch, (reps, 1, 1))[:m]; pos = np.tile(vf.pos, (reps, 1))[:m]; sl = np.tile(vf.search_level, reps)[:m]
    h = capi.Handle(cfg)
    h.set_stream(torch.cuda.current_stream().cuda_stream)
    if os.environ.get("FL_WIDE") is not None:
        h.set_option(14, int(os.environ["FL_WIDE"]))
    h.vio_set_frame(vf.img); h.vio_set_patches(ref, pos, sl); h.vio_begin(x0, x0)
    for _ in range(5): h.vio_iterate(0, 1, capi.FL_ITER_FORCE, want_info=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 30
    e0.record()
    for _ in range(K): h.vio_iterate(0, 1, capi.FL_ITER_FORCE, want_info=False)
    e1.record(); torch.cuda.synchronize()
    out[m] = round(e0.elapsed_time(e1) * 1e3 / K, 2)
    h.close()
print(json.dumps(out))
''' % ROOT

libs = sys.argv[1:] or ["-"]
for rnd in range(2):
    for lib in libs:
        env = dict(os.environ)
        if lib != "-":
            env["FL_LIB_PATH"] = os.path.join(ROOT, lib)
        r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]
        print(f"{lib:28s} round {rnd}: {line}", flush=True)
