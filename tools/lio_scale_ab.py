"""us per forced at-scale LIO pass (8 M and 32 M points, one launch per pass) for several library builds, each in a process of its own,
interleaved twice in ONE gpurun call:   python tools/lio_scale_ab.py build_ab/lib_a.so build_ab/lib_b.so ...   ("-" = the in-tree library)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, %r)
import fastlivo  # noqa
import torch
from fast_livo_amd import capi, synth
scene = synth.make_scene()
fr0 = synth.make_lio_frame(200000, scene=scene)
vf = synth.make_vio_frame(16, fr0)
cfg = capi.config_from_frames(fr0, vf, max_iterations=1)
x0 = capi.state18_from_frame(fr0)
nbr0, valid0 = synth.knn5(scene, fr0.world_at(fr0.R_prior, fr0.p_prior))
out = {}
for n in (8000000, 32000000):
    reps = (n + fr0.n - 1) // fr0.n
    body = np.tile(fr0.body_xyz, (reps, 1))[:n]; nbr = np.tile(nbr0, (reps, 1, 1))[:n]; valid = np.tile(valid0, reps)[:n]
    h = capi.Handle(cfg)
    h.set_stream(torch.cuda.current_stream().cuda_stream)
    h.lio_set_points(body); h.lio_begin18(x0, x0); h.lio_set_neighbours(nbr, valid)
    del nbr, body
    for _ in range(5): h.lio_iterate18(1, capi.FL_ITER_FORCE, want_info=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 30
    e0.record()
    for _ in range(K): h.lio_iterate18(1, capi.FL_ITER_FORCE, want_info=False)
    e1.record(); torch.cuda.synchronize()
    li = h.lio_iterate18(0, capi.FL_ITER_FORCE)
    out[n] = [round(e0.elapsed_time(e1) * 1e3 / K, 1), int(li.effct_feat_num), int(li.status)]
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
