"""At large patch counts: the fused VIO pass (vio_pass_kernel<0>: producers + auditor + solver in one launch, 205 VGPRs) against
accumulate (vio_pass_kernel<1>, 163 VGPRs: one more wavefront per SIMD) + solve as two launches.  python tools/vio_accum_vs_fused.py [m ...]"""
import importlib, os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
capi = importlib.import_module("fast-livo_amd.capi")
synth = importlib.import_module("fast-livo_amd.synth")
ms = [int(a) for a in sys.argv[1:]] or [200000, 1000000]
scene = synth.make_scene()
fr = synth.make_lio_frame(50000, scene=scene, point_seed=synth.SEED + 101)
vf = synth.make_vio_frame(2000, fr, patch_seed=synth.SEED + 103)
cfg = capi.config_from_frames(fr, vf, max_iterations=10)
x0 = capi.state18_from_frame(fr)
F = capi.FL_ITER_FORCE
for m in ms:
    reps = (m + vf.m - 1) // vf.m
    ref = np.tile(vf.ref_patch, (reps, 1, 1))[:m]; pos = np.tile(vf.pos, (reps, 1))[:m]; sl = np.tile(vf.search_level, reps)[:m]
    h = capi.Handle(cfg)
    h.set_stream(torch.cuda.current_stream().cuda_stream)
    h.vio_set_frame(vf.img); h.vio_set_patches(ref, pos, sl); h.vio_begin(x0, x0)
    d_sums = torch.zeros(64, dtype=torch.float64, device="cuda")
    K = 20
    res = {"patches": m}
    for name, fn in (("fused_pass_us", lambda: h.vio_iterate(0, 1, F, want_info=False)),
                     ("accumulate_plus_solve_us", lambda: (h.vio_accumulate(0, d_sums.data_ptr()), h.vio_solve(d_sums.data_ptr(), F)))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(K):
            fn()
        e1.record(); torch.cuda.synchronize()
        res[name] = round(e0.elapsed_time(e1) * 1e3 / K, 1)
    print(json.dumps(res))
    h.close()
