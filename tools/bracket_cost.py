"""Where the fixed cost of bench.py's 20-step bracket goes: enqueue time, and the wait by torch.cuda.synchronize() alone vs
hipStreamSynchronize (fl_sync) first."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import fastlivo  # noqa
from fast_livo_amd import capi, synth
scene = synth.make_scene()
fr = synth.make_lio_frame(50000, scene=scene); vf = synth.make_vio_frame(2000, fr)
nbr, valid = synth.knn5(scene, fr.world_at(fr.R_prior, fr.p_prior))
cfg = capi.config_from_frames(fr, vf, max_iterations=10); x0 = capi.state18_from_frame(fr)
side = torch.cuda.Stream(); torch.cuda.set_stream(side); stream = torch.cuda.current_stream().cuda_stream
hl, hv = capi.Handle(cfg), capi.Handle(cfg); hl.set_stream(stream); hv.set_stream(stream)
hl.lio_set_points(fr.body_xyz); hl.lio_begin18(x0, x0); hl.lio_set_neighbours(nbr, valid)
hv.vio_set_frame(vf.img); hv.vio_set_patches(vf.ref_patch, vf.pos, vf.search_level); hv.vio_begin(x0, x0)
F = capi.FL_ITER_FORCE
def steps(k, order):
    if order == "alt":
        for _ in range(k // 10):
            hl.lio_iterate18(10, F, want_info=False); hv.vio_iterate(0, 10, F, want_info=False)
    else:
        for _ in range(k // 10): hl.lio_iterate18(10, F, want_info=False)
        for _ in range(k // 10): hv.vio_iterate(0, 10, F, want_info=False)
for order in ("alt", "grouped"):
    for mode in ("torch", "stream+torch"):
        res = []
        for rep in range(30):
            steps(20, order); torch.cuda.synchronize(); time.sleep(0.002)
            t0 = time.perf_counter()
            steps(20, order)
            t1 = time.perf_counter()
            if mode == "stream+torch": hv.sync()
            t2 = time.perf_counter()
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            res.append(((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t3 - t0) * 1e6))
        r = np.median(np.array(res[5:]), axis=0)
        print(json.dumps({"order": order, "wait": mode, "enqueue_us": round(r[0], 1), "stream_sync_us": round(r[1], 1), "torch_sync_us": round(r[2], 1), "total_us": round(r[3], 1),
                          "us_per_step": round(r[3] / 20, 2)}))
# one-shot samples after the device has been idle (what the driver's `--steps 20 --warmup 5` run measures), with and without a wake-up phase
for wake in (0, 0, 200, 200, 2000, 0):
    torch.cuda.synchronize(); time.sleep(3.0)
    if wake: steps(wake, "alt"); torch.cuda.synchronize()
    steps(5 * 2, "alt"); torch.cuda.synchronize()
    hv.vio_iterate(0, 0, F)
    t0 = time.perf_counter(); steps(20, "alt"); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(json.dumps({"idle_s": 3.0, "wake_up_steps": wake, "one_shot_20_steps_us": round((t1 - t0) * 1e6, 1), "us_per_step": round((t1 - t0) * 1e6 / 20, 2)}))
