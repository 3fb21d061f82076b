"""SURVEY 8f N2 measurement: device patch selection / affine warp (HIP events around memset + 4 launches) vs the CPU restatement."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlivo  # noqa
from fast_livo_amd import capi, synth
from oracle import oracle as orc
res = []
for m in (208, 2000, 20000):
    sf = synth.make_select_frame(m)
    h = capi.Handle(capi.config_from_frames(sf.lio, sf.vio))
    ids = [h.vio_add_keyframe(k) for k in sf.keyframes]
    h.vio_set_frame(sf.vio.img)
    cand = capi.patch_candidates(sf, ids)
    run = lambda: h.vio_select_patches(sf.Rcw, sf.Pcw, sf.scan_world, cand, outlier_threshold=300.0, want_patches=False)
    out = run(); h.set_timing(True)
    ks, ws = [], []
    for _ in range(20):
        t0 = time.perf_counter(); run(); ws.append(time.perf_counter() - t0); ks.append(h.last_kernel_ms())
    cfg = orc.vio_config(sf.vio); oc = orc.patch_candidates(sf)
    cs = []
    for _ in range(3):
        t0 = time.perf_counter()
        depth = orc.vio_depth_image(cfg, sf.Rcw, sf.Pcw, sf.scan_world)
        orc.vio_select(cfg, sf.Rcw, sf.Pcw, sf.vio.img, sf.keyframes, depth, oc, outlier_threshold=300.0)
        cs.append(time.perf_counter() - t0)
    # algorithmic bytes per candidate: 200 B candidate + 81 depth words (8 B) + 3*64*4 taps (u8) + 64*4 current taps + 768 B patch out (+768 B copy)
    res.append({"candidates": m, "scan_points": int(sf.scan_world.shape[0]), "accepted": int(len(out["idx"])),
                "device_kernels_us": round(float(np.median(ks)) * 1e3, 1), "device_call_us": round(float(np.median(ws)) * 1e6, 1),
                "cpu_oracle_us": round(float(np.median(cs)) * 1e6, 1)})
print(json.dumps(res))
