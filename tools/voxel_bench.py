"""SURVEY 8f N3 measurement: pcl::VoxelGrid on the device (events around the 7 launches) vs the CPU restatement."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlivo  # noqa
from fast_livo_amd import capi, synth
from oracle import oracle as orc
res = []
for n, leaf in [(24000, 0.15), (100000, 0.15), (400000, 0.15), (2000000, 0.15)]:
    fr = synth.make_lio_frame(n)
    p = np.concatenate([fr.body_xyz, np.zeros((n, 1), np.float32)], 1).astype(np.float32)
    h = capi.Handle(capi.config_from_frames(fr))
    h.scan_voxel_filter(p, leaf, stage_as_scan=True, want=False)
    h.set_timing(True)
    ks, ws = [], []
    for _ in range(20):
        t0 = time.perf_counter(); _, m, _ = h.scan_voxel_filter(p, leaf, stage_as_scan=True, want=False); ws.append(time.perf_counter() - t0)
        ks.append(h.last_kernel_ms())
    cs = []
    for _ in range(3):
        t0 = time.perf_counter(); ref, _ = orc.voxel_grid(p, leaf); cs.append(time.perf_counter() - t0)
    res.append({"points": n, "leaf": leaf, "voxels": m, "device_kernels_us": round(float(np.median(ks)) * 1e3, 1),
                "device_call_incl_h2d_us": round(float(np.median(ws)) * 1e6, 1), "cpu_oracle_us": round(float(np.median(cs)) * 1e6, 1),
                "algorithmic_bytes": n * 16 + m * 28, "GBps_kernels": round((n * 16 + m * 28) / (float(np.median(ks)) * 1e-3) / 1e9, 1)})
print(json.dumps(res))
