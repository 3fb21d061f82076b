"""SURVEY 8f N4 measurement: UndistortPcl on the device (HIP events around forward + 3 undistortion launches) vs the CPU restatement."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlivo  # noqa
from fast_livo_amd import capi, synth
from oracle import oracle as orc
res = []
for n, k in [(24000, 20), (100000, 20), (100000, 100), (1000000, 20)]:
    f = synth.make_imu_frame(n, n_imu=k)
    h = capi.Handle(capi.config_from_frames(f.lio))
    run = lambda want: h.imu_undistort(capi.imu_proc_from_frame(f), capi.state18_from_frame(f.lio), f.imu, f.pcl_beg_time, f.pcl_end_time, f.pts_xyzt, want=want)
    run(False); h.set_timing(True)
    ks, ws = [], []
    for _ in range(20):
        t0 = time.perf_counter(); run(False); ws.append(time.perf_counter() - t0); ks.append(h.last_kernel_ms())
    cs = []
    for _ in range(3):
        po = orc.imu_proc_from_frame(f); xo = orc.state18_from_frame(f.lio)
        t0 = time.perf_counter(); orc.imu_undistort(po, xo, f.imu, f.pcl_beg_time, f.pcl_end_time, f.pts_xyzt); cs.append(time.perf_counter() - t0)
    res.append({"points": n, "imu_samples": k, "device_kernels_us": round(float(np.median(ks)) * 1e3, 1),
                "device_call_incl_h2d_us": round(float(np.median(ws)) * 1e6, 1), "cpu_oracle_us": round(float(np.median(cs)) * 1e6, 1)})
print(json.dumps(res))
