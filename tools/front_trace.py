"""fl_lidar_front a few times (fused, then staged) for tools/ktrace.sh:  bash tools/ktrace.sh 40 tools/front_trace.py 24000 [staged]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastlivo  # noqa
from fast_livo_amd import capi, synth
raw = int(sys.argv[1]) if len(sys.argv) > 1 else 24000
staged = len(sys.argv) > 2 and sys.argv[2] == "staged"
lio = synth.make_lio_frame(raw)
f = synth.make_imu_frame(raw, n_imu=20, lio=lio, quiet=True)
f.pts_xyzt[:, :3] = lio.body_xyz
h = capi.Handle(capi.config_from_frames(lio, max_iterations=10))
h.map_set_points(lio.scene.map_xyz, 0.5)
pts = h.host_alloc((raw, 4), np.float32)
pts[:] = f.pts_xyzt
for _ in range(4):
    x = capi.state18_from_frame(lio); pr = capi.imu_proc_from_frame(f)
    info, m = h.lidar_front(pr, x, f.imu, f.pcl_beg_time, f.pcl_end_time, pts, 0.15, staged=staged)
print("scan", m, "iterations", info.iterations, "neff", info.effct_feat_num)
h.close()
