"""Randomised cross-check (not part of the test suite; run on the GPU box): random scan sizes / iteration caps / cell sizes,
all-device frame vs host-kNN frame of the same library (1e-12) and vs the CPU oracle frame (1e-9); voxel filter and
undistortion on ragged sizes vs the oracle."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlivo  # noqa
from fast_livo_amd import capi, synth
from oracle import oracle as orc
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
scene = synth.make_scene()
bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 25):
    n = int(rng.choice([1, 2, 5, 63, 64, 65, 255, 256, 257, 1000, 4097, 20000, 65279, 65281, 70000, 130000]))
    max_iter = int(rng.integers(1, 11))
    cell = float(rng.choice([0.3, 0.5, 0.8, 1.5]))
    fr = synth.make_lio_frame(n, scene=scene, point_seed=int(rng.integers(1 << 30)))
    def knn(w):
        nb, _, va, _ = orc.knn5_bruteforce(scene.map_xyz, w)
        return nb, va
    h = capi.Handle(capi.config_from_frames(fr, max_iterations=max_iter))
    h.map_set_points(scene.map_xyz, cell)
    xg = capi.state18_from_frame(fr); ig = h.lio_frame18_dev(xg, fr.body_xyz)
    xh = capi.state18_from_frame(fr); ih = h.lio_frame18(xh, fr.body_xyz, knn)
    xo = orc.state18_from_frame(fr); ro = orc.lio18_frame(xo, fr.body_xyz, fr.R_LI, fr.t_LI, fr.laser_point_cov, max_iter, knn)
    e1 = np.abs(xg.vec() - xh.vec()).max(); e2 = np.abs(xg.vec() - xo.vec()).max(); e3 = np.abs(xg.cov_np() - xo.cov_np()).max()
    ok = e1 <= 1e-12 and e2 <= 1e-9 and e3 <= 1e-12 and ig.iterations == ro["out"].iterations and ig.effct_feat_num == ro["out"].effct_feat_num
    # ragged voxel / imu
    m = int(rng.integers(1, 5000))
    p = np.concatenate([rng.uniform(-20, 20, (m, 3)), rng.uniform(0, 99, (m, 1))], 1).astype(np.float32)
    leaf = float(rng.choice([0.1, 0.33, 1.0]))
    ref, _ = orc.voxel_grid(p, leaf); out, mm, _ = h.scan_voxel_filter(p, leaf)
    ok2 = mm == ref.shape[0] and np.array_equal(out.view(np.uint32), ref.view(np.uint32))
    f = synth.make_imu_frame(m, n_imu=int(rng.integers(1, 40)), seed=int(rng.integers(1 << 20)), time_sorted=bool(rng.integers(2)))
    xa = orc.state18_from_frame(f.lio); pa = orc.imu_proc_from_frame(f)
    rp, _ = orc.imu_undistort(pa, xa, f.imu, f.pcl_beg_time, f.pcl_end_time, f.pts_xyzt)
    xb = capi.state18_from_frame(f.lio); pb = capi.imu_proc_from_frame(f)
    gp, _ = h.imu_undistort(pb, xb, f.imu, f.pcl_beg_time, f.pcl_end_time, f.pts_xyzt)
    ok3 = np.abs(gp[:, :3].astype(np.float64) - rp[:, :3]).max() <= 4e-6 and np.abs(xb.vec() - xa.vec()).max() <= 1e-11
    if not (ok and ok2 and ok3):
        bad += 1
        print("MISMATCH", dict(n=n, max_iter=max_iter, cell=cell, e1=e1, e2=e2, e3=e3, it=(ig.iterations, ro["out"].iterations),
                               neff=(ig.effct_feat_num, ro["out"].effct_feat_num), ok2=bool(ok2), ok3=bool(ok3), m=m))
    h.close()
print(json.dumps({"trials": trial + 1, "mismatches": bad}))
