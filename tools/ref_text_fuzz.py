#!/usr/bin/env python3
"""Random frames through the CPU oracle AND through the reference's own text (oracle/_ref/libeigen_ref.so, recipe oracle/ref_eigen/): every
output must agree -- bit for bit where the library was built over the stand-in linear algebra (`linalg_kind() == "shim"`), at the
tolerances of tests/test_ref_eigen_cpu.py over a real Eigen.  CPU only.  TEST INFRASTRUCTURE (imports oracle/).

    python tools/ref_text_fuzz.py [--lio N] [--vio N] [--imu N] [--ikf N] [--sel N] [--seed S]

Prints one summary line per family and exits non-zero on the first disagreement (with the seed that reproduces it).
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fastlivo  # noqa: E402,F401
from fast_livo_amd import synth  # noqa: E402
from oracle import eigenref, ikdref, oracle as orc  # noqa: E402


def same(a, b, tol, exact):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a.shape != b.shape:
        return False
    if exact:
        return np.array_equal(a, b)
    return bool(a.size == 0 or np.abs(a - b).max() <= tol * max(1.0, float(np.abs(b).max())))


def fail(family, seed, what):
    print(f"MISMATCH {family} seed {seed}: {what}")
    sys.exit(1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lio", type=int, default=40)
    ap.add_argument("--vio", type=int, default=40)
    ap.add_argument("--imu", type=int, default=40)
    ap.add_argument("--ikf", type=int, default=20)
    ap.add_argument("--sel", type=int, default=20)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    exact = eigenref.linalg_kind() == "shim"
    print(f"library {eigenref.LIB_PATH}: linalg_kind {eigenref.linalg_kind()} -> {'bit for bit' if exact else 'tolerances'}")
    scene = synth.make_scene()
    rng = np.random.default_rng(a.seed)

    # ---- Mode-18 frames: random size, prior error, scan noise, iteration cap, extrinsic
    tree = ikdref.IkdTree()
    tree.build(scene.map_xyz)

    def knn(w):
        xyz, sq, found = tree.nearest(w, 5)
        return xyz, ((found == 5) & (sq[:, 4] <= 5.0)).astype(np.uint8)
    passes = 0
    for k in range(a.lio):
        seed = int(rng.integers(1 << 30))
        n = int(rng.choice([7, 60, 500, 3000, 12000]))
        fr = synth.make_lio_frame(n, seed=seed, scene=scene, scan_noise=float(rng.choice([0.0, 0.01, 0.05])),
                                  rot_pert_deg=float(rng.choice([0.05, 0.5, 3.0])), pos_pert=float(rng.choice([0.002, 0.02, 0.3])),
                                  t_LI=synth.NTU_T_LI if k % 3 == 0 else synth.AVIA_T_LI)
        mi = int(rng.choice([1, 2, 3, 4, 10]))
        xo, xr = orc.state18_from_frame(fr), orc.state18_from_frame(fr)
        ro = orc.lio18_frame(xo, fr.body_xyz, fr.R_LI, fr.t_LI, fr.laser_point_cov, mi, knn, nthreads=1)
        rr = eigenref.lio18_frame(xr, fr.body_xyz, scene.map_xyz, fr.R_LI, fr.t_LI, fr.laser_point_cov, mi)
        passes += ro["out"].iterations
        if ro["out"].effct_feat_num == 0:
            continue                                   # 0 / 0 in res_mean_last on both sides; nothing else to compare
        ok = (ro["out"].iterations == rr["out"].iterations and ro["out"].searches == rr["out"].searches and
              ro["out"].effct_feat_num == rr["out"].effct_feat_num and np.array_equal(ro["sel"], rr["sel"]) and
              np.array_equal(ro["normvec"][ro["sel"] != 0], rr["normvec"][rr["sel"] != 0]) and
              same(xo.vec(), xr.vec(), 1e-11, exact) and same(xo.cov_np(), xr.cov_np(), 1e-12, exact))
        if not ok:
            fail("lio", seed, f"n {n} max_iter {mi}: iterations {ro['out'].iterations}/{rr['out'].iterations} eff {ro['out'].effct_feat_num}/{rr['out'].effct_feat_num} "
                              f"dstate {np.abs(xo.vec() - xr.vec()).max():g}")
    tree.close()
    print(f"lio   {a.lio} frames, {passes} passes: identical")

    # ---- VIO: random patch count, level, search levels, camera, covariance scale, persistent G
    its = 0
    for k in range(a.vio):
        seed = int(rng.integers(1 << 30))
        lf = synth.make_lio_frame(500, seed=seed, scene=scene)
        ntu = k % 4 == 0
        m = int(rng.choice([1, 3, 40, 300, 1500]))
        vf = synth.make_vio_frame(m, lf, seed=seed, distortion=bool(k % 2), ref_noise=float(rng.choice([0.0, 2.0, 10.0])),
                                  img_point_cov=float(rng.choice([100.0, 1000.0])), max_iterations=int(rng.choice([1, 3, 10])),
                                  **(dict(cam=synth.NTU_CAM, Rcl=synth.NTU_RCL, Pcl=synth.NTU_PCL) if ntu else {}))
        if k % 5 == 0:
            vf.search_level[:] = rng.integers(0, 2, m).astype(np.int32)
        xp = orc.state18_from_frame(lf)
        if k % 2 == 0:
            level = int(rng.integers(0, 3 if not (k % 5 == 0) else 2))
            G0 = rng.standard_normal((18, 18)) * 1e-3
            xo, xr = orc.state18_from_frame(lf), orc.state18_from_frame(lf)
            ao = orc.vio_update_state(vf, xo, xp, 1e10, level, G=G0.copy())
            ar = eigenref.vio_update_state(vf, xr, xp, 1e10, level, G=G0.copy())
            its += ao["out"].iterations
            ok = (ao["error"] == ar["error"] and np.array_equal(ao["errors"], ar["errors"]) and same(ao["G"], ar["G"], 1e-11, exact) and
                  same(xo.vec(), xr.vec(), 1e-11, exact))
        else:
            xo, xr = orc.state18_from_frame(lf), orc.state18_from_frame(lf)
            ro = orc.vio_compute_j(vf, xo, xp)
            rr = eigenref.vio_compute_j(vf, xr, xp)
            its += sum(ro["outs"][lv].iterations for lv in range(3))
            ok = np.array_equal(ro["errors"], rr["errors"]) and same(xo.vec(), xr.vec(), 1e-11, exact) and same(xo.cov_np(), xr.cov_np(), 1e-12, exact)
        if not ok:
            fail("vio", seed, f"m {m} case {k}")
    print(f"vio   {a.vio} frames, {its} iterations: identical")

    # ---- undistortion (the reference selects the points up to `(t_end - t_beg) * 1000`, t_end derived from the last point: that product can
    # round below the float the last point -- and every point sharing its time stamp -- carries; those are dropped, see text/imu_2.inc)
    dropped = 0
    for k in range(a.imu):
        seed = int(rng.integers(1 << 30))
        n = int(rng.choice([2, 50, 1000, 8000]))
        f = synth.make_imu_frame(n, n_imu=int(rng.choice([1, 2, 7, 20, 60])), seed=seed, imu_before_frame=bool(k % 2), first_point_late=(k % 7 == 0),
                                 quiet=(k % 5 == 0))
        so, sr = orc.state18_from_frame(f.lio), orc.state18_from_frame(f.lio)
        po, pr = orc.imu_proc_from_frame(f), orc.imu_proc_from_frame(f)
        pts_r, poses_r, t_end = eigenref.imu_undistort(pr, sr, f.imu, f.pcl_beg_time, f.pts_xyzt)
        kept = len(pts_r)
        pts_o, poses_o = orc.imu_undistort(po, so, f.imu, f.pcl_beg_time, t_end, f.pts_xyzt[:kept])
        flat = lambda P: np.array([[q.offset_time, *q.acc, *q.gyr, *q.vel, *q.pos, *q.rot] for q in P])   # noqa: E731
        ok = (len(poses_o) == len(poses_r) and same(flat(poses_o), flat(poses_r), 1e-12, exact) and
              (np.array_equal(pts_o, pts_r) if exact else np.abs(pts_o - pts_r).max() <= 1e-6) and same(so.vec(), sr.vec(), 1e-12, exact) and
              same(so.cov_np(), sr.cov_np(), 1e-12, exact))
        dropped += n - kept
        if not ok:
            fail("imu", seed, f"n {n} kept {kept}")
    print(f"imu   {a.imu} frames: identical ({dropped} points dropped by the reference's own selection)")

    # ---- Mode-23 updates, both halves from the reference's text
    calls_t = 0
    for k in range(a.ikf):
        seed = int(rng.integers(1 << 30))
        n = int(rng.choice([30, 400, 3000]))
        fr = synth.make_lio_frame(n, seed=seed, scene=scene, rot_pert_deg=float(rng.choice([0.1, 0.5, 2.0])), pos_pert=float(rng.choice([0.005, 0.02, 0.2])))
        mi = int(rng.choice([2, 3, 4, 10]))
        A = rng.standard_normal((23, 23))
        P0 = fr.cov23.copy() if k % 2 else 1e-3 * (A @ A.T / 23 + 0.5 * np.eye(23))
        hm = eigenref.HShareModel(fr.body_xyz, scene.map_xyz)
        try:
            s_r, P_r, calls = eigenref.ikfom_update_text_c(orc.state23_from_frame(fr, synth.quat_from_R).vec(), P0.copy(), 0.001, mi, hm.callback)
        finally:
            hm.close()

        def knn2(w):
            nb, _, va, _ = orc.knn5_bruteforce(scene.map_xyz, w)
            return nb, va
        x_o, P_o = orc.state23_from_frame(fr, synth.quat_from_R), P0.copy()
        ro = orc.ikfom_update(x_o, P_o, fr.body_xyz, 0.001, mi, knn2, nthreads=1)
        calls_t += calls
        if not (calls == ro["out"].iterations and same(x_o.vec(), s_r, 1e-12, exact) and same(P_o, P_r, 1e-12, exact)):
            fail("ikf", seed, f"n {n} max_iter {mi}: calls {calls}/{ro['out'].iterations} dstate {np.abs(x_o.vec() - s_r).max():g}")
    print(f"ikf   {a.ikf} updates, {calls_t} calls of h_share_model: identical")

    # ---- patch selection
    acc = 0
    for k in range(a.sel):
        seed = int(rng.integers(1 << 30))
        sf = synth.make_select_frame(int(rng.choice([5, 80, 400])), seed=seed, n_keyframes=int(rng.choice([1, 3, 6])), distortion=bool(k % 2),
                                     discont_frac=float(rng.choice([0.0, 0.1, 0.4])))
        cfg = orc.vio_config(sf.vio)
        depth = orc.vio_depth_image(cfg, sf.Rcw, sf.Pcw, sf.scan_world)
        opt = dict(ncc_en=bool(k % 3 == 0), ncc_thre=float(rng.choice([0.0, 0.5, 0.9])), outlier_threshold=float(rng.choice([30.0, 300.0, 1e12])))
        ro = orc.vio_select(cfg, sf.Rcw, sf.Pcw, sf.vio.img, sf.keyframes, depth, orc.patch_candidates(sf), **opt)
        rr = eigenref.vio_select(cfg, sf.Rcw, sf.Pcw, sf.vio.img, sf.keyframes, depth, orc.patch_candidates(sf), **opt)
        acc += len(ro["idx"])
        ok = np.array_equal(ro["idx"], rr["idx"]) and np.array_equal(ro["levels"], rr["levels"]) and \
            (np.array_equal(ro["patches"], rr["patches"]) and np.array_equal(ro["errors"], rr["errors"]) if exact
             else np.abs(ro["patches"] - rr["patches"]).max(initial=0.0) <= 1e-4)
        if not ok:
            fail("sel", seed, f"candidates {len(sf.cand_pos)}")
    print(f"sel   {a.sel} frames, {acc} accepted patches: identical")


if __name__ == "__main__":
    main()
