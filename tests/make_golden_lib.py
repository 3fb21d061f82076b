"""Deterministic small cases whose oracle outputs are frozen under tests/golden/*.json."""
import numpy as np


def run_lio18_iter(orc, scene):
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(1500, scene=scene)
    x = orc.state18_from_frame(fr)
    nbr, valid = synth.knn5(scene, fr.world_at(fr.R_prior, fr.p_prior))
    sel = valid.copy()
    r = orc.lio18_iterate(x, x.copy(), fr.body_xyz, nbr, sel, fr.R_LI, fr.t_LI, fr.laser_point_cov, nthreads=2)
    o = r["out"]
    return {"solution": list(o.solution), "HTH": list(o.HTH), "HTz": list(o.HTz), "neff": [o.effct_feat_num],
            "total_residual": [o.total_residual], "state_after": x.vec().tolist(),
            "sel_checksum": [int(np.flatnonzero(sel).sum())]}


def run_vio_level(orc, scene):
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(256, scene=scene)
    vf = synth.make_vio_frame(40, fr)
    x = orc.state18_from_frame(fr)
    r = orc.vio_update_state(vf, x, x.copy(), 1e10, 1)
    o = r["out"]
    return {"solution": list(o.solution), "HTH": list(o.HTH), "HTz": list(o.HTz), "error": [r["error"]],
            "iterations": [o.iterations], "errors_head": r["errors"][:8].tolist(), "state_after": x.vec().tolist()}


def run_ikfom_update(orc, scene):
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(1200, scene=scene)
    x = orc.state23_from_frame(fr, synth.quat_from_R)
    P = fr.cov23.copy()
    r = orc.ikfom_update(x, P, fr.body_xyz, 0.001, 4, lambda w: synth.knn5(scene, w), nthreads=2)
    return {"dx": list(r["out"].dx), "iterations": [r["out"].iterations], "neff": [r["out"].effct_feat_num],
            "state_after": x.vec().tolist(), "P_diag": np.diag(P).tolist(), "P_row0": P[0].tolist()}


def _ck(a):
    """order-sensitive checksum of an array's bit pattern (fixtures stay small)"""
    b = np.ascontiguousarray(a).reshape(-1).view(np.uint8).astype(np.uint64)
    return [int((b * (np.arange(b.size, dtype=np.uint64) % 251 + 1)).sum() % (1 << 53))]


def run_knn5(orc, scene):
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(400, scene=scene)
    w = fr.world_at(fr.R_prior, fr.p_prior).astype(np.float32)
    nbr, sq, valid, idx = orc.knn5_bruteforce(scene.map_xyz, w, nthreads=2)
    return {"idx_head": idx[:6].reshape(-1).tolist(), "sq_head": sq[:6].reshape(-1).tolist(), "valid_sum": [int(valid.sum())],
            "idx_checksum": _ck(idx), "nbr_checksum": _ck(nbr)}


def run_voxel_grid(orc, scene):
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(3000, scene=scene)
    p = np.concatenate([fr.body_xyz, np.linspace(0, 50, fr.n, dtype=np.float32)[:, None]], 1).astype(np.float32)
    out, small = orc.voxel_grid(p, 0.4)
    return {"count": [int(out.shape[0])], "small": [int(small)], "head": out[:5].reshape(-1).tolist(), "checksum": _ck(out)}


def run_imu_undistort(orc, scene):
    from fast_livo_amd import synth
    f = synth.make_imu_frame(500, n_imu=12, seed=31)
    x = orc.state18_from_frame(f.lio); pr = orc.imu_proc_from_frame(f)
    pts, poses = orc.imu_undistort(pr, x, f.imu, f.pcl_beg_time, f.pcl_end_time, f.pts_xyzt)
    return {"n_poses": [len(poses)], "pose_last_pos": list(poses[-1].pos), "pose_last_rot": list(poses[-1].rot),
            "state_after": x.vec().tolist(), "cov_diag": np.diag(np.array(x.cov).reshape(18, 18)).tolist(),
            "pts_head": pts[:4].reshape(-1).tolist(), "pts_mean": pts[:, :3].astype(np.float64).mean(0).tolist(),
            "acc_s_last": list(pr.acc_s_last)}


def run_vio_select(orc, scene):
    from fast_livo_amd import synth
    sf = synth.make_select_frame(60, seed=41)
    cfg = orc.vio_config(sf.vio)
    depth = orc.vio_depth_image(cfg, sf.Rcw, sf.Pcw, sf.scan_world)
    r = orc.vio_select(cfg, sf.Rcw, sf.Pcw, sf.vio.img, sf.keyframes, depth, orc.patch_candidates(sf), outlier_threshold=300.0)
    return {"reason": r["reason"].tolist(), "idx": r["idx"].tolist(), "levels": r["levels"].tolist(), "errors": r["errors"].tolist(),
            "depth_nonzero": [int((depth > 0).sum())], "depth_sum": [float(depth.astype(np.float64).sum())],
            "patch0": r["patches"][0].tolist() if len(r["idx"]) else [], "patches_checksum": _ck(r["patches"])}
