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


def map_case(scene):
    rng = np.random.default_rng(808)
    m0 = scene.map_xyz[rng.choice(len(scene.map_xyz), 4000, replace=False)]
    new = (scene.map_xyz[rng.choice(len(scene.map_xyz), 1500, replace=False)] + rng.normal(0, 0.02, (1500, 3))).astype(np.float32)
    lo = m0.min(0)
    box = np.array([[lo[0], lo[1], lo[2], lo[0] + 4.0, lo[1] + 30.0, lo[2] + 10.0]], dtype=np.float32)
    return m0, new, box


def run_map_update(orc, scene):
    m0, new, box = map_case(scene)
    a, ia = orc.map_add_points(m0, new, 0.25)
    b, ib = orc.map_delete_boxes(a, box)
    return {"after_add": [ia.n_after, ia.n_added, ia.n_removed, ia.n_ambiguous], "after_delete": [ib.n_after, ib.n_removed],
            "add_checksum": _ck(a), "delete_checksum": _ck(b), "tail": a[-3:].reshape(-1).tolist()}


def vmap_case(scene):
    from fast_livo_amd import synth
    lio = synth.make_lio_frame(2500, scene=scene)
    vf = synth.make_vio_frame(8, lio)
    scan = lio.world_at(lio.R_true, lio.p_true).astype(np.float32)
    poses, imgs = [], []
    R_wi, p_wi = lio.R_true.copy(), lio.p_true.copy()
    for k in range(4):
        R_wi = R_wi @ synth.exp_so3(np.array([0.0, 0.0, 0.005]))
        p_wi = p_wi + np.array([0.2, 0.1, 0.0])
        poses.append(synth.cam_pose(vf.Rcl, vf.Pcl, lio.R_LI, lio.t_LI, R_wi, p_wi))
        imgs.append(np.ascontiguousarray(np.roll(vf.img, (k, -2 * k), axis=(0, 1))))
    return lio, vf, scan, poses, imgs


def vmap_summary(sel_counts, added, obs, points):
    """points: list of (pos, value, obs list) -- folded into checksums"""
    pos = np.array([p[0] for p in points], np.float64)
    val = np.array([p[1] for p in points], np.float32)
    nob = np.array([len(p[2]) for p in points], np.int32)
    px = np.array([[o.px[0], o.px[1]] for p in points for o in p[2]], np.float64)
    fr = np.array([o.frame_id for p in points for o in p[2]], np.int32)
    return {"selected": sel_counts, "added": added, "obs_added": obs, "n_points": [len(points)], "pos_checksum": _ck(pos),
            "value_checksum": _ck(val), "n_obs_checksum": _ck(nob), "px_checksum": _ck(px), "frame_checksum": _ck(fr)}


def run_vmap_sequence(orc, scene):
    lio, vf, scan, poses, imgs = vmap_case(scene)
    vm = orc.VMap(orc.vio_config(vf), 40)
    sel_counts, added, obs = [], [], []
    for k, ((Rcw, Pcw), img) in enumerate(zip(poses, imgs)):
        s = vm.select(Rcw, Pcw, img, imgs[:k + 1], scan, outlier_threshold=1e12)
        sel_counts.append(len(s["points"]))
        added.append(vm.add_sparse(Rcw, Pcw, img, scan, k, k))
        obs.append(vm.add_observation(Rcw, Pcw, img, s["points"], s["levels"], k, k))
    pts = [vm.get_point(i) for i in range(vm.size())]
    vm.close()
    return vmap_summary(sel_counts, added, obs, pts)
