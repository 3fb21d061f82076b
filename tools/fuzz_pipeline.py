"""End-to-end regression of the two one-enqueue frame halves against their staged forms (round 6): per frame, on TWO handles fed the same inputs,
    fl_lidar_front (fused | FL_FRONT_STAGED) -> fl_map_add_points(NULL: the staged scan, in place) -> fl_vio_detect with the scan on the device
    (FL_OPT_DETECT_FUSED 1 | 0: the staged fallback that materialises the clouds)
over a walk with rendered images; every few frames something awkward happens: a raw cloud out of time order, a scan nowhere near the visual map (nothing
selected), fl_map_compact, a raw size that changes. After EVERY call the two handles must agree bit for bit: state, covariance, ImuProcess members, scan size,
the three detect counts, per-patch errors; at the end the LiDAR map and the whole visual map.  usage: fuzz_pipeline.py [seed] [frames]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlivo  # noqa
from fast_livo_amd import capi, synth

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = np.random.default_rng(seed)
scene = synth.make_scene()
lio0 = synth.make_lio_frame(20000, scene=scene)
vf0 = synth.make_vio_frame(8, lio0, max_iterations=4)
cfg = capi.config_from_frames(lio0, vf0, max_iterations=4)
hs = [capi.Handle(cfg), capi.Handle(cfg)]
hs[1].set_option(capi.FL_OPT_DETECT_FUSED, 0)
for h in hs:
    h.map_set_points(scene.map_xyz, 0.5)
    h.vmap_clear(40)
Rci = vf0.Rcl @ lio0.R_LI.T
Pci = vf0.Rcl @ (-lio0.R_LI.T @ lio0.t_LI) + vf0.Pcl
bad = 0
events = []


def same(tag, a, b):
    global bad
    if a != b:
        bad += 1
        print("MISMATCH", tag)


R_t, p_t = lio0.R_true.copy(), lio0.p_true.copy()
xs = [capi.State18.make(R_t, p_t, lio0.vel, lio0.bg, lio0.ba, lio0.grav, lio0.cov18) for _ in range(2)]
prs = None
for k in range(frames):
    raw = int(rng.choice([12000, 20000, 31000]))
    R_t = R_t @ synth.exp_so3(np.array([0.0, 0.0, 0.004]))
    p_t = p_t + np.array([0.02, 0.012, 0.0])
    lio = synth.make_lio_frame(raw, scene=scene, seed=synth.SEED + k)
    f = synth.make_imu_frame(raw, n_imu=20, lio=lio, quiet=True, seed=synth.SEED + k)
    f.pts_xyzt[:, :3] = lio.body_xyz
    what = rng.random()
    if what < 0.15:
        f.pts_xyzt = np.ascontiguousarray(f.pts_xyzt[rng.permutation(raw)]); events.append((k, "unsorted"))
    if prs is None:
        prs = [capi.imu_proc_from_frame(f) for _ in range(2)]
    # (the synthetic frames all start from the scene's prior: the walk is in the camera half; the LiDAR half is exercised frame after frame on carried handles)
    xl = [capi.state18_from_frame(lio) for _ in range(2)]
    pl = [capi.imu_proc_from_frame(f) for _ in range(2)]
    out = []
    for i, h in enumerate(hs):
        info, m = h.lidar_front(pl[i], xl[i], f.imu, f.pcl_beg_time, f.pcl_end_time, f.pts_xyzt, 0.2, staged=(i == 1))
        out.append((bytes(xl[i]), bytes(pl[i]), m, info.iterations, info.effct_feat_num, info.status))
    same(f"frame {k} lidar_front", out[0], out[1])
    if 0.15 <= what < 0.3:
        for h in hs:
            h.map_compact()
        events.append((k, "compact"))
    for h in hs:
        h.map_add_points(None, 0.25, want_info=False)
    # camera: an image rendered at the pose the LiDAR state says (so that patches track); the scan stays on the device
    Rs = np.array(xl[0].rot).reshape(3, 3); ps = np.array(xl[0].pos[:])
    Rc_t, Pc_t = synth.cam_pose(vf0.Rcl, vf0.Pcl, lio0.R_LI, lio0.t_LI, Rs, ps)
    img = synth.render_image(scene, vf0.cam, Rc_t, Pc_t, seed=k)
    det = []
    for i, h in enumerate(hs):
        xc = xl[i].copy()
        c = h.vio_detect(img, None, None, Rci, Pci, xc, k, outlier_threshold=3000.0)
        e = h.vio_get_errors(c[0]).tobytes() if c[0] > 0 else b""
        det.append((c, bytes(xc), e))
    same(f"frame {k} detect", det[0], det[1])
n0, n1 = hs[0].vmap_size(), hs[1].vmap_size()
same("visual map size", n0, n1)
for i in range(min(n0, n1)):
    p, q = hs[0].vmap_get_point(i), hs[1].vmap_get_point(i)
    if not (np.array_equal(p[0], q[0]) and p[1] == q[1] and len(p[2]) == len(q[2]) and all(bytes(a) == bytes(b) for a, b in zip(p[2], q[2]))):
        bad += 1; print("MISMATCH visual map point", i); break
same("lidar map", hs[0].map_get_points().tobytes(), hs[1].map_get_points().tobytes())
tracked = det[0][0]
for h in hs:
    h.close()
print(json.dumps({"frames": frames, "mismatches": bad, "events": events[:12], "last_detect_counts": list(tracked), "visual_map_points": n0}))
