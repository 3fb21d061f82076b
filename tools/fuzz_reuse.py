"""One long-lived handle driven through a random sequence of differently sized frames / maps / patch sets, each result compared
bit for bit with a fresh handle doing only that call (state leaking between calls: stale records, flags, buffer growth)."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlivo  # noqa
from fast_livo_amd import capi, synth
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
T = int(sys.argv[2]) if len(sys.argv) > 2 else 40
scene = synth.make_scene()
base = synth.make_lio_frame(1000, scene=scene)
vf0 = synth.make_vio_frame(8, base)
cfg = capi.config_from_frames(base, vf0, max_iterations=6)
H = capi.Handle(cfg)
maps = [scene.map_xyz, scene.map_xyz[::3].copy(), scene.map_xyz[: 20000].copy()]
bad = 0
for t in range(T):
    op = int(rng.integers(0, 4))
    fresh = capi.Handle(cfg)
    if op == 0:      # all-device LIO frame
        n = int(rng.choice([3, 500, 4096, 30000, 65280, 90000])); mp = maps[int(rng.integers(3))]; cell = float(rng.choice([0.4, 0.5, 1.0]))
        fr = synth.make_lio_frame(n, scene=scene, point_seed=int(rng.integers(1 << 30)))
        outs = []
        for h in (H, fresh):
            h.map_set_points(mp, cell); x = capi.state18_from_frame(fr); info = h.lio_frame18_dev(x, fr.body_xyz)
            m, v = h.lio_get_selection(n); outs.append(bytes(x) + m.tobytes() + v.tobytes() + bytes([info.iterations]))
    elif op == 1:    # host-kNN passes, explicit iterate
        n = int(rng.choice([10, 2000, 50000]))
        fr = synth.make_lio_frame(n, scene=scene, point_seed=int(rng.integers(1 << 30)))
        nbr, valid = synth.knn5(scene, fr.world_at(fr.R_prior, fr.p_prior)); c = int(rng.integers(1, 7))
        outs = []
        for h in (H, fresh):
            x = capi.state18_from_frame(fr); h.lio_set_points(fr.body_xyz); h.lio_begin18(x, x); h.lio_set_neighbours(nbr, valid)
            h.lio_iterate18(c, capi.FL_ITER_FORCE | capi.FL_ITER_KEEP_NORMVEC, want_info=False)
            m, v = h.lio_get_selection(n); outs.append(h.lio_get_state18().vec().tobytes() + m.tobytes() + v.tobytes())
    elif op == 2:    # VIO ComputeJ with a new patch count
        mm = int(rng.choice([1, 9, 700, 2000, 2300]))
        vf = synth.make_vio_frame(mm, base, max_iterations=6, patch_seed=int(rng.integers(1 << 30)))
        outs = []
        for h in (H, fresh):
            x = capi.state18_from_frame(base); xp = capi.state18_from_frame(base)
            h.vio_set_frame(vf.img); h.vio_set_patches(vf.ref_patch, vf.pos, vf.search_level); h.vio_compute_j(x, xp)
            outs.append(bytes(x) + h.vio_get_errors(mm).tobytes())
    else:            # voxel filter staged as scan + frame on it
        n = int(rng.choice([100, 8000, 60000])); leaf = float(rng.choice([0.15, 0.3]))
        fr = synth.make_lio_frame(n, scene=scene, point_seed=int(rng.integers(1 << 30)))
        p = np.concatenate([fr.body_xyz, np.zeros((n, 1), np.float32)], 1).astype(np.float32)
        outs = []
        for h in (H, fresh):
            h.map_set_points(scene.map_xyz, 0.5); _, k, _ = h.scan_voxel_filter(p, leaf, stage_as_scan=True, want=False)
            x = capi.state18_from_frame(fr); h.lio_frame18_dev(x, None); outs.append(bytes(x) + bytes([k % 251]))
    if outs[0] != outs[1]:
        bad += 1
        print("MISMATCH at step", t, "op", op)
    fresh.close()
print(json.dumps({"steps": T, "mismatches": bad}))
