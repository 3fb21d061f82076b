"""Map-scale sweep (round 5): the per-frame map update (fl_map_add_points = map_incremental, 20 k new points, down-sampling 0.3 m) and the
Mode-18 frame (50 k points, device k-NN) against local maps of 55 k ... 5 M points at CONSTANT density (the synthetic room tiled over a
growing area), with the in-place update (FL_OPT_MAP_INCREMENTAL 1) and with the compact-and-rebuild form (0).
    python tools/map_scale_bench.py [sizes...]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlivo  # noqa
from fast_livo_amd import capi, synth


def tiled_map(scene, k, rng):
    base = scene.map_xyz
    tiles = (k + len(base) - 1) // len(base)
    side = int(np.ceil(np.sqrt(tiles)))
    parts = []
    for t in range(tiles):
        off = np.float32([22.0 * (t % side), 22.0 * (t // side), 0.0])
        parts.append(base + off + (rng.normal(0, 0.01, base.shape).astype(np.float32) if t else 0))
    return np.ascontiguousarray(np.concatenate(parts)[:k], dtype=np.float32)


def run(sizes, n_new=20000, ds=0.3, reps=12, scene=None):
    rng = np.random.default_rng(0)
    scene = scene or synth.make_scene()
    fr = synth.make_lio_frame(50000, scene=scene)
    out = []
    for K in sizes:
        m0 = tiled_map(scene, K, rng)
        row = {"map_points": int(len(m0))}
        for incr in (1, 0):
            h = capi.Handle(capi.config_from_frames(fr, max_iterations=10))
            h.set_option(capi.FL_OPT_MAP_INCREMENTAL, incr)
            h.map_set_points(m0, 0.5)
            h.set_timing(True)
            wall, dev = [], []
            for r in range(reps):
                new = (scene.map_xyz[rng.integers(0, len(scene.map_xyz), n_new)] + rng.normal(0, 0.03, (n_new, 3))).astype(np.float32)
                h.sync()
                t0 = time.perf_counter()
                h.map_add_points(new, ds, want_info=False)
                h.sync()
                wall.append(time.perf_counter() - t0); dev.append(h.last_kernel_ms())
            x = capi.state18_from_frame(fr)
            ft = []
            for r in range(8):
                xx = x.copy(); h.sync()
                t0 = time.perf_counter(); h.lio_frame18_dev(xx, fr.body_xyz); ft.append(time.perf_counter() - t0)
            comp = None
            if incr:          # the worst-case frame of the in-place form: the O(map) compaction + re-index its arrays ask for every capacity / points-per-frame
                ct = []       # frames (fl_map_compact = that step on demand), timed behind an update that left tombstones
                for r in range(3):
                    new = (scene.map_xyz[rng.integers(0, len(scene.map_xyz), n_new)] + rng.normal(0, 0.03, (n_new, 3))).astype(np.float32)
                    h.map_add_points(new, ds, want_info=False); h.sync()
                    t0 = time.perf_counter(); h.map_compact(); h.sync(); ct.append(time.perf_counter() - t0)
                comp = round(float(np.median(ct)) * 1e3, 4)
            key = "in_place" if incr else "rebuild"
            row[key] = {"map_add_wall_ms": round(float(np.median(wall[2:])) * 1e3, 4), "map_add_device_ms": round(float(np.median(dev[2:])), 4),
                        "lio_frame_ms": round(float(np.median(ft[2:])) * 1e3, 4)}
            if comp is not None:
                row[key]["compaction_ms"] = comp
                row[key]["worst_case_map_add_wall_ms"] = round(comp + row[key]["map_add_wall_ms"], 4)
            h.close()
        out.append(row)
    return out


if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [55000, 500000, 2000000, 5000000]
    print(json.dumps(run(sizes)))
