"""Run the device k-NN search kernel a few times (for rocprofv3 counter collection)."""
import argparse, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlivo  # noqa: F401
from fast_livo_amd import capi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=50000)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--cell", type=float, default=0.5)
ap.add_argument("--order", choices=("as_is", "voxel", "random"), default="as_is", help="order of the scan points: as synth makes them, sorted by voxel index as pcl::VoxelGrid emits its centroids, shuffled")
ap.add_argument("--leaf", type=float, default=0.5)
ap.add_argument("--move", type=float, default=-1.0, help=">= 0: also time a second search after the pose moved by this many metres (incremental on / off)")
ap.add_argument("--only-incremental", action="store_true", help="with --move: time the incremental form only (rocprofv3 --stats of this run = the rematch search kernel)")
a = ap.parse_args()
fr = synth.make_lio_frame(a.points)
if a.order != "as_is":
    b = fr.body_xyz
    if a.order == "random":
        perm = np.random.default_rng(1).permutation(len(b))
    else:      # pcl::VoxelGrid: idx = ijk0 + ijk1 * dx + ijk2 * dx * dy, points emitted in ascending idx
        ijk = np.floor(b / a.leaf).astype(np.int64); ijk -= ijk.min(axis=0); d = ijk.max(axis=0) + 1
        perm = np.argsort(ijk[:, 0] + ijk[:, 1] * d[0] + ijk[:, 2] * d[0] * d[1], kind="stable")
    fr.body_xyz = np.ascontiguousarray(b[perm])
h = capi.Handle(capi.config_from_frames(fr, max_iterations=10))
h.map_set_points(fr.scene.map_xyz, a.cell)
x = capi.state18_from_frame(fr); h.lio_set_points(fr.body_xyz); h.lio_begin18(x, x)
h.set_timing(True)
ks = []
for _ in range(a.reps):
    h.lio_search18(fr.n, want=False); ks.append(h.last_kernel_ms() * 1e3)
print("search_fit_us", ks)
# the rematch search of a frame: same scan, same map, the pose moved by --move metres (and --move / 10 rad)
if a.move >= 0:
    from oracle import np_oracle as npo          # Exp only (tool, not product)
    rng = np.random.default_rng(2)
    for incr in ((1,) if a.only_incremental else (1, 0)):
        h.set_option(capi.FL_OPT_INCR_SEARCH, incr)
        ks = []
        for r in range(a.reps):
            h.lio_begin18(x, x); h.lio_search18(fr.n, want=False)
            R = fr.R_prior @ npo.Exp(rng.standard_normal(3) * a.move / 10)
            x2 = capi.State18.make(R, fr.p_prior + rng.standard_normal(3) * a.move, fr.vel, fr.bg, fr.ba, fr.grav, fr.cov18)
            h.lio_begin18(x2, x2); h.lio_search18(fr.n, want=False); ks.append(h.last_kernel_ms() * 1e3)
        print("second_search_us incremental=%d move=%g" % (incr, a.move), ks)
