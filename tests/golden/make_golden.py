#!/usr/bin/env python3
"""Generates tests/golden/*.json from the CPU oracle on small seeded cases and cross-checks the
LIO / VIO cases against the independent numpy restatement (oracle/np_oracle.py) before writing.

Generated from the oracle.  Since round 4 the reference's own text runs (oracle/ref_eigen, over a stand-in for Eigen's API), and
tests/test_ref_eigen_cpu.py::test_committed_fixtures_are_outputs_of_the_reference_text re-derives the committed numbers of the VIO
level, the Mode-23 update, the undistortion, the patch selection and the visual-map sequence from it; the k-NN / map fixtures belong to
the ikd-Tree (pinned by oracle/ref_ikdtree), the VoxelGrid fixture to PCL (absent: unpinned).
Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fastlivo  # noqa: E402,F401
from fast_livo_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from oracle import np_oracle as npo  # noqa: E402
import make_golden_lib as mg  # noqa: E402

scene = synth.make_scene()
out = {}
for name in ("lio18_iter", "vio_level", "ikfom_update", "knn5", "voxel_grid", "imu_undistort", "vio_select", "map_update", "vmap_sequence"):
    out[name] = getattr(mg, "run_" + name)(orc, scene)

# cross-check LIO against numpy
fr = synth.make_lio_frame(1500, scene=scene)
nbr, valid = synth.knn5(scene, fr.world_at(fr.R_prior, fr.p_prior))
rest = np.concatenate([fr.vel, fr.bg, fr.ba, fr.grav])
sol, HTH, HTz, eff, nv, _ = npo.lio18_iterate(fr.R_prior, fr.p_prior, rest, fr.R_prior, fr.p_prior, rest, fr.cov18,
                                                fr.body_xyz, nbr, valid, fr.R_LI, fr.t_LI, fr.laser_point_cov)
g = out["lio18_iter"]
assert abs(int(eff.sum()) - g["neff"][0]) <= 2, (int(eff.sum()), g["neff"])
assert np.abs(sol - np.array(g["solution"])).max() <= 1e-5 * max(1.0, np.abs(sol).max()), "LIO numpy cross-check failed"
print("LIO cross-check |d_c - d_numpy| =", np.abs(sol - np.array(g["solution"])).max())

for name, d in out.items():
    with open(os.path.join(ROOT, "tests", "golden", name + ".json"), "w") as f:
        json.dump(d, f)
    print("wrote", name)
