"""Randomised parity of the device map maintenance against the sequential oracle (oracle/orc_map.c): random map / scan sizes,
down-sampling sizes (powers of two and not), lattice data with exact ties, interleaved box deletions, automatic and fixed k-NN cell
sizes, several updates per handle. usage: fuzz_map.py [trials]"""
import json
import sys

import numpy as np

sys.path.insert(0, "/root/repo")
import fastlivo  # noqa: F401,E402
from fast_livo_amd import capi, synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(2024)
scene = synth.make_scene()
fr = synth.make_lio_frame(64, scene=scene)
h = capi.Handle(capi.config_from_frames(fr))
mism, amb_cases, updates = 0, 0, 0
for tr in range(trials):
    n0 = int(rng.integers(1, 30000))
    span = float(rng.choice([3.0, 10.0, 40.0]))
    m = rng.uniform(-span, span, (n0, 3)).astype(np.float32)
    lattice = rng.random() < 0.4
    if lattice:
        m[: n0 // 2] = np.round(m[: n0 // 2] * 8) / 8
    cell = float(rng.choice([0.0, 0.3, 0.5, 1.0]))
    h.map_set_points(m, cell)
    cur = m.copy()
    for step in range(int(rng.integers(1, 5))):
        if rng.random() < 0.25 and len(cur):
            lo = rng.uniform(-span, span, 3)
            box = np.concatenate([lo, lo + rng.uniform(0.5, span, 3)]).astype(np.float32)[None]
            gi = h.map_delete_boxes(box)
            cur, oi = orc.map_delete_boxes(cur, box)
            ok = gi.n_after == oi.n_after and np.array_equal(h.map_get_points(), cur)
            amb = 0
        else:
            n1 = int(rng.integers(1, 8000))
            new = rng.uniform(-span, span, (n1, 3)).astype(np.float32)
            if lattice:
                new[: n1 // 2] = np.round(new[: n1 // 2] * 8) / 8
            ds = float(rng.choice([0.0, 0.125, 0.25, 0.5, 1.0, 0.1, 0.15, 0.2, 0.3, 0.37, 0.7]))
            gi = h.map_add_points(new, ds)
            want, oi = orc.map_add_points(cur, new, ds)
            amb = oi.n_ambiguous
            # n_ambiguous: the oracle counts EVERY point (old and new) whose box depends on a rounding; the in-place form (FL_OPT_MAP_INCREMENTAL,
            # round 5) only looks at the boxes an update touches, so it counts the new points and the old ones that could matter to this update.
            # The contract (include/fastlivo_hip.h): results may differ from the sequential reference ONLY where the device's count is not 0.
            ok = gi.n_ambiguous <= oi.n_ambiguous
            got = h.map_get_points().copy()
            if gi.n_ambiguous == 0:
                ok = ok and np.array_equal(got, want)
            if amb:
                amb_cases += 1
            cur = got if not np.array_equal(got, want) else want
        updates += 1
        if not ok:
            mism += 1
            print("MISMATCH", dict(trial=tr, step=step, n0=n0, lattice=lattice, cell=cell))
        if len(cur) == 0:
            break
h.close()
print(json.dumps({"trials": trials, "updates": updates, "mismatches": mism, "updates_with_ambiguous_points": amb_cases}))
