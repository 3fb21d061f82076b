"""The speculating accept of ComputeJ (FL_OPT_VIO_SPECULATE, solve18.h vio_spec_confirm) under fuzz, on the GPU box with the DEBUG library
(it counts: fragile accepts that went ahead / confirmed / rolled back): random frames, patch counts and iteration caps, with the prior
perturbed so that consecutive errors lie close together; the speculating form, the waiting form and the CPU oracle must agree in every
bit of the state, the covariance, the per-patch errors and the iteration counts.
    python tools/fuzz_vio_spec.py [seed] [trials]"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlivo  # noqa
from fast_livo_amd import capi, synth
from oracle import oracle as orc
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
T = int(sys.argv[2]) if len(sys.argv) > 2 else 40
bad, tot = 0, np.zeros(5, np.int64)
for trial in range(T):
    m = int(rng.choice([300, 700, 1000, 2000, 2040]))
    max_iter = int(rng.integers(3, 11))
    seed = int(rng.integers(1 << 20))
    lio = synth.make_lio_frame(500, seed=synth.SEED + seed % 11)
    noise = float(rng.choice([0.5, 2.0, 6.0]))
    vf = synth.make_vio_frame(m, lio, max_iterations=max_iter, patch_seed=seed, ref_noise=noise)
    res = []
    for spec in (2, 1, 0):           # all levels in one launch (default) / a launch per level / the waiting form
        h = capi.Handle(capi.config_from_frames(lio, vf, max_iterations=max_iter), debug=True)
        h.set_option(capi.FL_OPT_VIO_SPECULATE, spec)
        w0 = np.array(h.debug_wall(), dtype=np.int64)[2040:2045].copy()
        xg = capi.state18_from_frame(lio); xp = capi.state18_from_frame(lio)
        h.vio_set_frame(vf.img); h.vio_set_patches(vf.ref_patch, vf.pos, vf.search_level)
        ig = h.vio_compute_j(xg, xp)
        eg = h.vio_get_errors(m)
        w1 = np.array(h.debug_wall(), dtype=np.int64)[2040:2045]
        if spec == 2:
            tot += w1 - w0
            if (w1 - w0)[2]:
                print("ROLLED BACK", dict(m=m, max_iter=max_iter, seed=seed, lio_seed=seed % 11, noise=noise, counts=(w1 - w0).tolist()))
        res.append((bytes(xg), eg.copy(), [(int(i.iterations), int(i.accepted), int(i.status)) for i in ig]))
        h.close()
    xo = orc.state18_from_frame(lio); xq = orc.state18_from_frame(lio)
    ro = orc.vio_compute_j(vf, xo, xq)
    same = all(res[0][0] == r[0] and np.array_equal(res[0][1].view(np.uint32), r[1].view(np.uint32)) and res[0][2] == r[2] for r in res[1:])
    xs = np.frombuffer(res[0][0], np.float64)
    vs_orc = np.abs(xs[:24] - xo.vec()[:24]).max() <= 1e-9 and np.array_equal(res[0][1].view(np.uint32), ro["errors"].view(np.uint32)) and \
        [r[0] for r in res[0][2]] == [int(o.iterations) for o in ro["outs"]]
    if not (same and vs_orc):
        bad += 1
        print("MISMATCH", dict(m=m, max_iter=max_iter, seed=seed, same=bool(same), vs_oracle=bool(vs_orc), spec=res[0][2], per_level=res[1][2], wait=res[2][2]))
print(json.dumps({"trials": T, "mismatches": bad, "speculated": int(tot[0]), "confirmed": int(tot[1]), "rolled_back": int(tot[2]), "rolled_back_across_levels": int(tot[3]), "verdicts_carried_into_the_next_level": int(tot[4])}))
