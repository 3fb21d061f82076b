"""Randomised cross-check of the VIO side (run on the GPU box): random patch counts / iteration caps, ComputeJ vs the oracle
(state 1e-9, per-patch errors bit for bit), selection vs the oracle (bit-identical). FL_FUZZ_WIDE=2: ComputeJ on the one-patch-per-lane
producers (FL_OPT_VIO_WIDE). FL_FUZZ_SCALE=1: patch counts drawn from 16 384 ... 100 000 (the automatic switch to those producers)."""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fastlivo  # noqa
from fast_livo_amd import capi, synth
from oracle import oracle as orc
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
bad = 0
T = int(sys.argv[2]) if len(sys.argv) > 2 else 20
for trial in range(T):
    m = int(rng.choice([1, 2, 7, 8, 9, 63, 64, 65, 300, 1000, 2040, 2041, 2600]))
    max_iter = int(rng.integers(1, 11))
    if os.environ.get("FL_FUZZ_SCALE"):          # round 6: sizes at which the one-patch-per-lane producers switch on BY THEMSELVES (default option),
        m = int(rng.integers(16384, 100001))     # distinct patches; the oracle's patch loop over the host's cores (bit-identical for any thread count)
        max_iter = int(rng.integers(1, 5))
        orc.lib().orc_vio_set_threads(min(os.cpu_count() or 1, 64))
    seed = int(rng.integers(1 << 20))
    lio = synth.make_lio_frame(500, seed=synth.SEED + seed % 7)
    vf = synth.make_vio_frame(m, lio, max_iterations=max_iter, patch_seed=seed)
    h = capi.Handle(capi.config_from_frames(lio, vf, max_iterations=max_iter))
    if os.environ.get("FL_FUZZ_WIDE"):           # e.g. 2: every pass on the one-patch-per-lane producers (FL_OPT_VIO_WIDE)
        h.set_option(capi.FL_OPT_VIO_WIDE, int(os.environ["FL_FUZZ_WIDE"]))
    xg = capi.state18_from_frame(lio); xp = capi.state18_from_frame(lio)
    h.vio_set_frame(vf.img); h.vio_set_patches(vf.ref_patch, vf.pos, vf.search_level)
    ig = h.vio_compute_j(xg, xp)
    eg = h.vio_get_errors(m)
    xo = orc.state18_from_frame(lio); xq = orc.state18_from_frame(lio)
    ro = orc.vio_compute_j(vf, xo, xq)
    e1 = np.abs(xg.vec() - xo.vec()).max(); e2 = np.abs(xg.cov_np() - xo.cov_np()).max()
    e3 = np.abs(eg - ro["errors"]).max() / max(1.0, np.abs(ro["errors"]).max())
    its = [int(i.iterations) for i in ig]; ito = [int(o.iterations) for o in ro["outs"]]
    # an accept test decided inside the rounding noise of the reference's float running sum may go either way (status bit 16)
    fragile = any(i.status & 16 for i in ig)       # the exact replay of the reference's float error sum ran
    ok = e1 <= 1e-9 and e2 <= 1e-11 and np.array_equal(eg.view(np.uint32), ro["errors"].view(np.uint32)) and its == ito
    # selection
    k = int(rng.choice([1, 3, 64, 500]))
    sf = synth.make_select_frame(k, seed=seed, n_keyframes=int(rng.integers(1, 4)))
    hs = capi.Handle(capi.config_from_frames(sf.lio, sf.vio))
    ids = [hs.vio_add_keyframe(kf) for kf in sf.keyframes]
    hs.vio_set_frame(sf.vio.img)
    cfg = orc.vio_config(sf.vio)
    depth = orc.vio_depth_image(cfg, sf.Rcw, sf.Pcw, sf.scan_world)
    rr = orc.vio_select(cfg, sf.Rcw, sf.Pcw, sf.vio.img, sf.keyframes, depth, orc.patch_candidates(sf), outlier_threshold=300.0)
    dd = hs.vio_select_patches(sf.Rcw, sf.Pcw, sf.scan_world, capi.patch_candidates(sf, ids), outlier_threshold=300.0)
    ok2 = np.array_equal(dd["reason"], rr["reason"]) and np.array_equal(dd["patches"].view(np.uint32), rr["patches"].view(np.uint32)) and \
        np.array_equal(dd["errors"].view(np.uint32), rr["errors"].view(np.uint32))
    if not (ok and ok2):
        bad += 1
        print("MISMATCH", dict(m=m, max_iter=max_iter, e1=e1, e2=e2, e3=e3, its=its, ito=ito, fragile=bool(fragile), ok2=bool(ok2), k=k))
    h.close(); hs.close()
print(json.dumps({"trials": T, "mismatches": bad}))
