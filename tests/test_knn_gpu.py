"""GPU parity of the device k-NN against the brute-force oracle (oracle/orc_knn.c: the
ikd-Tree's float distance, ascending, ties by lower map index) and of the all-device frame drivers."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _handle(capi, synth, fr, max_iter=10):
    return capi.Handle(capi.config_from_frames(fr, max_iterations=max_iter))


@pytest.mark.parametrize("n,cell", [(1, 0.5), (777, 0.5), (20000, 0.5), (5000, 0.3), (5000, 1.2), (300, 0.02), (300, 0.0008)])   # (the last two: the smallest cell edge the library uses, and one below it -- raised to 2 cm)
def test_search_is_exact(gpu_lib, oracle_lib, scene, n, cell):
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(n, scene=scene)
    h = _handle(capi, synth, fr)
    h.map_set_points(scene.map_xyz, cell)
    x = capi.state18_from_frame(fr)
    h.lio_set_points(fr.body_xyz)
    h.lio_begin18(x, x)
    nbr_g, valid_g = h.lio_search18(n)
    world = h.lio_get_world_points(n)
    nbr_o, sq_o, valid_o, idx_o = orc.knn5_bruteforce(scene.map_xyz, world)
    assert np.array_equal(valid_g, valid_o)
    ok = valid_o != 0
    assert np.array_equal(nbr_g[ok], nbr_o[ok]), f"{int((np.abs(nbr_g[ok] - nbr_o[ok]).reshape(ok.sum(), -1).max(1) > 0).sum())} rows differ"
    # and it agrees with the cKDTree stand-in used everywhere else
    nb2, va2 = synth.knn5(scene, world)
    assert np.array_equal(valid_g, va2)
    h.close()


def test_search_sparse_map_and_far_queries(gpu_lib, oracle_lib, scene):
    """Few map points / queries outside the map: < 5 neighbours or sqdist[4] > 5 -> invalid, identically."""
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    rng = np.random.default_rng(5)
    fr = synth.make_lio_frame(3000, scene=scene)
    sparse = scene.map_xyz[rng.choice(len(scene.map_xyz), 300, replace=False)]
    h = _handle(capi, synth, fr)
    x = capi.state18_from_frame(fr)
    h.lio_set_points(fr.body_xyz)
    h.lio_begin18(x, x)
    for m in (sparse, sparse[:3], scene.map_xyz + np.float32(50.0)):
        h.map_set_points(m, 0.5)
        nbr_g, valid_g = h.lio_search18(fr.n)
        world = h.lio_get_world_points(fr.n)
        nbr_o, sq_o, valid_o, _ = orc.knn5_bruteforce(m, world)
        assert np.array_equal(valid_g, valid_o)
        ok = valid_o != 0
        assert np.array_equal(nbr_g[ok], nbr_o[ok])
    h.close()


@pytest.mark.parametrize("n,max_iter", [(5000, 3), (50000, 10)])
def test_all_device_frame_matches_oracle(gpu_lib, oracle_lib, scene, n, max_iter):
    """fl_lio_frame18_dev: search + passes + covariance update without leaving the device, against the
    CPU frame loop driven by the brute-force k-NN."""
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(n, scene=scene)

    def knn(w):
        nb, _, va, _ = orc.knn5_bruteforce(scene.map_xyz, w)
        return nb, va
    xo = orc.state18_from_frame(fr)
    ro = orc.lio18_frame(xo, fr.body_xyz, fr.R_LI, fr.t_LI, fr.laser_point_cov, max_iter, knn)
    h = _handle(capi, synth, fr, max_iter)
    h.map_set_points(scene.map_xyz, 0.5)
    xg = capi.state18_from_frame(fr)
    info = h.lio_frame18_dev(xg, fr.body_xyz)
    assert info.iterations == ro["out"].iterations
    assert info.effct_feat_num == ro["out"].effct_feat_num
    assert info.status == 0 and info.stop == 1
    assert np.abs(xg.vec() - xo.vec()).max() <= 1e-9
    assert np.abs(xg.cov_np() - xo.cov_np()).max() <= 1e-12
    # and equal to the host-kNN driver of the same library
    xh = capi.state18_from_frame(fr)
    h.lio_frame18(xh, fr.body_xyz, knn)
    assert np.abs(xg.vec() - xh.vec()).max() <= 1e-12
    h.close()


def test_all_device_ikfom_update_matches_oracle(gpu_lib, oracle_lib, scene):
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    n, max_iter = 20000, 4
    fr = synth.make_lio_frame(n, scene=scene)

    def knn(w):
        nb, _, va, _ = orc.knn5_bruteforce(scene.map_xyz, w)
        return nb, va
    xo = orc.state23_from_frame(fr, synth.quat_from_R)
    Po = fr.cov23.copy()
    ro = orc.ikfom_update(xo, Po, fr.body_xyz, 0.001, max_iter, knn)
    h = _handle(capi, synth, fr, max_iter)
    h.map_set_points(scene.map_xyz, 0.5)
    xg = capi.state23_from_frame(fr)
    Pg = fr.cov23.copy()
    info = h.ikfom_update_iterated_dev(xg, Pg, fr.body_xyz, 0.001)
    assert info.iterations == ro["out"].iterations
    assert info.effct_feat_num == ro["out"].effct_feat_num
    assert np.abs(xg.vec() - xo.vec()).max() <= 1e-9
    assert np.abs(Pg - Po).max() <= 1e-10
    h.close()


def test_exact_ties_and_far_from_origin(gpu_lib, oracle_lib, scene):
    """Lattice map + queries on lattice midpoints: many exactly equal float distances -> the lower map index wins, as in
    oracle/orc_knn.c; duplicated map points; and the same scene 3 km from the origin (cell assignment margin)."""
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(2000, scene=scene)
    h = _handle(capi, synth, fr)
    x = capi.state18_from_frame(fr)
    g = np.arange(-8, 8, dtype=np.float32) * np.float32(0.25)
    lattice = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    lattice = np.concatenate([lattice, lattice[::7]])                      # duplicates: equal coordinates, different indices
    rng = np.random.default_rng(11)
    q = (rng.integers(-6, 6, (3000, 3)).astype(np.float32) * np.float32(0.25) + np.float32(0.125))     # cell centres: 8-way ties
    q[::3] -= np.float32(0.125)                                             # on lattice points: 6-way ties at the next shell
    # queries are handed over as "body" points of an identity pose
    xi = capi.State18.make(np.eye(3), np.zeros(3), np.zeros(3), np.zeros(3), np.zeros(3), [0, 0, -9.81], np.eye(18) * 1e-3)
    cfg = capi.make_config(np.eye(3), np.zeros(3), synth.AVIA_RCL, synth.AVIA_PCL, dict(synth.PINHOLE, d=(0.0,) * 5))
    hi = capi.Handle(cfg)
    for off in (np.float32(0.0), np.float32(3000.0)):
        m = lattice + off
        hi.map_set_points(m, 0.5)
        hi.lio_set_points(q + off); hi.lio_begin18(xi, xi)
        nbr_g, valid_g = hi.lio_search18(q.shape[0])
        world = hi.lio_get_world_points(q.shape[0])
        assert np.array_equal(world, q + off)
        nbr_o, sq_o, valid_o, idx_o = orc.knn5_bruteforce(m, world)
        assert np.array_equal(valid_g, valid_o) and valid_o.all()
        assert np.array_equal(nbr_g, nbr_o)
        if off == 0:
            assert (np.diff(sq_o, axis=1) == 0).mean() > 0.3                # the ties are really there
    hi.close(); h.close()


def test_frame_without_any_valid_neighbour(gpu_lib, oracle_lib, scene):
    """Map 50 m away from the scan: every point is rejected (sqdist[4] > 5), no effective measurement. The reference's loop then
    solves with H = 0 (prior pull only; here state_propagat == state, so the state stays put) and stops on the convergence rule;
    the device must do the same, flag status bit 4 and leave the covariance unchanged."""
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(3000, scene=scene)
    far = scene.map_xyz + np.float32(50.0)

    def knn(w):
        nb, _, va, _ = orc.knn5_bruteforce(far, w)
        return nb, va
    xo = orc.state18_from_frame(fr)
    ro = orc.lio18_frame(xo, fr.body_xyz, fr.R_LI, fr.t_LI, fr.laser_point_cov, 10, knn)
    h = _handle(capi, synth, fr, 10)
    h.map_set_points(far, 0.5)
    xg = capi.state18_from_frame(fr)
    x0 = capi.state18_from_frame(fr)
    info = h.lio_frame18_dev(xg, fr.body_xyz)
    assert info.effct_feat_num == 0 == ro["out"].effct_feat_num
    assert info.iterations == ro["out"].iterations
    assert info.status & 4
    assert np.abs(xg.vec() - xo.vec()).max() <= 1e-12 and np.abs(xg.vec() - x0.vec()).max() <= 1e-12
    assert np.abs(xg.cov_np() - xo.cov_np()).max() <= 1e-15
    mask, _ = h.lio_get_selection(fr.n)
    assert mask.sum() == 0
    h.close()


def test_frame_timing_hook_splits_match_and_solve(gpu_lib, scene):
    """fl_get_frame_timing: the reference's match_time / solve_time (laserMapping.cpp:1604,1729) for the all-device frame."""
    capi = gpu_lib
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(30000, scene=scene)
    h = capi.Handle(capi.config_from_frames(fr, max_iterations=10))
    h.map_set_points(scene.map_xyz, 0.5)
    h.set_timing(True)
    for _ in range(3):
        x = capi.state18_from_frame(fr)
        info = h.lio_frame18_dev(x, fr.body_xyz)
    t = h.frame_timing()
    assert info.status == 0 and t["searches"] == 2
    assert 0.0 < t["match_ms"] < 1.0 and 0.0 < t["solve_ms"] < 1.0 and abs(t["total_ms"] - t["match_ms"] - t["solve_ms"]) < 1e-3
    print(f"\n[frame timing] match {t['match_ms'] * 1e3:.1f} us, solve {t['solve_ms'] * 1e3:.1f} us")
    h.close()


@pytest.mark.parametrize("cell", [0.5, 0.3, 1.2])
def test_incremental_search_is_the_full_search(gpu_lib, oracle_lib, scene, cell):
    """A search over a scan and a map an earlier search covered walks only the cells within reach of the earlier winners
    (knn_kernels.h, FL_OPT_INCR_SEARCH): same neighbours, bit for bit, as the full walk and as the brute-force oracle --
    for pose changes from a fraction of a millimetre (the rematch of a frame, laserMapping.cpp:1700-1705) to metres
    (bound beyond one cell: the query falls back to the full walk)."""
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    from oracle import np_oracle as npo
    n = 20000
    fr = synth.make_lio_frame(n, scene=scene)
    rng = np.random.default_rng(42)
    h_inc = _handle(capi, synth, fr)
    h_full = _handle(capi, synth, fr)
    h_full.set_option(capi.FL_OPT_INCR_SEARCH, 0)
    for h in (h_inc, h_full):
        h.map_set_points(scene.map_xyz, cell)
        h.lio_set_points(fr.body_xyz)
    x0 = capi.state18_from_frame(fr)
    for h in (h_inc, h_full):
        h.lio_begin18(x0, x0)
        h.lio_search18(n)                      # the first search of the scan: full in both, leaves the winners behind
    for d_pos, d_rot in ((0.0, 0.0), (2e-4, 1e-5), (3e-3, 2e-4), (3e-2, 2e-3), (0.3, 2e-2), (3.0, 0.2), (1e-3, 1e-4)):
        R = fr.R_prior @ npo.Exp(rng.standard_normal(3) * d_rot)
        p = fr.p_prior + rng.standard_normal(3) * d_pos
        x = capi.State18.make(R, p, fr.vel, fr.bg, fr.ba, fr.grav, fr.cov18)
        out = []
        for h in (h_inc, h_full):
            h.lio_begin18(x, x)
            nbr, valid = h.lio_search18(n)
            out.append((nbr, valid, h.lio_get_world_points(n)))
        (nbr_i, val_i, world), (nbr_f, val_f, world_f) = out
        assert np.array_equal(world, world_f)
        assert np.array_equal(val_i, val_f)
        assert np.array_equal(nbr_i, nbr_f), f"d_pos {d_pos}: {int((nbr_i != nbr_f).any(axis=(1, 2)).sum())} queries differ from the full walk"
        nbr_o, sq_o, valid_o, _ = orc.knn5_bruteforce(scene.map_xyz, world)
        assert np.array_equal(val_i, valid_o)
        ok = valid_o != 0
        assert np.array_equal(nbr_i[ok], nbr_o[ok])
    # another map under the same scan: the kept winners index the old map -- the next search must not use them
    rng2 = np.random.default_rng(3)
    thin = scene.map_xyz[rng2.choice(len(scene.map_xyz), len(scene.map_xyz) // 3, replace=False)]
    h_inc.map_set_points(thin, cell)
    nbr, valid = h_inc.lio_search18(n)
    nbr_o, _, valid_o, _ = orc.knn5_bruteforce(thin, h_inc.lio_get_world_points(n))
    assert np.array_equal(valid, valid_o) and np.array_equal(nbr[valid_o != 0], nbr_o[valid_o != 0])
    # and incremental over the thin map (winners often farther than a cell: mixed pruned / full queries in one wave)
    x = capi.State18.make(fr.R_prior, fr.p_prior + np.array([0.004, -0.003, 0.002]), fr.vel, fr.bg, fr.ba, fr.grav, fr.cov18)
    h_inc.lio_begin18(x, x)
    nbr, valid = h_inc.lio_search18(n)
    nbr_o, _, valid_o, _ = orc.knn5_bruteforce(thin, h_inc.lio_get_world_points(n))
    assert np.array_equal(valid, valid_o) and np.array_equal(nbr[valid_o != 0], nbr_o[valid_o != 0])
    # another scan under the same map
    fr2 = synth.make_lio_frame(n // 2, scene=scene, seed=9) if "seed" in synth.make_lio_frame.__code__.co_varnames else fr
    h_inc.lio_set_points(fr2.body_xyz[: n // 2][::-1].copy())
    h_inc.lio_begin18(x0, x0)
    nbr, valid = h_inc.lio_search18(n // 2)
    nbr_o, _, valid_o, _ = orc.knn5_bruteforce(thin, h_inc.lio_get_world_points(n // 2))
    assert np.array_equal(valid, valid_o) and np.array_equal(nbr[valid_o != 0], nbr_o[valid_o != 0])
    h_inc.close(); h_full.close()
