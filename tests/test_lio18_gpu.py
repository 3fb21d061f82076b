"""GPU parity: Mode-18 LiDAR ESKF (fl_lio_*) against the CPU oracle on identical seeded inputs.

Tolerances: identical selection sets; reduced sums to 1e-12 relative;
state delta |d_gpu - d_cpu|_inf <= 1e-9 * max(1, |d_cpu|_inf).
"""
import ctypes as C

import numpy as np
import pytest

from helpers import TOL_SUMS_REL, assert_delta_close, copy_state, sums_to_HTH

pytestmark = pytest.mark.gpu


def _setup(capi, synth, scene, n, max_iter=10, seed_off=0):
    fr = synth.make_lio_frame(n, scene=scene, seed=synth.SEED + seed_off)
    cfg = capi.config_from_frames(fr, max_iterations=max_iter)
    h = capi.Handle(cfg)
    world = fr.world_at(fr.R_prior, fr.p_prior)
    nbr, valid = synth.knn5(scene, world)
    return fr, h, nbr, valid


def _dev_sums(capi, h, flags=0):
    import torch
    t = torch.zeros(capi.FL_SUMS18, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    h.lio_accumulate18(t.data_ptr(), flags)
    h.sync()
    return t.cpu().numpy(), t


@pytest.mark.parametrize("n", [1, 63, 257, 5000, 50000])
def test_single_iteration_matches_oracle(gpu_lib, oracle_lib, scene, n):
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    fr, h, nbr, valid = _setup(capi, synth, scene, n)
    xo = orc.state18_from_frame(fr)
    xpo = xo.copy()
    sel_o = valid.copy()
    ro = orc.lio18_iterate(xo, xpo, fr.body_xyz, nbr, sel_o, fr.R_LI, fr.t_LI, fr.laser_point_cov)

    xg = capi.state18_from_frame(fr)
    h.lio_set_points(fr.body_xyz)
    h.lio_begin18(xg, xg)
    h.lio_set_neighbours(nbr, valid)
    # sharded form first (does not touch the state): exposes the reduced record
    sums, _ = _dev_sums(capi, h, capi.FL_ITER_KEEP_NORMVEC)
    S, HTz = sums_to_HTH(sums)
    HTH_o = np.array(ro["out"].HTH).reshape(6, 6)
    assert int(sums[27]) == ro["out"].effct_feat_num
    if ro["out"].effct_feat_num > 0:
        assert np.abs(S - HTH_o).max() <= TOL_SUMS_REL * np.abs(HTH_o).max()
        assert np.abs(HTz - np.array(ro["out"].HTz)).max() <= TOL_SUMS_REL * max(np.abs(np.array(ro["out"].HTz)).max(), 1e-300) * 10
        assert abs(sums[28] - ro["out"].total_residual) <= 1e-12 * max(1.0, ro["out"].total_residual)
    mask, nv = h.lio_get_selection(n)
    eff_o = (sel_o != 0) & (ro["res_last"] <= 2.0)
    assert np.array_equal(mask != 0, eff_o), f"selection flips: {int((mask.astype(bool) != eff_o).sum())}"
    assert np.array_equal(nv[sel_o != 0], ro["normvec"][sel_o != 0])   # bit-identical planes and pd2

    # fused pass on a fresh frame state
    h.lio_begin18(xg, xg)
    h.lio_set_neighbours(nbr, valid)
    info = h.lio_iterate18(1, 0)
    assert info.effct_feat_num == ro["out"].effct_feat_num
    assert info.iterations == 1
    assert_delta_close(np.array(info.solution)[:18], np.array(ro["out"].solution))
    assert info.converged == ro["out"].converged
    xs = h.lio_get_state18()
    assert np.abs(xs.vec() - xo.vec()).max() <= 1e-9
    h.close()


def test_three_passes_without_search(gpu_lib, oracle_lib, scene):
    """Passes 2 and 3 reuse the staged neighbours; selection flags persist."""
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    n = 20000
    fr, h, nbr, valid = _setup(capi, synth, scene, n)
    xo = orc.state18_from_frame(fr)
    xpo = xo.copy()
    sel_o = valid.copy()
    G = np.zeros((18, 18))
    nvo = np.zeros((n, 4), dtype=np.float32)
    rl = np.zeros(n)
    xg = capi.state18_from_frame(fr)
    h.lio_set_points(fr.body_xyz)
    h.lio_begin18(xg, xg)
    h.lio_set_neighbours(nbr, valid)
    for it in range(3):
        ro = orc.lio18_iterate(xo, xpo, fr.body_xyz, nbr, sel_o, fr.R_LI, fr.t_LI, fr.laser_point_cov, G=G,
                               normvec=nvo, res_last=rl)
        info = h.lio_iterate18(1, capi.FL_ITER_FORCE | capi.FL_ITER_KEEP_NORMVEC)
        assert info.effct_feat_num == ro["out"].effct_feat_num, it
        assert_delta_close(np.array(info.solution)[:18], np.array(ro["out"].solution))
        xs = h.lio_get_state18()
        assert np.abs(xs.vec() - xo.vec()).max() <= 1e-9, it
        mask, _ = h.lio_get_selection(n)
        assert np.array_equal(mask != 0, (sel_o != 0) & (rl <= 2.0)), it
    h.close()


@pytest.mark.parametrize("n,max_iter", [(5000, 3), (50000, 10)])
def test_frame_loop_matches_oracle(gpu_lib, oracle_lib, scene, n, max_iter):
    """Whole per-frame loop incl. rematch, stop logic and covariance update (laserMapping.cpp:1504-1733)."""
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(n, scene=scene)
    knn = lambda w: synth.knn5(scene, w)  # noqa: E731
    xo = orc.state18_from_frame(fr)
    ro = orc.lio18_frame(xo, fr.body_xyz, fr.R_LI, fr.t_LI, fr.laser_point_cov, max_iter, knn)
    h = capi.Handle(capi.config_from_frames(fr, max_iterations=max_iter))
    xg = capi.state18_from_frame(fr)
    info = h.lio_frame18(xg, fr.body_xyz, knn)
    assert info.iterations == ro["out"].iterations
    assert info.effct_feat_num == ro["out"].effct_feat_num
    assert info.status == 0
    assert np.abs(xg.vec() - xo.vec()).max() <= 1e-9
    assert np.abs(xg.cov_np() - xo.cov_np()).max() <= 1e-12
    mask, nv = h.lio_get_selection(n)
    sel_o = ro["sel"] != 0
    assert np.array_equal(nv[sel_o], ro["normvec"][sel_o])
    h.close()


def test_sharded_accumulate_then_solve_equals_fused(gpu_lib, scene):
    """point-range shards are additive; solve from the summed record == fused pass."""
    capi = gpu_lib
    import torch
    from fast_livo_amd import synth
    n = 30000
    fr, h, nbr, valid = _setup(capi, synth, scene, n)
    xg = capi.state18_from_frame(fr)
    h.lio_set_points(fr.body_xyz)
    h.lio_begin18(xg, xg)
    h.lio_set_neighbours(nbr, valid)
    info_f = h.lio_iterate18(1, capi.FL_ITER_FORCE)
    x_f = h.lio_get_state18()

    total = torch.zeros(capi.FL_SUMS18, dtype=torch.float64, device="cuda")
    cut = 13000
    hs = []
    for lo, hi in ((0, cut), (cut, n)):
        hh = capi.Handle(capi.config_from_frames(fr))
        hh.lio_set_points(fr.body_xyz[lo:hi])
        hh.lio_begin18(xg, xg)
        hh.lio_set_neighbours(nbr[lo:hi], valid[lo:hi])
        t = torch.zeros(capi.FL_SUMS18, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        hh.lio_accumulate18(t.data_ptr(), capi.FL_ITER_FORCE)
        hh.sync()
        total += t
        hs.append(hh)
    torch.cuda.synchronize()
    info_s = hs[0].lio_solve18(total.data_ptr(), capi.FL_ITER_FORCE, want_info=True)
    x_s = hs[0].lio_get_state18()
    assert info_s.effct_feat_num == info_f.effct_feat_num
    assert_delta_close(np.array(info_s.solution)[:18], np.array(info_f.solution)[:18], tol=1e-11)
    assert np.abs(x_s.vec() - x_f.vec()).max() <= 1e-11
    for hh in hs:
        hh.close()
    h.close()


def test_edge_cases_invalid_and_degenerate_neighbours(gpu_lib, oracle_lib, scene):
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    n = 4096
    fr, h, nbr, valid = _setup(capi, synth, scene, n)
    # (a) collinear / identical neighbours -> NaN planes must never be selected (same as the oracle)
    nbr2 = nbr.copy()
    nbr2[:100] = nbr2[:100, :1, :]              # 5 identical points
    nbr2[100:200, :, 1:] = nbr2[100:200, :1, 1:]  # collinear along x
    valid2 = valid.copy()
    valid2[200:300] = 0                          # kNN said invalid
    xo = orc.state18_from_frame(fr)
    sel_o = valid2.copy()
    ro = orc.lio18_iterate(xo, xo.copy(), fr.body_xyz, nbr2, sel_o, fr.R_LI, fr.t_LI, fr.laser_point_cov)
    xg = capi.state18_from_frame(fr)
    h.lio_set_points(fr.body_xyz)
    h.lio_begin18(xg, xg)
    h.lio_set_neighbours(nbr2, valid2)
    info = h.lio_iterate18(1, capi.FL_ITER_KEEP_NORMVEC)
    assert info.effct_feat_num == ro["out"].effct_feat_num
    mask, _ = h.lio_get_selection(n)
    assert not mask[200:300].any()
    assert np.array_equal(mask != 0, (sel_o != 0) & (ro["res_last"] <= 2.0))
    assert_delta_close(np.array(info.solution)[:18], np.array(ro["out"].solution))
    # (b) nothing valid: no measurement, delta is the prior pull only (zero), status flags it
    h.lio_begin18(xg, xg)
    h.lio_set_neighbours(nbr, np.zeros(n, dtype=np.uint8))
    info = h.lio_iterate18(1, 0)
    assert info.effct_feat_num == 0
    assert info.status & 4
    assert np.abs(np.array(info.solution)[:18]).max() == 0.0
    h.close()


def test_large_scan_properties(gpu_lib, oracle_lib, scene):
    """200k-point scan (BASELINE config 4 size): parity with the oracle and shard additivity."""
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    n = 200000
    fr, h, nbr, valid = _setup(capi, synth, scene, n)
    xo = orc.state18_from_frame(fr)
    sel_o = valid.copy()
    ro = orc.lio18_iterate(xo, xo.copy(), fr.body_xyz, nbr, sel_o, fr.R_LI, fr.t_LI, fr.laser_point_cov, nthreads=8)
    xg = capi.state18_from_frame(fr)
    h.lio_set_points(fr.body_xyz)
    h.lio_begin18(xg, xg)
    h.lio_set_neighbours(nbr, valid)
    info = h.lio_iterate18(1, 0)
    assert info.effct_feat_num == ro["out"].effct_feat_num
    assert_delta_close(np.array(info.solution)[:18], np.array(ro["out"].solution))
    h.close()


def test_begin_behind_enqueue_only_calls_waits_for_the_state_mirror(gpu_lib, scene):
    """fl_lio_begin18 stages the state in a page-locked mirror and skips the stream synchronisation when no copy out of that mirror can
    be in flight (frame drivers: every begin follows a read-back). After enqueue-only calls (iterate without info) a copy MAY be in
    flight: begin must wait. Ten begin / enqueue-only rounds with alternating states, then the result of the last one must be the
    result of the same sequence on a fresh handle -- and the scan staged from page-locked memory gives the same bits as from a numpy
    array."""
    capi = gpu_lib
    from fast_livo_amd import synth
    fr, h, nbr, valid = _setup(capi, synth, scene, 20000)
    F = capi.FL_ITER_FORCE
    h.lio_set_points(fr.body_xyz)
    xa = capi.state18_from_frame(fr)
    xb = capi.state18_from_frame(fr)
    xb.pos[0] += 0.01
    for r in range(10):
        x = xa if r % 2 == 0 else xb
        h.lio_begin18(x, x)
        h.lio_set_neighbours(nbr, valid)
        h.lio_iterate18(3, F, want_info=False)          # enqueue only: the H2D of the state mirror may still be in flight
    info = h.lio_iterate18(1, F)
    got = np.array(info.solution)
    h2 = capi.Handle(capi.config_from_frames(fr, max_iterations=10))
    pinned = h2.host_alloc(fr.body_xyz.shape, np.float32)
    pinned[...] = fr.body_xyz
    h2.lio_set_points(pinned)
    h2.lio_begin18(xb, xb)
    h2.lio_set_neighbours(nbr, valid)
    h2.lio_iterate18(3, F, want_info=False)
    ref = np.array(h2.lio_iterate18(1, F).solution)
    assert np.array_equal(got, ref)
    h2.host_free(pinned)
    h.close(); h2.close()
