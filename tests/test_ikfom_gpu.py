"""GPU parity: Mode-23 (IKFoM) path -- h_share_model bodies and update_iterated_dyn_share_modified --
against the CPU oracle.  Tolerances: identical selection / effct_feat_num; sums 1e-12 relative; state
delta 1e-9 absolute; covariance 1e-10 absolute (|P| ~ 1e-3)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(capi, synth, scene, n, max_iter=4):
    fr = synth.make_lio_frame(n, scene=scene)
    h = capi.Handle(capi.config_from_frames(fr, max_iterations=max_iter))
    return fr, h


@pytest.mark.parametrize("n", [15, 300, 20000])
def test_h_share_model_rows_and_sums(gpu_lib, oracle_lib, scene, n):
    capi, orc = gpu_lib, oracle_lib
    import ctypes as C
    from fast_livo_amd import synth
    fr, h = _setup(capi, synth, scene, n)
    so = orc.state23_from_frame(fr, synth.quat_from_R)
    world = fr.world_at(fr.R_prior, fr.p_prior)
    nbr, valid = synth.knn5(scene, world)
    sel = valid.copy()
    h_x = np.zeros((n, 12))
    hv = np.zeros(n)
    nv = np.zeros((n, 4), dtype=np.float32)
    rl = np.zeros(n)
    tr = C.c_double()
    p = lambda a, t: a.ctypes.data_as(C.POINTER(t))  # noqa: E731
    neff = orc.lib().orc_h_share_model(C.byref(so), p(fr.body_xyz, C.c_float), p(nbr, C.c_float), p(sel, C.c_uint8), n, 4, None,
                                       p(nv, C.c_float), p(rl, C.c_double), p(h_x, C.c_double), p(hv, C.c_double), C.byref(tr))
    sg = capi.state23_from_frame(fr)
    h.lio_set_points(fr.body_xyz)
    h.lio_set_neighbours(nbr, valid)
    HTH, HTh, neff_g, res_g = h.h_share_model_sums(sg)
    assert neff_g == neff
    HTH_o = h_x[:neff].T @ h_x[:neff]
    HTh_o = h_x[:neff].T @ hv[:neff]
    assert np.abs(HTH - HTH_o).max() <= 1e-11 * np.abs(HTH_o).max()
    assert np.abs(HTh - HTh_o).max() <= 1e-11 * max(np.abs(HTh_o).max(), 1e-300) * 10
    assert abs(res_g - tr.value) <= 1e-12 * max(1.0, tr.value)
    # row-compat body: rows in ascending point order
    h.lio_set_neighbours(nbr, valid)
    rows_g, h_g = h.h_share_model_rows(sg, n)
    assert rows_g.shape[0] == neff
    assert np.abs(rows_g - h_x[:neff]).max() <= 1e-12 * max(1.0, np.abs(h_x[:neff]).max())
    assert np.array_equal(h_g, hv[:neff])
    h.close()


@pytest.mark.parametrize("n,max_iter", [(15, 4), (5000, 4), (50000, 10)])
def test_update_iterated_matches_oracle(gpu_lib, oracle_lib, scene, n, max_iter):
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    fr, h = _setup(capi, synth, scene, n, max_iter)
    knn = lambda w: synth.knn5(scene, w)  # noqa: E731
    xo = orc.state23_from_frame(fr, synth.quat_from_R)
    Po = fr.cov23.copy()
    ro = orc.ikfom_update(xo, Po, fr.body_xyz, 0.001, max_iter, knn)
    xg = capi.state23_from_frame(fr)
    Pg = fr.cov23.copy()
    info = h.ikfom_update_iterated(xg, Pg, fr.body_xyz, 0.001, knn)
    assert info.iterations == ro["out"].iterations
    assert info.effct_feat_num == ro["out"].effct_feat_num
    assert info.status == 0
    assert np.abs(np.array(info.solution) - np.array(ro["out"].dx)).max() <= 1e-9
    assert np.abs(xg.vec() - xo.vec()).max() <= 1e-9
    assert np.abs(Pg - Po).max() <= 1e-10
    mask, nv = h.lio_get_selection(n)
    sel_o = ro["sel"] != 0
    # planes are bit-identical; pd2 is evaluated at states that agree to ~1e-13, so the float world
    # point may differ in its last bit for a few points
    assert np.array_equal(nv[sel_o][:, :3], ro["normvec"][sel_o][:, :3])
    assert np.abs(nv[sel_o][:, 3] - ro["normvec"][sel_o][:, 3]).max() <= 2e-6
    h.close()


def test_mode23_agrees_with_mode18_when_extrinsic_is_frozen(gpu_lib, scene):
    """SURVEY 8c analytic property: with a tiny prior covariance on the LiDAR-IMU extrinsic the
    23-state update moves rot/pos like the 18-state one (same measurements, same R)."""
    capi = gpu_lib
    from fast_livo_amd import synth
    n = 20000
    fr = synth.make_lio_frame(n, scene=scene)
    world = fr.world_at(fr.R_prior, fr.p_prior)
    nbr, valid = synth.knn5(scene, world)
    h = capi.Handle(capi.config_from_frames(fr, max_iterations=4))
    h.lio_set_points(fr.body_xyz)
    # Mode-18, diagonal prior
    P18 = np.eye(18) * 1e-3
    x18 = capi.State18.make(fr.R_prior, fr.p_prior, fr.vel, fr.bg, fr.ba, fr.grav, P18)
    h.lio_begin18(x18, x18)
    h.lio_set_neighbours(nbr, valid)
    i18 = h.lio_iterate18(1, capi.FL_ITER_FORCE)
    # Mode-23, same prior on rot/pos, extrinsic frozen
    P23 = np.eye(23) * 1e-3
    P23[6:12, 6:12] = np.eye(6) * 1e-14
    x23 = capi.state23_from_frame(fr)
    h.ikfom_begin(x23, P23)
    h.lio_set_neighbours(nbr, valid)
    i23 = h.ikfom_iterate(1, capi.FL_ITER_FORCE)
    d18 = np.array(i18.solution)[:6]          # rot, pos
    d23 = np.array(i23.solution)
    assert i18.effct_feat_num == i23.effct_feat_num
    assert np.abs(d23[3:6] - d18[0:3]).max() <= 1e-6      # rotation delta
    assert np.abs(d23[0:3] - d18[3:6]).max() <= 1e-6      # position delta
    assert np.abs(d23[6:12]).max() <= 1e-8                # extrinsic did not move
    h.close()
