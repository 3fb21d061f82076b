"""The Eigen-dependent restatements of oracle/orc_*.c held to the REFERENCE ITSELF -- wherever Eigen exists.

oracle/_ref/libeigen_ref.so is the reference's own include/common_lib.h + include/so3_math.h (and, with Boost, include/use-ikfom.hpp
+ IKFoM_toolkit) compiled where they lie behind a C driver (recipe: oracle/ref_eigen/).  This image and the GPU box have no Eigen
(NOTES.md), so here every test of this file SKIPS with the recipe's own message; on a box with Eigen `make -C oracle/ref_eigen`
builds the library and the same tests pin
    esti_plane<float>                       common_lib.h:448-493      bit for bit (float)
    StatesGroup += / -                      common_lib.h:343-365      bit for bit (double)
    Exp / Log                               so3_math.h:54-81          bit for bit
    state_ikfom boxplus / boxminus          use-ikfom.hpp, MTK        <= 1e-15 / 1e-13
    update_iterated_dyn_share_modified      esekfom.hpp:1619-1928     state <= 1e-12, covariance <= 1e-12 relative
The bit-for-bit rows are the claim the QR restatement makes (oracle/orc_lio_common.h: Eigen's published ColPivHouseholderQR order);
if a given Eigen version associates a reduction differently the test prints how many of the planes differ and by how much, and fails.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import eigenref

pytestmark = pytest.mark.skipif(not eigenref.available(), reason="parity of the Eigen rows unpinned here: " + eigenref.why_not())


def _neighbour_sets(orc, scene, n):
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(n, scene=scene)
    world = fr.world_at(fr.R_prior, fr.p_prior).astype(np.float32)
    nbr, _, valid, _ = orc.knn5_bruteforce(scene.map_xyz, world)
    return nbr[valid != 0]


def test_esti_plane_bit_for_bit(oracle_lib, scene):
    orc = oracle_lib
    sets = _neighbour_sets(orc, scene, 20000)
    # plus degenerate inputs: collinear, coincident, far from the origin
    rng = np.random.default_rng(3)
    extra = [np.repeat(rng.standard_normal((1, 3)).astype(np.float32), 5, 0),
             (np.arange(5, dtype=np.float32)[:, None] * np.array([[1.0, 2.0, 3.0]], np.float32)),
             (rng.standard_normal((5, 3)) * 1e-3 + 500.0).astype(np.float32)]
    sets = np.concatenate([sets, np.stack(extra)])
    differ = 0
    worst = 0.0
    for near in sets:
        near = np.ascontiguousarray(near, np.float32)
        a = np.zeros(4, np.float32)
        ok_a = orc.lib().orc_unit_esti_plane(near.ctypes.data_as(C.POINTER(C.c_float)), C.c_float(0.1), a.ctypes.data_as(C.POINTER(C.c_float)))
        b, ok_b = eigenref.esti_plane(near, 0.1)
        if bool(ok_a) != ok_b or not np.array_equal(a, b, equal_nan=True):
            differ += 1
            if np.isfinite(a).all() and np.isfinite(b).all():
                worst = max(worst, float(np.abs(a - b).max()))
    assert differ == 0, f"{differ} of {len(sets)} planes differ from Eigen {eigenref.lib().ref_eigen_version().decode()} (max |d| {worst:g})"


def test_state18_ops_and_so3_bit_for_bit(oracle_lib):
    orc = oracle_lib
    rng = np.random.default_rng(5)
    for k in range(2000):
        mag = [1.0, 1e-2, 1e-5, 1.1e-5, 1e-9, 0.0][k % 6]
        v = rng.standard_normal(3) * mag
        R_o = np.zeros(9)
        orc.lib().orc_unit_so3_exp(v.ctypes.data_as(C.POINTER(C.c_double)), R_o.ctypes.data_as(C.POINTER(C.c_double)))
        R_r = eigenref.so3_exp(v)
        assert np.array_equal(R_o.reshape(3, 3), R_r)
        l_o = np.zeros(3)
        orc.lib().orc_unit_so3_log(R_o.ctypes.data_as(C.POINTER(C.c_double)), l_o.ctypes.data_as(C.POINTER(C.c_double)))
        assert np.array_equal(l_o, eigenref.so3_log(R_r))
        # StatesGroup += and -
        a = orc.State18.make(R_r, *(rng.standard_normal(3) for _ in range(5)), np.eye(18))
        d = rng.standard_normal(18) * mag
        b = a.copy()
        orc.lib().orc_unit_state18_plus(C.byref(b), d.ctypes.data_as(C.POINTER(C.c_double)))
        v15 = np.concatenate([np.array(a.pos), np.array(a.vel), np.array(a.bg), np.array(a.ba), np.array(a.grav)])
        rot_r, v_r = eigenref.state18_plus(np.array(a.rot), v15, d)
        assert np.array_equal(np.array(b.rot).reshape(3, 3), rot_r)
        assert np.array_equal(np.concatenate([np.array(b.pos), np.array(b.vel), np.array(b.bg), np.array(b.ba), np.array(b.grav)]), v_r)
        out = np.zeros(18)
        orc.lib().orc_unit_state18_minus(C.byref(b), C.byref(a), out.ctypes.data_as(C.POINTER(C.c_double)))
        assert np.array_equal(out, eigenref.state18_minus(rot_r, v_r, np.array(a.rot), v15))


def test_mode23_against_the_reference_toolkit(oracle_lib, scene):
    if not eigenref.have_mtk():
        pytest.skip("libeigen_ref.so was built without Boost: the MTK / esekf part of the reference is not in it")
    import test_cross_oracle_cpu as xo
    from fast_livo_amd import synth
    orc = oracle_lib
    rng = np.random.default_rng(11)
    # box operators
    for k in range(200):
        mag = [1.0, 1e-2, 1e-3, 1e-6, 1e-13, 0.0][k % 6]
        s = orc.State23()
        for f, _ in s._fields_:
            getattr(s, f)[:] = rng.standard_normal(len(getattr(s, f)))
        for f in ("rot", "offset_R_L_I"):
            q = np.array(getattr(s, f)); getattr(s, f)[:] = q / np.linalg.norm(q)
        g = np.array(s.grav); s.grav[:] = g / np.linalg.norm(g) * 9.809
        d = rng.standard_normal(23) * mag
        b = s.copy()
        orc.lib().orc_state23_boxplus(C.byref(b), d.ctypes.data_as(C.POINTER(C.c_double)))
        assert np.abs(b.vec() - eigenref.state23_boxplus(s.vec(), d)).max() <= 1e-15 * 10
        out = np.zeros(23)
        orc.lib().orc_state23_boxminus(C.byref(b), C.byref(s), out.ctypes.data_as(C.POINTER(C.c_double)))
        assert np.abs(out - eigenref.state23_boxminus(b.vec(), s.vec())).max() <= 1e-13
    # the whole update, the reference's own esekf around the C oracle's h_share_model
    for n, max_iter, P0 in ((3000, 4, None), (3000, 10, xo._spd23(1, 1e-3)), (15, 4, xo._spd23(3, 1e-3))):
        fr = synth.make_lio_frame(n, scene=scene)
        P0 = fr.cov23.copy() if P0 is None else P0
        x_c = orc.state23_from_frame(fr, synth.quat_from_R)
        P_c = P0.copy()
        cb_c, _ = xo._c_rows_callback(orc, fr, scene.map_xyz)
        r_c = orc.ikfom_update_dyn_share(x_c, P_c, 0.001, max_iter, cb_c)
        cb_r, _ = xo._c_rows_callback(orc, fr, scene.map_xyz)
        s_r, P_r, calls = eigenref.ikfom_update_dyn_share(orc.state23_from_frame(fr, synth.quat_from_R).vec(), P0, 0.001, max_iter, cb_r)
        assert calls == r_c["out"].iterations
        assert np.abs(x_c.vec() - s_r).max() <= 1e-12
        assert np.abs(P_c - P_r).max() <= 1e-12 * max(1.0, np.abs(P_r).max())
