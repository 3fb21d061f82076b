"""CPU tests of the product's own arithmetic (fast-livo_amd/csrc/fl_math.h, fl_ikfom_math.h compiled
for the host by tests/host_emul -- test-only, never shipped) against the oracle, and of the sharded
orchestration over gloo with world_size 2."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import TOL_SUMS_REL, assert_delta_close, p, sums_to_HTH

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _emul_lio(E, fr, nbr, valid):
    n = fr.n
    plane = np.zeros((n, 4), np.float32)
    ok = np.zeros(n, np.uint8)
    E.emul_fit_planes(p(nbr, C.c_float), n, p(plane, C.c_float), p(ok, C.c_uint8))
    return plane, (valid & ok).astype(np.uint8)


@pytest.mark.parametrize("n", [1, 257, 6000])
def test_plane_fit_gates_rows_and_solve_match_oracle(oracle_lib, emul_lib, scene, n):
    from fast_livo_amd import synth
    orc, E = oracle_lib, emul_lib
    fr = synth.make_lio_frame(n, scene=scene)
    nbr, valid = synth.knn5(scene, fr.world_at(fr.R_prior, fr.p_prior))
    x = orc.state18_from_frame(fr)
    xp = x.copy()
    sel_o = valid.copy()
    r = orc.lio18_iterate(x, xp, fr.body_xyz, nbr, sel_o, fr.R_LI, fr.t_LI, fr.laser_point_cov)
    plane, sel_e = _emul_lio(E, fr, nbr, valid)
    x0 = xp.vec().copy()
    sums = np.zeros(32)
    nv = np.zeros((n, 4), np.float32)
    RLI = np.ascontiguousarray(fr.R_LI.reshape(9))
    tLI = np.ascontiguousarray(fr.t_LI)
    E.emul_lio18_accumulate(p(fr.body_xyz, C.c_float), p(plane, C.c_float), p(sel_e, C.c_uint8), n, p(x0, C.c_double),
                            p(RLI, C.c_double), p(tLI, C.c_double), p(sums, C.c_double), p(nv, C.c_float))
    assert np.array_equal(sel_e, sel_o)                                  # zero selection flips
    assert np.array_equal(nv[sel_o != 0], r["normvec"][sel_o != 0])      # bit-identical planes / pd2
    assert int(sums[27]) == r["out"].effct_feat_num
    if r["out"].effct_feat_num:
        S, HTz = sums_to_HTH(sums)
        HTH_o = np.array(r["out"].HTH).reshape(6, 6)
        assert np.abs(S - HTH_o).max() <= TOL_SUMS_REL * np.abs(HTH_o).max()
    for fn in (E.emul_solve18, E.emul_solve18_fast):                     # push-through and LDL^T forms
        xe = x0.copy()
        G6 = np.zeros(108)
        d = np.zeros(18)
        P = np.ascontiguousarray(fr.cov18.reshape(-1))
        st = fn(p(xe, C.c_double), p(xp.vec().copy(), C.c_double), p(P, C.c_double), C.c_double(fr.laser_point_cov),
                p(sums, C.c_double), C.c_double(1.0), p(G6, C.c_double), p(d, C.c_double))
        assert st == 0
        assert_delta_close(d, np.array(r["out"].solution))
        assert np.abs(xe - x.vec()).max() <= 1e-9
        assert np.abs(G6.reshape(18, 6) - r["G"][:, :6]).max() <= 1e-9


def test_gate_threshold_is_the_reference_expression(emul_lib, scene):
    """fl_gate_threshold (fl_math.h): `|pd2| <= T` must be the reference's `s > 0.9`, s = (float)(1 - 0.9*|pd2|/sqrt(|p_b|))
    (laserMapping.cpp:1574-1576), for every float |pd2| -- probed at T +- 3 ulp and at far values, for scan points, tiny and huge
    ranges, the origin and non-finite coordinates."""
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(20000, scene=scene)
    rng = np.random.default_rng(0)
    extra = np.array([[0, 0, 0], [1e-20, 0, 0], [1e18, 1e18, 1e18], [np.inf, 0, 0], [np.nan, 1, 1], [3e-4, 2e-4, 1e-4], [1e4, -2e4, 3e3]],
                     dtype=np.float32)
    body = np.concatenate([fr.body_xyz, (rng.normal(0, 1, (5000, 3)) * 10.0 ** rng.uniform(-6, 6, (5000, 1))).astype(np.float32), extra])
    body = np.ascontiguousarray(body, dtype=np.float32)
    T = np.zeros(len(body), dtype=np.float32)
    bad = C.c_int(0)
    emul_lib.emul_gate_thresholds(p(body, C.c_float), len(body), p(T, C.c_float), C.byref(bad))
    assert bad.value == 0
    assert T[len(body) - 7] == -1.0 and T[len(body) - 3] == -1.0          # the origin and a NaN coordinate never pass
    rngs = np.sqrt(np.linalg.norm(fr.body_xyz.astype(np.float64), axis=1))
    assert np.allclose(T[:fr.n], rngs / 9.0, rtol=1e-5)                  # the analytic crossing: |pd2| < sqrt(|p_b|) / 9


def test_degenerate_neighbours_never_selected(emul_lib):
    E = emul_lib
    nb = np.zeros((3, 5, 3), dtype=np.float32)
    nb[0] = 1.0                                        # five identical points
    nb[1, :, 0] = np.arange(5)                         # collinear
    nb[2] = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 1, 0.5], [0.5, 0.5, -0.5]])  # not planar within 0.1
    plane = np.zeros((3, 4), np.float32)
    ok = np.zeros(3, np.uint8)
    E.emul_fit_planes(p(nb, C.c_float), 3, p(plane, C.c_float), p(ok, C.c_uint8))
    body = np.ones((3, 3), np.float32)
    sel = np.ones(3, np.uint8)
    x = np.concatenate([np.eye(3).reshape(9), np.zeros(15)])
    sums = np.zeros(32)
    eye = np.eye(3).reshape(9).copy()
    z3 = np.zeros(3)
    E.emul_lio18_accumulate(p(body, C.c_float), p(plane, C.c_float), p(sel, C.c_uint8), 3, p(x, C.c_double), p(eye, C.c_double),
                            p(z3, C.c_double), p(sums, C.c_double), None)
    assert ok[2] == 0
    assert sums[27] <= 1.0 and np.isfinite(sums).all()  # NaN planes flow to "not selected", never into the sums


def test_mode23_iteration_matches_oracle(oracle_lib, emul_lib, scene):
    from fast_livo_amd import synth
    orc, E = oracle_lib, emul_lib
    for n, max_iter in ((15, 4), (3000, 4)):
        fr = synth.make_lio_frame(n, scene=scene)
        knn = lambda w: synth.knn5(scene, w)  # noqa: E731
        x23 = orc.state23_from_frame(fr, synth.quat_from_R)
        P = fr.cov23.copy()
        x0 = x23.vec().copy()
        ro = orc.ikfom_update(x23, P, fr.body_xyz, 0.001, max_iter, knn)
        x = x0.copy()
        Pw = np.zeros((23, 23))
        Pprop = fr.cov23.copy()
        limit = np.full(23, 0.001)
        ctl = np.array([-1, 0, 1, 0, max_iter, 0], dtype=np.int32)
        plane = np.zeros((n, 4), np.float32)
        ok = np.zeros(n, np.uint8)
        sel = np.zeros(n, np.uint8)
        sums = np.zeros(96)
        dx = np.zeros(23)
        iters = 0
        while not ctl[3] and ctl[0] < max_iter:
            if ctl[2]:
                w = np.zeros((n, 3), np.float32)
                E.emul_world_points23(p(x, C.c_double), p(fr.body_xyz, C.c_float), n, p(w, C.c_float))
                nbr, valid = knn(w)
                nbr = np.ascontiguousarray(nbr)
                E.emul_fit_planes(p(nbr, C.c_float), n, p(plane, C.c_float), p(ok, C.c_uint8))
                sel = (valid & ok).astype(np.uint8)
            E.emul_h_share_sums(p(x, C.c_double), p(fr.body_xyz, C.c_float), p(plane, C.c_float), p(sel, C.c_uint8), n,
                                p(sums, C.c_double), None)
            E.emul_ikfom_iterate(p(x, C.c_double), p(x0, C.c_double), p(Pprop, C.c_double), p(Pw, C.c_double), p(limit, C.c_double),
                                 C.c_double(0.001), p(sums, C.c_double), p(ctl, C.c_int32), p(dx, C.c_double))
            iters += 1
        assert iters == ro["out"].iterations
        assert int(sums[90]) == ro["out"].effct_feat_num
        assert np.abs(x - x23.vec()).max() <= 1e-9
        assert np.abs(Pw - P).max() <= 1e-10
        assert np.abs(dx - np.array(ro["out"].dx)).max() <= 1e-9


def test_mode23_shipped_series_forms_on_the_host(oracle_lib, emul_lib, emul_series_lib, scene):
    """the device compiles power-series forms of the manifold exp / log / Jacobians (fl_ikfom_math.h FL_IK_SERIES),
    the host emulation used the libm forms -- only GPU tests saw the shipped arithmetic. libemul_series.so is the same source with
    the series forms selected: (1) the whole update, series build vs the C oracle (1e-9 / 1e-10) and vs the libm build (rounding
    level); (2) boxplus / boxminus across the series range and beyond it (|delta| from 1e-13 to 2 rad: the forms hand over to libm
    at x^2 = 0.25) against the oracle's MTK restatement."""
    from fast_livo_amd import synth
    orc, E, ES = oracle_lib, emul_lib, emul_series_lib
    knn = lambda w: synth.knn5(scene, w)  # noqa: E731

    def run(lib, fr, max_iter):
        n = fr.n
        x0 = orc.state23_from_frame(fr, synth.quat_from_R).vec().copy()
        x = x0.copy()
        Pw = np.zeros((23, 23)); Pprop = fr.cov23.copy(); limit = np.full(23, 0.001)
        ctl = np.array([-1, 0, 1, 0, max_iter, 0], dtype=np.int32)
        plane = np.zeros((n, 4), np.float32); ok = np.zeros(n, np.uint8); sel = np.zeros(n, np.uint8)
        sums = np.zeros(96); dx = np.zeros(23)
        iters = 0
        while not ctl[3] and ctl[0] < max_iter:
            if ctl[2]:
                w = np.zeros((n, 3), np.float32)
                lib.emul_world_points23(p(x, C.c_double), p(fr.body_xyz, C.c_float), n, p(w, C.c_float))
                nbr, valid = knn(w)
                nbr = np.ascontiguousarray(nbr)
                lib.emul_fit_planes(p(nbr, C.c_float), n, p(plane, C.c_float), p(ok, C.c_uint8))
                sel = (valid & ok).astype(np.uint8)
            lib.emul_h_share_sums(p(x, C.c_double), p(fr.body_xyz, C.c_float), p(plane, C.c_float), p(sel, C.c_uint8), n, p(sums, C.c_double), None)
            lib.emul_ikfom_iterate(p(x, C.c_double), p(x0, C.c_double), p(Pprop, C.c_double), p(Pw, C.c_double), p(limit, C.c_double),
                                   C.c_double(0.001), p(sums, C.c_double), p(ctl, C.c_int32), p(dx, C.c_double))
            iters += 1
        return x, Pw, iters
    for n, max_iter in ((15, 4), (3000, 4), (3000, 10)):
        fr = synth.make_lio_frame(n, scene=scene)
        x23 = orc.state23_from_frame(fr, synth.quat_from_R)
        P = fr.cov23.copy()
        ro = orc.ikfom_update(x23, P, fr.body_xyz, 0.001, max_iter, knn)
        xs, Ps, its = run(ES, fr, max_iter)
        xl, Pl, itl = run(E, fr, max_iter)
        assert its == itl == ro["out"].iterations
        assert np.abs(xs - x23.vec()).max() <= 1e-9 and np.abs(Ps - P).max() <= 1e-10
        assert np.abs(xs - xl).max() <= 1e-13 and np.abs(Ps - Pl).max() <= 1e-13          # series vs libm: rounding level
    rng = np.random.default_rng(17)
    worst = 0.0
    for k in range(600):
        mag = [1e-13, 1e-9, 1e-5, 1e-3, 0.05, 0.3, 0.49, 0.51, 1.0, 2.0][k % 10]
        s = orc.State23()
        for f, _ in s._fields_:
            getattr(s, f)[:] = rng.standard_normal(len(getattr(s, f)))
        for f in ("rot", "offset_R_L_I"):
            q = np.array(getattr(s, f)); getattr(s, f)[:] = q / np.linalg.norm(q)
        g = np.array(s.grav); s.grav[:] = g / np.linalg.norm(g) * 9.809
        d = rng.standard_normal(23); d *= mag / np.linalg.norm(d[3:6])
        b = s.copy()
        orc.lib().orc_state23_boxplus(C.byref(b), p(d, C.c_double))
        xs = s.vec().copy()
        ES.emul_x23_boxplus(p(xs, C.c_double), p(d, C.c_double))
        worst = max(worst, np.abs(xs - b.vec()).max())
        out_o = np.zeros(23); out_s = np.zeros(23)
        orc.lib().orc_state23_boxminus(C.byref(b), C.byref(s), p(out_o, C.c_double))
        bv, sv = b.vec().copy(), s.vec().copy()
        ES.emul_x23_boxminus(p(bv, C.c_double), p(sv, C.c_double), p(out_s, C.c_double))
        worst = max(worst, np.abs(out_s - out_o).max())
    assert worst <= 1e-12, worst


def test_shard_ranges_cover_everything():
    import fastlivo  # noqa: F401
    from fast_livo_amd.sharded import shard_range
    for n in (1, 7, 50000, 200001):
        for world in (1, 2, 3, 8):
            edges = [shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(world - 1))


def test_sharded_pass_gloo_world2(oracle_lib, emul_lib):
    """N>1 path on CPU: 2 gloo ranks, point-range shards, all-reduce of the 32-double record, redundant
    solve -> both ranks hold the same state, equal to the unsharded oracle."""
    script = os.path.join(ROOT, "tests", "gloo_sharded_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", "29631", script],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "RANK0 OK" in out.stdout and "RANK1 OK" in out.stdout, out.stdout[-2000:]


def test_cpp_local_map_window_follows_the_oracle(tmp_path):
    """LocalMapDev::lasermap_fov_segment (C++ host mirror) against oracle/orc_map.c::orc_fov_segment along a random walk: the same
    boxes are cut off and the window is the same float for float. Runs without a device (the deletion call fails on a null handle)."""
    import subprocess
    from oracle import oracle as orc
    from fast_livo_amd import LIB_PATH, capi
    capi.build()
    src = os.path.join(os.path.dirname(__file__), "host_emul", "fov_walk.cpp")
    exe = tmp_path / "fov_walk"
    libdir = os.path.dirname(LIB_PATH)
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", str(exe), src, "-L" + libdir, "-lfastlivo_hip", "-Wl,-rpath," + libdir,
                           "-Wl,-rpath,/opt/rocm/lib"])
    rng = np.random.default_rng(12)
    steps = np.cumsum(rng.normal(0, 6.0, (300, 3)) + np.array([4.0, -2.0, 0.5]), axis=0)
    cube, det, mov = 120.0, 20.0, 1.5
    inp = f"{len(steps)} {cube} {det} {mov}\n" + "\n".join(" ".join(repr(float(v)) for v in p) for p in steps) + "\n"
    out = subprocess.run([str(exe)], input=inp, capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    win = np.zeros(6, dtype=np.float32)
    init = False
    moved = 0
    for p, ln in zip(steps, lines):
        boxes, init = orc.fov_segment(win, init, p, cube, det, mov)
        vals = ln.split()
        assert int(vals[0]) == len(boxes)
        assert np.array_equal(np.array(vals[1:], dtype=np.float32), win), ln
        moved += len(boxes) > 0
    assert moved > 10


def test_voxel_order_is_pcl_voxelgrid_output_order():
    """synth.voxel_order: ascending idx = i + j*dx + k*dx*dy over the leaf grid, stable inside a voxel (what pcl::VoxelGrid's sorted index
    vector gives; bench.py feeds the frame drivers scans in this order)"""
    import importlib
    synth = importlib.import_module("fast-livo_amd.synth")
    rng = np.random.default_rng(3)
    pts = rng.uniform(-5, 7, (4000, 3)).astype(np.float32)
    leaf = 0.5
    perm = synth.voxel_order(pts, leaf)
    assert sorted(perm.tolist()) == list(range(len(pts)))
    ijk = np.floor(pts.astype(np.float64) / leaf).astype(np.int64)
    mn = ijk.min(axis=0); d = ijk.max(axis=0) - mn + 1
    idx = (ijk[:, 0] - mn[0]) + (ijk[:, 1] - mn[1]) * d[0] + (ijk[:, 2] - mn[2]) * d[0] * d[1]
    si = idx[perm]
    assert np.all(np.diff(si) >= 0)
    same = np.diff(si) == 0
    assert np.all(np.diff(perm)[same] > 0)          # stable: equal voxels keep their input order
