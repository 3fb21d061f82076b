"""GPU parity of the one-patch-per-lane VIO producers (csrc/vio_kernels.h vio_produce_wide, FL_OPT_VIO_WIDE).

The at-scale form of the photometric pass gives every lane a patch of its own (shared taps and bilinear values, no cross-lane
reductions); passes over >= 16 384 patches take it by themselves, FL_OPT_VIO_WIDE = 2 forces it at any size. The float part follows
the reference's expressions and operand order (lidar_selection.cpp:826-829,837,849), so per-patch errors are compared BIT FOR BIT with
the CPU oracle and with the 16-lanes-per-patch form; the fp64 sums differ in their order only (state delta 1e-9 like every fp64 sum).
Covered: every pyramid level (tap scales 1, 2 and 4 each have row loads of their own; patches reaching over the image border, mixed
search levels and an image width that is not a multiple of 4 take the out-of-line byte path), ragged sizes (1, 63, 64, 65, 300, 2 000 patches: partial wavefronts, several workgroups), non-zero
search levels, the distorting camera, a whole ComputeJ with accept / revert, the accumulate-only kernel of the sharded form, and the
automatic switch at 16 384 patches -- against the 16-lane form at the same size AND against the CPU oracle (orc_vio_set_threads: its
patch loop spread over the host's cores, bit-identical by construction) at 16 384 and 70 000 DISTINCT patches, default option, every
level, plus a whole ComputeJ with its accept / revert decisions.
"""
import os
import numpy as np
import pytest

from helpers import assert_delta_close

pytestmark = pytest.mark.gpu


def _frames(synth, scene, m, n=2000, **kw):
    fr = synth.make_lio_frame(n, scene=scene)
    vf = synth.make_vio_frame(m, fr, **kw)
    return fr, vf


def _handle(capi, fr, vf, wide):
    h = capi.Handle(capi.config_from_frames(fr, vf, max_iterations=vf.max_iterations))
    h.set_option(capi.FL_OPT_VIO_WIDE, wide)
    h.vio_set_frame(vf.img)
    h.vio_set_patches(vf.ref_patch, vf.pos, vf.search_level)
    return h


@pytest.mark.parametrize("m,level", [(1, 0), (63, 0), (64, 0), (65, 1), (300, 0), (300, 2), (2000, 0), (2000, 1), (5000, 0)])
def test_wide_pass_matches_oracle_and_narrow(gpu_lib, oracle_lib, scene, m, level):
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    fr, vf = _frames(synth, scene, m)
    vf.max_iterations = 1
    xo = orc.state18_from_frame(fr)
    ro = orc.vio_update_state(vf, xo, xo.copy(), 1e10, level)
    res = {}
    for wide in (2, 0):
        h = _handle(capi, fr, vf, wide)
        xg = capi.state18_from_frame(fr)
        h.vio_begin(xg, xg)
        err, info = h.vio_update_state(1e10, level)
        assert info.iterations == 1 and info.accepted == 1
        assert info.effct_feat_num == ro["out"].n_meas == 64 * m
        assert abs(err - ro["error"]) <= 1e-5 * ro["error"]
        res[wide] = (h.vio_get_errors(m), np.array(info.solution)[:18], h.vio_get_state18().vec())
        h.close()
    assert np.array_equal(res[2][0], ro["errors"])            # per-patch errors: the reference's float chain, bit for bit
    assert np.array_equal(res[2][0], res[0][0])
    assert_delta_close(res[2][1], np.array(ro["out"].solution))
    assert np.abs(res[2][2] - xo.vec()).max() <= 1e-9
    assert np.abs(res[2][2] - res[0][2]).max() <= 1e-10


def test_wide_search_levels_match_oracle(gpu_lib, oracle_lib, scene):
    """Mixed search levels: the wavefronts take the out-of-line byte path for their whole sweep."""
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    fr, vf = _frames(synth, scene, 400)
    vf.search_level[::3] = 1
    vf.search_level[1::7] = 2
    vf.max_iterations = 1
    xo = orc.state18_from_frame(fr)
    ro = orc.vio_update_state(vf, xo, xo.copy(), 1e10, 0)
    h = _handle(capi, fr, vf, 2)
    xg = capi.state18_from_frame(fr)
    h.vio_begin(xg, xg)
    err, info = h.vio_update_state(1e10, 0)
    assert np.array_equal(h.vio_get_errors(vf.m), ro["errors"])
    assert_delta_close(np.array(info.solution)[:18], np.array(ro["out"].solution))
    assert np.abs(h.vio_get_state18().vec() - xo.vec()).max() <= 1e-9
    h.close()


def test_wide_border_patches_equal_narrow(gpu_lib, scene):
    """Patches whose taps reach over the image border: the reference (and the oracle) read unchecked there, both device forms clamp
    rows and columns -- the same taps, hence the same bits."""
    capi = gpu_lib
    from fast_livo_amd import synth
    fr, vf = _frames(synth, scene, 200)
    Rcw, Pcw = synth.cam_pose(vf.Rcl, vf.Pcl, fr.R_LI, fr.t_LI, fr.R_true, fr.p_true)
    W, H = vf.cam["width"], vf.cam["height"]
    px = np.array([[2.0, 2.0], [W - 3.0, H - 3.0], [W / 2, 1.0], [1.5, H / 2], [W - 2.0, 3.0], [W / 3, H - 2.5]])
    xyc = np.stack([(px[:, 0] - vf.cam["cx"]) / vf.cam["fx"], (px[:, 1] - vf.cam["cy"]) / vf.cam["fy"], np.ones(len(px))], -1) * 5.0
    vf.pos[10:10 + len(px)] = (xyc - Pcw) @ Rcw
    vf.pos[100:100 + len(px)] = (xyc * 1.7 - Pcw) @ Rcw
    vf.max_iterations = 1
    res = {}
    for wide in (2, 0):
        h = _handle(capi, fr, vf, wide)
        xg = capi.state18_from_frame(fr)
        h.vio_begin(xg, xg)
        for level in (2, 0):
            info = h.vio_iterate(level, 1, capi.FL_ITER_FORCE)
            res[(wide, level)] = (h.vio_get_errors(vf.m), np.array(info.solution)[:18])
        h.close()
    for level in (2, 0):
        assert np.array_equal(res[(2, level)][0], res[(0, level)][0]), level
        assert_delta_close(res[(2, level)][1], res[(0, level)][1])


@pytest.mark.parametrize("distortion", [False, True])
def test_wide_compute_j_matches_oracle(gpu_lib, oracle_lib, scene, distortion):
    """Whole ComputeJ (levels 2, 1, 0, accept / revert through the per-patch words, covariance update) on the wide producers."""
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    fr, vf = _frames(synth, scene, 700, distortion=distortion)
    xo = orc.state18_from_frame(fr)
    ro = orc.vio_compute_j(vf, xo, xo.copy())
    h = _handle(capi, fr, vf, 2)
    xg = capi.state18_from_frame(fr)
    infos = h.vio_compute_j(xg, xg.copy())
    for lv in (2, 1, 0):
        assert infos[lv].iterations == ro["outs"][lv].iterations, lv
        assert infos[lv].accepted == ro["outs"][lv].accepted, lv
        assert abs(infos[lv].total_residual - ro["outs"][lv].error) <= 1e-5 * ro["outs"][lv].error
    assert np.abs(xg.vec() - xo.vec()).max() <= 1e-9
    assert np.abs(xg.cov_np() - xo.cov_np()).max() <= 1e-12
    assert np.array_equal(h.vio_get_errors(vf.m), ro["errors"])
    h.close()


def test_wide_accumulate_equals_narrow(gpu_lib, scene):
    """The accumulate-only kernel of the sharded form (fl_vio_accumulate): same 32 sums from both producer forms (fp64 re-ordering)."""
    capi = gpu_lib
    import torch
    from fast_livo_amd import synth
    fr, vf = _frames(synth, scene, 1500)
    xg = capi.state18_from_frame(fr)
    sums = {}
    for wide in (2, 0):
        h = _handle(capi, fr, vf, wide)
        h.vio_begin(xg, xg)
        t = torch.zeros(capi.FL_SUMS18, dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        h.vio_accumulate(0, t.data_ptr())
        h.sync()
        torch.cuda.synchronize()
        sums[wide] = t.cpu().numpy().copy()
        h.close()
    scale = np.abs(sums[0]).max()
    assert np.abs(sums[2] - sums[0]).max() <= 1e-12 * scale
    assert sums[2][27] == 64.0 * vf.m


def test_wide_is_automatic_at_scale(gpu_lib, scene):
    """70 000 patches (the frame's 2 000 tiled): FL_OPT_VIO_WIDE = 1 (default) takes the wide producers, 0 the 16-lane ones; ten real
    passes (accept / revert decided through the per-patch words) end in the same state, per-patch errors bit for bit."""
    capi = gpu_lib
    from fast_livo_amd import synth
    fr, vf = _frames(synth, scene, 2000)
    m = 70000
    reps = (m + vf.m - 1) // vf.m
    ref = np.tile(vf.ref_patch, (reps, 1, 1))[:m]
    pos = np.tile(vf.pos, (reps, 1))[:m]
    sl = np.tile(vf.search_level, reps)[:m]
    out = {}
    for wide in (1, 0):
        h = capi.Handle(capi.config_from_frames(fr, vf, max_iterations=10))
        h.set_option(capi.FL_OPT_VIO_WIDE, wide)
        h.vio_set_frame(vf.img)
        h.vio_set_patches(ref, pos, sl)
        xg = capi.state18_from_frame(fr)
        h.vio_begin(xg, xg)
        err, info = h.vio_update_state(1e10, 0)
        out[wide] = (err, info.iterations, info.accepted, h.vio_get_errors(m), h.vio_get_state18().vec())
        h.close()
    assert out[1][1] == out[0][1] and out[1][2] == out[0][2]
    assert np.array_equal(out[1][3], out[0][3])
    assert np.abs(out[1][4] - out[0][4]).max() <= 1e-9
    assert abs(out[1][0] - out[0][0]) <= 1e-6 * abs(out[0][0])


def test_compute_j_at_scale_equals_the_16_lane_form(gpu_lib, scene):
    """ComputeJ over 70 000 patches with the default option: all three levels (tap scales 4, 2, 1: each with its own row loads) run on the
    one-patch-per-lane producers, one launch per pass. Against the 16-lane form throughout: same pass and accept counts per level,
    per-patch errors bit for bit, state 1e-9."""
    capi = gpu_lib
    from fast_livo_amd import synth
    fr, vf = _frames(synth, scene, 2000)
    m = 70000
    reps = (m + vf.m - 1) // vf.m
    ref = np.tile(vf.ref_patch, (reps, 1, 1))[:m]
    pos = np.tile(vf.pos, (reps, 1))[:m]
    sl = np.tile(vf.search_level, reps)[:m]
    out = {}
    for wide in (1, 0):
        h = capi.Handle(capi.config_from_frames(fr, vf, max_iterations=4))
        h.set_option(capi.FL_OPT_VIO_WIDE, wide)
        h.vio_set_frame(vf.img)
        h.vio_set_patches(ref, pos, sl)
        xg = capi.state18_from_frame(fr)
        infos = h.vio_compute_j(xg, xg.copy())
        out[wide] = ([(infos[lv].iterations, infos[lv].accepted) for lv in (2, 1, 0)], h.vio_get_errors(m), xg.vec().copy(), xg.cov_np().copy())
        h.close()
    assert out[1][0] == out[0][0]
    assert np.array_equal(out[1][1], out[0][1])
    assert np.abs(out[1][2] - out[0][2]).max() <= 1e-9
    assert np.abs(out[1][3] - out[0][3]).max() <= 1e-12


@pytest.mark.parametrize("wide", [2, 0])
def test_image_width_not_a_multiple_of_four(gpu_lib, oracle_lib, scene, wide):
    """A 642-pixel-wide image: the device copy's row stride (= the width) is not a multiple of 4, so the byte phase of a patch's tap rows
    changes from row to row and neither producer form may use its aligned row loads -- both take their byte paths at every level."""
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    cam = dict(synth.PINHOLE, width=642)
    fr, vf = _frames(synth, scene, 300, cam=cam)
    vf.max_iterations = 1
    for level in (0, 1):
        xo = orc.state18_from_frame(fr)
        ro = orc.vio_update_state(vf, xo, xo.copy(), 1e10, level)
        h = _handle(capi, fr, vf, wide)
        xg = capi.state18_from_frame(fr)
        h.vio_begin(xg, xg)
        err, info = h.vio_update_state(1e10, level)
        assert np.array_equal(h.vio_get_errors(vf.m), ro["errors"]), level
        assert_delta_close(np.array(info.solution)[:18], np.array(ro["out"].solution))
        assert np.abs(h.vio_get_state18().vec() - xo.vec()).max() <= 1e-9
        h.close()


def test_mid_size_update_takes_the_wide_form_per_pass(gpu_lib, scene):
    """20 000 patches, default option, a real update of several passes (fl_vio_update_state, up to 6): the wide form's grid (81 workgroups)
    would pass the admission test of a multi-pass launch, but the multi-pass kernels only exist for the 16-lane producers -- the passes
    must go out one by one on the wide producers (a 16-lane multi-pass kernel on 42 workgroups is correct and several times slower: the
    last assertion). Against the 16-lane form (1 024 workgroups: one launch per pass as well): same counts, per-patch errors bit for bit."""
    capi = gpu_lib
    from fast_livo_amd import synth
    fr, vf = _frames(synth, scene, 2000)
    m = 20000
    reps = (m + vf.m - 1) // vf.m
    ref = np.tile(vf.ref_patch, (reps, 1, 1))[:m]
    pos = np.tile(vf.pos, (reps, 1))[:m]
    sl = np.tile(vf.search_level, reps)[:m]
    out = {}
    for wide in (1, 0):
        h = capi.Handle(capi.config_from_frames(fr, vf, max_iterations=6))
        h.set_option(capi.FL_OPT_VIO_WIDE, wide)
        h.set_timing(True)
        h.vio_set_frame(vf.img)
        h.vio_set_patches(ref, pos, sl)
        xg = capi.state18_from_frame(fr)
        h.vio_begin(xg, xg)
        err, info = h.vio_update_state(1e10, 0)
        out[wide] = (info.iterations, info.accepted, h.vio_get_errors(m), h.vio_get_state18().vec(), h.last_kernel_ms())
        h.close()
    assert out[1][0] == out[0][0] and out[1][1] == out[0][1]
    assert np.array_equal(out[1][2], out[0][2])
    assert np.abs(out[1][3] - out[0][3]).max() <= 1e-9
    assert out[1][4] < 1.5 * out[0][4] + 0.05, (out[1][4], out[0][4])       # (a 16-lane multi-pass kernel on 81 workgroups would take several times longer)


def _oracle_threads(orc):
    n = min(os.cpu_count() or 1, 64)
    orc.lib().orc_vio_set_threads(n)
    return n


@pytest.mark.parametrize("m,level", [(16384, 0), (16384, 1), (16384, 2), (70000, 0), (70000, 1), (70000, 2), (16385, 0), (33000, 2)])
def test_auto_wide_pass_matches_oracle_at_scale(gpu_lib, oracle_lib, scene, m, level):
    """the sizes at which the wide producers switch on BY THEMSELVES (FL_OPT_VIO_WIDE left at its default 1: auto-switch
    at 16 384 patches, >= 2 sweeps per wavefront at 70 000, priority hint live), m DISTINCT patch positions (not the 2 000-set tiled),
    against the CPU oracle: per-patch errors bit for bit (lidar_selection.cpp:826-829,837,849-861), measurement count, accept decision,
    state and solution to 1e-9. 16 385 / 33 000: a ragged last wavefront behind full sweeps."""
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    fr, vf = _frames(synth, scene, m)
    assert len(np.unique(vf.pos, axis=0)) == m
    vf.max_iterations = 1
    _oracle_threads(orc)
    try:
        xo = orc.state18_from_frame(fr)
        ro = orc.vio_update_state(vf, xo, xo.copy(), 1e10, level)
    finally:
        orc.lib().orc_vio_set_threads(1)
    h = capi.Handle(capi.config_from_frames(fr, vf, max_iterations=1))        # (no set_option: the default takes the wide form from 16 384 patches on)
    h.vio_set_frame(vf.img)
    h.vio_set_patches(vf.ref_patch, vf.pos, vf.search_level)
    xg = capi.state18_from_frame(fr)
    h.vio_begin(xg, xg)
    err, info = h.vio_update_state(1e10, level)
    assert info.iterations == ro["out"].iterations == 1 and info.accepted == ro["out"].accepted == 1
    assert info.effct_feat_num == ro["out"].n_meas == 64 * m
    assert abs(err - ro["error"]) <= 1e-5 * ro["error"]
    assert np.array_equal(h.vio_get_errors(m), ro["errors"])
    assert_delta_close(np.array(info.solution)[:18], np.array(ro["out"].solution))
    assert np.abs(h.vio_get_state18().vec() - xo.vec()).max() <= 1e-9
    h.close()


@pytest.mark.parametrize("m,max_iter", [(16384, 5), (70000, 3)])
def test_auto_wide_compute_j_matches_oracle_at_scale(gpu_lib, oracle_lib, scene, m, max_iter):
    """The same sizes through a whole ComputeJ (levels 2, 1, 0; one launch per pass on the wide producers; the reference's float running
    sum over all patches decides accept / revert, lidar_selection.cpp:849-861,888-892): per-level pass and accept counts equal the
    oracle's, per-patch errors bit for bit, state 1e-9, covariance 1e-12."""
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    fr, vf = _frames(synth, scene, m)
    vf.max_iterations = max_iter
    _oracle_threads(orc)
    try:
        xo = orc.state18_from_frame(fr)
        ro = orc.vio_compute_j(vf, xo, xo.copy())
    finally:
        orc.lib().orc_vio_set_threads(1)
    h = capi.Handle(capi.config_from_frames(fr, vf, max_iterations=max_iter))
    h.vio_set_frame(vf.img)
    h.vio_set_patches(vf.ref_patch, vf.pos, vf.search_level)
    xg = capi.state18_from_frame(fr)
    infos = h.vio_compute_j(xg, xg.copy())
    for lv in (2, 1, 0):
        assert infos[lv].iterations == ro["outs"][lv].iterations, lv
        assert infos[lv].accepted == ro["outs"][lv].accepted, lv
        assert abs(infos[lv].total_residual - ro["outs"][lv].error) <= 1e-5 * ro["outs"][lv].error
    assert np.array_equal(h.vio_get_errors(m), ro["errors"])
    assert np.abs(xg.vec() - xo.vec()).max() <= 1e-9
    assert np.abs(xg.cov_np() - xo.cov_np()).max() <= 1e-12
    h.close()
