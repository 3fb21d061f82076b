"""Multi-pass LIO kernel (several ESKF passes in one launch, pose broadcast in-kernel) vs one launch per pass: bit-identical
states, selections and counters, with and without the convergence logic; whole all-device frame vs the per-pass path
(FL_NO_MULTIPASS=1 in a child process)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _prep(capi, synth, scene, n, max_iter=10):
    fr = synth.make_lio_frame(n, scene=scene)
    nbr, valid = synth.knn5(scene, fr.world_at(fr.R_prior, fr.p_prior))
    h = capi.Handle(capi.config_from_frames(fr, max_iterations=max_iter))
    x0 = capi.state18_from_frame(fr)
    h.lio_set_points(fr.body_xyz); h.lio_begin18(x0, x0); h.lio_set_neighbours(nbr, valid)
    return fr, h


@pytest.mark.parametrize("n", [300, 20000, 60000, 120000])   # 120000: grid-stride producers (255 workgroups)
def test_forced_multipass_equals_single_launches(gpu_lib, scene, n):
    capi = gpu_lib
    from fast_livo_amd import synth
    F = capi.FL_ITER_FORCE | capi.FL_ITER_KEEP_NORMVEC
    fr, h1 = _prep(capi, synth, scene, n)
    for _ in range(7):
        i1 = h1.lio_iterate18(1, F)
    fr, h2 = _prep(capi, synth, scene, n)
    i2 = h2.lio_iterate18(7, F)
    assert np.array_equal(h1.lio_get_state18().vec(), h2.lio_get_state18().vec())
    assert i1.iterations == i2.iterations == 7 and i1.effct_feat_num == i2.effct_feat_num
    assert list(i1.solution) == list(i2.solution)
    m1, v1 = h1.lio_get_selection(n); m2, v2 = h2.lio_get_selection(n)
    assert np.array_equal(m1, m2) and np.array_equal(v1.view(np.uint32), v2.view(np.uint32))
    # and a second multi-pass launch continues where the first stopped
    h1.lio_iterate18(1, F); h1.lio_iterate18(1, F)
    h2.lio_iterate18(2, F)
    assert np.array_equal(h1.lio_get_state18().vec(), h2.lio_get_state18().vec())


def test_convergence_logic_stops_the_launch(gpu_lib, scene):
    capi = gpu_lib
    from fast_livo_amd import synth
    fr, h1 = _prep(capi, synth, scene, 20000)
    infos = []
    for _ in range(11):
        infos.append(h1.lio_iterate18(1, 0))
        if infos[-1].need_search or infos[-1].stop:
            break
    fr, h2 = _prep(capi, synth, scene, 20000)
    i2 = h2.lio_iterate18(11, 0)
    assert i2.iterations == infos[-1].iterations and i2.need_search == infos[-1].need_search and i2.stop == infos[-1].stop
    assert np.array_equal(h1.lio_get_state18().vec(), h2.lio_get_state18().vec())
    # a further launch without a new search is a no-op, exactly like single launches
    before = h2.lio_get_state18().vec()
    i3 = h2.lio_iterate18(5, 0)
    assert i3.iterations == i2.iterations and np.array_equal(before, h2.lio_get_state18().vec())


_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import fastlivo
from fast_livo_amd import capi, synth
fr = synth.make_lio_frame(30000)
h = capi.Handle(capi.config_from_frames(fr, max_iterations=10))
h.map_set_points(fr.scene.map_xyz, 0.5)
x = capi.state18_from_frame(fr)
info = h.lio_frame18_dev(x, fr.body_xyz)
m, v = h.lio_get_selection(fr.n)
sys.stdout.buffer.write(bytes(x) + np.int32([info.iterations, info.effct_feat_num]).tobytes() + m.tobytes() + v.tobytes())
"""


def test_all_device_frame_same_with_and_without_multipass(gpu_lib):
    outs = []
    for env in ({}, {"FL_NO_MULTIPASS": "1"}):
        e = dict(os.environ); e.update(env)
        outs.append(subprocess.run([sys.executable, "-c", _CHILD % ROOT], env=e, check=True, capture_output=True).stdout)
    assert len(outs[0]) > 1000 and outs[0] == outs[1]


def _prep_vio(capi, synth, m, max_iter=10):
    lio = synth.make_lio_frame(2000)
    vf = synth.make_vio_frame(m, lio)
    h = capi.Handle(capi.config_from_frames(lio, vf, max_iterations=max_iter))
    x0 = capi.state18_from_frame(lio)
    h.vio_set_frame(vf.img); h.vio_set_patches(vf.ref_patch, vf.pos, vf.search_level); h.vio_begin(x0, x0)
    return lio, vf, h


@pytest.mark.parametrize("m,level", [(7, 0), (500, 1), (2000, 0)])
def test_vio_forced_multipass_equals_single_launches(gpu_lib, m, level):
    capi = gpu_lib
    from fast_livo_amd import synth
    F = capi.FL_ITER_FORCE
    _, _, h1 = _prep_vio(capi, synth, m)
    for _ in range(6):
        i1 = h1.vio_iterate(level, 1, F)
    _, _, h2 = _prep_vio(capi, synth, m)
    i2 = h2.vio_iterate(level, 6, F)
    assert np.array_equal(h1.vio_get_state18().vec(), h2.vio_get_state18().vec())
    assert list(i1.solution) == list(i2.solution) and i1.accepted == i2.accepted and i1.iterations == i2.iterations
    assert np.array_equal(h1.vio_get_errors(m).view(np.uint32), h2.vio_get_errors(m).view(np.uint32))


def test_vio_compute_j_same_with_and_without_multipass(gpu_lib):
    """ComputeJ (3 levels x up to max_iterations passes, accept/revert/stop logic) in a child process per mode."""
    child = r'''
import sys, numpy as np
sys.path.insert(0, %r)
import fastlivo
from fast_livo_amd import capi, synth
lio = synth.make_lio_frame(2000)
vf = synth.make_vio_frame(1500, lio)
h = capi.Handle(capi.config_from_frames(lio, vf, max_iterations=10))
x = capi.state18_from_frame(lio); xp = capi.state18_from_frame(lio)
h.vio_set_frame(vf.img); h.vio_set_patches(vf.ref_patch, vf.pos, vf.search_level)
infos = h.vio_compute_j(x, xp)
e = h.vio_get_errors(vf.m)
sys.stdout.buffer.write(bytes(x) + np.int32([i.iterations for i in infos] + [i.accepted for i in infos]).tobytes() + e.tobytes())
''' % ROOT
    outs = []
    for env in ({}, {"FL_NO_MULTIPASS": "1"}):
        e = dict(os.environ); e.update(env)
        outs.append(subprocess.run([sys.executable, "-c", child], env=e, check=True, capture_output=True).stdout)
    assert len(outs[0]) > 1000 and outs[0] == outs[1]


def test_non_finite_state_is_reported_not_hung(gpu_lib, scene):
    """A NaN in the incoming state poisons the sums: every pass must still terminate (bounded hand-offs) and report status bit 2;
    the handle stays usable for the next frame."""
    capi = gpu_lib
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(20000, scene=scene)
    nbr, valid = synth.knn5(scene, fr.world_at(fr.R_prior, fr.p_prior))
    h = capi.Handle(capi.config_from_frames(fr, max_iterations=10))
    bad = capi.state18_from_frame(fr)
    bad.pos[1] = float("nan")
    h.lio_set_points(fr.body_xyz); h.lio_begin18(bad, bad); h.lio_set_neighbours(nbr, valid)
    info = h.lio_iterate18(8, capi.FL_ITER_FORCE)
    assert info.iterations == 8 and (info.status & 2 or info.effct_feat_num == 0)
    assert not (info.status & 8)                       # no hand-off timeout
    # next frame on the same handle is clean
    x = capi.state18_from_frame(fr)
    h.lio_set_points(fr.body_xyz); h.lio_begin18(x, x); h.lio_set_neighbours(nbr, valid)
    good = h.lio_iterate18(3, capi.FL_ITER_FORCE)
    assert good.status == 0 and good.effct_feat_num > 1000 and np.isfinite(h.lio_get_state18().vec()).all()


def test_vio_kernel_variants_give_the_same_bits(gpu_lib):
    """The VIO multi-pass kernel exists in two register budgets (a CU per workgroup / two workgroups per CU, api_vio.inc
    vio_mp_variant) besides the one-launch-per-pass form: ComputeJ must not depend on which one a launch gets (the concurrent and
    sharded uses get the co-resident one). Fused multiply-adds are written out (fma()) wherever the two instantiations could contract
    differently."""
    capi = gpu_lib
    from fast_livo_amd import synth
    lio = synth.make_lio_frame(2000)
    vf = synth.make_vio_frame(1800, lio)
    outs = []
    for opts in ({}, {capi.FL_OPT_VIO_WHOLE_CU: 0}, {capi.FL_OPT_MULTIPASS: 0}):
        h = capi.Handle(capi.config_from_frames(lio, vf, max_iterations=10))
        for k, v in opts.items():
            h.set_option(k, v)
        h.vio_set_frame(vf.img); h.vio_set_patches(vf.ref_patch, vf.pos, vf.search_level)
        x = capi.state18_from_frame(lio)
        infos = h.vio_compute_j(x, capi.state18_from_frame(lio))
        outs.append((x.vec().copy(), x.cov_np().copy(), h.vio_get_errors(vf.m).view(np.uint32).copy(),
                     [(i.iterations, i.accepted, i.status, i.total_residual, tuple(i.solution)) for i in infos]))
        h.close()
    for o in outs[1:]:
        assert np.array_equal(o[0], outs[0][0]) and np.array_equal(o[1], outs[0][1]) and np.array_equal(o[2], outs[0][2]) and o[3] == outs[0][3]


def test_result_mailbox_gives_the_same_frame(gpu_lib, scene):
    """The frame drivers get their results through a host word the frame's last kernel writes (fl_publish_state, FL_OPT_MAILBOX) or
    through a copy + stream synchronisation: the same state, covariance, counters, level results -- also over repeated frames on one
    handle and for a frame whose chain is resumed."""
    capi = gpu_lib
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(20000, scene=scene)
    vf = synth.make_vio_frame(1500, fr)
    outs = []
    for mailbox in (3, 0, 2):
        h = capi.Handle(capi.config_from_frames(fr, vf, max_iterations=10))
        h.set_option(capi.FL_OPT_MAILBOX, mailbox)
        h.map_set_points(scene.map_xyz, 0.5)
        h.vio_set_frame(vf.img); h.vio_set_patches(vf.ref_patch, vf.pos, vf.search_level)
        rec = []
        for rep in range(4):
            x = capi.state18_from_frame(fr)
            info = h.lio_frame18_dev(x, fr.body_xyz)
            xv = capi.state18_from_frame(fr)
            infos = h.vio_compute_j(xv, capi.state18_from_frame(fr))
            rec.append((x.vec().copy(), x.cov_np().copy(), info.status, info.iterations, info.effct_feat_num, xv.vec().copy(), xv.cov_np().copy(),
                        [(i.iterations, i.accepted, i.status, i.total_residual, tuple(i.solution)) for i in infos]))
        outs.append(rec)
        h.close()
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:5] == b[2:5]
            assert np.array_equal(a[5], b[5]) and np.array_equal(a[6], b[6]) and a[7] == b[7]
    for rec in outs:                    # and frame after frame on one handle
        for r in rec[1:]:
            assert np.array_equal(r[0], rec[0][0]) and np.array_equal(r[5], rec[0][5])


def test_scan_fetched_by_the_search_kernel_gives_the_same_frame(gpu_lib, scene):
    """A scan in fl_host_alloc memory is fetched by the frame's first search kernel over the host link (FL_OPT_SCAN_PULL) instead of
    a copy command: same frame as from pageable memory and as with the option off; the device copy it leaves behind serves a
    following frame on the staged scan (body None); odd sizes (a last workgroup with fewer than 64 points)."""
    capi = gpu_lib
    from fast_livo_amd import synth
    for n in (20000, 12345, 65):
        fr = synth.make_lio_frame(n, scene=scene)
        ref = None
        for pull, pinned in ((1, True), (0, True), (1, False)):
            h = capi.Handle(capi.config_from_frames(fr, max_iterations=10))
            h.set_option(capi.FL_OPT_SCAN_PULL, pull)
            h.map_set_points(scene.map_xyz, 0.5)
            scan = fr.body_xyz
            if pinned:
                scan = h.host_alloc(fr.body_xyz.shape, np.float32)
                scan[:] = fr.body_xyz
            out = []
            for rep in range(2):
                x = capi.state18_from_frame(fr)
                info = h.lio_frame18_dev(x, scan)
                out.append((x.vec().copy(), x.cov_np().copy(), info.status, info.iterations, info.effct_feat_num))
            x = capi.state18_from_frame(fr)
            info = h.lio_frame18_dev(x, None)                  # the scan as the last frame left it on the device
            out.append((x.vec().copy(), x.cov_np().copy(), info.status, info.iterations, info.effct_feat_num))
            mask, normvec = h.lio_get_selection(n)
            out.append((mask.copy(), normvec.view(np.uint32).copy()))
            if pinned:
                h.host_free(scan)
            h.close()
            if ref is None:
                ref = out
            else:
                for a, b in zip(ref[:3], out[:3]):
                    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:] == b[2:], (n, pull, pinned)
                assert np.array_equal(ref[3][0], out[3][0]) and np.array_equal(ref[3][1], out[3][1])
            assert out[0][2] == 0 and np.array_equal(out[0][0], out[2][0])


def test_mode23_update_with_the_scan_fetched_by_the_search_kernel(gpu_lib, scene):
    """fl_ikfom_update_iterated_dev with the scan in fl_host_alloc memory (fetched by the first search kernel) against the same update
    from pageable memory (copy command): the same state, covariance and counters."""
    capi = gpu_lib
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(12345, scene=scene)
    outs = []
    for pinned in (True, False):
        h = capi.Handle(capi.config_from_frames(fr, max_iterations=10))
        h.map_set_points(scene.map_xyz, 0.5)
        scan = fr.body_xyz
        if pinned:
            scan = h.host_alloc(fr.body_xyz.shape, np.float32)
            scan[:] = fr.body_xyz
        rec = []
        for rep in range(2):
            x23 = capi.state23_from_frame(fr)
            P = fr.cov23.copy()
            info = h.ikfom_update_iterated_dev(x23, P, scan, 0.001)
            rec.append((x23.vec().copy(), P.copy(), info.status, info.iterations, info.effct_feat_num))
        outs.append(rec)
        if pinned:
            h.host_free(scan)
        h.close()
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2:] == b[2:]
    assert outs[0][0][2] == 0


@pytest.mark.parametrize("n", [63, 12345, 50000])
def test_mode23_update_driver_variants_give_the_same_update(gpu_lib, oracle_lib, scene, n):
    """Round 4: fl_ikfom_update_iterated_dev pulls the state block from the page-locked mirror in its first search kernel, lets that
    kernel write the gate thresholds, runs two launch segments and gets its result through the mailbox. Every combination of
    mailbox on / off, scan + state pulled / copied, multi-pass / one launch per pass, on consecutive updates of one handle (a
    stale mailbox or a stale winner record would show in the second one): the same state, covariance and counters, bit for
    bit -- and equal to the oracle's update."""
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(n, scene=scene)

    def knn(w):
        nb, _, va, _ = orc.knn5_bruteforce(scene.map_xyz, w)
        return nb, va
    xo = orc.state23_from_frame(fr, synth.quat_from_R)
    Po = fr.cov23.copy()
    ro = orc.ikfom_update(xo, Po, fr.body_xyz, 0.001, 10, knn)
    ref = None
    for mailbox in (3, 0):
        for pull in (1, 0):
            for multipass in (1, 0):
                h = capi.Handle(capi.config_from_frames(fr, max_iterations=10))
                h.set_option(capi.FL_OPT_MAILBOX, mailbox)
                h.set_option(capi.FL_OPT_SCAN_PULL, pull)
                h.set_option(capi.FL_OPT_MULTIPASS, multipass)
                h.map_set_points(scene.map_xyz, 0.5)
                scan = h.host_alloc(fr.body_xyz.shape, np.float32)
                scan[:] = fr.body_xyz
                for rep in range(3):
                    x23 = capi.state23_from_frame(fr)
                    P = fr.cov23.copy()
                    info = h.ikfom_update_iterated_dev(x23, P, scan, 0.001)
                    out = (x23.vec().copy(), P.copy(), int(info.status), int(info.iterations), int(info.effct_feat_num))
                    if ref is None:
                        ref = out
                        assert out[2] == 0 and out[3] == ro["out"].iterations and out[4] == ro["out"].effct_feat_num
                        assert np.abs(out[0] - xo.vec()).max() <= 1e-9 and np.abs(out[1] - Po).max() <= 1e-10
                    assert np.array_equal(out[0], ref[0]) and np.array_equal(out[1], ref[1]) and out[2:] == ref[2:], (mailbox, pull, multipass, rep)
                h.host_free(scan)
                h.close()
