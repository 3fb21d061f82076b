"""Abandon / resume, deterministically: the instrumented build's fault injector (fl_debug_drop_record) makes producer
workgroup 0 of a chosen pass withhold its record, the solver's bounded gather expires, the pass is ABANDONED (state untouched,
FL_NUM_TIMEOUT sticky, everything enqueued behind it skipped) and the synchronous entry points re-run exactly what was left:
same iteration count and the same bits as the undisturbed run, no status bit -- for one launch per pass (where the abandoned
pass itself used to go uncounted: a call with count == 1 came back FL_OK with the iteration silently dropped) and for the
multi-pass launches, for the 18-state LIO / VIO filters and the 23-state one."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _lio(capi, fr, nbr, valid, multipass):
    h = capi.Handle(capi.config_from_frames(fr, max_iterations=10), debug=True)
    h.set_option(capi.FL_OPT_MULTIPASS, 1 if multipass else 0)
    x0 = capi.state18_from_frame(fr)
    h.lio_set_points(fr.body_xyz); h.lio_begin18(x0, x0); h.lio_set_neighbours(nbr, valid)
    return h


@pytest.mark.parametrize("multipass", [False, True])
@pytest.mark.parametrize("count,drop", [(1, 0), (3, 0), (3, 1), (3, 2), (6, 4)])
def test_lio_dropped_record_is_resumed(gpu_lib, scene, multipass, count, drop):
    capi = gpu_lib
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(20000, scene=scene)
    nbr, valid = synth.knn5(scene, fr.world_at(fr.R_prior, fr.p_prior))
    F = capi.FL_ITER_FORCE
    ref = _lio(capi, fr, nbr, valid, multipass)
    iref = ref.lio_iterate18(count, F)
    xref = ref.lio_get_state18().vec()
    ref.close()
    h = _lio(capi, fr, nbr, valid, multipass)
    d0 = h.diagnostics()
    h.debug_drop_record(drop)
    info = h.lio_iterate18(count, F)
    d1 = h.diagnostics()
    assert d1["resumes"] - d0["resumes"] >= 1                 # the pass really was abandoned
    assert info.status == 0 and info.iterations == iref.iterations == count
    assert np.array_equal(h.lio_get_state18().vec(), xref)
    assert list(info.solution) == list(iref.solution)
    h.close()


@pytest.mark.parametrize("multipass", [False, True])
@pytest.mark.parametrize("count,drop", [(1, 0), (4, 2)])
def test_vio_dropped_record_is_resumed(gpu_lib, multipass, count, drop):
    capi = gpu_lib
    from fast_livo_amd import synth
    lio = synth.make_lio_frame(2000)
    vf = synth.make_vio_frame(700, lio)
    F = capi.FL_ITER_FORCE

    def make():
        h = capi.Handle(capi.config_from_frames(lio, vf, max_iterations=10), debug=True)
        h.set_option(capi.FL_OPT_MULTIPASS, 1 if multipass else 0)
        x0 = capi.state18_from_frame(lio)
        h.vio_set_frame(vf.img); h.vio_set_patches(vf.ref_patch, vf.pos, vf.search_level); h.vio_begin(x0, x0)
        return h
    ref = make()
    iref = ref.vio_iterate(0, count, F)
    xref = ref.vio_get_state18().vec()
    eref = ref.vio_get_errors(vf.m)
    ref.close()
    h = make()
    h.debug_drop_record(drop)
    info = h.vio_iterate(0, count, F)
    assert h.diagnostics()["resumes"] >= 1
    assert info.status == 0 and info.iterations == iref.iterations == count and info.accepted == iref.accepted
    assert np.array_equal(h.vio_get_state18().vec(), xref)
    assert np.array_equal(h.vio_get_errors(vf.m).view(np.uint32), eref.view(np.uint32))
    h.close()


@pytest.mark.parametrize("multipass", [False, True])
@pytest.mark.parametrize("count,drop", [(1, 0), (3, 1)])
def test_mode23_dropped_record_is_resumed(gpu_lib, scene, multipass, count, drop):
    capi = gpu_lib
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(15000, scene=scene)
    nbr, valid = synth.knn5(scene, fr.world_at(fr.R_prior, fr.p_prior))
    F = capi.FL_ITER_FORCE

    def make():
        h = capi.Handle(capi.config_from_frames(fr, max_iterations=10), debug=True)
        h.set_option(capi.FL_OPT_MULTIPASS, 1 if multipass else 0)
        h.lio_set_points(fr.body_xyz); h.ikfom_begin(capi.state23_from_frame(fr), fr.cov23.copy()); h.lio_set_neighbours(nbr, valid)
        return h
    ref = make()
    iref = ref.ikfom_iterate(count, F)
    xr, Pr = ref.ikfom_get()
    ref.close()
    h = make()
    h.debug_drop_record(drop)
    info = h.ikfom_iterate(count, F)
    x, P = h.ikfom_get()
    assert h.diagnostics()["resumes"] >= 1
    assert info.status == 0 and info.iterations == iref.iterations == count
    assert bytes(x) == bytes(xr) and np.array_equal(P, Pr)
    h.close()


def test_demotion_after_repeated_timeouts_and_recovery(gpu_lib, scene):
    """A foreign compute client the admission check cannot see makes multi-pass chains time out again and again, each costing a full
    bounded-wait stall before the per-pass resume. FL_OPT_DEMOTE_AFTER consecutive driver calls that ended so demote the handle to one
    launch per pass for FL_OPT_DEMOTE_CALLS calls; then it tries the multi-pass form again (a relapse on probation doubles the
    period). Deterministic here through the fault injector; results stay bit-identical through all of it."""
    capi = gpu_lib
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(20000, scene=scene)
    nbr, valid = synth.knn5(scene, fr.world_at(fr.R_prior, fr.p_prior))
    F = capi.FL_ITER_FORCE
    ref = _lio(capi, fr, nbr, valid, True)
    ref.lio_iterate18(4, F)
    xref = ref.lio_get_state18().vec()
    ref.close()

    h = capi.Handle(capi.config_from_frames(fr, max_iterations=10), debug=True)
    h.set_option(capi.FL_OPT_DEMOTE_AFTER, 2)
    h.set_option(capi.FL_OPT_DEMOTE_CALLS, 3)
    h.lio_set_points(fr.body_xyz); h.lio_set_neighbours(nbr, valid)
    h.debug_drop_record(1 << 30)                  # the injector's epoch is a global of the debug library: park what an earlier test left armed

    def call(drop):
        x0 = capi.state18_from_frame(fr)
        h.lio_begin18(x0, x0); h.lio_set_neighbours(nbr, valid)
        if drop is not None:
            h.debug_drop_record(drop)
        info = h.lio_iterate18(4, F)
        assert info.status == 0 and info.iterations == 4
        assert np.array_equal(h.lio_get_state18().vec(), xref)
        return h.diagnostics()
    d = call(None)
    assert d["demotions"] == 0 and d["demoted_calls_left"] == 0
    d = call(1)                                   # one time-out: not yet
    assert d["demotions"] == 0 and d["resumes"] >= 1
    d = call(None)                                # a clean call in between resets the count
    d = call(0)
    assert d["demotions"] == 0
    d = call(2)                                   # the second in a row: demoted for 3 calls
    assert d["demotions"] == 1 and d["demoted_calls_left"] == 3
    r0 = d["resumes"]
    for left in (2, 1, 0):
        d = call(None)                            # served per pass
        assert d["demoted_calls_left"] == left and d["resumes"] == r0
    d = call(1)                                   # probation: back on the multi-pass form, it times out at once -> demoted again, twice as long
    assert d["demotions"] == 2 and d["demoted_calls_left"] == 6
    for _ in range(6):
        d = call(None)
    assert d["demoted_calls_left"] == 0
    d = call(None)                                # a clean multi-pass call ends the probation and resets the period
    d = call(0); d = call(0)
    assert d["demotions"] == 3 and d["demoted_calls_left"] == 3
    h.set_option(capi.FL_OPT_DEMOTE_AFTER, 0)     # never demote: the override lifts a running demotion
    assert h.diagnostics()["demoted_calls_left"] == 0
    d = call(0); d = call(0); d = call(0)
    assert d["demotions"] == 3
    h.close()
