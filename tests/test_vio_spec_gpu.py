"""FL_OPT_VIO_SPECULATE (csrc/solve18.h vio_spec_confirm, round 5): in fl_vio_compute_j a pass whose accept test (lidar_selection.cpp:857-861)
is decided inside float rounding noise AND accepted by the fp64 test goes ahead without waiting for the reference's float running sum; the
sum's verdict is applied a pass later, with a roll-back (:888-892 semantics) when it disagrees.  The speculating form, the waiting form
(option 0) and the CPU oracle must agree in every bit of the state, of the per-patch errors and in the iteration / accept counts -- on
frames where the speculation is confirmed AND on the three frames a fuzz run (tools/fuzz_vio_spec.py, seeds 1 and 2) found where it is
rolled back.  The debug library counts what happened (fragile accepts that went ahead / confirmed / rolled back)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROLLBACK_CASES = [dict(m=2000, max_iter=7, seed=1028591, lio_seed=3, noise=6.0), dict(m=2000, max_iter=6, seed=25685, lio_seed=0, noise=6.0),
                  dict(m=700, max_iter=5, seed=607895, lio_seed=2, noise=2.0),
                  # (round 6, found with all pyramid levels in one launch)
                  dict(m=1000, max_iter=3, seed=641667, lio_seed=4, noise=6.0), dict(m=2040, max_iter=7, seed=1008255, lio_seed=6, noise=6.0)]
CONFIRM_CASES = [dict(m=2000, max_iter=10, seed=103, lio_seed=0, noise=2.0), dict(m=1000, max_iter=4, seed=7, lio_seed=5, noise=0.5)]


def _run(capi, orc, synth, c):
    lio = synth.make_lio_frame(500, seed=synth.SEED + c["lio_seed"])
    vf = synth.make_vio_frame(c["m"], lio, max_iterations=c["max_iter"], patch_seed=c["seed"], ref_noise=c["noise"])
    res, counts = [], None
    for spec in (2, 1, 0):           # 2 (default): all three pyramid levels in ONE launch (round 6), 1: a launch per level, 0: the waiting form
        h = capi.Handle(capi.config_from_frames(lio, vf, max_iterations=c["max_iter"]), debug=True)
        h.set_option(capi.FL_OPT_VIO_SPECULATE, spec)
        w0 = np.array(h.debug_wall(), dtype=np.int64)[2040:2043].copy()
        xg = capi.state18_from_frame(lio); xp = capi.state18_from_frame(lio)
        h.vio_set_frame(vf.img); h.vio_set_patches(vf.ref_patch, vf.pos, vf.search_level)
        ig = h.vio_compute_j(xg, xp)
        eg = h.vio_get_errors(c["m"])
        if spec:
            c_now = np.array(h.debug_wall(), dtype=np.int64)[2040:2043] - w0
            assert counts is None or np.array_equal(counts, c_now), (counts, c_now)      # both speculating forms speculate on the same passes
            counts = c_now
        res.append((bytes(xg), eg.copy(), [(int(i.iterations), int(i.accepted), int(i.status)) for i in ig]))
        h.close()
    xo = orc.state18_from_frame(lio)
    ro = orc.vio_compute_j(vf, xo, orc.state18_from_frame(lio))
    for r in res[1:]:
        assert res[0][0] == r[0]                                        # state + covariance: bit for bit
        assert np.array_equal(res[0][1].view(np.uint32), r[1].view(np.uint32)) and res[0][2] == r[2]
    xs = np.frombuffer(res[0][0], np.float64)
    assert np.abs(xs[:24] - xo.vec()[:24]).max() <= 1e-9
    assert np.array_equal(res[0][1].view(np.uint32), ro["errors"].view(np.uint32))
    assert [r[0] for r in res[0][2]] == [int(o.iterations) for o in ro["outs"]]
    return counts


@pytest.mark.parametrize("case", ROLLBACK_CASES, ids=lambda c: f"m{c['m']}-seed{c['seed']}")
def test_rejected_speculation_rolls_back_to_the_waiting_forms_result(gpu_lib, oracle_lib, case):
    from fast_livo_amd import synth
    counts = _run(gpu_lib, oracle_lib, synth, case)
    assert counts[2] == 1 and counts[0] == counts[1] + counts[2], counts      # one fragile accept went ahead and was taken back


@pytest.mark.parametrize("case", CONFIRM_CASES, ids=lambda c: f"m{c['m']}-seed{c['seed']}")
def test_confirmed_speculation_changes_nothing(gpu_lib, oracle_lib, case):
    from fast_livo_amd import synth
    counts = _run(gpu_lib, oracle_lib, synth, case)
    assert counts[2] == 0 and counts[0] == counts[1]
