"""FL_OPT_VIO_SPECULATE (csrc/solve18.h vio_spec_confirm, round 5): in fl_vio_compute_j a pass whose accept test (lidar_selection.cpp:857-861)
is decided inside float rounding noise AND accepted by the fp64 test goes ahead without waiting for the reference's float running sum; the
sum's verdict is applied a pass later, with a roll-back (:888-892 semantics) when it disagrees.  The speculating form, the waiting form
(option 0) and the CPU oracle must agree in every bit of the state, of the per-patch errors and in the iteration / accept counts -- on
frames where the speculation is confirmed AND on the ten frames fuzz runs (tools/fuzz_vio_spec.py, seeds 1-4) found where it is
rolled back.  Round 6: ComputeJ's three pyramid levels are one launch (option 2, the default) and a fragile accept that ENDS a level goes
ahead too -- the next level begins on its state and takes the verdict behind its first pass's records; a rejection re-opens the finished
level (state, result block) and starts the begun one again.  Six of the ten frames roll back across a level.  The debug library counts what
happened (fragile accepts that went ahead / confirmed / rolled back / rolled back across a level / verdicts carried into the next level)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# (m, iteration cap, patch seed, lio seed, noise; what the debug library counts in the one-launch form: roll-backs, of which ACROSS a pyramid level)
ROLLBACK_CASES = [dict(m=2000, max_iter=7, seed=1028591, lio_seed=3, noise=6.0, rolled=1, across=1), dict(m=2000, max_iter=6, seed=25685, lio_seed=0, noise=6.0, rolled=1, across=1),
                  dict(m=700, max_iter=5, seed=607895, lio_seed=2, noise=2.0, rolled=1, across=0),
                  # (round 6, found with all pyramid levels in one launch and the verdict of a level's last pass taken by the next level)
                  dict(m=1000, max_iter=3, seed=641667, lio_seed=4, noise=6.0, rolled=1, across=1), dict(m=2040, max_iter=7, seed=1008255, lio_seed=6, noise=6.0, rolled=1, across=0),
                  dict(m=1000, max_iter=9, seed=688278, lio_seed=8, noise=2.0, rolled=1, across=1), dict(m=2000, max_iter=4, seed=46139, lio_seed=5, noise=6.0, rolled=1, across=1),
                  dict(m=2040, max_iter=4, seed=843490, lio_seed=10, noise=6.0, rolled=2, across=1), dict(m=2000, max_iter=10, seed=601349, lio_seed=1, noise=6.0, rolled=1, across=0),
                  dict(m=1000, max_iter=6, seed=522677, lio_seed=1, noise=2.0, rolled=1, across=0)]
CONFIRM_CASES = [dict(m=2000, max_iter=10, seed=103, lio_seed=0, noise=2.0), dict(m=1000, max_iter=4, seed=7, lio_seed=5, noise=0.5)]


def _run(capi, orc, synth, c):
    lio = synth.make_lio_frame(500, seed=synth.SEED + c["lio_seed"])
    vf = synth.make_vio_frame(c["m"], lio, max_iterations=c["max_iter"], patch_seed=c["seed"], ref_noise=c["noise"])
    res, counts = [], None
    for spec in (2, 1, 0):           # 2 (default): all three pyramid levels in ONE launch (round 6), 1: a launch per level, 0: the waiting form
        h = capi.Handle(capi.config_from_frames(lio, vf, max_iterations=c["max_iter"]), debug=True)
        h.set_option(capi.FL_OPT_VIO_SPECULATE, spec)
        w0 = np.array(h.debug_wall(), dtype=np.int64)[2040:2045].copy()
        xg = capi.state18_from_frame(lio); xp = capi.state18_from_frame(lio)
        h.vio_set_frame(vf.img); h.vio_set_patches(vf.ref_patch, vf.pos, vf.search_level)
        ig = h.vio_compute_j(xg, xp)
        eg = h.vio_get_errors(c["m"])
        if spec:
            c_now = np.array(h.debug_wall(), dtype=np.int64)[2040:2045] - w0      # speculated, confirmed, rolled back, rolled back across a level, verdicts carried into the next level
            # (the two speculating forms need not speculate on the same passes: behind a roll-back across a level the auditor workgroup is a
            # pass late for a while -- it adds the dropped pass's chain up before it sees the restart -- and a fragile accept whose
            # predecessor's total is not in its ring waits instead of going ahead. Same bits either way; that is what is asserted below.)
            assert c_now[0] == c_now[1] + c_now[2], c_now                        # every verdict that was out has been taken
            if counts is None:
                counts = c_now                                                   # (the one-launch form's: only it carries verdicts across levels)
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
    # fragile accepts that went ahead and were taken back, and how many of them after the next pyramid level had begun on their state. The
    # FIRST roll-back of a frame is reproducible (the passes before it run at the regular cadence); a second one depends on the auditor's timing.
    assert 1 <= counts[2] <= case["rolled"] and counts[0] == counts[1] + counts[2], counts
    assert counts[3] <= case["across"] and (counts[3] == 1 or not (case["rolled"] == 1 and case["across"] == 1)), counts


@pytest.mark.parametrize("case", CONFIRM_CASES, ids=lambda c: f"m{c['m']}-seed{c['seed']}")
def test_confirmed_speculation_changes_nothing(gpu_lib, oracle_lib, case):
    from fast_livo_amd import synth
    counts = _run(gpu_lib, oracle_lib, synth, case)
    assert counts[2] == 0 and counts[0] == counts[1]
