"""The device forms of exact_chain.h -- the workgroup form the kernels run (double prefix scan, guessed binades, checked walk over the
events) and the wavefront form it falls back to (ballots, DPP prefix sum, readlane) -- against one lane adding one by one ON THE
DEVICE and against numpy's float32 loop on the host: the same bits, on the case list of tests/test_exact_chain_cpu.py."""
import numpy as np
import pytest

from test_exact_chain_cpu import chain_cases

pytestmark = pytest.mark.gpu


def _plain(e, init):
    s = np.float32(init)
    for v in e:
        s = np.float32(s + v)
    return s


def test_device_chain_equals_plain_chain(gpu_lib, scene):
    capi = gpu_lib
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(2000, scene=scene)
    vf = synth.make_vio_frame(64, fr)
    h = capi.Handle(capi.config_from_frames(fr, vf), debug=True)     # fl_debug_chain: instrumented build
    rng = np.random.default_rng(20240924)
    n_checked = 0
    with np.errstate(over="ignore", invalid="ignore"):
        for name, e, init in chain_cases(rng, 400):
            a, b, w, fell = h.debug_chain(e, init, full=True)
            assert a.tobytes() == b.tobytes() or (np.isnan(a) and np.isnan(b)), (name, a, b, fell)
            assert w.tobytes() == b.tobytes() or (np.isnan(w) and np.isnan(b)), (name, w, b)
            if len(e) <= 2048:
                c = _plain(e, init)
                assert b.tobytes() == c.tobytes() or (np.isnan(b) and np.isnan(c)), (name, b, c)
            n_checked += 1
        for seed in range(100):          # uniform over bit patterns
            m = int(rng.integers(1, 6000))
            lo, hi = sorted(rng.integers(0, 250, 2))
            bits = (rng.integers(0, 1 << 23, m).astype(np.uint32)) | (rng.integers(lo, hi + 1, m).astype(np.uint32) << np.uint32(23))
            a, b, w, fell = h.debug_chain(bits.view(np.float32), 0.0, full=True)
            assert a.tobytes() == b.tobytes() or (np.isnan(a) and np.isnan(b)), (seed, m, a, b, fell)
            assert w.tobytes() == b.tobytes() or (np.isnan(w) and np.isnan(b)), (seed, m, w, b)
    h.close()
    assert n_checked == 400


def test_workgroup_form_on_patch_errors_and_binade_edges(gpu_lib, scene):
    """patch-error-like data (the workgroup form must not fall back more than now and then) and sums built to land on powers of two
    (where its guesses go wrong and its checks must notice)"""
    capi = gpu_lib
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(2000, scene=scene)
    vf = synth.make_vio_frame(64, fr)
    h = capi.Handle(capi.config_from_frames(fr, vf), debug=True)
    rng = np.random.default_rng(77)
    fell_total = 0
    for _ in range(150):
        m = int(rng.choice([500, 1000, 2000, 2048, 5000]))
        e = ((rng.standard_normal((m, 64)).astype(np.float32) * rng.uniform(0.5, 30)) ** 2).sum(axis=1, dtype=np.float32)
        a, b, w, fell = h.debug_chain(e, 0.0, full=True)
        assert a.tobytes() == b.tobytes() == w.tobytes(), (m, a, b, w, fell)
        fell_total += fell
    assert fell_total <= 15, fell_total
    edge_fell = 0
    for trial in range(300):
        m = int(rng.integers(2, 1500))
        k = int(rng.integers(-20, 40))
        e = rng.uniform(0, 1, m).astype(np.float64)
        e *= (2.0 ** k) / e.sum()
        j = int(rng.integers(1, m))
        e[:j] *= (2.0 ** (k - 1)) / e[:j].sum()
        a, b, w, fell = h.debug_chain(e.astype(np.float32), 0.0, full=True)
        assert a.tobytes() == b.tobytes() == w.tobytes(), (trial, a, b, w, fell)
        edge_fell += fell
    print(f"\n[workgroup chain] fell back in {fell_total} chunks of patch-error data, {edge_fell} of 300 binade-edge chains")
    assert edge_fell > 0                     # the checks were exercised
    h.close()
