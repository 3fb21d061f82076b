"""The device form of exact_chain.h (ballots, DPP prefix sum, readlane) against one lane adding one by one ON THE DEVICE and
against numpy's float32 loop on the host: the same bits, on the case list of tests/test_exact_chain_cpu.py."""
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
            a, b = h.debug_chain(e, init)
            assert a.tobytes() == b.tobytes() or (np.isnan(a) and np.isnan(b)), (name, a, b)
            if len(e) <= 2048:
                c = _plain(e, init)
                assert b.tobytes() == c.tobytes() or (np.isnan(b) and np.isnan(c)), (name, b, c)
            n_checked += 1
        for seed in range(100):          # uniform over bit patterns
            m = int(rng.integers(1, 6000))
            lo, hi = sorted(rng.integers(0, 250, 2))
            bits = (rng.integers(0, 1 << 23, m).astype(np.uint32)) | (rng.integers(lo, hi + 1, m).astype(np.uint32) << np.uint32(23))
            a, b = h.debug_chain(bits.view(np.float32), 0.0)
            assert a.tobytes() == b.tobytes() or (np.isnan(a) and np.isnan(b)), (seed, m, a, b)
    h.close()
    assert n_checked == 400
