"""exact_chain.h: the binade-wise, lane-parallel form of the reference's float running sum `error += patch_error`
(lidar_selection.cpp:849-857) must equal the plain chain of additions bit for bit -- on patch-error-like data and on inputs built to
hit its corners (exact ties, binade crossings in every lane, subnormals, zeros, inf, huge dynamic range). The per-lane phases are
the product's own functions (compiled for the host by tests/host_emul); the wave-level glue (ballots, prefix sum) is emulated
here and tested on the device by tests/test_exact_chain_gpu.py."""
import ctypes as C
import numpy as np
import pytest


def _run(E, e, init=0.0):
    e = np.ascontiguousarray(e, dtype=np.float32)
    E.emul_chain_f32.restype = C.c_float
    E.emul_chain_f32_plain.restype = C.c_float
    steps = C.c_int(0)
    pe = e.ctypes.data_as(C.POINTER(C.c_float))
    a = E.emul_chain_f32(pe, len(e), C.c_float(init), C.byref(steps))
    b = E.emul_chain_f32_plain(pe, len(e), C.c_float(init))
    return np.float32(a), np.float32(b), steps.value


def _same(a, b):
    return a.tobytes() == b.tobytes() or (np.isnan(a) and np.isnan(b))


def chain_cases(rng, n_cases=300):
    """(name, array, init) -- shared with the GPU test"""
    out = []
    for i in range(n_cases):
        m = int(rng.choice([0, 1, 2, 15, 16, 17, 63, 64, 65, 255, 256, 257, 271, 272, 273, 1000, 2000, 2048, 5000]))
        kind = i % 10
        if kind == 0:      # patch errors: sums of 64 squared residuals of a few grey levels
            e = (rng.standard_normal((m, 64)).astype(np.float32) * rng.uniform(0.5, 30) ) ** 2
            e = e.sum(axis=1, dtype=np.float32)
        elif kind == 1:    # small integers: every addition exact until 2^24, then ties everywhere
            e = rng.integers(0, 5, m).astype(np.float32) * np.float32(2.0 ** rng.integers(-3, 20))
        elif kind == 2:    # exact ties against a large running sum: multiples of half an ulp of 2^k
            k = int(rng.integers(0, 30))
            e = (rng.integers(0, 64, m).astype(np.float64) * 2.0 ** (k - 24)).astype(np.float32)
            if m: e[0] = np.float32(2.0 ** k)
        elif kind == 3:    # huge dynamic range
            e = np.exp(rng.uniform(-80, 60, m)).astype(np.float32)
        elif kind == 4:    # subnormals and zeros
            e = (rng.integers(0, 1 << 10, m).astype(np.float64) * 2.0 ** -149).astype(np.float32)
            e[rng.random(m) < 0.3] = 0.0
        elif kind == 5:    # geometric growth: a binade crossing every element
            e = (2.0 ** np.arange(m, dtype=np.float64).clip(0, 120) * rng.uniform(1, 2, m)).astype(np.float32)
        elif kind == 6:    # near-equal values (the converged case) with ulp-level noise
            e = (np.float32(rng.uniform(1, 1e4)) * (1 + rng.integers(-3, 4, m) * 2.0 ** -23)).astype(np.float32)
        elif kind == 7:    # an inf in the middle
            e = rng.uniform(0, 100, m).astype(np.float32)
            if m > 3: e[m // 2] = np.inf
        elif kind == 8:    # ties with odd/even alternation: 0.5 ulp steps at 2^23
            e = np.full(m, 0.5, dtype=np.float32)
            if m: e[0] = np.float32(2.0 ** 23)
        else:              # all zeros / one big then tiny
            e = np.zeros(m, dtype=np.float32)
            if m > 2: e[1] = 1e30; e[2:] = rng.uniform(0, 1e22, m - 2).astype(np.float32)
        init = float(np.float32(rng.choice([0.0, 0.0, 1.0, 3.5e6, 2.0 ** 24, 1e-40])))
        out.append((f"case{i}_kind{kind}_m{m}", e.astype(np.float32), init))
    return out


def test_lane_parallel_chain_equals_the_plain_chain(emul_lib):
    rng = np.random.default_rng(20240924)
    worst_steps = 0
    for name, e, init in chain_cases(rng, 600):
        a, b, steps = _run(emul_lib, e, init)
        assert _same(a, b), (name, a, b)
        if "kind0" in name and len(e) == 2000:
            worst_steps = max(worst_steps, steps)
    # 2 k patch errors: ~8 full steps + one per binade climbed after the 16 leading additions
    assert 0 < worst_steps <= 24, worst_steps


def test_negative_or_nan_elements_fall_back_to_the_plain_chain(emul_lib):
    rng = np.random.default_rng(5)
    e = rng.uniform(0, 10, 1000).astype(np.float32)
    e[400] = -3.0
    a, b, _ = _run(emul_lib, e)
    assert _same(a, b)
    e[700] = np.nan
    a, b, _ = _run(emul_lib, e)
    assert np.isnan(a) and np.isnan(b)


@pytest.mark.parametrize("seed", range(4))
def test_random_bit_patterns(emul_lib, seed):
    """non-negative floats drawn uniformly over the BIT patterns (every exponent equally likely)"""
    rng = np.random.default_rng(100 + seed)
    for _ in range(200):
        m = int(rng.integers(1, 3000))
        bits = rng.integers(0, 0x7f000000, m, dtype=np.int64).astype(np.uint32)
        lo, hi = sorted(rng.integers(0, 250, 2))
        bits = (bits & np.uint32(0x007fffff)) | (rng.integers(lo, hi + 1, m).astype(np.uint32) << np.uint32(23))
        e = bits.view(np.float32)
        a, b, _ = _run(emul_lib, e, 0.0)
        assert _same(a, b), (seed, m, a, b)


# ---- the workgroup form (fl_chain_f32_block): binades guessed from a double prefix sum, every guess checked, wavefront form on a
# failed check
def _run_spec(E, e, init=0.0):
    e = np.ascontiguousarray(e, dtype=np.float32)
    E.emul_chain_f32_spec.restype = C.c_float
    E.emul_chain_f32_plain.restype = C.c_float
    fb, nev = C.c_int(0), C.c_int(0)
    pe = e.ctypes.data_as(C.POINTER(C.c_float))
    a = E.emul_chain_f32_spec(pe, len(e), C.c_float(init), C.byref(fb), C.byref(nev))
    b = E.emul_chain_f32_plain(pe, len(e), C.c_float(init))
    return np.float32(a), np.float32(b), fb.value, nev.value


def test_workgroup_chain_equals_the_plain_chain(emul_lib):
    rng = np.random.default_rng(20240924)
    n = fell = 0
    for name, e, init in chain_cases(rng, 600):
        if len(e) > 2048:
            continue
        a, b, fb, nev = _run_spec(emul_lib, e, init)
        assert _same(a, b), (name, a, b, fb, nev)
        n += 1
        fell += fb
    assert n > 400 and fell < n          # (the corner cases fall back often; the point is the bits)


def test_workgroup_chain_on_patch_errors_rarely_falls_back(emul_lib):
    """the data the kernels see: 2 k sums of 64 squared grey-level residuals, near-converged passes"""
    rng = np.random.default_rng(77)
    fell = 0
    worst_events = 0
    trials = 400
    for _ in range(trials):
        m = int(rng.choice([500, 1000, 2000, 2048]))
        e = ((rng.standard_normal((m, 64)).astype(np.float32) * rng.uniform(0.5, 30)) ** 2).sum(axis=1, dtype=np.float32)
        a, b, fb, nev = _run_spec(emul_lib, e, 0.0)
        assert _same(a, b), (m, a, b, fb, nev)
        fell += fb
        if not fb:
            worst_events = max(worst_events, nev)
    print(f"\n[workgroup chain] {fell} of {trials} chains fell back to the wavefront form; at most {worst_events} events walked")
    assert fell <= trials // 10
    assert 0 < worst_events <= 63


@pytest.mark.parametrize("seed", range(4))
def test_workgroup_chain_random_bit_patterns(emul_lib, seed):
    rng = np.random.default_rng(300 + seed)
    for _ in range(300):
        m = int(rng.integers(1, 2049))
        lo, hi = sorted(rng.integers(0, 250, 2))
        bits = (rng.integers(0, 1 << 23, m).astype(np.uint32)) | (rng.integers(lo, hi + 1, m).astype(np.uint32) << np.uint32(23))
        init = float(np.float32(rng.choice([0.0, 1.0, 2.0 ** 24, 1e-40, 3.0e38])))
        with np.errstate(over="ignore"):
            a, b, fb, nev = _run_spec(emul_lib, bits.view(np.float32), init)
        assert _same(a, b), (seed, m, a, b, fb, nev)


def test_workgroup_chain_binade_edges(emul_lib):
    """running sums that land exactly on / one ulp beside powers of two, where the double prefix and the float chain may disagree about
    the binade: the checks must catch every such case (result still the plain chain's)"""
    rng = np.random.default_rng(9)
    for trial in range(400):
        m = int(rng.integers(2, 1500))
        k = int(rng.integers(-20, 40))
        e = rng.uniform(0, 1, m).astype(np.float64)
        e *= (2.0 ** k) / e.sum()                         # the exact sum is (about) 2^k
        j = int(rng.integers(1, m))
        e[:j] *= (2.0 ** (k - 1)) / e[:j].sum()           # ... and a prefix is (about) 2^(k-1)
        e = e.astype(np.float32)
        if trial % 3 == 0:
            e = np.concatenate([e, e[: m // 2]])[:2048]
        a, b, fb, nev = _run_spec(emul_lib, e, 0.0)
        assert _same(a, b), (trial, a, b, fb, nev)
