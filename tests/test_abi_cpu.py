"""CPU-side checks of the C ABI: the library builds for gfx950, loads, and exports every symbol
include/fastlivo_hip.h declares.  No compute calls (no GPU here)."""
import os
import re

import pytest

import fastlivo  # noqa: F401
from fast_livo_amd import capi, LIB_PATH, REPO_ROOT


def test_library_builds_and_loads():
    capi.build()
    assert os.path.exists(LIB_PATH)
    capi.lib()


def test_every_declared_symbol_is_exported_and_bound():
    hdr = open(os.path.join(REPO_ROOT, "include", "fastlivo_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int32_t|const char \*)\s*\*?\s*(fl_[a-z0-9_]+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 30
    assert declared == set(capi.SYMBOLS), (declared ^ set(capi.SYMBOLS))
    L = capi.lib()
    for name in declared:
        assert hasattr(L, name), name


def test_struct_sizes_match_header():
    import ctypes as C
    assert C.sizeof(capi.State18) == 8 * (24 + 324)
    assert C.sizeof(capi.State23) == 8 * 26
    assert C.sizeof(capi.IterInfo) == 8 * 24 + 8 * 4
    assert C.sizeof(capi.Config) == 6 * 4 + 8 * (9 + 3 + 9 + 3 + 4 + 5 + 2)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU present")
def test_create_fails_loudly_without_gpu():
    from fast_livo_amd import synth
    cfg = capi.make_config(synth.AVIA_R_LI, synth.AVIA_T_LI, synth.AVIA_RCL, synth.AVIA_PCL, synth.PINHOLE)
    with pytest.raises(capi.FlError):
        capi.Handle(cfg)
