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


def test_debug_entry_points_live_in_the_instrumented_build_only():
    """include/fastlivo_hip_debug.h: declared == bound == exported by libfastlivo_hip_debug.so; the release library exports no
    fl_debug_* symbol, no stamp global, and reads nothing from the environment."""
    import subprocess
    hdr = open(os.path.join(REPO_ROOT, "include", "fastlivo_hip_debug.h")).read()
    declared = set(re.findall(r"^\s*int32_t\s+(fl_debug_[a-z0-9_]+)\s*\(", hdr, flags=re.M))
    assert declared == set(capi.DEBUG_SYMBOLS), declared ^ set(capi.DEBUG_SYMBOLS)
    Ld = capi.lib(debug=True)
    for name in list(declared) + list(capi.SYMBOLS):
        assert hasattr(Ld, name), name
    dyn = subprocess.check_output(["nm", "-D", "--defined-only", LIB_PATH], text=True)
    assert "fl_debug_" not in dyn
    exported = set(re.findall(r"\bT (fl_[a-z0-9_]+)$", dyn, flags=re.M))
    assert exported == set(capi.SYMBOLS), exported ^ set(capi.SYMBOLS)
    csrc = os.path.join(REPO_ROOT, "fast-livo_amd", "csrc")           # no tuning knob comes from the environment (fl_set_option)
    for f in os.listdir(csrc):
        assert "getenv" not in open(os.path.join(csrc, f)).read(), f


def test_struct_sizes_match_header():
    import ctypes as C
    assert C.sizeof(capi.State18) == 8 * (24 + 324)
    assert C.sizeof(capi.State23) == 8 * 26
    assert C.sizeof(capi.IterInfo) == 8 * 24 + 8 * 4
    assert C.sizeof(capi.Config) == 6 * 4 + 8 * (9 + 3 + 9 + 3 + 4 + 5 + 2)


def test_struct_layouts_match_a_c_compiler(tmp_path):
    """sizeof + offset of the last member of every struct of the header, as gcc lays them out, against the ctypes mirrors."""
    import ctypes as C
    import subprocess
    pairs = {"fl_config": capi.Config, "fl_state18": capi.State18, "fl_state23": capi.State23, "fl_iter_info": capi.IterInfo,
             "fl_map_info": capi.MapInfo, "fl_imu_sample": capi.ImuSample, "fl_pose6d": capi.Pose6d, "fl_imu_proc": capi.ImuProc,
             "fl_patch_candidate": capi.PatchCandidate, "fl_vmap_obs": capi.VmapObs, "fl_diagnostics": capi.Diagnostics, "fl_frame_timing": capi.FrameTiming}
    hdr = open(os.path.join(REPO_ROOT, "include", "fastlivo_hip.h")).read()
    declared = set(re.findall(r"}\s*(fl_[a-z0-9_]+);", hdr))
    assert declared == set(pairs), declared ^ set(pairs)
    src = tmp_path / "sz.c"
    body = "".join(f'printf("{n} %zu %zu\\n", sizeof({n}), offsetof({n}, {cls._fields_[-1][0]}));\n' for n, cls in pairs.items())
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "fastlivo_hip.h"\nint main(void){\n' + body + "return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(REPO_ROOT, "include"), "-o", str(exe), str(src)])
    out = subprocess.check_output([str(exe)], text=True)
    for line in out.strip().splitlines():
        name, size, off = line.split()
        cls = pairs[name]
        assert C.sizeof(cls) == int(size), name
        assert getattr(cls, cls._fields_[-1][0]).offset == int(off), name


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU present")
def test_create_fails_loudly_without_gpu():
    from fast_livo_amd import synth
    cfg = capi.make_config(synth.AVIA_R_LI, synth.AVIA_T_LI, synth.AVIA_RCL, synth.AVIA_PCL, synth.PINHOLE)
    with pytest.raises(capi.FlError):
        capi.Handle(cfg)
