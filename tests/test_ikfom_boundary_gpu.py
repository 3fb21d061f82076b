"""SURVEY 8b, IKFoM callback boundary on the device: the product's 2-ARGUMENT `void h_share_model(state_ikfom&,
esekfom::dyn_share_datastruct<double>&)` (fast-livo_amd/host/fastlivo_shim.hpp, file-scope context like the reference's globals)
is registered through `init_dyn_share`'s `measurementModel_dyn_share` function-pointer parameter (esekfom.hpp:129,238-254; call
site laserMapping.cpp:1233-1235) of a stand-in esekf (tests/host_emul/esekf_mock.hpp) whose `update_iterated_dyn_share_modified`
is the oracle's restatement of esekfom.hpp:1619-1928 -- an updater that knows nothing about surrogates.  The 23x12 surrogate the
callback returns from the device-reduced sums must drive it to the same state/covariance as fl_ikfom_update_iterated (the
product's own whole update) and as the oracle's update over the reference's N_eff x 12 rows: 1e-9."""
import os
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build():
    d = os.path.join(ROOT, "tests", "host_emul")
    exe = os.path.join(d, "ikfom_boundary.bin")
    srcs = [os.path.join(d, "ikfom_boundary.cpp"), os.path.join(d, "esekf_mock.hpp"),
            os.path.join(ROOT, "fast-livo_amd", "host", "fastlivo_shim.hpp")]
    if not os.path.exists(exe) or any(os.path.getmtime(exe) < os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-o", exe, srcs[0],
                               "-L" + os.path.join(ROOT, "fast-livo_amd"), "-lfastlivo_hip", "-L" + os.path.join(ROOT, "oracle"), "-loracle",
                               "-Wl,-rpath," + os.path.join(ROOT, "fast-livo_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
                               "-Wl,-rpath,/opt/rocm/lib"])
    return exe


@pytest.mark.parametrize("n,max_iter", [(8000, 4), (50000, 10)])
def test_two_argument_callback_through_the_unmodified_update(gpu_lib, oracle_lib, scene, tmp_path, n, max_iter):
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    exe = _build()
    R = 0.001
    fr = synth.make_lio_frame(n, scene=scene)
    x0 = capi.state23_from_frame(fr)
    P0 = fr.cov23.copy()
    limit = np.full(23, 0.001)
    f = tmp_path / "frame23.bin"
    with open(f, "wb") as fh:
        fh.write(struct.pack("<iii", n, len(scene.map_xyz), max_iter))
        fh.write(struct.pack("<d", R))
        fh.write(x0.vec().astype("<f8").tobytes())
        fh.write(P0.astype("<f8").tobytes())
        fh.write(limit.astype("<f8").tobytes())
        fh.write(fr.body_xyz.astype("<f4").tobytes())
        fh.write(np.ascontiguousarray(scene.map_xyz, dtype="<f4").tobytes())
    out = subprocess.run([exe, str(f)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    kv = dict(zip(lines[0].split()[0::2], map(int, lines[0].split()[1::2])))
    assert kv["iters_callback"] == kv["iters_device"] and kv["searches_callback"] == kv["searches_device"]
    assert kv["status_a"] == 0 and kv["status_hshare"] == 0 and kv["status_b"] == 0 and kv["neff"] > n // 4
    d = dict(zip(lines[1].split()[0::2], map(float, lines[1].split()[1::2])))
    assert d["moved"] > 1e-4                                            # the update did something
    assert d["max_state_diff"] <= 1e-9
    assert d["max_P_diff"] <= 1e-9 * max(1.0, d["P_scale"])
    # and both equal the oracle's update over the reference's own N_eff x 12 rows
    xo = orc.state23_from_frame(fr, synth.quat_from_R)
    Po = fr.cov23.copy()

    def knn(w):
        nb, _, va, _ = orc.knn5_bruteforce(scene.map_xyz, w)
        return nb, va
    ro = orc.ikfom_update(xo, Po, fr.body_xyz, R, max_iter, knn)
    xb = np.array(lines[2].split(), dtype=np.float64)
    assert ro["out"].iterations == kv["iters_device"]
    assert np.abs(xb - xo.vec()).max() <= 1e-9
