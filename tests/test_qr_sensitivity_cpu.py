"""How much can the one unpinnable piece of third-party arithmetic change?  esti_plane (common_lib.h:448-493) solves a 5x3 float least
squares with Eigen's ColPivHouseholderQR; the restatement (and the device) associate its short reductions left to right, an x86 Eigen
build may reduce one SSE packet of four horizontally first (orc_lio_common.h: orc_sum_terms, modes 1 and 2).  Eigen is not available in
this image, so the order cannot be pinned; this test MEASURES what it can flip on unfiltered data (no margin filter: raw synthetic scan,
raw 5-NN): planes that differ in any bit, planarity verdicts, first-pass selections and effective flags.  It asserts only that the
effect stays a rare last-bit event (otherwise the "bit-identical to the oracle" claims would say little about the reference) and prints
the counts -- the size of the risk is now known instead of unknown."""
import ctypes as C

import numpy as np

from helpers import p


def test_reduction_order_of_the_plane_fit_qr(oracle_lib, scene, capsys):
    orc = oracle_lib
    from fast_livo_amd import synth
    n = 200000
    fr = synth.make_lio_frame(n, scene=scene, scan_noise=0.01)
    world = fr.world_at(fr.R_prior, fr.p_prior).astype(np.float32)
    nbr, valid = synth.knn5(scene, world)
    L = orc.lib()
    L.orc_plane_sensitivity.restype = C.c_int
    res = {}
    for mode in (0, 1, 2):
        plane = np.zeros((n, 4), np.float32)
        planar = np.zeros(n, np.uint8); sel = np.zeros(n, np.uint8); eff = np.zeros(n, np.uint8)
        L.orc_plane_sensitivity(p(nbr, C.c_float), p(world, C.c_float), p(fr.body_xyz, C.c_float), n, mode, p(plane, C.c_float),
                                p(planar, C.c_uint8), p(sel, C.c_uint8), p(eff, C.c_uint8))
        res[mode] = (plane, planar & valid, sel & valid, eff & valid)
    base = res[0]
    lines = []
    for mode in (1, 2):
        plane, planar, sel, eff = res[mode]
        dif = (plane.view(np.uint32) != base[0].view(np.uint32)).any(axis=1) & (valid != 0)
        d = np.abs(plane[dif] - base[0][dif]).max(axis=1) if dif.any() else np.zeros(1)
        rel = float(np.median(d))
        flips = dict(planes_with_a_different_bit=int(dif.sum()), median_abs_plane_difference=rel, p99_abs_plane_difference=float(np.percentile(d, 99)),
                     max_abs_plane_difference=float(d.max()),
                     planarity_flips=int((planar != base[1]).sum()), selection_flips=int((sel != base[2]).sum()),
                     effective_flips=int((eff != base[3]).sum()))
        lines.append((mode, flips))
        # typically a last-bit event: a different association moves a plane coefficient by a few float ulps (ill-conditioned
        # neighbour sets -- nearly collinear points -- amplify it; those planes mostly fail the 0.1 planarity check anyway) ...
        assert rel <= 1e-4           # (the offset d of a plane 20-30 m from the origin has a float ulp of 2e-6)
        # ... and that crosses a gate for at most a handful of the 200 000 points
        assert flips["planarity_flips"] + flips["selection_flips"] <= n // 2000
    with capsys.disabled():
        print(f"\n[QR reduction-order sensitivity] {n} unfiltered points, {int(valid.sum())} with 5 neighbours within range; "
              f"baseline selected {int(base[2].sum())}")
        for mode, fl in lines:
            print(f"  mode {mode} ({'((t0+t2)+(t1+t3))+t4' if mode == 1 else '((t0+t1)+(t2+t3))+t4'}): {fl}")
