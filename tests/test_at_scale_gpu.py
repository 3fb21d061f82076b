"""The at-scale sections of bench.py (8 M / 32 M points, 1 023 producer workgroups per pass) stand on these checks.

The producers hand their fp64 partial sums to the solver workgroup as self-tagged 8-byte words: the epoch tag takes the low 6 bits
of the mantissa, a partial travels with 46 of its 52 bits (csrc/handoff.h, rounded to nearest since round 4).  That is narrower
than the reference's fp64 accumulation (laserMapping.cpp:1665-1666 forms H^T H in doubles), so it is measured where it is largest:
one pass over 8 M points = 1 023 partials per sum, against sums formed in extended precision (numpy longdouble) from the oracle's own
per-point planes and residuals, and against the C oracle on all host cores.  Expectation: a rounded 46-bit partial is off by at
most 2^-47 = 7e-15 relative, unbiased, so ~1 000 like-signed partials give <= 7e-15 / sqrt(1023) ~ 2e-16 relative in the total --
below the fp64 summation noise of either side (~1e-14 at 8 M terms); the bound asserted is the 1e-12 every parity test states.
"""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _skew(v):
    z = np.zeros(len(v))
    return np.stack([np.stack([z, -v[:, 2], v[:, 1]], -1), np.stack([v[:, 2], z, -v[:, 0]], -1), np.stack([-v[:, 1], v[:, 0], z], -1)], 1)


def test_8M_point_sums_against_extended_precision(gpu_lib, oracle_lib, scene):
    capi, orc = gpu_lib, oracle_lib
    import torch
    from fast_livo_amd import synth
    n0, reps = 200000, 40
    fr = synth.make_lio_frame(n0, scene=scene)
    nbr0, valid0 = synth.knn5(scene, fr.world_at(fr.R_prior, fr.p_prior))
    # the oracle's pass over the base frame: planes, residuals, selection (and its own sequentially accumulated sums)
    xo = orc.state18_from_frame(fr)
    sel = valid0.copy()
    ro = orc.lio18_iterate(xo, xo.copy(), fr.body_xyz, nbr0, sel, fr.R_LI, fr.t_LI, fr.laser_point_cov, nthreads=32)
    eff = (sel != 0) & (ro["res_last"] <= 2.0)
    nv = ro["normvec"][eff].astype(np.float64)
    pb = fr.body_xyz[eff].astype(np.float64)
    p_i = pb @ fr.R_LI.T + fr.t_LI
    C = nv[:, :3] @ fr.R_prior                                   # R^T n
    A = np.einsum("nij,nj->ni", _skew(p_i), C)
    rows = np.concatenate([A, nv[:, :3]], 1).astype(np.longdouble)
    z = (-nv[:, 3]).astype(np.longdouble)
    HTH = (rows.T @ rows) * np.longdouble(reps)
    HTz = (rows.T @ z) * np.longdouble(reps)
    ref = np.zeros(32, dtype=np.longdouble)
    k = 0
    for i in range(6):
        for j in range(i, 6):
            ref[k] = HTH[i, j]; k += 1
    ref[21:27] = HTz
    ref[27] = np.longdouble(int(eff.sum()) * reps)
    ref[28] = np.abs(nv[:, 3]).astype(np.longdouble).sum() * reps
    ref[29] = (nv[:, 3].astype(np.longdouble) ** 2).sum() * reps
    # scale of entry (i, j): sqrt(S_ii S_jj) -- an off-diagonal sum may nearly cancel, its error is that of its terms
    d = np.sqrt(np.array([HTH[i, i] for i in range(6)], dtype=np.float64))
    zz = float(np.sqrt((z * z).sum() * reps))
    scale = np.ones(32)
    k = 0
    for i in range(6):
        for j in range(i, 6):
            scale[k] = d[i] * d[j]; k += 1
    scale[21:27] = d * zz
    scale[27], scale[28], scale[29] = float(ref[27]), float(ref[28]), float(ref[29])

    # the device: the same frame tiled to 8 M points, one accumulate launch (1 023 producer workgroups -> 1 023 partials per sum)
    n = n0 * reps
    body = np.tile(fr.body_xyz, (reps, 1))
    nbr = np.tile(nbr0, (reps, 1, 1))
    valid = np.tile(valid0, reps)
    h = capi.Handle(capi.config_from_frames(fr, max_iterations=4))
    x0 = capi.state18_from_frame(fr)
    h.lio_set_points(body); h.lio_begin18(x0, x0); h.lio_set_neighbours(nbr, valid)
    buf = torch.zeros(capi.FL_SUMS18, dtype=torch.float64, device="cuda")
    h.lio_accumulate18(buf.data_ptr(), 0); h.sync()
    torch.cuda.synchronize()
    gpu = buf.cpu().numpy()
    assert gpu[27] == float(ref[27]), "another set of effective points than the oracle's"
    err_gpu = np.abs((gpu[:30].astype(np.longdouble) - ref[:30]).astype(np.float64)) / scale[:30]

    # the C oracle over the same 8 M points on all cores (its sums are plain fp64 accumulations in thread-partial order)
    xo2 = orc.state18_from_frame(fr)
    sel2 = valid.copy()
    ro2 = orc.lio18_iterate(xo2, xo2.copy(), body, nbr, sel2, fr.R_LI, fr.t_LI, fr.laser_point_cov, nthreads=64)
    orc_s = np.concatenate([np.array(ro2["out"].HTH).reshape(6, 6)[np.triu_indices(6)], np.array(ro2["out"].HTz)])
    err_orc = np.abs((orc_s.astype(np.longdouble) - ref[:27]).astype(np.float64)) / scale[:27]
    err_gpu_orc = np.abs(gpu[:27] - orc_s) / scale[:27]
    h.close()
    report = {"points": n, "partials_per_sum": 1023, "gpu_vs_extended_precision_max_rel": float(err_gpu.max()),
              "oracle64threads_vs_extended_precision_max_rel": float(err_orc.max()), "gpu_vs_oracle_max_rel": float(err_gpu_orc.max()),
              "expectation_rounded_46bit_partials": 2.0 ** -47 / np.sqrt(1023.0), "bound_asserted": 1e-12}
    print("AT_SCALE_SUMS " + json.dumps(report))
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        json.dump(report, open(os.path.join(out_dir, "r04_at_scale_sums_check.json"), "w"), indent=1)
    assert err_gpu.max() <= 1e-12 and err_gpu_orc.max() <= 1e-12
