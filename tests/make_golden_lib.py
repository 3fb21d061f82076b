"""Deterministic small cases whose oracle outputs are frozen under tests/golden/*.json."""
import numpy as np


def run_lio18_iter(orc, scene):
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(1500, scene=scene)
    x = orc.state18_from_frame(fr)
    nbr, valid = synth.knn5(scene, fr.world_at(fr.R_prior, fr.p_prior))
    sel = valid.copy()
    r = orc.lio18_iterate(x, x.copy(), fr.body_xyz, nbr, sel, fr.R_LI, fr.t_LI, fr.laser_point_cov, nthreads=2)
    o = r["out"]
    return {"solution": list(o.solution), "HTH": list(o.HTH), "HTz": list(o.HTz), "neff": [o.effct_feat_num],
            "total_residual": [o.total_residual], "state_after": x.vec().tolist(),
            "sel_checksum": [int(np.flatnonzero(sel).sum())]}


def run_vio_level(orc, scene):
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(256, scene=scene)
    vf = synth.make_vio_frame(40, fr)
    x = orc.state18_from_frame(fr)
    r = orc.vio_update_state(vf, x, x.copy(), 1e10, 1)
    o = r["out"]
    return {"solution": list(o.solution), "HTH": list(o.HTH), "HTz": list(o.HTz), "error": [r["error"]],
            "iterations": [o.iterations], "errors_head": r["errors"][:8].tolist(), "state_after": x.vec().tolist()}


def run_ikfom_update(orc, scene):
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(1200, scene=scene)
    x = orc.state23_from_frame(fr, synth.quat_from_R)
    P = fr.cov23.copy()
    r = orc.ikfom_update(x, P, fr.body_xyz, 0.001, 4, lambda w: synth.knn5(scene, w), nthreads=2)
    return {"dx": list(r["out"].dx), "iterations": [r["out"].iterations], "neff": [r["out"].effct_feat_num],
            "state_after": x.vec().tolist(), "P_diag": np.diag(P).tolist(), "P_row0": P[0].tolist()}
