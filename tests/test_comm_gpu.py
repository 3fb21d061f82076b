"""Native exchange of the sharded form (fl_comm_* + fl_*_iterate*_sharded): a 1-rank RCCL communicator on the GPU box.
accumulate -> ncclAllReduce -> solve must equal the accumulate/solve pair driven from the host bit for bit; the N>1
orchestration itself is covered by the gloo world-2 test (tests/test_host_logic_cpu.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_sharded_native_equals_accumulate_solve(gpu_lib, scene):
    capi = gpu_lib
    import torch
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(20000, scene=scene)
    vf = synth.make_vio_frame(500, fr)
    nbr, valid = synth.knn5(scene, fr.world_at(fr.R_prior, fr.p_prior))
    cfg = capi.config_from_frames(fr, vf)
    F = capi.FL_ITER_FORCE

    def prep():
        hl, hv = capi.Handle(cfg), capi.Handle(cfg)
        x0 = capi.state18_from_frame(fr)
        hl.lio_set_points(fr.body_xyz); hl.lio_begin18(x0, x0); hl.lio_set_neighbours(nbr, valid)
        hv.vio_set_frame(vf.img); hv.vio_set_patches(vf.ref_patch, vf.pos, vf.search_level); hv.vio_begin(x0, x0)
        return hl, hv
    # reference: host-driven accumulate / solve (what sharded.py does around torch.distributed)
    hl, hv = prep()
    buf = torch.zeros(capi.FL_SUMS18, dtype=torch.float64, device="cuda")
    for _ in range(3):
        hl.lio_accumulate18(buf.data_ptr(), F); hl.sync(); hl.lio_solve18(buf.data_ptr(), F)
        hv.vio_accumulate(0, buf.data_ptr()); hv.sync(); hv.vio_solve(buf.data_ptr(), F)
    ref_l, ref_v = hl.lio_get_state18().vec(), hv.vio_get_state18().vec()
    # native: one communicator per handle, world size 1
    hl2, hv2 = prep()
    for h in (hl2, hv2):
        h.comm_init(h.comm_unique_id(), 0, 1)
    hl2.lio_iterate18_sharded(3, F, want_info=False)
    hv2.vio_iterate_sharded(0, 3, F, want_info=False)
    assert np.array_equal(hl2.lio_get_state18().vec(), ref_l)
    assert np.array_equal(hv2.vio_get_state18().vec(), ref_v)
    hl2.comm_destroy(); hv2.comm_destroy()
    # without a communicator the call fails loudly
    hl3, _ = prep()
    with pytest.raises(RuntimeError):
        hl3.lio_iterate18_sharded(1, F)
