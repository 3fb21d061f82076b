"""Every exchange form of the sharded VIO pass decides `error <= last_error` like the reference (lidar_selection.cpp:849-861: a FLOAT
running sum of the per-patch errors over ALL patches in order):
  * in-kernel peer exchange          tests/test_p2p_gpu.py::test_exact_accept_replay_runs_through_the_ranks (chain handed rank to rank)
  * collective form (this file)      accumulate -> all-reduce(32 doubles) + all-gather(per-patch floats) -> fl_vio_solve_exact: every rank
                                     replays the chain over the gathered floats (solve18.h vio_exact_flat_sum) -- what
                                     fl_vio_iterate_sharded does with RCCL and fast-livo_amd/sharded.py with torch.distributed
The collectives are emulated in-process here (world handles on the one device of the test box, sums / concatenation through the
host in rank order): the kernels and the arithmetic are the ones the RCCL and torch forms run, only the transport is not.
Also: the bench.py fallback branch (FL_BENCH_NO_P2P=1: native RCCL must fail cleanly with two ranks on one device, every rank
must agree on torch.distributed) -- control flow no test covered before round 4."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _collective_level(capi, torch, hs, bufs, chunks, gathered, stride, level, max_iter, use_exact=True):
    """One pyramid level in the collective form on `world` handles; returns the per-rank infos of the last pass."""
    world = len(hs)
    infos = None
    for it in range(max_iter):
        for r, h in enumerate(hs):
            h.vio_accumulate(level, bufs[r].data_ptr())
            if use_exact:
                h.vio_errors_chunk(chunks[r].data_ptr(), stride)
            h.sync()
        total = torch.zeros_like(bufs[0])
        for r in range(world):                      # all-reduce: a fixed order, the same total on every rank
            total += bufs[r]
        for r in range(world):
            bufs[r].copy_(total)
        if use_exact:
            gathered.copy_(torch.cat(chunks))       # all-gather in rank order
        torch.cuda.synchronize()
        infos = [h.vio_solve_exact(bufs[r].data_ptr(), 0, gathered.data_ptr(), stride, world, want_info=True) if use_exact
                 else h.vio_solve(bufs[r].data_ptr(), 0, want_info=True) for r, h in enumerate(hs)]
        if all(i.stop for i in infos):
            break
    return infos


@pytest.mark.parametrize("world", [1, 2, 3])
def test_collective_form_takes_the_references_decisions(gpu_lib, oracle_lib, world):
    capi, orc = gpu_lib, oracle_lib
    import torch
    from fast_livo_amd import synth
    fragile_seen = 0
    for seed in range(1, 9):
        lio = synth.make_lio_frame(500, seed=synth.SEED + seed % 7)
        vf = synth.make_vio_frame(1000, lio, max_iterations=10, patch_seed=seed * 7919)
        cfg = capi.config_from_frames(lio, vf, max_iterations=10)
        cuts = np.linspace(0, vf.m, world + 1).astype(int)
        counts = [int(cuts[r + 1] - cuts[r]) for r in range(world)]
        stride = max(counts) + 1
        hs = [capi.Handle(cfg) for _ in range(world)]
        for r, h in enumerate(hs):
            sl = slice(cuts[r], cuts[r + 1])
            h.vio_set_frame(vf.img); h.vio_set_patches(vf.ref_patch[sl], vf.pos[sl], vf.search_level[sl])
        bufs = [torch.zeros(capi.FL_SUMS18, dtype=torch.float64, device="cuda") for _ in range(world)]
        chunks = [torch.zeros(stride, dtype=torch.float32, device="cuda") for _ in range(world)]
        gathered = torch.zeros(world * stride, dtype=torch.float32, device="cuda")
        # the oracle, level by level (ComputeJ, lidar_selection.cpp:967-977: every level starts from last_error = 1e10)
        xo = orc.state18_from_frame(lio)
        xprop_o = xo.copy()
        xg = [capi.state18_from_frame(lio) for _ in range(world)]
        xprop = capi.state18_from_frame(lio)
        for level in (2, 1, 0):
            ro = orc.vio_update_state(vf, xo, xprop_o, 1e10, level)
            for r, h in enumerate(hs):
                h.vio_begin(xg[r], xprop)
            infos = _collective_level(capi, torch, hs, bufs, chunks, gathered, stride, level, 10)
            for r, h in enumerate(hs):
                assert infos[r].iterations == ro["out"].iterations and infos[r].accepted == ro["out"].accepted, (seed, level, r)
                xg[r] = h.vio_get_state18()
                assert np.array_equal(xg[r].vec(), xg[0].vec())                      # ranks bitwise equal
                assert np.abs(xg[r].vec() - xo.vec()).max() <= 1e-9, (seed, level, r)
                eg = h.vio_get_errors(counts[r])
                assert np.array_equal(eg.view(np.uint32), ro["errors"][cuts[r]:cuts[r + 1]].view(np.uint32)), (seed, level, r)
            fragile_seen += bool(infos[0].status & 16)
        for h in hs:
            h.close()
    assert fragile_seen >= 1          # near-ties occurred: the float chain, not the fp64 mean, took those decisions


def test_native_rccl_form_with_decisions_one_rank(gpu_lib, oracle_lib):
    """fl_vio_iterate_sharded without FL_ITER_FORCE: chunk-size agreement (max all-reduce), ncclAllGather of the per-patch floats and
    the exact solve, on a 1-rank communicator (RCCL refuses two ranks on one device; world > 1 shares every kernel with the emulated
    collectives above)."""
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    for seed in (2, 5):
        lio = synth.make_lio_frame(500, seed=synth.SEED + seed % 7)
        vf = synth.make_vio_frame(1000, lio, max_iterations=10, patch_seed=seed * 7919)
        h = capi.Handle(capi.config_from_frames(lio, vf, max_iterations=10))
        h.vio_set_frame(vf.img); h.vio_set_patches(vf.ref_patch, vf.pos, vf.search_level)
        h.comm_init(h.comm_unique_id(), 0, 1)
        xo = orc.state18_from_frame(lio)
        xprop_o = xo.copy()
        xg = capi.state18_from_frame(lio)
        xprop = capi.state18_from_frame(lio)
        for level in (2, 1, 0):
            ro = orc.vio_update_state(vf, xo, xprop_o, 1e10, level)
            h.vio_begin(xg, xprop)
            info = h.vio_iterate_sharded(level, 10, 0)
            assert info.iterations == ro["out"].iterations and info.accepted == ro["out"].accepted, (seed, level)
            xg = h.vio_get_state18()
            assert np.abs(xg.vec() - xo.vec()).max() <= 1e-9
        h.comm_destroy(); h.close()


def test_bench_n2_fallback_on_one_device(gpu_lib):
    """bench.py --gpus 2 with the in-kernel exchange switched off (FL_BENCH_NO_P2P=1), both ranks on device 0, control plane on gloo:
    the native RCCL communicator cannot be built (two ranks, one device) -- that must fail cleanly on every rank, all ranks must
    agree, and the run must complete on torch.distributed with finite states and `exchange.used` saying so."""
    env = dict(os.environ, FL_BENCH_BACKEND="gloo", FL_BENCH_SINGLE_DEVICE="1", FL_BENCH_NO_P2P="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29547", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "10"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["state_finite"]
    assert d["exchange"]["used"] == "torch.distributed" and d["exchange"]["p2p_selftest"] is None
    assert "torch.distributed.all_reduce (gloo)" in d["config"]["parallelism"]
    # and with the torch exchange forced outright (no attempt at RCCL)
    env["FL_BENCH_TORCH_EXCHANGE"] = "1"
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29549", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "40", "--warmup", "10"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["exchange"]["used"] == "torch.distributed" and d["state_finite"]
