"""Sharded form with the exchange inside the pass kernels (api_p2p.inc / handoff.h peer_allreduce32): two ranks, each with half
of the frame's points / patches, exchange their 32 sums peer to peer in the solver workgroup. On the 1-GPU test box both ranks
live on device 0 -- in one process (fl_p2p_connect_local, one host thread per rank) and in two processes (hipIpc handles):
the kernels, the protocol and the IPC plumbing are the ones a multi-GPU node uses, only the wire is not xGMI.
Required: both ranks end bitwise equal; equal to the unsharded run within the re-association tolerance (1e-9).
World sizes 2, 3, 4, 5 and 8: with four or more senders peer_allreduce32 polls them in more than one round (one wavefront per
sender at a time) and the store fan-out reaches 7 peers -- the paths BASELINE configs 4 / 5 (8 GPUs) take. Ranks of one process
on one device need one hardware queue each (tests/conftest.py raises GPU_MAX_HW_QUEUES): two ranks multiplexed onto one queue
would wait for each other's kernels."""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_ranks(fns):
    out, err = [None] * len(fns), [None] * len(fns)

    def wrap(i):
        try:
            out[i] = fns[i]()
        except Exception as e:   # noqa: BLE001
            err[i] = e
    th = [threading.Thread(target=wrap, args=(i,)) for i in range(len(fns))]
    [t.start() for t in th]
    [t.join(120) for t in th]
    assert not any(t.is_alive() for t in th), "a rank hangs"
    for e in err:
        if e:
            raise e
    return out


@pytest.mark.parametrize("world", [2, 3, 4, 5, 8])
def test_lio_passes_sharded_in_kernel(gpu_lib, oracle_lib, scene, world):
    capi = gpu_lib
    from fast_livo_amd import synth
    n, max_iter = 30000, 6
    fr = synth.make_lio_frame(n, scene=scene)
    nbr, valid = synth.knn5(scene, fr.world_at(fr.R_prior, fr.p_prior))
    cfg = capi.config_from_frames(fr, max_iterations=max_iter)
    ref = capi.Handle(cfg)
    x0 = capi.state18_from_frame(fr)
    ref.lio_set_points(fr.body_xyz); ref.lio_begin18(x0, x0); ref.lio_set_neighbours(nbr, valid)
    iref = ref.lio_iterate18(max_iter + 1, 0)
    xref = ref.lio_get_state18()
    ref.close()
    hs = [capi.Handle(cfg) for _ in range(world)]
    capi.p2p_connect_local(hs)
    cuts = np.linspace(0, n, world + 1).astype(int)

    def rank(r):
        def go():
            h = hs[r]
            sl = slice(cuts[r], cuts[r + 1])
            h.lio_set_points(fr.body_xyz[sl]); h.lio_begin18(x0, x0); h.lio_set_neighbours(nbr[sl], valid[sl])
            info = h.lio_iterate18(max_iter + 1, 0)
            return info, h.lio_get_state18()
        return go
    res = _run_ranks([rank(r) for r in range(world)])
    for info, x in res:
        assert info.status == 0 and info.iterations == iref.iterations and info.effct_feat_num == iref.effct_feat_num
        assert np.array_equal(x.vec(), res[0][1].vec())                 # ranks bitwise equal
        assert np.abs(x.vec() - xref.vec()).max() <= 1e-9
    # forced passes, one launch per pass (the non-multi-pass kernels exchange as well)
    for h in hs:
        h.set_option(capi.FL_OPT_MULTIPASS, 0)
    try:
        def rank1(r):
            def go():
                h = hs[r]
                h.lio_begin18(x0, x0)
                h.lio_set_neighbours(nbr[cuts[r]:cuts[r + 1]], valid[cuts[r]:cuts[r + 1]])
                h.lio_iterate18(3, capi.FL_ITER_FORCE)
                return h.lio_get_state18()
            return go
        res1 = _run_ranks([rank1(r) for r in range(world)])
    finally:
        for h in hs:
            h.set_option(capi.FL_OPT_MULTIPASS, 1)
    assert all(np.array_equal(x.vec(), res1[0].vec()) for x in res1)
    for h in hs:
        h.close()


def test_vio_levels_and_all_device_frame_sharded_in_kernel(gpu_lib, oracle_lib, scene):
    capi = gpu_lib
    from fast_livo_amd import synth
    n, m, max_iter = 20000, 900, 5
    fr = synth.make_lio_frame(n, scene=scene)
    vf = synth.make_vio_frame(m, fr)
    cfg = capi.config_from_frames(fr, vf, max_iterations=max_iter)
    x0 = capi.state18_from_frame(fr)
    ref = capi.Handle(cfg)
    ref.map_set_points(scene.map_xyz, 0.5)
    xr = capi.state18_from_frame(fr)
    ir = ref.lio_frame18_dev(xr, fr.body_xyz)
    ref.vio_set_frame(vf.img); ref.vio_set_patches(vf.ref_patch, vf.pos, vf.search_level)
    xv = capi.state18_from_frame(fr)
    ref.vio_compute_j(xv, x0)
    ref.close()
    hs = [capi.Handle(cfg) for _ in range(2)]
    capi.p2p_connect_local(hs)

    def rank(r):
        def go():
            h = hs[r]
            h.map_set_points(scene.map_xyz, 0.5)                         # map replicated, scan sharded
            x = capi.state18_from_frame(fr)
            info = h.lio_frame18_dev(x, fr.body_xyz[r * n // 2:(r + 1) * n // 2])
            h.vio_set_frame(vf.img)
            sl = slice(r * m // 2, (r + 1) * m // 2)
            h.vio_set_patches(vf.ref_patch[sl], vf.pos[sl], vf.search_level[sl])
            xvr = capi.state18_from_frame(fr)
            h.vio_compute_j(xvr, x0)
            return info, x, xvr
        return go
    res = _run_ranks([rank(0), rank(1)])
    (i0, xa, va), (i1, xb, vb) = res
    assert i0.status == 0 and i1.status == 0
    assert i0.iterations == i1.iterations == ir.iterations and i0.effct_feat_num == i1.effct_feat_num == ir.effct_feat_num
    assert np.array_equal(xa.vec(), xb.vec()) and np.array_equal(xa.cov_np(), xb.cov_np())
    assert np.abs(xa.vec() - xr.vec()).max() <= 1e-9 and np.abs(xa.cov_np() - xr.cov_np()).max() <= 1e-11
    assert np.array_equal(va.vec(), vb.vec())
    assert np.abs(va.vec() - xv.vec()).max() <= 1e-9          # (the accept test runs through the ranks in the reference's float arithmetic)
    for h in hs:
        h.close()


@pytest.mark.parametrize("world", [2, 3, 4, 5, 8])
def test_exact_accept_replay_runs_through_the_ranks(gpu_lib, oracle_lib, world):
    """The reference decides `error <= last_error` on a float running sum over ALL patches in order (lidar_selection.cpp:849-859).
    Sharded, that sum runs through the ranks' contiguous patch ranges one after the other: on the fragile passes rank r continues the
    chain from the float rank r-1 ended with and the last rank sends the total back (solve18.h: vio_exact_chain). The accept/revert
    sequences, iteration counts and the final state then equal the oracle's on every frame -- at N > 1 as on one GPU -- and status
    bit 16 means what it means there ("decided in the reference's float arithmetic"), not "may differ"."""
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    replayed = 0
    for seed in range(1, 9):
        lio = synth.make_lio_frame(500, seed=synth.SEED + seed % 7)
        vf = synth.make_vio_frame(1000, lio, max_iterations=10, patch_seed=seed * 7919)
        cfg = capi.config_from_frames(lio, vf, max_iterations=10)
        xo = orc.state18_from_frame(lio)
        ro = orc.vio_compute_j(vf, xo, xo.copy())
        hs = [capi.Handle(cfg) for _ in range(world)]
        capi.p2p_connect_local(hs)
        cuts = np.linspace(0, vf.m, world + 1).astype(int)

        def rank(r):
            def go():
                h = hs[r]
                sl = slice(cuts[r], cuts[r + 1])
                h.vio_set_frame(vf.img); h.vio_set_patches(vf.ref_patch[sl], vf.pos[sl], vf.search_level[sl])
                xg = capi.state18_from_frame(lio); xp = capi.state18_from_frame(lio)
                infos = h.vio_compute_j(xg, xp)
                return infos, xg, h.vio_get_errors(cuts[r + 1] - cuts[r])
            return go
        res = _run_ranks([rank(r) for r in range(world)])
        for r, (infos, xg, eg) in enumerate(res):
            for l in range(3):
                assert infos[l].iterations == ro["outs"][l].iterations and infos[l].accepted == ro["outs"][l].accepted, (seed, r, l)
                assert not (infos[l].status & 8), (seed, r, l)
            assert np.array_equal(xg.vec(), res[0][1].vec()) and np.array_equal(xg.cov_np(), res[0][1].cov_np())      # ranks bitwise equal
            assert np.abs(xg.vec() - xo.vec()).max() <= 1e-9, (seed, r)
            assert np.abs(xg.cov_np() - xo.cov_np()).max() <= 1e-11, (seed, r)
            assert np.array_equal(eg.view(np.uint32), ro["errors"][cuts[r]:cuts[r + 1]].view(np.uint32)), (seed, r)
        replayed += any(i.status & 16 for i in res[0][0])
        for h in hs:
            h.close()
    assert replayed >= 1          # the chain through the ranks was exercised


@pytest.mark.parametrize("world", [2, 3, 4, 5, 8])
def test_mode23_passes_sharded_in_kernel(gpu_lib, scene, world):
    """The 23-state IKFoM update with the points spread over ranks and the 96-double record exchanged inside the pass kernels
    (three 32-double exchanges per pass, handoff.h peer_allreduce96): ranks bitwise equal, state and covariance equal to the
    unsharded update within the re-association tolerance -- multi-pass launches and one launch per pass."""
    capi = gpu_lib
    from fast_livo_amd import synth
    n, max_iter = 18000, 6
    fr = synth.make_lio_frame(n, scene=scene)
    nbr, valid = synth.knn5(scene, fr.world_at(fr.R_prior, fr.p_prior))
    cfg = capi.config_from_frames(fr, max_iterations=max_iter)
    x0 = capi.state23_from_frame(fr)
    P0 = fr.cov23.copy()

    def run_ref(count, flags):
        ref = capi.Handle(cfg)
        ref.lio_set_points(fr.body_xyz); ref.ikfom_begin(x0, P0); ref.lio_set_neighbours(nbr, valid)
        info = ref.ikfom_iterate(count, flags)
        x, P = ref.ikfom_get()
        ref.close()
        return info, np.frombuffer(bytes(x), dtype=np.float64).copy(), P
    iref, xref, Pref = run_ref(3, capi.FL_ITER_FORCE)
    hs = [capi.Handle(cfg) for _ in range(world)]
    capi.p2p_connect_local(hs)
    cuts = np.linspace(0, n, world + 1).astype(int)

    def rank(r):
        def go():
            h = hs[r]
            sl = slice(cuts[r], cuts[r + 1])
            h.lio_set_points(fr.body_xyz[sl]); h.ikfom_begin(x0, P0); h.lio_set_neighbours(nbr[sl], valid[sl])
            info = h.ikfom_iterate(3, capi.FL_ITER_FORCE)
            x, P = h.ikfom_get()
            return info, np.frombuffer(bytes(x), dtype=np.float64).copy(), P
        return go
    for no_multi in (False, True):
        for h in hs:
            h.set_option(capi.FL_OPT_MULTIPASS, 0 if no_multi else 1)
        res = _run_ranks([rank(r) for r in range(world)])
        for info, x, P in res:
            assert (info.status & 8) == 0 and info.effct_feat_num == iref.effct_feat_num
            assert np.array_equal(x, res[0][1]) and np.array_equal(P, res[0][2])      # ranks bitwise equal
            assert np.abs(x - xref).max() <= 1e-9
            assert np.abs(P - Pref).max() <= 1e-11
    for h in hs:
        h.close()


def test_two_processes_over_hip_ipc(gpu_lib, tmp_path):
    """Each rank its own process (as under torch.distributed.run), handles exchanged through files, both on device 0."""
    worker = os.path.join(ROOT, "tests", "p2p_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=300)
        assert p.returncode == 0, e[-2000:]
        outs.append(o.strip().splitlines()[-1])
    a, b = [np.array(o.split()[1:], dtype=np.float64) for o in outs]
    assert outs[0].split()[0] == outs[1].split()[0] == "OK"
    assert np.array_equal(a, b)


def test_bench_n2_path_on_one_device(gpu_lib):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank), with both ranks on device 0
    and the control plane on gloo (FL_BENCH_SINGLE_DEVICE / FL_BENCH_BACKEND, test aids): handle exchange, connection, self-test,
    timed region and the collective roofline launches of the in-kernel exchange path."""
    import json
    env = dict(os.environ, FL_BENCH_BACKEND="gloo", FL_BENCH_SINGLE_DEVICE="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "100", "--warmup", "20"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    # default N > 1 = STRONG scaling of the metric's own frame: 50 k points + 2 k patches split over the ranks, value = frame iterations/s
    assert d["n_gpus"] == 2 and d["state_finite"] and d["scaling"] == "strong"
    assert d["config"]["frame_points"] == 50000 and d["config"]["points_per_gpu"] == 25000 and d["config"]["patches_per_gpu"] == 1000
    assert d["value"] == d["frame_iterations_per_s"] and "shard_iterations_per_s" not in d
    assert "in-kernel peer-to-peer" in d["config"]["parallelism"] and "xGMI between" not in d["config"]["parallelism"]   # both ranks on ONE device here
    assert d["exchange"]["used"] == "in-kernel p2p" and d["exchange"]["p2p_selftest"] == "passed" and d["exchange"]["ranks_on_distinct_devices"] is False
    assert d["roofline"]["lio_pass_us"] < 100 and d["roofline"]["vio_pass_us"] < 100      # no time-outs hidden in the launches
    # weak scaling and BASELINE config 4 (200 k points over the ranks) are flags away
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29543", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "50", "--warmup", "10",
                          "--scaling", "weak"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["scaling"] == "weak" and d["config"]["frame_points"] == 100000 and d["config"]["points_per_gpu"] == 50000
    assert abs(d["shard_iterations_per_s"] - 2 * d["value"]) < 1e-6 * d["value"]


@pytest.mark.parametrize("points,config", [(50000, "3"), (200000, "4")])
def test_bench_n8_path_on_one_device(gpu_lib, points, config):
    """BASELINE configs 3 and 4 are defined at 8 GPUs: bench.py --gpus 8 exactly as the driver launches it (torch.distributed.run, one
    process per rank, hipIpc-mapped exchange buffers, the in-kernel exchange with 7 peers: two polling rounds and the full store
    fan-out of peer_allreduce32), all eight ranks on device 0 and the control plane on gloo -- so that the first real 8-GPU lease
    measures instead of debugging. No time-out bit, all ranks bitwise equal (the bench's self-test), finite states."""
    import json
    env = dict(os.environ, FL_BENCH_BACKEND="gloo", FL_BENCH_SINGLE_DEVICE="1")
    env.pop("GPU_MAX_HW_QUEUES", None)          # (conftest raises it for the in-process ranks; 8 PROCESSES need few queues each: bench.py)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                          "--master-port", "29571" if config == "3" else "29573", os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "100",
                          "--warmup", "20", "--points", str(points)], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 8 and d["state_finite"] and d["scaling"] == "strong"
    assert d["config"]["frame_points"] == points and d["config"]["points_per_gpu"] == points // 8 and d["config"]["patches_per_gpu"] == 250
    assert f"BASELINE config {config}" in d["config"]["workload"]
    assert d["exchange"]["used"] == "in-kernel p2p" and d["exchange"]["p2p_selftest"] == "passed" and d["exchange"]["ranks_on_distinct_devices"] is False
    assert 1.0 < d["roofline"]["lio_pass_us"] < 200 and 1.0 < d["roofline"]["vio_pass_us"] < 200      # neither skipped (abandoned) nor timed out
    print(f"\n[n8 on one device] config {config}: {d['value']:.0f} it/s, LIO pass {d['roofline']['lio_pass_us']:.1f} us, VIO pass {d['roofline']['vio_pass_us']:.1f} us")
