#!/usr/bin/env python3
"""bench.py -- ESKF iterations/s of the FAST-LIVO hot path on MI355X (BASELINE.json metric).

One *step* = one LIVO ESKF iteration over one synthetic frame already resident in HBM:
  * one LIO pass  : 50 000 down-sampled LiDAR points -> point-to-plane residuals + gates + 1x6 rows
                    -> H^T H / H^T z reduction -> 18-state gain solve -> state (+)= delta
  * one VIO pass  : 2 000 8x8 photometric patches (pyramid level 0) -> same reduction/solve
(BASELINE config 3, "LIVO: 50k pts + 2k 8x8 patches"; kNN excluded on both sides, neighbours
pre-staged -- SURVEY.md section 8d).  With --gpus N (launched by torch.distributed.run, one rank
per GPU) every rank holds its own 50k-point / 2k-patch shard of an N-times larger frame (weak
scaling), reduces its partial normal equations on the device, all-reduces the 32-double record
over RCCL and runs the gain solve redundantly.  `value` = shard-iterations/s summed over ranks
(= N x frame-iterations/s); `frame_iterations_per_s` is reported next to it.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_POINTS = 50000
N_PATCHES = 2000
PASSES_PER_LAUNCH = 10   # avia.yaml max_iteration
AT_SCALE_POINTS = 8000000  # roofline.at_scale: the LIO pass where the traffic, not the hand-off latency, is the time
VIO_LEVEL = 0
# algorithmic HBM bytes per unit and launch
LIO_BYTES_PER_POINT = 12 + 16 + 1      # body xyz + cached plane (n,d) + selection flag
VIO_BYTES_PER_PATCH = 405 + 4          # 256 ref + 121 image footprint + 24 pos + 4 level, + 4 written
HBM_PEAK_GBS = 8000.0                  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--points", type=int, default=N_POINTS)
    ap.add_argument("--patches", type=int, default=N_PATCHES)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU-oracle sample")
    ap.add_argument("--sweep", action="store_true", help="also print a kernel-only size sweep to stderr")
    ap.add_argument("--no-at-scale", action="store_true", help="skip the 8 M-point pass of roofline.at_scale")
    return ap.parse_args()


def cpu_baseline(fr, vf, nbr, valid, budget_s):
    """The CPU oracle (a port of the reference loop bodies; the reference itself cannot be built
    here) timed on this box's host cores: LIO pass with the reference's OpenMP width (4 threads,
    CMakeLists.txt:23-26), VIO pass single-threaded as in the reference."""
    from oracle import oracle as orc
    threads = min(4, os.cpu_count() or 1)
    x0 = orc.state18_from_frame(fr)
    vf1 = vf
    old_max = vf1.max_iterations
    vf1.max_iterations = 1
    times = []
    t_end = time.perf_counter() + budget_s
    reps = 0
    while reps < 3 or (time.perf_counter() < t_end and reps < 200):
        x = x0.copy()
        sel = valid.copy()
        t0 = time.perf_counter()
        orc.lio18_iterate(x, x0, fr.body_xyz, nbr, sel, fr.R_LI, fr.t_LI, fr.laser_point_cov, nthreads=threads)
        xv = x0.copy()
        orc.vio_update_state(vf1, xv, x0, 1e10, VIO_LEVEL)
        times.append(time.perf_counter() - t0)
        reps += 1
    vf1.max_iterations = old_max
    t = np.array(times[1:]) if len(times) > 1 else np.array(times)
    med = float(np.median(t))
    return {"value": 1.0 / med, "unit": "iterations/s", "cores": threads, "kind": "port",
            "host_cores_available": os.cpu_count(),
            "sample": f"{len(t)} LIVO iterations ({fr.n} pts LIO pass with {threads} OpenMP threads + {vf.m} patches VIO pass "
                      f"single-thread, level {VIO_LEVEL}); median {med * 1e3:.2f} ms, p10 {np.percentile(t, 10) * 1e3:.2f}, "
                      f"p90 {np.percentile(t, 90) * 1e3:.2f}"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("FL_BENCH_SINGLE_DEVICE") == "1":     # test aid: all ranks on device 0 (1-GPU box), control plane over gloo
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the ESKF hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    distributed = world > 1 or os.environ.get("FL_BENCH_FORCE_SHARDED") == "1"   # the env var exercises the N>1 code path on one GPU
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29617")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("FL_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    ctl = "cuda" if (not distributed or os.environ.get("FL_BENCH_BACKEND", "nccl") == "nccl") else "cpu"   # where control-plane tensors live
    import fastlivo  # noqa: F401
    from fast_livo_amd import capi, synth

    # ---- synthetic frame: this rank's shard of an (world x points)-point scan
    scene = synth.make_scene()
    fr = synth.make_lio_frame(args.points, scene=scene, point_seed=synth.SEED + 101 + 17 * rank)
    vf = synth.make_vio_frame(args.patches, fr, patch_seed=synth.SEED + 103 + 17 * rank)
    world_pts = fr.world_at(fr.R_prior, fr.p_prior)
    nbr, valid = synth.knn5(scene, world_pts)

    cfg = capi.config_from_frames(fr, vf, max_iterations=10, device=local_rank)
    hl = capi.Handle(cfg)   # LIO filter
    hv = capi.Handle(cfg)   # VIO filter
    side = torch.cuda.Stream()            # everything (kernels, RCCL ordering, events) on one non-default stream
    torch.cuda.set_stream(side)
    stream = torch.cuda.current_stream().cuda_stream
    hl.set_stream(stream)
    hv.set_stream(stream)
    x0 = capi.state18_from_frame(fr)
    F = capi.FL_ITER_FORCE

    def begin():
        hl.lio_set_points(fr.body_xyz)
        hl.lio_begin18(x0, x0)
        hl.lio_set_neighbours(nbr, valid)
        hv.vio_set_frame(vf.img)
        hv.vio_set_patches(vf.ref_patch, vf.pos, vf.search_level)
        hv.vio_begin(x0, x0)

    # N > 1, first choice: the exchange INSIDE the pass kernels (api_p2p.inc): every rank maps the peers' exchange buffers (hipIpc),
    # the solver workgroup of each pass trades the 32 sums peer to peer over xGMI, and a rank keeps enqueuing multi-pass launches
    # exactly like the single-GPU path. It is verified before it is relied on (no time-out bit, all ranks bitwise equal after a few
    # passes); every decision is agreed on by all ranks, otherwise they would wait for each other in different collectives.
    p2p = False
    if distributed and world >= 2 and os.environ.get("FL_BENCH_NO_P2P") != "1" and os.environ.get("FL_BENCH_TORCH_EXCHANGE") != "1":
        def all_agree(flag):
            t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=ctl)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return int(t.item()) == 1
        gathered = []
        exported = True
        for h_ in (hl, hv):
            mine = torch.zeros(65, dtype=torch.uint8, device=ctl)
            try:
                mine.copy_(torch.frombuffer(bytearray(h_.p2p_export(world)) + bytearray([1]), dtype=torch.uint8))
            except Exception as e:   # noqa: BLE001
                print(f"[bench] rank {rank}: p2p_export failed ({e})", file=sys.stderr)
            allh = [torch.zeros(65, dtype=torch.uint8, device=ctl) for _ in range(world)]
            dist.all_gather(allh, mine)
            torch.cuda.synchronize()
            raw = [t.cpu().numpy().tobytes() for t in allh]
            exported = exported and all(r[64] == 1 for r in raw)
            gathered.append([r[:64] for r in raw])
        connected = exported
        if exported:
            try:
                hl.p2p_connect(rank, world, gathered[0])
                hv.p2p_connect(rank, world, gathered[1])
            except Exception as e:   # noqa: BLE001
                print(f"[bench] rank {rank}: p2p_connect failed ({e})", file=sys.stderr)
                connected = False
        connected = all_agree(connected)         # (also the barrier: nobody publishes before everybody has mapped everybody)
        if connected:
            begin()
            for _ in range(3):           # the launches of the timed region: multi-pass kernels, both epoch parities several times
                i1 = hl.lio_iterate18(PASSES_PER_LAUNCH, F)
                i2 = hv.vio_iterate(VIO_LEVEL, PASSES_PER_LAUNCH, F)
            sv = np.concatenate([hl.lio_get_state18().vec(), hv.vio_get_state18().vec()])
            mine = torch.tensor(sv, dtype=torch.float64, device=ctl)
            alls = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(alls, mine)
            same = all(bool(torch.equal(a, alls[0])) for a in alls)
            good = (i1.status & 8) == 0 and (i2.status & 8) == 0 and bool(np.isfinite(sv).all()) and same
            p2p = all_agree(good)
            if not p2p and rank == 0:
                print(f"[bench] in-kernel peer exchange failed its self-test (status {i1.status}/{i2.status}, equal={same}); "
                      "using RCCL", file=sys.stderr)
        if not p2p:
            hl.p2p_disconnect(); hv.p2p_disconnect()
    begin()
    from fast_livo_amd.sharded import GpuLioBackend, GpuVioBackend, ShardedPass
    lio_pass = ShardedPass(GpuLioBackend(hl, F), dist)      # accumulate -> all_reduce(32 doubles) -> solve
    vio_pass = ShardedPass(GpuVioBackend(hv, VIO_LEVEL, F), dist)

    # N > 1: the exchange is done natively (ncclAllReduce on the handle's stream between accumulate and solve, api_comm.inc);
    # the unique ids travel over torch.distributed. If RCCL cannot be bound, fall back to torch.distributed.all_reduce.
    exchange = "none"
    if p2p:
        exchange = "in-kernel peer-to-peer exchange of the solver workgroups over xGMI (api_p2p.inc)"
    elif distributed:
        exchange = "torch.distributed.all_reduce"
        if os.environ.get("FL_BENCH_TORCH_EXCHANGE") != "1":
            # every rank must take the same branch: the unique id carries rank 0's verdict in its last byte, and the outcome of
            # comm_init is agreed on with a MIN all-reduce before anybody relies on the native communicator
            ok = 1
            for h_ in (hl, hv):
                uid = torch.zeros(129, dtype=torch.uint8, device=ctl)
                if rank == 0:
                    try:
                        raw = bytearray(h_.comm_unique_id()) + bytearray([1])
                        uid.copy_(torch.frombuffer(raw, dtype=torch.uint8))
                    except Exception as e:   # noqa: BLE001
                        print(f"[bench] native RCCL exchange unavailable ({e}); using torch.distributed", file=sys.stderr)
                dist.broadcast(uid, 0)
                torch.cuda.synchronize()
                host = uid.cpu().numpy().tobytes()
                if host[128] != 1:
                    ok = 0
                    break
                try:
                    h_.comm_init(bytes(host[:128]), rank, max(world, 1))
                except Exception as e:   # noqa: BLE001
                    print(f"[bench] rank {rank}: comm_init failed ({e})", file=sys.stderr)
                    ok = 0                       # keep going: the other ranks are in the next broadcast
            agree = torch.tensor([ok], dtype=torch.int32, device=ctl)
            dist.all_reduce(agree, op=dist.ReduceOp.MIN)
            if int(agree.item()) == 1:
                exchange = "ncclAllReduce on the pass stream (native)"
    native = exchange.startswith("ncclAllReduce")

    # A frame of the reference runs its passes back to back (<= max_iteration + 1 LIO passes, <= max_iteration VIO passes per
    # pyramid level): the single-GPU path enqueues PASSES_PER_LAUNCH consecutive passes as one multi-pass launch (pose handed
    # from the solver to the producers inside the kernel); every pass still does the full work of a step.
    def steps(k):
        """k steps = k LIO passes + k VIO passes"""
        if not distributed or p2p:
            done = 0
            while done < k:
                c = min(PASSES_PER_LAUNCH, k - done)
                hl.lio_iterate18(c, F, want_info=False)      # fused passes (reduce + solve + pose broadcast in-launch)
                hv.vio_iterate(VIO_LEVEL, c, F, want_info=False)
                done += c
        elif native:
            for _ in range(k):
                hl.lio_iterate18_sharded(1, F, want_info=False)  # accumulate -> ncclAllReduce(32 doubles) -> solve
                hv.vio_iterate_sharded(VIO_LEVEL, 1, F, want_info=False)
        else:
            for _ in range(k):
                lio_pass.step()
                vio_pass.step()

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
            torch.cuda.synchronize()

    steps(args.warmup)
    fence()
    t0 = time.perf_counter()
    steps(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device=ctl)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity: the filters must have produced finite states
    xs = hl.lio_get_state18().vec()
    xv = hv.vio_get_state18().vec()
    finite = bool(np.isfinite(xs).all() and np.isfinite(xv).all())

    # ---- roofline of the dominant kernel (LIO pass), HIP events on the launch stream
    roof = None
    if rank == 0 or p2p:       # with the in-kernel exchange a launch is a collective: every rank issues the same launches
        # the kernels of the timed region: one launch = PASSES_PER_LAUNCH passes (multi-pass kernels); average launch duration
        # from events on the launch stream around K back-to-back launches
        C = PASSES_PER_LAUNCH
        K = max(50, min(args.steps // C, 300))
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(10):
            hl.lio_iterate18(C, F, want_info=False)
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(K):
            hl.lio_iterate18(C, F, want_info=False)
        ev1.record()
        torch.cuda.synchronize()
        lio_launch_us = ev0.elapsed_time(ev1) * 1e3 / K
        ev0.record()
        for _ in range(K):
            hv.vio_iterate(VIO_LEVEL, C, F, want_info=False)
        ev1.record()
        torch.cuda.synchronize()
        vio_launch_us = ev0.elapsed_time(ev1) * 1e3 / K
        lio_us, vio_us = lio_launch_us / C, vio_launch_us / C
    if rank == 0:
        lio_bytes = LIO_BYTES_PER_POINT * args.points * C      # algorithmic bytes per launch = per pass x passes per launch
        vio_bytes = VIO_BYTES_PER_PATCH * args.patches * C
        dom = "lio18_multipass_kernel" if lio_us >= vio_us else "vio_multipass_kernel"
        dom_bytes, dom_us = (lio_bytes, lio_launch_us) if lio_us >= vio_us else (vio_bytes, vio_launch_us)
        ach = dom_bytes / (dom_us * 1e-6) / 1e9
        traffic, traffic_src = None, None
        try:   # PMC bytes are collected in separate rocprofv3 --pmc passes and committed under profiles/
            import glob
            latest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")))[-1]
            pm = json.load(open(latest))["kernels"]
            for kname, d in pm.items():
                if dom in kname:
                    traffic = d["read_bytes_x2_corrected"] + d["write_bytes_raw"]
                    traffic_src = os.path.basename(latest)
        except Exception:
            pass
        roof = {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_us": dom_us,
                "method": f"HIP events around {K} back-to-back launches on the launch stream (includes the "
                          "inter-kernel boundary)",
                "passes_per_launch": C, "lio_pass_us": lio_us, "vio_pass_us": vio_us,
                "note": "latency-bound at BASELINE sizes: a pass moves 1.45 MB (LIO) / 0.82 MB (VIO) = 0.2 us at 8 TB/s, against "
                        "two cross-workgroup hand-offs of ~2 us each per pass (SURVEY.md fact 5); see the DESIGN.md size sweep"}
        if world == 1 and not args.no_at_scale:
            # the same pass kernel where it is bandwidth-bound: at BASELINE sizes a pass is 0.2 us of traffic
            # behind ~6 us of hand-off latency, at 8 M points the traffic is the time
            us8, gb8 = lio_pass_at(capi, synth, scene, cfg, x0, AT_SCALE_POINTS)
            roof["at_scale"] = {"kernel": "lio18_pass_kernel", "points": AT_SCALE_POINTS, "pass_us": us8, "achieved": gb8, "unit": "GB/s",
                                "frac": gb8 / HBM_PEAK_GBS, "algorithmic_bytes": LIO_BYTES_PER_POINT * AT_SCALE_POINTS}
        if args.sweep:
            sweep(capi, synth, scene, cfg, x0, sys.stderr)

    cpu = None
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        cpu = cpu_baseline(fr, vf, nbr, valid, args.cpu_seconds)

    if distributed:
        torch.cuda.synchronize()
        dist.barrier()          # nobody unmaps an exchange buffer a peer may still be using
    hl.close()
    hv.close()
    if rank == 0:
        frame_it_s = args.steps / elapsed
        out = {
            "metric": "ESKF iterations/sec (50k LiDAR pts + 2k 8x8 patches)",
            "value": frame_it_s * world,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 point/pixel math + f64 Jacobian rows, reduction and solve",
            "data": "synthetic",
            "config": {"workload": f"LIVO (BASELINE config 3): {args.points} pts point-to-plane LIO pass + {args.patches} "
                                   f"8x8 patches VIO pass (level {VIO_LEVEL}) per GPU, 18-state (StatesGroup) ESKF, "
                                   "neighbours/planes/image resident in HBM",
                       "points_per_gpu": args.points, "patches_per_gpu": args.patches,
                       "iteration_definition": "one LIO pass + one VIO pass, each = residuals + Jacobian rows + "
                                               "normal equations + gain solve + state update",
                       "parallelism": (f"point/patch-range shards x{world}, the 32-double normal-equation record summed over the "
                                       f"ranks in every pass: {exchange}"
                                       + (f"; {PASSES_PER_LAUNCH} consecutive passes per launch" if p2p else "")) if world > 1 else
                                      f"single GPU, fused multi-pass kernels ({PASSES_PER_LAUNCH} consecutive passes per launch, as in one frame)"},
            "frame_iterations_per_s": frame_it_s,
            "state_finite": finite,
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        if cpu:
            out["speedup_vs_cpu_baseline"] = out["value"] / cpu["value"]
    if distributed:
        dist.destroy_process_group()       # (the handles' own communicators went with hl.close() / hv.close())
    if rank == 0:
        # RCCL writes its version banner to the C stdout when NCCL_DEBUG=VERSION is set in the environment: drain it first so
        # that the JSON line is the last line of this rank's stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


def lio_pass_at(capi, synth, scene, cfg, x0, n):
    """Average duration (us) and effective bandwidth (GB/s) of one forced LIO pass over n points, HIP events on the launch stream."""
    import torch
    fr = synth.make_lio_frame(min(n, 200000), scene=scene)
    reps = (n + fr.n - 1) // fr.n
    body = np.tile(fr.body_xyz, (reps, 1))[:n]
    w = fr.world_at(fr.R_prior, fr.p_prior)
    nbr, valid = synth.knn5(scene, w)
    nbr = np.tile(nbr, (reps, 1, 1))[:n]
    valid = np.tile(valid, reps)[:n]
    h = capi.Handle(cfg)
    h.set_stream(torch.cuda.current_stream().cuda_stream)
    h.lio_set_points(body)
    h.lio_begin18(x0, x0)
    h.lio_set_neighbours(nbr, valid)
    K = 100 if n <= 1000000 else 20
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
        h.lio_iterate18(1, capi.FL_ITER_FORCE, want_info=False)
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(K):
        h.lio_iterate18(1, capi.FL_ITER_FORCE, want_info=False)
    ev1.record()
    torch.cuda.synchronize()
    us = ev0.elapsed_time(ev1) * 1e3 / K
    gbs = LIO_BYTES_PER_POINT * n / (us * 1e-6) / 1e9
    h.close()
    return us, gbs


def sweep(capi, synth, scene, cfg, x0, fh):
    """Kernel-only effective bandwidth of the LIO pass over a size sweep."""
    for n in (50000, 200000, 1000000, 4000000, 8000000):
        us, gbs = lio_pass_at(capi, synth, scene, cfg, x0, n)
        print(json.dumps({"sweep_points": n, "lio_pass_us": us, "effective_GBps": gbs, "frac_of_8TBps": gbs / HBM_PEAK_GBS}),
              file=fh)


if __name__ == "__main__":
    main()
