"""Stress of the in-kernel peer exchange (api_p2p.inc): `world` ranks on one device, one host thread and ONE stream each, many
multi-pass launches back to back with no synchronisation in between; at the end no time-out bit and all ranks bitwise equal.
(One stream per rank on purpose: kernels that wait for each other must not sit behind one another in a hardware queue. On one
device HIP multiplexes streams onto a handful of hardware queues, so this emulation is limited to a few streams; on a multi-GPU
node every rank has its own device. Run it under `timeout`.)
usage: p2p_stress.py [world=2] [launches=1500] [points_per_rank=10000]"""
import json
import sys
import threading
import time

import numpy as np

sys.path.insert(0, "/root/repo")
import fastlivo  # noqa: F401,E402
from fast_livo_amd import capi, synth  # noqa: E402

world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
npr = int(sys.argv[3]) if len(sys.argv) > 3 else 10000
scene = synth.make_scene()
fr = synth.make_lio_frame(world * npr, scene=scene)
vf = synth.make_vio_frame(300 * world, fr)
nbr, valid = synth.knn5(scene, fr.world_at(fr.R_prior, fr.p_prior))
cfg = capi.config_from_frames(fr, vf, max_iterations=10)
hs = [capi.Handle(cfg) for _ in range(world)]
capi.p2p_connect_local(hs)
x0 = capi.state18_from_frame(fr)
F = capi.FL_ITER_FORCE
out = [None] * world


def rank(r):
    sl = slice(r * npr, (r + 1) * npr)
    h = hs[r]
    h.lio_set_points(fr.body_xyz[sl]); h.lio_begin18(x0, x0); h.lio_set_neighbours(nbr[sl], valid[sl])
    for k in range(launches):
        h.lio_iterate18(10, F, want_info=False)
        if k % 500 == 499:          # now and then a natural (non-forced) frame segment in between
            h.lio_begin18(x0, x0); h.lio_set_neighbours(nbr[sl], valid[sl])
            h.lio_iterate18(11, 0, want_info=False)
            h.lio_begin18(x0, x0); h.lio_set_neighbours(nbr[sl], valid[sl])
    i1 = h.lio_iterate18(1, F)
    out[r] = (i1.status, 0, h.lio_get_state18().vec(), np.zeros(1))


t0 = time.time()
th = [threading.Thread(target=rank, args=(r,)) for r in range(world)]
[t.start() for t in th]
[t.join() for t in th]
dt = time.time() - t0
ok = all(o is not None for o in out) and all((o[0] & 8) == 0 and (o[1] & 8) == 0 for o in out)
same = all(np.array_equal(o[2], out[0][2]) and np.array_equal(o[3], out[0][3]) for o in out)
print(json.dumps({"world": world, "multi_pass_launches_per_rank": launches, "passes_per_rank": 10 * launches, "seconds": round(dt, 2),
                  "no_timeout": bool(ok), "ranks_bitwise_equal": bool(same), "finite": bool(np.isfinite(out[0][2]).all())}))
