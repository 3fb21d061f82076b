"""Worker of test_two_processes_over_hip_ipc: rank r of `world` processes on device 0; fl_p2p_export -> handles through files ->
fl_p2p_connect -> a sharded LIO frame segment. Prints "OK <state vector>"."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fastlivo  # noqa: F401,E402
from fast_livo_amd import capi, synth  # noqa: E402

rank, world, d = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
scene = synth.make_scene()
n = 20000
fr = synth.make_lio_frame(n, scene=scene)
nbr, valid = synth.knn5(scene, fr.world_at(fr.R_prior, fr.p_prior))
h = capi.Handle(capi.config_from_frames(fr, max_iterations=5))
mine = h.p2p_export(world)
tmp = os.path.join(d, f"h{rank}.tmp")
open(tmp, "wb").write(mine)
os.rename(tmp, os.path.join(d, f"h{rank}.bin"))
handles = []
for r in range(world):
    f = os.path.join(d, f"h{r}.bin")
    t0 = time.time()
    while not os.path.exists(f):
        if time.time() - t0 > 120:
            raise SystemExit("peer handle never appeared")
        time.sleep(0.01)
    handles.append(open(f, "rb").read())
h.p2p_connect(rank, world, handles)
open(os.path.join(d, f"c{rank}.ok"), "w").write("1")          # nobody publishes before everybody has mapped everybody
for r in range(world):
    while not os.path.exists(os.path.join(d, f"c{r}.ok")):
        time.sleep(0.01)
x0 = capi.state18_from_frame(fr)
sl = slice(rank * n // world, (rank + 1) * n // world)
h.lio_set_points(fr.body_xyz[sl]); h.lio_begin18(x0, x0); h.lio_set_neighbours(nbr[sl], valid[sl])
info = h.lio_iterate18(6, 0)
x = h.lio_get_state18()
assert info.status == 0, info.status
print("OK " + " ".join(repr(float(v)) for v in x.vec()))
h.close()
