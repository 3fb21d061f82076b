"""per-kernel durations of the voxel filter: run under rocprofv3 --kernel-trace --stats (tools/ktrace.sh style)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastlivo  # noqa
from fast_livo_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24000
fr = synth.make_lio_frame(n)
p = np.concatenate([fr.body_xyz, np.zeros((n, 1), np.float32)], 1).astype(np.float32)
h = capi.Handle(capi.config_from_frames(fr))
for _ in range(30):
    h.scan_voxel_filter(p, 0.15, stage_as_scan=True, want=False)
h.close()
