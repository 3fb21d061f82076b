"""Writes the input of fast-livo_amd/host/demo_pipeline (the frame bench.py's `pipeline` section uses):  python tools/make_pipe_bin.py out.bin [raw]
then e.g.  FL_KT_CMD=1 FL_DEMO_TIME_REPS=3 bash tools/ktrace.sh 90 fast-livo_amd/host/demo_pipeline out.bin   (camera-half time line)"""
import os, struct, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fastlivo  # noqa
from fast_livo_amd import capi, synth
fn = sys.argv[1]
raw = int(sys.argv[2]) if len(sys.argv) > 2 else 100000
leaf, cell, n_imu = 0.15, 0.5, 20
scene = synth.make_scene()
lio = synth.make_lio_frame(raw, scene=scene)
f = synth.make_imu_frame(raw, n_imu=n_imu, lio=lio, quiet=True)
f.pts_xyzt[:, :3] = lio.body_xyz
x0 = capi.state18_from_frame(lio)
with open(fn, "wb") as fh:
    fh.write(struct.pack("<iiiiffdd", raw, f.imu.shape[0], scene.map_xyz.shape[0], 10, leaf, cell, f.pcl_beg_time, f.pcl_end_time))
    fh.write(np.asarray(lio.R_LI, dtype="<f8").tobytes()); fh.write(np.asarray(lio.t_LI, dtype="<f8").tobytes())
    fh.write(x0.vec().astype("<f8").tobytes()); fh.write(np.asarray(x0.cov_np(), dtype="<f8").tobytes())
    fh.write(bytes(capi.imu_proc_from_frame(f)))
    fh.write(np.ascontiguousarray(f.imu, dtype="<f8").tobytes())
    fh.write(f.pts_xyzt.astype("<f4").tobytes()); fh.write(scene.map_xyz.astype("<f4").tobytes())
    Rci = np.eye(3) @ lio.R_LI.T
    fh.write(np.asarray(Rci, dtype="<f8").tobytes()); fh.write(np.asarray(-lio.R_LI.T @ lio.t_LI, dtype="<f8").tobytes())
    fh.write(synth.make_image(640, 512, seed=3).tobytes())
print("wrote", fn)
