"""Soak: the same all-device LIO frame and VIO ComputeJ repeated many times -- results must be bit-identical every time and no
hand-off time-out may reach the caller.
  python tools/soak.py [frames]                                  undisturbed
  python tools/soak.py [frames] --compete gemm [--size 8192]     beside ANOTHER PROCESS that keeps the device busy with fp32 GEMMs
                                                                 (a foreign compute client: the library's admission check cannot see it)
  ... --demote-after K                                           FL_OPT_DEMOTE_AFTER (0 = never demote; default: the library's 2)
Reports frame-time percentiles (host wall time of fl_lio_frame18_dev + fl_vio_compute_j), abandoned-and-resumed chains, demotions."""
import argparse, json, os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("frames", type=int, nargs="?", default=3000)
ap.add_argument("--compete", choices=("none", "gemm"), default="none")
ap.add_argument("--size", type=int, default=8192)
ap.add_argument("--demote-after", type=int, default=-1)
ap.add_argument("--hog-main", action="store_true", help=argparse.SUPPRESS)
a = ap.parse_args()

if a.hog_main:          # the competing process: back-to-back GEMMs until killed
    import torch
    x = torch.randn(a.size, a.size, device="cuda"); y = torch.randn(a.size, a.size, device="cuda")
    print("HOG READY", flush=True)
    while True:
        for _ in range(20):
            x = (x @ y) * (1.0 / a.size)
        torch.cuda.synchronize()

import fastlivo  # noqa
from fast_livo_amd import capi, synth
N = a.frames
fr = synth.make_lio_frame(50000)
h = capi.Handle(capi.config_from_frames(fr, max_iterations=10))
h.map_set_points(fr.scene.map_xyz, 0.5)
vf = synth.make_vio_frame(2000, fr)
hv = capi.Handle(capi.config_from_frames(fr, vf, max_iterations=10))
hv.vio_set_frame(vf.img); hv.vio_set_patches(vf.ref_patch, vf.pos, vf.search_level)
if a.demote_after >= 0:
    h.set_option(capi.FL_OPT_DEMOTE_AFTER, a.demote_after); hv.set_option(capi.FL_OPT_DEMOTE_AFTER, a.demote_after)
scan = h.host_alloc(fr.body_xyz.shape, np.float32); scan[:] = fr.body_xyz


def frame():
    x = capi.state18_from_frame(fr)
    t0 = time.perf_counter()
    info = h.lio_frame18_dev(x, scan)
    t1 = time.perf_counter()
    xv = capi.state18_from_frame(fr); xp = capi.state18_from_frame(fr)
    infos = hv.vio_compute_j(xv, xp)
    t2 = time.perf_counter()
    return bytes(x), bytes(xv), info, infos, t1 - t0, t2 - t1


first, firstv = None, None
for _ in range(20):                      # warm, undisturbed: the reference bits
    first, firstv, *_ = frame()
hog = None
if a.compete == "gemm":
    hog = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--hog-main", "--size", str(a.size)], stdout=subprocess.PIPE, text=True)
    assert "HOG READY" in hog.stdout.readline()
    time.sleep(1.0)
bad = badstatus = 0
tl, tv = [], []
d0, dv0 = h.diagnostics(), hv.diagnostics()
t0 = time.perf_counter()
try:
    for i in range(N):
        b, bv, info, infos, dl, dvv = frame()
        bad += (b != first) + (bv != firstv)
        badstatus += int((info.status & ~16) != 0) + sum(int((i_.status & ~16) != 0) for i_ in infos)      # bit 16 = the exact accept test ran
        tl.append(dl); tv.append(dvv)
finally:
    if hog:
        hog.kill(); hog.wait()
d1, dv1 = h.diagnostics(), hv.diagnostics()
tl, tv = np.array(tl) * 1e3, np.array(tv) * 1e3
tf = tl + tv
pc = lambda v: {"p50": round(float(np.percentile(v, 50)), 4), "p90": round(float(np.percentile(v, 90)), 4), "p99": round(float(np.percentile(v, 99)), 4), "max": round(float(v.max()), 3)}  # noqa: E731
print(json.dumps({"frames": N, "compete": a.compete if a.compete == "none" else f"{a.compete} {a.size}^3 fp32 in another process", "demote_after": a.demote_after,
                  "different_results": int(bad), "status_other_than_fragile": int(badstatus),
                  "lio_frame_ms": pc(tl), "vio_computej_ms": pc(tv), "frame_ms": pc(tf),
                  "chains_resumed": {"lio": d1["resumes"] - d0["resumes"], "vio": dv1["resumes"] - dv0["resumes"]},
                  "admission_fallbacks": {"lio": d1["fallbacks"] - d0["fallbacks"], "vio": dv1["fallbacks"] - dv0["fallbacks"]},
                  "demotions": {"lio": d1["demotions"] - d0["demotions"], "vio": dv1["demotions"] - dv0["demotions"]},
                  "seconds": round(time.perf_counter() - t0, 1)}))
