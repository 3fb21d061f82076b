import sys, os
sys.path.insert(0, "/root/repo")
import numpy as np
import fastlivo
from fast_livo_amd import capi, synth
scene = synth.make_scene()
fr = synth.make_lio_frame(50000, scene=scene)
vf = synth.make_vio_frame(2000, fr)
outs = []
for name, opts in (("default", {}), ("whole_cu=0", {capi.FL_OPT_VIO_WHOLE_CU: 0}), ("multipass=0", {capi.FL_OPT_MULTIPASS: 0}), ("default again", {})):
    h = capi.Handle(capi.config_from_frames(fr, vf, max_iterations=10))
    for k, v in opts.items():
        h.set_option(k, v)
    h.vio_set_frame(vf.img); h.vio_set_patches(vf.ref_patch, vf.pos, vf.search_level)
    xv = capi.state18_from_frame(fr)
    infos = h.vio_compute_j(xv, capi.state18_from_frame(fr))
    e = h.vio_get_errors(vf.m)
    outs.append((xv.vec().copy(), e.copy(), [(i.iterations, i.accepted, i.status) for i in infos]))
    print(name, outs[-1][2], "state == default:", np.array_equal(outs[-1][0], outs[0][0]), "errors == default:", np.array_equal(outs[-1][1].view(np.uint32), outs[0][1].view(np.uint32)),
          "max |dx|", np.abs(outs[-1][0] - outs[0][0]).max())
    h.close()
