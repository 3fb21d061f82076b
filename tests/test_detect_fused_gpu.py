"""fl_vio_detect as ONE enqueue (round 6, csrc/api_vmap.inc: FL_OPT_DETECT_FUSED) -- LidarSelector::detect (lidar_selection.cpp:1027-1076:
addFromSparseMap -> addSparseMap -> ComputeJ -> addObservation) with the candidate / accepted-patch / founded / observed counts kept on the
device, the launches sized for their upper bound (one per grid cell), one result mailbox.

Against (a) the staged form (the six calls sequenced inside the library): counts, state, covariance, per-patch errors and the whole visual
map BIT FOR BIT over a multi-frame walk -- same partition, same record order, same arithmetic; (b) the CPU oracle driven as detect() drives it
(oracle/orc_vmap.c + orc_vio.c): the same points tracked / founded / observed every frame, state 1e-9, covariance 1e-11, visual maps equal.
Frames without a selection (the first one: empty map; one with the camera turned away) take ComputeJ's early return (:969): state untouched."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _VF:                       # what oracle.vio_compute_j / vio_config read
    pass


def _frame_pose(Rci, Pci, rot9, pos3):
    """vmap_frame_pose of api_vmap.inc, operation for operation (plain python floats: no fused multiply-add)"""
    Rcw = np.zeros((3, 3)); Pcw = np.zeros(3)
    for i in range(3):
        for j in range(3):
            a = 0.0
            for k in range(3):
                a += float(Rci[i, k]) * float(rot9[j * 3 + k])
            Rcw[i, j] = a
    for i in range(3):
        a = 0.0
        for k in range(3):
            a += float(Rcw[i, k]) * float(pos3[k])
        Pcw[i] = -a + float(Pci[i])
    return Rcw, Pcw


def _walk(capi, orc, synth, scene, frames, n_scan, max_iter, grid, fused, with_oracle, blind_frame=-1, outlier=3000.0, drop=None, refuse=None, spec=None):
    """drop = (frame, passes_ahead): the instrumented build's fault injector makes a producer of that ComputeJ pass withhold its record"""
    fr0 = synth.make_lio_frame(n_scan, scene=scene)
    vf0 = synth.make_vio_frame(8, fr0, max_iterations=max_iter)
    h = capi.Handle(capi.config_from_frames(fr0, vf0, max_iterations=max_iter), debug=drop is not None or refuse is not None)
    if drop is not None:
        h.debug_drop_record(1 << 30)            # (park the process-wide injector: an earlier test may have left it armed)
    h.set_option(capi.FL_OPT_DETECT_FUSED, 1 if fused else 0)
    if spec is not None:
        h.set_option(capi.FL_OPT_VIO_SPECULATE, spec)       # 2 (default): ComputeJ's three pyramid levels in one launch, 1: a launch per level
    h.vmap_clear(grid)
    Rli = fr0.R_LI.T
    Rci = vf0.Rcl @ Rli
    Pci = vf0.Rcl @ (-fr0.R_LI.T @ fr0.t_LI) + vf0.Pcl
    vm = orc.VMap(orc.vio_config(vf0), grid) if with_oracle else None
    R_t, p_t = fr0.R_true.copy(), fr0.p_true.copy()
    xg = capi.State18.make(R_t, p_t, fr0.vel, fr0.bg, fr0.ba, fr0.grav, fr0.cov18)
    xo = orc.State18.make(R_t, p_t, fr0.vel, fr0.bg, fr0.ba, fr0.grav, fr0.cov18)
    Q = np.diag([1e-5] * 3 + [1e-4] * 3 + [1e-3] * 3 + [1e-8] * 9)
    kf_imgs, log = [], []
    for k in range(frames):
        R_t = R_t @ synth.exp_so3(np.array([0.0, 0.0, 0.01]))
        p_t = p_t + np.array([0.05, 0.03, 0.0])
        body = synth.scan_from_pose(scene, R_t, p_t, n_scan, seed=600 + k)
        Rc_t, Pc_t = synth.cam_pose(vf0.Rcl, vf0.Pcl, fr0.R_LI, fr0.t_LI, R_t, p_t)
        img = synth.render_image(scene, vf0.cam, Rc_t, Pc_t, seed=k)
        # the LIO posterior stands in as "true pose + a small error": the camera half is what is under test
        dR = synth.exp_so3(np.array([0.002, -0.001, 0.0015])); dp = np.array([0.01, -0.008, 0.005])
        xg = capi.State18.make(R_t @ dR, p_t + dp, fr0.vel, fr0.bg, fr0.ba, fr0.grav, xg.cov_np() + Q)
        world = (body.astype(np.float64) @ (R_t @ fr0.R_LI).T + (R_t @ fr0.t_LI + p_t)).astype(np.float32)
        if k == blind_frame:
            world = (world + np.float32(500.0)).astype(np.float32)        # a scan nowhere near the map: nothing is selected
        down, _ = orc.voxel_grid(np.concatenate([world, np.zeros((n_scan, 1), np.float32)], axis=1), 0.2)
        down = np.ascontiguousarray(down[:, :3])
        x_in = xg.copy()
        if drop is not None and k == drop[0]:
            r0 = h.diagnostics()["resumes"]
            h.debug_drop_record(drop[1])
        if refuse is not None and k == refuse[0]:
            f0 = h.diagnostics()["fallbacks"]
            h.debug_mp_refuse(refuse[1], refuse[2])
        ns, na, no = h.vio_detect(img, world, down, Rci, Pci, xg, k, outlier_threshold=outlier)
        if refuse is not None and k == refuse[0]:
            assert h.diagnostics()["fallbacks"] - f0 >= 1, "no reservation was refused"
            h.debug_mp_refuse(0, 0)
        if drop is not None and k == drop[0]:
            assert h.diagnostics()["resumes"] - r0 >= 1, "the pass was not abandoned: the injector missed ComputeJ's launches"
            h.debug_drop_record(1 << 30)
        rec = dict(counts=(ns, na, no), x=xg.vec().copy(), P=xg.cov_np().copy(), errors=h.vio_get_errors(ns).copy() if ns > 0 else np.zeros(0, np.float32))
        if ns == 0:
            assert np.array_equal(xg.vec(), x_in.vec()) and np.array_equal(xg.cov_np(), x_in.cov_np()), f"frame {k}: no selection, state must be untouched"
        log.append(rec)
        if not with_oracle:
            continue
        xo = orc.State18.make(R_t @ dR, p_t + dp, fr0.vel, fr0.bg, fr0.ba, fr0.grav, xo.cov_np() + Q)
        kf_imgs.append(img)
        Rcw, Pcw = _frame_pose(Rci, Pci, xo.rot, xo.pos)
        o = vm.select(Rcw, Pcw, img, kf_imgs, down, outlier_threshold=outlier)
        na_o = vm.add_sparse(Rcw, Pcw, img, world, k, k)
        m = len(o["points"])
        assert (ns, na) == (m, na_o), f"frame {k}: selected / founded {ns, na} vs oracle {m, na_o}"
        if m > 0:
            vf = _VF()
            for a in ("Rcl", "Pcl", "R_LI", "t_LI", "cam", "img_point_cov", "max_iterations", "patch_size"):
                setattr(vf, a, getattr(vf0, a))
            vf.m = m; vf.img = img
            vf.ref_patch = np.ascontiguousarray(o["patches"].reshape(m, 3, 64))
            vf.pos = np.ascontiguousarray(np.stack([vm.get_point(int(i))[0] for i in o["points"]]))
            vf.search_level = np.ascontiguousarray(o["levels"].astype(np.int32))
            ro = orc.vio_compute_j(vf, xo, xo.copy())
            assert np.abs(xg.vec() - xo.vec()).max() <= 1e-9, f"frame {k} state"
            assert np.abs(xg.cov_np() - xo.cov_np()).max() <= 1e-11, f"frame {k} cov"
            assert np.array_equal(rec["errors"], ro["errors"]), f"frame {k}: per-patch errors"
        Rc2, Pc2 = _frame_pose(Rci, Pci, xo.rot, xo.pos)
        no_o = vm.add_observation(Rc2, Pc2, img, o["points"], o["levels"], k, k)
        assert no == no_o, f"frame {k}: observed {no} vs oracle {no_o}"
        # the next frame starts from the device's state on both sides (the comparison is per frame, not of accumulated drift)
        xo = orc.State18.make(np.array(xg.rot).reshape(3, 3), xg.pos[:], xg.vel[:], xg.bg[:], xg.ba[:], xg.grav[:], xg.cov_np())
    vmap = [h.vmap_get_point(i) for i in range(h.vmap_size())]
    if with_oracle:
        assert h.vmap_size() == vm.size()
        for i in range(vm.size()):
            pg, vg, obg = vmap[i]
            po, vo, obo = vm.get_point(i)
            assert np.array_equal(pg, po) and vg == vo and len(obg) == len(obo), f"visual map point {i}"
        vm.close()
    h.close()
    return log, vmap


def _same_obs(a, b):
    return all(np.array_equal(np.frombuffer(bytes(x), np.uint8), np.frombuffer(bytes(y), np.uint8)) for x, y in zip(a, b))


@pytest.mark.parametrize("n_scan,grid,max_iter", [(3000, 40, 4), (6000, 20, 10)])
def test_fused_detect_equals_staged_bit_for_bit(gpu_lib, oracle_lib, scene, n_scan, grid, max_iter):
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    logs = {}
    for fused in (True, False):
        logs[fused] = _walk(capi, orc, synth, scene, 7, n_scan, max_iter, grid, fused, False, blind_frame=4)
    (lf, mf), (ls, ms) = logs[True], logs[False]
    assert sum(r["counts"][0] for r in lf) > 100 and lf[4]["counts"][0] == 0, [r["counts"] for r in lf]
    for k, (a, b) in enumerate(zip(lf, ls)):
        assert a["counts"] == b["counts"], f"frame {k}: {a['counts']} vs {b['counts']}"
        assert np.array_equal(a["x"], b["x"]) and np.array_equal(a["P"], b["P"]), f"frame {k}: state"
        assert np.array_equal(a["errors"], b["errors"]), f"frame {k}: per-patch errors"
    assert len(mf) == len(ms)
    for i, (p, q) in enumerate(zip(mf, ms)):
        assert np.array_equal(p[0], q[0]) and p[1] == q[1] and len(p[2]) == len(q[2]) and _same_obs(p[2], q[2]), f"visual map point {i}"


def test_fused_detect_matches_the_oracle(gpu_lib, oracle_lib, scene):
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    log, _ = _walk(capi, orc, synth, scene, 6, 3000, 4, 40, True, True, blind_frame=3)
    assert sum(r["counts"][0] for r in log) > 60 and sum(r["counts"][2] for r in log) > 0, [r["counts"] for r in log]


@pytest.mark.parametrize("fused", [True, False])
def test_detect_with_the_scan_on_the_device(gpu_lib, oracle_lib, scene, fused):
    """n_pg = FL_DETECT_SCAN_ON_DEVICE: pg = the handle's staged scan under state_io (pointBodyToWorld), its 0.2 m down-sampling by the device
    voxel filter, the count never leaving the device (lidar_selection.cpp:352-353). Against a second handle that is handed both clouds (computed
    by fl_lio_get_world_points / fl_scan_voxel_filter, i.e. the same kernels): counts, states, per-patch errors and the visual map bit for bit.
    fused = False: FL_OPT_DETECT_FUSED 0 on the device-scan handle -- the fallback that materialises the clouds and runs the staged calls."""
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    n_scan, max_iter, grid, frames = 4000, 4, 40, 6
    fr0 = synth.make_lio_frame(n_scan, scene=scene)
    vf0 = synth.make_vio_frame(8, fr0, max_iterations=max_iter)
    hs = [capi.Handle(capi.config_from_frames(fr0, vf0, max_iterations=max_iter)) for _ in range(2)]
    hs[1].set_option(capi.FL_OPT_DETECT_FUSED, 1 if fused else 0)
    for h in hs:
        h.vmap_clear(grid)
    Rci = vf0.Rcl @ fr0.R_LI.T
    Pci = vf0.Rcl @ (-fr0.R_LI.T @ fr0.t_LI) + vf0.Pcl
    R_t, p_t = fr0.R_true.copy(), fr0.p_true.copy()
    Q = np.diag([1e-5] * 3 + [1e-4] * 3 + [1e-3] * 3 + [1e-8] * 9)
    cov = fr0.cov18.copy()
    tracked = 0
    for k in range(frames):
        R_t = R_t @ synth.exp_so3(np.array([0.0, 0.0, 0.01]))
        p_t = p_t + np.array([0.05, 0.03, 0.0])
        body = synth.scan_from_pose(scene, R_t, p_t, n_scan, seed=700 + k)
        Rc_t, Pc_t = synth.cam_pose(vf0.Rcl, vf0.Pcl, fr0.R_LI, fr0.t_LI, R_t, p_t)
        img = synth.render_image(scene, vf0.cam, Rc_t, Pc_t, seed=k)
        dR = synth.exp_so3(np.array([0.002, -0.001, 0.0015])); dp = np.array([0.01, -0.008, 0.005])
        xs = [capi.State18.make(R_t @ dR, p_t + dp, fr0.vel, fr0.bg, fr0.ba, fr0.grav, cov + Q) for _ in range(2)]
        # handle 0: both clouds through the host
        hs[0].lio_set_points(body); hs[0].lio_begin18(xs[0], xs[0])
        world = hs[0].lio_get_world_points(n_scan)
        down, nd, _ = hs[0].scan_voxel_filter(np.ascontiguousarray(np.concatenate([world, np.zeros((n_scan, 1), np.float32)], axis=1)), 0.2)
        down = np.ascontiguousarray(down[:nd, :3])
        c0 = hs[0].vio_detect(img, world, down, Rci, Pci, xs[0], k, outlier_threshold=3000.0)
        # handle 1: nothing but the image goes up
        hs[1].lio_set_points(body)
        c1 = hs[1].vio_detect(img, None, None, Rci, Pci, xs[1], k, outlier_threshold=3000.0)
        assert c0 == c1, f"frame {k}: {c0} vs {c1}"
        assert np.array_equal(xs[0].vec(), xs[1].vec()) and np.array_equal(xs[0].cov_np(), xs[1].cov_np()), f"frame {k}: state"
        if c0[0] > 0:
            assert np.array_equal(hs[0].vio_get_errors(c0[0]), hs[1].vio_get_errors(c1[0])), f"frame {k}: per-patch errors"
        tracked += c0[0]
        cov = xs[0].cov_np()
    assert tracked > 60 and hs[0].vmap_size() == hs[1].vmap_size()
    for i in range(hs[0].vmap_size()):
        p, q = hs[0].vmap_get_point(i), hs[1].vmap_get_point(i)
        assert np.array_equal(p[0], q[0]) and p[1] == q[1] and len(p[2]) == len(q[2]) and _same_obs(p[2], q[2]), f"visual map point {i}"
    for h in hs:
        h.close()


@pytest.mark.parametrize("passes_ahead", [0, 1, 3])
def test_fused_detect_with_an_abandoned_computej_pass(gpu_lib, oracle_lib, scene, passes_ahead):
    """A hand-off time-out inside ComputeJ's launches of the fused frame (fault injector of the instrumented build): the pass is abandoned, every
    launch behind it is a no-op, vmap_addobs_dev_kernel leaves the map alone and publishes; the host resumes ComputeJ per pass and adds the
    observations through the staged call. Counts, state, per-patch errors and the visual map must equal the undisturbed walk's, bit for bit --
    in the faulted frame and in the frames after it."""
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    good, map_good = _walk(capi, orc, synth, scene, 5, 3000, 4, 40, True, False)
    bad, map_bad = _walk(capi, orc, synth, scene, 5, 3000, 4, 40, True, False, drop=(2, passes_ahead))
    assert good[2]["counts"][0] > 10
    for k, (a, b) in enumerate(zip(good, bad)):
        assert a["counts"] == b["counts"], f"frame {k}: {a['counts']} vs {b['counts']}"
        assert np.array_equal(a["x"], b["x"]) and np.array_equal(a["P"], b["P"]), f"frame {k}: state"
        assert np.array_equal(a["errors"], b["errors"]), f"frame {k}: per-patch errors"
    assert len(map_good) == len(map_bad)
    for i, (p, q) in enumerate(zip(map_good, map_bad)):
        assert np.array_equal(p[0], q[0]) and p[1] == q[1] and len(p[2]) == len(q[2]) and _same_obs(p[2], q[2]), f"visual map point {i}"


@pytest.mark.parametrize("spec,nth,count", [(1, 1, 1), (1, 1, 2), (1, 2, 1), (1, 2, 2), (1, 3, 2), (1, 1, 6), (2, 1, 1), (2, 1, 2), (2, 1, 6)])
def test_fused_detect_losing_the_multipass_admission_half_way(gpu_lib, oracle_lib, scene, spec, nth, count):
    """The fused frame's ComputeJ launches take their patch count from the device, which only the multi-pass kernels can do. If the admission
    is lost at some level (another handle of the process launched in between; here: the debug library refuses `count` reservations from the nth
    on -- with a launch per pyramid level (FL_OPT_VIO_SPECULATE 1) each level tries the whole-CU variant first, then the shared one: (1, 1)
    moves level 2 to the shared variant, (1, 2) refuses level 2 outright, (2, 2) level 1, (3, 2) level 0, (1, 6) every level), nothing more is
    enqueued, the counts are read back and the remaining levels run per pass with launches that know the count; addObservation goes through the
    staged call. With all levels in ONE launch (2, the default) there is one reservation: (1, 1) sends level 2 to the shared variant and
    levels 1-0 into one launch, (1, 2) and (1, 6) refuse. Same bits as the undisturbed walk of the default form."""
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    good, map_good = _walk(capi, orc, synth, scene, 4, 3000, 4, 40, True, False)
    bad, map_bad = _walk(capi, orc, synth, scene, 4, 3000, 4, 40, True, False, refuse=(2, nth, count), spec=spec)
    for k, (a, b) in enumerate(zip(good, bad)):
        assert a["counts"] == b["counts"], f"frame {k}: {a['counts']} vs {b['counts']}"
        assert np.array_equal(a["x"], b["x"]) and np.array_equal(a["P"], b["P"]), f"frame {k}: state"
        assert np.array_equal(a["errors"], b["errors"]), f"frame {k}: per-patch errors"
    assert len(map_good) == len(map_bad)
    for i, (p, q) in enumerate(zip(map_good, map_bad)):
        assert np.array_equal(p[0], q[0]) and p[1] == q[1] and len(p[2]) == len(q[2]) and _same_obs(p[2], q[2]), f"visual map point {i}"


def test_device_scan_detect_when_the_downsampling_grid_outgrows_its_bitmap(gpu_lib, oracle_lib, scene):
    """A scan with a few far-away returns: the 0.2 m down-sampling grid of its bounding box has more cells than the occupancy bitmap is allocated
    for (2^27). The fused frame with the scan on the device notices that on the device (nothing has touched the visual map: vmap_commit_kernel
    leaves, nothing is selected), the host grows the bitmap, gives the keyframe slot back and runs the frame again. Against a handle that is
    handed both clouds (fl_scan_voxel_filter grows its bitmap the same way): every count, state and map point equal."""
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    n_scan, max_iter, grid = 4000, 4, 40
    fr0 = synth.make_lio_frame(n_scan, scene=scene)
    vf0 = synth.make_vio_frame(8, fr0, max_iterations=max_iter)
    hs = [capi.Handle(capi.config_from_frames(fr0, vf0, max_iterations=max_iter)) for _ in range(2)]
    for h in hs:
        h.vmap_clear(grid)
    Rci = vf0.Rcl @ fr0.R_LI.T
    Pci = vf0.Rcl @ (-fr0.R_LI.T @ fr0.t_LI) + vf0.Pcl
    R_t, p_t = fr0.R_true.copy(), fr0.p_true.copy()
    cov = fr0.cov18.copy()
    for k in range(4):
        R_t = R_t @ synth.exp_so3(np.array([0.0, 0.0, 0.01]))
        p_t = p_t + np.array([0.05, 0.03, 0.0])
        body = synth.scan_from_pose(scene, R_t, p_t, n_scan, seed=900 + k).copy()
        if k == 2:                          # three returns a few hundred metres out: 700 x 500 x 120 m / 0.2^3 = 5.3e9 ... too many; 350 x 300 x 60 m = 7.9e8 cells > 2^27
            body[-3:] = np.float32([[330.0, 5.0, 2.0], [10.0, 280.0, 1.0], [4.0, -3.0, 55.0]])
        img = synth.render_image(scene, vf0.cam, *synth.cam_pose(vf0.Rcl, vf0.Pcl, fr0.R_LI, fr0.t_LI, R_t, p_t), seed=k)
        xs = [capi.State18.make(R_t, p_t, fr0.vel, fr0.bg, fr0.ba, fr0.grav, cov) for _ in range(2)]
        hs[0].lio_set_points(body); hs[0].lio_begin18(xs[0], xs[0])
        world = hs[0].lio_get_world_points(n_scan)
        down, nd, _ = hs[0].scan_voxel_filter(np.ascontiguousarray(np.concatenate([world, np.zeros((n_scan, 1), np.float32)], axis=1)), 0.2)
        c0 = hs[0].vio_detect(img, world, np.ascontiguousarray(down[:nd, :3]), Rci, Pci, xs[0], k, outlier_threshold=3000.0)
        hs[1].lio_set_points(body)
        c1 = hs[1].vio_detect(img, None, None, Rci, Pci, xs[1], k, outlier_threshold=3000.0)
        assert c0 == c1, f"frame {k}: {c0} vs {c1}"
        assert np.array_equal(xs[0].vec(), xs[1].vec()) and np.array_equal(xs[0].cov_np(), xs[1].cov_np()), f"frame {k}"
        cov = xs[0].cov_np()
    assert hs[0].vmap_size() == hs[1].vmap_size() > 50
    for i in range(hs[0].vmap_size()):
        p, q = hs[0].vmap_get_point(i), hs[1].vmap_get_point(i)
        assert np.array_equal(p[0], q[0]) and p[1] == q[1] and len(p[2]) == len(q[2]) and _same_obs(p[2], q[2]), f"visual map point {i}"
    for h in hs:
        h.close()
