"""SURVEY 8f N2 (pixel-level part of addFromSparseMap): device selection / affine warp vs oracle/orc_select.c through the C ABI.

The device mirrors the oracle's operations one by one (IEEE double/float, no contraction, sequential float error sum), so
depth image, reasons, search levels, errors and patches are required to be bit-identical."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _setup(m, seed=20241108, **kw):
    from fast_livo_amd import capi, synth
    sf = synth.make_select_frame(m, seed=seed, **kw)
    h = capi.Handle(capi.config_from_frames(sf.lio, sf.vio))
    ids = [h.vio_add_keyframe(k) for k in sf.keyframes]
    h.vio_set_frame(sf.vio.img)
    return capi, synth, sf, h, ids


def _oracle(sf, cand, **kw):
    cfg = orc.vio_config(sf.vio)
    depth = orc.vio_depth_image(cfg, sf.Rcw, sf.Pcw, sf.scan_world)
    return depth, orc.vio_select(cfg, sf.Rcw, sf.Pcw, sf.vio.img, sf.keyframes, depth, cand, **kw)


def _same(dev, ref):
    assert np.array_equal(dev["reason"], ref["reason"])
    assert np.array_equal(dev["idx"], ref["idx"])
    assert np.array_equal(dev["levels"], ref["levels"])
    assert np.array_equal(dev["errors"].view(np.uint32), ref["errors"].view(np.uint32))
    assert np.array_equal(dev["patches"].view(np.uint32), ref["patches"].view(np.uint32))


@pytest.mark.parametrize("m,distortion", [(1, False), (37, False), (600, False), (2000, False), (600, True)])
def test_selection_matches_oracle(gpu_lib, m, distortion):
    """distortion=True: the shipped camera files carry radtan coefficients; world2cam applies them, cam2world is
    cv::undistortPoints' five sweeps (restated on both sides)"""
    capi, synth, sf, h, ids = _setup(m, distortion=distortion)
    depth, ref = _oracle(sf, orc.patch_candidates(sf), outlier_threshold=sf.outlier_threshold)
    dev = h.vio_select_patches(sf.Rcw, sf.Pcw, sf.scan_world, capi.patch_candidates(sf, ids), outlier_threshold=sf.outlier_threshold,
                               want_depth=True)
    assert np.array_equal(dev["depth"].view(np.uint32), depth.view(np.uint32))
    _same(dev, ref)
    if m >= 600:
        r = ref["reason"]
        assert (r == 0).sum() > 0.2 * m and (r == 1).sum() > 0 and (r == 4).sum() > 0      # all branches exercised


def test_ncc_gate(gpu_lib):
    capi, synth, sf, h, ids = _setup(500, seed=7)
    for thre in (0.2, 0.8):
        _, ref = _oracle(sf, orc.patch_candidates(sf), ncc_en=True, ncc_thre=thre, outlier_threshold=1e9)
        dev = h.vio_select_patches(sf.Rcw, sf.Pcw, sf.scan_world, capi.patch_candidates(sf, ids), ncc_en=True, ncc_thre=thre,
                                   outlier_threshold=1e9)
        # the NCC sums are doubles reduced in a different order on the device: allow flips only within 1e-9 of the threshold
        diff = np.nonzero(dev["reason"] != ref["reason"])[0]
        assert len(diff) == 0
        _same(dev, ref)
    assert (ref["reason"] == 3).sum() > 0


def test_identity_warp_reproduces_current_patch(gpu_lib):
    """Reference observation taken from the current pose in the current image: A = I, error ~ 0, everything accepted."""
    capi, synth, sf, h, ids = _setup(300, seed=11, n_keyframes=1, discont_frac=0.0)
    dev = h.vio_select_patches(sf.Rcw, sf.Pcw, sf.scan_world, capi.patch_candidates(sf, ids), outlier_threshold=sf.outlier_threshold)
    # the only rejections left are depth discontinuities caused by the unrelated scan returns of the synthetic scene
    assert set(np.unique(dev["reason"])) <= {0, 1} and len(dev["idx"]) > 200 and (dev["levels"] == 0).all()
    assert dev["errors"].max() < 1.0


def test_accepted_patches_feed_compute_j(gpu_lib):
    """The patch set staged by the selection == fl_vio_set_patches of the oracle's accepted patches: same ComputeJ result."""
    capi, synth, sf, h, ids = _setup(800, seed=3)
    lio = sf.lio
    _, ref = _oracle(sf, orc.patch_candidates(sf), outlier_threshold=sf.outlier_threshold)
    k = len(ref["idx"])
    assert k > 50
    dev = h.vio_select_patches(sf.Rcw, sf.Pcw, sf.scan_world, capi.patch_candidates(sf, ids), outlier_threshold=sf.outlier_threshold,
                               want_patches=False)
    xa = capi.state18_from_frame(lio); xp = capi.state18_from_frame(lio)
    h.vio_compute_j(xa, xp)
    ea = h.vio_get_errors(k)
    h2 = capi.Handle(capi.config_from_frames(lio, sf.vio))
    h2.vio_set_frame(sf.vio.img)
    h2.vio_set_patches(ref["patches"].reshape(k, 3, 64), sf.cand_pos[ref["idx"]], ref["levels"])
    xb = capi.state18_from_frame(lio)
    h2.vio_compute_j(xb, xp)
    eb = h2.vio_get_errors(k)
    assert bytes(xa) == bytes(xb)
    assert np.array_equal(ea.view(np.uint32), eb.view(np.uint32))


def test_keyframe_pool_reuse_and_errors(gpu_lib):
    capi, synth, sf, h, ids = _setup(50, seed=5)
    h.vio_drop_keyframe(ids[1])
    with pytest.raises(RuntimeError):
        h.vio_select_patches(sf.Rcw, sf.Pcw, sf.scan_world, capi.patch_candidates(sf, ids))
    new_id = h.vio_add_keyframe(sf.keyframes[1])
    assert new_id == ids[1]                     # the freed slot is reused
    _, ref = _oracle(sf, orc.patch_candidates(sf), outlier_threshold=sf.outlier_threshold)
    dev = h.vio_select_patches(sf.Rcw, sf.Pcw, sf.scan_world, capi.patch_candidates(sf, ids), outlier_threshold=sf.outlier_threshold)
    _same(dev, ref)
    # no candidates, no scan
    dev = h.vio_select_patches(sf.Rcw, sf.Pcw, np.zeros((0, 3), np.float32), (capi.PatchCandidate * 0)())
    assert len(dev["idx"]) == 0


@pytest.mark.parametrize("k,grid_size", [(1, 40), (500, 40), (30000, 40), (30000, 11), (5000, 64)])
def test_grid_competition_matches_oracle(gpu_lib, k, grid_size):
    """Projection + per-cell competition (lidar_selection.cpp:412-466): winners, distances, values and cell types identical,
    including exact distance ties (duplicated points: the later one of the list wins) and points behind / outside the image."""
    capi = gpu_lib
    from fast_livo_amd import synth
    sf = synth.make_select_frame(8, seed=19)
    h = capi.Handle(capi.config_from_frames(sf.lio, sf.vio))
    rng = np.random.default_rng(k + grid_size)
    cam = sf.vio.cam
    px = np.stack([rng.uniform(-60, cam["width"] + 60, k), rng.uniform(-60, cam["height"] + 60, k)], -1)
    depth = rng.uniform(-2.0, 25.0, k)                        # some behind the camera
    xyc = np.stack([(px[:, 0] - cam["cx"]) / cam["fx"], (px[:, 1] - cam["cy"]) / cam["fy"], np.ones(k)], -1) * depth[:, None]
    pos = (xyc - sf.Pcw) @ sf.Rcw
    if k > 100:
        pos[k // 2:k // 2 + 40] = pos[:40]                    # exact duplicates -> equal float distances
    value = rng.uniform(0, 500, k).astype(np.float32)
    ref = orc.vio_grid_select(orc.vio_config(sf.vio), sf.Rcw, sf.Pcw, pos, value, grid_size)
    dev = h.vio_grid_select(sf.Rcw, sf.Pcw, pos, value, grid_size)
    for key in ("winner", "grid_num"):
        assert np.array_equal(dev[key], ref[key]), key
    for key in ("map_dist", "map_value"):
        assert np.array_equal(dev[key].view(np.uint32), ref[key].view(np.uint32)), key
    if k >= 500:
        assert (ref["winner"] >= 0).sum() > 3
    # the FIRST point of the list alone in the image: it must be reported as the winner of its cell (index 0 is not "none")
    centre = (np.array([[0.0, 0.0, 6.0]]) - sf.Pcw) @ sf.Rcw
    ref1 = orc.vio_grid_select(orc.vio_config(sf.vio), sf.Rcw, sf.Pcw, centre, np.array([5.0], np.float32), grid_size)
    dev1 = h.vio_grid_select(sf.Rcw, sf.Pcw, centre, np.array([5.0], np.float32), grid_size)
    assert (ref1["winner"] == 0).sum() == 1 and np.array_equal(dev1["winner"], ref1["winner"])
