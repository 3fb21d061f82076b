"""pcl::VoxelGrid on the device vs the CPU restatement (oracle/orc_voxel.c), through the C ABI.

Voxel membership and output order must be identical; the centroids are bit-identical because the device sums each voxel
in the same (ascending input index) order with the same float operations."""
import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def _cloud(n, seed, span=20.0):
    rng = np.random.default_rng(seed)
    p = np.empty((n, 4), np.float32)
    p[:, :3] = rng.uniform(-span, span, (n, 3)).astype(np.float32)
    p[:, 3] = rng.uniform(0, 255, n).astype(np.float32)
    return p


def _check(h, p, leaf):
    ref, ref_small = orc.voxel_grid(p, leaf)
    out, m, small = h.scan_voxel_filter(p, leaf)
    assert small == ref_small
    assert m == ref.shape[0]
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))
    return out


@pytest.mark.parametrize("n,leaf", [(1, 0.5), (7, 0.5), (1000, 0.15), (24000, 0.15), (24000, (0.2, 0.3, 0.5)), (200000, 0.5)])
def test_random_clouds(gpu_lib, n, leaf):
    from fast_livo_amd import capi, synth
    fr = synth.make_lio_frame(1000)
    h = capi.Handle(capi.config_from_frames(fr))
    _check(h, _cloud(n, 100 + n), leaf)


def test_scan_like_cloud_and_reuse(gpu_lib):
    from fast_livo_amd import capi, synth
    fr = synth.make_lio_frame(120000)
    h = capi.Handle(capi.config_from_frames(fr))
    p = np.concatenate([fr.body_xyz, np.linspace(0, 100, fr.n, dtype=np.float32)[:, None]], 1).astype(np.float32)
    out = _check(h, p, 0.15)
    assert 1000 < out.shape[0] < fr.n
    _check(h, p[:5000], 0.3)          # smaller cloud on the same handle
    _check(h, p, 0.5)


def test_all_in_one_voxel_and_duplicates(gpu_lib):
    from fast_livo_amd import capi, synth
    h = capi.Handle(capi.config_from_frames(synth.make_lio_frame(1000)))
    p = _cloud(5000, 5, span=0.01) + np.float32([3.3, -1.2, 0.7, 0])
    out = _check(h, p, 1.0)
    assert out.shape[0] <= 8
    q = np.repeat(_cloud(10, 6), 50, axis=0)
    _check(h, q, 0.25)


@pytest.mark.parametrize("n,span,leaf", [(300000, 4.0, 0.25), (60000, 2.0, 0.2), (400000, 20.0, 0.15)])
def test_dense_clouds_mix_voxels_of_few_and_of_many_members(gpu_lib, n, span, leaf):
    """vx_centroid_kernel (round 6) puts a voxel's members in order itself: up to 8 in its thread's registers (a sorting network), more
    than 8 by the workgroup (ranking spread over (voxel, member) pairs, then a thread per voxel sums). ~9 members per voxel on average:
    both paths in every workgroup, some voxels with dozens of members."""
    from fast_livo_amd import capi, synth
    h = capi.Handle(capi.config_from_frames(synth.make_lio_frame(1000)))
    out = _check(h, _cloud(n, 4242 + n, span=span), leaf)
    assert out.shape[0] < n


def test_non_finite_points_are_skipped(gpu_lib):
    from fast_livo_amd import capi, synth
    h = capi.Handle(capi.config_from_frames(synth.make_lio_frame(1000)))
    p = _cloud(3000, 7)
    p[::17, 0] = np.nan
    p[5::29, 2] = np.inf
    p[3::31, 1] = -np.inf
    _check(h, p, 0.4)
    allbad = np.full((10, 4), np.nan, np.float32)
    out, m, small = h.scan_voxel_filter(allbad, 0.5)
    assert m == 0


def test_leaf_too_small_returns_input(gpu_lib):
    from fast_livo_amd import capi, synth
    h = capi.Handle(capi.config_from_frames(synth.make_lio_frame(1000)))
    p = _cloud(2000, 8, span=500.0)
    ref, ref_small = orc.voxel_grid(p, 0.001)
    assert ref_small
    out, m, small = h.scan_voxel_filter(p, 0.001)
    assert small and m == p.shape[0]
    assert np.array_equal(out.view(np.uint32), p.view(np.uint32))


def test_stage_as_scan_feeds_the_lio_frame(gpu_lib):
    """Down-sampled scan staged on the device == fl_lio_set_points of the oracle's centroids: same frame result."""
    from fast_livo_amd import capi, synth
    fr = synth.make_lio_frame(60000)
    scene = fr.scene
    h = capi.Handle(capi.config_from_frames(fr, max_iterations=4))
    h.map_set_points(scene.map_xyz, 0.5)
    p = np.concatenate([fr.body_xyz, np.zeros((fr.n, 1), np.float32)], 1).astype(np.float32)
    ref, _ = orc.voxel_grid(p, 0.2)
    # path A: filter on the device, result stays there
    _, m, _ = h.scan_voxel_filter(p, 0.2, stage_as_scan=True, want=False)
    assert m == ref.shape[0]
    xa = capi.state18_from_frame(fr)
    ia = h.lio_frame18_dev(xa, None)
    # path B: host-side centroids through fl_lio_set_points
    xb = capi.state18_from_frame(fr)
    ib = h.lio_frame18_dev(xb, np.ascontiguousarray(ref[:, :3]))
    assert ia.iterations == ib.iterations and ia.effct_feat_num == ib.effct_feat_num
    assert bytes(xa) == bytes(xb)


def test_sort_free_path_equals_the_sorted_path(gpu_lib):
    """fl_set_option(FL_OPT_VOXEL_SORT): the round-1 form (radix sort of (voxel index, point)) and the occupancy-bitmap form give the same
    bits; dense voxels (hundreds of members each: the per-point ordering), the same handle run after run (the bitmap is left clean)."""
    from fast_livo_amd import capi, synth
    h = capi.Handle(capi.config_from_frames(synth.make_lio_frame(1000)))
    clouds = [(_cloud(30000, 21), 0.15), (_cloud(30000, 22, span=3.0), 0.5), (_cloud(5000, 23, span=1.0), 0.8), (_cloud(70000, 24), 0.3),
              (_cloud(30000, 21), 0.15)]
    for p, leaf in clouds:
        h.set_option(capi.FL_OPT_VOXEL_SORT, 1)
        a, ma, _ = h.scan_voxel_filter(p, leaf)
        h.set_option(capi.FL_OPT_VOXEL_SORT, 0)
        b, mb, _ = h.scan_voxel_filter(p, leaf)
        assert ma == mb and np.array_equal(a.view(np.uint32), b.view(np.uint32))
        _check(h, p, leaf)


def test_grid_larger_than_the_first_bitmap(gpu_lib):
    """a 150 m x 150 m x 60 m cloud at leaf 0.1: 1500 x 1500 x 600 = 1.35e9 cells > the 2^27 the bitmap starts with -- the run reports it, the
    bitmap grows, the filter runs again; smaller grids afterwards reuse the larger bitmap"""
    from fast_livo_amd import capi, synth
    h = capi.Handle(capi.config_from_frames(synth.make_lio_frame(1000)))
    rng = np.random.default_rng(31)
    p = np.empty((40000, 4), np.float32)
    p[:, 0] = rng.uniform(-75, 75, 40000); p[:, 1] = rng.uniform(-75, 75, 40000); p[:, 2] = rng.uniform(-30, 30, 40000)
    p[:, 3] = rng.uniform(0, 255, 40000)
    _check(h, p, 0.1)
    _check(h, _cloud(24000, 32), 0.15)
    _check(h, p, 0.1)
