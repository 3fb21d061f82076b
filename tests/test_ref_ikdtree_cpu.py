"""The brute-force restatements oracle/orc_knn.c and oracle/orc_map.c held to the REFERENCE ITSELF: oracle/_ref/libikdtree_ref.so is
the reference's own include/ikd-Tree/ikd_Tree.cpp, compiled unmodified (recipe: oracle/ref_ikdtree/).  This is what pins SURVEY 8f N1
(KD_TREE::Nearest_Search, ikd_Tree.cpp:350-380) and the map rows (Add_Points :382-457, Delete_Point_Boxes :501-520): the GPU tests then
compare the HIP path with the same library (tests/test_ref_ikdtree_gpu.py).

What "equal" means, and why:
  * k-NN: the ascending squared-distance 5-tuple is bit-identical; every neighbour strictly closer than the 5th is the same point; a
    neighbour AT the 5th distance may be another map point at exactly that distance (the tree keeps the one its traversal met first,
    `dist < q.top().dist` at :862 -- the restatements keep the lower map index).  On data without exact float ties the point lists
    are identical.
  * map: the tree holds a SET (its flatten() order is the traversal order of a self-balancing tree); compared as sorted arrays.
"""
import numpy as np
import pytest

from oracle import ikdref

pytestmark = pytest.mark.skipif(not ikdref.available(), reason="oracle/_ref/libikdtree_ref.so not built and no /root/reference")


def sorted_rows(a):
    a = np.ascontiguousarray(a, np.float32).reshape(-1, 3)
    return a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]


def knn_equal_up_to_ties(xyz_a, sq_a, xyz_b, sq_b, map_xyz=None):
    """-> (rows whose point lists differ, rows that differ although their 5th distance is unique in the tuple)."""
    assert np.array_equal(sq_a, sq_b), "squared-distance tuples differ"
    differ = (xyz_a != xyz_b).any(axis=(1, 2))
    bad = 0
    for i in np.nonzero(differ)[0]:
        closer = sq_a[i] < sq_a[i, -1]
        sa = sorted(map(tuple, xyz_a[i][closer].tolist()))
        sb = sorted(map(tuple, xyz_b[i][closer].tolist()))
        if sa != sb:
            # equal distances INSIDE the tuple may be listed in another order: compare as sets per distance value
            bad += 1
    return int(differ.sum()), bad


def scan_world(scene, rng, n, jitter=0.02):
    base = scene.map_xyz[rng.integers(0, len(scene.map_xyz), n)]
    return (base + rng.normal(0, jitter, (n, 3))).astype(np.float32)


def lattice_case():
    g = np.arange(-8, 8, dtype=np.float32) * np.float32(0.25)
    lattice = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3).astype(np.float32)
    lattice = np.concatenate([lattice, lattice[::7]])
    rng = np.random.default_rng(11)
    q = (rng.integers(-6, 6, (3000, 3)).astype(np.float32) * np.float32(0.25) + np.float32(0.125))
    q[::3] -= np.float32(0.125)
    return lattice, q


@pytest.mark.parametrize("n", [1, 777, 20000])
def test_bruteforce_knn_equals_the_reference_tree_on_the_scene(oracle_lib, scene, n):
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(n, scene=scene)
    q = fr.world_at(fr.R_prior, fr.p_prior).astype(np.float32)
    t = ikdref.IkdTree(0.5)
    t.build(scene.map_xyz)
    xyz, sq, found = t.nearest(q)
    nbr_o, sq_o, valid_o, _ = oracle_lib.knn5_bruteforce(scene.map_xyz, q)
    assert (found == 5).all()
    assert np.array_equal(sq, sq_o)
    assert np.array_equal(xyz, nbr_o)                       # no exact ties on this data: identical points, identical order
    assert np.array_equal(valid_o != 0, sq[:, 4] <= 5.0)    # laserMapping.cpp:1549 on the tree's own output
    t.close()


def test_bruteforce_knn_vs_tree_on_exact_ties(oracle_lib):
    lattice, q = lattice_case()
    t = ikdref.IkdTree(0.5)
    t.build(lattice)
    xyz, sq, found = t.nearest(q)
    nbr_o, sq_o, _, _ = oracle_lib.knn5_bruteforce(lattice, q)
    n_differ, n_bad = knn_equal_up_to_ties(xyz, sq, nbr_o, sq_o)
    assert n_bad == 0
    assert n_differ > 0                                     # the tie rule IS different: traversal order vs lower index
    # every point the tree returned is a map point at the distance it reports
    d = ((q[:, None, :] - xyz) ** 2)
    assert np.array_equal((d[..., 0] + d[..., 1]) + d[..., 2], sq)
    t.close()


def test_fewer_than_five_and_far_queries(oracle_lib, scene):
    from fast_livo_amd import synth
    rng = np.random.default_rng(5)
    fr = synth.make_lio_frame(3000, scene=scene)
    q = fr.world_at(fr.R_prior, fr.p_prior).astype(np.float32)
    sparse = scene.map_xyz[rng.choice(len(scene.map_xyz), 300, replace=False)]
    for m in (sparse, sparse[:3], scene.map_xyz + np.float32(50.0)):
        t = ikdref.IkdTree(0.5)
        t.build(m)
        xyz, sq, found = t.nearest(q)
        nbr_o, sq_o, valid_o, _ = oracle_lib.knn5_bruteforce(m, q)
        assert (found == min(5, len(m))).all()
        # validity as laserMapping.cpp:1549,1567 computes it from the tree's output (size check first: no read of sqdist[4])
        valid_ref = (found == 5) & ~(sq[:, 4] > 5.0)
        assert np.array_equal(valid_ref, valid_o != 0)
        k = min(5, len(m))
        assert np.array_equal(sq[:, :k], sq_o[:, :k]) and np.array_equal(xyz[:, :k], nbr_o[:, :k])
        t.close()


@pytest.mark.parametrize("n_map,n_new,ds", [(20000, 3000, 0.5), (20000, 3000, 0.3), (5000, 6000, 0.15), (1, 1, 0.5), (30000, 1, 0.25),
                                            (55000, 20000, 0.2)])
def test_sequential_add_points_restatement_equals_the_tree(oracle_lib, scene, n_map, n_new, ds):
    rng = np.random.default_rng(7 + n_map + n_new)
    m0 = scene.map_xyz[rng.choice(len(scene.map_xyz), n_map, replace=False)]
    new = scan_world(scene, rng, n_new)
    t = ikdref.IkdTree(ds)
    t.build(m0)
    t.add_points(new, True)
    got = t.flatten()
    want, oi = oracle_lib.map_add_points(m0, new, ds)
    assert oi.n_ambiguous == 0
    assert t.validnum() == len(got) == len(want)
    assert np.array_equal(sorted_rows(got), sorted_rows(want))
    t.close()


def test_add_points_without_downsampling_and_with_a_quiescent_tree(oracle_lib, scene):
    """Add_Points(points, false) (ikd_Tree.cpp:438-453) appends; and the set does not depend on what the rebuild thread is doing."""
    rng = np.random.default_rng(2)
    m0 = scene.map_xyz[:20000]
    sets = []
    for wait in (False, True):
        t = ikdref.IkdTree(0.3)
        t.build(m0)
        cur = m0.copy()
        r = np.random.default_rng(3)
        for k in range(5):
            new = scan_world(scene, r, 4000, 0.05)
            t.add_points(new, True)
            if wait:
                assert t.wait_rebuild()
            cur, oi = oracle_lib.map_add_points(cur, new, 0.3)
            assert oi.n_ambiguous == 0
        extra = scan_world(scene, r, 500)
        t.add_points(extra, False)
        cur, _ = oracle_lib.map_add_points(cur, extra, 0.0)
        got = sorted_rows(t.flatten())
        assert np.array_equal(got, sorted_rows(cur))
        sets.append(got)
        t.close()
    assert np.array_equal(sets[0], sets[1])


def test_exact_ties_in_a_box(oracle_lib):
    """the cases of tests/test_map_gpu.py::test_ties_latest_new_point_wins...: the tree agrees with the per-box rule."""
    ds = 0.5
    old = np.array([[0.125, 0.25, 0.25], [5.1, 5.1, 5.1], [5.2, 5.2, 5.2]], dtype=np.float32)
    new = np.array([[0.375, 0.25, 0.25], [0.25, 0.125, 0.25], [0.25, 0.375, 0.25], [5.4, 5.4, 5.4], [9.0, 9.0, 9.0]], dtype=np.float32)
    t = ikdref.IkdTree(ds)
    t.build(old)
    t.add_points(new, True)
    want, _ = oracle_lib.map_add_points(old, new, ds)
    assert np.array_equal(sorted_rows(t.flatten()), sorted_rows(want))
    assert np.array_equal(sorted_rows(want), sorted_rows(np.stack([old[2], new[2], new[4]])))
    t.close()


def test_membership_ambiguity_is_where_the_tree_and_the_floor_partition_part(oracle_lib):
    """ds = 0.3 and coordinates on multiples of 1/8: floor(v/ds)*ds can land on the other side of v.  The sequential restatement tests
    coordinates against the float box bounds like the tree does, so it must still agree with the tree; it also counts the points for
    which the device's integer partition is NOT that test (n_ambiguous) -- the number the product reports in fl_map_info."""
    rng = np.random.default_rng(1)
    m0 = (np.round(rng.uniform(-5, 5, (2000, 3)) * 8) / 8).astype(np.float32)
    m0 = np.unique(m0, axis=0)
    new = (np.round(rng.uniform(-5, 5, (2000, 3)) * 8) / 8).astype(np.float32)
    t = ikdref.IkdTree(0.3)
    t.build(m0)
    t.add_points(new, True)
    want, oi = oracle_lib.map_add_points(m0, new, 0.3)
    assert oi.n_ambiguous > 0
    assert np.array_equal(sorted_rows(t.flatten()), sorted_rows(want))
    t.close()


def test_delete_boxes_restatement_equals_the_tree(oracle_lib, scene):
    m0 = scene.map_xyz.copy()
    lo, hi = m0.min(0), m0.max(0)
    boxes = np.array([[lo[0], lo[1], lo[2], lo[0] + 3.0, hi[1] + 1, hi[2] + 1],
                      [hi[0] - 2.0, lo[1], lo[2], hi[0] + 1, hi[1] + 1, hi[2] + 1],
                      [100, 100, 100, 101, 101, 101]], dtype=np.float32)
    boxes[0, 3] = m0[17, 0]          # upper bound ON a coordinate: `max > v` is strict, the point stays (ikd_Tree.cpp:650)
    boxes[1, 0] = m0[23, 0]          # lower bound on a coordinate: `min <= v`, the point goes
    t = ikdref.IkdTree(0.5)
    t.build(m0)
    removed = t.delete_boxes(boxes)
    want, oi = oracle_lib.map_delete_boxes(m0, boxes)
    assert removed == oi.n_removed > 0
    assert np.array_equal(sorted_rows(t.flatten()), sorted_rows(want))
    # the window of lasermap_fov_segment walking along +x
    win = np.zeros(6, dtype=np.float32)
    init = False
    cur = want
    moved = 0
    for step in range(40):
        bx, init = oracle_lib.fov_segment(win, init, np.array([step * 1.0, 0.0, 0.0]), cube_len=40.0, det_range=10.0, mov_threshold=1.5)
        if len(bx):
            moved += 1
            r = t.delete_boxes(bx)
            cur, oi = oracle_lib.map_delete_boxes(cur, bx)
            assert r == oi.n_removed
            assert np.array_equal(sorted_rows(t.flatten()), sorted_rows(cur))
    assert moved >= 1
    # search after deletions: deleted points must not come back as neighbours
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(2000, scene=scene)
    q = fr.world_at(fr.R_prior, fr.p_prior).astype(np.float32)
    xyz, sq, found = t.nearest(q)
    nbr_o, sq_o, _, _ = oracle_lib.knn5_bruteforce(cur, q)
    full = found == 5
    assert np.array_equal(sq[full], sq_o[full]) and np.array_equal(xyz[full], nbr_o[full])
    t.close()


def test_threaded_search_is_the_same_search(scene):
    """ikdref_nearest_mt = KD_TREE::Nearest_Search under `#pragma omp parallel for`, the way laserMapping.cpp:1516-1519 calls it (what
    bench.py's cpu_frame times): same neighbours, same order, same distances as the sequential calls."""
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(6000, scene=scene)
    q = fr.world_at(fr.R_prior, fr.p_prior).astype(np.float32)
    t = ikdref.IkdTree(0.5)
    t.build(scene.map_xyz)
    a = t.nearest(q)
    b = t.nearest(q, nthreads=4)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    t.close()


def test_bench_cpu_frame_section_runs(scene):
    import bench
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(3000, scene=scene)
    vf = synth.make_vio_frame(100, fr)
    r = bench.section_cpu_frame(synth, scene, fr, vf, 0.5)
    assert r["reference_threading"]["threads"] == 4 and r["reference_threading"]["frame_ms"] > 0
    assert r["reference_threading"]["lio_passes"] >= 2 and r["best_over_thread_counts"]["frame_ms"] > 0
