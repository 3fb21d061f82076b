"""GPU parity of the device map maintenance (the map side): fl_map_add_points / fl_map_delete_boxes against
oracle/orc_map.c, which replays KD_TREE::Add_Points(points, downsample) / Delete_Point_Boxes one point after the other
(ikd_Tree.cpp:382-457, :501-520) on a flat array. Bit-exact: same points, same order."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _handle(capi, synth, fr, max_iter=10):
    return capi.Handle(capi.config_from_frames(fr, max_iterations=max_iter))


def _scan_world(scene, rng, n, jitter=0.02):
    """points near the map's surfaces, like a registered scan"""
    base = scene.map_xyz[rng.integers(0, len(scene.map_xyz), n)]
    return (base + rng.normal(0, jitter, (n, 3))).astype(np.float32)


@pytest.mark.parametrize("n_map,n_new,ds", [(20000, 3000, 0.5), (20000, 3000, 0.3), (5000, 6000, 0.15), (1, 1, 0.5), (30000, 1, 0.25)])
def test_add_points_matches_sequential_oracle(gpu_lib, oracle_lib, scene, n_map, n_new, ds):
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    rng = np.random.default_rng(7 + n_map + n_new)
    fr = synth.make_lio_frame(100, scene=scene)
    h = _handle(capi, synth, fr)
    m0 = scene.map_xyz[rng.choice(len(scene.map_xyz), n_map, replace=False)]
    new = _scan_world(scene, rng, n_new)
    h.map_set_points(m0, 0.5)
    info = h.map_add_points(new, ds)
    got = h.map_get_points()
    want, oi = orc.map_add_points(m0, new, ds)
    assert oi.n_ambiguous == info.n_ambiguous
    if oi.n_ambiguous == 0:
        assert (info.n_before, info.n_after, info.n_added, info.n_removed) == (oi.n_before, oi.n_after, oi.n_added, oi.n_removed)
        assert np.array_equal(got, want)
    assert info.status == 0
    # every touched box now holds exactly one point
    key = np.floor(got / np.float32(ds)).astype(np.int64)
    tk = np.unique(np.floor(new / np.float32(ds)).astype(np.int64), axis=0)
    allk, cnt = np.unique(key, axis=0, return_counts=True)
    d = {tuple(k): c for k, c in zip(allk, cnt)}
    assert all(d.get(tuple(k), 0) == 1 for k in tk)
    h.close()


def test_sequence_of_frames_and_search_on_the_updated_map(gpu_lib, oracle_lib, scene):
    """first frame: Build (no down-sampling) on an empty map; then several map_incremental calls; the k-NN of the maintained
    map equals brute force over the oracle's map."""
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    rng = np.random.default_rng(3)
    fr = synth.make_lio_frame(4000, scene=scene)
    h = _handle(capi, synth, fr)
    h.map_clear(0.5)
    first = _scan_world(scene, rng, 8000, 0.05)
    info = h.map_add_points(first, 0.0)                       # ikdtree.Build(feats_down_world): everything goes in
    assert info.n_after == 8000 and info.n_added == 8000
    cur = first.copy()
    for k in range(4):
        new = _scan_world(scene, rng, 5000, 0.05)
        info = h.map_add_points(new, 0.3)
        cur, oi = orc.map_add_points(cur, new, 0.3)
        assert oi.n_ambiguous == 0 == info.n_ambiguous
        assert np.array_equal(h.map_get_points(), cur), f"frame {k}"
    x = capi.state18_from_frame(fr)
    h.lio_set_points(fr.body_xyz); h.lio_begin18(x, x)
    nbr_g, valid_g = h.lio_search18(fr.n)
    world = h.lio_get_world_points(fr.n)
    nbr_o, _, valid_o, _ = orc.knn5_bruteforce(cur, world)
    assert np.array_equal(valid_g, valid_o)
    ok = valid_o != 0
    assert ok.sum() > 100 and np.array_equal(nbr_g[ok], nbr_o[ok])
    h.close()


def test_ties_latest_new_point_wins_and_old_point_needs_strictly_closer(gpu_lib, oracle_lib, scene):
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(100, scene=scene)
    h = _handle(capi, synth, fr)
    ds = 0.5
    # box [0,0.5)^3, centre 0.25: points mirrored at +-0.125 (exact in float) are exactly equally far from the centre
    old = np.array([[0.125, 0.25, 0.25], [5.1, 5.1, 5.1], [5.2, 5.2, 5.2]], dtype=np.float32)     # + one box holding 2 old points
    new = np.array([[0.375, 0.25, 0.25],      # ties with old[0] -> the new point wins (an old point needs strictly <)
                    [0.25, 0.125, 0.25],      # same distance again -> the later one wins
                    [0.25, 0.375, 0.25],
                    [5.4, 5.4, 5.4],          # touches the 2-point box: reduced to the single closest, old (5.2,..) (centre 5.25)
                    [9.0, 9.0, 9.0]], dtype=np.float32)
    h.map_set_points(old, 0.5)
    info = h.map_add_points(new, ds)
    got = h.map_get_points()
    want, oi = orc.map_add_points(old, new, ds)
    assert np.array_equal(want, np.stack([old[2], new[2], new[4]]))
    assert np.array_equal(got, want)
    assert info.n_after == oi.n_after == 3 and info.n_removed == 2 and info.n_added == 2
    # adding the same scan again changes nothing in the set (every point ties with itself; the later copy replaces the earlier)
    info2 = h.map_add_points(new, ds)
    assert sorted(map(tuple, h.map_get_points().tolist())) == sorted(map(tuple, got.tolist()))
    h.close()


def test_delete_boxes_and_fov_window(gpu_lib, oracle_lib, scene):
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    rng = np.random.default_rng(9)
    fr = synth.make_lio_frame(100, scene=scene)
    h = _handle(capi, synth, fr)
    m0 = scene.map_xyz.copy()
    h.map_set_points(m0, 0.5)
    lo, hi = m0.min(0), m0.max(0)
    boxes = np.array([[lo[0], lo[1], lo[2], lo[0] + 3.0, hi[1] + 1, hi[2] + 1],
                      [hi[0] - 2.0, lo[1], lo[2], hi[0] + 1, hi[1] + 1, hi[2] + 1],
                      [100, 100, 100, 101, 101, 101]], dtype=np.float32)
    boxes[0, 3] = m0[17, 0]          # an upper bound ON a point's coordinate: max > v is strict, the point stays
    boxes[1, 0] = m0[23, 0]          # a lower bound on a coordinate: min <= v, the point goes
    info = h.map_delete_boxes(boxes)
    want, oi = orc.map_delete_boxes(m0, boxes)
    assert (info.n_after, info.n_removed) == (oi.n_after, oi.n_removed) and oi.n_removed > 0
    assert np.array_equal(h.map_get_points(), want)
    # window logic of lasermap_fov_segment driving the deletion: LiDAR walking along +x until the window moves
    win = np.zeros(6, dtype=np.float32)
    init = False
    cur = want
    moved = 0
    for step in range(40):
        pos = np.array([step * 1.0, 0.0, 0.0])
        bx, init = orc.fov_segment(win, init, pos, cube_len=40.0, det_range=10.0, mov_threshold=1.5)
        if len(bx):
            moved += 1
            info = h.map_delete_boxes(bx)
            cur, oi = orc.map_delete_boxes(cur, bx)
            assert np.array_equal(h.map_get_points(), cur)
    assert moved >= 1
    # deleting everything leaves an empty map that refuses searches but accepts points again
    h.map_delete_boxes(np.array([[-1e6, -1e6, -1e6, 1e6, 1e6, 1e6]], dtype=np.float32))
    assert len(h.map_get_points()) == 0
    h.map_add_points(m0[:100], 0.5)
    assert len(h.map_get_points()) > 0
    h.close()


def test_resident_scan_goes_into_the_map_under_the_updated_state(gpu_lib, oracle_lib, scene):
    """fl_lio_frame18_dev then fl_map_add_points(NULL): map_incremental's pointBodyToWorld under the frame's final state."""
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(6000, scene=scene)
    h = _handle(capi, synth, fr, 4)
    h.map_set_points(scene.map_xyz, 0.5)
    x = capi.state18_from_frame(fr)
    h.lio_frame18_dev(x, fr.body_xyz)
    world = h.lio_get_world_points(fr.n)                     # under the final state
    info = h.map_add_points(None, 0.25)
    want, oi = orc.map_add_points(scene.map_xyz, world, 0.25)
    if oi.n_ambiguous == 0:
        assert np.array_equal(h.map_get_points(), want)
    assert info.n_after == oi.n_after
    h.close()


def test_growth_beyond_capacity_keeps_the_map(gpu_lib, oracle_lib, scene):
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    rng = np.random.default_rng(21)
    fr = synth.make_lio_frame(100, scene=scene)
    h = _handle(capi, synth, fr)
    m0 = rng.uniform(-20, 20, (60000, 3)).astype(np.float32)           # just below the initial 65536 capacity
    new = rng.uniform(-20, 20, (30000, 3)).astype(np.float32)
    h.map_set_points(m0, 0.5)
    info = h.map_add_points(new, 0.0)
    assert info.n_after == 90000
    got = h.map_get_points()
    assert np.array_equal(got[:60000], m0) and np.array_equal(got[60000:], new)
    h.close()


def test_automatic_cell_size_follows_the_density_and_changes_no_result(gpu_lib, oracle_lib, scene):
    """cell_size <= 0: the k-NN cell edge is chosen from the map's density and re-chosen when map_incremental thins the map.
    The search is exact for any cell size: neighbour sets stay identical to brute force before and after."""
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    rng = np.random.default_rng(5)
    fr = synth.make_lio_frame(3000, scene=scene)
    h = _handle(capi, synth, fr)
    x = capi.state18_from_frame(fr)
    h.map_set_points(scene.map_xyz, 0.0)                      # automatic
    h.lio_set_points(fr.body_xyz); h.lio_begin18(x, x)

    def check(map_now):
        nbr_g, valid_g = h.lio_search18(fr.n)
        world = h.lio_get_world_points(fr.n)
        nbr_o, _, valid_o, _ = orc.knn5_bruteforce(map_now, world)
        assert np.array_equal(valid_g, valid_o)
        ok = valid_o != 0
        assert np.array_equal(nbr_g[ok], nbr_o[ok])
    check(scene.map_xyz)
    # thin the map to one point per 0.6 m box: every box of the scene is touched by a dense "scan" of the map's own points
    dense = (scene.map_xyz + rng.normal(0, 0.01, scene.map_xyz.shape)).astype(np.float32)
    i1 = h.map_add_points(dense, 0.6)
    cur, oi = orc.map_add_points(scene.map_xyz, dense, 0.6)
    assert oi.n_ambiguous == info_amb(i1) and np.array_equal(h.map_get_points(), cur)
    i2 = h.map_add_points(dense[:10], 0.6)                    # the density of the thinned map is known by now: the cell grows
    cur, _ = orc.map_add_points(cur, dense[:10], 0.6)
    assert i2.cell_size > i1.cell_size >= 0.3
    assert len(cur) < len(scene.map_xyz) // 3
    check(cur)
    h.close()


def info_amb(i):
    return i.n_ambiguous


def test_resident_scan_under_the_ikfom_state(gpu_lib, oracle_lib, scene):
    """Mode-23: after fl_ikfom_update_iterated_dev, fl_map_add_points(NULL) registers the scan under the updated state_ikfom."""
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(5000, scene=scene)
    h = _handle(capi, synth, fr, 3)
    h.map_set_points(scene.map_xyz, 0.5)
    x = capi.state23_from_frame(fr)
    P = fr.cov23.copy()
    h.ikfom_update_iterated_dev(x, P, fr.body_xyz, 0.001)
    world = h.ikfom_world_points(x, fr.n)                    # pointBodyToWorld under the updated state (the same kernel the search uses)
    info = h.map_add_points(None, 0.25)
    want, oi = orc.map_add_points(scene.map_xyz, world, 0.25)
    assert info.n_after == oi.n_after
    if oi.n_ambiguous == 0:
        assert np.array_equal(h.map_get_points(), want)
    h.close()


def _search(h, capi, fr):
    x = capi.state18_from_frame(fr)
    h.lio_set_points(fr.body_xyz); h.lio_begin18(x, x)
    nbr, valid = h.lio_search18(fr.n)
    return nbr, valid, h.lio_get_world_points(fr.n)


def test_in_place_updates_keep_the_index_exact_between_read_backs(gpu_lib, oracle_lib, scene):
    """FL_OPT_MAP_INCREMENTAL (default): 12 updates in a row -- map_incremental with and without down-sampling, slab deletions, scans that
    open new cells far from the map, cells that outgrow their slack -- with NO read-back of the map in between (fl_map_get_points would
    compact and re-index). After every update the device search over the in-place index must equal brute force over the oracle's map,
    neighbour for neighbour; at the end the compacted array equals the oracle's, order included."""
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    rng = np.random.default_rng(41)
    fr = synth.make_lio_frame(3000, scene=scene)
    h = _handle(capi, synth, fr)
    cur = scene.map_xyz[rng.choice(len(scene.map_xyz), 15000, replace=False)].copy()
    h.map_set_points(cur, 0.5)
    lo, hi = cur.min(0), cur.max(0)
    for k in range(12):
        if k % 4 == 3:                                               # lasermap_fov_segment: a slab leaves
            a = lo[0] + rng.uniform(0, 10)
            boxes = np.array([[a, lo[1] - 1, lo[2] - 1, a + 1.5, hi[1] + 1, hi[2] + 1]], dtype=np.float32)
            h.map_delete_boxes(boxes, want_info=False)
            cur, _ = orc.map_delete_boxes(cur, boxes)
        else:
            new = _scan_world(scene, rng, 4000, 0.05)
            if k == 5:
                new = np.concatenate([new, (rng.uniform(40, 44, (500, 3))).astype(np.float32)])      # cells the table has never seen
            if k == 6:
                new = np.concatenate([new, (np.float32([1.0, 1.0, 0.0]) + rng.normal(0, 0.1, (3000, 3))).astype(np.float32)])   # one spot, many points: cells move
            ds = (0.3, 0.2, 0.0)[k % 3]
            h.map_add_points(new, ds, want_info=False)
            cur, oi = orc.map_add_points(cur, new, ds)
            assert oi.n_ambiguous == 0
        nbr_g, valid_g, world = _search(h, capi, fr)
        nbr_o, _, valid_o, _ = orc.knn5_bruteforce(cur, world)
        assert np.array_equal(valid_g, valid_o), f"update {k}"
        ok = valid_o != 0
        assert np.array_equal(nbr_g[ok], nbr_o[ok]), f"update {k}"
    assert np.array_equal(h.map_get_points(), cur)
    h.close()


def test_in_place_and_rebuilding_forms_agree(gpu_lib, scene):
    """fl_set_option(FL_OPT_MAP_INCREMENTAL, 0 / 1): same counters, same map, same neighbours over a short sequence"""
    capi = gpu_lib
    from fast_livo_amd import synth
    fr = synth.make_lio_frame(2000, scene=scene)
    outs = []
    for incr in (0, 1):
        rng = np.random.default_rng(5)
        h = _handle(capi, synth, fr)
        h.set_option(capi.FL_OPT_MAP_INCREMENTAL, incr)
        h.map_set_points(scene.map_xyz[::3], 0.0)                   # automatic cell size
        rec = []
        for k in range(6):
            new = _scan_world(scene, rng, 5000, 0.04)
            i = h.map_add_points(new, 0.25)
            rec.append((i.n_before, i.n_after, i.n_added, i.n_removed, i.n_ambiguous))
            if k == 3:
                lo = scene.map_xyz.min(0)
                i = h.map_delete_boxes(np.array([[lo[0], lo[1], lo[2], lo[0] + 4, lo[1] + 30, lo[2] + 10]], dtype=np.float32))
                rec.append((i.n_before, i.n_after, i.n_removed))
        nbr, valid, _ = _search(h, capi, fr)
        outs.append((rec, h.map_get_points(), nbr, valid))
        h.close()
    assert outs[0][0] == outs[1][0]
    assert np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][3], outs[1][3]) and np.array_equal(outs[0][2][outs[0][3] != 0], outs[1][2][outs[1][3] != 0])


def test_many_in_place_updates_run_into_the_full_rebuild(gpu_lib, oracle_lib, scene):
    """60 updates of 6 000 points on a 10 k-point map: the raw array (appended to by every update) and the pool fill up and the lazy
    status check compacts / re-indexes in between -- nothing of that may show in the map or in the search"""
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    rng = np.random.default_rng(77)
    fr = synth.make_lio_frame(2000, scene=scene)
    h = _handle(capi, synth, fr)
    cur = scene.map_xyz[rng.choice(len(scene.map_xyz), 10000, replace=False)].copy()
    h.map_set_points(cur, 0.5)
    for k in range(60):
        new = _scan_world(scene, rng, 6000, 0.05)
        h.map_add_points(new, 0.3, want_info=False)
        cur, _ = orc.map_add_points(cur, new, 0.3)
        if k % 20 == 19:
            nbr_g, valid_g, world = _search(h, capi, fr)
            nbr_o, _, valid_o, _ = orc.knn5_bruteforce(cur, world)
            ok = valid_o != 0
            assert np.array_equal(valid_g, valid_o) and np.array_equal(nbr_g[ok], nbr_o[ok]), f"update {k}"
    assert np.array_equal(h.map_get_points(), cur)
    h.close()


def test_search_after_an_update_that_ran_out_of_pool_sees_every_point(gpu_lib, oracle_lib, scene):
    """an in-place update that runs out of pool room drops its queued points from the INDEX and raises needs_rebuild; the
    next search must read that status (map_index_ready) and re-index first. The pool is starved through the debug library's
    fl_debug_map_pool_limit (device-side only, so the host's pre-check does not see it coming); no fl_map_* call between the update
    and the search."""
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    rng = np.random.default_rng(91)
    fr = synth.make_lio_frame(2000, scene=scene)
    h = capi.Handle(capi.config_from_frames(fr, max_iterations=10), debug=True)
    cur = scene.map_xyz[rng.choice(len(scene.map_xyz), 20000, replace=False)].copy()
    h.map_set_points(cur, 0.5)
    h.debug_map_pool_limit(16)
    # dense: many points per touched cell, so cells outgrow their slack and want to move to the (exhausted) pool top
    world0 = fr.world_at(fr.R_prior, fr.p_prior).astype(np.float32)
    new = (world0[rng.integers(0, len(world0), 8000)] + rng.normal(0, 0.01, (8000, 3))).astype(np.float32)
    h.map_add_points(new, 0.0, want_info=False)
    cur, _ = orc.map_add_points(cur, new, 0.0)
    nbr_g, valid_g, world = _search(h, capi, fr)
    nbr_o, _, valid_o, _ = orc.knn5_bruteforce(cur, world)
    ok = valid_o != 0
    assert np.array_equal(valid_g, valid_o) and np.array_equal(nbr_g[ok], nbr_o[ok])
    assert np.array_equal(h.map_get_points(), cur)
    h.close()


def test_map_compact_on_demand_changes_nothing(gpu_lib, oracle_lib, scene):
    """fl_map_compact: the O(map) compaction + re-index of the in-place form, run when the caller chooses. Map content, order and search results
    are those of the oracle before and after; a second call (nothing dead) is a no-op."""
    capi, orc = gpu_lib, oracle_lib
    from fast_livo_amd import synth
    rng = np.random.default_rng(123)
    fr = synth.make_lio_frame(2000, scene=scene)
    h = _handle(capi, synth, fr)
    cur = scene.map_xyz[rng.choice(len(scene.map_xyz), 30000, replace=False)].copy()
    h.map_set_points(cur, 0.5)
    for k in range(3):
        new = _scan_world(scene, rng, 5000, 0.04)
        h.map_add_points(new, 0.3, want_info=False)
        cur, _ = orc.map_add_points(cur, new, 0.3)
    nbr0, valid0, world = _search(h, capi, fr)
    h.map_compact()
    nbr1, valid1, _ = _search(h, capi, fr)
    h.map_compact()
    nbr_o, _, valid_o, _ = orc.knn5_bruteforce(cur, world)
    ok = valid_o != 0
    assert np.array_equal(valid0, valid_o) and np.array_equal(valid1, valid_o)
    assert np.array_equal(nbr0[ok], nbr_o[ok]) and np.array_equal(nbr1[ok], nbr_o[ok])
    assert np.array_equal(h.map_get_points(), cur)
    new = _scan_world(scene, rng, 5000, 0.04)
    h.map_add_points(new, 0.3, want_info=False)          # in place again on the compacted arrays
    cur, _ = orc.map_add_points(cur, new, 0.3)
    assert np.array_equal(h.map_get_points(), cur)
    h.close()
