// mapinc_kernels.h -- the device map updated IN PLACE (round 5; SURVEY.md 8f rows N1/N3, the reference's side of it:
// KD_TREE::Add_Points, include/ikd-Tree/ikd_Tree.cpp:382-457, is O(new points x log map) and laserMapping.cpp:1758 calls it every frame).
//
// Until round 4 an update compacted the whole map array and rebuilt the whole k-NN index (mapupd_kernels.h + knn_kernels.h): fine at
// 55 k - 200 k points, O(map) at the 1-5 M points a cube_side_length-1000 map holds, and one host synchronisation per update. Here an
// update touches only the cells of the new and of the deleted points and never waits for the host:
//   * the map array (positions by index: the k-NN's tie rule and its neighbour fetch) is APPEND-ONLY between compactions: the new
//     points of an update take the indices n_raw .. n_raw + n - 1 in input order (losers of the down-sampling included, flagged dead),
//     deleted points are flagged dead and keep their slot -- the relative order of the live points, all the tie rule looks at, is
//     what the compacting form produces;
//   * the index keeps every cell's points contiguous in a pool with SLACK behind them (capacity > count); a cell that outgrows its
//     capacity moves to a fresh, larger region at the pool's end (bump allocation; the hole is reclaimed by the next full rebuild).
//     The search kernel is unchanged: it still sees (start, count) per cell;
//   * what only a full rebuild can do (raw array or pool nearly full, table load, cell size out of tune) is decided by the host from a
//     status block the update's last kernel leaves in page-locked memory -- read lazily, before the NEXT use of the map.
// Per-box rule of Add_Points: see mapupd_kernels.h (the arithmetic and the tie rules are shared with it).
//
//   mapupd_new_kernel (mapupd_kernels.h)  every new point claims its down-sampling box and competes for "best new point of the box"
//   mapinc_resolve_kernel   one thread per claimed box: the box's old points are found through the k-NN grid (the <= 8 cells the box
//                           overlaps when ds <= cell), best old vs best new decided, the losers among the old points tombstoned in the
//                           pool and flagged dead, the winning new point queued at its cell
//   mapinc_apply_kernel     one thread per (box, overlapped cell), the first to arrive owns the cell: tombstones squeezed out, queued
//                           points appended, the cell moved if it no longer fits
//   mapinc_append_*         the same without down-sampling (Add_Points(points, false)): every new point is queued at its cell
//   mapinc_delete_kernel    Delete_Point_Boxes: one thread per table slot, cells that intersect a box are squeezed in place
//   mapinc_status_kernel    counters -> the page-locked status block
#pragma once

#include "mapupd_kernels.h"

#define FL_MI_DEAD 0x7ffffffe          /* id of a tombstoned pool entry (never a map index; FL_KNN_NO_ID is 0x7fffffff) */
#define FL_MI_NONE 0xFFFFFFFFu

struct FlMapIncCtl {
    unsigned pool_top, pool_cap;       // bump pointer / size of the pool (float4 entries)
    unsigned ncells, slots;            // occupied / all table slots
    int live;                          // live map points
    int needs_rebuild;                 // pool or table exhausted in an update: the index is INCOMPLETE until the next full rebuild
    int added, removed, ambiguous, range_error;      // of the update in flight
    unsigned long long seq;            // of the update whose status this is (status block only)
};

// (fl_mi_cap / FlCellCap: knn_kernels.h -- the capacity a cell gets for its points)

struct FlMapIncView {                  // what the update kernels need of the index (all writable)
    float4 *pts;
    FlCellEntry *htab;
    unsigned hmask;
    unsigned long long *ckeys;
    unsigned cmask;
    unsigned *cellcap;                 // per table slot: capacity of the cell's region
    unsigned *dirty;                   // per table slot: the cell holds tombstones and / or queued points
    unsigned *pend_head;               // per table slot: first queued new point (FL_MI_NONE: none)
    float inv_cell;
};

// The update's counters (added / removed / ambiguous): n per lane, ONE atomic per WORKGROUP, striped over FL_MI_STRIPES addresses in
// separate 128-byte lines behind the control block. (Round 6: they were one atomic per wavefront on one address each -- 2 048 wavefronts of
// mapinc_resolve_kernel x up to three counters at ~12 ns per same-address atomic were more than half of that kernel's 47-60 us.)
// mapinc_begin_kernel zeroes the stripes, mapinc_status_kernel folds them into the control block. All threads of the workgroup call it.
#define FL_MI_STRIPES 16
#define FL_MI_CNT_ADDED 0
#define FL_MI_CNT_REMOVED 1
#define FL_MI_CNT_AMBIGUOUS 2
struct FlMapIncStripes { int v[3][FL_MI_STRIPES][32]; };
__device__ __forceinline__ FlMapIncStripes *fl_mi_stripes(FlMapIncCtl *ctl) { return reinterpret_cast<FlMapIncStripes *>(ctl + 1); }
__device__ __forceinline__ void fl_mi_count_block(int n, int which, FlMapIncCtl *ctl)
{
    __shared__ int s_cnt[3][FL_BLOCK / 64];
    int s = n;
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) s += __shfl_xor(s, k);
    if ((threadIdx.x & 63u) == 0) s_cnt[which][threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
#pragma unroll
        for (int w = 0; w < FL_BLOCK / 64; w++) t += s_cnt[which][w];
        if (t) atomicAdd(&fl_mi_stripes(ctl)->v[which][blockIdx.x % FL_MI_STRIPES][0], t);
    }
}

// read-only probe: table slot of a cell, -1 if the cell does not exist
__device__ __forceinline__ int fl_mi_find(const FlMapIncView &V, unsigned long long key)
{
    unsigned h = fl_hash64(key) & V.hmask;
    for (unsigned probes = 0; probes <= V.hmask; probes++) {
        const unsigned long long k = *(volatile unsigned long long *)&V.htab[h].key;
        if (k == key) return (int)h;
        if (k == FL_KNN_EMPTY) return -1;
        h = (h + 1) & V.hmask;
    }
    return -1;
}
// find or create; -1: the table is full (needs_rebuild). A created cell also enters the coarse occupancy set.
__device__ __forceinline__ int fl_mi_find_or_insert(const FlMapIncView &V, unsigned long long key, FlMapIncCtl *ctl)
{
    unsigned h = fl_hash64(key) & V.hmask;
    for (unsigned probes = 0; probes <= V.hmask; probes++) {
        const unsigned long long prev = atomicCAS((unsigned long long *)&V.htab[h].key, FL_KNN_EMPTY, key);
        if (prev == key) return (int)h;
        if (prev == FL_KNN_EMPTY) {
            atomicAdd(&ctl->ncells, 1u);
            const unsigned long long ck = fl_coarse_key(key);
            unsigned hc = fl_hash64(ck * 0x9E3779B97F4A7C15ull >> 1) & V.cmask;
            for (unsigned p2 = 0; p2 <= V.cmask; p2++) {
                const unsigned long long pc = atomicCAS(&V.ckeys[hc], FL_KNN_EMPTY, ck);
                if (pc == FL_KNN_EMPTY || pc == ck) break;
                hc = (hc + 1) & V.cmask;
            }
            return (int)h;
        }
        h = (h + 1) & V.hmask;
    }
    return -1;
}
__device__ __forceinline__ unsigned long long fl_mi_cell_key_of(float x, float y, float z, float inv_cell)
{
    return fl_cell_key((int)floorf(x * inv_cell), (int)floorf(y * inv_cell), (int)floorf(z * inv_cell));
}

// the cells a down-sampling box [mn, mn + ds) overlaps, per axis [c0, c1] (conservative: the upper face counts)
struct FlBoxCells { int c0[3], c1[3]; float cx, cy, cz; };
__device__ __forceinline__ FlBoxCells fl_mi_box_cells(unsigned long long box_key, float ds, float inv_cell)
{
    FlBoxCells b;
    const int bi[3] = {(int)((box_key >> 42) & 0x1FFFFFu) - (1 << 20), (int)((box_key >> 21) & 0x1FFFFFu) - (1 << 20),
                       (int)(box_key & 0x1FFFFFu) - (1 << 20)};
    float c[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float f = (float)bi[k];
        const float mn = f * ds, mx = mn + ds;                                   // vertex_min / vertex_max (ikd_Tree.cpp:392-393)
        c[k] = (float)((double)mn + (double)(mx - mn) / 2.0);                    // mid_point (:398)
        // the box's members are the floats v with floorf(v / ds) == f (the partition mapupd_kernels.h uses): an interval whose ends lie
        // within an ulp of mn and mx (the division rounds) -- found by probing, so that aligned grids (ds a power of two) get exactly
        // [mn, mx) and no neighbour cell is scanned for nothing
        float lo = mn, hi = nextafterf(mx, -INFINITY);
        if (floorf(nextafterf(mn, -INFINITY) / ds) == f) lo = nextafterf(mn, -INFINITY);
        if (floorf(mx / ds) == f) hi = mx;
        if (floorf(nextafterf(mx, INFINITY) / ds) == f) hi = nextafterf(mx, INFINITY);
        b.c0[k] = (int)floorf(lo * inv_cell);
        b.c1[k] = (int)floorf(hi * inv_cell);
    }
    b.cx = c[0]; b.cy = c[1]; b.cz = c[2];
    return b;
}

// FL_MI_RESOLVE_LANES lanes per slot of the box table, one per overlapped cell (a box of the down-sampling grid overlaps up to 2 x 2 x 2 cells of
// the k-NN grid when ds <= cell; more cells: the lanes stride). Round 6: it was ONE lane per box walking all its cells' points twice, one
// dependent load after the other -- up to 160 sequential visits on the slowest lane of a wavefront, 45-60 us for 20 k new points. Now a
// lane walks ONE cell's points (pass 1: its candidate for the box's closest old point), the eight lanes of a box take the minimum between
// them (three shuffles), and pass 2 marks each lane's own cell. Same candidates, same minimum, same marks: same map.
#define FL_MI_RESOLVE_LANES 8
#define FL_MI_RESOLVE_KEEP 6
__global__ __launch_bounds__(FL_BLOCK) void mapinc_resolve_kernel(FlBoxSlot *__restrict__ tab, unsigned tab_cap, const float *__restrict__ new_pts,
                                                                 float ds, FlMapIncView V, unsigned char *__restrict__ dead, int n_raw,
                                                                 unsigned *__restrict__ pend_next, FlMapIncCtl *__restrict__ ctl)
{
    const unsigned t = blockIdx.x * FL_BLOCK + threadIdx.x;
    const unsigned s = t / FL_MI_RESOLVE_LANES, sub = t % FL_MI_RESOLVE_LANES;
    int removed = 0, added = 0, amb_old = 0;
    // is box `k` claimed by a new point of this update? (open addressing, as mapupd_new_kernel filled it; tab_cap is a power of two)
    auto touched = [&](unsigned long long k) -> bool {
        unsigned hh = fl_hash64(k) & (tab_cap - 1u);
        while (true) {
            const unsigned long long kk = tab[hh].key;
            if (kk == k) return true;
            if (kk == FL_KNN_EMPTY) return false;
            hh = (hh + 1u) & (tab_cap - 1u);
        }
    };
    const unsigned long long bkey = (s < tab_cap) ? tab[s].key : FL_KNN_EMPTY;
    const bool live = bkey != FL_KNN_EMPTY;                     // (uniform over the box's lanes)
    FlBoxCells B;
    int nx = 0, ny = 0, ncell = 0;
    if (live) {
        B = fl_mi_box_cells(bkey, ds, V.inv_cell);
        nx = B.c1[0] - B.c0[0] + 1; ny = B.c1[1] - B.c0[1] + 1;
        ncell = nx * ny * (B.c1[2] - B.c0[2] + 1);
    }
    // pass 1: this lane's cells' candidate for the box's closest old point (lowest index among equals). The box's points it meets on the way
    // are remembered (pool slot + map index, up to FL_MI_RESOLVE_KEEP of them -- a 0.3 m box holds a handful), so that pass 2 does not walk
    // the cells again; a lane that meets more falls back to the walk.
    unsigned long long best_old = FL_KNN_EMPTY;
    unsigned keep_j[FL_MI_RESOLVE_KEEP];
    int keep_id[FL_MI_RESOLVE_KEEP], keep_hs[FL_MI_RESOLVE_KEEP], nkeep = 0;
    bool overflow = false;
    for (int c = (int)sub; c < ncell; c += FL_MI_RESOLVE_LANES) {
        const int ix = B.c0[0] + c % nx, iy = B.c0[1] + (c / nx) % ny, iz = B.c0[2] + c / (nx * ny);
        const int hs = fl_mi_find(V, fl_cell_key(ix, iy, iz));
        if (hs < 0) continue;
        const unsigned st = V.htab[hs].start, cn = V.htab[hs].count;
        for (unsigned j = st; j < st + cn; j++) {
            const float4 p = V.pts[j];
            if (__float_as_int(p.w) == FL_MI_DEAD) continue;
            const FlBoxGeom g = fl_box_of(p.x, p.y, p.z, ds);
            // n_ambiguous of the update (include/fastlivo_hip.h): the OLD points whose box depends on a rounding and that could
            // matter to it -- those of a touched box (counted by their own box) and those next to a touched box whose own box
            // no new point claimed (the reference's coordinate test could have put them into the neighbour)
            if (g.ambiguous && !g.range_error) {      // (rare: ~1e-7 per coordinate on real data)
                const unsigned long long own = fl_cell_key(g.ix, g.iy, g.iz);
                if (own == bkey) amb_old++;
                else if (!touched(own)) {
                    // counted ONCE: by the touched box with the smallest key among the 26 neighbours of its own box
                    unsigned long long first = FL_KNN_EMPTY;
                    for (int dz = -1; dz <= 1; dz++)
                        for (int dy = -1; dy <= 1; dy++)
                            for (int dx = -1; dx <= 1; dx++) {
                                if (!(dx | dy | dz)) continue;
                                const unsigned long long nk = fl_cell_key(g.ix + dx, g.iy + dy, g.iz + dz);
                                if (nk < first && touched(nk)) first = nk;
                            }
                    if (first == bkey) amb_old++;
                }
            }
            if (g.range_error || fl_cell_key(g.ix, g.iy, g.iz) != bkey) continue;
            const float d = fl_calc_dist(p.x, p.y, p.z, g.cx, g.cy, g.cz);
            const unsigned long long cand = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(unsigned)__float_as_int(p.w);
            best_old = cand < best_old ? cand : best_old;
            if (nkeep < FL_MI_RESOLVE_KEEP) {
#pragma unroll
                for (int q = 0; q < FL_MI_RESOLVE_KEEP; q++)          // (constant indices: a run-time index would put the arrays into scratch)
                    if (q == nkeep) { keep_j[q] = j; keep_id[q] = __float_as_int(p.w); keep_hs[q] = hs; }
                nkeep++;
            } else overflow = true;
        }
    }
    // the box's minimum over its lanes (aligned groups of FL_MI_RESOLVE_LANES inside a wavefront: the xor pattern stays inside the group)
#pragma unroll
    for (int m = 1; m < FL_MI_RESOLVE_LANES; m <<= 1) {
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)best_old, m), hi = (unsigned)__shfl_xor((int)(unsigned)(best_old >> 32), m);
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;
        best_old = o < best_old ? o : best_old;
    }
    const unsigned long long best_new = live ? tab[s].best_new : 0ull;
    const bool old_wins = live && best_old != FL_KNN_EMPTY && (unsigned)(best_old >> 32) < (unsigned)(best_new >> 32);      // strictly closer than every new point
    if (live && sub == 0) tab[s].best_old = best_old;
    // pass 2: every other old point of the box goes
    if (!overflow) {
#pragma unroll
        for (int q = 0; q < FL_MI_RESOLVE_KEEP; q++) {
            if (q >= nkeep) continue;
            if (old_wins && (unsigned)keep_id[q] == (unsigned)best_old) continue;
            V.pts[keep_j[q]].w = __int_as_float(FL_MI_DEAD);      // (a point belongs to exactly one box: nobody else writes it)
            dead[keep_id[q]] = 1;
            V.dirty[keep_hs[q]] = 1u;
            removed++;
        }
    }
    for (int c = (int)sub; overflow && c < ncell; c += FL_MI_RESOLVE_LANES) {
        const int ix = B.c0[0] + c % nx, iy = B.c0[1] + (c / nx) % ny, iz = B.c0[2] + c / (nx * ny);
        const int hs = fl_mi_find(V, fl_cell_key(ix, iy, iz));
        if (hs < 0) continue;
        const unsigned st = V.htab[hs].start, cn = V.htab[hs].count;
        bool dirtied = false;
        for (unsigned j = st; j < st + cn; j++) {
            const float4 p = V.pts[j];
            if (__float_as_int(p.w) == FL_MI_DEAD) continue;        // (a tombstone of this or an earlier update: its coordinates are kept)
            const FlBoxGeom g = fl_box_of(p.x, p.y, p.z, ds);
            if (g.range_error || fl_cell_key(g.ix, g.iy, g.iz) != bkey) continue;
            const int id = __float_as_int(p.w);
            if (old_wins && (unsigned)id == (unsigned)best_old) continue;
            V.pts[j].w = __int_as_float(FL_MI_DEAD);      // (a point belongs to exactly one box: nobody else writes it)
            dead[id] = 1;
            removed++;
            dirtied = true;
        }
        if (dirtied) V.dirty[hs] = 1u;
    }
    if (live && sub == 0 && !old_wins) {         // the closest new point (the latest among equals) enters: queued at its cell
        const unsigned jn = 0xFFFFFFFFu - (unsigned)best_new;
        const float x = new_pts[jn * 3], y = new_pts[jn * 3 + 1], z = new_pts[jn * 3 + 2];
        //  a point that is not finite or whose box index does not fit the key (|p / ds| >= 2^20) is never admitted: it stays
        // dead in the array, the update's status says FL_NUM_NONFINITE (mapupd_new_kernel counted it) -- (int)floorf(NaN) below is undefined
        const bool ok = isfinite(x) && isfinite(y) && isfinite(z) && !fl_box_of(x, y, z, ds).range_error;
        if (ok) {
            // the cell the point is queued at must be one of those mapinc_apply_kernel visits for this box (B.c0 .. B.c1, derived by probing
            // the box's faces within one ulp): if the rounding of p * inv_cell ever puts it outside, the cell would stay dirty with a queue
            // into pend_next, which the next update overwrites. Checked here instead of argued: the point then enters the array only and
            // the index is re-built before anybody searches it (needs_rebuild -> map_index_ready, api_knn.inc).
            const int qx = (int)floorf(x * V.inv_cell), qy = (int)floorf(y * V.inv_cell), qz = (int)floorf(z * V.inv_cell);
            const bool covered = qx >= B.c0[0] && qx <= B.c1[0] && qy >= B.c0[1] && qy <= B.c1[1] && qz >= B.c0[2] && qz <= B.c1[2];
            const int hs = covered ? fl_mi_find_or_insert(V, fl_cell_key(qx, qy, qz), ctl) : -1;
            if (hs < 0) {                       // (not covered, or the table is full: the point lives in the array, the index follows at the rebuild)
                ctl->needs_rebuild = 1;
                dead[n_raw + (int)jn] = 0;
                added = 1;
            } else {
                pend_next[jn] = atomicExch(&V.pend_head[hs], jn);
                V.dirty[hs] = 1u;
                dead[n_raw + (int)jn] = 0;
                added = 1;
            }
        }
    }
    fl_mi_count_block(removed, FL_MI_CNT_REMOVED, ctl);
    fl_mi_count_block(added, FL_MI_CNT_ADDED, ctl);
    fl_mi_count_block(amb_old, FL_MI_CNT_AMBIGUOUS, ctl);
}

// the owner of a dirty cell: tombstones out, queued points in, moved if it no longer fits
__device__ __forceinline__ void fl_mi_apply_cell(const FlMapIncView &V, int hs, const float *__restrict__ new_pts, int n_raw,
                                                 const unsigned *__restrict__ pend_next, FlMapIncCtl *__restrict__ ctl)
{
    unsigned st = V.htab[hs].start;
    const unsigned cn = V.htab[hs].count, cap = V.cellcap[hs];
    unsigned j = 0;
    for (unsigned i = 0; i < cn; i++) {
        const float4 p = V.pts[st + i];
        if (__float_as_int(p.w) == FL_MI_DEAD) continue;
        if (i != j) V.pts[st + j] = p;
        j++;
    }
    unsigned k = 0;
    for (unsigned q = V.pend_head[hs]; q != FL_MI_NONE; q = pend_next[q]) k++;
    unsigned newcap = cap;
    if (j + k > cap) {
        newcap = fl_mi_cap(j + k);
        const unsigned pos = atomicAdd(&ctl->pool_top, newcap);
        if (pos + newcap > ctl->pool_cap) {              // the pool is full: the queued points cannot enter -- the index is incomplete until the rebuild
            ctl->needs_rebuild = 1;
            V.htab[hs].count = j;
            V.pend_head[hs] = FL_MI_NONE;
            return;
        }
        for (unsigned i = 0; i < j; i++) V.pts[pos + i] = V.pts[st + i];
        st = pos;
    }
    for (unsigned q = V.pend_head[hs]; q != FL_MI_NONE; q = pend_next[q])
        V.pts[st + j++] = make_float4(new_pts[q * 3], new_pts[q * 3 + 1], new_pts[q * 3 + 2], __int_as_float(n_raw + (int)q));
    V.pend_head[hs] = FL_MI_NONE;
    V.cellcap[hs] = newcap;
    // (start and count in one 8-byte store: the search kernels of LATER launches read the entry whole)
    *reinterpret_cast<uint2 *>(&V.htab[hs].start) = make_uint2(st, j);
}

// one thread per (box, overlapped cell): up to FL_MI_CELLS_PER_BOX cells per box are covered by threads, the rest by a loop
#define FL_MI_CELLS_PER_BOX 8
__global__ __launch_bounds__(FL_BLOCK) void mapinc_apply_kernel(const FlBoxSlot *__restrict__ tab, unsigned tab_cap, const float *__restrict__ new_pts,
                                                               float ds, FlMapIncView V, int n_raw, const unsigned *__restrict__ pend_next,
                                                               FlMapIncCtl *__restrict__ ctl)
{
    const unsigned t = blockIdx.x * FL_BLOCK + threadIdx.x;
    const unsigned s = t / FL_MI_CELLS_PER_BOX, sub = t % FL_MI_CELLS_PER_BOX;
    if (s >= tab_cap) return;
    const unsigned long long bkey = tab[s].key;
    if (bkey == FL_KNN_EMPTY) return;
    const FlBoxCells B = fl_mi_box_cells(bkey, ds, V.inv_cell);
    const int nx = B.c1[0] - B.c0[0] + 1, ny = B.c1[1] - B.c0[1] + 1, nz = B.c1[2] - B.c0[2] + 1;
    const int ncell = nx * ny * nz;
    for (int c = (int)sub; c < ncell; c += FL_MI_CELLS_PER_BOX) {
        const int ix = B.c0[0] + c % nx, iy = B.c0[1] + (c / nx) % ny, iz = B.c0[2] + c / (nx * ny);
        const int hs = fl_mi_find(V, fl_cell_key(ix, iy, iz));
        if (hs < 0) continue;
        if (atomicExch(&V.dirty[hs], 0u) != 1u) continue;            // clean, or another thread owns it
        fl_mi_apply_cell(V, hs, new_pts, n_raw, pend_next, ctl);
    }
}

// Add_Points(points, false): no boxes -- every new point is queued at its cell ...
__global__ __launch_bounds__(FL_BLOCK) void mapinc_append_queue_kernel(const float *__restrict__ new_pts, int n, FlMapIncView V,
                                                                      unsigned *__restrict__ pend_next, int *__restrict__ slot_of,
                                                                      FlMapIncCtl *__restrict__ ctl)
{
    const int j = blockIdx.x * FL_BLOCK + threadIdx.x;
    int added = 0;
    if (j < n) {
        const int hs = fl_mi_find_or_insert(V, fl_mi_cell_key_of(new_pts[j * 3], new_pts[j * 3 + 1], new_pts[j * 3 + 2], V.inv_cell), ctl);
        slot_of[j] = hs;
        if (hs < 0) ctl->needs_rebuild = 1;
        else {
            pend_next[j] = atomicExch(&V.pend_head[hs], (unsigned)j);
            V.dirty[hs] = 1u;
            added = 1;
        }
    }
    fl_mi_count_block(added, FL_MI_CNT_ADDED, ctl);
}
// ... and the first of a cell's new points to arrive applies the queue
__global__ __launch_bounds__(FL_BLOCK) void mapinc_append_apply_kernel(const float *__restrict__ new_pts, int n, FlMapIncView V, int n_raw,
                                                                      const unsigned *__restrict__ pend_next, const int *__restrict__ slot_of,
                                                                      FlMapIncCtl *__restrict__ ctl)
{
    const int j = blockIdx.x * FL_BLOCK + threadIdx.x;
    if (j >= n) return;
    const int hs = slot_of[j];
    if (hs < 0 || atomicExch(&V.dirty[hs], 0u) != 1u) return;
    fl_mi_apply_cell(V, hs, new_pts, n_raw, pend_next, ctl);
}

// Delete_Point_Boxes (ikd_Tree.cpp:501-520; box test of Delete_by_range :626-650: min <= v && max > v on every axis): one thread per
// table slot; a cell whose cube meets a box is squeezed in place
__global__ __launch_bounds__(FL_BLOCK) void mapinc_delete_kernel(FlMapIncView V, unsigned slots, float cell, const float *__restrict__ boxes, int nb,
                                                                unsigned char *__restrict__ dead, FlMapIncCtl *__restrict__ ctl)
{
    const unsigned hs = blockIdx.x * FL_BLOCK + threadIdx.x;
    int removed = 0;
    if (hs < slots) {
        const unsigned long long key = V.htab[hs].key;
        const unsigned cn = key != FL_KNN_EMPTY ? V.htab[hs].count : 0u;
        if (cn) {
            const float lo[3] = {(float)((int)((key >> 42) & 0x1FFFFFu) - (1 << 20)) * cell, (float)((int)((key >> 21) & 0x1FFFFFu) - (1 << 20)) * cell,
                                 (float)((int)(key & 0x1FFFFFu) - (1 << 20)) * cell};
            bool meets = false;
            const float m = 1e-3f * cell + 1e-4f;             // (the cell assignment is floorf(x * inv_cell): margin for its rounding)
            for (int b = 0; b < nb && !meets; b++) {
                const float *B = boxes + b * 6;
                meets = B[0] <= lo[0] + cell + m && B[3] > lo[0] - m && B[1] <= lo[1] + cell + m && B[4] > lo[1] - m && B[2] <= lo[2] + cell + m &&
                        B[5] > lo[2] - m;
            }
            if (meets) {
                const unsigned st = V.htab[hs].start;
                unsigned j = 0;
                for (unsigned i = 0; i < cn; i++) {
                    const float4 p = V.pts[st + i];
                    bool gone = false;
                    for (int b = 0; b < nb; b++) {
                        const float *B = boxes + b * 6;
                        if (B[0] <= p.x && B[3] > p.x && B[1] <= p.y && B[4] > p.y && B[2] <= p.z && B[5] > p.z) gone = true;
                    }
                    if (gone) { dead[__float_as_int(p.w)] = 1; removed++; continue; }
                    if (i != j) V.pts[st + j] = p;
                    j++;
                }
                V.htab[hs].count = j;
            }
        }
    }
    fl_mi_count_block(removed, FL_MI_CNT_REMOVED, ctl);
}

// all slots of a fresh table: no capacity, clean, nothing queued
__global__ __launch_bounds__(FL_BLOCK) void mapinc_clear_kernel(unsigned *__restrict__ cellcap, unsigned *__restrict__ dirty, unsigned *__restrict__ pend_head,
                                                               unsigned slots, FlMapIncCtl *__restrict__ ctl)
{
    const unsigned i = blockIdx.x * FL_BLOCK + threadIdx.x;
    if (i == 0) ctl->ncells = 0u;
    if (i >= slots) return;
    cellcap[i] = 0u; dirty[i] = 0u; pend_head[i] = FL_MI_NONE;
}
__global__ __launch_bounds__(FL_BLOCK) void mapinc_count_cells_kernel(const FlCellEntry *__restrict__ htab, unsigned slots, FlMapIncCtl *__restrict__ ctl)
{
    // (grid-stride over at most 256 workgroups, one atomic each: one atomic per wavefront on this one address was 3 ms of a 5 M-point map's
    // re-index -- 260 k same-address atomics at ~12 ns)
    __shared__ unsigned s_c[FL_BLOCK / 64];
    unsigned c = 0;
    for (unsigned i = blockIdx.x * FL_BLOCK + threadIdx.x; i < slots; i += gridDim.x * FL_BLOCK) c += (htab[i].key != FL_KNN_EMPTY) ? 1u : 0u;
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) c += __shfl_xor(c, k);
    if ((threadIdx.x & 63u) == 0) s_c[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned t = 0;
        for (int w = 0; w < FL_BLOCK / 64; w++) t += s_c[w];
        if (t) atomicAdd(&ctl->ncells, t);
    }
}
// flags of the compaction that precedes a full rebuild: 1 = live
__global__ __launch_bounds__(FL_BLOCK) void mapinc_live_flags_kernel(const unsigned char *__restrict__ dead, int n, int *__restrict__ flags)
{
    const int i = blockIdx.x * FL_BLOCK + threadIdx.x;
    if (i < n) flags[i] = dead[i] ? 0 : 1;
}
// (upd, nullable: the down-sampling update's info block is reset here too -- mapupd_init_kernel was a launch of its own)
__global__ void mapinc_begin_kernel(FlMapIncCtl *__restrict__ ctl, FlMapUpdInfo *__restrict__ upd = nullptr)
{
    if (blockIdx.x != 0) return;
    if (upd && threadIdx.x == 1) { upd->total = 0; upd->kept_old = 0; upd->kept_new = 0; upd->ambiguous = 0; upd->range_error = 0; }
    if (threadIdx.x == 0) { ctl->added = 0; ctl->removed = 0; ctl->ambiguous = 0; ctl->range_error = 0; }
    FlMapIncStripes *S = fl_mi_stripes(ctl);
    for (int e = (int)threadIdx.x; e < 3 * FL_MI_STRIPES; e += (int)blockDim.x) S->v[e / FL_MI_STRIPES][e % FL_MI_STRIPES][0] = 0;
}
// the update's last kernel: live count + the status block in page-locked memory (system-scope stores; the host polls seq)
__global__ void mapinc_status_kernel(FlMapIncCtl *__restrict__ ctl, const FlMapUpdInfo *__restrict__ upd, FlMapIncCtl *__restrict__ status, unsigned long long seq)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    {
        FlMapIncStripes *S = fl_mi_stripes(ctl);
        int a = 0, r = 0, m = 0;
        for (int k = 0; k < FL_MI_STRIPES; k++) { a += S->v[FL_MI_CNT_ADDED][k][0]; r += S->v[FL_MI_CNT_REMOVED][k][0]; m += S->v[FL_MI_CNT_AMBIGUOUS][k][0]; }
        for (int k = 0; k < FL_MI_STRIPES; k++) { S->v[FL_MI_CNT_ADDED][k][0] = 0; S->v[FL_MI_CNT_REMOVED][k][0] = 0; S->v[FL_MI_CNT_AMBIGUOUS][k][0] = 0; }
        ctl->added += a; ctl->removed += r; ctl->ambiguous += m;
    }
    ctl->live += ctl->added - ctl->removed;
    if (upd) { ctl->ambiguous += upd->ambiguous; ctl->range_error = upd->range_error; }      // (new points: mapupd_new_kernel; old ones: mapinc_resolve_kernel)
    FlMapIncCtl c = *ctl;
    c.seq = 0ull;
    *status = c;
    __threadfence_system();
    __hip_atomic_store(&status->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// index build with slack (map_rebuild): the pool's top and the cell count of a fresh index
__global__ void mapinc_init_ctl_kernel(FlMapIncCtl *__restrict__ ctl, const FlCellEntry *__restrict__ htab, const unsigned *__restrict__ first, unsigned slots,
                                      unsigned pool_cap, int live)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const unsigned last = slots - 1u;
    ctl->pool_top = first[last] + FlCellCap()(htab[last]);
    ctl->pool_cap = pool_cap; ctl->slots = slots; ctl->live = live; ctl->needs_rebuild = 0;
    ctl->added = ctl->removed = ctl->ambiguous = ctl->range_error = 0;
}
