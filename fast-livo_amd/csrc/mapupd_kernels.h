// mapupd_kernels.h -- the map side of the device k-NN: the LiDAR map stays on the
// device between frames and is updated there, instead of being rebuilt on the host and re-staged.
//   fl_map_add_points    <- map_incremental (src/laserMapping.cpp:692-706) -> KD_TREE::Add_Points(points, downsample_on = true)
//                           (include/ikd-Tree/ikd_Tree.cpp:382-457)
//   fl_map_delete_boxes  <- lasermap_fov_segment (src/laserMapping.cpp:363-417) -> KD_TREE::Delete_Point_Boxes (ikd_Tree.cpp:501-520,
//                           box test of Delete_by_range :626-650: min <= v && max > v on every axis)
//
// What Add_Points does, stated as a set operation. Every new point p owns the box [floor(p/ds)*ds, +ds) of the down-sampling
// grid and that box's centre. The points are processed one after the other: the map points inside the box are collected
// (Search_by_range), the candidate closest to the centre among {p} + collected wins (a collected point only if STRICTLY closer
// than p), and unless the box held exactly one point that beats p, the box is emptied and the winner (re)inserted. Boxes are
// independent of each other and inside a box the outcome does not depend on the interleaving, so the sequential loop equals:
//   * a box no new point falls into is untouched (it may hold several points: the first frame's Build does not down-sample);
//   * a touched box ends with exactly one point: the old point closest to the centre if it is strictly closer than EVERY new
//     point of the box, else the new point closest to the centre, the LATEST one among equals.
// (Equal distances among old points of one box are decided by the tree's traversal order in the reference; here by the lower
// map index, as in the k-NN.) Distances are the tree's float arithmetic (calc_dist, ikd_Tree.cpp:1291-1295), the box and its
// centre the floats of :392-400.
// The partition is by the integer floor(p/ds) per axis. floor(v/ds)*ds can round to the other side of v when ds is not a power
// of two, in which case the reference's coordinate test puts the point into a different (or no) box than its own; such points
// are counted (`n_ambiguous`, probability ~1e-7 per coordinate) so that a caller or a test can tell.
//
// The map array keeps its order: surviving points first, in their old order, then the added points in input order. The k-NN
// index (cell sort + hash table, knn_kernels.h) is rebuilt from it -- sort + build of the whole local map is cheaper here than
// pointer surgery on a tree.
#pragma once

#include "knn_kernels.h"

struct FlBoxSlot {
    unsigned long long key;        // fl_cell_key of the box, FL_KNN_EMPTY: free
    unsigned long long best_new;   // (float bits of the distance to the centre << 32) | ~input index : min = closest, latest among equals
    unsigned long long best_old;   // (float bits << 32) | map index : min = closest, lowest index among equals
};

struct FlMapUpdInfo {
    int total;          // points after the update
    int kept_old;
    int kept_new;
    int ambiguous;
    int range_error;    // |floor(p/ds)| >= 2^20 somewhere: the box key does not hold it
    int pad[3];
};

struct FlBoxGeom {
    int ix, iy, iz;
    float cx, cy, cz;
    int ambiguous, range_error;
};

__device__ __forceinline__ void fl_box_axis(float v, float ds, int &i, float &c, int &amb, int &range_err)
{
    const float f = floorf(v / ds);
    const float mn = f * ds;                         // vertex_min (:392)
    const float mx = mn + ds;                        // vertex_max (:393)
    c = (float)((double)mn + (double)(mx - mn) / 2.0);      // mid_point (:398), a float member
    // the reference's membership test is on coordinates (min <= v && max > v): the point must lie in its own box and in neither
    // neighbour's
    const float lo_max = (f - 1.0f) * ds + ds;
    const float hi_min = (f + 1.0f) * ds;
    if (!(mn <= v && mx > v) || lo_max > v || hi_min <= v) amb = 1;
    if (!(fabsf(f) < 1048576.0f)) { range_err = 1; i = 0; }
    else i = (int)f;
}
__device__ __forceinline__ FlBoxGeom fl_box_of(float x, float y, float z, float ds)
{
    FlBoxGeom g;
    g.ambiguous = 0; g.range_error = 0;
    fl_box_axis(x, ds, g.ix, g.cx, g.ambiguous, g.range_error);
    fl_box_axis(y, ds, g.iy, g.cy, g.ambiguous, g.range_error);
    fl_box_axis(z, ds, g.iz, g.cz, g.ambiguous, g.range_error);
    return g;
}
__device__ __forceinline__ float fl_calc_dist(float ax, float ay, float az, float bx, float by, float bz)
{
    return (ax - bx) * (ax - bx) + (ay - by) * (ay - by) + (az - bz) * (az - bz);      // ikd_Tree.cpp:1293, no contraction
}
__device__ __forceinline__ void fl_count_wave(int flag, int *counter)
{
    const unsigned long long b = __ballot(flag != 0);
    if (b && (threadIdx.x & 63) == (unsigned)(__ffsll((long long)b) - 1)) atomicAdd(counter, __popcll(b));
}

__global__ __launch_bounds__(FL_BLOCK) void mapupd_init_kernel(FlMapUpdInfo *__restrict__ info)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        info->total = 0; info->kept_old = 0; info->kept_new = 0; info->ambiguous = 0; info->range_error = 0;
    }
}

// new points: claim the box, compete for "best new point of the box"
// raw_out / dead_out (in-place map update, nullable): the point also goes to the end of the map array, flagged dead until its box admits it
// (they were a device-to-device copy command and a fill in front of this launch: two DMA-to-kernel transitions per update)
__global__ __launch_bounds__(FL_BLOCK) void mapupd_new_kernel(const float *__restrict__ pts, int n, float ds, FlBoxSlot *__restrict__ tab,
                                                             unsigned mask, int *__restrict__ slot_of, FlMapUpdInfo *__restrict__ info,
                                                             float *__restrict__ raw_out = nullptr, unsigned char *__restrict__ dead_out = nullptr)
{
    const int j = blockIdx.x * FL_BLOCK + threadIdx.x;
    int amb = 0, rerr = 0;
    if (j < n) {
        const float x = pts[j * 3], y = pts[j * 3 + 1], z = pts[j * 3 + 2];
        if (raw_out) { raw_out[j * 3] = x; raw_out[j * 3 + 1] = y; raw_out[j * 3 + 2] = z; dead_out[j] = 1; }
        const FlBoxGeom g = fl_box_of(x, y, z, ds);
        amb = g.ambiguous; rerr = g.range_error;
        const unsigned long long key = fl_cell_key(g.ix, g.iy, g.iz);
        const float d = fl_calc_dist(x, y, z, g.cx, g.cy, g.cz);
        unsigned h = fl_hash64(key) & mask;
        while (true) {
            const unsigned long long prev = atomicCAS(&tab[h].key, FL_KNN_EMPTY, key);
            if (prev == FL_KNN_EMPTY || prev == key) break;
            h = (h + 1) & mask;
        }
        atomicMin(&tab[h].best_new, ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)j));
        slot_of[j] = (int)h;
    }
    fl_count_wave(amb, &info->ambiguous);
    fl_count_wave(rerr, &info->range_error);
}

// map points: those whose box was claimed by a new point compete for "best old point of the box"
__global__ __launch_bounds__(FL_BLOCK) void mapupd_old_kernel(const float *__restrict__ pts, int n, float ds, FlBoxSlot *__restrict__ tab,
                                                             unsigned mask, int *__restrict__ slot_of, FlMapUpdInfo *__restrict__ info)
{
    const int i = blockIdx.x * FL_BLOCK + threadIdx.x;
    int amb = 0, rerr = 0;
    if (i < n) {
        const float x = pts[i * 3], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
        const FlBoxGeom g = fl_box_of(x, y, z, ds);
        amb = g.ambiguous; rerr = g.range_error;
        const unsigned long long key = fl_cell_key(g.ix, g.iy, g.iz);
        unsigned h = fl_hash64(key) & mask;
        int slot = -1;
        while (true) {
            const unsigned long long k = tab[h].key;
            if (k == key) { slot = (int)h; break; }
            if (k == FL_KNN_EMPTY) break;
            h = (h + 1) & mask;
        }
        if (slot >= 0) {
            const float d = fl_calc_dist(x, y, z, g.cx, g.cy, g.cz);
            atomicMin(&tab[slot].best_old, ((unsigned long long)__float_as_uint(d) << 32) | (unsigned long long)(unsigned)i);
        }
        slot_of[i] = slot;
    }
    fl_count_wave(amb, &info->ambiguous);
    fl_count_wave(rerr, &info->range_error);
}

// survivors: flags[0..n_old) for the map points, flags[n_old..n_old+n_new) for the new points
__global__ __launch_bounds__(FL_BLOCK) void mapupd_flags_kernel(const FlBoxSlot *__restrict__ tab, const int *__restrict__ slot_old, int n_old,
                                                               const int *__restrict__ slot_new, int n_new, int downsample,
                                                               int *__restrict__ flags)
{
    const int t = blockIdx.x * FL_BLOCK + threadIdx.x;
    int keep = 0;
    if (t < n_old + n_new) {
        if (!downsample) keep = 1;
        else if (t < n_old) {
            const int s = slot_old[t];
            if (s < 0) keep = 1;
            else {
                const unsigned long long bo = tab[s].best_old, bn = tab[s].best_new;
                keep = ((unsigned)bo == (unsigned)t) && ((unsigned)(bo >> 32) < (unsigned)(bn >> 32));      // strictly closer than every new point
            }
        } else {
            const int j = t - n_old;
            const FlBoxSlot e = tab[slot_new[j]];
            const bool old_wins = (e.best_old != FL_KNN_EMPTY) && ((unsigned)(e.best_old >> 32) < (unsigned)(e.best_new >> 32));
            keep = ((unsigned)e.best_new == 0xFFFFFFFFu - (unsigned)j) && !old_wins;
        }
        flags[t] = keep;
    }
}

// box deletion: a map point goes iff it lies in one of the boxes (min <= v && max > v on every axis)
__global__ __launch_bounds__(FL_BLOCK) void mapupd_boxflags_kernel(const float *__restrict__ pts, int n, const float *__restrict__ boxes, int nb,
                                                                  int *__restrict__ flags)
{
    const int i = blockIdx.x * FL_BLOCK + threadIdx.x;
    if (i < n) {
        const float x = pts[i * 3], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
        int keep = 1;
        for (int b = 0; b < nb; b++) {
            const float *B = boxes + b * 6;
            if (B[0] <= x && B[3] > x && B[1] <= y && B[4] > y && B[2] <= z && B[5] > z) keep = 0;
        }
        flags[i] = keep;
    }
}

// stable compaction: pos = exclusive prefix sum of flags. The survivor counts fall out of the prefix sums (no atomics: one
// same-address atomic per wave cost 38 us on a 200 k-point map).
__global__ __launch_bounds__(FL_BLOCK) void mapupd_scatter_kernel(const float *__restrict__ old_pts, int n_old, const float *__restrict__ new_pts,
                                                                 int n_new, const int *__restrict__ flags, const int *__restrict__ pos,
                                                                 float *__restrict__ out, FlMapUpdInfo *__restrict__ info)
{
    const int t = blockIdx.x * FL_BLOCK + threadIdx.x;
    if (t >= n_old + n_new) return;
    const int f = flags[t], p = pos[t];
    if (f) {
        const float *src = (t < n_old) ? old_pts + (size_t)t * 3 : new_pts + (size_t)(t - n_old) * 3;
        out[(size_t)p * 3] = src[0]; out[(size_t)p * 3 + 1] = src[1]; out[(size_t)p * 3 + 2] = src[2];
    }
    if (t == n_old + n_new - 1) {
        info->total = p + f;
        if (n_new == 0) { info->kept_old = p + f; info->kept_new = 0; }
    }
    if (n_new > 0 && t == n_old) {          // first new point: everything before it is a surviving map point
        info->kept_old = p;
    }
}
