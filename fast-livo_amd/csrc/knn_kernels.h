// knn_kernels.h -- SURVEY.md section 8(f) row N1: exact 5-nearest-neighbour search of the scan points in the
// LiDAR map ON THE DEVICE, replacing the host ikd-Tree search of the 2 search passes per frame
// (KD_TREE::Nearest_Search, include/ikd-Tree/ikd_Tree.cpp:350-380, call sites src/laserMapping.cpp:1543,
// :1002) and with it the 0.6 MB world-point read-back and the 3 MB neighbour restage.
//
// Index: uniform voxel grid over the map, stored sparsely -- map points sorted by a 63-bit cell key
// (hipCUB radix sort, once per map update), an open-addressing hash table cell-key -> first sorted
// point. It is rebuilt whenever the host map changes (map_incremental, laserMapping.cpp:692-706).
// Search (hand-written): one lane per scan point; world point from the current state exactly like the
// residual kernels; cells are visited in growing Chebyshev rings around the query's cell and the
// search stops as soon as the 5th best distance is <= (ring * cell)^2, which proves that no
// unvisited point can be closer (any point outside the visited cube is farther than ring*cell).
// Rings are capped where (ring * cell)^2 > 5: farther neighbours make the point invalid anyway
// (laserMapping.cpp:1549: sqdist[4] > 5). Distances are the ikd-Tree's float arithmetic
// (ikd_Tree.cpp:1291-1295); exact ties are broken by the lower map index (the tree's traversal
// order is not reproducible; oracle/orc_knn.c uses the same rule).
// The plane fit (K0) is fused: the kernel writes the plane and the selection flag directly.
#pragma once

#include "fl_device.h"
#include "fl_math.h"
#include "fl_ikfom_math.h"

#define FL_KNN_EMPTY 0xFFFFFFFFFFFFFFFFull

struct FlMapGrid {
    const float4 *pts;               // sorted by cell key: xyz + original index (as int bits) in w
    const unsigned long long *keys;  // sorted cell key of every point
    const unsigned long long *hkeys; // hash table: cell key or FL_KNN_EMPTY
    const unsigned *hstart;          // hash table: index of the cell's first sorted point
    unsigned hmask;                  // table size - 1 (power of two)
    int npts;
    float cell;                      // edge length
    float inv_cell;
    int max_ring;                    // smallest r with (r*cell)^2 > 5
};

__device__ __forceinline__ unsigned long long fl_cell_key(int ix, int iy, int iz)
{
    return ((unsigned long long)(unsigned)(ix + (1 << 20)) << 42) | ((unsigned long long)(unsigned)(iy + (1 << 20)) << 21) |
           (unsigned long long)(unsigned)(iz + (1 << 20));
}
__device__ __forceinline__ unsigned fl_hash64(unsigned long long k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (unsigned)k;
}

__global__ __launch_bounds__(FL_BLOCK) void knn_keys_kernel(const float *__restrict__ map_xyz, int k, float inv_cell,
                                                           unsigned long long *__restrict__ keys, unsigned *__restrict__ idx)
{
    const int i = blockIdx.x * FL_BLOCK + threadIdx.x;
    if (i >= k) return;
    const int ix = (int)floorf(map_xyz[i * 3] * inv_cell), iy = (int)floorf(map_xyz[i * 3 + 1] * inv_cell),
              iz = (int)floorf(map_xyz[i * 3 + 2] * inv_cell);
    keys[i] = fl_cell_key(ix, iy, iz);
    idx[i] = (unsigned)i;
}

__global__ __launch_bounds__(FL_BLOCK) void knn_build_kernel(const float *__restrict__ map_xyz, const unsigned long long *__restrict__ skeys,
                                                            const unsigned *__restrict__ sidx, int k, float4 *__restrict__ pts,
                                                            unsigned long long *__restrict__ hkeys, unsigned *__restrict__ hstart,
                                                            unsigned hmask)
{
    const int i = blockIdx.x * FL_BLOCK + threadIdx.x;
    if (i >= k) return;
    const unsigned o = sidx[i];
    pts[i] = make_float4(map_xyz[o * 3], map_xyz[o * 3 + 1], map_xyz[o * 3 + 2], __int_as_float((int)o));
    const unsigned long long key = skeys[i];
    if (i == 0 || skeys[i - 1] != key) {          // first point of its cell: claim a slot
        unsigned h = fl_hash64(key) & hmask;
        while (true) {
            const unsigned long long prev = atomicCAS((unsigned long long *)&hkeys[h], FL_KNN_EMPTY, key);
            if (prev == FL_KNN_EMPTY || prev == key) { hstart[h] = (unsigned)i; break; }
            h = (h + 1) & hmask;
        }
    }
}

struct FlTop5 {
    float d[5];
    int id[5];      // original map index (tie-break)
    int at[5];      // position in the sorted array
};
__device__ __forceinline__ void fl_top5_insert(FlTop5 &t, float d, int id, int at)
{
    if (!(d < t.d[4] || (d == t.d[4] && id < t.id[4]))) return;
#pragma unroll
    for (int p = 4; p >= 0; p--) {
        const bool before_prev = (p > 0) && (d < t.d[p - 1] || (d == t.d[p - 1] && id < t.id[p - 1]));
        if (before_prev) { t.d[p] = t.d[p - 1]; t.id[p] = t.id[p - 1]; t.at[p] = t.at[p - 1]; }
        else { t.d[p] = d; t.id[p] = id; t.at[p] = at; break; }
    }
}

__device__ __forceinline__ void fl_scan_cell(const FlMapGrid &G, int ix, int iy, int iz, float qx, float qy, float qz, FlTop5 &t)
{
    const unsigned long long key = fl_cell_key(ix, iy, iz);
    unsigned h = fl_hash64(key) & G.hmask;
    while (true) {
        const unsigned long long hk = G.hkeys[h];
        if (hk == FL_KNN_EMPTY) return;
        if (hk == key) break;
        h = (h + 1) & G.hmask;
    }
    for (int j = (int)G.hstart[h]; j < G.npts && G.keys[j] == key; j++) {
        const float4 p = G.pts[j];
        const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
        const float d = dx * dx + dy * dy + dz * dz;        // ikd_Tree.cpp:1293 (no contraction)
        fl_top5_insert(t, d, __float_as_int(p.w), j);
    }
}

// MODE 18: world point from FlDev18 ; MODE 23: from FlDev23. `cond` != 0: run only when the device
// raised need_search and not stop (the frame drivers enqueue it before every pass).
template <int MODE, typename DEV>
__global__ __launch_bounds__(FL_BLOCK) void lio_search_fit_kernel(const float *__restrict__ body, int n, FlMapGrid G, DEV *__restrict__ D,
                                                                 float4 *__restrict__ plane, uint8_t *__restrict__ sel,
                                                                 float *__restrict__ nbr_out /* nullable n x 15 */,
                                                                 uint8_t *__restrict__ valid_out /* nullable */, int cond)
{
    if (cond && (!D->need_search || D->stop)) return;
    const int i = blockIdx.x * FL_BLOCK + threadIdx.x;
    if (i < n) {
        const float pb[3] = {body[i * 3], body[i * 3 + 1], body[i * 3 + 2]};
        float pw[3];
        if constexpr (MODE == 18) {
            const FlDev18 *D18 = D;
            const double b0 = (double)pb[0], b1 = (double)pb[1], b2 = (double)pb[2];
            const double q0 = (D18->R_LI[0] * b0 + D18->R_LI[1] * b1 + D18->R_LI[2] * b2) + D18->t_LI[0];
            const double q1 = (D18->R_LI[3] * b0 + D18->R_LI[4] * b1 + D18->R_LI[5] * b2) + D18->t_LI[1];
            const double q2 = (D18->R_LI[6] * b0 + D18->R_LI[7] * b1 + D18->R_LI[8] * b2) + D18->t_LI[2];
            pw[0] = (float)((D18->x[0] * q0 + D18->x[1] * q1 + D18->x[2] * q2) + D18->x[9]);
            pw[1] = (float)((D18->x[3] * q0 + D18->x[4] * q1 + D18->x[5] * q2) + D18->x[10]);
            pw[2] = (float)((D18->x[6] * q0 + D18->x[7] * q1 + D18->x[8] * q2) + D18->x[11]);
        } else {
            double x[FL_X23_LEN], p_i[3];
#pragma unroll
            for (int k = 0; k < FL_X23_LEN; k++) x[k] = D->x[k];
            fl_world_point23(x, pb, p_i, pw);
        }
        const int cx = (int)floorf(pw[0] * G.inv_cell), cy = (int)floorf(pw[1] * G.inv_cell), cz = (int)floorf(pw[2] * G.inv_cell);
        FlTop5 t;
#pragma unroll
        for (int k = 0; k < 5; k++) { t.d[k] = INFINITY; t.id[k] = 0x7fffffff; t.at[k] = -1; }
        for (int r = 0; r <= G.max_ring; r++) {
            for (int dz = -r; dz <= r; dz++)
                for (int dy = -r; dy <= r; dy++) {
                    const bool face = (dz == -r || dz == r || dy == -r || dy == r);
                    if (face) {
                        for (int dx = -r; dx <= r; dx++) fl_scan_cell(G, cx + dx, cy + dy, cz + dz, pw[0], pw[1], pw[2], t);
                    } else {            // only the two x-faces of the shell
                        fl_scan_cell(G, cx - r, cy + dy, cz + dz, pw[0], pw[1], pw[2], t);
                        fl_scan_cell(G, cx + r, cy + dy, cz + dz, pw[0], pw[1], pw[2], t);
                    }
                }
            // 1 mm safety margin: cell boundaries are evaluated in float (floorf(x * inv_cell)), exact to
            // well under a millimetre for maps of several kilometres
            const float reach = (float)r * G.cell - 1e-3f;
            if (reach > 0.f && t.d[4] <= reach * reach) break;
        }
        const int found5 = t.at[4] >= 0;
        const int valid = found5 && !(t.d[4] > 5.0f);
        float nb[15];
#pragma unroll
        for (int k = 0; k < 5; k++) {
            if (t.at[k] >= 0) {
                const float4 p = G.pts[t.at[k]];
                nb[k * 3] = p.x; nb[k * 3 + 1] = p.y; nb[k * 3 + 2] = p.z;
            } else {
                nb[k * 3] = 0.f; nb[k * 3 + 1] = 0.f; nb[k * 3 + 2] = 0.f;
            }
        }
        float pl[4];
        const int ok = fl_esti_plane(nb, pl);
        plane[i] = make_float4(pl[0], pl[1], pl[2], pl[3]);
        sel[i] = (uint8_t)(valid && ok);
        if (nbr_out) {
#pragma unroll
            for (int k = 0; k < 15; k++) nbr_out[(size_t)i * 15 + k] = nb[k];
        }
        if (valid_out) valid_out[i] = (uint8_t)valid;
    }
    // the search pass is done: nearest_search_en = false for the passes that follow. All threads of the
    // grid read need_search before anyone clears it only if the clear happens in a later launch, so the
    // flag is cleared by a separate tiny kernel (knn_clear_flag_kernel) enqueued right after this one.
}

template <typename DEV>
__global__ void knn_clear_flag_kernel(DEV *__restrict__ D, int cond)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (cond && (!D->need_search || D->stop)) return;
    D->need_search = 0;
}
