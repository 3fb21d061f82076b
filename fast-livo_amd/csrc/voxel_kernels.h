// voxel_kernels.h -- SURVEY.md section 8(f) row N3: the scan's voxel down-sampling on the device, i.e.
// pcl::VoxelGrid<PointType>::applyFilter as called at src/laserMapping.cpp:1398-1399 (downSizeFilterSurf,
// leaf = filter_size_surf, :1186) and src/lidar_selection.cpp:352-353 (leaf 0.2, :7). PCL itself is not part of the
// reference tree; the algorithm is restated from pcl/filters/impl/voxel_grid.hpp (see oracle/orc_voxel.c for the
// step list). Arithmetic that decides voxel membership is PCL's float arithmetic, expression by expression:
//   inverse_leaf = 1.0f / leaf ; min_b = (int)floor(min * inverse_leaf) ; ijk = (int)(floor(p * inverse_leaf) - (float)min_b)
// Output order = ascending voxel index (PCL's sorted order). Inside a voxel PCL's summation order is whatever
// std::sort leaves; here it is ascending cloud index (stable radix sort), the same rule as the oracle.
//
//   vox_minmax_kernel    bounding box of the finite points (wave reduction + ordered-int atomics)
//   vox_keys_kernel      voxel index of every point (0xFFFFFFFF: not finite)        -> hipCUB radix sort (idx, point)
//   vox_heads_kernel     1 at the first sorted entry of every voxel                -> hipCUB exclusive scan = output slot
//   vox_centroid_kernel  the lane that owns a voxel's first entry walks the run, sums in float, divides by the count
#pragma once

#include "fl_device.h"

struct FlVoxCtl {
    int mn[3], mx[3];       // ordered-int images of the float bounds
    int count;              // number of output points
    int leaf_too_small;     // PCL's "Leaf size is too small" case: output = input
    int nfinite;
};

__device__ __forceinline__ int fl_ordered_int(float f)
{
    const int i = __float_as_int(f);
    return i ^ ((i >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float fl_ordered_float(int i) { return __int_as_float(i ^ ((i >> 31) & 0x7fffffff)); }

__global__ void vox_init_kernel(FlVoxCtl *__restrict__ C)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int k = 0; k < 3; k++) { C->mn[k] = 0x7fffffff; C->mx[k] = (int)0x80000000; }
    C->count = 0; C->leaf_too_small = 0; C->nfinite = 0;
}

__global__ __launch_bounds__(FL_BLOCK) void vox_minmax_kernel(const float4 *__restrict__ in, int n, FlVoxCtl *__restrict__ C)
{
    int mn[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, mx[3] = {(int)0x80000000, (int)0x80000000, (int)0x80000000};
    int cnt = 0;
    for (int i = blockIdx.x * FL_BLOCK + threadIdx.x; i < n; i += gridDim.x * FL_BLOCK) {
        const float4 p = in[i];
        if (!isfinite(p.x) || !isfinite(p.y) || !isfinite(p.z)) continue;
        const int o[3] = {fl_ordered_int(p.x), fl_ordered_int(p.y), fl_ordered_int(p.z)};
#pragma unroll
        for (int k = 0; k < 3; k++) { mn[k] = min(mn[k], o[k]); mx[k] = max(mx[k], o[k]); }
        cnt++;
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
#pragma unroll
        for (int k = 0; k < 3; k++) { mn[k] = min(mn[k], __shfl_xor(mn[k], s)); mx[k] = max(mx[k], __shfl_xor(mx[k], s)); }
        cnt += __shfl_xor(cnt, s);
    }
    // one set of atomics per workgroup (they all hit the same 7 words: ~12 ns each, serialised)
    __shared__ int s_red[FL_BLOCK / 64][7];
    const int w = (int)(threadIdx.x >> 6);
    if ((threadIdx.x & 63u) == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) { s_red[w][k] = mn[k]; s_red[w][3 + k] = mx[k]; }
        s_red[w][6] = cnt;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int v = 1; v < FL_BLOCK / 64; v++) {
#pragma unroll
            for (int k = 0; k < 3; k++) { mn[k] = min(mn[k], s_red[v][k]); mx[k] = max(mx[k], s_red[v][3 + k]); }
            cnt += s_red[v][6];
        }
#pragma unroll
        for (int k = 0; k < 3; k++) { atomicMin(&C->mn[k], mn[k]); atomicMax(&C->mx[k], mx[k]); }
        atomicAdd(&C->nfinite, cnt);
    }
}

struct FlVoxGrid {
    float inv[3];
    int min_b[3];
    int mul[3];
    int too_small;
};
__device__ __forceinline__ FlVoxGrid fl_vox_grid(const FlVoxCtl *C, float ilx, float ily, float ilz)
{
    FlVoxGrid g;
    g.inv[0] = ilx; g.inv[1] = ily; g.inv[2] = ilz;
    int div_b[3];
    long long d[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float mn = fl_ordered_float(C->mn[k]), mx = fl_ordered_float(C->mx[k]);
        d[k] = (long long)((mx - mn) * g.inv[k]) + 1;                 // voxel_grid.hpp: dx, dy, dz
        g.min_b[k] = (int)floorf(mn * g.inv[k]);
        div_b[k] = (int)floorf(mx * g.inv[k]) - g.min_b[k] + 1;
    }
    g.too_small = (d[0] * d[1] * d[2]) > 2147483647ll;
    g.mul[0] = 1; g.mul[1] = div_b[0]; g.mul[2] = div_b[0] * div_b[1];
    return g;
}

__global__ __launch_bounds__(FL_BLOCK) void vox_keys_kernel(const float4 *__restrict__ in, int n, FlVoxCtl *__restrict__ C, float ilx,
                                                           float ily, float ilz, unsigned *__restrict__ keys, unsigned *__restrict__ vals)
{
    const int i = blockIdx.x * FL_BLOCK + threadIdx.x;
    if (i >= n) return;
    unsigned key = 0xFFFFFFFFu;
    if (C->nfinite > 0) {
        const FlVoxGrid g = fl_vox_grid(C, ilx, ily, ilz);
        if (i == 0) C->leaf_too_small = g.too_small;
        const float4 p = in[i];
        if (isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
            if (g.too_small) {
                key = (unsigned)i;                                      // output = input: every point its own "voxel"
            } else {
                const int i0 = (int)(floorf(p.x * g.inv[0]) - (float)g.min_b[0]);
                const int i1 = (int)(floorf(p.y * g.inv[1]) - (float)g.min_b[1]);
                const int i2 = (int)(floorf(p.z * g.inv[2]) - (float)g.min_b[2]);
                key = (unsigned)(i0 * g.mul[0] + i1 * g.mul[1] + i2 * g.mul[2]);
            }
        }
    }
    keys[i] = key;
    vals[i] = (unsigned)i;
}

__global__ __launch_bounds__(FL_BLOCK) void vox_heads_kernel(const unsigned *__restrict__ skeys, int n, unsigned *__restrict__ heads)
{
    const int i = blockIdx.x * FL_BLOCK + threadIdx.x;
    if (i >= n) return;
    const unsigned k = skeys[i];
    heads[i] = (k != 0xFFFFFFFFu && (i == 0 || skeys[i - 1] != k)) ? 1u : 0u;
}

// out: centroids (x, y, z, intensity); body (nullable): xyz only, the staged-scan layout of the LIO kernels
__global__ __launch_bounds__(FL_BLOCK) void vox_centroid_kernel(const float4 *__restrict__ in, const unsigned *__restrict__ skeys,
                                                               const unsigned *__restrict__ svals, const unsigned *__restrict__ heads,
                                                               const unsigned *__restrict__ slot, int n, float4 *__restrict__ out,
                                                               float *__restrict__ body, FlVoxCtl *__restrict__ C)
{
    const int i = blockIdx.x * FL_BLOCK + threadIdx.x;
    if (i >= n) return;
    if (i == n - 1) C->count = (int)(slot[i] + heads[i]);
    if (!heads[i]) return;
    const unsigned k = skeys[i];
    float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
    int j = i;
    for (; j < n && skeys[j] == k; j++) {
        const float4 p = in[svals[j]];
        sx = sx + p.x; sy = sy + p.y; sz = sz + p.z; si = si + p.w;
    }
    const float cnt = (float)(j - i);
    const float4 c = make_float4(sx / cnt, sy / cnt, sz / cnt, si / cnt);
    const unsigned o = slot[i];
    out[o] = c;
    if (body) { body[3 * o] = c.x; body[3 * o + 1] = c.y; body[3 * o + 2] = c.z; }
}


// ================================================================================================================================
// Round 5: the same filter WITHOUT a sort (the hipCUB radix sort + scan above were 10 of the 13 launches and 48 of the 69 us of a
// 24 k-point scan; they stay as the path for very large clouds and as the A/B reference, fl_set_option(FL_OPT_VOXEL_SORT, 1)).
//
// PCL's output order is ascending voxel index; the index space of a grid is bounded (dx*dy*dz <= INT32_MAX or PCL does not filter
// at all), so the ORDER comes from an occupancy bitmap over the cells instead of a sort:
//     rank(cell) = number of occupied cells with a smaller index
//                = l2pre[cell >> 18] + l1pre[cell >> 8] + popcount(bits of the cell's 256-cell group below it)
// and the summation order inside a voxel (ascending cloud index -- PCL leaves it to std::sort, the oracle and the old path fix it
// so) from counting, per point, the members of its voxel with a smaller index. Eight small launches, plain loads and stores between
// them, integer atomics only where their order cannot matter (bit claims, member counters):
//   vx_minmax      bounding box of the finite points, one record per workgroup (fl_lidar_front: written by the undistortion's last kernel
//                  instead -- seven launches); vx_claim folds the records
//   vx_claim       voxel index per point; a fire-and-forget atomicOr raises the cell's bit; the point marks its 2^18-cell block (l2flag)
//   vx_scan_l2     one workgroup per TOUCHED 2^18-cell block: group counts (popcounts of the bitmap), exclusive prefix over its 1024
//                  groups, block total
//   vx_rank        per point: rank of its voxel = output slot o; arrival position pos = atomicAdd(cnt[o])
//   vx_segments    per voxel: a segment of cnt[o] entries (workgroup-aggregated bump allocation)
//   vx_scatter     per point: member list, arrival order
//   vx_order       per point: its place among its voxel's members by ascending cloud index
//   vx_centroid    per voxel: float sums in that order, / count -- and the clean-up: every word this frame set in the bitmap, the
//                  block flags and the voxel counters is zeroed again (touched words only; nothing is memset)
// Everything the sequence sets it clears, so the (large) bitmap is zeroed once, at allocation. Cells beyond the bitmap's capacity:
// cells_short is raised, nothing is filtered, the host grows the bitmap and runs the frame again (first frames of a run only).
// Bit-identical to the sorted path and to oracle/orc_voxel.c: same keys, same output order, same summation order.
// (Round 6, built, measured and removed: FIVE launches instead of seven -- a member counter per grid CELL (512 MB for 2^27 cells) raised where
// the voxel is claimed, plus one per 32-cell bitmap word, so that the counts are final before the prefix pass, the pass scans members along
// with voxels, and rank + segment start + scatter are one kernel. Bit-identical, and no faster: 35.7 vs 38 us at 24 k points stand-alone, but
// fl_lidar_front 0.196 -> 0.217 ms and the camera half 0.167 -> 0.171 ms -- 100 k atomics WITH a return value scattered over a half-gigabyte
// array (DRAM and TLB misses in a frame whose caches other kernels have just used) cost more than the two kernel boundaries they save; the
// per-voxel counters of the seven-launch form live in a compact, cache-resident array. A first version that walked the set bits of a
// group and loaded their counters one after the other took 60 us for the prefix pass alone.)
#define FL_VX_NT 256
#define FL_VX_SLOTS 8             /* members a voxel holds in its own slots (vx_rank_kernel); more are chained */
#define FL_VX_L1_SHIFT 8            /* 256 cells (8 bitmap words = one 32-byte line) per group */
#define FL_VX_L2_SHIFT 18           /* 1024 groups per block */
#define FL_VX_L2_MAX 8192           /* 2^31 cells */

struct FlVxCtl {
    unsigned mn_enc[3], mx_enc[3];  // atomicMax of ~u / u, u = the ordered image of the float with the sign bit flipped; 0 = no point yet
    int nfinite;
    int count;                      // voxels = output points
    int leaf_too_small;
    int cells_short;
    unsigned cursor;                // segment allocator
    unsigned pad;
    long long cells;
};

// bounding box of one workgroup's finite points, in the encodings of FlVxCtl (0 = no point). vx_minmax_kernel writes one per workgroup;
// in fl_lidar_front the undistortion's last kernel does (it holds every final point in a register anyway: one launch less);
// vx_claim_kernel folds them -- every workgroup for itself, a few KB from L2.
struct FlVxPartial { unsigned mn[3], mx[3]; int cnt, pad; };
__device__ __forceinline__ unsigned fl_vx_enc(float f) { return (unsigned)fl_ordered_int(f) ^ 0x80000000u; }
__device__ __forceinline__ float fl_vx_dec(unsigned u) { return fl_ordered_float((int)(u ^ 0x80000000u)); }

struct FlVxGrid {
    float inv[3];
    int min_b[3];
    int mul[3];
    int too_small;
    long long cells;
};
__device__ __forceinline__ FlVxGrid fl_vx_grid(const unsigned *mn_enc, const unsigned *mx_enc, float ilx, float ily, float ilz, int n)
{
    FlVxGrid g;
    g.inv[0] = ilx; g.inv[1] = ily; g.inv[2] = ilz;
    int div_b[3];
    long long d[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float mn = fl_vx_dec(~mn_enc[k]), mx = fl_vx_dec(mx_enc[k]);
        d[k] = (long long)((mx - mn) * g.inv[k]) + 1;                 // voxel_grid.hpp: dx, dy, dz
        g.min_b[k] = (int)floorf(mn * g.inv[k]);
        div_b[k] = (int)floorf(mx * g.inv[k]) - g.min_b[k] + 1;
    }
    g.too_small = (d[0] * d[1] * d[2]) > 2147483647ll;
    g.mul[0] = 1; g.mul[1] = div_b[0]; g.mul[2] = div_b[0] * div_b[1];
    // "Leaf size is too small": output = input -- every point its own cell, in cloud order
    g.cells = g.too_small ? (long long)n : (long long)div_b[0] * (long long)div_b[1] * (long long)div_b[2];
    return g;
}

// exclusive prefix over the workgroup (FL_VX_NT threads); total = the sum over all of them
__device__ __forceinline__ unsigned fl_vx_block_scan(unsigned v, unsigned *lds /* FL_VX_NT / 64 + 1 */, unsigned *total)
{
    const int lane = (int)(threadIdx.x & 63u), w = (int)(threadIdx.x >> 6);
    unsigned inc = v;
#pragma unroll
    for (int s = 1; s < 64; s <<= 1) {
        const unsigned o = __shfl_up(inc, s);
        if (lane >= s) inc += o;
    }
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    unsigned base = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < FL_VX_NT / 64; k++) { const unsigned t = lds[k]; if (k < w) base += t; tot += t; }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// the calling workgroup's (256 threads) bounding box: every thread brings the encodings of its own point(s) (0 = none) and its count;
// thread 0 returns the workgroup's record
__device__ __forceinline__ FlVxPartial fl_vx_block_box(unsigned (&mn)[3], unsigned (&mx)[3], int cnt, unsigned (*s_red)[7] /* [4][7] LDS */)
{
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) {
#pragma unroll
        for (int k = 0; k < 3; k++) { mn[k] = max(mn[k], (unsigned)__shfl_xor((int)mn[k], s)); mx[k] = max(mx[k], (unsigned)__shfl_xor((int)mx[k], s)); }
        cnt += __shfl_xor(cnt, s);
    }
    const int w = (int)(threadIdx.x >> 6);
    if ((threadIdx.x & 63u) == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) { s_red[w][k] = mn[k]; s_red[w][3 + k] = mx[k]; }
        s_red[w][6] = (unsigned)cnt;
    }
    __syncthreads();
    FlVxPartial r;
#pragma unroll
    for (int k = 0; k < 3; k++) { r.mn[k] = 0u; r.mx[k] = 0u; }
    r.cnt = 0; r.pad = 0;
    if (threadIdx.x == 0) {
        for (int v = 0; v < 4; v++) {
#pragma unroll
            for (int k = 0; k < 3; k++) { r.mn[k] = max(r.mn[k], s_red[v][k]); r.mx[k] = max(r.mx[k], s_red[v][3 + k]); }
            r.cnt += (int)s_red[v][6];
        }
    }
    return r;
}

__global__ __launch_bounds__(FL_VX_NT) void vx_minmax_kernel(const float4 *__restrict__ in, int n, FlVxPartial *__restrict__ partial)
{
    unsigned mn[3] = {0u, 0u, 0u}, mx[3] = {0u, 0u, 0u};
    int cnt = 0;
    for (int i = blockIdx.x * FL_VX_NT + threadIdx.x; i < n; i += gridDim.x * FL_VX_NT) {
        const float4 p = in[i];
        if (!isfinite(p.x) || !isfinite(p.y) || !isfinite(p.z)) continue;
        const unsigned u[3] = {fl_vx_enc(p.x), fl_vx_enc(p.y), fl_vx_enc(p.z)};
#pragma unroll
        for (int k = 0; k < 3; k++) { mn[k] = max(mn[k], ~u[k]); mx[k] = max(mx[k], u[k]); }
        cnt++;
    }
    __shared__ unsigned s_red[FL_VX_NT / 64][7];
    const FlVxPartial r = fl_vx_block_box(mn, mx, cnt, s_red);
    if (threadIdx.x == 0) partial[blockIdx.x] = r;
}

__global__ __launch_bounds__(FL_VX_NT) void vx_claim_kernel(const float4 *__restrict__ in, int n, FlVxCtl *__restrict__ C, float ilx, float ily,
                                                           float ilz, long long cells_cap, unsigned *__restrict__ keys,
                                                           unsigned *__restrict__ bits, unsigned *__restrict__ l2flag,
                                                           const FlVxPartial *__restrict__ partial, int npartial)
{
    const int i = blockIdx.x * FL_VX_NT + threadIdx.x;
    // the cloud's bounding box from the workgroup records (every workgroup folds them itself: npartial x 32 bytes)
    __shared__ unsigned s_red[FL_VX_NT / 64][7];
    __shared__ unsigned s_box[7];
    {
        unsigned mn[3] = {0u, 0u, 0u}, mx[3] = {0u, 0u, 0u};
        int cnt = 0;
        for (int b = (int)threadIdx.x; b < npartial; b += FL_VX_NT) {
            const FlVxPartial q = partial[b];
#pragma unroll
            for (int k = 0; k < 3; k++) { mn[k] = max(mn[k], q.mn[k]); mx[k] = max(mx[k], q.mx[k]); }
            cnt += q.cnt;
        }
        const FlVxPartial r = fl_vx_block_box(mn, mx, cnt, s_red);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int k = 0; k < 3; k++) { s_box[k] = r.mn[k]; s_box[3 + k] = r.mx[k]; }
            s_box[6] = (unsigned)r.cnt;
        }
        __syncthreads();
    }
    const unsigned mn_enc[3] = {s_box[0], s_box[1], s_box[2]}, mx_enc[3] = {s_box[3], s_box[4], s_box[5]};
    const int nfinite = (int)s_box[6];
    if (i == 0) {
#pragma unroll
        for (int k = 0; k < 3; k++) { C->mn_enc[k] = mn_enc[k]; C->mx_enc[k] = mx_enc[k]; }
        C->nfinite = nfinite;
    }
    if (nfinite <= 0) { if (i < n) keys[i] = 0xFFFFFFFFu; return; }
    const FlVxGrid g = fl_vx_grid(mn_enc, mx_enc, ilx, ily, ilz, n);
    const bool is_short = g.cells > cells_cap;
    if (i == 0) { C->leaf_too_small = g.too_small; C->cells = g.cells; C->cells_short = is_short ? 1 : 0; }
    unsigned key = 0xFFFFFFFFu;
    if (i < n) {
        const float4 p = in[i];
        if (!is_short && isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
            if (g.too_small) {
                key = (unsigned)i;
            } else {
                const int i0 = (int)(floorf(p.x * g.inv[0]) - (float)g.min_b[0]);
                const int i1 = (int)(floorf(p.y * g.inv[1]) - (float)g.min_b[1]);
                const int i2 = (int)(floorf(p.z * g.inv[2]) - (float)g.min_b[2]);
                key = (unsigned)(i0 * g.mul[0] + i1 * g.mul[1] + i2 * g.mul[2]);
            }
        }
    }
    // The occupancy bits. The 100 k fire-and-forget atomics of a scan were 7 of this kernel's 12 us (a launch ends when its atomics have
    // retired). Clouds that arrive in scan-line or voxel order -- a LiDAR driver's, and the camera half's input, which IS the first filter's
    // output -- put runs of consecutive points into one 32-cell word: the lanes of a run OR their bits together (six shuffle steps; the
    // doubling stays inside a run because a lane only takes from a lane with ITS word) and the run's first lane issues one atomic.
    // The same bits either way; a cloud in random order pays thirty instructions per lane for nothing.
    {
        const int lane = (int)(threadIdx.x & 63u);
        const bool live = key != 0xFFFFFFFFu;
        const unsigned w = live ? (key >> 5) : (0xFFFFFFC0u + (unsigned)lane);        // (dead lanes: words of their own, no atomic)
        unsigned bv = live ? (1u << (key & 31u)) : 0u;
        const unsigned wprev = (unsigned)__shfl_up((int)w, 1);
        const bool head = (lane == 0) || (wprev != w);
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned wo = (unsigned)__shfl_down((int)w, d), bo = (unsigned)__shfl_down((int)bv, d);
            if (lane + d < 64 && wo == w) bv |= bo;
        }
        if (live && head) {
            atomicOr(&bits[key >> 5], bv);                   // (result unused: the counts come from popcounts, vx_scan_l2)
            l2flag[key >> FL_VX_L2_SHIFT] = 1u;              // (same value from everybody; a 32-cell word lies inside one 2^18-cell block)
        }
    }
    if (i < n) keys[i] = key;
}

// grid = blocks the bitmap can hold; block b serves cells [b << 18, (b + 1) << 18)
// (group counts = popcounts of the group's eight bitmap words: no counter to raise while claiming, none to clear afterwards. The prefixes
// are written for all 1024 groups of a touched block and read only for groups that hold points of this frame: nothing of them needs clearing.)
__global__ __launch_bounds__(FL_VX_NT) void vx_scan_l2_kernel(const FlVxCtl *__restrict__ C, const unsigned *__restrict__ bits, unsigned *__restrict__ l1pre,
                                                             const unsigned *__restrict__ l2flag, unsigned *__restrict__ l2tot)
{
    __shared__ unsigned s_w[FL_VX_NT / 64 + 1];
    const int b = (int)blockIdx.x;
    if (C->nfinite <= 0 || C->cells_short || ((long long)b << FL_VX_L2_SHIFT) >= C->cells) return;
    if (!l2flag[b]) { if (threadIdx.x == 0) l2tot[b] = 0u; return; }       // (uniform)
    const size_t g0 = ((size_t)b << (FL_VX_L2_SHIFT - FL_VX_L1_SHIFT)) + 4u * threadIdx.x;   // four consecutive groups per thread: 32 words
    const uint4 *w4 = reinterpret_cast<const uint4 *>(bits + (g0 << 3));
    unsigned c[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const uint4 a = w4[2 * k], d = w4[2 * k + 1];
        c[k] = (unsigned)(__popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w) + __popc(d.x) + __popc(d.y) + __popc(d.z) + __popc(d.w));
    }
    unsigned total;
    const unsigned base = fl_vx_block_scan(c[0] + c[1] + c[2] + c[3], s_w, &total);
    reinterpret_cast<uint4 *>(l1pre + g0)[0] = make_uint4(base, base + c[0], base + c[0] + c[1], base + c[0] + c[1] + c[2]);
    if (threadIdx.x == 0) l2tot[b] = total;
}

__global__ __launch_bounds__(FL_VX_NT) void vx_rank_kernel(int n, FlVxCtl *__restrict__ C, const unsigned *__restrict__ keys, const unsigned *__restrict__ bits,
                                                          const unsigned *__restrict__ l1pre, const unsigned *__restrict__ l2tot,
                                                          unsigned *__restrict__ slots, unsigned *__restrict__ ohead, unsigned *__restrict__ onext,
                                                          unsigned *__restrict__ cnt)
{
    __shared__ unsigned s_l2pre[FL_VX_L2_MAX];
    __shared__ unsigned s_w[FL_VX_NT / 64 + 1];
    if (C->nfinite <= 0 || C->cells_short) { if (blockIdx.x == 0 && threadIdx.x == 0) C->count = 0; return; }
    const int nl2 = (int)((C->cells + (1ll << FL_VX_L2_SHIFT) - 1) >> FL_VX_L2_SHIFT);
    unsigned run = 0;
    for (int b0 = 0; b0 < nl2; b0 += FL_VX_NT) {               // exclusive prefix over the block totals (nl2 <= 8192; a handful for a scan)
        const int b = b0 + (int)threadIdx.x;
        const unsigned v = b < nl2 ? l2tot[b] : 0u;
        unsigned total;
        const unsigned ex = fl_vx_block_scan(v, s_w, &total);
        if (b < nl2) s_l2pre[b] = run + ex;
        run += total;
    }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) C->count = (int)run;
    const int i = blockIdx.x * FL_VX_NT + threadIdx.x;
    if (i >= n) return;
    const unsigned key = keys[i];
    if (key == 0xFFFFFFFFu) return;
    const unsigned grp = key >> FL_VX_L1_SHIFT;
    const uint4 *w4 = reinterpret_cast<const uint4 *>(bits + ((size_t)grp << 3));
    const uint4 a = w4[0], b = w4[1];
    const unsigned wv[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    const unsigned wi = (key >> 5) & 7u;
    unsigned below = 0;
#pragma unroll
    for (unsigned k = 0; k < 8; k++) {
        const unsigned m = (k < wi) ? 0xFFFFFFFFu : ((k == wi) ? ((1u << (key & 31u)) - 1u) : 0u);
        below += (unsigned)__popc(wv[k] & m);
    }
    const unsigned o = s_l2pre[key >> FL_VX_L2_SHIFT] + l1pre[grp] + below;
    // Round 6 (second half): a voxel's first FL_VX_SLOTS members go straight into the voxel's own slots, in the order the atomic hands the
    // positions out; what comes later -- rare in a scan -- is chained behind the voxel (ohead / onext). vx_centroid_kernel puts either in
    // ascending point index. (Until then: position + voxel per point, a prefix over the voxels' counts for segment starts, a scatter of
    // the point indices into the segments -- vx_segments_kernel and vx_scatter_kernel, two launches of their own.)
    const unsigned p = atomicAdd(&cnt[o], 1u);
    if (p < (unsigned)FL_VX_SLOTS) slots[(size_t)o * FL_VX_SLOTS + p] = (unsigned)i;
    else onext[i] = atomicExch(&ohead[o], (unsigned)i);
}

// out: centroids (x, y, z, intensity); body (nullable): xyz only, the staged-scan layout. The launch covers max(points, voxels)
// threads: thread t serves voxel t and cleans up behind point t. `next`: the control block of the NEXT frame, zeroed here.
// Round 6: the members of a voxel arrive in its slots (and, beyond eight, in its chain) in the order vx_rank_kernel's atomics gave them;
// PCL sums them in ascending point index (the stable sort of downSizeFilter). A voxel of <= 8 members -- nearly all of them -- is put in
// order in its thread's registers (a 19-comparator network over 8 slots padded with 0xFFFFFFFF) and summed; the rest are left to the
// workgroup behind a barrier: every thread ranks a member among the voxel's others, a thread per voxel sums in order. Same order, same
// sums as the three kernels this replaces (per-voxel segments by a prefix over the counts, a scatter of the point indices, a launch
// that ranked every point within its voxel): the sort-free filter is 4 launches (claim, prefix, rank, centroid), it was 7.
#define FL_VX_CE(a, b) do { const unsigned lo_ = min(a, b), hi_ = max(a, b); a = lo_; b = hi_; } while (0)
__global__ __launch_bounds__(FL_VX_NT) void vx_centroid_kernel(const float4 *__restrict__ in, int n, FlVxCtl *__restrict__ C,
                                                              const unsigned *__restrict__ keys, unsigned *__restrict__ cnt,
                                                              const unsigned *__restrict__ slots, unsigned *__restrict__ ohead,
                                                              const unsigned *__restrict__ onext, unsigned *__restrict__ scratch,
                                                              unsigned *__restrict__ ordered /* both: the voxels with > 8 members */,
                                                              float4 *__restrict__ out, float *__restrict__ body, unsigned *__restrict__ bits,
                                                              unsigned *__restrict__ l2flag, FlVxCtl *__restrict__ next,
                                                              FlFrontTail *__restrict__ tail = nullptr)
{
    static_assert(FL_VX_SLOTS == 8, "the sorting network below has eight inputs");
    __shared__ int s_big[FL_VX_NT];
    __shared__ int s_nbig;
    const int t = blockIdx.x * FL_VX_NT + threadIdx.x;
    if (threadIdx.x == 0) s_nbig = 0;
    if (t == 0) {
        FlVxCtl z; memset(&z, 0, sizeof z); *next = z;
        if (tail) {      // fl_lidar_front: what the host wants to know of the filter rides back behind the state block
            tail->vox_count = C->count; tail->vox_leaf_too_small = C->leaf_too_small; tail->vox_cells_short = C->cells_short;
            tail->vox_nfinite = C->nfinite; tail->vox_cells = C->cells;
        }
    }
    if (C->nfinite <= 0 || C->cells_short) return;           // (uniform over the grid)
    const int nvox = C->count;
    __syncthreads();
    if (t < nvox) {
        const unsigned c = cnt[t];
        if (c <= 8u) {
            const uint4 sa = reinterpret_cast<const uint4 *>(slots + (size_t)t * FL_VX_SLOTS)[0], sb = reinterpret_cast<const uint4 *>(slots + (size_t)t * FL_VX_SLOTS)[1];
            unsigned m0 = c > 0u ? sa.x : 0xFFFFFFFFu, m1 = c > 1u ? sa.y : 0xFFFFFFFFu, m2 = c > 2u ? sa.z : 0xFFFFFFFFu, m3 = c > 3u ? sa.w : 0xFFFFFFFFu;
            unsigned m4 = c > 4u ? sb.x : 0xFFFFFFFFu, m5 = c > 5u ? sb.y : 0xFFFFFFFFu, m6 = c > 6u ? sb.z : 0xFFFFFFFFu, m7 = c > 7u ? sb.w : 0xFFFFFFFFu;
            if (c > 1u) {
                FL_VX_CE(m0, m1); FL_VX_CE(m2, m3); FL_VX_CE(m4, m5); FL_VX_CE(m6, m7);
                FL_VX_CE(m0, m2); FL_VX_CE(m1, m3); FL_VX_CE(m4, m6); FL_VX_CE(m5, m7);
                FL_VX_CE(m1, m2); FL_VX_CE(m5, m6); FL_VX_CE(m0, m4); FL_VX_CE(m3, m7);
                FL_VX_CE(m1, m5); FL_VX_CE(m2, m6);
                FL_VX_CE(m1, m4); FL_VX_CE(m3, m6);
                FL_VX_CE(m2, m4); FL_VX_CE(m3, m5);
                FL_VX_CE(m3, m4);
            }
            const unsigned mm[8] = {m0, m1, m2, m3, m4, m5, m6, m7};
            float4 p[8];
#pragma unroll
            for (int k = 0; k < 8; k++) p[k] = ((unsigned)k < c) ? in[mm[k]] : make_float4(0.f, 0.f, 0.f, 0.f);      // (in flight together)
            float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
#pragma unroll
            for (int k = 0; k < 8; k++)
                if ((unsigned)k < c) { sx = sx + p[k].x; sy = sy + p[k].y; sz = sz + p[k].z; si = si + p[k].w; }
            const float fc = (float)c;
            const float4 ce = make_float4(sx / fc, sy / fc, sz / fc, si / fc);
            out[t] = ce;
            if (body) { body[3 * t] = ce.x; body[3 * t + 1] = ce.y; body[3 * t + 2] = ce.z; }
            cnt[t] = 0u;
        } else {
            s_big[atomicAdd(&s_nbig, 1)] = t;
        }
    }
    __syncthreads();
    const int nbig = s_nbig;                                  // (uniform over the workgroup)
    if (nbig > 0) {
        // the workgroup's voxels with more than 8 members. Thread b takes voxel b: a segment of the scratch array (one atomic on the
        // frame's cursor), its eight slots and its chain copied there. Then the ranking, spread over (voxel, member) pairs -- every thread
        // ranks members among their voxel's others, c comparisons each, as vx_order_kernel did for every point of the scan --, then
        // thread b sums voxel b in order (the sums of different voxels run side by side).
        __shared__ unsigned s_off[FL_VX_NT + 1], s_seg[FL_VX_NT], s_scan[FL_VX_NT / 64 + 1];
        const bool mine = (int)threadIdx.x < nbig;
        const int v = mine ? s_big[threadIdx.x] : 0;
        const unsigned vc = mine ? cnt[v] : 0u;
        unsigned vs = 0u;
        if (mine) {
            vs = atomicAdd(&C->cursor, vc);
            const uint4 sa = reinterpret_cast<const uint4 *>(slots + (size_t)v * FL_VX_SLOTS)[0], sb = reinterpret_cast<const uint4 *>(slots + (size_t)v * FL_VX_SLOTS)[1];
            scratch[vs] = sa.x; scratch[vs + 1] = sa.y; scratch[vs + 2] = sa.z; scratch[vs + 3] = sa.w;
            scratch[vs + 4] = sb.x; scratch[vs + 5] = sb.y; scratch[vs + 6] = sb.z; scratch[vs + 7] = sb.w;
            unsigned idx = ohead[v];
            for (unsigned k = FL_VX_SLOTS; k < vc && idx != 0xFFFFFFFFu; k++) { scratch[vs + k] = idx; idx = onext[idx]; }
            ohead[v] = 0xFFFFFFFFu;
            s_seg[threadIdx.x] = vs;
        }
        unsigned total = 0;
        const unsigned off = fl_vx_block_scan(vc, s_scan, &total);
        s_off[threadIdx.x] = off;
        if (threadIdx.x == 0) s_off[FL_VX_NT] = total;
        __threadfence_block();
        __syncthreads();
        for (unsigned w = threadIdx.x; w < total; w += FL_VX_NT) {
            int lo = 0, hi = nbig - 1;                        // the voxel of pair w: the last b with s_off[b] <= w
            while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_off[mid] <= w) lo = mid; else hi = mid - 1; }
            const unsigned s = s_seg[lo], c = s_off[lo + 1] - s_off[lo], j = w - s_off[lo];
            const unsigned mj = scratch[s + j];
            unsigned r = 0;
            for (unsigned k = 0; k < c; k++) r += (scratch[s + k] < mj) ? 1u : 0u;
            ordered[s + r] = mj;
        }
        __threadfence_block();
        __syncthreads();
        if (mine) {
            float sx = 0.f, sy = 0.f, sz = 0.f, si = 0.f;
            for (unsigned k = 0; k < vc; k++) {
                const float4 p = in[ordered[vs + k]];
                sx = sx + p.x; sy = sy + p.y; sz = sz + p.z; si = si + p.w;
            }
            const float fc = (float)vc;
            const float4 ce = make_float4(sx / fc, sy / fc, sz / fc, si / fc);
            out[v] = ce;
            if (body) { body[3 * v] = ce.x; body[3 * v + 1] = ce.y; body[3 * v + 2] = ce.z; }
            cnt[v] = 0u;
        }
    }
    if (t < n) {
        const unsigned key = keys[t];
        if (key != 0xFFFFFFFFu) { bits[key >> 5] = 0u; l2flag[key >> FL_VX_L2_SHIFT] = 0u; }
    }
}
