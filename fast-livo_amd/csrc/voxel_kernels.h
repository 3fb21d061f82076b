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
