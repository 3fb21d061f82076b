/*
 * oracle/orc_voxel.c -- TEST INFRASTRUCTURE ONLY (see fastlivo_oracle.h).
 *
 * CPU restatement of pcl::VoxelGrid<PointT>::applyFilter as FAST-LIVO uses it:
 *   downSizeFilterSurf on the undistorted scan  src/laserMapping.cpp:1186 (setLeafSize), :1398-1399 (filter)
 *   downSizeFilter (leaf 0.2) on the world scan  src/lidar_selection.cpp:7, :352-353
 * PCL is a third-party dependency that is NOT under /root/reference (README.md: "PCL >= 1.8", unpinned); the
 * algorithm below is restated from the published source pcl/filters/impl/voxel_grid.hpp (identical in 1.8 .. 1.12
 * for clouds without a filter field, downsample_all_data_ = true, min_points_per_voxel_ = 0):
 *   1. getMinMax3D over the finite points;
 *   2. dx,dy,dz = (int64)((max-min)*inverse_leaf)+1 ; if dx*dy*dz > INT32_MAX: warn, output = input, return;
 *   3. min_b = (int)floor(min*inverse_leaf), max_b likewise, div_b = max_b-min_b+1, divb_mul = (1, div_b0, div_b0*div_b1);
 *   4. per finite point: ijk = (int)(floor(p*inverse_leaf) - (float)min_b)  [float arithmetic], idx = ijk . divb_mul;
 *   5. std::sort of (idx, cloud index) by idx;
 *   6. one output point per run of equal idx, in ascending idx order: the centroid, every field accumulated in
 *      float and divided by (float)count (CentroidPoint / NdCopyPointEigenFunctor).
 * PARITY UNPINNED, and one rule is OURS: std::sort is not stable, so PCL's summation order inside a voxel is
 * unspecified; here (and on the device) points are accumulated in ascending cloud index. Fields carried: x, y, z
 * and one scalar channel (intensity); the normal/curvature fields of PointXYZINormal are not read downstream.
 */
#include "fastlivo_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { unsigned idx; int pt; } vox_pair;

static int vox_cmp(const void *a, const void *b)
{
    const vox_pair *p = (const vox_pair *)a, *q = (const vox_pair *)b;
    if (p->idx != q->idx) return p->idx < q->idx ? -1 : 1;
    return (p->pt > q->pt) - (p->pt < q->pt);
}

int orc_voxel_grid(const float *xyzi, int n, float leaf_x, float leaf_y, float leaf_z, float *out_xyzi, int32_t *out_n,
                   int32_t *leaf_too_small)
{
    const float inv[3] = {1.0f / leaf_x, 1.0f / leaf_y, 1.0f / leaf_z};
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    int nfinite = 0;
    *out_n = 0;
    *leaf_too_small = 0;
    for (int i = 0; i < n; i++) {
        const float *p = xyzi + 4 * (size_t)i;
        if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
        for (int k = 0; k < 3; k++) { if (p[k] < mn[k]) mn[k] = p[k]; if (p[k] > mx[k]) mx[k] = p[k]; }
        nfinite++;
    }
    if (nfinite == 0) return 0;
    const int64_t dx = (int64_t)((mx[0] - mn[0]) * inv[0]) + 1, dy = (int64_t)((mx[1] - mn[1]) * inv[1]) + 1,
                  dz = (int64_t)((mx[2] - mn[2]) * inv[2]) + 1;
    if (dx * dy * dz > (int64_t)INT32_MAX) {          /* "Leaf size is too small for the input dataset" */
        memcpy(out_xyzi, xyzi, sizeof(float) * 4 * (size_t)n);
        *out_n = n;
        *leaf_too_small = 1;
        return 0;
    }
    int min_b[3], max_b[3], div_b[3], mul[3];
    for (int k = 0; k < 3; k++) {
        min_b[k] = (int)floorf(mn[k] * inv[k]);
        max_b[k] = (int)floorf(mx[k] * inv[k]);
        div_b[k] = max_b[k] - min_b[k] + 1;
    }
    mul[0] = 1; mul[1] = div_b[0]; mul[2] = div_b[0] * div_b[1];
    vox_pair *v = (vox_pair *)malloc(sizeof(vox_pair) * (size_t)nfinite);
    if (!v) return -1;
    int m = 0;
    for (int i = 0; i < n; i++) {
        const float *p = xyzi + 4 * (size_t)i;
        if (!isfinite(p[0]) || !isfinite(p[1]) || !isfinite(p[2])) continue;
        const int i0 = (int)(floorf(p[0] * inv[0]) - (float)min_b[0]);
        const int i1 = (int)(floorf(p[1] * inv[1]) - (float)min_b[1]);
        const int i2 = (int)(floorf(p[2] * inv[2]) - (float)min_b[2]);
        v[m].idx = (unsigned)(i0 * mul[0] + i1 * mul[1] + i2 * mul[2]);
        v[m].pt = i;
        m++;
    }
    qsort(v, (size_t)m, sizeof(vox_pair), vox_cmp);
    int o = 0;
    for (int a = 0; a < m;) {
        int b = a;
        float s[4] = {0.f, 0.f, 0.f, 0.f};
        while (b < m && v[b].idx == v[a].idx) {
            const float *p = xyzi + 4 * (size_t)v[b].pt;
            for (int k = 0; k < 4; k++) s[k] = s[k] + p[k];
            b++;
        }
        const float cnt = (float)(b - a);
        for (int k = 0; k < 4; k++) out_xyzi[4 * (size_t)o + k] = s[k] / cnt;
        o++;
        a = b;
    }
    free(v);
    *out_n = o;
    return 0;
}
