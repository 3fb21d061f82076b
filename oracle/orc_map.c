/*
 * oracle/orc_map.c -- TEST INFRASTRUCTURE ONLY (CPU oracle). Never linked into the product.
 *
 * The map side of the ikd-Tree as the reference drives it, restated on a flat point array by brute force, one new point
 * after the other exactly like the loop it follows:
 *   orc_map_add_points    KD_TREE::Add_Points(PointToAdd, downsample_on)   /root/reference/include/ikd-Tree/ikd_Tree.cpp:382-457
 *                         box + centre :392-400, Search_by_range membership :988-1000 (min <= v && max > v),
 *                         strict `tmp_dist < min_dist` :404-409, the size()>1 || same_point rule :411-415, EPSS ikd_Tree.h:12,
 *                         calc_dist :1291-1295; caller map_incremental /root/reference/src/laserMapping.cpp:692-706
 *   orc_map_delete_boxes  KD_TREE::Delete_Point_Boxes :501-520 -> Delete_by_range point test :650; caller lasermap_fov_segment
 *                         laserMapping.cpp:363-417
 *   orc_fov_segment       lasermap_fov_segment's window logic itself (laserMapping.cpp:363-417; MOV_THRESHOLD 1.5f :90, DET_RANGE :83,
 *                         cube_len :118,1118)
 * The tree's internals (balancing, lazy deletion, the rebuild thread) do not change WHICH points the map holds, only the order
 * Search_by_range reports them in; that order decides between old points of one box that are exactly equally far from its
 * centre. Not reproducible without the tree: the array order is used (lower index first), as in orc_knn.c. A point that wins
 * its box again keeps its place in the array; added points are appended. PINNED to the reference's own ikd-Tree (oracle/ref_ikdtree);
 * orc_fov_segment also to lasermap_fov_segment's text (oracle/ref_eigen, tests/test_ref_eigen_cpu.py).
 */
#include "fastlivo_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float mn[3], mx[3], mid[3]; } box_t;

static void box_of_point(const float *p, float ds, box_t *b)
{
    for (int k = 0; k < 3; k++) {
        b->mn[k] = floorf(p[k] / ds) * ds;
        b->mx[k] = b->mn[k] + ds;
        b->mid[k] = (float)((double)b->mn[k] + (double)(b->mx[k] - b->mn[k]) / 2.0);
    }
}
static int in_box(const float *p, const float *mn, const float *mx)
{
    return mn[0] <= p[0] && mx[0] > p[0] && mn[1] <= p[1] && mx[1] > p[1] && mn[2] <= p[2] && mx[2] > p[2];
}
static float calc_dist(const float *a, const float *b)
{
    return (a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]);
}
static int same_point(const float *a, const float *b)
{
    return fabs(a[0] - b[0]) < 1e-6 && fabs(a[1] - b[1]) < 1e-6 && fabs(a[2] - b[2]) < 1e-6;      /* EPSS is a double literal */
}
/* is the membership of coordinate v rounding-dependent? (own box by floor() vs the coordinate test, and the two neighbours) */
static int axis_ambiguous(float v, float ds)
{
    const float f = floorf(v / ds);
    const float mn = f * ds, mx = mn + ds;
    const float lo_max = (f - 1.0f) * ds + ds, hi_min = (f + 1.0f) * ds;
    return !(mn <= v && mx > v) || lo_max > v || hi_min <= v;
}

int orc_map_add_points(const float *map_xyz, int n_map, const float *new_xyz, int n_new, float downsample_size, float *out_xyz /* (n_map+n_new) x 3 */,
                       orc_map_info *info)
{
    const int cap = n_map + n_new;
    float *pts = (float *)malloc(sizeof(float) * 3 * (size_t)(cap > 0 ? cap : 1));
    unsigned char *alive = (unsigned char *)calloc((size_t)(cap > 0 ? cap : 1), 1);
    int *storage = (int *)malloc(sizeof(int) * (size_t)(cap > 0 ? cap : 1));
    if (!pts || !alive || !storage) { free(pts); free(alive); free(storage); return -1; }
    memcpy(pts, map_xyz, sizeof(float) * 3 * (size_t)n_map);
    for (int i = 0; i < n_map; i++) alive[i] = 1;
    int cnt = n_map, amb = 0;
    if (downsample_size > 0.f) {
        for (int i = 0; i < n_map; i++) amb += axis_ambiguous(pts[i * 3], downsample_size) || axis_ambiguous(pts[i * 3 + 1], downsample_size) ||
                                               axis_ambiguous(pts[i * 3 + 2], downsample_size);
    }
    for (int j = 0; j < n_new; j++) {
        const float *p = new_xyz + (size_t)j * 3;
        if (!(downsample_size > 0.f)) {                 /* downsample_on == false: Add_by_point */
            memcpy(pts + (size_t)cnt * 3, p, sizeof(float) * 3); alive[cnt++] = 1;
            continue;
        }
        amb += axis_ambiguous(p[0], downsample_size) || axis_ambiguous(p[1], downsample_size) || axis_ambiguous(p[2], downsample_size);
        box_t b;
        box_of_point(p, downsample_size, &b);
        int ns = 0;
        for (int i = 0; i < cnt; i++) {
            const float *q = pts + (size_t)i * 3;
            if (!(b.mn[0] <= q[0] && b.mx[0] > q[0])) continue;
            if (alive[i] && in_box(q, b.mn, b.mx)) storage[ns++] = i;      /* Search_by_range */
        }
        float min_dist = calc_dist(p, b.mid);
        int result = -1;                                 /* -1: PointToAdd[i] itself */
        for (int s = 0; s < ns; s++) {
            const float tmp = calc_dist(pts + (size_t)storage[s] * 3, b.mid);
            if (tmp < min_dist) { min_dist = tmp; result = storage[s]; }
        }
        const float *res = result < 0 ? p : pts + (size_t)result * 3;
        if (ns > 1 || same_point(p, res)) {
            for (int s = 0; s < ns; s++) alive[storage[s]] = 0;            /* Delete_by_range(box) */
            if (result >= 0) alive[result] = 1;                            /* Add_by_point(downsample_result): the same point again */
            else { memcpy(pts + (size_t)cnt * 3, p, sizeof(float) * 3); alive[cnt++] = 1; }
        }
    }
    int m = 0, kept_old = 0;
    for (int i = 0; i < cnt; i++)
        if (alive[i]) {
            memcpy(out_xyz + (size_t)m * 3, pts + (size_t)i * 3, sizeof(float) * 3); m++;
            if (i < n_map) kept_old++;
        }
    if (info) {
        info->n_before = n_map; info->n_after = m; info->n_added = m - kept_old; info->n_removed = n_map - kept_old; info->n_ambiguous = amb;
    }
    free(pts); free(alive); free(storage);
    return 0;
}

int orc_map_delete_boxes(const float *map_xyz, int n_map, const float *boxes /* nb x 6 */, int nb, float *out_xyz, orc_map_info *info)
{
    int m = 0;
    for (int i = 0; i < n_map; i++) {
        const float *q = map_xyz + (size_t)i * 3;
        int gone = 0;
        for (int b = 0; b < nb && !gone; b++) gone = in_box(q, boxes + b * 6, boxes + b * 6 + 3);
        if (!gone) { memcpy(out_xyz + (size_t)m * 3, q, sizeof(float) * 3); m++; }
    }
    if (info) { info->n_before = n_map; info->n_after = m; info->n_added = 0; info->n_removed = n_map - m; info->n_ambiguous = 0; }
    return 0;
}

/* lasermap_fov_segment (laserMapping.cpp:363-417): the local-map window follows the LiDAR; returns the number of boxes to delete
 * (0..3) in boxes_out (x 6 floats). win: vertex_min[3], vertex_max[3] (BoxPointType floats), *initialized as Localmap_Initialized. */
int orc_fov_segment(float *win /* 6, in/out */, int *initialized, const double *pos_lid /* 3 */, double cube_len, float det_range,
                    float mov_threshold, float *boxes_out /* 3 x 6 */)
{
    float *vmin = win, *vmax = win + 3;
    if (!*initialized) {
        for (int i = 0; i < 3; i++) {
            vmin[i] = (float)(pos_lid[i] - cube_len / 2.0);
            vmax[i] = (float)(pos_lid[i] + cube_len / 2.0);
        }
        *initialized = 1;
        return 0;
    }
    float dist_to_map_edge[3][2];
    int need_move = 0;
    for (int i = 0; i < 3; i++) {
        dist_to_map_edge[i][0] = (float)fabs(pos_lid[i] - (double)vmin[i]);
        dist_to_map_edge[i][1] = (float)fabs(pos_lid[i] - (double)vmax[i]);
        if (dist_to_map_edge[i][0] <= mov_threshold * det_range || dist_to_map_edge[i][1] <= mov_threshold * det_range) need_move = 1;
    }
    if (!need_move) return 0;
    float nmin[3], nmax[3];
    memcpy(nmin, vmin, sizeof nmin); memcpy(nmax, vmax, sizeof nmax);
    const double a = (cube_len - 2.0 * (double)mov_threshold * (double)det_range) * 0.5 * 0.9, b = (double)(det_range * (mov_threshold - 1));
    const float mov_dist = (float)(a > b ? a : b);
    int nb = 0;
    for (int i = 0; i < 3; i++) {
        float tmin[3], tmax[3];
        memcpy(tmin, vmin, sizeof tmin); memcpy(tmax, vmax, sizeof tmax);
        if (dist_to_map_edge[i][0] <= mov_threshold * det_range) {
            nmax[i] -= mov_dist; nmin[i] -= mov_dist;
            tmin[i] = vmax[i] - mov_dist;
        } else if (dist_to_map_edge[i][1] <= mov_threshold * det_range) {
            nmax[i] += mov_dist; nmin[i] += mov_dist;
            tmax[i] = vmin[i] + mov_dist;
        } else continue;
        memcpy(boxes_out + nb * 6, tmin, sizeof tmin); memcpy(boxes_out + nb * 6 + 3, tmax, sizeof tmax);
        nb++;
    }
    memcpy(vmin, nmin, sizeof nmin); memcpy(vmax, nmax, sizeof nmax);
    return nb;
}
