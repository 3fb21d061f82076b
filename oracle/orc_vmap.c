/*
 * oracle/orc_vmap.c -- TEST INFRASTRUCTURE ONLY (see fastlivo_oracle.h).
 *
 * The visual map of LidarSelector and the three per-frame steps that touch it, restated on flat arrays
 * (/root/reference/src/lidar_selection.cpp, /root/reference/src/point.cpp):
 *   orc_vmap_add_sparse       addSparseMap :142-197 + AddPoint :199-230 (voxel key rule :203-212)
 *   orc_vmap_select           addFromSparseMap :346-587: sub_feat_map keys :383-391, depth image :393-409, voxel walk +
 *                             grid competition :412-466, per winning cell getCloseViewObs (point.cpp:141-178) and the
 *                             pixel-level part (orc_select.c)
 *   orc_vmap_add_observation  addObservation :913-965 with getFurthestViewObs (point.cpp:219-247), deleteFeatureRef :88-98,
 *                             addFrameRef :61-65 (push_front)
 * Call order within a frame as in detect() :1050-1064: select -> add_sparse -> (ComputeJ) -> add_observation; map_value
 * is NOT reset between select and add_sparse (reset_grid :81-90 leaves it alone), so a scan point only founds a new map
 * point where it scores higher than every map point that projected into its grid cell.
 * A point's observations are kept in list order (obs[0] = front of the std::list, the newest: addFrameRef pushes to the
 * front); `obs_.back()` of :929 is therefore the OLDEST observation.
 * Not reproducible: the iteration order of the two unordered_maps (:412, feat_map buckets) -- it only decides between map
 * points at exactly the same float distance from the camera in one grid cell; here points are visited in the order they
 * entered the map. The reference's own lines are held to their text since round 4 (oracle/ref_eigen,
 * tests/test_ref_eigen_cpu.py: a 36-frame walk, bit for bit).  Third-party arithmetic restated from the published sources (unpinned):
 *   vk::shiTomasiScore (rpg_vikit vision.cpp): 8x8 box of central differences, (dXX, dYY, dXY) / (2 * 64), smaller eigenvalue;
 *   Sophus::SE3 products / inverse stated with rotation matrices.
 */
#include "fastlivo_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct vobs { double px[2], f[3], R[9], t[3]; float score; int32_t level, kf_id, frame_id; } vobs;
typedef struct vpoint { double pos[3]; float value; int32_t n_obs; int64_t key[3]; vobs obs[ORC_VMAP_MAX_OBS]; } vpoint;
struct orc_vmap {
    orc_vio_config cfg;
    int grid_size, gw, gh, length;
    float *map_value, *map_dist;
    int32_t *grid_num, *winner;
    vpoint *pts;
    int n, cap;
};

static int in_frame(int u, int v, int boundary, int w, int h) { return u >= boundary && u < w - boundary && v >= boundary && v < h - boundary; }
static void mv(const double *A, const double *x, double *o)
{
    double t[3];
    for (int i = 0; i < 3; i++) t[i] = A[i * 3] * x[0] + A[i * 3 + 1] * x[1] + A[i * 3 + 2] * x[2];
    memcpy(o, t, sizeof t);
}
static void w2f(const double *R, const double *t, const double *p, double *o) { mv(R, p, o); o[0] += t[0]; o[1] += t[1]; o[2] += t[2]; }
static void cam2world(const orc_vio_config *c, double u, double v, double *f) { orc_cam2world(c, u, v, f); }
/* T_f_w.inverse().translation() = -R^T t */
static void frame_pos(const double *R, const double *t, double *o)
{
    for (int i = 0; i < 3; i++) o[i] = -(R[i] * t[0] + R[3 + i] * t[1] + R[6 + i] * t[2]);
}

float orc_shi_tomasi(const uint8_t *img, int width, int height, int u, int v)
{
    float dXX = 0.0f, dYY = 0.0f, dXY = 0.0f;
    const int halfbox = 4, box = 8, area = 64;
    const int x_min = u - halfbox, x_max = u + halfbox, y_min = v - halfbox, y_max = v + halfbox;
    if (x_min < 1 || x_max >= width - 1 || y_min < 1 || y_max >= height - 1) return 0.0f;
    for (int y = y_min; y < y_max; y++)
        for (int x = 0; x < box; x++) {
            const float dx = (float)((int)img[width * y + x_min + 1 + x] - (int)img[width * y + x_min - 1 + x]);
            const float dy = (float)((int)img[width * (y + 1) + x_min + x] - (int)img[width * (y - 1) + x_min + x]);
            dXX += dx * dx; dYY += dy * dy; dXY += dx * dy;
        }
    dXX = (float)((double)dXX / (2.0 * area));
    dYY = (float)((double)dYY / (2.0 * area));
    dXY = (float)((double)dXY / (2.0 * area));
    const float s = dXX + dYY;
    return (float)(0.5 * (double)(s - sqrtf(s * s - 4 * (dXX * dYY - dXY * dXY))));
}

orc_vmap *orc_vmap_create(const orc_vio_config *cfg, int grid_size)
{
    orc_vmap *m = (orc_vmap *)calloc(1, sizeof *m);
    m->cfg = *cfg; m->grid_size = grid_size;
    m->gw = cfg->width / grid_size; m->gh = cfg->height / grid_size; m->length = m->gw * m->gh;
    m->map_value = (float *)calloc((size_t)m->length, sizeof(float));
    m->map_dist = (float *)calloc((size_t)m->length, sizeof(float));
    m->grid_num = (int32_t *)calloc((size_t)m->length, sizeof(int32_t));
    m->winner = (int32_t *)calloc((size_t)m->length, sizeof(int32_t));
    return m;
}
void orc_vmap_destroy(orc_vmap *m) { if (!m) return; free(m->map_value); free(m->map_dist); free(m->grid_num); free(m->winner); free(m->pts); free(m); }
int orc_vmap_size(const orc_vmap *m) { return m->n; }
int orc_vmap_get_point(const orc_vmap *m, int i, double *pos, float *value, int32_t *n_obs, orc_vmap_obs *obs /* ORC_VMAP_MAX_OBS */)
{
    if (i < 0 || i >= m->n) return -1;
    const vpoint *p = &m->pts[i];
    memcpy(pos, p->pos, sizeof p->pos); *value = p->value; *n_obs = p->n_obs;
    for (int k = 0; k < p->n_obs; k++) {
        memcpy(obs[k].px, p->obs[k].px, sizeof obs[k].px); memcpy(obs[k].f, p->obs[k].f, sizeof obs[k].f);
        memcpy(obs[k].R, p->obs[k].R, sizeof obs[k].R); memcpy(obs[k].t, p->obs[k].t, sizeof obs[k].t);
        obs[k].score = p->obs[k].score; obs[k].level = p->obs[k].level; obs[k].kf_id = p->obs[k].kf_id; obs[k].frame_id = p->obs[k].frame_id;
    }
    return 0;
}
void orc_vmap_get_grid(const orc_vmap *m, float *map_value, int32_t *grid_num)
{
    memcpy(map_value, m->map_value, sizeof(float) * (size_t)m->length);
    memcpy(grid_num, m->grid_num, sizeof(int32_t) * (size_t)m->length);
}

static void reset_grid(orc_vmap *m)                           /* :81-90 */
{
    for (int i = 0; i < m->length; i++) { m->grid_num[i] = 3; m->map_dist[i] = 10000.f; m->winner[i] = -1; }
}

/* addSparseMap + AddPoint */
int orc_vmap_add_sparse(orc_vmap *m, const double *Rcw, const double *Pcw, const uint8_t *img, const float *scan_world_xyz, int n,
                        int kf_id, int frame_id)
{
    const orc_vio_config *c = &m->cfg;
    const int W = c->width, H = c->height, half = c->patch_size / 2;
    reset_grid(m);
    for (int i = 0; i < n; i++) {
        const double pt[3] = {scan_world_xyz[3 * i], scan_world_xyz[3 * i + 1], scan_world_xyz[3 * i + 2]};
        double pc3[3], pc[2];
        w2f(Rcw, Pcw, pt, pc3);
        orc_world2cam(c, pc3, pc);                                                       /* no test of the depth sign (:153) */
        if (!in_frame((int)pc[0], (int)pc[1], (half + 1) * 8, W, H)) continue;
        const int index = (int)(pc[0] / m->grid_size) * m->gh + (int)(pc[1] / m->grid_size);
        if (index < 0 || index >= m->length) continue;
        const float cur_value = orc_shi_tomasi(img, W, H, (int)pc[0], (int)pc[1]);
        if (cur_value > m->map_value[index]) { m->map_value[index] = cur_value; m->winner[index] = i; m->grid_num[index] = 2; }
    }
    int add = 0;
    for (int g = 0; g < m->length; g++) {
        if (m->grid_num[g] != 2) continue;
        const int i = m->winner[g];
        const double pt[3] = {scan_world_xyz[3 * i], scan_world_xyz[3 * i + 1], scan_world_xyz[3 * i + 2]};
        double pc3[3], pc[2];
        w2f(Rcw, Pcw, pt, pc3);
        orc_world2cam(c, pc3, pc);
        if (m->n == m->cap) { m->cap = m->cap ? 2 * m->cap : 1024; m->pts = (vpoint *)realloc(m->pts, sizeof(vpoint) * (size_t)m->cap); }
        vpoint *p = &m->pts[m->n++];
        memset(p, 0, sizeof *p);
        memcpy(p->pos, pt, sizeof pt);
        p->value = m->map_value[g];
        p->n_obs = 1;
        vobs *o = &p->obs[0];
        o->px[0] = pc[0]; o->px[1] = pc[1];
        cam2world(c, pc[0], pc[1], o->f);
        memcpy(o->R, Rcw, sizeof o->R); memcpy(o->t, Pcw, sizeof o->t);
        o->score = m->map_value[g]; o->level = 0; o->kf_id = kf_id; o->frame_id = frame_id;
        for (int j = 0; j < 3; j++) {                                                     /* AddPoint :203-212 */
            float loc = (float)(pt[j] / 0.5);
            if (loc < 0) loc -= 1.0f;
            p->key[j] = (int64_t)loc;
        }
        add++;
    }
    return add;
}

/* Point::getCloseViewObs: index of the chosen observation or -1 */
static int close_view_obs(const vpoint *p, const double *fpos)
{
    if (p->n_obs <= 0) return -1;
    double od[3] = {fpos[0] - p->pos[0], fpos[1] - p->pos[1], fpos[2] - p->pos[2]};
    const double on = sqrt(od[0] * od[0] + od[1] * od[1] + od[2] * od[2]);
    od[0] /= on; od[1] /= on; od[2] /= on;
    int best = 0;
    double min_cos = 0;
    for (int k = 0; k < p->n_obs; k++) {
        double c[3];
        frame_pos(p->obs[k].R, p->obs[k].t, c);
        double d[3] = {c[0] - p->pos[0], c[1] - p->pos[1], c[2] - p->pos[2]};
        const double dn = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        d[0] /= dn; d[1] /= dn; d[2] /= dn;
        const double cosa = od[0] * d[0] + od[1] * d[1] + od[2] * d[2];
        if (cosa > min_cos) { min_cos = cosa; best = k; }
    }
    return (min_cos < 0.5) ? -1 : best;
}

int orc_vmap_select(orc_vmap *m, const double *Rcw, const double *Pcw, const uint8_t *cur_img, const uint8_t *const *keyframes,
                    const float *scan_down_world_xyz, int n, int ncc_en, double ncc_thre, double outlier_threshold,
                    int32_t *sel_point /* room for length */, float *errors, int32_t *search_levels, float *patches /* length x 192 */,
                    int32_t *n_selected)
{
    const orc_vio_config *c = &m->cfg;
    const int W = c->width, H = c->height, half = c->patch_size / 2;
    *n_selected = 0;
    if (m->n <= 0) return 0;                                                             /* :348 */
    reset_grid(m);
    memset(m->map_value, 0, sizeof(float) * (size_t)m->length);                           /* :356 */
    /* sub_feat_map keys :383-391 */
    int64_t *keys = (int64_t *)malloc(sizeof(int64_t) * 3 * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++)
        for (int j = 0; j < 3; j++) keys[3 * i + j] = (int64_t)(int)floor((double)scan_down_world_xyz[3 * i + j] / (double)0.5f);
    float *depth = (float *)malloc(sizeof(float) * (size_t)W * (size_t)H);
    orc_vio_depth_image(c, Rcw, Pcw, scan_down_world_xyz, n, depth);
    double fpos[3];
    frame_pos(Rcw, Pcw, fpos);
    for (int v = 0; v < m->n; v++) {
        const vpoint *p = &m->pts[v];
        int hit = 0;
        for (int i = 0; i < n && !hit; i++) hit = keys[3 * i] == p->key[0] && keys[3 * i + 1] == p->key[1] && keys[3 * i + 2] == p->key[2];
        if (!hit) continue;
        double pc3[3], pc[2];
        w2f(Rcw, Pcw, p->pos, pc3);
        if (pc3[2] < 0) continue;                                                        /* :430 */
        orc_world2cam(c, pc3, pc);
        if (!in_frame((int)pc[0], (int)pc[1], (half + 1) * 8, W, H)) continue;
        const int index = (int)(pc[0] / m->grid_size) * m->gh + (int)(pc[1] / m->grid_size);
        if (index < 0 || index >= m->length) continue;
        m->grid_num[index] = 1;
        const double o0 = fpos[0] - p->pos[0], o1 = fpos[1] - p->pos[1], o2 = fpos[2] - p->pos[2];
        const float cur_dist = (float)sqrt(o0 * o0 + o1 * o1 + o2 * o2);
        if (cur_dist <= m->map_dist[index]) { m->map_dist[index] = cur_dist; m->winner[index] = v; }
        if (p->value >= m->map_value[index]) m->map_value[index] = p->value;
    }
    /* winners in ascending cell order -> candidates (getCloseViewObs) -> pixel-level part */
    orc_patch_candidate *cand = (orc_patch_candidate *)malloc(sizeof(orc_patch_candidate) * (size_t)m->length);
    int nc = 0;
    for (int g = 0; g < m->length; g++) {
        if (m->grid_num[g] != 1 || m->winner[g] < 0) continue;
        const vpoint *p = &m->pts[m->winner[g]];
        const int k = close_view_obs(p, fpos);
        if (k < 0) continue;
        orc_patch_candidate *cd = &cand[nc++];
        memset(cd, 0, sizeof *cd);
        memcpy(cd->pos, p->pos, sizeof cd->pos);
        memcpy(cd->px_ref, p->obs[k].px, sizeof cd->px_ref); memcpy(cd->f_ref, p->obs[k].f, sizeof cd->f_ref);
        memcpy(cd->R_ref, p->obs[k].R, sizeof cd->R_ref); memcpy(cd->t_ref, p->obs[k].t, sizeof cd->t_ref);
        cd->keyframe_id = p->obs[k].kf_id; cd->level_ref = p->obs[k].level; cd->grid_index = g; cd->reserved = m->winner[g];
    }
    int32_t *acc = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nc > 0 ? nc : 1));
    int32_t na = 0;
    int rc = 0;
    if (nc > 0) rc = orc_vio_select(c, Rcw, Pcw, cur_img, keyframes, depth, cand, nc, ncc_en, ncc_thre, outlier_threshold, acc, patches, errors,
                                    search_levels, &na, NULL);
    for (int i = 0; i < na; i++) sel_point[i] = cand[acc[i]].reserved;
    *n_selected = na;
    free(keys); free(depth); free(cand); free(acc);
    return rc;
}

/* addObservation */
int orc_vmap_add_observation(orc_vmap *m, const double *Rcw, const double *Pcw, const uint8_t *img, const int32_t *sel_point,
                             const int32_t *search_levels, int n_sel, int kf_id, int frame_id)
{
    const orc_vio_config *c = &m->cfg;
    const int W = c->width, H = c->height;
    double fpos[3];
    frame_pos(Rcw, Pcw, fpos);
    int added = 0;
    for (int i = 0; i < n_sel; i++) {
        vpoint *p = &m->pts[sel_point[i]];
        double pc3[3], pc[2];
        w2f(Rcw, Pcw, p->pos, pc3);
        orc_world2cam(c, pc3, pc);
        int add_flag = 0;
        const vobs *last = &p->obs[p->n_obs - 1];                                        /* obs_.back() */
        /* delta_pose = pose_ref * pose_cur.inverse(): R = R_ref R_cur^T, t = t_ref - R t_cur */
        double Rd[9], td[3];
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) Rd[a * 3 + b] = last->R[a * 3] * Rcw[b * 3] + last->R[a * 3 + 1] * Rcw[b * 3 + 1] + last->R[a * 3 + 2] * Rcw[b * 3 + 2];
        mv(Rd, Pcw, td);
        for (int a = 0; a < 3; a++) td[a] = last->t[a] - td[a];
        const double delta_p = sqrt(td[0] * td[0] + td[1] * td[1] + td[2] * td[2]);
        const double tr = Rd[0] + Rd[4] + Rd[8];
        const double delta_theta = (tr > 3.0 - 1e-6) ? 0.0 : acos(0.5 * (tr - 1));
        if (delta_p > 0.5 || delta_theta > 10) add_flag = 1;                              /* :939 (radians compared with 10) */
        const double e0 = pc[0] - last->px[0], e1 = pc[1] - last->px[1];
        if (sqrt(e0 * e0 + e1 * e1) > 40) add_flag = 1;                                   /* :942-944 */
        if (p->n_obs >= 20) {                                                             /* :947-953 getFurthestViewObs + deleteFeatureRef */
            int far = 0;
            double maxdist = 0.0;
            for (int k = 0; k < p->n_obs; k++) {
                double cpos[3];
                frame_pos(p->obs[k].R, p->obs[k].t, cpos);
                const double d0 = cpos[0] - fpos[0], d1 = cpos[1] - fpos[1], d2 = cpos[2] - fpos[2];
                const double dist = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
                if (dist > maxdist) { maxdist = dist; far = k; }
            }
            for (int k = far; k + 1 < p->n_obs; k++) p->obs[k] = p->obs[k + 1];
            p->n_obs--;
        }
        if (add_flag) {
            p->value = orc_shi_tomasi(img, W, H, (int)pc[0], (int)pc[1]);
            for (int k = p->n_obs; k > 0; k--) p->obs[k] = p->obs[k - 1];                 /* push_front */
            vobs *o = &p->obs[0];
            o->px[0] = pc[0]; o->px[1] = pc[1];
            cam2world(c, pc[0], pc[1], o->f);
            memcpy(o->R, Rcw, sizeof o->R); memcpy(o->t, Pcw, sizeof o->t);
            o->score = p->value; o->level = search_levels[i]; o->kf_id = kf_id; o->frame_id = frame_id;
            p->n_obs++;
            added++;
        }
    }
    return added;
}
