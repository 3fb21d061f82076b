/*
 * oracle/orc_select.c -- TEST INFRASTRUCTURE ONLY (see fastlivo_oracle.h).
 *
 * CPU restatement of the pixel-level part of LidarSelector::addFromSparseMap, /root/reference/src/lidar_selection.cpp:
 *   depth image of the scan                          :376-410   (orc_vio_depth_image)
 *   per grid winner (the loop at :470-583):
 *     depth-continuity test                          :484-506
 *     Warp_map: warp + search level per reference frame :530-546 (first candidate of a frame computes them, the rest reuse them)
 *     getWarpMatrixAffine                            :232-256  (call :537-538, level_ref = pyramid_level = 0)
 *     getBestSearchLevel                             :315-329  (call :534, max_level 2)
 *     warpAffine for pyramid levels 0..2             :258-296  (call :545-548)
 *     getpatch of the current image, level 0         :119-140  (call :557)
 *     NCC gate (if ncc_en)                           :298-313, :559-563
 *     squared-error outlier gate                     :565-570
 *   accepted candidates are appended in ascending grid index (:572-579).
 * What stays with the caller (pointer-chasing over the visual map, out of scope): the voxel lookups and the grid
 * competition (:412-466) and Point::getCloseViewObs (src/point.cpp:141-178), which pick, per grid cell, the map point
 * and the reference observation handed in here as a candidate.
 * Third-party arithmetic not under /root/reference, restated from the published sources (unpinned; the reference's own
 * lines of this file are held to their text since round 4, oracle/ref_eigen, tests/test_ref_eigen_cpu.py):
 *   vk::interpolateMat_8u (rpg_vikit vision.h): w00=(1-sx)(1-sy), w01=(1-sx)sy, w10=sx(1-sy), w11=1-w00-w01-w10, floats;
 *   vk::PinholeCamera::cam2world: ((u-cx)/fx, (v-cy)/fy, 1).normalized(); with distortion vikit calls cv::undistortPoints,
 *   restated as its five fixed-point sweeps (orc_cam2world in orc_vio.c); world2cam as in orc_vio.c;
 *   vk::AbstractCamera::isInFrame(obs, boundary): boundary <= obs < size - boundary;
 *   Sophus::SE3 (T*p, inverse, product) stated with rotation matrices (Sophus a621ff keeps a quaternion: results
 *   agree to rounding, not bitwise); Eigen 2x2 inverse = adjugate * (1/det).
 */
#include "fastlivo_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

static void s_mv(const double *A, const double *x, double *o)
{
    double t[3];
    for (int i = 0; i < 3; i++) t[i] = A[i * 3] * x[0] + A[i * 3 + 1] * x[1] + A[i * 3 + 2] * x[2];
    memcpy(o, t, sizeof t);
}
static int in_frame(int u, int v, int boundary, int w, int h) { return u >= boundary && u < w - boundary && v >= boundary && v < h - boundary; }

static void cam2world(const orc_vio_config *c, double u, double v, double *f) { orc_cam2world(c, u, v, f); }

static float interpolate_8u(const uint8_t *img, int stride, float u, float v)
{
    const int x = (int)floorf(u), y = (int)floorf(v);
    const float sx = u - x, sy = v - y;
    const float w00 = (1.0f - sx) * (1.0f - sy), w01 = (1.0f - sx) * sy, w10 = sx * (1.0f - sy);
    const float w11 = 1.0f - w00 - w01 - w10;
    const uint8_t *p = img + y * stride + x;
    return w00 * p[0] + w01 * p[stride] + w10 * p[1] + w11 * p[stride + 1];
}

/* :376-410. depth: width*height floats, zeroed here. */
void orc_vio_depth_image(const orc_vio_config *cfg, const double *Rcw, const double *Pcw, const float *scan_world_xyz, int n,
                         float *depth)
{
    const int W = cfg->width, H = cfg->height, half = cfg->patch_size / 2;
    memset(depth, 0, sizeof(float) * (size_t)W * (size_t)H);
    for (int i = 0; i < n; i++) {
        const double pw[3] = {scan_world_xyz[3 * i], scan_world_xyz[3 * i + 1], scan_world_xyz[3 * i + 2]};
        double pc[3];
        s_mv(Rcw, pw, pc);
        pc[0] += Pcw[0]; pc[1] += Pcw[1]; pc[2] += Pcw[2];
        if (pc[2] > 0) {
            const double px0 = cfg->fx * pc[0] / pc[2] + cfg->cx, px1 = cfg->fy * pc[1] / pc[2] + cfg->cy;   /* :398-399 */
            if (in_frame((int)px0, (int)px1, (half + 1) * 8, W, H)) depth[W * (int)px1 + (int)px0] = (float)pc[2];
        }
    }
}

int orc_vio_select(const orc_vio_config *cfg, const double *Rcw, const double *Pcw, const uint8_t *cur_img,
                   const uint8_t *const *keyframes, const float *depth, const orc_patch_candidate *cand, int m,
                   int ncc_en, double ncc_thre, double outlier_threshold, int32_t *accepted_idx, float *patches,
                   float *errors, int32_t *search_levels, int32_t *n_accepted, int32_t *reason)
{
    const int W = cfg->width, H = cfg->height, ps = cfg->patch_size, half = ps / 2, pst = ps * ps;
    int na = 0, n_warp = 0;
    int32_t *warp_kf = (int32_t *)malloc(sizeof(int32_t) * (size_t)(m > 0 ? m : 1)), *warp_owner = (int32_t *)malloc(sizeof(int32_t) * (size_t)(m > 0 ? m : 1));
    for (int ci = 0; ci < m; ci++) {
        const orc_patch_candidate *c = &cand[ci];
        if (reason) reason[ci] = 0;
        double pt_cam[3], pc[2];
        s_mv(Rcw, c->pos, pt_cam);
        pt_cam[0] += Pcw[0]; pt_cam[1] += Pcw[1]; pt_cam[2] += Pcw[2];
        orc_world2cam(cfg, pt_cam, pc);                                                                /* :480-481 */
        int discont = 0;
        for (int u = -half; u <= half && !discont; u++)                                                /* :484-506 */
            for (int v = -half; v <= half; v++) {
                if (u == 0 && v == 0) continue;
                const float d = depth[W * (v + (int)pc[1]) + u + (int)pc[0]];
                if (d == 0.f) continue;
                if (fabs(pt_cam[2] - (double)d) > 1.5) { discont = 1; break; }
            }
        if (discont) { if (reason) reason[ci] = 1; continue; }
        /* Warp_map (:530-546): the affine warp and the search level are computed for the FIRST candidate of a reference frame that
         * gets this far and REUSED for every later candidate observed in the same frame (key ref_ftr->id_; one image per frame, so the
         * keyframe id stands for it) -- with that first candidate's pixel, bearing, depth and pose, not the current one's */
        int owner = -1;
        for (int k = 0; k < n_warp; k++)
            if (warp_kf[k] == c->keyframe_id) { owner = warp_owner[k]; break; }
        if (owner < 0) { owner = ci; warp_kf[n_warp] = c->keyframe_id; warp_owner[n_warp] = ci; n_warp++; }
        const orc_patch_candidate *w = &cand[owner];
        /* getWarpMatrixAffine :232-256 */
        double Rt[9], ref_pos[3], T_R[9], T_t[3];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) Rt[i * 3 + j] = w->R_ref[j * 3 + i];
        s_mv(Rt, w->t_ref, ref_pos);
        for (int k = 0; k < 3; k++) ref_pos[k] = -ref_pos[k];                                          /* Feature::pos() */
        const double dv[3] = {ref_pos[0] - w->pos[0], ref_pos[1] - w->pos[1], ref_pos[2] - w->pos[2]};
        const double depth_ref = sqrt(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2]);
        for (int i = 0; i < 3; i++)                                                                    /* T_cur_ref = T_cur * T_ref^-1 */
            for (int j = 0; j < 3; j++) T_R[i * 3 + j] = Rcw[i * 3] * Rt[j] + Rcw[i * 3 + 1] * Rt[3 + j] + Rcw[i * 3 + 2] * Rt[6 + j];
        s_mv(Rcw, ref_pos, T_t);
        for (int k = 0; k < 3; k++) T_t[k] += Pcw[k];
        const double xyz_ref[3] = {w->f_ref[0] * depth_ref, w->f_ref[1] * depth_ref, w->f_ref[2] * depth_ref};
        double du[3], dw[3];
        cam2world(cfg, w->px_ref[0] + (double)half, w->px_ref[1], du);
        cam2world(cfg, w->px_ref[0], w->px_ref[1] + (double)half, dw);
        const double su = xyz_ref[2] / du[2], sw = xyz_ref[2] / dw[2];
        for (int k = 0; k < 3; k++) { du[k] *= su; dw[k] *= sw; }
        double q[3], px_cur[2], px_du[2], px_dv[2];
        s_mv(T_R, xyz_ref, q); for (int k = 0; k < 3; k++) q[k] += T_t[k];
        orc_world2cam(cfg, q, px_cur);
        s_mv(T_R, du, q); for (int k = 0; k < 3; k++) q[k] += T_t[k];
        orc_world2cam(cfg, q, px_du);
        s_mv(T_R, dw, q); for (int k = 0; k < 3; k++) q[k] += T_t[k];
        orc_world2cam(cfg, q, px_dv);
        const double A00 = (px_du[0] - px_cur[0]) / half, A10 = (px_du[1] - px_cur[1]) / half;
        const double A01 = (px_dv[0] - px_cur[0]) / half, A11 = (px_dv[1] - px_cur[1]) / half;
        /* getBestSearchLevel :315-329 */
        int search_level = 0;
        double D = A00 * A11 - A01 * A10;
        const double det = D;
        while (D > 3.0 && search_level < 2) { search_level += 1; D *= 0.25; }
        /* warpAffine :258-296 */
        float *P = patches + (size_t)na * 3 * pst;
        memset(P, 0, sizeof(float) * 3 * (size_t)pst);
        const double invdet = 1.0 / det;
        const float B00 = (float)(A11 * invdet), B01 = (float)(-A01 * invdet), B10 = (float)(-A10 * invdet), B11 = (float)(A00 * invdet);
        const uint8_t *ref = keyframes[c->keyframe_id];
        if (!isnan(B00)) {
            for (int lvl = 0; lvl <= 2; lvl++)
                for (int y = 0; y < ps; y++)
                    for (int x = 0; x < ps; x++) {
                        float p0 = (float)(x - half), p1 = (float)(y - half);
                        p0 *= (float)(1 << search_level); p1 *= (float)(1 << search_level);
                        p0 *= (float)(1 << lvl); p1 *= (float)(1 << lvl);
                        const float u = (B00 * p0 + B01 * p1) + (float)c->px_ref[0];
                        const float v = (B10 * p0 + B11 * p1) + (float)c->px_ref[1];
                        if (u < 0 || v < 0 || u >= W - 1 || v >= H - 1) P[pst * lvl + y * ps + x] = 0;
                        else P[pst * lvl + y * ps + x] = interpolate_8u(ref, W, u, v);
                    }
        }
        /* getpatch(img, pc, patch_cache, 0) :119-140 */
        float cur[64 * 4];
        {
            const float u_ref = (float)pc[0], v_ref = (float)pc[1];
            const int u_i = (int)floorf((float)pc[0]), v_i = (int)floorf((float)pc[1]);
            const float su_ = u_ref - u_i, sv_ = v_ref - v_i;
            const float w_tl = (float)((1.0 - su_) * (1.0 - sv_)), w_tr = (float)(su_ * (1.0 - sv_)), w_bl = (float)((1.0 - su_) * sv_),
                        w_br = su_ * sv_;
            for (int x = 0; x < ps; x++) {
                const uint8_t *ip = cur_img + (v_i - half + x) * W + (u_i - half);
                for (int y = 0; y < ps; y++, ip++) cur[x * ps + y] = w_tl * ip[0] + w_tr * ip[1] + w_bl * ip[W] + w_br * ip[W + 1];
            }
        }
        if (ncc_en) {                                                                                  /* :298-313 */
            double sr = 0.0, sc = 0.0;
            for (int i = 0; i < pst; i++) sr += P[i];
            for (int i = 0; i < pst; i++) sc += cur[i];
            const double mr = sr / pst, mc = sc / pst;
            double num = 0, d1 = 0, d2 = 0;
            for (int i = 0; i < pst; i++) {
                num += (P[i] - mr) * (cur[i] - mc);
                d1 += (P[i] - mr) * (P[i] - mr);
                d2 += (cur[i] - mc) * (cur[i] - mc);
            }
            if (num / sqrt(d1 * d2 + 1e-10) < ncc_thre) { if (reason) reason[ci] = 3; continue; }
        }
        float error = 0.0f;
        for (int i = 0; i < pst; i++) error += (P[i] - cur[i]) * (P[i] - cur[i]);                      /* :565-569 */
        if (error > outlier_threshold * pst) { if (reason) reason[ci] = 4; continue; }
        accepted_idx[na] = ci;
        errors[na] = error;
        search_levels[na] = search_level;
        na++;
    }
    *n_accepted = na;
    free(warp_kf); free(warp_owner);
    return 0;
}

/* Projection + grid competition over the map points of the visible voxels, lidar_selection.cpp:412-466, given those points as a
 * flat list in the order the reference's loops visit them (the walk over sub_feat_map / feat_map stays with the caller).
 * winner[length]: index into the list of the point each grid cell keeps (voxel_points_[index]), -1 if none; map_dist starts at
 * 10000 (reset_grid, :85), map_value at 0 (:356), grid_num: 1 = TYPE_MAP where a point projected, else 3 = TYPE_UNKNOWN.
 * A cell index >= length (possible in the reference when width is not a multiple of grid_size) is skipped. */
int orc_vio_grid_select(const orc_vio_config *cfg, const double *Rcw, const double *Pcw, const double *pos, const float *value, int k,
                        int grid_size, int32_t *winner, float *map_dist, float *map_value, int32_t *grid_num)
{
    const int W = cfg->width, H = cfg->height, half = cfg->patch_size / 2;
    const int gw = W / grid_size, gh = H / grid_size, length = gw * gh;
    for (int i = 0; i < length; i++) { winner[i] = -1; map_dist[i] = 10000.f; map_value[i] = 0.f; grid_num[i] = 3; }
    double fpos[3];                                    /* new_frame_->pos() = T_f_w_.inverse().translation() = -(Rcw^T Pcw) */
    for (int i = 0; i < 3; i++) fpos[i] = -(Rcw[i] * Pcw[0] + Rcw[3 + i] * Pcw[1] + Rcw[6 + i] * Pcw[2]);
    for (int j = 0; j < k; j++) {
        const double *p = pos + 3 * (size_t)j;
        double pc3[3], px[2];
        s_mv(Rcw, p, pc3);
        pc3[0] += Pcw[0]; pc3[1] += Pcw[1]; pc3[2] += Pcw[2];
        if (pc3[2] < 0) continue;                                                      /* :430 */
        orc_world2cam(cfg, pc3, px);                                                   /* :432 */
        if (!in_frame((int)px[0], (int)px[1], (half + 1) * 8, W, H)) continue;         /* :436 */
        const int index = (int)(px[0] / grid_size) * gh + (int)(px[1] / grid_size);   /* :438 */
        if (index < 0 || index >= length) continue;
        grid_num[index] = 1;
        const double o0 = fpos[0] - p[0], o1 = fpos[1] - p[1], o2 = fpos[2] - p[2];
        const float cur_dist = (float)sqrt(o0 * o0 + o1 * o1 + o2 * o2);
        const float cur_value = value[j];
        if (cur_dist <= map_dist[index]) { map_dist[index] = cur_dist; winner[index] = j; }      /* :445-449 */
        if (cur_value >= map_value[index]) map_value[index] = cur_value;                           /* :451-454 */
    }
    return length;
}
