/*
 * oracle/orc_vio.c -- TEST INFRASTRUCTURE ONLY (CPU oracle). Never linked into the product.
 *
 * Photometric (8x8 patch) ESKF update of FAST-LIVO's VIO side, single-threaded like the
 * reference.
 *
 * Reference lines restated (file:line under /root/reference):
 *   set_extrinsic / init extrinsic Jacobians   src/lidar_selection.cpp:35-59
 *   dpi                                       src/lidar_selection.cpp:92-103
 *   UpdateState                               src/lidar_selection.cpp:743-902
 *   ComputeJ                                  src/lidar_selection.cpp:967-983
 *   patch layout P[64*level + 8*x + y]         src/lidar_selection.cpp:837 (producer :258-296)
 *   vk::PinholeCamera::world2cam              rpg_vikit (third party, unpinned git master,
 *                                              README.md:73-76) -- NOT in the tree; restated from
 *                                              memory of vikit_common/src/pinhole_camera.cpp:
 *                                              project2d then radial-tangential distortion
 *                                              when |d0| > 1e-7. d = 0 is the primary parity config.
 * The reference's own lines are held to their text since round 4 (oracle/ref_eigen, tests/test_ref_eigen_cpu.py: bit for bit over a stand-in
 * for Eigen); vikit's camera and Eigen's arithmetic stay unpinned -- see fastlivo_oracle.h.
 */
#include "fastlivo_oracle.h"
#include "orc_math.h"

#include <stdlib.h>

int orc_solve18(orc_state18 *x, const orc_state18 *x_prop, const double *HTH6, const double *HTz6,
                double meas_cov, double sign, double *G, double *solution);

/* Sensitivity study only (tests/test_radtan_sensitivity_cpu.py): vikit is absent from this image, so the operation order of its radtan
 * branch cannot be pinned. g_radtan_mode selects plausible alternatives of the SAME formula: bit 0 = Horner form of the radial
 * polynomial (1 + r2 (k1 + r2 (k2 + r2 k3))) instead of the powers r2, r4, r6; bit 1 = the final `xd * fx + cx` contracted into a
 * fused multiply-add (what -march=native / -ffp-contract=fast would do); bit 2 = the tangential terms associated the other way
 * (x cdist + (p1 a1 + p2 a2)). Mode 0 is the restatement every parity test uses. */
static int g_radtan_mode = 0;
void orc_vio_set_radtan_mode(int mode) { g_radtan_mode = mode; }

void orc_world2cam(const orc_vio_config *cfg, const double *xyz_c, double *px)
{
    double u = xyz_c[0] / xyz_c[2], v = xyz_c[1] / xyz_c[2];
    if (!(fabs(cfg->d[0]) > 0.0000001)) {
        px[0] = cfg->fx * u + cfg->cx;
        px[1] = cfg->fy * v + cfg->cy;
    } else {
        double x = u, y = v;
        double r2 = x * x + y * y, r4 = r2 * r2, r6 = r4 * r2;
        double a1 = 2 * x * y, a2 = r2 + 2 * x * x, a3 = r2 + 2 * y * y;
        double cdist = 1 + cfg->d[0] * r2 + cfg->d[1] * r4 + cfg->d[4] * r6;
        if (g_radtan_mode & 1) cdist = 1 + r2 * (cfg->d[0] + r2 * (cfg->d[1] + r2 * cfg->d[4]));
        double xd = x * cdist + cfg->d[2] * a1 + cfg->d[3] * a2;
        double yd = y * cdist + cfg->d[3] * a1 + cfg->d[2] * a3;
        if (g_radtan_mode & 4) {
            xd = x * cdist + (cfg->d[2] * a1 + cfg->d[3] * a2);
            yd = y * cdist + (cfg->d[3] * a1 + cfg->d[2] * a3);
        }
        px[0] = xd * cfg->fx + cfg->cx;
        px[1] = yd * cfg->fy + cfg->cy;
        if (g_radtan_mode & 2) {
            px[0] = fma(xd, cfg->fx, cfg->cx);
            px[1] = fma(yd, cfg->fy, cfg->cy);
        }
    }
}

/* vk::PinholeCamera::cam2world (rpg_vikit pinhole_camera.cpp; unpinned, restated): without distortion the bearing of the pixel;
 * with distortion vikit packs (u, v) into a cv::Point2f, calls cv::undistortPoints(src, dst, K, D) and normalises (px.x, px.y, 1).
 * cv::undistortPoints (OpenCV 3.x/4.x imgproc/undistort.cpp, the 4-argument overload: TermCriteria(MAX_ITER, 5, 0.01) = five
 * fixed-point sweeps, no tolerance test), restated for the 5-coefficient model k1,k2,p1,p2,k3: float in, doubles inside, float out. */
void orc_cam2world(const orc_vio_config *c, double u, double v, double *f)
{
    double x, y;
    if (!(fabs(c->d[0]) > 0.0000001)) {
        x = (u - c->cx) / c->fx; y = (v - c->cy) / c->fy;
    } else {
        const double ifx = 1. / c->fx, ify = 1. / c->fy;
        x = (double)(float)u; y = (double)(float)v;
        x = (x - c->cx) * ifx; y = (y - c->cy) * ify;
        const double x0 = x, y0 = y;
        const double k1 = c->d[0], k2 = c->d[1], p1 = c->d[2], p2 = c->d[3], k3 = c->d[4];
        for (int j = 0; j < 5; j++) {
            const double r2 = x * x + y * y;
            const double icdist = (1 + ((0 * r2 + 0) * r2 + 0) * r2) / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
            const double deltaX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
            const double deltaY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
            x = (x0 - deltaX) * icdist; y = (y0 - deltaY) * icdist;
        }
        x = (double)(float)x; y = (double)(float)y;
    }
    const double z = 1.0, n = sqrt(x * x + y * y + z * z);
    f[0] = x / n; f[1] = y / n; f[2] = z / n;
}

typedef struct vio_extr {
    double Rci[9], Pci[3], Jdphi_dR[9], Jdp_dR[9];
    double fx, fy;
} vio_extr;

/* lidar_selection.cpp:35-59 */
static void vio_init_extr(const orc_vio_config *cfg, vio_extr *e)
{
    double Rli[9], Pli[3], t[3];
    m3_tr(cfg->R_LI, Rli);                       /* Rli = rot.transpose() */
    m3_vec(Rli, cfg->t_LI, t);
    Pli[0] = -t[0]; Pli[1] = -t[1]; Pli[2] = -t[2]; /* Pli = -rot^T * transl */
    m3_mul(cfg->Rcl, Rli, e->Rci);
    m3_vec(cfg->Rcl, Pli, e->Pci);
    for (int i = 0; i < 3; i++) e->Pci[i] += cfg->Pcl[i];
    memcpy(e->Jdphi_dR, e->Rci, sizeof e->Rci);
    double Rcit[9], Pic[3], tmp[9], nR[9];
    m3_tr(e->Rci, Rcit);
    for (int i = 0; i < 9; i++) nR[i] = -Rcit[i];
    m3_vec(nR, e->Pci, Pic);                     /* Pic = -Rci^T * Pci */
    skew3(Pic, tmp);
    for (int i = 0; i < 9; i++) nR[i] = -e->Rci[i];
    m3_mul(nR, tmp, e->Jdp_dR);                  /* Jdp_dR = -Rci * [Pic]x */
    e->fx = fabs(cfg->fx);                       /* errorMultiplier2() */
    e->fy = fabs(4.0 * cfg->fx * cfg->fy) / (4. * e->fx); /* errorMultiplier()/(4 fx) */
}

/* The reference runs this loop on ONE thread (lidar_selection.cpp:789-853 has no OpenMP pragma). For the "generous" CPU baseline of
 * bench.py the patch loop and the 42 column sums can be spread over threads WITHOUT changing a bit of the result: rows
 * of different patches are independent, the float running sum `error += patch_error` is taken afterwards in patch order, and every
 * H^T H / H^T z entry is still one sequential sum. 1 = the reference. */
static int g_vio_threads = 1;
void orc_vio_set_threads(int n) { g_vio_threads = n > 1 ? n : 1; }

/* camera pose of state x (lidar_selection.cpp:780-784), for the sensitivity study of tests/test_radtan_sensitivity_cpu.py */
void orc_vio_cam_pose(const orc_vio_config *cfg, const orc_state18 *x, double *Rcw, double *Pcw)
{
    vio_extr ex;
    vio_init_extr(cfg, &ex);
    double Rwit[9], nRci[9], T[9], t3[3];
    m3_tr(x->rot, Rwit);
    m3_mul(ex.Rci, Rwit, Rcw);
    for (int i = 0; i < 9; i++) nRci[i] = -ex.Rci[i];
    m3_mul(nRci, Rwit, T);
    m3_vec(T, x->pos, t3);
    for (int i = 0; i < 3; i++) Pcw[i] = t3[i] + ex.Pci[i];
}

float orc_vio_update_state(const orc_vio_config *cfg, orc_state18 *x, const orc_state18 *x_prop,
                           const uint8_t *img, const float *ref_patch, const double *pos,
                           const int32_t *search_level, int m, float total_residual, int level,
                           float *errors, double *G, orc_vio_level_out *out)
{
    if (out) memset(out, 0, sizeof *out);
    if (m == 0) return 0.f;
    vio_extr ex;
    vio_init_extr(cfg, &ex);
    const int patch_size = cfg->patch_size, pst = patch_size * patch_size, half = patch_size / 2;
    const int width = cfg->width;
    const int H_DIM = m * pst;
    orc_state18 old_state = *x;
    double *z = (double *)calloc((size_t)H_DIM, sizeof(double));
    double *H_sub = (double *)calloc((size_t)H_DIM * 6, sizeof(double));
    int EKF_end = 0;
    float error = 0.0f, last_error = total_residual, patch_error = 0.0f;
    int iters = 0, accepted = 0, n_meas = 0, fragile = 0;
    double HTH[36], HTz[6], solution[18];
    memset(HTH, 0, sizeof HTH); memset(HTz, 0, sizeof HTz); memset(solution, 0, sizeof solution);

    for (int iteration = 0; iteration < cfg->max_iterations; iteration++) {
        iters++;
        error = 0.0f;
        n_meas = 0;
        double Rwit[9], Rcw[9], Pcw[3], Jdp_dt[9], t3[3];
        m3_tr(x->rot, Rwit);
        m3_mul(ex.Rci, Rwit, Rcw);                       /* Rcw = Rci * Rwi^T */
        {                                                /* Pcw = -Rci*Rwi^T*Pwi + Pci */
            double nRci[9], T[9];
            for (int i = 0; i < 9; i++) nRci[i] = -ex.Rci[i];
            m3_mul(nRci, Rwit, T);
            m3_vec(T, x->pos, t3);
            for (int i = 0; i < 3; i++) Pcw[i] = t3[i] + ex.Pci[i];
        }
        memcpy(Jdp_dt, Rcw, sizeof Rcw);                 /* Jdp_dt = Rci * Rwi^T */

#ifdef _OPENMP
#pragma omp parallel for num_threads(g_vio_threads) if (g_vio_threads > 1) schedule(static) private(patch_error)
#endif
        for (int i = 0; i < m; i++) {
            patch_error = 0.0f;
            const int pyramid_level = level + search_level[i];
            const int scale = (1 << pyramid_level);
            double pf[3], pc[2];
            m3_vec(Rcw, pos + (size_t)i * 3, pf);
            pf[0] += Pcw[0]; pf[1] += Pcw[1]; pf[2] += Pcw[2];
            orc_world2cam(cfg, pf, pc);
            /* dpi, :92-103 */
            double Jdpi[6];
            {
                const double xx = pf[0], yy = pf[1], z_inv = 1. / pf[2], z_inv_2 = z_inv * z_inv;
                Jdpi[0] = ex.fx * z_inv; Jdpi[1] = 0.0; Jdpi[2] = -ex.fx * xx * z_inv_2;
                Jdpi[3] = 0.0; Jdpi[4] = ex.fy * z_inv; Jdpi[5] = -ex.fy * yy * z_inv_2;
            }
            double p_hat[9];
            skew3(pf, p_hat);
            const float u_ref = (float)pc[0];
            const float v_ref = (float)pc[1];
            const int u_ref_i = (int)(floorf((float)(pc[0] / scale)) * scale);
            const int v_ref_i = (int)(floorf((float)(pc[1] / scale)) * scale);
            const float subpix_u_ref = (u_ref - u_ref_i) / scale;
            const float subpix_v_ref = (v_ref - v_ref_i) / scale;
            const float w_ref_tl = (float)((1.0 - subpix_u_ref) * (1.0 - subpix_v_ref));
            const float w_ref_tr = (float)(subpix_u_ref * (1.0 - subpix_v_ref));
            const float w_ref_bl = (float)((1.0 - subpix_u_ref) * subpix_v_ref);
            const float w_ref_br = subpix_u_ref * subpix_v_ref;
            const float *P = ref_patch + (size_t)i * 3 * pst;

            for (int xr = 0; xr < patch_size; xr++) {
                const uint8_t *img_ptr = img + (v_ref_i + xr * scale - half * scale) * width + u_ref_i - half * scale;
                for (int y = 0; y < patch_size; ++y, img_ptr += scale) {
                    float du = 0.5f * ((w_ref_tl * img_ptr[scale] + w_ref_tr * img_ptr[scale * 2] + w_ref_bl * img_ptr[scale * width + scale] + w_ref_br * img_ptr[scale * width + scale * 2])
                                     - (w_ref_tl * img_ptr[-scale] + w_ref_tr * img_ptr[0] + w_ref_bl * img_ptr[scale * width - scale] + w_ref_br * img_ptr[scale * width]));
                    float dv = 0.5f * ((w_ref_tl * img_ptr[scale * width] + w_ref_tr * img_ptr[scale + scale * width] + w_ref_bl * img_ptr[width * scale * 2] + w_ref_br * img_ptr[width * scale * 2 + scale])
                                     - (w_ref_tl * img_ptr[-scale * width] + w_ref_tr * img_ptr[-scale * width + scale] + w_ref_bl * img_ptr[0] + w_ref_br * img_ptr[scale]));
                    double Jimg[2] = {(double)du, (double)dv};
                    Jimg[0] = Jimg[0] * (1.0 / scale);
                    Jimg[1] = Jimg[1] * (1.0 / scale);
                    double JJ[3], Jdphi[3], Jdp[3], JdR[3], Jdt[3];
                    /* Jdphi = Jimg * Jdpi * p_hat */
                    for (int c = 0; c < 3; c++) JJ[c] = Jimg[0] * Jdpi[c] + Jimg[1] * Jdpi[3 + c];
                    for (int c = 0; c < 3; c++) Jdphi[c] = JJ[0] * p_hat[c] + JJ[1] * p_hat[3 + c] + JJ[2] * p_hat[6 + c];
                    /* Jdp = -Jimg * Jdpi */
                    for (int c = 0; c < 3; c++) Jdp[c] = (-Jimg[0]) * Jdpi[c] + (-Jimg[1]) * Jdpi[3 + c];
                    /* JdR = Jdphi * Jdphi_dR + Jdp * Jdp_dR ; Jdt = Jdp * Jdp_dt */
                    for (int c = 0; c < 3; c++) {
                        double a = Jdphi[0] * ex.Jdphi_dR[c] + Jdphi[1] * ex.Jdphi_dR[3 + c] + Jdphi[2] * ex.Jdphi_dR[6 + c];
                        double b = Jdp[0] * ex.Jdp_dR[c] + Jdp[1] * ex.Jdp_dR[3 + c] + Jdp[2] * ex.Jdp_dR[6 + c];
                        JdR[c] = a + b;
                        Jdt[c] = Jdp[0] * Jdp_dt[c] + Jdp[1] * Jdp_dt[3 + c] + Jdp[2] * Jdp_dt[6 + c];
                    }
                    double res = (double)(w_ref_tl * img_ptr[0] + w_ref_tr * img_ptr[scale] + w_ref_bl * img_ptr[scale * width] + w_ref_br * img_ptr[scale * width + scale] - P[pst * level + xr * patch_size + y]);
                    const int row = i * pst + xr * patch_size + y;
                    z[row] = res;
                    patch_error = (float)(patch_error + res * res);
                    H_sub[row * 6 + 0] = JdR[0]; H_sub[row * 6 + 1] = JdR[1]; H_sub[row * 6 + 2] = JdR[2];
                    H_sub[row * 6 + 3] = Jdt[0]; H_sub[row * 6 + 4] = Jdt[1]; H_sub[row * 6 + 5] = Jdt[2];
                }
            }
            errors[i] = patch_error;
        }
        for (int i = 0; i < m; i++) { error += errors[i]; n_meas += pst; }      /* `error += patch_error; n_meas++` in patch order (:849-857) */
        error = error / n_meas;

        /* not part of the reference: note when the test below is decided within the rounding noise of the float running sum
         * (see solve18.h); the parity tests use it to choose their tolerance */
        if (last_error < 1e9f && fabsf(error - last_error) <= 3e-5f * fabsf(error)) fragile = 1;
        if (error <= last_error) {
            old_state = *x;
            last_error = error;
#ifdef _OPENMP
#pragma omp parallel for num_threads(g_vio_threads) if (g_vio_threads > 1) schedule(static)
#endif
            for (int ab = 0; ab < 42; ab++) {
                const int a = ab / 7, b = ab % 7;
                double s = 0.0;
                if (b < 6) {
                    for (int k = 0; k < H_DIM; k++) s += H_sub[k * 6 + a] * H_sub[k * 6 + b];
                    HTH[a * 6 + b] = s;
                } else {
                    for (int k = 0; k < H_DIM; k++) s += H_sub[k * 6 + a] * z[k];
                    HTz[a] = s;
                }
            }
            orc_solve18(x, x_prop, HTH, HTz, cfg->img_point_cov, -1.0, G, solution);
            accepted++;
            double rn = sqrt(solution[0] * solution[0] + solution[1] * solution[1] + solution[2] * solution[2]);
            double tn = sqrt(solution[3] * solution[3] + solution[4] * solution[4] + solution[5] * solution[5]);
            if ((rn * 57.3f < 0.001f) && (tn * 100.0f < 0.001f)) EKF_end = 1;
        } else {
            *x = old_state;
            EKF_end = 1;
        }
        if (EKF_end) break;
    }
    if (out) {
        memcpy(out->HTH, HTH, sizeof HTH);
        memcpy(out->HTz, HTz, sizeof HTz);
        memcpy(out->solution, solution, sizeof solution);
        out->error = last_error;
        out->iterations = iters;
        out->n_meas = n_meas;
        out->accepted = accepted;
        out->fragile = fragile;
    }
    free(z); free(H_sub);
    return last_error;
}

int orc_vio_compute_j(const orc_vio_config *cfg, orc_state18 *x, const orc_state18 *x_prop,
                      const uint8_t *img, const float *ref_patch, const double *pos,
                      const int32_t *search_level, int m, float *errors, orc_vio_level_out *out3)
{
    if (m == 0) return 0;
    double G[18 * 18];
    memset(G, 0, sizeof G);                     /* lidar_selection.cpp:8 */
    float error = 1e10f, now_error = error;
    for (int level = 2; level >= 0; level--)
        now_error = orc_vio_update_state(cfg, x, x_prop, img, ref_patch, pos, search_level, m, error, level,
                                         errors, G, out3 ? &out3[level] : NULL);
    if (now_error < error) {                    /* state->cov -= G*state->cov */
        double GP[18 * 18];
        for (int i = 0; i < 18; i++)
            for (int j = 0; j < 18; j++) {
                double s = 0.0;
                for (int k = 0; k < 18; k++) s += G[i * 18 + k] * x->cov[k * 18 + j];
                GP[i * 18 + j] = s;
            }
        for (int i = 0; i < 18 * 18; i++) x->cov[i] -= GP[i];
    }
    return 0;
}
