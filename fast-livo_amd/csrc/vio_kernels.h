// vio_kernels.h -- gfx950 kernels for the 8x8-patch photometric ESKF update
// (LidarSelector::UpdateState / ComputeJ, src/lidar_selection.cpp:743-983).
//
//  K2  vio_iterate_kernel   one iteration at one pyramid level: ONE WAVEFRONT PER PATCH, lane =
//                           8*x + y = one pixel. Each lane gathers its 12 u8 taps (bilinear value +
//                           central-difference gradient), forms the 1x6 Jacobian row in fp64,
//                           the wave reduces the 32-double record with the transposing butterfly,
//                           the workgroup publishes a partial (write-through), and the last
//                           workgroup does the fixed-order final reduce + accept/revert + gain
//                           solve + state update.  One launch per iteration, no host round trip.
#pragma once

#include "fl_device.h"
#include "fl_math.h"
#include "lio_kernels.h"

// Photometric measurement for one pixel of one patch. All wave-uniform inputs are precomputed by
// the caller (same values in every lane). Operation order mirrors lidar_selection.cpp:826-837 so the
// float part rounds identically to the reference.
struct FlPatchGeom {
    double Jdpi[6];     // dpi(pf), :92-103
    double pf[3];
    float wtl, wtr, wbl, wbr;
    int u_i, v_i, scale;
};

// a patch's projection as the geometry batches of vio_produce stage it in LDS (96 bytes: six 16-byte reads per lane)
struct __attribute__((aligned(16))) FlGeomLds {
    double d[7];        // Jdpi[0], Jdpi[2], Jdpi[4], Jdpi[5], pf[0..2]
    double pad;
    float w[4];         // wtl, wtr, wbl, wbr
    int q[4];           // u_i, v_i, scale, -
};

// 1 / scale for scale = 2^k (a pyramid scale, 1 ... 2^30): exact in both precisions, so building it from the exponent gives the very
// bits of the IEEE divisions it replaces (~15 dependent instructions each in fp64 on the device) -- host build: the division itself
FL_HD double fl_inv_pow2_f64(int scale)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __hiloint2double((1023 - (__ffs(scale) - 1)) << 20, 0);
#else
    return 1.0 / (double)scale;
#endif
}
FL_HD float fl_inv_pow2_f32(int scale)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __int_as_float((127 - (__ffs(scale) - 1)) << 23);
#else
    return 1.0f / (float)scale;
#endif
}

FL_HD void fl_world2cam(const FlVioConst &c, const double *pf, double *pc)
{
    const double u = pf[0] / pf[2], v = pf[1] / pf[2];
    if (!c.distort) {
        pc[0] = c.fx * u + c.cx;
        pc[1] = c.fy * v + c.cy;
    } else {   // vk::PinholeCamera radtan (rpg_vikit, restated from memory -- see oracle/orc_vio.c)
        const double r2 = u * u + v * v, r4 = r2 * r2, r6 = r4 * r2;
        const double a1 = 2 * u * v, a2 = r2 + 2 * u * u, a3 = r2 + 2 * v * v;
        const double cdist = 1 + c.d[0] * r2 + c.d[1] * r4 + c.d[4] * r6;
        const double xd = u * cdist + c.d[2] * a1 + c.d[3] * a2;
        const double yd = v * cdist + c.d[3] * a1 + c.d[2] * a3;
        pc[0] = xd * c.fx + c.cx;
        pc[1] = yd * c.fy + c.cy;
    }
}

FL_HD void fl_patch_geom(const FlVioConst &c, const double *Rcw, const double *Pcw, const double *pos, int scale, FlPatchGeom &g)
{
    g.pf[0] = (Rcw[0] * pos[0] + Rcw[1] * pos[1] + Rcw[2] * pos[2]) + Pcw[0];
    g.pf[1] = (Rcw[3] * pos[0] + Rcw[4] * pos[1] + Rcw[5] * pos[2]) + Pcw[1];
    g.pf[2] = (Rcw[6] * pos[0] + Rcw[7] * pos[1] + Rcw[8] * pos[2]) + Pcw[2];
    double pc[2];
    fl_world2cam(c, g.pf, pc);
    const double z_inv = 1. / g.pf[2], z_inv_2 = z_inv * z_inv;
    g.Jdpi[0] = c.fx_abs * z_inv; g.Jdpi[1] = 0.0; g.Jdpi[2] = -c.fx_abs * g.pf[0] * z_inv_2;
    g.Jdpi[3] = 0.0; g.Jdpi[4] = c.fy_abs * z_inv; g.Jdpi[5] = -c.fy_abs * g.pf[1] * z_inv_2;
    const float u_ref = (float)pc[0], v_ref = (float)pc[1];
    g.scale = scale;
    // scale is a power of two: x / scale == x * (1 / scale) bit for bit (no rounding in either), one multiply instead of an IEEE
    // division sequence on the patch's critical path (lidar_selection.cpp:806-809 divides)
    const double inv_sd = fl_inv_pow2_f64(scale);
    const float inv_sf = fl_inv_pow2_f32(scale);
    g.u_i = (int)(floorf((float)(pc[0] * inv_sd)) * scale);
    g.v_i = (int)(floorf((float)(pc[1] * inv_sd)) * scale);
    const float su = (u_ref - g.u_i) * inv_sf;
    const float sv = (v_ref - g.v_i) * inv_sf;
    g.wtl = (float)((1.0 - su) * (1.0 - sv));
    g.wtr = (float)(su * (1.0 - sv));
    g.wbl = (float)((1.0 - su) * sv);
    g.wbr = su * sv;
}

// taps: t[r][c] = img[(row0 + (r-1)*scale) , (col0 + (c-1)*scale)], r,c in 0..3 (only 12 used)
FL_HD void fl_pixel_row(const FlPatchGeom &g, const float t[4][4], float ref, const double *Jdphi_dR, const double *Jdp_dR,
                        const double *Jdp_dt, double *row /*6*/, double *res_out)
{
    const float wtl = g.wtl, wtr = g.wtr, wbl = g.wbl, wbr = g.wbr;
    // t[1][1] = img_ptr[0]; columns: [0]=-s [1]=0 [2]=+s [3]=+2s ; rows: [0]=-sW [1]=0 [2]=+sW [3]=+2sW
    const float du = 0.5f * ((wtl * t[1][2] + wtr * t[1][3] + wbl * t[2][2] + wbr * t[2][3])
                           - (wtl * t[1][0] + wtr * t[1][1] + wbl * t[2][0] + wbr * t[2][1]));
    const float dv = 0.5f * ((wtl * t[2][1] + wtr * t[2][2] + wbl * t[3][1] + wbr * t[3][2])
                           - (wtl * t[0][1] + wtr * t[0][2] + wbl * t[1][1] + wbr * t[1][2]));
    const double inv_s = fl_inv_pow2_f64(g.scale);
    const double J0 = (double)du * inv_s, J1 = (double)dv * inv_s;
    double JJ[3], Jdphi[3], Jdp[3];
#pragma unroll
    for (int c = 0; c < 3; c++) JJ[c] = J0 * g.Jdpi[c] + J1 * g.Jdpi[3 + c];
    // p_hat = skew(pf): [0,-z,y; z,0,-x; -y,x,0]
    const double px = g.pf[0], py = g.pf[1], pz = g.pf[2];
    Jdphi[0] = JJ[0] * 0.0 + JJ[1] * pz + JJ[2] * (-py);
    Jdphi[1] = JJ[0] * (-pz) + JJ[1] * 0.0 + JJ[2] * px;
    Jdphi[2] = JJ[0] * py + JJ[1] * (-px) + JJ[2] * 0.0;
#pragma unroll
    for (int c = 0; c < 3; c++) Jdp[c] = (-J0) * g.Jdpi[c] + (-J1) * g.Jdpi[3 + c];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const double a = Jdphi[0] * Jdphi_dR[c] + Jdphi[1] * Jdphi_dR[3 + c] + Jdphi[2] * Jdphi_dR[6 + c];
        const double b = Jdp[0] * Jdp_dR[c] + Jdp[1] * Jdp_dR[3 + c] + Jdp[2] * Jdp_dR[6 + c];
        row[c] = a + b;
        row[3 + c] = Jdp[0] * Jdp_dt[c] + Jdp[1] * Jdp_dt[3 + c] + Jdp[2] * Jdp_dt[6 + c];
    }
    *res_out = (double)(wtl * t[1][1] + wtr * t[1][2] + wbl * t[2][1] + wbr * t[2][2] - ref);
}

// ---- per-patch factorisation of the photometric rows --------------------------------------------
// Every pixel row of a patch is  row = [du dv] * M  with ONE 2x6 matrix per patch
//     M = (1/scale) [ Jdpi (p_hat Jdphi_dR - Jdp_dR) | -Jdpi Jdp_dt ]          (lidar_selection.cpp:830-835)
// so the patch's contribution to the normal equations is  M^T G M  and  M^T g  with the 2x2 Gram
// matrix G = sum [du dv]^T [du dv] and g = sum [du dv]^T res over its 64 pixels. A lane therefore
// accumulates 6 numbers instead of forming a 1x6 row and 27 products; the wave reduces the 6 numbers
// and the (wave-uniform) 6x6 update is done once per patch. Algebraically identical to summing the
// rows; compared with the oracle by tolerance.
FL_HD void fl_patch_M(const FlPatchGeom &g, const double *Jdphi_dR, const double *Jdp_dR, const double *Jdp_dt, double (&M)[2][6])
{
    FL_FP_CONTRACT
    const double px = g.pf[0], py = g.pf[1], pz = g.pf[2];
    const double ph[9] = {0.0, -pz, py, pz, 0.0, -px, -py, px, 0.0};
    double B[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            B[i * 3 + j] = (ph[i * 3 + 0] * Jdphi_dR[0 * 3 + j] + ph[i * 3 + 1] * Jdphi_dR[1 * 3 + j] + ph[i * 3 + 2] * Jdphi_dR[2 * 3 + j]) - Jdp_dR[i * 3 + j];
    const double inv_s = fl_inv_pow2_f64(g.scale);
    const double a = g.Jdpi[0] * inv_s, c = g.Jdpi[2] * inv_s, b = g.Jdpi[4] * inv_s, d = g.Jdpi[5] * inv_s;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        M[0][j] = a * B[0 * 3 + j] + c * B[2 * 3 + j];
        M[1][j] = b * B[1 * 3 + j] + d * B[2 * 3 + j];
        M[0][3 + j] = -(a * Jdp_dt[0 * 3 + j] + c * Jdp_dt[2 * 3 + j]);
        M[1][3 + j] = -(b * Jdp_dt[1 * 3 + j] + d * Jdp_dt[2 * 3 + j]);
    }
}
// float part of one pixel, reference operation order (lidar_selection.cpp:826-829,837)
FL_HD void fl_pixel_grad(const FlPatchGeom &g, const float t[4][4], float ref, float *du_o, float *dv_o, float *res_o)
{
    const float wtl = g.wtl, wtr = g.wtr, wbl = g.wbl, wbr = g.wbr;
    *du_o = 0.5f * ((wtl * t[1][2] + wtr * t[1][3] + wbl * t[2][2] + wbr * t[2][3])
                  - (wtl * t[1][0] + wtr * t[1][1] + wbl * t[2][0] + wbr * t[2][1]));
    *dv_o = 0.5f * ((wtl * t[2][1] + wtr * t[2][2] + wbl * t[3][1] + wbr * t[3][2])
                  - (wtl * t[0][1] + wtr * t[0][2] + wbl * t[1][1] + wbr * t[1][2]));
    *res_o = wtl * t[1][1] + wtr * t[1][2] + wbl * t[2][1] + wbr * t[2][2] - ref;
}
// record += M^T G M, M^T gz, counts (all arguments wave-uniform)
FL_HD void fl_patch_accum(double *v /*32*/, const double (&M)[2][6], const double *T /*Suu,Suv,Svv,Sur,Svr,Srr*/)
{
    FL_FP_CONTRACT
    double GM0[6], GM1[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        GM0[j] = T[0] * M[0][j] + T[1] * M[1][j];
        GM1[j] = T[1] * M[0][j] + T[2] * M[1][j];
    }
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = i; j < 6; j++) { v[k] += M[0][i] * GM0[j] + M[1][i] * GM1[j]; k++; }
#pragma unroll
    for (int i = 0; i < 6; i++) v[21 + i] += M[0][i] * T[3] + M[1][i] * T[4];
    v[27] += 64.0;
    v[28] += T[5];
}

// Camera pose of the current state, Rcw = Rci Rwi^T, Pcw = -Rci Rwi^T Pwi + Pci (lidar_selection.cpp:780-784),
// in exactly the reference's operation order (it feeds the float sub-pixel weights). Threads 0..11 of
// the calling workgroup each form one element from xn = {rot(9), pos(3)}.
__device__ __forceinline__ void vio_derive_pose(const double *xn, const FlVioConst *__restrict__ VC, FlDev18 *__restrict__ D, int t)
{
    if (t < 9) {
        const int i = t / 3, j = t % 3;   // Rcw[i][j] = sum_k Rci[i][k] * Rwi[j][k]
        D->Rcw[t] = VC->Rci[i * 3 + 0] * xn[j * 3 + 0] + VC->Rci[i * 3 + 1] * xn[j * 3 + 1] + VC->Rci[i * 3 + 2] * xn[j * 3 + 2];
    } else if (t < 12) {
        const int i = t - 9;              // T = (-Rci) Rwi^T ; Pcw = T Pwi + Pci
        double T[3];
#pragma unroll
        for (int j = 0; j < 3; j++)
            T[j] = (-VC->Rci[i * 3 + 0]) * xn[j * 3 + 0] + (-VC->Rci[i * 3 + 1]) * xn[j * 3 + 1] + (-VC->Rci[i * 3 + 2]) * xn[j * 3 + 2];
        D->Pcw[i] = (T[0] * xn[9] + T[1] * xn[10] + T[2] * xn[11]) + VC->Pci[i];
    }
}
__global__ void vio_derive_kernel(FlDev18 *__restrict__ D, const FlVioConst *__restrict__ VC)
{
    __shared__ double xn[12];
    if (threadIdx.x < 12) xn[threadIdx.x] = D->x[threadIdx.x];
    __syncthreads();
    vio_derive_pose(xn, VC, D, (int)threadIdx.x);
}
// fl_vio_begin in one launch: the gain-solve constants of the state block (eskf18_prepare_kernel) and, by the last 12 threads, the
// camera pose of the initial state
// x18_host != nullptr (fl_vio_compute_j, round 6): the state block waits in the handle's page-locked mirror and this kernel fetches it itself
// (as vmap_frame_init_kernel does for fl_vio_detect) -- one copy command (4-6 us of stream time) less in front of ComputeJ
__global__ __launch_bounds__(128) void vio_prepare_kernel(FlDev18 *__restrict__ D, const FlVioConst *__restrict__ VC, const FlDev18 *__restrict__ x18_host)
{
    if (x18_host) {      // (uniform)
        FlPull<128, (int)sizeof(FlDev18)> blk;
        blk.load(x18_host);
        blk.store(D);
        __threadfence_block();
        __syncthreads();
    }
    if (threadIdx.x >= 116) vio_derive_pose(D->x, VC, D, (int)threadIdx.x - 116);
    eskf18_prepare_body(D);
}

#define FL_VIO_NT 256
// Lanes per patch: 32 (a half-wave per patch, 2 pixels per lane) or 16 (a 16-lane DPP row per patch, 4 pixels per lane, 4 patches per
// wavefront). The work that is uniform over a patch -- projection with its fp64 divisions, the 2x6 matrix M, the 6x6 update -- is
// executed by the whole wavefront whatever the number of patches it serves, so 4 patches per wavefront halve it again, halve the
// number of producer workgroups (125 records instead of 250 at 2000 patches) and shorten the per-patch cross-lane reduction by a stage.
#ifndef FL_VIO_LPP
#define FL_VIO_LPP 16
#endif
#ifdef FL_AUDIT_STAMPS                      /* debug build: time line of the auditor and the solver (tools/vio_audit_stamps.py) */
#define FL_AUDIT_STAMP(i, v) do { if (threadIdx.x == 0) g_fl_wall[(i)] = (long long)(v); } while (0)
#else
#define FL_AUDIT_STAMP(i, v) do { } while (0)
#endif
#define FL_VIO_SOLVER_BLOCK(nprod) (nprod)
#define FL_VIO_AUDITOR_BLOCK(nprod) ((nprod) + 1)
#define FL_VIO_PPL (64 / FL_VIO_LPP)          /* pixels per lane */
#define FL_VIO_CHAIN_BATCH 4                 /* iterations whose per-patch float chains run together (vio_produce) */
#define FL_VIO_GPW (64 / FL_VIO_LPP)          /* patches (lane groups) per wavefront */

// grid = producers + 2 (the auditor workgroup, then the solver workgroup) ; MODE 0: fused pass ; MODE 1: accumulate only (sums -> sums_out).
// 256-thread workgroups: 8 (16) patches each. The first patch's inputs do not depend on the state: their loads are issued before
// the state round trip.
struct FlVioFirst {
    int slevel;
    double pos0, pos1, pos2;
    float ref[FL_VIO_PPL];
    bool have;
};
__device__ __forceinline__ FlVioFirst vio_prefetch_first(const float *__restrict__ ref, const double *__restrict__ pos,
                                                        const int32_t *__restrict__ slevel, int m, int level_arg, int nprod)
{
    constexpr int WPB = FL_VIO_NT / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / FL_VIO_LPP, hl = lane % FL_VIO_LPP;
    const int i_first = (blockIdx.x * WPB + wave) * FL_VIO_GPW + grp;
    FlVioFirst f;
    f.have = ((int)blockIdx.x < nprod) && (i_first < m);
    f.slevel = 0; f.pos0 = 0.0; f.pos1 = 0.0; f.pos2 = 0.0;
#pragma unroll
    for (int k = 0; k < FL_VIO_PPL; k++) f.ref[k] = 0.f;
    if (f.have) {
        f.slevel = slevel[i_first];
        f.pos0 = pos[i_first * 3 + 0]; f.pos1 = pos[i_first * 3 + 1]; f.pos2 = pos[i_first * 3 + 2];
        if (level_arg >= 0) {
#pragma unroll
            for (int k = 0; k < FL_VIO_PPL; k++) f.ref[k] = ref[(size_t)i_first * 192 + 64 * level_arg + hl + FL_VIO_LPP * k];
        }
    }
    return f;
}

// float patch_error of one patch from its 64 residuals in LDS (pixel order x*8+y), by the first lane of the patch's lane group
__device__ __forceinline__ void vio_patch_error(const float *r, bool mine, int i, float *__restrict__ errors,
                                                unsigned long long *__restrict__ err_words, unsigned epoch)
{
    if (!mine) return;
    float pe = 0.0f;
#pragma unroll
    for (int k = 0; k < 64; k += 16) {               // operands from LDS sixteen at a time: the chain waits for the adder only
        float q[16];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float4 t = *reinterpret_cast<const float4 *>(r + k + 4 * j);
            q[4 * j] = t.x; q[4 * j + 1] = t.y; q[4 * j + 2] = t.z; q[4 * j + 3] = t.w;
        }
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const double rd = (double)q[j];
            pe = (float)fma(rd, rd, (double)pe);     // the product of two floats is exact in double: fused or not, same rounding
        }
    }
    errors[i] = pe;
    if (err_words)
        __hip_atomic_store(err_words + i, ((unsigned long long)__float_as_uint(pe) << 32) | (unsigned long long)epoch, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}

// Totals of 6 values over the 16 lanes of a DPP row, in every lane of the row: a cyclic all-reduce (rotate by 8, by 4, then the
// two quad exchanges) -- a fixed order, no selects, no final shuffles.
#define FL_DPP_ROW_ROR4 0x124
__device__ __forceinline__ void row_sum6(double (&w)[8], double (&T)[6])
{
#pragma unroll
    for (int k = 0; k < 6; k++) {
        double x = w[k];
        x = x + dpp_f64<FL_DPP_ROW_ROR8>(x);
        x = x + dpp_f64<FL_DPP_ROW_ROR4>(x);
        x = x + dpp_f64<FL_DPP_QUAD_XOR2>(x);
        x = x + dpp_f64<FL_DPP_QUAD_XOR1>(x);
        T[k] = x;
    }
}

// ---- the 6x6 update of a patch, spread over the 16 lanes of its DPP row (round 3) ------------------------------------------
// The patch's contribution is M^T G M (21 numbers), M^T g (6), the pixel count and sum res^2: 29 outputs of a 2x6 matrix M and six
// row totals T. Every lane used to form all of M (60 fp64 instructions) and all 29 outputs (110) into a 32-double accumulator; now
//   * lane e < 12 forms ONE entry M(e / 6, e % 6) from per-lane constants set up once per pass and drops it into LDS,
//   * lane hl forms outputs hl and hl + 16 of the record from the four / eight entries of M they need (read back from LDS) and
//     keeps TWO accumulators,
//   * at the end of the pass the four rows of a wavefront are added up with two lane swaps -- value index == lane -- and the
//     wavefronts through LDS.
// Same algebra as fl_patch_M / fl_patch_accum (M^T G M as sum over the pixels of row_i row_j, row = [du dv] M); fused multiply-adds;
// compared with the oracle by tolerance like every fp64 sum.
#define FL_VIO_MROW 20                   /* doubles per lane group in the M exchange: M0[0..5], M1[0..5], 1.0 at 12, 0.0 at 18 */
struct FlVioLaneRole {
    double X1, X2, X3, Y1, Y2, Y3;      // B_r = q1 X1 + q2 X2 - X3, B_2 = -py Y1 + px Y2 - Y3 for this lane's column of M
    int r;                              // row of M this lane forms
    int o1a, o1b, o2a, o2b;             // entries (column i, column j) of outputs 1 and 2 inside the lane group's FL_VIO_MROW doubles
    int kind2;                          // output 2: 0 H^T H entry, 1 H^T z entry, 2 pixel count, 3 sum res^2, 4 none
};
__device__ __forceinline__ void fl_vio_upper_ij(int k, int &i, int &j)       // record index k < 21 -> (i, j), i <= j, row-major
{
    int rowlen = 6;
    i = 0;
    while (k >= rowlen) { k -= rowlen; rowlen--; i++; }
    j = i + k;
}
// Once per launch; the three entries that follow the pose (Jdp_dt = Rcw) are refreshed per pass by fl_vio_lane_role_pose. The
// lane-dependent entries come straight from memory (VC: the constants in HBM; cam: Rcw, D->Rcw or the broadcast pose in LDS):
// selecting them out of the by-value copies would index a register array at run time, i.e. put it in scratch.
__device__ __forceinline__ FlVioLaneRole fl_vio_lane_role(int hl, const FlVioConst *__restrict__ VC)
{
    FlVioLaneRole R;
    const int e = hl < 12 ? hl : 0, col = e % 6;
    R.r = e / 6;
    R.X1 = 0.0; R.X2 = 0.0; R.X3 = 0.0; R.Y1 = 0.0; R.Y2 = 0.0; R.Y3 = 0.0;
    if (col < 3) {
        const double A0 = VC->Jdphi_dR[col], A1 = VC->Jdphi_dR[3 + col], A2 = VC->Jdphi_dR[6 + col];
        const double D0 = VC->Jdp_dR[col], D1 = VC->Jdp_dR[3 + col], D2 = VC->Jdp_dR[6 + col];
        R.X1 = R.r ? A0 : A1; R.X2 = A2; R.X3 = R.r ? D1 : D0;
        R.Y1 = A0; R.Y2 = A1; R.Y3 = D2;
    }
    int i, j;
    fl_vio_upper_ij(hl, i, j);            // output 1: record index hl (< 21: an entry of H^T H)
    R.o1a = i; R.o1b = j;
    const int k2 = hl + 16;
    R.o2a = 0; R.o2b = 0;
    if (k2 < 21) { fl_vio_upper_ij(k2, i, j); R.o2a = i; R.o2b = j; R.kind2 = 0; }
    else if (k2 < 27) { R.o2a = k2 - 21; R.o2b = 12; R.kind2 = 1; }       // column "12": M0 = 1, M1 = 0
    else R.kind2 = (k2 == FL_S_NEFF) ? 2 : ((k2 == FL_S_RES) ? 3 : 4);
    return R;
}
__device__ __forceinline__ void fl_vio_lane_role_pose(FlVioLaneRole &R, int hl, const double *cam /* Rcw, row-major */)
{
    const int e = hl < 12 ? hl : 0, col = e % 6;
    if (col >= 3) {
        const int cc = col - 3;
        R.X3 = cam[(R.r ? 3 : 0) + cc];
        R.Y3 = cam[6 + cc];
    }
}

// One producer workgroup's share of a pass: residuals, rows, 6x6 update for its patches, reduced to one record and published.
// CB: iterations whose per-patch float chains run together (FL_VIO_CHAIN_BATCH in the per-pass kernels, which iterate over many
// patches at scale; 1 in the multi-pass kernels, whose wavefronts have one iteration each -- the bookkeeping cost them 0.3 us per pass).
template <int CB>
__device__ __forceinline__ void vio_produce(const uint8_t *__restrict__ img, const float *__restrict__ ref, const double *__restrict__ pos,
                                            const int32_t *__restrict__ slevel, float *__restrict__ errors, int m, int level_arg, int level,
                                            const FlVioConst &vc, const double (&Rcw)[9], const double (&Pcw)[3], const FlVioFirst &pf,
                                            int nprod, double *s_red, unsigned epoch, void *__restrict__ records, int flags,
                                            unsigned long long *__restrict__ err_words /* this pass's half, nullable */,
                                            float *s_res /* LDS: GPW * WPB * CB * 64 */, int *s_pidx /* LDS: GPW * WPB * CB */,
                                            const FlVioLaneRole &role /* fl_vio_lane_role + fl_vio_lane_role_pose of this pass */)
{
    constexpr int WPB = FL_VIO_NT / 64;
    constexpr int LPP = FL_VIO_LPP, PPL = FL_VIO_PPL, GPW = FL_VIO_GPW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane / LPP, hl = lane % LPP;
    const int slot = wave * GPW + grp;            // this lane group's slot in the workgroup's LDS arrays
    const int i_first = (blockIdx.x * WPB + wave) * GPW + grp;
    const bool have_first = pf.have;
    const int pf_slevel = pf.slevel;
    const double pf_pos0 = pf.pos0, pf_pos1 = pf.pos1, pf_pos2 = pf.pos2;
    const int xr = hl >> 3, yc = hl & 7;          // this lane's pixels: (xr + (LPP / 8) k, yc), k = 0 .. PPL-1
    const int W = vc.stride, Hm1 = vc.height - 1, Wm1 = vc.width - 1;
    // the tap rows of the fast paths come through a buffer descriptor: per-lane byte offset (one VGPR) + a scalar row offset -- no
    // 64-bit pointer arithmetic per row (10 rows x ~6 vector instructions per iteration before). Needs rows that keep the dword phase
    // of a lane's first row (stride % 4 == 0: true for every shipped camera); other strides take the byte-tap path.
    const __amdgpu_buffer_rsrc_t img_rs = __builtin_amdgcn_make_buffer_rsrc((void *)img, 0, W * vc.height + 16, 0x00020000);
    const bool rows_dword_phase = (W & 3) == 0;

    static_assert(FL_VIO_LPP == 16, "the lane-distributed 6x6 update maps one patch to one 16-lane DPP row");
    double *s_M = s_red + slot * FL_VIO_MROW;                     // this lane group's M exchange (GPW * WPB * FL_VIO_MROW doubles)
    double *s_fin = s_red + GPW * WPB * FL_VIO_MROW;              // cross-wavefront sum of the record: WPB * 32 doubles
    if (hl == 12) s_M[12] = 1.0;
    if (hl == 13) s_M[18] = 0.0;
    double acc1 = 0.0, acc2 = 0.0;

    FL_INSTR(if (blockIdx.x == 0) fl_stamp(flags, 0);)
    // trip count uniform over the wave: all lane groups iterate together, an inactive group (m not a multiple of GPW) computes
    // on patch 0 and contributes nothing
    // The per-patch float chains (vio_patch_error: 64 dependent steps, ~250 instructions) run on ONE lane per patch, i.e. with 4 of
    // the 64 lanes when they follow every iteration -- a third of the loop's instructions at 1 M patches. They are batched: the
    // residuals of up to CB iterations wait in LDS and lanes hl < batch of every row each take one of them, so a
    // chain pass serves 16 patches. The last batch is deferred until the record is published (off the hand-off's critical path).
    int bslot = 0;                                   // iterations waiting in the batch
    // Geometry batches (GB > 1, the per-pass kernels that iterate over many patches): the projection of a patch -- three fp64
    // divisions, the radtan polynomial, dpi, the sub-pixel weights: ~130 of the loop's ~610 vector instructions -- is the same for the
    // 16 lanes of its row, i.e. it ran with 4 useful lanes out of 64. Every GB iterations the wavefront therefore projects the 64
    // patches of its next GB iterations at once, one patch per lane (lane 4k + g: iteration k, lane group g), into LDS, and an
    // iteration reads its patch's 96 bytes back (broadcast reads: the 16 lanes of a row ask for the same address). The arithmetic
    // is fl_patch_geom's on another lane: the same bits. GB = 1 (multi-pass kernels: one iteration per wavefront and pass): as before.
    constexpr int GB = (CB > 1) ? 16 : 1;
    const int stride_it = nprod * WPB * GPW;
    for (int ib0 = (blockIdx.x * WPB + wave) * GPW; ib0 < m; ib0 += stride_it * GB) {
    FlGeomLds *geom_w = nullptr;
    if constexpr (GB > 1) {
        __shared__ __attribute__((aligned(16))) FlGeomLds s_geom[WPB][64];
        geom_w = s_geom[wave];
        const int gi = ib0 + (lane >> 2) * stride_it + (lane & 3);
        const int gii = gi < m ? gi : 0;
        const int gscale = 1 << (level + slevel[gii]);
        const double gps[3] = {pos[gii * 3 + 0], pos[gii * 3 + 1], pos[gii * 3 + 2]};
        FlPatchGeom gg;
        fl_patch_geom(vc, Rcw, Pcw, gps, gscale, gg);
        __builtin_amdgcn_wave_barrier();                       // the rows have read the previous batch
        FlGeomLds o;
        o.d[0] = gg.Jdpi[0]; o.d[1] = gg.Jdpi[2]; o.d[2] = gg.Jdpi[4]; o.d[3] = gg.Jdpi[5]; o.d[4] = gg.pf[0]; o.d[5] = gg.pf[1]; o.d[6] = gg.pf[2];
        o.w[0] = gg.wtl; o.w[1] = gg.wtr; o.w[2] = gg.wbl; o.w[3] = gg.wbr;
        o.q[0] = gg.u_i; o.q[1] = gg.v_i; o.q[2] = gg.scale; o.q[3] = 0;
        o.pad = 0.0;
        geom_w[lane] = o;
        __builtin_amdgcn_wave_barrier();
    }
    for (int kk = 0; kk < GB; kk++) {
        const int ib = ib0 + kk * stride_it;
        if (ib >= m) break;                                    // (uniform over the wavefront)
        const int i = ib + grp;
        const bool active = i < m;
        const int ii = active ? i : 0;
        const bool first = (i == i_first) && have_first;
        FlPatchGeom g;
        int scale;
        if constexpr (GB > 1) {
            const FlGeomLds gl = geom_w[kk * 4 + grp];
            g.Jdpi[0] = gl.d[0]; g.Jdpi[1] = 0.0; g.Jdpi[2] = gl.d[1]; g.Jdpi[3] = 0.0; g.Jdpi[4] = gl.d[2]; g.Jdpi[5] = gl.d[3];
            g.pf[0] = gl.d[4]; g.pf[1] = gl.d[5]; g.pf[2] = gl.d[6];
            g.wtl = gl.w[0]; g.wtr = gl.w[1]; g.wbl = gl.w[2]; g.wbr = gl.w[3];
            g.u_i = gl.q[0]; g.v_i = gl.q[1]; g.scale = gl.q[2];
            scale = gl.q[2];
        } else {
            scale = 1 << (level + (first ? pf_slevel : slevel[ii]));
            double ps[3];
            if (first) { ps[0] = pf_pos0; ps[1] = pf_pos1; ps[2] = pf_pos2; }
            else { ps[0] = pos[ii * 3 + 0]; ps[1] = pos[ii * 3 + 1]; ps[2] = pos[ii * 3 + 2]; }
            fl_patch_geom(vc, Rcw, Pcw, ps, scale, g);
        }
        FL_INSTR(if ((flags & FL_ITER_STAMP) && blockIdx.x == 0) { asm volatile("" ::"v"(g.wbr), "v"(g.u_i)); fl_stamp(flags, 40); })
        const int col0 = g.u_i + (yc - 4) * scale;
        float t[PPL][4][4];
        // taps span [anchor - 5*scale, anchor + 5*scale]: no clamping needed inside the image
        const bool inside = (g.v_i - 5 * scale >= 0) && (g.v_i + 5 * scale <= Hm1) && (g.u_i - 5 * scale >= 0) && (g.u_i + 5 * scale <= Wm1);
        // Every patch of the wavefront inside the image and at the same pyramid scale (the common case): a lane's four pixels sit
        // two rows apart in one column, so at the finest level their 4 x 4 tap windows are 10 image rows x 4 consecutive bytes: ten
        // aligned 8-byte loads + v_alignbyte instead of 48 single-byte loads (each a 64-address instruction for the texture addresser).
        // Coarser pyramid levels (scale 2, 4: the first two levels of ComputeJ) read the same 10 rows at stride scale: the four
        // taps of a row are bytes 0, s, 2s, 3s of a 7- / 13-byte span -- three / four aligned dwords per row and byte shifts.
        const int scale_u = __builtin_amdgcn_readfirstlane(scale);
        const bool fast_taps = (LPP == 16) && rows_dword_phase && (scale_u == 1 || scale_u == 2 || scale_u == 4) && (__ballot(inside && scale == scale_u) == ~0ull);
        if (fast_taps && scale_u == 1) {          // (one straight-line path per scale: a branch per row cost the finest level 1.3 us)
            const int off0 = (g.v_i + xr - 5) * W + (col0 - 1);       // >= 0: the patch is inside the image
            const unsigned sh = (unsigned)off0 & 3u;
            const int voff = off0 & ~3;
            unsigned rowv[10];
#pragma unroll
            for (int m2 = 0; m2 < 10; m2++) {
                const fl_u2 w2 = __builtin_bit_cast(fl_u2, __builtin_amdgcn_raw_buffer_load_b64(img_rs, voff, m2 * W, 0));   // (dword-aligned; the frame buffer is padded: api_vio.inc)
                rowv[m2] = __builtin_amdgcn_alignbyte(w2.y, w2.x, sh);
            }
#pragma unroll
            for (int px = 0; px < PPL; px++)
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const bool used = !((r == 0 && (c == 0 || c == 3)) || (r == 3 && (c == 0 || c == 3)));
                        t[px][r][c] = used ? (float)((rowv[2 * px + r] >> (8 * c)) & 0xffu) : 0.f;
                    }
        } else if (fast_taps && scale_u == 2) {
            const int off0 = (g.v_i + (xr - 5) * 2) * W + (col0 - 2);
            const unsigned sh = (unsigned)off0 & 3u;
            const int voff = off0 & ~3;
            unsigned lo[10], hi[10];               // bytes 0, 2 of lo: taps 0, 1; of hi: taps 2, 3
#pragma unroll
            for (int m2 = 0; m2 < 10; m2++) {
                typedef unsigned int fl_u3 __attribute__((ext_vector_type(3)));
                const fl_u3 w3 = __builtin_bit_cast(fl_u3, __builtin_amdgcn_raw_buffer_load_b96(img_rs, voff, m2 * 2 * W, 0));
                lo[m2] = __builtin_amdgcn_alignbyte(w3.y, w3.x, sh);
                hi[m2] = __builtin_amdgcn_alignbyte(w3.z, w3.y, sh);
            }
#pragma unroll
            for (int px = 0; px < PPL; px++)
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const bool used = !((r == 0 && (c == 0 || c == 3)) || (r == 3 && (c == 0 || c == 3)));
                        const unsigned wv = (c < 2) ? lo[2 * px + r] : hi[2 * px + r];
                        t[px][r][c] = used ? (float)((wv >> (16 * (c & 1))) & 0xffu) : 0.f;
                    }
        } else if (fast_taps) {                   // scale 4
            const int off0 = (g.v_i + (xr - 5) * 4) * W + (col0 - 4);
            const unsigned sh = (unsigned)off0 & 3u;
            const int voff = off0 & ~3;
            unsigned tv[10][4];
#pragma unroll
            for (int m2 = 0; m2 < 10; m2++) {
                const fl_u4 w4 = __builtin_bit_cast(fl_u4, __builtin_amdgcn_raw_buffer_load_b128(img_rs, voff, m2 * 4 * W, 0));
                tv[m2][0] = __builtin_amdgcn_alignbyte(w4.y, w4.x, sh); tv[m2][1] = __builtin_amdgcn_alignbyte(w4.z, w4.y, sh);
                tv[m2][2] = __builtin_amdgcn_alignbyte(w4.w, w4.z, sh); tv[m2][3] = w4.w >> (8 * sh);
            }
#pragma unroll
            for (int px = 0; px < PPL; px++)
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const bool used = !((r == 0 && (c == 0 || c == 3)) || (r == 3 && (c == 0 || c == 3)));
                        t[px][r][c] = used ? (float)(tv[2 * px + r][c] & 0xffu) : 0.f;
                    }
        } else {
#pragma unroll
        for (int px = 0; px < PPL; px++) {
            const int row0 = g.v_i + (xr + (LPP / 8) * px - 4) * scale;
            if (inside) {
                const uint8_t *q = img + row0 * W + col0;
#pragma unroll
                for (int r = 0; r < 4; r++)
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const bool used = !((r == 0 && (c == 0 || c == 3)) || (r == 3 && (c == 0 || c == 3)));
                        t[px][r][c] = used ? (float)q[(r - 1) * scale * W + (c - 1) * scale] : 0.f;
                    }
            } else {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    int rr = row0 + (r - 1) * scale;
                    rr = rr < 0 ? 0 : (rr > Hm1 ? Hm1 : rr);  // the reference reads unchecked; clamp instead of faulting
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const bool used = !((r == 0 && (c == 0 || c == 3)) || (r == 3 && (c == 0 || c == 3)));
                        if (used) {
                            int cc = col0 + (c - 1) * scale;
                            cc = cc < 0 ? 0 : (cc > Wm1 ? Wm1 : cc);
                            t[px][r][c] = (float)img[rr * W + cc];
                        } else {
                            t[px][r][c] = 0.f;
                        }
                    }
                }
            }
        }
        }
        float refv[PPL];
        if (first && level_arg >= 0) {
#pragma unroll
            for (int k = 0; k < PPL; k++) refv[k] = pf.ref[k];
        } else {
#pragma unroll
            for (int k = 0; k < PPL; k++) refv[k] = ref[(size_t)ii * 192 + 64 * level + hl + LPP * k];
        }
        FL_INSTR(if ((flags & FL_ITER_STAMP) && blockIdx.x == 0) { asm volatile("" ::"v"(t[0][1][1] + t[PPL - 1][2][2] + refv[PPL - 1])); fl_stamp(flags, 41); })
        {   // this lane's entry of M (lanes 0..11), overlaps the tap loads. Explicit fma() in a fixed form: the kernel variants (launch
            // bounds) must produce the same bits, and a contraction left to the compiler differs between instantiations.
            const double inv_s = fl_inv_pow2_f64(g.scale);
            const double s1 = (role.r ? g.Jdpi[4] : g.Jdpi[0]) * inv_s, s2 = (role.r ? g.Jdpi[5] : g.Jdpi[2]) * inv_s;
            const double q1 = role.r ? g.pf[2] : -g.pf[2], q2 = role.r ? -g.pf[0] : g.pf[1];
            const double Br = fma(q1, role.X1, fma(q2, role.X2, -role.X3));
            const double B2 = fma(-g.pf[1], role.Y1, fma(g.pf[0], role.Y2, -role.Y3));
            const double Me = fma(s1, Br, s2 * B2);
            if (hl < 12) s_M[hl] = Me;
        }
        FL_INSTR(if ((flags & FL_ITER_STAMP) && blockIdx.x == 0) { fl_stamp(flags, 42); })
        double w8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        // the float part two pixels at a time (v_pk_mul_f32 / v_pk_add_f32: IEEE per element, no contraction -- the same roundings as the
        // scalar expressions of fl_pixel_grad, lidar_selection.cpp:826-829,837), the Gram sums in fp64 with fused multiply-adds
        // (compared by tolerance)
#pragma unroll
        for (int px = 0; px < PPL; px += 2) {
            typedef float fl_f2 __attribute__((ext_vector_type(2)));
            fl_f2 tt[4][4];
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int c = 0; c < 4; c++) { tt[r][c].x = t[px][r][c]; tt[r][c].y = t[px + 1][r][c]; }
            const fl_f2 wtl = {g.wtl, g.wtl}, wtr = {g.wtr, g.wtr}, wbl = {g.wbl, g.wbl}, wbr = {g.wbr, g.wbr};
            const fl_f2 half = {0.5f, 0.5f};
            const fl_f2 rf = {refv[px], refv[px + 1]};
            const fl_f2 du2 = half * ((wtl * tt[1][2] + wtr * tt[1][3] + wbl * tt[2][2] + wbr * tt[2][3])
                                    - (wtl * tt[1][0] + wtr * tt[1][1] + wbl * tt[2][0] + wbr * tt[2][1]));
            const fl_f2 dv2 = half * ((wtl * tt[2][1] + wtr * tt[2][2] + wbl * tt[3][1] + wbr * tt[3][2])
                                    - (wtl * tt[0][1] + wtr * tt[0][2] + wbl * tt[1][1] + wbr * tt[1][2]));
            const fl_f2 rs2 = wtl * tt[1][1] + wtr * tt[1][2] + wbl * tt[2][1] + wbr * tt[2][2] - rf;
            s_res[(slot * CB + bslot) * 64 + LPP * px + hl] = rs2.x;       // pixel order of the reference: x * 8 + y
            s_res[(slot * CB + bslot) * 64 + LPP * (px + 1) + hl] = rs2.y;
#pragma unroll
            for (int e = 0; e < 2; e++) {
                const double dud = (double)(e ? du2.y : du2.x), dvd = (double)(e ? dv2.y : dv2.x), res = (double)(e ? rs2.y : rs2.x);
                w8[0] = fma(dud, dud, w8[0]); w8[1] = fma(dud, dvd, w8[1]); w8[2] = fma(dvd, dvd, w8[2]);
                w8[3] = fma(dud, res, w8[3]); w8[4] = fma(dvd, res, w8[4]); w8[5] = fma(res, res, w8[5]);
            }
        }
        FL_INSTR(if ((flags & FL_ITER_STAMP) && blockIdx.x == 0) { asm volatile("" ::"v"(w8[5] + w8[0])); fl_stamp(flags, 43); })
        double T6[6];
        if (LPP == 32) half_sum6(w8, lane, T6); else row_sum6(w8, T6);
        FL_INSTR(if ((flags & FL_ITER_STAMP) && blockIdx.x == 0) { asm volatile("" ::"v"(T6[5] + T6[0])); fl_stamp(flags, 44); })
        // patch_error exactly as the reference rounds it (lidar_selection.cpp:849: float patch_error; patch_error += res*res with a
        // double res): one lane per patch replays the 64 additions in pixel order (vio_patch_error). It feeds only the errors[]
        // output and the rare exact accept test, never the record: for the wave's LAST patch group it is deferred until the record is
        // published, off the hand-off's critical path (the residuals wait in LDS).
        const bool last_iter = (ib + stride_it >= m);
        {   // this lane's two outputs of the patch (explicit fma(), see above)
            __builtin_amdgcn_wave_barrier();
            const double Ma1 = s_M[role.o1a], Mb1 = s_M[6 + role.o1a], Mc1 = s_M[role.o1b], Md1 = s_M[6 + role.o1b];
            const double Ma2 = s_M[role.o2a], Mb2 = s_M[6 + role.o2a], Mc2 = s_M[role.o2b], Md2 = s_M[6 + role.o2b];
            __builtin_amdgcn_wave_barrier();
            const double out1 = fma(Ma1, fma(T6[0], Mc1, T6[1] * Md1), Mb1 * fma(T6[1], Mc1, T6[2] * Md1));
            const bool htz = role.kind2 == 1;
            const double ta = htz ? T6[3] : T6[0], tb2 = htz ? T6[4] : T6[1];
            double out2 = fma(Ma2, fma(ta, Mc2, T6[1] * Md2), Mb2 * fma(tb2, Mc2, T6[2] * Md2));
            if (role.kind2 >= 2) out2 = (role.kind2 == 2) ? 64.0 : ((role.kind2 == 3) ? T6[5] : 0.0);
            if (active) { acc1 += out1; acc2 += out2; }
        }
        FL_INSTR(if ((flags & FL_ITER_STAMP) && blockIdx.x == 0) { asm volatile("" ::"v"(acc1 + acc2)); fl_stamp(flags, 45); })
        if (hl == 0) s_pidx[slot * CB + bslot] = active ? i : -1;
        bslot++;
        if (bslot == CB && !last_iter) {
            __builtin_amdgcn_wave_barrier();
            const int pi = s_pidx[slot * CB + (hl < CB ? hl : 0)];
            vio_patch_error(s_res + (slot * CB + hl) * 64, hl < CB && pi >= 0, pi, errors, err_words, epoch);
            __builtin_amdgcn_wave_barrier();
            bslot = 0;
        }
    }
    }
    FL_INSTR(if (blockIdx.x == 0) fl_stamp(flags, 1);)
    // lane hl of every row holds the row's outputs hl and hl + 16: the four rows of a wavefront with two lane swaps (even rows then
    // hold index hl, odd rows index hl + 16, i.e. lane L < 32 holds the wavefront's total of record value L), the wavefronts via LDS
    swap16_f64(acc1, acc2);
    double c = acc1 + acc2;
    {
        double c2 = c;
        swap32_f64(c, c2);
        c = c + c2;
    }
    if (lane < 32) s_fin[wave * 32 + lane] = c;
    __syncthreads();
    double mine = 0.0;
    if (threadIdx.x < FL_SUMS18) {
        mine = s_fin[threadIdx.x];
#pragma unroll
        for (int w = 1; w < WPB; w++) mine += s_fin[w * FL_SUMS18 + threadIdx.x];
    }
    publish_record<FL_SUMS18>(mine, epoch, records);
    FL_INSTR(if (blockIdx.x == 0) fl_stamp(flags, 3);)
    FL_INSTR(if ((flags & FL_ITER_STAMP) && threadIdx.x == 0 && blockIdx.x < 1024) g_fl_wall[blockIdx.x] = (long long)wall_clock64();)   // every producer's publish time
    __builtin_amdgcn_wave_barrier();
    {
        const int pi = s_pidx[slot * CB + (hl < bslot ? hl : 0)];
        vio_patch_error(s_res + (slot * CB + hl) * 64, hl < bslot && pi >= 0, pi, errors, err_words, epoch);
    }
}

// ---- the at-scale producer: ONE PATCH PER LANE (round 5) -----------------------------------------------------------------------
// vio_produce gives a patch to the 16 lanes of a DPP row, which is what a 2 000-patch pass wants (125 workgroups with something to
// do, one iteration each: the pass is a chain of hand-offs). A pass over 10^4 .. 10^6 patches has no such concern and pays for the
// spreading (tools/vio_pmc.sh: 438 vector instructions per wavefront iteration = 110 per patch, 60 % VALU utilisation at two wavefronts
// per SIMD); most of those instructions exist only because a patch is spread over lanes:
//   * every pixel fetches and converts its own 12 taps and forms its own 5 bilinear values (centre, left, right, up, down): 768 tap
//     conversions and 320 interpolations per patch, where the patch has 121 taps and 96 distinct bilinear values -- the value right
//     of pixel (x, y) IS the centre value of pixel (x, y + 1), the same expression over the same operands, hence the same bits;
//   * the six Gram sums cross the 16 lanes (72 DPP instructions per iteration), the 2x6 matrix M and the 29 outputs go through LDS,
//     the per-patch float chain runs with a quarter of the lanes.
// Here a lane walks its own patch: 11 tap rows of 11 bytes (tap scale 1: one 16-byte load per row; scales 2 and 4: 24 / 44 bytes per row
// and a v_perm_b32 gather; all three end in the same three packed words per row), a rolling window of three
// rows of bilinear values, pixels in the reference's order x * 8 + y -- so the float chain `patch_error += res * res` is simply the
// lane's own running value --, the Gram sums in the lane's registers, M and the 29 outputs once per patch with all 64 lanes busy
// (fl_patch_M / fl_patch_accum: the round-1 arithmetic). No cross-lane traffic until the record is reduced at the end of the pass.
// The reference patches (256 contiguous bytes per patch and level) are fetched by the wavefront together -- 16 loads of 1 KB, four
// patches each, straight into LDS (global_load_lds_dwordx4: no staging registers) -- and read back by their lanes.
// Float part: the reference's expressions and operand order (lidar_selection.cpp:826-829,837), no contraction: per-patch errors are
// bit-identical to vio_produce's and to the oracle's. fp64 sums: other order than vio_produce, compared by tolerance like every sum.
// (40.7 vector instructions per patch; the pass is then bound by its memory requests and by the slower of the two workgroups of a CU:
// DESIGN.md 4.2.1.)
// One tap row of a patch that is not served by the row loads (a wavefront whose patches are at different pyramid scales, a patch
// reaching over the image border, a row stride that is not a multiple of 4): 11 bytes at column c0 + b * scale of image row `row`,
// packed like the row loads' words. The reference reads unchecked; rows and columns are clamped instead of faulting (vio_produce does
// the same). Out of line: 11 x 11 inlined copies of the clamped address arithmetic cost the common path its registers.
typedef unsigned int fl_u3 __attribute__((ext_vector_type(3)));
__device__ __attribute__((noinline)) fl_u3 vio_tap_row_bytes(const uint8_t *__restrict__ img, int W, int Hm1, int Wm1, int row, int c0, int scale)
{
    row = row < 0 ? 0 : (row > Hm1 ? Hm1 : row);
    const uint8_t *q = img + (size_t)row * W;
    unsigned t[12];
#pragma unroll
    for (int b = 0; b < 11; b++) {
        int cc = c0 + b * scale;
        cc = cc < 0 ? 0 : (cc > Wm1 ? Wm1 : cc);
        t[b] = q[cc];
    }
    t[11] = 0u;
    fl_u3 d;
    d.x = t[0] | (t[1] << 8) | (t[2] << 16) | (t[3] << 24);
    d.y = t[4] | (t[5] << 8) | (t[6] << 16) | (t[7] << 24);
    d.z = t[8] | (t[9] << 8) | (t[10] << 16) | (t[11] << 24);
    return d;
}
#if defined(FL_INSTRUMENT) && defined(FL_WIDE_STAMPS)     /* tools/vio_wide_stamps.py: phases of the sweeps of producer wavefront 0 */
#define FL_WSTAMP(k, j) do { if (blockIdx.x == 0 && threadIdx.x == 0 && (k) < 250) g_fl_wall[8 * (k) + (j)] = (long long)wall_clock64(); } while (0)
#else
#define FL_WSTAMP(k, j) do { } while (0)
#endif
#define FL_VIO_WIDE_PRIO_ROWS 0x5B6           /* tap rows (bit a) on which the second workgroup of a CU runs at priority 1: a % 3 != 0 */
#define FL_VIO_WIDE_CUS_SHIFT 16             /* launch flags (internal), bits 16..: the device's CU count = the first block that shares a CU */
#define FL_VIO_WIDE_RS 64                    /* floats per patch in the LDS copy of the reference patches */
#define FL_VIO_WIDE_PPB FL_VIO_NT            /* patches per workgroup and sweep */
template <int NT>
__device__ __forceinline__ void vio_produce_wide(const uint8_t *__restrict__ img, const float *__restrict__ ref, const double *__restrict__ pos,
                                                 const int32_t *__restrict__ slevel, float *__restrict__ errors, int m, int level,
                                                 const FlVioConst *__restrict__ VC, const FlDev18 *__restrict__ D, int nprod,
                                                 double *s_red, unsigned epoch, void *__restrict__ records,
                                                 unsigned long long *__restrict__ err_words /* this pass's half, nullable */,
                                                 float *s_ref /* LDS: (NT / 64) * 64 * FL_VIO_WIDE_RS */,
                                                 int first_younger /* first block that shares its CU with an earlier one; 0: none */)
{
    constexpr int WPB = NT / 64;
    typedef float fl_f4 __attribute__((ext_vector_type(4)));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *sr = s_ref + wave * (64 * FL_VIO_WIDE_RS);
    // The pass's fp64 constants (camera pose, intrinsics, the extrinsic Jacobians: 41 doubles) are wave-uniform, but fp64 vector
    // instructions take them from vector registers: carried through the loop they are 90 registers of every lane (and the scalar file
    // is full: FlVioConst by value is 82 SGPRs). They wait in LDS instead and are fetched where an iteration needs them -- through an
    // offset the compiler cannot see through, or it hoists the reads out of the loop again.
    double *s_const = s_red + WPB * FL_SUMS18;      // [0..8] Rcw [9..11] Pcw [12..15] fx fy cx cy [16..20] d [21..22] fx_abs fy_abs [23..31] Jdphi_dR [32..40] Jdp_dR
    static_assert(FL_VIO_GPW * WPB * FL_SUMS18 >= WPB * FL_SUMS18 + 41, "the constants fit behind the record sums");
    {
        const int t = threadIdx.x;
        double cv = 0.0;
        if (t < 9) cv = D->Rcw[t];
        else if (t < 12) cv = D->Pcw[t - 9];
        else if (t == 12) cv = VC->fx;
        else if (t == 13) cv = VC->fy;
        else if (t == 14) cv = VC->cx;
        else if (t == 15) cv = VC->cy;
        else if (t < 21) cv = VC->d[t - 16];
        else if (t == 21) cv = VC->fx_abs;
        else if (t == 22) cv = VC->fy_abs;
        else if (t < 32) cv = VC->Jdphi_dR[t - 23];
        else if (t < 41) cv = VC->Jdp_dR[t - 32];
        if (t < 41) s_const[t] = cv;
    }
    __syncthreads();
    const int distort = VC->distort;
    const int W = VC->stride, Hm1 = VC->height - 1, Wm1 = VC->width - 1;
    const __amdgpu_buffer_rsrc_t img_rs = __builtin_amdgcn_make_buffer_rsrc((void *)img, 0, W * (Hm1 + 1) + 16, 0x00020000);
    const bool rows_dword_phase = (W & 3) == 0;
    double acc = 0.0;                              // lane L: this wavefront's total of record value L >> 1
    const int wave_stride = nprod * WPB * 64;
    // Two producer workgroups share a CU, and a SIMD issues for its OLDER wavefront whenever both are ready: the second workgroup of a CU
    // (the dispatcher hands out blocks 0 .. CUs-1 first) ran a quarter slower and ended 20 us behind the first one at 1 M patches -- and
    // the launch lasts as long as its slowest producer (end stamps of all producers: tools/vio_wide_stamps.py). The second workgroup
    // therefore raises its priority on 7 of the 11 tap rows: the ends' means move from 78 / 99 us to 91 / 86, the last producer from
    // 107 to 100, the pass from 123-125 to 113-116 us. A scheduling hint only: results do not depend on it.
    const bool younger = first_younger > 0 && (int)blockIdx.x >= first_younger;
    int sweep = 0;
    for (int ib0 = (blockIdx.x * WPB + wave) * 64; ib0 < m; ib0 += wave_stride, sweep++) {
        const int i = ib0 + lane;
        const bool active = i < m;
        const int ii = active ? i : 0;
        FL_WSTAMP(sweep, 0);
        // the 64 reference patches of this sweep straight into LDS, four patches (1 KB) per load instruction j: lane (cc, q) fetches
        // chunk (cc - j) & 15 (16 bytes) of patch 4 j + q, so that patch p = 4 j + q finds its chunk c at slot ((c + j) & 15, q) of
        // group j -- the 16-byte reads of 16 consecutive lanes then fall on 16 different bank groups (the destination of an LDS-direct
        // load is linear in the lane, the permutation has to sit in the source address).
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // (the lanes have read the previous sweep's patches)
        // (addresses from a few per-iteration values, not from 16 + 16 loop-invariant registers: `lq` is opaque to the compiler)
        int lq = lane >> 2;
        asm volatile("" : "+v"(lq));
        {
            const char *rb = reinterpret_cast<const char *>(ref + (size_t)ib0 * 192 + 64 * level);      // (wave-uniform)
            const int last = m - 1 - ib0;                                                            // last patch of the sweep that exists
#pragma unroll
            for (int j = 0; j < 16; j++) {
                int pq = 4 * j + (lane & 3);
                pq = pq < last ? pq : last;
                const unsigned voff = (unsigned)pq * 768u + (unsigned)((lq - j) & 15) * 16u;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(rb + voff),
                                                 (__attribute__((address_space(3))) void *)(sr + j * 256), 16, 0, 0);
            }
        }
        FL_WSTAMP(sweep, 5);
        const int scale = 1 << (level + slevel[ii]);
        const double ps[3] = {pos[ii * 3 + 0], pos[ii * 3 + 1], pos[ii * 3 + 2]};
        FlPatchGeom g;
        {
            int co = 0;
            asm volatile("" : "+v"(co));
            const double *cp = s_const + co;
            double Rcw[9], Pcw[3];
#pragma unroll
            for (int k = 0; k < 9; k++) Rcw[k] = cp[k];
#pragma unroll
            for (int k = 0; k < 3; k++) Pcw[k] = cp[9 + k];
            FlVioConst vcl;
            vcl.fx = cp[12]; vcl.fy = cp[13]; vcl.cx = cp[14]; vcl.cy = cp[15];
#pragma unroll
            for (int k = 0; k < 5; k++) vcl.d[k] = cp[16 + k];
            vcl.fx_abs = cp[21]; vcl.fy_abs = cp[22];
            vcl.distort = distort;
            fl_patch_geom(vcl, Rcw, Pcw, ps, scale, g);
        }
        const bool inside = (g.v_i - 5 * scale >= 0) && (g.v_i + 5 * scale <= Hm1) && (g.u_i - 5 * scale >= 0) && (g.u_i + 5 * scale <= Wm1);
        // (row loads need the whole wavefront inside the image and at ONE pyramid scale, 1, 2 or 4)
        const int scale_u = __builtin_amdgcn_readfirstlane(scale);
        const bool uniform = rows_dword_phase && (__ballot(inside && scale == scale_u) == ~0ull);
        // tap rows a = 0 .. 10 = image rows v_i + (a - 5) scale, tap columns b = 0 .. 10 = image columns u_i + (b - 5) scale: 11 bytes
        // per row, as three words
#if defined(FL_INSTRUMENT) && defined(FL_WIDE_STAMPS)
        asm volatile("" :: "v"(g.u_i), "v"(g.v_i));
#endif
        FL_WSTAMP(sweep, 6);
        fl_u3 d[11];
        if (uniform && scale_u == 1) {
            const int off0 = (g.v_i - 5) * W + (g.u_i - 5);            // >= 0: the patch is inside the image
            const unsigned sh = (unsigned)off0 & 3u;                   // the same for every row: W % 4 == 0
            const int voff = off0 & ~3;
#pragma unroll
            for (int a = 0; a < 11; a++) {
                const fl_u4 rw = __builtin_bit_cast(fl_u4, __builtin_amdgcn_raw_buffer_load_b128(img_rs, voff, a * W, 0));
                d[a].x = __builtin_amdgcn_alignbyte(rw.y, rw.x, sh); d[a].y = __builtin_amdgcn_alignbyte(rw.z, rw.y, sh);
                d[a].z = __builtin_amdgcn_alignbyte(rw.w, rw.z, sh);
            }
        } else if (uniform && scale_u == 2) {
            // tap scale 2 (level 1 of a ComputeJ): the 11 taps of a row are every other byte of a 21-byte span -- six dwords (16 + 8
            // bytes), lined up, then the even bytes of each pair of dwords gathered into one (v_perm_b32): the same three words per row
            const int off0 = (g.v_i - 10) * W + (g.u_i - 10);
            const unsigned sh = (unsigned)off0 & 3u;
            const int voff = off0 & ~3;
#pragma unroll
            for (int a = 0; a < 11; a++) {
                const fl_u4 w4 = __builtin_bit_cast(fl_u4, __builtin_amdgcn_raw_buffer_load_b128(img_rs, voff, a * 2 * W, 0));
                const fl_u2 w2 = __builtin_bit_cast(fl_u2, __builtin_amdgcn_raw_buffer_load_b64(img_rs, voff + 16, a * 2 * W, 0));
                const unsigned e0 = __builtin_amdgcn_alignbyte(w4.y, w4.x, sh), e1 = __builtin_amdgcn_alignbyte(w4.z, w4.y, sh),
                               e2 = __builtin_amdgcn_alignbyte(w4.w, w4.z, sh), e3 = __builtin_amdgcn_alignbyte(w2.x, w4.w, sh),
                               e4 = __builtin_amdgcn_alignbyte(w2.y, w2.x, sh), e5 = __builtin_amdgcn_alignbyte(0u, w2.y, sh);
                d[a].x = __builtin_amdgcn_perm(e1, e0, 0x06040200u); d[a].y = __builtin_amdgcn_perm(e3, e2, 0x06040200u);
                d[a].z = __builtin_amdgcn_perm(e5, e4, 0x06040200u);
            }
        } else if (uniform && scale_u == 4) {
            // tap scale 4 (level 2): tap b is byte `sh` of dword b of an 11-dword span (16 + 16 + 12 bytes) -- gathered straight out of the
            // loaded dwords with per-lane selectors. Two batches of rows (6 + 5): all 121 dwords at once would not fit the registers.
            const int off0 = (g.v_i - 20) * W + (g.u_i - 20);
            const unsigned sh = (unsigned)off0 & 3u;
            const int voff = off0 & ~3;
            const unsigned sel2 = sh | ((4u + sh) << 8) | 0x0c0c0000u;       // [lo.b(sh), hi.b(sh), 0, 0]
            const unsigned sel1 = sh | 0x0c0c0c00u;                          // [lo.b(sh), 0, 0, 0]
#pragma unroll
            for (int half = 0; half < 2; half++) {
                fl_u4 wa[6], wb[6];
                fl_u3 wc[6];
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    const int a = half * 6 + k;
                    if (a < 11) {
                        wa[k] = __builtin_bit_cast(fl_u4, __builtin_amdgcn_raw_buffer_load_b128(img_rs, voff, a * 4 * W, 0));
                        wb[k] = __builtin_bit_cast(fl_u4, __builtin_amdgcn_raw_buffer_load_b128(img_rs, voff + 16, a * 4 * W, 0));
                        wc[k] = __builtin_bit_cast(fl_u3, __builtin_amdgcn_raw_buffer_load_b96(img_rs, voff + 32, a * 4 * W, 0));
                    }
                }
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    const int a = half * 6 + k;
                    if (a < 11) {
                        const unsigned p01 = __builtin_amdgcn_perm(wa[k].y, wa[k].x, sel2), p23 = __builtin_amdgcn_perm(wa[k].w, wa[k].z, sel2),
                                       p45 = __builtin_amdgcn_perm(wb[k].y, wb[k].x, sel2), p67 = __builtin_amdgcn_perm(wb[k].w, wb[k].z, sel2),
                                       p89 = __builtin_amdgcn_perm(wc[k].y, wc[k].x, sel2), pA = __builtin_amdgcn_perm(0u, wc[k].z, sel1);
                        d[a].x = __builtin_amdgcn_perm(p23, p01, 0x05040100u); d[a].y = __builtin_amdgcn_perm(p67, p45, 0x05040100u);
                        d[a].z = __builtin_amdgcn_perm(pA, p89, 0x0c040100u);
                    }
                }
            }
        } else {
#pragma unroll
            for (int a = 0; a < 11; a++) d[a] = vio_tap_row_bytes(img, W, Hm1, Wm1, g.v_i + (a - 5) * scale, g.u_i - 5 * scale, scale);
        }
        FL_WSTAMP(sweep, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the reference patches have landed (the tap rows too)
        FL_WSTAMP(sweep, 2);
        const float wtl = g.wtl, wtr = g.wtr, wbl = g.wbl, wbr = g.wbr;
        const float *srp = sr + lq * 256 + (lane & 3) * 4;
        float Tp[11], Ia[10], Ib[10], Ic[10];
        double S[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        float pe = 0.0f;
#pragma unroll
        for (int b = 0; b < 11; b++) Tp[b] = 0.f;
#pragma unroll
        for (int b = 0; b < 10; b++) { Ia[b] = 0.f; Ib[b] = 0.f; Ic[b] = 0.f; }
#pragma unroll
        for (int a = 0; a < 11; a++) {
            float Tc[11];
            if (younger) { if ((FL_VIO_WIDE_PRIO_ROWS >> a) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
#pragma unroll
            for (int b = 0; b < 11; b++) {
                const unsigned dd = (b < 4) ? d[a].x : ((b < 8) ? d[a].y : d[a].z);
                Tc[b] = (float)((dd >> (8 * (b & 3))) & 0xffu);
            }
            if (a >= 1) {
                // bilinear values of row r = a - 1 (taps of rows r and r + 1); the four corners of the 10 x 10 array are never used
#pragma unroll
                for (int b = 0; b < 10; b++) { Ia[b] = Ib[b]; Ib[b] = Ic[b]; }
#pragma unroll
                for (int b = 0; b < 10; b++) {
                    const bool corner = (a == 1 || a == 10) && (b == 0 || b == 9);
                    Ic[b] = corner ? 0.f : (wtl * Tp[b] + wtr * Tp[b + 1] + wbl * Tc[b] + wbr * Tc[b + 1]);
                }
            }
            if (a >= 3) {
                // pixel row x = a - 3: centre values Ib (row x + 1), the rows above / below Ia, Ic
                const int x = a - 3;
                int lr = lq;
                asm volatile("" : "+v"(lr));                      // (keeps the reads in their row: hoisted, they are 64 registers)
                const fl_f4 r0 = *reinterpret_cast<const fl_f4 *>(srp + ((2 * x + lr) & 15) * 16),
                            r1 = *reinterpret_cast<const fl_f4 *>(srp + ((2 * x + 1 + lr) & 15) * 16);
                const float rf[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
                for (int y = 0; y < 8; y++) {
                    const float du = 0.5f * (Ib[y + 2] - Ib[y]);
                    const float dv = 0.5f * (Ic[y + 1] - Ia[y + 1]);
                    const float res = Ib[y + 1] - rf[y];
                    const double dud = (double)du, dvd = (double)dv, rd = (double)res;
                    S[0] = fma(dud, dud, S[0]); S[1] = fma(dud, dvd, S[1]); S[2] = fma(dvd, dvd, S[2]);
                    S[3] = fma(dud, rd, S[3]); S[4] = fma(dvd, rd, S[4]); S[5] = fma(rd, rd, S[5]);
                    pe = (float)fma(rd, rd, (double)pe);           // lidar_selection.cpp:849 (the product of two floats is exact in double)
                }
            }
#pragma unroll
            for (int b = 0; b < 11; b++) Tp[b] = Tc[b];
            // A row at a time: left alone, the compiler converts all 121 taps and forms all 96 bilinear values up front and sinks the
            // pixels' sums behind them -- 480 registers. Everything a row hands to the next one passes through this statement.
            asm volatile("" : "+v"(Tp[0]), "+v"(Tp[1]), "+v"(Tp[2]), "+v"(Tp[3]), "+v"(Tp[4]), "+v"(Tp[5]), "+v"(Tp[6]), "+v"(Tp[7]), "+v"(Tp[8]),
                              "+v"(Tp[9]), "+v"(Tp[10]), "+v"(Ic[0]), "+v"(Ic[1]), "+v"(Ic[2]), "+v"(Ic[3]), "+v"(Ic[4]), "+v"(Ic[5]), "+v"(Ic[6]),
                              "+v"(Ic[7]), "+v"(Ic[8]), "+v"(Ic[9]), "+v"(S[0]), "+v"(S[1]), "+v"(S[2]), "+v"(S[3]), "+v"(S[4]), "+v"(S[5]), "+v"(pe));
        }
        FL_WSTAMP(sweep, 3);
        // the patch's 29 outputs (M^T G M, M^T g, count, sum res^2), summed over the wavefront's 64 patches right away: 32 accumulators
        // per lane carried through the pixel rows would be 64 registers of the loop's budget
        double w[FL_SUMS18];
#pragma unroll
        for (int k = 0; k < FL_SUMS18; k++) w[k] = 0.0;
        if (active) {
            errors[i] = pe;
            if (err_words)
                __hip_atomic_store(err_words + i, ((unsigned long long)__float_as_uint(pe) << 32) | (unsigned long long)epoch, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            int co = 0;
            asm volatile("" : "+v"(co));
            const double *cp = s_const + co;
            double Rcw[9], J1[9], J2[9];
#pragma unroll
            for (int k = 0; k < 9; k++) { Rcw[k] = cp[k]; J1[k] = cp[23 + k]; J2[k] = cp[32 + k]; }
            double M[2][6];
            fl_patch_M(g, J1, J2, Rcw, M);
            fl_patch_accum(w, M, S);
        }
        wave_transpose_reduce32(w, lane);
        acc += w[0];
        FL_WSTAMP(sweep, 4);
    }
    double *s_fin = s_red;                          // WPB * 32 doubles
    if ((lane & 1) == 0) s_fin[wave * FL_SUMS18 + (lane >> 1)] = acc;
    __syncthreads();
    double mine = 0.0;
    if (threadIdx.x < FL_SUMS18) {
        mine = s_fin[threadIdx.x];
#pragma unroll
        for (int wv = 1; wv < WPB; wv++) mine += s_fin[wv * FL_SUMS18 + threadIdx.x];
    }
    publish_record<FL_SUMS18>(mine, epoch, records);
#if defined(FL_INSTRUMENT) && defined(FL_WIDE_STAMPS)
    if (threadIdx.x == 0 && blockIdx.x < 1000) g_fl_wall[1024 + blockIdx.x] = (long long)wall_clock64();      // every producer's end
#endif
}

// The AUDITOR workgroup (block `nprod`): the reference's float running sum `error += patch_error` over the patches in order
// (lidar_selection.cpp:849-857; solve18.h, eskf18_solve_block) for EVERY pass, computed beside the producers and the solver instead of
// by the solver when it finds the accept test fragile: the m additions start as soon as the
// per-patch words arrive and overlap the gather and the solve, so a fragile pass waits for the tail of the chain only, and the value
// of the last accepted pass is always at hand (no second chain). The result goes into a 16-slot ring behind the per-patch words,
// slot = epoch & 15, tagged with the epoch like every hand-off word. The chain itself is exact_chain.h's lane-parallel form (~3 us
// at 2 k patches): the auditor is done before the solver asks.
// (not inlined: its registers and its staging stay out of the pass kernels' allocation)
// EXT: the staging buffer is the caller's (s_ext, FL_EXACT_LDS floats) -- the one-patch-per-lane pass kernel lends the auditor workgroup
// its reference-patch array, which keeps the kernel's LDS under half a CU's
template <bool EXT = false>
__device__ __attribute__((noinline)) int vio_audit_pass(unsigned long long *__restrict__ err_base, int err_cap, int buf, int m, unsigned epoch,
                                                        float *s_ext = nullptr)
{
    __shared__ __attribute__((aligned(16))) float s_own[EXT ? 4 : FL_EXACT_LDS];
    float *s_aud = EXT ? s_ext : s_own;
    __shared__ int s_to;
    if (threadIdx.x == 0) {
        s_to = 0;
        __hip_atomic_store(err_base + 2 * (size_t)err_cap + FL_AUDIT_RING, (unsigned long long)epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const float f = vio_exact_sum_inl(err_base + (size_t)buf * err_cap, m, epoch, s_aud, &s_to);
    const int to = s_to;
    if (threadIdx.x == 0)         // (no total: the solver workgroup replays the pass itself, solve18.h)
        __hip_atomic_store(err_base + 2 * (size_t)err_cap + (epoch & (FL_AUDIT_RING - 1)),
                           ((unsigned long long)(to ? FL_AUDIT_NONE : __float_as_uint(f)) << 32) | (unsigned long long)epoch, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    return to;
}

// WIDE 1: the producers are vio_produce_wide's (one patch per lane; passes over >= FL_VIO_WIDE_MIN patches, api_vio.inc)
template <int MODE, int WIDE = 0>
__global__ __launch_bounds__(FL_VIO_NT, WIDE ? 2 : 1) void vio_pass_kernel(const uint8_t *__restrict__ img, const float *__restrict__ ref,
                                                            const double *__restrict__ pos, const int32_t *__restrict__ slevel,
                                                            float *__restrict__ errors, int m, int level_arg,
                                                            const FlVioConst *__restrict__ VC, FlDev18 *__restrict__ D,
                                                            void *__restrict__ records, unsigned *__restrict__ epoch_ptr,
                                                            double *__restrict__ sums_out, int flags)
{
    constexpr int NT = FL_VIO_NT;
    constexpr int WPB = NT / 64;
    // chain batch: 8 iterations in the fused kernel (16 KB more LDS than 4; the solver workgroup's gather and replay buffers live in the
    // producers' arrays -- a workgroup has ONE role --, which keeps two workgroups per CU), 4 in the accumulate-only form (its third
    // workgroup per CU is worth more than the longer batch)
    constexpr int CB = (MODE == 0) ? 8 : FL_VIO_CHAIN_BATCH;
    constexpr int RES_FLOATS = WIDE ? WPB * 64 * FL_VIO_WIDE_RS : FL_VIO_GPW * WPB * CB * 64;      // (WIDE: the lanes' reference patches)
    __shared__ double s_red[FL_VIO_GPW * WPB * FL_SUMS18];
    __shared__ __attribute__((aligned(16))) float s_res[RES_FLOATS];
    static_assert(sizeof(double) * FL_VIO_GPW * WPB * FL_SUMS18 >= sizeof(double) * 2 * NT, "the solver's gather buffer fits the producers' reduction buffer");
    static_assert(MODE != 0 || sizeof(float) * RES_FLOATS >= sizeof(float) * FL_EXACT_LDS, "the solver's replay buffer fits the producers' residual buffer");
    const int nprod = gridDim.x - 2;              // then the solver and the auditor (FL_VIO_SOLVER_BLOCK / FL_VIO_AUDITOR_BLOCK)
    const int solver_block = FL_VIO_SOLVER_BLOCK(nprod), auditor_block = FL_VIO_AUDITOR_BLOCK(nprod);
    FL_INSTR(if ((flags & FL_ITER_STAMP) && threadIdx.x == 0 && blockIdx.x < 1024) g_fl_wall[blockIdx.x] = (long long)wall_clock64();)
    FlVioFirst pf;
    if constexpr (!WIDE) pf = vio_prefetch_first(ref, pos, slevel, m, level_arg, nprod);
    double pf_solver = 0.0;
    if (MODE == 0 && blockIdx.x == solver_block) pf_solver = eskf18_prefetch_issue(D);
    if (D->status & FL_NUM_TIMEOUT) {             // an earlier pass of the chain was abandoned (solve18.h): the host resumes
        if (blockIdx.x == solver_block && threadIdx.x == 0) D->resume_count += 1;      // (the solver workgroup is the only writer)
        return;
    }
    if (!(flags & FL_ITER_FORCE) && D->stop) return;
    const unsigned epoch = *epoch_ptr;

    if (blockIdx.x == auditor_block) {
        // ------------------------------------------------------------------ auditor workgroup (single-rank fused passes only: the
        // sharded form chains the sum through the ranks, solve18.h vio_exact_chain)
        if (MODE != 0 || (flags & FL_ITER_FORCE) || !D->err_words || D->xchg_world > 1) return;
        if constexpr (WIDE) vio_audit_pass<true>(D->err_words, D->err_cap, D->iters_run & 1, m, epoch, s_res);
        else vio_audit_pass(D->err_words, D->err_cap, D->iters_run & 1, m, epoch);
        return;
    }
    if (blockIdx.x == solver_block) {
        // ------------------------------------------------------------------ solver workgroup
        double *s_fin = s_red;                               // (2 * NT doubles: see the static_assert above)
        __shared__ double s_sums[FL_SUMS18];
        __shared__ FlSolveLds s_solve;
        FL_INSTR(fl_stamp(flags, 8);)
        FlSolveRegs G;
        if (MODE == 0) { eskf18_prefetch_commit(pf_solver, s_solve); eskf18_load_regs(s_solve, G, VC); }
        FL_INSTR(fl_stamp(flags, 9);)
#if defined(FL_INSTRUMENT) && defined(FL_WIDE_STAMPS)
        if (WIDE && threadIdx.x == 0) g_fl_wall[2040] = (long long)wall_clock64();
#endif
        int gst = gather_records<NT, FL_SUMS18>(records, nprod, epoch, s_fin, s_sums);
#if defined(FL_INSTRUMENT) && defined(FL_WIDE_STAMPS)
        if (WIDE && threadIdx.x == 0) g_fl_wall[2041] = (long long)wall_clock64();
#endif
        FL_INSTR(fl_stamp(flags, 10);)
        if (threadIdx.x == 0) *epoch_ptr = epoch + 1u;
        const int world = (MODE == 0) ? D->xchg_world : 1;
        unsigned xe_pass = 0u;                              // exchange epoch of this pass (tag of the replay mail, solve18.h)
        if (world > 1) {                                    // sharded form: totals over the ranks (handoff.h)
            __shared__ double s_xchg[FL_MAX_PEERS * 32];
            const FlPeerView PV = fl_peer_view(D);
            const unsigned xe = *D->xchg_epoch;
            xe_pass = xe;
            gst |= peer_allreduce32(PV, xe, s_sums, s_xchg);
            if (threadIdx.x == 0) *D->xchg_epoch = xe + 1u;
        }
        if (MODE == 0) {
            float *s_ex = s_res;                             // (FL_EXACT_LDS floats)
            FlVioExact ex;
            ex.words = D->err_words; ex.m = m; ex.cap = D->err_cap; ex.epoch = epoch; ex.scratch = s_ex; ex.enabled = !(flags & FL_ITER_FORCE);
            ex.own = (world > 1) ? D->xchg_peer[D->xchg_rank] : nullptr; ex.peer = D->xchg_peer; ex.rank = D->xchg_rank; ex.world = world;
            ex.xe = xe_pass;
            eskf18_solve_block<FL_EPI_VIO>(D, s_sums, s_solve, G, gst, nullptr, 0u, ex, VC);   // incl. the camera pose for the next pass's producers
        } else {
            if (threadIdx.x < FL_SUMS18) sums_out[threadIdx.x] = s_sums[threadIdx.x];
        }
        FL_INSTR(fl_stamp(flags, 11);)
#if defined(FL_INSTRUMENT) && defined(FL_WIDE_STAMPS)
        if (WIDE && threadIdx.x == 0) g_fl_wall[2042] = (long long)wall_clock64();
#endif
        return;
    }

    // -------------------------------------------------------------------- producer workgroups
    const int level = (level_arg >= 0) ? level_arg : D->level;
    if constexpr (WIDE) {
        unsigned long long *ew = D->err_words ? D->err_words + (size_t)(D->iters_run & 1) * D->err_cap : nullptr;
        vio_produce_wide<NT>(img, ref, pos, slevel, errors, m, level, VC, D, nprod, s_red, epoch, records, ew, s_res,
                             (int)((unsigned)flags >> FL_VIO_WIDE_CUS_SHIFT));
    } else {
        // wave-uniform camera pose, derived from the state by the previous pass's solver (vio_derive_pose)
        const FlVioConst vc = *VC;
        double Rcw[9], Pcw[3];
#pragma unroll
        for (int i = 0; i < 9; i++) Rcw[i] = D->Rcw[i];
#pragma unroll
        for (int i = 0; i < 3; i++) Pcw[i] = D->Pcw[i];
        unsigned long long *ew = D->err_words ? D->err_words + (size_t)(D->iters_run & 1) * D->err_cap : nullptr;
        __shared__ int s_pidx[FL_VIO_GPW * WPB * CB];
        FlVioLaneRole role = fl_vio_lane_role((int)(threadIdx.x & 15), VC);
        fl_vio_lane_role_pose(role, (int)(threadIdx.x & 15), D->Rcw);
        vio_produce<CB>(img, ref, pos, slevel, errors, m, level_arg, level, vc, Rcw, Pcw, pf, nprod, s_red, epoch, records, flags, ew, s_res, s_pidx, role);
    }
    FL_INSTR(if (blockIdx.x == 0) fl_stamp(flags, 2);)
    FL_INSTR(if ((flags & FL_ITER_STAMP) && threadIdx.x == 0 && blockIdx.x < 1024) g_fl_wall[1024 + blockIdx.x] = (long long)wall_clock64();)
}

#ifdef FL_INSTRUMENT
// Debug / test: exact_chain.h's workgroup chain (out4[0]) and wavefront chain (out4[2]) beside the plain one-lane chain (out4[1]) over
// the same floats; out4[3] = number of chunks in which the workgroup form's checks failed (it then ran the wavefront form)
__global__ __launch_bounds__(256) void fl_chain_debug_kernel(const float *__restrict__ e, int n, float init, float *__restrict__ out4)
{
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(16))) float scr[FL_EXACT_LDS];
    float f = init, g = init, w = init;
    int fell = 0;
    for (int base = 0; base < n; base += FL_EXACT_CHUNK) {
        const int cnt = min(FL_EXACT_CHUNK, n - base);
        bool bad = false;
        for (int k = threadIdx.x; k < cnt; k += blockDim.x) { const float v = e[base + k]; scr[k] = v; bad |= !(v >= 0.0f); }
        for (int k = cnt + threadIdx.x; k < cnt + FL_CHAIN_STEP; k += blockDim.x) scr[k] = 0.0f;
        const bool any_bad = __syncthreads_or(bad ? 1 : 0) != 0;
        int fb = 0;
        __shared__ long long s_prof[12];
        f = fl_chain_f32_block(scr, cnt, f, any_bad, &fb, s_prof);
        fell += fb;
        if (threadIdx.x == 0) s_prof[10] = (long long)clock64();
        if (threadIdx.x < 64) w = fl_chain_f32_wave(scr, cnt, w, any_bad);
        if (threadIdx.x == 0) { s_prof[11] = (long long)clock64(); for (int i = 0; i < 12; i++) out4[4 + i] = (float)(s_prof[i] - s_prof[0]); }
        if (threadIdx.x == 64)
            for (int k = 0; k < cnt; k++) g = g + scr[k];
        __syncthreads();
    }
    if (threadIdx.x == 0) { out4[0] = f; out4[2] = w; out4[3] = (float)fell; }
    if (threadIdx.x == 64) out4[1] = g;
}
#endif


// ComputeJ tail: if (now_error < error) state->cov -= G*state->cov  (:978-981); then the frame's result mailbox (fl_publish_state).
// Any workgroup of >= 128 threads, barriers inside.
__device__ __forceinline__ void vio_cov_update_body(FlDev18 *__restrict__ D)
{
    __shared__ double sP[324];
    __shared__ double sG[108];
    const int t = threadIdx.x, nt = blockDim.x;
    const bool apply = D->last_error < 1e10f;
    if (!(D->status & FL_NUM_TIMEOUT)) {          // abandoned frame: enqueued again after the resume (uniform)
        for (int e = t; e < 324; e += nt) sP[e] = D->P[e];
        eskf18_gain_block(D, sG);     // G[:,0:6] of the last ACCEPTED iteration (G is only rewritten on acceptance, :877)
        if (apply) {
            for (int e = t; e < 324; e += nt) {
                const int r = e / 18, c = e % 18;
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < 6; k++) s += sG[r * 6 + k] * sP[k * 18 + c];
                D->P[e] = sP[e] - s;
            }
        }
    }
    fl_publish_state(D);
}
// (out of line: its registers and LDS addressing stay out of the pass loop's allocation, as eskf18_cov_outofline for the LIO kernel)
__device__ __attribute__((noinline)) void vio_cov_outofline(FlDev18 *D) { vio_cov_update_body(D); }
#define FL_VIO_DO_COV 0x200            /* launch flag (internal): the LAST level launch of fl_vio_compute_j ends with the covariance update */
#define FL_VIO_M_DEV 0x400             /* launch flag (internal, fl_vio_detect's fused form): the patch count is FlDev18::m_dev -- the launch's grid
                                          is sized for an upper bound, the kernel uses the workgroups api_vio.inc's vio_grid() would have launched
                                          for the real count (same partition, same record order, same bits), the others leave at once; no
                                          patches: no passes (ComputeJ returns at once then, lidar_selection.cpp:969) */

#define FL_VIO_LEVELS 0x800            /* launch flag (internal, vio_multipass_kernel<1, 1> only): the launch runs pyramid level `level` and then the
                                          finer ones down to 0 -- ComputeJ's whole coarse-to-fine schedule (lidar_selection.cpp:971-977) in one kernel:
                                          every level begins with `begin_residual`, `count` = max_iterations, level_info = the array of the three
                                          results (indexed by level). A level's last broadcast (stop) is the next level's first pose. */
// Up to `count` passes of one pyramid level in ONE launch (see lio18_multipass_kernel): the solver broadcasts the derived camera
// pose (Rcw, Pcw: what the producers consume) and the stop bit; a rejected solve (error went up, lidar_selection.cpp:888-892)
// reverts and stops like the reference. Bit-identical to `count` launches of vio_pass_kernel<0>.
// WAVES = the register budget (launch bound): 2 = two workgroups per CU (256 VGPRs), the form every concurrent or sharded use needs;
// 1 = a CU's registers to one workgroup (256 VGPRs + AGPRs, no scratch): 8.5 instead of 9.2 us per pass, taken when the launch has the
// device to itself (api_vio.inc vio_mp_variant). Same code, same arithmetic, same bits.
// SPEC = 1 (fl_vio_compute_j / fl_vio_update_state launches that have the device to themselves; never forced passes): fragile accepts go
// ahead on the fp64 decision and are confirmed a pass later (solve18.h vio_spec_confirm).
template <int WAVES, int SPEC = 0>
__global__ __launch_bounds__(FL_VIO_NT, WAVES) void vio_multipass_kernel(const uint8_t *__restrict__ img, const float *__restrict__ ref,
                                                                 const double *__restrict__ pos, const int32_t *__restrict__ slevel,
                                                                 float *__restrict__ errors, int m, int level,
                                                                 const FlVioConst *__restrict__ VC, FlDev18 *__restrict__ D,
                                                                 void *__restrict__ records, unsigned *__restrict__ epoch_ptr,
                                                                 unsigned long long *__restrict__ bcast, int count, int flags,
                                                                 float begin_residual, FlVioLevelInfo *__restrict__ level_info,
                                                                 unsigned *__restrict__ done_word, unsigned done_seq)
{
    // begin_residual >= 0: the launch starts a pyramid level, i.e. it first does what vio_level_begin_kernel does (UpdateState
    // prologue, lidar_selection.cpp:747,756); level_info != nullptr: it ends with what vio_level_end_kernel does. ComputeJ then
    // needs one launch per level instead of three.
    constexpr int NT = FL_VIO_NT;
    constexpr int WPB = NT / 64;
    // every word of the launch's prologue in flight at once (behind one another's branches they were four L2 round trips in a row at the
    // head of the launch, in every workgroup)
    const int status0 = D->status, stop0 = D->stop, iters0 = D->iters_run, m_dev0 = D->m_dev;
    const unsigned epoch0 = *epoch_ptr;
    unsigned long long *err_base = D->err_words;
    const int err_cap = D->err_cap;
    int nprod = gridDim.x - 2;                    // then the solver and the auditor (FL_VIO_SOLVER_BLOCK / FL_VIO_AUDITOR_BLOCK)
    if (flags & FL_VIO_M_DEV) {                   // (uniform over the grid)
        m = m_dev0;
        int want = (m + FL_VIO_GPW * WPB - 1) / (FL_VIO_GPW * WPB);
        want = want < 1 ? 1 : want;
        nprod = want < nprod ? want : nprod;
        if ((int)blockIdx.x >= nprod + 2) return;
        if (m <= 0) count = 0;
    }
    const int solver_block = FL_VIO_SOLVER_BLOCK(nprod), auditor_block = FL_VIO_AUDITOR_BLOCK(nprod);
    const bool force = (flags & FL_ITER_FORCE) != 0;
    const bool begin = begin_residual >= 0.f;
    if (status0 & FL_NUM_TIMEOUT) {               // an earlier pass of the chain was abandoned: nothing runs until the host has resumed
        if (blockIdx.x == solver_block && threadIdx.x == 0) D->resume_count += count;  // (the solver workgroup is the only writer)
        if ((flags & FL_VIO_DO_COV) && blockIdx.x == solver_block) { __syncthreads(); vio_cov_outofline(D); }    // (publishes the abandoned block)
        fl_mp_done(done_word, done_seq, blockIdx.x == solver_block);
        return;
    }
    if (!force && !begin && stop0) { fl_mp_done(done_word, done_seq, blockIdx.x == solver_block); return; }
    const int pass0 = begin ? 0 : iters0;                // index of this launch's first pass within its pyramid level
    // SPEC: all levels in this launch (FL_VIO_LEVELS). Everybody follows the broadcasts by the same rule: a level ends with a broadcast
    // that carries "stop" (tag t); it holds the pose the next level starts from (the accepted state's, or the reverted one's) and the
    // next level's first pass has epoch t. If the level's last pass was a fragile accept that went ahead, the float chain's verdict on it
    // is taken behind the next level's first gather (cross-level speculation): confirmed, nothing changes for the producers; rejected
    // (rare), that pass's records are dropped and its broadcast (tag t + 1) carries the reverted pose and bit 4, "the level starts again".
    // The halves of the per-patch word buffer go on alternating across levels (pb), so that the verdict's words outlive the next pass.
    const bool levels = SPEC != 0 && (flags & FL_VIO_LEVELS) != 0;

    if (blockIdx.x == auditor_block) {
        // auditor workgroup (see vio_audit_pass): follows the passes through the broadcast like a producer
        if (force || !err_base || D->xchg_world > 1) return;
#ifndef FL_AB_NO_AUDITOR
        __shared__ double s_apose[12];
        __shared__ int s_actrl;
        unsigned ebase = epoch0;
        int pb = pass0, lv = level;
        for (;;) {
            int ps;
            for (ps = 0; ps < count; ps++) {
                const unsigned epoch = ebase + (unsigned)ps;
                FL_AUDIT_STAMP(16 * (epoch & 15) + 0, wall_clock64());
                if (ps > 0) {
                    bcast_wait(bcast, epoch, s_apose, &s_actrl, FL_GATHER_SPIN_LIMIT);
                    __syncthreads();
                    if (s_actrl & 7) break;
                    if constexpr (SPEC != 0) if (s_actrl & 16) { ebase += (unsigned)ps; pb = (pb + ps) & 1; ps = 0; }      // the level starts again
                }
                FL_AUDIT_STAMP(16 * (epoch & 15) + 1, wall_clock64());
                const int to = vio_audit_pass(err_base, err_cap, (pb + ps) & 1, m, epoch);
                FL_AUDIT_STAMP(16 * (epoch & 15) + 2, wall_clock64());
                FL_AUDIT_STAMP(16 * (epoch & 15) + 3, to);
            }
            if constexpr (SPEC == 0) break;
            if (!levels || lv == 0 || count <= 0) break;
            if (ps == count) {                            // (a level that used all its passes: its last broadcast has not been read)
                __syncthreads();
                bcast_wait(bcast, ebase + (unsigned)ps, s_apose, &s_actrl, FL_GATHER_SPIN_LIMIT);
                __syncthreads();
            }
            if (s_actrl & 4) break;
            ebase += (unsigned)ps;
            lv--; pb = (pb + ps) & 1;
        }
#endif
        return;
    }
    if (blockIdx.x == solver_block) {
        __shared__ double s_fin[2 * NT];
        __shared__ double s_sums[FL_SUMS18];
        __shared__ FlSolveLds s_solve;
        __shared__ __attribute__((aligned(16))) float s_ex[FL_EXACT_LDS];
        eskf18_prefetch(D, s_solve);
        int lv = level;            // SPEC under FL_VIO_LEVELS: the level this workgroup is on
        unsigned ebase = epoch0;   // ... and the epoch of that level's first pass (otherwise: epoch0)
        if (begin) {
            __syncthreads();
            if (threadIdx.x < 24) D->xold[threadIdx.x] = s_solve.x[threadIdx.x];      // old_state = *state
            if (threadIdx.x == 32) {
                s_solve.last_error = begin_residual; s_solve.iters_run = 0; s_solve.accepted = 0; s_solve.fragile = 0; s_solve.sticky = 0;
                s_solve.last_exact = begin_residual; s_solve.last_exact_valid = 1; s_solve.acc_buf = 0; s_solve.acc_epoch = 0u;
                D->last_exact = begin_residual; D->last_exact_valid = 1; D->err_acc_buf = 0; D->err_acc_epoch = 0u;
                D->last_error = begin_residual; D->level = level; D->stop = 0; D->converged = 0; D->iters_run = 0; D->accepted = 0;
                D->status = 0;
                if constexpr (SPEC != 0) { s_solve.levels = levels ? 1 : 0; s_solve.ctrl = 0; s_solve.li_base = level_info; }
            }
            __syncthreads();
        }
        __shared__ double s_xchg[FL_MAX_PEERS * 32];
        __shared__ unsigned long long *s_peers[FL_MAX_PEERS];
        const FlPeerView PV = fl_peer_view_lds(D, s_peers);
        const unsigned xe0 = PV.world > 1 ? *D->xchg_epoch : 0u;
        int done = 0;
        int rollback = 0;          // SPEC: 1 = a verdict taken at the top of pass `done - 1` rejected the pass before it, 2 = taken behind the launch's last pass
      for (;;) {                   // (one round per pyramid level; one round unless FL_VIO_LEVELS)
        for (int p = 0; p < count; p++) {
            const unsigned epoch = ebase + (unsigned)p;
            // The float chain's verdict on the PREVIOUS pass, if that one went ahead without it (a fragile accept; possibly the last pass of
            // the pyramid level before this one): taken HERE, while this pass's records are on their way -- the auditor's total arrives
            // ~6 us after that pass's records, this pass's gather completes ~7.5 us after them. (Until round 6 it was taken behind the
            // gather: 1.4 us of barriers, a ring read and the deferred writes between every speculated pass's successor and its solve.)
            // A rejection ends the level here; this pass's records are never read.
            if constexpr (SPEC != 0) if (s_solve.spec_pending) {      // (uniform: LDS, written before the barriers of the last pass)
                if (vio_spec_confirm(D, &s_solve, err_base, err_cap, m, s_ex, errors, VC, bcast, epoch + 1u)) { done = p + 1; rollback = 1; break; }
            }
            FlSolveRegs G;
            FL_AUDIT_STAMP(16 * (epoch & 15) + 13, wall_clock64());
            eskf18_load_regs(s_solve, G, VC);                    // solve operands into wave 0's registers while the producers work
            FL_AUDIT_STAMP(16 * (epoch & 15) + 14, wall_clock64());
            FL_INSTR(if (p == 5) fl_stamp(flags, 16);)
            int gst = gather_records<NT, FL_SUMS18>(records, nprod, epoch, s_fin, s_sums);
            if (PV.world > 1) gst |= peer_allreduce32(PV, xe0 + (unsigned)p, s_sums, s_xchg);      // sharded form: totals over the ranks
            FL_AUDIT_STAMP(16 * (epoch & 15) + 8, wall_clock64()); FL_AUDIT_STAMP(16 * (epoch & 15) + 11, epoch); FL_AUDIT_STAMP(16 * (epoch & 15) + 12, level);
            FL_INSTR(if (p == 5) fl_stamp(flags, 17);)
            FlVioExact ex;
            ex.words = err_base; ex.m = m; ex.cap = err_cap; ex.epoch = epoch; ex.scratch = s_ex; ex.enabled = !force;
            ex.own = PV.own; ex.peer = PV.peer; ex.rank = PV.rank; ex.world = PV.world; ex.xe = xe0 + (unsigned)p;
            // wave 0 solves, derives the camera pose of the new state and publishes it (+ the control word) for the producers
            eskf18_solve_block<FL_EPI_VIO, SPEC>(D, s_sums, s_solve, G, gst, bcast, epoch + 1u, ex, VC, (p == 5) ? (flags & FL_ITER_STAMP) : 0);
            FL_INSTR(if (p == 5) fl_stamp(flags, 35);)
            __syncthreads();
            // ... and on THIS pass if the launch ends behind it (stop raised, or the last pass asked for): the wait of the old form, but
            // the solve has been done in the meantime
            // (not if another pyramid level follows in this launch: that level's first pass takes the verdict)
            if constexpr (SPEC != 0) if (s_solve.spec_pending && !(s_solve.ctrl & 4) && ((!force && (s_solve.ctrl & 3)) || p + 1 == count) && !(levels && lv > 0))
                if (vio_spec_confirm(D, &s_solve, err_base, err_cap, m, s_ex, errors, VC, nullptr, 0u)) rollback = 2;
            FL_AUDIT_STAMP(16 * (epoch & 15) + 9, wall_clock64());
            FL_AUDIT_STAMP(16 * (epoch & 15) + 10, s_solve.fragile + 2 * s_solve.audited + 4 * s_solve.exact_timeout + 8 * s_solve.accept);
            FL_INSTR(if (p == 5) fl_stamp(flags, 18);)
            done = p + 1;
            const int ctrl = s_solve.ctrl;
            if (ctrl & 4) {                                      // abandoned (hand-off time-out): this pass and the rest are still to do
                if (threadIdx.x == 0) D->resume_count = count - p;
                break;
            }
            if (!force && (ctrl & 3)) break;
            if (p + 1 < count) eskf18_restage(s_solve);
        }
        if constexpr (SPEC != 0) {
            // A launch left on the ABANDON path (gather time-out of pass p, or the solve raised ctrl & 4) with the verdict on pass p - 1 still
            // out: the deferred old_state / sums_acc / solution / error bookkeeping of that pass live in LDS only and would be
            // lost with the kernel -- the device block would then hold x_{p} beside last_error and old_state of pass p - 2, and the host's
            // resume would revert to the wrong state. Pass p - 1's records and per-patch words are complete (it was gathered and solved):
            // take the chain's verdict now (replay fallback inside), commit or roll back, then leave.
            if (s_solve.spec_pending && !rollback && (s_solve.ctrl & 4)) {      // (uniform: LDS, behind the loop's last barrier)
                if (vio_spec_confirm(D, &s_solve, err_base, err_cap, m, s_ex, errors, VC, nullptr, 0u)) rollback = 2;
            }
            // the revert of a fragile accept the float chain did not confirm (rare; out of the loop and out of line). rollback == 1: the
            // producers of pass done - 1 wait for a control word -- they get "stop"; their records are dropped
            if (rollback) vio_spec_rollback(D, &s_solve, err_base, err_cap, m, errors, VC, rollback == 1 ? bcast : (unsigned long long *)nullptr, ebase + (unsigned)done);
        }
        ebase += (unsigned)done;
        bool restart = false, carry = false;
        if constexpr (SPEC != 0) {
            restart = s_solve.xl_restart != 0;                 // (uniform: LDS, behind a barrier) the level that had begun starts again
            carry = s_solve.spec_pending != 0 && !restart;     // the level's last pass went ahead: its verdict travels into the next level
        }
        if (carry) {
            // what the level's result block gets once the float chain has spoken, and the old_state a rejection goes back to
            FL_INSTR(if (threadIdx.x == 0) g_fl_wall[2044]++;)        // (debug build: verdicts carried into the next level)
            if (threadIdx.x < 24) s_solve.xl_xold[threadIdx.x] = D->xold[threadIdx.x];
            if (threadIdx.x == 32) {
                s_solve.xl_pending = 1; s_solve.xl_level = lv; s_solve.xl_iters = s_solve.iters_run; s_solve.xl_accepted = s_solve.accepted;
                s_solve.xl_status = s_solve.sticky | s_solve.fragile; s_solve.xl_converged = D->converged; s_solve.xl_neff = D->neff;
                s_solve.xl_err = s_solve.last_error;
            }
        } else if (level_info && !restart) {
            FlVioLevelInfo *li = level_info + (levels ? lv : 0);
            __syncthreads();
            if (threadIdx.x < 18) li->solution[threadIdx.x] = D->solution[threadIdx.x];
            if (threadIdx.x == 32) {
                li->error = D->last_error; li->iterations = D->iters_run; li->n_meas = D->neff;
                li->accepted = D->accepted; li->status = D->status; li->converged = D->converged;
            }
        }
        if constexpr (SPEC == 0) break;
        if (!levels || (lv == 0 && !restart) || (s_solve.ctrl & 4)) break;
        // ---- the next pyramid level (UpdateState's prologue, lidar_selection.cpp:747,756, as `begin` above): the state the level
        // ended with -- accepted, reverted or rolled back -- is in xn / xadd, the producers have it as the pose of the level's last broadcast
        __syncthreads();
        if (threadIdx.x == 0) { s_solve.pb = (s_solve.pb + done) & 1; s_solve.xl_restart = 0; }
        if (!restart) lv--;
        const int ran = done;
        done = 0; rollback = 0;
        if (count > 0 && ran > 0) eskf18_restage(s_solve);     // (no patches: no passes ran, x is what it was)
        if (threadIdx.x < 24) D->xold[threadIdx.x] = s_solve.x[threadIdx.x];
        if (threadIdx.x == 32) {
            s_solve.last_error = begin_residual; s_solve.iters_run = 0; s_solve.accepted = 0; s_solve.fragile = 0; s_solve.sticky = 0;
            s_solve.last_exact = begin_residual; s_solve.last_exact_valid = 1; s_solve.acc_buf = 0; s_solve.acc_epoch = 0u;
            D->last_exact = begin_residual; D->last_exact_valid = 1; D->err_acc_buf = 0; D->err_acc_epoch = 0u;
            D->last_error = begin_residual; D->level = lv; D->stop = 0; D->converged = 0; D->iters_run = 0; D->accepted = 0;
            D->status = 0;
        }
        __syncthreads();
      }
        if (threadIdx.x == 0) {
            *epoch_ptr = ebase;
            if (PV.world > 1) *D->xchg_epoch = xe0 + (unsigned)done;
        }
        if (flags & FL_VIO_DO_COV) {                              // the frame's last launch: covariance update + result mailbox
            __threadfence();
            __syncthreads();
            vio_cov_outofline(D);
        }
        fl_mp_done(done_word, done_seq, true);
        return;
    }

    __shared__ double s_red[FL_VIO_GPW * WPB * FL_SUMS18];
    __shared__ double s_pose[12];
    __shared__ int s_ctrl;
    __shared__ __attribute__((aligned(16))) float s_res[FL_VIO_GPW * WPB * 64];
    __shared__ int s_pidx[FL_VIO_GPW * WPB];
    const int spin_limit = D->xchg_world > 1 ? FL_XCHG_SPIN_LIMIT : FL_GATHER_SPIN_LIMIT;   // the solver may be waiting for another process
    const FlVioConst vc = *VC;
    double Rcw[9], Pcw[3];
#pragma unroll
    for (int i = 0; i < 9; i++) Rcw[i] = D->Rcw[i];
#pragma unroll
    for (int i = 0; i < 3; i++) Pcw[i] = D->Pcw[i];
    FlVioLaneRole role = fl_vio_lane_role((int)(threadIdx.x & 15), VC);
    fl_vio_lane_role_pose(role, (int)(threadIdx.x & 15), D->Rcw);
    unsigned ebase = epoch0;
    int pb = pass0, lv = level;
  for (;;) {                       // (one round per pyramid level; one round unless FL_VIO_LEVELS)
    int ps;
    for (ps = 0; ps < count; ps++) {
        const unsigned epoch = ebase + (unsigned)ps;
        const FlVioFirst pf = vio_prefetch_first(ref, pos, slevel, m, lv, nprod);
        if (blockIdx.x == 0) FL_AUDIT_STAMP(16 * (epoch & 15) + 4, wall_clock64());
        if (ps > 0) {
            FL_INSTR(if (blockIdx.x == 0 && (ps == 5 || ps == 6)) fl_stamp(flags, 20 + 4 * (ps - 5));)
            bcast_wait(bcast, epoch, s_pose, &s_ctrl, spin_limit);
            __syncthreads();
            FL_INSTR(if (blockIdx.x == 0 && (ps == 5 || ps == 6)) fl_stamp(flags, 21 + 4 * (ps - 5));)
            if (!force && (s_ctrl & 3)) break;
            if (s_ctrl & 4) break;
            if constexpr (SPEC != 0) if (s_ctrl & 16) { ebase += (unsigned)ps; pb = (pb + ps) & 1; ps = 0; }      // the level starts again (with this pose)
#pragma unroll
            for (int i = 0; i < 9; i++) Rcw[i] = s_pose[i];
#pragma unroll
            for (int i = 0; i < 3; i++) Pcw[i] = s_pose[9 + i];
            fl_vio_lane_role_pose(role, (int)(threadIdx.x & 15), s_pose);
        }
        if (blockIdx.x == 0) FL_AUDIT_STAMP(16 * (epoch & 15) + 5, wall_clock64());
        if (blockIdx.x < 127) FL_AUDIT_STAMP(256 + 2 * blockIdx.x, wall_clock64());
        vio_produce<1>(img, ref, pos, slevel, errors, m, lv, lv, vc, Rcw, Pcw, pf, nprod, s_red, epoch, records,
                    (ps == 5) ? flags : (flags & ~FL_ITER_STAMP), err_base ? err_base + (size_t)((pb + ps) & 1) * err_cap : nullptr, s_res, s_pidx, role);
        FL_INSTR(if (blockIdx.x == 0 && (ps == 5 || ps == 6)) fl_stamp(flags, 23 + 4 * (ps - 5));)
        if (blockIdx.x == 0) FL_AUDIT_STAMP(16 * (epoch & 15) + 6, wall_clock64());
        if (blockIdx.x < 127) FL_AUDIT_STAMP(256 + 2 * blockIdx.x + 1, wall_clock64());
        __syncthreads();
    }
    if constexpr (SPEC != 0) {
        // the level ended with the ROLL-BACK of the pass before the one just produced (solve18.h vio_spec_rollback): the per-patch error array
        // holds the dropped pass's values and has to hold the rejecting pass's -- every workgroup puts its own patches back (the entries it
        // wrote itself: same L2, program order) out of that pass's half of the word buffer
        if (ps < count && (s_ctrl & 32) && err_base) {      // (uniform; ps < count: the loop was left at a wait, s_ctrl is that wait's)
            const unsigned long long *w = err_base + (size_t)((s_ctrl >> 6) & 1) * err_cap;
            const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
            for (int ib = ((int)blockIdx.x * WPB + wave) * FL_VIO_GPW; ib < m; ib += nprod * WPB * FL_VIO_GPW) {
                const int i = ib + lane / FL_VIO_LPP;
                if (lane % FL_VIO_LPP == 0 && i < m)
                    errors[i] = __uint_as_float((unsigned)(__hip_atomic_load(w + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32));
            }
        }
    }
    if constexpr (SPEC == 0) break;
    if (!levels || lv == 0 || count <= 0) break;      // (no patches: no passes, nothing to wait for)
    if (ps == count) {                                // (a level that used all its passes: its last broadcast has not been read)
        bcast_wait(bcast, ebase + (unsigned)ps, s_pose, &s_ctrl, spin_limit);
        __syncthreads();
    }
    if (s_ctrl & 4) break;
    ebase += (unsigned)ps;
#pragma unroll
    for (int i = 0; i < 9; i++) Rcw[i] = s_pose[i];
#pragma unroll
    for (int i = 0; i < 3; i++) Pcw[i] = s_pose[9 + i];
    fl_vio_lane_role_pose(role, (int)(threadIdx.x & 15), s_pose);
    __syncthreads();
    lv--; pb = (pb + ps) & 1;
  }
}

// UpdateState prologue: old_state = *state, last_error = total_residual (:747,756); per-level counters.
__global__ void vio_level_begin_kernel(FlDev18 *__restrict__ D, int level, float total_residual)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (D->status & FL_NUM_TIMEOUT) return;       // abandoned chain (solve18.h): the host resumes, nothing is reset behind its back
    for (int i = 0; i < 24; i++) D->xold[i] = D->x[i];
    D->last_error = total_residual;
    D->last_exact = total_residual; D->last_exact_valid = 1; D->err_acc_buf = 0; D->err_acc_epoch = 0u;
    D->level = level;
    D->stop = 0;
    D->converged = 0;
    D->iters_run = 0;
    D->accepted = 0;
    D->status = 0;
}

__global__ void vio_level_end_kernel(const FlDev18 *__restrict__ D, FlVioLevelInfo *__restrict__ out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int i = 0; i < 18; i++) out->solution[i] = D->solution[i];
    out->error = D->last_error;
    out->iterations = D->iters_run;
    out->n_meas = D->neff;
    out->accepted = D->accepted;
    out->status = D->status;
    out->converged = D->converged;
}

__global__ __launch_bounds__(384) void vio_cov_update_kernel(FlDev18 *__restrict__ D) { vio_cov_update_body(D); }
