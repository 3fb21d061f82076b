// select_kernels.h -- SURVEY.md section 8(f) row N2, pixel-level part of LidarSelector::addFromSparseMap
// (src/lidar_selection.cpp:346-587): depth image of the scan (:376-410) and, per grid winner, the depth-continuity test
// (:484-506), getWarpMatrixAffine (:232-256), getBestSearchLevel (:315-329), warpAffine for 3 pyramid levels (:258-296),
// getpatch of the current image (:119-140), the NCC gate (:298-313, :559-563) and the squared-error gate (:565-570).
// The accepted patches are written straight into the VIO patch tensor of the handle (d_ref/d_pos/d_slevel, the layout
// UpdateState reads) in ascending candidate order (:572-579): the 768-byte patches never visit the host.
// The voxel lookups / grid competition (:412-466) and Point::getCloseViewObs (src/point.cpp:141-178) walk the
// pointer-linked visual map and stay with the caller, who hands in one candidate per winning grid cell.
//
//  vio_depth_kernel     1 lane = 1 scan point; the reference overwrites a pixel in scan order (last point wins), here
//                       a 64-bit atomicMax of (point index << 32 | depth bits) gives the same winner.
//  vio_select_kernel    1 wavefront = 1 candidate, 1 lane = 1 patch pixel. Doubles/floats in the oracle's order; the
//                       squared error is summed sequentially in float by one lane (the gate compares it to a threshold).
//  vio_select_compact_kernel / vio_select_scatter_kernel   exclusive scan of the accept flags, copy to the patch tensor.
#pragma once

#include "fl_device.h"
#include "vio_kernels.h"

struct FlPatchCandidate {      // == fl_patch_candidate (C ABI)
    double pos[3], px_ref[2], f_ref[3], R_ref[9], t_ref[3];
    int32_t keyframe_id, level_ref, grid_index, reserved;
};

struct FlSelectParams {
    double Rcw[9], Pcw[3];
    double ncc_thre, outlier_threshold;
    int32_t ncc_en, m;
};

// one scan point of the depth image (:393-409)
template <int STRIDE = 3>      // floats per scan point: 3 (x, y, z) or 4 (the voxel filter's x, y, z, intensity)
__device__ __forceinline__ void fl_depth_point(const float *__restrict__ scan, int i, const FlSelectParams *__restrict__ S,
                                               const FlVioConst *__restrict__ VC, unsigned long long *__restrict__ depth64)
{
    const double pw[3] = {(double)scan[STRIDE * i], (double)scan[STRIDE * i + 1], (double)scan[STRIDE * i + 2]};
    double pc[3];
    pc[0] = (S->Rcw[0] * pw[0] + S->Rcw[1] * pw[1] + S->Rcw[2] * pw[2]) + S->Pcw[0];
    pc[1] = (S->Rcw[3] * pw[0] + S->Rcw[4] * pw[1] + S->Rcw[5] * pw[2]) + S->Pcw[1];
    pc[2] = (S->Rcw[6] * pw[0] + S->Rcw[7] * pw[1] + S->Rcw[8] * pw[2]) + S->Pcw[2];
    if (!(pc[2] > 0)) return;
    const double px0 = VC->fx_abs * pc[0] / pc[2] + VC->cx, px1 = VC->fy_abs * pc[1] / pc[2] + VC->cy;      // :398-399
    const int u = (int)px0, v = (int)px1, W = VC->width, H = VC->height, b = 40;                                // (patch_size_half+1)*8
    if (!(u >= b && u < W - b && v >= b && v < H - b)) return;
    const unsigned long long key = ((unsigned long long)(unsigned)i << 32) | (unsigned long long)__float_as_uint((float)pc[2]);
    atomicMax(&depth64[(size_t)W * v + u], key);
}
__global__ __launch_bounds__(FL_BLOCK) void vio_depth_kernel(const float *__restrict__ scan, int n, const FlSelectParams *__restrict__ S,
                                                            const FlVioConst *__restrict__ VC, unsigned long long *__restrict__ depth64)
{
    const int i = blockIdx.x * FL_BLOCK + threadIdx.x;
    if (i >= n) return;
    fl_depth_point(scan, i, S, VC, depth64);
}

// vk::PinholeCamera::cam2world; with distortion = cv::undistortPoints' five fixed-point sweeps on a float pixel (oracle/orc_vio.c)
__device__ __forceinline__ void fl_cam2world(const FlVioConst &c, double u, double v, double *f)
{
    double x, y;
    if (!c.distort) {
        x = (u - c.cx) / c.fx; y = (v - c.cy) / c.fy;
    } else {
        const double ifx = 1. / c.fx, ify = 1. / c.fy;
        x = (double)(float)u; y = (double)(float)v;
        x = (x - c.cx) * ifx; y = (y - c.cy) * ify;
        const double x0 = x, y0 = y;
        const double k1 = c.d[0], k2 = c.d[1], p1 = c.d[2], p2 = c.d[3], k3 = c.d[4];
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const double r2 = x * x + y * y;
            const double icdist = 1.0 / (1 + ((k3 * r2 + k2) * r2 + k1) * r2);
            const double deltaX = 2 * p1 * x * y + p2 * (r2 + 2 * x * x);
            const double deltaY = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y;
            x = (x0 - deltaX) * icdist; y = (y0 - deltaY) * icdist;
        }
        x = (double)(float)x; y = (double)(float)y;
    }
    const double z = 1.0, n = sqrt(x * x + y * y + z * z);
    f[0] = x / n; f[1] = y / n; f[2] = z / n;
}
__device__ __forceinline__ void fl_se3_apply(const double *R, const double *t, const double *x, double *o)
{
    o[0] = (R[0] * x[0] + R[1] * x[1] + R[2] * x[2]) + t[0];
    o[1] = (R[3] * x[0] + R[4] * x[1] + R[5] * x[2]) + t[1];
    o[2] = (R[6] * x[0] + R[7] * x[1] + R[8] * x[2]) + t[2];
}
// vk::interpolateMat_8u (rpg_vikit vision.h, restated -- see oracle/orc_select.c)
__device__ __forceinline__ float fl_interpolate_8u(const uint8_t *__restrict__ img, int stride, float u, float v)
{
    const int x = (int)floorf(u), y = (int)floorf(v);
    const float sx = u - x, sy = v - y;
    const float w00 = (1.0f - sx) * (1.0f - sy), w01 = (1.0f - sx) * sy, w10 = sx * (1.0f - sy);
    const float w11 = 1.0f - w00 - w01 - w10;
    const uint8_t *p = img + (size_t)y * stride + x;
    return w00 * p[0] + w01 * p[stride] + w10 * p[1] + w11 * p[stride + 1];
}

#define FL_SEL_NT 256
// depth-continuity test of a candidate (:484-506), all 64 lanes of its wavefront; true = discontinuous
__device__ __forceinline__ bool fl_depth_discontinuous(const FlVioConst &VC, const unsigned long long *__restrict__ depth64, const double *pt_cam,
                                                       const double *pc, int lane)
{
    const int W = VC.width, H = VC.height, half = 4;
    const int pu = min(max((int)pc[0], half), W - 1 - half), pv = min(max((int)pc[1], half), H - 1 - half);   // clamp: the reference reads unchecked
    bool bad = false;
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int k = lane + 64 * r;
        if (k < 81 && k != 40) {
            const int u = k / 9 - half, v = k % 9 - half;
            const float d = __uint_as_float((unsigned)depth64[(size_t)W * (v + pv) + u + pu]);
            if (d != 0.f && fabs(pt_cam[2] - (double)d) > 1.5) bad = true;
        }
    }
    return __any(bad) != 0;
}
// Warp_map of addFromSparseMap (:530-546): the affine warp and the search level are computed for the FIRST candidate of a reference
// frame that passes the depth test and reused for every later candidate observed in the same frame (one image per frame: the
// keyframe id is the key). owner_of_kf[kf] = that first candidate, by atomicMin over the candidate index.
__global__ __launch_bounds__(FL_SEL_NT) void vio_select_owner_kernel(const FlPatchCandidate *__restrict__ cand, const FlSelectParams *__restrict__ S,
                                                                    const FlVioConst *__restrict__ VCp,
                                                                    const unsigned long long *__restrict__ depth64, int *__restrict__ owner_of_kf)
{
    const int wv = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63u);
    const int ci = blockIdx.x * (FL_SEL_NT / 64) + wv;
    if (ci >= S->m) return;
    const FlPatchCandidate &c = cand[ci];
    double pt_cam[3], pc[2];
    fl_se3_apply(S->Rcw, S->Pcw, c.pos, pt_cam);
    fl_world2cam(*VCp, pt_cam, pc);
    if (fl_depth_discontinuous(*VCp, depth64, pt_cam, pc, lane)) return;
    if (lane == 0) atomicMin(&owner_of_kf[c.keyframe_id], ci);
}
// reason: 0 accepted, 1 depth discontinuity, 3 NCC gate, 4 outlier gate
__global__ __launch_bounds__(FL_SEL_NT) void vio_select_kernel(const FlPatchCandidate *__restrict__ cand, const FlSelectParams *__restrict__ S,
                                                              const FlVioConst *__restrict__ VCp, const uint8_t *__restrict__ cur_img,
                                                              const uint8_t *const *__restrict__ keyframes,
                                                              const unsigned long long *__restrict__ depth64, const int *__restrict__ owner_of_kf,
                                                              float *__restrict__ patches /* m x 192 */,
                                                              float *__restrict__ errors, int32_t *__restrict__ slevel, int32_t *__restrict__ reason)
{
    __shared__ float s_d2[FL_SEL_NT / 64][64];
    const int wv = (int)(threadIdx.x >> 6), lane = (int)(threadIdx.x & 63u);
    const int ci = blockIdx.x * (FL_SEL_NT / 64) + wv;
    if (ci >= S->m) return;                                    // whole wavefront
    const FlVioConst &VC = *VCp;
    const FlPatchCandidate &c = cand[ci];
    const int W = VC.width, H = VC.height, half = 4, ps = 8, pst = 64;
    double pt_cam[3], pc[2];
    fl_se3_apply(S->Rcw, S->Pcw, c.pos, pt_cam);
    fl_world2cam(VC, pt_cam, pc);                                                                        // :480-481
    if (fl_depth_discontinuous(VC, depth64, pt_cam, pc, lane)) { if (lane == 0) reason[ci] = 1; return; }      // :484-506
    const FlPatchCandidate &w = cand[owner_of_kf[c.keyframe_id]];      // Warp_map: whose warp this reference frame uses (vio_select_owner_kernel)
    // getWarpMatrixAffine :232-256 (every lane, same values)
    double Rt[9], ref_pos[3], T_R[9], T_t[3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) Rt[i * 3 + j] = w.R_ref[j * 3 + i];
#pragma unroll
    for (int i = 0; i < 3; i++) ref_pos[i] = -(Rt[i * 3] * w.t_ref[0] + Rt[i * 3 + 1] * w.t_ref[1] + Rt[i * 3 + 2] * w.t_ref[2]);   // Feature::pos()
    const double dv0 = ref_pos[0] - w.pos[0], dv1 = ref_pos[1] - w.pos[1], dv2 = ref_pos[2] - w.pos[2];
    const double depth_ref = sqrt(dv0 * dv0 + dv1 * dv1 + dv2 * dv2);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) T_R[i * 3 + j] = S->Rcw[i * 3] * Rt[j] + S->Rcw[i * 3 + 1] * Rt[3 + j] + S->Rcw[i * 3 + 2] * Rt[6 + j];
    fl_se3_apply(S->Rcw, S->Pcw, ref_pos, T_t);
    const double xyz_ref[3] = {w.f_ref[0] * depth_ref, w.f_ref[1] * depth_ref, w.f_ref[2] * depth_ref};
    double du[3], dw[3];
    fl_cam2world(VC, w.px_ref[0] + (double)half, w.px_ref[1], du);
    fl_cam2world(VC, w.px_ref[0], w.px_ref[1] + (double)half, dw);
    const double su = xyz_ref[2] / du[2], sw = xyz_ref[2] / dw[2];
#pragma unroll
    for (int k = 0; k < 3; k++) { du[k] *= su; dw[k] *= sw; }
    double q[3], px_cur[2], px_du[2], px_dv[2];
    fl_se3_apply(T_R, T_t, xyz_ref, q); fl_world2cam(VC, q, px_cur);
    fl_se3_apply(T_R, T_t, du, q);      fl_world2cam(VC, q, px_du);
    fl_se3_apply(T_R, T_t, dw, q);      fl_world2cam(VC, q, px_dv);
    const double A00 = (px_du[0] - px_cur[0]) / half, A10 = (px_du[1] - px_cur[1]) / half;
    const double A01 = (px_dv[0] - px_cur[0]) / half, A11 = (px_dv[1] - px_cur[1]) / half;
    int search_level = 0;                                                                                // :315-329
    const double det = A00 * A11 - A01 * A10;
    double D = det;
    while (D > 3.0 && search_level < 2) { search_level += 1; D *= 0.25; }
    // warpAffine :258-296, lane = y*8 + x
    const double invdet = 1.0 / det;
    const float B00 = (float)(A11 * invdet), B01 = (float)(-A01 * invdet), B10 = (float)(-A10 * invdet), B11 = (float)(A00 * invdet);
    const uint8_t *ref = keyframes[c.keyframe_id];
    const int x = lane & 7, y = lane >> 3;
    float P0 = 0.f;
    float *Pout = patches + (size_t)ci * 192;
#pragma unroll
    for (int lvl = 0; lvl <= 2; lvl++) {
        float val = 0.f;
        if (!isnan(B00)) {
            float p0 = (float)(x - half), p1 = (float)(y - half);
            p0 *= (float)(1 << search_level); p1 *= (float)(1 << search_level);
            p0 *= (float)(1 << lvl); p1 *= (float)(1 << lvl);
            const float u = (B00 * p0 + B01 * p1) + (float)c.px_ref[0];
            const float v = (B10 * p0 + B11 * p1) + (float)c.px_ref[1];
            if (!(u < 0 || v < 0 || u >= W - 1 || v >= H - 1)) val = fl_interpolate_8u(ref, W, u, v);
        }
        Pout[pst * lvl + lane] = val;
        if (lvl == 0) P0 = val;
    }
    // getpatch(img, pc, patch_cache, 0) :119-140 ; lane = x_row*8 + y_col
    float cur;
    {
        const float u_ref = (float)pc[0], v_ref = (float)pc[1];
        const int u_i = min(max((int)floorf((float)pc[0]), half), W - 2 - half), v_i = min(max((int)floorf((float)pc[1]), half), H - 2 - half);
        const float su_ = u_ref - (int)floorf((float)pc[0]), sv_ = v_ref - (int)floorf((float)pc[1]);
        const float w_tl = (float)((1.0 - su_) * (1.0 - sv_)), w_tr = (float)(su_ * (1.0 - sv_)), w_bl = (float)((1.0 - su_) * sv_), w_br = su_ * sv_;
        const uint8_t *ip = cur_img + (size_t)(v_i - half + y) * W + (u_i - half) + x;
        cur = w_tl * ip[0] + w_tr * ip[1] + w_bl * ip[W] + w_br * ip[W + 1];
    }
    if (S->ncc_en) {                                                                                     // :298-313 (doubles)
        const double sr = wave_sum((double)P0), sc = wave_sum((double)cur);
        const double mr = sr / pst, mc = sc / pst;
        const double num = wave_sum(((double)P0 - mr) * ((double)cur - mc));
        const double d1 = wave_sum(((double)P0 - mr) * ((double)P0 - mr));
        const double d2 = wave_sum(((double)cur - mc) * ((double)cur - mc));
        if (num / sqrt(d1 * d2 + 1e-10) < S->ncc_thre) { if (lane == 0) reason[ci] = 3; return; }
    }
    s_d2[wv][lane] = (P0 - cur) * (P0 - cur);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0): the LDS writes of this wavefront have landed
    if (lane == 0) {
        float error = 0.0f;
        for (int i = 0; i < pst; i++) error += s_d2[wv][i];                                              // :565-569, sequential float sum
        const bool out = (double)error > S->outlier_threshold * pst;
        reason[ci] = out ? 4 : 0;
        errors[ci] = error;
        slevel[ci] = search_level;
    }
    (void)ps;
}

// one workgroup: exclusive scan of (reason == 0) over the m candidates -> slot, count
__global__ __launch_bounds__(1024) void vio_select_compact_kernel(const int32_t *__restrict__ reason, int m, int32_t *__restrict__ slot,
                                                                 int32_t *__restrict__ count)
{
    __shared__ int s_cnt[1024];
    const int t = (int)threadIdx.x;
    const int chunk = (m + 1023) / 1024;
    const int b = t * chunk, e = min(b + chunk, m);
    int cnt = 0;
    for (int i = b; i < e; i++) cnt += (reason[i] == 0);
    s_cnt[t] = cnt;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = (t >= d) ? s_cnt[t - d] : 0;
        __syncthreads();
        s_cnt[t] += v;
        __syncthreads();
    }
    int run = s_cnt[t] - cnt;
    for (int i = b; i < e; i++) {
        const int acc = (reason[i] == 0);
        slot[i] = acc ? run : -1;
        run += acc;
    }
    if (t == 1023) *count = s_cnt[1023];
}

// 1 wavefront per candidate: copy the accepted patch and its metadata to the VIO patch tensor
__global__ __launch_bounds__(64) void vio_select_scatter_kernel(const FlPatchCandidate *__restrict__ cand, const int32_t *__restrict__ slot,
                                                               const float *__restrict__ patches, const float *__restrict__ errors,
                                                               const int32_t *__restrict__ slevel, float *__restrict__ d_ref,
                                                               double *__restrict__ d_pos, int32_t *__restrict__ d_slevel,
                                                               float *__restrict__ d_errors, int32_t *__restrict__ accepted_idx,
                                                               float *__restrict__ acc_errors, int32_t *__restrict__ acc_slevel)
{
    const int ci = blockIdx.x, lane = (int)threadIdx.x;
    const int s = slot[ci];
    if (s < 0) return;
#pragma unroll
    for (int k = 0; k < 3; k++) d_ref[(size_t)s * 192 + 64 * k + lane] = patches[(size_t)ci * 192 + 64 * k + lane];
    if (lane < 3) d_pos[(size_t)s * 3 + lane] = cand[ci].pos[lane];
    if (lane == 3) { d_slevel[s] = slevel[ci]; acc_slevel[s] = slevel[ci]; }
    if (lane == 4) { d_errors[s] = errors[ci]; acc_errors[s] = errors[ci]; }
    if (lane == 5) accepted_idx[s] = ci;
}

// fl_vio_detect's fused form: vio_select_compact_kernel + vio_select_scatter_kernel (+ vmap_selected_kernel) in one launch sized for an
// UPPER BOUND of the candidate count (one per grid cell): the count is S->m, written by the kernel that formed the candidates. One
// wavefront per candidate; its slot = the number of accepted candidates in front of it (a ballot walk over reason[0 .. ci): at most
// cells / 64 trips), so the patch tensor comes out in ascending candidate order as after the scan (:572-579). The wavefront of the
// last candidate also leaves the number of accepted patches for the host (count_out) and for ComputeJ's launches (FlDev18::m_dev).
__global__ __launch_bounds__(64) void vio_select_finish_kernel(const FlPatchCandidate *__restrict__ cand, const FlSelectParams *__restrict__ S,
                                                              const int32_t *__restrict__ reason, const float *__restrict__ patches,
                                                              const float *__restrict__ errors, const int32_t *__restrict__ slevel,
                                                              float *__restrict__ d_ref, double *__restrict__ d_pos, int32_t *__restrict__ d_slevel,
                                                              float *__restrict__ d_errors, int32_t *__restrict__ accepted_idx,
                                                              float *__restrict__ acc_errors, int32_t *__restrict__ acc_slevel,
                                                              int32_t *__restrict__ sel_point, int32_t *__restrict__ count_out, FlDev18 *__restrict__ D)
{
    const int ci = blockIdx.x, lane = (int)threadIdx.x;
    const int m = S->m;
    if (ci >= m) return;
    int s = 0;
    for (int i0 = 0; i0 < ci; i0 += 64) {
        const int i = i0 + lane;
        s += (int)__popcll(__ballot(i < ci && reason[i] == 0));
    }
    const bool acc = reason[ci] == 0;
    if (ci == m - 1 && lane == 0) { *count_out = s + (acc ? 1 : 0); D->m_dev = s + (acc ? 1 : 0); }
    if (!acc) return;
#pragma unroll
    for (int k = 0; k < 3; k++) d_ref[(size_t)s * 192 + 64 * k + lane] = patches[(size_t)ci * 192 + 64 * k + lane];
    if (lane < 3) d_pos[(size_t)s * 3 + lane] = cand[ci].pos[lane];
    if (lane == 3) { d_slevel[s] = slevel[ci]; acc_slevel[s] = slevel[ci]; }
    if (lane == 4) { d_errors[s] = errors[ci]; acc_errors[s] = errors[ci]; }
    if (lane == 5) accepted_idx[s] = ci;
    if (lane == 6) sel_point[s] = cand[ci].reserved;          // sub_sparse_map->voxel_points as indices into the map (vmap_selected_kernel)
}


// ---- projection + grid competition over the map points of the visible voxels (lidar_selection.cpp:412-466) ----------------
// 1 lane = 1 map point of the caller's flat list. The reference keeps, per grid cell, the LAST point of its loop whose distance
// is <= the running minimum: the point with the smallest float distance, the later one on ties -- a 64-bit atomicMin over
// (distance bits, ~index). map_value is a running maximum of non-negative scores (atomicMax on the float bits).
struct FlGridParams {
    double Rcw[9], Pcw[3], fpos[3];
    int32_t k, grid_size, gh, length;
};
__global__ __launch_bounds__(FL_BLOCK) void vio_grid_init_kernel(unsigned long long *__restrict__ key, int *__restrict__ val, int32_t *__restrict__ gnum,
                                                                int length)
{
    const int i = blockIdx.x * FL_BLOCK + threadIdx.x;
    if (i >= length) return;
    key[i] = ((unsigned long long)__float_as_uint(10000.f) << 32) | 0xFFFFFFFFull;
    val[i] = 0;                 // bits of 0.0f
    gnum[i] = 3;                // TYPE_UNKNOWN
}
__global__ __launch_bounds__(FL_BLOCK) void vio_grid_kernel(const double *__restrict__ pos, const float *__restrict__ value,
                                                           const FlGridParams *__restrict__ G, const FlVioConst *__restrict__ VC,
                                                           unsigned long long *__restrict__ key, int *__restrict__ val, int32_t *__restrict__ gnum)
{
    const int j = blockIdx.x * FL_BLOCK + threadIdx.x;
    if (j >= G->k) return;
    const double p[3] = {pos[3 * (size_t)j], pos[3 * (size_t)j + 1], pos[3 * (size_t)j + 2]};
    double pc3[3], px[2];
    fl_se3_apply(G->Rcw, G->Pcw, p, pc3);
    if (pc3[2] < 0) return;
    fl_world2cam(*VC, pc3, px);
    const int u = (int)px[0], v = (int)px[1], W = VC->width, H = VC->height, b = 40;
    if (!(u >= b && u < W - b && v >= b && v < H - b)) return;
    const int index = (int)(px[0] / G->grid_size) * G->gh + (int)(px[1] / G->grid_size);
    if (index < 0 || index >= G->length) return;
    gnum[index] = 1;            // TYPE_MAP
    const double o0 = G->fpos[0] - p[0], o1 = G->fpos[1] - p[1], o2 = G->fpos[2] - p[2];
    const float cur_dist = (float)sqrt(o0 * o0 + o1 * o1 + o2 * o2);
    if (cur_dist <= 10000.f)
        atomicMin(&key[index], ((unsigned long long)__float_as_uint(cur_dist) << 32) | (unsigned long long)(0xFFFFFFFEu - (unsigned)j));   // 0xFFFFFFFF = no point yet
    const float cv = value[j];
    if (cv >= 0.f) atomicMax(&val[index], __float_as_int(cv));
}
