// vmap_kernels.h -- the visual map of LidarSelector kept on the device, and the three per-frame steps that touch it
// (src/lidar_selection.cpp, src/point.cpp), completing SURVEY 8f N2's "addFromSparseMap" on the side of its caller:
//   fl_vmap_select           addFromSparseMap :346-587 in one call: voxel keys of the down-sampled scan (sub_feat_map :383-391), depth
//                            image (:393-409, vio_depth_kernel), every map point of those voxels projected + grid competition
//                            (:412-466), per winning cell Point::getCloseViewObs (point.cpp:141-178) -> candidate -> the
//                            pixel-level part (select_kernels.h); the accepted patches are the staged VIO patch set
//   fl_vmap_add_sparse       addSparseMap :142-197 + AddPoint :199-230: Shi-Tomasi score of every scan point in its grid cell,
//                            the best one founds a map point where it beats what the cell already holds
//   fl_vmap_add_observation  addObservation :913-965: pose / pixel distance tests against the OLDEST observation (obs_.back():
//                            addFrameRef pushes to the front), getFurthestViewObs + deleteFeatureRef at 20 observations, push_front
// A map point = position, score, voxel key (AddPoint's rule: truncation after "-1 if negative", :203-212 -- not floor() for
// negative multiples of the voxel size, replicated) and up to 20 observations in list order (obs[0] = front).
// Order-dependent spots of the reference and what replaces them: the unordered_map walk of :412 decides only between map points
// at exactly the same float distance in one cell -- here the later point in map order wins, as in the reference's `<=`;
// "first scan point with the highest score" (:160, strict >) and "last scan point on a depth pixel" are kept with 64-bit atomics
// over (value bits, index).
#pragma once

#include "select_kernels.h"
#include "knn_kernels.h"
#include "voxel_kernels.h"
#include "lio_kernels.h"

#define FL_VMAP_MAX_OBS 20
struct FlVObs {                 // == fl_vmap_obs (C ABI)
    double px[2], f[3], R[9], t[3];
    float score;
    int32_t level, kf_id, frame_id;
};
struct FlVPoint {
    double pos[3];
    float value;
    int32_t n_obs;
    int32_t key[3];
    int32_t pad;
    FlVObs obs[FL_VMAP_MAX_OBS];
};
struct FlVmapParams {
    double Rcw[9], Pcw[3], fpos[3];
    int32_t grid_size, gh, length, n_pts, kf_id, frame_id, pad0, pad1;
};
struct FlVmapCount { int32_t cand, added, obs_added, pad; };

// vk::shiTomasiScore (rpg_vikit vision.cpp; restated, see oracle/orc_vmap.c): sums of integers, exact in float
__device__ __forceinline__ float fl_shi_tomasi(const uint8_t *__restrict__ img, int width, int height, int u, int v)
{
    const int x_min = u - 4, x_max = u + 4, y_min = v - 4, y_max = v + 4;
    if (x_min < 1 || x_max >= width - 1 || y_min < 1 || y_max >= height - 1) return 0.0f;
    float dXX = 0.0f, dYY = 0.0f, dXY = 0.0f;
    for (int y = y_min; y < y_max; y++) {
        const uint8_t *r = img + (size_t)width * y + x_min;
#pragma unroll
        for (int x = 0; x < 8; x++) {
            const float dx = (float)((int)r[x + 1] - (int)r[x - 1]);
            const float dy = (float)((int)r[x + width] - (int)r[x - width]);
            dXX += dx * dx; dYY += dy * dy; dXY += dx * dy;
        }
    }
    dXX = (float)((double)dXX / 128.0);
    dYY = (float)((double)dYY / 128.0);
    dXY = (float)((double)dXY / 128.0);
    const float s = dXX + dYY;
    return (float)(0.5 * (double)(s - sqrtf(s * s - 4 * (dXX * dYY - dXY * dXY))));
}
__device__ __forceinline__ void fl_frame_pos(const double *R, const double *t, double *o)      // T_f_w.inverse().translation()
{
#pragma unroll
    for (int i = 0; i < 3; i++) o[i] = -(R[i] * t[0] + R[3 + i] * t[1] + R[6 + i] * t[2]);
}

// ---- fl_vmap_select ----------------------------------------------------------------------------------------------------------
// one scan point's voxel key into the set of voxels the scan touches (sub_feat_map, :383-391)
template <int STRIDE = 3>
__device__ __forceinline__ void fl_subkey_point(const float *__restrict__ scan, int i, unsigned long long *__restrict__ set, unsigned mask)
{
    const int kx = (int)floor((double)scan[STRIDE * i] / (double)0.5f), ky = (int)floor((double)scan[STRIDE * i + 1] / (double)0.5f),
              kz = (int)floor((double)scan[STRIDE * i + 2] / (double)0.5f);                       // :387-389
    const unsigned long long key = fl_cell_key(kx, ky, kz);
    unsigned h = fl_hash64(key) & mask;
    while (true) {
        const unsigned long long prev = atomicCAS(&set[h], FL_KNN_EMPTY, key);
        if (prev == FL_KNN_EMPTY || prev == key) break;
        h = (h + 1) & mask;
    }
}
__global__ __launch_bounds__(FL_BLOCK) void vmap_subkeys_kernel(const float *__restrict__ scan, int n, unsigned long long *__restrict__ set, unsigned mask)
{
    const int i = blockIdx.x * FL_BLOCK + threadIdx.x;
    if (i >= n) return;
    fl_subkey_point(scan, i, set, mask);
}

__global__ __launch_bounds__(FL_BLOCK) void vmap_project_kernel(const FlVPoint *__restrict__ pts, const FlVmapParams *__restrict__ G,
                                                               const FlVioConst *__restrict__ VC, const unsigned long long *__restrict__ set,
                                                               unsigned mask, unsigned long long *__restrict__ key, int *__restrict__ val,
                                                               int32_t *__restrict__ gnum)
{
    const int j = blockIdx.x * FL_BLOCK + threadIdx.x;
    if (j >= G->n_pts) return;
    const FlVPoint *P = pts + j;
    const unsigned long long vk = fl_cell_key(P->key[0], P->key[1], P->key[2]);
    unsigned h = fl_hash64(vk) & mask;
    while (true) {                                                     // is the point's voxel one the scan touches? (:415-418)
        const unsigned long long k = set[h];
        if (k == vk) break;
        if (k == FL_KNN_EMPTY) return;
        h = (h + 1) & mask;
    }
    const double p[3] = {P->pos[0], P->pos[1], P->pos[2]};
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
    const float cv = P->value;
    if (cv >= 0.f) atomicMax(&val[index], __float_as_int(cv));
}

// Point::getCloseViewObs: observation index or -1
__device__ __forceinline__ int fl_close_view_obs(const FlVPoint *P, const double *fpos)
{
    if (P->n_obs <= 0) return -1;
    double od[3] = {fpos[0] - P->pos[0], fpos[1] - P->pos[1], fpos[2] - P->pos[2]};
    const double on = sqrt(od[0] * od[0] + od[1] * od[1] + od[2] * od[2]);
    od[0] /= on; od[1] /= on; od[2] /= on;
    int best = 0;
    double min_cos = 0;
    for (int k = 0; k < P->n_obs; k++) {
        double c[3];
        fl_frame_pos(P->obs[k].R, P->obs[k].t, c);
        double d[3] = {c[0] - P->pos[0], c[1] - P->pos[1], c[2] - P->pos[2]};
        const double dn = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        d[0] /= dn; d[1] /= dn; d[2] /= dn;
        const double cosa = od[0] * d[0] + od[1] * d[1] + od[2] * d[2];
        if (cosa > min_cos) { min_cos = cosa; best = k; }
    }
    return (min_cos < 0.5) ? -1 : best;
}

// one workgroup: winners in ascending cell order -> candidates (stable compaction by a running offset)
#define FL_VMAP_WG 256
__global__ __launch_bounds__(FL_VMAP_WG) void vmap_candidates_kernel(const FlVPoint *__restrict__ pts, const FlVmapParams *__restrict__ G,
                                                                    const unsigned long long *__restrict__ key, const int32_t *__restrict__ gnum,
                                                                    FlPatchCandidate *__restrict__ cand, FlVmapCount *__restrict__ cnt,
                                                                    int32_t *__restrict__ sel_m /* nullable; fused detect: FlSelectParams::m */)
{
    __shared__ int s_scan[FL_VMAP_WG];
    __shared__ int s_base;
    const int t = (int)threadIdx.x;
    if (t == 0) s_base = 0;
    __syncthreads();
    for (int g0 = 0; g0 < G->length; g0 += FL_VMAP_WG) {
        const int g = g0 + t;
        int obs = -1, v = -1;
        if (g < G->length && gnum[g] == 1) {
            const unsigned lo = (unsigned)key[g];
            if (lo != 0xFFFFFFFFu) {
                v = (int)(0xFFFFFFFEu - lo);
                obs = fl_close_view_obs(pts + v, G->fpos);
            }
        }
        const int flag = obs >= 0;
        s_scan[t] = flag;
        __syncthreads();
        for (int off = 1; off < FL_VMAP_WG; off <<= 1) {           // inclusive Hillis-Steele
            const int x = (t >= off) ? s_scan[t - off] : 0;
            __syncthreads();
            s_scan[t] += x;
            __syncthreads();
        }
        const int base = s_base;
        if (flag) {
            const FlVPoint *P = pts + v;
            FlPatchCandidate c;
            for (int k = 0; k < 3; k++) { c.pos[k] = P->pos[k]; c.f_ref[k] = P->obs[obs].f[k]; c.t_ref[k] = P->obs[obs].t[k]; }
            for (int k = 0; k < 2; k++) c.px_ref[k] = P->obs[obs].px[k];
            for (int k = 0; k < 9; k++) c.R_ref[k] = P->obs[obs].R[k];
            c.keyframe_id = P->obs[obs].kf_id; c.level_ref = P->obs[obs].level; c.grid_index = g; c.reserved = v;
            cand[base + s_scan[t] - 1] = c;
        }
        __syncthreads();
        if (t == FL_VMAP_WG - 1) s_base = base + s_scan[t];
        __syncthreads();
    }
    if (t == 0) { cnt->cand = s_base; if (sel_m) *sel_m = s_base; }
}

// sub_sparse_map->voxel_points of the accepted candidates, as indices into the map
__global__ __launch_bounds__(FL_BLOCK) void vmap_selected_kernel(const FlPatchCandidate *__restrict__ cand, const int32_t *__restrict__ acc_idx,
                                                                const int32_t *__restrict__ count, int32_t *__restrict__ sel_point)
{
    const int i = blockIdx.x * FL_BLOCK + threadIdx.x;
    if (i < *count) sel_point[i] = cand[acc_idx[i]].reserved;
}

// ---- fl_vmap_add_sparse ------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(FL_BLOCK) void vmap_score_kernel(const float *__restrict__ scan, int n, const FlVmapParams *__restrict__ G,
                                                             const FlVioConst *__restrict__ VC, const uint8_t *__restrict__ img,
                                                             unsigned long long *__restrict__ best)
{
    const int i = blockIdx.x * FL_BLOCK + threadIdx.x;
    if (i >= n) return;
    const double pt[3] = {(double)scan[3 * i], (double)scan[3 * i + 1], (double)scan[3 * i + 2]};
    double pc3[3], pc[2];
    fl_se3_apply(G->Rcw, G->Pcw, pt, pc3);
    fl_world2cam(*VC, pc3, pc);                                        // no test of the depth sign in the reference (:153)
    const int u = (int)pc[0], v = (int)pc[1], W = VC->width, H = VC->height, b = 40;
    if (!(u >= b && u < W - b && v >= b && v < H - b)) return;
    const int index = (int)(pc[0] / G->grid_size) * G->gh + (int)(pc[1] / G->grid_size);
    if (index < 0 || index >= G->length) return;
    const float s = fl_shi_tomasi(img, W, H, u, v);
    if (!(s > 0.f)) return;                                            // must beat map_value >= 0 (:160)
    atomicMax(&best[index], ((unsigned long long)__float_as_uint(s) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)i));   // first among equals
}

__global__ __launch_bounds__(FL_VMAP_WG) void vmap_commit_kernel(const float *__restrict__ scan, const FlVmapParams *__restrict__ G,
                                                                const FlVioConst *__restrict__ VC, const unsigned long long *__restrict__ best,
                                                                int *__restrict__ val /* map_value bits, carried from the select step */,
                                                                int32_t *__restrict__ gnum, FlVPoint *__restrict__ pts, FlVmapCount *__restrict__ cnt,
                                                                const struct FlVxCtl *__restrict__ vx /* nullable; fused detect with the scan on the device */)
{
    __shared__ int s_scan[FL_VMAP_WG];
    __shared__ int s_base;
    const int t = (int)threadIdx.x;
    // (the down-sampling did not fit its occupancy bitmap: the host grows it and runs the frame again -- nothing may have changed the map)
    if (vx && vx->cells_short) { if (t == 0) cnt->added = 0; return; }
    if (t == 0) s_base = 0;
    __syncthreads();
    for (int g0 = 0; g0 < G->length; g0 += FL_VMAP_WG) {
        const int g = g0 + t;
        int flag = 0;
        float score = 0.f;
        int idx = -1;
        if (g < G->length) {
            gnum[g] = 3;                                               // reset_grid (:83)
            const unsigned long long b = best[g];
            if (b != 0ull) {
                score = __uint_as_float((unsigned)(b >> 32));
                idx = (int)(0xFFFFFFFFu - (unsigned)b);
                flag = score > __int_as_float(val[g]);
            }
        }
        s_scan[t] = flag;
        __syncthreads();
        for (int off = 1; off < FL_VMAP_WG; off <<= 1) {
            const int x = (t >= off) ? s_scan[t - off] : 0;
            __syncthreads();
            s_scan[t] += x;
            __syncthreads();
        }
        const int base = s_base;
        if (flag) {
            val[g] = __float_as_int(score);
            gnum[g] = 2;                                               // TYPE_POINTCLOUD
            FlVPoint *P = pts + G->n_pts + base + s_scan[t] - 1;
            const double pt[3] = {(double)scan[3 * idx], (double)scan[3 * idx + 1], (double)scan[3 * idx + 2]};
            double pc3[3], pc[2];
            fl_se3_apply(G->Rcw, G->Pcw, pt, pc3);
            fl_world2cam(*VC, pc3, pc);
            for (int k = 0; k < 3; k++) {
                P->pos[k] = pt[k];
                float loc = (float)(pt[k] / 0.5);                      // AddPoint :203-212
                if (loc < 0) loc -= 1.0f;
                P->key[k] = (int32_t)(long long)loc;
            }
            P->value = score; P->n_obs = 1; P->pad = 0;
            FlVObs *o = &P->obs[0];
            o->px[0] = pc[0]; o->px[1] = pc[1];
            fl_cam2world(*VC, pc[0], pc[1], o->f);
            for (int k = 0; k < 9; k++) o->R[k] = G->Rcw[k];
            for (int k = 0; k < 3; k++) o->t[k] = G->Pcw[k];
            o->score = score; o->level = 0; o->kf_id = G->kf_id; o->frame_id = G->frame_id;
        }
        __syncthreads();
        if (t == FL_VMAP_WG - 1) s_base = base + s_scan[t];
        __syncthreads();
    }
    if (t == 0) cnt->added = s_base;
}

// ---- fl_vmap_add_observation -------------------------------------------------------------------------------------------------
// one selected map point of addObservation; pose (Rcw, Pcw, fpos) = the frame's after ComputeJ. Returns 1 if an observation was added.
__device__ __forceinline__ int fl_addobs_point(FlVPoint *__restrict__ P, const double *__restrict__ Rcw, const double *__restrict__ Pcw,
                                               const double *__restrict__ fpos, int kf_id, int frame_id, const FlVioConst *__restrict__ VC,
                                               const uint8_t *__restrict__ img, int level)
{
    int added = 0;
    const double p[3] = {P->pos[0], P->pos[1], P->pos[2]};
    double pc3[3], pc[2];
    fl_se3_apply(Rcw, Pcw, p, pc3);
    fl_world2cam(*VC, pc3, pc);
    const FlVObs *last = &P->obs[P->n_obs - 1];                     // obs_.back(): the oldest
    double Rd[9], td[3];
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++)
            Rd[a * 3 + b] = last->R[a * 3] * Rcw[b * 3] + last->R[a * 3 + 1] * Rcw[b * 3 + 1] + last->R[a * 3 + 2] * Rcw[b * 3 + 2];
#pragma unroll
    for (int a = 0; a < 3; a++) td[a] = last->t[a] - (Rd[a * 3] * Pcw[0] + Rd[a * 3 + 1] * Pcw[1] + Rd[a * 3 + 2] * Pcw[2]);
    const double delta_p = sqrt(td[0] * td[0] + td[1] * td[1] + td[2] * td[2]);
    const double tr = Rd[0] + Rd[4] + Rd[8];
    const double delta_theta = (tr > 3.0 - 1e-6) ? 0.0 : acos(0.5 * (tr - 1));
    bool add_flag = (delta_p > 0.5 || delta_theta > 10);           // :939
    const double e0 = pc[0] - last->px[0], e1 = pc[1] - last->px[1];
    if (sqrt(e0 * e0 + e1 * e1) > 40) add_flag = true;              // :942-944
    int n = P->n_obs;
    if (n >= 20) {                                                  // getFurthestViewObs + deleteFeatureRef
        int far = 0;
        double maxdist = 0.0;
        for (int k = 0; k < n; k++) {
            double c[3];
            fl_frame_pos(P->obs[k].R, P->obs[k].t, c);
            const double d0 = c[0] - fpos[0], d1 = c[1] - fpos[1], d2 = c[2] - fpos[2];
            const double dist = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
            if (dist > maxdist) { maxdist = dist; far = k; }
        }
        for (int k = far; k + 1 < n; k++) P->obs[k] = P->obs[k + 1];
        n--;
    }
    if (add_flag) {
        const float score = fl_shi_tomasi(img, VC->width, VC->height, (int)pc[0], (int)pc[1]);
        P->value = score;
        for (int k = n; k > 0; k--) P->obs[k] = P->obs[k - 1];      // push_front
        FlVObs *o = &P->obs[0];
        o->px[0] = pc[0]; o->px[1] = pc[1];
        fl_cam2world(*VC, pc[0], pc[1], o->f);
        for (int k = 0; k < 9; k++) o->R[k] = Rcw[k];
        for (int k = 0; k < 3; k++) o->t[k] = Pcw[k];
        o->score = score; o->level = level; o->kf_id = kf_id; o->frame_id = frame_id;
        n++;
        added = 1;
    }
    P->n_obs = n;
    return added;
}
__global__ __launch_bounds__(FL_BLOCK) void vmap_addobs_kernel(FlVPoint *__restrict__ pts, const FlVmapParams *__restrict__ G,
                                                              const FlVioConst *__restrict__ VC, const uint8_t *__restrict__ img,
                                                              const int32_t *__restrict__ sel_point, const int32_t *__restrict__ levels, int n_sel,
                                                              FlVmapCount *__restrict__ cnt)
{
    const int i = blockIdx.x * FL_BLOCK + threadIdx.x;
    int added = 0;
    if (i < n_sel) added = fl_addobs_point(pts + sel_point[i], G->Rcw, G->Pcw, G->fpos, G->kf_id, G->frame_id, VC, img, levels[i]);
    const unsigned long long b = __ballot(added != 0);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(&cnt->obs_added, (int)__popcll(b));
}

// ---- fl_vio_detect, fused form (api_vmap.inc): LidarSelector::detect (:1027-1076) as ONE enqueue ---------------------------------
// Nothing returns to the host between the frame's first copy command and its result mailbox: the candidate count, the number of
// accepted patches (ComputeJ's size) and the counts of founded / observed points stay on the device; launches that depend on them are
// sized for their upper bound (one candidate per grid cell) and read the real count from memory.
struct FlDetectParams {             // one small upload per frame
    FlVmapParams vm;                // pose at the call: addFromSparseMap + addSparseMap (updateFrameState at :1039)
    FlSelectParams sel;             // the same pose + the gates; m is written by vmap_candidates_kernel
    double Rci[9], Pci[3];          // camera <- IMU: the pose after ComputeJ is derived on the device (updateFrameState, :904-911)
};

// every clear of the frame in one launch (they were 4 fills + vio_grid_init_kernel) + the keyframe copy of the staged image + the keyframe
// table entry + the record slots of ComputeJ's launches (sized for the upper bound; what a launch does not rewrite must carry no tag)
// + fl_vio_begin's launch (vio_prepare_kernel) with the state block and the frame's parameter block FETCHED from page-locked memory by
// the kernel itself (as imu_forward_kernel does for fl_lidar_front): two copy commands (~4-5 us of stream time each) less in front of the frame
__global__ __launch_bounds__(FL_BLOCK) void vmap_frame_init_kernel(unsigned long long *__restrict__ depth64, int n_depth, unsigned long long *__restrict__ set,
                                                                  int n_set, int *__restrict__ owner, int n_owner, unsigned long long *__restrict__ best,
                                                                  unsigned long long *__restrict__ key, int *__restrict__ val, int32_t *__restrict__ gnum,
                                                                  int length, unsigned long long *__restrict__ records, int n_rec_words,
                                                                  const uint4 *__restrict__ img /* the staged image, or the caller's page-locked one */,
                                                                  uint4 *__restrict__ img_cur /* nullable: then `img` IS the staged copy */, uint4 *__restrict__ kf_img, int n_img16,
                                                                  const uint8_t **__restrict__ kf_table, int kf_id, FlVmapCount *__restrict__ cnt,
                                                                  int32_t *__restrict__ sel_count, unsigned *__restrict__ ticket,
                                                                  FlDev18 *__restrict__ D, const FlVioConst *__restrict__ VC, const FlDev18 *__restrict__ x18_host,
                                                                  FlDetectParams *__restrict__ prm, const FlDetectParams *__restrict__ prm_host)
{
    // x18_host != nullptr: the launch's LAST workgroup does what fl_vio_begin's launch does (state block + parameter block fetched, gain-solve constants,
    // camera pose) -- it used to be a launch of its own in front of this one (8 us), now it runs beside the image's trip over the host link
    const int nblk = (int)gridDim.x - (x18_host ? 1 : 0);
    if ((int)blockIdx.x == nblk) {
        FlPull<FL_BLOCK, (int)sizeof(FlDev18)> blk;
        FlPull<FL_BLOCK, (int)sizeof(FlDetectParams)> pblk;
        blk.load(x18_host);
        pblk.load(prm_host);
        blk.store(D);
        pblk.store(prm);
        __threadfence_block();
        __syncthreads();
        if (threadIdx.x >= 116 && threadIdx.x < 128) vio_derive_pose(D->x, VC, D, (int)threadIdx.x - 116);
        eskf18_prepare_body(D);
        return;
    }
    const int t = blockIdx.x * FL_BLOCK + threadIdx.x, nt = nblk * FL_BLOCK;
    for (int i = t; i < n_depth; i += nt) depth64[i] = 0ull;
    for (int i = t; i < n_set; i += nt) set[i] = FL_KNN_EMPTY;
    for (int i = t; i < n_owner; i += nt) owner[i] = 0x7F7F7F7F;                                   // "no owner yet" = a huge index
    for (int i = t; i < length; i += nt) {
        best[i] = 0ull;
        key[i] = ((unsigned long long)__float_as_uint(10000.f) << 32) | 0xFFFFFFFFull;             // reset_grid + map_value = 0 (:355-356)
        val[i] = 0;
        gnum[i] = 3;
    }
    for (int i = t; i < n_rec_words; i += nt) records[i] = 0ull;
    if (img_cur) {                                                                              // (one trip over the host link)
        const fl_u4 *src = reinterpret_cast<const fl_u4 *>(img);
        fl_u4 *d0 = reinterpret_cast<fl_u4 *>(kf_img), *d1 = reinterpret_cast<fl_u4 *>(img_cur);
        for (int i = t; i < n_img16; i += nt) { const fl_u4 v = __builtin_nontemporal_load(src + i); d0[i] = v; d1[i] = v; }
    }
    else for (int i = t; i < n_img16; i += nt) kf_img[i] = img[i];
    if (t == 0) {
        kf_table[kf_id] = reinterpret_cast<const uint8_t *>(kf_img);
        cnt->cand = 0; cnt->added = 0; cnt->obs_added = 0;
        *sel_count = 0;
        *ticket = 0u;
    }
}

// vio_depth_kernel + vmap_subkeys_kernel: one walk over the down-sampled scan
__global__ __launch_bounds__(FL_BLOCK) void vmap_scan_kernel(const float *__restrict__ scan, int n, const FlSelectParams *__restrict__ S,
                                                            const FlVioConst *__restrict__ VC, unsigned long long *__restrict__ depth64,
                                                            unsigned long long *__restrict__ set, unsigned mask)
{
    const int i = blockIdx.x * FL_BLOCK + threadIdx.x;
    if (i >= n) return;
    fl_depth_point(scan, i, S, VC, depth64);
    fl_subkey_point(scan, i, set, mask);
}

// The scan stays on the device (fl_vio_detect with n_pg = FL_DETECT_SCAN_ON_DEVICE): pg = the handle's staged scan registered under
// the state block (pointBodyToWorld, laserMapping.cpp:695-698 -- what fl_lio_get_world_points hands to the host), written twice: as
// x, y, z (what addSparseMap's kernels read) and as x, y, z, 0 for the voxel filter (downSizeFilter, lidar_selection.cpp:352-353), whose
// per-workgroup bounding boxes come out of the same launch (as undistort_apply_kernel does for the LiDAR front).
__global__ __launch_bounds__(FL_BLOCK) void detect_world_kernel(const float *__restrict__ body, int n, const FlDev18 *__restrict__ D,
                                                               float *__restrict__ world3, float4 *__restrict__ world4, FlVxPartial *__restrict__ box)
{
    const int i = blockIdx.x * FL_BLOCK + threadIdx.x;
    float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) {
        fl_world_point(D, body[i * 3], body[i * 3 + 1], body[i * 3 + 2], p.x, p.y, p.z);
        world3[i * 3] = p.x; world3[i * 3 + 1] = p.y; world3[i * 3 + 2] = p.z;
        world4[i] = p;
    }
    static_assert(FL_BLOCK == 256, "fl_vx_block_box");
    __shared__ unsigned s_red[4][7];
    unsigned mn[3] = {0u, 0u, 0u}, mx[3] = {0u, 0u, 0u};
    int cnt = 0;
    if (i < n && isfinite(p.x) && isfinite(p.y) && isfinite(p.z)) {
        mn[0] = ~fl_vx_enc(p.x); mn[1] = ~fl_vx_enc(p.y); mn[2] = ~fl_vx_enc(p.z);
        mx[0] = fl_vx_enc(p.x); mx[1] = fl_vx_enc(p.y); mx[2] = fl_vx_enc(p.z);
        cnt = 1;
    }
    const FlVxPartial r = fl_vx_block_box(mn, mx, cnt, s_red);
    if (threadIdx.x == 0) box[blockIdx.x] = r;
}
// vmap_scan_kernel over the voxel filter's output (x, y, z, intensity; its count on the device), launched for the scan's full size
__global__ __launch_bounds__(FL_BLOCK) void vmap_scan4_kernel(const float4 *__restrict__ scan4, const FlVxCtl *__restrict__ vx, const FlSelectParams *__restrict__ S,
                                                             const FlVioConst *__restrict__ VC, unsigned long long *__restrict__ depth64,
                                                             unsigned long long *__restrict__ set, unsigned mask)
{
    const int i = blockIdx.x * FL_BLOCK + threadIdx.x;
    if (vx->cells_short || i >= vx->count) return;
    fl_depth_point<4>(reinterpret_cast<const float *>(scan4), i, S, VC, depth64);
    fl_subkey_point<4>(reinterpret_cast<const float *>(scan4), i, set, mask);
}

// addObservation under the pose of the state ComputeJ left (derived here as the host's vmap_frame_pose derives it: Rcw = Rci R^T,
// Pcw = -Rcw p + Pci, the same operations in the same order), sized for the upper bound of the selection; the workgroup that finishes
// last writes the frame's counts behind the state block and serves the result mailbox.
__global__ __launch_bounds__(FL_BLOCK) void vmap_addobs_dev_kernel(FlVPoint *__restrict__ pts, const FlDetectParams *__restrict__ P,
                                                                  const FlVioConst *__restrict__ VC, const uint8_t *__restrict__ img,
                                                                  const int32_t *__restrict__ sel_point, const int32_t *__restrict__ levels,
                                                                  const int32_t *__restrict__ n_sel_ptr, FlVmapCount *__restrict__ cnt,
                                                                  FlDev18 *__restrict__ D, unsigned *__restrict__ ticket,
                                                                  unsigned long long *pub_flag, void *pub_dst, unsigned long long pub_seq,
                                                                  const FlVxCtl *__restrict__ vx /* nullable: the down-sampling's control block */)
{
    __shared__ double s_pose[15];            // Rcw, Pcw, fpos
    __shared__ int s_last;
    const int n_sel = *n_sel_ptr;
    const bool abandoned = (D->status & FL_NUM_TIMEOUT) != 0;      // ComputeJ did not finish: the host resumes it and adds the observations then
    if (threadIdx.x == 0) {
        const double *rot = D->x, *pos = D->x + 9;
        double Rcw[9], Pcw[3];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                double a = 0.0;
                for (int k = 0; k < 3; k++) a += P->Rci[i * 3 + k] * rot[j * 3 + k];
                Rcw[i * 3 + j] = a;
            }
        for (int i = 0; i < 3; i++) {
            double a = 0.0;
            for (int k = 0; k < 3; k++) a += Rcw[i * 3 + k] * pos[k];
            Pcw[i] = -a + P->Pci[i];
        }
        for (int i = 0; i < 9; i++) s_pose[i] = Rcw[i];
        for (int i = 0; i < 3; i++) s_pose[9 + i] = Pcw[i];
        for (int i = 0; i < 3; i++) s_pose[12 + i] = -(Rcw[i] * Pcw[0] + Rcw[3 + i] * Pcw[1] + Rcw[6 + i] * Pcw[2]);   // new_frame_->pos()
    }
    __syncthreads();
    const int i = blockIdx.x * FL_BLOCK + threadIdx.x;
    int added = 0;
    if (i < n_sel && !abandoned)
        added = fl_addobs_point(pts + sel_point[i], s_pose, s_pose + 9, s_pose + 12, P->vm.kf_id, P->vm.frame_id, VC, img, levels[i]);
    const unsigned long long b = __ballot(added != 0);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(&cnt->obs_added, (int)__popcll(b));
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) s_last = (atomicAdd(ticket, 1u) == gridDim.x - 1u);
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) {
        FlDetectTail *T = reinterpret_cast<FlDetectTail *>(reinterpret_cast<char *>(D) + sizeof(FlDev18) + FL_DETECT_TAIL_OFF);
        // (the six words in flight together, then the stores: one L2 round trip on the frame's last stretch, not six)
        const int c_cand = __hip_atomic_load(&cnt->cand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int c_added = __hip_atomic_load(&cnt->added, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int c_obs = __hip_atomic_load(&cnt->obs_added, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const FlVxCtl *vxs = vx ? vx : reinterpret_cast<const FlVxCtl *>(D);        // (a readable address either way: no branch between the loads)
        const int v_count = vxs->count, v_short = vxs->cells_short;
        const long long v_cells = vxs->cells;
        T->n_cand = c_cand; T->n_selected = n_sel; T->n_added = c_added; T->n_observed = c_obs;
        T->n_down = vx ? v_count : 0; T->vox_cells_short = vx ? v_short : 0; T->vox_cells = vx ? v_cells : 0ll; T->pad0 = T->pad1 = 0;
        D->pub_flag = pub_flag; D->pub_dst = pub_dst; D->pub_seq = pub_seq;
        *ticket = 0u;
        __threadfence();
    }
    fl_publish_state(D);
}

// how many observations of the map refer to keyframe `kf` (fl_vio_drop_keyframe refuses while any does: the reference keeps such an
// image alive through the Feature's cv::Mat)
__global__ __launch_bounds__(FL_BLOCK) void vmap_kfrefs_kernel(const FlVPoint *__restrict__ pts, int n, int kf, int *__restrict__ count)
{
    const int i = blockIdx.x * FL_BLOCK + threadIdx.x;
    int c = 0;
    if (i < n)
        for (int k = 0; k < pts[i].n_obs; k++) c += pts[i].obs[k].kf_id == kf;
    if (c) atomicAdd(count, c);
}

// references per keyframe over the whole map (fl_vmap_release_keyframes)
__global__ __launch_bounds__(FL_BLOCK) void vmap_kfhist_kernel(const FlVPoint *__restrict__ pts, int n, int n_kf, int *__restrict__ hist)
{
    const int i = blockIdx.x * FL_BLOCK + threadIdx.x;
    if (i >= n) return;
    for (int k = 0; k < pts[i].n_obs; k++) {
        const int kf = pts[i].obs[k].kf_id;
        if (kf >= 0 && kf < n_kf) atomicAdd(&hist[kf], 1);
    }
}
