// knn_kernels.h -- SURVEY.md section 8(f) row N1: exact 5-nearest-neighbour search of the scan points in the
// LiDAR map ON THE DEVICE, replacing the host ikd-Tree search of the 2 search passes per frame
// (KD_TREE::Nearest_Search, include/ikd-Tree/ikd_Tree.cpp:350-380, call sites src/laserMapping.cpp:1543,
// :1002) and with it the 0.6 MB world-point read-back and the 3 MB neighbour restage.
//
// Index: uniform voxel grid over the map, stored sparsely -- map points grouped by their 63-bit cell key
// (counting placement, once per map update), an open-addressing hash table cell-key -> first point of
// the cell. It is rebuilt whenever the map changes (map_incremental, laserMapping.cpp:692-706).
// Search (hand-written): one lane per scan point; world point from the current state exactly like the
// residual kernels; cells are visited in growing Chebyshev rings around the query's cell and the
// search stops as soon as the 5th best distance is <= (ring * cell)^2, which proves that no
// unvisited point can be closer (any point outside the visited cube is farther than ring*cell).
// Rings are capped where (ring * cell)^2 > 5: farther neighbours make the point invalid anyway
// (laserMapping.cpp:1549: sqdist[4] > 5). Distances are the ikd-Tree's float arithmetic
// (ikd_Tree.cpp:1291-1295); exact ties are broken by the lower map index (the tree's traversal
// order is not reproducible; oracle/orc_knn.c uses the same rule).
// The plane fit (K0) is fused: the kernel writes the plane and the selection flag directly.
#pragma once

#include "fl_device.h"
#include "fl_math.h"
#include "fl_ikfom_math.h"

#define FL_KNN_EMPTY 0xFFFFFFFFFFFFFFFFull
#define FL_KNN_QPB 64            // queries per workgroup of the search kernel
#define FL_KNN_NT 256
#ifndef FL_KNN_BATCH
#define FL_KNN_BATCH 8           // point loads in flight per lane (4: the same 39 us per 50 k-point search, 16: 42.5 us -- the walk is VALU bound)
#endif

// hash-table entry: cell key, first sorted point of the cell, number of points in the cell
struct __attribute__((aligned(16))) FlCellEntry {
    unsigned long long key;
    unsigned start;
    unsigned count;
};

struct FlMapGrid {
    const float4 *pts;               // grouped by cell: xyz + original index (as int bits) in w
    const float *raw;                // the map as staged (k x 3), addressed by original index
    const FlCellEntry *htab;         // open-addressing table, key == FL_KNN_EMPTY: free slot
    unsigned hmask;                  // table size - 1 (power of two)
    const unsigned long long *ckeys; // coarse occupancy set: keys of the 4x4x4-cell blocks that hold at least one point
    unsigned cmask;
    int npts;
    float cell;                      // edge length
    float inv_cell;
    int max_ring;                    // smallest r with (r*cell)^2 > 5 (+1 for the margin of the stop rule)
};

__device__ __forceinline__ unsigned long long fl_cell_key(int ix, int iy, int iz)
{
    return ((unsigned long long)(unsigned)(ix + (1 << 20)) << 42) | ((unsigned long long)(unsigned)(iy + (1 << 20)) << 21) |
           (unsigned long long)(unsigned)(iz + (1 << 20));
}
// table slot of a cell. The three 21-bit biased coordinates are mixed with full-rate 24-bit multiplies (the 64-bit
// multiplies of a general-purpose mixer are quarter rate and the search kernel is VALU bound).
__device__ __forceinline__ unsigned fl_hash64(unsigned long long k)
{
    const unsigned z = (unsigned)k & 0x1FFFFFu, y = (unsigned)(k >> 21) & 0x1FFFFFu, x = (unsigned)(k >> 42) & 0x1FFFFFu;
    unsigned h = __umul24(x, 0x9E3779u) ^ __umul24(y, 0x85EBCBu) ^ __umul24(z, 0xC2B2AFu);
    h ^= h >> 15;
    h = __umul24(h & 0xFFFFFFu, 0x27D4EBu) ^ (h >> 11);
    return h;
}

// coarse block (4 x 4 x 4 cells) of a cell, from the biased 21-bit coordinates packed in its key
__device__ __forceinline__ unsigned long long fl_coarse_key(unsigned long long cell_key)
{
    const unsigned long long m = 0x1FFFFFull;
    return ((((cell_key >> 42) & m) >> 2) << 42) | ((((cell_key >> 21) & m) >> 2) << 21) | ((cell_key & m) >> 2);
}
__device__ __forceinline__ bool fl_coarse_occupied(const unsigned long long *ckeys, unsigned cmask, unsigned long long ck)
{
    unsigned h = fl_hash64(ck * 0x9E3779B97F4A7C15ull >> 1) & cmask;
    while (true) {
        const unsigned long long k = ckeys[h];
        if (k == ck) return true;
        if (k == FL_KNN_EMPTY) return false;
        h = (h + 1) & cmask;
    }
}

// Index build = grouping by cell, not a sort: the search only needs each cell's points contiguous (its tie rule compares the
// original indices stored in w, so the order inside a cell is irrelevant). Three light passes instead of a 64-bit radix/merge
// sort of the whole map (~110 us -> ~35 us at 200 k points):
//   knn_count_kernel    every point finds/claims its cell's slot and takes a rank inside the cell (atomicAdd on the slot's count;
//                       a cell holds a handful of points, so the atomics do not pile up); the first point of a cell also marks
//                       the cell's 4x4x4 block in the coarse occupancy set
//   (exclusive prefix sum over the slots' counts -> first point of every cell)
//   knn_place_kernel    every point goes to first[slot] + rank
__global__ __launch_bounds__(FL_BLOCK) void knn_table_init_kernel(FlCellEntry *__restrict__ htab, unsigned long long *__restrict__ ckeys,
                                                                 unsigned slots, unsigned *__restrict__ occ_found)
{
    const unsigned i = blockIdx.x * FL_BLOCK + threadIdx.x;
    if (i == 0) *occ_found = 0u;
    if (i >= slots) return;
    FlCellEntry e;
    e.key = FL_KNN_EMPTY; e.start = 0u; e.count = 0u;
    htab[i] = e;
    ckeys[i] = FL_KNN_EMPTY;
}

__global__ __launch_bounds__(FL_BLOCK) void knn_count_kernel(const float *__restrict__ map_xyz, int k, float inv_cell,
                                                            FlCellEntry *__restrict__ htab, unsigned hmask,
                                                            unsigned long long *__restrict__ ckeys, unsigned cmask,
                                                            unsigned *__restrict__ slot_of, unsigned *__restrict__ rank_of)
{
    const int i = blockIdx.x * FL_BLOCK + threadIdx.x;
    if (i >= k) return;
    const int ix = (int)floorf(map_xyz[i * 3] * inv_cell), iy = (int)floorf(map_xyz[i * 3 + 1] * inv_cell),
              iz = (int)floorf(map_xyz[i * 3 + 2] * inv_cell);
    const unsigned long long key = fl_cell_key(ix, iy, iz);
    unsigned h = fl_hash64(key) & hmask;
    while (true) {
        const unsigned long long prev = atomicCAS((unsigned long long *)&htab[h].key, FL_KNN_EMPTY, key);
        if (prev == FL_KNN_EMPTY || prev == key) break;
        h = (h + 1) & hmask;
    }
    const unsigned r = atomicAdd(&htab[h].count, 1u);
    slot_of[i] = h;
    rank_of[i] = r;
    if (r == 0u) {                                             // first point of its cell: the cell's 4x4x4 block is occupied
        const unsigned long long ck = fl_coarse_key(key);
        unsigned hc = fl_hash64(ck * 0x9E3779B97F4A7C15ull >> 1) & cmask;
        while (true) {
            const unsigned long long prev = atomicCAS(&ckeys[hc], FL_KNN_EMPTY, ck);
            if (prev == FL_KNN_EMPTY || prev == ck) break;
            hc = (hc + 1) & cmask;
        }
    }
}

// density estimate for the self-tuning cell size: occupied slots among the first `sample` slots of the table (the hash spreads
// the cells uniformly, so occupied cells ~= found * slots / sample)
__global__ __launch_bounds__(FL_BLOCK) void knn_occupancy_kernel(const FlCellEntry *__restrict__ htab, unsigned sample, unsigned *__restrict__ found)
{
    const unsigned i = blockIdx.x * FL_BLOCK + threadIdx.x;
    const bool occ = i < sample && htab[i].key != FL_KNN_EMPTY;
    const unsigned long long b = __ballot(occ);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(found, (unsigned)__popcll(b));
}

struct FlCellCount {
    __host__ __device__ __forceinline__ unsigned operator()(const FlCellEntry &e) const { return e.count; }
};
// capacity of a cell's region in the point pool for c points (round 5, mapinc_kernels.h: the map is updated in place, so every cell gets
// room behind its points): half as many again, at least 4, a multiple of 4
__host__ __device__ __forceinline__ unsigned fl_mi_cap(unsigned c) { const unsigned w = c + (c >> 1) + 3u; return w < 4u ? 4u : (w & ~3u); }
struct FlCellCap {
    __host__ __device__ __forceinline__ unsigned operator()(const FlCellEntry &e) const { return e.count ? fl_mi_cap(e.count) : 0u; }
};

__global__ __launch_bounds__(FL_BLOCK) void knn_place_kernel(const float *__restrict__ map_xyz, int k, const unsigned *__restrict__ slot_of,
                                                            const unsigned *__restrict__ rank_of, const unsigned *__restrict__ first,
                                                            float4 *__restrict__ pts, FlCellEntry *__restrict__ htab,
                                                            unsigned *__restrict__ cellcap)
{
    const int i = blockIdx.x * FL_BLOCK + threadIdx.x;
    if (i >= k) return;
    const unsigned s = slot_of[i], r = rank_of[i], f = first[s];      // first[]: prefix over the cells' CAPACITIES (FlCellCap)
    pts[f + r] = make_float4(map_xyz[i * 3], map_xyz[i * 3 + 1], map_xyz[i * 3 + 2], __int_as_float(i));
    if (r == 0u) { htab[s].start = f; cellcap[s] = fl_mi_cap(htab[s].count); }
}

// Sorted best-5 list of keys. key bits = (float bits of the squared distance + 0x00100000) << 32 | original map
// index. Squared distances are non-negative floats, so the unsigned order of the keys is (distance, then lower map
// index) -- the tie rule of oracle/orc_knn.c. The bit pattern is handled as a positive, normal, finite DOUBLE (the
// bias keeps the exponent field in 1..0x7F9), whose IEEE order equals the unsigned order: a compare-exchange is one
// v_min_f64 + one v_max_f64, both exact selections.
struct FlTop5 {
    double key[5];
};
__device__ __forceinline__ double fl_knn_key(float d, int id) { return __hiloint2double((int)(__float_as_uint(d) + 0x00100000u), id); }
__device__ __forceinline__ float fl_knn_key_d(double key) { return __uint_as_float((unsigned)__double2hiint(key) - 0x00100000u); }
__device__ __forceinline__ int fl_knn_key_id(double key) { return __double2loint(key); }
#define FL_KNN_NO_ID 0x7fffffff
__device__ __forceinline__ void fl_top5_clear(FlTop5 &t)
{
#pragma unroll
    for (int k = 0; k < 5; k++) t.key[k] = fl_knn_key(INFINITY, FL_KNN_NO_ID);
}
// branch-free insertion: 5 compare-exchanges carry the displaced key down the list
__device__ __forceinline__ void fl_top5_insert(FlTop5 &t, double key)
{
#pragma unroll
    for (int p = 0; p < 5; p++) {
        const double lo = fmin(key, t.key[p]);
        key = fmax(key, t.key[p]);
        t.key[p] = lo;
    }
}

__device__ __forceinline__ void fl_scan_cell(const FlMapGrid &G, int ix, int iy, int iz, float qx, float qy, float qz, FlTop5 &t)
{
    const unsigned long long key = fl_cell_key(ix, iy, iz);
    unsigned h = fl_hash64(key) & G.hmask;
    unsigned start, count;
    while (true) {
        const uint4 e = *reinterpret_cast<const uint4 *>(&G.htab[h]);
        const unsigned long long hk = ((unsigned long long)e.y << 32) | e.x;
        if (hk == FL_KNN_EMPTY) return;
        if (hk == key) { start = e.z; count = e.w; break; }
        h = (h + 1) & G.hmask;
    }
    for (unsigned j = start; j < start + count; j++) {
        const float4 p = G.pts[j];
        const float dx = qx - p.x, dy = qy - p.y, dz = qz - p.z;
        const float d = dx * dx + dy * dy + dz * dz;        // ikd_Tree.cpp:1293 (no contraction)
        fl_top5_insert(t, fl_knn_key(d, __float_as_int(p.w)));
    }
}

// ---- quad (4 lanes) cross-lane helpers -------------------------------------------------------
__device__ __forceinline__ double quad_min_f64(double v)
{
    v = fmin(v, dpp_f64<FL_DPP_QUAD_XOR1>(v));
    return fmin(v, dpp_f64<FL_DPP_QUAD_XOR2>(v));
}
__device__ __forceinline__ unsigned quad_sum_u32(unsigned v)
{
    v += __builtin_amdgcn_update_dpp(0u, v, FL_DPP_QUAD_XOR1, 0xf, 0xf, false);
    return v + __builtin_amdgcn_update_dpp(0u, v, FL_DPP_QUAD_XOR2, 0xf, 0xf, false);
}

// Merge the 4 lanes' sorted local lists into the quad's global best 5 (same in the 4 lanes). Keys are unique
// per map point, so a round has one winner (or only INF keys are left).
__device__ __forceinline__ void quad_merge_top5(const FlTop5 &loc, FlTop5 &g)
{
    FlTop5 w = loc;
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const double best = quad_min_f64(w.key[0]);
        const bool win = (w.key[0] == best);
        g.key[k] = best;
#pragma unroll
        for (int p = 0; p < 4; p++) w.key[p] = win ? w.key[p + 1] : w.key[p];
        w.key[4] = win ? fl_knn_key(INFINITY, FL_KNN_NO_ID) : w.key[4];
    }
}

// hash lookup of one cell: returns count (0: empty) and start
__device__ __forceinline__ unsigned fl_cell_lookup(const FlMapGrid &G, unsigned long long key, uint4 first, unsigned h, unsigned *start)
{
    uint4 e = first;
    while (true) {
        const unsigned long long hk = ((unsigned long long)e.y << 32) | e.x;
        if (hk == FL_KNN_EMPTY) return 0u;
        if (hk == key) { *start = e.z; return e.w; }
        h = (h + 1) & G.hmask;
        e = *reinterpret_cast<const uint4 *>(&G.htab[h]);
    }
}

// MODE 18: world point from FlDev18 ; MODE 23: from FlDev23. `cond` != 0: run only when the device raised
// need_search and not stop (the frame drivers enqueue it before every pass).
//
// Workgroup = 256 threads = 64 queries, one quad (4 lanes) per query -- 50 k scan points alone are fewer
// than one wave per SIMD, so a lane-per-query kernel is purely latency bound; 4 lanes per query give
// 3+ waves per SIMD and short dependent chains. Phase 1a: the quad looks up the 27 cells of rings 0..1
// (7 independent table loads per lane). Phase 1b: the ~100 candidate points of those cells are split
// evenly over the 4 lanes (prefix over the cell counts in LDS) -- the cells themselves are very unevenly
// filled, a plane crosses 9 of the 27. Each lane keeps a sorted local best-5; 5 rounds of a quad DPP
// min-reduction over (distance, index) keys merge them. Farther rings (sparse regions only) are scanned
// cell-round-robin. Phase 2: one lane per query gathers the 5 neighbours and fits the plane (K0 fused) on
// full waves.
template <typename DEV> __device__ __forceinline__ void fl_search_prepare(DEV *D) {}
// (inline: as a `noinline` function it is compiled for 180 VGPRs, which the kernel inherits -- occupancy 2 instead of 4, the frame
// 13 us slower; inlined the kernel needs 106)
template <> __device__ __forceinline__ void fl_search_prepare<FlDev18>(FlDev18 *D)
{
    if (threadIdx.x < 128) eskf18_prepare_body(D);      // (threads 128.. leave: the body's barriers count arrivals of live wavefronts only)
}
template <int MODE, typename DEV>
__global__ __launch_bounds__(FL_KNN_NT) void lio_search_fit_kernel(const float *__restrict__ body, int n, FlMapGrid G, DEV *__restrict__ D,
                                                                  float4 *__restrict__ plane, uint8_t *__restrict__ sel,
                                                                  float *__restrict__ nbr_out /* nullable n x 15 */,
                                                                  uint8_t *__restrict__ valid_out /* nullable */, int cond,
                                                                  float4 *__restrict__ gate_out /* nullable */,
                                                                  float *__restrict__ body_keep /* nullable */,
                                                                  const DEV *__restrict__ host_state /* nullable */,
                                                                  float4 *ids /* nullable n x 5: the 5 winners of every query (position, map index), kept between searches */,
                                                                  int incremental, const int *__restrict__ n_dev = nullptr)
{
    // n_dev != nullptr (fl_lidar_front): the scan was produced by the voxel filter earlier in the stream and only the device knows its
    // size -- `n` is the capacity the grid was sized for, the workgroups beyond the real scan leave at once.
    if (n_dev) n = *n_dev;
    // ids / incremental (the SECOND search of a frame, and every later one over the same scan and map): the winners of the search
    // before (kept with their positions, so the bound needs no second trip through the map) are 5 distinct map points, so the largest of their distances to the query's NEW world point bounds the new 5th-best
    // distance from above (B2) whatever the pose did in between. Only cells whose box lies within sqrt(B2) of the query can hold a
    // point that enters the best 5 (or ties with its 5th): the quad looks up and walks those cells -- typically 3-6 of the 27 --
    // and the result is the full search's, key for key (same float distances, same index tie rule). Where the old winners are
    // farther than one cell edge (sparse map) or fewer than 5, the query falls back to the full walk.
    // host_state != nullptr (same launch as cond & 4): the state block has NOT been copied to the device -- it is still in the
    // caller's page-locked mirror (this is its device address). The prepare workgroup copies it into D (7 KB over the host link)
    // before it forms the constants; the search workgroups read the pose and the extrinsics (36 doubles) from the mirror directly,
    // next to their scan bytes. The copy command of the state block and the gap behind it (3.5 + 4.6 us) leave the frame.
    // body_keep != nullptr (fl_lio_frame18_dev with the scan in page-locked host memory): `body` is that HOST buffer as the device
    // addresses it -- every workgroup fetches its 64 points over the host link itself (768 B, each byte once, through LDS) and leaves
    // them in body_keep (the device copy the second search and later calls read). The fetches of the workgroups that are served
    // later overlap the searches of those served first: the scan's copy command (20 us at 50 k points) and the 8 us between a DMA
    // copy and the first kernel behind it are gone from the frame.
    // cond & 4 (the FIRST search of an 18-state frame, fl_lio_frame18_dev): the launch has one workgroup more than the scan needs,
    // and that workgroup forms the gain-solve constants of the state block (what eskf18_prepare_kernel does) beside the search;
    // gate_out != nullptr (same launch): every point's gate threshold (what lio_gate_kernel does) is written with its plane. Two
    // launches and two kernel boundaries less per frame; the first search of a frame always runs (begin raises need_search).
    if ((cond & 4) && blockIdx.x == gridDim.x - 1) {      // (Mode-23: the copy only -- fl_search_prepare<FlDev23> is empty)
        if (host_state) {
            const unsigned long long *src = reinterpret_cast<const unsigned long long *>(host_state);
            unsigned long long *dst = reinterpret_cast<unsigned long long *>(D);
            constexpr int WORDS = (int)(sizeof(DEV) / 8), PER = (WORDS + FL_KNN_NT - 1) / FL_KNN_NT;
            unsigned long long v[PER];
#pragma unroll
            for (int k = 0; k < PER; k++) { const int w = (int)threadIdx.x + FL_KNN_NT * k; v[k] = w < WORDS ? __builtin_nontemporal_load(src + w) : 0ull; }
#pragma unroll
            for (int k = 0; k < PER; k++) { const int w = (int)threadIdx.x + FL_KNN_NT * k; if (w < WORDS) dst[w] = v[k]; }
            __threadfence();
            __syncthreads();
        }
        if constexpr (MODE == 18) { if (n_dev && threadIdx.x == 0) D->n_scan = n; }      // (behind the copy of the block: the mirror does not know it)
        fl_search_prepare<DEV>(D);
        return;
    }
    if ((int)blockIdx.x * FL_KNN_QPB >= n) return;         // capacity-sized grid (n_dev), or an empty scan
    int vb = (int)blockIdx.x;
    {
        const int nqb = (n + FL_KNN_QPB - 1) / FL_KNN_QPB, per = nqb >> 3;
        // XCD-aware mapping (round 6): workgroup b runs on XCD b % 8, each XCD has an L2 of its own -- XCD x serves a CONTIGUOUS eighth of the
        // queries (which arrive in voxel order: neighbours probe the same map cells) instead of every eighth workgroup, so a map cell is
        // fetched into one L2, not eight. Worth 2 us of the frame's two searches (0.116 -> 0.114 ms, interleaved A/B); results are per query.
        if (vb < per * 8) vb = (vb & 7) * per + (vb >> 3);
    }
    // (the first search of a frame always runs: begin raises need_search; with host_state the block is not on the device yet)
    if (!host_state && (cond & 1)) {
        const int need = D->need_search, stop = D->stop, status = D->status;      // (in flight together, not one round trip after the other)
        if (!need || stop || (status & 8 /* FL_NUM_TIMEOUT: abandoned chain */)) return;
    }
    const bool stamp = (cond & 2) && threadIdx.x == 0 && blockIdx.x < 512;
    FL_INSTR(if (stamp) g_fl_wall[blockIdx.x] = (long long)wall_clock64();)
    __shared__ int s_at[FL_KNN_QPB][5];
    __shared__ float s_d5[FL_KNN_QPB];
    __shared__ unsigned s_cstart[FL_KNN_QPB][28];
    __shared__ unsigned s_ccnt[FL_KNN_QPB][28];
    const int ql = (int)(threadIdx.x >> 2), j = (int)(threadIdx.x & 3u);
    const int q0 = vb * FL_KNN_QPB;
    const int i = min(q0 + ql, n - 1);                // tail quads repeat the last query (results unused)
    // previous winners of this query (incremental): position + map index, 5 x float4 per point
    float4 rec_mine = make_float4(0.f, 0.f, 0.f, 0.f), rec_4th = rec_mine;
    if (incremental) { rec_mine = ids[(size_t)i * 5 + j]; rec_4th = ids[(size_t)i * 5 + 4]; }

    __shared__ float s_body[FL_KNN_QPB * 3];
    if (body_keep) {
        const int cnt = min(FL_KNN_QPB, n - q0) * 3;
        if ((int)threadIdx.x < cnt) {
            const float v = __builtin_nontemporal_load(body + (size_t)q0 * 3 + threadIdx.x);
            s_body[threadIdx.x] = v;
            body_keep[(size_t)q0 * 3 + threadIdx.x] = v;
        }
        __syncthreads();
    }
    const int il = i - q0;
    const float pb[3] = {body_keep ? s_body[il * 3] : body[i * 3], body_keep ? s_body[il * 3 + 1] : body[i * 3 + 1],
                         body_keep ? s_body[il * 3 + 2] : body[i * 3 + 2]};
    float pw[3];
    if constexpr (MODE == 18) {
        const FlDev18 *D18 = host_state ? host_state : D;
        const double b0 = (double)pb[0], b1 = (double)pb[1], b2 = (double)pb[2];
        const double u0 = (D18->R_LI[0] * b0 + D18->R_LI[1] * b1 + D18->R_LI[2] * b2) + D18->t_LI[0];
        const double u1 = (D18->R_LI[3] * b0 + D18->R_LI[4] * b1 + D18->R_LI[5] * b2) + D18->t_LI[1];
        const double u2 = (D18->R_LI[6] * b0 + D18->R_LI[7] * b1 + D18->R_LI[8] * b2) + D18->t_LI[2];
        pw[0] = (float)((D18->x[0] * u0 + D18->x[1] * u1 + D18->x[2] * u2) + D18->x[9]);
        pw[1] = (float)((D18->x[3] * u0 + D18->x[4] * u1 + D18->x[5] * u2) + D18->x[10]);
        pw[2] = (float)((D18->x[6] * u0 + D18->x[7] * u1 + D18->x[8] * u2) + D18->x[11]);
    } else {
        const FlDev23 *D23 = host_state ? host_state : D;
        double xs[FL_X23_LEN], p_i[3];
#pragma unroll
        for (int k = 0; k < FL_X23_LEN; k++) xs[k] = D23->x[k];
        fl_world_point23(xs, pb, p_i, pw);
    }
    const int cx = (int)floorf(pw[0] * G.inv_cell), cy = (int)floorf(pw[1] * G.inv_cell), cz = (int)floorf(pw[2] * G.inv_cell);

    // ---- incremental bound: B2 = largest new distance of the previous winners (lane j takes winner j, lane 0 also winner 4)
    float B2s = INFINITY;                             // prune threshold, INFINITY = full search
    if (incremental) {
        const float4 wa = rec_mine, wb = rec_4th;     // (loaded at the top: the addresses depend on the query index alone)
        float dm = INFINITY;
        if (__float_as_int(wa.w) != FL_KNN_NO_ID && __float_as_int(wb.w) != FL_KNN_NO_ID) {
            const float d0x = pw[0] - wa.x, d0y = pw[1] - wa.y, d0z = pw[2] - wa.z;
            const float d1x = pw[0] - wb.x, d1y = pw[1] - wb.y, d1z = pw[2] - wb.z;
            dm = fmaxf(d0x * d0x + d0y * d0y + d0z * d0z, d1x * d1x + d1y * d1y + d1z * d1z);
        }
        dm = fmaxf(dm, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, dm), FL_DPP_QUAD_XOR1, 0xf, 0xf, false)));
        dm = fmaxf(dm, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, dm), FL_DPP_QUAD_XOR2, 0xf, 0xf, false)));
        const float reach1 = fmaxf(G.cell - 1e-3f, 0.f);
        // inside ring 1 only (farther winners: the full path with its ring logic); 1e-5 relative slack covers the float rounding of
        // the box distance below, the 1 mm margin the float cell assignment (floorf(x * inv_cell)) -- as the stop rule of the full walk
        if (dm <= reach1 * reach1) B2s = dm * (1.0f + 1e-5f);
    }

    // ---- phase 1: rings 0..1 in at most two rounds. Cells c = j, j+4, ... < 27 ; c = (dz+1)*9 + (dy+1)*3 + (dx+1).
    // Round 0 takes the cells that must be looked at: with a bound from the previous winners (incremental) the cells within its
    // reach; without one the 2 x 2 x 2 block of cells around the query's nearest cell corner -- every map point within
    // dmin >= cell / 2 of the query lies in that block, so when the block's 5th-best distance is below dmin the search is over after
    // 8 look-ups and ~30 candidates instead of 27 and ~100 (the usual case where the map is dense: the 5th neighbour of a scan point
    // is a fraction of a cell away). Round 1 (only if some query of the wavefront needs it) takes the other 19 cells, pruned by
    // the bound round 0 produced. The lane-local lists run through both rounds: the result is the best 5 of everything visited, key
    // for key what one walk over all 27 cells gives.
    FlTop5 t, g;
    fl_top5_clear(t);
    const float reach1 = fmaxf(G.cell - 1e-3f, 0.f);      // (a cell edge below the 1 mm margin: no bound ever holds, every query walks on)
    const float fx = pw[0] * G.inv_cell - (float)cx, fy = pw[1] * G.inv_cell - (float)cy, fz = pw[2] * G.inv_cell - (float)cz;
    const int bx = fx < 0.5f ? -1 : 0, by = fy < 0.5f ? -1 : 0, bz = fz < 0.5f ? -1 : 0;      // the block: offsets {b, b + 1} per axis
    const float dmin = fmaxf(G.cell * fminf(fminf(fx < 0.5f ? 1.0f - fx : fx, fy < 0.5f ? 1.0f - fy : fy), fz < 0.5f ? 1.0f - fz : fz) - 1e-3f, 0.f);
    const bool blockmode = !(B2s < INFINITY);
    bool needB = false;
    float B2r = INFINITY;
    for (int round = 0; round < 2; round++) {
        if (round == 1 && __ballot(needB) == 0ull) break;             // (uniform over the wavefront)
        unsigned long long key[7];
        unsigned hs[7];
        uint4 first[7];
#pragma unroll
        for (int m = 0; m < 7; m++) {
            const int c = min(j + 4 * m, 26);
            const int ox = (c % 3) - 1, oy = ((c / 3) % 3) - 1, oz = (c / 9) - 1;
            key[m] = fl_cell_key(cx + ox, cy + oy, cz + oz);
            hs[m] = fl_hash64(key[m]) & G.hmask;
            // squared distance from the query to the cell's box (0 inside), each face pushed out by the 1 mm margin
            const float lx = (float)(cx + ox) * G.cell, ly = (float)(cy + oy) * G.cell, lz = (float)(cz + oz) * G.cell;
            const float gx = fmaxf(fmaxf(lx - pw[0], pw[0] - (lx + G.cell)) - 1e-3f, 0.f);
            const float gy = fmaxf(fmaxf(ly - pw[1], pw[1] - (ly + G.cell)) - 1e-3f, 0.f);
            const float gz = fmaxf(fmaxf(lz - pw[2], pw[2] - (lz + G.cell)) - 1e-3f, 0.f);
            const float box2 = gx * gx + gy * gy + gz * gz;
            const bool inblock = (ox == bx || ox == bx + 1) && (oy == by || oy == by + 1) && (oz == bz || oz == bz + 1);
            const bool look = (j + 4 * m < 27) &&
                              (round == 0 ? (blockmode ? inblock : !(box2 > B2s)) : (needB && !inblock && !(box2 > B2r)));
            first[m] = look ? *reinterpret_cast<const uint4 *>(&G.htab[hs[m]]) : make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u);   // (skipped: reads as an empty slot)
        }
        // compact the occupied cells (a plane crosses ~9 of the 27) into the quad's LDS list, in cell order:
        // rank = occupied cells of the earlier rows m + occupied cells of the lower lanes in this row
        unsigned mysum = 0;
        int nocc = 0;
        const int quad_shift = (int)(threadIdx.x & 60u);
#pragma unroll
        for (int m = 0; m < 7; m++) {
            unsigned st = 0;
            const unsigned cn = fl_cell_lookup(G, key[m], first[m], hs[m], &st);
            const unsigned bits = (unsigned)(__ballot(cn != 0u) >> quad_shift) & 0xFu;
            const int r = nocc + __popc(bits & ((1u << j) - 1u));
            if (cn != 0u) { s_cstart[ql][r] = st; s_ccnt[ql][r] = cn; }
            nocc += __popc(bits);
            mysum += cn;
        }
        if (j == 0) s_ccnt[ql][nocc] = 0x40000000u;      // terminator: the walk never advances past it
        const unsigned T = quad_sum_u32(mysum);
        // (the list of a query is written and read by its own quad only -- one wavefront: no workgroup barrier, which the
        // wavefront-uniform `break` above could not pair up anyway)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        FL_INSTR(if (stamp && round == 0) g_fl_wall[512 + blockIdx.x] = (long long)wall_clock64();)
        // ---- the walk: the T candidates of the concatenated cell ranges are split evenly over the 4 lanes, lane j
        // takes [j*T/4, (j+1)*T/4) -- balanced however unevenly the cells are filled. One flat loop over the
        // candidates (the 16 queries of a wave have their points in different cells: a loop over cells would run
        // every cell's longest range for the whole wave); every listed cell holds >= 1 point, so stepping to the
        // next candidate crosses at most one cell boundary (branch-free). Loads are issued FL_KNN_BATCH at a time.
        const unsigned seg_b = (T * (unsigned)j) >> 2, seg_e = (T * (unsigned)(j + 1)) >> 2;
        int c = 0;
        unsigned cell_end = s_ccnt[ql][0], base = s_cstart[ql][0];
        while (seg_b >= cell_end) {                      // position on the cell of the first candidate
            c++;
            base = s_cstart[ql][c] - cell_end;
            cell_end += s_ccnt[ql][c];
        }
        for (unsigned k0 = seg_b; k0 < seg_e; k0 += FL_KNN_BATCH) {
            unsigned pos[FL_KNN_BATCH];
#pragma unroll
            for (int u = 0; u < FL_KNN_BATCH; u++) {
                const unsigned k = min(k0 + (unsigned)u, seg_e - 1u);
                const bool adv = k >= cell_end;
                c += adv ? 1 : 0;
                const unsigned nst = s_cstart[ql][c], ncn = s_ccnt[ql][c];
                base = adv ? nst - cell_end : base;
                cell_end = adv ? cell_end + ncn : cell_end;
                pos[u] = base + k;
            }
            float4 p[FL_KNN_BATCH];
#pragma unroll
            for (int u = 0; u < FL_KNN_BATCH; u++) p[u] = G.pts[pos[u]];
#pragma unroll
            for (int u = 0; u < FL_KNN_BATCH; u++) {
                const bool live = k0 + (unsigned)u < seg_e;      // the tail of a batch repeats the last candidate: never inserted
                const float dx = pw[0] - p[u].x, dy = pw[1] - p[u].y, dz = pw[2] - p[u].z;
                const float d = dx * dx + dy * dy + dz * dz;        // ikd_Tree.cpp:1293 (no contraction)
                fl_top5_insert(t, fl_knn_key(live ? d : INFINITY, live ? __float_as_int(p[u].w) : FL_KNN_NO_ID));
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // (the quad's list is rewritten by the next round)
        __builtin_amdgcn_wave_barrier();
        if (round == 0) {
            if (blockmode) {
                quad_merge_top5(t, g);
                const float d5 = fl_knn_key_d(g.key[4]);
                needB = !(d5 <= dmin * dmin);                                    // a 5th neighbour may lie outside the block
                B2r = (d5 <= reach1 * reach1) ? d5 * (1.0f + 1e-5f) : INFINITY;   // (no bound inside ring 1: all of the other 19 cells)
            }
        }
    }
    {
        quad_merge_top5(t, g);
        // 1 mm safety margin: cell boundaries are evaluated in float (floorf(x * inv_cell)), exact to
        // well under a millimetre for maps of several kilometres
        float reach = reach1;
        if (!(fl_knn_key_d(g.key[4]) <= reach * reach)) {
            // ring 2 cell by cell (98 cells, round-robin over the quad): enough wherever the map is merely a little thin
            bool done = false;
            if (G.max_ring >= 2) {
                for (int c = j; c < 125; c += 4) {
                    const int dz = c / 25 - 2, rem = c % 25, dy = rem / 5 - 2, dx = rem % 5 - 2;
                    if (max(max(abs(dx), abs(dy)), abs(dz)) < 2) continue;      // visited by rings 0..1
                    fl_scan_cell(G, cx + dx, cy + dy, cz + dz, pw[0], pw[1], pw[2], t);
                }
                quad_merge_top5(t, g);
                reach = 2.0f * G.cell - 1e-3f;
                done = fl_knn_key_d(g.key[4]) <= reach * reach;
            }
            // sparse region: everything out to the ring cap, but only inside the 4x4x4-cell blocks that hold points at all (coarse
            // occupancy set) -- a query with nothing around costs <= 125 coarse probes instead of (2R+1)^3 cell probes
            if (!done && G.max_ring >= 3) {
                const int R = G.max_ring, B = 1 << 20;
                const int lx = (cx - R + B) >> 2, ly = (cy - R + B) >> 2, lz = (cz - R + B) >> 2;
                const int nx = ((cx + R + B) >> 2) - lx + 1, ny = ((cy + R + B) >> 2) - ly + 1, nz = ((cz + R + B) >> 2) - lz + 1;
                const int total = nx * ny * nz;
                for (int cc = j; cc < total; cc += 4) {
                    const int bz = lz + cc / (nx * ny), brem = cc % (nx * ny), by = ly + brem / nx, bx = lx + brem % nx;
                    const unsigned long long ck = ((unsigned long long)(unsigned)bx << 42) | ((unsigned long long)(unsigned)by << 21) | (unsigned long long)(unsigned)bz;
                    if (!fl_coarse_occupied(G.ckeys, G.cmask, ck)) continue;
                    for (int f = 0; f < 64; f++) {
                        const int ix = (bx << 2) + (f & 3) - B, iy = (by << 2) + ((f >> 2) & 3) - B, iz = (bz << 2) + (f >> 4) - B;
                        const int cheb = max(max(abs(ix - cx), abs(iy - cy)), abs(iz - cz));
                        if (cheb <= 2 || cheb > R) continue;                    // visited already / beyond the cap
                        fl_scan_cell(G, ix, iy, iz, pw[0], pw[1], pw[2], t);
                    }
                }
                quad_merge_top5(t, g);
            }
        }
        if (j == 0) {
#pragma unroll
            for (int k = 0; k < 5; k++) s_at[ql][k] = fl_knn_key_id(g.key[k]);
            s_d5[ql] = fl_knn_key_d(g.key[4]);
        }
    }
    __syncthreads();
    FL_INSTR(if (stamp) g_fl_wall[1024 + blockIdx.x] = (long long)wall_clock64();)

    // "the search for pass iters_run has been made": the pass kernels run when need_search is down or
    // searched_at == iters_run. No workgroup of this kernel reads searched_at, so one of them may write it.
    if (blockIdx.x == 0 && threadIdx.x == 0 && !host_state) D->searched_at = D->iters_run;      // (host_state: the block arrives with it set)
    const int qf = (int)threadIdx.x;
    const int iq = q0 + qf;
    if (qf >= FL_KNN_QPB || iq >= n) return;
    const int found5 = s_at[qf][4] != FL_KNN_NO_ID;
    const int valid = found5 && !(s_d5[qf] > 5.0f);
    float nb[15];
#pragma unroll
    for (int k = 0; k < 5; k++) {
        const int id = s_at[qf][k];                    // original map index of the k-th neighbour
        if (id != FL_KNN_NO_ID) {
            nb[k * 3] = G.raw[(size_t)id * 3]; nb[k * 3 + 1] = G.raw[(size_t)id * 3 + 1]; nb[k * 3 + 2] = G.raw[(size_t)id * 3 + 2];
        } else {
            nb[k * 3] = 0.f; nb[k * 3 + 1] = 0.f; nb[k * 3 + 2] = 0.f;
        }
    }
    if (ids) {
#pragma unroll
        for (int k = 0; k < 5; k++) ids[(size_t)iq * 5 + k] = make_float4(nb[k * 3], nb[k * 3 + 1], nb[k * 3 + 2], __int_as_float(s_at[qf][k]));
    }
    float pl[4];
    const int ok = fl_esti_plane(nb, pl);
    const bool keep = valid && ok && (pl[0] == pl[0]);
    plane[iq] = keep ? make_float4(pl[0], pl[1], pl[2], pl[3]) : make_float4(__builtin_nanf(""), 0.f, 0.f, 0.f);   // see lio_fit_planes_kernel
    sel[iq] = (uint8_t)keep;
    if (gate_out) {
        const float pg[3] = {body_keep ? s_body[qf * 3] : body[(size_t)iq * 3], body_keep ? s_body[qf * 3 + 1] : body[(size_t)iq * 3 + 1],
                             body_keep ? s_body[qf * 3 + 2] : body[(size_t)iq * 3 + 2]};
        gate_out[iq] = make_float4(pg[0], pg[1], pg[2], fl_gate_threshold(pg));
    }
    if (nbr_out) {
#pragma unroll
        for (int k = 0; k < 15; k++) nbr_out[(size_t)iq * 15 + k] = nb[k];
    }
    if (valid_out) valid_out[iq] = (uint8_t)valid;
    FL_INSTR(if (stamp) g_fl_wall[1536 + blockIdx.x] = (long long)wall_clock64();)
}
