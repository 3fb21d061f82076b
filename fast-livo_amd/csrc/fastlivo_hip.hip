// fastlivo_hip.hip -- C ABI (include/fastlivo_hip.h) over the gfx950 ESKF kernels.
// Built by hipcc --offload-arch=gfx950 -ffp-contract=off into fast-livo_amd/libfastlivo_hip.so.
// No CPU fallback: every entry point needs a live HIP device and fails loudly otherwise.
#include "../../include/fastlivo_hip.h"
#ifdef FL_INSTRUMENT
#include "../../include/fastlivo_hip_debug.h"
#endif

#include "fl_device.h"
#include "fl_math.h"
#include "lio_kernels.h"
#include "vio_kernels.h"
#include "ikfom_kernels.h"
#include "knn_kernels.h"
#include "voxel_kernels.h"
#include "mapupd_kernels.h"
#include "mapinc_kernels.h"
#include "vmap_kernels.h"
#include "imu_kernels.h"
#include "select_kernels.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <mutex>
#include <string>
#include <vector>

#define FL_MAX_BLOCKS 1024

struct fl_context {
    fl_config cfg;
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipStream_t aux_stream = nullptr;      // overlapped uploads (api_imu.inc)
    hipEvent_t aux_event = nullptr;
    // LIO
    float *d_body = nullptr, *d_nbr = nullptr, *d_world = nullptr;
    float4 *d_gate = nullptr;     // staged scan as (x, y, z, T), T = per-point selection threshold (fl_math.h: fl_gate_threshold); valid when gate_valid
    bool gate_valid = false;
    uint8_t *d_valid = nullptr, *d_sel = nullptr;
    float4 *d_knn_ids = nullptr;        // the 5 winners of every scan point's last search (position + map index, 5 x float4 per point)
    bool knn_ids_valid = false;         // ... filled for the staged scan and the staged map by a search that certainly ran
    bool search_certain = false;        // the next conditional search will run (a begin raised need_search)
    bool opt_incr_search = true;        // FL_OPT_INCR_SEARCH
    float4 *d_plane = nullptr, *d_normvec = nullptr;
    int cap_points = 0, n = 0;
    bool have_nbr = false;
    bool begun18 = false;         // an 18-state is on the device (fl_lio_begin18 / fl_vio_begin / frame drivers)
    int last_state_mode = 0;      // 18 / 23: which filter state was staged last (fl_map_add_points(NULL) registers the scan under it)
    int num_cus = 0;              // compute units of the device: the multi-pass kernels need every workgroup resident (<= 1 per CU)
    int mp_capacity = 0;          // workgroups of the LIO / VIO multi-pass kernels the device can hold at once (occupancy x CUs)
    int mp_capacity_ik = 0;       // ... of the Mode-23 multi-pass kernel (more registers: fewer per CU)
    int mp_capacity_q = 0, mp_capacity_ik_q = 0;   // what the occupancy query said (FL_OPT_MP_CAPACITY 0 restores it)
    unsigned *h_mp_done = nullptr;   // pinned host word the solver workgroup of a multi-pass launch writes its sequence number to when it ends
    unsigned *d_mp_done = nullptr;   // ... as the device addresses it
    unsigned mp_seq = 0;             // sequence number of this handle's last multi-pass launch
    int mp_last_grid = 0;
    int mp_fallbacks = 0, mp_resumes = 0;   // diagnostics: launches sent down the per-pass path by the admission check / frames resumed
    // Demotion (another compute client on the device that the admission check cannot see): a synchronous driver call whose chain ended
    // in an abandoned pass costs a full FL_GATHER_SPIN_LIMIT stall before the resume. After `demote_after` such calls in a row the
    // handle launches per pass for `demote_calls` calls (doubling on every relapse, up to 1024), then tries the multi-pass form again.
    int opt_demote_after = 2, opt_demote_calls = 64;
    int mp_consec_timeouts = 0, mp_demoted_left = 0, mp_demote_period = 0, mp_demotions = 0;
    // fl_set_option (include/fastlivo_hip.h)
    int opt_multipass = 1, opt_max_producers = 0, opt_ik_producers = 0, opt_vio_whole_cu = 1;
    bool normvec_valid = false;   // a pass with FL_ITER_KEEP_NORMVEC has run on the staged scan
    // 18-state block, reduction scratch
    FlDev18 *d_dev = nullptr;
    FlDev18 *h_dev = nullptr;      // pinned mirror
    // result mailbox of the frame drivers (fl_device.h fl_publish_state): the word the frame's last kernel writes, its device address,
    // the device address of the mirror, the sequence number of the last published frame
    unsigned long long *h_pub = nullptr, *d_pub = nullptr;
    void *d_hdev = nullptr;
    void *d_hdev23 = nullptr;           // device address of the page-locked FlDev23 mirror (state pull + result mailbox of the Mode-23 update)
    unsigned long long pub_seq = 0;
    int opt_scan_pull = 1;
    struct PinnedRange { void *host; size_t bytes; void *dev; };
    std::vector<PinnedRange> pinned;   // allocations of fl_host_alloc (kernels read scans in them in place)
    std::mutex pinned_mu;
    int opt_mailbox = 3;            // bit 0: fl_vio_compute_j, bit 1: fl_lio_frame18_dev
#define FL_UP_SLOTS 12
    void *d_small = nullptr;        // device address of h_small
    void *h_small = nullptr;        // page-locked scratch: 4 KB for the small per-call read-backs (counts, control blocks) + FL_UP_SLOTS x 1 KB for parameter uploads
    unsigned up_slot = 0;
    unsigned up_pending = 0;        // uploads through the slot ring since the last stream synchronisation this code knows of (upload_small)
    bool hdev_busy = false, hdev23_busy = false;   // an async copy from the pinned mirror may still be in flight (begin without a read-back since)
    void *d_records = nullptr;      // tagged per-workgroup records (handoff.h)
    size_t rec_fresh_bytes = 0;     // bytes of d_records the previous pass launch covered (records_for)
    unsigned *d_epoch = nullptr;    // launch epoch of the records, advanced on the device
    double *d_sums_tmp = nullptr;
    unsigned long long *d_bcast = nullptr;   // pose broadcast words of the multi-pass kernels (handoff.h)
    // visual map on the device (vmap_kernels.h)
    struct FlVPoint *d_vm_pts = nullptr;
    unsigned long long *d_vm_key = nullptr, *d_vm_best = nullptr, *d_vm_set = nullptr;
    int *d_vm_val = nullptr;
    int32_t *d_vm_num = nullptr, *d_vm_sel = nullptr;
    struct FlVmapParams *d_vm_prm = nullptr;
    struct FlVmapCount *d_vm_cnt = nullptr;
    float *d_vm_scan = nullptr;
    int vm_n = 0, vm_cap = 0, vm_length = 0, vm_grid = 0, vm_scan_cap = 0, vm_nsel = 0;
    bool vm_defer_count = false, vm_added_pending = false;   // fl_vio_detect: fl_vmap_add_sparse's count read with the frame's last block
    int vm_added_last = 0;
    unsigned vm_set_cap = 0;
    // peer exchange of the sharded form (api_p2p.inc)
    unsigned long long *d_xchg = nullptr;    // this rank's exchange buffer (fine-grained), [2][world][64] words
    unsigned long long *xchg_peer[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool xchg_ipc_open[8] = {false, false, false, false, false, false, false, false};
    unsigned *d_xepoch = nullptr;
    int xchg_alloc_world = 0, xchg_world = 1, xchg_rank = 0;
    // VIO
    FlVioConst *d_vc = nullptr;
    float *d_flat = nullptr;            // all-gathered per-patch floats of the sharded VIO pass (api_comm.inc), world + 1 chunks
    size_t flat_cap = 0;
    FlVioConst h_vc;
    uint8_t *d_img = nullptr;
    size_t cap_img = 0;
    float *d_ref = nullptr, *d_errors = nullptr;
    unsigned long long *d_err_words = nullptr;   // [2][cap_patches]: per-patch errors of a pass as self-validating words (solve18.h)
    double *d_pos = nullptr;
    int32_t *d_slevel = nullptr;
    int cap_patches = 0, m = 0;
    bool have_img = false;
    // Mode-23
    FlDev23 *d_dev23 = nullptr;
    FlDev23 *h_dev23 = nullptr;
    // device map grid for the k-NN (knn_kernels.h)
    float *d_map_raw = nullptr, *d_map_raw2 = nullptr;   // the map array (original order) and its ping-pong partner (map updates)
    // map maintenance on the device (mapupd_kernels.h)
    float *d_mu_new = nullptr, *d_mu_boxes = nullptr;
    int *d_mu_flags = nullptr, *d_mu_pos = nullptr, *d_mu_slot = nullptr;
    struct FlBoxSlot *d_mu_tab = nullptr;
    struct FlMapUpdInfo *d_mu_info = nullptr, *h_mu_info = nullptr;
    void *d_mu_tmp = nullptr;
    size_t mu_tmp_bytes = 0;
    int mu_cap = 0, mu_new_cap = 0;
    unsigned mu_tab_cap = 0;
    float4 *d_map_pts = nullptr;
    unsigned *d_map_slot = nullptr, *d_map_rank = nullptr, *d_map_first = nullptr;   // index build: cell slot / rank in cell per point, first point per slot
    struct FlCellEntry *d_map_htab = nullptr;
    unsigned long long *d_map_ckeys = nullptr;
    void *d_map_scan_tmp = nullptr;
    size_t map_scan_bytes = 0;
    int map_cap = 0, map_n = 0, map_max_ring = 0;
    unsigned map_hcap = 0, map_hslots = 0;   // allocated / used slots of the cell table (power of two >= 2 x points)
    float map_cell = 0.f;
    // the map updated in place (mapinc_kernels.h, round 5): per-slot capacity / dirty flag / queue head, per-new-point queue link, per-index
    // dead flag, device control block + its page-locked status copy (read lazily: before the NEXT use of the map)
    unsigned *d_mi_cellcap = nullptr, *d_mi_dirty = nullptr, *d_mi_pend_head = nullptr, *d_mi_pend_next = nullptr;
    unsigned char *d_mi_dead = nullptr;
    FlMapIncCtl *d_mi_ctl = nullptr, *h_mi_status = nullptr, *d_mi_status = nullptr;
    unsigned long long mi_seq = 0, mi_seen = 0;      // status blocks requested / read
    int mi_pend_cap = 0;
    int mi_live = 0;                       // live points as of the last status read (map_n is the RAW length: live + dead)
    bool map_has_dead = false;             // the raw array holds dead entries (compacted by the next full rebuild)
    bool map_index_stale = false;          // the last status said needs_rebuild
    int opt_map_incr = 1;                  // FL_OPT_MAP_INCREMENTAL
    int opt_vio_spec = 2;                  // FL_OPT_VIO_SPECULATE (2: + all pyramid levels of ComputeJ in one launch)
    int opt_vio_wide = 1;                  // FL_OPT_VIO_WIDE
    int opt_detect_fused = 1;              // FL_OPT_DETECT_FUSED
    int front_unsorted_left = 0;           // fl_lidar_front: frames left on the general undistortion kernels (api_front.inc)
    struct FlDetectParams *d_det_prm = nullptr;    // fl_vio_detect, fused form (api_vmap.inc)
    unsigned *d_det_ticket = nullptr;
    size_t map_pool_cap = 0;               // float4 entries of d_map_pts
    bool map_cell_auto = false;            // cell size follows the map's density (cell_size <= 0 at fl_map_set_points / fl_map_clear)
    unsigned *d_map_occ = nullptr, *h_map_occ = nullptr;    // occupied slots among the sampled ones (device counter, pinned copy)
    unsigned map_occ_sample = 0, map_occ_slots = 0;          // of the build the pinned copy belongs to
    int map_occ_n = 0;
    // scan voxel filter (voxel_kernels.h)
    float4 *d_vox_in = nullptr, *d_vox_out = nullptr;
    unsigned *d_vox_keys = nullptr, *d_vox_keys_s = nullptr, *d_vox_vals = nullptr, *d_vox_vals_s = nullptr;
    unsigned *d_vox_heads = nullptr, *d_vox_slot = nullptr;
    FlVoxCtl *d_vox_ctl = nullptr;
    void *d_vox_tmp = nullptr;
    size_t vox_tmp_bytes = 0;
    int vox_cap = 0;
    int vox_resident = 0;          // points left in d_vox_in by fl_imu_undistort
    // the sort-free path (voxel_kernels.h, round 5): occupancy bitmap over the grid's cells + counters; zeroed at allocation, left clean by every run
    unsigned *d_vx_bits = nullptr, *d_vx_l1pre = nullptr, *d_vx_l2flag = nullptr, *d_vx_l2tot = nullptr, *d_vx_cnt = nullptr, *d_vx_ordered = nullptr, *d_vx_slots = nullptr, *d_vx_ohead = nullptr;
    FlVxPartial *d_vx_partial = nullptr;       // workgroup bounding boxes (vx_minmax_kernel / undistort_apply_kernel)
    FlVxCtl *d_vx_ctl = nullptr;   // two blocks: a run uses [vx_parity] and leaves [1 - vx_parity] zeroed for the next one
    int vx_parity = 0;
    long long vx_cells_cap = 0;
    int opt_voxel_sort = 0;        // FL_OPT_VOXEL_SORT
    // fl_lidar_front: while its launches are enqueued the scan's size lives on the device only (the voxel filter's count); h->n is then
    // the CAPACITY the grids are sized for and the kernels read the size through this pointer (nullptr everywhere else)
    const int *n_dev = nullptr;
    bool front_refused = false;    // a pass launch of fl_lidar_front lost its multi-pass admission (launch_lio_passes)
    bool imu_busy = false;         // h_imu (page-locked) may still be read by a copy command of the last fl_lidar_front
    // IMU propagation / undistortion (imu_kernels.h)
    FlImuDev *d_imu = nullptr, *h_imu = nullptr;
    FlImuDev *d_himu = nullptr;                // h_imu (page-locked) as the device addresses it (fl_lidar_front: the kernel fetches it)
    FlImuSample *h_imu_samples = nullptr, *d_himu_samples = nullptr;    // page-locked staging of v_imu for the same purpose
    FlImuSample *d_imu_samples = nullptr;
    FlPose6 *d_imu_poses = nullptr;
    int *d_imu_head = nullptr, *d_imu_blockmin = nullptr;
    int imu_cap_samples = 0, imu_cap_points = 0;
    // VIO patch selection (select_kernels.h)
    std::vector<uint8_t *> kf_ptrs;            // device copies of the reference images (Feature::img), by keyframe id
    uint8_t **d_kf_table = nullptr;
    int *d_sel_owner = nullptr;             // Warp_map: first depth-passing candidate per reference keyframe (select_kernels.h)
    int kf_table_cap = 0;
    bool kf_table_dirty = true;
    unsigned long long *d_depth64 = nullptr;
    struct FlPatchCandidate *d_sel_cand = nullptr;
    struct FlSelectParams *d_sel_prm = nullptr;
    float *d_sel_patches = nullptr, *d_sel_errors = nullptr, *d_sel_acc_err = nullptr, *d_sel_scan = nullptr;
    int32_t *d_sel_slevel = nullptr, *d_sel_reason = nullptr, *d_sel_slot = nullptr, *d_sel_count = nullptr, *d_sel_acc_idx = nullptr,
            *d_sel_acc_lvl = nullptr;
    int sel_cap = 0, sel_scan_cap = 0;
    double *d_grid_pos = nullptr;
    float *d_grid_val_in = nullptr;
    unsigned long long *d_grid_key = nullptr;
    int *d_grid_val = nullptr;
    int32_t *d_grid_num = nullptr;
    void *d_grid_prm = nullptr;
    int grid_cap_pts = 0, grid_cap_cells = 0;
    // native exchange of the sharded form (api_comm.inc)
    void *comm = nullptr;          // ncclComm_t
    int comm_world = 0, comm_rank = 0;
    // misc
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_frame[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // fl_get_frame_timing
    int ev_frame_n = 0;            // events recorded by the last frame driver: start, then (search end, passes end) per segment, covariance end
    int ev_frame_kind[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // what ended at event i: 1 search + plane fit, 2 passes, 3 covariance update
    bool timing = false;
    bool dbg_knn_stamp = false;   // FL_INSTRUMENT build only
    int dbg_mp_refuse_in = 0, dbg_mp_refuse_n = 0;     // FL_INSTRUMENT build only: reservations nth .. nth + count - 1 from now are refused (fl_debug_mp_refuse)
    float last_ms = 0.f;
    int last_launches = 0;
    std::string err;
};

static thread_local std::string g_err;

#define HIPCHK(h, call)                                                                            \
    do {                                                                                           \
        hipError_t e_ = (call);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            char b_[512];                                                                          \
            snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            if (h) (h)->err = b_; else g_err = b_;                                                 \
            return FL_ERR_HIP;                                                                     \
        }                                                                                          \
    } while (0)

static int32_t fail_arg(fl_handle h, const char *msg)
{
    if (h) h->err = msg; else g_err = msg;
    return FL_ERR_ARG;
}

// lidar_selection.cpp:35-59 (set_extrinsic + init): constants of the photometric Jacobian chain.
static void build_vio_const(const fl_config &c, FlVioConst &v)
{
    auto mul = [](const double *A, const double *B, double *C) {
        double T[9];
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) T[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
        memcpy(C, T, sizeof T);
    };
    auto mv = [](const double *A, const double *x, double *o) {
        double t[3];
        for (int i = 0; i < 3; i++) t[i] = A[i * 3] * x[0] + A[i * 3 + 1] * x[1] + A[i * 3 + 2] * x[2];
        memcpy(o, t, sizeof t);
    };
    double Rli[9], Pli[3], t[3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) Rli[i * 3 + j] = c.R_LI[j * 3 + i];
    mv(Rli, c.t_LI, t);
    for (int i = 0; i < 3; i++) Pli[i] = -t[i];
    mul(c.Rcl, Rli, v.Rci);
    mv(c.Rcl, Pli, v.Pci);
    for (int i = 0; i < 3; i++) v.Pci[i] += c.Pcl[i];
    memcpy(v.Jdphi_dR, v.Rci, sizeof v.Rci);
    double nRt[9], Pic[3], K[9], nR[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) nRt[i * 3 + j] = -v.Rci[j * 3 + i];
    mv(nRt, v.Pci, Pic);
    K[0] = 0; K[1] = -Pic[2]; K[2] = Pic[1]; K[3] = Pic[2]; K[4] = 0; K[5] = -Pic[0]; K[6] = -Pic[1]; K[7] = Pic[0]; K[8] = 0;
    for (int i = 0; i < 9; i++) nR[i] = -v.Rci[i];
    mul(nR, K, v.Jdp_dR);
    v.fx_abs = fabs(c.fx);
    v.fy_abs = fabs(4.0 * c.fx * c.fy) / (4. * v.fx_abs);
    v.fx = c.fx; v.fy = c.fy; v.cx = c.cx; v.cy = c.cy;
    for (int i = 0; i < 5; i++) v.d[i] = c.d[i];
    v.width = c.img_width; v.height = c.img_height; v.stride = c.img_width;
    v.distort = (fabs(c.d[0]) > 0.0000001) ? 1 : 0;
}

// ---- co-residency of the multi-pass kernels, made explicit --------------------------------------------------------------
// A multi-pass kernel waits for its own other workgroups (handoff.h), so ALL of them must be resident. What the device can
// hold is occupancy x CUs workgroups (hipOccupancyMaxActiveBlocksPerMultiprocessor, queried per handle); what is already
// on it are the multi-pass launches of OTHER streams of this process that have not completed (launches of one stream run
// one after the other and do not compete). Every multi-pass launch ends by writing its sequence number to a pinned host
// word of its handle; a launch is admitted only while in-flight workgroups + its own fit, otherwise the passes go down the one-launch-per-pass path, whose
// kernels never wait for anything that is not already running. Other processes / foreign kernels are invisible to this
// check: for them the bounded waits end in an ABANDONED pass (solve18.h) and the drivers resume the frame per pass
// (resume_after_timeout below) -- the caller sees neither.
static std::mutex g_mp_mu;
static std::vector<fl_context *> g_mp_handles;      // live handles of this process
// in flight = launched and its completion word not written yet (one store by the launch's last action; no events, no
// extra commands in the stream -- an event per launch cost ~0.5 us per pass of GPU time)
static int mp_busy_locked(fl_handle h)
{
    int busy = 0;
    for (fl_context *o : g_mp_handles) {
        if (o == h || o->cfg.device != h->cfg.device || o->stream == h->stream) continue;
        if (o->mp_seq != __atomic_load_n(o->h_mp_done, __ATOMIC_RELAXED)) busy += o->mp_last_grid;
    }
    return busy;
}
// non-binding look (the frame drivers choose their launch plan with it; the launches themselves reserve)
static bool mp_would_admit(fl_handle h, int slots, int capacity)
{
    std::lock_guard<std::mutex> lk(g_mp_mu);
    return mp_busy_locked(h) + slots <= capacity;
}
// Check AND reserve in one critical section (two host threads on different handles could both pass a separate check and
// oversubscribe the device): on success the handle counts as in flight with `slots` workgroup slots from this moment, and the returned
// sequence number is the one the launch carries and writes to h_mp_done when it ends. 0 = refused (the caller launches per pass).
// need_idle: only if nothing else of the process is on the device (the whole-CU VIO variant).
static unsigned mp_reserve(fl_handle h, int slots, int capacity, bool need_idle = false)
{
    std::lock_guard<std::mutex> lk(g_mp_mu);
    const int busy = mp_busy_locked(h);
    FL_INSTR(if (h->dbg_mp_refuse_in > 1) h->dbg_mp_refuse_in--;                                              // test aid: as if another handle's launch were in flight
             else if (h->dbg_mp_refuse_in == 1) { if (--h->dbg_mp_refuse_n <= 0) h->dbg_mp_refuse_in = 0; h->mp_fallbacks++; return 0u; })
    if (need_idle ? (busy != 0 || slots > capacity) : (busy + slots > capacity)) {
        if (!need_idle) h->mp_fallbacks++;
        return 0u;
    }
    h->mp_last_grid = slots;
    if (++h->mp_seq == 0u) ++h->mp_seq;          // (0 is "refused")
    return h->mp_seq;
}
// FL_NUM_TIMEOUT handling: clears the abandoned mark so that the enqueued per-pass chain runs (solve18.h, fl_pass_skipped)
__global__ void eskf18_resume_kernel(FlDev18 *__restrict__ D)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) { D->status &= ~FL_NUM_TIMEOUT; D->resume_count = 0; D->pub_flag = nullptr; }    // (a resumed frame is read back with a copy)
}
#ifdef FL_INSTRUMENT
// debug / test aid: occupies `blocks` workgroup slots for ~`usec` microseconds (tests/test_coresidency_gpu.py)
__global__ __launch_bounds__(256) void fl_hog_kernel(long long ticks, int *sink)
{
    extern __shared__ int s_hog[];                 // dynamic LDS: what keeps other workgroups off the CU
    const long long t0 = (long long)wall_clock64();
    int v = 0;
    s_hog[threadIdx.x] = 0;
    while ((long long)wall_clock64() - t0 < ticks) { v++; __builtin_amdgcn_s_sleep(8); }
    if (v == -1) *sink = s_hog[0];
}
#endif

extern "C" {

int32_t fl_create(const fl_config *cfg, fl_handle *out)
{
    if (!cfg || !out) return fail_arg(nullptr, "fl_create: null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_err = "fl_create: no HIP device visible (this library has no CPU path)";
        return FL_ERR_NODEVICE;
    }
    if (cfg->device < 0 || cfg->device >= ndev) return fail_arg(nullptr, "fl_create: bad device ordinal");
    if (cfg->patch_size != 8) return fail_arg(nullptr, "fl_create: patch_size must be 8");
    fl_context *h = new fl_context();
    h->cfg = *cfg;
    HIPCHK(h, hipSetDevice(cfg->device));
    HIPCHK(h, hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->stream = h->own_stream;
    // + FL_DEV18_TAIL bytes behind the block for per-call results that travel back with it in the same copy (fl_vio_compute_j's
    // three FlVioLevelInfo)
    HIPCHK(h, hipMalloc(&h->d_dev, sizeof(FlDev18) + FL_DEV18_TAIL));
    HIPCHK(h, hipHostMalloc(&h->h_dev, sizeof(FlDev18) + FL_DEV18_TAIL));
    HIPCHK(h, hipHostMalloc(&h->h_pub, 64));
    memset(h->h_pub, 0, 64);
    HIPCHK(h, hipHostGetDevicePointer((void **)&h->d_pub, h->h_pub, 0));
    HIPCHK(h, hipHostGetDevicePointer(&h->d_hdev, h->h_dev, 0));
    HIPCHK(h, hipMalloc(&h->d_dev23, sizeof(FlDev23)));
    HIPCHK(h, hipHostMalloc(&h->h_dev23, sizeof(FlDev23)));
    HIPCHK(h, hipHostGetDevicePointer(&h->d_hdev23, h->h_dev23, 0));
    HIPCHK(h, hipHostMalloc(&h->h_small, 4096 + FL_UP_SLOTS * 1024));
    HIPCHK(h, hipHostGetDevicePointer(&h->d_small, h->h_small, 0));      // (kernels that fetch a parameter block themselves: fl_vio_detect)
    HIPCHK(h, hipMalloc(&h->d_records, (size_t)16 * FL_MAX_BLOCKS * FL_SUMS23));
    HIPCHK(h, hipMalloc(&h->d_epoch, 64));
    {
        hipDeviceProp_t prop;
        HIPCHK(h, hipGetDeviceProperties(&prop, h->cfg.device));
        h->num_cus = prop.multiProcessorCount;
    }
    HIPCHK(h, hipMalloc(&h->d_sums_tmp, sizeof(double) * FL_SUMS23));
    HIPCHK(h, hipMalloc(&h->d_bcast, sizeof(unsigned long long) * FL_BCAST_STRIDE * 16));     // up to 16 copies of the words (handoff.h)
    HIPCHK(h, hipMemset(h->d_bcast, 0, sizeof(unsigned long long) * FL_BCAST_STRIDE * 16));
    HIPCHK(h, hipMalloc(&h->d_vc, sizeof(FlVioConst)));
    HIPCHK(h, hipMemset(h->d_records, 0, (size_t)16 * FL_MAX_BLOCKS * FL_SUMS23));
    {
        const unsigned one = 1u;   // epoch 0 is the "never written" tag
        HIPCHK(h, hipMemset(h->d_epoch, 0, 64));
        HIPCHK(h, hipMemcpy(h->d_epoch, &one, sizeof one, hipMemcpyHostToDevice));
    }
    HIPCHK(h, hipMemset(h->d_dev, 0, sizeof(FlDev18) + FL_DEV18_TAIL));
    HIPCHK(h, hipMemset(h->d_dev23, 0, sizeof(FlDev23)));
    build_vio_const(h->cfg, h->h_vc);
    HIPCHK(h, hipMemcpy(h->d_vc, &h->h_vc, sizeof(FlVioConst), hipMemcpyHostToDevice));
    HIPCHK(h, hipEventCreate(&h->ev0));
    HIPCHK(h, hipEventCreate(&h->ev1));
    HIPCHK(h, hipHostMalloc((void **)&h->h_mp_done, 64, hipHostMallocMapped));
    *h->h_mp_done = 0u;
    HIPCHK(h, hipHostGetDevicePointer((void **)&h->d_mp_done, h->h_mp_done, 0));
    {
        int b_lio = 0, b_vio = 0, b_ik = 0;
        HIPCHK(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&b_lio, lio18_multipass_kernel, FL_LIO_NT, 0));
        HIPCHK(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&b_vio, vio_multipass_kernel<2>, FL_VIO_NT, 0));
        HIPCHK(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&b_ik, ikfom_multipass_kernel, FL_IK_NT, 0));
        const int b = b_lio < b_vio ? b_lio : b_vio;
        h->mp_capacity = h->mp_capacity_q = b * h->num_cus;
        h->mp_capacity_ik = h->mp_capacity_ik_q = b_ik * h->num_cus;
    }
    { std::lock_guard<std::mutex> lk(g_mp_mu); g_mp_handles.push_back(h); }
    *out = h;
    return FL_OK;
}

static void vox_free(fl_handle h);
static void vx_free_cells(fl_handle h);
static int32_t map_free(fl_handle h);
static void mapupd_free(fl_handle h);
static void imu_free(fl_handle h);
static void select_free(fl_handle h);
static void vmap_free(fl_handle h);
extern "C" int32_t fl_comm_destroy(fl_handle h);
extern "C" int32_t fl_p2p_disconnect(fl_handle h);

int32_t fl_destroy(fl_handle h)
{
    if (!h) return FL_OK;
    hipSetDevice(h->cfg.device);
    hipStreamSynchronize(h->stream);
    hipFree(h->d_body); hipFree(h->d_nbr); hipFree(h->d_world); hipFree(h->d_valid); hipFree(h->d_sel); hipFree(h->d_gate); hipFree(h->d_knn_ids); hipFree(h->d_flat);
    hipFree(h->d_plane); hipFree(h->d_normvec); hipFree(h->d_dev); hipFree(h->d_dev23); hipFree(h->d_records);
    hipFree(h->d_epoch); hipFree(h->d_sums_tmp); hipFree(h->d_bcast); hipFree(h->d_vc); hipFree(h->d_img); hipFree(h->d_ref);
    hipFree(h->d_errors); hipFree(h->d_err_words); hipFree(h->d_pos); hipFree(h->d_slevel);
    map_free(h);
    mapupd_free(h);
    vox_free(h);
    vx_free_cells(h);
    imu_free(h);
    select_free(h);
    vmap_free(h);
    fl_comm_destroy(h);
    fl_p2p_disconnect(h);
    if (h->h_dev) hipHostFree(h->h_dev);
    if (h->h_pub) hipHostFree(h->h_pub);
    if (h->h_dev23) hipHostFree(h->h_dev23);
    if (h->h_small) hipHostFree(h->h_small);
    {
        std::lock_guard<std::mutex> lk(g_mp_mu);
        for (size_t i = 0; i < g_mp_handles.size(); i++)
            if (g_mp_handles[i] == h) { g_mp_handles[i] = g_mp_handles.back(); g_mp_handles.pop_back(); break; }
    }
    if (h->h_mp_done) hipHostFree(h->h_mp_done);
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    for (int i = 0; i < 8; i++) if (h->ev_frame[i]) hipEventDestroy(h->ev_frame[i]);
    if (h->aux_event) hipEventDestroy(h->aux_event);
    if (h->aux_stream) hipStreamDestroy(h->aux_stream);
    if (h->own_stream) hipStreamDestroy(h->own_stream);
    delete h;
    return FL_OK;
}

const char *fl_last_error_string(fl_handle h) { return h ? h->err.c_str() : g_err.c_str(); }

int32_t fl_set_stream(fl_handle h, void *s)
{
    if (!h) return fail_arg(nullptr, "null handle");
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->stream = (hipStream_t)s;   // exactly the caller's stream; NULL is HIP's default (null) stream
    return FL_OK;
}

int32_t fl_sync(fl_handle h)
{
    if (!h) return fail_arg(nullptr, "null handle");
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->up_pending = 0;
    return FL_OK;
}

int32_t fl_host_alloc(fl_handle h, size_t bytes, void **out)
{
    if (!h || !out || bytes == 0) return fail_arg(h, "fl_host_alloc: bad argument");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipHostMalloc(out, bytes, hipHostMallocDefault));
    void *dev = nullptr;
    if (hipHostGetDevicePointer(&dev, *out, 0) == hipSuccess && dev) {      // (kernels may read it in place: fl_lio_frame18_dev)
        std::lock_guard<std::mutex> lk(h->pinned_mu);
        h->pinned.push_back({*out, bytes, dev});
    }
    return FL_OK;
}

int32_t fl_host_free(fl_handle h, void *p)
{
    if (!h) return fail_arg(nullptr, "null handle");
    if (p) {
        {
            std::lock_guard<std::mutex> lk(h->pinned_mu);
            for (size_t k = 0; k < h->pinned.size(); k++)
                if (h->pinned[k].host == p) { h->pinned.erase(h->pinned.begin() + (long)k); break; }
        }
        HIPCHK(h, hipStreamSynchronize(h->stream));       // (a kernel may still be reading it)
        HIPCHK(h, hipHostFree(p));
    }
    return FL_OK;
}

int32_t fl_set_option(fl_handle h, int32_t option, int32_t value)
{
    if (!h) return fail_arg(nullptr, "null handle");
    HIPCHK(h, hipSetDevice(h->cfg.device));            // (the record memsets below go to this handle's stream)
    std::lock_guard<std::mutex> lk(g_mp_mu);           // mp_reserve of other threads reads opt_multipass / mp_capacity
    switch (option) {
    case FL_OPT_MULTIPASS: h->opt_multipass = value != 0; break;
    case FL_OPT_MAX_PRODUCERS:
        if (value < 0 || value > FL_MAX_BLOCKS - 1) return fail_arg(h, "fl_set_option: FL_OPT_MAX_PRODUCERS out of range");
        if (value != h->opt_max_producers)   // the grid (number of records) changes: stale records must not carry a live tag
            HIPCHK(h, hipMemsetAsync(h->d_records, 0, (size_t)16 * FL_MAX_BLOCKS * FL_SUMS23, h->stream));
        h->opt_max_producers = value; break;
    case FL_OPT_IK_PRODUCERS:
        if (value < 0 || value > FL_MAX_BLOCKS - 1) return fail_arg(h, "fl_set_option: FL_OPT_IK_PRODUCERS out of range");
        if (value != h->opt_ik_producers)
            HIPCHK(h, hipMemsetAsync(h->d_records, 0, (size_t)16 * FL_MAX_BLOCKS * FL_SUMS23, h->stream));
        h->opt_ik_producers = value; break;
    case FL_OPT_MP_CAPACITY:
        if (value < 0) return fail_arg(h, "fl_set_option: FL_OPT_MP_CAPACITY out of range");
        if (value > 0) h->mp_capacity = h->mp_capacity_ik = value;
        else { h->mp_capacity = h->mp_capacity_q; h->mp_capacity_ik = h->mp_capacity_ik_q; }
        break;
    case FL_OPT_VIO_WHOLE_CU: h->opt_vio_whole_cu = value != 0; break;
    case FL_OPT_MAILBOX: h->opt_mailbox = value & 3; break;
    case FL_OPT_SCAN_PULL: h->opt_scan_pull = value != 0; break;
    case FL_OPT_INCR_SEARCH: h->opt_incr_search = value != 0; break;
    case FL_OPT_VOXEL_SORT: h->opt_voxel_sort = value != 0; break;
    case FL_OPT_MAP_INCREMENTAL: h->opt_map_incr = value != 0; break;
    case FL_OPT_VIO_SPECULATE: h->opt_vio_spec = value < 0 ? 0 : (value > 2 ? 2 : (int)value); break;
    case FL_OPT_DETECT_FUSED: h->opt_detect_fused = value != 0; break;
    case FL_OPT_VIO_WIDE:
        if (value < 0 || value > 2) return fail_arg(h, "fl_set_option: FL_OPT_VIO_WIDE out of range");
        if (value != h->opt_vio_wide)         // another grid for the same patches: stale records must not carry a live tag
            HIPCHK(h, hipMemsetAsync(h->d_records, 0, (size_t)16 * FL_MAX_BLOCKS * FL_SUMS23, h->stream));
        h->opt_vio_wide = value; break;
    case FL_OPT_DEMOTE_AFTER:
        if (value < 0) return fail_arg(h, "fl_set_option: FL_OPT_DEMOTE_AFTER out of range");
        h->opt_demote_after = value; h->mp_consec_timeouts = 0;
        if (value == 0) { h->mp_demoted_left = 0; h->mp_demote_period = 0; }
        break;
    case FL_OPT_DEMOTE_CALLS:
        if (value < 1 || value > 1024) return fail_arg(h, "fl_set_option: FL_OPT_DEMOTE_CALLS out of range");
        h->opt_demote_calls = value; break;
    default: return fail_arg(h, "fl_set_option: unknown option");
    }
    return FL_OK;
}

int32_t fl_abi_revision(void) { return FL_ABI_REVISION; }

int32_t fl_get_diagnostics(fl_handle h, fl_diagnostics *out)
{
    if (!h || !out) return fail_arg(h, "fl_get_diagnostics: null argument");
    out->multipass_fallbacks = h->mp_fallbacks; out->frames_resumed = h->mp_resumes;
    out->multipass_capacity = h->mp_capacity; out->compute_units = h->num_cus;
    out->demotions = h->mp_demotions; out->demoted_calls_left = h->mp_demoted_left;
    return FL_OK;
}

int32_t fl_set_timing(fl_handle h, int32_t enable)
{
    if (!h) return fail_arg(nullptr, "null handle");
    h->timing = enable != 0;
    return FL_OK;
}

// frame drivers under fl_set_timing: one event behind each stage (created on first use; ~0.5 us of GPU time each, only when timing is on)
static void frame_mark(fl_handle h, int kind)
{
    if (!h->timing) return;
    if (kind == 0) h->ev_frame_n = 0;
    if (h->ev_frame_n >= 8) return;
    hipEvent_t &e = h->ev_frame[h->ev_frame_n];
    if (!e && hipEventCreate(&e) != hipSuccess) return;
    if (hipEventRecord(e, h->stream) != hipSuccess) return;
    h->ev_frame_kind[h->ev_frame_n++] = kind;
}

int32_t fl_get_frame_timing(fl_handle h, fl_frame_timing *out)
{
    if (!h || !out) return fail_arg(h, "fl_get_frame_timing: null argument");
    memset(out, 0, sizeof *out);
    if (h->ev_frame_n < 2) return fail_arg(h, "fl_get_frame_timing: no frame was timed (fl_set_timing, then fl_lio_frame18_dev)");
    HIPCHK(h, hipEventSynchronize(h->ev_frame[h->ev_frame_n - 1]));
    for (int i = 1; i < h->ev_frame_n; i++) {
        float ms = 0.f;
        HIPCHK(h, hipEventElapsedTime(&ms, h->ev_frame[i - 1], h->ev_frame[i]));
        if (h->ev_frame_kind[i] == 1) { out->match_ms += ms; out->searches++; }
        else out->solve_ms += ms;
        out->total_ms += ms;
    }
    return FL_OK;
}

int32_t fl_get_last_kernel_ms(fl_handle h, float *ms)
{
    if (!h || !ms) return fail_arg(h, "null argument");
    HIPCHK(h, hipEventSynchronize(h->ev1));
    HIPCHK(h, hipEventElapsedTime(&h->last_ms, h->ev0, h->ev1));
    *ms = h->last_ms;
    return FL_OK;
}

// ------------------------------------------------------------------------------------------ LIO
static int32_t ensure_points(fl_handle h, int n)
{
    if (n <= h->cap_points) return FL_OK;
    int cap = h->cap_points ? h->cap_points : 4096;
    while (cap < n) cap *= 2;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    hipFree(h->d_body); hipFree(h->d_nbr); hipFree(h->d_world); hipFree(h->d_valid); hipFree(h->d_sel);
    hipFree(h->d_plane); hipFree(h->d_normvec); hipFree(h->d_gate); hipFree(h->d_knn_ids);
    h->d_knn_ids = nullptr; h->knn_ids_valid = false;
    h->d_body = h->d_nbr = h->d_world = nullptr; h->d_valid = h->d_sel = nullptr; h->d_plane = h->d_normvec = nullptr; h->d_gate = nullptr;
    h->cap_points = 0; h->gate_valid = false;
    HIPCHK(h, hipMalloc(&h->d_body, sizeof(float) * 3 * (size_t)cap));
    HIPCHK(h, hipMalloc(&h->d_nbr, sizeof(float) * 15 * (size_t)cap));
    HIPCHK(h, hipMalloc(&h->d_world, sizeof(float) * 3 * (size_t)cap));
    HIPCHK(h, hipMalloc(&h->d_valid, (size_t)cap));
    HIPCHK(h, hipMalloc(&h->d_sel, (size_t)cap));
    HIPCHK(h, hipMalloc(&h->d_gate, sizeof(float4) * (size_t)cap));
    HIPCHK(h, hipMalloc(&h->d_plane, sizeof(float4) * (size_t)cap));
    HIPCHK(h, hipMalloc(&h->d_normvec, sizeof(float4) * (size_t)cap));
    // (d_knn_ids, 80 B per point, is allocated by the first device search: launch_search -- a handle that is fed neighbours by the host,
    // like the 32 M-point passes of bench.py, never pays for it)
    h->cap_points = cap;
    return FL_OK;
}

// producers + 1 solver workgroup (handoff.h). fl_set_option(FL_OPT_MAX_PRODUCERS) caps the producers.
static inline int lio_grid(fl_handle h, int n)
{
    // One gather sweep covers 256 records (handoff.h): keep small/medium scans to a single sweep and
    // let each lane take several points; spread over more workgroups only when the point loop
    // dominates (measured crossover, bench.py --sweep: 200k pts 9.8 us @255 vs 14.4 us @1023;
    // 1M pts 17.5 us @511; 8M pts 75 us @1023).
    // Round 2, multi-pass kernel: a wavefront of the gather pays ~0.1 us per 16-byte load instruction (one per 16 records) while a
    // producer lane's second point costs about as much -- 50 k pts: 7.05 us @196 (one point per lane), 6.72 @160; 65 k: 7.56 @254,
    // 6.8 @96..160; 100 k: 7.64 @255, 7.25..7.32 @128..160; 200 k: 8.05 @255, 8.27 @160.
    return fl_lio_producers(n, h->opt_max_producers) + 1;     // (lio_kernels.h: the device derives the same number from the scan's size, fl_lidar_front)
}

// The records of a pass (handoff.h) validate themselves with a 6-bit tag of the launch epoch, so a record that the PREVIOUS launch
// did not rewrite must not be looked at by the next one: LIO, VIO and Mode-23 passes of one handle share the buffer with different
// grids and record sizes, and a stale tail could carry a matching tag once the epoch has advanced by a multiple of 63 in between.
// Whatever lies beyond the bytes the previous launch covered is zeroed (tag 0 = "never written") before a larger launch reads it.
static inline int vio_grid_h(fl_handle h, int level);
static inline int ik_grid(fl_handle h, int n);
static void *records_for(fl_handle h, size_t need_bytes)
{
    if (need_bytes > h->rec_fresh_bytes)
        (void)hipMemsetAsync((char *)h->d_records + h->rec_fresh_bytes, 0, need_bytes - h->rec_fresh_bytes, h->stream);
    h->rec_fresh_bytes = need_bytes;
    return h->d_records;
}
static inline void *records_lio(fl_handle h) { return records_for(h, (size_t)lio_grid(h, h->n) * FL_SUMS18 * 8); }
static inline void *records_vio(fl_handle h, int level) { return records_for(h, (size_t)vio_grid_h(h, level) * FL_SUMS18 * 8); }
static inline void *records_ik(fl_handle h) { return records_for(h, (size_t)ik_grid(h, h->n) * FL_SUMS23I * 8); }

// everything fl_lio_set_points does except moving the points (and clearing the selection flags, which the search + fit kernel
// writes for every point)
static int32_t stage_points_meta(fl_handle h, int32_t n)
{
    HIPCHK(h, hipSetDevice(h->cfg.device));
    int32_t st = ensure_points(h, n);
    if (st) return st;
    if (n != h->n)   // the grid (number of records) changes: stale records must not carry a live tag
        HIPCHK(h, hipMemsetAsync(h->d_records, 0, (size_t)16 * FL_MAX_BLOCKS * FL_SUMS23, h->stream));
    h->n = n;
    h->have_nbr = false;
    h->normvec_valid = false;
    h->gate_valid = false;
    h->knn_ids_valid = false;          // another scan: the kept winners belong to the old one
    return FL_OK;
}
// the device address of [p, p + bytes) if it lies inside an allocation of fl_host_alloc, else nullptr
static const void *pinned_device_ptr(fl_handle h, const void *p, size_t bytes)
{
    std::lock_guard<std::mutex> lk(h->pinned_mu);
    for (const auto &a : h->pinned)
        if ((const char *)p >= (const char *)a.host && (const char *)p + bytes <= (const char *)a.host + a.bytes)
            return (const char *)a.dev + ((const char *)p - (const char *)a.host);
    return nullptr;
}

int32_t fl_lio_set_points(fl_handle h, const float *body_xyz, int32_t n)
{
    if (!h || !body_xyz || n <= 0) return fail_arg(h, "fl_lio_set_points: bad argument");
    int32_t st = stage_points_meta(h, n);
    if (st) return st;
    HIPCHK(h, hipMemcpyAsync(h->d_body, body_xyz, sizeof(float) * 3 * (size_t)n, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_sel, 0, (size_t)n, h->stream));
    return FL_OK;
}

int32_t fl_lio_set_neighbours(fl_handle h, const float *nbr_xyz, const uint8_t *valid, int32_t n)
{
    if (!h || !nbr_xyz || !valid) return fail_arg(h, "fl_lio_set_neighbours: null argument");
    if (n != h->n || n <= 0) return fail_arg(h, "fl_lio_set_neighbours: n differs from fl_lio_set_points");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipMemcpyAsync(h->d_nbr, nbr_xyz, sizeof(float) * 15 * (size_t)n, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_valid, valid, (size_t)n, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(lio_fit_planes_kernel, dim3((n + FL_BLOCK - 1) / FL_BLOCK), dim3(FL_BLOCK), 0, h->stream,
                       h->d_nbr, h->d_valid, h->d_plane, h->d_sel, n);
    HIPCHK(h, hipGetLastError());
    // the search pass is done: nearest_search_en = false for the passes that follow
    HIPCHK(h, hipMemsetAsync((char *)h->d_dev + offsetof(FlDev18, need_search), 0, sizeof(int32_t), h->stream));
    HIPCHK(h, hipMemsetAsync((char *)h->d_dev23 + offsetof(FlDev23, need_search), 0, sizeof(int32_t), h->stream));
    h->have_nbr = true;
    return FL_OK;
}

int32_t fl_lio_get_selection(fl_handle h, uint8_t *mask, float *normvec)
{
    if (!h || h->n <= 0) return fail_arg(h, "fl_lio_get_selection: no points");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int n = h->n;
    std::vector<float> nv((size_t)n * 4);
    std::vector<uint8_t> sel((size_t)n);
    HIPCHK(h, hipMemcpyAsync(nv.data(), h->d_normvec, sizeof(float) * 4 * (size_t)n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(sel.data(), h->d_sel, (size_t)n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < n; i++) {
        // effective = point_selected_surf && res_last <= 2.0 (laserMapping.cpp:1593)
        // without a FL_ITER_KEEP_NORMVEC pass d_normvec holds nothing: mask = the selection flags, normvec = 0
        if (mask) mask[i] = (uint8_t)(sel[i] && (!h->normvec_valid || (double)fabsf(nv[(size_t)i * 4 + 3]) <= 2.0));
        if (normvec) {
            if (sel[i] && h->normvec_valid) memcpy(normvec + (size_t)i * 4, nv.data() + (size_t)i * 4, sizeof(float) * 4);
            else memset(normvec + (size_t)i * 4, 0, sizeof(float) * 4);
        }
    }
    return FL_OK;
}

int32_t fl_lio_get_world_points(fl_handle h, float *world_xyz)
{
    if (!h || !world_xyz || h->n <= 0) return fail_arg(h, "fl_lio_get_world_points: bad argument");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    hipLaunchKernelGGL(lio_world_points_kernel, dim3((h->n + FL_BLOCK - 1) / FL_BLOCK), dim3(FL_BLOCK), 0, h->stream,
                       h->d_body, h->d_world, h->n, h->d_dev);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(world_xyz, h->d_world, sizeof(float) * 3 * (size_t)h->n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return FL_OK;
}

static void pack_state18(const fl_state18 *s, double *x24)
{
    memcpy(x24, s->rot, sizeof(double) * 9);
    memcpy(x24 + 9, s->pos, sizeof(double) * 3);
    memcpy(x24 + 12, s->vel, sizeof(double) * 3);
    memcpy(x24 + 15, s->bg, sizeof(double) * 3);
    memcpy(x24 + 18, s->ba, sizeof(double) * 3);
    memcpy(x24 + 21, s->grav, sizeof(double) * 3);
}
static void unpack_state18(const double *x24, const double *P, fl_state18 *s)
{
    memcpy(s->rot, x24, sizeof(double) * 9);
    memcpy(s->pos, x24 + 9, sizeof(double) * 3);
    memcpy(s->vel, x24 + 12, sizeof(double) * 3);
    memcpy(s->bg, x24 + 15, sizeof(double) * 3);
    memcpy(s->ba, x24 + 18, sizeof(double) * 3);
    memcpy(s->grav, x24 + 21, sizeof(double) * 3);
    memcpy(s->cov, P, sizeof(double) * 324);
}

// A small device block into a host variable: copy into the handle's page-locked scratch, wait, copy out. (hipMemcpyAsync straight
// into a stack variable -- pageable memory -- is a staged, synchronous copy of its own: ~20 us instead of ~5.)
static int32_t read_small(fl_handle h, void *dst, const void *d_src, size_t bytes)
{
    if (bytes > 4096) return fail_arg(h, "read_small: block too large");
    HIPCHK(h, hipMemcpyAsync(h->h_small, d_src, bytes, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->up_pending = 0;
    memcpy(dst, h->h_small, bytes);
    return FL_OK;
}

// A small parameter block from a host variable to the device through the page-locked scratch (an asynchronous copy from pageable
// memory is staged synchronously). The slot is reused FL_UP_SLOTS uploads later: every entry point that uploads also ends with a
// synchronisation, and none uploads more than a handful of blocks.
static int32_t upload_small(fl_handle h, void *d_dst, const void *src, size_t bytes)
{
    if (bytes > 1024) return fail_arg(h, "upload_small: block too large");
    // a slot is reused FL_UP_SLOTS uploads later: its asynchronous copy must have left the page-locked scratch by then --
    // a synchronisation the code knows of resets the count, otherwise the ring waits here before it wraps
    if (h->up_pending >= FL_UP_SLOTS) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        h->up_pending = 0;
    }
    h->up_pending++;
    char *slot = (char *)h->h_small + 4096 + (size_t)(h->up_slot++ % FL_UP_SLOTS) * 1024;
    memcpy(slot, src, bytes);
    HIPCHK(h, hipMemcpyAsync(d_dst, slot, bytes, hipMemcpyHostToDevice, h->stream));
    return FL_OK;
}

static int32_t begin18_common(fl_handle h, const fl_state18 *state, const fl_state18 *prop, double meas_cov, bool vio = false, bool prepare_in_search = false,
                              bool publish = false, bool leave_in_mirror = false)
{
    HIPCHK(h, hipSetDevice(h->cfg.device));
    // h_dev is reused: wait only if a copy out of it can still be in flight. (A frame driver's begin follows the previous frame's
    // read-back, and then this wait would only serialise the host with the scan copy just enqueued in front of it.)
    if (h->hdev_busy) HIPCHK(h, hipStreamSynchronize(h->stream));
    FlDev18 *D = h->h_dev;
    memset(D, 0, sizeof(FlDev18));
    pack_state18(state, D->x);
    pack_state18(prop, D->xprop);
    memcpy(D->xold, D->x, sizeof D->x);
    memcpy(D->P, state->cov, sizeof(double) * 324);
    memcpy(D->R_LI, h->cfg.R_LI, sizeof(double) * 9);
    memcpy(D->t_LI, h->cfg.t_LI, sizeof(double) * 3);
    D->meas_cov = meas_cov;
    D->iterCount = -1;
    D->rematch_num = 0;
    D->need_search = 1;
    h->search_certain = true;          // the next conditional search runs (launch_search)
    D->searched_at = -1;
    D->stop = 0;
    D->max_iter = h->cfg.max_iterations;
    D->last_error = 1e10f;
    D->last_exact = 1e10f;
    D->last_exact_valid = 1;
    D->err_words = h->d_err_words;
    D->err_cap = h->cap_patches;
    for (int r = 0; r < 8; r++) D->xchg_peer[r] = h->xchg_peer[r];
    D->xchg_epoch = h->d_xepoch;
    D->xchg_rank = h->xchg_rank;
    D->xchg_world = h->xchg_world;
    // frame drivers: the frame's last kernel publishes the block (read_info18 polls). Not for a sharded handle: its chain waits for
    // peers (seconds under FL_XCHG_SPIN_LIMIT if one is late) and the calling thread would spin on the word all that time
    if (publish && (h->opt_mailbox & (vio ? 1 : 2)) && h->xchg_world <= 1) {
        D->pub_flag = h->d_pub; D->pub_dst = h->d_hdev; D->pub_seq = ++h->pub_seq;
    }
    // leave_in_mirror (fl_lio_frame18_dev, prepare_in_search): no copy command -- the frame's first search kernel fetches the block
    // from the mirror (knn_kernels.h host_state); it arrives as that search would leave it (searched_at = iters_run = 0)
    // (vio && leave_in_mirror: fl_vio_compute_j -- vio_prepare_kernel fetches the block)
    const bool vio_pull = vio && leave_in_mirror && h->d_hdev && h->opt_scan_pull;
    if (leave_in_mirror && prepare_in_search) D->searched_at = 0;
    else if (!vio_pull) HIPCHK(h, hipMemcpyAsync(h->d_dev, D, sizeof(FlDev18), hipMemcpyHostToDevice, h->stream));
    h->hdev_busy = true;
    if (vio) hipLaunchKernelGGL(vio_prepare_kernel, dim3(1), dim3(128), 0, h->stream, h->d_dev, (const FlVioConst *)h->d_vc,
                                vio_pull ? (const FlDev18 *)h->d_hdev : (const FlDev18 *)nullptr);      // + the camera pose
    else if (!prepare_in_search) hipLaunchKernelGGL(eskf18_prepare_kernel, dim3(1), dim3(128), 0, h->stream, h->d_dev);
    // (prepare_in_search: fl_lio_frame18_dev -- the first search launch of the frame carries the prepare workgroup, api_knn.inc)
    HIPCHK(h, hipGetLastError());
    h->begun18 = true;
    h->last_state_mode = 18;
    return FL_OK;
}

int32_t fl_lio_begin18(fl_handle h, const fl_state18 *state, const fl_state18 *prop)
{
    if (!h || !state || !prop) return fail_arg(h, "fl_lio_begin18: null argument");
    int32_t st = begin18_common(h, state, prop, h->cfg.laser_point_cov);
    if (st) return st;
    if (h->have_nbr) {  // neighbours staged before begin (benchmark order): the search pass is done
        HIPCHK(h, hipMemsetAsync((char *)h->d_dev + offsetof(FlDev18, need_search), 0, sizeof(int32_t), h->stream));
        h->search_certain = false;     // ... and the device will skip the next conditional search (launch_search)
    }
    return FL_OK;
}

// The frame's last kernel publishes the state block into the pinned mirror and raises h_pub (fl_publish_state): wait for it. False:
// the stream ran dry without the word (abandoned chain: nothing behind the abandoned pass ran) -- the caller reads the block back.
static bool wait_published(fl_handle h, unsigned long long seq)
{
    // Spin budget: a frame is 0.1-0.3 ms; past 1 ms the chain is slow for a reason (a large scan, another client on the device, a
    // time-out being served) and the thread blocks in the runtime instead of burning a core.
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned n = 1;; n++) {
        if (__atomic_load_n(h->h_pub, __ATOMIC_ACQUIRE) == seq) return true;
        if ((n & 1023u) == 0u && std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(1000)) {
            if (hipStreamSynchronize(h->stream) != hipSuccess) return false;
            return __atomic_load_n(h->h_pub, __ATOMIC_ACQUIRE) == seq;
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
}

static int32_t read_info18(fl_handle h, fl_iter_info *info, bool published = false)
{
    // published: the block was uploaded with a mailbox request (begin18_common) and the chain ends in a kernel that serves it
    if (!(published && h->h_dev->pub_flag && wait_published(h, h->pub_seq))) {
        HIPCHK(h, hipMemcpyAsync(h->h_dev, h->d_dev, sizeof(FlDev18) + FL_DEV18_TAIL, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    // (either way everything enqueued before is complete: the stream is in order and the publishing kernel is its last command)
    h->hdev_busy = false; h->hdev23_busy = false; h->up_pending = 0;
    if (!info) return FL_OK;
    const FlDev18 *D = h->h_dev;
    memset(info, 0, sizeof *info);
    memcpy(info->solution, D->solution, sizeof(double) * 18);
    info->total_residual = D->total_residual;
    info->effct_feat_num = D->neff;
    info->converged = D->converged;
    info->status = D->status;
    info->iterations = D->iters_run;
    info->need_search = D->need_search;
    info->stop = D->stop;
    info->accepted = D->accepted;
    return FL_OK;
}

// `count` passes: one multi-pass launch when the grid is certainly co-resident (<= 256 workgroups, one per CU at most),
// else one launch per pass. fl_set_option(FL_OPT_MULTIPASS, 0) forces the latter (A/B measurements, tests).
// per-point gate thresholds of the staged scan (fl_math.h: fl_gate_threshold): once per scan, before its first pass
static void ensure_gates(fl_handle h)
{
    if (h->gate_valid || h->n <= 0) return;
    hipLaunchKernelGGL(lio_gate_kernel, dim3((h->n + FL_BLOCK - 1) / FL_BLOCK), dim3(FL_BLOCK), 0, h->stream, h->d_body, h->d_gate, h->n);
    h->gate_valid = true;
}
// may this handle use the multi-pass form for a grid of `grid` workgroups right now?
static bool multipass_ok(fl_handle h, int grid, bool mode23 = false)          // a look, not a reservation
{
    return grid <= h->num_cus && h->opt_multipass && h->mp_demoted_left <= 0 && mp_would_admit(h, grid, mode23 ? h->mp_capacity_ik : h->mp_capacity);
}
// the outcome of one synchronous driver call (resume_after_timeout / resume23_after_timeout call this exactly once per call)
static void mp_note_outcome(fl_handle h, bool timed_out)
{
    if (h->mp_demoted_left > 0) {                       // serving a demotion: one call less (time-outs of per-pass launches do not extend it)
        if (--h->mp_demoted_left == 0) h->mp_consec_timeouts = h->opt_demote_after > 0 ? h->opt_demote_after - 1 : 0;   // probation
        return;
    }
    if (!timed_out) { h->mp_consec_timeouts = 0; h->mp_demote_period = 0; return; }
    if (h->opt_demote_after <= 0 || !h->opt_multipass) return;
    if (++h->mp_consec_timeouts >= h->opt_demote_after) {
        h->mp_demote_period = h->mp_demote_period ? (h->mp_demote_period * 2 > 1024 ? 1024 : h->mp_demote_period * 2) : h->opt_demote_calls;
        h->mp_demoted_left = h->mp_demote_period;
        h->mp_demotions++;
    }
}
// the reservation of a multi-pass launch of `grid` workgroups: its sequence number, or 0 (launch per pass)
static unsigned multipass_reserve(fl_handle h, int grid, bool mode23 = false)
{
    if (!(grid <= h->num_cus && h->opt_multipass) || h->mp_demoted_left > 0) return 0u;
    return mp_reserve(h, grid, mode23 ? h->mp_capacity_ik : h->mp_capacity);
}
// returns true if the launch carried `extra` (FL_LIO_DO_COV: the covariance update at the end of the launch), i.e. it was a multi-pass one
static bool launch_lio_passes(fl_handle h, int grid, int count, int flags, bool allow_multi = true, int extra = 0)
{
    if (flags & FL_ITER_KEEP_NORMVEC) h->normvec_valid = true;
    ensure_gates(h);
    if (const unsigned seq = (allow_multi && count > 1) ? multipass_reserve(h, grid) : 0u) {
        hipLaunchKernelGGL(lio18_multipass_kernel, dim3(grid), dim3(FL_LIO_NT), 0, h->stream, h->d_gate, h->d_plane, h->d_sel, h->d_normvec,
                           h->n, h->d_dev, records_lio(h), h->d_epoch, h->d_bcast, (int)count, (int)flags, h->d_mp_done, seq, (int)extra, h->n_dev,
                           (int)h->opt_max_producers);
        return extra != 0;
    }
    // (fl_lidar_front: the per-pass kernels take the scan's size from the host, which does not have it yet -- nothing is launched, the
    // driver reads the size back and finishes the frame with launches that know it)
    if (h->n_dev) { h->front_refused = true; return false; }
    for (int i = 0; i < count; i++)
        hipLaunchKernelGGL(lio18_pass_kernel<0>, dim3(grid), dim3(FL_LIO_NT), 0, h->stream, h->d_gate, h->d_plane, h->d_sel,
                           h->d_normvec, h->n, h->d_dev, records_lio(h), h->d_epoch, (double *)nullptr, (int)flags);
    return false;
}
// A pass of the enqueued chain was abandoned after a hand-off time-out (status bit FL_NUM_TIMEOUT, state untouched, everything
// behind it skipped): clear the mark and run what is left with one launch per pass. `enqueue(remaining)` re-enqueues the tail
// of the caller's chain; up to three attempts, then the bit is surfaced. Not in the sharded form: there the ranks may have
// parted ways (one completed the pass, the other did not) and only the caller can restart them together (fl_p2p_connect).
extern "C++" {
template <class F>
static int32_t resume_after_timeout(fl_handle h, fl_iter_info *li, F enqueue)
{
    mp_note_outcome(h, (li->status & FL_NUM_TIMEOUT) != 0 && h->xchg_world <= 1);
    for (int attempt = 0; attempt < 3 && (li->status & FL_NUM_TIMEOUT) && h->xchg_world <= 1; attempt++) {
        h->mp_resumes++;
        const int remaining = h->h_dev->resume_count;
        hipLaunchKernelGGL(eskf18_resume_kernel, dim3(1), dim3(1), 0, h->stream, h->d_dev);
        int32_t st = enqueue(remaining);
        if (st) return st;
        if ((st = read_info18(h, li))) return st;
    }
    return FL_OK;
}
}  // extern "C++"

int32_t fl_lio_iterate18(fl_handle h, int32_t count, int32_t flags, fl_iter_info *info)
{
    flags &= FL_PUBLIC_ITER_FLAGS;          // internal launch bits (FL_IK_PUBLISH, FL_VIO_DO_COV) are not the caller's to set
    if (!h || count < 0) return fail_arg(h, "fl_lio_iterate18: bad argument");
    if (h->n <= 0 || !h->have_nbr) return fail_arg(h, "fl_lio_iterate18: points/neighbours not staged");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    const int grid = lio_grid(h, h->n);
    if (h->timing) HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    launch_lio_passes(h, grid, count, flags);
    if (h->timing) HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    HIPCHK(h, hipGetLastError());
    h->last_launches = count;
    if (!info) return FL_OK;          // enqueue only: a time-out stays visible in the status of a later read-back
    int32_t st = read_info18(h, info);
    if (st) return st;
    return resume_after_timeout(h, info, [&](int remaining) -> int32_t {
        launch_lio_passes(h, grid, remaining < count ? remaining : count, flags, false);
        HIPCHK(h, hipGetLastError());
        return FL_OK;
    });
}

int32_t fl_lio_get_state18(fl_handle h, fl_state18 *out)
{
    if (!h || !out) return fail_arg(h, "fl_lio_get_state18: null argument");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    int32_t st = read_info18(h, nullptr);
    if (st) return st;
    unpack_state18(h->h_dev->x, h->h_dev->P, out);
    return FL_OK;
}

int32_t fl_lio_finish18(fl_handle h, fl_state18 *out)
{
    if (!h) return fail_arg(h, "fl_lio_finish18: null handle");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    hipLaunchKernelGGL(eskf18_cov_update_kernel, dim3(1), dim3(384), 0, h->stream, h->d_dev);
    HIPCHK(h, hipGetLastError());
    if (out) return fl_lio_get_state18(h, out);
    return FL_OK;
}

int32_t fl_lio_frame18(fl_handle h, fl_state18 *state_io, const float *body_xyz, int32_t n, fl_knn_fn knn, void *knn_ctx,
                       fl_iter_info *info)
{
    if (!h || !state_io || !body_xyz || !knn || n <= 0) return fail_arg(h, "fl_lio_frame18: bad argument");
    int32_t st;
    if ((st = fl_lio_set_points(h, body_xyz, n))) return st;
    if ((st = fl_lio_begin18(h, state_io, state_io))) return st;   // state_propagat = state (:1292)
    std::vector<float> world((size_t)n * 3), nbr((size_t)n * 15);
    std::vector<uint8_t> valid((size_t)n);
    fl_iter_info li;
    memset(&li, 0, sizeof li);
    const int total = h->cfg.max_iterations + 1;       // iterCount = -1 .. max-1
    int acc_status = 0;
    while (li.iterations < total) {
        // nearest_search_en: transform at the current state, search on the host, restage (:1527-1549)
        if ((st = fl_lio_get_world_points(h, world.data()))) return st;
        knn(knn_ctx, world.data(), n, nbr.data(), valid.data());
        if ((st = fl_lio_set_neighbours(h, nbr.data(), valid.data(), n))) return st;
        if ((st = fl_lio_iterate18(h, total - li.iterations, FL_ITER_KEEP_NORMVEC, &li))) return st;
        acc_status |= li.status;
        if (li.stop || !li.need_search) break;
    }
    if ((st = fl_lio_finish18(h, state_io))) return st;
    if (info) { *info = li; info->status = acc_status; }
    return FL_OK;
}

int32_t fl_lio_accumulate18(fl_handle h, double *d_sums, int32_t flags)
{
    flags &= FL_PUBLIC_ITER_FLAGS;          // internal launch bits (FL_IK_PUBLISH, FL_VIO_DO_COV) are not the caller's to set
    if (h && (flags & FL_ITER_KEEP_NORMVEC)) h->normvec_valid = true;
    if (!h || !d_sums) return fail_arg(h, "fl_lio_accumulate18: null argument");
    if (h->n <= 0 || !h->have_nbr) return fail_arg(h, "fl_lio_accumulate18: points/neighbours not staged");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (h->timing) HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    ensure_gates(h);
    hipLaunchKernelGGL(lio18_pass_kernel<1>, dim3(lio_grid(h, h->n)), dim3(FL_LIO_NT), 0, h->stream, h->d_gate, h->d_plane,
                       h->d_sel, h->d_normvec, h->n, h->d_dev, records_lio(h), h->d_epoch, d_sums, (int)flags);
    if (h->timing) HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    HIPCHK(h, hipGetLastError());
    return FL_OK;
}

int32_t fl_lio_solve18(fl_handle h, const double *d_sums, int32_t flags, fl_iter_info *info)
{
    flags &= FL_PUBLIC_ITER_FLAGS;          // internal launch bits (FL_IK_PUBLISH, FL_VIO_DO_COV) are not the caller's to set
    if (!h || !d_sums) return fail_arg(h, "fl_lio_solve18: null argument");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    hipLaunchKernelGGL(eskf18_solve_kernel, dim3(1), dim3(FL_BLOCK), 0, h->stream, h->d_dev, d_sums, 0, (int)flags, (const FlVioConst *)h->d_vc);
    HIPCHK(h, hipGetLastError());
    if (info) return read_info18(h, info);
    return FL_OK;
}

#ifdef FL_INSTRUMENT     /* ---- include/fastlivo_hip_debug.h: present in libfastlivo_hip_debug.so only */
// Debug / test aid: a foreign kernel that occupies `blocks` workgroup slots (256 threads, lds_bytes of LDS each) for ~usec
// microseconds on a stream of its own -- what another process on the same GPU looks like to the multi-pass kernels.
int32_t fl_debug_hog(fl_handle h, int32_t blocks, int32_t lds_bytes, int32_t usec)
{
    if (!h || blocks <= 0 || lds_bytes < 1024 || usec <= 0) return fail_arg(h, "fl_debug_hog: bad argument");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    if (!h->aux_stream) HIPCHK(h, hipStreamCreateWithFlags(&h->aux_stream, hipStreamNonBlocking));
    HIPCHK(h, hipFuncSetAttribute((const void *)fl_hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipLaunchKernelGGL(fl_hog_kernel, dim3(blocks), dim3(256), (size_t)lds_bytes, h->aux_stream, (long long)usec * 100ll /* 100 MHz */,
                       (int *)h->d_epoch + 8);
    HIPCHK(h, hipGetLastError());
    return FL_OK;
}
// Debug / test: init + e[0] + ... + e[n-1] as one chain of float additions, by exact_chain.h's workgroup form (out4[0]), by one lane
// adding one by one (out4[1]), by the wavefront form (out4[2]); out4[3] = chunks in which the workgroup form fell back. e is a host array.
int32_t fl_debug_chain(fl_handle h, const float *e, int32_t n, float init, float *out4)
{
    if (!h || n < 0 || (n > 0 && !e) || !out4) return fail_arg(h, "fl_debug_chain: bad argument");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    float *d = nullptr;
    HIPCHK(h, hipMalloc(&d, sizeof(float) * ((size_t)n + 16)));
    if (n > 0) HIPCHK(h, hipMemcpyAsync(d + 16, e, sizeof(float) * (size_t)n, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(fl_chain_debug_kernel, dim3(1), dim3(256), 0, h->stream, (const float *)(d + 16), (int)n, init, d);
    hipError_t err = hipMemcpyAsync(out4, d, sizeof(float) * 16, hipMemcpyDeviceToHost, h->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(h->stream);
    hipFree(d);
    HIPCHK(h, err);
    return FL_OK;
}

// Debug: phase timestamps (shader clock) of the last pass launched with FL_ITER_STAMP (tools/kstamps.py).
int32_t fl_debug_get_stamps(fl_handle h, long long *out64)
{
    if (!h || !out64) return fail_arg(h, "fl_debug_get_stamps: null argument");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_fl_stamps), sizeof(long long) * 64));
    return FL_OK;
}

int32_t fl_debug_get_wall(fl_handle h, long long *out2048)
{
    if (!h || !out2048) return fail_arg(h, "fl_debug_get_wall: null argument");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipMemcpyFromSymbol(out2048, HIP_SYMBOL(g_fl_wall), sizeof(long long) * 2048));
    return FL_OK;
}

// Debug: per-workgroup wall-clock stamps of the device k-NN search launches (tools/knn_wall.py).
int32_t fl_debug_knn_stamp(fl_handle h, int32_t enable)
{
    if (!h) return fail_arg(nullptr, "null handle");
    h->dbg_knn_stamp = enable != 0;
    return FL_OK;
}

// Fault injection for the abandon / resume machinery (tests/test_resume_gpu.py): producer workgroup 0 of the pass launched
// `passes_ahead` passes from now does not publish its record -- the solver's bounded gather expires and the pass is abandoned.
int32_t fl_debug_drop_record(fl_handle h, int32_t passes_ahead)
{
    if (!h || passes_ahead < 0) return fail_arg(h, "fl_debug_drop_record: bad argument");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    unsigned e = 0u;
    HIPCHK(h, hipMemcpy(&e, h->d_epoch, sizeof e, hipMemcpyDeviceToHost));
    e += (unsigned)passes_ahead;
    HIPCHK(h, hipMemcpyToSymbol(HIP_SYMBOL(g_fl_fault_epoch), &e, sizeof e));
    return FL_OK;
}
int32_t fl_debug_mp_refuse(fl_handle h, int32_t nth, int32_t count)
{
    if (!h || nth < 0 || count < 0) return fail_arg(h, "fl_debug_mp_refuse: bad argument");
    std::lock_guard<std::mutex> lk(g_mp_mu);
    h->dbg_mp_refuse_in = count > 0 ? nth : 0;
    h->dbg_mp_refuse_n = count;
    return FL_OK;
}
int32_t fl_debug_map_pool_limit(fl_handle h, int32_t spare_entries)
{
    if (!h || spare_entries < 0 || !h->d_mi_ctl) return fail_arg(h, "fl_debug_map_pool_limit: bad argument / no map index");
    HIPCHK(h, hipSetDevice(h->cfg.device));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    FlMapIncCtl c;
    HIPCHK(h, hipMemcpy(&c, h->d_mi_ctl, sizeof c, hipMemcpyDeviceToHost));
    c.pool_cap = c.pool_top + (unsigned)spare_entries;
    HIPCHK(h, hipMemcpy(h->d_mi_ctl, &c, sizeof c, hipMemcpyHostToDevice));
    return FL_OK;
}
#endif   /* FL_INSTRUMENT */

#include "api_vio.inc"
#include "api_ikfom.inc"
#include "api_knn.inc"
#include "api_map.inc"
#include "api_voxel.inc"
#include "api_imu.inc"
#include "api_front.inc"
#include "api_select.inc"
#include "api_vmap.inc"
#include "api_comm.inc"
#include "api_p2p.inc"

}  // extern "C"
