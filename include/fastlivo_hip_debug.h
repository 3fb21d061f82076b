/*
 * include/fastlivo_hip_debug.h -- measurement and test aids of the INSTRUMENTED build only.
 *
 * libfastlivo_hip_debug.so = the same sources compiled with -DFL_INSTRUMENT (fast-livo_amd/build.sh debug): every entry point of
 * include/fastlivo_hip.h plus the ones below, phase stamps inside the pass kernels (flag FL_ITER_STAMP) and a fault injector.
 * The release library (libfastlivo_hip.so) exports none of these and carries no stamp code. tools/ and the tests of the
 * abandon / resume machinery and of the exact float chain load this build; nothing else does.
 */
#ifndef FASTLIVO_HIP_DEBUG_H
#define FASTLIVO_HIP_DEBUG_H

#include "fastlivo_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Phase stamps (100 MHz wall clock) of the last pass launched with flag FL_ITER_STAMP: 64 slots of workgroup 0 / the solver. */
#define FL_ITER_STAMP 4
int32_t fl_debug_get_stamps(fl_handle h, long long *out64);
/* Per-workgroup start/end wall-clock stamps of the last stamped pass / search launch. */
int32_t fl_debug_get_wall(fl_handle h, long long *out2048);
/* Stamp the device k-NN search launches of this handle (tools/knn_wall.py). */
int32_t fl_debug_knn_stamp(fl_handle h, int32_t enable);
/* A foreign kernel that occupies `blocks` workgroup slots (256 threads + lds_bytes of LDS each) for ~usec microseconds on a stream
 * of its own: what another process on the same GPU looks like to the multi-pass kernels (tests/test_coresidency_gpu.py). */
int32_t fl_debug_hog(fl_handle h, int32_t blocks, int32_t lds_bytes, int32_t usec);
/* init + e[0] + ... + e[n-1] (host array) as ONE chain of float additions, as the reference's `error += patch_error`
 * (lidar_selection.cpp:857) rounds it: out4[0] by the workgroup form the kernels use (csrc/exact_chain.h), out4[1] by one lane adding
 * one by one, out4[2] by the wavefront form (the workgroup form's fallback). They must be the same bits. out4[3] = number of
 * 2048-element chunks in which the workgroup form's checks failed and it fell back; out4[4..15]: shader-clock offsets of the phases
 * of the last chunk (out4 holds 16 floats). */
int32_t fl_debug_chain(fl_handle h, const float *e, int32_t n, float init, float *out4);
/* Fault injection: producer workgroup 0 of the pass launched `passes_ahead` passes from now (0 = the next one) does not publish its
 * record, so that pass's bounded gather expires and the pass is ABANDONED (FL_NUM_TIMEOUT): exercises the resume paths. */
int32_t fl_debug_drop_record(fl_handle h, int32_t passes_ahead);
/* Leaves the in-place map index `spare_entries` pool entries (device-side control block only: the host's pre-check of the next
 * update keeps seeing the old size), so that update runs out of pool, drops its queued points from the index and raises
 * needs_rebuild -- the path every search must notice before it reads the index (tests/test_map_gpu.py). */
int32_t fl_debug_map_pool_limit(fl_handle h, int32_t spare_entries);
/* `count` consecutive multi-pass reservations of this handle, starting with the nth from now (1 = the next one; nth or count 0 disarms), are refused as if
 * another handle's launch were in flight:
 * the caller's per-pass path -- or, for the frame drivers that keep a size on the device (fl_lidar_front, fl_vio_detect), the path that reads
 * the size back and finishes with launches that know it. Every variant the admission code tries counts as one reservation (the VIO launches try
 * the whole-CU variant first, then the shared one: count = 2 refuses a level outright). */
int32_t fl_debug_mp_refuse(fl_handle h, int32_t nth, int32_t count);

#ifdef __cplusplus
}
#endif
#endif
