// tests/host_emul/flabi_emul.cpp -- TEST INFRASTRUCTURE ONLY (never linked into libfastlivo_hip.so; not a fallback path).
//
// The four C-ABI entry points the product's h_share_model (fast-livo_amd/host/fastlivo_shim.hpp) calls -- fl_lio_set_points,
// fl_lio_set_neighbours, fl_ikfom_world_points, fl_h_share_model_sums (include/fastlivo_hip.h) -- over the HOST build of the
// product's own per-point arithmetic (fl_math.h / fl_ikfom_math.h, the code the kernels run), so the CPU suite can drive the
// product's callback through the reference's updater text without a GPU.  Sums are accumulated sequentially here (the device
// reduces them in a different order: that is why the GPU test exists).
#include "../../include/fastlivo_hip.h"
#include "../../fast-livo_amd/csrc/fl_math.h"
#include "../../fast-livo_amd/csrc/fl_ikfom_math.h"

#include <string.h>
#include <vector>

struct fl_context {
    int n = 0;
    std::vector<float> body, plane;
    std::vector<unsigned char> sel;
    bool have_nbr = false;
};

extern "C" {
int32_t flabi_emul_create(fl_handle *out) { *out = new fl_context; return 0; }
int32_t flabi_emul_destroy(fl_handle h) { delete h; return 0; }

int32_t fl_lio_set_points(fl_handle h, const float *body_xyz, int32_t n)
{
    if (!h || !body_xyz || n <= 0) return -1;
    h->n = n;
    h->body.assign(body_xyz, body_xyz + (size_t)n * 3);
    h->have_nbr = false;
    return 0;
}
// lio_fit_planes_kernel (lio_kernels.h): plane once per staging, selection = valid && plane ok
int32_t fl_lio_set_neighbours(fl_handle h, const float *nbr_xyz, const uint8_t *valid, int32_t n)
{
    if (!h || !nbr_xyz || !valid || n != h->n) return -1;
    h->plane.assign((size_t)n * 4, 0.f);
    h->sel.assign((size_t)n, 0);
    for (int i = 0; i < n; i++) {
        float pl[4];
        const int ok = fl_esti_plane(nbr_xyz + (size_t)i * 15, pl);
        const bool keep = valid[i] && ok && pl[0] == pl[0];
        h->sel[(size_t)i] = keep;
        if (keep) memcpy(&h->plane[(size_t)i * 4], pl, sizeof pl);
    }
    h->have_nbr = true;
    return 0;
}
int32_t fl_ikfom_world_points(fl_handle h, const fl_state23 *s, float *world_xyz)
{
    if (!h || !s || !world_xyz || h->n <= 0) return -1;
    const double *x = (const double *)s;                  // fl_state23 is the 26-double device layout (pack_state23)
    for (int i = 0; i < h->n; i++) { double p_i[3]; fl_world_point23(x, &h->body[(size_t)i * 3], p_i, world_xyz + (size_t)i * 3); }
    return 0;
}
// ikfom_pass_kernel<1> (forced pass at state s): gates, 1x12 rows, sums
int32_t fl_h_share_model_sums(fl_handle h, const fl_state23 *s, double *HTH12, double *HTh12, int32_t *effct_feat_num, double *total_residual)
{
    if (!h || !s || !HTH12 || !HTh12 || h->n <= 0 || !h->have_nbr) return -1;
    const double *x = (const double *)s;
    double sums[FL_SUMS23];
    for (int k = 0; k < FL_SUMS23; k++) sums[k] = 0.0;
    for (int i = 0; i < h->n; i++) {
        if (!h->sel[(size_t)i]) continue;
        const float *pb = &h->body[(size_t)i * 3], *pl = &h->plane[(size_t)i * 4];
        double p_i[3]; float pw[3], pd2; int eff;
        fl_world_point23(x, pb, p_i, pw);
        h->sel[(size_t)i] = (unsigned char)fl_gates_from_pw(pb, pl, pw, &pd2, &eff);
        if (!eff) continue;
        double row[12], z;
        fl_row23(x, pb, p_i, pl, pd2, row, &z);
        fl_accum12(sums, row, z);
        sums[FL_S23_NEFF] += 1.0;
        sums[FL_S23_RES] += (double)fabsf(pd2);
    }
    int k = 0;
    for (int i = 0; i < 12; i++)
        for (int j = i; j < 12; j++) { HTH12[i * 12 + j] = sums[k]; HTH12[j * 12 + i] = sums[k]; k++; }
    for (int i = 0; i < 12; i++) HTh12[i] = sums[FL_S23_HTZ + i];
    if (effct_feat_num) *effct_feat_num = (int32_t)sums[FL_S23_NEFF];
    if (total_residual) *total_residual = sums[FL_S23_RES];
    return 0;
}
}  // extern "C"
