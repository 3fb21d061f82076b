// tests/host_emul/hshare_product.cpp -- TEST INFRASTRUCTURE ONLY.
//
// The PRODUCT's 2-argument `void h_share_model(state_ikfom&, esekfom::dyn_share_datastruct<double>&)`
// (fast-livo_amd/host/fastlivo_shim.hpp: device-reduced sums returned as the 23x12 sum-compat surrogate, SURVEY 8b body 2) behind the
// flat C callback shape that the REFERENCE's updater text takes (oracle/ref_eigen/text/ikf_3.inc `ref_h_fn`;
// `ref_ikfom_update_text` = esekf::update_iterated_dyn_share_modified, esekfom.hpp:1619-1928, compiled from the reference's source
// text).  That closes the boundary the way laserMapping.cpp wires it (:1233-1235 registration, :1484 call, esekfom.hpp:1636
// invocation): the unmodified reference updater consumes what the product's callback returns.
//
// Two builds of this one file (tests/test_ikfom_boundary_ref_*.py):
//   libhshare_product.so       linked against libfastlivo_hip.so (the device computes the sums)
//   libhshare_product_emul.so  linked against libflabi_emul.so (tests/host_emul/flabi_emul.cpp: the four entry points the callback
//                              uses, over the host build of the product's per-point arithmetic) -- the CPU suite's form
#include "../../fast-livo_amd/host/fastlivo_shim.hpp"

#include <vector>

using namespace fastlivo_host;

namespace {
std::vector<double> g_hx, g_h;
int g_calls = 0, g_searches = 0, g_invalid = 0;
}

extern "C" {

// what the frame loop fills where the reference fills its globals (handle + k-NN provider + the scan's size after fl_lio_set_points)
void product_hsm_setup(fl_handle h, fl_knn_fn knn, void *knn_ctx, int n)
{
    g_hshare.handle = h; g_hshare.knn = knn; g_hshare.knn_ctx = knn_ctx; g_hshare.n = n;
    g_hshare.last_status = 0; g_hshare.effct_feat_num = 0; g_hshare.total_residual = 0.0;
    g_calls = g_searches = g_invalid = 0;
}

// ref_h_fn: state26 = pos 3, rot 4 (x, y, z, w), offset_R_L_I 4, offset_T_L_I 3, vel 3, bg 3, ba 3, grav 3
void product_hsm_callback(void *ctx, double *state26, int *valid, int *converge, int *rows, const double **h_x, const double **h)
{
    (void)ctx;
    state_ikfom s;
    for (int i = 0; i < 3; i++) {
        s.pos.v[i] = state26[i]; s.offset_T_L_I.v[i] = state26[11 + i]; s.vel.v[i] = state26[14 + i];
        s.bg.v[i] = state26[17 + i]; s.ba.v[i] = state26[20 + i]; s.grav.v[i] = state26[23 + i];
    }
    s.rot = Quat{state26[3], state26[4], state26[5], state26[6]};
    s.offset_R_L_I = Quat{state26[7], state26[8], state26[9], state26[10]};
    esekfom::dyn_share_datastruct<double> d;
    d.valid = *valid != 0;
    d.converge = *converge != 0;
    if (d.converge) g_searches++;
    h_share_model(s, d);                      // the registered callback, by its reference name and signature
    g_calls++;
    if (!d.valid) g_invalid++;
    *valid = d.valid ? 1 : 0;
    const int r = d.valid ? d.h_x.rows() : 0;
    g_hx.assign((size_t)r * 12, 0.0);
    g_h.assign((size_t)r, 0.0);
    for (int i = 0; i < r; i++) {
        for (int j = 0; j < 12; j++) g_hx[(size_t)i * 12 + j] = d.h_x(i, j);
        g_h[(size_t)i] = d.h(i);
    }
    *rows = r;
    *h_x = g_hx.data();
    *h = g_h.data();
}

// calls, searches (converge passes), invalid passes, effct_feat_num of the last pass, accumulated status
void product_hsm_stats(int *out5, double *total_residual)
{
    out5[0] = g_calls; out5[1] = g_searches; out5[2] = g_invalid; out5[3] = g_hshare.effct_feat_num; out5[4] = g_hshare.last_status;
    if (total_residual) *total_residual = g_hshare.total_residual;
}

}  // extern "C"
