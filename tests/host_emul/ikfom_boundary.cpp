// tests/host_emul/ikfom_boundary.cpp -- TEST INFRASTRUCTURE ONLY (links the CPU oracle).
//
// The IKFoM callback boundary of SURVEY 8b, exercised the way laserMapping.cpp wires it:
//     kf.init_dyn_share(get_f, df_dx, df_dw, h_share_model, NUM_MAX_ITERATIONS, epsi);          laserMapping.cpp:1233-1235
//     kf.update_iterated_dyn_share_modified(LASER_POINT_COV, solve_time);                       laserMapping.cpp:1484
// with h_share_model = the product's 2-argument callback (fast-livo_amd/host/fastlivo_shim.hpp: device-reduced sums returned as
// a 23x12 surrogate) and kf = esekf_mock.hpp, whose updater is the oracle's restatement of esekfom.hpp:1619-1928 and knows
// nothing about the surrogate.  Compared with the product's own whole-update entry point fl_ikfom_update_iterated on the same
// frame and the same k-NN provider.  Prints the differences; tests/test_ikfom_boundary_gpu.py asserts them.
#include "../../fast-livo_amd/host/fastlivo_shim.hpp"
#include "esekf_mock.hpp"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace fastlivo_host;

struct MapKnn { std::vector<float> map; int k = 0; int calls = 0; };
static void knn_bruteforce(void *ctx, const float *world, int32_t n, float *nbr, uint8_t *valid)
{
    MapKnn *m = (MapKnn *)ctx;
    std::vector<float> sq((size_t)n * 5);
    orc_knn5(m->map.data(), m->k, world, n, nbr, sq.data(), valid, nullptr, 8);
    m->calls++;
}
template <typename T> static void rd(FILE *f, T *p, size_t n) { if (fread(p, sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); } }

typedef esekfom::esekf<state_ikfom, 12> esekf_t;

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: ikfom_boundary frame.bin\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 2; }
    int32_t n, k, max_iter;
    rd(f, &n, 1); rd(f, &k, 1); rd(f, &max_iter, 1);
    double R;
    rd(f, &R, 1);
    fl_state23 x0;
    rd(f, (double *)&x0, sizeof(fl_state23) / sizeof(double));
    std::vector<double> P0(23 * 23), limit(23);
    rd(f, P0.data(), P0.size()); rd(f, limit.data(), 23);
    std::vector<float> body((size_t)n * 3);
    rd(f, body.data(), body.size());
    MapKnn mk;
    mk.k = k; mk.map.resize((size_t)k * 3);
    rd(f, mk.map.data(), mk.map.size());
    fclose(f);

    fl_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.device = 0; cfg.max_iterations = max_iter; cfg.img_width = 640; cfg.img_height = 512; cfg.patch_size = 8;
    for (int i = 0; i < 9; i++) cfg.R_LI[i] = cfg.Rcl[i] = (i % 4 == 0);
    cfg.fx = cfg.fy = 400; cfg.cx = 320; cfg.cy = 256;
    cfg.laser_point_cov = R; cfg.img_point_cov = 100;
    fl_handle h = nullptr;
    int32_t st = fl_create(&cfg, &h);
    if (st) { fprintf(stderr, "fl_create: %d %s\n", st, fl_last_error_string(nullptr)); return 1; }

    // ---- A: the reference's wiring.  state_ikfom from the ABI struct
    state_ikfom s0;
    memcpy(s0.pos.v, x0.pos, sizeof x0.pos);
    s0.rot = Quat{x0.rot[0], x0.rot[1], x0.rot[2], x0.rot[3]};
    s0.offset_R_L_I = Quat{x0.offset_R_L_I[0], x0.offset_R_L_I[1], x0.offset_R_L_I[2], x0.offset_R_L_I[3]};
    memcpy(s0.offset_T_L_I.v, x0.offset_T_L_I, sizeof x0.offset_T_L_I);
    memcpy(s0.vel.v, x0.vel, sizeof x0.vel); memcpy(s0.bg.v, x0.bg, sizeof x0.bg);
    memcpy(s0.ba.v, x0.ba, sizeof x0.ba); memcpy(s0.grav.v, x0.grav, sizeof x0.grav);
    esekf_t kf;
    double epsi[23];
    for (int i = 0; i < 23; i++) epsi[i] = limit[i];
    kf.init_dyn_share(nullptr, nullptr, nullptr, h_share_model, max_iter, epsi);      // the 2-argument callback, by name
    esekf_t::cov Pc;
    memcpy(Pc.d, P0.data(), sizeof Pc.d);
    kf.change_x(s0); kf.change_P(Pc);
    // what the frame loop fills where the reference fills its globals
    g_hshare.handle = h; g_hshare.knn = knn_bruteforce; g_hshare.knn_ctx = &mk; g_hshare.n = n;
    if ((st = fl_lio_set_points(h, body.data(), n))) { fprintf(stderr, "set_points %d\n", st); return 1; }
    double solve_time = 0;
    kf.update_iterated_dyn_share_modified(R, solve_time);
    const int calls_a = mk.calls;
    fl_state23 xa;
    to_abi(kf.get_x(), xa);
    const esekf_t::cov Pa = kf.get_P();

    // ---- B: the product's whole-update entry point on the same inputs
    fl_state23 xb = x0;
    std::vector<double> Pb = P0;
    fl_iter_info info;
    mk.calls = 0;
    st = fl_ikfom_update_iterated(h, &xb, Pb.data(), body.data(), n, R, limit.data(), knn_bruteforce, &mk, &info);
    if (st) { fprintf(stderr, "fl_ikfom_update_iterated: %d %s\n", st, fl_last_error_string(h)); return 1; }

    double dx = 0, dp = 0, pscale = 0;
    const double *a = (const double *)&xa, *b = (const double *)&xb;
    for (size_t i = 0; i < sizeof(fl_state23) / sizeof(double); i++) dx = std::fmax(dx, std::fabs(a[i] - b[i]));
    for (int i = 0; i < 529; i++) { dp = std::fmax(dp, std::fabs(Pa.d[i] - Pb[i])); pscale = std::fmax(pscale, std::fabs(Pb[i])); }
    double moved = 0;
    const double *z = (const double *)&x0;
    for (size_t i = 0; i < sizeof(fl_state23) / sizeof(double); i++) moved = std::fmax(moved, std::fabs(b[i] - z[i]));
    printf("iters_callback %d iters_device %d searches_callback %d searches_device %d neff %d status_a %d status_hshare %d status_b %d\n",
           kf.iterations, info.iterations, calls_a, mk.calls, g_hshare.effct_feat_num, kf.last_status, g_hshare.last_status, info.status);
    printf("max_state_diff %.6e max_P_diff %.6e P_scale %.6e moved %.6e\n", dx, dp, pscale, moved);
    for (size_t i = 0; i < sizeof(fl_state23) / sizeof(double); i++) printf("%.17g ", b[i]);
    printf("\n");
    fl_destroy(h);
    return 0;
}
