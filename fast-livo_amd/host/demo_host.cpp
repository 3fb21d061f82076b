// demo_host.cpp -- exercises the host mirror (fastlivo_shim.hpp) over libfastlivo_hip.so from plain C++.
// Input: a little-endian binary frame written by tests (see tests/test_host_mirror_gpu.py); the kNN
// provider replays neighbour sets recorded by the test (one per search pass), standing in for ikd-Tree.
// Output: the updated StatesGroup / state_ikfom as text, compared by the test with the Python path.
#include "fastlivo_shim.hpp"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace fastlivo_host;

struct Replay {
    std::vector<std::vector<float>> nbr;
    std::vector<std::vector<uint8_t>> valid;
    int next = 0;
};
static void knn_replay(void *ctx, const float *, int32_t n, float *nbr, uint8_t *valid)
{
    Replay *r = (Replay *)ctx;
    int k = r->next < (int)r->nbr.size() ? r->next : (int)r->nbr.size() - 1;
    memcpy(nbr, r->nbr[k].data(), sizeof(float) * 15 * (size_t)n);
    memcpy(valid, r->valid[k].data(), (size_t)n);
    r->next++;
}
template <typename T> static void rd(FILE *f, T *p, size_t n) { if (fread(p, sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); } }

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: demo_host frame.bin\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 2; }
    int32_t n, nsearch, max_iter;
    rd(f, &n, 1); rd(f, &nsearch, 1); rd(f, &max_iter, 1);
    fl_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.device = 0; cfg.max_iterations = max_iter; cfg.img_width = 640; cfg.img_height = 512; cfg.patch_size = 8;
    rd(f, cfg.R_LI, 9); rd(f, cfg.t_LI, 3);
    for (int i = 0; i < 9; i++) cfg.Rcl[i] = (i % 4 == 0);
    cfg.fx = cfg.fy = 400; cfg.cx = 320; cfg.cy = 256;
    cfg.laser_point_cov = 0.001; cfg.img_point_cov = 100;
    StatesGroup state;
    rd(f, state.rot_end.m, 9); rd(f, state.pos_end.v, 3); rd(f, state.vel_end.v, 3); rd(f, state.bias_g.v, 3);
    rd(f, state.bias_a.v, 3); rd(f, state.gravity.v, 3); rd(f, state.cov, 324);
    std::vector<float> body((size_t)n * 3);
    rd(f, body.data(), body.size());
    Replay rp;
    for (int k = 0; k < nsearch; k++) {
        rp.nbr.emplace_back((size_t)n * 15); rp.valid.emplace_back((size_t)n);
        rd(f, rp.nbr.back().data(), (size_t)n * 15); rd(f, rp.valid.back().data(), (size_t)n);
    }
    fclose(f);

    fl_handle h = nullptr;
    int32_t st = fl_create(&cfg, &h);
    if (st) { fprintf(stderr, "fl_create: %d %s\n", st, fl_last_error_string(nullptr)); return 1; }
    LioMode18 lio;
    lio.handle = h; lio.knn = knn_replay; lio.knn_ctx = &rp;
    lio.update(state, body.data(), n);
    printf("status %d iter %d neff %d\n", lio.last_status, lio.iterCount, lio.effct_feat_num);
    for (int i = 0; i < 9; i++) printf("%.17g ", state.rot_end.m[i]);
    for (int i = 0; i < 3; i++) printf("%.17g ", state.pos_end.v[i]);
    printf("\n");
    for (int i = 0; i < 18; i++) printf("%.17g ", state.cov[i * 18 + i]);
    printf("\n");

    // h_share_model surrogate check: S^T S == H^T H and S^T h == H^T z
    rp.next = 0;
    state_ikfom s;
    s.pos = state.pos_end;
    s.offset_T_L_I = V3D{{cfg.t_LI[0], cfg.t_LI[1], cfg.t_LI[2]}};
    // rotation matrix -> quaternion (trace positive for the test frames)
    const double *R = state.rot_end.m;
    const double tr = R[0] + R[4] + R[8], qs = std::sqrt(tr + 1.0) * 2;
    s.rot.w = 0.25 * qs; s.rot.x = (R[7] - R[5]) / qs; s.rot.y = (R[2] - R[6]) / qs; s.rot.z = (R[3] - R[1]) / qs;
    esekfom::dyn_share_datastruct<double> dyn;
    HShareContext hc;
    hc.handle = h; hc.knn = knn_replay; hc.knn_ctx = &rp; hc.n = n;
    h_share_model(s, dyn, hc);
    fl_state23 st23; to_abi(s, st23);
    double HTH[144], HTh[12]; int32_t neff; double tr2;
    fl_h_share_model_sums(h, &st23, HTH, HTh, &neff, &tr2);
    double e1 = 0, e2 = 0, sc = 0;
    for (int a = 0; a < 12; a++) {
        for (int b = 0; b < 12; b++) {
            double v = 0; for (int k = 0; k < 23; k++) v += dyn.h_x(k, a) * dyn.h_x(k, b);
            e1 = std::fmax(e1, std::fabs(v - HTH[a * 12 + b])); sc = std::fmax(sc, std::fabs(HTH[a * 12 + b]));
        }
        double v = 0; for (int k = 0; k < 23; k++) v += dyn.h_x(k, a) * dyn.h(k);
        e2 = std::fmax(e2, std::fabs(v - HTh[a]));
    }
    printf("surrogate rows %d valid %d neff %d relerr_HTH %.3e abserr_HTh %.3e scale %.3e\n", dyn.h_x.rows(), (int)dyn.valid, hc.effct_feat_num,
           e1 / sc, e2, sc);
    fl_destroy(h);
    return 0;
}
