// demo_pipeline.cpp -- the LiDAR front of a frame through the host mirror, from plain C++ over the C ABI:
// ImuProcessDev::UndistortPcl -> VoxelGridDev::filter_to_scan -> LioMode18Dev::update (map search on the device) ->
// LocalMapDev::lasermap_fov_segment / map_incremental (the map stays on the device).
// Input: little-endian binary written by tests/test_host_mirror_gpu.py; output: the updated StatesGroup as text.
#include "fastlivo_shim.hpp"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace fastlivo_host;
template <typename T> static void rd(FILE *f, T *p, size_t n) { if (fread(p, sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); } }

int main(int argc, char **argv)
{
    if (argc < 2) { fprintf(stderr, "usage: demo_pipeline frame.bin\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 2; }
    int32_t n, n_imu, k_map, max_iter;
    float leaf, cell;
    double beg, end;
    rd(f, &n, 1); rd(f, &n_imu, 1); rd(f, &k_map, 1); rd(f, &max_iter, 1); rd(f, &leaf, 1); rd(f, &cell, 1); rd(f, &beg, 1); rd(f, &end, 1);
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
    ImuProcessDev imu;
    rd(f, &imu.proc, 1);
    std::vector<fl_imu_sample> samples((size_t)n_imu);
    rd(f, samples.data(), samples.size());
    std::vector<float> pts((size_t)n * 4), map((size_t)k_map * 3);
    rd(f, pts.data(), pts.size()); rd(f, map.data(), map.size());
    // optional: camera extrinsics of the state frame (Rci, Pci) and a 640 x 512 grey image
    double Rci[9], Pci[3];
    std::vector<uint8_t> image((size_t)cfg.img_width * cfg.img_height);
    const bool have_cam = fread(Rci, sizeof(double), 9, f) == 9;
    if (have_cam) { rd(f, Pci, 3); rd(f, image.data(), image.size()); }
    fclose(f);

    fl_handle h = nullptr;
    int32_t st = fl_create(&cfg, &h);
    if (st) { fprintf(stderr, "fl_create: %d %s\n", st, fl_last_error_string(nullptr)); return 1; }
    imu.handle = h;
    VoxelGridDev vg; vg.handle = h; vg.setLeafSize(leaf, leaf, leaf);
    LioMode18Dev lio; lio.handle = h;
    if (lio.set_map(map.data(), k_map, cell)) { fprintf(stderr, "set_map: %s\n", fl_last_error_string(h)); return 1; }
    const StatesGroup state0 = state;                  // (for the timed repetitions at the end)
    const fl_imu_proc proc0 = imu.proc;
    imu.UndistortPcl(samples, beg, end, state, pts, /*keep_on_device=*/true);
    vg.setInputCloudOnDevice(n);
    const int feats_down_size = vg.filter_to_scan();
    lio.update(state, nullptr, 0);
    if (imu.last_status < 0 || vg.last_status < 0 || lio.last_status < 0) { fprintf(stderr, "error: %s\n", fl_last_error_string(h)); return 1; }
    printf("status %d iter %d neff %d scan %d\n", lio.last_status, lio.iterCount, lio.effct_feat_num, feats_down_size);
    for (int i = 0; i < 9; i++) printf("%.17g ", state.rot_end.m[i]);
    for (int i = 0; i < 3; i++) printf("%.17g ", state.pos_end.v[i]);
    for (int i = 0; i < 3; i++) printf("%.17g ", state.vel_end.v[i]);
    printf("\n");
    for (int i = 0; i < 18; i++) printf("%.17g ", state.cov[i * 18 + i]);
    printf("\n%.17g %.17g\n", imu.proc.last_lidar_end_time, imu.proc.acc_s_last[2]);
    // (timed HERE, before the map changes below: the figures are for the map the frame was registered against)
    // FL_DEMO_TIME_REPS=N: the LiDAR front (undistortion -> voxel filter -> Mode-18 update, everything between them on the device) repeated
    // N times from the same inputs, host wall time per frame from plain C++ -- what tools/pipeline_bench.py measures through python.
    // Printed on stderr (tests parse stdout).
    if (const char *reps_s = getenv("FL_DEMO_TIME_REPS")) {
        const int reps = atoi(reps_s);
        std::vector<double> ms;
        for (int r = 0; r < reps + 3; r++) {
            StatesGroup x = state0;
            imu.proc = proc0;
            std::vector<float> p = pts;
            const auto t0 = std::chrono::steady_clock::now();
            imu.UndistortPcl(samples, beg, end, x, p, /*keep_on_device=*/true);
            vg.setInputCloudOnDevice(n);
            vg.filter_to_scan();
            lio.update(x, nullptr, 0);
            const auto t1 = std::chrono::steady_clock::now();
            if (imu.last_status < 0 || vg.last_status < 0 || lio.last_status < 0) { fprintf(stderr, "error: %s\n", fl_last_error_string(h)); return 1; }
            if (r >= 3) ms.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
        }
        if (!ms.empty()) {
            std::sort(ms.begin(), ms.end());
            fprintf(stderr, "lidar_front_ms median %.4f min %.4f p90 %.4f (%d raw points, %zu frames, C++ over the C ABI)\n", ms[ms.size() / 2], ms[0],
                    ms[(ms.size() * 9) / 10], n, ms.size());
        }
        // the same frames through fl_lidar_front (LidarFrontDev): one enqueue, one wait -- and the state must come out bit for bit the same
        LidarFrontDev front; front.handle = h; front.imu = &imu; front.filter_size_surf = leaf;
        // the raw scan in page-locked memory of the library (fl_host_alloc): the frame's first launch fetches it itself, no copy command
        float *pinned = nullptr;
        if (fl_host_alloc(h, sizeof(float) * pts.size(), (void **)&pinned) || !pinned) { fprintf(stderr, "fl_host_alloc: %s\n", fl_last_error_string(h)); return 1; }
        memcpy(pinned, pts.data(), sizeof(float) * pts.size());
        std::vector<double> mf, mp;
        StatesGroup x_staged = state0, x_fused = state0;
        {
            imu.proc = proc0;
            std::vector<float> p = pts;
            imu.UndistortPcl(samples, beg, end, x_staged, p, true);
            vg.setInputCloudOnDevice(n); vg.filter_to_scan(); lio.update(x_staged, nullptr, 0);
        }
        for (int r = 0; r < reps + 3; r++) {
            StatesGroup x = state0;
            imu.proc = proc0;
            const auto t0 = std::chrono::steady_clock::now();
            front.update(samples, beg, end, x, pts);
            const auto t1 = std::chrono::steady_clock::now();
            if (front.last_status < 0) { fprintf(stderr, "fl_lidar_front: %s\n", fl_last_error_string(h)); return 1; }
            if (r >= 3) mf.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
            x_fused = x;
        }
        if (!mf.empty()) {
            std::sort(mf.begin(), mf.end());
            const bool same = memcmp(&x_staged, &x_fused, sizeof(StatesGroup)) == 0;
            fprintf(stderr, "lidar_front_fused_ms median %.4f min %.4f p90 %.4f (fl_lidar_front, scan %d, state %s the staged calls')\n", mf[mf.size() / 2],
                    mf[0], mf[(mf.size() * 9) / 10], front.feats_down_size, same ? "bit-identical to" : "DIFFERS from");
        }
        StatesGroup x_pinned = state0;
        for (int r = 0; r < reps + 3; r++) {
            StatesGroup x = state0;
            imu.proc = proc0;
            fl_state18 st18;
            to_abi(x, st18);
            fl_iter_info info;
            int32_t m = 0;
            const auto t0 = std::chrono::steady_clock::now();
            const int32_t rc = fl_lidar_front(h, &imu.proc, &st18, samples.data(), (int)samples.size(), beg, end, pinned, n, leaf, 0, &info, &m);
            const auto t1 = std::chrono::steady_clock::now();
            if (rc < 0) { fprintf(stderr, "fl_lidar_front (pinned): %s\n", fl_last_error_string(h)); return 1; }
            from_abi(st18, x);
            if (r >= 3) mp.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
            x_pinned = x;
        }
        if (!mp.empty()) {
            std::sort(mp.begin(), mp.end());
            const bool same = memcmp(&x_staged, &x_pinned, sizeof(StatesGroup)) == 0;
            fprintf(stderr, "lidar_front_fused_pinned_ms median %.4f min %.4f p90 %.4f (fl_lidar_front, raw scan in fl_host_alloc memory, state %s the staged calls')\n",
                    mp[mp.size() / 2], mp[0], mp[(mp.size() * 9) / 10], same ? "bit-identical to" : "DIFFERS from");
        }
        fl_host_free(h, pinned);
    }
    // the frame's tail: window check at the new position, then the scan goes into the map (laserMapping.cpp:1395,1758)
    LocalMapDev lm; lm.handle = h; lm.cube_len = 24.0; lm.DET_RANGE = 4.0f; lm.downsample_size = leaf;
    V3D pos_lid = state.pos_end;
    lm.lasermap_fov_segment(pos_lid);                 // first call: centres the window
    pos_lid.v[0] += 8.0;                               // pretend the sensor moved: the window follows, one slab is cut off
    const int cut = lm.lasermap_fov_segment(pos_lid);
    const int removed = lm.kdtree_delete_counter;
    lm.map_incremental();
    if (lm.last_status < 0) { fprintf(stderr, "map: %s\n", fl_last_error_string(h)); return 1; }
    printf("map %d boxes %d removed %d before %d after %d added %d\n", lm.last_status, cut, removed, lm.last.n_before, lm.last.n_after,
           lm.last.n_added);
    // optional camera half (LidarSelectorDev::detect), two frames on the same image: the first founds map points, the second tracks them
    if (have_cam) {
        std::vector<float> pg((size_t)feats_down_size * 3), pg_down((size_t)feats_down_size * 4);
        if (fl_lio_get_world_points(h, pg.data())) { fprintf(stderr, "world points: %s\n", fl_last_error_string(h)); return 1; }
        std::vector<float> pg4((size_t)feats_down_size * 4, 0.0f);
        for (int i = 0; i < feats_down_size; i++) for (int k = 0; k < 3; k++) pg4[(size_t)i * 4 + k] = pg[(size_t)i * 3 + k];
        int32_t n_down = 0, small = 0;
        if (fl_scan_voxel_filter(h, pg4.data(), feats_down_size, 0.2f, 0.2f, 0.2f, 0, pg_down.data(), &n_down, &small) < 0) return 1;   // downSizeFilter (:352-353)
        std::vector<float> down3((size_t)n_down * 3);
        for (int i = 0; i < n_down; i++) for (int k = 0; k < 3; k++) down3[(size_t)i * 3 + k] = pg_down[(size_t)i * 4 + k];
        LidarSelectorDev sel; sel.handle = h; sel.grid_size = 40; sel.outlier_threshold = 1e12;
        if (sel.init() < 0) { fprintf(stderr, "vmap: %s\n", fl_last_error_string(h)); return 1; }
        for (int f2 = 0; f2 < 2; f2++) {
            sel.detect(image.data(), cfg.img_width, cfg.img_height, cfg.img_width, pg.data(), feats_down_size, down3.data(), n_down, Rci, Pci, state);
            if (sel.last_status < 0) { fprintf(stderr, "detect: %s\n", fl_last_error_string(h)); return 1; }
            printf("cam %d selected %d founded %d observed %d\n", f2, sel.n_selected, sel.n_founded, sel.n_observed);
        }
        lm.compact();                                    // (exercises fl_map_compact from the host mirror; the map's content does not change)
        if (lm.last_status < 0) { fprintf(stderr, "compact: %s\n", fl_last_error_string(h)); return 1; }
        for (int i = 0; i < 9; i++) printf("%.17g ", state.rot_end.m[i]);
        for (int i = 0; i < 3; i++) printf("%.17g ", state.pos_end.v[i]);
        printf("\n");
        // FL_DEMO_TIME_REPS: the camera half (LidarSelectorDev::detect = the body of LidarSelector::detect, lidar_selection.cpp:1027-1076) again and
        // again on the same image from the same state, host wall time per frame from plain C++ (stderr)
        if (const char *reps_s = getenv("FL_DEMO_TIME_REPS")) {
            const int reps = atoi(reps_s);
            std::vector<double> ms;
            const StatesGroup xc0 = state;
            for (int r = 0; r < reps + 3; r++) {
                StatesGroup xc = xc0;
                const auto t0 = std::chrono::steady_clock::now();
                sel.detect(image.data(), cfg.img_width, cfg.img_height, cfg.img_width, pg.data(), feats_down_size, down3.data(), n_down, Rci, Pci, xc);
                const auto t1 = std::chrono::steady_clock::now();
                if (sel.last_status < 0) { fprintf(stderr, "detect: %s\n", fl_last_error_string(h)); return 1; }
                if (r >= 3) ms.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
            }
            if (!ms.empty()) {
                std::sort(ms.begin(), ms.end());
                fprintf(stderr, "camera_half_ms median %.4f min %.4f p90 %.4f (detect: %d patches tracked, %d scan points, %d down-sampled; C++ over the C ABI)\n",
                        ms[ms.size() / 2], ms[0], ms[(ms.size() * 9) / 10], sel.n_selected, feats_down_size, n_down);
            }
            // the same with the scan left on the device (FL_DETECT_SCAN_ON_DEVICE): pointBodyToWorld + the 0.2 m down-sampling (lidar_selection.cpp:352-353)
            // are part of the timed call now, nothing but the image goes up
            std::vector<double> md;
            int32_t ns = 0, nf = 0, no = 0;
            // (the image in page-locked memory of the library: the frame's first kernel fetches it, no copy command at all)
            uint8_t *img_pinned = nullptr;
            if (fl_host_alloc(h, image.size(), (void **)&img_pinned) || !img_pinned) { fprintf(stderr, "fl_host_alloc: %s\n", fl_last_error_string(h)); return 1; }
            memcpy(img_pinned, image.data(), image.size());
            for (int r = 0; r < reps + 3; r++) {
                StatesGroup xc = xc0;
                fl_state18 st18;
                to_abi(xc, st18);
                const auto t0 = std::chrono::steady_clock::now();
                const int32_t rc = fl_vio_detect(h, img_pinned, cfg.img_width, cfg.img_height, cfg.img_width, nullptr, FL_DETECT_SCAN_ON_DEVICE, nullptr, 0, Rci, Pci,
                                                 &st18, 100 + r, 0, 0.0, 1e12, &ns, &nf, &no);
                const auto t1 = std::chrono::steady_clock::now();
                if (rc < 0) { fprintf(stderr, "detect (device scan): %s\n", fl_last_error_string(h)); return 1; }
                if (r >= 3) md.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
            }
            if (!md.empty()) {
                std::sort(md.begin(), md.end());
                fprintf(stderr, "camera_half_device_scan_ms median %.4f min %.4f p90 %.4f (detect with the scan on the device: world points + 0.2 m voxel filter inside the call, image in fl_host_alloc memory; %d patches tracked)\n",
                        md[md.size() / 2], md[0], md[(md.size() * 9) / 10], ns);
            }
            fl_host_free(h, img_pinned);
        }
    }
    fl_destroy(h);
    return 0;
}
