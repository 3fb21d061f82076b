// fov_walk.cpp -- CPU test aid: LocalMapDev::lasermap_fov_segment (fast-livo_amd/host/fastlivo_shim.hpp, the host mirror of
// laserMapping.cpp:363-417) driven along a walk read from stdin; prints the window after every step. No device is touched: the
// handle is null, so the deletion call returns an error which this program ignores.
#include "../../fast-livo_amd/host/fastlivo_shim.hpp"

#include <cstdio>
using namespace fastlivo_host;
int main()
{
    LocalMapDev lm;
    int n = 0;
    double cube = 0;
    float det = 0, mov = 0;
    if (scanf("%d %lf %f %f", &n, &cube, &det, &mov) != 4) return 2;
    lm.cube_len = cube; lm.DET_RANGE = det; lm.MOV_THRESHOLD = mov;
    for (int i = 0; i < n; i++) {
        V3D p;
        if (scanf("%lf %lf %lf", &p.v[0], &p.v[1], &p.v[2]) != 3) return 2;
        const int nb = lm.lasermap_fov_segment(p);
        printf("%d %.9g %.9g %.9g %.9g %.9g %.9g\n", nb, lm.vertex_min[0], lm.vertex_min[1], lm.vertex_min[2], lm.vertex_max[0], lm.vertex_max[1],
               lm.vertex_max[2]);
    }
    return 0;
}
