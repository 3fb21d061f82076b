// Is a captured hipGraph a cheaper way to submit a chain of small dependent kernels than N launches? (round 6: the front part of
// fl_vio_detect runs at the host's launch rate, 4 us per launch.) N dependent kernels of ~1 us each, submitted (a) as N launches,
// (b) as one hipGraphLaunch of the captured chain; wall time from the first submission to the stream's completion, and the host
// time of the submission alone.   hipcc --offload-arch=gfx950 -O2 -o tools/graph_bench.bin tools/graph_bench.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void step(int *p, int k) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = p[0] + k; else if (blockIdx.x * 256 + threadIdx.x < 50000) p[1 + blockIdx.x * 256 + threadIdx.x] = k; }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    int *d; hipMalloc(&d, sizeof(int) * 60000); hipMemset(d, 0, sizeof(int) * 60000);
    hipStream_t s; hipStreamCreate(&s);
    for (int N : {4, 8, 14, 20}) {
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int k = 0; k < N; k++) hipLaunchKernelGGL(step, dim3(196), dim3(256), 0, s, d, k);
        hipStreamEndCapture(s, &g);
        hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        std::vector<double> a_sub, a_all, b_sub, b_all;
        for (int rep = 0; rep < 60; rep++) {
            hipStreamSynchronize(s);
            double t0 = now();
            for (int k = 0; k < N; k++) hipLaunchKernelGGL(step, dim3(196), dim3(256), 0, s, d, k);
            double t1 = now(); hipStreamSynchronize(s); double t2 = now();
            a_sub.push_back(t1 - t0); a_all.push_back(t2 - t0);
            t0 = now(); hipGraphLaunch(ge, s); t1 = now(); hipStreamSynchronize(s); t2 = now();
            b_sub.push_back(t1 - t0); b_all.push_back(t2 - t0);
        }
        auto med = [](std::vector<double> &v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        printf("N %2d dependent kernels: launches submit %.1f us, done %.1f us | graph submit %.1f us, done %.1f us\n", N, med(a_sub), med(a_all), med(b_sub), med(b_all));
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
