// sync_cost.hip -- host cost of waiting for a small piece of GPU work: hipStreamSynchronize vs spinning on hipEventQuery vs
// spinning on a pinned host word the GPU writes. Build: hipcc --offload-arch=gfx950 -O2 -o sync_cost.bin sync_cost.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void k(volatile unsigned *flag, unsigned v, float *out) { out[threadIdx.x] = (float)v; if (threadIdx.x == 0) { __threadfence_system(); *flag = v; } }
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    hipStream_t s; (void)hipStreamCreate(&s);
    float *d; (void)hipMalloc(&d, 1024);
    unsigned *hflag; (void)hipHostMalloc(&hflag, 4, hipHostMallocMapped); *hflag = 0;
    float *hbuf; (void)hipHostMalloc(&hbuf, 1024);
    hipEvent_t ev; (void)hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    const int R = 2000;
    for (int mode = 0; mode < 3; mode++) {
        double t = 0;
        for (int i = 0; i < R + 100; i++) {
            const double t0 = now();
            hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, s, hflag, (unsigned)(mode * 100000 + i + 1), d);
            (void)hipMemcpyAsync(hbuf, d, 256, hipMemcpyDeviceToHost, s);
            if (mode == 0) (void)hipStreamSynchronize(s);
            else if (mode == 1) { (void)hipEventRecord(ev, s); while (hipEventQuery(ev) == hipErrorNotReady) { } }
            else { while (*(volatile unsigned *)hflag != (unsigned)(mode * 100000 + i + 1)) { } (void)hipStreamSynchronize(s); }
            if (i >= 100) t += now() - t0;
        }
        printf("%s: %.1f us per launch+copy+wait\n", mode == 0 ? "hipStreamSynchronize" : mode == 1 ? "spin on hipEventQuery" : "spin on pinned flag (+sync)", t / R);
    }
    return 0;
}
