// gather_bench.hip -- how long does ONE workgroup need to read records that other CUs just wrote
// write-through (sc1)?  Sweep over record count / width; answers whether the solver's gather
// (csrc/handoff.h) is byte-bound or latency-bound.  Build: hipcc --offload-arch=gfx950 -O3 gather_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
typedef unsigned int u2 __attribute__((ext_vector_type(2)));

__global__ void writer(u4 *buf, int words16, unsigned tag)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < words16) {
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)buf, 0, words16 * 16, 0x00020000);
        u4 g; g.x = tag; g.y = i; g.z = i * 3; g.w = tag;
        __builtin_amdgcn_raw_buffer_store_b128(g, rs, i * 16, 0, 16);
    }
}
// one workgroup of NT threads reads words16 16-byte words, LOADS per thread in flight
template <int LOADS, int SC1>
__global__ void reader(const u4 *buf, int words16, long long *t, unsigned *sink)
{
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)buf, 0, words16 * 16, 0x00020000);
    long long t0 = wall_clock64();
    unsigned acc = 0;
    for (int base = 0; base < words16; base += blockDim.x * LOADS) {
        u4 v[LOADS];
#pragma unroll
        for (int j = 0; j < LOADS; j++) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (base + j * blockDim.x + threadIdx.x) * 16, 0, SC1 ? 16 : 0);
#pragma unroll
        for (int j = 0; j < LOADS; j++) acc += v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
    }
    __syncthreads();
    long long t1 = wall_clock64();
    if (threadIdx.x == 0) { t[0] = t0; t[1] = t1; }
    sink[threadIdx.x] = acc;
}
int main()
{
    const int maxw = 1 << 16;
    u4 *buf; long long *t; unsigned *sink;
    hipMalloc(&buf, maxw * 16); hipMalloc(&t, 16); hipMalloc(&sink, 4096);
    for (int nt : {256, 512, 1024}) {
        for (int kb : {4, 12, 25, 50, 100, 200}) {
            int words = kb * 1024 / 16;
            double best[2] = {1e9, 1e9}, sum[2] = {0, 0};
            for (int sc1 = 0; sc1 < 2; sc1++) {
                for (int rep = 0; rep < 20; rep++) {
                    hipLaunchKernelGGL(writer, dim3((words + 31) / 32), dim3(32), 0, 0, buf, words, (unsigned)(rep + 1));
                    hipDeviceSynchronize();
                    if (sc1) hipLaunchKernelGGL((reader<8, 1>), dim3(1), dim3(nt), 0, 0, buf, words, t, sink);
                    else hipLaunchKernelGGL((reader<8, 0>), dim3(1), dim3(nt), 0, 0, buf, words, t, sink);
                    hipDeviceSynchronize();
                    long long h[2]; hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
                    double us = (h[1] - h[0]) * 0.01;
                    if (rep >= 2) { sum[sc1] += us; if (us < best[sc1]) best[sc1] = us; }
                }
            }
            printf("NT %4d  %3d KB : plain %.2f us (min %.2f)   sc1 %.2f us (min %.2f)   => sc1 %.1f GB/s\n", nt, kb, sum[0] / 18, best[0], sum[1] / 18, best[1], kb * 1.024e-3 / (sum[1] / 18 * 1e-6) / 1e3 * 1e3 / 1e3);
        }
    }
    return 0;
}
