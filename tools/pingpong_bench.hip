// pingpong_bench.hip -- one-way latency of the sc1 "data is the flag" hand-off (csrc/handoff.h) between two workgroups that
// sit on the SAME XCD vs on DIFFERENT XCDs. Workgroups are dispatched round-robin over the 8 XCDs (blockIdx % 8 on an idle
// device); every workgroup records its HW_REG_XCC_ID so the placement is verified, not assumed.
// Build: hipcc --offload-arch=gfx950 -O3 -o pingpong_bench.bin pingpong_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

// block A (blockIdx == a) and block B (blockIdx == b) bounce a counter `rounds` times through two words
__global__ void pingpong(unsigned long long *w, int a, int b, int rounds, long long *t, unsigned *xcc)
{
    if (threadIdx.x == 0) xcc[blockIdx.x] = xcc_id();
    if (threadIdx.x != 0 || (blockIdx.x != a && blockIdx.x != b)) return;
    const bool isA = blockIdx.x == a;
    unsigned long long *mine = w + (isA ? 0 : 16), *other = w + (isA ? 16 : 0);   // separate 128-B lines
    long long t0 = wall_clock64();
    for (int i = 1; i <= rounds; i++) {
        if (isA) {
            __hip_atomic_store(mine, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spin = 0;
            while (__hip_atomic_load(other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)i && ++spin < (1 << 22)) {}
        } else {
            int spin = 0;
            while (__hip_atomic_load(other, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)i && ++spin < (1 << 22)) {}
            __hip_atomic_store(mine, (unsigned long long)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    long long t1 = wall_clock64();
    if (isA) { t[0] = t0; t[1] = t1; }
}

int main()
{
    unsigned long long *w; long long *t; unsigned *xcc;
    hipMalloc(&w, 4096); hipMalloc(&t, 64); hipMalloc(&xcc, 4096);
    const int grid = 64, rounds = 2000;
    std::vector<unsigned> hx(grid);
    for (int rep = 0; rep < 2; rep++) {
        for (int b : {8, 16, 1, 2, 4, 7, 9}) {
            hipMemset(w, 0, 4096);
            hipLaunchKernelGGL(pingpong, dim3(grid), dim3(64), 0, 0, w, 0, b, rounds, t, xcc);
            hipDeviceSynchronize();
            long long ht[2];
            hipMemcpy(ht, t, 16, hipMemcpyDeviceToHost);
            hipMemcpy(hx.data(), xcc, 4 * grid, hipMemcpyDeviceToHost);
            printf("A=block0(xcc %u) B=block%d(xcc %u): one-way %.0f ns\n", hx[0], b, hx[b], (ht[1] - ht[0]) * 10.0 / rounds / 2);
        }
    }
    printf("xcc of blocks 0..15:");
    for (int i = 0; i < 16; i++) printf(" %u", hx[i]);
    printf("\n");
    return 0;
}
