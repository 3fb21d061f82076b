// valu_rate.hip -- issue cost of the vector instructions the at-scale VIO producers are made of, at SATURATION (several wavefronts
// per SIMD, independent operands): shader-clock ticks per instruction and SIMD. clock_bench.hip answers the lone-wavefront question.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate.bin valu_rate.hip        (tools/README.md)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define BODY(name, ins)                                                                                         \
    __global__ void k_##name(long long *t, int n)                                                               \
    {                                                                                                           \
        long long c0 = clock64();                                                                               \
        for (int i = 0; i < n; i++) { asm volatile(REP8(REP8(ins "\n")) ::: "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25"); } \
        long long c1 = clock64();                                                                               \
        if (threadIdx.x == 0) t[blockIdx.x] = c1 - c0;                                                          \
    }
// 64 instructions per loop trip; destinations rotate over v10..v25 through the assembler's \@-free form: fixed registers, independent
BODY(fma_f64, "v_fma_f64 v[10:11], v[2:3], v[4:5], v[6:7]\n v_fma_f64 v[12:13], v[2:3], v[4:5], v[6:7]\n v_fma_f64 v[14:15], v[2:3], v[4:5], v[6:7]\n v_fma_f64 v[16:17], v[2:3], v[4:5], v[6:7]\n")
BODY(fmac_f64, "v_fmac_f64 v[10:11], v[2:3], v[4:5]\n v_fmac_f64 v[12:13], v[2:3], v[4:5]\n v_fmac_f64 v[14:15], v[2:3], v[4:5]\n v_fmac_f64 v[16:17], v[2:3], v[4:5]\n")
BODY(add_f64, "v_add_f64 v[10:11], v[2:3], v[4:5]\n v_add_f64 v[12:13], v[2:3], v[4:5]\n v_add_f64 v[14:15], v[2:3], v[4:5]\n v_add_f64 v[16:17], v[2:3], v[4:5]\n")
BODY(mul_f64, "v_mul_f64 v[10:11], v[2:3], v[4:5]\n v_mul_f64 v[12:13], v[2:3], v[4:5]\n v_mul_f64 v[14:15], v[2:3], v[4:5]\n v_mul_f64 v[16:17], v[2:3], v[4:5]\n")
BODY(cvt_f64_f32, "v_cvt_f64_f32 v[10:11], v2\n v_cvt_f64_f32 v[12:13], v3\n v_cvt_f64_f32 v[14:15], v4\n v_cvt_f64_f32 v[16:17], v5\n")
BODY(cvt_f32_f64, "v_cvt_f32_f64 v10, v[2:3]\n v_cvt_f32_f64 v11, v[4:5]\n v_cvt_f32_f64 v12, v[6:7]\n v_cvt_f32_f64 v13, v[2:3]\n")
BODY(mul_f32, "v_mul_f32 v10, v2, v3\n v_mul_f32 v11, v2, v3\n v_mul_f32 v12, v2, v3\n v_mul_f32 v13, v2, v3\n")
BODY(fma_f32, "v_fma_f32 v10, v2, v3, v4\n v_fma_f32 v11, v2, v3, v4\n v_fma_f32 v12, v2, v3, v4\n v_fma_f32 v13, v2, v3, v4\n")
BODY(pk_mul_f32, "v_pk_mul_f32 v[10:11], v[2:3], v[4:5]\n v_pk_mul_f32 v[12:13], v[2:3], v[4:5]\n v_pk_mul_f32 v[14:15], v[2:3], v[4:5]\n v_pk_mul_f32 v[16:17], v[2:3], v[4:5]\n")
BODY(pk_add_f32, "v_pk_add_f32 v[10:11], v[2:3], v[4:5]\n v_pk_add_f32 v[12:13], v[2:3], v[4:5]\n v_pk_add_f32 v[14:15], v[2:3], v[4:5]\n v_pk_add_f32 v[16:17], v[2:3], v[4:5]\n")
BODY(pk_fma_f32, "v_pk_fma_f32 v[10:11], v[2:3], v[4:5], v[6:7]\n v_pk_fma_f32 v[12:13], v[2:3], v[4:5], v[6:7]\n v_pk_fma_f32 v[14:15], v[2:3], v[4:5], v[6:7]\n v_pk_fma_f32 v[16:17], v[2:3], v[4:5], v[6:7]\n")
BODY(cvt_ubyte, "v_cvt_f32_ubyte0 v10, v2\n v_cvt_f32_ubyte1 v11, v2\n v_cvt_f32_ubyte2 v12, v2\n v_cvt_f32_ubyte3 v13, v2\n")
BODY(mov_b32, "v_mov_b32 v10, v2\n v_mov_b32 v11, v3\n v_mov_b32 v12, v4\n v_mov_b32 v13, v5\n")
BODY(mov_b64, "v_mov_b64 v[10:11], v[2:3]\n v_mov_b64 v[12:13], v[4:5]\n v_mov_b64 v[14:15], v[6:7]\n v_mov_b64 v[16:17], v[2:3]\n")
BODY(mix_f64_f32, "v_fma_f64 v[10:11], v[2:3], v[4:5], v[6:7]\n v_mul_f32 v12, v2, v3\n v_fma_f64 v[14:15], v[2:3], v[4:5], v[6:7]\n v_mul_f32 v13, v2, v3\n")
typedef void (*kfn)(long long *, int);
static void run(const char *name, kfn k, long long *t)
{
    const int n = 400;
    printf("%-14s", name);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wps : {1, 2, 4, 8}) {                                     // wavefronts per SIMD (8: two 1024-thread workgroups per CU)
        const int threads = wps >= 4 ? 1024 : 256 * wps, blocks = wps == 8 ? 512 : 256;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, t, 10);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, t, n);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        long long h[512]; hipMemcpy(h, t, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
        double s = 0; for (int i = 0; i < blocks; i++) s += (double)h[i];
        const double per_wave = s / blocks / (n * 256.0);              // ticks per instruction as one wavefront sees it (256 instr / trip)
        // the whole launch by the host's events: ns per wave-instruction and SIMD (1024 SIMDs)
        const double ns_simd = (double)ms * 1e6 / ((double)n * 256.0 * wps);
        printf("  %dw: %5.2f tick/wave %5.2f ns/SIMD", wps, per_wave, ns_simd);
    }
    printf("\n");
}
int main()
{
    long long *t; hipMalloc(&t, 512 * 8);
    int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("device clock attribute %d kHz; clock64 ticks (s_memtime)\n", clk);
#define R(x) run(#x, k_##x, t)
    R(fma_f64); R(fmac_f64); R(add_f64); R(mul_f64); R(cvt_f64_f32); R(cvt_f32_f64); R(mul_f32); R(fma_f32); R(pk_mul_f32); R(pk_add_f32);
    R(pk_fma_f32); R(cvt_ubyte); R(mov_b32); R(mov_b64); R(mix_f64_f32);
    return 0;
}
