// clock_bench.hip -- what does an instruction of a LONE wavefront cost? Shader clock (s_memtime) per instruction for dependent and
// independent chains of fp64 / fp32 / int / conversion / DPP instructions, at 1, 2 and 4 wavefronts per SIMD.
// (Round 3 finding: a wavefront issues one fp64 instruction every ~12 cycles whether or not it depends on the previous one, so a
// single-wavefront phase costs instructions x 5 ns; several wavefronts per SIMD overlap almost perfectly.)
// Build: hipcc --offload-arch=gfx950 -O3 -o clock_bench.bin clock_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND, int INDEP>
__global__ void chain(double *out, long long *t, int n)
{
    double a = out[threadIdx.x], a2 = a + 1, a3 = a + 2, a4 = a + 3;
    const double b = 1.0000001, c = 1e-9;
    float f = (float)a, f2 = f + 1, f3 = f + 2, f4 = f + 3;
    const float fb = 1.0000001f, fc = 1e-9f;
    int k = (int)a, k2 = k + 1, k3 = k + 2, k4 = k + 3;
    long long w0 = wall_clock64(); long long c0 = clock64();
    for (int i = 0; i < n / 16; i++) {
#pragma unroll
      for (int u = 0; u < 16; u++) {
        if (KIND == 0) {          // fp64 fma
            if (INDEP) { a = __builtin_fma(a, b, c); a2 = __builtin_fma(a2, b, c); a3 = __builtin_fma(a3, b, c); a4 = __builtin_fma(a4, b, c); }
            else { a = __builtin_fma(a, b, c); a = __builtin_fma(a, b, c); a = __builtin_fma(a, b, c); a = __builtin_fma(a, b, c); }
        } else if (KIND == 1) {   // fp32 fma
            if (INDEP) { f = __builtin_fmaf(f, fb, fc); f2 = __builtin_fmaf(f2, fb, fc); f3 = __builtin_fmaf(f3, fb, fc); f4 = __builtin_fmaf(f4, fb, fc); }
            else { f = __builtin_fmaf(f, fb, fc); f = __builtin_fmaf(f, fb, fc); f = __builtin_fmaf(f, fb, fc); f = __builtin_fmaf(f, fb, fc); }
        } else if (KIND == 2) {   // int mad
            if (INDEP) { k = k * 3 + 1; k2 = k2 * 3 + 1; k3 = k3 * 3 + 1; k4 = k4 * 3 + 1; }
            else { k = k * 3 + 1; k = k * 5 + 1; k = k * 7 + 1; k = k * 9 + 1; }
        } else if (KIND == 3) {   // fp64 add
            if (INDEP) { a = a + c; a2 = a2 + c; a3 = a3 + c; a4 = a4 + c; }
            else { a = a + c; a = a + b; a = a + c; a = a + b; }
        } else if (KIND == 4) {   // cvt f32 -> f64 -> f32 (2 instructions per step)
            if (INDEP) { f = (float)((double)f + 0.0); f2 = (float)((double)f2); f3 = (float)((double)f3); f4 = (float)((double)f4); asm volatile("" : "+v"(f), "+v"(f2), "+v"(f3), "+v"(f4)); }
            else { f = (float)(double)f; asm volatile("" : "+v"(f)); f = (float)(double)f; asm volatile("" : "+v"(f)); f = (float)(double)f; asm volatile("" : "+v"(f)); f = (float)(double)f; asm volatile("" : "+v"(f)); }
        } else if (KIND == 5) {   // DPP mov (32-bit) + add
            f = f + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), 0x128, 0xf, 0xf, false));
            f = f + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), 0x124, 0xf, 0xf, false));
            f = f + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), 0x4E, 0xf, 0xf, false));
            f = f + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, f), 0xB1, 0xf, 0xf, false));
        } else if (KIND == 6) {   // fp32 add, non-fused mul+add pairs (what -ffp-contract=off float math looks like)
            if (INDEP) { f = f * fb; f2 = f2 * fb; f3 = f3 + fc; f4 = f4 + fc; }
            else { f = f * fb; f = f + fc; f = f * fb; f = f + fc; }
        }
      }
    }
    long long c1 = clock64(); long long w1 = wall_clock64();
    out[threadIdx.x] = a + a2 + a3 + a4 + f + f2 + f3 + f4 + k + k2 + k3 + k4;
    if (threadIdx.x == 0 && blockIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}
template <int KIND, int INDEP>
static void run(double *o, long long *t, const char *name)
{
    const int n = 3200;
    printf("%-34s %-11s", name, INDEP ? "independent" : "dependent");
    for (int threads : {256, 512, 1024}) {
        hipLaunchKernelGGL((chain<KIND, INDEP>), dim3(1), dim3(threads), 0, 0, o, t, n);
        hipDeviceSynchronize();
        long long h[2]; hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
        printf("  %d w/SIMD: %5.2f tick %5.2f ns", threads / 256, (double)h[0] / (4.0 * n), h[1] * 10.0 / (4.0 * n));
    }
    printf("\n");
}
int main()
{
    double *o; long long *t; hipMalloc(&o, 8 * 1024); hipMalloc(&t, 64); hipMemset(o, 0, 8 * 1024);
    run<0, 0>(o, t, "fp64 fma"); run<0, 1>(o, t, "fp64 fma");
    run<3, 0>(o, t, "fp64 add"); run<3, 1>(o, t, "fp64 add");
    run<1, 0>(o, t, "fp32 fma"); run<1, 1>(o, t, "fp32 fma");
    run<6, 0>(o, t, "fp32 mul / add"); run<6, 1>(o, t, "fp32 mul / add");
    run<2, 0>(o, t, "int32 mad"); run<2, 1>(o, t, "int32 mad");
    run<4, 0>(o, t, "cvt f32->f64->f32 (2 instr/step)"); run<4, 1>(o, t, "cvt f32->f64->f32 (2 instr/step)");
    run<5, 0>(o, t, "dpp mov + fp32 add (2 instr/step)");
    return 0;
}
