// hop_bench.hip -- what does ONE cross-workgroup hop cost by store / load flavour and by placement?
// Two workgroups bounce an 8-byte tagged word (csrc/handoff.h "the data IS the flag") through two separate 128-B lines.
// Flavours (gfx950 cache-policy bits): store plain / sc0 / sc1 / sc0 sc1 ; load sc1 / sc0 sc1 (an L1 hit can never see the word).
// Placement: block b runs on XCD b % 8 (verified through HW_REG_XCC_ID, not assumed): b = 8 same XCD, b = 1 another XCD.
// A plain or sc0 store leaves the line dirty in the writer's XCD L2 (MI355X_MICROARCH.md): a same-XCD reader finds it there at L2
// latency, a reader on another XCD must never see it (time-out column) -- this bench measures both.
// Build: hipcc --offload-arch=gfx950 -O3 -o hop_bench.bin hop_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>
#include <algorithm>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned xcc_id()
{
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
template <int ST>
__device__ __forceinline__ void st64(unsigned long long *p, unsigned long long v)
{
    if (ST == 0) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (ST == 1) asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    if (ST == 2) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    if (ST == 3) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
}
template <int LD>
__device__ __forceinline__ unsigned long long ld64(const unsigned long long *p)
{
    unsigned long long v;
    if (LD == 0) asm volatile("global_load_dwordx2 %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (LD == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (LD == 2) asm volatile("global_load_dwordx2 %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int ST, int LD>
__global__ void pingpong(unsigned long long *w, int a, int b, int rounds, long long *t, unsigned *xcc)
{
    if (threadIdx.x == 0) xcc[blockIdx.x] = xcc_id();
    if (threadIdx.x != 0 || (blockIdx.x != a && blockIdx.x != b)) return;
    const bool isA = blockIdx.x == a;
    unsigned long long *mine = w + (isA ? 0 : 16), *other = w + (isA ? 16 : 0);   // separate 128-B lines
    long long t0 = wall_clock64();
    int timeouts = 0;
    for (int i = 1; i <= rounds; i++) {
        if (isA) st64<ST>(mine, (unsigned long long)i);
        int spin = 0;
        while (ld64<LD>(other) < (unsigned long long)i && ++spin < 20000) {}
        if (spin >= 20000) { timeouts++; if (timeouts > 3) break; }
        if (!isA) st64<ST>(mine, (unsigned long long)i);
    }
    long long t1 = wall_clock64();
    if (isA) { t[0] = t0; t[1] = t1; t[2] = timeouts; }
}

// ---- fan-in: P producer workgroups (blocks p * stride, i.e. one XCD when stride == 8) publish a 256-B record each (32 tagged
// 8-byte words, ST flavour) when the collector's "go" word (sc1) reaches them; the collector (block `cblk`) polls all records with
// 16-byte LD-flavour loads (one 16-lane row per record) until every tag matches. Time: go -> collector holds all records.
template <int ST, int LD>
__global__ void fanin(unsigned long long *rec, unsigned long long *go, int P, int stride, int cblk, int rounds, long long *t, unsigned *xcc, int words = 32)
{
    const int S = P * (words / 32);                      // 256-byte slots the collector polls (<= 256)
    if (threadIdx.x == 0) xcc[blockIdx.x] = xcc_id();
    const int b = blockIdx.x;
    const bool producer = (b % stride == 0) && (b / stride < P) && b != cblk;
    if (!producer && b != cblk) return;
    if (b == cblk) {
        long long tot = 0;
        int timeouts = 0;
        for (int r = 1; r <= rounds; r++) {
            __syncthreads();
            long long t0 = wall_clock64();
            if (threadIdx.x == 0) st64<3>(go, (unsigned long long)r);
            // 256 threads: row = tid / 16 covers record row + 16 k; all loads of a sweep in flight together (as handoff.h gather_records)
            const int kp = threadIdx.x & 15, row = threadIdx.x >> 4;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)rec, 0, S * 256, 0x00020000);
            bool all = false;
            int spin = 0;
            while (!all && spin < 4000) {
                bool ok = true;
                u4 v[16];
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int q = row + 16 * k;
                    if (q < S) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, q * 256 + kp * 16, 0, LD == 0 ? 16 : 17);
                }
#pragma unroll
                for (int k = 0; k < 16; k++) {
                    const int q = row + 16 * k;
                    if (q < S) ok = ok && (v[k].x == (unsigned)r) && (v[k].z == (unsigned)r);
                }
                all = __syncthreads_and(ok ? 1 : 0) != 0;
                spin++;
            }
            if (!all) timeouts++;
            long long t1 = wall_clock64();
            tot += t1 - t0;
            if (threadIdx.x == 0) t[8 + r] = t1 - t0;
        }
        if (threadIdx.x == 0) { t[0] = tot; t[2] = timeouts; }
        return;
    }
    // producers: records indexed by producer ordinal
    const int q = b / stride;
    for (int r = 1; r <= rounds; r++) {
        if (threadIdx.x < 64) {
            int spin = 0;
            while (ld64<0>(go) < (unsigned long long)r && ++spin < 400000) {}
        }
        __syncthreads();
        if ((int)threadIdx.x < words) st64<ST>(rec + (size_t)q * words + threadIdx.x, ((unsigned long long)(threadIdx.x + 1) << 32) | (unsigned long long)r);
    }
}

template <int ST, int LD>
static void run_pp(unsigned long long *w, long long *t, unsigned *xcc, int b)
{
    const int grid = 64, rounds = 2000;
    std::vector<unsigned> hx(grid);
    hipMemset(w, 0, 4096);
    hipLaunchKernelGGL((pingpong<ST, LD>), dim3(grid), dim3(64), 0, 0, w, 0, b, rounds, t, xcc);
    hipDeviceSynchronize();
    long long ht[3];
    hipMemcpy(ht, t, 24, hipMemcpyDeviceToHost);
    hipMemcpy(hx.data(), xcc, 4 * grid, hipMemcpyDeviceToHost);
    static const char *sn[] = {"plain", "sc0", "sc1", "sc0sc1"}, *ln[] = {"sc1", "sc0sc1", "sc0"};
    printf("pingpong store %-6s load %-6s  block0(xcc %u) <-> block%d(xcc %u): one-way %6.0f ns  timeouts %lld\n", sn[ST], ln[LD], hx[0], b, hx[b],
           (ht[1] - ht[0]) * 10.0 / rounds / 2, ht[2]);
}
template <int ST, int LD>
static void run_fi(unsigned long long *rec, unsigned long long *go, long long *t, unsigned *xcc, int P, int stride, int cblk)
{
    const int rounds = 500;
    const int grid = 256;
    hipMemset(rec, 0, 1 << 20); hipMemset(go, 0, 4096);
    hipLaunchKernelGGL((fanin<ST, LD>), dim3(grid), dim3(256), 0, 0, rec, go, P, stride, cblk, rounds, t, xcc);
    hipDeviceSynchronize();
    std::vector<long long> ht(8 + rounds + 1);
    hipMemcpy(ht.data(), t, 8 * ht.size(), hipMemcpyDeviceToHost);
    std::vector<unsigned> hx(grid);
    hipMemcpy(hx.data(), xcc, 4 * grid, hipMemcpyDeviceToHost);
    std::vector<long long> d(ht.begin() + 9 + 20, ht.end());        // skip the first rounds
    std::sort(d.begin(), d.end());
    static const char *sn[] = {"plain", "sc0", "sc1", "sc0sc1"}, *ln[] = {"sc1", "sc0sc1", "sc0"};
    printf("fan-in %3d producers (stride %d) -> block %d (xcc %u; producer 1 on xcc %u)  store %-6s load %-6s : go -> all records min %5lld median %5lld p90 %5lld ns  timeouts %lld\n", P, stride, cblk,
           hx[cblk], hx[stride], sn[ST], ln[LD], d[0] * 10, d[d.size() / 2] * 10, d[d.size() * 9 / 10] * 10, ht[2]);
}

// --json P:words [P:words ...]: what bench.py's `roofline.latency_model` is built from, measured in the SAME run as the bench line -- one hop
// (sc1 store -> sc1 load, today's protocol) inside an XCD and across, and the fan-in of P records of `words` 8-byte words each from
// producers spread over all XCDs into one collector workgroup (go word -> all records held), median over 480 rounds. One JSON line.
static double pp_ns(unsigned long long *w, long long *t, unsigned *xcc, int b)
{
    const int grid = 64, rounds = 2000;
    hipMemset(w, 0, 4096);
    hipLaunchKernelGGL((pingpong<2, 0>), dim3(grid), dim3(64), 0, 0, w, 0, b, rounds, t, xcc);
    hipDeviceSynchronize();
    long long ht[3];
    hipMemcpy(ht, t, 24, hipMemcpyDeviceToHost);
    return ht[2] ? -1.0 : (ht[1] - ht[0]) * 10.0 / rounds / 2;
}
static double fi_ns(unsigned long long *rec, unsigned long long *go, long long *t, unsigned *xcc, int P, int words)
{
    const int rounds = 500, grid = 256;
    hipMemset(rec, 0, 1 << 20); hipMemset(go, 0, 4096);
    hipLaunchKernelGGL((fanin<2, 0>), dim3(grid), dim3(256), 0, 0, rec, go, P, 1, 255, rounds, t, xcc, words);
    hipDeviceSynchronize();
    std::vector<long long> ht(8 + rounds + 1);
    hipMemcpy(ht.data(), t, 8 * ht.size(), hipMemcpyDeviceToHost);
    if (ht[2]) return -1.0;
    std::vector<long long> d(ht.begin() + 9 + 20, ht.end());
    std::sort(d.begin(), d.end());
    return d[d.size() / 2] * 10.0;
}
static int json_mode(int argc, char **argv)
{
    unsigned long long *w, *rec, *go; long long *t; unsigned *xcc;
    if (hipMalloc(&t, 65536) != hipSuccess) { printf("{\"error\": \"no device\"}\n"); return 1; }
    hipMalloc(&xcc, 4096); hipMalloc(&w, 4096); hipMalloc(&rec, 1 << 20); hipMalloc(&go, 4096);
    pp_ns(w, t, xcc, 8); fi_ns(rec, go, t, xcc, 64, 32);            // warm up
    printf("{\"hop_same_xcd_ns\": %.0f, \"hop_cross_xcd_ns\": %.0f, \"fan_in_ns\": {", pp_ns(w, t, xcc, 8), pp_ns(w, t, xcc, 1));
    bool first = true;
    for (int a = 2; a < argc; a++) {
        int P = 0, words = 32;
        if (sscanf(argv[a], "%d:%d", &P, &words) < 1 || P < 1 || (words != 32 && words != 64) || P * (words / 32) > 255) continue;
        const double a1 = fi_ns(rec, go, t, xcc, P, words), a2 = fi_ns(rec, go, t, xcc, P, words);
        printf("%s\"%d:%d\": %.0f", first ? "" : ", ", P, words, a1 < a2 ? a1 : a2);
        first = false;
    }
    printf("}, \"protocol\": \"8-byte tagged words, sc1 stores, sc1 16-byte loads (csrc/handoff.h); fan-in = collector's go word -> every record held, producers on all XCDs\"}\n");
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 1 && std::string(argv[1]) == "--json") return json_mode(argc, argv);
    unsigned long long *w, *rec, *go; long long *t; unsigned *xcc;
    hipMalloc(&t, 65536); hipMalloc(&xcc, 4096);
    const int memkind = argc > 1 ? atoi(argv[1]) : 0;      // 0 hipMalloc, 1 fine-grained, 2 uncached
    if (memkind == 0) { hipMalloc(&w, 4096); hipMalloc(&rec, 1 << 20); hipMalloc(&go, 4096); }
    else {
        const unsigned fl = memkind == 1 ? hipDeviceMallocFinegrained : hipDeviceMallocUncached;
        hipError_t e1 = hipExtMallocWithFlags((void **)&w, 4096, fl), e2 = hipExtMallocWithFlags((void **)&rec, 1 << 20, fl), e3 = hipExtMallocWithFlags((void **)&go, 4096, fl);
        printf("memory kind %d: %s %s %s\n", memkind, hipGetErrorString(e1), hipGetErrorString(e2), hipGetErrorString(e3));
    }
    for (int rep = 0; rep < 2; rep++) {
        for (int b : {8, 1}) {
            run_pp<2, 0>(w, t, xcc, b);      // today's protocol: sc1 / sc1
            run_pp<3, 1>(w, t, xcc, b);
            run_pp<0, 0>(w, t, xcc, b);      // plain store, L1-bypassing load
            run_pp<1, 0>(w, t, xcc, b);
            run_pp<0, 1>(w, t, xcc, b);
            run_pp<1, 2>(w, t, xcc, b);      // sc0 / sc0: workgroup scope both sides (expected stale: L1 hit)
        }
    }
    for (int rep = 0; rep < 2; rep++) {
        // 24 producers on ONE XCD + collector on the same XCD (block 248 = xcc 0), by flavour
        run_fi<2, 0>(rec, go, t, xcc, 24, 8, 248);
        run_fi<0, 0>(rec, go, t, xcc, 24, 8, 248);
        run_fi<1, 0>(rec, go, t, xcc, 24, 8, 248);
        run_fi<2, 0>(rec, go, t, xcc, 16, 8, 248);
        run_fi<0, 0>(rec, go, t, xcc, 16, 8, 248);
        run_fi<0, 0>(rec, go, t, xcc, 31, 8, 248);
        // the same producers, collector on ANOTHER XCD (block 249)
        run_fi<2, 0>(rec, go, t, xcc, 24, 8, 249);
        if (rep == 0) run_fi<0, 0>(rec, go, t, xcc, 24, 8, 249);      // expected: never visible (500 time-outs)
        // 8 producers spread over the XCDs (stride 1) -> one collector: the second level
        run_fi<2, 0>(rec, go, t, xcc, 8, 1, 248);
        // today's flat gather: 192 producers over all XCDs -> one collector
        run_fi<2, 0>(rec, go, t, xcc, 192, 1, 255);
        run_fi<2, 0>(rec, go, t, xcc, 96, 1, 255);
        run_fi<2, 0>(rec, go, t, xcc, 160, 1, 255);
        run_fi<2, 0>(rec, go, t, xcc, 24, 1, 255);
    }
    return 0;
}
