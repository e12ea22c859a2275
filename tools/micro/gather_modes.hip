// gather_modes.hip -- does the size of the fabric request behind a random 16-byte gather depend on HOW the memory is
// allocated (coarse-grained / fine-grained / uncached) or on the load's cache-policy bits (nt, sc0, sc1)?
// Every L2 miss of the search kernels leaves the XCD as one 128-byte request (profiles/r02_ceilings.txt); this measures
// the alternatives.  Build: hipcc --offload-arch=gfx950 -O3 -o gather_modes gather_modes.hip    Run: ./gather_modes [MB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u64 mix(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
// MODE 0 plain, 1 nt builtin, 2 asm sc0, 3 asm sc1, 4 asm sc0 sc1, 5 asm nt, 6 asm sc0 sc1 nt
// (the asm forms issue their four loads and wait for them inside ONE statement: the compiler does not track the
// asynchronous register writes of a load it did not emit)
#define LD4(MOD)                                                                                                      \
    asm volatile("global_load_dwordx4 %0, %4, off " MOD "\n\tglobal_load_dwordx4 %1, %5, off " MOD "\n\t"             \
                 "global_load_dwordx4 %2, %6, off " MOD "\n\tglobal_load_dwordx4 %3, %7, off " MOD "\n\ts_waitcnt vmcnt(0)" \
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]) : "memory")
template <int MODE>
__global__ void __launch_bounds__(256) k_gather(const u32x4 *__restrict__ table, u64 mask, int iters, u64 *sink) {
    const u64 tid = (u64)blockIdx.x * 256 + threadIdx.x;
    u64 st = mix(tid + 1);
    unsigned acc = 0;
    for (int it = 0; it < iters; it++) {
        u32x4 v[4];
        const u32x4 *p[4];
#pragma unroll
        for (int g = 0; g < 4; g++) p[g] = table + (mix(st + g) & mask);
        if (MODE == 0) {
#pragma unroll
            for (int g = 0; g < 4; g++) v[g] = *p[g];
        } else if (MODE == 1) {
#pragma unroll
            for (int g = 0; g < 4; g++) v[g] = __builtin_nontemporal_load(p[g]);
        } else if (MODE == 2) { LD4("sc0"); }
        else if (MODE == 3) { LD4("sc1"); }
        else if (MODE == 4) { LD4("sc0 sc1"); }
        else if (MODE == 5) { LD4("nt"); }
        else if (MODE == 6) { LD4("sc0 sc1 nt"); }
        else { LD4(""); }
#pragma unroll
        for (int g = 0; g < 4; g++) acc ^= v[g].x;
        st = mix(st + 0x9E3779B9ull);
    }
    if (acc == 0x12345u) sink[0] = acc;
}
template <typename F> static double timed(F launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(); (void)hipDeviceSynchronize();
    double best = 1e30;
    for (int r = 0; r < 3; r++) {
        (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms * 1e-3 < best) best = ms * 1e-3;
    }
    return best;
}
template <int MODE> static void run(const char *alloc, const char *name, const u32x4 *t, u64 bytes, u64 *sink) {
    const int blocks = 8192, iters = 32;
    double s = timed([&] { hipLaunchKernelGGL((k_gather<MODE>), dim3(blocks), dim3(256), 0, 0, t, bytes / 16 - 1, iters, sink); });
    double n = (double)blocks * 256 * 4 * iters;
    printf("%-14s %-16s %8.1f G gathers/s\n", alloc, name, n / s / 1e9);
    fflush(stdout);
}
int main(int argc, char **argv) {
    const u64 bytes = (u64)(argc > 1 ? atol(argv[1]) : 1024) << 20;
    u64 *sink;
    if (hipMalloc(&sink, 64) != hipSuccess) return 1;
    struct { const char *name; unsigned flags; int ext; } allocs[] = {
        {"hipMalloc", 0, 0}, {"finegrained", hipDeviceMallocFinegrained, 1}, {"uncached", hipDeviceMallocUncached, 1}};
    int only = argc > 2 ? atoi(argv[2]) : -1;
    for (int a = 0; a < 3; a++) {
        if (only >= 0 && only != a) continue;
        u32x4 *t = nullptr;
        hipError_t e = allocs[a].ext ? hipExtMallocWithFlags((void **)&t, bytes, allocs[a].flags) : hipMalloc((void **)&t, bytes);
        if (e != hipSuccess) { printf("%s: allocation failed (%s)\n", allocs[a].name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
        (void)hipMemset(t, 1, bytes);
        (void)hipDeviceSynchronize();
        run<0>(allocs[a].name, "plain", t, bytes, sink);
        run<1>(allocs[a].name, "nt (builtin)", t, bytes, sink);
        run<2>(allocs[a].name, "sc0", t, bytes, sink);
        run<3>(allocs[a].name, "sc1", t, bytes, sink);
        run<4>(allocs[a].name, "sc0 sc1", t, bytes, sink);
        run<5>(allocs[a].name, "nt", t, bytes, sink);
        run<6>(allocs[a].name, "sc0 sc1 nt", t, bytes, sink);
        run<7>(allocs[a].name, "asm, no bits", t, bytes, sink);
        (void)hipFree(t);
    }
    return 0;
}
