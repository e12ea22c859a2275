// stream_bench.hip -- microbenchmark behind DESIGN.md's "fabric request ceiling": how many 64-byte requests per
// second the memory system takes for (a) a coalesced streaming copy, (b) coalesced streaming writes only,
// (c) whole 64-byte lines written at random addresses, (d) a mix like the search kernel's (random 16-byte
// gathers + whole-line writes, 5:3).  Build: hipcc --offload-arch=gfx950 -O3 -o stream_bench stream_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
__device__ __forceinline__ u64 mix(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
__global__ void __launch_bounds__(256) k_copy(const uint4 *__restrict__ in, uint4 *__restrict__ out, u64 n) {
    for (u64 t = (u64)blockIdx.x * 256 + threadIdx.x; t < n; t += (u64)gridDim.x * 256) out[t] = in[t];
}
__global__ void __launch_bounds__(256) k_fill(uint4 *__restrict__ out, u64 n) {
    for (u64 t = (u64)blockIdx.x * 256 + threadIdx.x; t < n; t += (u64)gridDim.x * 256) out[t] = make_uint4(1, 2, 3, (unsigned)t);
}
// groups of 4 lanes write one whole 64-byte line each, at a random line per group and iteration
__global__ void __launch_bounds__(256) k_line_writes(uint4 *__restrict__ out, u64 line_mask, int iters) {
    const u64 tid = (u64)blockIdx.x * 256 + threadIdx.x, grp = tid >> 2, sub = tid & 3;
    u64 st = mix(grp + 1);
    for (int it = 0; it < iters; it++) {
        out[((st & line_mask) << 2) + sub] = make_uint4((unsigned)st, 1, 2, 3);
        st = mix(st);
    }
}
// per group of 4 lanes and iteration: GATHERS random 16-byte loads per lane ... and one whole-line write per group
template <int GATHERS>
__global__ void __launch_bounds__(256) k_mix(const uint4 *__restrict__ table, u64 tmask, uint4 *__restrict__ out, u64 line_mask,
                                             int iters, u64 *sink) {
    const u64 tid = (u64)blockIdx.x * 256 + threadIdx.x, grp = tid >> 2, sub = tid & 3;
    u64 st = mix(tid + 1), ws = mix(grp + 77), acc = 0;
    for (int it = 0; it < iters; it++) {
        uint4 v[GATHERS];
#pragma unroll
        for (int g = 0; g < GATHERS; g++) v[g] = table[(mix(st + g) & tmask) << 2];
#pragma unroll
        for (int g = 0; g < GATHERS; g++) acc ^= v[g].x;
        out[((ws & line_mask) << 2) + sub] = make_uint4((unsigned)acc, 1, 2, 3);
        st = mix(st + 0x9E37);                         // addresses do not depend on the data: throughput, not latency
        ws = mix(ws);
    }
    if (acc == 0x12345) sink[0] = acc;
}
template <typename F> static double timed(F launch) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3;
}
int main() {
    const u64 bytes = 8ull << 30, nq = bytes / 16, lines = bytes / 64;
    uint4 *a, *b; u64 *sink;
    (void)hipMalloc(&a, bytes); (void)hipMalloc(&b, bytes); (void)hipMalloc(&sink, 64);
    (void)hipMemset(a, 1, bytes); (void)hipMemset(b, 0, bytes);
    const int blocks = 256 * 8;
    double t = timed([&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, a, b, nq); });
    printf("streaming copy      : %.2f TB/s moved, %.1f G 64-byte requests/s\n", 2.0 * bytes / t / 1e12, 2.0 * lines / t / 1e9);
    t = timed([&] { hipLaunchKernelGGL(k_fill, dim3(blocks), dim3(256), 0, 0, b, nq); });
    printf("streaming write     : %.2f TB/s, %.1f G requests/s\n", bytes / t / 1e12, lines / t / 1e9);
    const int iters = 256;
    t = timed([&] { hipLaunchKernelGGL(k_line_writes, dim3(blocks * 4), dim3(256), 0, 0, b, lines - 1, iters); });
    const double nl = (double)blocks * 4 * 256 / 4 * iters;
    printf("random line writes  : %.2f TB/s, %.1f G requests/s\n", nl * 64 / t / 1e12, nl / t / 1e9);
    t = timed([&] { hipLaunchKernelGGL((k_mix<1>), dim3(blocks * 4), dim3(256), 0, 0, a, (bytes / 64) - 1, b, lines - 1, iters, sink); });
    const double ng = (double)blocks * 4 * 256 * iters;
    printf("mix 4 gathers : 1 line write: %.1f G requests/s (%.1f reads + %.1f writes), %.2f TB/s of 64-byte sectors\n",
           (ng + nl) / t / 1e9, ng / t / 1e9, nl / t / 1e9, (ng + nl) * 64 / t / 1e12);
    t = timed([&] { hipLaunchKernelGGL((k_mix<2>), dim3(blocks * 4), dim3(256), 0, 0, a, (bytes / 64) - 1, b, lines - 1, iters, sink); });
    printf("mix 8 gathers : 1 line write: %.1f G requests/s, %.2f TB/s of 64-byte sectors\n", (2 * ng + nl) / t / 1e9,
           (2 * ng + nl) * 64 / t / 1e12);
    return 0;
}
