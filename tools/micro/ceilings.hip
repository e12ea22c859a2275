// ceilings.hip -- what the MI355X memory system delivers for the access shapes of the search kernels, measured with
// enough loads in flight per wave (unrolled, independent addresses) that the numbers are ceilings of the memory
// system, not of the test loop:
//   copy / read / write        coalesced streams, 16 B per lane, UNROLL independent accesses per lane per trip
//   gather G x S               random reads of S-byte segments (S = 16, 64, 128), each by S/16 adjacent lanes, from a
//                              table of a given size; addresses do not depend on loaded data (throughput, not latency)
//   scatter S                  whole S-byte segments (64, 128) written at random addresses
// Build: hipcc --offload-arch=gfx950 -O3 -o ceilings ceilings.hip       Run: ./ceilings [table MB ...]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ u64 mix(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

template <int UNROLL, bool NT>
__global__ void __launch_bounds__(256) k_copy(const u32x4 *__restrict__ in, u32x4 *__restrict__ out, u64 n) {
    const u64 stride = (u64)gridDim.x * 256;
    for (u64 t = (u64)blockIdx.x * 256 + threadIdx.x; t + (UNROLL - 1) * stride < n; t += UNROLL * stride) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = NT ? __builtin_nontemporal_load(in + t + u * stride) : in[t + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            if (NT) __builtin_nontemporal_store(v[u], out + t + u * stride);
            else out[t + u * stride] = v[u];
        }
    }
}
template <int UNROLL>
__global__ void __launch_bounds__(256) k_read(const u32x4 *__restrict__ in, u64 n, u64 *sink) {
    const u64 stride = (u64)gridDim.x * 256;
    unsigned acc = 0;
    for (u64 t = (u64)blockIdx.x * 256 + threadIdx.x; t + (UNROLL - 1) * stride < n; t += UNROLL * stride) {
        u32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = __builtin_nontemporal_load(in + t + u * stride);
#pragma unroll
        for (int u = 0; u < UNROLL; u++) acc ^= v[u].x ^ v[u].w;
    }
    if (acc == 0x12345u) sink[0] = acc;
}
template <int UNROLL>
__global__ void __launch_bounds__(256) k_write(u32x4 *__restrict__ out, u64 n) {
    const u64 stride = (u64)gridDim.x * 256;
    for (u64 t = (u64)blockIdx.x * 256 + threadIdx.x; t + (UNROLL - 1) * stride < n; t += UNROLL * stride) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            u32x4 v = {(unsigned)t, 1u, 2u, (unsigned)u};
            __builtin_nontemporal_store(v, out + t + u * stride);
        }
    }
}
// SEG bytes per segment: SEG/16 adjacent lanes read one segment together; G independent segments per lane-group per trip
template <int SEG, int G>
__global__ void __launch_bounds__(256) k_gather(const u32x4 *__restrict__ table, u64 seg_mask, int iters, u64 *sink) {
    constexpr int LPS = SEG / 16;
    const u64 tid = (u64)blockIdx.x * 256 + threadIdx.x, grp = tid / LPS, sub = tid % LPS;
    u64 st = mix(grp + 1);
    unsigned acc = 0;
    for (int it = 0; it < iters; it++) {
        u32x4 v[G];
#pragma unroll
        for (int g = 0; g < G; g++) v[g] = table[(mix(st + g) & seg_mask) * LPS + sub];
#pragma unroll
        for (int g = 0; g < G; g++) acc ^= v[g].x;
        st = mix(st + 0x9E3779B9ull);
    }
    if (acc == 0x12345u) sink[0] = acc;
}
template <int SEG>
__global__ void __launch_bounds__(256) k_scatter(u32x4 *__restrict__ out, u64 seg_mask, int iters) {
    constexpr int LPS = SEG / 16;
    const u64 tid = (u64)blockIdx.x * 256 + threadIdx.x, grp = tid / LPS, sub = tid % LPS;
    u64 st = mix(grp + 1);
    for (int it = 0; it < iters; it++) {
        u32x4 v = {(unsigned)st, 1u, 2u, 3u};
        __builtin_nontemporal_store(v, out + (st & seg_mask) * LPS + sub);
        st = mix(st);
    }
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
int main(int argc, char **argv) {
    const u64 bytes = 4ull << 30, nq = bytes / 16;
    u32x4 *a, *b; u64 *sink;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess) return 1;
    (void)hipMemset(a, 1, bytes); (void)hipMemset(b, 0, bytes);
    printf("== coalesced streams over %llu MiB, 16 B per lane ==\n", bytes >> 20);
    for (int blocks : {1024, 2048, 4096, 8192}) {
        double t;
        t = timed([&] { hipLaunchKernelGGL((k_copy<4, true>), dim3(blocks), dim3(256), 0, 0, a, b, nq); });
        printf("copy  nt unroll 4  %5d blocks: %.2f TB/s moved (read + write)\n", blocks, 2.0 * bytes / t / 1e12);
        t = timed([&] { hipLaunchKernelGGL((k_copy<8, true>), dim3(blocks), dim3(256), 0, 0, a, b, nq); });
        printf("copy  nt unroll 8  %5d blocks: %.2f TB/s moved\n", blocks, 2.0 * bytes / t / 1e12);
        t = timed([&] { hipLaunchKernelGGL((k_copy<8, false>), dim3(blocks), dim3(256), 0, 0, a, b, nq); });
        printf("copy     unroll 8  %5d blocks: %.2f TB/s moved\n", blocks, 2.0 * bytes / t / 1e12);
        t = timed([&] { hipLaunchKernelGGL((k_read<8>), dim3(blocks), dim3(256), 0, 0, a, nq, sink); });
        printf("read  nt unroll 8  %5d blocks: %.2f TB/s\n", blocks, bytes / t / 1e12);
        t = timed([&] { hipLaunchKernelGGL((k_write<8>), dim3(blocks), dim3(256), 0, 0, b, nq); });
        printf("write nt unroll 8  %5d blocks: %.2f TB/s\n", blocks, bytes / t / 1e12);
    }
    std::vector<size_t> sizes = {12, 128, 1024, 4096};
    if (argc > 1) { sizes.clear(); for (int i = 1; i < argc; i++) sizes.push_back((size_t)atol(argv[i])); }
    printf("== random segment reads (independent addresses), 8192 blocks x 256 lanes ==\n");
    printf("%8s %5s %3s %14s %10s\n", "table", "seg", "G", "G segments/s", "TB/s");
    const int blocks = 8192, iters = 64;
    for (size_t mb : sizes) {
        size_t tb = mb << 20;
        if (tb > bytes) tb = bytes;
        size_t p2 = 1; while (p2 * 2 <= tb) p2 *= 2;                  // power-of-two bytes of the table actually used
#define RUNG(SEG, G)                                                                                                   \
        {                                                                                                              \
            double t = timed([&] { hipLaunchKernelGGL((k_gather<SEG, G>), dim3(blocks), dim3(256), 0, 0, a, p2 / SEG - 1, iters, sink); }); \
            double ns = (double)blocks * 256 / (SEG / 16) * G * iters;                                                 \
            printf("%6zuMB %5d %3d %14.1f %10.2f\n", p2 >> 20, SEG, G, ns / t / 1e9, ns * SEG / t / 1e12);             \
        }
        RUNG(16, 1) RUNG(16, 4) RUNG(16, 8) RUNG(64, 4) RUNG(64, 8) RUNG(128, 4) RUNG(128, 8)
    }
    printf("== random whole-segment writes over 4096 MiB ==\n");
    {
        double t = timed([&] { hipLaunchKernelGGL((k_scatter<64>), dim3(blocks), dim3(256), 0, 0, b, bytes / 64 - 1, iters); });
        double ns = (double)blocks * 256 / 4 * iters;
        printf("scatter  64 B: %.1f G segments/s, %.2f TB/s\n", ns / t / 1e9, ns * 64 / t / 1e12);
        t = timed([&] { hipLaunchKernelGGL((k_scatter<128>), dim3(blocks), dim3(256), 0, 0, b, bytes / 128 - 1, iters); });
        ns = (double)blocks * 256 / 8 * iters;
        printf("scatter 128 B: %.1f G segments/s, %.2f TB/s\n", ns / t / 1e9, ns * 128 / t / 1e12);
    }
    return 0;
}
