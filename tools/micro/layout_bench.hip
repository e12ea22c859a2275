// layout_bench.hip -- what would a denser block layout buy?  Dependent random lookups over N columns with
//   A: 64 B per 64 columns   (today's layout): two 16-B loads from one 64-B block
//   B: 128 B per 192 columns (5.33 bits/column): four 16-B loads from one 128-B block + one 16-B load
//      from a tiny superblock table (L2 resident)
// Reports G lookups/s for both at the two index sizes of the bench configs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
__device__ __forceinline__ u64 mix(u64 x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

template <int LAYOUT>
__global__ void __launch_bounds__(256) k(const uint4 *__restrict__ t, const uint4 *__restrict__ sb, u64 n_cols, int iters, u64 *sink) {
    u64 st = mix((u64)blockIdx.x * 256 + threadIdx.x + 1);
    for (int it = 0; it < iters; it++) {
        u64 col = st % n_cols;
        u64 x;
        if (LAYOUT == 0) {
            const uint4 *p = t + (col >> 6) * 4 + ((st >> 50) & 2);
            uint4 a = p[0], b = p[1];
            x = (u64)a.x ^ ((u64)b.w << 32) ^ a.z;
        } else {
            u64 blk = col / 192;
            const uint4 *p = t + blk * 8;
            int c2 = (int)((st >> 50) & 3);                 // bits of symbol c: quads c2*3/2.. (two quads), ssup+counts: quads 6,7
            uint4 a = p[(c2 * 3) >> 1], b = p[((c2 * 3) >> 1) + 1], s6 = p[6], s7 = p[7];
            uint4 base = sb[blk >> 8];
            x = (u64)a.x ^ ((u64)b.w << 32) ^ s6.y ^ s7.w ^ base.z;
        }
        st = mix(st ^ x);
    }
    if (st == 0x1234567) sink[0] = st;
}
template <int LAYOUT>
double run(const uint4 *t, const uint4 *sb, u64 n_cols, int blocks, u64 *sink) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 300;
    hipLaunchKernelGGL((k<LAYOUT>), dim3(blocks), dim3(256), 0, 0, t, sb, n_cols, 8, sink); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<LAYOUT>), dim3(blocks), dim3(256), 0, 0, t, sb, n_cols, iters, sink);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    return (double)blocks * 256 * iters / (ms * 1e-3) / 1e9;
}
int main() {
    u64 *sink; (void)hipMalloc(&sink, 64);
    for (u64 n_cols : {12835509ull, 142349118ull}) {
        size_t bytesA = (n_cols / 64 + 1) * 64, bytesB = (n_cols / 192 + 1) * 128;
        uint4 *tA, *tB, *sb;
        (void)hipMalloc(&tA, bytesA); (void)hipMalloc(&tB, bytesB); (void)hipMalloc(&sb, (n_cols / 192 / 256 + 2) * 16);
        (void)hipMemset(tA, 0x5a, bytesA); (void)hipMemset(tB, 0x5a, bytesB); (void)hipMemset(sb, 1, (n_cols / 192 / 256 + 2) * 16);
        for (int blocks : {1024, 2048}) {
            double a = run<0>(tA, sb, n_cols, blocks, sink), b = run<1>(tB, sb, n_cols, blocks, sink);
            printf("n_cols %llu blocks %d: A (64B/64col, %.1f MB) %.1f G/s   B (128B/192col, %.1f MB) %.1f G/s\n", n_cols, blocks, bytesA / 1e6, a, bytesB / 1e6, b);
        }
        (void)hipFree(tA); (void)hipFree(tB); (void)hipFree(sb);
    }
    return 0;
}
