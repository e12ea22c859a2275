// gather_bench.hip -- microbenchmark: dependent chains of random 16-byte gathers from a table,
// the access pattern of the SBWT search kernels.  Reports G loads/s for table sizes (L2 / Infinity
// Cache / HBM resident), chains per lane (memory-level parallelism per lane) and occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;

__device__ __forceinline__ u64 mix(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

template <int CHAINS, int BYTES>   // BYTES per load: 16 or 32 (two adjacent quads)
__global__ void __launch_bounds__(256) k_gather(const uint4 *__restrict__ table, u64 n_lines_mask, int iters,
                                                u64 *__restrict__ sink) {
    u64 tid = (u64)blockIdx.x * 256 + threadIdx.x;
    u64 st[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; c++) st[c] = mix(tid * CHAINS + c + 1);
    for (int it = 0; it < iters; it++) {
        uint4 v[CHAINS], w[CHAINS];
#pragma unroll
        for (int c = 0; c < CHAINS; c++) {
            const uint4 *p = table + ((st[c] & n_lines_mask) << 2) + ((st[c] >> 40) & 2);
            v[c] = p[0];
            if (BYTES == 32) w[c] = p[1];
        }
#pragma unroll
        for (int c = 0; c < CHAINS; c++) {
            u64 x = (u64)v[c].x | ((u64)v[c].y << 32);
            if (BYTES == 32) x ^= (u64)w[c].z << 7;
            st[c] = mix(st[c] ^ x);
        }
    }
    u64 acc = 0;
#pragma unroll
    for (int c = 0; c < CHAINS; c++) acc ^= st[c];
    if (acc == 0x1234567) sink[0] = acc;
}

template <int CHAINS, int BYTES>
double run(const uint4 *table, u64 mask, int iters, int blocks, u64 *sink) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_gather<CHAINS, BYTES>), dim3(blocks), dim3(256), 0, 0, table, mask, 8, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_gather<CHAINS, BYTES>), dim3(blocks), dim3(256), 0, 0, table, mask, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return (double)blocks * 256 * CHAINS * iters / (ms * 1e-3) / 1e9;
}

int main() {
    u64 *sink; hipMalloc(&sink, 64);
    const size_t sizes_mb[] = {2, 12, 32, 128, 1024, 4096};
    printf("%8s %7s %6s %6s %12s %12s\n", "table", "blocks", "chains", "bytes", "Gloads/s", "ns/iter/wave");
    for (size_t mb : sizes_mb) {
        size_t bytes = mb << 20;
        size_t lines = 1; while (lines * 2 * 64 <= bytes) lines *= 2;   // power-of-two number of 64-B lines
        uint4 *table; if (hipMalloc(&table, lines * 64) != hipSuccess) { printf("alloc %zu MB failed\n", mb); continue; }
        std::vector<u64> h(lines * 8);
        u64 s = 88172645463325252ULL;
        for (auto &x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = s; }
        hipMemcpy(table, h.data(), lines * 64, hipMemcpyHostToDevice);
        for (int blocks : {512, 1024, 2048}) {
            int iters = 400;
            double g;
            g = run<1, 16>(table, lines - 1, iters, blocks, sink);
            printf("%6zuMB %7d %6d %6d %12.2f %12.1f\n", lines * 64 >> 20, blocks, 1, 16, g, (double)blocks * 4 * 1e9 / (g * 1e9 / 64) / 1.0);
            g = run<1, 32>(table, lines - 1, iters, blocks, sink);
            printf("%6zuMB %7d %6d %6d %12.2f\n", lines * 64 >> 20, blocks, 1, 32, g);
            g = run<2, 16>(table, lines - 1, iters, blocks, sink);
            printf("%6zuMB %7d %6d %6d %12.2f\n", lines * 64 >> 20, blocks, 2, 16, g);
            g = run<4, 16>(table, lines - 1, iters, blocks, sink);
            printf("%6zuMB %7d %6d %6d %12.2f\n", lines * 64 >> 20, blocks, 4, 16, g);
            g = run<4, 32>(table, lines - 1, iters, blocks, sink);
            printf("%6zuMB %7d %6d %6d %12.2f\n", lines * 64 >> 20, blocks, 4, 32, g);
        }
        hipFree(table);
    }
    return 0;
}
