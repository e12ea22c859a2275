// mixed_rw.hip -- do random reads and streaming writes share the memory system additively?  The search kernel's launch is
// priced as reads / 56 G/s + writes / 80 G/s (bench.py request_model); this measures the premise: ONE kernel whose lanes do
// R random 16-byte gathers (each its own 128-byte line of a 1 GB table; addresses independent of loaded data) and W coalesced
// nontemporal 16-byte stores per trip, for several R : W mixes, and reports the time against the two ceilings measured alone
// in the same run.
// Build: hipcc --offload-arch=gfx950 -O3 -o mixed_rw mixed_rw.hip       Run: ./mixed_rw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u64 mix(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
// per trip and lane: R gathers, W stores (16 B each); trips per lane = n_trips
template <int R, int W>
__global__ void __launch_bounds__(256) k_mixed(const u32x4 *__restrict__ table, u64 n_lines, u32x4 *__restrict__ out, u64 n_trips, u64 *sink) {
    const u64 tid = (u64)blockIdx.x * 256 + threadIdx.x, nthreads = (u64)gridDim.x * 256;
    unsigned acc = 0;
    for (u64 t = 0; t < n_trips; t++) {
        u32x4 v[R > 0 ? R : 1];
#pragma unroll
        for (int r = 0; r < R; r++) {
            const u64 line = mix((tid * n_trips + t) * 8 + r + 1) % n_lines;
            v[r] = table[line * 8 + (tid & 7)];                       // 16 bytes of a random 128-byte line
        }
#pragma unroll
        for (int r = 0; r < R; r++) acc ^= v[r].x;
#pragma unroll
        for (int w = 0; w < W; w++) {
            u32x4 o = {(unsigned)t, acc & 1u, 2u, (unsigned)w};
            __builtin_nontemporal_store(o, out + ((t * W + w) * nthreads + tid));      // coalesced: consecutive lanes, 16 B each
        }
    }
    if (acc == 0x12345u) sink[0] = acc;
}
template <int R, int W>
static double run(const u32x4 *table, u64 n_lines, u32x4 *out, u64 n_trips, u64 *sink, int grid) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(a);
        hipLaunchKernelGGL((k_mixed<R, W>), dim3(grid), dim3(256), 0, 0, table, n_lines, out, n_trips, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep && ms < best) best = ms;
    }
    return best;
}
int main() {
    const u64 table_bytes = 1ull << 30, n_lines = table_bytes / 128;
    const int grid = 1280 * 4;                                        // 20 waves per CU like the search kernel, x 4 for the loop-free shape
    const u64 nthreads = (u64)grid * 256, n_trips = 64;
    u32x4 *table, *out; u64 *sink;
    hipMalloc(&table, table_bytes); hipMemset(table, 1, table_bytes);
    hipMalloc(&out, nthreads * n_trips * 4 * 16); hipMalloc(&sink, 8);
    const double lanes = (double)nthreads * n_trips;
    const double t_r = run<4, 0>(table, n_lines, out, n_trips, sink, grid);      // reads alone: 4 gathers per trip
    const double t_w = run<0, 4>(table, n_lines, out, n_trips, sink, grid);      // writes alone: 4 stores per trip
    const double rd = lanes * 4 / (t_r * 1e-3), wr = lanes * 4 * 16 / (t_w * 1e-3);
    printf("alone: random 16-byte reads %.1f G/s (each a 128-byte line); coalesced nontemporal writes %.2f TB/s = %.1f G 64-byte requests/s\n",
           rd / 1e9, wr / 1e12, wr / 64 / 1e9);
#define MIX(R, W) do { const double t = run<R, W>(table, n_lines, out, n_trips, sink, grid); \
        const double model = (lanes * R / rd + lanes * W * 16 / wr) * 1e3; \
        printf("mix %d gathers : %d stores per trip: %.3f ms; additive model %.3f ms (%.0f %%); reads %.1f G/s, writes %.2f TB/s\n", R, W, t, model, \
               100.0 * model / t, lanes * R / (t * 1e-3) / 1e9, lanes * W * 16 / (t * 1e-3) / 1e12); } while (0)
    MIX(4, 4); MIX(4, 2); MIX(2, 4); MIX(4, 1); MIX(1, 4); MIX(3, 4);
    return 0;
}
