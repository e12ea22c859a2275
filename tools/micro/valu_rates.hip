// Issue cost of the integer instructions the search kernel leans on (gfx950): a dependent chain per lane would measure latency,
// so each thread runs 8 independent chains; 5 waves per SIMD like the search kernel.  Prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHAINS 8
#define ITERS 4096
template <int OP>
__global__ void __launch_bounds__(256) k(unsigned *out, unsigned a, unsigned b) {
    unsigned x[CHAINS];
    unsigned long long y[CHAINS];
    for (int q = 0; q < CHAINS; q++) { x[q] = threadIdx.x * 7u + q + a; y[q] = ((unsigned long long)x[q] << 32) | (b + q); }
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int q = 0; q < CHAINS; q++) {
            if (OP == 0) x[q] = x[q] + b;                                            // v_add_u32
            else if (OP == 1) x[q] = x[q] * b;                                       // v_mul_lo_u32
            else if (OP == 2) x[q] = __umulhi(x[q], b);                              // v_mul_hi_u32
            else if (OP == 3) y[q] = (unsigned long long)(unsigned)y[q] * b + y[q];  // v_mad_u64_u32
            else if (OP == 4) y[q] = (y[q] << 2) + ((unsigned long long)a << 7);     // v_lshl_add_u64
            else if (OP == 5) x[q] = __builtin_amdgcn_perm(x[q], b, 0x05020301u);     // v_perm_b32
            else if (OP == 6) x[q] = __popc(x[q]) + b;                               // v_bcnt_u32_b32
            else if (OP == 7) x[q] = (x[q] & b) | (~x[q] & a);                       // v_bfi / bitop3
            else if (OP == 8) x[q] = x[q] > b ? x[q] - b : x[q] + a;                 // cmp + cndmask + ...
            else if (OP == 9) x[q] = (x[q] & 0xFFFFFFu) * (b & 0xFFFFFFu);              // v_mul_u32_u24
            else if (OP == 10) y[q] = y[q] * 0x9E3779B97F4A7C15ull + b;              // 64-bit multiply
            else if (OP == 11) x[q] = (unsigned)__umul64hi(y[q], (unsigned long long)x[q] | 1ull), y[q] += x[q];   // umul64hi by a 32-bit value
            else if (OP == 12) x[q] = __builtin_amdgcn_readlane(x[q], (it + q) & 63) + x[q];   // v_readlane (+ add)
        }
    }
    unsigned s = 0;
    for (int q = 0; q < CHAINS; q++) s += x[q] + (unsigned)y[q] + (unsigned)(y[q] >> 32);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP> double run(unsigned *d, const char *name, double base) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 5;
    k<OP><<<blocks, 256>>>(d, 3, 5);
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, 3, 5);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // waves per SIMD = 5; each executes ITERS * CHAINS ops
    const double ops_per_simd = 5.0 * ITERS * CHAINS;
    printf("%-44s %8.3f ms  %.2f ns per wave-op per SIMD (x %.2f of v_add_u32)\n", name, ms, ms * 1e6 / ops_per_simd, base > 0 ? ms / base : 1.0);
    return ms;
}
int main() {
    unsigned *d; hipMalloc(&d, 256 * 5 * 256 * 4);
    double b = run<0>(d, "v_add_u32", 0);
    run<1>(d, "v_mul_lo_u32", b); run<2>(d, "v_mul_hi_u32", b); run<3>(d, "v_mad_u64_u32", b); run<4>(d, "v_lshl_add_u64", b);
    run<5>(d, "v_perm_b32", b); run<6>(d, "v_bcnt_u32_b32 (+add)", b); run<7>(d, "and/or/not (bitop3)", b); run<8>(d, "cmp+cndmask+sub/add", b);
    run<9>(d, "v_mul_u32_u24", b); run<10>(d, "64-bit multiply by a constant + add", b); run<11>(d, "umul64hi(y, 32-bit) + add64", b);
    run<12>(d, "v_readlane + add", b);
    return 0;
}
