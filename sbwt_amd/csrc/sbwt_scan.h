// sbwt_scan.h -- exclusive prefix sum of n int64 values (three small kernels), used by the output formatting
// and by the path-order build.  `static`: every translation unit that includes this gets its own copy.
#pragma once
#include "sbwt_kernels_common.h"

// exclusive scan of n int64 values in[] -> out[] (out has n+1 entries, out[n] = total); 1024 per block
static __global__ void __launch_bounds__(256) k_scan_block_sums(const i64 *__restrict__ in, i64 n, i64 *__restrict__ bsum) {
    __shared__ i64 sh[4];
    const i64 base = (i64)blockIdx.x * 1024;
    i64 s = 0;
    for (int t = threadIdx.x; t < 1024; t += 256) s += (base + t < n) ? in[base + t] : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
static __global__ void __launch_bounds__(1024) k_scan_sums(i64 *bsum, i64 nb) {   // one block: in-place exclusive scan
    __shared__ i64 sh[1024];
    const i64 chunk = (nb + 1023) / 1024;
    const i64 lo = (i64)threadIdx.x * chunk, hi = (lo + chunk < nb) ? lo + chunk : nb;
    i64 loc = 0;
    for (i64 b = lo; b < hi; b++) loc += bsum[b];
    sh[threadIdx.x] = loc;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        i64 add = (threadIdx.x >= (unsigned)off) ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += add;
        __syncthreads();
    }
    i64 run = sh[threadIdx.x] - loc;
    for (i64 b = lo; b < hi; b++) { i64 v = bsum[b]; bsum[b] = run; run += v; }
    if (threadIdx.x == 1023) bsum[nb] = sh[1023];
}
static __global__ void __launch_bounds__(256) k_scan_apply(const i64 *__restrict__ in, i64 n, const i64 *__restrict__ bsum,
                                                    i64 *__restrict__ out) {
    __shared__ i64 sh[256];
    const i64 base = (i64)blockIdx.x * 1024;
    i64 v[4], loc = 0;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const i64 idx = base + threadIdx.x * 4 + t;
        v[t] = (idx < n) ? in[idx] : 0;
        loc += v[t];
    }
    sh[threadIdx.x] = loc;
    __syncthreads();
    for (int off = 1; off < 256; off <<= 1) {        // Hillis-Steele over the 256 partial sums
        i64 add = (threadIdx.x >= off) ? sh[threadIdx.x - off] : 0;
        __syncthreads();
        sh[threadIdx.x] += add;
        __syncthreads();
    }
    i64 run = bsum[blockIdx.x] + sh[threadIdx.x] - loc;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const i64 idx = base + threadIdx.x * 4 + t;
        if (idx < n) out[idx] = run;
        run += v[t];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = bsum[gridDim.x];
}

