// sbwt_format.hip -- print_vector (src/CLI/sbwt_search.cpp:21-43) for a whole batch on the device.
#include "sbwt_kernels_common.h"
#include "sbwt_scan.h"

// ---------------------------------------------------------------------------------------------
// Output formatting on the device: print_vector of src/CLI/sbwt_search.cpp:21-43 for a whole batch.
// One line per read, every value followed by one space, '\n' per read, -1 printed as "-1", and the
// reference's quirk kept: 0 prints as an empty token.  One wave per read.
//   k_fmt_len    line length of every read
//   k_scan_*     exclusive prefix sum of the line lengths (three small kernels)
//   k_fmt_write  the characters
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int fmt_len(i64 v) {      // characters of the token incl. its trailing space
    if (v < 0) return 3;                             // "-1 "
    int n = 1;                                       // the space; 0 -> empty token
    u64 x = (u64)v;
    while (x > 0) { n++; x /= 10; }
    return n;
}

__global__ void __launch_bounds__(256) k_fmt_len(const i64 *__restrict__ vals, const i64 *__restrict__ out_off,
                                                 i64 n_reads, i64 *__restrict__ line_len) {
    const i64 r = ((i64)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (r >= n_reads) return;
    const i64 lo = out_off[r], hi = out_off[r + 1];
    i64 sum = 0;
    for (i64 t = lo + lane; t < hi; t += 64) sum += fmt_len(vals[t]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
    if (lane == 0) line_len[r] = sum + 1;            // + '\n'
}

__global__ void __launch_bounds__(256) k_fmt_write(const i64 *__restrict__ vals, const i64 *__restrict__ out_off,
                                                   i64 n_reads, const i64 *__restrict__ line_off,
                                                   char *__restrict__ text) {
    const i64 r = ((i64)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (r >= n_reads) return;
    const i64 lo = out_off[r], hi = out_off[r + 1];
    i64 pos = line_off[r];
    for (i64 t0 = lo; t0 < hi; t0 += 64) {
        const i64 t = t0 + lane;
        const i64 v = (t < hi) ? vals[t] : 0;
        const int len = (t < hi) ? fmt_len(v) : 0;
        int incl = len;                              // inclusive wave prefix sum of the token lengths
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            int up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        if (t < hi) {
            char *p = text + pos + (incl - len);
            if (v < 0) { p[0] = '-'; p[1] = '1'; p[2] = ' '; }
            else {
                p[len - 1] = ' ';
                u64 x = (u64)v;
                for (int d = len - 2; d >= 0; d--) { p[d] = (char)('0' + (int)(x % 10)); x /= 10; }
            }
        }
        pos += __shfl(incl, 63);
    }
    if (lane == 0) text[pos] = '\n';
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
long long sbwt_format_scratch_bytes(long long n_reads) {
    return (n_reads + (n_reads + 1023) / 1024 + 2) * 8 + 256;
}

void sbwt_launch_format(const long long *d_vals, const long long *d_out_off, long long n_reads, char *d_text,
                        long long *d_line_off, void *d_scratch, hipStream_t stream) {
    if (n_reads <= 0) return;
    i64 *line_len = reinterpret_cast<i64 *>(d_scratch);
    i64 *bsum = line_len + n_reads;
    const unsigned wave_blocks = (unsigned)((n_reads * 64 + 255) / 256);
    const unsigned nb = (unsigned)((n_reads + 1023) / 1024);
    hipLaunchKernelGGL(k_fmt_len, dim3(wave_blocks), dim3(256), 0, stream, d_vals, d_out_off, (i64)n_reads, line_len);
    hipLaunchKernelGGL(k_scan_block_sums, dim3(nb), dim3(256), 0, stream, line_len, (i64)n_reads, bsum);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, stream, bsum, (i64)nb);
    hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(256), 0, stream, line_len, (i64)n_reads, bsum, d_line_off);
    hipLaunchKernelGGL(k_fmt_write, dim3(wave_blocks), dim3(256), 0, stream, d_vals, d_out_off, (i64)n_reads,
                       d_line_off, d_text);
}

// scratch: last[n] + next[n] bytes + acc[n_blocks] words

