// sbwt_sort.hip -- optional pre-pass of the path-order search: hand the reads to the search kernel sorted by WHERE IN THE
// PATH ORDER they start.  The reads of a batch arrive in random order, so the 64 lanes of a wave walk 64 unrelated places
// of col / pq / trans and every step drags its own 128-byte line through the fabric (25 lines per read on config 2).  Reads
// that start next to each other walk the same paths: sorted, the lanes of a wave share most of those lines (L1/L2 hits;
// measured with perfectly sorted input: 255 M -> 136 M lines per 10 M reads).  The results do not change -- a read's results
// depend on the read alone (SBWT::streaming_search is const, SBWT.hh:544) and go to the same place of `out`.
//
//   k_anchor   per read: the path position of its first k-mer that the index holds, looking at offsets 0, k, 2k, 3k
//              (one exact lookup each in the depth-k sparse table, stop at the first hit); reads without one sort last
//   rocPRIM radix sort of (position, read) pairs -> perm[ticket] = read
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include "sbwt_kernels_common.h"

__global__ void __launch_bounds__(256) k_anchor(SbwtIndexView ix, const uint4 *__restrict__ packed,
                                                const i64 *__restrict__ read_off, i64 n_reads, const SbwtWorkHeader *ws,
                                                unsigned *__restrict__ keys, unsigned *__restrict__ ids) {
    const i64 r = (i64)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_reads) return;
    const bool uni = ws->u_bad == 0 && ws->u_len > 0;
    const i64 P0 = uni ? ws->u_read0 + r * ws->u_len : read_off[r];
    const i64 len = uni ? ws->u_len : read_off[r + 1] - P0;
    const int k = ix.k;
    unsigned key_out = (unsigned)ix.n_nodes;               // no anchor: after every path position
    for (int o = 0; o < 4 * k && o + k <= len; o += k) {
        const i64 P = P0 + o;
        const uint4 g0 = packed[P >> 5], g1 = packed[(P >> 5) + 1];
        const int s = (int)(P & 31);
        u64 w = quad_bits(g0) >> (2 * s);
        if (s) w |= quad_bits(g1) << (64 - 2 * s);
        const u64 vr = (((u64)g1.w << 32) | (u64)g0.w) >> s;
        const u64 vm = low_mask(k);
        if ((vr & vm) != vm) continue;                     // a non-ACGT base in the window
        const u64 key = w & low_mask(2 * k);
        size_t bkt = sbwt_sp_bucket(key, ix.n_sb, 0u);
        bool found = false;
        for (int tries = 0; tries < 64; tries++) {         // buckets a key had to skip carry the overflow flag
            const uint4 e0 = ix.stab[2 * bkt], e1 = ix.stab[2 * bkt + 1];
            const u64 w0 = quad_bits(e0), w1 = quad_bits(e1);
            if ((w0 & ~SBWT_SP_OVERFLOW) == key) { key_out = e0.w; found = true; break; }
            if (w1 == key) { key_out = e1.w; found = true; break; }
            if (!(w0 & SBWT_SP_OVERFLOW)) break;
            bkt = bkt + 1 < ix.n_sb ? bkt + 1 : 0;
        }
        if (found) break;
    }
    keys[r] = key_out;
    ids[r] = (unsigned)r;
}

// scratch layout: keys_in | ids_in | keys_out | ids_out (4 x n_reads x 4 B, 256-byte aligned parts) | rocPRIM temporary
static inline size_t part_bytes(long long n_reads) { return (((size_t)n_reads * 4) + 255) & ~(size_t)255; }
long long sbwt_sort_scratch_bytes(long long n_reads, int key_bits) {
    size_t tmp = 0;
    (void)rocprim::radix_sort_pairs(nullptr, tmp, (const unsigned *)nullptr, (unsigned *)nullptr, (const unsigned *)nullptr,
                                    (unsigned *)nullptr, (size_t)n_reads, 0u, (unsigned)key_bits, (hipStream_t)0);
    return (long long)(4 * part_bytes(n_reads) + tmp + 256);
}
// Returns the permutation (device pointer inside d_scratch), or nullptr if the sort could not be set up.
const unsigned *sbwt_launch_sort_reads(const SbwtIndexView &ix, const uint4 *d_packed, const long long *d_read_off,
                                       long long n_reads, const SbwtWorkHeader *ws, void *d_scratch, long long scratch_bytes,
                                       int key_bits, hipStream_t stream) {
    char *base = static_cast<char *>(d_scratch);
    const size_t pb = part_bytes(n_reads);
    unsigned *keys_in = reinterpret_cast<unsigned *>(base), *ids_in = reinterpret_cast<unsigned *>(base + pb);
    unsigned *keys_out = reinterpret_cast<unsigned *>(base + 2 * pb), *ids_out = reinterpret_cast<unsigned *>(base + 3 * pb);
    void *tmp = base + 4 * pb;
    size_t tmp_bytes = (size_t)scratch_bytes - 4 * pb;
    hipLaunchKernelGGL(k_anchor, dim3(grid_for(n_reads)), dim3(256), 0, stream, ix, d_packed, d_read_off, (i64)n_reads, ws,
                       keys_in, ids_in);
    if (rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, ids_in, ids_out, (size_t)n_reads, 0u, (unsigned)key_bits,
                                  stream) != hipSuccess)
        return nullptr;
    return ids_out;
}
