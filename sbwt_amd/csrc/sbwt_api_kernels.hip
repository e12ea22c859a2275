// sbwt_api_kernels.hip -- the batched API neighbours of the search path: rank, prefix-table precalc,
// update_sbwt_interval, forward (one thread per item; see sbwt_search.hip for the search kernels).
#include "sbwt_kernels_common.h"

// ---------------------------------------------------------------------------------------------
// k_rank: SubsetMatrixRank::rank(pos, c) for n independent (pos, sym) pairs
// ---------------------------------------------------------------------------------------------
template <bool MEGA>
__global__ void __launch_bounds__(256) k_rank(SbwtIndexView ix, const i64 *__restrict__ pos,
                                              const char *__restrict__ sym, i64 n, i64 *__restrict__ out) {
    i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    unsigned b = (unsigned char)sym[t];
    i64 ps = pos[t];
    i64 res = 0;
    if (is_ACGT(b)) {
        int c = (int)dna_code(b);
        uint4 q = ix.blocks[((ps >> 6) << 2) + c];
        res = (i64)quad_rank<MEGA>(ix, q, ps, c) - ix.C[c];
    }
    out[t] = res;
}

// ---------------------------------------------------------------------------------------------
// k_precalc: do_kmer_prefix_precalc (SBWT.hh:616-645): entry d = interval of the p-mer whose
// i-th char is (d >> 2i) & 3, starting from {0, n_nodes-1}
// ---------------------------------------------------------------------------------------------
template <bool MEGA>
__global__ void __launch_bounds__(256) k_precalc(SbwtIndexView ix, int p, longlong2 *__restrict__ table) {
    u64 d = (u64)blockIdx.x * 256 + threadIdx.x;
    if (d >= (1ull << (2 * p))) return;
    i64 l = 0, r = ix.n_nodes - 1;
    for (int i = 0; i < p; i++) {
        int c = (int)((d >> (2 * i)) & 3ull);
        uint4 q1 = ix.blocks[((l >> 6) << 2) + c];
        uint4 q2 = ix.blocks[(((r + 1) >> 6) << 2) + c];
        l = (i64)quad_rank<MEGA>(ix, q1, l, c);
        r = (i64)quad_rank<MEGA>(ix, q2, r + 1, c) - 1;
        if (l > r) { l = -1; r = -1; break; }
    }
    table[d] = make_longlong2(l, r);
}

// ---------------------------------------------------------------------------------------------
// k_update_interval: SBWT::update_sbwt_interval (SBWT.hh:422-437), one lane per query
// ---------------------------------------------------------------------------------------------
template <bool MEGA>
__global__ void __launch_bounds__(256) k_update_interval(SbwtIndexView ix, const char *__restrict__ bases,
                                                         const i64 *__restrict__ off, i64 n,
                                                         i64 *__restrict__ first, i64 *__restrict__ second) {
    i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    i64 l = first[t], r = second[t];
    if (l == -1) return;
    for (i64 i = off[t]; i < off[t + 1]; i++) {
        unsigned b = (unsigned char)bases[i];
        if (!is_ACGT(b)) { l = -1; r = -1; break; }      // raw char validated (SBWT.hh:427-428)
        int c = (int)dna_code(b);
        uint4 q1 = ix.blocks[((l >> 6) << 2) + c];
        uint4 q2 = ix.blocks[(((r + 1) >> 6) << 2) + c];
        l = (i64)quad_rank<MEGA>(ix, q1, l, c);
        r = (i64)quad_rank<MEGA>(ix, q2, r + 1, c) - 1;
        if (l > r) { l = -1; r = -1; break; }
    }
    first[t] = l;
    second[t] = r;
}

// ---------------------------------------------------------------------------------------------
// k_forward: SBWT::forward (SBWT.hh:368-381), one lane per (node, sym)
// ---------------------------------------------------------------------------------------------
template <bool MEGA>
__global__ void __launch_bounds__(256) k_forward(SbwtIndexView ix, const i64 *__restrict__ node,
                                                 const char *__restrict__ sym, i64 n, i64 *__restrict__ out) {
    i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    unsigned b = (unsigned char)sym[t];
    i64 res = -1;                                          // rank() of a non-ACGT char is 0 -> r1 == r2 -> -1
    if (is_ACGT(b)) {
        int c = (int)dna_code(b);
        i64 v = node[t];
        i64 blk = v >> 6;
        const uint4 *pa = ix.blocks + ((blk << 2) + (c & 2));
        uint4 e = pa[0], o = pa[1];
        u64 msk = ((u64)e.w | ((u64)o.w << 32)) & ((2ull << (v & 63)) - 1ull);
        while (msk == 0 && blk > 0) {
            blk--;
            pa = ix.blocks + ((blk << 2) + (c & 2));
            e = pa[0]; o = pa[1];
            msk = (u64)e.w | ((u64)o.w << 32);
        }
        if (msk == 0) msk = 1;
        int gb = 63 - __clzll((i64)msk);
        uint4 mine = (c & 1) ? o : e;
        u64 bits = quad_bits(mine);
        if ((bits >> gb) & 1ull) res = (i64)quad_rank<MEGA>(ix, mine, (blk << 6) | gb, c);
    }
    out[t] = res;
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
void sbwt_launch_rank(const SbwtIndexView &ix, const long long *d_pos, const char *d_sym, long long n,
                      long long *d_out, hipStream_t stream) {
    if (n <= 0) return;
    if (ix.n_mega > 1)
        hipLaunchKernelGGL(k_rank<true>, dim3(grid_for(n)), dim3(256), 0, stream, ix, d_pos, d_sym, (i64)n, d_out);
    else
        hipLaunchKernelGGL(k_rank<false>, dim3(grid_for(n)), dim3(256), 0, stream, ix, d_pos, d_sym, (i64)n, d_out);
}

void sbwt_launch_precalc(const SbwtIndexView &ix, int p, longlong2 *d_table, hipStream_t stream) {
    i64 n = 1ll << (2 * p);
    if (ix.n_mega > 1)
        hipLaunchKernelGGL(k_precalc<true>, dim3(grid_for(n)), dim3(256), 0, stream, ix, p, d_table);
    else
        hipLaunchKernelGGL(k_precalc<false>, dim3(grid_for(n)), dim3(256), 0, stream, ix, p, d_table);
}

void sbwt_launch_update_interval(const SbwtIndexView &ix, const char *d_bases, const long long *d_off, long long n,
                                 long long *d_first, long long *d_second, hipStream_t stream) {
    if (n <= 0) return;
    if (ix.n_mega > 1)
        hipLaunchKernelGGL(k_update_interval<true>, dim3(grid_for(n)), dim3(256), 0, stream, ix, d_bases, d_off,
                           (i64)n, d_first, d_second);
    else
        hipLaunchKernelGGL(k_update_interval<false>, dim3(grid_for(n)), dim3(256), 0, stream, ix, d_bases, d_off,
                           (i64)n, d_first, d_second);
}

void sbwt_launch_forward(const SbwtIndexView &ix, const long long *d_node, const char *d_sym, long long n,
                         long long *d_out, hipStream_t stream) {
    if (n <= 0) return;
    if (ix.n_mega > 1)
        hipLaunchKernelGGL(k_forward<true>, dim3(grid_for(n)), dim3(256), 0, stream, ix, d_node, d_sym, (i64)n, d_out);
    else
        hipLaunchKernelGGL(k_forward<false>, dim3(grid_for(n)), dim3(256), 0, stream, ix, d_node, d_sym, (i64)n, d_out);
}

// scratch: line_len[n_reads] + bsum[n_reads/1024 + 2]
