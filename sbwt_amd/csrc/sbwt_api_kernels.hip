// sbwt_api_kernels.hip -- the batched API neighbours of the search path: rank, prefix-table precalc,
// update_sbwt_interval, forward (one thread per item; see sbwt_search.hip for the search kernels).
#include "sbwt_kernels_common.h"

// ---------------------------------------------------------------------------------------------
// k_rank4 / k_rank1: SubsetMatrixRank::rank(pos, c) (SubsetMatrixRank.hh:31-37) for n independent (pos, sym) pairs
// ---------------------------------------------------------------------------------------------
template <bool MEGA>
__device__ __forceinline__ i64 rank_one(const SbwtIndexView &ix, unsigned b, i64 ps, const uint4 &q) {
    return is_ACGT(b) ? (i64)quad_rank<MEGA>(ix, q, ps, (int)dna_code(b)) - ix.C[dna_code(b)] : 0;   // non-ACGT: 0 (:36)
}

// Four consecutive pairs per lane: the (pos, sym) stream arrives as two 16-byte loads + one 4-byte load per lane
// (whole lines per wave), the four block quads are gathered with all four loads in flight before the first use,
// and the results leave as two 16-byte non-temporal stores.  One quad = the 16 bytes { bits, count } of symbol c in
// the 64-column block of pos (sbwt_device.h), so a rank is ONE gather whatever the symbol.  The scalar tail
// (and misaligned caller buffers) go through k_rank1.
template <bool MEGA>
__global__ void __launch_bounds__(256) k_rank4(SbwtIndexView ix, const i64 *__restrict__ pos,
                                               const char *__restrict__ sym, i64 n4, i64 *__restrict__ out) {
    const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
    if (t >= n4) return;
    typedef i64 i64x2 __attribute__((ext_vector_type(2)));
    const i64x2 p01 = __builtin_nontemporal_load(reinterpret_cast<const i64x2 *>(pos) + 2 * t);
    const i64x2 p23 = __builtin_nontemporal_load(reinterpret_cast<const i64x2 *>(pos) + 2 * t + 1);
    const unsigned s4 = __builtin_nontemporal_load(reinterpret_cast<const unsigned *>(sym) + t);
    const i64 ps[4] = {p01.x, p01.y, p23.x, p23.y};
    unsigned b[4];
    uint4 q[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
        b[u] = (s4 >> (8 * u)) & 0xFFu;
        // non-ACGT symbols gather quad 0 of their block (the value is not used): no divergent load
        q[u] = ix.blocks[((ps[u] >> 6) << 2) + (is_ACGT(b[u]) ? (i64)dna_code(b[u]) : 0)];
    }
    i64 r[4];
#pragma unroll
    for (int u = 0; u < 4; u++) r[u] = rank_one<MEGA>(ix, b[u], ps[u], q[u]);
    const i64x2 r01 = {r[0], r[1]}, r23 = {r[2], r[3]};
    __builtin_nontemporal_store(r01, reinterpret_cast<i64x2 *>(out) + 2 * t);
    __builtin_nontemporal_store(r23, reinterpret_cast<i64x2 *>(out) + 2 * t + 1);
}

template <bool MEGA>
__global__ void __launch_bounds__(256) k_rank1(SbwtIndexView ix, const i64 *__restrict__ pos,
                                               const char *__restrict__ sym, i64 first, i64 n, i64 *__restrict__ out) {
    i64 t = first + (i64)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    unsigned b = (unsigned char)sym[t];
    i64 ps = pos[t];
    uint4 q = ix.blocks[((ps >> 6) << 2) + (is_ACGT(b) ? (i64)dna_code(b) : 0)];
    out[t] = rank_one<MEGA>(ix, b, ps, q);
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
// k_partial_search: SBWT::partial_search (SBWT.hh:525-537), one lane per query: the longest prefix of the query
// that is a suffix of some column label, its interval and its length.  Every char is upper-cased before its
// update_sbwt_interval step (:529), so lower-case acgt match here (unlike in search()).
// ---------------------------------------------------------------------------------------------
template <bool MEGA>
__global__ void __launch_bounds__(256) k_partial_search(SbwtIndexView ix, const char *__restrict__ bases,
                                                        const i64 *__restrict__ off, i64 n, i64 *__restrict__ first,
                                                        i64 *__restrict__ second, i64 *__restrict__ matched) {
    i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    i64 l = 0, r = ix.n_nodes - 1;
    const i64 b0 = off[t], b1 = off[t + 1];
    i64 i = b0;
    for (; i < b1; i++) {
        unsigned b = (unsigned char)bases[i];
        if (b >= 'a' && b <= 'z') b -= 32;                  // toupper, C locale
        if (!is_ACGT(b)) break;                             // update_sbwt_interval gives {-1,-1} (SBWT.hh:427-428)
        int c = (int)dna_code(b);
        uint4 q1 = ix.blocks[((l >> 6) << 2) + c];
        uint4 q2 = ix.blocks[(((r + 1) >> 6) << 2) + c];
        i64 nl = (i64)quad_rank<MEGA>(ix, q1, l, c);
        i64 nr = (i64)quad_rank<MEGA>(ix, q2, r + 1, c) - 1;
        if (nl > nr) break;                                 // SBWT.hh:433 -> :531
        l = nl;
        r = nr;
    }
    first[t] = l;
    second[t] = r;
    matched[t] = i - b0;
}

// ---------------------------------------------------------------------------------------------
// k_get_kmer: SBWT::get_kmer / get_kmer_fast (SBWT.hh:700-746) for n columns: the k-mer (with leading '$' for
// dummy columns) spelled by walking the incoming edges backwards.  One backward step needs
// SubsetMatrixSelectSupport::select(j, c) (SubsetMatrixSelectSupport.hh:27-33) = the column holding the j-th
// one of row c; here it is found in the block counts themselves -- the last block whose count is <= the
// target (interpolated first guess, gallop, bisection), then the j-th set bit inside the block's 64 bits.
// ---------------------------------------------------------------------------------------------
// (block_count / select_in_row: sbwt_kernels_common.h -- the path order's build uses them too)
template <bool MEGA>
__global__ void __launch_bounds__(256) k_get_kmer(SbwtIndexView ix, const i64 *__restrict__ colex, i64 n,
                                                  char *__restrict__ out) {
    i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const int k = ix.k;
    i64 v = colex[t];
    char *buf = out + t * (i64)k;
    for (int i = 0; i < k; i++) {
        if (v == 0) { buf[k - 1 - i] = '$'; continue; }     // the root: SBWT.hh:703-704
        int c = 0;
        while (c + 1 < 4 && v >= ix.C[c + 1]) c++;          // :706-707
        buf[k - 1 - i] = "ACGT"[c];
        const i64 hi_c = (c < 3) ? ix.C[c + 1] : ix.n_nodes;        // an SBWT holds n_nodes - 1 ones: "C[4]" is n_nodes
        v = select_in_row<MEGA>(ix, c, v, hi_c - ix.C[c]);  // step backward (:711-721 / :739-743)
    }
}

// k_select: SubsetMatrixSelectSupport::select(j, c) (SubsetMatrixSelectSupport.hh:27-33): the column of the j-th one
// (1-based) of row c; non-ACGT -> 0.  row_ones[c] = ones in row c (the caller has them from the C array / the counts).
template <bool MEGA>
__global__ void __launch_bounds__(256) k_select(SbwtIndexView ix, const i64 *__restrict__ j, const char *__restrict__ sym,
                                                i64 n, i64 o0, i64 o1, i64 o2, i64 o3, i64 *__restrict__ out) {
    i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const unsigned b = (unsigned char)sym[t];
    i64 res = 0;
    if (is_ACGT(b)) {
        const int c = (int)dna_code(b);
        const i64 ones = c == 0 ? o0 : c == 1 ? o1 : c == 2 ? o2 : o3;
        res = select_in_row<MEGA>(ix, c, ix.C[c] + j[t] - 1, ones);
    }
    out[t] = res;
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
void sbwt_launch_rank(const SbwtIndexView &ix, const long long *d_pos, const char *d_sym, long long n,
                      long long *d_out, hipStream_t stream) {
    if (n <= 0) return;
    const bool mega = ix.n_mega > 1 || ix.force_mega;
    // the 4-per-lane kernel wants 16-byte aligned pos/out and a 4-byte aligned sym
    const bool al = (((uintptr_t)d_pos | (uintptr_t)d_out) & 15) == 0 && ((uintptr_t)d_sym & 3) == 0;
    const i64 n4 = al ? n / 4 : 0;
    if (n4 > 0) {
        if (mega) hipLaunchKernelGGL(k_rank4<true>, dim3(grid_for(n4)), dim3(256), 0, stream, ix, d_pos, d_sym, n4, d_out);
        else hipLaunchKernelGGL(k_rank4<false>, dim3(grid_for(n4)), dim3(256), 0, stream, ix, d_pos, d_sym, n4, d_out);
    }
    const i64 rest = n - 4 * n4;
    if (rest > 0) {
        if (mega) hipLaunchKernelGGL(k_rank1<true>, dim3(grid_for(rest)), dim3(256), 0, stream, ix, d_pos, d_sym, 4 * n4, (i64)n, d_out);
        else hipLaunchKernelGGL(k_rank1<false>, dim3(grid_for(rest)), dim3(256), 0, stream, ix, d_pos, d_sym, 4 * n4, (i64)n, d_out);
    }
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

void sbwt_launch_partial_search(const SbwtIndexView &ix, const char *d_bases, const long long *d_off, long long n,
                                long long *d_first, long long *d_second, long long *d_matched, hipStream_t stream) {
    if (n <= 0) return;
    if (ix.n_mega > 1)
        hipLaunchKernelGGL(k_partial_search<true>, dim3(grid_for(n)), dim3(256), 0, stream, ix, d_bases, d_off, (i64)n,
                           d_first, d_second, d_matched);
    else
        hipLaunchKernelGGL(k_partial_search<false>, dim3(grid_for(n)), dim3(256), 0, stream, ix, d_bases, d_off, (i64)n,
                           d_first, d_second, d_matched);
}

void sbwt_launch_get_kmer(const SbwtIndexView &ix, const long long *d_colex, long long n, char *d_out,
                          hipStream_t stream) {
    if (n <= 0) return;
    if (ix.n_mega > 1)
        hipLaunchKernelGGL(k_get_kmer<true>, dim3(grid_for(n)), dim3(256), 0, stream, ix, d_colex, (i64)n, d_out);
    else
        hipLaunchKernelGGL(k_get_kmer<false>, dim3(grid_for(n)), dim3(256), 0, stream, ix, d_colex, (i64)n, d_out);
}

void sbwt_launch_select(const SbwtIndexView &ix, const long long *d_j, const char *d_sym, long long n,
                        const long long row_ones[4], long long *d_out, hipStream_t stream) {
    if (n <= 0) return;
    if (ix.n_mega > 1 || ix.force_mega)
        hipLaunchKernelGGL(k_select<true>, dim3(grid_for(n)), dim3(256), 0, stream, ix, d_j, d_sym, (i64)n, row_ones[0],
                           row_ones[1], row_ones[2], row_ones[3], d_out);
    else
        hipLaunchKernelGGL(k_select<false>, dim3(grid_for(n)), dim3(256), 0, stream, ix, d_j, d_sym, (i64)n, row_ones[0],
                           row_ones[1], row_ones[2], row_ones[3], d_out);
}
