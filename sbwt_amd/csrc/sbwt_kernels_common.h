// sbwt_kernels_common.h -- device helpers shared by the HIP translation units of libsbwtgpu
// (sbwt_search.hip, sbwt_api_kernels.hip, sbwt_derived.hip, sbwt_format.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "sbwt_device.h"

typedef unsigned long long u64;
typedef long long i64;

#define SBWT_ERR_NOT_SINGLETON (-7)

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
// ASCII -> 0..3 for A,C,G,T (globals.hh:38-47); only meaningful when the validity bit is set.
__device__ __forceinline__ unsigned dna_code(unsigned b) { return ((b >> 1) & 3u) ^ ((b >> 2) & 1u); }
__device__ __forceinline__ bool is_ACGT(unsigned b) { return b == 'A' || b == 'C' || b == 'G' || b == 'T'; }
__device__ __forceinline__ u64 quad_bits(const uint4 &q) { return (u64)q.x | ((u64)q.y << 32); }
__device__ __forceinline__ u64 low_mask(int n) { return (1ull << n) - 1ull; }   // n in [0,63]

// wave-uniform values the compiler cannot prove uniform: pin them to scalar registers
__device__ __forceinline__ unsigned uniform32(unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ u64 uniform64(u64 v) {
    return (u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v) |
           ((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32);
}

// streaming (read-once / write-once) 16-byte accesses that should not displace the index in L2
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#ifdef SBWT_PLAIN_STORES        // experiments: the result stores without the non-temporal hint
#define SBWT_NT_STORE(v, p) (*(p) = (v))
#else
#define SBWT_NT_STORE(v, p) __builtin_nontemporal_store(v, p)
#endif
__device__ __forceinline__ void st_stream(i64 *p, i64 v) { SBWT_NT_STORE(v, p); }
// two consecutive results / columns at their natural (8-byte / 4-byte) alignment
typedef i64 i64x2_a8 __attribute__((ext_vector_type(2), aligned(8)));
typedef unsigned u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ void st_stream2(i64 *p, i64 a, i64 b) {
    i64x2_a8 v = {a, b};
    SBWT_NT_STORE(v, reinterpret_cast<i64x2_a8 *>(p));
}
// a result / two consecutive results at slot `at` of the caller's array: int64, or int32 when the call asked for that
// (o32 is wave-uniform: one launch writes one kind)
typedef int i32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ void st_res(i64 *out, i64 at, i64 v, int o32) {
    if (o32) SBWT_NT_STORE((int)v, reinterpret_cast<int *>(out) + at);
    else st_stream(out + at, v);
}
__device__ __forceinline__ void st_res2(i64 *out, i64 at, i64 a, i64 b, int o32) {
    if (o32) {
        i32x2_a4 v = {(int)a, (int)b};
        SBWT_NT_STORE(v, reinterpret_cast<i32x2_a4 *>(reinterpret_cast<int *>(out) + at));
    } else {
        st_stream2(out + at, a, b);
    }
}
__device__ __forceinline__ uint4 ld_stream(const uint4 *p) {
    u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}

// value of (C[c] + rank_c(pos)) from the quad of pos's block
template <bool MEGA>
__device__ __forceinline__ u64 quad_rank(const SbwtIndexView &ix, const uint4 &q, i64 pos, int c) {
    u64 v = (u64)q.z + (u64)__popcll(quad_bits(q) & low_mask((int)(pos & 63)));
    if (MEGA) v += ix.mega[(i64)c * ix.n_mega + (pos >> SBWT_MEGA_SHIFT)];
    return v;
}

// ---- select in a row of the matrix, from the block counts (SubsetMatrixSelectSupport.hh:27-33) ----
template <bool MEGA>
__device__ __forceinline__ u64 block_count(const SbwtIndexView &ix, i64 blk, int c, uint4 *q) {
    *q = ix.blocks[(blk << 2) + c];
    u64 v = (u64)q->z;
    if (MEGA) v += ix.mega[(i64)c * ix.n_mega + (blk >> (SBWT_MEGA_SHIFT - 6))];
    return v;
}

// the column whose row-c bit is the one numbered `target` - C[c] (0-based): the last block whose count (= C[c] + rank_c
// of its first column) is <= target, then the right set bit inside it.  target in [C[c], C[c] + ones of row c).
template <bool MEGA>
__device__ __forceinline__ i64 select_in_row(const SbwtIndexView &ix, int c, i64 target_i, i64 row_ones) {
    const u64 target = (u64)target_i;
    const i64 n_blocks = ix.n_nodes / 64 + 1;
    i64 g = row_ones > 0 ? (i64)((double)(target_i - ix.C[c]) / (double)row_ones * (double)n_blocks) : 0;
    if (g < 0) g = 0;
    if (g > n_blocks - 1) g = n_blocks - 1;
    uint4 q;
    i64 lo, hi;                                             // invariant: count(lo) <= target < count(hi) (hi may be n_blocks)
    if (block_count<MEGA>(ix, g, c, &q) <= target) {
        lo = g;
        i64 step = 1;
        hi = g + step;
        while (hi < n_blocks && block_count<MEGA>(ix, hi, c, &q) <= target) { lo = hi; step <<= 1; hi = lo + step; }
        if (hi > n_blocks) hi = n_blocks;
    } else {
        hi = g;
        i64 step = 1;
        lo = g - step;
        while (lo > 0 && block_count<MEGA>(ix, lo, c, &q) > target) { hi = lo; step <<= 1; lo = hi - step; }
        if (lo < 0) lo = 0;
    }
    while (hi - lo > 1) {
        const i64 mid = lo + ((hi - lo) >> 1);
        if (block_count<MEGA>(ix, mid, c, &q) <= target) lo = mid; else hi = mid;
    }
    const u64 cnt = block_count<MEGA>(ix, lo, c, &q);
    u64 bits = quad_bits(q);
    const int skip = (int)(target - cnt);                   // ones of this block before the wanted one
    for (int s = 0; s < skip; s++) bits &= bits - 1;
    return (lo << 6) + (bits ? (__ffsll((i64)bits) - 1) : 0);
}


// What a read meets at the path step it does not follow.  sA, sB: the step's two state bits (k_path_reencode): go = ~A | B,
// safe = A & B, only successor = ~A & B.  `mismatch`: the path goes on there with another char than the read's (else the
// path ends at the step).
#define PS_NONE 0               // nothing: the window was used up
#define PS_TRANS 1              // a lookup in the transition table (k_trans_insert): a successor, a verdict, or a free slot
#define PS_BRIDGE 2             // no other successor, and the step is substitution-safe: compare the next k-1 bases
#define PS_ABSENT 3             // no other successor: the streaming step's answer is -1 (SBWT.hh:572-575), no gather
__device__ __forceinline__ int path_stop_kind(bool mismatch, unsigned sA, unsigned sB, bool has_safe) {
    if (!mismatch || !sB) return PS_TRANS;              // the path ends, or the step has other successors
    return (sA && has_safe) ? PS_BRIDGE : PS_ABSENT;
}
#define SBWT_TRANS_NEG 0x100u          // transition entry: (t, c) has no successor ...
#define SBWT_TRANS_NEG_SAFE 0x200u     // ... and the step is substitution-safe for exactly this char
// bucket of a key in the sparse prefix table of n_buckets 32-byte buckets (any number: the hash is scaled, not masked),
// j buckets past its home
__device__ __forceinline__ unsigned sbwt_sp_bucket(u64 key, unsigned n_buckets, unsigned j) {
    unsigned b = (unsigned)__umul64hi(key * SBWT_SP_HASH, (u64)n_buckets) + j;
    while (b >= n_buckets) b -= n_buckets;
    return b;
}
// bucket of (first column of the 31-prefix's interval, rest of the k-mer) in the second-level table of n_entries 32-byte
// buckets, j buckets past its home
__device__ __forceinline__ unsigned sbwt_sp2_entry(unsigned origin, u64 key2, unsigned n_entries, unsigned j) {
    unsigned e = (unsigned)__umul64hi(sp2_hash(origin, key2), (u64)n_entries) + j;
    while (e >= n_entries) e -= n_entries;
    return e;
}
// slot of (path position t, char c) in the transition table of n_slots entries (any number: the hash is scaled, not masked),
// j slots past its home
__device__ __forceinline__ unsigned sbwt_trans_slot(unsigned t, unsigned c, unsigned n_slots, unsigned j) {
    const u64 h = ((u64)t * 4ull + c) * SBWT_SP_HASH;
    unsigned s = (unsigned)__umul64hi(h, (u64)n_slots) + j;
    while (s >= n_slots) s -= n_slots;
    return s;
}

// The fused route (sbwt_search_fused.hip) takes a batch when all reads have one length of 32 .. 32 * SBWT_FUSED_MAXG bases
// (k_check_uniform2 has filled in the header) that holds at least one k-mer; the kernels chained behind it ask the same
// question to know what is left for them.
#define SBWT_FUSED_MAXG 5
// 0: the batch is the general kernel's; 1: reads of one length (offsets by arithmetic); 2: reads of any lengths, the
// fused kernel fetches their offsets and hands on what is too long for it (at most one in eight of the reads the check
// kernel sampled: more, and its refills would mostly encode reads it cannot take).
// Round 4: a read of more than 32 * SBWT_FUSED_MAXG bases is taken as up to SBWT_FUSED_MAXP PIECES of that many bases that
// overlap by k-1 (piece pc holds the read's k-mers pc * kpp .. with kpp = SBWT_FUSED_MAXLEN - k + 1): a ticket is (read,
// piece), P tickets per read for the whole batch (sbwt_fused_pieces).  Exact: the fused kernel only walks pieces of
// upper-case ACGT, where a k-mer's result does not depend on what came before it (SBWT.hh:557-575; tests/test_large.hh:104-115).
#define SBWT_FUSED_MAXLEN (32 * SBWT_FUSED_MAXG)
#define SBWT_FUSED_MAXP 3
__device__ __forceinline__ long long sbwt_fused_limit(int P, int k) {       // the longest read P pieces hold
    const int kpp = SBWT_FUSED_MAXLEN - k + 1;
    return (P <= 1 || kpp < 64) ? (long long)SBWT_FUSED_MAXLEN : (long long)P * kpp + k - 1;
}
// rg_long packs three counters of 20 bits: sampled reads that need more than 1, 2, 3 pieces
// (rg_sample: the sample's size in bits 0-15, the most pieces a read may be taken as in bits 16-: "fused_pieces", 1 .. 3)
__device__ __forceinline__ int sbwt_fused_pieces(const SbwtWorkHeader *ws, int k) {
    const int sample = ws->rg_sample & 0xFFFF;
    int maxp = ws->rg_sample >> 16;
    if (maxp < 1) maxp = 1;
    if (maxp > SBWT_FUSED_MAXP) maxp = SBWT_FUSED_MAXP;
    if (ws->u_bad == 0) {
        for (int P = 1; P <= maxp; P++)
            if (ws->u_len <= sbwt_fused_limit(P, k)) return P;
        return 0;
    }
    if (sample <= 0) return 0;
    for (int P = 1; P <= maxp; P++) {
        const unsigned long long longer = (ws->rg_long >> (20 * (P - 1))) & 0xFFFFFull;
        if (longer * 8ull <= (unsigned long long)sample && (P == 1 || sbwt_fused_limit(P, k) > SBWT_FUSED_MAXLEN)) return P;
    }
    return 0;
}
__device__ __forceinline__ int sbwt_fused_mode(const SbwtWorkHeader *ws, int k) {
    const long long len = ws->u_len;
    const int pieces = sbwt_fused_pieces(ws, k);
    if (ws->u_bad == 0 && pieces > 0) return (len >= 32 && len >= k) ? 1 : 0;
    if (ws->u_bad != 0 && pieces > 0) return 2;
    // 3 (round 6): reads of any lengths, too many of them longer than three pieces -- the batch was cut into tickets of <= 160 bases
    // listed in a table (k_fused_tickets), every read of it is the fused kernel's
    return (ws->n_ftick > 0 && ws->ftick_over == 0) ? 3 : 0;
}
// may k_fused_tickets build the table?  (a ragged batch the fused kernel would otherwise decline; "fused_ragged" on)
__device__ __forceinline__ bool sbwt_fused_wants_table(const SbwtWorkHeader *ws, int k) {
    // (reads of one length of more than three pieces included: 1 kbp or 10 kbp reads)
    return (ws->rg_sample & 0xFFFF) > 0 && sbwt_fused_pieces(ws, k) == 0;
}
// the check kernels: thread t of the first blocks counts into the sample
#define SBWT_RG_SAMPLE 4096
__device__ __forceinline__ void fused_sample_of_wave(i64 t, i64 len, bool valid, i64 n_reads, int rg_enable, int k, SbwtWorkHeader *ws) {
    // rg_enable: bit 0 = batches of mixed read lengths ("fused_ragged"), bits 8- = the most pieces per read ("fused_pieces")
    if (t == 0) ws->rg_sample = ((rg_enable & 1) ? (int)(n_reads < SBWT_RG_SAMPLE ? n_reads : SBWT_RG_SAMPLE) : 0) | ((rg_enable >> 8) << 16);
    if (t - (threadIdx.x & 63) >= SBWT_RG_SAMPLE) return;          // (wave-uniform)
    const bool in = valid && t < SBWT_RG_SAMPLE;
    const u64 l1 = __ballot(in && len > sbwt_fused_limit(1, k)), l2 = __ballot(in && len > sbwt_fused_limit(2, k)),
              l3 = __ballot(in && len > sbwt_fused_limit(3, k));
    if (l1 && (threadIdx.x & 63) == 0)
        atomicAdd(&ws->rg_long, (unsigned long long)__popcll(l1) | ((unsigned long long)__popcll(l2) << 20) |
                                    ((unsigned long long)__popcll(l3) << 40));
}

// ---- long reads (SbwtPieceTab, sbwt_device.h) ----
// is read r one that its pieces answer?  (the search kernels skip it then.)  More than two zones' worth of k-mers -- and more
// than the fused kernel takes as its own pieces (at most 3 x 160 bases): a read that kernel may answer is never cut into
// zones as well, whose tickets the kernel behind it would walk over bases that nobody encoded.
__device__ __forceinline__ bool piece_read_is_cut(i64 m, int piece) {
    return m > 2 * (i64)piece && m > (i64)SBWT_FUSED_MAXP * SBWT_FUSED_MAXLEN;
}
// the check kernels call this with every lane of a wave (valid: the lane has a read): a long read reserves its zones, and
// the wave together notes { read, zone, zones of the read } for each of them (a genome as one read has 40 000)
__device__ __forceinline__ void piece_zones_of_wave(i64 r, i64 len, bool valid, int k, SbwtWorkHeader *ws, const SbwtPieceTab &pt) {
    const i64 m = len - k + 1;
    const bool isl = valid && pt.pairs != nullptr && piece_read_is_cut(m, pt.piece);
    u64 mask = __ballot(isl);
    if (!mask) return;
    const i64 nz = isl ? m / pt.piece : 0;
    const i64 base = isl ? (i64)atomicAdd(&ws->n_pieces, (unsigned long long)nz) : 0;
    const int lane = threadIdx.x & 63;
    while (mask) {
        const int src = __ffsll((i64)mask) - 1;
        mask &= mask - 1;
        const i64 rr = __shfl(r, src), bb = __shfl(base, src), nn = __shfl(nz, src);
        for (i64 j = lane; j < nn && bb + j < pt.cap; j += 64)
            pt.outs[bb + j] = make_uint4((unsigned)rr, (unsigned)((u64)rr >> 32), (unsigned)j, (unsigned)nn);
    }
}

// workgroups of 256 for n threads.  A launch holds fewer than 2^32 work-items (32-bit grid size in the dispatch packet; what is
// beyond is dropped silently -- it cost the sparse table of a 2.25 x 10^9-column index most of its entries before round 5): a
// caller with more slices its launch (SBWT_LAUNCH_SLICE, sbwt_derived.hip); one that does not is stopped here.
static inline unsigned grid_for(i64 n) {
    if (n + 255 >= ((i64)1 << 32)) {
        fprintf(stderr, "sbwtgpu: internal error: a launch of %lld threads exceeds the 2^32 work-items of one dispatch\n", (long long)n);
        abort();
    }
    return (unsigned)((n + 255) / 256);
}
