// sbwt_derived.hip -- structures derived from the index on the device when the image is created: suffix-group
// marks for indexes without streaming support, the sparse prefix table (+ second level), the probe filter, the
// path order with its packed chars, substitution-safe bits and transition table (DESIGN.md sections 2-3).
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include "sbwt_kernels_common.h"
#include "sbwt_scan.h"

// ---------------------------------------------------------------------------------------------
// Suffix-group marks derived on the device (mark_suffix_groups, src/suffix_group_optimization.cpp:66-121)
// for indexes saved with --no-streaming-support: the marks are a function of the four columns, so the
// per-k-mer search loop can use streaming steps internally (with the raw-character validation of
// SBWT::search) although the index carries no suffix_group_starts vector.  k-1 rounds of
//   mark:       column i starts a group in this round iff its label differs from column i-1's
//   propagate:  every edge (i --c--> C[c] + rank_c(i)) hands column i's label to its target
// starting from label(v) = the symbol whose C-array range holds v ('$' for the root).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_sg_init(SbwtIndexView ix, unsigned char *__restrict__ last) {
    i64 v = (i64)blockIdx.x * 256 + threadIdx.x;
    if (v >= ix.n_nodes) return;
    last[v] = (unsigned char)((v >= ix.C[0]) + (v >= ix.C[1]) + (v >= ix.C[2]) + (v >= ix.C[3]));
}
__global__ void __launch_bounds__(256) k_sg_mark(const unsigned char *__restrict__ last, i64 n, u64 *__restrict__ acc) {
    i64 w = (i64)blockIdx.x * 256 + threadIdx.x;      // one 64-column word per thread
    if (w * 64 >= n) return;
    u64 m = 0;
    for (int t = 0; t < 64; t++) {
        i64 i = w * 64 + t;
        if (i < n && (i == 0 || last[i] != last[i - 1])) m |= 1ull << t;
    }
    acc[w] |= m;
}
template <bool MEGA>
__global__ void __launch_bounds__(256) k_sg_propagate(SbwtIndexView ix, const unsigned char *__restrict__ last,
                                                      unsigned char *__restrict__ next) {
    i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
    if (i >= ix.n_nodes) return;
    if (i == 0) next[0] = 0;                           // nothing points at the root: '$'
    const unsigned char lab = last[i];
    const uint4 *blk = ix.blocks + ((i >> 6) << 2);
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const uint4 q = blk[c];
        if ((quad_bits(q) >> (i & 63)) & 1ull) next[(i64)quad_rank<MEGA>(ix, q, i, c)] = lab;
    }
}
__global__ void __launch_bounds__(256) k_sg_patch(uint4 *__restrict__ blocks, const u64 *__restrict__ acc, i64 n_blocks) {
    i64 b = (i64)blockIdx.x * 256 + threadIdx.x;
    if (b >= n_blocks) return;
    const u64 s = acc[b];
    for (int c = 0; c < 4; c++) blocks[b * 4 + c].w = (c & 1) ? (unsigned)(s >> 32) : (unsigned)s;
}

// ---------------------------------------------------------------------------------------------
// Sparse prefix table: kmer_prefix_precalc (SBWT.hh:40,616-645) at depth p_sparse (default 20, the
// deepest the reference allows), holding only the prefixes whose interval is not empty -- at that depth
// about one entry per k-mer instead of 4^20.  Built by expanding the non-empty entries of the dense
// device table one character at a time (each expansion is the interval update of SBWT.hh:430-431) and
// hashing the survivors into buckets of two entries; a bucket that a key had to skip carries an overflow
// flag, so a lookup that meets a bucket without the flag knows the prefix is absent.
// ---------------------------------------------------------------------------------------------
// Output slot of a compacting append: ONE atomic per 256-thread workgroup (every expansion level appends ~n_nodes items
// to one counter, and a single address takes only ~10^8 atomics per second: one per lane or even one per wave made the
// 17 levels of a 142 M-column index cost 1.4 s).  Returns ~0 for threads that append nothing.  Every thread of the
// workgroup must call it (it synchronises the workgroup).
__device__ __forceinline__ u64 block_append_slot(u64 *counter, bool ok) {
    __shared__ unsigned wcount[4];
    __shared__ u64 wbase;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const u64 m = __ballot(ok);
    if (lane == 0) wcount[wv] = (unsigned)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned total = wcount[0] + wcount[1] + wcount[2] + wcount[3];
        wbase = total ? atomicAdd(counter, (u64)total) : 0ull;
    }
    __syncthreads();
    unsigned before = 0;
    for (int w = 0; w < wv; w++) before += wcount[w];
    const u64 slot = wbase + before + (u64)__popcll(m & low_mask(lane));
    __syncthreads();                                    // (the shared words are reused by the caller's next append)
    return ok ? slot : ~0ull;
}

// Output slots of a compacting append of 0 .. 4 items per thread: one atomic per workgroup, like block_append_slot.
__device__ __forceinline__ u64 block_append_n(u64 *counter, unsigned cnt) {
    __shared__ unsigned wtot[4];
    __shared__ u64 wbase_n;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned v = __shfl_up(incl, o); if (lane >= o) incl += v; }
    if (lane == 63) wtot[wv] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned total = wtot[0] + wtot[1] + wtot[2] + wtot[3];
        wbase_n = total ? atomicAdd(counter, (u64)total) : 0ull;
    }
    __syncthreads();
    unsigned before = 0;
    for (int w = 0; w < wv; w++) before += wtot[w];
    const u64 slot = wbase_n + before + (u64)(incl - cnt);
    __syncthreads();
    return slot;
}
struct SpItem { u64 key; i64 l; i64 r; };

__global__ void __launch_bounds__(256) k_sp_collect(const longlong2 *__restrict__ ptab, u64 n_entries,
                                                    SpItem *__restrict__ out, u64 *counter) {
    u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
    longlong2 e = make_longlong2(-1, -1);
    if (t < n_entries) e = ptab[t];
    const u64 slot = block_append_slot(counter, e.x >= 0);
    if (slot != ~0ull) out[slot] = SpItem{t, e.x, e.y};
}
// ---- second level (31 < k <= 63): the 31-prefix's interval (named by its first column) + the remaining bases ----
struct SpItem2 { u64 key2; unsigned origin, l, r, pad; };
static_assert(sizeof(SpItem2) == sizeof(SpItem), "the two item lists share their buffers");
__global__ void __launch_bounds__(256) k_sp2_seed(SpItem *items, const u64 *n) {     // in place: depth-31 items
    u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
    if (t >= *n) return;
    const SpItem it = items[t];
    reinterpret_cast<SpItem2 *>(items)[t] = SpItem2{0ull, (unsigned)it.l, (unsigned)it.l, (unsigned)it.r, 0u};
}
__global__ void __launch_bounds__(256) k_sp2_expand(SbwtIndexView ix, const SpItem2 *__restrict__ in, const u64 *n_in,
                                                    int d2, SpItem2 *__restrict__ out, u64 *n_out, u64 t0) {
    // (one thread per item and all four chars, like k_sp_expand)
    u64 t = t0 + (u64)blockIdx.x * 256 + threadIdx.x;
    const bool live = t < *n_in;
    SpItem2 it = SpItem2{0ull, 0u, 0u, 0u, 0u};
    if (live) it = in[t];
    i64 L[4], R[4];
    unsigned cnt = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) { L[c] = 1; R[c] = 0; }
    if (live) {
        const i64 bl = ((i64)it.l >> 6) << 2, br = (((i64)it.r + 1) >> 6) << 2;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint4 q1 = ix.blocks[bl + c];
            const uint4 q2 = (br == bl) ? q1 : ix.blocks[br + c];
            L[c] = (i64)quad_rank<false>(ix, q1, (i64)it.l, c);
            R[c] = (i64)quad_rank<false>(ix, q2, (i64)it.r + 1, c) - 1;
            cnt += (L[c] <= R[c]) ? 1u : 0u;
        }
    }
    u64 slot = block_append_n(n_out, cnt);
#pragma unroll
    for (int c = 0; c < 4; c++)
        if (L[c] <= R[c]) out[slot++] = SpItem2{it.key2 | ((u64)c << (2 * d2)), it.origin, (unsigned)L[c], (unsigned)R[c], 0u};
}
__global__ void __launch_bounds__(256) k_sp2_insert(const SpItem2 *__restrict__ items, const u64 *n, uint4 *table,
                                                    unsigned n_entries, const unsigned *__restrict__ pos, int *wide_flag, int big) {
    u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
    if (t >= *n) return;
    const SpItem2 it = items[t];
    if (it.l != it.r) *wide_flag = 1;                  // a k-mer's interval is one column in an SBWT
    // payload: the k-mer's path position when there is a path order (its column is col[position]), else its column
    // (big layout: position + 1, every bit of the word is the position's; no overflow flag either -- the origin word is a
    // full 32-bit column -- so a lookup goes on past every full bucket)
    const unsigned payload = big ? (pos ? pos[it.l] : it.l) + 1u : (SBWT_SP2_USED | (pos ? pos[it.l] : it.l));
    size_t bkt = sbwt_sp2_entry(it.origin, it.key2, n_entries, 0u);
    for (;;) {
        unsigned *b0 = reinterpret_cast<unsigned *>(&table[2 * bkt]);
        for (int e = 0; e < 2; e++) {
            unsigned *w = b0 + 4 * e;
            if (atomicCAS(&w[3], 0u, payload) == 0u) {       // keys are distinct: an empty entry is simply taken
                w[0] = (unsigned)it.key2;
                w[1] = (unsigned)(it.key2 >> 32);
                atomicOr(&w[2], it.origin);                  // (entry 0's word also carries the overflow flag)
                return;
            }
        }
        if (!big) atomicOr(&b0[2], SBWT_SP2_OVERFLOW);     // both entries taken: mark and move on
        bkt = bkt + 1 < n_entries ? bkt + 1 : 0;
    }
}
__global__ void __launch_bounds__(256) k_sp_wide(const SpItem *__restrict__ items, const u64 *n, int *flag) {
    u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
    if (t < *n && items[t].l != items[t].r) *flag = 1;
}
// (A launch holds fewer than 2^32 work-items -- the dispatch packet's grid size is 32 bits, and more are dropped SILENTLY: a table
// of 5.6 x 10^9 entries, or four threads for each of 2.25 x 10^9 items, goes in slices of 2^31 threads that start at t0; round 5)
#define SBWT_LAUNCH_SLICE ((u64)1 << 31)
__global__ void __launch_bounds__(256) k_sp_clear(uint4 *table, u64 n_entries, u64 t0) {
    u64 t = t0 + (u64)blockIdx.x * 256 + threadIdx.x;
    if (t < n_entries) table[t] = make_uint4(0u, (unsigned)(SBWT_SP_EMPTY >> 32), 0u, 0u);
}
// One level of the expansion: every prefix of depth `depth` is carried on by each of the four chars whose interval is not empty.
// ONE thread per prefix (round 6; it was four, one per char): the four quads of a block are one 64-byte line -- a prefix whose
// interval is one column, which is nearly every prefix from depth log4(n) + 2 on, costs one line and one thread instead of
// four threads that each synchronise the workgroup three times for their slot (k_sp_expand was 44 % of an image's build time).
template <bool MEGA>
__global__ void __launch_bounds__(256) k_sp_expand(SbwtIndexView ix, const SpItem *__restrict__ in, const u64 *n_in,
                                                   int depth, SpItem *__restrict__ out, u64 *n_out, u64 t0) {
    u64 t = t0 + (u64)blockIdx.x * 256 + threadIdx.x;
    const bool live = t < *n_in;
    SpItem it = SpItem{0, 0, 0};
    if (live) it = in[t];
    i64 L[4], R[4];
    unsigned cnt = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) { L[c] = 1; R[c] = 0; }
    if (live) {
        const i64 bl = (it.l >> 6) << 2, br = ((it.r + 1) >> 6) << 2;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint4 q1 = ix.blocks[bl + c];
            const uint4 q2 = (br == bl) ? q1 : ix.blocks[br + c];
            L[c] = (i64)quad_rank<MEGA>(ix, q1, it.l, c);
            R[c] = (i64)quad_rank<MEGA>(ix, q2, it.r + 1, c) - 1;
            cnt += (L[c] <= R[c]) ? 1u : 0u;
        }
    }
    u64 slot = block_append_n(n_out, cnt);
#pragma unroll
    for (int c = 0; c < 4; c++)
        if (L[c] <= R[c]) out[slot++] = SpItem{it.key | ((u64)c << (2 * depth)), L[c], R[c]};   // char `depth` of the prefix is c
}
// Probe filter: a blocked Bloom filter (128-bit blocks, two bits per key) over every p_filter-mer the index
// holds.  A certificate probe asks "is this window absent?": a clear bit answers yes in one gather; two set bits
// answer "perhaps not", and the exact walk through the dense table decides.
__global__ void __launch_bounds__(256) k_pf_insert(const SpItem *__restrict__ items, const u64 *n, unsigned *filter,
                                                   int log2f) {
    u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
    if (t >= *n) return;
    const u64 h = sbwt_pf_hash(items[t].key);
    unsigned *blk = filter + ((h >> (64 - log2f)) << 2);
    const unsigned hb = sbwt_pf_bits(h);
    const unsigned b1 = hb & 127u, b2 = (hb >> 7) & 127u;
    atomicOr(&blk[b1 >> 5], 1u << (b1 & 31u));
    atomicOr(&blk[b2 >> 5], 1u << (b2 & 31u));
}
// pos != nullptr: the items are whole k-mers (one column each); the second payload word is the column's path position
__global__ void __launch_bounds__(256) k_sp_insert(const SpItem *__restrict__ items, const u64 *n, uint4 *table,
                                                   unsigned n_buckets, const unsigned *__restrict__ pos,
                                                   const unsigned *__restrict__ upos) {
    u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
    if (t >= *n) return;
    const SpItem it = items[t];
    size_t bkt = sbwt_sp_bucket(it.key, n_buckets, 0u);
    for (;;) {
        for (int e = 0; e < 2; e++) {
            u64 *word = reinterpret_cast<u64 *>(&table[2 * bkt + e]);
            u64 old = atomicCAS(word, SBWT_SP_EMPTY, it.key);
            if (old == SBWT_SP_EMPTY) {
                unsigned *pay = reinterpret_cast<unsigned *>(word) + 2;
                pay[0] = (unsigned)it.l;
                // depth k: the column's path position.  Depth < k: the interval's width -- or, for a prefix with ONE column when
                // the image has a second level and a path order, that column's position (a seed for an alignment, no pos[] gather)
                pay[1] = pos ? pos[it.l] : (upos && it.r == it.l) ? (SBWT_SP_UNIQ | upos[it.l]) : (unsigned)(it.r - it.l);
                return;
            }
        }
        atomicOr(reinterpret_cast<u64 *>(&table[2 * bkt]), SBWT_SP_OVERFLOW);   // both entries taken: mark and move on
        bkt = bkt + 1 < n_buckets ? bkt + 1 : 0;
    }
}

// ---------------------------------------------------------------------------------------------
// Path order: the streaming steps of SBWT::streaming_search (SBWT.hh:562-575), precomputed.
//
// A streaming step maps (column v, char c) to the column of the k-mer that follows, and that map does
// not depend on the query.  Give every column ONE outgoing step (a char its suffix group offers) and
// every column at most one incoming one: the columns fall apart into vertex-disjoint paths -- in a
// genome, the unitigs strung together through their branch points.  Number the columns along the
// paths: t = pos[v], v = col[t].  A query that sits on column col[t] and whose next base equals the
// path's char at t sits on col[t+1] next, and so on: while the read follows the path, its answers are
// the CONTIGUOUS run col[t+1], col[t+2], ... and checking that it does is a 2-bit compare against the
// path's packed chars, 32 bases at a time.  The random 64-byte block gather per k-mer becomes
// sequential 4-byte reads; the blocks are only touched where a read leaves its path (a branch taken
// the other way, a substitution, the end of a path), by the generic step.
//
//   k_path_succ   per column: the group's start, the chars it offers, the char this member takes
//                 (member r of a group with d chars takes the (r mod d)-th, so the members of a
//                 bubble fan out), the target column; claims the target with atomicMin
//   k_path_keep   a step survives if its source won the claim
//   k_path_jump   pointer doubling over the predecessor links: head of the path + distance from it
//                 (and the minimum column seen, which names a cycle's cut point)
//   k_path_cut    columns that never reached a head lie on a cycle: cut it at its minimum
//   k_path_len / scan / k_path_place   paths laid out head by head: pos, col and the packed chars
// ---------------------------------------------------------------------------------------------
#define PATH_NONE 0xFFFFFFFFu
__global__ void __launch_bounds__(256) k_path_fill(unsigned *a, i64 n, unsigned v) {
    i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
    if (t < n) a[t] = v;
}
__global__ void __launch_bounds__(256) k_path_succ(SbwtIndexView ix, unsigned *__restrict__ succ,
                                                   unsigned char *__restrict__ sch, unsigned *prv) {
    const i64 v = (i64)blockIdx.x * 256 + threadIdx.x;
    if (v >= ix.n_nodes) return;
    i64 blk = v >> 6;
    u64 msk = ((u64)ix.blocks[blk * 4].w | ((u64)ix.blocks[blk * 4 + 1].w << 32)) & ((2ull << (int)(v & 63)) - 1ull);
    while (msk == 0 && blk > 0) {
        blk--;
        msk = (u64)ix.blocks[blk * 4].w | ((u64)ix.blocks[blk * 4 + 1].w << 32);
    }
    if (msk == 0) msk = 1;
    const int gb = 63 - __clzll((i64)msk);
    const i64 g = (blk << 6) | gb;
    uint4 q[4];
    int deg = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        q[c] = ix.blocks[blk * 4 + c];
        deg += (int)((quad_bits(q[c]) >> gb) & 1ull);
    }
    if (deg == 0) { succ[v] = PATH_NONE; sch[v] = 0; return; }
    int want = (int)((v - g) % deg), pick = 0;
    unsigned target = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const u64 bits = quad_bits(q[c]);
        if ((bits >> gb) & 1ull) {
            if (want == 0) { pick = c; target = q[c].z + (unsigned)__popcll(bits & low_mask(gb)); }
            want--;
        }
    }
    succ[v] = target;
    sch[v] = (unsigned char)pick;
    atomicMin(&prv[target], (unsigned)v);
}
// Which successor does a column take, which predecessor does a column keep?  Any choice gives a valid path order; a good
// one keeps the paths on the graph's CORE.  In a pan-genome nearly every column of the shared sequence has a variant
// branching off somewhere within k steps, and a rule that is blind to that sends the path into every second private
// bubble: the reads (most of which follow the majority) then leave their path every few k-mers (config 3: 21
// transitions per read).  The core is recognisable from the matrix alone: it has branch points AHEAD (F: columns
// with two or more successors met within the next D steps, following the best successor) and merges BEHIND (G: columns
// that join another within the last D steps); a private bubble has neither.  D rounds of
//     F'[v] = [v branches] + max over successors F[s]        G'[s] = [s's predecessors are a group of >= 2] + max over them G[p]
// (streaming passes: consecutive columns share blocks, successors by one char are consecutive), then within every suffix
// group the members ranked by G take the successors ranked by F, best to best.  With all weights equal this is the
// rule above (member r takes char r, the smallest claimant wins).
struct PathGroup {
    i64 g, gend;                    // the suffix group [g, gend) of the column
    int deg;                        // chars the group offers
    unsigned target[4];             // successor column by char (PATH_NONE: none)
};
__device__ __forceinline__ PathGroup path_group(const SbwtIndexView &ix, i64 v) {
    PathGroup pg;
    i64 blk = v >> 6;
    const u64 here = (u64)ix.blocks[blk * 4].w | ((u64)ix.blocks[blk * 4 + 1].w << 32);
    u64 msk = here & ((2ull << (int)(v & 63)) - 1ull);
    i64 gblk = blk;
    while (msk == 0 && gblk > 0) {
        gblk--;
        msk = (u64)ix.blocks[gblk * 4].w | ((u64)ix.blocks[gblk * 4 + 1].w << 32);
    }
    if (msk == 0) msk = 1;
    const int gb = 63 - __clzll((i64)msk);
    pg.g = (gblk << 6) | gb;
    // the next group start after v.  (A suffix group has at most five members -- one per first char and '$'; the bound
    // below only keeps the loops short on marks that are not an SBWT's.)
    u64 nxt = here & ~((2ull << (int)(v & 63)) - 1ull);
    i64 eblk = blk;
    const i64 nblk = ix.n_nodes / 64 + 1;
    if (nxt == 0 && eblk + 1 < nblk) {
        eblk++;
        nxt = (u64)ix.blocks[eblk * 4].w | ((u64)ix.blocks[eblk * 4 + 1].w << 32);
    }
    pg.gend = nxt ? ((eblk << 6) | (i64)(__ffsll((i64)nxt) - 1)) : ix.n_nodes;
    if (pg.gend > pg.g + 8) pg.gend = pg.g + 8;
    if (pg.gend > ix.n_nodes) pg.gend = ix.n_nodes;
    if (pg.gend <= v) pg.gend = v + 1;
    pg.deg = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const uint4 q = ix.blocks[gblk * 4 + c];
        const u64 bits = quad_bits(q);
        const bool has = ((bits >> gb) & 1ull) != 0;
        pg.target[c] = has ? (q.z + (unsigned)__popcll(bits & low_mask(gb))) : PATH_NONE;
        pg.deg += has;
    }
    return pg;
}
__global__ void __launch_bounds__(256) k_path_weigh(SbwtIndexView ix, const unsigned char *__restrict__ Fin,
                                                    unsigned char *__restrict__ Fout, const unsigned char *__restrict__ Gin,
                                                    unsigned char *__restrict__ Gout) {
    const i64 v = (i64)blockIdx.x * 256 + threadIdx.x;
    if (v >= ix.n_nodes) return;
    const PathGroup pg = path_group(ix, v);
    unsigned f = 0;
#pragma unroll
    for (int c = 0; c < 4; c++)
        if (pg.target[c] != PATH_NONE) { const unsigned x = Fin[pg.target[c]]; f = x > f ? x : f; }
    f += (pg.deg >= 2);
    Fout[v] = (unsigned char)(f > 255u ? 255u : f);
    if (v == 0) Gout[0] = 0;                            // the root has no incoming edge
    if (v == pg.g) {                                    // the group's first column speaks for its successors
        unsigned gm = 0;
        for (i64 p = pg.g; p < pg.gend; p++) { const unsigned x = Gin[p]; gm = x > gm ? x : gm; }
        gm += (pg.gend - pg.g >= 2);
        const unsigned char out = (unsigned char)(gm > 255u ? 255u : gm);
#pragma unroll
        for (int c = 0; c < 4; c++)
            if (pg.target[c] != PATH_NONE) Gout[pg.target[c]] = out;
    }
}
__global__ void __launch_bounds__(256) k_path_succ_weighted(SbwtIndexView ix, const unsigned char *__restrict__ F,
                                                            const unsigned char *__restrict__ G, unsigned *__restrict__ succ,
                                                            unsigned char *__restrict__ sch, unsigned *__restrict__ prv) {
    const i64 v = (i64)blockIdx.x * 256 + threadIdx.x;
    if (v >= ix.n_nodes) return;
    const PathGroup pg = path_group(ix, v);
    succ[v] = PATH_NONE;
    sch[v] = 0;
    if (pg.deg == 0) return;
    // this member's rank among the group's members: larger G first, then the smaller column
    const unsigned gv = G[v];
    int rk = 0;
    for (i64 p = pg.g; p < pg.gend; p++) {
        const unsigned gp = G[p];
        rk += (gp > gv) || (gp == gv && p < v);
    }
    if (rk >= pg.deg) return;                           // more members than chars: this one's path ends here
    // the successor of that rank: larger F first, then the smaller char
    unsigned fw[4];
#pragma unroll
    for (int c = 0; c < 4; c++) fw[c] = (pg.target[c] != PATH_NONE) ? (unsigned)F[pg.target[c]] : 0u;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        if (pg.target[c] == PATH_NONE) continue;
        int before = 0;
#pragma unroll
        for (int d = 0; d < 4; d++)
            before += (pg.target[d] != PATH_NONE) && (fw[d] > fw[c] || (fw[d] == fw[c] && d < c));
        if (before == rk) {
            succ[v] = pg.target[c];
            sch[v] = (unsigned char)c;
            prv[pg.target[c]] = (unsigned)v;            // the only claimant: ranks within a group are distinct
        }
    }
}
__global__ void __launch_bounds__(256) k_path_keep(i64 n, unsigned *__restrict__ succ, const unsigned *__restrict__ prv,
                                                   uint4 *__restrict__ jb) {
    const i64 v = (i64)blockIdx.x * 256 + threadIdx.x;
    if (v >= n) return;
    const unsigned sv = succ[v];
    if (sv != PATH_NONE && prv[sv] != (unsigned)v) succ[v] = PATH_NONE;
    if (!jb) return;                                    // (the splitter ranking below makes its own entries)
    const unsigned pv = prv[v];
    // { jump (heads point at themselves), distance to it, smallest column seen, - }: one 16-byte gather per doubling step
    jb[v] = make_uint4((pv == PATH_NONE) ? (unsigned)v : pv, (pv == PATH_NONE) ? 0u : 1u, (unsigned)v, 0u);
}
__global__ void __launch_bounds__(256) k_path_jump(i64 n, const uint4 *__restrict__ in, uint4 *__restrict__ out, int *moved) {
    const i64 v = (i64)blockIdx.x * 256 + threadIdx.x;
    if (v >= n) return;
    const uint4 a = in[v], b = in[a.x];
    out[v] = make_uint4(b.x, a.y + b.y, a.z < b.z ? a.z : b.z, 0u);
    if (b.x != a.x && moved) *moved = 1;                // (every writer stores the same value)
}
// ---- the same { head, distance, smallest column } per column by SPLITTERS instead of doubling over every column (round 6).
// Doubling gathers 16 bytes per column and round: log2(n) + 1 rounds when the graph is one long path (10^9 columns: 31 rounds,
// 0.85 s of a 2.9 s image; 2.25 x 10^9 with one cycle to cut: 65 rounds, 4.0 s of 11.9).  Here the heads and one column in 64
// (by a hash of its number) are splitters: every splitter walks to the next one (64 steps on average) and leaves there how far
// it came and the smallest column it saw; the SPLITTERS are ranked by doubling (n / 64 of them); every splitter walks its
// stretch again and writes the entries.  Two random reads and one random write per column instead of 31 of each.  The
// entries of columns on open paths are those of the doubling ({ head, distance }; the third word, used for cycles only, is the
// cycle's smallest column on cycles as there), so the path order is the same bit for bit (tests: test_gpu_derived).
// A cycle without a splitter on it is found at the end (entries still unset) and walked by each of its columns; a walk of more
// than RANK_LIMIT steps -- (63/64)^65536: a hash that fails the graph -- raises the flag and the doubling runs instead, as it
// does for graphs of very short paths (more splitters than n / 8: the doubling is done after a few rounds there).
#define RANK_SENT 0xFFFFFFFFu
#define RANK_LIMIT (1 << 16)
__device__ __forceinline__ bool rank_sampled(unsigned v) {
    unsigned h = v * 0x9E3779B1u;
    h ^= h >> 15; h *= 0x85EBCA77u; h ^= h >> 13;
    return (h & 63u) == 0u;
}
__global__ void __launch_bounds__(256) k_rank_list(i64 n, const unsigned *__restrict__ prv, unsigned *__restrict__ S, u64 *m,
                                                   uint4 *__restrict__ ja, uint4 *__restrict__ jb) {
    const i64 v = (i64)blockIdx.x * 256 + threadIdx.x;
    const bool in = v < n;
    const bool head = in && prv[v] == PATH_NONE;
    const bool spl = in && (head || rank_sampled((unsigned)v));
    const u64 slot = block_append_n(m, spl ? 1u : 0u);
    if (spl) S[slot] = (unsigned)v;
    if (in) {
        const uint4 e = head ? make_uint4((unsigned)v, 0u, (unsigned)v, 0u) : make_uint4(RANK_SENT, 0u, 0u, 0u);
        ja[v] = e;
        jb[v] = e;
    }
}
__global__ void __launch_bounds__(256) k_rank_walk1(const u64 *__restrict__ m, const unsigned *__restrict__ S,
                                                    const unsigned *__restrict__ succ, uint4 *ja, int *flag, unsigned limit) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= *m) return;
    const unsigned s = S[i];
    unsigned u = succ[s], d = 1, mn = 0xFFFFFFFFu;
    while (u != PATH_NONE && !rank_sampled(u)) {
        mn = u < mn ? u : mn;
        u = succ[u];
        if (++d > limit) { *flag = 2; return; }
    }
    if (u != PATH_NONE) ja[u] = make_uint4(s, d, u < mn ? u : mn, 0u);    // (a column has one predecessor: one writer)
}
__global__ void __launch_bounds__(256) k_rank_jump(const u64 *__restrict__ m, const unsigned *__restrict__ S,
                                                   const uint4 *__restrict__ in, uint4 *__restrict__ out) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= *m) return;
    const unsigned v = S[i];
    const uint4 a = in[v];
    if (a.x == RANK_SENT) { out[v] = a; return; }       // (never written: a walk gave up, the flag is up)
    const uint4 b = in[a.x];
    if (b.x == RANK_SENT) { out[v] = a; return; }
    out[v] = make_uint4(b.x, a.y + b.y, a.z < b.z ? a.z : b.z, 0u);
}
__global__ void __launch_bounds__(256) k_rank_walk2(const u64 *__restrict__ m, const unsigned *__restrict__ S,
                                                    const unsigned *__restrict__ succ, uint4 *fin, int *flag, unsigned limit) {
    const u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i >= *m) return;
    const unsigned s = S[i];
    const uint4 e = fin[s];
    if (e.x == RANK_SENT) return;
    unsigned u = succ[s], d = 1;
    while (u != PATH_NONE && !rank_sampled(u)) {
        fin[u] = make_uint4(e.x, e.y + d, e.z, 0u);
        u = succ[u];
        if (++d > limit) { *flag = 2; return; }
    }
}
__global__ void __launch_bounds__(256) k_rank_leftover(i64 n, const unsigned *__restrict__ succ, uint4 *fin, int *flag, unsigned limit) {
    const i64 v = (i64)blockIdx.x * 256 + threadIdx.x;
    if (v >= n) return;
    if (fin[v].x != RANK_SENT) return;
    unsigned u = succ[v], mn = (unsigned)v, steps = 0;  // on a cycle no splitter lies on
    while (u != (unsigned)v) {
        if (u == PATH_NONE || ++steps > limit) { *flag = 2; return; }
        mn = u < mn ? u : mn;
        u = succ[u];
    }
    fin[v] = make_uint4((unsigned)v, 0u, mn, 0u);
}
__global__ void __launch_bounds__(256) k_path_cut(i64 n, const uint4 *__restrict__ jb, unsigned *prv, unsigned *succ, int *flag) {
    const i64 v = (i64)blockIdx.x * 256 + threadIdx.x;
    if (v >= n) return;
    const uint4 me = jb[v];
    // reached a head?  (a cycle's cut point may be cut by its own thread while others look: they then
    // return here, and the flag is raised by the cutting thread)
    if (prv[me.x] == PATH_NONE) return;
    if (me.z == (unsigned)v) {                          // the cycle's smallest column becomes a head
        const unsigned pv = prv[v];
        if (pv != PATH_NONE) { succ[pv] = PATH_NONE; prv[v] = PATH_NONE; }
        *flag = 1;
    }
}
// (the last column of a path writes its length: one writer per path -- an atomicMax by every column serialises on a
// genome that is one long path, 3.4 s for 3 x 10^8 columns)
__global__ void __launch_bounds__(256) k_path_len(i64 n, const uint4 *__restrict__ jb, const unsigned *__restrict__ succ,
                                                  unsigned long long *len) {
    const i64 v = (i64)blockIdx.x * 256 + threadIdx.x;
    if (v >= n) return;
    if (succ[v] == PATH_NONE) { const uint4 me = jb[v]; len[me.x] = (unsigned long long)me.y + 1ull; }
}
__global__ void __launch_bounds__(256) k_path_place(i64 n, const uint4 *__restrict__ jb,
                                                    const i64 *__restrict__ base, const unsigned *__restrict__ succ,
                                                    const unsigned char *__restrict__ sch, unsigned *__restrict__ pos,
                                                    unsigned *__restrict__ col, unsigned *pq) {
    const i64 v = (i64)blockIdx.x * 256 + threadIdx.x;
    if (v >= n) return;
    const uint4 me = jb[v];
    const unsigned t = (unsigned)base[me.x] + me.y;
    pos[v] = t;
    col[t] = (unsigned)v;
    if (succ[v] != PATH_NONE) {                         // quad t>>5 = { chars lo, chars hi, go mask, - }
        unsigned *quad = pq + (size_t)(t >> 5) * 4;
        const unsigned s = t & 31u;
        if (sch[v]) atomicOr(&quad[s >> 4], (unsigned)sch[v] << (2 * (s & 15u)));
        atomicOr(&quad[2], 1u << s);
    }
}

// ---------------------------------------------------------------------------------------------
// Stitched paths.  Vertex-disjoint paths end wherever two strains merge: the shared stretch belongs to ONE path, the other
// strain's path stops at the merge and a new one starts where the strains part again -- a read of that strain makes two
// transitions per shared stretch (config 2: 21 % of the k-mers are shared, 1.4 transitions per read), and its path pieces
// are too short for most substitution-safe bits.  So the path ORDER may repeat columns: behind the last column e of a path
// comes a COPY of the stretch of the other path that e's successor lies on, up to a column that has another way out onto
// the head of a third path, and that path follows -- path, copied stretch, path: one chain, numbered consecutively.
// Nothing about a position changes: col[t+1] is still the successor of col[t] by the position's char, a column's state bits
// and transitions are its own wherever it stands; pos[v] names the position in v's own path.  Which head a tail is
// joined to is a guess (the first free one within ST_MAXCOPY steps, smallest path number first where several tails want
// one head); a wrong guess costs the transition it would have cost anyway.  The copies are bounded to a fifth of the columns; a graph that would need more keeps its disjoint paths.
// Works on the positions of the disjoint layout (k_path_place), where a path is a run of positions: GO ... GO, end.
// ---------------------------------------------------------------------------------------------
#define ST_MAXCOPY 256
struct StLink { int next; unsigned t0, len, chars; };          // next path (-1: none), first copied position, copy length, c_in | c_out << 2
__device__ __forceinline__ bool st_go(const uint4 *__restrict__ pq, i64 t) { return (pq[t >> 5].z >> (int)(t & 31)) & 1u; }
__device__ __forceinline__ unsigned st_char(const uint4 *__restrict__ pq, i64 t) {
    return (unsigned)(quad_bits(pq[t >> 5]) >> (2 * (int)(t & 31))) & 3u;
}
__global__ void __launch_bounds__(256) k_st_flags(const uint4 *__restrict__ pq, i64 n, i64 *__restrict__ flag) {
    const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
    if (t < n) flag[t] = (t == 0 || !st_go(pq, t - 1)) ? 1 : 0;
}
__global__ void __launch_bounds__(256) k_st_paths(const i64 *__restrict__ flag, const i64 *__restrict__ pex, i64 n,
                                                  unsigned *__restrict__ start, unsigned *__restrict__ pid_of) {
    const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    if (flag[t]) start[pex[t]] = (unsigned)t;
    pid_of[t] = (unsigned)(pex[t] + flag[t] - 1);
    if (t == n - 1) start[pex[n]] = (unsigned)n;              // sentinel: the end of the last path
}
// One round of joining tails to heads.  Every path whose tail is still free looks, along the path its column's successor
// lies on, for the first column with a way out onto the head of a third path that nobody holds yet, and PROPOSES to that
// head (atomicMin on the proposer's number); k_st_accept gives every head to its smallest proposer.  Rounds repeat until
// nobody proposes: the outcome does not depend on the order the threads ran in.
// cand[p] = { head's path, first copied position, copy length, c_in | c_out << 2 | branching columns copied << 4 }
__global__ void __launch_bounds__(256) k_st_propose(SbwtIndexView ix, const unsigned *__restrict__ col, const unsigned *__restrict__ pos,
                                                    const uint4 *__restrict__ pq, const unsigned *__restrict__ pid_of,
                                                    const unsigned *__restrict__ start, i64 np, i64 n, const int *__restrict__ claim,
                                                    const StLink *__restrict__ link, unsigned char *__restrict__ spent,
                                                    unsigned *__restrict__ prop, uint4 *__restrict__ cand, int min_copy, int *any) {
    const i64 p = (i64)blockIdx.x * 256 + threadIdx.x;
    if (p >= np) return;
    cand[p] = make_uint4(0xFFFFFFFFu, 0u, 0u, 0u);
    if (link[p].next >= 0 || spent[p]) return;
    const i64 end = (i64)start[p + 1] - 1;
    const PathGroup pg = path_group(ix, (i64)col[end]);
    for (unsigned c0 = 0; c0 < 4; c0++) {                       // where the tail's column goes on (usually one way: a merge)
        if (pg.target[c0] == PATH_NONE) continue;
        const i64 t0 = pos[pg.target[c0]];
        const unsigned q0 = pid_of[t0];
        if ((i64)q0 == p) break;
        unsigned nbr = 0;
        for (int j = 0; j < ST_MAXCOPY; j++) {
            const i64 t = t0 + j;
            if (t >= n || pid_of[t] != q0) break;              // the other path ended first
            const PathGroup pv = path_group(ix, (i64)col[t]);
            nbr += pv.deg >= 2;
            if (j + 1 < min_copy) continue;
            const bool go = st_go(pq, t);
            const unsigned y = st_char(pq, t);
            for (unsigned c = 0; c < 4; c++) {
                const unsigned h = pv.target[c];
                if (h == PATH_NONE || (go && c == y)) continue;
                const i64 th = pos[h];
                const unsigned q1 = pid_of[th];
                if (th != (i64)start[q1] || (i64)q1 == p || q1 == q0 || claim[q1] != -1) continue;   // not a free head of a third path
                atomicMin(&prop[q1], (unsigned)p);
                cand[p] = make_uint4(q1, (unsigned)t0, (unsigned)(j + 1), c0 | (c << 2) | (nbr << 4));
                *any = 1;
                return;
            }
        }
        break;                                                 // (the first way on only)
    }
    spent[p] = 1;                                              // nothing within reach
}
__global__ void __launch_bounds__(256) k_st_accept(i64 np, const unsigned *__restrict__ prop, const uint4 *__restrict__ cand,
                                                   int *__restrict__ claim, StLink *__restrict__ link,
                                                   unsigned long long *__restrict__ totals) {
    const i64 p = (i64)blockIdx.x * 256 + threadIdx.x;
    if (p >= np) return;
    const uint4 c = cand[p];
    if (c.x == 0xFFFFFFFFu || prop[c.x] != (unsigned)p) return;
    claim[c.x] = (int)p;
    link[p] = StLink{(int)c.x, c.y, c.z, c.w & 15u};
    atomicAdd(&totals[0], (unsigned long long)c.z);            // positions copied
    atomicAdd(&totals[1], (unsigned long long)(c.w >> 4));     // ... of them columns with two or more successors
}
__global__ void __launch_bounds__(256) k_st_link_init(i64 np, StLink *__restrict__ link) {
    const i64 p = (i64)blockIdx.x * 256 + threadIdx.x;
    if (p < np) link[p] = StLink{-1, 0u, 0u, 0u};
}
// list ranking over the paths of a chain: { jump towards the chain's first path, positions before this path, smallest path seen }
__global__ void __launch_bounds__(256) k_st_rank_init(const int *__restrict__ claim, const StLink *__restrict__ link,
                                                      const unsigned *__restrict__ start, i64 np, uint4 *__restrict__ R) {
    const i64 p = (i64)blockIdx.x * 256 + threadIdx.x;
    if (p >= np) return;
    const int prev = claim[p];
    unsigned before = 0;
    if (prev >= 0) before = (start[prev + 1] - start[prev]) + link[prev].len;
    R[p] = make_uint4(prev >= 0 ? (unsigned)prev : (unsigned)p, before, (unsigned)p, 0u);
}
__global__ void __launch_bounds__(256) k_st_rank_jump(i64 np, const uint4 *__restrict__ in, uint4 *__restrict__ out, int *moved) {
    const i64 p = (i64)blockIdx.x * 256 + threadIdx.x;
    if (p >= np) return;
    const uint4 a = in[p], b = in[a.x];
    out[p] = make_uint4(b.x, a.y + b.y, a.z < b.z ? a.z : b.z, 0u);
    if (b.x != a.x) *moved = 1;
}
__global__ void __launch_bounds__(256) k_st_rank_cut(i64 np, const uint4 *__restrict__ R, int *__restrict__ claim,
                                                     StLink *__restrict__ link, int *flag) {
    const i64 p = (i64)blockIdx.x * 256 + threadIdx.x;
    if (p >= np) return;
    const uint4 me = R[p];
    if (claim[me.x] == -1) return;                             // reached a chain's first path
    if (me.z == (unsigned)p) {                                 // a cycle of paths: its smallest member becomes a first path
        const int prev = claim[p];
        if (prev >= 0) { link[prev].next = -1; link[prev].len = 0; claim[p] = -1; }
        *flag = 1;
    }
}
__global__ void __launch_bounds__(256) k_st_chainlen(i64 np, const uint4 *__restrict__ R, const StLink *__restrict__ link,
                                                     const unsigned *__restrict__ start, i64 *__restrict__ chainlen) {
    const i64 p = (i64)blockIdx.x * 256 + threadIdx.x;
    if (p >= np) return;
    const i64 seg = (i64)(start[p + 1] - start[p]) + (link[p].next >= 0 ? (i64)link[p].len : 0);
    atomicAdd(reinterpret_cast<unsigned long long *>(&chainlen[R[p].x]), (unsigned long long)seg);
}
__device__ __forceinline__ void st_put(unsigned *__restrict__ pq_words, i64 t, unsigned ch) {     // chars + GO of a step that goes on
    unsigned *quad = pq_words + (size_t)(t >> 5) * 4;
    const unsigned s = (unsigned)(t & 31);
    if (ch) atomicOr(&quad[s >> 4], ch << (2 * (s & 15u)));
    atomicOr(&quad[2], 1u << s);
}
__global__ void __launch_bounds__(256) k_st_place(i64 n, const unsigned *__restrict__ col_old, const uint4 *__restrict__ pq_old,
                                                  const unsigned *__restrict__ pid_of, const unsigned *__restrict__ start,
                                                  const uint4 *__restrict__ R, const i64 *__restrict__ cbase,
                                                  const StLink *__restrict__ link, unsigned *__restrict__ col,
                                                  unsigned *__restrict__ pos, unsigned *__restrict__ pq_words) {
    const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    const unsigned p = pid_of[t];
    const uint4 r = R[p];
    const i64 nt = cbase[r.x] + (i64)r.y + (t - (i64)start[p]);
    const unsigned v = col_old[t];
    col[nt] = v;
    pos[v] = (unsigned)nt;
    if (st_go(pq_old, t)) st_put(pq_words, nt, st_char(pq_old, t));
    else if (link[p].next >= 0) st_put(pq_words, nt, link[p].chars & 3u);        // the tail's step into the copied stretch
}
__global__ void __launch_bounds__(256) k_st_copy(i64 np, const unsigned *__restrict__ col_old, const uint4 *__restrict__ pq_old,
                                                 const unsigned *__restrict__ start, const uint4 *__restrict__ R,
                                                 const i64 *__restrict__ cbase, const StLink *__restrict__ link,
                                                 unsigned *__restrict__ col, unsigned *__restrict__ pq_words) {
    const i64 p = (i64)blockIdx.x * 256 + threadIdx.x;
    if (p >= np) return;
    const StLink lk = link[p];
    if (lk.next < 0) return;
    const uint4 r = R[p];
    const i64 first = cbase[r.x] + (i64)r.y + (i64)(start[p + 1] - start[p]);     // behind the path's own positions
    for (unsigned j = 0; j < lk.len; j++) {
        col[first + j] = col_old[lk.t0 + j];
        st_put(pq_words, first + j, j + 1 < lk.len ? st_char(pq_old, (i64)lk.t0 + j) : (lk.chars >> 2) & 3u);
    }
}

// Substitution-safe bits.  Path index u carries the char ch[u] of the step from position u to u+1.  Bit u says:
// the 2k steps around u lie on one path, and replacing ch[u] by any other base gives, in each of the k windows
// of k chars that contain it, a k-mer that is NOT in the index (3k exact lookups in the depth-k sparse table).
// A read that follows the path, differs from it in exactly the base at u and agrees again on the next k-1 bases
// therefore has -1 for all k k-mers that contain that base -- no probe needed (M_BRIDGE).
__device__ __forceinline__ bool sp_present(const SbwtIndexView &ix, u64 key) {
    size_t bkt = sbwt_sp_bucket(key, ix.n_sb, 0u);
    for (;;) {
        const uint4 e0 = ix.stab[2 * bkt], e1 = ix.stab[2 * bkt + 1];
        const u64 w0 = quad_bits(e0), w1 = quad_bits(e1);
        if ((w0 & ~SBWT_SP_OVERFLOW) == key || w1 == key) return true;
        if (!(w0 & SBWT_SP_OVERFLOW)) return false;
        bkt = bkt + 1 < ix.n_sb ? bkt + 1 : 0;
    }
}
__global__ void __launch_bounds__(256) k_path_safe(SbwtIndexView ix, unsigned *pq_words) {
    const i64 u = (i64)blockIdx.x * 256 + threadIdx.x;
    const int k = ix.k;
    if (u < k || u + k > ix.n_pos) return;
    const i64 lo = u - k;                               // steps lo .. lo+2k-1 must all be kept
    const uint4 *q = ix.pq + (lo >> 5);
    const uint4 a = q[0], b = q[1], c = q[2];
    const int s = (int)(lo & 31);
    const u64 A = quad_bits(a), B = quad_bits(b), C = quad_bits(c);
    const u64 w0 = s ? ((A >> (2 * s)) | (B << (64 - 2 * s))) : A;       // chars lo .. lo+31
    const u64 w1 = s ? ((B >> (2 * s)) | (C << (64 - 2 * s))) : B;       // chars lo+32 .. lo+63
    const u64 ga = ((u64)b.z << 32) | (u64)a.z, gb = ((u64)c.z << 32) | (u64)b.z;
    const u64 g = (ga >> s) | (s ? ((gb >> 32) << (64 - s)) : 0ull);      // go bits lo .. lo+63
    if ((g & low_mask(2 * k)) != low_mask(2 * k)) return;
    const u64 km = low_mask(2 * k);
    for (int w = 0; w < k; w++) {
        const int st = k - w;                           // the window starts st steps after lo; ch[u] is its char w
        u64 key = (st < 32) ? ((w0 >> (2 * st)) | (st ? (w1 << (64 - 2 * st)) : 0ull)) : (w1 >> (2 * (st - 32)));
        key &= km;
        for (u64 alt = 1; alt < 4; alt++)
            if (sp_present(ix, key ^ (alt << (2 * w)))) return;
    }
    atomicOr(&pq_words[(size_t)(u >> 5) * 4 + 3], 1u << (int)(u & 31));
}

// The same bit by a wider rule (the default): the k-1 steps BEFORE u need not lie on u's path.  What the k windows hold
// left of the substituted base are the last k-1 chars of the label of column col[u], and near the head of a path those
// come from the head's own label (k_path_head_labels: k backward steps, SBWT.hh:700-746) followed by the path's chars.
// Only the k steps u .. u+k-1 must be on the path -- M_BRIDGE compares the read with exactly those, and goes on from
// position u+k.  On genomes whose paths are ~100 columns long this makes about half as many steps again bridgeable.
// (the heads are listed first: one position in ~100 is a head, and a wave that walks k backward steps for one or two
// of its lanes costs as much as one that does it for 64 -- 69 -> 6 ms at 142 M columns)
__global__ void __launch_bounds__(256) k_path_list_heads(SbwtIndexView ix, unsigned *__restrict__ list, u64 *count) {
    const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
    const bool head = t < ix.n_pos && (t == 0 || !((ix.pq[(t - 1) >> 5].z >> (int)((t - 1) & 31)) & 1u));   // step t-1 does not go on
    const u64 slot = block_append_slot(count, head);
    if (slot != ~0ull) list[slot] = (unsigned)t;
}
template <bool MEGA>
__global__ void __launch_bounds__(256) k_path_head_labels(SbwtIndexView ix, const unsigned *__restrict__ list,
                                                          const u64 *__restrict__ count, u64 *__restrict__ hlab) {
    const i64 e = (i64)blockIdx.x * 256 + threadIdx.x;
    if (e >= (i64)*count) return;
    const i64 t = list[e];
    const int k = ix.k;
    i64 v = ix.col[t];
    u64 lab = 0;                                        // char j of the label at bits 2j (first char lowest, as the table keys)
    for (int i = 0; i < k; i++) {
        if (v == 0) { lab = ~0ull; break; }             // a dummy column: its label starts with '$' -- never vouched for
        int c = 0;
        while (c + 1 < 4 && v >= ix.C[c + 1]) c++;
        lab |= (u64)c << (2 * (k - 1 - i));
        const i64 hi_c = (c < 3) ? ix.C[c + 1] : ix.n_nodes;
        v = select_in_row<MEGA>(ix, c, v, hi_c - ix.C[c]);
    }
    hlab[t] = lab;
}
// Is this p_filter-mer (perhaps) in the index?  (the search kernels' probe filter: no false negatives)
__device__ __forceinline__ bool pf_maybe(const SbwtIndexView &ix, u64 key) {
    const u64 h = sbwt_pf_hash(key);
    const uint4 blk = ix.pfil[h >> (64 - ix.log2f)];
    const unsigned hb = sbwt_pf_bits(h), b1 = hb & 127u, b2 = (hb >> 7) & 127u;
    const unsigned w1 = (b1 < 64) ? (b1 < 32 ? blk.x : blk.y) : (b1 < 96 ? blk.z : blk.w);
    const unsigned w2 = (b2 < 64) ? (b2 < 32 ? blk.x : blk.y) : (b2 < 96 ? blk.z : blk.w);
    return ((w1 >> (b1 & 31u)) & (w2 >> (b2 & 31u)) & 1u) != 0;
}
// The safe-bit kernels ask, for every step u and each of the three substitutes of its char, whether ANY of the k k-mers that hold
// the substituted base is in the index: 3k table lookups per position, nearly all of them misses (a third of an image's build
// time).  Round 6: the probe filter first.  A k-mer that is there has every one of its L-mers in the filter (L = p_filter), so a
// few L-windows that hold the substituted base and between them lie inside every one of the k windows -- starts c0-L+1,
// c0-L+1+g, ..., c0 with g = k-L+1 (c0: the base's place in the 2k-1 chars around it) -- rule a substitute out when ALL of them
// are absent; only a substitute some window could not rule out takes the exact lookups.  Returns the substitutes ruled out
// (bit a-1).  The bits that come out are the same: the filter only ever proves absence.
template <typename W>
__device__ __forceinline__ unsigned safe_prefilter(const SbwtIndexView &ix, W S, int k) {
    const int L = ix.p_filter;
    if (!ix.pfil || L <= 0 || L > k || L > 31) return 0u;
    const int c0 = k - 1, g = k - L + 1;
    const u64 lm = low_mask(2 * L);
    unsigned out = 7u;
    for (int t = c0 - L + 1;; t = (t + g < c0) ? t + g : c0) {
        const u64 key = (u64)(S >> (2 * t)) & lm;
        for (u64 alt = 1; alt < 4; alt++)
            if (((out >> (alt - 1)) & 1u) && pf_maybe(ix, key ^ (alt << (2 * (c0 - t))))) out &= ~(1u << (alt - 1));
        if (t == c0 || !out) break;
    }
    return out;
}
// alt[u] (may be null): bit a-1 set <=> the step is safe for the one substitute ch[u] ^ a (the transition entry of that char,
// which says "no successor", passes it on: a read with exactly that base is bridged after its transition lookup -- in a
// pan-genome most steps have ONE real variant somewhere, and a sequencing error is usually one of the other two bases)
__global__ void __launch_bounds__(256) k_path_safe_labels(SbwtIndexView ix, unsigned *pq_words, const u64 *__restrict__ hlab,
                                                          unsigned char *__restrict__ alt_safe) {
    const i64 u = (i64)blockIdx.x * 256 + threadIdx.x;
    const int k = ix.k;
    if (u >= ix.n_pos) return;
    // steps 32(q-1) .. 32(q+2)-1 in three quads; step u is number o = 32 + s of them
    const i64 q = u >> 5;
    const int s = (int)(u & 31), o = 32 + s;
    const uint4 zero = make_uint4(0u, 0u, 0u, 0u);
    const uint4 qa = q > 0 ? ix.pq[q - 1] : zero, qb = ix.pq[q], qc = ix.pq[q + 1];
    const u64 A = quad_bits(qa), B = quad_bits(qb), C = quad_bits(qc);
    // the k steps u .. u+k-1 and their chars
    const u64 gr = ((((u64)qc.z << 32) | (u64)qb.z) >> s) & low_mask(k);
    if (gr != low_mask(k)) return;
    const u64 right = ((B >> (2 * s)) | (s ? (C << (64 - 2 * s)) : 0ull)) & low_mask(2 * k);
    // the k-1 steps before u: on this path, or as far back as its head and then the head's label
    const int f = o - (k - 1);                          // first of them, in [2, 62]
    const u64 gl = (((((u64)qb.z << 32) | (u64)qa.z) >> f)) & low_mask(k - 1);
    const u64 val = ((f < 32) ? ((A >> (2 * f)) | (B << (64 - 2 * f))) : (B >> (2 * (f - 32)))) & low_mask(2 * (k - 1));
    u64 left = val;
    if (gl != low_mask(k - 1)) {
        const int jz = 63 - __clzll((i64)(~gl & low_mask(k - 1)));      // the last step that does not go on
        const int d = k - 2 - jz;                       // u is d steps after its path's head
        const u64 H = hlab[u - d];
        if (H == ~0ull) return;
        const u64 lm = low_mask(2 * (k - 1 - d));
        left = ((H >> (2 * (d + 1))) & lm) | (val & ~lm);
    }
    // S = left (k-1 chars) . ch[u] . right's other k-1 chars; window number w starts at char w, ch[u] is its char k-1-w
    const u64 Slo = left | (right << (2 * (k - 1))), Shi = right >> (64 - 2 * (k - 1));
    const u64 km = low_mask(2 * k);
    unsigned ok = 7u;                                   // substitutes not seen in any window yet
    // (the probe filter rules most substitutes out with two or three probes each; the others take the k exact lookups)
    const unsigned ruled_out = safe_prefilter(ix, (unsigned __int128)Slo | ((unsigned __int128)Shi << 64), k);
    for (int w = 0; w < k && (ok & ~ruled_out); w++) {
        const u64 key = ((Slo >> (2 * w)) | (w ? (Shi << (64 - 2 * w)) : 0ull)) & km;
        for (u64 alt = 1; alt < 4; alt++)
            if (((ok & ~ruled_out) >> (alt - 1) & 1u) && sp_present(ix, key ^ (alt << (2 * (k - 1 - w))))) ok &= ~(1u << (alt - 1));
    }
    if (alt_safe) alt_safe[u] = (unsigned char)ok;
    if (ok == 7u) atomicOr(&pq_words[(size_t)q * 4 + 3], 1u << s);
}

// ---- the same bits for 31 < k <= 63: whole k-mers live in the two-level table (stab at depth 31 + stab2), windows are up
// to 126 bits wide (unsigned __int128), the k-1 steps left of u span up to three path groups.  Rule 2 only (head labels).
typedef unsigned __int128 u128;
__device__ __forceinline__ u128 mask128(int bits) { return bits >= 128 ? ~(u128)0 : ((((u128)1) << bits) - 1); }
__device__ __forceinline__ bool sp_find(const SbwtIndexView &ix, u64 key, unsigned *first) {
    size_t bkt = sbwt_sp_bucket(key, ix.n_sb, 0u);
    for (;;) {
        const uint4 e0 = ix.stab[2 * bkt], e1 = ix.stab[2 * bkt + 1];
        const u64 w0 = quad_bits(e0), w1 = quad_bits(e1);
        if ((w0 & ~SBWT_SP_OVERFLOW) == key) { *first = e0.z; return true; }
        if (w1 == key) { *first = e1.z; return true; }
        if (!(w0 & SBWT_SP_OVERFLOW)) return false;
        bkt = bkt + 1 < ix.n_sb ? bkt + 1 : 0;
    }
}
__device__ __forceinline__ bool sp2_present(const SbwtIndexView &ix, unsigned origin, u64 key2) {
    unsigned e = sbwt_sp2_entry(origin, key2, ix.n_sb2, 0u);
    for (;;) {
        const uint4 a = ix.stab2[2 * (size_t)e], b = ix.stab2[2 * (size_t)e + 1];
        if (ix.big) {
            if (a.w != 0u && quad_bits(a) == key2 && a.z == origin) return true;
            if (b.w != 0u && quad_bits(b) == key2 && b.z == origin) return true;
            if (a.w == 0u || b.w == 0u) return false;           // (a bucket with a free entry ends the search)
        } else {
            if ((a.w & SBWT_SP2_USED) && quad_bits(a) == key2 && (a.z & ~SBWT_SP2_OVERFLOW) == origin) return true;
            if ((b.w & SBWT_SP2_USED) && quad_bits(b) == key2 && b.z == origin) return true;
            if (!(a.z & SBWT_SP2_OVERFLOW)) return false;
        }
        e = e + 1 < ix.n_sb2 ? e + 1 : 0;
    }
}
template <bool MEGA>
__global__ void __launch_bounds__(256) k_path_head_labels_wide(SbwtIndexView ix, const unsigned *__restrict__ list,
                                                               const u64 *__restrict__ count, u64 *__restrict__ hlab) {
    const i64 e = (i64)blockIdx.x * 256 + threadIdx.x;
    if (e >= (i64)*count) return;
    const i64 t = list[e];
    const int k = ix.k;
    i64 v = ix.col[t];
    u128 lab = 0;                                       // char j of the label at bits 2j (first char lowest, as the table keys)
    bool dummy = false;
    for (int i = 0; i < k; i++) {
        if (v == 0) { dummy = true; break; }            // a dummy column: its label starts with '$' -- never vouched for
        int c = 0;
        while (c + 1 < 4 && v >= ix.C[c + 1]) c++;
        lab |= (u128)(unsigned)c << (2 * (k - 1 - i));
        const i64 hi_c = (c < 3) ? ix.C[c + 1] : ix.n_nodes;
        v = select_in_row<MEGA>(ix, c, v, hi_c - ix.C[c]);
    }
    hlab[2 * t] = dummy ? ~0ull : (u64)lab;
    hlab[2 * t + 1] = dummy ? ~0ull : (u64)(lab >> 64);
}
__global__ void __launch_bounds__(256) k_path_safe_labels_wide(SbwtIndexView ix, unsigned *pq_words, const u64 *__restrict__ hlab,
                                                               unsigned char *__restrict__ alt_safe) {
    const i64 u = (i64)blockIdx.x * 256 + threadIdx.x;
    const int k = ix.k;                                 // 32 .. 63
    if (u >= ix.n_pos) return;
    // steps 32(q-2) .. 32(q+3)-1 in five quads; step u is number o = 64 + s of them
    const i64 q = u >> 5;
    const int s = (int)(u & 31), o = 64 + s;
    u64 Cw[7];
    unsigned Gw[7];
#pragma unroll
    for (int j = 0; j < 5; j++) {
        const i64 qq = q - 2 + j;
        const uint4 v = qq >= 0 ? ix.pq[qq] : make_uint4(0u, 0u, 0u, 0u);
        Cw[j] = quad_bits(v);
        Gw[j] = v.z;                                    // (build-time layout: z = GO)
    }
    Cw[5] = Cw[6] = 0ull;
    Gw[5] = Gw[6] = 0u;
    const u128 glo = (u128)Gw[0] | ((u128)Gw[1] << 32) | ((u128)Gw[2] << 64) | ((u128)Gw[3] << 96);
    const u64 ghi = (u64)Gw[4];
    auto go_bits = [&](int f, int n) -> u64 {           // go bits of steps number f .. f+n-1 (n <= 63)
        u128 v = glo >> f;
        if (f) v |= (u128)ghi << (128 - f);
        return (u64)v & low_mask(n);
    };
    auto chars = [&](int f, int n) -> u128 {            // chars of steps number f .. f+n-1 (n <= 63), 2 bits each
        const int w = f >> 5, sh = 2 * (f & 31);
        const u128 lo = (u128)Cw[w] | ((u128)Cw[w + 1] << 64);
        u128 v = lo >> sh;
        if (sh) v |= (u128)Cw[w + 2] << (128 - sh);
        return v & mask128(2 * n);
    };
    // the k steps u .. u+k-1 and their chars
    if (go_bits(o, k) != low_mask(k)) return;
    const u128 right = chars(o, k);
    // the k-1 steps before u: on this path, or as far back as its head and then the head's label
    const int f = o - (k - 1);                          // first of them, in [2, 64]
    const u64 gl = go_bits(f, k - 1);
    const u128 val = chars(f, k - 1);
    u128 left = val;
    if (gl != low_mask(k - 1)) {
        const int jz = 63 - __clzll((i64)(~gl & low_mask(k - 1)));      // the last step that does not go on
        const int d = k - 2 - jz;                       // u is d steps after its path's head
        const u64 H0 = hlab[2 * (u - d)], H1 = hlab[2 * (u - d) + 1];
        if (H0 == ~0ull && H1 == ~0ull) return;
        const u128 H = (u128)H0 | ((u128)H1 << 64);
        const u128 lm = mask128(2 * (k - 1 - d));
        left = ((H >> (2 * (d + 1))) & lm) | (val & ~lm);
    }
    // S = left (k-1 chars) . ch[u] . right's other k-1 chars; window number w starts at char w, ch[u] is its char k-1-w
    const u128 Slo = left | (right << (2 * (k - 1)));
    const u128 Shi = right >> (128 - 2 * (k - 1));      // (62 <= 2(k-1) <= 124)
    const u128 km = mask128(2 * k);
    const u64 m62 = low_mask(62);
    const int rest = k - 31;                            // chars of the second-level key, 1 .. 32
    const u64 mrest = rest >= 32 ? ~0ull : low_mask(2 * rest);
    unsigned ok = 7u;                                   // substitutes not seen in any window yet
    // (the probe filter first, as in k_path_safe_labels: a few L-windows that hold the substituted base rule a substitute out)
    unsigned ruled_out = 0u;
    {
        const int L = ix.p_filter;
        if (ix.pfil && L > 0 && L <= 31) {
            const int c0 = k - 1, g = k - L + 1;
            const u64 lm = low_mask(2 * L);
            ruled_out = 7u;
            for (int t = c0 - L + 1;; t = (t + g < c0) ? t + g : c0) {
                u128 v = 2 * t < 128 ? (Slo >> (2 * t)) : (Shi >> (2 * t - 128));
                if (t && 2 * t < 128) v |= Shi << (128 - 2 * t);
                const u64 key = (u64)v & lm;
                for (u64 alt = 1; alt < 4; alt++)
                    if (((ruled_out >> (alt - 1)) & 1u) && pf_maybe(ix, key ^ (alt << (2 * (c0 - t))))) ruled_out &= ~(1u << (alt - 1));
                if (t == c0 || !ruled_out) break;
            }
        }
    }
    for (int w = 0; w < k && (ok & ~ruled_out); w++) {
        u128 key = Slo >> (2 * w);
        if (w) key |= Shi << (128 - 2 * w);
        key &= km;
        const u64 P = (u64)key & m62, R = (u64)(key >> 62) & mrest;
        const int a = k - 1 - w;                        // ch[u] is char a of the window
        if (a < 31) {
            for (u64 alt = 1; alt < 4; alt++) {
                unsigned first = 0;
                if ((((ok & ~ruled_out) >> (alt - 1)) & 1u) && sp_find(ix, P ^ (alt << (2 * a)), &first) && sp2_present(ix, first, R))
                    ok &= ~(1u << (alt - 1));
            }
        } else {
            unsigned first = 0;
            if (sp_find(ix, P, &first))
                for (u64 alt = 1; alt < 4; alt++)
                    if ((((ok & ~ruled_out) >> (alt - 1)) & 1u) && sp2_present(ix, first, R ^ (alt << (2 * (a - 31))))) ok &= ~(1u << (alt - 1));
        }
    }
    if (alt_safe) alt_safe[u] = (unsigned char)ok;
    if (ok == 7u) atomicOr(&pq_words[(size_t)q * 4 + 3], 1u << s);
}

// Where a read can leave its path, and the transition table.  Position t (column v = col[t]) offers the successors of v's
// suffix group; the path itself takes one of them (char y).  ONLY[t]: y is the only one -- a read that differs from the path
// there gets -1 without any gather (SBWT.hh:572-575: no column of the group carries its char), or a bridge where the SAFE bit
// vouches for the step.  Every other way off a path is ONE gather of a 32-byte entry of the transition table, hashed on
// (t, c) with linear probing:
//   quad 0 = { t + 1 (0 = free slot), c | flags, successor column (the streaming step's answer, SBWT.hh:562-575), its path position p }
//   quad 1 = the successor's path from p on, steps p .. p+31, in the path groups' own encoding { chars lo, chars hi, A, B }
//            (pre-shifted: a read that takes the transition runs on along them without a look at pq)
// Entries exist for (i) every successor of a column whose path ends there (a char without one finds a free slot: -1), and
// (ii) at a step with two or more successors, for ALL three substitutes of the path's char: a successor, or a NEGATIVE entry
// (flag 0x100) that says "none" and whether the step is substitution-safe for exactly this char (0x200, k_path_safe_labels:
// in a pan-genome most branching steps have ONE real variant, and a sequencing error is one of the other two bases).
__global__ void __launch_bounds__(256) k_path_oth(SbwtIndexView ix, unsigned *__restrict__ only,
                                                  unsigned long long *__restrict__ counters) {
    const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
    int n_ent = 0, branch = 0;
    if (t < ix.n_pos) {
        const PathGroup pg = path_group(ix, (i64)ix.col[t]);
        const bool go = (ix.pq[t >> 5].z >> (int)(t & 31)) & 1u;        // (build-time layout: z = GO, w = SAFE)
        if (!go) n_ent = pg.deg;
        else if (pg.deg >= 2) n_ent = 3;
        else atomicOr(&only[t >> 5], 1u << (unsigned)(t & 31));         // exactly one successor: where the path goes on
        branch = pg.deg >= 2;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { n_ent += __shfl_down(n_ent, off); branch += __shfl_down(branch, off); }
    if ((threadIdx.x & 63) == 0) {
        if (n_ent) atomicAdd(&counters[0], (unsigned long long)n_ent);
        if (branch) atomicAdd(&counters[1], (unsigned long long)branch);
    }
}
// (runs on the FINAL encoding of the path groups, k_path_reencode: go = ~A | B)
__global__ void __launch_bounds__(256) k_trans_insert(SbwtIndexView ix, unsigned *__restrict__ table, unsigned n_slots,
                                                      const unsigned char *__restrict__ alt_safe) {
    const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
    if (t >= ix.n_pos) return;
    const uint4 Q = ix.pq[t >> 5];
    const int s = (int)(t & 31);
    const bool go = ((~Q.z | Q.w) >> s) & 1u, only = ((~Q.z & Q.w) >> s) & 1u;
    if (go && only) return;
    const PathGroup pg = path_group(ix, (i64)ix.col[t]);
    if (go && pg.deg < 2) return;
    const unsigned y = go ? ((unsigned)(quad_bits(Q) >> (2 * s)) & 3u) : 0u;
    const unsigned asafe = (go && alt_safe) ? alt_safe[t] : 0u;
#pragma unroll
    for (unsigned c = 0; c < 4; c++) {
        if (go ? (c == y) : (pg.target[c] == PATH_NONE)) continue;
        uint4 e0, e1 = make_uint4(0u, 0u, 0u, 0u);
        if (pg.target[c] == PATH_NONE) {
            const unsigned a = c ^ y;
            e0 = make_uint4((unsigned)t + 1u, c | SBWT_TRANS_NEG | (((asafe >> (a - 1)) & 1u) ? SBWT_TRANS_NEG_SAFE : 0u), PATH_NONE, 0u);
        } else {
            const unsigned nc = pg.target[c], np = ix.pos[nc];
            const uint4 *q = ix.pq + (np >> 5);         // the successor's path from np on: 32 steps out of two path groups
            const uint4 a0 = q[0], a1 = q[1];
            const int sp = (int)(np & 31u);
            const u64 ch = (quad_bits(a0) >> (2 * sp)) | (sp ? (quad_bits(a1) << (64 - 2 * sp)) : 0ull);
            e0 = make_uint4((unsigned)t + 1u, c, nc, np);
            e1 = make_uint4((unsigned)ch, (unsigned)(ch >> 32), (unsigned)((((u64)a1.z << 32) | (u64)a0.z) >> sp),
                            (unsigned)((((u64)a1.w << 32) | (u64)a0.w) >> sp));
        }
        size_t slot = sbwt_trans_slot((unsigned)t, c, n_slots, 0u);
        for (;;) {
            if (atomicCAS(&table[slot * 8], 0u, e0.x) == 0u) break;
            slot = slot + 1 < n_slots ? slot + 1 : 0;
        }
        table[slot * 8 + 1] = e0.y;
        table[slot * 8 + 2] = e0.z;
        table[slot * 8 + 3] = e0.w;
        reinterpret_cast<uint4 *>(table + slot * 8)[1] = e1;
    }
}

// Final form of the path groups.  While the path order is built, quad t>>5 = { chars lo, chars hi, GO, SAFE }.  The search
// kernels want a third bit per position (ONLY: the path's char is the only successor) in the same 16 bytes, and a
// substitution-safe step is necessarily an only-successor step, so two words encode the four states of a position:
//     A B
//     0 0   the path goes on                        A = SAFE | ~GO
//     0 1   ... and its char is the only successor   B = SAFE | (ONLY & GO)
//     1 1   ... and the step is substitution-safe
//     1 0   the path ends here
// decoded by the kernels as  go = ~A | B,  safe = A & B,  only = ~A & B.
__global__ void __launch_bounds__(256) k_path_reencode(uint4 *__restrict__ pq, i64 n_quads, const unsigned *__restrict__ only) {
    const i64 q = (i64)blockIdx.x * 256 + threadIdx.x;
    if (q >= n_quads) return;
    uint4 v = pq[q];
    const unsigned go = v.z, safe = v.w, on = only[q];
    v.z = safe | ~go;
    v.w = safe | (on & go);
    pq[q] = v;
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
long long sbwt_derive_scratch_bytes(long long n_nodes) {
    return 2 * ((n_nodes + 255) & ~255ll) + (n_nodes / 64 + 1) * 8 + 256;
}
void sbwt_launch_derive_marks(const SbwtIndexView &ix, uint4 *d_blocks, void *d_scratch, hipStream_t stream) {
    const i64 n = ix.n_nodes, nb = n / 64 + 1, nal = (n + 255) & ~255ll;
    unsigned char *last = reinterpret_cast<unsigned char *>(d_scratch), *next = last + nal;
    u64 *acc = reinterpret_cast<u64 *>(next + nal);
    (void)hipMemsetAsync(acc, 0, (size_t)nb * 8, stream);
    hipLaunchKernelGGL(k_sg_init, dim3(grid_for(n)), dim3(256), 0, stream, ix, last);
    for (int round = 0; round < ix.k - 1; round++) {
        hipLaunchKernelGGL(k_sg_mark, dim3(grid_for(nb)), dim3(256), 0, stream, last, n, acc);
        if (ix.n_mega > 1)
            hipLaunchKernelGGL(k_sg_propagate<true>, dim3(grid_for(n)), dim3(256), 0, stream, ix, last, next);
        else
            hipLaunchKernelGGL(k_sg_propagate<false>, dim3(grid_for(n)), dim3(256), 0, stream, ix, last, next);
        unsigned char *t = last; last = next; next = t;
    }
    hipLaunchKernelGGL(k_sg_patch, dim3(grid_for(nb)), dim3(256), 0, stream, d_blocks, acc, nb);
}

// SBWTGPU_VERBOSE=2: the builders below say what they are at (stderr, after a stream synchronise)
static void derived_log(hipStream_t stream, const char *what, long long a = 0, const u64 *d_count = nullptr) {
    static const int verbose = [] { const char *e = getenv("SBWTGPU_VERBOSE"); return e ? atoi(e) : 0; }();
    if (verbose < 2) return;
    const hipError_t e = hipStreamSynchronize(stream);
    u64 c = 0;
    if (d_count) (void)hipMemcpy(&c, d_count, 8, hipMemcpyDeviceToHost);
    static std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "sbwtgpu:   derived: %s %lld (count %llu)%s  [+%.3f s]\n", what, a, (unsigned long long)c,
            e == hipSuccess ? "" : " -- STREAM ERROR", std::chrono::duration<double>(now - last).count());
    last = now;
}
// scratch of the sparse-table build: two item lists of n_nodes entries + two counters
long long sbwt_sparse_scratch_bytes(long long n_nodes) { return 2 * (n_nodes + 64) * (long long)sizeof(SpItem) + 256; }

// d_pos: path positions to store with depth-k entries (nullptr = none).  Returns 1 if they were stored,
// 0 if not (no d_pos, p_sparse < k, or some k-mer's interval is wider than one column), < 0 on error.
int sbwt_launch_build_sparse(const SbwtIndexView &ix, int p_dense, int p_sparse, long long n_buckets, uint4 *d_table,
                             void *d_scratch, const unsigned *d_pos, int p_filter, int log2f, uint4 *d_filter,
                             long long n_entries2, uint4 *d_table2, hipStream_t stream) {
    u64 *counters = reinterpret_cast<u64 *>(d_scratch);                    // [0], [1]: list lengths
    SpItem *listA = reinterpret_cast<SpItem *>(reinterpret_cast<char *>(d_scratch) + 256);
    SpItem *listB = listA + (ix.n_nodes + 64);
    (void)hipMemsetAsync(counters, 0, 256, stream);
    for (u64 t0 = 0; t0 < (u64)2 * (u64)n_buckets; t0 += SBWT_LAUNCH_SLICE) {
        const u64 left = (u64)2 * (u64)n_buckets - t0;
        hipLaunchKernelGGL(k_sp_clear, dim3(grid_for((i64)(left < SBWT_LAUNCH_SLICE ? left : SBWT_LAUNCH_SLICE))), dim3(256), 0, stream,
                           d_table, (u64)2 * (u64)n_buckets, t0);
    }
    const u64 n_dense = 1ull << (2 * p_dense);
    hipLaunchKernelGGL(k_sp_collect, dim3(grid_for((i64)n_dense)), dim3(256), 0, stream, ix.ptab, n_dense, listA,
                       counters + 0);
    SpItem *in = listA, *outl = listB;
    int ci = 0;
    for (int d = p_dense; d < p_sparse; d++) {
        (void)hipMemsetAsync(counters + (ci ^ 1), 0, 8, stream);
        const u64 threads = (u64)(ix.n_nodes + 64);            // (one per prefix; there are at most n_nodes of a depth)
        for (u64 t0 = 0; t0 < threads; t0 += SBWT_LAUNCH_SLICE) {
            const unsigned g = grid_for((i64)(threads - t0 < SBWT_LAUNCH_SLICE ? threads - t0 : SBWT_LAUNCH_SLICE));
            if (ix.n_mega > 1)
                hipLaunchKernelGGL(k_sp_expand<true>, dim3(g), dim3(256), 0, stream, ix, in, counters + ci, d, outl, counters + (ci ^ 1), t0);
            else
                hipLaunchKernelGGL(k_sp_expand<false>, dim3(g), dim3(256), 0, stream, ix, in, counters + ci, d, outl, counters + (ci ^ 1), t0);
        }
        SpItem *t = in; in = outl; outl = t;
        ci ^= 1;
        derived_log(stream, "sparse table: prefixes expanded to depth", d + 1, counters + ci);
        if (d_filter && d + 1 == p_filter) {
            (void)hipMemsetAsync(d_filter, 0, (size_t)16 << log2f, stream);
            hipLaunchKernelGGL(k_pf_insert, dim3(grid_for(ix.n_nodes + 64)), dim3(256), 0, stream, in, counters + ci,
                               reinterpret_cast<unsigned *>(d_filter), log2f);
        }
    }
    int with_pos = 0;
    if (d_pos && p_sparse == ix.k) {
        int *flag = reinterpret_cast<int *>(counters + 8);
        hipLaunchKernelGGL(k_sp_wide, dim3(grid_for(ix.n_nodes + 64)), dim3(256), 0, stream, in, counters + ci, flag);
        int h_flag = 1;
        if (hipMemcpyAsync(&h_flag, flag, 4, hipMemcpyDeviceToHost, stream) != hipSuccess) return -1;
        if (hipStreamSynchronize(stream) != hipSuccess) return -1;
        with_pos = h_flag ? 0 : 1;
    }
    derived_log(stream, "sparse table: items to insert, with positions =", with_pos, counters + ci);
    hipLaunchKernelGGL(k_sp_insert, dim3(grid_for(ix.n_nodes + 64)), dim3(256), 0, stream, in, counters + ci, d_table,
                       (unsigned)n_buckets, with_pos ? d_pos : (const unsigned *)nullptr,
                       (!with_pos && d_pos && d_table2 && ix.k > p_sparse && !ix.big) ? d_pos : (const unsigned *)nullptr);
    derived_log(stream, "sparse table: inserted", 0);
    if (d_table2 && ix.k > p_sparse) {
        // second level: carry every depth-p_sparse prefix on to depth k, remembering where it started
        (void)hipMemsetAsync(d_table2, 0, (size_t)32 * (size_t)n_entries2, stream);
        hipLaunchKernelGGL(k_sp2_seed, dim3(grid_for(ix.n_nodes + 64)), dim3(256), 0, stream, in, counters + ci);
        SpItem2 *in2 = reinterpret_cast<SpItem2 *>(in), *out2 = reinterpret_cast<SpItem2 *>(outl);
        for (int d = p_sparse; d < ix.k; d++) {
            (void)hipMemsetAsync(counters + (ci ^ 1), 0, 8, stream);
            const u64 threads = (u64)(ix.n_nodes + 64);
            for (u64 t0 = 0; t0 < threads; t0 += SBWT_LAUNCH_SLICE)
                hipLaunchKernelGGL(k_sp2_expand, dim3(grid_for((i64)(threads - t0 < SBWT_LAUNCH_SLICE ? threads - t0 : SBWT_LAUNCH_SLICE))),
                                   dim3(256), 0, stream, ix, in2, counters + ci, d - p_sparse, out2, counters + (ci ^ 1), t0);
            SpItem2 *t = in2; in2 = out2; out2 = t;
            ci ^= 1;
        }
        int *flag = reinterpret_cast<int *>(counters + 9);
        hipLaunchKernelGGL(k_sp2_insert, dim3(grid_for(ix.n_nodes + 64)), dim3(256), 0, stream, in2, counters + ci, d_table2,
                           (unsigned)n_entries2, d_pos, flag, ix.big);
        int h_flag = 1;
        if (hipMemcpyAsync(&h_flag, flag, 4, hipMemcpyDeviceToHost, stream) != hipSuccess) return -1;
        if (hipStreamSynchronize(stream) != hipSuccess) return -1;
        if (h_flag) return -3;                          // not an SBWT: the caller drops the second level
    }
    return with_pos;
}

// ---- path order (see k_path_*) ----
static inline long long path_pad(long long n) { return (n + 64 + 255) & ~255ll; }
long long sbwt_path_scratch_bytes(long long n_nodes) {
    const long long np = path_pad(n_nodes);
    const long long nb = (n_nodes + 1023) / 1024;
    return np * (8 + 32) + np + np * 8 * 2 + (nb + 2) * 8 + 4096;     // succ, prv; two arrays of { jump, dist, min, - }; ...
}
long long sbwt_path_quads(long long n_nodes) { return n_nodes / 32 + 4; }
// number of paths = positions whose "path goes on" bit is clear
__global__ void __launch_bounds__(256) k_path_count_ends(const uint4 *__restrict__ pq, i64 n, unsigned long long *counter) {
    const i64 q = (i64)blockIdx.x * 256 + threadIdx.x;
    unsigned ends = 0;
    if (q * 32 < n) {
        const i64 left = n - q * 32;
        const unsigned valid = left >= 32 ? 0xFFFFFFFFu : ((1u << (int)left) - 1u);
        ends = (unsigned)__popc(pq[q].z & ~pq[q].w & valid);      // A & ~B: the path ends (final encoding, k_path_reencode)
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ends += __shfl_down(ends, off);
    if ((threadIdx.x & 63) == 0 && ends) atomicAdd(counter, (unsigned long long)ends);
}
long long sbwt_count_paths(const SbwtIndexView &ix, hipStream_t stream) {
    unsigned long long *d = nullptr, h = 0;
    if (hipMalloc((void **)&d, 8) != hipSuccess) return -1;
    (void)hipMemsetAsync(d, 0, 8, stream);
    hipLaunchKernelGGL(k_path_count_ends, dim3(grid_for(ix.n_pos / 32 + 1)), dim3(256), 0, stream, ix.pq, (i64)ix.n_pos, d);
    hipError_t e = hipMemcpyAsync(&h, d, 8, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(d);
    return e == hipSuccess ? (long long)h : -1;
}
// rule 1: 2k steps around u on one path; rule 2 (needs d_hlab: sbwt_path_safe_scratch_bytes of scratch): k steps from u on.
// 31 < k <= 63 (whole k-mers in the two-level table): rule 2 only, head labels of 16 bytes.
long long sbwt_path_safe_scratch_bytes(long long n_pos, int k) {
    return (((long long)n_pos * (k > 31 ? 20 : 12) + 15) & ~15ll) + 256;
}
void sbwt_launch_path_safe(const SbwtIndexView &ix, uint4 *d_pq, int rule, void *d_hlab, unsigned char *d_alt_safe,
                           hipStream_t stream) {
    if (d_alt_safe) (void)hipMemsetAsync(d_alt_safe, 0, (size_t)ix.n_pos, stream);
    const bool wide = ix.k > 31;
    if (wide && !d_hlab) return;                        // (the caller does not set has_safe then)
    if (!wide && (rule < 2 || !d_hlab)) {
        hipLaunchKernelGGL(k_path_safe, dim3(grid_for(ix.n_pos)), dim3(256), 0, stream, ix, reinterpret_cast<unsigned *>(d_pq));
        return;
    }
    // d_hlab: [ hlab : n x 8 B (16 B for k > 31) ][ list of heads : n x 4 B ][ their number : 8 B ]
    u64 *hlab = reinterpret_cast<u64 *>(d_hlab);
    const size_t lab_words = wide ? 2 : 1;
    unsigned *list = reinterpret_cast<unsigned *>(hlab + lab_words * (size_t)ix.n_pos);
    u64 *count = reinterpret_cast<u64 *>(reinterpret_cast<char *>(d_hlab) + (((size_t)ix.n_pos * (wide ? 20 : 12) + 15) & ~(size_t)15));
    (void)hipMemsetAsync(count, 0, 8, stream);
    hipLaunchKernelGGL(k_path_list_heads, dim3(grid_for(ix.n_pos)), dim3(256), 0, stream, ix, list, count);
    // (the grid covers every position; blocks beyond the number of heads return at once)
    if (wide) {
        if (ix.n_mega > 1)
            hipLaunchKernelGGL(k_path_head_labels_wide<true>, dim3(grid_for(ix.n_pos)), dim3(256), 0, stream, ix, list, count, hlab);
        else
            hipLaunchKernelGGL(k_path_head_labels_wide<false>, dim3(grid_for(ix.n_pos)), dim3(256), 0, stream, ix, list, count, hlab);
        hipLaunchKernelGGL(k_path_safe_labels_wide, dim3(grid_for(ix.n_pos)), dim3(256), 0, stream, ix,
                           reinterpret_cast<unsigned *>(d_pq), hlab, d_alt_safe);
        return;
    }
    if (ix.n_mega > 1)
        hipLaunchKernelGGL(k_path_head_labels<true>, dim3(grid_for(ix.n_pos)), dim3(256), 0, stream, ix, list, count, hlab);
    else
        hipLaunchKernelGGL(k_path_head_labels<false>, dim3(grid_for(ix.n_pos)), dim3(256), 0, stream, ix, list, count, hlab);
    hipLaunchKernelGGL(k_path_safe_labels, dim3(grid_for(ix.n_pos)), dim3(256), 0, stream, ix,
                       reinterpret_cast<unsigned *>(d_pq), hlab, d_alt_safe);
}

// Joins the disjoint paths (col_o / pos_o / pq_o, k_path_place) into chains with copied stretches (see "Stitched paths"):
// d_col / d_pq have room for pos_cap positions.  Returns the number of positions (>= n), or -1.
static long long stitch_paths(const SbwtIndexView &ix, const unsigned *col_o, const unsigned *pos_o, const uint4 *pq_o,
                              unsigned *d_col, unsigned *d_pos, uint4 *d_pq, long long pos_cap, int min_copy, hipStream_t stream) {
    const i64 n = ix.n_nodes;
    const unsigned g = grid_for(n);
    const i64 nbk = (n + 1023) / 1024;
    i64 *flag = nullptr, *pex = nullptr, *bsum = nullptr, *chainlen = nullptr, *cbase = nullptr;
    unsigned *pid_of = nullptr, *start = nullptr;
    int *claim = nullptr, *ctl = nullptr;
    StLink *link = nullptr;
    unsigned char *spent = nullptr;
    unsigned *prop = nullptr;
    uint4 *cand = nullptr;
    uint4 *R[2] = {nullptr, nullptr};
    unsigned long long *budget = nullptr;
    long long n_pos = -1;
    i64 np = 0, total = 0;
    int cur = 0;
#define ST_TRY(x) do { if ((x) != hipSuccess) goto out; } while (0)
    ST_TRY(hipMalloc((void **)&flag, (size_t)(n + 1) * 8));
    ST_TRY(hipMalloc((void **)&pex, (size_t)(n + 2) * 8));
    ST_TRY(hipMalloc((void **)&bsum, (size_t)(nbk + 2) * 8));
    ST_TRY(hipMalloc((void **)&pid_of, (size_t)(n + 4) * 4));
    ST_TRY(hipMalloc((void **)&ctl, 64));
    ST_TRY(hipMalloc((void **)&budget, 16));
    hipLaunchKernelGGL(k_st_flags, dim3(g), dim3(256), 0, stream, pq_o, n, flag);
    hipLaunchKernelGGL(k_scan_block_sums, dim3((unsigned)nbk), dim3(256), 0, stream, flag, n, bsum);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, stream, bsum, nbk);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nbk), dim3(256), 0, stream, flag, n, bsum, pex);
    ST_TRY(hipMemcpyAsync(&np, pex + n, 8, hipMemcpyDeviceToHost, stream));
    ST_TRY(hipStreamSynchronize(stream));
    {
        const unsigned gp = grid_for(np);
        const i64 nbp = (np + 1023) / 1024;
        ST_TRY(hipMalloc((void **)&start, (size_t)(np + 2) * 4));
        ST_TRY(hipMalloc((void **)&claim, (size_t)(np + 1) * 4));
        ST_TRY(hipMalloc((void **)&link, (size_t)(np + 1) * sizeof(StLink)));
        ST_TRY(hipMalloc((void **)&R[0], (size_t)(np + 1) * 16));
        ST_TRY(hipMalloc((void **)&R[1], (size_t)(np + 1) * 16));
        ST_TRY(hipMalloc((void **)&chainlen, (size_t)(np + 2) * 8));
        ST_TRY(hipMalloc((void **)&cbase, (size_t)(np + 2) * 8));
        hipLaunchKernelGGL(k_st_paths, dim3(g), dim3(256), 0, stream, flag, pex, n, start, pid_of);
        ST_TRY(hipMemsetAsync(claim, 0xFF, (size_t)(np + 1) * 4, stream));
        ST_TRY(hipMemsetAsync(budget, 0, 16, stream));
        ST_TRY(hipMalloc((void **)&spent, (size_t)np + 16));
        ST_TRY(hipMalloc((void **)&prop, (size_t)(np + 1) * 4));
        ST_TRY(hipMalloc((void **)&cand, (size_t)(np + 1) * 16));
        ST_TRY(hipMemsetAsync(spent, 0, (size_t)np + 16, stream));
        hipLaunchKernelGGL(k_st_link_init, dim3(gp), dim3(256), 0, stream, np, link);
        for (int round = 0; round < 64; round++) {
            ST_TRY(hipMemsetAsync(prop, 0xFF, (size_t)(np + 1) * 4, stream));
            ST_TRY(hipMemsetAsync(ctl, 0, 4, stream));
            hipLaunchKernelGGL(k_st_propose, dim3(gp), dim3(256), 0, stream, ix, col_o, pos_o, pq_o, pid_of, start, np, n, claim, link,
                               spent, prop, cand, min_copy, ctl);
            hipLaunchKernelGGL(k_st_accept, dim3(gp), dim3(256), 0, stream, np, prop, cand, claim, link, budget);
            int h_any = 0;
            ST_TRY(hipMemcpyAsync(&h_any, ctl, 4, hipMemcpyDeviceToHost, stream));
            ST_TRY(hipStreamSynchronize(stream));
            if (!h_any) break;
        }
        {
            // A graph whose tails need more copied positions than there is room for, or whose copies would repeat many
            // BRANCHING columns (three transition entries each), is too branchy for this -- a pan-genome of many strains:
            // every merge is followed by somebody else's bubble within a few columns; the copies bought 0.7 % of the time
            // for 27 % of the image on config 3.  It keeps its vertex-disjoint paths.
            unsigned long long h_b[2] = {0, 0};
            ST_TRY(hipMemcpyAsync(h_b, budget, 16, hipMemcpyDeviceToHost, stream));
            ST_TRY(hipStreamSynchronize(stream));
            const unsigned long long cap = pos_cap > n ? (unsigned long long)(pos_cap - n) : 0ull;
            if (h_b[0] > cap || h_b[1] > (unsigned long long)(n / 32)) goto out;
        }
        // rank the paths within their chains; chains of paths that close on themselves are opened at their smallest path
        int rounds = 1;
        while (((i64)1 << rounds) < np) rounds++;
        rounds++;
        for (int attempt = 0; attempt < 4; attempt++) {
            cur = 0;
            hipLaunchKernelGGL(k_st_rank_init, dim3(gp), dim3(256), 0, stream, claim, link, start, np, R[0]);
            for (int r = 0; r < rounds; r++) {
                ST_TRY(hipMemsetAsync(ctl, 0, 4, stream));
                hipLaunchKernelGGL(k_st_rank_jump, dim3(gp), dim3(256), 0, stream, np, R[cur], R[cur ^ 1], ctl);
                cur ^= 1;
                if (r >= 5) {
                    int h_moved = 1;
                    ST_TRY(hipMemcpyAsync(&h_moved, ctl, 4, hipMemcpyDeviceToHost, stream));
                    ST_TRY(hipStreamSynchronize(stream));
                    if (!h_moved) break;
                }
            }
            ST_TRY(hipMemsetAsync(ctl + 1, 0, 4, stream));
            hipLaunchKernelGGL(k_st_rank_cut, dim3(gp), dim3(256), 0, stream, np, R[cur], claim, link, ctl + 1);
            int h_flag = 0;
            ST_TRY(hipMemcpyAsync(&h_flag, ctl + 1, 4, hipMemcpyDeviceToHost, stream));
            ST_TRY(hipStreamSynchronize(stream));
            if (!h_flag) break;
            if (attempt == 3) goto out;                 // cannot happen: one cut per cycle opens every cycle
        }
        ST_TRY(hipMemsetAsync(chainlen, 0, (size_t)(np + 2) * 8, stream));
        hipLaunchKernelGGL(k_st_chainlen, dim3(gp), dim3(256), 0, stream, np, R[cur], link, start, chainlen);
        // (chainlen is zero for paths that are not the first of a chain: its exclusive scan is the chains' first positions)
        hipLaunchKernelGGL(k_scan_block_sums, dim3((unsigned)nbp), dim3(256), 0, stream, chainlen, np, bsum);
        hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, stream, bsum, nbp);
        hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nbp), dim3(256), 0, stream, chainlen, np, bsum, cbase);
        ST_TRY(hipMemcpyAsync(&total, cbase + np, 8, hipMemcpyDeviceToHost, stream));
        ST_TRY(hipStreamSynchronize(stream));
        if (total < n || total > pos_cap) goto out;
        ST_TRY(hipMemsetAsync(d_pq, 0, (size_t)sbwt_path_quads(pos_cap) * 16, stream));
        hipLaunchKernelGGL(k_st_place, dim3(g), dim3(256), 0, stream, n, col_o, pq_o, pid_of, start, R[cur], cbase, link, d_col, d_pos,
                           reinterpret_cast<unsigned *>(d_pq));
        hipLaunchKernelGGL(k_st_copy, dim3(gp), dim3(256), 0, stream, np, col_o, pq_o, start, R[cur], cbase, link, d_col,
                           reinterpret_cast<unsigned *>(d_pq));
        ST_TRY(hipStreamSynchronize(stream));
        n_pos = total;
    }
out:
#undef ST_TRY
    (void)hipFree(flag); (void)hipFree(pex); (void)hipFree(bsum); (void)hipFree(pid_of); (void)hipFree(start); (void)hipFree(claim);
    (void)hipFree(link); (void)hipFree(R[0]); (void)hipFree(R[1]); (void)hipFree(chainlen); (void)hipFree(cbase); (void)hipFree(ctl);
    (void)hipFree(budget); (void)hipFree(spent); (void)hipFree(prop); (void)hipFree(cand);
    if (n_pos < 0) (void)hipGetLastError();
    return n_pos;
}

// d_col: pos_cap (+4 padding) u32, d_pos: n_nodes (+4) u32; d_pq: sbwt_path_quads(pos_cap) quads.  *n_pos: the positions of
// the path order (n_nodes, or more when paths were stitched: stitch != 0 and pos_cap > n_nodes).  Synchronises the stream.
int sbwt_launch_build_path(const SbwtIndexView &ix, unsigned *d_col, unsigned *d_pos, uint4 *d_pq, long long pos_cap,
                           long long *n_pos, int stitch, int min_copy, void *d_scratch, int lookahead, hipStream_t stream) {
    const i64 n = ix.n_nodes;
    const long long np = path_pad(n);
    const i64 nb = (n + 1023) / 1024;
    char *base = reinterpret_cast<char *>(d_scratch);
    int *flag = reinterpret_cast<int *>(base);
    int *moved = flag + 16;
    base += 4096;
    unsigned *succ = reinterpret_cast<unsigned *>(base); base += np * 4;
    unsigned *prv = reinterpret_cast<unsigned *>(base); base += np * 4;
    uint4 *jb[2];
    for (int a = 0; a < 2; a++) { jb[a] = reinterpret_cast<uint4 *>(base); base += np * 16; }
    unsigned char *sch = reinterpret_cast<unsigned char *>(base); base += np;
    unsigned long long *len = reinterpret_cast<unsigned long long *>(base); base += np * 8;
    i64 *pbase = reinterpret_cast<i64 *>(base); base += np * 8;
    i64 *bsum = reinterpret_cast<i64 *>(base);
    const unsigned g = grid_for(n);
    hipLaunchKernelGGL(k_path_fill, dim3(g), dim3(256), 0, stream, prv, n, PATH_NONE);
    if (lookahead > 0) {
        // weights in the room of `len` / `pbase` (not in use yet): four byte arrays
        unsigned char *W = reinterpret_cast<unsigned char *>(len);
        unsigned char *F[2] = {W, W + np}, *G[2] = {W + 2 * np, W + 3 * np};
        (void)hipMemsetAsync(W, 0, (size_t)np * 4, stream);
        int w = 0;
        for (int round = 0; round < lookahead; round++, w ^= 1)
            hipLaunchKernelGGL(k_path_weigh, dim3(g), dim3(256), 0, stream, ix, F[w], F[w ^ 1], G[w], G[w ^ 1]);
        hipLaunchKernelGGL(k_path_succ_weighted, dim3(g), dim3(256), 0, stream, ix, F[w], G[w], succ, sch, prv);
    } else {
        hipLaunchKernelGGL(k_path_succ, dim3(g), dim3(256), 0, stream, ix, succ, sch, prv);
    }
    derived_log(stream, "path order: successors chosen, columns", n);
    int rounds = 1;
    while (((i64)1 << rounds) < n) rounds++;
    rounds++;
    int cur = 0;
    // (SBWTGPU_PATH_RANK=0: the doubling over every column, which is also the fallback; 2: splitters even where most columns are
    // heads; read per call so that a test can build all three)
    const int rank_splitters = [] { const char *e = getenv("SBWTGPU_PATH_RANK"); return e ? atoi(e) : 1; }();
    // (SBWTGPU_PATH_RANK_LIMIT: the longest walk, for tests of the way back to the doubling)
    const unsigned rank_limit = [] { const char *e = getenv("SBWTGPU_PATH_RANK_LIMIT"); return e && atoi(e) > 0 ? (unsigned)atoi(e) : (unsigned)RANK_LIMIT; }();
    int *rflag = flag + 8;
    u64 *m_dev = reinterpret_cast<u64 *>(flag + 32);
    unsigned *S = reinterpret_cast<unsigned *>(len);    // (`len` is not in use before k_path_len: 8 bytes per column)
    for (int attempt = 0; attempt < 3; attempt++) {
        cur = 0;
        bool ranked = false;
        if (rank_splitters) {
            hipLaunchKernelGGL(k_path_keep, dim3(g), dim3(256), 0, stream, n, succ, prv, (uint4 *)nullptr);
            (void)hipMemsetAsync(flag, 0, 4096, stream);
            hipLaunchKernelGGL(k_rank_list, dim3(g), dim3(256), 0, stream, n, prv, S, m_dev, jb[0], jb[1]);
            u64 m = 0;
            if (hipMemcpyAsync(&m, m_dev, 8, hipMemcpyDeviceToHost, stream) != hipSuccess) return -1;
            if (hipStreamSynchronize(stream) != hipSuccess) return -1;
            if (m > 0 && (m <= (u64)n / 8 || rank_splitters >= 2)) {      // (2: whatever the share of splitters -- tests)
                const unsigned gm = (unsigned)((m + 255) / 256);
                hipLaunchKernelGGL(k_rank_walk1, dim3(gm), dim3(256), 0, stream, m_dev, S, succ, jb[0], rflag, rank_limit);
                int rr = 1;
                while (((u64)1 << rr) < m) rr++;
                rr++;
                for (int r = 0; r < rr; r++, cur ^= 1)
                    hipLaunchKernelGGL(k_rank_jump, dim3(gm), dim3(256), 0, stream, m_dev, S, jb[cur], jb[cur ^ 1]);
                hipLaunchKernelGGL(k_rank_walk2, dim3(gm), dim3(256), 0, stream, m_dev, S, succ, jb[cur], rflag, rank_limit);
                hipLaunchKernelGGL(k_rank_leftover, dim3(g), dim3(256), 0, stream, n, succ, jb[cur], rflag, rank_limit);
                int h_r = 0;
                if (hipMemcpyAsync(&h_r, rflag, 4, hipMemcpyDeviceToHost, stream) != hipSuccess) return -1;
                if (hipStreamSynchronize(stream) != hipSuccess) return -1;
                ranked = (h_r == 0);
                derived_log(stream, ranked ? "path order: ranked by splitters, attempt" : "path order: the splitter ranking gave up, attempt", attempt, m_dev);
            }
        }
        if (!ranked) {
        cur = 0;
        hipLaunchKernelGGL(k_path_keep, dim3(g), dim3(256), 0, stream, n, succ, prv, jb[0]);
        for (int r = 0; r < rounds; r++) {
            // from the 8th round on, every other round asks whether any pointer still moved: most graphs' longest path is far
            // shorter than 2^rounds (cycles never settle and take all rounds; k_path_cut then opens them)
            const bool ask = r >= 7 && (r & 1);
            if (ask) (void)hipMemsetAsync(moved, 0, 4, stream);
            hipLaunchKernelGGL(k_path_jump, dim3(g), dim3(256), 0, stream, n, jb[cur], jb[cur ^ 1], ask ? moved : (int *)nullptr);
            cur ^= 1;
            if (ask) {
                int h_moved = 1;
                if (hipMemcpyAsync(&h_moved, moved, 4, hipMemcpyDeviceToHost, stream) != hipSuccess) return -1;
                if (hipStreamSynchronize(stream) != hipSuccess) return -1;
                if (!h_moved) break;
            }
        }
        }   // !ranked
        (void)hipMemsetAsync(flag, 0, 4, stream);
        hipLaunchKernelGGL(k_path_cut, dim3(g), dim3(256), 0, stream, n, jb[cur], prv, succ, flag);
        int h_flag = 0;
        if (hipMemcpyAsync(&h_flag, flag, 4, hipMemcpyDeviceToHost, stream) != hipSuccess) return -1;
        if (hipStreamSynchronize(stream) != hipSuccess) return -1;
        derived_log(stream, "path order: cycles cut (flag)", h_flag);
        if (!h_flag) break;
        if (attempt == 2) return -2;                    // cannot happen: one cut per cycle opens every cycle
    }
    (void)hipMemsetAsync(len, 0, (size_t)np * 8, stream);
    hipLaunchKernelGGL(k_path_len, dim3(g), dim3(256), 0, stream, n, jb[cur], succ, len);
    hipLaunchKernelGGL(k_scan_block_sums, dim3((unsigned)nb), dim3(256), 0, stream, reinterpret_cast<const i64 *>(len), n, bsum);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, stream, bsum, nb);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(256), 0, stream, reinterpret_cast<const i64 *>(len), n, bsum, pbase);
    derived_log(stream, "path order: path lengths scanned", n);
    *n_pos = n;
    if (stitch && pos_cap > n) {
        unsigned *col_o = nullptr, *pos_o = nullptr;
        uint4 *pq_o = nullptr;
        long long got = -1;
        if (hipMalloc((void **)&col_o, (size_t)(n + 4) * 4) == hipSuccess && hipMalloc((void **)&pos_o, (size_t)(n + 4) * 4) == hipSuccess &&
            hipMalloc((void **)&pq_o, (size_t)sbwt_path_quads(n) * 16) == hipSuccess) {
            (void)hipMemsetAsync(pq_o, 0, (size_t)sbwt_path_quads(n) * 16, stream);
            hipLaunchKernelGGL(k_path_place, dim3(g), dim3(256), 0, stream, n, jb[cur], pbase, succ, sch, pos_o, col_o,
                               reinterpret_cast<unsigned *>(pq_o));
            got = stitch_paths(ix, col_o, pos_o, pq_o, d_col, d_pos, d_pq, pos_cap, min_copy, stream);
        } else {
            (void)hipGetLastError();
        }
        (void)hipFree(col_o); (void)hipFree(pos_o); (void)hipFree(pq_o);
        if (got >= n) { *n_pos = got; return 0; }
        // (no room or a failure: the disjoint paths as they are)
    }
    (void)hipMemsetAsync(d_pq, 0, (size_t)sbwt_path_quads(pos_cap > n ? pos_cap : n) * 16, stream);
    hipLaunchKernelGGL(k_path_place, dim3(g), dim3(256), 0, stream, n, jb[cur], pbase, succ, sch, d_pos,
                       d_col, reinterpret_cast<unsigned *>(d_pq));
    if (hipStreamSynchronize(stream) != hipSuccess) return -1;
    derived_log(stream, "path order: placed", n);
    return 0;
}
// The ONLY bits, then the path groups' final encoding (k_path_reencode); returns the number of transition entries
// (*n_branch: columns with two or more successors).  Synchronises the stream.  -1 on error.
// count_only: the entries and the branching columns only (the table's size, which the image's final layout needs before the sparse
// tables are built); the path groups stay in their build-time encoding
long long sbwt_launch_path_oth(const SbwtIndexView &ix, uint4 *d_pq, long long *n_branch, hipStream_t stream, int count_only) {
    unsigned long long *d = nullptr, h[2] = {0, 0};
    unsigned *only = nullptr;
    const i64 n_quads = sbwt_path_quads(ix.n_pos);
    if (hipMalloc((void **)&d, 16) != hipSuccess) return -1;
    if (hipMalloc((void **)&only, (size_t)n_quads * 4) != hipSuccess) { (void)hipFree(d); return -1; }
    (void)hipMemsetAsync(d, 0, 16, stream);
    (void)hipMemsetAsync(only, 0, (size_t)n_quads * 4, stream);
    hipLaunchKernelGGL(k_path_oth, dim3(grid_for(ix.n_pos)), dim3(256), 0, stream, ix, only, d);
    if (!count_only) hipLaunchKernelGGL(k_path_reencode, dim3(grid_for(n_quads)), dim3(256), 0, stream, d_pq, n_quads, only);
    hipError_t e = hipMemcpyAsync(h, d, 16, hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(d);
    (void)hipFree(only);
    if (e != hipSuccess) return -1;
    if (n_branch) *n_branch = (long long)h[1];
    return (long long)h[0];
}
// Fills the transition table of n_slots 32-byte entries (zeroed here).  d_alt_safe: the per-substitute verdicts of
// k_path_safe_labels (a byte per position), or nullptr.
void sbwt_launch_trans_insert(const SbwtIndexView &ix, uint4 *d_trans, long long n_slots, const unsigned char *d_alt_safe,
                              hipStream_t stream) {
    (void)hipMemsetAsync(d_trans, 0, (size_t)32 * (size_t)n_slots, stream);
    hipLaunchKernelGGL(k_trans_insert, dim3(grid_for(ix.n_pos)), dim3(256), 0, stream, ix, reinterpret_cast<unsigned *>(d_trans), (unsigned)n_slots,
                       d_alt_safe);
}
