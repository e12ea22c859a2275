// sbwt_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4) for the plain-matrix SBWT
// k-mer search path.  Integer / bit-manipulation only; the bound is the rate of 64-byte fabric
// requests (random gathers from the index image, result lines), so the design rules are: one dependent
// memory round trip per query step, 16-byte vector loads issued unconditionally and back to back, as
// few requests per k-mer as the data structure allows, all 64 lanes of a wave kept busy by a work
// queue, whole-line result writes, no MFMA.
//
// Contents, in file order:
//   k_encode                     ASCII bases -> packed groups of 32 bases
//   k_search                     lane-per-read search in the reference's order of searches (variant 0, cross-check)
//   k_search_cert<WIDE,WPS,PATH> the product kernel: certificates; PATH = along the path order
//   k_rank, k_precalc, k_update_interval, k_forward    the batched API neighbours
//   k_sg_*                       suffix-group marks derived on the device (indexes without streaming support)
//   k_sp_*, k_sp2_*, k_pf_insert sparse prefix table (+ second level for 31 < k <= 63), probe filter
//   k_path_*                     path order, packed path chars, transition table
//   k_fmt_*, k_scan_*            print_vector on the device
//   sbwt_launch_*                host launchers
//
// Reference semantics restated here (paths relative to the reference repo):
//   SBWT::streaming_search      include/sbwt/SBWT.hh:544-581
//   SBWT::search                include/sbwt/SBWT.hh:389-415
//   SBWT::update_sbwt_interval  include/sbwt/SBWT.hh:422-437
//   SBWT::forward               include/sbwt/SBWT.hh:368-381
//   SBWT::do_kmer_prefix_precalc include/sbwt/SBWT.hh:616-645
//   SubsetMatrixRank::rank      include/sbwt/SubsetMatrixRank.hh:31-37
#include <hip/hip_runtime.h>
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
__device__ __forceinline__ void st_stream(i64 *p, i64 v) { __builtin_nontemporal_store(v, p); }
// two consecutive results / columns at their natural (8-byte / 4-byte) alignment
typedef i64 i64x2_a8 __attribute__((ext_vector_type(2), aligned(8)));
typedef unsigned u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ void st_stream2(i64 *p, i64 a, i64 b) {
    i64x2_a8 v = {a, b};
    __builtin_nontemporal_store(v, reinterpret_cast<i64x2_a8 *>(p));
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

// ---------------------------------------------------------------------------------------------
// k_encode: ASCII bases -> packed groups of 32 bases (coalesced streaming pre-pass)
//   group = { codes[31:0], codes[63:32], validU, validRaw }
//   codes   2 bits per base, base t of the group at bits 2t..2t+1 (toupper'd, globals.hh:38-47)
//   validU  bit t set iff toupper(base) is ACGT   (streaming step validates this: SBWT.hh:565-568)
//   validRaw bit t set iff base itself is ACGT    (search validates the raw char: SBWT.hh:398-399,427-428)
// Also resets the workspace header for the search launch that follows on the same stream.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_encode(const unsigned char *__restrict__ bases, i64 total,
                                                uint4 *__restrict__ packed, i64 n_groups,
                                                SbwtWorkHeader *ws, int aligned16) {
    i64 g = (i64)blockIdx.x * 256 + threadIdx.x;
    if (g == 0) { ws->ticket = 0; ws->status = 0; ws->n_stream = 0; ws->n_search = 0; ws->n_lf = 0; ws->n_tab_hit = 0; ws->n_ext = 0; ws->n_bridge = 0; }
    if (g >= n_groups) return;
    i64 base = g * SBWT_GROUP_BASES;
    u64 codes = 0;
    unsigned vu = 0, vr = 0;
    if (aligned16 && base + SBWT_GROUP_BASES <= total) {
        const uint4 *src = reinterpret_cast<const uint4 *>(bases + base);
        uint4 x0 = ld_stream(src), x1 = ld_stream(src + 1);
        unsigned w[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
        for (int d = 0; d < 8; d++) {
#pragma unroll
            for (int t = 0; t < 4; t++) {
                unsigned b = (w[d] >> (8 * t)) & 0xFFu;
                unsigned up = b & 0xDFu;
                int pos = d * 4 + t;
                codes |= (u64)dna_code(up) << (2 * pos);
                vu |= (unsigned)is_ACGT(up) << pos;
                vr |= (unsigned)is_ACGT(b) << pos;
            }
        }
    } else {
        for (int pos = 0; pos < SBWT_GROUP_BASES; pos++) {
            if (base + pos < total) {
                unsigned b = bases[base + pos];
                unsigned up = b & 0xDFu;
                codes |= (u64)dna_code(up) << (2 * pos);
                vu |= (unsigned)is_ACGT(up) << pos;
                vr |= (unsigned)is_ACGT(b) << pos;
            }
        }
    }
    packed[g] = make_uint4((unsigned)codes, (unsigned)(codes >> 32), vu, vr);
}

// ---------------------------------------------------------------------------------------------
// k_search: streaming_search (streaming != 0) or the per-k-mer search loop (streaming == 0).
//
// One lane owns one read at a time and walks it as a small state machine; every loop iteration
// performs exactly one dependent gather from the index image for every busy lane, whatever its
// state, so that lanes in different states do not serialise their memory round trips:
//   M_STREAM  one-step extension of the previous answer (SBWT.hh:562-575): the 32-byte quad
//             pair holding the suffix-group word and column c of the block of the previous answer
//   M_INIT    start of a full search (SBWT.hh:389-404): one 16-byte prefix-table entry
//   M_STEP    one update_sbwt_interval character (SBWT.hh:425-433): quad c at `first` and, if it
//             lies in another block, quad c at `second+1`
// Finished lanes pull the next read from a device-wide ticket counter, so lanes stay busy although
// a read with a mismatch costs ~10x one without.
// ---------------------------------------------------------------------------------------------
#define M_IDLE 0
#define M_STREAM 1
#define M_INIT 2
#define M_STEP 3
#define M_DEAD 4

template <bool MEGA>
__global__ void __launch_bounds__(256) k_search(SbwtIndexView ix, const uint4 *__restrict__ packed,
                                                const i64 *__restrict__ read_off,
                                                const i64 *__restrict__ out_off, i64 *__restrict__ out,
                                                i64 n_reads, SbwtWorkHeader *ws, int streaming) {
    const int lane = threadIdx.x & 63;
    const int k = ix.k, p = ix.p_dev;
    const i64 last_node = ix.n_nodes - 1;

    int mode = M_IDLE;
    bool dead = false;
    i64 P0 = 0, obase = 0;
    int m = 0, i = 0, j = 0;
    i64 a = -1, l = 0, r = 0;
    i64 tag = -1;
    uint4 g0 = make_uint4(0, 0, 0, 0), g1 = make_uint4(0, 0, 0, 0);
    unsigned c_stream = 0, c_search = 0, c_lf = 0, c_tab = 0;   // per-lane work counters

    for (;;) {
        // ---- hand out reads to idle lanes ----
        bool want = (mode == M_IDLE) && !dead;
        u64 need = __ballot(want);
        if (need) {
            int leader = __ffsll((i64)need) - 1;
            u64 first = 0;
            if (lane == leader) first = atomicAdd(&ws->ticket, (u64)__popcll(need));
            first = __shfl(first, leader);
            if (want) {
                u64 rd = first + (u64)__popcll(need & low_mask(lane));
                if (rd < (u64)n_reads) {
                    i64 s = read_off[rd], e = read_off[rd + 1];
                    P0 = s;
                    obase = out_off[rd];
                    m = (int)(e - s) - k + 1;
                    i = 0;
                    if (m > 0) {
                        if (p > 0) mode = M_INIT;
                        else { mode = M_STEP; l = 0; r = last_node; j = 0; c_search++; }
                    }
                } else {
                    dead = true;
                }
            }
        }
        if (__ballot(!dead) == 0) break;

        // ---- bases: make sure the group pair holding the next needed base(s) is in registers ----
        const bool act = (mode != M_IDLE);
        const int q = (mode == M_STREAM) ? (i + k - 1) : ((mode == M_INIT) ? i : (i + j));
        const i64 P = P0 + q;
        if (act && (P >> 5) != tag) {
            tag = P >> 5;
            g0 = packed[tag];
            g1 = packed[tag + 1];
        }
        const int s = (int)(P & 31);
        const u64 codes0 = quad_bits(g0);
        const int c = (int)((codes0 >> (2 * s)) & 3ull);

        // ---- form this iteration's gather addresses ----
        const uint4 *a1 = nullptr, *a2 = nullptr;
        bool emit = false;
        i64 res = -1;
        if (mode == M_STREAM) {
            c_stream++;
            if ((g0.z >> s) & 1u) {
                a1 = ix.blocks + (((a >> 6) << 2) + (c & 2));
                a2 = a1 + 1;
            } else {
                emit = true;   // non-ACGT after toupper -> -1 (SBWT.hh:568)
            }
        } else if (mode == M_INIT) {
            c_search++;
            u64 w = codes0 >> (2 * s);
            if (s) w |= quad_bits(g1) << (64 - 2 * s);
            u64 vr = (((u64)g1.w << 32) | (u64)g0.w) >> s;
            u64 vm = low_mask(p);
            if ((vr & vm) == vm) a1 = reinterpret_cast<const uint4 *>(ix.ptab + (w & low_mask(2 * p)));
            else emit = true;  // non-ACGT among the first p chars (SBWT.hh:398-399)
        } else if (mode == M_STEP) {
            if ((g0.w >> s) & 1u) {
                a1 = ix.blocks + (((l >> 6) << 2) + c);
                const uint4 *t = ix.blocks + ((((r + 1) >> 6) << 2) + c);
                if (t != a1) a2 = t;
            } else {
                emit = true;   // raw char invalid (SBWT.hh:427-428)
            }
        }

        // ---- the one dependent round trip of this iteration ----
        uint4 v1 = make_uint4(0, 0, 0, 0), v2 = make_uint4(0, 0, 0, 0);
        if (a1) v1 = *a1;
        if (a2) v2 = *a2;

        // ---- consume ----
        if (a1) {
            if (mode == M_STREAM) {
                u64 ss = (u64)v1.w | ((u64)v2.w << 32);
                uint4 mine = (c & 1) ? v2 : v1;
                const int b = (int)(a & 63);
                u64 msk = ss & ((2ull << b) - 1ull);
                i64 blk = a >> 6;
                while (msk == 0) {   // suffix group starts in an earlier block (rare)
                    if (blk == 0) { msk = 1; break; }   // cannot happen: column 0 is always marked
                    blk--;
                    const uint4 *pa = ix.blocks + ((blk << 2) + (c & 2));
                    uint4 e = pa[0], o = pa[1];
                    msk = (u64)e.w | ((u64)o.w << 32);
                    mine = (c & 1) ? o : e;
                }
                const int gb = 63 - __clzll((i64)msk);
                const u64 bits = quad_bits(mine);
                u64 val = (u64)mine.z + (u64)__popcll(bits & low_mask(gb));
                if (MEGA) val += ix.mega[(i64)c * ix.n_mega + (((blk << 6) | gb) >> SBWT_MEGA_SHIFT)];
                // node_left == node_right  <=>  column c has its bit set at the group start (SBWT.hh:572-575)
                res = ((bits >> gb) & 1ull) ? (i64)val : -1;
                emit = true;
            } else if (mode == M_INIT) {
                l = (i64)quad_bits(v1);
                r = (i64)((u64)v1.z | ((u64)v1.w << 32));
                c_tab += (l != -1);
                if (l == -1) {
                    emit = true;                       // SBWT.hh:424
                } else if (p == k) {
                    emit = true;
                    res = l;
                    if (l != r) ws->status = SBWT_ERR_NOT_SINGLETON;
                } else {
                    j = p;
                    mode = M_STEP;
                }
            } else {   // M_STEP
                if (!a2) v2 = v1;
                c_lf++;
                u64 va = quad_rank<MEGA>(ix, v1, l, c);
                u64 vb = quad_rank<MEGA>(ix, v2, r + 1, c);
                l = (i64)va;
                r = (i64)vb - 1;
                if (l > r) {
                    emit = true;                       // SBWT.hh:433
                } else if (++j == k) {
                    emit = true;
                    res = l;
                    if (l != r) ws->status = SBWT_ERR_NOT_SINGLETON;   // SBWT.hh:410-413
                }
            }
        }

        // ---- write one k-mer result and pick the next state ----
        if (emit) {
            out[obase + i] = res;
            i++;
            if (i == m) {
                mode = M_IDLE;
            } else if (streaming && res != -1) {
                mode = M_STREAM;                       // SBWT.hh:560-
                a = res;
            } else if (p > 0) {
                mode = M_INIT;                         // SBWT.hh:557-559
            } else {
                mode = M_STEP; l = 0; r = last_node; j = 0;   // SBWT.hh:408
                c_search++;
            }
        }
    }

    // ---- work counters: wave reduction, one atomic per counter per wave ----
    u64 t0 = c_stream, t1 = c_search, t2 = c_lf, t3 = c_tab;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        t0 += __shfl_down(t0, off);
        t1 += __shfl_down(t1, off);
        t2 += __shfl_down(t2, off);
        t3 += __shfl_down(t3, off);
    }
    if (lane == 0) {
        atomicAdd(&ws->n_stream, t0);
        atomicAdd(&ws->n_search, t1);
        atomicAdd(&ws->n_lf, t2);
        atomicAdd(&ws->n_tab_hit, t3);
    }
}

// ---------------------------------------------------------------------------------------------
// k_search_cert: same results as k_search, far fewer gathers, one memory round trip per iteration.
//
// (1) Absent-substring certificates.  After a miss the reference runs a full search for every
//     following k-mer; on a read with one substituted base that is ~k failing searches which all
//     die at the same base.  A walk (prefix table + interval updates) that starts at read position
//     s and becomes empty at position t proves that read[s..t] occurs nowhere in the index, hence
//     EVERY k-mer that contains [s..t] is absent -- the reference would print -1 for each of them
//     (a k-mer's result is SBWT::search(kmer) whenever the previous result is -1, SBWT.hh:557-559,
//     and search() of an absent k-mer is -1).  So when the position b of the last failure lies
//     inside the window of the next unresolved k-mer i, the walk is started close to b instead of
//     at i (a short probe ending at b, then a probe starting at b): one short walk certifies the
//     whole run [i..s] at once.  Walk starts are only a heuristic; a walk that stays alive to the
//     end of k-mer i's window proves nothing for s > i and is followed by the reference's own walk
//     from i, so results never depend on the heuristic.
// (2) One gather slot per iteration.  Whatever a lane needs next -- a new read's offsets, the packed
//     group holding its next base, the block of a streaming step, the block left of it when the
//     suffix group starts there, a prefix-table entry, the two quads of an interval update -- is
//     loaded in the single load/wait point of the loop, so a wave never serialises several memory
//     round trips in one iteration because different lanes need different things
//     (profiles/r01_v1_rocprof_summary.txt: the first kernel spent ~8 us per wave iteration).
//     Reads are handed out from a per-wave pool refilled by one atomic per 64 reads.
// (3) Results are staged per lane in LDS and leave as line-aligned runs written by groups of lanes, so
//     that a result costs a share of a whole-line write instead of one memory write per 8 bytes.
// ---------------------------------------------------------------------------------------------
#define M_FETCH 5
#define M_BACK 6
#define M_EXT 7                 // PATH: follow the path from position r while the read agrees with it
#define M_TRANS 8               // PATH: the read left the path at position r: the streaming step, from the transition table
#define M_POS 9                 // PATH: r = pos[l]  (a k-mer was found by a walk: onto its path)
#define M_BRIDGE 10             // PATH: the read differs from the path at a substitution-safe base: do the next k-1 agree?
#define EV_NONE 0
#define EV_EMIT1 1
#define EV_FAIL 2
#define EV_END 3
#define EV_PRES 4               // a range probe's window is in the index: the bad base is not inside it
#define K_NONE 0
#define K_FETCH 1
#define K_RELOAD 2
#define K_MODE 3
#define STAGE_DEPTH 8
#ifndef SBWT_COPY_PIPE
#define SBWT_COPY_PIPE 4
#endif

// WIDE = false: every column index fits 31 bits (n_nodes < 2^31 - 64): positions, results and the LDS
// stage are 32-bit (16 staged results = one 128-byte line per flush).  WIDE = true: 64-bit
// positions, mega-block counts, 8 staged 64-bit results.
template <bool WIDE> struct SearchTypes;
template <> struct SearchTypes<false> { typedef int pos_t; typedef unsigned stage_t; static constexpr int DEPTH = 16; };
template <> struct SearchTypes<true> { typedef i64 pos_t; typedef u64 stage_t; static constexpr int DEPTH = 8; };

template <bool WIDE>
__device__ __forceinline__ typename SearchTypes<WIDE>::pos_t quad_rank_t(const SbwtIndexView &ix, const uint4 &q,
                                                                        typename SearchTypes<WIDE>::pos_t pos, int c) {
    typedef typename SearchTypes<WIDE>::pos_t pos_t;
    pos_t v = (pos_t)q.z + (pos_t)__popcll(quad_bits(q) & low_mask((int)(pos & 63)));
    if (WIDE) v += (pos_t)ix.mega[(i64)c * ix.n_mega + ((i64)pos >> SBWT_MEGA_SHIFT)];
    return v;
}

// (4) PATH = true (32-bit indexes with a path order, see k_path_*): after a k-mer is found at column l
//     the lane moves to its path position and resolves up to 32 following k-mers per iteration by a
//     2-bit compare of the read against the path's chars; their answers are the contiguous run
//     col[t+1..], copied to `out` by the wave together.  Where the read leaves the path the streaming
//     step is one 32-byte entry of the transition table (the four successors of position t, as columns
//     and as path positions).  A third load per iteration prefetches the next packed group of the
//     read, so that 32-base windows rarely wait for a reload.
template <bool WIDE, int WPS, bool PATH>
__global__ void __launch_bounds__(256, WPS) k_search_cert(SbwtIndexView ix, const uint4 *__restrict__ packed,
                                                        const i64 *__restrict__ read_off,
                                                        const i64 *__restrict__ out_off, i64 *__restrict__ out,
                                                        i64 n_reads, SbwtWorkHeader *ws, int streaming) {
    typedef typename SearchTypes<WIDE>::pos_t pos_t;
    typedef typename SearchTypes<WIDE>::stage_t stage_t;
    constexpr int DEPTH = SearchTypes<WIDE>::DEPTH;
    __shared__ stage_t stage[DEPTH][256];
    __shared__ uint4 desc[PATH ? 4 : 1][PATH ? 128 : 1];      // PATH: run descriptors, per wave
    const int tid = threadIdx.x, lane = tid & 63;
    const int k = ix.k, p = ix.p_dev, L0 = ix.probe_len;
    const int ps = WIDE ? 0 : ix.p_sparse;          // sparse table: 32-bit intervals only
    const bool pfon = ix.pfil && ix.p_filter == L0 && L0 > p;
    const u64 m2 = (k - ps >= 32) ? ~0ull : low_mask(2 * ((k - ps) & 31));   // key mask of the second-level window
    const int pw = pfon ? L0 : p;                   // window of a range probe: the filter's when there is one
    const pos_t last_node = (pos_t)(ix.n_nodes - 1);

    int wk = 0;                     // how this walk starts: 0 dense prefix table, 1 sparse table (walks from k-mer i
                                    // itself), 2 probe filter (certificate probes; a "maybe" falls back to 0),
                                    // 3 range probe (is the bad base inside this window?),
                                    // 5 second-level sparse lookup (31 < k <= 63: l = the 31-prefix's first column)
    u64 hk = 0;                     // M_INIT: the window's key (filter: the bit positions), kept across the gather
    int blo = -1;                   // the last failure is known to lie in [blo, b] (blo >= b: exactly at b)
    int mode = M_IDLE;              // M_DEAD once the ticket counter has run past the last read
    i64 obase = 0;                  // first result slot of the current read
    int pgrp = 0, poff = 0;         // the read starts at base poff of packed group pgrp
    int m = 0, i = 0, j = 0, b = -1, wstart = 0, cnt = 0;
    pos_t l = 0, r = 0;             // walk interval; M_STREAM: l = previous answer; M_BACK: r = block
    i64 rd = 0;                     // M_FETCH: the read whose offsets are being fetched
    int tag = -2;                   // g0 = packed group `tag`; g1 = group tag+1 if g1ok
    bool g1ok = false;
    unsigned c_ext = 0;             // PATH: k-mers answered along paths (per lane)
    unsigned c_brg = 0;             // PATH: substitutions bridged (per lane)
    uint4 g0 = make_uint4(0, 0, 0, 0), g1 = make_uint4(0, 0, 0, 0);
    u64 pool_next = 0, pool_end = 0;                              // wave-uniform pool of read tickets
    unsigned c_stream = 0, c_search = 0, c_lf = 0, c_tab = 0;     // wave-uniform (scalar) work counters

    for (;;) {
        // ---- hand out reads to idle lanes from the wave's ticket pool ----
        const u64 need = __ballot(mode == M_IDLE);
        if (need) {
            if (pool_next == pool_end) {
                u64 t = 0;
                if (lane == 0) t = atomicAdd(&ws->ticket, 64ull);
                pool_next = uniform64(t);              // lane 0's value, kept in scalar registers
                pool_end = pool_next + 64;
            }
            const unsigned avail = (unsigned)(pool_end - pool_next);
            const unsigned n = (unsigned)__popcll(need);
            const unsigned rank = (unsigned)__popcll(need & low_mask(lane));
            if (mode == M_IDLE && rank < avail) {
                rd = (i64)(pool_next + rank);
                mode = (rd < n_reads) ? M_FETCH : M_DEAD;
            }
            pool_next = uniform64(pool_next + ((n < avail) ? n : avail));
        }
        if (__ballot(mode != M_DEAD) == 0) break;

        // ---- what does this lane gather this iteration?  Always two 16-byte loads; lanes that need
        //      one (or none) load a duplicate (or the first block), so that the wave issues both
        //      loads back to back and waits once. ----
        int kind = K_NONE, ev = EV_NONE, tfail = 0, c = 0, grp = 0;
        const uint4 *a1 = ix.blocks, *a2 = ix.blocks;
        pos_t res = -1;
        const bool strm = !PATH && (mode == M_STREAM || mode == M_BACK);
        const bool ext = PATH && (mode == M_EXT);
        const bool trn = PATH && (mode == M_TRANS);
        const bool brg = PATH && (mode == M_BRIDGE);
        bool rknown = false;                           // PATH: this iteration's answer came with its path position (in r)
        pos_t tpos = -1;
        int seg_n = 0;                                 // PATH: k-mers i .. i+seg_n-1 are col[seg_src ..]
        unsigned seg_src = 0;
        if (mode == M_FETCH) {
            kind = K_FETCH;                            // {read_off[rd], read_off[rd+1]}, {out_off[rd], ..}
            a1 = reinterpret_cast<const uint4 *>(read_off + rd);
            a2 = reinterpret_cast<const uint4 *>(out_off + rd);
        } else if (PATH && mode == M_POS) {
            kind = K_MODE;                             // the aligned 16 bytes holding pos[l]
            a1 = reinterpret_cast<const uint4 *>(ix.pos + ((unsigned)l & ~3u));
            a2 = a1;
        } else if (mode != M_IDLE && mode != M_DEAD) {
            // M_INIT reads the window at wstart (j counts extra hash buckets there); M_STEP the base at wstart + j
            const int woff = (mode == M_INIT && wk == 5) ? ps : 0;   // the second-level window starts after the prefix
            const int P = poff + ((strm || ext || trn) ? (i + k - 1) : brg ? (i + k) : ((mode == M_INIT) ? (wstart + woff) : (wstart + j)));
            const int s = P & 31;
            // bases the table window of this walk covers
            const int wl = (wk == 1) ? ps : (wk == 2) ? L0 : (wk == 3) ? pw : (wk == 5) ? k - ps : p;
            grp = pgrp + (P >> 5);
            if (grp == tag + 1 && g1ok && (mode != M_INIT || s + wl <= 32)) {
                g0 = g1;                               // crossed into the group that is already here
                g1ok = false;
                tag = grp;
            }
            if (grp != tag || (mode == M_INIT && s + wl > 32 && !g1ok)) {
                kind = K_RELOAD;                       // the packed group pair holding the next base(s)
                a1 = packed + grp;
                a2 = a1 + 1;
            } else {
                kind = K_MODE;
                const u64 codes0 = quad_bits(g0);
                c = (int)((unsigned)(codes0 >> (2 * s)) & 3u);
                if (ext || brg) {
                    a1 = ix.pq + (((unsigned)r + (brg ? 1u : 0u)) >> 5);   // the two quads holding path chars r (+1) .. +31
                    a2 = a1 + 1;
                } else if (trn) {
                    if (((streaming == 2 ? g0.w : g0.z) >> s) & 1u) {     // validity as in M_STREAM below
                        a1 = ix.trans + 2 * (size_t)(unsigned)r;
                        a2 = a1 + 1;
                    } else {
                        ev = EV_EMIT1;
                        b = blo = i + k - 1;
                    }
                } else if (strm) {
                    // streaming == 1: SBWT::streaming_search validates the upper-cased char (SBWT.hh:565-568);
                    // streaming == 2: internal streaming inside the search loop keeps SBWT::search's raw-char
                    // validation (SBWT.hh:398-399,427-428)
                    if (((streaming == 2 ? g0.w : g0.z) >> s) & 1u) {
                        const i64 blk = (mode == M_BACK) ? (i64)r : ((i64)l >> 6);
                        a1 = ix.blocks + ((blk << 2) + (c & 2));
                        a2 = a1 + 1;
                    } else {
                        ev = EV_EMIT1;                 // non-ACGT after toupper -> -1 (SBWT.hh:568)
                        b = blo = i + k - 1;
                    }
                } else if (mode == M_INIT) {
                    u64 w = codes0 >> (2 * s);
                    if (s) w |= quad_bits(g1) << (64 - 2 * s);
                    const u64 vr = (((u64)g1.w << 32) | (u64)g0.w) >> s;
                    const u64 vm = low_mask(wl);
                    if ((vr & vm) == vm) {
                        if (wk == 1) {                 // bucket (hash + j) of the sparse table: two entries
                            const u64 key = w & low_mask(2 * ps);
                            hk = key;
                            const u64 bkt = (((key * SBWT_SP_HASH) >> (64 - ix.log2b)) + (u64)j) & low_mask(ix.log2b);
                            a1 = ix.stab + 2 * bkt;
                            a2 = a1 + 1;
                        } else if (wk == 5) {          // second level: (prefix interval, rest of the k-mer) -> one entry
                            hk = w & m2;
                            const u64 bkt = ((sp2_hash((unsigned)l, hk) >> (64 - ix.log2b2)) + (u64)j) & low_mask(ix.log2b2);
                            a1 = ix.stab2 + 2 * bkt;
                            a2 = a1 + 1;
                        } else if (wk == 2 || (wk == 3 && pfon)) {   // the window's block of the probe filter
                            const u64 h = sbwt_pf_hash(w & low_mask(2 * L0));
                            hk = (u64)sbwt_pf_bits(h);
                            a1 = ix.pfil + (h >> (64 - ix.log2f));
                            a2 = a1;
                        } else {
                            a1 = reinterpret_cast<const uint4 *>(ix.ptab + (w & low_mask(2 * p)));
                            a2 = a1;
                        }
                    } else {
                        ev = EV_FAIL;                  // a non-ACGT char inside the table window
                        tfail = wstart + woff + (__ffsll((i64)(~vr & vm)) - 1);
                    }
                } else {   // M_STEP
                    if ((g0.w >> s) & 1u) {
                        a1 = ix.blocks + ((((i64)l >> 6) << 2) + c);
                        a2 = ix.blocks + (((((i64)r + 1) >> 6) << 2) + c);
                    } else {
                        ev = EV_FAIL;                  // SBWT.hh:427-428
                        tfail = wstart + j;
                    }
                }
            }
        }
        const bool have = (kind == K_MODE && ev == EV_NONE);
#ifdef SBWT_STATS
        {   // lane-iterations by kind: pad[8..]: fetch, reload, init, step, trans, pos, ext, idle/dead, waves-iterations
            const int cls = (kind == K_FETCH) ? 0 : (kind == K_RELOAD) ? 1 : (kind == K_NONE) ? 7 :
                            (mode == M_INIT) ? 2 : (mode == M_STEP) ? 3 : (mode == M_TRANS) ? 4 : (mode == M_POS) ? 5 : (mode == M_EXT) ? 6 : 7;
            for (int q = 0; q < 8; q++) {
                const unsigned long long cq = __popcll(__ballot(cls == q));
                if (lane == 0 && cq) atomicAdd(&ws->pad[8 + q], cq);
            }
            if (lane == 0) atomicAdd(&ws->pad[16], 1ull);
        }
#endif
        c_search = uniform32(c_search + (unsigned)__popcll(__ballot(kind == K_MODE && (mode == M_INIT || (p == 0 && mode == M_STEP && j == 0)))));
        c_lf = uniform32(c_lf + (unsigned)__popcll(__ballot(have && mode == M_STEP)));

        // ---- the one memory round trip of this iteration ----
        // PATH: the read's next packed group rides along when it is not here yet
        const bool pf = PATH && !g1ok && kind == K_MODE;
        const uint4 *a3 = pf ? (packed + (tag + 1)) : a1;
        const uint4 v1 = *a1;
        const uint4 v2 = *a2;
        if (PATH) {
            const uint4 v3 = *a3;
            if (pf) { g1 = v3; g1ok = true; }
        }

        // ---- consume ----
        bool tabhit = false, do_plan = false, force = false;
        bool imprecise = false;                        // this iteration's failure is a table-level miss
        int burst_to = -1;                             // M_BRIDGE: k-mers i .. burst_to are certified absent
        if (kind == K_FETCH) {
            const i64 P0 = (i64)quad_bits(v1);
            obase = (i64)quad_bits(v2);
            pgrp = (int)(P0 >> 5);
            poff = (int)(P0 & 31);
            m = (int)((i64)((u64)v1.z | ((u64)v1.w << 32)) - P0) - k + 1;
            i = 0;
            b = -1;
            blo = -1;
            if (m > 0) { do_plan = true; force = true; }
            else mode = M_IDLE;
        } else if (kind == K_RELOAD) {
            g0 = v1;
            g1 = v2;
            g1ok = true;
            tag = grp;
        } else if (PATH && have && mode == M_POS) {
            const unsigned sel = (unsigned)l & 3u;
            r = (pos_t)(sel == 0 ? v1.x : sel == 1 ? v1.y : sel == 2 ? v1.z : v1.w);
            mode = M_EXT;
        } else if (trn && have) {
            // successors of path position r: v1 = their columns, v2 = their path positions (SBWT.hh:562-575)
            const unsigned nc = (c == 0 ? v1.x : c == 1 ? v1.y : c == 2 ? v1.z : v1.w);
            ev = EV_EMIT1;
            if (nc == 0xFFFFFFFFu) {
                b = blo = i + k - 1;
            } else {
                res = (pos_t)nc;
                r = (pos_t)(c == 0 ? v2.x : c == 1 ? v2.y : c == 2 ? v2.z : v2.w);
                rknown = true;
            }
        } else if (brg && have) {
            // k-mer i (ending at the mismatching base) .. : if the bases after it agree with the path again, every
            // k-mer that contains the mismatching base is a one-base variant of a path k-mer, absent by the safe bit
            const int P = poff + i + k, s = P & 31, sp = (int)(((unsigned)r + 1u) & 31u);
            u64 rw = quad_bits(g0) >> (2 * s), pw = quad_bits(v1) >> (2 * sp);
            if (s) rw |= quad_bits(g1) << (64 - 2 * s);
            if (sp) pw |= quad_bits(v2) << (64 - 2 * sp);
            const u64 x = rw ^ pw;
            const u64 mm = (x | (x >> 1)) & 0x5555555555555555ull;
            const int nm = mm ? ((__ffsll((i64)mm) - 1) >> 1) : 32;
            const int need = (k - 1 < m - 1 - i) ? (k - 1) : (m - 1 - i);
            if (nm >= need) {
                ev = EV_FAIL;
                burst_to = i + need;
                c_brg++;
            } else {
                mode = M_TRANS;
            }
        } else if (ext && have) {
            // k-mer i-1 sits at path position r.  Read bases i+k-1.. against path chars r..: while they agree
            // (and the read's bases are valid and the path goes on), k-mer i+x sits at r+1+x.
            const int P = poff + i + k - 1, s = P & 31, sp = (int)((unsigned)r & 31u);
            u64 rw = quad_bits(g0) >> (2 * s), pw = quad_bits(v1) >> (2 * sp);
            if (s) rw |= quad_bits(g1) << (64 - 2 * s);
            if (sp) pw |= quad_bits(v2) << (64 - 2 * sp);
            const u64 rv = ((streaming == 2) ? (((u64)g1.w << 32) | (u64)g0.w) : (((u64)g1.z << 32) | (u64)g0.z)) >> s;
            const u64 pg = (((u64)v2.z << 32) | (u64)v1.z) >> sp;
            const u64 x = rw ^ pw;
            const u64 mm = (x | (x >> 1)) & 0x5555555555555555ull;
            const int nm = mm ? ((__ffsll((i64)mm) - 1) >> 1) : 32;
            const u64 bad = ~(rv & pg) | (1ull << 32);
            const int nv = __ffsll((i64)bad) - 1;
            int n = nm < nv ? nm : nv;
            bool stopped = n < 32;                     // a mismatch, an invalid base or the end of the path
            if (n > m - i) n = m - i;
            if (n > 32 - cnt) { n = 32 - cnt; stopped = false; }   // staged results + run travel in one descriptor
            seg_n = n;
            seg_src = (unsigned)r + 1u;
            r += (pos_t)n;
            c_ext += (unsigned)n;
            bool sbit = false;                         // stopped at a char mismatch whose path step is substitution-safe?
            if (ix.has_safe && stopped && nm < nv) sbit = ((((((u64)v2.w << 32) | (u64)v1.w) >> sp) >> nm) & 1ull) != 0;
            if (i + n == m) mode = M_IDLE;
            else if (stopped) mode = sbit ? M_BRIDGE : M_TRANS;
        } else if (have) {
            if (strm) {
                const i64 blk = (mode == M_BACK) ? (i64)r : ((i64)l >> 6);
                u64 msk = (u64)v1.w | ((u64)v2.w << 32);
                if (mode == M_STREAM) msk &= (2ull << (int)(l & 63)) - 1ull;
                if (msk == 0 && blk > 0) {
                    mode = M_BACK;                     // the suffix group starts in an earlier block
                    r = (pos_t)(blk - 1);
                } else {
                    if (msk == 0) msk = 1;             // cannot happen: column 0 is always marked
                    const uint4 mine = (c & 1) ? v2 : v1;
                    const int gb = 63 - __clzll((i64)msk);
                    const u64 bits = quad_bits(mine);
                    pos_t val = (pos_t)mine.z + (pos_t)__popcll(bits & low_mask(gb));
                    if (WIDE) val += (pos_t)ix.mega[(i64)c * ix.n_mega + (((blk << 6) | gb) >> SBWT_MEGA_SHIFT)];
                    // node_left == node_right <=> column c has its bit set at the group start (SBWT.hh:572-575)
                    res = ((bits >> gb) & 1ull) ? val : (pos_t)-1;
                    ev = EV_EMIT1;
                    if (res == -1) b = blo = i + k - 1;
                }
            } else if (mode == M_INIT) {
                int wl = p;
                bool again = false;
                const bool viaf = (wk == 2) || (wk == 3 && pfon);
                if (wk == 1 || wk == 5 || viaf) {
                  if (viaf) {
                    const unsigned b1 = (unsigned)hk & 127u, b2 = ((unsigned)hk >> 7) & 127u;
                    const unsigned w1 = (b1 < 64) ? (b1 < 32 ? v1.x : v1.y) : (b1 < 96 ? v1.z : v1.w);
                    const unsigned w2 = (b2 < 64) ? (b2 < 32 ? v1.x : v1.y) : (b2 < 96 ? v1.z : v1.w);
                    wl = L0;
                    if (((w1 >> (b1 & 31u)) & (w2 >> (b2 & 31u)) & 1u) != 0) {
                        if (wk == 3) {
                            l = 0;                     // range probe: "perhaps present" only moves the guess
                        } else {
                            again = true;              // perhaps present: the dense table walks the window exactly
                            wk = 0;
                        }
                    } else {
                        l = -1;                        // read[wstart .. wstart+L0-1] is not in the index
                    }
                  } else if (wk == 5) {
                    wl = k;                            // a hit completes the k-mer; a miss says read[wstart .. wstart+k-1] is absent
                    if ((v1.w & SBWT_SP2_USED) && quad_bits(v1) == hk && v1.z == (unsigned)l) {
                        l = (pos_t)v2.x;
                        r = l;
                        if (PATH) tpos = (pos_t)v2.y;
                    } else if (v1.w & SBWT_SP2_OVERFLOW) {
                        again = true;
                        j++;
                    } else {
                        l = -1;
                    }
                  } else {
                    const u64 key = hk;
                    const u64 w0 = quad_bits(v1), w1 = quad_bits(v2);
                    const bool m0 = (w0 & ~SBWT_SP_OVERFLOW) == key, m1 = w1 == key;
                    wl = ps;
                    if (m0 | m1) {
                        l = (pos_t)(m0 ? v1.z : v2.z);
                        if (ix.stab_pos) {             // depth-k entries: one column, stored with its path position
                            r = l;
                            if (PATH) tpos = (pos_t)(m0 ? v1.w : v2.w);
                        } else {
                            r = l + (pos_t)(m0 ? v1.w : v2.w);
                        }
                    } else if (w0 & SBWT_SP_OVERFLOW) {
                        again = true;                  // a later bucket may hold the key
                        j++;
                    } else {
                        l = -1;                        // read[wstart .. wstart+ps-1] is not in the index
                    }
                  }
                } else {
                    l = (pos_t)(i64)quad_bits(v1);
                    r = (pos_t)(i64)((u64)v1.z | ((u64)v1.w << 32));
                }
                if (!again) {
                    tabhit = (l != -1);
                    if (l == -1) {
                        ev = EV_FAIL;                  // read[wstart .. wstart+wl-1] is not in the index
                        tfail = wstart + wl - 1;
                        imprecise = (wk != 2);         // ... but where inside the window it fails is not known
                    } else if (wk == 3) {
                        ev = EV_PRES;
                    } else if (wk == 1 && ps < k && ix.stab2) {
                        wk = 5;                        // the prefix is there (l = its first column): the rest in one more gather
                        j = 0;
                    } else {
                        j = wl;
                        if (wstart + j == i + k) ev = EV_END;
                        else mode = M_STEP;
                    }
                }
            } else {   // M_STEP
                l = quad_rank_t<WIDE>(ix, v1, l, c);
                r = quad_rank_t<WIDE>(ix, v2, r + 1, c) - 1;
                if (l > r) {
                    ev = EV_FAIL;                      // SBWT.hh:433
                    tfail = wstart + j;
                } else if (wstart + (++j) == i + k) {
                    ev = EV_END;
                }
            }
        }
        c_tab = uniform32(c_tab + (unsigned)__popcll(__ballot(tabhit)));
        c_stream = uniform32(c_stream + (unsigned)__popcll(__ballot(ev == EV_EMIT1 && (strm || trn))));

        // ---- events: results, certificates, next state ----
        int burst_hi = -1;                             // >= i: k-mers i..burst_hi are certified absent
        if (ev == EV_END) {
            if (wstart == i) {                         // k chars matched from i: the k-mer is there
                res = l;
                if (l != r) ws->status = SBWT_ERR_NOT_SINGLETON;   // SBWT.hh:410-413
                if (PATH && tpos >= 0) { r = tpos; rknown = true; }
                ev = EV_EMIT1;
                b = -1;
            } else {
                do_plan = true;                        // probe inconclusive: the reference's own walk
                force = true;
            }
        } else if (ev == EV_FAIL) {
            // read[wstart..tfail] is not in the index: k-mers i..min(wstart, m-1) all contain it
            burst_hi = (wstart < m - 1) ? wstart : (m - 1);
            if (burst_to >= 0) {                       // bridged substitution: nothing is known about the next one
                burst_hi = burst_to;
                b = -1;
            } else if (wk == 3) {                             // range probe: the bad base is in [wstart, b]
                if (wstart >= b) b = -1;
                else if (blo < wstart + 1) blo = wstart + 1;
            } else if (imprecise && !(wstart == b && blo >= b)) {
                blo = wstart;                          // the bad base is somewhere in [wstart, tfail]
                b = tfail;
            } else {
                // a walk that started AT the known-bad position says nothing about where the next one is
                b = (wstart == b) ? -1 : tfail;
                blo = b;
            }
            if (burst_hi == i) { ev = EV_EMIT1; burst_hi = -1; }   // a single -1 goes through the stage
        }
        if (ev == EV_PRES) {                           // no bad base in [wstart, wstart+p-1]: shrink the range
            const int lo = blo > i ? blo : i;
            if (wstart > lo) b = wstart - 1;
            else blo = wstart + pw;
            if (blo > b) b = -1;
            do_plan = true;
        }
        if (ev == EV_EMIT1) {
            stage[cnt][tid] = (stage_t)res;
            cnt++;
            i++;
        }
        // ---- result writes, wave-cooperative ----
        // Staged results are those of k-mers [i-cnt, i); they leave as one run that ends on a line
        // boundary of `out` (DEPTH results = one 128-byte / 64-byte line), at a burst, or at the read's
        // end.  A certified burst is a run of -1.  Each run is written by a group of DEPTH lanes with one
        // coalesced store (64/DEPTH runs per store instruction) instead of a per-lane loop of 8-byte
        // stores that would execute in almost every iteration for a handful of lanes.
        if (PATH) {
            // One writer, whole lines.  The results of a read leave in order.  What a lane has not written yet --
            // `cnt` results staged in LDS, always starting on a 64-byte line of `out` (or at the read's first
            // result) -- is joined with this iteration's run (a certified burst of -1, or a path run out of col[]),
            // and the part that ends on a line boundary is written; the tail goes (back) into the lane's stage.
            // So every store covers whole 64-byte lines except at the two ends of a read.
            // A lane with something to write posts a 16-byte descriptor in LDS at its rank among the posting
            // lanes; group g of 16 lanes takes descriptors g, g+4, ... and handles two results per lane
            // (stage + run <= 32 results), PIPE descriptors per trip with their loads in flight together.
            int nleft = (burst_hi >= 0) ? (burst_hi - i + 1) : seg_n;
            unsigned s2 = (burst_hi >= 0) ? 0xFFFFFFFFu : seg_src;
            uint4 *mydesc = desc[tid >> 6];
            const int sub = lane & 15, grpl = lane >> 4;
            const u64 lt = low_mask(lane);
            for (;;) {                                 // more than one round only for bursts longer than a descriptor
                const i64 dst0 = obase + (i - cnt);
                const int nn = nleft < 32 - cnt ? nleft : 32 - cnt;
                const int total = cnt + nn;
                const bool end = (i + nn == m);
                const int over = (int)((unsigned)(dst0 + total) & 7u);   // results past the last line boundary
                const bool post = nn > 0 || (cnt > 0 && (end || over == 0));
                const int w = !post ? 0 : (end ? total : (over <= total ? total - over : 0));
                const u64 pm = __ballot(post);
                if (pm == 0) break;
                const int ndesc = __popcll(pm);
                if (post)
                    mydesc[__popcll(pm & lt)] = make_uint4((unsigned)dst0, (unsigned)((u64)dst0 >> 32),
                                                           (unsigned)cnt | ((unsigned)total << 8) | ((unsigned)w << 16) | ((unsigned)tid << 24), s2);
                for (int base = 0; base < ndesc; base += 4 * SBWT_COPY_PIPE) {
                    constexpr int PIPE = SBWT_COPY_PIPE;
                    i64 dd[PIPE];
                    int sa[PIPE], sb[PIPE], ca[PIPE], cb[PIPE], fa[PIPE], fb[PIPE], wl[PIPE], tl[PIPE], ht[PIPE];
#pragma unroll
                    for (int u = 0; u < PIPE; u++) {
                        const int idx = base + 4 * u + grpl;
                        const bool on = idx < ndesc && !(ix.debug & 1);
                        const uint4 ds = mydesc[idx < ndesc ? idx : 0];
                        const int dc = (int)(ds.z & 0xFFu), dt = (int)((ds.z >> 8) & 0xFFu);
                        const int j0 = 2 * sub;
                        ht[u] = (int)(ds.z >> 24);
                        wl[u] = on ? (int)((ds.z >> 16) & 0xFFu) - j0 : 0;      // results of this lane's pair that go to `out`
                        tl[u] = on ? dt - j0 : 0;                                 // ... that exist at all
                        dd[u] = (i64)((u64)ds.x | ((u64)ds.y << 32)) + j0;
                        if (ix.debug & 4) dd[u] &= 0xFFFE;            // timing experiment: all stores into one small region
                        // both sources are read unconditionally (clamped addresses) and selected afterwards: loads
                        // under divergent branches would be waited for one by one
                        const bool isc = ds.w != 0xFFFFFFFFu && on;
                        const int r0 = j0 < 15 ? j0 : 14;
                        const unsigned c0 = (isc && j0 >= dc && tl[u] > 0) ? ds.w + (unsigned)(j0 - dc) : 0u;
                        const unsigned c1 = (isc && j0 + 1 >= dc && tl[u] > 1) ? ds.w + (unsigned)(j0 + 1 - dc) : 0u;
                        sa[u] = (int)stage[r0][ht[u]];
                        sb[u] = (int)stage[r0 + 1][ht[u]];
                        ca[u] = (int)ix.col[c0];
                        cb[u] = (int)ix.col[c1];
                        fa[u] = (j0 < dc) ? 0 : (isc ? 1 : 2);           // where the value comes from: stage, col, constant
                        fb[u] = (j0 + 1 < dc) ? 0 : (isc ? 1 : 2);
                    }
#pragma unroll
                    for (int u = 0; u < PIPE; u++) {
                        const int va = fa[u] == 0 ? sa[u] : (fa[u] == 1 ? ca[u] : -1);
                        const int vb = fb[u] == 0 ? sb[u] : (fb[u] == 1 ? cb[u] : -1);
                        if (wl[u] >= 2) st_stream2(out + dd[u], (i64)va, (i64)vb);
                        else if (wl[u] == 1) st_stream(out + dd[u], (i64)va);
                        // the tail stays with the holder: stage slot = position past the written part
                        if (tl[u] > 0 && wl[u] < 1) stage[-wl[u]][ht[u]] = (stage_t)va;
                        if (tl[u] > 1 && wl[u] < 2) stage[1 - wl[u]][ht[u]] = (stage_t)vb;
                    }
                }
                cnt = total - w;
                i += nn;
                nleft -= nn;
                if (s2 != 0xFFFFFFFFu) s2 += (unsigned)nn;
                if (__ballot(nleft > 0) == 0) break;
            }
        } else
        {
            const bool fl = (cnt > 0) && (burst_hi >= 0 || seg_n > 0 || i == m || (((unsigned)obase + (unsigned)i) & (DEPTH - 1)) == 0);
            i64 dst = obase + (i - cnt);               // run of staged results
            int nrun = fl ? cnt : 0;
            u64 fm = __ballot(fl);
            const int sub = lane & (DEPTH - 1), grpl = lane / DEPTH;
            constexpr int NG = 64 / DEPTH;
            while (fm) {
                int src = -1;
#pragma unroll
                for (int g = 0; g < NG; g++) {
                    int f = fm ? (__ffsll((i64)fm) - 1) : -1;
                    fm &= fm - 1;
                    src = (grpl == g) ? f : src;
                }
                const int srcl = src < 0 ? 0 : src;
                const i64 d = __shfl(dst, srcl);
                const int nn = __shfl(nrun, srcl);
                if (src >= 0 && sub < nn && !(ix.debug & 1)) {
                    const i64 val = (i64)(pos_t)stage[sub][(tid & ~63) + src];
                    if (ix.debug & 2) out[d + sub] = val;
                    else st_stream(out + d + sub, val);
                }
            }
            if (fl) cnt = 0;
            // certified bursts: k-mers i..burst_hi are -1
            dst = obase + i;
            nrun = (burst_hi >= 0) ? (burst_hi - i + 1) : 0;
            fm = __ballot(nrun > 0);
            while (fm) {
                int src = -1;
                u64 served = 0;
#pragma unroll
                for (int g = 0; g < NG; g++) {
                    int f = fm ? (__ffsll((i64)fm) - 1) : -1;
                    if (f >= 0) served |= 1ull << f;
                    fm &= fm - 1;
                    src = (grpl == g) ? f : src;
                }
                const int srcl = src < 0 ? 0 : src;
                const i64 d = __shfl(dst, srcl);
                const int nn = __shfl(nrun, srcl);
                if (src >= 0 && sub < nn && !(ix.debug & 1)) {
                    if (ix.debug & 2) out[d + sub] = -1;
                    else st_stream(out + d + sub, -1);
                }
                if ((served >> lane) & 1ull) { dst += DEPTH; nrun -= DEPTH; }
                fm = __ballot(nrun > 0);
            }
            if (burst_hi >= 0) i = burst_hi + 1;
        }
        if (ev == EV_EMIT1 || burst_hi >= 0) {
            if (i == m) {
                mode = M_IDLE;
            } else if (ev == EV_EMIT1 && res != -1 && streaming) {
                mode = PATH ? (rknown ? M_EXT : M_POS) : M_STREAM;   // SBWT.hh:560-
                l = res;
            } else {
                do_plan = true;                        // SBWT.hh:557-559 (with certificates)
            }
        }
        if (do_plan) {
            // where the next walk starts (see the header comment): at k-mer i itself, or close to
            // the last failure position b when b lies inside k-mer i's window
            int s0 = i, nwk = (ps > 0) ? 1 : 0;
            if (!force && L0 > 0 && b >= i && b <= i + k - 1) {
                const int lo = blo > i ? blo : i;
                if (lo < b && p > 0 && k - pw >= 1) {
                    // the bad base is somewhere in [lo, b]: halve the range with a window that starts inside it
                    // (absent: k-mers i..x are certified; present: the bad base is left of x)
                    int x = lo + ((b - lo + 1) >> 1);
                    if (x > i + k - pw) x = i + k - pw;
                    if (x <= i) x = i + 1;
                    s0 = x;
                    nwk = 3;
                } else {
                    s0 = (b - i >= L0 - 1) ? (b - L0 + 1) : b;
                    if (s0 + p - 1 > i + k - 1) s0 = i;
                    if (s0 != i) nwk = (pfon && s0 + L0 - 1 <= i + k - 1) ? 2 : 0;
                }
            }
            wstart = s0;
            j = 0;
            // walks from k-mer i itself: sparse table; certificate probes: the filter when the whole probe window
            // lies inside k-mer i's window (a clear bit certifies; otherwise the dense table finds the exact position)
            wk = nwk;
            if (p > 0) mode = M_INIT;
            else { mode = M_STEP; l = 0; r = last_node; }
        }
    }

    if (PATH) {
        u64 e = c_ext, eb = c_brg;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { e += __shfl_down(e, off); eb += __shfl_down(eb, off); }
        if (lane == 0) { atomicAdd(&ws->n_ext, e); atomicAdd(&ws->n_bridge, eb); }
    }
    if (lane == 0) {   // the counters are wave-uniform
        atomicAdd(&ws->n_stream, (u64)c_stream);
        atomicAdd(&ws->n_search, (u64)c_search);
        atomicAdd(&ws->n_lf, (u64)c_lf);
        atomicAdd(&ws->n_tab_hit, (u64)c_tab);
    }
}

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
struct SpItem { u64 key; i64 l; i64 r; };

__global__ void __launch_bounds__(256) k_sp_collect(const longlong2 *__restrict__ ptab, u64 n_entries,
                                                    SpItem *__restrict__ out, u64 *counter) {
    u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_entries) return;
    longlong2 e = ptab[t];
    if (e.x < 0) return;
    u64 slot = atomicAdd(counter, 1ull);
    out[slot] = SpItem{t, e.x, e.y};
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
                                                    int d2, SpItem2 *__restrict__ out, u64 *n_out) {
    u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
    if ((t >> 2) >= *n_in) return;
    const SpItem2 it = in[t >> 2];
    const int c = (int)(t & 3);
    uint4 q1 = ix.blocks[(((i64)it.l >> 6) << 2) + c];
    uint4 q2 = ix.blocks[((((i64)it.r + 1) >> 6) << 2) + c];
    i64 l = (i64)quad_rank<false>(ix, q1, (i64)it.l, c);
    i64 r = (i64)quad_rank<false>(ix, q2, (i64)it.r + 1, c) - 1;
    if (l > r) return;
    u64 slot = atomicAdd(n_out, 1ull);
    out[slot] = SpItem2{it.key2 | ((u64)c << (2 * d2)), it.origin, (unsigned)l, (unsigned)r, 0u};
}
__global__ void __launch_bounds__(256) k_sp2_insert(const SpItem2 *__restrict__ items, const u64 *n, uint4 *table,
                                                    int log2b2, const unsigned *__restrict__ pos, int *wide_flag) {
    u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
    if (t >= *n) return;
    const SpItem2 it = items[t];
    if (it.l != it.r) *wide_flag = 1;                  // a k-mer's interval is one column in an SBWT
    const u64 mask = (1ull << log2b2) - 1ull;
    u64 bkt = sp2_hash(it.origin, it.key2) >> (64 - log2b2);
    for (;;) {
        unsigned *e = reinterpret_cast<unsigned *>(&table[2 * bkt]);
        if (atomicCAS(&e[3], 0u, SBWT_SP2_USED) == 0u) {     // keys are distinct: an empty entry is simply taken
            e[0] = (unsigned)it.key2;
            e[1] = (unsigned)(it.key2 >> 32);
            e[2] = it.origin;
            e[4] = it.l;
            e[5] = pos ? pos[it.l] : 0u;
            return;
        }
        atomicOr(&e[3], SBWT_SP2_OVERFLOW);
        bkt = (bkt + 1) & mask;
    }
}
__global__ void __launch_bounds__(256) k_sp_wide(const SpItem *__restrict__ items, const u64 *n, int *flag) {
    u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
    if (t < *n && items[t].l != items[t].r) *flag = 1;
}
__global__ void __launch_bounds__(256) k_sp_clear(uint4 *table, u64 n_entries) {
    u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
    if (t < n_entries) table[t] = make_uint4(0u, (unsigned)(SBWT_SP_EMPTY >> 32), 0u, 0u);
}
template <bool MEGA>
__global__ void __launch_bounds__(256) k_sp_expand(SbwtIndexView ix, const SpItem *__restrict__ in, const u64 *n_in,
                                                   int depth, SpItem *__restrict__ out, u64 *n_out) {
    u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
    if ((t >> 2) >= *n_in) return;
    const SpItem it = in[t >> 2];
    const int c = (int)(t & 3);
    uint4 q1 = ix.blocks[((it.l >> 6) << 2) + c];
    uint4 q2 = ix.blocks[(((it.r + 1) >> 6) << 2) + c];
    i64 l = (i64)quad_rank<MEGA>(ix, q1, it.l, c);
    i64 r = (i64)quad_rank<MEGA>(ix, q2, it.r + 1, c) - 1;
    if (l > r) return;
    u64 slot = atomicAdd(n_out, 1ull);
    out[slot] = SpItem{it.key | ((u64)c << (2 * depth)), l, r};   // char `depth` of the prefix is c
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
                                                   int log2b, const unsigned *__restrict__ pos) {
    u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
    if (t >= *n) return;
    const SpItem it = items[t];
    const u64 mask = (1ull << log2b) - 1ull;
    u64 bkt = (it.key * SBWT_SP_HASH) >> (64 - log2b);
    for (;;) {
        for (int e = 0; e < 2; e++) {
            u64 *word = reinterpret_cast<u64 *>(&table[2 * bkt + e]);
            u64 old = atomicCAS(word, SBWT_SP_EMPTY, it.key);
            if (old == SBWT_SP_EMPTY) {
                unsigned *pay = reinterpret_cast<unsigned *>(word) + 2;
                pay[0] = (unsigned)it.l;
                pay[1] = pos ? pos[it.l] : (unsigned)(it.r - it.l);
                return;
            }
        }
        atomicOr(reinterpret_cast<u64 *>(&table[2 * bkt]), SBWT_SP_OVERFLOW);   // both entries taken: mark and move on
        bkt = (bkt + 1) & mask;
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
__global__ void __launch_bounds__(256) k_path_keep(i64 n, unsigned *__restrict__ succ, const unsigned *__restrict__ prv,
                                                   unsigned *__restrict__ jump, unsigned *__restrict__ dist,
                                                   unsigned *__restrict__ mn) {
    const i64 v = (i64)blockIdx.x * 256 + threadIdx.x;
    if (v >= n) return;
    const unsigned sv = succ[v];
    if (sv != PATH_NONE && prv[sv] != (unsigned)v) succ[v] = PATH_NONE;
    const unsigned pv = prv[v];
    jump[v] = (pv == PATH_NONE) ? (unsigned)v : pv;     // heads point at themselves
    dist[v] = (pv == PATH_NONE) ? 0u : 1u;
    mn[v] = (unsigned)v;
}
__global__ void __launch_bounds__(256) k_path_jump(i64 n, const unsigned *__restrict__ jin, const unsigned *__restrict__ din,
                                                   const unsigned *__restrict__ min_, unsigned *__restrict__ jout,
                                                   unsigned *__restrict__ dout, unsigned *__restrict__ mout) {
    const i64 v = (i64)blockIdx.x * 256 + threadIdx.x;
    if (v >= n) return;
    const unsigned j = jin[v];
    dout[v] = din[v] + din[j];
    const unsigned a = min_[v], bq = min_[j];
    mout[v] = a < bq ? a : bq;
    jout[v] = jin[j];
}
__global__ void __launch_bounds__(256) k_path_cut(i64 n, const unsigned *__restrict__ jump, const unsigned *__restrict__ mn,
                                                  unsigned *prv, unsigned *succ, int *flag) {
    const i64 v = (i64)blockIdx.x * 256 + threadIdx.x;
    if (v >= n) return;
    // reached a head?  (a cycle's cut point may be cut by its own thread while others look: they then
    // return here, and the flag is raised by the cutting thread)
    if (prv[jump[v]] == PATH_NONE) return;
    if (mn[v] == (unsigned)v) {                         // the cycle's smallest column becomes a head
        const unsigned pv = prv[v];
        if (pv != PATH_NONE) { succ[pv] = PATH_NONE; prv[v] = PATH_NONE; }
        *flag = 1;
    }
}
__global__ void __launch_bounds__(256) k_path_len(i64 n, const unsigned *__restrict__ head, const unsigned *__restrict__ dist,
                                                  unsigned long long *len) {
    const i64 v = (i64)blockIdx.x * 256 + threadIdx.x;
    if (v >= n) return;
    atomicMax(&len[head[v]], (unsigned long long)dist[v] + 1ull);
}
__global__ void __launch_bounds__(256) k_path_place(i64 n, const unsigned *__restrict__ head, const unsigned *__restrict__ dist,
                                                    const i64 *__restrict__ base, const unsigned *__restrict__ succ,
                                                    const unsigned char *__restrict__ sch, unsigned *__restrict__ pos,
                                                    unsigned *__restrict__ col, unsigned *pq) {
    const i64 v = (i64)blockIdx.x * 256 + threadIdx.x;
    if (v >= n) return;
    const unsigned t = (unsigned)base[head[v]] + dist[v];
    pos[v] = t;
    col[t] = (unsigned)v;
    if (succ[v] != PATH_NONE) {                         // quad t>>5 = { chars lo, chars hi, go mask, - }
        unsigned *quad = pq + (size_t)(t >> 5) * 4;
        const unsigned s = t & 31u;
        if (sch[v]) atomicOr(&quad[s >> 4], (unsigned)sch[v] << (2 * (s & 15u)));
        atomicOr(&quad[2], 1u << s);
    }
}

// Substitution-safe bits.  Path index u carries the char ch[u] of the step from position u to u+1.  Bit u says:
// the 2k steps around u lie on one path, and replacing ch[u] by any other base gives, in each of the k windows
// of k chars that contain it, a k-mer that is NOT in the index (3k exact lookups in the depth-k sparse table).
// A read that follows the path, differs from it in exactly the base at u and agrees again on the next k-1 bases
// therefore has -1 for all k k-mers that contain that base -- no probe needed (M_BRIDGE).
__device__ __forceinline__ bool sp_present(const SbwtIndexView &ix, u64 key) {
    u64 bkt = (key * SBWT_SP_HASH) >> (64 - ix.log2b);
    for (;;) {
        const uint4 e0 = ix.stab[2 * bkt], e1 = ix.stab[2 * bkt + 1];
        const u64 w0 = quad_bits(e0), w1 = quad_bits(e1);
        if ((w0 & ~SBWT_SP_OVERFLOW) == key || w1 == key) return true;
        if (!(w0 & SBWT_SP_OVERFLOW)) return false;
        bkt = (bkt + 1) & low_mask(ix.log2b);
    }
}
__global__ void __launch_bounds__(256) k_path_safe(SbwtIndexView ix, unsigned *pq_words) {
    const i64 u = (i64)blockIdx.x * 256 + threadIdx.x;
    const int k = ix.k;
    if (u < k || u + k > ix.n_nodes) return;
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

// transition table: entry t = { columns of the four successors of col[t] } { their path positions }
__global__ void __launch_bounds__(256) k_path_trans(SbwtIndexView ix, const unsigned *__restrict__ pos,
                                                    uint4 *__restrict__ trans) {
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
    unsigned nc[4], np[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const uint4 q = ix.blocks[blk * 4 + c];
        const u64 bits = quad_bits(q);
        nc[c] = ((bits >> gb) & 1ull) ? (q.z + (unsigned)__popcll(bits & low_mask(gb))) : PATH_NONE;
        np[c] = (nc[c] != PATH_NONE) ? pos[nc[c]] : PATH_NONE;
    }
    const size_t t = pos[v];
    trans[2 * t] = make_uint4(nc[0], nc[1], nc[2], nc[3]);
    trans[2 * t + 1] = make_uint4(np[0], np[1], np[2], np[3]);
}

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

// exclusive scan of n int64 values in[] -> out[] (out has n+1 entries, out[n] = total); 1024 per block
__global__ void __launch_bounds__(256) k_scan_block_sums(const i64 *__restrict__ in, i64 n, i64 *__restrict__ bsum) {
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
__global__ void __launch_bounds__(1024) k_scan_sums(i64 *bsum, i64 nb) {   // one block: in-place exclusive scan
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
__global__ void __launch_bounds__(256) k_scan_apply(const i64 *__restrict__ in, i64 n, const i64 *__restrict__ bsum,
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
static inline unsigned grid_for(i64 n) { return (unsigned)((n + 255) / 256); }

void sbwt_launch_encode(const char *d_bases, long long total_bases, uint4 *d_packed, SbwtWorkHeader *ws,
                        hipStream_t stream) {
    i64 n_groups = (total_bases + SBWT_GROUP_BASES - 1) / SBWT_GROUP_BASES + 2;
    int aligned = ((uintptr_t)d_bases & 15) == 0;
    hipLaunchKernelGGL(k_encode, dim3(grid_for(n_groups)), dim3(256), 0, stream,
                       reinterpret_cast<const unsigned char *>(d_bases), (i64)total_bases, d_packed, n_groups, ws,
                       aligned);
}

void sbwt_launch_search(const SbwtIndexView &ix, const uint4 *d_packed, const long long *d_read_off,
                        const long long *d_out_off, long long *d_out, long long n_reads, SbwtWorkHeader *ws,
                        int streaming, hipStream_t stream, int variant, long long total_groups) {
    if (n_reads <= 0) return;
    if (variant >= 1) {
        i64 want1 = (n_reads + 255) / 256;
        unsigned grid1 = (unsigned)(want1 < 2048 ? want1 : 2048);
        // 32-bit positions need every column index (and n_nodes + 64) below 2^31 and < 2^31 packed groups
        const bool wide = ix.n_nodes >= ((1ll << 31) - 128) || total_groups >= (1ll << 31) - 4 || (ix.debug & 16);
        // no-spill build: 72 VGPRs (7 waves/SIMD max); 4 workgroups per CU measured best (tools/ab_bench.py)
        unsigned cap = (ix.debug >> 8) ? (unsigned)(ix.debug >> 8) : (variant >= 2 ? 1280u : 1024u);
        unsigned g = grid1 < cap ? grid1 : cap;
        if (wide)
            hipLaunchKernelGGL((k_search_cert<true, 4, false>), dim3(g), dim3(256), 0, stream, ix, d_packed, d_read_off,
                               d_out_off, d_out, (i64)n_reads, ws, streaming);
        else if (variant >= 2 && ix.col && streaming)      // path order (the default when the index has one)
            hipLaunchKernelGGL((k_search_cert<false, 4, true>), dim3(g), dim3(256), 0, stream, ix, d_packed, d_read_off,
                               d_out_off, d_out, (i64)n_reads, ws, streaming);
        else
            hipLaunchKernelGGL((k_search_cert<false, 4, false>), dim3(g), dim3(256), 0, stream, ix, d_packed, d_read_off,
                               d_out_off, d_out, (i64)n_reads, ws, streaming);
        return;
    }
    // persistent-style grid: enough 256-thread workgroups to fill 256 CUs x 8 workgroups, never
    // more lanes than reads
    i64 want = (n_reads + 255) / 256;
    unsigned grid = (unsigned)(want < 2048 ? want : 2048);
    if (ix.n_mega > 1)
        hipLaunchKernelGGL(k_search<true>, dim3(grid), dim3(256), 0, stream, ix, d_packed, d_read_off, d_out_off,
                           d_out, (i64)n_reads, ws, streaming == 1 ? 1 : 0);
    else
        hipLaunchKernelGGL(k_search<false>, dim3(grid), dim3(256), 0, stream, ix, d_packed, d_read_off, d_out_off,
                           d_out, (i64)n_reads, ws, streaming == 1 ? 1 : 0);
}

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

// scratch of the sparse-table build: two item lists of n_nodes entries + two counters
long long sbwt_sparse_scratch_bytes(long long n_nodes) { return 2 * (n_nodes + 64) * (long long)sizeof(SpItem) + 256; }

// d_pos: path positions to store with depth-k entries (nullptr = none).  Returns 1 if they were stored,
// 0 if not (no d_pos, p_sparse < k, or some k-mer's interval is wider than one column), < 0 on error.
int sbwt_launch_build_sparse(const SbwtIndexView &ix, int p_dense, int p_sparse, int log2b, uint4 *d_table,
                             void *d_scratch, const unsigned *d_pos, int p_filter, int log2f, uint4 *d_filter,
                             int log2b2, uint4 *d_table2, hipStream_t stream) {
    u64 *counters = reinterpret_cast<u64 *>(d_scratch);                    // [0], [1]: list lengths
    SpItem *listA = reinterpret_cast<SpItem *>(reinterpret_cast<char *>(d_scratch) + 256);
    SpItem *listB = listA + (ix.n_nodes + 64);
    (void)hipMemsetAsync(counters, 0, 256, stream);
    hipLaunchKernelGGL(k_sp_clear, dim3(grid_for((i64)2 << log2b)), dim3(256), 0, stream, d_table, (u64)2 << log2b);
    const u64 n_dense = 1ull << (2 * p_dense);
    hipLaunchKernelGGL(k_sp_collect, dim3(grid_for((i64)n_dense)), dim3(256), 0, stream, ix.ptab, n_dense, listA,
                       counters + 0);
    SpItem *in = listA, *outl = listB;
    int ci = 0;
    for (int d = p_dense; d < p_sparse; d++) {
        (void)hipMemsetAsync(counters + (ci ^ 1), 0, 8, stream);
        const i64 threads = (ix.n_nodes + 64) * 4;
        if (ix.n_mega > 1)
            hipLaunchKernelGGL(k_sp_expand<true>, dim3(grid_for(threads)), dim3(256), 0, stream, ix, in, counters + ci, d,
                               outl, counters + (ci ^ 1));
        else
            hipLaunchKernelGGL(k_sp_expand<false>, dim3(grid_for(threads)), dim3(256), 0, stream, ix, in, counters + ci, d,
                               outl, counters + (ci ^ 1));
        SpItem *t = in; in = outl; outl = t;
        ci ^= 1;
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
    hipLaunchKernelGGL(k_sp_insert, dim3(grid_for(ix.n_nodes + 64)), dim3(256), 0, stream, in, counters + ci, d_table,
                       log2b, with_pos ? d_pos : (const unsigned *)nullptr);
    if (d_table2 && ix.k > p_sparse) {
        // second level: carry every depth-p_sparse prefix on to depth k, remembering where it started
        (void)hipMemsetAsync(d_table2, 0, (size_t)32 << log2b2, stream);
        hipLaunchKernelGGL(k_sp2_seed, dim3(grid_for(ix.n_nodes + 64)), dim3(256), 0, stream, in, counters + ci);
        SpItem2 *in2 = reinterpret_cast<SpItem2 *>(in), *out2 = reinterpret_cast<SpItem2 *>(outl);
        for (int d = p_sparse; d < ix.k; d++) {
            (void)hipMemsetAsync(counters + (ci ^ 1), 0, 8, stream);
            hipLaunchKernelGGL(k_sp2_expand, dim3(grid_for((ix.n_nodes + 64) * 4)), dim3(256), 0, stream, ix, in2,
                               counters + ci, d - p_sparse, out2, counters + (ci ^ 1));
            SpItem2 *t = in2; in2 = out2; out2 = t;
            ci ^= 1;
        }
        int *flag = reinterpret_cast<int *>(counters + 9);
        hipLaunchKernelGGL(k_sp2_insert, dim3(grid_for(ix.n_nodes + 64)), dim3(256), 0, stream, in2, counters + ci, d_table2,
                           log2b2, d_pos, flag);
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
    return np * 4 * 8 + np + np * 8 * 2 + (nb + 2) * 8 + 4096;
}
long long sbwt_path_quads(long long n_nodes) { return n_nodes / 32 + 4; }
void sbwt_launch_path_safe(const SbwtIndexView &ix, uint4 *d_pq, hipStream_t stream) {
    hipLaunchKernelGGL(k_path_safe, dim3(grid_for(ix.n_nodes)), dim3(256), 0, stream, ix, reinterpret_cast<unsigned *>(d_pq));
}

// d_col, d_pos: n_nodes (+4 padding) u32 each; d_pq: sbwt_path_quads() quads.  Synchronises the stream.
int sbwt_launch_build_path(const SbwtIndexView &ix, unsigned *d_col, unsigned *d_pos, uint4 *d_pq, uint4 *d_trans,
                           void *d_scratch, hipStream_t stream) {
    const i64 n = ix.n_nodes;
    const long long np = path_pad(n);
    const i64 nb = (n + 1023) / 1024;
    char *base = reinterpret_cast<char *>(d_scratch);
    int *flag = reinterpret_cast<int *>(base);
    base += 4096;
    unsigned *succ = reinterpret_cast<unsigned *>(base); base += np * 4;
    unsigned *prv = reinterpret_cast<unsigned *>(base); base += np * 4;
    unsigned *buf[2][3];
    for (int a = 0; a < 2; a++)
        for (int f = 0; f < 3; f++) { buf[a][f] = reinterpret_cast<unsigned *>(base); base += np * 4; }
    unsigned char *sch = reinterpret_cast<unsigned char *>(base); base += np;
    unsigned long long *len = reinterpret_cast<unsigned long long *>(base); base += np * 8;
    i64 *pbase = reinterpret_cast<i64 *>(base); base += np * 8;
    i64 *bsum = reinterpret_cast<i64 *>(base);
    const unsigned g = grid_for(n);
    hipLaunchKernelGGL(k_path_fill, dim3(g), dim3(256), 0, stream, prv, n, PATH_NONE);
    hipLaunchKernelGGL(k_path_succ, dim3(g), dim3(256), 0, stream, ix, succ, sch, prv);
    int rounds = 1;
    while (((i64)1 << rounds) < n) rounds++;
    rounds++;
    int cur = 0;
    for (int attempt = 0; attempt < 3; attempt++) {
        cur = 0;
        hipLaunchKernelGGL(k_path_keep, dim3(g), dim3(256), 0, stream, n, succ, prv, buf[0][0], buf[0][1], buf[0][2]);
        for (int r = 0; r < rounds; r++) {
            hipLaunchKernelGGL(k_path_jump, dim3(g), dim3(256), 0, stream, n, buf[cur][0], buf[cur][1], buf[cur][2],
                               buf[cur ^ 1][0], buf[cur ^ 1][1], buf[cur ^ 1][2]);
            cur ^= 1;
        }
        (void)hipMemsetAsync(flag, 0, 4, stream);
        hipLaunchKernelGGL(k_path_cut, dim3(g), dim3(256), 0, stream, n, buf[cur][0], buf[cur][2], prv, succ, flag);
        int h_flag = 0;
        if (hipMemcpyAsync(&h_flag, flag, 4, hipMemcpyDeviceToHost, stream) != hipSuccess) return -1;
        if (hipStreamSynchronize(stream) != hipSuccess) return -1;
        if (!h_flag) break;
        if (attempt == 2) return -2;                    // cannot happen: one cut per cycle opens every cycle
    }
    (void)hipMemsetAsync(len, 0, (size_t)np * 8, stream);
    hipLaunchKernelGGL(k_path_len, dim3(g), dim3(256), 0, stream, n, buf[cur][0], buf[cur][1], len);
    hipLaunchKernelGGL(k_scan_block_sums, dim3((unsigned)nb), dim3(256), 0, stream, reinterpret_cast<const i64 *>(len), n, bsum);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(1024), 0, stream, bsum, nb);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(256), 0, stream, reinterpret_cast<const i64 *>(len), n, bsum, pbase);
    (void)hipMemsetAsync(d_pq, 0, (size_t)sbwt_path_quads(n) * 16, stream);
    hipLaunchKernelGGL(k_path_place, dim3(g), dim3(256), 0, stream, n, buf[cur][0], buf[cur][1], pbase, succ, sch, d_pos,
                       d_col, reinterpret_cast<unsigned *>(d_pq));
    hipLaunchKernelGGL(k_path_trans, dim3(g), dim3(256), 0, stream, ix, d_pos, d_trans);
    if (hipStreamSynchronize(stream) != hipSuccess) return -1;
    return 0;
}
