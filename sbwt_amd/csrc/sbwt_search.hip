// sbwt_search.hip -- hand-written HIP kernels (gfx950 / CDNA4) for the plain-matrix SBWT
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
//   sbwt_launch_encode / sbwt_launch_search
// Elsewhere: sbwt_api_kernels.hip (rank, precalc, update_interval, forward), sbwt_derived.hip (suffix-group marks,
// sparse prefix table, probe filter, path order), sbwt_format.hip (print_vector), sbwt_kernels_common.h (helpers).
//
// Reference semantics restated here (paths relative to the reference repo):
//   SBWT::streaming_search      include/sbwt/SBWT.hh:544-581
//   SBWT::search                include/sbwt/SBWT.hh:389-415
//   SBWT::update_sbwt_interval  include/sbwt/SBWT.hh:422-437
//   SBWT::forward               include/sbwt/SBWT.hh:368-381
//   SBWT::do_kmer_prefix_precalc include/sbwt/SBWT.hh:616-645
//   SubsetMatrixRank::rank      include/sbwt/SubsetMatrixRank.hh:31-37
#include "sbwt_kernels_common.h"

// ---------------------------------------------------------------------------------------------
// k_encode: ASCII bases -> packed groups of 32 bases (coalesced streaming pre-pass)
//   group = { codes[31:0], codes[63:32], validU, validRaw }
//   codes   2 bits per base, base t of the group at bits 2t..2t+1 (toupper'd, globals.hh:38-47)
//   validU  bit t set iff toupper(base) is ACGT   (streaming step validates this: SBWT.hh:565-568)
//   validRaw bit t set iff base itself is ACGT    (search validates the raw char: SBWT.hh:398-399,427-428)
// Also resets the workspace header for the search launch that follows on the same stream.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 encode_group(const unsigned char *__restrict__ bases, i64 total, i64 g, int aligned16) {
    i64 base = g * SBWT_GROUP_BASES;
    u64 codes = 0;
    unsigned vu = 0, vr = 0;
    if (aligned16 && base + SBWT_GROUP_BASES <= total) {
        const uint4 *src = reinterpret_cast<const uint4 *>(bases + base);
        uint4 x0 = ld_stream(src), x1 = ld_stream(src + 1);
        unsigned w[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        // four bases per 32-bit operation: upper-case, 2-bit codes, and "byte equals A/C/G/T" by exact per-byte
        // zero detection; the per-byte bits are gathered with one multiply each
#pragma unroll
        for (int d = 0; d < 8; d++) {
            const unsigned up = w[d] & 0xDFDFDFDFu;
            const unsigned t = ((up >> 1) & 0x03030303u) ^ ((up >> 2) & 0x01010101u);        // dna_code of every byte
            const unsigned c8 = (t * 0x01041040u) >> 24;                                     // 4 x 2 bits -> one byte
            auto nz = [](unsigned v) { return ((v & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | v; };       // bit 7 of a byte set <=> byte != 0
            const unsigned none = nz(up ^ 0x41414141u) & nz(up ^ 0x43434343u) & nz(up ^ 0x47474747u) & nz(up ^ 0x54545454u);
            const unsigned isl = (~none >> 7) & 0x01010101u;                                 // 1 per byte that is ACGT after toupper
            const unsigned u4 = (isl * 0x10204080u) >> 28;
            const unsigned low4 = (((w[d] >> 5) & 0x01010101u) * 0x10204080u) >> 28;          // bit 5 set: lower case
            codes |= (u64)c8 << (8 * d);
            vu |= u4 << (4 * d);
            vr |= (u4 & ~low4) << (4 * d);
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
    return make_uint4((unsigned)codes, (unsigned)(codes >> 32), vu, vr);
}

__global__ void __launch_bounds__(256) k_encode(const unsigned char *__restrict__ bases, i64 total,
                                                uint4 *__restrict__ packed, i64 n_groups,
                                                SbwtWorkHeader *ws, int aligned16) {
    i64 g = (i64)blockIdx.x * 256 + threadIdx.x;
    if (g == 0) { ws->ticket = 0; ws->status = 0; ws->n_stream = 0; ws->n_search = 0; ws->n_lf = 0; ws->n_tab_hit = 0; ws->n_ext = 0; ws->n_bridge = 0; }
    if (g >= n_groups) return;
    packed[g] = encode_group(bases, total, g, aligned16);
}

// The encode pass of the fused route (sbwt_search_fused.hip), behind k_search_fused: every group when that kernel declined
// the batch (reads of different lengths), else only the groups of the reads it handed on (usually none).  Leaves the
// workspace header alone: the fused kernel's counters are in it.
__global__ void __launch_bounds__(256) k_encode_chained(const unsigned char *__restrict__ bases, i64 total,
                                                        uint4 *__restrict__ packed, i64 n_groups, const SbwtWorkHeader *ws,
                                                        const unsigned *__restrict__ defer_list,
                                                        const i64 *__restrict__ read_off, int k, int aligned16) {
    const i64 t0 = (i64)blockIdx.x * 256 + threadIdx.x, stride = (i64)gridDim.x * 256;
    const int fmode = sbwt_fused_mode(ws, k);
    if (fmode == 0) {
        for (i64 g = t0; g < n_groups; g += stride) packed[g] = encode_group(bases, total, g, aligned16);
        return;
    }
    if (fmode >= 2) {
        // reads of any lengths: a wave per read that was handed on, its lanes over the read's groups (it may be a genome)
        const i64 nd = (i64)ws->n_deferred, nw = stride >> 6;
        const int lane = threadIdx.x & 63;
        for (i64 d = t0 >> 6; d < nd; d += nw) {
            const i64 r = (i64)defer_list[d], P0 = read_off[r], P1 = read_off[r + 1];
            if (P1 <= P0) continue;
            for (i64 g = (P0 >> 5) + lane; g <= ((P1 - 1) >> 5) && g < n_groups; g += 64)
                packed[g] = encode_group(bases, total, g, aligned16);
        }
        return;
    }
    const i64 nd = (i64)ws->n_deferred, len = ws->u_len, r0 = ws->u_read0;
    const i64 GP = (len + 31) / 32 + 1;                      // groups a read of len bases can touch
    for (i64 t = t0; t < nd * GP; t += stride) {
        const i64 P0 = r0 + (i64)defer_list[t / GP] * len;
        const i64 g = (P0 >> 5) + (t % GP);
        if (g <= ((P0 + len - 1) >> 5) && g < n_groups) packed[g] = encode_group(bases, total, g, aligned16);
    }
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
            if (ix.out32) reinterpret_cast<int *>(out)[obase + i] = (int)res;
            else out[obase + i] = res;
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
//     step is one 16-byte quad of the transition table (the successor of position t by that char: column,
//     path position and the next 8 steps of its path, so that short runs end in the same iteration).  A third load per iteration prefetches the next packed group of the
//     read, so that 32-base windows rarely wait for a reload.
// (5) SEG = true (with PATH): results are not staged as values.  What a read has produced since its last flush is kept as
//     a short list of SEGMENTS in LDS -- { source, first k-mer }: a run out of col[] (source = path position of its first
//     result), a stretch of -1, or one literal column -- and when the read ends (or the list is full) the whole wave
//     writes the read's results together: every lane finds the segment of its two results, fetches col[] and stores 16
//     bytes.  One cooperative pass per read instead of one descriptor per run and iteration (the descriptor writer is
//     53 % of this kernel's vector instructions).
#define SBWT_NSEG 12
template <bool WIDE, int WPS, bool PATH>
__global__ void __launch_bounds__(256, WPS) k_search_cert(SbwtIndexView ix, const uint4 *__restrict__ packed,
                                                        const i64 *__restrict__ read_off,
                                                        const i64 *__restrict__ out_off, i64 *__restrict__ out,
                                                        i64 n_reads, SbwtWorkHeader *ws, int streaming,
                                                        const unsigned *__restrict__ perm,
                                                        const unsigned *__restrict__ defer_list, SbwtPieceTab pt) {
    // perm != nullptr: ticket t is read perm[t] (the reads sorted by where they start in the path order, sbwt_sort.hip:
    // the lanes of a wave then walk the same paths and share their lines of col / pq / trans)
    // defer_list != nullptr: this launch runs behind k_search_fused (sbwt_search_fused.hip).  If that kernel took the batch,
    // what is left are the reads it handed on: n_deferred tickets, ticket t is read defer_list[t].
    unsigned long long *const ticket = defer_list ? &ws->ticket2 : &ws->ticket;
    if (defer_list && sbwt_fused_mode(ws, ix.k)) {
        n_reads = (i64)ws->n_deferred;
        perm = defer_list;
        if (n_reads == 0) return;                      // the usual case: nothing was handed on
    }
    // long reads: tickets n_reads .. n_tickets-1 are their pieces (SbwtPieceTab), the reads themselves are skipped.
    // (a lane holds piece z as rd = -(z + 1): with perm, read numbers and ticket numbers are different things)
    const bool cut = pt.pairs != nullptr;
    i64 n_tickets = n_reads;
    if (cut) {
        const i64 np = (i64)ws->n_pieces;
        n_tickets += np < pt.cap ? np : pt.cap;
    }
    typedef typename SearchTypes<WIDE>::pos_t pos_t;
    typedef typename SearchTypes<WIDE>::stage_t stage_t;
    constexpr int DEPTH = SearchTypes<WIDE>::DEPTH;
    constexpr bool SEG = PATH;                      // the path-order kernel writes through per-read segment lists
    static_assert(!PATH || !WIDE, "path order: 32-bit indexes only");
    // a third cached packed group, fetched in pairs (see the loads below).  SEG only: the descriptor-writer kernel would
    // pay for the registers with a wave per SIMD (config 3: 123 -> 150 ms)
    constexpr bool G3 = SEG;
    __shared__ stage_t stage[SEG ? 1 : DEPTH][SEG ? 1 : 256];
    __shared__ uint2 segs[SEG ? SBWT_NSEG : 1][SEG ? 256 : 1];                      // SEG: { source, first k-mer } per lane
    const int tid = threadIdx.x, lane = tid & 63;
    int nseg = 0, i0 = 0, last_start = 0;           // SEG: segments listed; first result of the read not written yet
    unsigned last_src = 0, emit_pos = 0;
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
    // Blind mode (as in k_search_fused): bits 0-1 = own searches of consecutive k-mers that answered nothing but their own
    // k-mer, each behind probes that found every window (perhaps) present -- a substitution that is another strain's base;
    // bit 2 = such probes since the last own search.  From two on the planner stops probing until a k-mer is found
    // (the reference's own loop, SBWT.hh:557-559: 5 iterations per k-mer -> 1).
    unsigned mz = 0;
    int mode = M_IDLE;              // M_DEAD once the ticket counter has run past the last read
    i64 obase = 0;                  // first result slot of the current read
    int pgrp = 0, poff = 0;         // the read starts at base poff of packed group pgrp
    int m = 0, i = 0, j = 0, b = -1, wstart = 0, cnt = 0;
    pos_t l = 0, r = 0;             // walk interval; M_STREAM: l = previous answer; M_BACK: r = block
    i64 rd = 0;                     // M_FETCH: the read whose offsets are being fetched
    bool rdok = true;               // false: rd is still a ticket, the fetch step reads perm[rd] first
    int tag = -2;                   // g0 = packed group `tag`; g1 = group tag+1 if g1ok; PATH: g2 = group tag+2 if g2ok
    bool g1ok = false, g2ok = false;
    // PATH: fixed-length reads (checked by k_check_uniform just before this launch)
    const bool uni = PATH && ws->u_bad == 0 && ws->u_len > 0;
    const i64 u_read0 = ws->u_read0, u_len = ws->u_len, u_out0 = ws->u_out0, u_stride = ws->u_stride;
    unsigned c_ext = 0;             // PATH: k-mers answered along paths (per lane)
    unsigned c_brg = 0;             // PATH: substitutions bridged (per lane)
    uint4 g0 = make_uint4(0, 0, 0, 0), g1 = make_uint4(0, 0, 0, 0), g2 = make_uint4(0, 0, 0, 0);
    u64 pool_next = 0, pool_end = 0;                              // wave-uniform pool of read tickets
    unsigned c_stream = 0, c_search = 0, c_lf = 0, c_tab = 0;     // wave-uniform (scalar) work counters

    for (;;) {
        // ---- hand out reads to idle lanes from the wave's ticket pool ----
        const u64 need = __ballot(mode == M_IDLE);
        if (need) {
            if (pool_next == pool_end) {
                u64 t = 0;
                if (lane == 0) t = atomicAdd(ticket, 64ull);
                pool_next = uniform64(t);              // lane 0's value, kept in scalar registers
                pool_end = pool_next + 64;
            }
            const unsigned avail = (unsigned)(pool_end - pool_next);
            const unsigned n = (unsigned)__popcll(need);
            const unsigned rank = (unsigned)__popcll(need & low_mask(lane));
            if (mode == M_IDLE && rank < avail) {
                rd = (i64)(pool_next + rank);
                mode = (rd < n_tickets) ? M_FETCH : M_DEAD;
                rdok = (perm == nullptr);
                if (rd >= n_reads) { rd = -(rd - n_reads) - 1; rdok = true; }
                if (uni && mode == M_FETCH && rdok && rd >= 0) {
                    // reads of one length: offsets by arithmetic, the walk for the first k-mer starts right away
                    const i64 P0 = u_read0 + rd * u_len;
                    obase = u_out0 + rd * u_stride;
                    pgrp = (int)(P0 >> 5);
                    poff = (int)(P0 & 31);
                    m = (int)u_len - k + 1;
                    i = 0;
                    if (SEG) { nseg = 0; i0 = 0; }
                    b = -1;
                    blo = -1;
                    mz = 0;
                    wstart = 0;
                    j = 0;
                    wk = (ps > 0) ? 1 : 0;
                    if (m <= 0 || (cut && piece_read_is_cut(m, pt.piece))) mode = M_IDLE;
                    else if (p > 0) mode = M_INIT;
                    else { mode = M_STEP; l = 0; r = last_node; }
                }
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
        bool ext_absent = false;                       // SEG: the k-mer after this iteration's run is absent (only-successor step)
        int tnext = M_EXT;                             // PATH: where a transition's quoted steps already end the run
        pos_t tpos = -1;
        int seg_n = 0;                                 // PATH: k-mers i .. i+seg_n-1 are col[seg_src ..]
        unsigned seg_src = 0;
        if (mode == M_FETCH) {
            kind = K_FETCH;                            // {read_off[rd], read_off[rd+1]}, {out_off[rd], ..}
            if (rdok && rd < 0) {                      // a piece of a long read
                a1 = pt.pairs + (-rd - 1);
                a2 = pt.outs + (-rd - 1);
            } else if (rdok) {
                a1 = reinterpret_cast<const uint4 *>(read_off + rd);
                a2 = reinterpret_cast<const uint4 *>(out_off + rd);
            } else {                                   // the aligned 16 bytes holding perm[ticket]
                a1 = reinterpret_cast<const uint4 *>(perm + (rd & ~(i64)3));
                a2 = a1;
            }
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
            if (grp == tag + 1 && g1ok && (mode != M_INIT || s + wl <= 32 || (G3 && g2ok))) {
                g0 = g1;                               // crossed into the group that is already here
                g1ok = false;
                if (G3) { g1 = g2; g1ok = g2ok; g2ok = false; }
                tag = grp;
            } else if (G3 && grp == tag + 2 && g1ok && g2ok && (mode != M_INIT || s + wl <= 32)) {
                g0 = g2;                               // ... or two groups on (after a certified burst)
                g1ok = false;
                g2ok = false;
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
                    // (a base that is not ACGT never gets here: M_EXT treats it as "no successor")
                    // the entry of (position r, char c); j counts the slots probed
                    a1 = ix.trans + 2 * (size_t)sbwt_trans_slot((unsigned)r, (unsigned)c, ix.n_tslots, (unsigned)j);
                    a2 = a1 + 1;
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
                            const size_t bkt = sbwt_sp_bucket(key, ix.n_sb, (unsigned)j);
                            a1 = ix.stab + 2 * bkt;
                            a2 = a1 + 1;
                        } else if (wk == 5) {          // second level: (prefix interval, rest of the k-mer) -> one entry
                            hk = w & m2;
                            const size_t bkt = sbwt_sp2_entry((unsigned)l, hk, ix.n_sb2, (unsigned)j);
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
        {   // lane-iterations by kind: pad[8..]: fetch, reload, init, step, trans, pos/bridge, ext, idle/dead, waves-iterations
            const int cls = (kind == K_FETCH) ? 0 : (kind == K_RELOAD) ? 1 : (kind == K_NONE) ? 7 :
                            (mode == M_INIT) ? 2 : (mode == M_STEP) ? 3 : (mode == M_TRANS) ? 4 : (mode == M_POS || mode == M_BRIDGE) ? 5 : (mode == M_EXT) ? 6 : 7;
            for (int q = 0; q < 8; q++) {
                const unsigned long long cq = __popcll(__ballot(cls == q));
                if (lane == 0 && cq) atomicAdd(&ws->pad[6 + q], cq);
            }
            if (lane == 0) atomicAdd(&ws->pad[14], 1ull);
        }
#endif
        c_search = uniform32(c_search + (unsigned)__popcll(__ballot(kind == K_MODE && (mode == M_INIT || (p == 0 && mode == M_STEP && j == 0)))));
        c_lf = uniform32(c_lf + (unsigned)__popcll(__ballot(have && mode == M_STEP)));

        // ---- the one memory round trip of this iteration ----
        // PATH: the read's next packed group -- G3: the next TWO -- ride along when the next one is not here yet (two at a time: a group
        // fetched in a later iteration is another request to the fabric even when it lies in the same line -- the L2 turns
        // over faster than a lane comes back); a reload brings three
        const bool pf = PATH && !g1ok && kind == K_MODE;
        const bool rl3 = G3 && kind == K_RELOAD;
        const uint4 *a3 = pf ? (packed + (tag + 1)) : rl3 ? (a1 + 2) : a1;
        const uint4 v1 = *a1;
        const uint4 v2 = *a2;
        if (PATH) {
            const uint4 v3 = *a3;
            if (pf) { g1 = v3; g1ok = true; }
            if (G3) {
                const uint4 v4 = *(pf ? (a3 + 1) : a1);
                if (pf) { g2 = v4; g2ok = true; }
                if (rl3) { g2 = v3; g2ok = true; }
            }
        }

        // ---- consume ----
        bool tabhit = false, do_plan = false, force = false;
        bool imprecise = false;                        // this iteration's failure is a table-level miss
        int burst_to = -1;                             // M_BRIDGE: k-mers i .. burst_to are certified absent
        bool bridged = false;                          // M_BRIDGE: ... and the read goes on along the path
        if (kind == K_FETCH) {
            bool have_off = rdok;
            i64 P0 = (i64)quad_bits(v1);
            i64 P1 = (i64)((u64)v1.z | ((u64)v1.w << 32));
            obase = (i64)quad_bits(v2);
            if (!rdok) {
                const unsigned sl = (unsigned)rd & 3u;
                rd = (i64)(sl == 0 ? v1.x : sl == 1 ? v1.y : sl == 2 ? v1.z : v1.w);
                rdok = true;
                if (uni) {                             // fixed-length reads: offsets by arithmetic
                    P0 = u_read0 + rd * u_len;
                    P1 = P0 + u_len;
                    obase = u_out0 + rd * u_stride;
                    have_off = true;
                }                                      // else: the next iteration fetches the offsets of read rd
            }
            if (have_off) {
                pgrp = (int)(P0 >> 5);
                poff = (int)(P0 & 31);
                m = (int)(P1 - P0) - k + 1;
                i = 0;
                if (SEG) { nseg = 0; i0 = 0; }
                b = -1;
                blo = -1;
                mz = 0;
                if (m > 0 && !(cut && rd >= 0 && piece_read_is_cut(m, pt.piece))) { do_plan = true; force = true; }
                else mode = M_IDLE;
            }
        } else if (kind == K_RELOAD) {
            g0 = v1;
            g1 = v2;
            g1ok = true;                               // (PATH: g2 and g2ok were set with the loads)
            tag = grp;
        } else if (PATH && have && mode == M_POS) {
            const unsigned sel = (unsigned)l & 3u;
            r = (pos_t)(sel == 0 ? v1.x : sel == 1 ? v1.y : sel == 2 ? v1.z : v1.w);
            mode = M_EXT;
        } else if (trn && have) {
            // v1 = { r + 1, c | flags, successor column (SBWT.hh:562-575), its path position }, v2 = its path's next 32 steps
            if (v1.x == 0u) {
                ev = EV_EMIT1;                         // a free slot: (r, c) has no entry -- a path's last column without a
                b = blo = i + k - 1;                   // successor by c: -1
            } else if (v1.x != (unsigned)r + 1u || (v1.y & 3u) != (unsigned)c) {
                j++;                                   // another entry's slot: the next one (linear probing)
                if (j > 4096) { ws->status = SBWT_ERR_NOT_SINGLETON; ev = EV_EMIT1; b = blo = i + k - 1; }   // damaged image
            } else if (v1.y & SBWT_TRANS_NEG) {
                // no successor by this char at a step that has others; the entry says whether the step is safe for it
                if (ix.has_safe && (v1.y & SBWT_TRANS_NEG_SAFE)) {
                    mode = M_BRIDGE;
                } else {
                    ev = EV_EMIT1;
                    b = blo = i + k - 1;
                }
            } else {
                ev = EV_EMIT1;
                res = (pos_t)v1.z;
                r = (pos_t)v1.w;
                emit_pos = v1.w;
                rknown = true;
                // the read's next bases against the steps quoted in the entry, as far as the cached groups reach
                const int P1 = poff + i + k, s1 = P1 & 31;
                int have1 = (s1 == 0) ? (g1ok ? 32 : 0) : (32 - s1) + (g1ok ? 32 : 0);   // bases from P1 on that are cached
                if (have1 > 31) have1 = 31;
                u64 rq, rv;
                const u64 va = (streaming == 2) ? (((u64)g1.w << 32) | (u64)g0.w) : (((u64)g1.z << 32) | (u64)g0.z);
                if (s1 != 0) {
                    rq = (quad_bits(g0) >> (2 * s1)) | (quad_bits(g1) << (64 - 2 * s1));
                    rv = va >> s1;
                } else {
                    rq = quad_bits(g1);
                    rv = va >> 32;
                }
                const u64 x = (rq ^ quad_bits(v2)) & low_mask(62);
                const u64 mm = (x | (x >> 1)) & 0x5555555555555555ull;
                const int nm = mm ? ((__ffsll((i64)mm) - 1) >> 1) : 31;
                const int ng = __ffs((int)((v2.z & ~v2.w) | 0x80000000u)) - 1;  // the quoted path ends: A & ~B
                const int nr = __ffs((int)(~(unsigned)rv | 0x80000000u)) - 1;   // a base of the read that is not ACGT
                const int nv = ng < nr ? ng : nr;
                int n2 = nm < nv ? nm : nv;
                bool stop2 = n2 < 31;
                if (n2 >= have1) { n2 = have1; stop2 = false; }
                if (n2 >= m - 1 - i) { n2 = m - 1 - i; stop2 = false; }
                if (n2 < 0) n2 = 0;
                seg_n = n2;
                seg_src = (unsigned)r + 1u;
                r += (pos_t)n2;
                c_ext += (unsigned)n2;
                if (stop2) {
                    int kind2 = PS_ABSENT;             // a base that is not ACGT: -1 (SBWT.hh:568)
                    if (!(nr <= nm && nr <= ng))
                        kind2 = path_stop_kind(nm < ng, (v2.z >> n2) & 1u, (v2.w >> n2) & 1u, ix.has_safe != 0);
                    if (kind2 == PS_ABSENT) ext_absent = true;
                    else tnext = (kind2 == PS_TRANS) ? M_TRANS : M_BRIDGE;
                }
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
            // (a base that is not ACGT among them: packed as 'A' it may seem to agree -- no bridge, the certificates sort it out)
            const u64 rvb = ((streaming == 2) ? (((u64)g1.w << 32) | (u64)g0.w) : (((u64)g1.z << 32) | (u64)g0.z)) >> s;
            if (nm >= need && (~rvb & low_mask(need)) == 0) {
                ev = EV_FAIL;
                burst_to = i + need;
                c_brg++;
                // the read is back on the path: its k-mer i+k-1 (the last one with the substituted base) stands where the path's
                // own k-mer would, at position r + k, as far as the next streaming step is concerned -- that step depends on the
                // last k-1 bases only, and those are the path's.  On with M_EXT from there, no walk.
                // The reference finds that next k-mer with SBWT::search (the one before it is -1, SBWT.hh:557-559), which
                // takes upper-case ACGT only (SBWT.hh:398-399,427-428): all k of its bases must be that, else the walk decides.
                const u64 vraw = (((u64)g1.w << 32) | (u64)g0.w) >> s;
                bridged = need == k - 1 && (~vraw & low_mask(k)) == 0;
            } else {
                // no bridge (a second difference within k-1 bases).  The streaming step itself needs no gather either: a
                // bridgeable step has no successor by the read's char
                ev = EV_EMIT1;
                b = blo = i + k - 1;
            }
        } else if (ext && have) {
            // k-mer i-1 sits at path position r.  Read bases i+k-1.. against path chars r..: while they agree
            // (and the read's bases are valid and the path goes on), k-mer i+x sits at r+1+x.
            const int P = poff + i + k - 1, s = P & 31, sp = (int)((unsigned)r & 31u);
            u64 rw = quad_bits(g0) >> (2 * s), pw = quad_bits(v1) >> (2 * sp);
            if (s) rw |= quad_bits(g1) << (64 - 2 * s);
            if (sp) pw |= quad_bits(v2) << (64 - 2 * sp);
            const u64 rv = ((streaming == 2) ? (((u64)g1.w << 32) | (u64)g0.w) : (((u64)g1.z << 32) | (u64)g0.z)) >> s;
            // the path groups' two state words (k_path_reencode): go = ~A | B, safe = A & B, only successor = ~A & B
            const u64 fA = (((u64)v2.z << 32) | (u64)v1.z) >> sp, fB = (((u64)v2.w << 32) | (u64)v1.w) >> sp;
            const u64 pgo = ~fA | fB;
            const u64 x = rw ^ pw;
            const u64 mm = (x | (x >> 1)) & 0x5555555555555555ull;
            int nm = mm ? ((__ffsll((i64)mm) - 1) >> 1) : 32;
            int ng = __ffsll((i64)(~pgo | (1ull << 32))) - 1;                 // the path ends
            int nr = __ffsll((i64)(~rv | (1ull << 32))) - 1;                  // a base that is not ACGT
            int nv = ng < nr ? ng : nr;
            int n = nm < nv ? nm : nv;
            bool stopped = n < 32;                     // a mismatch, an invalid base or the end of the path
            if (!stopped && g1ok) {
                // the cached group pair and the two path quads reach further than 32 steps: 32 - s more bases of the read
                // (all 32 with the third cached group), 32 - sp more chars of the path
                const bool r3 = g2ok && s != 0;
                const int w2 = (r3 || s <= sp) ? 32 - sp : 32 - s;
                u64 rw2 = quad_bits(g1) >> (2 * s);
                u64 rv2 = rv >> 32;
                if (r3) {
                    rw2 |= quad_bits(g2) << (64 - 2 * s);
                    const u64 vb = (streaming == 2) ? (((u64)g2.w << 32) | (u64)g1.w) : (((u64)g2.z << 32) | (u64)g1.z);
                    rv2 = (vb >> s) & 0xFFFFFFFFull;
                }
                const u64 x2 = rw2 ^ (quad_bits(v2) >> (2 * sp));
                const u64 mm2 = (x2 | (x2 >> 1)) & 0x5555555555555555ull;
                const int nm2 = mm2 ? ((__ffsll((i64)mm2) - 1) >> 1) : 32;
                const int ng2 = __ffsll((i64)(~(pgo >> 32) | (1ull << 32))) - 1;
                const int nr2 = __ffsll((i64)(~rv2 | (1ull << 32))) - 1;
                const int nv2 = ng2 < nr2 ? ng2 : nr2;
                int n2 = nm2 < nv2 ? nm2 : nv2;
                if (n2 >= w2) n2 = w2;                 // the end of what is cached is not a stop
                else stopped = true;
                n = 32 + n2;
                nm = 32 + nm2;
                ng = 32 + ng2;
                nr = 32 + nr2;
            }
            if (n >= m - i) { n = m - i; stopped = false; }
            seg_n = n;
            seg_src = (unsigned)r + 1u;
            c_ext += (unsigned)n;
#ifdef SBWT_STATS
            if (!stopped && i + n != m) atomicAdd(&ws->pad[15], 1ull);      // limited by the window
#endif
            if (i + n == m) {
                mode = M_IDLE;
            } else if (stopped) {
                int kind2 = PS_ABSENT;                 // a base that is not ACGT: -1 (SBWT.hh:568)
                if (!(nr <= nm && nr <= ng)) {
                    kind2 = path_stop_kind(nm < ng, (unsigned)(fA >> n) & 1u, (unsigned)(fB >> n) & 1u, ix.has_safe != 0);
                    if (kind2 == PS_BRIDGE && n < 32) {
                        // a bridge needs the next k-1 bases to agree with the path; a second difference already in this
                        // window: skip the attempt
                        const int after = 31 - n, want = (k - 1 < m - 1 - (i + n)) ? (k - 1) : (m - 1 - (i + n));
                        const int chk = after < want ? after : want;
                        if (chk > 0 && ((mm >> (2 * (n + 1))) & low_mask(2 * chk)) != 0) kind2 = PS_ABSENT;
                    }
                }
                if (kind2 == PS_ABSENT) ext_absent = true;
                else { mode = (kind2 == PS_TRANS) ? M_TRANS : M_BRIDGE; j = 0; }
            }
            r += (pos_t)n;
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
                        mz |= 4u;
                        if (wk == 3) {
                            l = 0;                     // range probe: "perhaps present" only moves the guess
                        } else {
                            // perhaps present (in a pan-genome usually truly: the window is another strain's variant):
                            // no certificate from this window -- the reference's own search of k-mer i decides
                            again = true;
                            do_plan = true;
                            force = true;
                        }
                    } else {
                        l = -1;                        // read[wstart .. wstart+L0-1] is not in the index
                    }
                  } else if (wk == 5) {
                    wl = k;                            // a hit completes the k-mer; a miss says read[wstart .. wstart+k-1] is absent
                    const bool s0 = (v1.w & SBWT_SP2_USED) && quad_bits(v1) == hk && (v1.z & ~SBWT_SP2_OVERFLOW) == (unsigned)l;
                    const bool s1 = (v2.w & SBWT_SP2_USED) && quad_bits(v2) == hk && v2.z == (unsigned)l;
                    if (s0 | s1) {
                        // the payload: the k-mer's path position on an image with a path order (column = col[position]), else
                        // its column.  (The blocks-only kernel on a path-order image -- a cross-check route -- fetches it.)
                        const unsigned pv = (s0 ? v1.w : v2.w) & ~SBWT_SP2_USED;
                        if (PATH) { tpos = (pos_t)pv; l = (pos_t)pv; }
                        else l = ix.col ? (pos_t)ix.col[pv] : (pos_t)pv;
                        r = l;
                    } else if (v1.z & SBWT_SP2_OVERFLOW) {
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
                            const unsigned wv = m0 ? v1.w : v2.w;          // (SBWT_SP_UNIQ: one column, the rest is its position)
                            r = (wv & SBWT_SP_UNIQ) ? l : l + (pos_t)wv;
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
                if (PATH && tpos >= 0) { r = tpos; rknown = true; emit_pos = (unsigned)tpos; }
                ev = EV_EMIT1;
                b = -1;
                mz = 0;
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
                // a walk that started AT the known-bad position says nothing about where the next one is; a window around
                // it leaves it where it is
                if (wstart == b) b = -1;
                else if (!(b > wstart && b <= tfail)) b = tfail;
                blo = b;
            }
            if (burst_hi > i || wk == 2 || wk == 3) mz = 0;                    // a window that IS absent: probes work here
            else if (wstart == i && (mz & 3u) < 2u) mz = (mz & 4u) ? (mz & 3u) + 1u : 0u;   // k-mer i's own search answered only k-mer i
            mz &= ~4u;
            if (burst_hi == i) { ev = EV_EMIT1; burst_hi = -1; }   // a single -1 goes through the stage
        }
        if (ev == EV_PRES) {                           // no bad base in [wstart, wstart+p-1]: shrink the range
            const int lo = blo > i ? blo : i;
            if (wstart > lo) b = wstart - 1;
            else blo = wstart + pw;
            if (blo > b) b = -1;
            do_plan = true;
        }
        if (ev == EV_EMIT1 && !SEG) {
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
        if (SEG) {
            // ---- append this iteration's results to the lane's segment list (at most two segments; contiguous ones merge) ----
            auto append = [&](int at, unsigned src) {
                const bool merge = nseg > 0 && ((src == 0xFFFFFFFFu && last_src == 0xFFFFFFFFu) ||
                                                (!(src >> 31) && !(last_src >> 31) && last_src + (unsigned)(at - last_start) == src));
                if (!merge) {
                    segs[nseg][tid] = make_uint2(src, (unsigned)at);
                    nseg++;
                    last_src = src;
                    last_start = at;
                }
            };
            if (ev == EV_EMIT1) {
                append(i, res == -1 ? 0xFFFFFFFFu : (rknown ? emit_pos : (0x80000000u | (unsigned)res)));
                i++;
            }
            {
                const int nleft = (burst_hi >= 0) ? (burst_hi - i + 1) : seg_n;
                if (nleft > 0) {
                    append(i, (burst_hi >= 0) ? 0xFFFFFFFFu : seg_src);
                    i += nleft;
                }
                if (ext_absent) {                      // the k-mer that left the path: -1, then the certificates as after
                    append(i, 0xFFFFFFFFu);            // any failed streaming step
                    b = blo = i + k - 1;
                    i++;
                    if (i == m) mode = M_IDLE;
                    else do_plan = true;
                }
            }
            // ---- flush: the read is done, or the list could overflow in the next iteration.  Up to four reads per trip: the
            //      col[] loads of all four are in flight before the first store (a trip would otherwise wait for one
            //      memory round trip per read) ----
            u64 fm = __ballot(nseg > 0 && (i == m || nseg > SBWT_NSEG - 2));
            while (fm) {
                constexpr int FP = 4;
                int fL[FP], fe[FP], fj[FP], w0[FP], w1[FP];
                i64 fob[FP];
                unsigned x0[FP], x1[FP];
                bool long_read = false;
#pragma unroll
                for (int u = 0; u < FP; u++) {
                    fL[u] = -1; fe[u] = 0; fj[u] = 0; fob[u] = 0; x0[u] = x1[u] = 0xFFFFFFFFu; w0[u] = w1[u] = 0;
                    if (fm == 0) continue;                       // wave-uniform: unused slots cost nothing
                    const int L = __ffsll((i64)fm) - 1;
                    fm &= fm - 1;
                    // (L is wave-uniform: the owner's registers are read with v_readlane)
                    const int ns = __builtin_amdgcn_readlane(nseg, L), a = __builtin_amdgcn_readlane(i0, L), e = __builtin_amdgcn_readlane(i, L);
                    fL[u] = L;
                    fe[u] = e;
                    fob[u] = (i64)((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u64)obase, L) |
                                   ((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)((u64)obase >> 32), L) << 32));
                    long_read = long_read || (e - a > 128);
                    const int tl = (tid & ~63) + L;
                    const int j0 = a + 2 * lane, j1 = j0 + 1;
                    fj[u] = j0;
                    // the segment of result j0: the last one that starts at or before it (lane t holds the start of segment t:
                    // the starts ascend, so that is a count); result j1 is in the same segment or the next
                    const int my_at = (int)segs[lane < SBWT_NSEG ? lane : SBWT_NSEG - 1][tl].y;
                    int idx = 0;
                    for (int t = 1; t < ns; t++) idx += (__builtin_amdgcn_readlane(my_at, t) <= j0) ? 1 : 0;
                    const uint2 c0 = segs[idx][tl];
                    const uint2 nx = segs[idx + 1 < SBWT_NSEG ? idx + 1 : SBWT_NSEG - 1][tl];
                    const uint2 c1 = (idx + 1 < ns && (int)nx.y <= j1) ? nx : c0;
                    x0[u] = c0.x;
                    x1[u] = c1.x;
                    const unsigned p0 = (j0 < e && !(c0.x >> 31)) ? c0.x + (unsigned)(j0 - (int)c0.y) : 0u;
                    const unsigned p1 = (j1 < e && !(c1.x >> 31)) ? c1.x + (unsigned)(j1 - (int)c1.y) : 0u;
                    w0[u] = (int)ix.col[p0];
                    w1[u] = (int)ix.col[p1];
                }
                // every col[] value is consumed before the first store is issued (loads and stores share vmcnt on gfx9: with a
                // store in flight the wait for a load is a wait for that store's acknowledgement; sbwt_search_fused.hip)
                int v0[FP], v1[FP];
#pragma unroll
                for (int u = 0; u < FP; u++) {
                    v0[u] = (x0[u] >> 31) ? ((x0[u] == 0xFFFFFFFFu) ? -1 : (int)(x0[u] & 0x7FFFFFFFu)) : w0[u];
                    v1[u] = (x1[u] >> 31) ? ((x1[u] == 0xFFFFFFFFu) ? -1 : (int)(x1[u] & 0x7FFFFFFFu)) : w1[u];
                }
#pragma unroll
                for (int u = 0; u < FP; u++) asm volatile("" : "+v"(v0[u]), "+v"(v1[u]));
                if (!(ix.debug & 1)) {
#pragma unroll
                    for (int u = 0; u < FP; u++) {
                        if (fj[u] + 1 < fe[u]) st_res2(out, fob[u] + fj[u], (i64)v0[u], (i64)v1[u], ix.out32);
                        else if (fj[u] < fe[u]) st_res(out, fob[u] + fj[u], (i64)v0[u], ix.out32);
                    }
                }
                if (long_read) {
                    // reads of more than 128 results since their last flush: the remaining passes, one read at a time
#pragma unroll
                    for (int u = 0; u < FP; u++) {
                        if (fL[u] < 0) continue;
                        const int L = fL[u], ns = __shfl(nseg, L), a = __shfl(i0, L), e = fe[u];
                        const int tl = (tid & ~63) + L;
                        for (int base = a + 128; base < e; base += 128) {
                            const int j0 = base + 2 * lane, j1 = j0 + 1;
                            uint2 c0 = segs[0][tl], c1 = c0;
                            for (int sx = 1; sx < ns; sx++) {
                                const uint2 sg = segs[sx][tl];
                                if ((int)sg.y <= j0) c0 = sg;
                                if ((int)sg.y <= j1) c1 = sg;
                            }
                            const unsigned p0 = (j0 < e && !(c0.x >> 31)) ? c0.x + (unsigned)(j0 - (int)c0.y) : 0u;
                            const unsigned p1 = (j1 < e && !(c1.x >> 31)) ? c1.x + (unsigned)(j1 - (int)c1.y) : 0u;
                            const int q0 = (int)ix.col[p0], q1 = (int)ix.col[p1];
                            const int v0 = (c0.x >> 31) ? ((c0.x == 0xFFFFFFFFu) ? -1 : (int)(c0.x & 0x7FFFFFFFu)) : q0;
                            const int v1 = (c1.x >> 31) ? ((c1.x == 0xFFFFFFFFu) ? -1 : (int)(c1.x & 0x7FFFFFFFu)) : q1;
                            if (!(ix.debug & 1)) {
                                if (j1 < e) st_res2(out, fob[u] + j0, (i64)v0, (i64)v1, ix.out32);
                                else if (j0 < e) st_res(out, fob[u] + j0, (i64)v0, ix.out32);
                            }
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < FP; u++)
                    if (lane == fL[u]) { nseg = 0; i0 = i; }
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
                    if ((ix.debug & 2) && !ix.out32) out[d + sub] = val;
                    else st_res(out, d + sub, val, ix.out32);
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
                    if ((ix.debug & 2) && !ix.out32) out[d + sub] = -1;
                    else st_res(out, d + sub, -1, ix.out32);
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
                mode = PATH ? (rknown ? tnext : M_POS) : M_STREAM;   // SBWT.hh:560-
                l = res;
                if (PATH) j = 0;
            } else if (PATH && bridged) {
                mode = M_EXT;                          // (i < m: the burst was k k-mers long)
                r += (pos_t)k;
            } else {
                do_plan = true;                        // SBWT.hh:557-559 (with certificates)
            }
        }
        if (do_plan) {
            // where the next walk starts (see the header comment): at k-mer i itself, or close to
            // the last failure position b when b lies inside k-mer i's window
            int s0 = i, nwk = (ps > 0) ? 1 : 0;
            if ((mz & 3u) >= 2u) force = true;         // blind: the k-mer's own search
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
                    // a window that starts at b but runs past k-mer i: the last window inside k-mer i holds b as well
                    if (pfon && s0 == b && b + L0 - 1 > i + k - 1 && k > L0) s0 = i + k - L0;
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
        if (lane == 0) { if (e) atomicAdd(&ws->n_ext, e); if (eb) atomicAdd(&ws->n_bridge, eb); }
    }
    if (lane == 0) {   // the counters are wave-uniform (a wave that found no work adds nothing: single-address atomics are slow)
        if (c_stream) atomicAdd(&ws->n_stream, (u64)c_stream);
        if (c_search) atomicAdd(&ws->n_search, (u64)c_search);
        if (c_lf) atomicAdd(&ws->n_lf, (u64)c_lf);
        if (c_tab) atomicAdd(&ws->n_tab_hit, (u64)c_tab);
    }
}

// Do all reads have one length and all result ranges one stride?  (Sequencing reads usually do.)  Then the search
// kernel computes a read's offsets instead of fetching them: one iteration and two gathers less per read.
__global__ void __launch_bounds__(256) k_check_uniform(const i64 *__restrict__ read_off, const i64 *__restrict__ out_off,
                                                       i64 n_reads, SbwtWorkHeader *ws, int k, SbwtPieceTab pt) {
    // (rg_sample stays 0: a launch without the fused kernel in front)
    const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
    const i64 len = read_off[1] - read_off[0], stride = (n_reads > 1) ? out_off[1] - out_off[0] : 0;
    if (t == 0) { ws->u_read0 = read_off[0]; ws->u_len = len; ws->u_out0 = out_off[0]; ws->u_stride = stride; }
    const bool valid = t < n_reads;
    const i64 mylen = valid ? read_off[t + 1] - read_off[t] : 0;
    piece_zones_of_wave(t, mylen, valid, k, ws, pt);
    if (!valid) return;
    const bool bad = (mylen != len) || (t + 1 < n_reads && out_off[t + 1] - out_off[t] != stride);
    // (a plain store: every writer stores the same value.  An atomicAdd per wave serialises on one address -- 0.75 ms for
    // 4 M ragged reads, a quarter of the search itself)
    if (__ballot(bad) != 0 && (threadIdx.x & 63) == 0) ws->u_bad = 1ull;
}

// ---- long reads: zones -> pieces (SbwtPieceTab, sbwt_device.h) ----
// the first k-mer x' >= x of the read at base P0 whose window holds no lower-case acgt (packed groups: .z = ACGT after
// toupper, .w = ACGT as written), or m
__device__ static i64 piece_adjust(const uint4 *__restrict__ packed, i64 P0, i64 x, i64 m, int k) {
    i64 cand = x, gi = -1;
    unsigned low = 0;
    for (i64 t = x; cand < m; t++) {
        if (t - cand >= k) return cand;                // bases cand .. cand + k - 1 are clean
        const i64 g = (P0 + t) >> 5;
        if (g != gi) { const uint4 q = packed[g]; low = q.z & ~q.w; gi = g; }
        if ((low >> ((P0 + t) & 31)) & 1u) cand = t + 1;
    }
    return m;
}
__global__ void __launch_bounds__(256) k_piece_bounds(const uint4 *__restrict__ packed, const i64 *__restrict__ read_off,
                                                      const i64 *__restrict__ out_off, int k, const SbwtWorkHeader *ws,
                                                      SbwtPieceTab pt, int behind_fused, const unsigned *__restrict__ defer_bits) {
    // (the fused kernel took a batch of reads of one length, none of them long enough to be cut into zones)
    if (behind_fused && sbwt_fused_mode(ws, k) == 1 && !piece_read_is_cut(ws->u_len - k + 1, pt.piece)) return;
    const i64 z = (i64)blockIdx.x * 256 + threadIdx.x;
    const i64 np = (i64)ws->n_pieces < pt.cap ? (i64)ws->n_pieces : pt.cap;
    if (z >= np) return;
    const uint4 d = pt.outs[z];
    const i64 r = (i64)(((u64)d.y << 32) | (u64)d.x), j = (i64)d.z, nz = (i64)d.w;
    // (the fused kernel took the whole batch through its ticket table: the zones of a long read are searched behind it only when
    // that kernel handed the read on -- a read it answered itself gets empty zones)
    if (behind_fused && defer_bits && sbwt_fused_mode(ws, k) == 3 && !((defer_bits[r >> 5] >> (r & 31)) & 1u)) {
        const i64 P0e = read_off[r];
        pt.pairs[z] = make_uint4((unsigned)P0e, (unsigned)((u64)P0e >> 32), (unsigned)P0e, (unsigned)((u64)P0e >> 32));
        pt.outs[z] = make_uint4(0u, 0u, 0u, 0u);
        return;
    }
    const i64 P0 = read_off[r], m = read_off[r + 1] - P0 - k + 1, ob = out_off[r];
    const i64 s = j ? piece_adjust(packed, P0, j * pt.piece, m, k) : 0;
    const i64 e = (j + 1 == nz) ? m : piece_adjust(packed, P0, (j + 1) * pt.piece, m, k);
    const i64 b0 = P0 + (e > s ? s : 0), b1 = e > s ? P0 + e + k - 1 : P0;     // (an empty zone: a read of no bases)
    const i64 o0 = ob + (e > s ? s : 0);
    pt.pairs[z] = make_uint4((unsigned)b0, (unsigned)((u64)b0 >> 32), (unsigned)b1, (unsigned)((u64)b1 >> 32));
    pt.outs[z] = make_uint4((unsigned)o0, (unsigned)((u64)o0 >> 32), 0u, 0u);
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
void sbwt_launch_encode(const char *d_bases, long long total_bases, uint4 *d_packed, SbwtWorkHeader *ws,
                        hipStream_t stream) {
    i64 n_groups = (total_bases + SBWT_GROUP_BASES - 1) / SBWT_GROUP_BASES + 2;
    int aligned = ((uintptr_t)d_bases & 15) == 0;
    hipLaunchKernelGGL(k_encode, dim3(grid_for(n_groups)), dim3(256), 0, stream,
                       reinterpret_cast<const unsigned char *>(d_bases), (i64)total_bases, d_packed, n_groups, ws,
                       aligned);
}

void sbwt_launch_encode_chained(const char *d_bases, long long total_bases, uint4 *d_packed, SbwtWorkHeader *ws,
                                const unsigned *d_defer, const long long *d_read_off, int k, hipStream_t stream) {
    i64 n_groups = (total_bases + SBWT_GROUP_BASES - 1) / SBWT_GROUP_BASES + 2;
    int aligned = ((uintptr_t)d_bases & 15) == 0;
    const i64 want = (n_groups + 255) / 256;
    hipLaunchKernelGGL(k_encode_chained, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(256), 0, stream,
                       reinterpret_cast<const unsigned char *>(d_bases), (i64)total_bases, d_packed, n_groups, ws, d_defer, d_read_off, k,
                       aligned);
}

// the general path kernel behind k_search_fused: all reads when that kernel declined the batch, else the reads it handed on
// The general kernel's bridge compares the k-1 bases after a substitution in ONE iteration (at most 32 of them): for
// k > 32 it runs without the safe bits (a safe step is an only-successor step: -1 and the certificates, as before the
// bits existed); the fused kernel's multi-iteration compare (F_CMP, sbwt_search_fused.hip) uses them for any k <= 64.
// An index of 2^31 columns or more is served by this file's kernels on its blocks and dense prefix table alone (the 64-bit
// instantiation, as on a level-2 image): its derived structures hold full 32-bit unsigned columns and positions, which only
// the fused kernel's BIG instantiation reads (sbwt_search_fused.hip).  Here that is the route of the few reads the fused
// kernel hands on, of batches it declines, and of the cross-check variants.
static inline bool view_is_big(const SbwtIndexView &ix) { return ix.big || ix.n_nodes >= ((1ll << 31) - 128); }
static inline SbwtIndexView general_view(const SbwtIndexView &ix) {
    SbwtIndexView v = ix;
    if (v.k > 32) v.has_safe = 0;
    if (view_is_big(v)) {
        v.stab = nullptr; v.p_sparse = 0; v.n_sb = 0; v.stab_pos = 0;
        v.stab2 = nullptr; v.n_sb2 = 0;
        v.pfil = nullptr; v.p_filter = 0; v.log2f = 0;
        v.col = v.pos = nullptr; v.pq = nullptr; v.trans = nullptr; v.n_tslots = 0; v.has_safe = 0;
        v.n_pos = v.n_nodes;
    }
    return v;
}

void sbwt_launch_search_chained(const SbwtIndexView &ix_in, const uint4 *d_packed, const long long *d_read_off,
                                const long long *d_out_off, long long *d_out, long long n_reads, SbwtWorkHeader *ws,
                                int streaming, hipStream_t stream, const unsigned *d_defer, SbwtPieceTab pt) {
    const i64 want1 = (n_reads + pt.cap + 255) / 256;
    const SbwtIndexView ix = general_view(ix_in);
    const unsigned cap = (ix.debug >> 8) ? (unsigned)(ix.debug >> 8) : 1280u;
    const unsigned g = (unsigned)(want1 < (i64)cap ? want1 : (i64)cap);
    if (view_is_big(ix))
        hipLaunchKernelGGL((k_search_cert<true, 4, false>), dim3(g), dim3(256), 0, stream, ix, d_packed, d_read_off,
                           d_out_off, d_out, (i64)n_reads, ws, streaming, (const unsigned *)nullptr, d_defer, pt);
    else
        hipLaunchKernelGGL((k_search_cert<false, 4, true>), dim3(g), dim3(256), 0, stream, ix, d_packed, d_read_off,
                           d_out_off, d_out, (i64)n_reads, ws, streaming, (const unsigned *)nullptr, d_defer, pt);
}

void sbwt_launch_piece_bounds_tt(const uint4 *d_packed, const long long *d_read_off, const long long *d_out_off, int k,
                                 SbwtWorkHeader *ws, SbwtPieceTab pt, int behind_fused, hipStream_t stream, const unsigned *defer_bits) {
    if (!pt.pairs || pt.cap <= 0) return;
    hipLaunchKernelGGL(k_piece_bounds, dim3(grid_for(pt.cap)), dim3(256), 0, stream, d_packed, d_read_off, d_out_off, k, ws, pt,
                       behind_fused, defer_bits);
}
void sbwt_launch_piece_bounds(const uint4 *d_packed, const long long *d_read_off, const long long *d_out_off, int k,
                              SbwtWorkHeader *ws, SbwtPieceTab pt, int behind_fused, hipStream_t stream) {
    sbwt_launch_piece_bounds_tt(d_packed, d_read_off, d_out_off, k, ws, pt, behind_fused, stream, nullptr);
}

void sbwt_launch_search(const SbwtIndexView &ix_in, const uint4 *d_packed, const long long *d_read_off,
                        const long long *d_out_off, long long *d_out, long long n_reads, SbwtWorkHeader *ws,
                        int streaming, hipStream_t stream, int variant, long long total_groups, void *d_sort_scratch,
                        long long sort_scratch_bytes, int sort_key_bits, SbwtPieceTab pt) {
    if (n_reads <= 0) return;
    const SbwtIndexView ix = general_view(ix_in);
    const unsigned *d_perm = nullptr;
    if (variant >= 1) {
        if (d_sort_scratch) pt = SbwtPieceTab();       // (sorted tickets: whole reads)
        i64 want1 = (n_reads + pt.cap + 255) / 256;
        // the check kernel notes what the path kernel wants to know about the offsets and lists the zones of long reads
        if (!(ix.debug & 8) || pt.pairs)
            hipLaunchKernelGGL(k_check_uniform, dim3(grid_for(n_reads)), dim3(256), 0, stream, d_read_off, d_out_off,
                               (i64)n_reads, ws, ix.k, pt);
        if (ix.debug & 8) (void)hipMemsetAsync(&ws->u_bad, 0xFF, 8, stream);       // experiment: the general offset path
        sbwt_launch_piece_bounds(d_packed, d_read_off, d_out_off, ix.k, ws, pt, 0, stream);
        unsigned grid1 = (unsigned)(want1 < 2048 ? want1 : 2048);
        // 32-bit positions need every column index (and n_nodes + 64) below 2^31 and < 2^31 packed groups
        const bool wide = view_is_big(ix) || total_groups >= (1ll << 31) - 4 || (ix.debug & 16);
        // no-spill build: 72 VGPRs (7 waves/SIMD max); 4 workgroups per CU measured best (tools/ab_bench.py)
        unsigned cap = (ix.debug >> 8) ? (unsigned)(ix.debug >> 8) : (variant >= 2 ? 1280u : 1024u);
        unsigned g = grid1 < cap ? grid1 : cap;
        if (wide)
            hipLaunchKernelGGL((k_search_cert<true, 4, false>), dim3(g), dim3(256), 0, stream, ix, d_packed, d_read_off,
                               d_out_off, d_out, (i64)n_reads, ws, streaming, d_perm, (const unsigned *)nullptr, pt);
        else if (variant >= 2 && ix.col && streaming) {    // path order (the default when the index has one)
            if (d_sort_scratch)
                d_perm = sbwt_launch_sort_reads(ix, d_packed, d_read_off, n_reads, ws, d_sort_scratch, sort_scratch_bytes,
                                                sort_key_bits, stream);
            hipLaunchKernelGGL((k_search_cert<false, 4, true>), dim3(g), dim3(256), 0, stream, ix, d_packed, d_read_off,
                               d_out_off, d_out, (i64)n_reads, ws, streaming, d_perm, (const unsigned *)nullptr, pt);
        } else
            hipLaunchKernelGGL((k_search_cert<false, 4, false>), dim3(g), dim3(256), 0, stream, ix, d_packed, d_read_off,
                               d_out_off, d_out, (i64)n_reads, ws, streaming, d_perm, (const unsigned *)nullptr, pt);
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
