// sbwt_search_fused.hip -- the streaming-search kernel for batches of equal-length reads (sequencing reads): the 2-bit
// re-encoding of the bases is part of the search kernel, and a read lives in LDS while its lane works on it.
//
// Why (round-2 profile of k_search_cert<PATH,SEG>): the separate k_encode pass cost 0.3-0.4 ms and
// 2.25 GB of traffic per 10 M reads, and the search kernel then fetched a read's packed groups again from HBM, two at a
// time, 2.7 lines and 1.3 extra lane-iterations per read, with 12 registers and the tag / reload logic to cache them.
// Here a wave's pool of 64 tickets is 64 CONTIGUOUS reads: when the pool runs dry every lane loads one of them (150 bytes
// each, 1.2 lines per read, every byte of a line used), encodes it with all 64 lanes busy and parks the 2-bit codes in a
// pool area in LDS; a lane that takes a ticket copies its read into its own LDS slot.  Bases are then LDS reads at any
// offset: no reload state, no prefetch loads, no validity words --
//
// -- because this kernel only walks reads whose bases are all upper-case ACGT.  A read with any other byte (N, lower
// case, ...) is HANDED ON: its number goes to a list, and the general kernel (k_search_cert<PATH,SEG>, sbwt_search.hip)
// runs behind this one over that list with the reference's validity rules (SBWT.hh:398-399,427-428,565-568).  With only
// valid bases the streaming step's and the full search's validation agree, so `streaming` 1 and 2 are the same here.
//
// F_CMP (round 4): the read compared with the chars of the path it is ALIGNED to (read base j <-> path step co + j), up to
// 64 bases per iteration, until the range is done or a second difference shows.  Two ways in:
//   bridge  the read leaves its path at a substitution-safe step u: the next k-1 bases agree again -> the k k-mers that hold
//           u are absent by the safe bit, on along the path (any k <= 64: as many iterations as the k-1 bases need)
//   anchor  a k-mer's lookup failed and nothing says where in its window the bad base is: instead of bisecting with range
//           probes, look the k-mer just past the suspect range up (the ANCHOR: one exact lookup); if it is there, its path
//           position aligns the whole stretch, one or two compares find every difference exactly: k-mers without one are
//           the path's own (a run out of col[]), k-mers with exactly one safe difference are absent by its safe bit, and
//           what is left (two differences in one window, an unsafe step) goes to the certificates with the bad base KNOWN.
// Every conclusion is exact -- a k-mer whose k bases equal k consecutive chars of one path IS that path's k-mer; a safe bit
// is a precomputed certificate (k_path_safe_labels*) -- so a wrong guess only costs the lookup.
//
// The state machine is otherwise the one of k_search_cert<PATH,SEG> (same certificates, same path-order steps, same segment-list
// writer; see the header comments there).  Reference semantics: SBWT::streaming_search include/sbwt/SBWT.hh:544-581,
// SBWT::search :389-415, update_sbwt_interval :422-437.
#include "sbwt_kernels_common.h"

#define F_IDLE 0
#define F_INIT 2
#define F_STEP 3
#define F_DEAD 4
#define F_EXT 7
#define F_TRANS 8
#define F_POS 9
#define F_BRIDGE 10             // entry of F_CMP from a substitution-safe step (set up at the top of the next iteration)
#define F_CMP 11                // the read against its path's chars, 33 .. 64 bases per iteration: bridges and anchors (below)
#define F_PLAN 12               // SORT: the planner has to choose this read's next walk (it runs at the top of a searcher wave's iteration)
#define FE_NONE 0
#define FE_EMIT1 1
#define FE_FAIL 2
#define FE_END 3
#define FE_PRES 4
#define FE_ANCH 5               // an anchor lookup found its k-mer: tpos = its path position
#ifndef FZ_ANCHORS
#define FZ_ANCHORS 3            // anchor lookups per read at most (a read of unrelated sequence pays that many lines for nothing)
#endif
#ifndef FZ_ALIGNS
#define FZ_ALIGNS 7             // seeds + resumed compares per read at most
#endif
// F_CMP flags: the remembered differences' step states (S: substitution-safe, A / B: the path group's state bits), ...
#define CF_S1 1u
#define CF_A1 2u
#define CF_B1 4u
#define CF_S2 8u                // (the second difference's three bits: the first one's << 3)
#define CF_A2 16u
#define CF_B2 32u
#define CF_ONP 64u              // ... k-mer i-1 is the path's own, at position co + i - 1 + k
#define CF_ALIGNED 128u         // co is the read's last known alignment (a compare may be resumed on it)
#define CF_ANCH 256u            // the lookup in flight is an anchor (wstart = its k-mer), not a walk inside k-mer i's window
#define CF_SEED 512u            // the position lookup in flight is a seed's
#define CA_GOON 0               // F_CMP goes on with the next bases
#define CA_DONE 1               // the lane's last k-mer is answered
#define CA_CERT 2               // to the certificates, b = m1
#define CA_TRANS 3              // a transition lookup at position ctr
#define CA_ABSENT 4             // one -1 (an only-successor step), then the certificates
#define CA_LOST 5               // the alignment ran off its path: forget it
#define CF_MISS 1024u            // two bits: own lookups of consecutive k-mers that failed without a certificate reaching further
#define CF_MISS_MASK 3072u
#define CF_WIN31 8192u           // k > 31: the sparse lookup in flight is a 31-base WINDOW probe (a certificate), not a k-mer's prefix
#define CF_SEED2 16384u          // k > 31: the seed's 31-prefix has TWO columns (l, l + 1): after the position lookup l holds the second
                                 // one's path position -- tried when the compare on the first answers nothing
#define CF_NOCERT 4096u          // since the last own lookup: windows were probed, and every one of them is (perhaps) in the index
#define CF_FORCE (1u << 30)      // SORT: the planner is to start k-mer i's own search (a probe was inconclusive)
#define CF_ANC_LEFT(f) (((f) >> 16) & 3u)
#define CF_CMP_LEFT(f) (((f) >> 18) & 7u)
#define CF_ANC_TRIED(f) ((int)(((f) >> 21) & 511u) - 1)
#define CF_SET_TRIED(f, a) ((f) = ((f) & ~(511u << 21)) | ((((unsigned)((a) + 1)) & 511u) << 21))
#define CF_BUDGETS (((unsigned)FZ_ANCHORS << 16) | ((unsigned)FZ_ALIGNS << 18))
#define CF_M1 (CF_S1 | CF_A1 | CF_B1)
#define CF_M2 (CF_S2 | CF_A2 | CF_B2)

#ifndef FZ_SPLIT_MIN
#define FZ_SPLIT_MIN 16         // the tail: a lane gives away half of its remaining k-mers when it has at least this many left
#endif
#define FZ_NSEG 9               // segments per lane: 256 x (9 x 4 + 9 x 1) B + 2 x 256 x 40 B of codes = 32 000 B = 5 workgroups per CU

typedef unsigned fz_u32x4 __attribute__((ext_vector_type(4)));

#ifdef SBWT_STATS
#ifndef SBWT_SLOW_LO
#define SBWT_SLOW_LO 100
#define SBWT_SLOW_HI 100000
#endif
// stats builds (tools/build_stats_lib.sh): lane-iterations per read (ticket or tail piece), bucket = iterations / 2
__device__ unsigned long long g_iter_hist[64];
__device__ unsigned long long g_tail_prof[192];       // [q], [64 + q], [128 + q]: waves / busy lanes / donor lanes at tail iteration q
extern "C" int sbwtgpu_debug_tail_prof(unsigned long long *out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tail_prof), sizeof(g_tail_prof)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[192] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_tail_prof), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
__device__ unsigned long long g_iter_max[8];          // (iterations << 32 | read) of the slowest reads seen, by iterations mod 8
extern "C" int sbwtgpu_debug_iter_hist(unsigned long long *out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_iter_hist), sizeof(g_iter_hist)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[64] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_iter_hist), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
extern "C" int sbwtgpu_debug_iter_max(unsigned long long *out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_iter_max), sizeof(g_iter_max)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_iter_max), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
// why substitutions are (not) bridged, k <= 31 (tools/lane_stats_fused.py prints the names)
__device__ unsigned long long g_fz_why[24];
extern "C" int sbwtgpu_debug_why(unsigned long long *out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fz_why), sizeof(g_fz_why)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[24] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_fz_why), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#define FZ_WHY(q, cond) do { const unsigned long long n_ = __popcll(__ballot(cond)); if (lane == 0 && n_) atomicAdd(&g_fz_why[q], n_); } while (0)
// what the planner starts and what comes of it (tools/lane_stats_fused.py prints the names)
__device__ unsigned long long g_fz_plan[48];
extern "C" int sbwtgpu_debug_plan(unsigned long long *out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fz_plan), sizeof(g_fz_plan)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[48] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_fz_plan), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#define FZ_PLAN(q, cond) do { const unsigned long long n_ = __popcll(__ballot(cond)); if (lane == 0 && n_) atomicAdd(&g_fz_plan[q], n_); } while (0)
#define FZ_PLAN_SUM(q, val) do { unsigned long long v_ = (unsigned long long)(val); for (int o_ = 32; o_ > 0; o_ >>= 1) v_ += __shfl_down(v_, o_); if (lane == 0 && v_) atomicAdd(&g_fz_plan[q], v_); } while (0)
#define FZ_HIST_FLUSH() do { if (it_cnt) { atomicAdd(&g_iter_hist[(it_cnt >> 1) < 63 ? (it_cnt >> 1) : 63], 1ull); \
    if (it_cnt >= SBWT_SLOW_LO && it_cnt < SBWT_SLOW_HI) atomicMax(&g_iter_max[it_cnt & 7u], ((unsigned long long)it_cnt << 32) | rd); it_cnt = 0; } } while (0)
#else
#define FZ_HIST_FLUSH() do { } while (0)
#define FZ_WHY(q, cond) do { } while (0)
#define FZ_PLAN(q, cond) do { } while (0)
#define FZ_PLAN_SUM(q, val) do { } while (0)
#endif

// timeline builds (tools/build_stats_lib.sh timeline): when the waves of one launch start, see the ticket counter run out, and
// leave (100 MHz clock).  g_fz_tl: [0] earliest start, [1] earliest / [2] latest "drained", [3] latest exit, [4] waves,
// [5] sum of exit - drained, [6] sum of exit - start, [7] iterations before "drained"; g_fz_tl_hist[b]: waves that left in the
// b-th 10 us after [0].  (Per-iteration clocks in the tail were tried and dropped: 5 120 waves adding to the same counters every
// iteration slow the tail they measure.)
#ifdef SBWT_TIMELINE
__device__ unsigned long long g_fz_tl[8];
__device__ unsigned long long g_fz_tl_hist[1024];
extern "C" int sbwtgpu_debug_timeline(unsigned long long *out, int reset) {
    if (out && (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fz_tl), sizeof(g_fz_tl)) != hipSuccess ||
                hipMemcpyFromSymbol(out + 8, HIP_SYMBOL(g_fz_tl_hist), sizeof(g_fz_tl_hist)) != hipSuccess)) return -1;
    if (reset) {
        static unsigned long long z[1024];
        unsigned long long t0[8] = {~0ull, ~0ull, 0, 0, 0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_fz_tl), t0, sizeof(t0)) != hipSuccess ||
            hipMemcpyToSymbol(HIP_SYMBOL(g_fz_tl_hist), z, sizeof(g_fz_tl_hist)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

// 32 ASCII bases (8 dwords) -> 64 bits of 2-bit codes; `bad` becomes non-zero when a byte of the read is not one of "ACGT":
// the codes select the letters they stand for out of "ACGT" (one v_perm_b32 per four bases), and a byte that is not that
// letter differs from it; tm[d] masks the bytes of dword d that belong to the read (0xFF each)
__device__ __forceinline__ u64 fz_encode32(const unsigned w[8], const unsigned tm[8], unsigned &bad) {
    u64 codes = 0;
#pragma unroll
    for (int d = 0; d < 8; d++) {
        const unsigned x = w[d];
        const unsigned t = ((x >> 1) & 0x03030303u) ^ ((x >> 2) & 0x01010101u);          // dna_code of every byte
        const unsigned c8 = (t * 0x01041040u) >> 24;                                     // 4 x 2 bits -> one byte
        const unsigned letters = __builtin_amdgcn_perm(0x54474341u, 0x54474341u, t);     // "ACGT"[code] for every byte
        bad |= (letters ^ x) & tm[d];
        codes |= (u64)c8 << (8 * d);
    }
    return codes;
}

// O32: the results are written as int32 (sbwtgpu_*_dev_i32).  WIDE = false: k <= 31 -- a read follows its path in F_EXT, a substitution-safe step is bridged in one compare (F_BRIDGE):
// round 3's walk, the leanest code for k-mers that cost one lookup.  WIDE = true: 31 < k <= 63 (and "debug" bit 64) -- both
// are the one state F_CMP, with anchors, seeds and resumed compares around it.
// BIG (round 5): an index of 2^31 .. 2^32 - 2^24 columns (k <= 31, int64 results).  Columns and path positions are then full
// 32-bit UNSIGNED values: nothing may sign-extend them, bit 31 of a segment's source is part of a position (such an image
// answers every found k-mer with its position: no result is known by its column only), and the block counts are absolute
// (C[c] + rank fits 32 bits; the image's mega table is all zero).  0xFFFFFFFF stays "none" / -1.
// UNI (round 5): the instantiation for batches of ONE read length taken as one ticket per read -- sequencing reads, the
// usual case.  Offsets, lengths and piece numbers are then compile-time facts: no read_off / out_off, no per-ticket lengths,
// no division of a ticket number by the pieces per read.  Both instantiations are launched; each returns at once from a
// batch that is the other's ("debug" bit 128: the general one takes everything).
// SORT (round 6): lanes sorted by state.  Waves 0-1 of a workgroup are SEARCHERS (they take the tickets and run F_INIT / F_STEP /
// F_POS, the events and the planner), waves 2-3 are PATH FOLLOWERS (F_EXT / F_TRANS / F_BRIDGE); every wave runs the writer for
// the lists its own lanes finished.  A read lives in a SLOT of the workgroup's LDS (its codes, its segment list, four words of
// state); a lane whose read changes class writes the state, pushes the slot's number into the other class's ring and goes
// idle; idle lanes hold a place in their class's ring and take what arrives there.  Each wave then executes only its class's
// blocks of the state machine (the rest are skipped by wave-uniform branches).  See DESIGN.md section 3.
template <bool WIDE, bool O32, bool BIG = false, bool UNI = false, bool SORT = false>
__global__ void __launch_bounds__(256, 5) k_search_fused(SbwtIndexView ix, const unsigned char *__restrict__ bases,
                                                          i64 total_bases, i64 *__restrict__ out, i64 n_reads,
                                                          SbwtWorkHeader *ws, unsigned *__restrict__ defer_list,
                                                          const i64 *__restrict__ read_off, const i64 *__restrict__ out_off) {
    static_assert(!SORT || (!WIDE && !BIG), "lanes sorted by state: k <= 31, fewer than 2^31 columns (so far)");
    __shared__ u64 pool_codes[SBWT_FUSED_MAXG][SORT ? 128 : 256];   // the wave's pool of 64 tickets, encoded ([group][wave * 64 + ticket]; SORT: searcher waves only)
    __shared__ u64 cur_codes[SBWT_FUSED_MAXG][256];         // the read this lane is working on (SORT: by slot)
    __shared__ unsigned seg_src[FZ_NSEG][256];              // segment lists: source ...
    __shared__ unsigned char seg_at[FZ_NSEG][256];          // ... and first k-mer
    // SORT: a slot's state while it waits in a ring -- [0] read, [1] path position r (to a follower) / b+1 | blo+1 << 8 (to a
    // searcher), [2] i | mend << 8 | nseg << 16 | kind << 20 (0 free slot, 1 F_EXT, 2 F_PLAN) | force << 22 | fl's miss and nocert
    // bits << 23 | piece << 26 | list holds a column-only result << 28, [3] first result not written | bnext+1 << 8
    __shared__ unsigned st_w[SORT ? 4 : 1][SORT ? 256 : 1];
    __shared__ unsigned short q_ring[SORT ? 2 : 1][SORT ? 256 : 1];   // the rings: [0] to the searchers, [1] to the followers; slot + 1, 0 = not written yet
    __shared__ unsigned q_ctl[8];                           // [0], [1] places handed out in ring 0 / 1; [2], [3] entries written; [4] reads in flight; [5] searcher waves drained
    const int fmode = sbwt_fused_mode(ws, ix.k);
    if (fmode == 0) return;                                 // the general route does it all
    const int P_batch = sbwt_fused_pieces(ws, ix.k);
    if (UNI != (fmode == 1 && P_batch == 1 && !(ix.debug & 128))) return;      // the other instantiation's batch
    const bool ragged = !UNI && fmode == 2;                 // reads of any lengths: offsets fetched with every refill
    const int tid = threadIdx.x, lane = tid & 63, wbase = tid & ~63;
    const bool isP = SORT && __builtin_amdgcn_readfirstlane(tid) >= 128;     // a path-follower wave (wave-uniform, in a scalar register)
    const bool runS = !SORT || !isP, runP = !SORT || isP;   // (wave-uniform: which blocks of the state machine this wave executes)
    const int qc = isP ? 1 : 0;                             // this wave's ring
    int slot = SORT ? (isP ? -1 : tid) : tid;               // SORT: the slot this lane holds (-1: none)
    int qpos = -1;                                          // SORT: this lane's place in its ring (-1: none)
    if (SORT) {
        // slots 0 .. 127 start with the searcher lanes, 128 .. 255 wait in the searchers' ring as free slots
        if (tid < 8) q_ctl[tid] = (tid == 2) ? 128u : 0u;
        if (tid < 128) { q_ring[0][tid] = (unsigned short)(128 + tid + 1); q_ring[0][128 + tid] = 0; st_w[2][128 + tid] = 0u; }
        else { q_ring[1][tid - 128] = 0; q_ring[1][tid] = 0; }
        __syncthreads();
    }
    volatile unsigned *const vctl = q_ctl;
    // entries of a ring: written after the slot's state (LDS operations of a wave complete in order; the wait keeps the compiler
    // and the hardware from letting the entry pass the state), read before it (the state's address depends on the entry)
    auto ring_push = [&](const u64 mask, const int ring, const int s) {
        // (wave-uniform call: every lane of `mask` appends slot s to `ring`)
        unsigned base = 0;
        if (lane == 0) base = __hip_atomic_fetch_add(&q_ctl[2 + ring], (unsigned)__popcll(mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        base = uniform32(base);
        if ((mask >> lane) & 1ull) {
            const unsigned at = (base + (unsigned)__popcll(mask & low_mask(lane))) & 255u;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            *(volatile unsigned short *)&q_ring[SORT ? ring : 0][SORT ? at : 0] = (unsigned short)(s + 1);
        }
    };
    const int k = ix.k, p = ix.p_dev, L0 = ix.probe_len, ps = ix.p_sparse;
    // tickets: P per read (reads of more than 160 bases as pieces that overlap by k-1; sbwt_kernels_common.h)
    const int P = UNI ? 1 : P_batch, kpp = SBWT_FUSED_MAXLEN - k + 1;
    const bool varlen = ragged || P > 1;                    // the tickets' lengths differ: each refill notes them
    const i64 n_tickets = n_reads * P;
    const int ulen = ragged ? SBWT_FUSED_MAXLEN : (int)ws->u_len, m = ulen - k + 1, G = varlen ? SBWT_FUSED_MAXG : ((ulen + 31) >> 5);
    const i64 u_read0 = ws->u_read0, u_out0 = ws->u_out0, u_stride = ws->u_stride;
    const bool pfon = ix.pfil && ix.p_filter == L0 && L0 > p;
    const u64 mk2 = (k - ps >= 32) ? ~0ull : low_mask(2 * ((k - ps) & 31));   // key mask of the second-level window
    const int pw = pfon ? L0 : p;                   // window of a range probe: the filter's when there is one
    const int last_node = (int)(ix.n_nodes - 1);
    static_assert(!BIG || (!WIDE && !O32), "2^31 columns and more: k <= 31, int64 results");
    auto zx = [](int v) -> i64 { return BIG ? (i64)(unsigned)v : (i64)v; };                    // a column / position as an index
    auto is_run = [](unsigned src) -> bool { return BIG ? src != 0xFFFFFFFFu : !(src >> 31); };   // a segment's source is a path position

    int nseg = 0, i0 = 0, last_start = 0;           // segments listed; first result of the read not written yet
    u64 fm_pend = 0;                // wave-uniform: lanes whose segment list waits for the writer (handed over at the end of an iteration)
    int pend = 0;                   // ... that list: first result | one past the last << 8 | segments << 16 | piece << 20 | holds a result known by column only << 22
    unsigned pend_rd = 0;           // ... and its read
    unsigned last_src = 0, emit_pos = 0;
    int wk = 0;                     // how this walk starts (as in k_search_cert): 0 dense table, 1 sparse table, 2 probe filter,
                                    // 3 range probe, 5 second-level sparse lookup
    u64 hk = 0;                     // F_INIT: the window's key (filter: the bit positions), kept across the gather
    int blo = -1;                   // the last failure is known to lie in [blo, b]
    // F_CMP: read base j <-> path step co + j; bases [cP, cE) still to compare; the first two differences seen (read
    // positions, -1: none) and whether the first one's step is substitution-safe
    int co = 0, cP = 0, m1 = -1;    // (m2 lives inside one iteration: a second difference is dealt with at once)
    unsigned fl = 0;                // CF_* flags
    // (in fl's upper bits, to save registers: anchor lookups this read may still make, bits 16-17; alignments -- seeds, resumed
    // compares -- it may still start, bits 18-20; the last anchor k-mer that was not there + 1, bits 21-29)
    // Anchors, seeds and resumed compares pay where a whole k-mer costs two lookups (31 < k <= 63: second-level table) and the
    // certificates' probes are short against k; for k <= 31 the certificates alone measured faster (config 2: -1.3 %).
    // "debug" bits: 32 = no anchors / seeds / resumes, 64 = all of them for k <= 31 as well.
    const bool wide_k = WIDE && ((ix.stab2 != nullptr && ps < k) || (ix.stab_pos && ps == k)) && !(ix.debug & 32);
    const bool anch_ok = wide_k;
    const bool seed_ok = wide_k && ps < k;
    int bnext = -1;                 // a hint: the read's next difference from its path after b (a failed bridge's compare saw it)
    int mode = F_IDLE;
    unsigned rd = 0;                // the read this lane works on
    int i = 0, j = 0, b = -1, wstart = 0;
    int mend = 0;                   // this lane answers k-mers [i, mend) of its read (mend = m unless the read was split, below)
    bool drained = false;           // wave-uniform: the ticket counter has run past the last read
    int l = 0, r = 0;               // walk interval; F_EXT ..: r = path position
    unsigned c_ext = 0, c_brg = 0;  // per lane: k-mers answered along paths, substitutions bridged
    u64 pool_next = 0, pool_end = 0, pool_bad = 0;  // wave-uniform pool of read tickets; tickets of it that are handed on
    int pool_len = 0;               // varlen: the length of the piece this lane encoded at the last refill (161: too long) | piece << 16
    // (the piece number of the lane's ticket rides in i0's bits 16..23: its first k-mer within its read is piece * kpp; bit 24:
    // the list holds a result known by its column only)
    unsigned c_stream = 0, c_search = 0, c_lf = 0, c_tab = 0;     // wave-uniform work counters
#ifdef SBWT_STATS
    unsigned it_cnt = 0, tail_it = 0;
#endif
#ifdef SBWT_TIMELINE
    const u64 tl_start = wall_clock64();
    u64 tl_drain = 0;
    unsigned tl_n = 0;
    if (lane == 0) atomicMin(&g_fz_tl[0], tl_start);
#endif

    for (;;) {
#ifdef SBWT_STATS
        int pl_kind = -1, oc_kind = -1, cert_n = 0;      // planner statistics (g_fz_plan)
#endif
        bool force = false;                            // the planner starts k-mer i's own search
        bool w31 = false;                              // k > 31: a filter window was (perhaps) present: try the 31-base window around b
        // the planner: where the next walk of this lane's read starts (SORT: called at the top of a searcher wave's iteration,
        // otherwise at the end of the iteration that asked for it)
        auto plan_walk = [&](bool force, const bool w31) {
            // where the next walk starts (see k_search_cert): at k-mer i itself, or close to the last failure position b
            // when b lies inside k-mer i's window
            int s0 = i, nwk = (ps > 0) ? 1 : 0;
            // k > 31: the 16-base window at b is (perhaps) in the index -- as another strain's variant, usually.  The 31-base
            // window that holds b and starts as late as k-mer i allows is an exact lookup in the sparse table, and absent far
            // more often; it answers up to k - 30 k-mers.
            bool win31 = false;
            if (w31 && (fl & CF_MISS_MASK) < 2u * CF_MISS && b >= i && b <= i + k - 1) {
                const int ws = b < i + k - ps ? b : i + k - ps;
                if (ws > i) { win31 = true; force = false; s0 = ws; }
            }
            if ((fl & CF_MISS_MASK) >= 2u * CF_MISS) force = true;      // blind: the k-mer's own search
            // nothing known about k-mer i's window, but a bridge compare has seen the read's next difference inside it: two
            // substitutions within k-1 bases -- start the certificates there instead of bisecting for it (a hint like b
            // itself: the probes prove what they prove wherever they start)
            bool hinted = false;
            if (!force && pfon && !(b >= i && b <= i + k - 1) && bnext >= i && bnext <= i + k - 1) { b = blo = bnext; bnext = -1; hinted = true; }
            if (win31) {
                nwk = 1;
                fl |= CF_WIN31;
            } else if (!force && L0 > 0 && b >= i && b <= i + k - 1) {
                const int lo = blo > i ? blo : i;
                if (lo < b && anch_ok && CF_ANC_LEFT(fl) > 0 && b + 1 != CF_ANC_TRIED(fl) && b + 1 <= mend - 1) {
                    // the bad base is somewhere in [lo, b]: the k-mer just past the range as an anchor (F_CMP) instead of
                    // halving the range probe by probe
                    s0 = b + 1;
                    nwk = 1;
                    fl |= CF_ANCH;
                    fl -= 1u << 16;
                } else if (lo < b && p > 0 && k - pw >= 1) {
                    // the bad base is somewhere in [lo, b]: halve the range with a window that starts inside it
                    int x = lo + ((b - lo + 1) >> 1);
                    if (x > i + k - pw) x = i + k - pw;
                    if (x <= i) x = i + 1;
                    s0 = x;
                    nwk = 3;
                } else {
                    s0 = (b - i >= L0 - 1) ? (b - L0 + 1) : b;
                    // a window that starts at b but runs past k-mer i (k = 31 with probes of 18 bases: three windows per bad
                    // base, not two): the last window inside k-mer i holds b as well
                    if (pfon && s0 == b && b + L0 - 1 > i + k - 1 && k > L0) s0 = i + k - L0;
                    if (s0 + p - 1 > i + k - 1) s0 = i;
                    if (s0 != i) nwk = (pfon && s0 + L0 - 1 <= i + k - 1) ? 2 : 0;
                    // a hinted probe that finds its window in the index gives up the hint instead of walking the window
                    if (hinted) { if (nwk == 2) nwk = 6; else { s0 = i; nwk = (ps > 0) ? 1 : 0; b = -1; } }
                }
            }
#ifdef SBWT_STATS
            pl_kind = nwk == 1 ? ((fl & CF_MISS_MASK) >= 2u * CF_MISS ? 2 : force ? 1 : (b >= i && b <= i + k - 1) ? 8 : 0) :
                      nwk == 2 ? (s0 == b ? 4 : s0 == b - L0 + 1 ? 3 : 9) : nwk == 3 ? 5 : nwk == 6 ? 6 : 7;
#endif
            wstart = s0;
            j = 0;
            wk = nwk;
            if (p > 0) mode = F_INIT;
            else { mode = F_STEP; l = 0; r = last_node; }
                };
        if (SORT) {
            // ---- the workgroup is done when both searcher waves have seen the tickets run out and no read is in flight ----
            const unsigned n_drained = vctl[5], n_live = vctl[4];
            if (isP) drained = n_drained >= 2u;
            if (n_drained >= 2u && n_live == 0u && fm_pend == 0) break;
            // ---- idle lanes without a slot hold a place in their class's ring and take what arrives there: a read that changed
            //      class (with its state), or -- searchers -- a free slot for the next ticket ----
            const u64 wantq = __ballot(mode == F_IDLE && slot < 0 && qpos < 0);
            if (wantq) {
                unsigned base = 0;
                if (lane == 0) base = __hip_atomic_fetch_add(&q_ctl[qc], (unsigned)__popcll(wantq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                base = uniform32(base);
                if ((wantq >> lane) & 1ull) qpos = (int)((base + (unsigned)__popcll(wantq & low_mask(lane))) & 0x7FFFFFFFu);
            }
            if (qpos >= 0) {
                volatile unsigned short *const e = &q_ring[SORT ? qc : 0][SORT ? (qpos & 255) : 0];
                const unsigned v = *e;
                if (v) {
                    *e = 0;
                    qpos = -1;
                    slot = (int)v - 1;
                    const int sx = SORT ? slot : 0;
                    const unsigned w2 = st_w[SORT ? 2 : 0][sx];
                    const unsigned kind = (w2 >> 20) & 3u;
                    if (kind != 0u) {
                        const unsigned w1 = st_w[SORT ? 1 : 0][sx], w3 = st_w[SORT ? 3 : 0][sx];
                        rd = st_w[0][sx];
                        i = (int)(w2 & 255u);
                        mend = (int)((w2 >> 8) & 255u);
                        nseg = (int)((w2 >> 16) & 15u);
                        i0 = (int)(w3 & 255u) | (int)(((w2 >> 26) & 3u) << 16) | (int)(((w2 >> 28) & 1u) << 24);
                        fl = CF_BUDGETS | (((w2 >> 23) & 7u) << 10) | (((w2 >> 22) & 1u) ? CF_FORCE : 0u);
                        bnext = (int)((w3 >> 8) & 255u) - 1;
                        const int nl = nseg > 0 ? nseg - 1 : 0;
                        last_src = seg_src[nl][slot];
                        last_start = (int)seg_at[nl][slot];
                        j = 0;
                        if (kind == 1u) { mode = F_EXT; r = (int)w1; b = -1; blo = -1; }
                        else { mode = F_PLAN; b = (int)(w1 & 255u) - 1; blo = (int)((w1 >> 8) & 255u) - 1; }
                    }
                }
            }
        }
        // ---- hand out reads to idle lanes from the wave's ticket pool; an empty pool is refilled with 64 encoded reads ----
        const u64 need = SORT ? (isP ? 0ull : __ballot(mode == F_IDLE && slot >= 0)) : __ballot(mode == F_IDLE);
        if (need && !drained) {
            if (pool_next == pool_end) {
                u64 t = 0;
                if (lane == 0) t = atomicAdd(&ws->ticket, 64ull);
                pool_next = uniform64(t);
                pool_end = pool_next + 64;
                pool_bad = 0;
                if ((i64)pool_next < n_tickets) {
                    i64 woff = u_read0 + (i64)pool_next * ulen;                       // first byte of the pool's 64 reads
                    unsigned vrel = (unsigned)lane * (unsigned)ulen;                  // this lane's read in there
                    int len_l = ulen;
                    bool toolong = false;
                    int pc = 0;
                    if (varlen) {
                        // this lane's ticket: piece pc of read tr
                        i64 tk = (i64)pool_next + lane;
                        if (tk > n_tickets) tk = n_tickets;
                        const i64 tr = P == 1 ? tk : P == 2 ? (tk >> 1) : tk / 3;
                        pc = (int)(tk - tr * P);
                        i64 rstart, rlen;
                        if (ragged) {
                            const i64 ta = tr < n_reads ? tr : n_reads, tb = tr + 1 < n_reads ? tr + 1 : n_reads;
                            rstart = read_off[ta];
                            rlen = read_off[tb] - rstart;
                        } else {
                            rstart = u_read0 + tr * ulen;
                            rlen = tr < n_reads ? ulen : 0;
                        }
                        toolong = rlen > sbwt_fused_limit(P, k);
                        const i64 pstart = rstart + (i64)pc * kpp, plen = rlen - (i64)pc * kpp;
                        woff = (i64)uniform64((u64)rstart);      // (lane 0's read: no later ticket starts before it)
                        const i64 rel = pstart - woff;
                        vrel = (rel < 0 || rel > 0xFFFF0000ll) ? 0xFFFF0000u : (unsigned)rel;     // (beyond the descriptor: reads as zeros, handed on)
                        len_l = toolong ? SBWT_FUSED_MAXLEN + 1 : plen <= 0 ? 0 : plen > SBWT_FUSED_MAXLEN ? SBWT_FUSED_MAXLEN : (int)plen;
                        pool_len = len_l | (pc << 16);
                    }
                    const int len_e = len_l > SBWT_FUSED_MAXLEN ? SBWT_FUSED_MAXLEN : len_l;
                    i64 remain = total_bases - woff;
                    if (remain < 0) remain = 0;
                    if (remain > 0xFFFFFFF0ll) remain = 0xFFFFFFF0ll;
                    // Bounds-checked loads of ALIGNED dwords (the descriptor's base is the pool's first byte rounded down to
                    // four; a read's bytes are shifted into place afterwards): the last dword that holds a byte of `bases`
                    // is an aligned word of memory, so nothing beyond it is touched, and dwords past it read as 0.
                    const unsigned char *p0 = bases + woff;
                    const unsigned delta = (unsigned)((uintptr_t)p0 & 3u);
                    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                        const_cast<unsigned char *>(p0 - delta), (short)0, (int)(((unsigned)remain + delta + 3u) & ~3u), 0x00020000);
                    const unsigned vo = delta + vrel;
                    unsigned bad = 0;
#pragma unroll
                    for (int g = 0; g < SBWT_FUSED_MAXG; g++) {
                        if (g < G) {
                            const unsigned o = vo + 32u * (unsigned)g, oa = o & ~3u, sh = o & 3u;
                            const fz_u32x4 x0 = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)oa, 0, 0);
                            const fz_u32x4 x1 = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)oa + 16, 0, 0);
                            const unsigned x2 = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)oa + 32, 0, 0);
                            const unsigned raw[9] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w, x2};
                            unsigned w[8], tm[8];
#pragma unroll
                            for (int d = 0; d < 8; d++) {
                                w[d] = __builtin_amdgcn_alignbyte(raw[d + 1], raw[d], sh);
                                const int nb = len_e - 32 * g - 4 * d;                 // bytes of this dword inside the read
                                tm[d] = nb >= 4 ? 0xFFFFFFFFu : nb <= 0 ? 0u : ((1u << (8 * nb)) - 1u);
                            }
                            pool_codes[g][wbase + lane] = fz_encode32(w, tm, bad);
                        }
                    }
                    const bool isbad = (bad != 0 || toolong) && (i64)(pool_next + (u64)lane) < n_tickets;
                    pool_bad = __ballot(isbad);
                    // hand them on (rare): a read that is too long once (by its first piece's ticket); a read with other bytes
                    // by every piece that holds some -- the general kernel may then answer it more than once, with the same
                    // results, and its clean pieces are answered here as well (they do not depend on the rest)
                    const u64 defer_m = __ballot(isbad && (!toolong || pc == 0));
                    if (defer_m) {
                        unsigned long long at = 0;
                        if (lane == 0) at = atomicAdd(&ws->n_deferred, (unsigned long long)__popcll(defer_m));
                        at = uniform64(at);
                        if ((defer_m >> lane) & 1ull)
                            defer_list[at + (u64)__popcll(defer_m & low_mask(lane))] =
                                (unsigned)(P == 1 ? (pool_next + (u64)lane) : P == 2 ? ((pool_next + (u64)lane) >> 1) : (pool_next + (u64)lane) / 3ull);
                    }
                } else {
                    drained = true;                    // no read left anywhere: from now on idle lanes help busy ones (below)
                    if (SORT && lane == 0) (void)__hip_atomic_fetch_add(&q_ctl[5], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#ifdef SBWT_TIMELINE
                    tl_drain = wall_clock64();
#endif
                    pool_end = pool_next;
                }
                __builtin_amdgcn_wave_barrier();
            }
            const unsigned avail = (unsigned)(pool_end - pool_next);
            const unsigned n = (unsigned)__popcll(need);
            const unsigned rank = (unsigned)__popcll(need & low_mask(lane));
            // (ragged batches: the length of ticket q's read sits in lane q -- fetched with every lane of the wave active)
            const int qsrc = (int)((pool_next + rank) & 63ull);
            const int plq = varlen ? __shfl(pool_len, qsrc) : 0;
            const int mq_r = varlen ? (plq & 0xFFFF) - k + 1 : m;
            bool started = false;
            if (((need >> lane) & 1ull) && rank < avail) {
                const u64 tk = pool_next + rank;
                const int q = (int)(tk & 63ull);
                const int mq = mq_r;                   // (a read shorter than k: nothing to answer)
                if ((i64)tk < n_tickets && !((pool_bad >> q) & 1ull) && mq > 0) {
                    FZ_HIST_FLUSH();
                    rd = (unsigned)(P == 1 ? tk : P == 2 ? (tk >> 1) : tk / 3ull);
#pragma unroll
                    for (int g = 0; g < SBWT_FUSED_MAXG; g++)
                        if (g < G) cur_codes[g][slot] = pool_codes[g][wbase + q];
                    started = true;
                    i = 0;
                    mend = mq;
                    nseg = 0;
                    i0 = (plq >> 16) << 16;
                    b = -1;
                    blo = -1;
                    bnext = -1;
                    fl = CF_BUDGETS;
                    wstart = 0;
                    j = 0;
                    wk = (ps > 0) ? 1 : 0;
                    if (p > 0) mode = F_INIT;
                    else { mode = F_STEP; l = 0; r = last_node; }
                }                                      // else: past the last read, or handed on -- the lane stays idle
            }
            pool_next = uniform64(pool_next + ((n < avail) ? n : avail));
            if (SORT) {
                const u64 st_m = __ballot(started);
                if (st_m && lane == 0) (void)__hip_atomic_fetch_add(&q_ctl[4], (unsigned)__popcll(st_m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
#ifdef SBWT_TIMELINE
        if (!drained) tl_n++;
#endif
        if (drained && runS) {
            // ---- the tail of the batch: one lane per read means a wave waits for its slowest read.  An idle lane takes over
            //      the second half of what a busy lane still has to answer.  Exact: this kernel only walks reads of upper-case
            //      ACGT, where a k-mer's result does not depend on what came before it (streaming step and full search agree,
            //      tests/test_large.hh:104-115), so the taker starts with a full search at its first k-mer.
            // (SORT: among the searcher lanes of a wave -- a taker needs a slot; reads that are with the followers are not split)
            const u64 idle = __ballot(mode == F_IDLE && slot >= 0);
            const u64 busy_m = __ballot(mode != F_IDLE);
            if (!SORT && busy_m == 0 && fm_pend == 0) break;    // everything this wave took is answered and written
            // (round 5: a lower threshold in the thin end of the tail only -- at most 8 / 16 / 32 lanes busy, halves of 2 or 4
            // k-mers -- moves nothing either: config 2 4.91-4.92 vs 4.91 ms, 1 M reads 0.92-0.94 vs 0.93, config 5 +1 %)
            u64 donors = __ballot(mode != F_IDLE && mend - i >= FZ_SPLIT_MIN);
            if (idle && donors) {
                const int n_pairs = min(__popcll(idle), __popcll(donors));
                const int my_idle_rank = __popcll(idle & low_mask(lane)), my_donor_rank = __popcll(donors & low_mask(lane));
                const bool giving = ((donors >> lane) & 1ull) && my_donor_rank < n_pairs;
                const bool taking = ((idle >> lane) & 1ull) && my_idle_rank < n_pairs;
                // the lane number of the donor of rank q, for every taker
                int src = 0;
                {
                    u64 dm = donors;
                    for (int q = 0; q < my_idle_rank && taking; q++) dm &= dm - 1;
                    src = taking ? (__ffsll((i64)dm) - 1) : lane;
                }
                const int d_i = __shfl(i, src), d_end = __shfl(mend, src);
                const unsigned d_rd = (unsigned)__shfl((int)rd, src);
                const int d_pc = __shfl(i0, src) & 0x00FF0000;
                const int d_slot = SORT ? __shfl(slot, src) : wbase + src;
                const int mid = d_i + ((d_end - d_i + 1) >> 1);
                if (giving) mend = i + ((mend - i + 1) >> 1);                        // (the same mid its taker computed)
#ifdef SBWT_STATS
                if (lane == 0) atomicAdd(&ws->pad[13], (unsigned long long)n_pairs);
#endif
                if (taking) {
                    FZ_HIST_FLUSH();
#pragma unroll
                    for (int g = 0; g < SBWT_FUSED_MAXG; g++)
                        if (g < G) cur_codes[g][slot] = cur_codes[g][d_slot];
                    rd = d_rd;
                    i = mid;
                    mend = d_end;
                    nseg = 0;
                    i0 = mid | d_pc;
                    b = -1;
                    blo = -1;
                    bnext = -1;
                    fl = CF_BUDGETS;
                    wstart = mid;
                    j = 0;
                    wk = (ps > 0) ? 1 : 0;
                    if (p > 0) mode = F_INIT;
                    else { mode = F_STEP; l = 0; r = last_node; }
                }
                if (SORT && lane == 0) (void)__hip_atomic_fetch_add(&q_ctl[4], (unsigned)n_pairs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __builtin_amdgcn_wave_barrier();
            }
            if (SORT) {
                // entries wait in the searchers' ring and nobody holds a place for them: idle lanes give up their (empty) slots,
                // take places in the next iteration, and the reads that came back from the followers go on
                const int backlog = (int)(vctl[2] - vctl[0]);
                const u64 still = __ballot(mode == F_IDLE && slot >= 0);
                if (backlog > 0 && ((still >> lane) & 1ull) && __popcll(still & low_mask(lane)) < backlog) slot = -1;
            }
        }
        // SORT: the planner runs here, at the top of a searcher wave's iteration, for the reads whose last iteration (in this
        // wave or in a follower wave) asked for it
        if (SORT && runS && mode == F_PLAN) {
            force = (fl & CF_FORCE) != 0;
            fl &= ~CF_FORCE;
            plan_walk(force, w31);
        }


        // F_EXT and F_BRIDGE are the two ways into F_CMP from a k-mer that sits on its path:
        if (!WIDE) {
        } else if (mode == F_EXT) {                    // k-mer i-1 sits at path position r: on along the path
            co = r - (i + k - 1);
            cP = i + k - 1;
            m1 = -1;
            fl = (fl & ~(CF_M1 | CF_M2)) | CF_ONP;
            mode = F_CMP;
        } else if (mode == F_BRIDGE) {
            // ... and the read's base u = i+k-1 differs from the path's char at step r, a substitution-safe step: the k
            // k-mers that hold u are absent if nothing else in their windows differs
            const int u = i + k - 1;
            co = r - u;
            cP = u + 1;
            m1 = u;
            fl = (fl & ~(CF_M1 | CF_M2 | CF_ONP)) | CF_M1;
            mode = F_CMP;
        }

        // ---- this iteration's gather: two 16-byte loads per lane, issued back to back, one wait ----
        int ev = FE_NONE, tfail = 0, c = 0;
        const uint4 *a1 = ix.blocks, *a2 = ix.blocks;
        int res = -1;
        const bool trn = runP && (mode == F_TRANS), cmp = WIDE && (mode == F_CMP);
        const bool ext = runP && !WIDE && (mode == F_EXT), brg = runP && !WIDE && (mode == F_BRIDGE);
        // (SORT: a lane whose read is of the other class by now -- its hand-over waits for the writer, below -- does nothing)
        const bool own = !SORT || (isP ? (mode == F_EXT || mode == F_TRANS || mode == F_BRIDGE) : (mode == F_INIT || mode == F_STEP || mode == F_POS));
        const bool busy = own && (mode != F_IDLE && mode != F_DEAD);
        const int mode0 = mode;                        // (the state this iteration's gather is for)
        const int sli = SORT ? (slot < 0 ? tid : slot) : tid;      // this lane's slot in LDS (a lane without one reads its own number's: unused)
        bool rknown = false, ext_absent = false;
        int tnext = F_EXT;
        int tpos = -1;
        int seg_n = 0;
        unsigned seg_src_run = 0;
        // the read's bases from position P on: three words of codes out of the lane's LDS slot
        const int woff5 = (mode == F_INIT && wk == 5) ? ps : 0;       // the second-level window starts after the prefix
        const int P = (ext || trn) ? (i + k - 1) : brg ? (i + k) : cmp ? cP : ((mode == F_INIT) ? (wstart + woff5) : (wstart + j));
        const int s = P & 31, pg = busy ? (P >> 5) : 0;
        const u64 cw0 = cur_codes[pg < SBWT_FUSED_MAXG ? pg : SBWT_FUSED_MAXG - 1][sli];
        const u64 cw1 = cur_codes[pg + 1 < SBWT_FUSED_MAXG ? pg + 1 : SBWT_FUSED_MAXG - 1][sli];
        const u64 cw2 = cur_codes[pg + 2 < SBWT_FUSED_MAXG ? pg + 2 : SBWT_FUSED_MAXG - 1][sli];
        const u64 rw = s ? ((cw0 >> (2 * s)) | (cw1 << (64 - 2 * s))) : cw0;      // bases P .. P+31
        if (runS && busy && mode == F_POS) {
            a1 = reinterpret_cast<const uint4 *>(ix.pos + ((unsigned)l & ~3u));  // the aligned 16 bytes holding pos[l]
            a2 = a1 + ((((unsigned)l & 3u) == 3u && (fl & CF_SEED2)) ? 1 : 0);   // (a seed of two columns: pos[l + 1] as well)
        } else if (busy) {
            const int wl = (wk == 1) ? ps : (wk == 2 || wk == 6) ? L0 : (wk == 3) ? pw : (wk == 5) ? k - ps : p;
            c = (int)((unsigned)rw & 3u);
            if (cmp) {
                a1 = ix.pq + ((unsigned)(co + cP) >> 5);               // the two quads holding path steps co + cP .. (33 to 64 of them)
                a2 = a1 + 1;
            } else if (ext || brg) {
                a1 = ix.pq + (((unsigned)r + (brg ? 1u : 0u)) >> 5);   // the two quads holding path chars r (+1) .. +31
                a2 = a1 + 1;
            } else if (trn) {
                // the entry of (position r, char c); j counts the slots probed
                a1 = ix.trans + 2 * (size_t)sbwt_trans_slot((unsigned)r, (unsigned)c, ix.n_tslots, (unsigned)j);
                a2 = a1 + 1;
            } else if (!runS) {
                // (SORT, a follower wave: no other state)
            } else if (mode == F_INIT) {
                (void)wl;
                if (wk == 1) {                 // bucket (hash + j) of the sparse table: two entries
                    const u64 key = rw & low_mask(2 * ps);
                    hk = key;
                    const size_t bkt = sbwt_sp_bucket(key, ix.n_sb, (unsigned)j);
                    a1 = ix.stab + 2 * bkt;
                    a2 = a1 + 1;
                } else if (wk == 5) {          // second level: (prefix interval, rest of the k-mer) -> one entry
                    hk = rw & mk2;
                    const size_t bkt = sbwt_sp2_entry((unsigned)l, hk, ix.n_sb2, (unsigned)j);
                    a1 = ix.stab2 + 2 * bkt;
                    a2 = a1 + 1;
                } else if (wk == 2 || wk == 6 || (wk == 3 && pfon)) {   // the window's block of the probe filter
                    const u64 h = sbwt_pf_hash(rw & low_mask(2 * L0));
                    hk = (u64)sbwt_pf_bits(h);
                    a1 = ix.pfil + (h >> (64 - ix.log2f));
                    a2 = a1;
                } else {
                    a1 = reinterpret_cast<const uint4 *>(ix.ptab + (rw & low_mask(2 * p)));
                    a2 = a1;
                }
            } else {   // F_STEP
                a1 = ix.blocks + (((zx(l) >> 6) << 2) + c);
                a2 = ix.blocks + ((((zx(r) + 1) >> 6) << 2) + c);
            }
        }
#ifdef SBWT_STATS
        {   // lane-iterations by kind: pad[0..]: sparse lookup, filter probe, dense table, second level, interval update, path run,
            // transition, bridge, pos, idle/dead; pad[10] = wave-iterations
            const int cls = !busy ? 9 : mode == F_INIT ? (wk == 1 ? 0 : (wk == 2 || wk == 6 || (wk == 3 && pfon)) ? 1 : wk == 5 ? 3 : 2) :
                            mode == F_STEP ? 4 : (ext || (cmp && m1 < 0)) ? 5 : trn ? 6 : (cmp || brg) ? 7 : 8;
            for (int q = 0; q < 10; q++) {
                const unsigned long long cq = __popcll(__ballot(cls == q));
                if (lane == 0 && cq) atomicAdd(&ws->pad[q], cq);
            }
            if (lane == 0) atomicAdd(&ws->pad[10], 1ull);
            it_cnt += busy ? 1u : 0u;
            // the tail: wave-iterations and idle lane-iterations after the tickets ran out; k-mers still open then
            if (drained) {
                {       // the tail by iteration since the tickets ran out: waves still running, their busy lanes, donors
                    const unsigned q = tail_it < 63u ? tail_it : 63u;
                    const unsigned long long nb = __popcll(__ballot(busy)), nd = __popcll(__ballot(busy && mend - i >= FZ_SPLIT_MIN));
                    if (lane == 0) { atomicAdd(&g_tail_prof[q], 1ull); atomicAdd(&g_tail_prof[64 + q], nb); atomicAdd(&g_tail_prof[128 + q], nd); }
                }
                tail_it++;
                const unsigned long long idl = __popcll(__ballot(!busy));
                const unsigned long long dn = __popcll(__ballot(busy && mend - i >= FZ_SPLIT_MIN));
                if (lane == 0) { atomicAdd(&ws->pad[11], 1ull); atomicAdd(&ws->pad[12], idl); atomicAdd(&ws->pad[14], dn); }
            }
        }
#endif
        c_search = uniform32(c_search + (unsigned)__popcll(__ballot(mode == F_INIT || (p == 0 && mode == F_STEP && j == 0))));
        c_lf = uniform32(c_lf + (unsigned)__popcll(__ballot(mode == F_STEP)));

        // ---- the writer: the lists handed over at the end of the last iteration (fm_pend; pend, pend_rd).  Here, behind the
        //      issue of this iteration's gather and ahead of its use: the col[] loads share the gather's round trip, and the
        //      stores have the whole iteration to be acknowledged before the next wait (loads and stores share vmcnt on gfx9:
        //      a wait for a load issued after a store waits for that store).  Up to four reads per trip, the col[] loads of all
        //      four in flight before the first store ----
        constexpr int FP = 4;
        int fL[FP], fe[FP], fa[FP], w0[FP], w1[FP];           // (fL, fe, fa, fob: wave-uniform)
        i64 fob[FP];
        const unsigned col_minus1 = (unsigned)ix.n_pos;       // col[n_pos] = 0xFFFFFFFF: the "position" of a -1 (sbwtgpu_index_create)
        // the source and first k-mer of the segments that hold results j0 and j0 + 1 of lane L's list of ns segments
        auto seg_of = [&](int tl, int ns, int j0, unsigned &c0s, int &c0a, unsigned &c1s, int &c1a) {
            // (lane t holds the start of segment t: the starts ascend, so the segment of j0 is a count; j0 + 1 is in it or the next)
            const int my_at = (int)seg_at[lane < FZ_NSEG ? lane : FZ_NSEG - 1][tl];
            int idx = 0;
            for (int t = 1; t < ns; t++) idx += (__builtin_amdgcn_readlane(my_at, t) <= j0) ? 1 : 0;
            c0s = seg_src[idx][tl];
            c0a = (int)seg_at[idx][tl];
            const int nxi = idx + 1 < FZ_NSEG ? idx + 1 : FZ_NSEG - 1;
            const unsigned nxs = seg_src[nxi][tl];
            const int nxa = (int)seg_at[nxi][tl];
            const bool usenx = (idx + 1 < ns && nxa <= j0 + 1);
            c1s = usenx ? nxs : c0s;
            c1a = usenx ? nxa : c0a;
        };
        // first half of a trip: up to FP lists, the col[] loads of their first 128 results.  A list that holds a result known
        // by its column only (bit 22 of pend: walks through the blocks -- images without whole k-mers in the tables) and what
        // lies beyond 128 results is left to the one-read-at-a-time pass of the second half (returns whether there is any)
        auto flush_issue = [&](u64 &fm) -> bool {
            bool slow = false;
#pragma unroll
            for (int u = 0; u < FP; u++) {
                fL[u] = -1; fe[u] = 0; fa[u] = 0; fob[u] = 0; w0[u] = w1[u] = 0;
                if (fm == 0) continue;                 // wave-uniform: unused slots cost nothing
                const int L = __ffsll((i64)fm) - 1;    // (wave-uniform: the owner's registers are read with v_readlane)
                fm &= fm - 1;
                const int pdL = __builtin_amdgcn_readlane(pend, L);
                const int ns = (pdL >> 16) & 15, a = pdL & 0xFF, e = (pdL >> 8) & 0xFF;
                const bool direct = (pdL >> 22) & 1;
                fL[u] = L;
                fa[u] = a;
                {
                    const i64 rdu = (i64)(unsigned)__builtin_amdgcn_readlane((int)pend_rd, L);
                    fob[u] = (ragged ? out_off[rdu] : u_out0 + rdu * u_stride) + (UNI ? 0 : (i64)(((pdL >> 20) & 3) * kpp));
                }
                slow = slow || direct || (e - a > 128);
                if (direct) continue;                  // (fe = 0: the first half writes nothing of it)
                fe[u] = e;
                const int j0 = a + 2 * lane, j1 = j0 + 1;
                unsigned c0s, c1s;
                int c0a, c1a;
                seg_of(SORT ? (int)((unsigned)pdL >> 24) : wbase + L, ns, j0, c0s, c0a, c1s, c1a);
                const unsigned p0 = (j0 < e && c0s != 0xFFFFFFFFu) ? c0s + (unsigned)(j0 - c0a) : col_minus1;
                const unsigned p1 = (j1 < e && c1s != 0xFFFFFFFFu) ? c1s + (unsigned)(j1 - c1a) : col_minus1;
                w0[u] = (int)ix.col[p0];
                w1[u] = (int)ix.col[p1];
            }
            return slow;
        };
        auto flush_store = [&](const bool slow) {
            // Every col[] value has arrived BEFORE the first store is issued: on gfx9 loads and stores share vmcnt, and with a
            // store in flight the compiler's wait for a load is vmcnt(0) -- a wait for that store's acknowledgement.  With the
            // stores of a trip between the loads' uses, each store waited for the one before it (round 5: four exposed write
            // latencies per trip, a quarter of the kernel's time).
#pragma unroll
            for (int u = 0; u < FP; u++) asm volatile("" : "+v"(w0[u]), "+v"(w1[u]));
            if (!(ix.debug & 1)) {
#pragma unroll
                for (int u = 0; u < FP; u++) {
                    const int fj = fa[u] + 2 * lane;
                    // (col[] values: int32 columns, -1 from the sentinel; BIG: uint32 columns, 0xFFFFFFFF is the -1)
                    const i64 r0 = BIG ? (w0[u] == -1 ? -1ll : (i64)(unsigned)w0[u]) : (i64)w0[u];
                    const i64 r1 = BIG ? (w1[u] == -1 ? -1ll : (i64)(unsigned)w1[u]) : (i64)w1[u];
                    if (fj + 1 < fe[u]) st_res2(out, fob[u] + fj, r0, r1, O32);
                    else if (fj < fe[u]) st_res(out, fob[u] + fj, r0, O32);
                }
            }
            if (slow) {
                // (rare) lists with a result known by its column only, from their first result; more than 128 results since the
                // last flush (reads of up to 160 bases), from the 129th: one read, 128 results at a time
#pragma unroll
                for (int u = 0; u < FP; u++) {
                    if (fL[u] < 0) continue;
                    const int L = fL[u], pdL = __builtin_amdgcn_readlane(pend, L);
                    const int ns = (pdL >> 16) & 15, a = pdL & 0xFF, e = (pdL >> 8) & 0xFF;
                    for (int base = ((pdL >> 22) & 1) ? a : a + 128; base < e; base += 128) {
                        const int j0 = base + 2 * lane, j1 = j0 + 1;
                        unsigned c0s, c1s;
                        int c0a, c1a;
                        seg_of(SORT ? (int)((unsigned)pdL >> 24) : wbase + L, ns, j0, c0s, c0a, c1s, c1a);
                        const unsigned p0 = (j0 < e && is_run(c0s)) ? c0s + (unsigned)(j0 - c0a) : col_minus1;
                        const unsigned p1 = (j1 < e && is_run(c1s)) ? c1s + (unsigned)(j1 - c1a) : col_minus1;
                        const int y0 = (int)ix.col[p0], y1 = (int)ix.col[p1];
                        const int q0 = is_run(c0s) ? y0 : ((c0s == 0xFFFFFFFFu) ? -1 : (int)(c0s & 0x7FFFFFFFu));
                        const int q1 = is_run(c1s) ? y1 : ((c1s == 0xFFFFFFFFu) ? -1 : (int)(c1s & 0x7FFFFFFFu));
                        const i64 r0 = BIG ? (q0 == -1 ? -1ll : (i64)(unsigned)q0) : (i64)q0;
                        const i64 r1 = BIG ? (q1 == -1 ? -1ll : (i64)(unsigned)q1) : (i64)q1;
                        if (!(ix.debug & 1)) {
                            if (j1 < e) st_res2(out, fob[u] + j0, r0, r1, O32);
                            else if (j0 < e) st_res(out, fob[u] + j0, r0, O32);
                        }
                    }
                }
            }
        };
        // (the gather's two loads are issued BEHIND the first trip's col[] loads: one wait covers both, and the gathered quads
        // are not live while the writer computes its addresses)
        uint4 v1, v2;
        if (fm_pend) {
            u64 fm = fm_pend;
            bool lr = flush_issue(fm);
            v1 = *a1;
            v2 = *a2;
            // (measured, round 5: also holding the stores back until the gathered quads have arrived -- so that the wait for
            // them is not a wait for the stores -- is 1.5 % slower: the stores start a gather's latency later)
            flush_store(lr);
            while (fm) {
                lr = flush_issue(fm);
                flush_store(lr);
            }
            if (SORT) {
                // the lists are written (their stores are issued): the last list of a read ends it -- the workgroup counts its
                // reads in flight -- and a follower lane's slot goes back to the searchers as a free one
                const u64 fin = fm_pend & __ballot((pend >> 23) & 1);
                if (fin) {
                    if (lane == 0) (void)__hip_atomic_fetch_sub(&q_ctl[4], (unsigned)__popcll(fin), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (isP) {
                        const int ps_ = (int)((unsigned)pend >> 24);
                        if ((fin >> lane) & 1ull) st_w[SORT ? 2 : 0][SORT ? ps_ : 0] = 0u;       // kind 0: a free slot
                        ring_push(fin, 0, ps_);
                    }
                }
            }
        } else {
            v1 = *a1;
            v2 = *a2;
        }

        // ---- consume ----
#ifdef SBWT_STATS
        unsigned why = 0;
#define WHY(q) (why |= 1u << (q))
#else
#define WHY(q) ((void)0)
#endif
        bool tabhit = false, do_plan = false;
        int m2 = -1;                                   // F_CMP: the second difference (this iteration's)
        int seed_col = -1;                             // k > 31: the column of a unique 31-mer whose k-mer was not there: a SEED for an alignment
        bool seed_is_pos = false;                      // ... seed_col is that column's path position already (the table entry carried it)
        bool imprecise = false;                        // this iteration's failure is a table-level miss
        int pre_n = 0, abs_n = 0, post_n = 0;          // F_CMP's verdicts: a run along the path, k-mers absent by a safe bit, a run
        int cact = 0, ctr = 0;                         // ... and what follows (CA_*)
        int burst_to = -1;                             // F_BRIDGE: k-mers i .. burst_to are certified absent
        bool bridged = false;                          // F_BRIDGE: ... and the read goes on along the path
        if (runS && busy && mode0 == F_POS) {
            const unsigned sel = (unsigned)l & 3u;
            r = (int)(sel == 0 ? v1.x : sel == 1 ? v1.y : sel == 2 ? v1.z : v1.w);
            if (fl & CF_SEED) {
                // a seed: the k-mer at position r ENDS with the 31 bases before read position -co; align the rest of the read
                fl &= ~(CF_SEED | CF_M1 | CF_M2 | CF_ONP);
                if (fl & CF_SEED2)                     // (the second column's position waits in l)
                    l = (int)(sel == 0 ? v1.y : sel == 1 ? v1.z : sel == 2 ? v1.w : v2.x);
                co += r;
                if ((unsigned)(co + i) < (unsigned)ix.n_pos) { cP = i; m1 = -1; mode = F_CMP; }
                else if (fl & CF_SEED2) {              // (the first column's alignment does not fit: the second at once)
                    co += l - r;
                    fl &= ~CF_SEED2;
                    if ((unsigned)(co + i) < (unsigned)ix.n_pos) { cP = i; m1 = -1; mode = F_CMP; }
                    else { mode = F_DEAD; do_plan = true; }
                }
                else { mode = F_DEAD; do_plan = true; }
            } else {
                mode = F_EXT;
            }
        } else if (trn) {
            // v1 = { r + 1, c | flags, successor column (SBWT.hh:562-575), its path position }, v2 = its path's next 32 steps
            if (v1.x == 0u) {
                WHY(5);
                ev = FE_EMIT1;                         // a free slot: (r, c) has no entry -- a path's last column without a
                b = blo = i + k - 1;                   // successor by c: -1
            } else if (v1.x != (unsigned)r + 1u || (v1.y & 3u) != (unsigned)c) {
                j++;                                   // another entry's slot: the next one (linear probing)
                if (j > 4096) { ws->status = SBWT_ERR_NOT_SINGLETON; ev = FE_EMIT1; b = blo = i + k - 1; }   // damaged image
            } else if (v1.y & SBWT_TRANS_NEG) {
                // no successor by this char at a step that has others; the entry says whether the step is safe for it
                if (ix.has_safe && (v1.y & SBWT_TRANS_NEG_SAFE)) {
                    WHY(6);
                    mode = F_BRIDGE;
                } else {
                    WHY(7);
                    ev = FE_EMIT1;
                    b = blo = i + k - 1;
                    co = r - (i + k - 1);              // (the read was on this path up to here: an alignment to resume on)
                    fl |= CF_ALIGNED;
                }
            } else {
                WHY(8);
                ev = FE_EMIT1;
                res = (int)v1.z;
                r = (int)v1.w;
                emit_pos = v1.w;
                rknown = true;
                // the read's next bases against the steps quoted in the entry: most runs after a transition end here
                const u64 rq = rw >> 2;                // bases P+1 .. P+31
                const u64 x = (rq ^ quad_bits(v2)) & low_mask(62);
                const u64 mm = (x | (x >> 1)) & 0x5555555555555555ull;
                const int nm = mm ? ((__ffsll((i64)mm) - 1) >> 1) : 31;
                const int nv = __ffs((int)((v2.z & ~v2.w) | 0x80000000u)) - 1;      // the quoted path ends: A & ~B
                int n2 = nm < nv ? nm : nv;
                bool stop2 = n2 < 31;
                if (n2 >= mend - 1 - i) { n2 = mend - 1 - i; stop2 = false; }
                if (n2 < 0) n2 = 0;
                seg_n = n2;
                seg_src_run = (unsigned)r + 1u;
                r += n2;
                c_ext += (unsigned)n2;
                if (stop2) {
                    const int kind = path_stop_kind(nm < nv, (v2.z >> n2) & 1u, (v2.w >> n2) & 1u, ix.has_safe != 0);
                    WHY(kind == PS_ABSENT ? 9 : kind == PS_TRANS ? 10 : 11);
                    if (kind == PS_ABSENT) ext_absent = true;
                    else tnext = (kind == PS_TRANS) ? F_TRANS : F_BRIDGE;
                }
            }
        } else if (brg) {
            // k-mer i ends at the mismatching base: if the bases after it agree with the path again, every k-mer that
            // contains the mismatching base is a one-base variant of a path k-mer, absent by the safe bit
            const int sp = (int)(((unsigned)r + 1u) & 31u);
            u64 pwd = quad_bits(v1) >> (2 * sp);
            if (sp) pwd |= quad_bits(v2) << (64 - 2 * sp);
            const u64 x = rw ^ pwd;
            const u64 mm = (x | (x >> 1)) & 0x5555555555555555ull;
            const int nm = mm ? ((__ffsll((i64)mm) - 1) >> 1) : 32;
            const int needb = (k - 1 < mend - 1 - i) ? (k - 1) : (mend - 1 - i);
            if (nm >= needb) {
                ev = FE_FAIL;
                WHY(needb == k - 1 ? 12 : 13);
                burst_to = i + needb;
                c_brg++;
                bridged = needb == k - 1;              // back on the path: on with F_EXT from position r + k, no walk
            } else {
                WHY(14);
                ev = FE_EMIT1;                         // no bridge: a bridgeable step has no successor by the read's char
                b = blo = i + k - 1;
                bnext = i + k + nm;                    // ... and the compare has seen where the read differs next
            }
        } else if (ext) {
            // k-mer i-1 sits at path position r.  Read bases i+k-1.. against path chars r..: while they agree (and the
            // path goes on), k-mer i+x sits at r+1+x.
            const int sp = (int)((unsigned)r & 31u);
            u64 pwd = quad_bits(v1) >> (2 * sp);
            if (sp) pwd |= quad_bits(v2) << (64 - 2 * sp);
            // the path groups' two state words (k_path_reencode): go = ~A | B, safe = A & B, only successor = ~A & B
            const u64 fA = (((u64)v2.z << 32) | (u64)v1.z) >> sp, fB = (((u64)v2.w << 32) | (u64)v1.w) >> sp;
            const u64 pgo = ~fA | fB;
            const u64 x = rw ^ pwd;
            const u64 mm = (x | (x >> 1)) & 0x5555555555555555ull;
            int nm = mm ? ((__ffsll((i64)mm) - 1) >> 1) : 32;
            int nv = __ffsll((i64)(~pgo | (1ull << 32))) - 1;
            int n = nm < nv ? nm : nv;
            bool stopped = n < 32;                     // a mismatch or the end of the path
            if (!stopped) {
                // the read's side is whole (LDS); the two path quads hold 32 - sp more steps
                const int w2 = 32 - sp;
                const u64 rw2 = s ? ((cw1 >> (2 * s)) | (cw2 << (64 - 2 * s))) : cw1;
                const u64 x2 = rw2 ^ (quad_bits(v2) >> (2 * sp));
                const u64 mm2 = (x2 | (x2 >> 1)) & 0x5555555555555555ull;
                const int nm2 = mm2 ? ((__ffsll((i64)mm2) - 1) >> 1) : 32;
                const int nv2 = __ffsll((i64)(~(pgo >> 32) | (1ull << 32))) - 1;
                int n2 = nm2 < nv2 ? nm2 : nv2;
                if (n2 >= w2) n2 = w2;                 // the end of what is loaded is not a stop
                else stopped = true;
                n = 32 + n2;
                nm = 32 + nm2;
                nv = 32 + nv2;
            }
            if (n >= mend - i) { n = mend - i; stopped = false; }
            seg_n = n;
            seg_src_run = (unsigned)r + 1u;
            c_ext += (unsigned)n;
            if (i + n == mend) {
                mode = F_IDLE;
            } else if (stopped) {
                int kind = path_stop_kind(nm < nv, (unsigned)(fA >> n) & 1u, (unsigned)(fB >> n) & 1u, ix.has_safe != 0);
                WHY(0);
                WHY(kind == PS_TRANS ? 1 : kind == PS_BRIDGE ? 2 : 4);
                if (kind == PS_TRANS && nm < nv) WHY(15);      // (a mismatch at a step with other successors, not the path's end)
                if (kind == PS_BRIDGE && n < 32) {
                    // a bridge needs the next k-1 bases to agree with the path; a second difference already in this window:
                    // skip the attempt (the step has no successor by the read's char either way)
                    const int after = 31 - n, want = (k - 1 < mend - 1 - (i + n)) ? (k - 1) : (mend - 1 - (i + n));
                    const int chk = after < want ? after : want;
                    const u64 m2nd = chk > 0 ? ((mm >> (2 * (n + 1))) & low_mask(2 * chk)) : 0ull;
                    if (m2nd) {
                        WHY(3);
                        kind = PS_ABSENT;
                        bnext = (i + k - 1) + n + 1 + ((__ffsll((i64)m2nd) - 1) >> 1);     // the second difference
                    }
                }
                if (kind == PS_ABSENT) ext_absent = true;
                else { mode = (kind == PS_TRANS) ? F_TRANS : F_BRIDGE; j = 0; }
            }
            r += n;
        } else if (cmp) {
            // The read against the path it is aligned to: base j <-> step co + j.  Bases [.., cP) are compared; of the
            // differences not behind k-mer i yet the first two are remembered (m1 < m2).  k-mer x is the path's own (position
            // co + x + k) when the k steps of its window lie on one path and hold no difference, and absent when they hold
            // exactly one, m1, at a substitution-safe step (k_path_safe_labels*).  Whatever else -- two differences in one
            // window, an unsafe step, the path's end -- leaves this state: a transition where k-mer x-1 is the path's own and
            // x ends at the difference (SBWT.hh:562-575 at a branching step), the certificates with b = m1 known exactly
            // otherwise.
            const int cE = mend + k - 1, last = mend - 1;
            const unsigned pp = (unsigned)(co + cP);
            const int sp = (int)(pp & 31u);
            u64 pwd = quad_bits(v1) >> (2 * sp);
            if (sp) pwd |= quad_bits(v2) << (64 - 2 * sp);
            // the path groups' two state words (k_path_reencode): go = ~A | B, safe = A & B, only successor = ~A & B
            const u64 fA = (((u64)v2.z << 32) | (u64)v1.z) >> sp, fB = (((u64)v2.w << 32) | (u64)v1.w) >> sp;
            int n = cE - cP;
            if (n > 64 - sp) n = 64 - sp;
            if (n < 0) n = 0;
            int pend = -1;                             // the path ends at the step of this base
            {
                const u64 ends = fA & ~fB;
                if (ends) {
                    const int nb = __ffsll((i64)ends) - 1;
                    if (nb < n) { n = nb; pend = cP + nb; }
                }
            }
            const u64 rw2 = s ? ((cw1 >> (2 * s)) | (cw2 << (64 - 2 * s))) : cw1;       // bases cP+32 ..
            const u64 x1 = rw ^ pwd, x2 = rw2 ^ (quad_bits(v2) >> (2 * sp));
            u64 d1 = (x1 | (x1 >> 1)) & 0x5555555555555555ull, d2 = (x2 | (x2 >> 1)) & 0x5555555555555555ull;
            if (n < 32) { d1 &= low_mask(2 * n); d2 = 0; }
            else if (n < 64) d2 &= low_mask(2 * (n - 32));
            int c1 = -1, c2 = -1;                      // the first two differences of this window
            if (d1) {
                c1 = (__ffsll((i64)d1) - 1) >> 1;
                d1 &= d1 - 1;
                if (d1) c2 = (__ffsll((i64)d1) - 1) >> 1;
            }
            if (c2 < 0 && d2) {
                const int t0 = 32 + ((__ffsll((i64)d2) - 1) >> 1);
                d2 &= d2 - 1;
                if (c1 < 0) { c1 = t0; if (d2) c2 = 32 + ((__ffsll((i64)d2) - 1) >> 1); }
                else c2 = t0;
            }
            auto state_of = [&](int cc) -> unsigned {  // { S, A, B } of the step of this window's base cc, as CF_*1 bits
                const unsigned A = (unsigned)(fA >> cc) & 1u, B = (unsigned)(fB >> cc) & 1u;
                return ((A & B & (ix.has_safe ? 1u : 0u)) ? CF_S1 : 0u) | (A ? CF_A1 : 0u) | (B ? CF_B1 : 0u);
            };
            if (c1 >= 0) {
                if (m1 < 0) {
                    m1 = cP + c1;
                    fl = (fl & ~CF_M1) | state_of(c1);
                    if (c2 >= 0) { m2 = cP + c2; fl = (fl & ~CF_M2) | (state_of(c2) << 3); }
                } else {
                    m2 = cP + c1;
                    fl = (fl & ~CF_M2) | (state_of(c1) << 3);
                }
            }
            // (a second difference: what lies behind it in this window is compared again once m1's k-mers are answered)
            if (m2 >= 0) { cP = m2 + 1; pend = -1; }
            else cP += n;
            // ---- verdicts: at most a run and a burst per iteration ----
            int x = i;
            bool onp = (fl & CF_ONP) != 0;
            auto clean_run = [&](int bound, int &cnt) {    // k-mers x .. bound hold no difference
                if (bound > last) bound = last;
                if (bound >= x) { cnt = bound - x + 1; x = bound + 1; onp = true; }
            };
            if (m1 < 0) {
                clean_run(cP - k, pre_n);
                if (x > last) cact = CA_DONE;
                else if (pend >= 0) { if (onp && x == pend - k + 1) { cact = CA_TRANS; ctr = co + pend; } else cact = CA_LOST; }
            } else {
                clean_run(m1 - k, pre_n);
                if (x > last) {
                    cact = CA_DONE;
                } else if (fl & CF_S1) {
                    const int lim = m2 >= 0 ? m2 : cP;
                    int hi = m1 < lim - k ? m1 : lim - k;
                    if (hi > last) hi = last;
                    if (hi >= x) { abs_n = hi - x + 1; x = hi + 1; onp = false; }
                    if (x > last) {
                        cact = CA_DONE;
                    } else if (x > m1) {               // every k-mer that holds m1 is answered: the next difference takes its place
                        m1 = m2;
                        m2 = -1;
                        fl = (fl & ~(CF_M1 | CF_M2)) | ((fl & CF_M2) >> 3);
                        if (pre_n == 0) {
                            clean_run((m1 >= 0 ? m1 : cP) - k, post_n);
                            if (x > last) cact = CA_DONE;
                        }
                    } else if (m2 >= 0) {
                        cact = CA_CERT;                // k-mer x holds both differences
                    } else if (pend >= 0) {
                        cact = CA_CERT;                // the path ends inside the windows that hold m1
                        fl &= ~CF_ALIGNED;
                    }
                } else if (onp && x == m1 - k + 1) {
                    // k-mer x-1 is the path's own and the read's next base differs from the path's char: where F_EXT used to stop
                    if (!(fl & CF_B1)) { cact = CA_TRANS; ctr = co + m1; }      // the step has other successors (or none)
                    else cact = CA_ABSENT;             // only successor: the streaming step's answer is -1 (SBWT.hh:572-575)
                } else {
                    cact = CA_CERT;
                }
            }
            fl = onp ? (fl | CF_ONP) : (fl & ~CF_ONP);
            if (cact == CA_ABSENT) ext_absent = true;
        } else if (runS && mode0 == F_INIT) {
            int wl = p;
            bool again = false;
            const bool viaf = (wk == 2) || (wk == 6) || (wk == 3 && pfon);
            if (viaf) {
                const unsigned b1 = (unsigned)hk & 127u, b2 = ((unsigned)hk >> 7) & 127u;
                const unsigned w1 = (b1 < 64) ? (b1 < 32 ? v1.x : v1.y) : (b1 < 96 ? v1.z : v1.w);
                const unsigned w2 = (b2 < 64) ? (b2 < 32 ? v1.x : v1.y) : (b2 < 96 ? v1.z : v1.w);
                wl = L0;
                if (((w1 >> (b1 & 31u)) & (w2 >> (b2 & 31u)) & 1u) != 0) {
                    fl |= CF_NOCERT;
                    if (wk == 3 || wk == 6) {
                        l = 0;                         // range probe / hinted probe: "perhaps present" only moves the guess
                    } else {
                        // perhaps present (in a pan-genome usually truly: the window is another strain's variant): no
                        // certificate from this window -- the reference's own search of k-mer i decides
                        again = true;
                        do_plan = true;
                        force = true;
                        w31 = seed_ok;
                    }
                } else {
                    l = -1;                            // read[wstart .. wstart+L0-1] is not in the index
                }
            } else if (wk == 5) {
                wl = k;                                // a hit completes the k-mer; a miss: read[wstart .. wstart+k-1] is absent
                const bool hit0 = (v1.w & SBWT_SP2_USED) && quad_bits(v1) == hk && (v1.z & ~SBWT_SP2_OVERFLOW) == (unsigned)l;
                const bool hit1 = (v2.w & SBWT_SP2_USED) && quad_bits(v2) == hk && v2.z == (unsigned)l;
                if (hit0 | hit1) {
                    tpos = (int)((hit0 ? v1.w : v2.w) & ~SBWT_SP2_USED);     // the k-mer's path position: its column is col[tpos]
                    l = tpos;                          // (stands in for the column: the result is emitted by position)
                    r = l;
                } else if (v1.z & SBWT_SP2_OVERFLOW) {
                    again = true;
                    j++;
                } else {
#ifdef SBWT_STATS
                    why |= (r < 0 || r == l) ? (1u << 16) : (r == l + 1) ? (1u << 17) : (r == l + 2) ? (1u << 18) : (1u << 19);
#endif
                    if (seed_ok && (r < 0 || r == l)) {    // the 31-mer is there, in ONE column: a seed for an alignment
                        seed_is_pos = r < 0;               // ... whose path position came with the entry
                        seed_col = r < 0 ? -1 - r : l;
                    } else if (seed_ok && r == l + 1) {    // ... or in two (the stretch two strains share): the read is one of them
                        seed_col = -2 - l;                 // (two columns, l and l + 1: encoded below -1)
                    }
                    l = -1;
                }
            } else if (wk == 1) {
                const u64 key = hk;
                const u64 w0 = quad_bits(v1), w1 = quad_bits(v2);
                const bool hit0 = (w0 & ~SBWT_SP_OVERFLOW) == key, hit1 = w1 == key;
                wl = ps;
                if ((hit0 | hit1) && (fl & CF_WIN31)) {
                    again = true;                      // the 31-base window is in the index: no certificate, k-mer i's own search
                    do_plan = true;
                    force = true;
                    fl = (fl & ~CF_WIN31) | CF_NOCERT;
                } else if (hit0 | hit1) {
                    l = (int)(hit0 ? v1.z : v2.z);
                    if (ix.stab_pos) {                 // depth-k entries: one column, stored with its path position
                        r = l;
                        tpos = (int)(hit0 ? v1.w : v2.w);
                    } else {
                        // (a prefix with one column carries that column's path position: kept in r, negative, across the
                        // second-level lookup -- the interval's end is not needed on that route)
                        const unsigned wv = hit0 ? v1.w : v2.w;
                        r = !(wv & SBWT_SP_UNIQ) ? l + (int)wv : (WIDE && ix.stab2 && ps < k) ? -1 - (int)(wv & ~SBWT_SP_UNIQ) : l;
                    }
                } else if (w0 & SBWT_SP_OVERFLOW) {
                    again = true;                      // a later bucket may hold the key
                    j++;
                } else {
                    l = -1;                            // read[wstart .. wstart+ps-1] is not in the index
                }
            } else {
                l = (int)(i64)quad_bits(v1);
                r = (int)(i64)((u64)v1.z | ((u64)v1.w << 32));
            }
#ifdef SBWT_STATS
            if (viaf) oc_kind = (wk == 2 ? 10 : wk == 3 ? 12 : 14) + ((again || l != -1) ? 1 : 0);
            else if (wk == 1 && again && !do_plan) oc_kind = 18;
            else if (wk == 1) oc_kind = (wstart == i ? 16 : 19) + ((l != -1) ? 0 : 1);
            if (wk == 1 && !again && i == 0 && wstart == 0) oc_kind = 21 + ((l != -1) ? 0 : 1);
#endif
            if (!again) {
                tabhit = (l != -1);
                if (l == -1) {
                    ev = FE_FAIL;                      // read[wstart .. wstart+wl-1] is not in the index
                    tfail = wstart + wl - 1;
                    imprecise = (wk != 2 && wk != 6 && !(fl & CF_WIN31));  // ... but where inside the window it fails is not known
                } else if (wk == 3 || wk == 6) {
                    ev = FE_PRES;
                } else if (wk == 1 && ps < k && ix.stab2 && (fl & CF_ANCH) && seed_ok && (r < 0 || r == l) && CF_CMP_LEFT(fl) > 0) {
                    seed_is_pos = r < 0;
                    seed_col = r < 0 ? -1 - r : l;     // an anchor's 31-mer in ONE column: that is a seed already -- its position
                    ev = FE_FAIL;                      // aligns the read without the k-mer's other 32 bases having to be clean
                } else if (wk == 1 && ps < k && ix.stab2) {
                    wk = 5;                            // the prefix is there (l = its first column): the rest in one more gather
                    j = 0;
                } else {
                    j = wl;
                    if (fl & CF_ANCH) ev = FE_ANCH;    // (only whole-k-mer lookups are anchors: wl == k)
                    else if (wstart + j == i + k) ev = FE_END;
                    else mode = F_STEP;
                }
            }
        } else if (runS && mode0 == F_STEP) {
            // (the counts as 64-bit sums: l > r without a signed compare of what may be 32-bit unsigned columns)
            const u64 Lq = (u64)v1.z + (u64)__popcll(quad_bits(v1) & low_mask(l & 63));
            const u64 Rq = (u64)v2.z + (u64)__popcll(quad_bits(v2) & low_mask((r + 1) & 63));
            l = (int)(unsigned)Lq;
            r = (int)(unsigned)(Rq - 1ull);
            if (Lq >= Rq) {
                ev = FE_FAIL;                          // SBWT.hh:433
                tfail = wstart + j;
            } else if (wstart + (++j) == i + k) {
                ev = FE_END;
            }
        }
#ifdef SBWT_STATS
        for (int q = 0; q < 20; q++) FZ_WHY(q, (why >> q) & 1u);
#endif
        c_tab = uniform32(c_tab + (unsigned)__popcll(__ballot(tabhit)));
        c_stream = uniform32(c_stream + (unsigned)__popcll(__ballot(ev == FE_EMIT1 && trn)));

#ifdef SBWT_TRACE
        if (n_reads == 1 && lane == 0 && busy)
            printf("it: was ext%d trn%d brg%d init%d step%d | now mode %d i %d r %d l %d j %d ev %d res %d seg_n %d absent %d tnext %d wk %d wstart %d b %d | v1 %08x %08x %08x %08x v2 %08x %08x %08x %08x\n",
                   (int)ext, (int)trn, (int)(cmp || brg), 0, 0, mode, i, r, l, j, ev, res, seg_n, (int)ext_absent, tnext, wk, wstart, b,
                   v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w);
#endif
        // ---- events: results, certificates, next state ----
        int burst_hi = -1;                             // >= i: k-mers i..burst_hi are certified absent
        if (!runS) {
            // (SORT, a follower wave: the only event beside FE_EMIT1 is a bridge's burst)
            if (ev == FE_FAIL) {
                burst_hi = burst_to;
                b = -1;
                if (burst_hi > i) fl &= ~(CF_MISS_MASK | CF_NOCERT);
                if (burst_hi == i) { ev = FE_EMIT1; burst_hi = -1; }
            }
        } else {
        if ((fl & CF_ANCH) && (ev == FE_FAIL || ev == FE_ANCH)) {
            // an anchor lookup is over.  Its window is NOT inside k-mer i's, so a miss certifies nothing here: the
            // certificates go on as if it had not been tried (or a seed it left is taken up).  A hit aligns the read.
            const int o = tpos - (wstart + k);
            if (ev == FE_ANCH && tpos >= 0 && (unsigned)(o + i) < (unsigned)ix.n_pos) {
                co = o;
                cP = i;
                m1 = -1;
                fl &= ~(CF_M1 | CF_M2 | CF_ONP);
                mode = F_CMP;
            } else {
                CF_SET_TRIED(fl, wstart);
                mode = F_DEAD;
                do_plan = true;
            }
            fl &= ~CF_ANCH;
            ev = FE_NONE;
        }
        if (ev == FE_END) {
            if (wstart == i) {                         // k chars matched from i: the k-mer is there
                res = l;
                if (l != r) ws->status = SBWT_ERR_NOT_SINGLETON;   // SBWT.hh:410-413
                if (tpos != -1) { r = tpos; rknown = true; emit_pos = (unsigned)tpos; }
                ev = FE_EMIT1;
                b = -1;
                fl &= ~CF_MISS_MASK;
            } else {
                do_plan = true;                        // probe inconclusive: the reference's own walk
                force = true;
            }
        } else if (ev == FE_FAIL) {
            // read[wstart..tfail] is not in the index: k-mers i..min(wstart, m-1) all contain it
            burst_hi = (wstart < mend - 1) ? wstart : (mend - 1);
            if (burst_to >= 0) {                       // bridged substitution: nothing is known about the next one
                burst_hi = burst_to;
                b = -1;
            } else if (wk == 3) {                      // range probe: the bad base is in [wstart, b]
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
            // Certificates that keep failing: a substitution that happens to be ANOTHER strain's base makes every short window
            // around it present, and only each k-mer's own search says it is absent (the reference's loop, SBWT.hh:557-559).
            // Two own searches in a row that certified nothing but their own k-mer, each after probes that found every window (perhaps) present: stop probing (the planner goes blind: own
            // searches only) until something is found again.  Without this such a read cost five to seven iterations per k-mer.
            if (burst_hi > i || wk == 2 || wk == 3 || wk == 6 || (fl & CF_WIN31)) {
                fl &= ~(CF_MISS_MASK | CF_NOCERT);     // a window that IS absent: probes work here
            } else if (wstart == i) {                  // k-mer i's own search, and only k-mer i is answered by it
                if ((fl & CF_MISS_MASK) >= 2u * CF_MISS) { }                 // blind already: until something is found
                else if (!(fl & CF_NOCERT)) fl &= ~CF_MISS_MASK;
                else fl += CF_MISS;
                fl &= ~CF_NOCERT;
            }
            fl &= ~CF_WIN31;
#ifdef SBWT_STATS
            if (mode == F_INIT && (wk == 2 || wk == 3 || wk == 6)) cert_n = burst_hi >= i ? burst_hi - i + 1 : 0;
#endif
            if (burst_hi == i) { ev = FE_EMIT1; burst_hi = -1; }
        }
        if (ev == FE_PRES) {                           // no bad base in [wstart, wstart+pw-1]: shrink the range
            const int lo = blo > i ? blo : i;
            if (wk == 6) b = -1;                       // (the hint was no substitution -- a variant the index knows: forget it)
            else if (wstart > lo) b = wstart - 1;
            else blo = wstart + pw;
            if (blo > b) b = -1;
            do_plan = true;
        }
        }

        // ---- append this iteration's results to the lane's segment list (at most two segments; contiguous ones merge) ----
        auto append = [&](int at, unsigned src) {
            const bool merge = nseg > 0 && ((src == 0xFFFFFFFFu && last_src == 0xFFFFFFFFu) ||
                                            (is_run(src) && is_run(last_src) && last_src + (unsigned)(at - last_start) == src));
            if (!merge) {
                seg_src[nseg][sli] = src;
                seg_at[nseg][sli] = (unsigned char)at;
                nseg++;
                last_src = src;
                last_start = at;
                if (!BIG && (src >> 31) && src != 0xFFFFFFFFu) i0 |= 1 << 24;     // a result known by its column only: the writer's slow pass
            }
        };
        if (ev == FE_EMIT1) {
            // (BIG: a k-mer found without its position cannot be listed -- such an image stores a position with every k-mer,
            // so that is a damaged image: the call fails)
            // (... and the k-mer is listed as "none": for BIG every source but 0xFFFFFFFF is a position, and col[0x80000000 | res]
            // would be a read far outside col[]; the status fails the call either way)
            const bool big_lost = BIG && res != -1 && !rknown;
            if (big_lost) ws->status = SBWT_ERR_NOT_SINGLETON;
            append(i, (res == -1 || big_lost) ? 0xFFFFFFFFu : (rknown ? emit_pos : (0x80000000u | (unsigned)res)));
            i++;
        }
        {
            const int nleft = (burst_hi >= 0) ? (burst_hi - i + 1) : seg_n;
            if (nleft > 0) {
                append(i, (burst_hi >= 0) ? 0xFFFFFFFFu : seg_src_run);
                i += nleft;
            }
            if (cmp && (fl & CF_SEED2)) {
                // a seed of two columns whose first alignment answered nothing (two differences at once: the other strain's path,
                // usually): the same compare on the second column's path, for one more of the read's alignments
                fl &= ~CF_SEED2;
                if (pre_n + abs_n + post_n == 0 && (cact == CA_CERT || cact == CA_LOST) && CF_CMP_LEFT(fl) > 0 &&
                    (unsigned)(l - (wstart + ps) + i) < (unsigned)ix.n_pos) {
                    fl -= 1u << 18;
                    co = l - (wstart + ps);
                    cP = i;
                    m1 = -1;
                    fl &= ~(CF_M1 | CF_M2 | CF_ONP);
                    cact = CA_GOON;
                }
            }
            if (cmp) {                                 // F_CMP's verdicts (at most two segments)
                if (pre_n > 0) { append(i, (unsigned)(co + i + k)); i += pre_n; c_ext += (unsigned)pre_n; fl &= ~CF_MISS_MASK; }
                if (abs_n > 0) { append(i, 0xFFFFFFFFu); i += abs_n; c_brg++; }
                if (post_n > 0) { append(i, (unsigned)(co + i + k)); i += post_n; c_ext += (unsigned)post_n; }
                if (cact == CA_DONE) {
                    mode = F_IDLE;
                } else if (cact == CA_TRANS) {
                    mode = F_TRANS;                    // k-mer i-1 sits at ctr, the read's base i+k-1 is not the path's char there
                    r = ctr;
                    j = 0;
                    fl &= ~CF_ALIGNED;
                } else if (cact == CA_CERT) {
                    b = blo = m1;                      // the certificates, with the bad base known exactly
                    bnext = m2;
                    fl |= CF_ALIGNED;
                    mode = F_DEAD;
                    do_plan = true;
                } else if (cact == CA_LOST) {
                    fl &= ~CF_ALIGNED;
                    CF_SET_TRIED(fl, wstart);          // (an anchor whose stretch is not one path: no second try at it)
                    mode = F_DEAD;
                    do_plan = true;
                } else if (cact == CA_ABSENT) {
                    fl |= CF_ALIGNED;                  // (ext_absent, below: -1, b = m1, the certificates)
                    mode = F_DEAD;
                }
            }
            if (ext_absent) {                          // the k-mer that left the path: -1, then the certificates
                if (trn) { co = r - (i + k - 1); fl |= CF_ALIGNED; }     // (after a transition's quoted steps: the alignment to resume on)
                append(i, 0xFFFFFFFFu);
                b = blo = i + k - 1;
                i++;
                if (i == mend) mode = F_IDLE;
                else do_plan = true;
            }
        }
        // ---- flush: the read is done, or the list could overflow in the next iteration.  The lane hands its list over (it
        //      stays in LDS; the next appends come after the writer's pass) and the whole wave writes it in the NEXT iteration,
        //      between the issue of that iteration's gather and its use (above) ----
        bool want_flush;
        {
            const bool want = nseg > 0 && (i == mend || nseg > FZ_NSEG - 2);
            want_flush = want;
            fm_pend = (!SORT && (ix.debug & 4)) ? 0ull : __ballot(want);      // ("debug" bit 4, experiments: no writer at all)
            if (want) {
                pend_rd = rd;
                // (SORT: bit 23 = the read's last list -- the writer then counts the read as done -- and the list's slot in bits 24 ..)
                pend = (i0 & 0xFF) | (i << 8) | (nseg << 16) | (((i0 >> 16) & 3) << 20) | (((i0 >> 24) & 1) << 22) |
                       (SORT ? (((i == mend) ? 1 : 0) << 23) | (int)((unsigned)slot << 24) : 0);
                nseg = 0;
                i0 = i | (i0 & 0x00FF0000);
            }
        }
        if (ev == FE_EMIT1 || burst_hi >= 0) {
            if (i == mend) {
                mode = F_IDLE;
            } else if (ev == FE_EMIT1 && res != -1) {
                mode = rknown ? tnext : F_POS;         // SBWT.hh:560-
                l = res;
                j = 0;
            } else if (bridged) {
                mode = F_EXT;                          // (i < m: the burst was k k-mers long)
                r += k;
            } else {
                do_plan = true;                        // SBWT.hh:557-559 (with certificates)
            }
        }
        if (do_plan && !force && CF_CMP_LEFT(fl) > 0 && (fl & CF_MISS_MASK) < 2u * CF_MISS &&
            (seed_col != -1 || ((fl & CF_ALIGNED) && wide_k && !(b >= i && b <= i + k - 1) && (unsigned)(co + i) < (unsigned)ix.n_pos))) {
            // k > 31: an alignment instead of a walk.  A seed (the 31-mer [wstart, wstart + 31) is in the index in one column, the
            // k-mer it began is not): that column's path position aligns the read.  Or the alignment the read had when it
            // left its path, once every k-mer that holds the known bad base is answered.  F_CMP's conclusions are exact
            // whatever the alignment is worth.
            fl -= 1u << 18;
            bool keep_plan = false;
            if (seed_col >= 0 && seed_is_pos) {
                // (the position is known: what F_POS would do with it)
                co = seed_col - (wstart + ps);
                fl &= ~(CF_SEED | CF_M1 | CF_M2 | CF_ONP);
                if ((unsigned)(co + i) < (unsigned)ix.n_pos) { cP = i; m1 = -1; mode = F_CMP; }
                else keep_plan = true;                 // (no such alignment: the planner, one alignment poorer)
            } else if (seed_col != -1) {
                l = seed_col >= 0 ? seed_col : -2 - seed_col;
                co = -(wstart + ps);
                fl = (fl & ~CF_SEED2) | CF_SEED | (seed_col < 0 ? CF_SEED2 : 0u);
                mode = F_POS;
            } else {
                cP = i;
                m1 = -1;
                fl &= ~(CF_M1 | CF_M2 | CF_ONP);
                mode = F_CMP;
            }
            do_plan = keep_plan;
        }
        if (do_plan) {
            if (SORT) { mode = F_PLAN; if (force) fl |= CF_FORCE; }      // (the planner runs at the top of a searcher wave's next iteration)
            else plan_walk(force, w31);
        }
        if (SORT) {
            // a follower lane whose read is done lets go of its slot: it travels in `pend` to the writer, which frees it
            if (isP && want_flush && mode == F_IDLE) slot = -1;
            // ---- hand-over: the read is of the other class now.  Its state goes into the slot, the slot's number into the other
            //      class's ring, the lane is idle.  (A list handed to the writer in this very iteration is written first -- the
            //      receiver appends to the same list -- so such a lane waits one iteration.) ----
            const bool foreign = slot >= 0 && (isP ? (mode == F_PLAN) : (mode == F_EXT || mode == F_TRANS || mode == F_BRIDGE));
            const u64 pm = __ballot(foreign && !want_flush);
            if (pm) {
                if ((pm >> lane) & 1ull) {
                    const int sx = SORT ? slot : 0;
                    st_w[0][sx] = rd;
                    st_w[SORT ? 1 : 0][sx] = isP ? ((unsigned)(b + 1) | ((unsigned)(blo + 1) << 8)) : (unsigned)r;
                    st_w[SORT ? 2 : 0][sx] = (unsigned)i | ((unsigned)mend << 8) | ((unsigned)nseg << 16) | ((isP ? 2u : 1u) << 20) |
                                             (((fl & CF_FORCE) ? 1u : 0u) << 22) | (((fl >> 10) & 7u) << 23) |
                                             ((unsigned)((i0 >> 16) & 3) << 26) | ((unsigned)((i0 >> 24) & 1) << 28);
                    st_w[SORT ? 3 : 0][sx] = (unsigned)(i0 & 0xFF) | ((unsigned)(bnext + 1) << 8);
                }
                ring_push(pm, isP ? 0 : 1, slot);
                if ((pm >> lane) & 1ull) { slot = -1; mode = F_IDLE; }
            }
        }
#ifdef SBWT_STATS
        for (int q = 0; q < 24; q++) FZ_PLAN(q, pl_kind == q || oc_kind == q);
        FZ_PLAN_SUM(24, cert_n);
#endif
    }

    FZ_HIST_FLUSH();
#ifdef SBWT_TIMELINE
    if (lane == 0) {
        const u64 tl_exit = wall_clock64(), t0 = g_fz_tl[0];
        atomicMin(&g_fz_tl[1], tl_drain); atomicMax(&g_fz_tl[2], tl_drain); atomicMax(&g_fz_tl[3], tl_exit);
        atomicAdd(&g_fz_tl[4], 1ull); atomicAdd(&g_fz_tl[5], tl_exit - tl_drain); atomicAdd(&g_fz_tl[6], tl_exit - tl_start);
        atomicAdd(&g_fz_tl[7], (unsigned long long)tl_n);
        const u64 bk = (tl_exit - t0) / 1000ull;
        atomicAdd(&g_fz_tl_hist[bk < 1023 ? bk : 1023], 1ull);
    }
#endif
    {
        u64 e = c_ext, eb = c_brg;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { e += __shfl_down(e, off); eb += __shfl_down(eb, off); }
        if (lane == 0) { if (e) atomicAdd(&ws->n_ext, e); if (eb) atomicAdd(&ws->n_bridge, eb); }
    }
    if (lane == 0) {   // the counters are wave-uniform
        if (c_stream) atomicAdd(&ws->n_stream, (u64)c_stream);
        if (c_search) atomicAdd(&ws->n_search, (u64)c_search);
        if (c_lf) atomicAdd(&ws->n_lf, (u64)c_lf);
        if (c_tab) atomicAdd(&ws->n_tab_hit, (u64)c_tab);
    }
}

// Do all reads have one length and all result ranges one stride?  Thread 0 also notes the first offsets.
__global__ void __launch_bounds__(256) k_check_uniform2(const i64 *__restrict__ read_off, const i64 *__restrict__ out_off,
                                                        i64 n_reads, SbwtWorkHeader *ws, int k, SbwtPieceTab pt,
                                                        int rg_enable) {
    const i64 t = (i64)blockIdx.x * 256 + threadIdx.x;
    const i64 len = read_off[1] - read_off[0], stride = (n_reads > 1) ? out_off[1] - out_off[0] : 0;
    if (t == 0) { ws->u_read0 = read_off[0]; ws->u_len = len; ws->u_out0 = out_off[0]; ws->u_stride = stride; }
    const bool valid = t < n_reads;
    const i64 mylen = valid ? read_off[t + 1] - read_off[t] : 0;
    fused_sample_of_wave(t, mylen, valid, n_reads, rg_enable, k, ws);
    piece_zones_of_wave(t, mylen, valid, k, ws, pt);                       // long reads: for the general kernel behind
    if (!valid) return;
    const bool bad = (mylen != len) || (t + 1 < n_reads && out_off[t + 1] - out_off[t] != stride);
    if (__ballot(bad) != 0 && (threadIdx.x & 63) == 0) ws->u_bad = 1ull;
}

void sbwt_launch_search_fused(const SbwtIndexView &ix, const char *d_bases, long long total_bases, uint4 *d_packed,
                              const long long *d_read_off, const long long *d_out_off, long long *d_out, long long n_reads,
                              SbwtWorkHeader *ws, int streaming, hipStream_t stream, unsigned *d_defer,
                              hipEvent_t ev_begin, hipEvent_t ev_end, SbwtPieceTab pt, int ragged_ok) {
    if (n_reads <= 0) return;
    hipLaunchKernelGGL(k_check_uniform2, dim3(grid_for(n_reads)), dim3(256), 0, stream, d_read_off, d_out_off, (i64)n_reads, ws,
                       ix.k, pt, ragged_ok);
    const i64 want = (n_reads + 255) / 256;
    // Workgroups: five per CU fill the chip and give the most k-mers per second -- and the slowest iterations (the vector ALU is
    // shared by five waves per SIMD), which is what a launch's END is made of: the last reads' chains of 20-40 iterations.  A
    // small batch is mostly end, so it gets fewer, faster waves (measured in one process, config 2, kernel ms at 1280 workgroups
    // and at the best number: 200 K reads 0.447 -> 0.357 at 512, 1 M reads 0.895 -> 0.792 at 768, 4 M and 10 M reads: 1024 and
    // 1280 the same, 896 +2-3 %; config 5, 1 M reads: 0.882 -> 0.816 at 768; with the steps below against 1280 throughout:
    // 400 K reads 73 -> 93 G k-mers/s, 1 M reads 129 -> 144, 2 M reads 174 -> 179, from 4 M reads on the same).
    // "debug" >> 8 sets the number.
    unsigned cap = 1280u;
    if (ix.debug >> 8) cap = (unsigned)(ix.debug >> 8);
    else if (total_bases < 8000000ll) cap = 256u;
    else if (total_bases < 20000000ll) cap = 384u;
    else if (total_bases < 60000000ll) cap = 512u;
    else if (total_bases < 110000000ll) cap = 640u;
    else if (total_bases < 170000000ll) cap = 768u;
    else if (total_bases < 250000000ll) cap = 896u;
    else if (total_bases < 500000000ll) cap = 1024u;
    const unsigned g = (unsigned)(want < (i64)cap ? want : (i64)cap);
    if (ev_begin) (void)hipEventRecord(ev_begin, stream);
    // (k > 31: whole k-mers in the two-level table; "debug" bit 64: the wide walk for every k -- experiments and the fuzzer)
    const bool wide = (ix.stab2 != nullptr && ix.p_sparse < ix.k) || (ix.debug & 64);
    // (two launches: the instantiation for one read length and one ticket per read, then the one for everything else; the one
    // whose batch it is not returns at once)
#define FZ_LAUNCH1(W, O, B, U) hipLaunchKernelGGL((k_search_fused<W, O, B, U>), dim3(g), dim3(256), 0, stream, ix, \
                                           reinterpret_cast<const unsigned char *>(d_bases), (i64)total_bases, d_out, (i64)n_reads, ws, d_defer, \
                                           d_read_off, d_out_off)
#define FZ_LAUNCH(W, O, B) do { FZ_LAUNCH1(W, O, B, true); FZ_LAUNCH1(W, O, B, false); } while (0)
#define FZ_LAUNCH_S(O) do { hipLaunchKernelGGL((k_search_fused<false, O, false, true, true>), dim3(g), dim3(256), 0, stream, ix, \
                                           reinterpret_cast<const unsigned char *>(d_bases), (i64)total_bases, d_out, (i64)n_reads, ws, d_defer, \
                                           d_read_off, d_out_off); \
                            hipLaunchKernelGGL((k_search_fused<false, O, false, false, true>), dim3(g), dim3(256), 0, stream, ix, \
                                           reinterpret_cast<const unsigned char *>(d_bases), (i64)total_bases, d_out, (i64)n_reads, ws, d_defer, \
                                           d_read_off, d_out_off); } while (0)
    const bool big = ix.n_nodes >= ((i64)1 << 31) - 64;        // (the C ABI sends such an index here only with k <= 31 and int64 results)
    if (big) FZ_LAUNCH(false, false, true);
    else if (wide) { if (ix.out32) FZ_LAUNCH(true, true, false); else FZ_LAUNCH(true, false, false); }
    else if (ix.fused_sort > 0) { if (ix.out32) FZ_LAUNCH_S(true); else FZ_LAUNCH_S(false); }      // lanes sorted by state (k <= 31)
    else      { if (ix.out32) FZ_LAUNCH(false, true, false); else FZ_LAUNCH(false, false, false); }
#undef FZ_LAUNCH_S
#undef FZ_LAUNCH
#undef FZ_LAUNCH1
    if (ev_end) (void)hipEventRecord(ev_end, stream);
    // what the fused kernel did not take: everything when the reads are not of one length, else the reads it handed on
    sbwt_launch_encode_chained(d_bases, total_bases, d_packed, ws, d_defer, d_read_off, ix.k, stream);
    sbwt_launch_piece_bounds(d_packed, d_read_off, d_out_off, ix.k, ws, pt, 1, stream);
    sbwt_launch_search_chained(ix, d_packed, d_read_off, d_out_off, d_out, n_reads, ws, streaming, stream, d_defer, pt);
}
