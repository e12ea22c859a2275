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
// SORT: [0] / [4] wave-iterations of searcher / follower waves, [1] / [5] their busy lanes, [2] / [6] lanes waiting at their ring,
// [3] / [7] idle lanes that hold a slot, [8] hand-overs to the followers, [9] to the searchers, [10] free slots sent back,
// [11] hand-overs that waited for the writer, [12] iterations skipped (a wave with nothing to do)
__device__ unsigned long long g_sort_stats[16];
extern "C" int sbwtgpu_debug_sort_stats(unsigned long long *out, int reset) {
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sort_stats), sizeof(g_sort_stats)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_sort_stats), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#define FZ_SORT_STAT(q, val) do { const unsigned long long v_ = (unsigned long long)(val); if (lane == 0 && v_) atomicAdd(&g_sort_stats[q], v_); } while (0)
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
#define FZ_SORT_STAT(q, val) do { } while (0)
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
// SORT runs four workgroups per CU instead of five: 128 VGPRs (the slot and the hand-over's temporaries do not fit the 96 of five
// waves per SIMD: 32-49 registers went to scratch), and the LDS of the fifth workgroup buys FZ_SLOTS = 320 slots for 256 lanes --
// reads wait in the rings without idling a lane.
#ifdef FZ_SORT_WG5              // (experiment: five workgroups per CU, as many slots as lanes)
#define FZ_SLOTS 256
#define FZ_RING 256
#define FZ_SORT_WGS 5
#else
#define FZ_SLOTS 320
#define FZ_RING 512
#define FZ_SORT_WGS 4
#endif
// TAB (round 6): the instantiation for batches that are cut into a ticket table (fused mode 3; a !UNI one).  Its own instantiation
// because the table's code in the refill cost the batches of mixed lengths and of pieces 2-3 % (ragged 80-150: 4.05 -> 4.12 ms,
// 250 bp reads 5.66 -> 5.84) -- three kernels are launched per call now, two of which return at once.
template <bool WIDE, bool O32, bool BIG = false, bool UNI = false, bool SORT = false, bool TAB = false>
// (measured and dropped: the k > 31 instantiation for batches of mixed lengths with 128 VGPRs / four workgroups per CU instead of its two
// spilled registers: ragged reads 3.82 -> 4.12 ms, 1 kbp reads 186 -> 169 G k-mers/s)
__global__ void __launch_bounds__(256, SORT ? FZ_SORT_WGS : 5) k_search_fused(SbwtIndexView ix, const unsigned char *__restrict__ bases,
                                                          i64 total_bases, i64 *__restrict__ out, i64 n_reads,
                                                          SbwtWorkHeader *ws, unsigned *__restrict__ defer_list,
                                                          const i64 *__restrict__ read_off, const i64 *__restrict__ out_off,
                                                          SbwtTickTab tt) {
    static_assert(!SORT || (!WIDE && !BIG), "lanes sorted by state: k <= 31, fewer than 2^31 columns (so far)");
    __shared__ u64 pool_codes[SBWT_FUSED_MAXG][SORT ? 128 : 256];   // the wave's pool of 64 tickets, encoded ([group][wave * 64 + ticket]; SORT: searcher waves only)
    constexpr int NSLOT = SORT ? FZ_SLOTS : 256;
    __shared__ u64 cur_codes[SBWT_FUSED_MAXG][NSLOT];       // the read this lane is working on (SORT: by slot)
    __shared__ unsigned seg_src[FZ_NSEG][NSLOT];            // segment lists: source ...
    __shared__ unsigned char seg_at[FZ_NSEG][NSLOT];        // ... and first k-mer
    // SORT: a slot's state while it waits in a ring -- [0] read, [1] path position r (to a follower) / b+1 | blo+1 << 8 (to a
    // searcher), [2] i | mend << 8 | nseg << 16 | kind << 20 (0 free slot, 1 F_EXT, 2 F_PLAN) | force << 22 | fl's miss and nocert
    // bits << 23 | piece << 26 | list holds a column-only result << 28, [3] first result not written | bnext+1 << 8
    __shared__ unsigned st_w[SORT ? 4 : 1][SORT ? FZ_SLOTS : 1];
    // (read and written through pointers cast to the LDS address space, FZ_LDS16 / FZ_LDS32: a volatile access through a generic
    // pointer -- or to a volatile __shared__ array -- is compiled to flat_load / flat_store "sc0 sc1", which count on vmcnt as well:
    // a wait for the writer's stores at every look at a ring)
    __shared__ unsigned short q_ring[SORT ? 2 : 1][SORT ? FZ_RING : 1];   // the rings: [0] to the searchers, [1] to the followers; slot + 1, 0 = not written yet
    __shared__ unsigned q_ctl[8];                           // [0], [1] places handed out in ring 0 / 1; [2], [3] entries written; [4] reads in flight; [5] searcher waves drained; [6] iterations with work (progress)
    const int fmode = sbwt_fused_mode(ws, ix.k);
    if (fmode == 0) return;                                 // the general route does it all
    const int P_batch = sbwt_fused_pieces(ws, ix.k);
    if (UNI != (fmode == 1 && P_batch == 1 && !(ix.debug & 128))) return;      // the other instantiation's batch
    const bool ragged = !UNI && fmode == 2;                 // reads of any lengths: offsets fetched with every refill
    // (round 6) reads of any lengths, many of them long: the batch as a TABLE of tickets of <= 160 bases (k_fused_tickets): a
    // ticket is a read of its own -- where its bases are, where its results go -- and `rd` is the ticket's number
    static_assert(!TAB || !UNI, "the ticket table's instantiation is a general one");
    const bool table = TAB && fmode == 3 && tt.tick != nullptr;
    if (TAB != (fmode == 3)) return;                        // (a table batch is the TAB instantiation's, and nothing else is)
    if (!WIDE && !BIG && ix.fused_sort > 0) {
        // the sorted or the unsorted instantiation?  "fused_sort" bit 12: always the sorted one; else by the hint the call before left
        // (not for batches of pieces -- reads of 161 .. 480 bases as two or three tickets each: measured 5 % slower sorted)
        const bool sorted_call = (ix.fused_sort & 4096) ||
                                 (ws->hint == (unsigned long long)(SBWT_HINT_MAGIC | (unsigned)SBWT_HINT_CALLS) && (UNI || TAB || P_batch == 1));
        if (SORT != sorted_call) return;
    } else if (SORT) return;
    const int tid = threadIdx.x, lane = tid & 63, wbase = tid & ~63;
    const bool follower_wave = SORT && __builtin_amdgcn_readfirstlane(tid) >= 128;     // a path-follower wave (wave-uniform, in a scalar register)
    int slot = SORT ? (follower_wave ? -1 : tid) : tid;     // SORT: the slot this lane holds; -1: none; <= -2: none, and the lane holds
                                                            // place -2 - slot of its class's ring
    if (SORT) {
        // slots 0 .. 127 start with the searcher lanes, the others wait in the searchers' ring as free slots
        if (tid < 8) q_ctl[tid] = (tid == 2) ? (unsigned)(FZ_SLOTS - 128) : 0u;
        for (int q = tid; q < (SORT ? FZ_RING : 0); q += 256) {
            q_ring[0][SORT ? q : 0] = q < FZ_SLOTS - 128 ? (unsigned short)(128 + q + 1) : (unsigned short)0;
            q_ring[SORT ? 1 : 0][SORT ? q : 0] = 0;
        }
        for (int q = 128 + tid; q < (SORT ? FZ_SLOTS : 0); q += 256) st_w[SORT ? 2 : 0][SORT ? q : 0] = 0u;
        __syncthreads();
    }
#define FZ_LDS32(p) (*(__attribute__((address_space(3))) volatile unsigned *)(p))
#define FZ_LDS16(p) (*(__attribute__((address_space(3))) volatile unsigned short *)(p))
    // entries of a ring: written after the slot's state (LDS operations of a wave complete in order; the wait keeps the compiler
    // and the hardware from letting the entry pass the state), read before it (the state's address depends on the entry)
    auto ring_push = [&](const u64 mask, const int ring, const int s) {
        // (wave-uniform call: every lane of `mask` appends slot s to `ring`)
        unsigned base = 0;
        if (lane == 0) base = __hip_atomic_fetch_add(&q_ctl[2 + ring], (unsigned)__popcll(mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        base = uniform32(base);
        if ((mask >> lane) & 1ull) {
            const unsigned at = (base + (unsigned)__popcll(mask & low_mask(lane))) & (unsigned)(FZ_RING - 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            FZ_LDS16(&q_ring[SORT ? ring : 0][SORT ? at : 0]) = (unsigned short)(s + 1);
        }
    };
    const int k = ix.k, p = ix.p_dev, L0 = ix.probe_len, ps = ix.p_sparse;
    // tickets: P per read (reads of more than 160 bases as pieces that overlap by k-1; sbwt_kernels_common.h)
    const int P = (UNI || table) ? 1 : P_batch, kpp = SBWT_FUSED_MAXLEN - k + 1;
    const bool varlen = ragged || table || P > 1;           // the tickets' lengths differ: each refill notes them
    const i64 n_tickets = table ? (i64)ws->n_ftick : n_reads * P;
    const int ulen = (ragged || table) ? SBWT_FUSED_MAXLEN : (int)ws->u_len, m = ulen - k + 1, G = varlen ? SBWT_FUSED_MAXG : ((ulen + 31) >> 5);
    const i64 u_read0 = ws->u_read0, u_out0 = ws->u_out0, u_stride = ws->u_stride;
    const bool pfon = ix.pfil && ix.p_filter == L0 && L0 > p;
    const u64 mk2 = (k - ps >= 32) ? ~0ull : low_mask(2 * ((k - ps) & 31));   // key mask of the second-level window
    const int pw = pfon ? L0 : p;                   // window of a range probe: the filter's when there is one
    const int last_node = (int)(ix.n_nodes - 1);
    static_assert(!BIG || !O32, "2^31 columns and more: int64 results");
    auto zx = [](int v) -> i64 { return BIG ? (i64)(unsigned)v : (i64)v; };                    // a column / position as an index
    auto is_run = [](unsigned src) -> bool { return BIG ? src != 0xFFFFFFFFu : !(src >> 31); };   // a segment's source is a path position

    int nseg = 0, i0 = 0, last_start = 0;           // segments listed; first result of the read not written yet
    unsigned sort_spins = 0;        // SORT, wave-uniform: iterations this wave found nothing to do in
    unsigned sort_seen = 0;         // ... the workgroup's progress counter (q_ctl[6]) when it last looked
    unsigned sort_waits = 0;        // ... and times in a row it has waited for more of its lanes to have work
#ifdef SBWT_SORT_COUNTERS           // (tools/build_variant_lib.sh sortcnt -DSBWT_SORT_COUNTERS=1; tools/ab_step.py prints them)
    unsigned c_iter = 0;            // SORT, wave-uniform: iterations with a gather, and the busy lanes in them (ws->pad[11..14])
    unsigned long long c_busy = 0;
    unsigned c_lists = 0, c_part = 0;       // SORT, wave-uniform: lists written, and those of them that were not a read's last (ws->pad[9], pad[10])
#endif
    const int sort_thr = (SORT && (ix.fused_sort & 255) > 1) ? ((ix.fused_sort & 255) < 64 ? (ix.fused_sort & 255) : 64) : 0;    // "fused_sort" & 255 = n > 1: that many busy lanes
    const int sort_ship = SORT ? ((ix.fused_sort >> 8) & 15) : 0;    // "fused_sort" >> 8: of eight follower lanes, those whose finished reads the searchers write
    u64 fm_pend = 0;                // wave-uniform: lanes whose segment list waits for the writer (handed over at the end of an iteration)
    int pend = 0;                   // ... that list: first result | one past the last << 8 | segments << 16 | piece << 20 | holds a result known by column only << 22
    unsigned pend_rd = 0;           // ... and its read (SORT: the list's slot instead -- the writer finds the read there)
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
    const bool seed_ok = wide_k && ps < k && !BIG;       // (seeds travel as signed columns / positions: not in the 32-bit unsigned layout)
    int bnext = -1;                 // a hint: the read's next difference from its path after b (a failed bridge's compare saw it)
    int mode = F_IDLE;
    unsigned rd = 0;                // the read this lane works on
    int i = 0, j = 0, b = -1, wstart = 0;
    int mend = 0;                   // this lane answers k-mers [i, mend) of its read (mend = m unless the read was split, below)
    bool drained = false;           // wave-uniform: the ticket counter has run past the last read
    int l = 0, r = 0;               // walk interval; F_EXT ..: r = path position
    unsigned c_ext = 0;             // per lane: k-mers answered along paths (low 20 bits), substitutions bridged (12 bits)
    constexpr bool LSEG_LDS = SORT || !UNI;
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

    // The main loop, compiled ONCE PER WAVE CLASS for SORT: the other class's blocks are not in a wave's code at all, and
    // neither are their registers -- the two loops are allocated independently (a follower needs no table pointers, hash
    // constants or planner state; a searcher no path-compare temporaries).
    if (SORT && follower_wave) {
        constexpr bool isP = true;
        constexpr bool runS = !SORT || !isP, runP = !SORT || isP;   // which blocks of the state machine this loop holds
        constexpr int qc = isP ? 1 : 0;                             // this wave's ring
#include "sbwt_search_fused_loop.inc"
    } else {
        constexpr bool isP = false;
        constexpr bool runS = !SORT || !isP, runP = !SORT || isP;
        constexpr int qc = isP ? 1 : 0;
#include "sbwt_search_fused_loop.inc"
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
        u64 e = c_ext & 0xFFFFFu, eb = c_ext >> 20;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { e += __shfl_down(e, off); eb += __shfl_down(eb, off); }
        if (lane == 0) { if (e) atomicAdd(&ws->n_ext, e); if (eb) atomicAdd(&ws->n_bridge, eb); }
    }
#if defined(SBWT_SORT_COUNTERS) && !defined(SBWT_STATS)
    if (SORT && lane == 0) {
        // (what a wave class's iterations cost: their number and the busy lanes in them; tools/ab_step.py prints the ratio)
        atomicAdd(&ws->pad[follower_wave ? 13 : 11], (unsigned long long)c_iter);
        atomicAdd(&ws->pad[follower_wave ? 14 : 12], c_busy);
        atomicAdd(&ws->pad[9], (unsigned long long)c_lists);
        atomicAdd(&ws->pad[10], (unsigned long long)c_part);
    }
#endif
    if (lane == 0) {   // the counters are wave-uniform
        if (c_stream) atomicAdd(&ws->n_stream, (u64)c_stream);
        if (c_search) atomicAdd(&ws->n_search, (u64)c_search);
        if (c_lf) atomicAdd(&ws->n_lf, (u64)c_lf);
        if (c_tab) atomicAdd(&ws->n_tab_hit, (u64)c_tab);
    }
}

// The ticket table of the fused route (fused mode 3, round 6).  A batch of reads of any lengths of which more than one in eight
// is longer than three pieces used to be the general kernel's altogether (for 31 < k <= 63 at half the fused kernel's rate: it
// has no aligned compare).  Now every read of such a batch is cut into tickets of <= 160 bases that overlap by k-1 -- exact:
// the fused kernel only walks upper-case ACGT, where a k-mer's result does not depend on what came before it -- and the
// tickets are listed here: one atomic per wave reserves the wave's entries, a read of many tickets (a genome: 50 000) is
// listed by the whole wave.  Ticket order is not read order; nothing depends on it.
__global__ void __launch_bounds__(256) k_fused_tickets(const i64 *__restrict__ read_off, const i64 *__restrict__ out_off, i64 n_reads,
                                                       SbwtWorkHeader *ws, int k, SbwtTickTab tt) {
    if (!sbwt_fused_wants_table(ws, k)) return;
    const int lane = threadIdx.x & 63;
    const int kpp = SBWT_FUSED_MAXLEN - k + 1;
    const i64 stride = (i64)gridDim.x * 256;
    for (i64 r0 = (i64)blockIdx.x * 256 + (threadIdx.x & ~63); r0 < n_reads; r0 += stride) {     // (wave-uniform loop)
        const i64 r = r0 + lane;
        const bool valid = r < n_reads;
        const i64 P0 = valid ? read_off[r] : 0, len = valid ? read_off[r + 1] - P0 : 0, ob = valid ? out_off[r] : 0;
        const i64 m = len - k + 1;
        const u64 np = m > 0 ? (u64)((m + kpp - 1) / kpp) : 0ull;
        u64 incl = np;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const u64 v = __shfl_up(incl, o); if (lane >= o) incl += v; }
        const u64 total = __shfl(incl, 63);
        if (total == 0) continue;
        u64 base = 0;
        if (lane == 0) base = atomicAdd(&ws->n_ftick, (unsigned long long)total);
        base = uniform64(base);
        if (base + total > (u64)tt.cap) {              // the table is too small for this batch: the general route takes it
            if (lane == 0) ws->ftick_over = 1ull;
            continue;
        }
        const u64 first = base + incl - np;
        auto entry = [&](i64 rr, i64 p0, i64 ln, i64 o0, u64 t0, u64 j) {
            const i64 b = p0 + (i64)j * kpp, left = ln - (i64)j * kpp;
            const unsigned nb = (unsigned)(left > SBWT_FUSED_MAXLEN ? SBWT_FUSED_MAXLEN : left);
            const i64 o = o0 + (i64)j * kpp;
            tt.tick[t0 + j] = make_uint4((unsigned)b, ((unsigned)((u64)b >> 32) & 0xFFFFu) | (nb << 16), (unsigned)o, (unsigned)((u64)o >> 32));
            tt.tick_read[t0 + j] = (unsigned)rr;
        };
        for (u64 j = 0; j < np && j < 4; j++) entry(r, P0, len, ob, first, j);      // (short reads: by their own lane)
        u64 big = __ballot(np > 4);
        while (big) {                                   // (long reads: by the whole wave)
            const int src = __ffsll((i64)big) - 1;
            big &= big - 1;
            const i64 rr = __shfl(r, src), p0 = __shfl(P0, src), ln = __shfl(len, src), o0 = __shfl(ob, src);
            const u64 t0 = __shfl(first, src), nn = __shfl(np, src);
            for (u64 j = 4 + (u64)lane; j < nn; j += 64) entry(rr, p0, ln, o0, t0, j);
        }
    }
}

// The hint for the next call on this workspace (SbwtWorkHeader::hint): did this call's reads mostly follow their paths?
__global__ void k_fused_hint(SbwtWorkHeader *ws, int k) {
    if (threadIdx.x != 0 || sbwt_fused_mode(ws, k) == 0) return;          // (the general route took the batch: nothing learnt)
    const unsigned long long ext = ws->n_ext, walks = ws->n_search, prev = ws->hint;
    const bool fits = walks > 0 && ext >= (unsigned long long)SBWT_HINT_RATIO * walks;
    // how many calls in a row (up to SBWT_HINT_CALLS) had the sorted kernel's work mix; one that has not starts the count again:
    // the wrong kernel costs 5 % one way and 40 % the other, so a workspace whose batches alternate stays on the unsorted one
    const unsigned run = ((prev & ~0xFFull) == (unsigned long long)SBWT_HINT_MAGIC) ? (unsigned)(prev & 0xFFull) : 0u;
    ws->hint = (unsigned long long)(SBWT_HINT_MAGIC | (fits ? (run < SBWT_HINT_CALLS ? run + 1u : (unsigned)SBWT_HINT_CALLS) : 0u));
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
                              hipEvent_t ev_begin, hipEvent_t ev_end, SbwtPieceTab pt, int ragged_ok, SbwtTickTab tt) {
    if (n_reads <= 0) return;
    hipLaunchKernelGGL(k_check_uniform2, dim3(grid_for(n_reads)), dim3(256), 0, stream, d_read_off, d_out_off, (i64)n_reads, ws,
                       ix.k, pt, ragged_ok);
    // the ticket table of a batch with many long reads (returns at once for every other batch)
    if (tt.tick && n_reads < ((i64)1 << 32)) {
        const i64 wantt = (n_reads + 255) / 256;
        hipLaunchKernelGGL(k_fused_tickets, dim3((unsigned)(wantt < 1024 ? wantt : 1024)), dim3(256), 0, stream, d_read_off, d_out_off,
                           (i64)n_reads, ws, ix.k, tt);
    }
    // (one lane per read -- or, with the ticket table of a batch of long reads, per ticket of <= 160 bases: how many there are is
    // known on the device only, total_bases / 160 is what they are at least)
    const i64 want_r = (n_reads + 255) / 256, want_t = tt.tick ? (total_bases / SBWT_FUSED_MAXLEN + 255) / 256 : 0;
    const i64 want = want_r > want_t ? want_r : want_t;
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
    const bool sorted = !ix.big && !(ix.n_nodes >= ((i64)1 << 31) - 64) && !((ix.stab2 != nullptr && ix.p_sparse < ix.k) || (ix.debug & 64)) && ix.fused_sort > 0;
    const unsigned cap_s = cap > 256u * FZ_SORT_WGS ? 256u * FZ_SORT_WGS : cap;      // (the SORT instantiation: four workgroups per CU)
    const unsigned g = (unsigned)(want < (i64)cap ? want : (i64)cap), g_s = (unsigned)(want < (i64)cap_s ? want : (i64)cap_s);
    if (ev_begin) (void)hipEventRecord(ev_begin, stream);
    // (k > 31: whole k-mers in the two-level table; "debug" bit 64: the wide walk for every k -- experiments and the fuzzer)
    const bool wide = (ix.stab2 != nullptr && ix.p_sparse < ix.k) || (ix.debug & 64);
    // (two launches: the instantiation for one read length and one ticket per read, then the one for everything else; the one
    // whose batch it is not returns at once)
#define FZ_LAUNCH1(W, O, B, U) hipLaunchKernelGGL((k_search_fused<W, O, B, U>), dim3(g), dim3(256), 0, stream, ix, \
                                           reinterpret_cast<const unsigned char *>(d_bases), (i64)total_bases, d_out, (i64)n_reads, ws, d_defer, \
                                           d_read_off, d_out_off, tt)
#define FZ_LAUNCH(W, O, B) do { FZ_LAUNCH1(W, O, B, true); FZ_LAUNCH1(W, O, B, false); \
                                if (tt.tick) hipLaunchKernelGGL((k_search_fused<W, O, B, false, false, true>), dim3(g), dim3(256), 0, stream, ix, \
                                           reinterpret_cast<const unsigned char *>(d_bases), (i64)total_bases, d_out, (i64)n_reads, ws, d_defer, \
                                           d_read_off, d_out_off, tt); } while (0)
#define FZ_LAUNCH_S(O) do { hipLaunchKernelGGL((k_search_fused<false, O, false, true, true>), dim3(g_s), dim3(256), 0, stream, ix, \
                                           reinterpret_cast<const unsigned char *>(d_bases), (i64)total_bases, d_out, (i64)n_reads, ws, d_defer, \
                                           d_read_off, d_out_off, tt); \
                            hipLaunchKernelGGL((k_search_fused<false, O, false, false, true>), dim3(g_s), dim3(256), 0, stream, ix, \
                                           reinterpret_cast<const unsigned char *>(d_bases), (i64)total_bases, d_out, (i64)n_reads, ws, d_defer, \
                                           d_read_off, d_out_off, tt); \
                            if (tt.tick) hipLaunchKernelGGL((k_search_fused<false, O, false, false, true, true>), dim3(g_s), dim3(256), 0, stream, ix, \
                                           reinterpret_cast<const unsigned char *>(d_bases), (i64)total_bases, d_out, (i64)n_reads, ws, d_defer, \
                                           d_read_off, d_out_off, tt); } while (0)
    const bool big = ix.big || ix.n_nodes >= ((i64)1 << 31) - 64;     // (the C ABI sends such an index here only with int64 results)
    if (big) { if (wide) FZ_LAUNCH(true, false, true); else FZ_LAUNCH(false, false, true); }
    else if (wide) { if (ix.out32) FZ_LAUNCH(true, true, false); else FZ_LAUNCH(true, false, false); }
    else {
        // k <= 31: the unsorted instantiations and (when "fused_sort" is on) the sorted ones; each returns at once from a call that is
        // the other's (the hint at the end of the workspace, SbwtTickTab::hint)
        if (ix.out32) FZ_LAUNCH(false, true, false); else FZ_LAUNCH(false, false, false);
        if (sorted) { if (ix.out32) FZ_LAUNCH_S(true); else FZ_LAUNCH_S(false); }
    }
#undef FZ_LAUNCH_S
#undef FZ_LAUNCH
#undef FZ_LAUNCH1
    if (ev_end) (void)hipEventRecord(ev_end, stream);
    // what this call's work mix says about the next call's kernel
    if (sorted) hipLaunchKernelGGL(k_fused_hint, dim3(1), dim3(64), 0, stream, ws, ix.k);
    // what the fused kernel did not take: everything when the reads are not of one length, else the reads it handed on
    sbwt_launch_encode_chained(d_bases, total_bases, d_packed, ws, d_defer, d_read_off, ix.k, stream);
    sbwt_launch_piece_bounds_tt(d_packed, d_read_off, d_out_off, ix.k, ws, pt, 1, stream, tt.defer_bits);
    sbwt_launch_search_chained(ix, d_packed, d_read_off, d_out_off, d_out, n_reads, ws, streaming, stream, d_defer, pt);
}
