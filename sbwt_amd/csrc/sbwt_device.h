// sbwt_device.h -- device-side data layout shared by the kernels and the C-ABI host code.
//
// Device image ("blob") of a plain-matrix SBWT index, one contiguous allocation:
//
//   [ blocks   : n_blocks x 64 B ]   n_blocks = n_nodes/64 + 1  (so pos == n_nodes is addressable)
//   [ ptab     : 4^p_dev x 16 B  ]   device prefix table, (first,second) int64 pairs
//   [ ftab     : 4^p_file x 16 B ]   the index file's own table (only if p_file != p_dev, p_file > 0)
//   [ mega     : 4 x n_mega x 8 B]   absolute (C[c] + rank_c) at every 2^31-column boundary
//   [ col, pos : n_nodes x 4 B each ]  path order (see k_path_* in sbwt_derived.hip); col[n_pos] = 0xFFFFFFFF (the "position" of a -1)
//   [ pq       : (n/32+4) x 16 B ]   path groups of 32 positions: packed chars + two state bits per position
//   [ trans    : n_tslots x 32 B ]   transition table: one hashed entry per way off a path -- (position, char) -> column,
//                                    path position and the next 32 steps of its path; sized once the path order is known:
//                                    3 entries per branching column + the successors of every path's last column
//   [ pfil     : 2^log2f x 16 B  ]   probe filter (see k_pf_insert)
//   [ stab2    : n_sb2 x 32 B    ]   second level for 31 < k <= 63: buckets of two 16-byte entries { rest key (8 B), first
//                                    column of the 31-prefix's interval | overflow flag (entry 0), payload | USED }; the
//                                    payload is the k-mer's PATH POSITION when the image has a path order (its column is
//                                    col[position]), else its column
//   [ stab     : n_sb x 32 B     ]   sparse prefix table at depth p_sparse (only non-empty prefixes), hashed:
//                                    bucket = two 16-byte entries { key | flags (u64), first (u32), second-first (u32) }
//
// Indexes of 2^31 .. 2^32 - 2^24 columns (round 5, k <= 31): the same layout with every column and path position a full
// 32-bit UNSIGNED value (0xFFFFFFFF stays "none"), block counts absolute (C[c] + rank_c < 2^32: the mega table is all zero) and
// no second-level table; read by k_search_fused<false, false, BIG> (sbwt_search_fused.hip), the general kernels see blocks +
// ptab only (general_view, sbwt_search.hip).
//
// One 64-byte block covers 64 consecutive columns and carries everything both query kinds need,
// as four 16-byte quads, quad c for symbol c in {A,C,G,T}:
//
//   quad c = { bits_c[31:0], bits_c[63:32], cnt_c, ssup_piece }
//     bits_c     the 64 bits of column c's row for these columns (sdsl bit order: LSB first)
//     cnt_c      C[c] + rank_c(block start) - mega[c][block_start >> 31]   (u32)
//     ssup_piece low 32 bits of the block's suffix_group_starts word in quads 0 and 2,
//                high 32 bits in quads 1 and 3 (so either quad pair {0,1} / {2,3} holds the
//                whole word)
//
// A search LF step (SBWT.hh:430-431) therefore needs ONE 16-byte quad per rank position; a
// streaming step (SBWT.hh:562-575) needs the 32-byte quad pair that contains quad c -- both
// inside one 64-byte line.  This replaces sdsl's per-column rank_support_v5 (2 cache lines per
// rank) and the separate suffix_group_starts vector (a third line).
#pragma once
#include <stdint.h>

#define SBWT_MEGA_SHIFT 31          // columns per mega block = 2^31
#define SBWT_GROUP_BASES 32         // bases per packed read group

#ifndef SBWT_FUSED_SORT_DEFAULT
#define SBWT_FUSED_SORT_DEFAULT 3632       // waves wait for 48 busy lanes | 14 << 8: the followers' finished reads go to the searchers while more than 40 reads wait in the followers' ring
#endif
struct SbwtIndexView {
    const uint4 *blocks;            // n_blocks * 4 quads
    const longlong2 *ptab;          // device prefix table, depth p_dev (nullptr if p_dev == 0)
    const unsigned long long *mega; // [4][n_mega]
    long long n_nodes;
    long long n_pos;                // positions of the path order: n_nodes, or more when paths were stitched (sbwt_derived.hip)
    long long C[4];
    int k;
    int p_dev;
    int n_mega;
    int has_ssup;
    int probe_len;                  // length of the certificate probes of k_search_cert (0 = off)
    const uint4 *stab;              // sparse prefix table (nullptr if p_sparse == 0)
    int p_sparse;                   // its depth (0 = none)
    unsigned n_sb;                  // its number of buckets (any: the hash is scaled to it)
    const unsigned *col, *pos;      // path order: column at path position t, path position of column v (nullptr = none)
    const uint4 *pq;                // packed path chars: quad t>>5 = { 32 chars (2 bits each), A, B }: go = ~A | B, safe = A & B,
                                    // only successor = ~A & B (k_path_reencode, sbwt_derived.hip)
    int has_safe;                   // the safe states are filled in (k_path_safe*)
    const uint4 *stab2;             // second-level sparse table for 31 < k <= 63 (nullptr = none): key = (first column of the
    unsigned n_sb2;                 // 31-prefix's interval, the remaining k-31 bases) -> the k-mer's path position (column
                                    // without a path order); its number of 32-byte buckets of two entries
    const uint4 *pfil;              // probe filter: blocked Bloom filter over the p_filter-mers of the index (nullptr = none)
    int p_filter, log2f;            // its depth and log2 of its number of 16-byte blocks
    const uint4 *trans;             // transition table: hashed 32-byte entries (k_trans_insert, sbwt_derived.hip)
    unsigned n_tslots;              // ... and their number (any: the hash is scaled to it)
    int stab_pos;                   // sparse entries are whole k-mers stored with their path position
    int debug;                      // experiments only: bit0 = skip result stores
    int out32 = 0;                  // results are written as int32 (the *_i32 entry points; n_nodes < 2^31): the result pointer is an int32 array
    int big = 0;                    // the image has the layout of 2^31 .. 2^32 columns (SbwtBlobHeader::big_layout): full 32-bit unsigned
                                    // columns and positions, no flag in bit 31 of any of them
    int fused_sort = 0;             // the fused kernel's SORT instantiation (lanes sorted by state; sbwt_search_fused.hip)
    int force_mega;                 // one mega block whose counts do not fit 32 bits (dense rank-only images): cnt is relative
                                    // to mega[c][0] although n_mega == 1
};

// Position-independent description of a blob (what index_export_header hands out).
struct SbwtBlobHeader {
    uint64_t magic;                 // 'SBWTGPU3'
    int64_t n_nodes, n_kmers, k, p_file, p_dev;
    int64_t C[4];
    int64_t n_blocks, n_mega;
    int64_t off_blocks, off_ptab, off_ftab, off_mega, blob_bytes;
    int32_t has_ssup;
    int32_t rank_only;              // the columns are not SBWT-consistent: only rank() is served
    int32_t ssup_derived;           // no suffix_group_starts given: marks derived on the device (internal use)
    int32_t p_sparse;               // depth of the sparse prefix table (0 = none)
    int64_t off_stab;
    int32_t big_layout;             // the derived structures hold full 32-bit UNSIGNED columns and positions (2^31 <= n < 2^32 - 2^24, or
                                    // "big_path" 2: forced, for tests): no stitched chains; sparse entries of depth < k carry no position
                                    // (SBWT_SP_UNIQ is bit 31); second-level entries hold position + 1 (0 = free) and no overflow flag (a
                                    // lookup goes on past a full bucket)
    int32_t has_path;               // path order present (col, pos, pq)
    int64_t off_col, off_pos, off_pq, off_trans;
    int32_t stab_pos;
    int32_t p_filter;               // depth of the probe filter (0 = none)
    int64_t off_pfil;
    int32_t log2f;
    int32_t has_safe;               // pq carries the substitution-safe bits
    int32_t force_mega;             // block counts are relative to mega[c][0] although n_mega == 1 (see SbwtIndexView)
    int64_t n_tslots;               // transition table: its number of 32-byte slots
    int64_t n_sb;                   // sparse prefix table: its number of 32-byte buckets
    int64_t n_pos;                  // positions of the path order (>= n_nodes: stitched chains repeat the columns of shared stretches)
    int64_t n_trans;                // ... and how many of them are in use
    int64_t n_paths;                // paths of the path order
    int64_t n_branch;               // columns with two or more successors (n_nodes / n_branch = columns between choices)
    int64_t image_level;            // 0 full, 1 no path order, 2 blocks + dense prefix table only
    int64_t row_ones[4];            // set bits of the rows A, C, G, T (select: valid j are 1 .. row_ones[c])
    int32_t log2b2_unused;
    int64_t n_sb2;                  // second-level sparse table: its number of 32-byte buckets (0 = none)
    int64_t off_stab2;
    int64_t path_lookahead;         // steps the path order looked ahead / behind when it chose successors (0: blind rule)
};
#define SBWT_BLOB_MAGIC 0x3355504754574253ull   // "SBWTGPU3" little endian (round 5: col[n_pos] = 0xFFFFFFFF; an image of an
                                                // earlier layout is refused by sbwtgpu_index_adopt)

// Sparse prefix table entry words
#define SBWT_SP_EMPTY (1ull << 63)              // the whole word of a free entry
#define SBWT_SP_OVERFLOW (1ull << 62)           // set in entry 0 of a bucket some key had to skip
#define SBWT_SP_UNIQ 0x80000000u                  // depth < k, payload word .w: the prefix has ONE column; the low 31 bits are its path position
#define SBWT_SP_MAX_DEPTH 31                    // keys are 2 bits per base in the low 62 bits
#define SBWT_SP_HASH 0x9E3779B97F4A7C15ull
// second-level sparse table hash: (first column of the prefix's interval, remaining bases)
static __host__ __device__ inline unsigned long long sp2_hash(unsigned origin, unsigned long long key2) {
    unsigned long long h = (key2 ^ ((unsigned long long)origin << 17)) * SBWT_SP_HASH;
    h ^= h >> 31;
    return (h + origin) * 0xD6E8FEB86659FD93ull;
}
#define SBWT_SP2_USED 0x80000000u               // in a second-level entry's payload word (.w)
#define SBWT_SP2_OVERFLOW 0x80000000u           // in entry 0's origin word (.z): some key had to skip this bucket
// probe filter hash (one multiply): block index in the top bits, two bit positions (7 bits each) from the
// well-mixed middle of the product (returned in the low 14 bits of sbwt_pf_bits)
static __host__ __device__ inline unsigned long long sbwt_pf_hash(unsigned long long key) { return key * SBWT_SP_HASH; }
static __host__ __device__ inline unsigned sbwt_pf_bits(unsigned long long h) { return (unsigned)(h >> 24) ^ (unsigned)(h >> 41); }

// Workspace header (first 256 bytes of the search workspace).
struct SbwtWorkHeader {
    unsigned long long ticket;      // next read to hand out
    int status;                     // 0 or SBWTGPU_ERR_NOT_SINGLETON
    int rg_sample;                  // reads the check kernel sampled for rg_long (0: ragged batches go the two-pass route)
    // work done by the last search launch (for the roofline's algorithmic-byte accounting)
    unsigned long long n_stream;    // streaming one-step extensions (SBWT.hh:562-575)
    unsigned long long n_search;    // full searches started (SBWT.hh:389-415)
    unsigned long long n_lf;        // interval updates executed past the device prefix table (SBWT.hh:430-431)
    unsigned long long n_tab_hit;   // prefix-table lookups that returned a non-empty interval
    unsigned long long n_ext;       // k-mers answered along path runs (k_search_cert<PATH>), not counted in n_stream
    unsigned long long n_bridge;    // substitutions bridged by the path's safe bits (no probes)
    // set by k_check_uniform before the search: all reads have one length and the result ranges one stride
    // (u_bad == 0): read r starts at u_read0 + r*u_len, its results at u_out0 + r*u_stride -- no offset fetch needed
    unsigned long long u_bad;
    long long u_read0, u_len, u_out0, u_stride;
    // the fused route (k_search_fused, sbwt_search_fused.hip): reads it handed on to the general kernel that runs behind it
    // (a base that is not upper-case ACGT), and that kernel's own ticket counter
    unsigned long long n_deferred;
    unsigned long long ticket2;
    unsigned long long n_pieces;    // pieces of long reads planned for this launch (SbwtPieceTab)
    unsigned long long pad[15];
    unsigned long long rg_long;     // of the sampled reads, those too long for the fused kernel
    // the fused route's TICKET TABLE (round 6; SbwtTickTab): a batch with many reads of more than three pieces is cut into
    // tickets of <= 160 bases listed in a table -- how many there are, and whether the table was too small for them
    unsigned long long n_ftick;
    unsigned long long ftick_over;
    unsigned long long pad2[5];
    // Which instantiation of the fused kernel suits the caller's batches (k <= 31): the LAST word of the header is not cleared
    // with it (SBWT_WS_CLEAR_BYTES) -- it is what the calls before on this workspace left there.  SBWT_HINT_MAGIC | c: the last c
    // calls in a row (counted up to SBWT_HINT_CALLS) had reads that followed their paths for at least SBWT_HINT_RATIO k-mers per
    // search started, the work mix the SORT instantiation is faster on (by 3-14 %; on batches that mostly search -- unrelated
    // reads, 5 % substitutions -- it is a third slower: half of its waves only follow paths).  c == SBWT_HINT_CALLS: the sorted
    // kernel; anything else (a fresh workspace, batches of alternating kinds): the unsorted one.  Both are launched; the one
    // whose call it is not returns at once.
    unsigned long long hint;
};
#define SBWT_WS_CLEAR_BYTES (sizeof(SbwtWorkHeader) - 8)

// Long reads on the device.  One lane walks one read, so a read of more than 2 * piece k-mers is cut into pieces of
// about `piece` k-mers (128 for small batches, so that a few genomes fill the chip, .. SBWT_PIECE for large ones, so that
// the table stays small) that lanes take like reads of their own: ticket n_reads + z is piece z.  The check kernel that
// runs ahead of every search lists the pieces' zones (k_check_uniform*), k_piece_bounds turns each zone into
// pairs[z] = { first base, one past the last base } (global base offsets, like read_off[r], read_off[r + 1]) and
// outs[z] = { first result slot, - }.  A piece starts at a k-mer whose window holds no lower-case acgt: for such a k-mer
// the reference's result does not depend on what came before it (SBWT.hh:565-566 vs :427), so the results tile the read's.
#define SBWT_PIECE 1024
struct SbwtPieceTab {
    uint4 *pairs = nullptr;         // nullptr: no splitting
    uint4 *outs = nullptr;          // (the check kernel parks { read, zone, zones of the read } here first)
    long long cap = 0;              // entries: total_bases / piece + 2 always suffice
    int piece = SBWT_PIECE;         // k-mers per zone
};
static_assert(sizeof(SbwtWorkHeader) == 320, "workspace header is 320 bytes");
// The fused kernel's ticket table (k_fused_tickets, sbwt_search_fused.hip): ticket t is the piece of <= 160 bases that starts at
// base tick[t].{x, y & 0xFFFF} (48 bits) and holds tick[t].y >> 16 bases; its results go to slot tick[t].{z, w} (64 bits) on;
// tick_read[t] is its read (for the list of reads handed on).  defer_bits: one bit per read, set when the fused kernel hands the
// read on (the zones of a long read are searched by the kernel behind only then).
struct SbwtTickTab {
    uint4 *tick = nullptr;          // nullptr: no table
    unsigned *tick_read = nullptr;
    unsigned *defer_bits = nullptr;
    long long cap = 0;              // entries
};
#define SBWT_HINT_MAGIC 0x5B377A00u
#define SBWT_HINT_RATIO 12
#define SBWT_HINT_CALLS 2

// launchers implemented in sbwt_search.hip, sbwt_api_kernels.hip, sbwt_derived.hip, sbwt_format.hip (all asynchronous on `stream`)
void sbwt_launch_encode(const char *d_bases, long long total_bases, uint4 *d_packed, SbwtWorkHeader *ws,
                        hipStream_t stream);
void sbwt_launch_search(const SbwtIndexView &ix, const uint4 *d_packed, const long long *d_read_off,
                        const long long *d_out_off, long long *d_out, long long n_reads, SbwtWorkHeader *ws,
                        int streaming, hipStream_t stream, int variant, long long total_groups, void *d_sort_scratch,
                        long long sort_scratch_bytes, int sort_key_bits, SbwtPieceTab pt);
// the fused route for batches of equal-length reads (sbwt_search_fused.hip): check + fused kernel + (for what it hands on)
// selective encode + the general path kernel.  d_defer: room for one 32-bit entry per read.
void sbwt_launch_search_fused(const SbwtIndexView &ix, const char *d_bases, long long total_bases, uint4 *d_packed,
                              const long long *d_read_off, const long long *d_out_off, long long *d_out, long long n_reads,
                              SbwtWorkHeader *ws, int streaming, hipStream_t stream, unsigned *d_defer,
                              hipEvent_t ev_begin, hipEvent_t ev_end, SbwtPieceTab pt, int ragged_ok, SbwtTickTab tt = SbwtTickTab());
void sbwt_launch_encode_chained(const char *d_bases, long long total_bases, uint4 *d_packed, SbwtWorkHeader *ws,
                                const unsigned *d_defer, const long long *d_read_off, int k, hipStream_t stream);
void sbwt_launch_search_chained(const SbwtIndexView &ix, const uint4 *d_packed, const long long *d_read_off,
                                const long long *d_out_off, long long *d_out, long long n_reads, SbwtWorkHeader *ws,
                                int streaming, hipStream_t stream, const unsigned *d_defer, SbwtPieceTab pt);
// turns the zones the check kernel listed into pieces (after the bases are packed, before the search kernel)
void sbwt_launch_piece_bounds_tt(const uint4 *d_packed, const long long *d_read_off, const long long *d_out_off, int k,
                                 SbwtWorkHeader *ws, SbwtPieceTab pt, int behind_fused, hipStream_t stream, const unsigned *defer_bits);
void sbwt_launch_piece_bounds(const uint4 *d_packed, const long long *d_read_off, const long long *d_out_off, int k,
                              SbwtWorkHeader *ws, SbwtPieceTab pt, int behind_fused, hipStream_t stream);
long long sbwt_sort_scratch_bytes(long long n_reads, int key_bits);
const unsigned *sbwt_launch_sort_reads(const SbwtIndexView &ix, const uint4 *d_packed, const long long *d_read_off,
                                       long long n_reads, const SbwtWorkHeader *ws, void *d_scratch, long long scratch_bytes,
                                       int key_bits, hipStream_t stream);
void sbwt_launch_rank(const SbwtIndexView &ix, const long long *d_pos, const char *d_sym, long long n,
                      long long *d_out, hipStream_t stream);
void sbwt_launch_precalc(const SbwtIndexView &ix, int p, longlong2 *d_table, hipStream_t stream);
void sbwt_launch_update_interval(const SbwtIndexView &ix, const char *d_bases, const long long *d_off, long long n,
                                 long long *d_first, long long *d_second, hipStream_t stream);
void sbwt_launch_forward(const SbwtIndexView &ix, const long long *d_node, const char *d_sym, long long n,
                         long long *d_out, hipStream_t stream);
void sbwt_launch_partial_search(const SbwtIndexView &ix, const char *d_bases, const long long *d_off, long long n,
                                long long *d_first, long long *d_second, long long *d_matched, hipStream_t stream);
void sbwt_launch_get_kmer(const SbwtIndexView &ix, const long long *d_colex, long long n, char *d_out,
                          hipStream_t stream);
void sbwt_launch_select(const SbwtIndexView &ix, const long long *d_j, const char *d_sym, long long n,
                        const long long row_ones[4], long long *d_out, hipStream_t stream);
long long sbwt_format_scratch_bytes(long long n_reads);
void sbwt_launch_format(const long long *d_vals, const long long *d_out_off, long long n_reads, char *d_text,
                        long long *d_line_off, void *d_scratch, hipStream_t stream);
long long sbwt_blocks_scratch_bytes(long long n_nodes);
int sbwt_blocks_count(const unsigned long long *d_bits, long long n_nodes, void *d_scratch, long long totals[5], hipStream_t st);
void sbwt_blocks_fill(const unsigned long long *d_bits, const unsigned long long *d_ssup, long long n_nodes, void *d_scratch,
                      const long long C[4], int use_mega, int n_mega, uint4 *d_blocks, unsigned long long *d_mega, hipStream_t st);
long long sbwt_derive_scratch_bytes(long long n_nodes);
void sbwt_launch_derive_marks(const SbwtIndexView &ix, uint4 *d_blocks, void *d_scratch, hipStream_t stream);
long long sbwt_sparse_scratch_bytes(long long n_nodes);
long long sbwt_path_scratch_bytes(long long n_nodes);
long long sbwt_path_quads(long long n_nodes);
long long sbwt_count_paths(const SbwtIndexView &ix, hipStream_t stream);
void sbwt_launch_path_safe(const SbwtIndexView &ix, uint4 *d_pq, int rule, void *d_hlab, unsigned char *d_alt_safe,
                           hipStream_t stream);
long long sbwt_path_safe_scratch_bytes(long long n_pos, int k);
long long sbwt_launch_path_oth(const SbwtIndexView &ix, uint4 *d_pq, long long *n_branch, hipStream_t stream, int count_only = 0);
void sbwt_launch_trans_insert(const SbwtIndexView &ix, uint4 *d_trans, long long n_slots, const unsigned char *d_alt_safe,
                              hipStream_t stream);
int sbwt_launch_build_path(const SbwtIndexView &ix, unsigned *d_col, unsigned *d_pos, uint4 *d_pq, long long pos_cap,
                           long long *n_pos, int stitch, int min_copy, void *d_scratch, int lookahead, hipStream_t stream);
int sbwt_launch_build_sparse(const SbwtIndexView &ix, int p_dense, int p_sparse, long long n_buckets, uint4 *d_table,
                             void *d_scratch, const unsigned *d_pos, int p_filter, int log2f, uint4 *d_filter,
                             long long n_entries2, uint4 *d_table2, hipStream_t stream);

// device builder (sbwt_build.hip): phase A = text -> sorted distinct k-mers, edges, predecessor-less k-mers; phase B =
// dummies + k-mers -> the five rows in host memory
struct SbwtBuildState {
    int k = 0, rc = 0, ssup = 0;
    int key_bytes = 8;                           // 8: k <= 32 (one 64-bit word per k-mer), 16: 32 < k <= 64 (__uint128_t)
    long long n_text = 0, nk = 0, ng = 0, n_nopred = 0;
    void *km = nullptr;                          // sorted distinct k-mers (key_bytes each)
    unsigned *edges = nullptr;                   // per k-mer (set on suffix-group starts)
    void *nopred_keys = nullptr;                 // the k-mers without a predecessor
};
int sbwt_build_phase_a(const char *h_text, long long n_text, int k, int rc, SbwtBuildState *S, hipStream_t st);
int sbwt_build_copy_nopred(const SbwtBuildState *S, void *h_keys);
int sbwt_build_phase_b(SbwtBuildState *S, const void *h_ddata, const unsigned *h_dedges, long long nd, int ssup,
                       unsigned long long *h_rows, hipStream_t st);
void sbwt_build_release(SbwtBuildState *S);
