/*
 * sbwtgpu.h -- C ABI of the MI355X (gfx950) k-mer search path for plain-matrix SBWT indexes.
 *
 * This is the drop-in boundary: the functions below are what a binding in the reference
 * (algbio/SBWT, C++17) would call instead of running its CPU loop.  Each entry point cites
 * the reference interface it replaces (paths relative to the reference repository root).
 * INTEGRATION.md shows the reference-side glue.
 *
 * Conventions
 *   - plain C, no C++/torch types; all sizes/ranks are int64_t like the reference's API.
 *   - return value: 0 = ok, negative = error (codes below); sbwtgpu_last_error() gives the
 *     message (thread-local).  Nothing throws or aborts across this boundary.
 *   - ownership: handles are owned by the library (destroy them); every buffer is owned by
 *     the caller and only borrowed for the duration of the call (index_create copies).
 *   - threading: a handle is immutable after creation; concurrent query calls on one handle
 *     are allowed (host-buffer calls use a private stream each).
 *   - "host" entry points take host pointers and do H2D/D2H themselves; "_dev" entry points
 *     take device pointers on the index's device plus a hipStream_t (as void*), enqueue
 *     asynchronously and never synchronise.
 *   - bit vectors are arrays of little-endian uint64_t words, bit i of the vector is bit
 *     (i % 64) of word i / 64 -- the sdsl::bit_vector convention used by the reference.
 */
#ifndef SBWTGPU_H
#define SBWTGPU_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SBWTGPU_OK                      0
#define SBWTGPU_ERR_INVALID_ARG        -1
#define SBWTGPU_ERR_NO_DEVICE          -2   /* no usable HIP device / wrong device index */
#define SBWTGPU_ERR_HIP                -3   /* a HIP runtime call failed */
#define SBWTGPU_ERR_NO_STREAMING       -4   /* "Error: streaming search support not built" (SBWT.hh:546-547) */
#define SBWTGPU_ERR_PRECALC_TOO_LONG   -5   /* precalc > 20 (SBWT.hh:619-621) */
#define SBWTGPU_ERR_PRECALC_GT_K       -6   /* precalc > k  (SBWT.hh:623-624) */
#define SBWTGPU_ERR_NOT_SINGLETON      -7   /* "Bug: k-mer search did not give a singleton interval" (SBWT.hh:410-413) */
#define SBWTGPU_ERR_OOM                -8
#define SBWTGPU_ERR_READ_TOO_LONG      -9   /* a single read of >= 2^31 bases */

typedef struct sbwtgpu_index sbwtgpu_index;

/* What sbwt::SBWT<SubsetMatrixRank<sdsl::bit_vector, sdsl::rank_support_v5<>>> holds
 * (include/sbwt/SBWT.hh:36-45, include/sbwt/SubsetMatrixRank.hh:19-23).  Rank supports
 * are not passed: the library builds its own device-side directory from the bits. */
typedef struct {
    int64_t         n_nodes;      /* number of columns (SBWT.hh:43); > 0 (the root always exists) */
    const uint64_t *A_bits;       /* ceil(n_nodes/64) words each */
    const uint64_t *C_bits;
    const uint64_t *G_bits;
    const uint64_t *T_bits;
    const uint64_t *suffix_group_starts; /* NULL => no streaming support (SBWT.hh:253) */
    int64_t         k;            /* SBWT.hh:45 */
    int64_t         n_kmers;      /* SBWT.hh:44 (informational) */
    int64_t         precalc_k;    /* SBWT.hh:41; 0 => no precalc table */
    const int64_t  *precalc;      /* 4^precalc_k (first,second) pairs as loaded from an index
                                     file (SBWT.hh:40,510), or NULL => computed on the device
                                     (do_kmer_prefix_precalc, SBWT.hh:616-645) */
} sbwtgpu_index_desc;

typedef struct {
    int64_t n_nodes, n_kmers, k, precalc_k;
    int64_t C[4];                 /* SBWT.hh:344-349 */
    int32_t has_streaming_support;
    int32_t device;
    int64_t device_precalc_k;     /* depth of the library's own device-side prefix table (>= precalc_k) */
    int64_t blob_bytes;           /* size of the device image (what index_bcast moves) */
    int64_t image_level;          /* what the image holds: 0 = all derived structures, 1 = no path order / transition
                                     table, 2 = blocks + dense prefix table only (see "image_level" below) */
    int64_t n_paths;              /* path order: number of paths (0 without one) */
    int64_t n_branch;             /* columns with two or more successors */
    int64_t default_search_variant; /* the kernel "search_variant" = -1 picks for this index (see below) */
} sbwtgpu_index_info;

/* ---- library ---- */
const char *sbwtgpu_version(void);
const char *sbwtgpu_last_error(void);
int         sbwtgpu_device_count(int *count);
/* Process-wide tuning knobs for experiments (results never depend on them):
 *   "search_variant"  -1 (default) = by the index: 5 on an index with a path order, else the blocks kernel.  5 = the fused route
 *                     (k_search_fused re-encodes the bases itself and takes batches of reads of up to 160 bases; the
 *                     general kernel runs behind it for what it hands on or declines); 4 = always two passes (encode +
 *                     k_search_cert along the path order, per-read segment lists written by the whole wave); 1 =
 *                     k_search_cert on the blocks only; 0 = k_search (the reference's order of searches; cross-checks)
 *   "fused_ragged"    1 (default): the fused route also takes batches of reads of different lengths (it fetches their
 *                     offsets); 0: only batches of reads of one length (SBWTGPU_FUSED_RAGGED)
 *   "fused_pieces"    1 .. 3: the fused route takes a read of more than 160 bases as up to this many pieces of 160 bases
 *                     that overlap by k-1 (tickets are (read, piece)); 1: such reads go to the general kernel.  -1 / 0
 *                     (default) = 3 (k = 63, 250-base reads: 115 -> 243 G k-mers/s; k = 30: 228 -> 235 G since round 5).
 *                     SBWTGPU_FUSED_PIECES.
 *   "split_long"      1 (default): the device entry points cut reads of more than two pieces' worth of k-mers (a piece:
 *                     128 .. 1024 k-mers by batch size) into pieces that lanes take separately, at k-mers whose result
 *                     does not depend on history; 0: one lane per read whatever its length (SBWTGPU_SPLIT_LONG)
 *   "kernel_events"   1: HIP events around the dominant kernel of every search call (sbwtgpu_kernel_times)
 *   "sort_reads"      1: the path-order kernels take the reads sorted by where they start in the path order (a lookup of one
 *                     k-mer per read + a radix sort of (position, read) pairs before the search; lanes of a wave then
 *                     share lines of the index); default 0: the pre-pass costs more than it saves while the path order
 *                     numbers its paths in column order.  Reads that ARRIVE sorted by genome
 *                     position need no switch and run 18 % faster.  Set it before sizing workspaces.  SBWTGPU_SORT_READS.
 *   "probe_len"       length of the certificate probes (-1 = automatic, 0 = off)
 *   "derive_ssup"     1 (default): indexes created without suffix_group_starts get the marks derived on
 *                     the device so that the per-k-mer search loop can use streaming steps internally
 *   "debug"           kernel experiment bits (0 = product behaviour); fused kernel: 32 = no anchors / seeds / resumed compares,
 *                     64 = the k > 31 walk (F_CMP) for every k
 *   "poison_results"  1: every search first fills its result range with 0xA5 (parity tests)
 *   "trans_ext", "trans_wide"   accepted and ignored (round-2 table formats)
 * Read when an index is CREATED (derived acceleration structures inside the device image; environment
 * variables of the same meaning: SBWTGPU_SPARSE_PRECALC, SBWTGPU_PROBE_FILTER, SBWTGPU_PATH_ORDER):
 *   "sparse_depth"    depth of the sparse (hashed) prefix table, 0 = none, default 31 (capped at k)
 *   "probe_filter"    1 (default): Bloom filter over the probe_len-mers of the index for the certificate probes
 *   "path_order"      1 (default): path order + transition table (indexes with suffix-group marks, given or derived)
 *   "big_path"        1 (default): an index of 2^31 .. 2^32 - 2^24 columns with k <= 31 gets the full image too -- columns and
 *                     path positions as 32-bit unsigned values, read by the fused kernel's BIG instantiation (2.25 x 10^9
 *                     columns: 120 GB, 207 G k-mers/s); 0: such an index gets blocks + dense table only, as before round 5
 *                     (23 G k-mers/s).  k > 31 beyond 2^31 columns always steps down.  SBWTGPU_BIG_PATH.
 *   "image_level"     0 (default): the image carries every derived structure (path order + transition table, sparse
 *                     prefix table, probe filter: 43-53 bytes per column for k <= 31, 92 for k = 63); 1: no path order; 2:
 *                     blocks + dense prefix table only (1 byte per column + the table).  Results are the same at every
 *                     level; throughput is not.  SBWTGPU_IMAGE_LEVEL.  A step-down that index_create makes by
 *                     itself (memory) is announced with one line on stderr.
 *   (environment only) SBWTGPU_SPARSE_BUCKETS_PCT: two-entry buckets of the sparse tables per 100 columns (default: 100 for
 *                     k <= 31, 125 for 31 < k <= 63); SBWTGPU_FILTER_LOG2_ADJ: +1 doubles, -1 halves the probe filter (default: 128-bit
 *                     blocks of 8-16 windows, up to 20 where that keeps a filter of more than 128 MB small); SBWTGPU_DEVICE_PRECALC: depth of the dense device prefix table (default:
 *                     log4 n, at most 8 on an image with sparse table and filter, at most 14 otherwise); SBWTGPU_VERBOSE=1 / 2:
 *                     index_create names the parts of the image it builds, with their times, on stderr.
 *   "max_image_bytes" > 0: index_create moves to the next level while the image would be larger than this (and fails
 *                     with SBWTGPU_ERR_OOM if level 2 is still larger); it also steps down by itself when device
 *                     memory runs out.  SBWTGPU_MAX_IMAGE_BYTES.
 *   "force_mega"      1: rank-only images (arbitrary bit vectors) store their block counts relative to a 64-bit base as
 *                     images whose counts pass 2^32 do (tests of that layout at small sizes); default 0
 *   "sort_reads"      1: the path-order kernels take the reads in the order of their first k-mer's path position (a lookup
 *                     and a radix sort per batch inside the caller's workspace; pays only when nothing upstream orders the
 *                     reads and the index is large); default off (SBWTGPU_SORT_READS)
 *   "path_lookahead"  8 (default): how many steps ahead / behind the path order looks for branch points when it chooses
 *                     which successor a column's path takes (paths follow the core of a pan-genome); 0: blind choice
 *                     (SBWTGPU_PATH_LOOKAHEAD)
 *   "path_safe"       substitution-safe bits along the paths (k <= 31; SBWTGPU_PATH_SAFE): 2 (default) wherever the next k
 *                     steps lie on the path, 1 only where the k steps before do too (the first rule), 0 none
 *   "path_stitch"     1 (default): the vertex-disjoint paths are joined into chains through copies of the stretches that
 *                     strains share (at most n/5 copied positions; a graph whose copies would repeat more than n/32
 *                     branching columns keeps its disjoint paths); 0: disjoint paths (SBWTGPU_PATH_STITCH)
 *   "path_stitch_min" shortest stretch worth a copy, default 1 (SBWTGPU_PATH_STITCH_MIN) */
int         sbwtgpu_set_tuning(const char *key, int64_t value);

/* ---- index life cycle ---- */
/* Replaces SBWT(A,C,G,T,ssup,k,n_kmers,precalc_k) ctor (SBWT.hh:335-353) and the tail of
 * SBWT::load (SBWT.hh:500-516): builds the device image (interleaved 64-column blocks with
 * rank counts), the C array, and the prefix table(s), on HIP device `device`. */
int  sbwtgpu_index_create(const sbwtgpu_index_desc *desc, int device, sbwtgpu_index **out);
void sbwtgpu_index_destroy(sbwtgpu_index *idx);
int  sbwtgpu_index_get_info(const sbwtgpu_index *idx, sbwtgpu_index_info *info);
/* get_precalc() (SBWT.hh:131): copies the 4^precalc_k (first,second) pairs to host memory. */
int  sbwtgpu_index_get_precalc(const sbwtgpu_index *idx, int64_t *out_pairs);

/* ---- construction on the device ---- */
/* The plain-matrix SBWT of a set of sequences, built on the GPU: what the reference's constructors produce
 * (build_nodeboss_in_memory, include/sbwt/NodeBOSSInMemoryConstructor.hh:98-213; the KMC-based SBWT(config) ctor,
 * SBWT.hh:300-332, gives the same bits): the four rows A, C, G, T and suffix_group_starts as sdsl-ordered words.
 * k-mers with anything but upper-case ACGT are skipped (:156-159); add_revcomp adds every reverse complement
 * (src/CLI/sbwt_build.cpp:108-123).  2 <= k <= 64 (a k-mer is packed into 64 or 128 bits); for longer k the C++ host
 * mirror (sbwt::build_plain_matrix_bits) builds on the CPU.  Release with sbwtgpu_free_plain_matrix(). */
typedef struct {
    int64_t   n_nodes, n_kmers, k;
    uint64_t *A_bits, *C_bits, *G_bits, *T_bits;   /* ceil(n_nodes/64) words each */
    uint64_t *suffix_group_starts;                 /* NULL when not requested */
} sbwtgpu_plain_matrix_bits;
int  sbwtgpu_build_plain_matrix(const char *const *seqs, const int64_t *seq_len, int64_t n_seqs, int64_t k,
                                int add_revcomp, int build_streaming_support, int device,
                                sbwtgpu_plain_matrix_bits *out);
void sbwtgpu_free_plain_matrix(sbwtgpu_plain_matrix_bits *bits);

/* ---- multi-GPU replication (no reference equivalent) ---- */
/* One process per GPU (torch.distributed / any launcher): rank 0 exports the device image,
 * the launcher broadcasts header (host bytes) and blob (device bytes, e.g. RCCL broadcast over
 * xGMI), every other rank adopts the received bytes.  The blob is position independent. */
int  sbwtgpu_index_export_header(const sbwtgpu_index *idx, void *header_out, int64_t header_cap,
                                 int64_t *header_bytes);
int  sbwtgpu_index_blob(const sbwtgpu_index *idx, void **dev_ptr, int64_t *bytes);
/* Copies the device image into caller-owned device memory (`bytes` must equal blob_bytes),
 * asynchronously on `stream` -- e.g. into the tensor a launcher hands to its broadcast. */
int  sbwtgpu_index_copy_blob(const sbwtgpu_index *idx, void *dst_dev, int64_t bytes, void *stream);
/* Adopts a caller-owned device blob (must stay alive and unchanged while the handle lives). */
int  sbwtgpu_index_adopt(const void *header, int64_t header_bytes, void *dev_blob, int64_t blob_bytes,
                         int device, sbwtgpu_index **out);
/* Single-process form used by the C++ CLI (--gpus N): replicates `root` onto the listed
 * devices with one RCCL ncclBroadcast; out[i] receives the handle for devs[i] (out[i] == root
 * where devs[i] is the root's device).  devs[] may name a device more than once: the image is sent once
 * per distinct device and the duplicates receive the SAME handle (destroy every distinct handle once). */
int  sbwtgpu_index_bcast(sbwtgpu_index *root, int n_dev, const int *devs, sbwtgpu_index **out);

/* ---- queries, host buffers ---- */
/* SubsetMatrixRank::rank(pos, c) (SubsetMatrixRank.hh:31-37) for n (pos, sym) pairs;
 * pos in [0, n_nodes]; sym is the ASCII char; non-ACGT (upper case) => 0. */
int  sbwtgpu_rank_batch(const sbwtgpu_index *idx, const int64_t *pos, const char *sym, int64_t n,
                        int64_t *out);
/* SBWT::streaming_search(const char*, int64_t) (SBWT.hh:544-581) for n_reads reads:
 * read r = bases[read_off[r] .. read_off[r+1]); its max(0, len-k+1) results are written to
 * out[out_off[r] ..] (out_off[r+1]-out_off[r] must equal that count).  -1 = not found.
 * Returns SBWTGPU_ERR_NO_STREAMING where the reference throws. */
int  sbwtgpu_streaming_search_batch(const sbwtgpu_index *idx, const char *bases, const int64_t *read_off,
                                    int64_t n_reads, int64_t *out, const int64_t *out_off);
/* The non-streaming loop of run_queries_not_streaming (src/CLI/sbwt_search.cpp:78-84):
 * out[out_off[r] + i] = SBWT::search(read_r + i) (SBWT.hh:389-415).  Works with or without
 * streaming support. */
int  sbwtgpu_search_batch(const sbwtgpu_index *idx, const char *bases, const int64_t *read_off,
                          int64_t n_reads, int64_t *out, const int64_t *out_off);
/* The same two calls with int32 results (result compaction; consumer: print_vector,
 * src/CLI/sbwt_search.cpp:21-43, which only prints the values): the kernels write int32 themselves, so a result costs
 * 4 bytes of HBM writes and of PCIe instead of 8.  Only for indexes of fewer than 2^31 columns (SBWTGPU_ERR_INVALID_ARG otherwise); -1
 * stays -1.  out[] is indexed like the int64 calls' (out_off in results, not bytes). */
int  sbwtgpu_streaming_search_batch_i32(const sbwtgpu_index *idx, const char *bases, const int64_t *read_off,
                                        int64_t n_reads, int32_t *out, const int64_t *out_off);
int  sbwtgpu_search_batch_i32(const sbwtgpu_index *idx, const char *bases, const int64_t *read_off,
                              int64_t n_reads, int32_t *out, const int64_t *out_off);
/* SBWT::update_sbwt_interval(S, len, I) (SBWT.hh:422-437) for n independent queries:
 * query q extends interval (first[q], second[q]) by bases[off[q] .. off[q+1]) in place. */
int  sbwtgpu_update_interval_batch(const sbwtgpu_index *idx, const char *bases, const int64_t *off,
                                   int64_t n, int64_t *first, int64_t *second);
/* SBWT::forward(node, c) (SBWT.hh:368-381) for n (node, sym) pairs. */
int  sbwtgpu_forward_batch(const sbwtgpu_index *idx, const int64_t *node, const char *sym, int64_t n,
                           int64_t *out);
/* SBWT::partial_search(input, len) (SBWT.hh:525-537) for n queries: query q = bases[off[q] .. off[q+1]); first/second
 * receive the interval of the longest matched prefix (every char upper-cased first, :529), matched its length. */
int  sbwtgpu_partial_search_batch(const sbwtgpu_index *idx, const char *bases, const int64_t *off, int64_t n,
                                  int64_t *first, int64_t *second, int64_t *matched);
/* SBWT::get_kmer / get_kmer_fast (SBWT.hh:700-746) for n columns: out[q*k .. q*k+k) = the k-mer of column
 * colex_rank[q] ('$'-padded on the left for dummy columns), not NUL-terminated. */
int  sbwtgpu_get_kmer_batch(const sbwtgpu_index *idx, const int64_t *colex_rank, int64_t n, char *out);
/* SubsetMatrixSelectSupport::select(j, c) (SubsetMatrixSelectSupport.hh:27-33) for n (j, sym) pairs: the column
 * holding the j-th set bit (1-based) of row sym; non-ACGT => 0.  j must be in [1, ones of the row]. */
int  sbwtgpu_select_batch(const sbwtgpu_index *idx, const int64_t *j, const char *sym, int64_t n, int64_t *out);

/* ---- queries, device buffers (asynchronous on `stream`) ----
 * The device entry points cannot look at the offsets: the caller guarantees that read_off/out_off are
 * non-decreasing, that out_off matches max(0, len-k+1) per read, that no read has 2^31 or more bases
 * and that one call carries fewer than 2^36 bases. */
/* Scratch the search kernels need: a work-queue header, the 2-bit re-encoding of the bases (total_bases/2 + 64
 * bytes), the list of reads the fused kernel hands on (total_bases/8), the pieces of long reads (total_bases/32, at
 * least 8 MB for batches of more than 32 Mbases) and, while "sort_reads" is on, room to sort the reads (about one more
 * byte per base).  Non-decreasing in total_bases: the caller allocates it once for its largest batch and may reuse it
 * across calls on the same stream.  Calls on DIFFERENT streams may run concurrently on one handle when each has its own
 * workspace (and result range): two batches in flight keep the chip full while the waves of the earlier launch leave one
 * by one -- +6 % on 10 M-read batches, +20 % on 1 M-read batches. */
int64_t sbwtgpu_search_workspace_bytes(int64_t total_bases);
int  sbwtgpu_streaming_search_dev(const sbwtgpu_index *idx, const char *d_bases, int64_t total_bases,
                                  const int64_t *d_read_off, int64_t n_reads, int64_t *d_out,
                                  const int64_t *d_out_off, void *d_workspace, int64_t workspace_bytes,
                                  void *stream);
int  sbwtgpu_search_dev(const sbwtgpu_index *idx, const char *d_bases, int64_t total_bases,
                        const int64_t *d_read_off, int64_t n_reads, int64_t *d_out,
                        const int64_t *d_out_off, void *d_workspace, int64_t workspace_bytes,
                        void *stream);
/* The same two calls with int32 results on the device (d_out is an int32 array indexed by d_out_off, in results): every kernel
 * of the route writes 4 bytes per k-mer, half the write requests of a launch (a third of its time: +10-13 % k-mers/s on
 * BASELINE config 2, 226 -> 256 G).  Indexes of fewer than 2^31 columns only (SBWTGPU_ERR_INVALID_ARG otherwise); -1 stays -1. */
int  sbwtgpu_streaming_search_dev_i32(const sbwtgpu_index *idx, const char *d_bases, int64_t total_bases,
                                      const int64_t *d_read_off, int64_t n_reads, int32_t *d_out,
                                      const int64_t *d_out_off, void *d_workspace, int64_t workspace_bytes,
                                      void *stream);
int  sbwtgpu_search_dev_i32(const sbwtgpu_index *idx, const char *d_bases, int64_t total_bases,
                            const int64_t *d_read_off, int64_t n_reads, int32_t *d_out,
                            const int64_t *d_out_off, void *d_workspace, int64_t workspace_bytes,
                            void *stream);
int  sbwtgpu_rank_dev(const sbwtgpu_index *idx, const int64_t *d_pos, const char *d_sym, int64_t n,
                      int64_t *d_out, void *stream);
/* The two halves of the calls above, for callers that re-run a search on bases that are already
 * encoded or that want to time the kernels separately: (1) 2-bit re-encoding of the bases into
 * the workspace, (2) the search over an encoded workspace.  streaming != 0 selects
 * streaming_search, 0 the per-k-mer search loop. */
int  sbwtgpu_encode_bases_dev(const sbwtgpu_index *idx, const char *d_bases, int64_t total_bases,
                              void *d_workspace, int64_t workspace_bytes, void *stream);
int  sbwtgpu_search_encoded_dev(const sbwtgpu_index *idx, int64_t total_bases, const int64_t *d_read_off,
                                int64_t n_reads, int64_t *d_out, const int64_t *d_out_off,
                                void *d_workspace, int64_t workspace_bytes, int streaming, void *stream);
/* Synchronises `stream`, then reports the status word of the last search on this workspace
 * (0, or SBWTGPU_ERR_NOT_SINGLETON). */
int  sbwtgpu_workspace_status(const void *d_workspace, void *stream, int *status);
/* Synchronises `stream`, then reports the work the last search on this workspace performed:
 * stats[0] streaming one-step extensions, [1] full searches, [2] interval updates executed past the
 * device prefix table, [3] device prefix-table lookups that returned a non-empty interval, [4] k-mers
 * answered along path runs (streaming steps that needed no block access), [5] substitutions bridged by the
 * path's safe bits, [6..7] reserved (0). */
int  sbwtgpu_workspace_stats(const void *d_workspace, void *stream, int64_t stats[8]);
/* Measurement aid (bench.py's roofline leg): after sbwtgpu_set_tuning("kernel_events", 1), sbwtgpu_streaming_search_dev /
 * sbwtgpu_search_dev record a pair of HIP events on their stream around the dominant kernel (k_search_fused) of every
 * call, without synchronising; this waits for them and reports the durations in ms, oldest first (the last 256 calls
 * since the switch was set; *n = how many). */
int  sbwtgpu_kernel_times(double *ms, int64_t cap, int64_t *n);

/* ---- output formatting on the device ---- */
/* print_vector of src/CLI/sbwt_search.cpp:21-43 for a whole batch: one line per read, every value
 * followed by one space, '\n' per read, -1 as "-1", 0 as an empty token (the reference's behaviour).
 * d_line_off (n_reads+1 entries) receives the byte offset of every read's line; d_line_off[n_reads] is
 * the length of the text.  text_cap must be >= sbwtgpu_format_text_bound(). */
int64_t sbwtgpu_format_text_bound(const sbwtgpu_index *idx, int64_t n_values, int64_t n_reads);
int64_t sbwtgpu_format_scratch_bytes(int64_t n_reads);
int  sbwtgpu_format_results_dev(const sbwtgpu_index *idx, const int64_t *d_values, const int64_t *d_out_off,
                                int64_t n_reads, int64_t n_values, char *d_text, int64_t text_cap,
                                int64_t *d_line_off, void *d_scratch, int64_t scratch_bytes, void *stream);
/* The whole `sbwt search` inner loop for one batch of reads in host memory (run_queries_streaming /
 * run_queries_not_streaming, src/CLI/sbwt_search.cpp:46-91): searches every read (streaming != 0:
 * streaming_search, else the per-k-mer search loop) and returns the formatted output text.  Large
 * batches are cut into chunks that are pipelined over two HIP streams with pinned staging buffers
 * (H2D of chunk i+1 and D2H of chunk i-1 overlap the kernels of chunk i).  *text is malloc'ed by the
 * library: release it with sbwtgpu_free_host().  *n_queries receives the number of k-mers searched. */
int  sbwtgpu_search_text_batch(const sbwtgpu_index *idx, const char *bases, const int64_t *read_off,
                               int64_t n_reads, int streaming, char **text, int64_t *text_bytes,
                               int64_t *n_queries);
/* The same, streamed: `sink` is called from the calling thread with consecutive pieces of the output text, in order, each
 * straight out of a pinned staging buffer that is only valid during the call (what `sbwt search` writes to its output
 * file: no copy of the whole text is ever made).  A non-zero return of the sink aborts the call.
 * Concurrency: several host threads may call this (and every other search entry point) on ONE index at the same time --
 * the handle is immutable and a call owns its stream, its staging slots (three per call here: pinned text, pinned input and
 * device buffers sized by the batch; at most eight are parked between calls) and its workspace; `sbwt search` drives two
 * such calls in turn (SBWT_CLI_SEARCH_THREADS=1: one).  NOT thread-safe: sbwtgpu_set_tuning() and the measurement aids
 * behind it ("kernel_events" / sbwtgpu_kernel_times: one unsynchronised event ring per process) -- set them before the
 * threads start and keep "kernel_events" off while calls overlap. */
typedef int (*sbwtgpu_text_sink)(void *ctx, const char *text, int64_t bytes);
int  sbwtgpu_search_text_stream(const sbwtgpu_index *idx, const char *bases, const int64_t *read_off,
                                int64_t n_reads, int streaming, sbwtgpu_text_sink sink, void *sink_ctx,
                                int64_t *n_queries);
void sbwtgpu_free_host(void *p);
/* sbwtgpu_search_text_batch keeps its pinned staging and device buffers for the next call, and every host thread
 * keeps one small (1 MiB) pinned + device buffer pair and a stream per device for small host-buffer calls (the
 * reference's scalar API arrives as batches of one); this frees the parked buffers and the calling thread's. */
void sbwtgpu_release_cached_buffers(void);

#ifdef __cplusplus
}
#endif
#endif
