/*
 * sbwt_oracle.h -- CPU restatement of the reference's plain-matrix SBWT search path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under sbwt_amd/ (the product) may include,
 * link or call this.  Allowed users: tests/, __graft_entry__.smoke(), and the
 * `cpu_baseline` leg of bench.py.
 *
 * Parity status: PINNED for results (ranks / -1) against the reference's own
 * known-answer tests (tests/test_CLI.hh:90 exact string; tests/test_small.hh:290
 * n_nodes==9; tests/test_small.hh:101-126 partial_search; exhaustive 4^k checks of
 * tests/test_small.hh:24-43; streaming==search of tests/test_large.hh:104-115).
 * The reference itself cannot be compiled here (sdsl-lite / SeqIO / KMC submodules
 * are empty in /root/reference), so there is no oracle/_ref build; see DESIGN.md.
 * UNPINNED: the byte layout of sdsl's serialized rank_support_v5 (no golden index
 * file exists in the reference tree).
 *
 * Every function cites the reference file:line it restates (paths relative to
 * /root/reference).
 */
#ifndef SBWT_ORACLE_H
#define SBWT_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* A packed bit vector with an sdsl::rank_support_v5-shaped directory
 * (2048-bit superblocks: one absolute 64-bit count + five packed counts for the
 * 384-bit blocks inside).  [UPSTREAM-KNOWLEDGE for the shape; the mathematical
 * contract rank(i) = popcount(bits[0,i)) is what results depend on.] */
typedef struct {
    int64_t   n_bits;
    int64_t   n_words;   /* ceil(n_bits/64) (+1 guard word, zero) */
    uint64_t *words;
    int64_t   n_dir;     /* number of 64-bit directory words */
    uint64_t *dir;
} orc_bitvec;

/* Mirrors the private members of sbwt::SBWT<SubsetMatrixRank<...>>
 * (include/sbwt/SBWT.hh:36-45, include/sbwt/SubsetMatrixRank.hh:19-28). */
typedef struct {
    orc_bitvec col[4];        /* A_bits, C_bits, G_bits, T_bits (+ their rank supports) */
    orc_bitvec ssup;          /* suffix_group_starts; n_bits == 0 => no streaming support */
    int64_t    C[4];          /* SBWT.hh:344-349 */
    int64_t   *precalc;       /* 4^precalc_k pairs (first, second); SBWT.hh:40 */
    int64_t    precalc_k;
    int64_t    n_nodes;
    int64_t    n_kmers;
    int64_t    k;
} orc_index;

/* ---- bit vector + rank (sdsl stand-in) ---- */
int      orc_bitvec_init(orc_bitvec *bv, const uint64_t *words, int64_t n_bits);
void     orc_bitvec_free(orc_bitvec *bv);
int64_t  orc_bitvec_rank(const orc_bitvec *bv, int64_t idx);     /* #ones in [0, idx) */
int      orc_bitvec_get(const orc_bitvec *bv, int64_t idx);

/* ---- index life cycle ---- */
/* SBWT(A,C,G,T,ssup,k,n_kmers,precalc_k) ctor, SBWT.hh:335-353.  ssup_words may be NULL. */
orc_index *orc_index_from_bits(const uint64_t *A, const uint64_t *C, const uint64_t *G,
                               const uint64_t *T, const uint64_t *ssup_words,
                               int64_t n_nodes, int64_t k, int64_t n_kmers, int64_t precalc_k);
/* build_nodeboss_in_memory(), NodeBOSSInMemoryConstructor.hh:98-213.  `seqs` are
 * NUL-terminated strings; k-mers containing non-ACGT are skipped (:156-159).
 * add_revcomp mirrors `sbwt build --add-reverse-complements` (sbwt_build.cpp:108-123). */
orc_index *orc_index_build(const char *const *seqs, int64_t n_seqs, int64_t k,
                           int streaming_support, int add_revcomp, int64_t precalc_k);
void       orc_index_free(orc_index *idx);
/* do_kmer_prefix_precalc, SBWT.hh:616-645.  Returns 0, or -1 (p>20) / -2 (p>k). */
int        orc_do_precalc(orc_index *idx, int64_t p);

/* ---- queries ---- */
int64_t orc_rank(const orc_index *idx, int64_t pos, char c);               /* SubsetMatrixRank.hh:31-37 */
int     orc_contains(const orc_index *idx, int64_t pos, char c);           /* SubsetMatrixRank.hh:39-48 */
void    orc_update_interval(const orc_index *idx, const char *S, int64_t len,
                            int64_t *first, int64_t *second);              /* SBWT.hh:422-437 */
/* SBWT.hh:389-415.  Returns the rank, -1 if absent, -2 on the "Bug: not a singleton" exit(1) path. */
int64_t orc_search(const orc_index *idx, const char *kmer);
/* SBWT.hh:544-581.  Writes max(0,len-k+1) values; returns that count, or -1 when there
 * is no streaming support (the reference throws, :546-547). */
int64_t orc_streaming_search(const orc_index *idx, const char *input, int64_t len, int64_t *out);
/* sbwt_search.cpp:67-91 inner loop: out[i] = search(input+i). Returns count. */
int64_t orc_search_all(const orc_index *idx, const char *input, int64_t len, int64_t *out);
int64_t orc_forward(const orc_index *idx, int64_t node, char c);           /* SBWT.hh:368-381; -2 if no ssup */
/* SBWT.hh:525-537.  Returns matched length, interval in *l,*r. */
int64_t orc_partial_search(const orc_index *idx, const char *input, int64_t len, int64_t *l, int64_t *r);

/* mark_suffix_groups, src/suffix_group_optimization.cpp:66-121: recompute the streaming
 * support bit vector from the four columns alone (independent check of ssup). out has
 * ceil(n/64) words. */
void orc_mark_suffix_groups(const orc_index *idx, uint64_t *out_words);

/* print_vector, src/CLI/sbwt_search.cpp:21-43.  Appends to buf (caller sizes it: <= 21
 * bytes per value + 1).  Returns bytes written. */
int64_t orc_print_vector(const int64_t *v, int64_t n, char *buf);

/* ---- batched drivers used for the CPU baseline (bench.py cpu_baseline leg) ---- */
/* Runs streaming_search (or search_all when the index has no ssup) for reads
 * [0,n_reads) with `n_threads` pthreads over contiguous read ranges; read r is
 * bases[read_off[r] .. read_off[r+1]), results at out[out_off[r]..].  Returns the
 * summed wall time in seconds of the query calls only, like the reference's
 * "us/query (excluding I/O etc)" timer (sbwt_search.cpp:54-56), max over threads. */
double orc_batch_search(const orc_index *idx, const char *bases, const int64_t *read_off,
                        int64_t n_reads, int64_t *out, const int64_t *out_off, int n_threads);

/* run_file + run_queries_* + print_vector, src/CLI/sbwt_search.cpp:21-105: the reference CLI's single-threaded loop over one
 * FASTQ / FASTA file (parse, query, format, write).  Returns the wall seconds of the whole call (-1 on I/O error);
 * *query_secs = the summed query time alone ("us/query (excluding I/O etc)", :54-56).  bench.py's end-to-end CPU figure. */
double orc_search_file(const orc_index *idx, const char *query_path, const char *out_path, int64_t *n_reads_out,
                       int64_t *n_kmers_out, double *query_secs);

/* SBWT.hh:700-725: writes the k chars of column colex_rank's k-mer ('$' for dummy positions), no NUL. */
void    orc_get_kmer(const orc_index *idx, int64_t colex_rank, char *buf);
/* SubsetMatrixSelectSupport.hh:27-33: column of the j-th (1-based) set bit of row c; non-ACGT -> 0; -1 if j too large. */
int64_t orc_select(const orc_index *idx, int64_t j, char c);

/* out[i] = orc_rank(pos[i], sym[i]) over n_threads pthreads; returns the slowest thread's loop time (seconds). */
double orc_batch_rank(const orc_index *idx, const int64_t *pos, const char *sym, int64_t n, int64_t *out,
                      int n_threads);

/* Work accounting for the roofline's algorithmic bytes: replays the reference algorithm on
 * the batch and counts (a) streaming one-step extensions taken, (b) full search() calls,
 * (c) LF steps (interval updates actually executed inside update_sbwt_interval). */
void orc_count_work(const orc_index *idx, const char *bases, const int64_t *read_off,
                    int64_t n_reads, int64_t *n_stream_steps, int64_t *n_searches,
                    int64_t *n_lf_steps);

#ifdef __cplusplus
}
#endif
#endif
