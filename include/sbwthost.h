/*
 * sbwthost.h -- C ABI of the GPU-free host helpers (libsbwthost.so): in-memory construction of the
 * plain-matrix SBWT columns and the reference's index file format.  These sit either side of the
 * hot path; tests and bench.py use them to make and move indexes.
 */
#ifndef SBWTHOST_H
#define SBWTHOST_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct sbwthost_bits sbwthost_bits;     /* A/C/G/T(+ssup) bit vectors of one index */
typedef struct sbwthost_file sbwthost_file;     /* a parsed index file */

const char *sbwthost_last_error(void);

/* Sort-based in-memory builder (same columns as NodeBOSSInMemoryConstructor.hh:98-213 / the KMC
 * constructor of the reference).  seqs[i] has seq_lens[i] bytes; k-mers containing anything but
 * upper-case ACGT are skipped.  2 <= k <= 255. */
int  sbwthost_build(const char *const *seqs, const int64_t *seq_lens, int64_t n_seqs, int64_t k,
                    int add_revcomp, int build_streaming_support, int n_threads, sbwthost_bits **out);
void sbwthost_bits_free(sbwthost_bits *b);
int  sbwthost_bits_info(const sbwthost_bits *b, int64_t *n_nodes, int64_t *n_kmers, int64_t *k, int *has_ssup);
/* which: 0..3 = A,C,G,T, 4 = suffix_group_starts (NULL if absent); ceil(n_nodes/64) words */
const uint64_t *sbwthost_bits_words(const sbwthost_bits *b, int which);

/* Index files in the reference format: serialize_string("plain-matrix") (sbwt_build.cpp:142)
 * followed by SBWT::serialize (SBWT.hh:462-491). */
int  sbwthost_file_write(const char *path, int64_t n_nodes, const uint64_t *A, const uint64_t *C,
                         const uint64_t *G, const uint64_t *T, const uint64_t *ssup /* nullable */,
                         const int64_t C_array[4], const int64_t *precalc_pairs, int64_t precalc_k,
                         int64_t n_kmers, int64_t k);
int  sbwthost_file_read(const char *path, sbwthost_file **out);
void sbwthost_file_free(sbwthost_file *f);
int  sbwthost_file_info(const sbwthost_file *f, int64_t *n_nodes, int64_t *n_kmers, int64_t *k,
                        int64_t *precalc_k, int64_t C_array[4], int *has_ssup);
const uint64_t *sbwthost_file_words(const sbwthost_file *f, int which);
const int64_t  *sbwthost_file_precalc(const sbwthost_file *f);

/* FASTA/FASTQ(.gz) reader (SeqIO stand-in): concatenates all reads of `path`.
 * *bases / *read_off are malloc'ed; free with sbwthost_free. */
int  sbwthost_read_sequences(const char *path, char **bases, int64_t **read_off, int64_t *n_reads);
void sbwthost_free(void *p);
/* The same reads by the CLI's reader for plain regular files: the file cut at record starts about chunk_bytes apart,
 * the pieces parsed by n_threads threads and put together in file order (seqio.hh read_file_chunked).  Returns 1 (and
 * reads nothing) when the file is not one that can be cut: gzip data, a pipe, an unknown extension. */
int  sbwthost_read_sequences_chunked(const char *path, int64_t chunk_bytes, int n_threads, char **bases, int64_t **read_off,
                                     int64_t *n_reads);
/* Writes n bytes through the CLI's buffered writer; gzip_output != 0 compresses 1 MiB blocks on n_threads
 * threads (0 = automatic) into a multi-member gzip file (-z of `sbwt search`, sbwt_search.cpp:120). */
int  sbwthost_write_file(const char *path, const char *data, int64_t n, int gzip_output, int n_threads);

/* The host rank directory of the scalar API (host/bitvector.hh: rank_support_v5_blob, the directory an index file carries
 * beside every bit vector): out[i] = number of set bits in bits[0, pos[i]) for n positions, pos[i] in [0, n_bits].
 * (SubsetMatrixRank::rank of ONE position through the C++ mirror answers from this structure; batches of the search
 * path go to the GPU -- this entry point exists so that the directory's arithmetic is tested where there is no GPU.) */
int  sbwthost_rank_batch(const uint64_t *bits, int64_t n_bits, const int64_t *pos, int64_t n, int64_t *out);

#ifdef __cplusplus
}
#endif
#endif
