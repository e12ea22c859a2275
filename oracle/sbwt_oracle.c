/*
 * sbwt_oracle.c -- CPU restatement of the reference's plain-matrix SBWT search path.
 *
 * TEST INFRASTRUCTURE ONLY (see sbwt_oracle.h).  Plain C11, no dependencies beyond libc
 * and pthreads.  Citations are file:line relative to /root/reference.
 */
#define _POSIX_C_SOURCE 200809L
#include "sbwt_oracle.h"

#include <ctype.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------------- */
/* globals.hh:38-53  ACGT <-> 0123 (upper case only; everything else -1)      */
/* ------------------------------------------------------------------------- */
static inline int dna_to_idx(char c) {
    switch (c) {
        case 'A': return 0;
        case 'C': return 1;
        case 'G': return 2;
        case 'T': return 3;
        default:  return -1;
    }
}
static const char IDX_TO_DNA[4] = {'A', 'C', 'G', 'T'};

/* globals.hh:19-35,56-58 reverse complement (upper->upper, lower->lower, others kept) */
static inline char rc_char(char c) {
    switch (c) {
        case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
        case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a';
        default:  return c;
    }
}

/* ------------------------------------------------------------------------- */
/* sdsl::bit_vector + sdsl::rank_support_v5<> stand-in                         */
/* ------------------------------------------------------------------------- */
int orc_bitvec_init(orc_bitvec *bv, const uint64_t *words, int64_t n_bits) {
    memset(bv, 0, sizeof(*bv));
    bv->n_bits = n_bits;
    int64_t nw = (n_bits + 63) / 64;
    bv->n_words = nw;
    /* 32 guard words so that word(idx) for idx == n_bits and whole superblocks are readable */
    bv->words = (uint64_t *)calloc((size_t)nw + 33, sizeof(uint64_t));
    if (!bv->words) return -1;
    if (nw && words) memcpy(bv->words, words, (size_t)nw * 8);
    if (n_bits & 63) bv->words[nw - 1] &= (~0ULL) >> (64 - (n_bits & 63)); /* clear tail bits */
    int64_t n_sb = ((nw * 64) >> 11) + 1;
    bv->n_dir = 2 * n_sb;
    bv->dir = (uint64_t *)calloc((size_t)bv->n_dir, sizeof(uint64_t));
    if (!bv->dir) return -1;
    uint64_t total = 0;
    for (int64_t sb = 0; sb < n_sb; sb++) {
        bv->dir[2 * sb] = total;
        uint64_t packed = 0, in_sb = 0;
        for (int w = 0; w < 32; w++) {
            int64_t wi = sb * 32 + w;
            /* sdsl's construction loop packs a field only once the vector has the 6*m words before it [UPSTREAM-KNOWLEDGE];
             * fields past the last word stay 0 (no rank query ever reads them) */
            if (w > 0 && w % 6 == 0 && wi <= nw) packed |= in_sb << (60 - 12 * (w / 6));
            if (wi < nw) in_sb += (uint64_t)__builtin_popcountll(bv->words[wi]);
        }
        bv->dir[2 * sb + 1] = packed;
        total += in_sb;
    }
    return 0;
}

void orc_bitvec_free(orc_bitvec *bv) {
    free(bv->words);
    free(bv->dir);
    memset(bv, 0, sizeof(*bv));
}

/* rank_support_v5::rank(idx) [UPSTREAM-KNOWLEDGE, SURVEY App. A]: #ones in [0, idx). */
int64_t orc_bitvec_rank(const orc_bitvec *bv, int64_t idx) {
    const uint64_t *p = bv->dir + ((idx >> 11) << 1);
    int64_t blk = (idx & 0x7FF) / 384;
    uint64_t r = p[0] + ((p[1] >> (60 - 12 * blk)) & 0x7FF);
    int64_t w = idx >> 6;
    int64_t first = ((idx >> 11) << 5) + blk * 6;
    for (int64_t i = first; i < w; i++) r += (uint64_t)__builtin_popcountll(bv->words[i]);
    if (idx & 63) r += (uint64_t)__builtin_popcountll(bv->words[w] & ((1ULL << (idx & 63)) - 1));
    return (int64_t)r;
}

int orc_bitvec_get(const orc_bitvec *bv, int64_t idx) {
    return (int)((bv->words[idx >> 6] >> (idx & 63)) & 1);
}

/* ------------------------------------------------------------------------- */
/* SubsetMatrixRank                                                            */
/* ------------------------------------------------------------------------- */
/* SubsetMatrixRank.hh:31-37 */
int64_t orc_rank(const orc_index *idx, int64_t pos, char c) {
    if (c == 'A') return orc_bitvec_rank(&idx->col[0], pos);
    if (c == 'C') return orc_bitvec_rank(&idx->col[1], pos);
    if (c == 'G') return orc_bitvec_rank(&idx->col[2], pos);
    if (c == 'T') return orc_bitvec_rank(&idx->col[3], pos);
    return 0;
}

/* SubsetMatrixRank.hh:39-48 */
int orc_contains(const orc_index *idx, int64_t pos, char c) {
    switch (c) {
        case 'A': return orc_bitvec_get(&idx->col[0], pos);
        case 'C': return orc_bitvec_get(&idx->col[1], pos);
        case 'G': return orc_bitvec_get(&idx->col[2], pos);
        case 'T': return orc_bitvec_get(&idx->col[3], pos);
        default:  return 0;
    }
}

/* ------------------------------------------------------------------------- */
/* SBWT                                                                        */
/* ------------------------------------------------------------------------- */
/* SBWT.hh:422-437.  Note (Q1): the RAW char is validated (:427) although the
 * upper-cased one is used for rank (:426) -- lower case therefore fails here. */
static inline void update_interval(const orc_index *idx, const char *S, int64_t len,
                                   int64_t *first, int64_t *second, int64_t *lf_steps) {
    if (*first == -1) return;
    for (int64_t i = 0; i < len; i++) {
        char c = (char)toupper((unsigned char)S[i]);
        int ci = dna_to_idx(S[i]);
        if (ci == -1) { *first = -1; *second = -1; return; }
        if (lf_steps) (*lf_steps)++;
        *first  = idx->C[ci] + orc_rank(idx, *first, c);
        *second = idx->C[ci] + orc_rank(idx, *second + 1, c) - 1;
        if (*first > *second) { *first = -1; *second = -1; return; }
    }
}

void orc_update_interval(const orc_index *idx, const char *S, int64_t len,
                         int64_t *first, int64_t *second) {
    update_interval(idx, S, len, first, second, NULL);
}

/* SBWT.hh:389-415 */
static inline int64_t search_counted(const orc_index *idx, const char *kmer, int64_t *lf_steps) {
    int64_t first, second;
    if (idx->precalc_k > 0) {
        uint64_t pidx = 0;
        for (int64_t i = 0; i < idx->precalc_k; i++) {              /* :396-401 */
            int ci = dna_to_idx(kmer[idx->precalc_k - 1 - i]);
            if (ci == -1) return -1;
            pidx = (pidx << 2) | (uint64_t)ci;
        }
        first = idx->precalc[2 * pidx];
        second = idx->precalc[2 * pidx + 1];
        update_interval(idx, kmer + idx->precalc_k, idx->k - idx->precalc_k, &first, &second, lf_steps);
    } else {
        first = 0;
        second = idx->n_nodes - 1;
        update_interval(idx, kmer, idx->k, &first, &second, lf_steps);
    }
    if (first != second) return -2; /* :410-413 "Bug: ... not a singleton" -> exit(1) */
    return first;
}

int64_t orc_search(const orc_index *idx, const char *kmer) {
    return search_counted(idx, kmer, NULL);
}

/* SBWT.hh:544-581 */
int64_t orc_streaming_search(const orc_index *idx, const char *input, int64_t len, int64_t *out) {
    if (idx->ssup.n_bits == 0) return -1; /* throws "Error: streaming search support not built" */
    int64_t k = idx->k;
    if (len < k) return 0;
    int64_t n = 0;
    out[n++] = orc_search(idx, input);
    for (int64_t i = 1; i < len - k + 1; i++) {
        if (out[n - 1] == -1) {
            out[n] = orc_search(idx, input + i);
            n++;
        } else {
            int64_t column = out[n - 1];
            while (orc_bitvec_get(&idx->ssup, column) == 0) column--;
            char c = (char)toupper((unsigned char)input[i + k - 1]);
            int ci = dna_to_idx(c);
            if (ci == -1) {
                out[n++] = -1;
            } else {
                int64_t node_left  = idx->C[ci] + orc_rank(idx, column, c);
                int64_t node_right = idx->C[ci] + orc_rank(idx, column + 1, c) - 1;
                out[n++] = (node_left == node_right) ? node_left : -1;
            }
        }
    }
    return n;
}

/* sbwt_search.cpp:78-84 */
int64_t orc_search_all(const orc_index *idx, const char *input, int64_t len, int64_t *out) {
    int64_t n = 0;
    for (int64_t i = 0; i < len - idx->k + 1; i++) out[n++] = orc_search(idx, input + i);
    return n;
}

/* SBWT.hh:368-381 */
int64_t orc_forward(const orc_index *idx, int64_t node, char c) {
    if (idx->ssup.n_bits == 0) return -2;
    while (!orc_bitvec_get(&idx->ssup, node)) node--;
    int64_t r1 = orc_rank(idx, node, c);
    int64_t r2 = orc_rank(idx, node + 1, c);
    if (r1 == r2) return -1;
    return idx->C[dna_to_idx(c)] + r1;
}

/* SBWT.hh:525-537 */
int64_t orc_partial_search(const orc_index *idx, const char *input, int64_t len, int64_t *lo, int64_t *ro) {
    int64_t l = 0, r = idx->n_nodes - 1;
    for (int64_t i = 0; i < len; i++) {
        char c = (char)toupper((unsigned char)input[i]);
        int64_t ln = l, rn = r;
        orc_update_interval(idx, &c, 1, &ln, &rn);
        if (ln == -1) { *lo = l; *ro = r; return i; }
        l = ln; r = rn;
    }
    *lo = l; *ro = r;
    return len;
}

/* SBWT.hh:616-645 */
int orc_do_precalc(orc_index *idx, int64_t p) {
    if (p == 0) return 0;
    if (p > 20) return -1;
    if (p > idx->k) return -2;
    uint64_t n = 1ULL << (2 * p);
    free(idx->precalc);
    idx->precalc = (int64_t *)malloc((size_t)n * 16);
    idx->precalc_k = p;
    char prefix[32];
    for (uint64_t data = 0; data < n; data++) {
        for (int64_t i = 0; i < p; i++) prefix[i] = IDX_TO_DNA[(data >> (2 * i)) & 3]; /* :636-638 */
        int64_t first = 0, second = idx->n_nodes - 1;
        orc_update_interval(idx, prefix, p, &first, &second);
        idx->precalc[2 * data] = first;
        idx->precalc[2 * data + 1] = second;
    }
    return 0;
}

/* SBWT.hh:335-353 */
orc_index *orc_index_from_bits(const uint64_t *A, const uint64_t *C, const uint64_t *G,
                               const uint64_t *T, const uint64_t *ssup_words,
                               int64_t n_nodes, int64_t k, int64_t n_kmers, int64_t precalc_k) {
    orc_index *idx = (orc_index *)calloc(1, sizeof(orc_index));
    if (!idx) return NULL;
    const uint64_t *cols[4] = {A, C, G, T};
    for (int i = 0; i < 4; i++) orc_bitvec_init(&idx->col[i], cols[i], n_nodes);
    if (ssup_words) orc_bitvec_init(&idx->ssup, ssup_words, n_nodes);
    idx->n_nodes = n_nodes;
    idx->k = k;
    idx->n_kmers = n_kmers;
    idx->C[0] = 1;                                          /* :346 ghost dollar into the root */
    idx->C[1] = idx->C[0] + orc_rank(idx, n_nodes, 'A');
    idx->C[2] = idx->C[1] + orc_rank(idx, n_nodes, 'C');
    idx->C[3] = idx->C[2] + orc_rank(idx, n_nodes, 'G');
    if (orc_do_precalc(idx, precalc_k) != 0) { orc_index_free(idx); return NULL; }
    return idx;
}

void orc_index_free(orc_index *idx) {
    if (!idx) return;
    for (int i = 0; i < 4; i++) orc_bitvec_free(&idx->col[i]);
    orc_bitvec_free(&idx->ssup);
    free(idx->precalc);
    free(idx);
}

/* ------------------------------------------------------------------------- */
/* In-memory construction, NodeBOSSInMemoryConstructor.hh                      */
/* ------------------------------------------------------------------------- */
/* A string of length len<=64 packed like sbwt::Kmer (Kmer.hh:27-30,70-75): 2 bits per
 * char, the LAST char in the most significant bits, unused low bits zero.  Comparing
 * (d, len) lexicographically is Kmer::operator< (Kmer.hh:108-123): colexicographic,
 * the shorter string first on a tie. */
typedef struct { u128 d; uint8_t len; uint8_t edges; } onode;

static inline int km_cmp(u128 ad, int al, u128 bd, int bl) {
    if (ad < bd) return -1;
    if (ad > bd) return 1;
    return (al < bl) ? -1 : (al > bl) ? 1 : 0;
}
static inline u128 km_dropleft_d(u128 d, int len) {           /* Kmer.hh:131-139 */
    return d & ~((u128)3 << (2 * (64 - len)));
}
static inline u128 km_appendright_d(u128 d, int c) {           /* Kmer.hh:156-170 */
    return (d >> 2) | ((u128)c << 126);
}
static inline int km_last(u128 d) { return (int)(d >> 126); }

static int onode_cmp(const void *a, const void *b) {           /* Node::operator< :54-58 */
    const onode *x = (const onode *)a, *y = (const onode *)b;
    int c = km_cmp(x->d, x->len, y->d, y->len);
    if (c) return c;
    return (x->edges < y->edges) ? -1 : (x->edges > y->edges) ? 1 : 0;
}
static int u128_cmp(const void *a, const void *b) {
    u128 x = *(const u128 *)a, y = *(const u128 *)b;
    return (x < y) ? -1 : (x > y) ? 1 : 0;
}

typedef struct { onode *v; int64_t n, cap; } nodevec;
static void nv_push(nodevec *nv, onode x) {
    if (nv->n == nv->cap) {
        nv->cap = nv->cap ? nv->cap * 2 : 1024;
        nv->v = (onode *)realloc(nv->v, (size_t)nv->cap * sizeof(onode));
    }
    nv->v[nv->n++] = x;
}

/* add_prefixes, :70-79 */
static void add_prefixes(u128 zd, int zlen, nodevec *nodes) {
    u128 d = zd;
    int len = zlen;
    while (len > 0) {
        int edge = km_last(d);
        d <<= 2;            /* dropright, Kmer.hh:143-152 */
        len--;
        onode nd = {d, (uint8_t)len, (uint8_t)(1u << edge)};
        nv_push(nodes, nd);
    }
}

orc_index *orc_index_build(const char *const *seqs, int64_t n_seqs, int64_t k,
                           int streaming_support, int add_revcomp, int64_t precalc_k) {
    if (k < 1 || k > 64) return NULL;
    /* get_distinct_kmers, :161-172 (hash set there; sort+unique here) */
    int64_t cap = 0;
    for (int64_t s = 0; s < n_seqs; s++) {
        int64_t L = (int64_t)strlen(seqs[s]);
        if (L >= k) cap += L - k + 1;
    }
    if (add_revcomp) cap *= 2;
    u128 *kmers = (u128 *)malloc((size_t)(cap ? cap : 1) * sizeof(u128));
    int64_t nk = 0;
    for (int pass = 0; pass < (add_revcomp ? 2 : 1); pass++) {
        for (int64_t s = 0; s < n_seqs; s++) {
            int64_t L = (int64_t)strlen(seqs[s]);
            char *buf = (char *)malloc((size_t)L + 1);
            if (pass == 0) memcpy(buf, seqs[s], (size_t)L + 1);
            else { for (int64_t i = 0; i < L; i++) buf[i] = rc_char(seqs[s][L - 1 - i]); buf[L] = 0; }
            for (int64_t i = 0; i + k <= L; i++) {
                u128 d = 0;
                int ok = 1;
                for (int64_t j = 0; j < k; j++) {
                    int ci = dna_to_idx(buf[i + j]);            /* is_valid_kmer :156-159 */
                    if (ci < 0) { ok = 0; break; }
                    d = km_appendright_d(d, ci);
                }
                if (ok) kmers[nk++] = d;
            }
            free(buf);
        }
    }
    qsort(kmers, (size_t)nk, sizeof(u128), u128_cmp);           /* std::sort(kmers) :192 */
    int64_t m = 0;
    for (int64_t i = 0; i < nk; i++) if (i == 0 || kmers[i] != kmers[i - 1]) kmers[m++] = kmers[i];
    nk = m;
    int K = (int)k;

    /* get_nodes, :98-154 */
    int64_t char_ptrs[4];
    for (int c = 0; c < 4; c++) {                                /* get_char_ptr :62-67 */
        int64_t p = nk;
        for (int64_t i = 0; i < nk; i++) if (km_last(kmers[i]) == c) { p = i; break; }
        char_ptrs[c] = p;
    }
    nodevec nodes = {0};
    onode root = {0, 0, 0};
    nv_push(&nodes, root);                                       /* :104-105 always a root */
    for (int64_t i = 0; i < nk; i++) {
        int sgs = (i == 0) || (km_dropleft_d(kmers[i], K) != km_dropleft_d(kmers[i - 1], K));
        onode x = {kmers[i], (uint8_t)K, 0};
        if (sgs) {
            for (int c = 0; c < 4; c++) {
                if (char_ptrs[c] == nk) continue;
                u128 y = km_appendright_d(km_dropleft_d(kmers[i], K), c);   /* x[1..k-1] c */
                u128 z = kmers[char_ptrs[c]];
                while (y > z) {                                  /* z has no incoming edge */
                    add_prefixes(z, K, &nodes);
                    char_ptrs[c]++;
                    if (char_ptrs[c] == nk) break;
                    z = kmers[char_ptrs[c]];
                }
                if (y == z) { char_ptrs[c]++; x.edges |= (uint8_t)(1u << c); }
            }
        }
        nv_push(&nodes, x);
    }
    for (int c = 0; c < 4; c++)                                  /* :140-145 remaining */
        while (char_ptrs[c] < nk && km_last(kmers[char_ptrs[c]]) == c)
            add_prefixes(kmers[char_ptrs[c]++], K, &nodes);
    qsort(nodes.v, (size_t)nodes.n, sizeof(onode), onode_cmp);   /* :148 */
    int64_t j = 0;                                               /* merge_equal_nodes :82-94 */
    for (int64_t i = 0; i < nodes.n; i++) {
        if (i > 0 && nodes.v[i].d == nodes.v[i - 1].d && nodes.v[i].len == nodes.v[i - 1].len)
            nodes.v[j - 1].edges |= nodes.v[i].edges;
        else
            nodes.v[j++] = nodes.v[i];
    }
    nodes.n = j;

    /* build(), :188-213 */
    int64_t n = nodes.n, nw = (n + 63) / 64;
    uint64_t *bits[5];
    for (int b = 0; b < 5; b++) bits[b] = (uint64_t *)calloc((size_t)nw + 1, 8);
    for (int64_t i = 0; i < n; i++)
        for (int c = 0; c < 4; c++)
            if (nodes.v[i].edges & (1u << c)) bits[c][i >> 6] |= 1ULL << (i & 63);
    if (streaming_support) {                                     /* build_streaming_support :174-185 */
        bits[4][0] |= 1;
        for (int64_t i = 1; i < n; i++) {
            u128 a = nodes.v[i - 1].d, b = nodes.v[i].d;
            int al = nodes.v[i - 1].len, bl = nodes.v[i].len;
            if (al == K) { a = km_dropleft_d(a, K); al--; }
            if (bl == K) { b = km_dropleft_d(b, K); bl--; }
            if (a != b || al != bl) bits[4][i >> 6] |= 1ULL << (i & 63);
        }
    }
    orc_index *idx = orc_index_from_bits(bits[0], bits[1], bits[2], bits[3],
                                         streaming_support ? bits[4] : NULL, n, k, nk, precalc_k);
    for (int b = 0; b < 5; b++) free(bits[b]);
    free(nodes.v);
    free(kmers);
    return idx;
}

/* src/suffix_group_optimization.cpp:66-121 */
void orc_mark_suffix_groups(const orc_index *idx, uint64_t *out_words) {
    int64_t n = idx->n_nodes, k = idx->k;
    char *last = (char *)malloc((size_t)n + 1), *prop = (char *)malloc((size_t)n + 1);
    int64_t Carr[4], m = 0;
    last[m++] = '$';
    for (int c = 0; c < 4; c++) {
        Carr[c] = m;
        for (int64_t i = 0; i < n; i++) if (orc_bitvec_get(&idx->col[c], i)) last[m++] = IDX_TO_DNA[c];
    }
    memset(out_words, 0, (size_t)((n + 63) / 64) * 8);
    for (int64_t round = 0; round < k - 1; round++) {
        for (int64_t i = 0; i < n; i++)
            if (i == 0 || last[i] != last[i - 1]) out_words[i >> 6] |= 1ULL << (i & 63);
        memset(prop, '$', (size_t)n);
        int64_t ptr[4] = {Carr[0], Carr[1], Carr[2], Carr[3]};
        for (int64_t i = 0; i < n; i++)
            for (int c = 0; c < 4; c++)
                if (orc_bitvec_get(&idx->col[c], i)) prop[ptr[c]++] = last[i];
        char *t = last; last = prop; prop = t;
    }
    free(last);
    free(prop);
}

/* src/CLI/sbwt_search.cpp:21-43.  Quirk kept: x == 0 prints an empty token. */
int64_t orc_print_vector(const int64_t *v, int64_t n, char *buf) {
    char *o = buf;
    for (int64_t t = 0; t < n; t++) {
        int64_t x = v[t];
        char tmp[32];
        int i = 0;
        if (x == -1) { tmp[0] = '1'; tmp[1] = '-'; i = 2; }
        else while (x > 0) { tmp[i++] = (char)('0' + (x % 10)); x /= 10; }
        while (i > 0) *o++ = tmp[--i];
        *o++ = ' ';
    }
    *o++ = '\n';
    return (int64_t)(o - buf);
}

/* ------------------------------------------------------------------------- */
/* Batched drivers (CPU baseline + work accounting)                            */
/* ------------------------------------------------------------------------- */
static double now_sec(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

typedef struct {
    const orc_index *idx; const char *bases; const int64_t *read_off; int64_t r0, r1;
    int64_t *out; const int64_t *out_off; double secs;
} batch_job;

static void *batch_worker(void *arg) {
    batch_job *jb = (batch_job *)arg;
    double acc = 0;
    int streaming = jb->idx->ssup.n_bits > 0;
    for (int64_t r = jb->r0; r < jb->r1; r++) {
        const char *s = jb->bases + jb->read_off[r];
        int64_t len = jb->read_off[r + 1] - jb->read_off[r];
        double t0 = now_sec();                                   /* sbwt_search.cpp:54-56 */
        if (streaming) orc_streaming_search(jb->idx, s, len, jb->out + jb->out_off[r]);
        else orc_search_all(jb->idx, s, len, jb->out + jb->out_off[r]);
        acc += now_sec() - t0;
    }
    jb->secs = acc;
    return NULL;
}

double orc_batch_search(const orc_index *idx, const char *bases, const int64_t *read_off,
                        int64_t n_reads, int64_t *out, const int64_t *out_off, int n_threads) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    pthread_t th[256];
    batch_job jobs[256];
    for (int t = 0; t < n_threads; t++) {
        jobs[t] = (batch_job){idx, bases, read_off, n_reads * t / n_threads, n_reads * (t + 1) / n_threads,
                              out, out_off, 0.0};
        pthread_create(&th[t], NULL, batch_worker, &jobs[t]);
    }
    double mx = 0;
    for (int t = 0; t < n_threads; t++) {
        pthread_join(th[t], NULL);
        if (jobs[t].secs > mx) mx = jobs[t].secs;
    }
    return mx;
}

/* SBWT::get_kmer, SBWT.hh:700-725: the k-mer of column colex_rank, spelled backwards one incoming edge at a
 * time; the step backward is the reference's own galloping search on rank() (:713-720).  buf gets k chars, no NUL. */
void orc_get_kmer(const orc_index *idx, int64_t colex_rank, char *buf) {
    const int64_t k = idx->k, n = idx->n_nodes;
    for (int64_t i = 0; i < k; i++) {
        if (colex_rank == 0) {
            buf[k - 1 - i] = '$';
        } else {
            int64_t char_idx = 0;
            while (char_idx + 1 < 4 && colex_rank >= idx->C[char_idx + 1]) char_idx++;
            char c = IDX_TO_DNA[char_idx];
            buf[k - 1 - i] = c;
            int64_t char_rel_rank = colex_rank - idx->C[char_idx];
            int64_t p = 0, step = n;
            while (step > 0) {
                while (p + step <= n && orc_rank(idx, p + step, c) <= char_rel_rank) p += step;
                step /= 2;
            }
            colex_rank = p;
        }
    }
}

/* SubsetMatrixSelectSupport::select(pos, c), SubsetMatrixSelectSupport.hh:27-33 with sdsl select_1 semantics
 * [UPSTREAM-KNOWLEDGE: select(j) = index of the j-th set bit, j >= 1]; non-ACGT -> 0.  Linear scan: test sizes only. */
int64_t orc_select(const orc_index *idx, int64_t j, char c) {
    int ci = dna_to_idx(c);
    if (ci < 0) return 0;
    int64_t seen = 0;
    for (int64_t i = 0; i < idx->n_nodes; i++)
        if (orc_bitvec_get(&idx->col[ci], i) && ++seen == j) return i;
    return -1;
}

/* ---- batched rank driver (bench.py --kernel rank cpu_baseline leg, rank parity tests) ----
 * out[i] = SubsetMatrixRank::rank(pos[i], sym[i]) (SubsetMatrixRank.hh:31-37) with n_threads pthreads over
 * contiguous ranges; returns the slowest thread's wall time of its loop. */
typedef struct { const orc_index *idx; const int64_t *pos; const char *sym; int64_t i0, i1; int64_t *out; double secs; } rank_job;

static void *rank_worker(void *arg) {
    rank_job *jb = (rank_job *)arg;
    double t0 = now_sec();
    for (int64_t i = jb->i0; i < jb->i1; i++) jb->out[i] = orc_rank(jb->idx, jb->pos[i], jb->sym[i]);
    jb->secs = now_sec() - t0;
    return NULL;
}

double orc_batch_rank(const orc_index *idx, const int64_t *pos, const char *sym, int64_t n, int64_t *out,
                      int n_threads) {
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    pthread_t th[256];
    rank_job jobs[256];
    for (int t = 0; t < n_threads; t++) {
        jobs[t] = (rank_job){idx, pos, sym, n * t / n_threads, n * (t + 1) / n_threads, out, 0.0};
        pthread_create(&th[t], NULL, rank_worker, &jobs[t]);
    }
    double mx = 0;
    for (int t = 0; t < n_threads; t++) {
        pthread_join(th[t], NULL);
        if (jobs[t].secs > mx) mx = jobs[t].secs;
    }
    return mx;
}

void orc_count_work(const orc_index *idx, const char *bases, const int64_t *read_off,
                    int64_t n_reads, int64_t *n_stream_steps, int64_t *n_searches,
                    int64_t *n_lf_steps) {
    int64_t ns = 0, nf = 0, nl = 0, k = idx->k;
    int streaming = idx->ssup.n_bits > 0;
    for (int64_t r = 0; r < n_reads; r++) {
        const char *s = bases + read_off[r];
        int64_t len = read_off[r + 1] - read_off[r];
        int64_t prev = -1;
        for (int64_t i = 0; i < len - k + 1; i++) {
            if (!streaming || i == 0 || prev == -1) {
                nf++;
                prev = search_counted(idx, s + i, &nl);
            } else {
                ns++;
                int64_t column = prev;
                while (orc_bitvec_get(&idx->ssup, column) == 0) column--;
                char c = (char)toupper((unsigned char)s[i + k - 1]);
                int ci = dna_to_idx(c);
                if (ci == -1) prev = -1;
                else {
                    int64_t L = idx->C[ci] + orc_rank(idx, column, c);
                    int64_t R = idx->C[ci] + orc_rank(idx, column + 1, c) - 1;
                    prev = (L == R) ? L : -1;
                }
            }
        }
    }
    *n_stream_steps = ns; *n_searches = nf; *n_lf_steps = nl;
}

/* ------------------------------------------------------------------------- */
/* The reference CLI's loop over one query file, for the end-to-end CPU figure */
/* ------------------------------------------------------------------------- */
/* run_file + run_queries_streaming / run_queries_not_streaming + print_vector, src/CLI/sbwt_search.cpp:21-105: one thread,
 * read by read -- parse, query, format, write.  The reader is a minimal 4-line FASTQ / single- or multi-line FASTA parser
 * (the reference uses its SeqIO submodule, absent here; sequences are upper-cased on read like SeqIO does [UPSTREAM-KNOWLEDGE]).
 * Returns the wall time of the whole call in seconds (-1.0 on an I/O error); *query_secs gets the summed query time alone,
 * the reference's "us/query (excluding I/O etc)" (sbwt_search.cpp:54-56). */
double orc_search_file(const orc_index *idx, const char *query_path, const char *out_path, int64_t *n_reads_out,
                       int64_t *n_kmers_out, double *query_secs) {
    const double t0 = now_sec();
    FILE *in = fopen(query_path, "rb");
    if (!in) return -1.0;
    FILE *out = fopen(out_path, "wb");
    if (!out) { fclose(in); return -1.0; }
    setvbuf(in, NULL, _IOFBF, 1 << 22);
    setvbuf(out, NULL, _IOFBF, 1 << 22);
    char *line = NULL;
    size_t cap = 0;
    ssize_t n;
    int64_t n_reads = 0, n_kmers = 0, res_cap = 0, txt_cap = 0, seq_cap = 0, seq_len = 0;
    int64_t *res = NULL;
    char *txt = NULL, *seq = NULL;
    double qsecs = 0.0;
    int fastq = -1, ok = 1;
    ssize_t pending = getline(&line, &cap, in);
    while (ok && pending > 0) {
        if (fastq < 0) fastq = line[0] == '@';
        if (line[0] != (fastq ? '@' : '>')) { ok = 0; break; }
        seq_len = 0;
        if (fastq) {
            n = getline(&line, &cap, in);
            if (n <= 0) { ok = 0; break; }
            while (n > 0 && (line[n - 1] == '\n' || line[n - 1] == '\r')) n--;
            if (n + 1 > seq_cap) { seq_cap = 2 * (n + 1); seq = (char *)realloc(seq, (size_t)seq_cap); }
            memcpy(seq, line, (size_t)n);
            seq_len = n;
            if (getline(&line, &cap, in) <= 0 || getline(&line, &cap, in) <= 0) { ok = 0; break; }   /* '+' and quality lines */
            pending = getline(&line, &cap, in);
        } else {
            for (;;) {
                pending = getline(&line, &cap, in);
                if (pending <= 0 || line[0] == '>') break;
                n = pending;
                while (n > 0 && (line[n - 1] == '\n' || line[n - 1] == '\r')) n--;
                if (seq_len + n + 1 > seq_cap) { seq_cap = 2 * (seq_len + n + 1); seq = (char *)realloc(seq, (size_t)seq_cap); }
                memcpy(seq + seq_len, line, (size_t)n);
                seq_len += n;
            }
        }
        if (seq_len == 0) break;   /* sbwt_search.cpp:49-50,72-73: `if(len == 0) break;` -- an empty record ends the file like EOF */
        for (int64_t t = 0; t < seq_len; t++)
            if (seq[t] >= 'a' && seq[t] <= 'z') seq[t] = (char)(seq[t] - 32);
        int64_t m = seq_len - idx->k + 1;
        if (m < 0) m = 0;
        if (m + 1 > res_cap) { res_cap = 2 * (m + 1); res = (int64_t *)realloc(res, (size_t)res_cap * 8); }
        if (21 * m + 2 > txt_cap) { txt_cap = 2 * (21 * m + 2); txt = (char *)realloc(txt, (size_t)txt_cap); }
        const double q0 = now_sec();
        const int64_t got = idx->ssup.n_bits > 0 ? orc_streaming_search(idx, seq, seq_len, res) : orc_search_all(idx, seq, seq_len, res);
        qsecs += now_sec() - q0;
        const int64_t bytes = orc_print_vector(res, got > 0 ? got : 0, txt);
        if ((int64_t)fwrite(txt, 1, (size_t)bytes, out) != bytes) { ok = 0; break; }
        n_reads++;
        n_kmers += got > 0 ? got : 0;
    }
    free(line); free(res); free(txt); free(seq);
    fclose(in);
    if (fclose(out) != 0) ok = 0;
    if (n_reads_out) *n_reads_out = n_reads;
    if (n_kmers_out) *n_kmers_out = n_kmers;
    if (query_secs) *query_secs = qsecs;
    return ok ? now_sec() - t0 : -1.0;
}
