// host_capi.cpp -- implementation of include/sbwthost.h (GPU-free host helpers).
#include "../../../include/sbwthost.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "index_builder.hh"
#include "index_file.hh"
#include "seqio.hh"

namespace {
thread_local char g_err[512] = "";
int fail(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return -1;
}
}  // namespace

struct sbwthost_bits { sbwt::PlainMatrixBits b; };
struct sbwthost_file { sbwt::IndexFileData f; };

extern "C" {

const char *sbwthost_last_error(void) { return g_err; }

int sbwthost_build(const char *const *seqs, const int64_t *seq_lens, int64_t n_seqs, int64_t k, int add_revcomp,
                   int build_ssup, int n_threads, sbwthost_bits **out) {
    if (!out || n_seqs < 0 || (n_seqs > 0 && (!seqs || !seq_lens))) return fail("invalid argument");
    if (k < 2 || k > 255) return fail("Error: this builder supports 2 <= k <= 255");
    try {
        std::vector<std::string> v;
        v.reserve((size_t)n_seqs);
        for (int64_t i = 0; i < n_seqs; i++) v.emplace_back(seqs[i], (size_t)seq_lens[i]);
        sbwthost_bits *r = new sbwthost_bits();
        r->b = sbwt::build_plain_matrix_bits(v, (int)k, add_revcomp != 0, build_ssup != 0, n_threads);
        *out = r;
        return 0;
    } catch (const std::exception &e) {
        return fail("%s", e.what());
    }
}
void sbwthost_bits_free(sbwthost_bits *b) { delete b; }
int sbwthost_bits_info(const sbwthost_bits *b, int64_t *n_nodes, int64_t *n_kmers, int64_t *k, int *has_ssup) {
    if (!b) return fail("NULL handle");
    if (n_nodes) *n_nodes = b->b.n_nodes;
    if (n_kmers) *n_kmers = b->b.n_kmers;
    if (k) *k = b->b.k;
    if (has_ssup) *has_ssup = !b->b.ssup.empty();
    return 0;
}
const uint64_t *sbwthost_bits_words(const sbwthost_bits *b, int which) {
    if (!b) return nullptr;
    switch (which) {
        case 0: return b->b.A.data();
        case 1: return b->b.C.data();
        case 2: return b->b.G.data();
        case 3: return b->b.T.data();
        case 4: return b->b.ssup.empty() ? nullptr : b->b.ssup.data();
        default: return nullptr;
    }
}

int sbwthost_file_write(const char *path, int64_t n_nodes, const uint64_t *A, const uint64_t *C, const uint64_t *G,
                        const uint64_t *T, const uint64_t *ssup, const int64_t C_array[4], const int64_t *precalc_pairs,
                        int64_t precalc_k, int64_t n_kmers, int64_t k) {
    if (!path || !A || !C || !G || !T || !C_array || n_nodes <= 0 || precalc_k < 0 || precalc_k > 20 ||
        (precalc_k > 0 && !precalc_pairs))
        return fail("invalid argument");
    try {
        sbwt::IndexFileData f;
        f.A_bits = sbwt::bit_vector(A, n_nodes);
        f.C_bits = sbwt::bit_vector(C, n_nodes);
        f.G_bits = sbwt::bit_vector(G, n_nodes);
        f.T_bits = sbwt::bit_vector(T, n_nodes);
        if (ssup) f.suffix_group_starts = sbwt::bit_vector(ssup, n_nodes);
        f.C.assign(C_array, C_array + 4);
        size_t np = precalc_k ? ((size_t)1 << (2 * precalc_k)) : 0;
        f.kmer_prefix_precalc.resize(np);
        if (np) memcpy((void *)f.kmer_prefix_precalc.data(), precalc_pairs, np * 16);
        f.precalc_k = precalc_k; f.n_nodes = n_nodes; f.n_kmers = n_kmers; f.k = k;
        std::ofstream out(path, std::ios::binary);
        if (!out.good()) return fail("Error opening file: %s", path);
        sbwt::serialize_string("plain-matrix", out);
        f.serialize(out);
        out.flush();
        if (!out.good()) return fail("Error writing to file %s", path);
        return 0;
    } catch (const std::exception &e) {
        return fail("%s", e.what());
    }
}

int sbwthost_file_read(const char *path, sbwthost_file **out) {
    if (!path || !out) return fail("invalid argument");
    try {
        std::ifstream in(path, std::ios::binary);
        if (!in.good()) return fail("Error opening file: %s", path);
        std::string variant = sbwt::load_string(in);
        if (variant != "plain-matrix") return fail("Error: not a plain-matrix index (variant '%s')", variant.c_str());
        sbwthost_file *r = new sbwthost_file();
        try {
            r->f.load(in);
        } catch (...) {
            delete r;
            throw;
        }
        *out = r;
        return 0;
    } catch (const std::exception &e) {
        return fail("%s", e.what());
    }
}
void sbwthost_file_free(sbwthost_file *f) { delete f; }
int sbwthost_file_info(const sbwthost_file *f, int64_t *n_nodes, int64_t *n_kmers, int64_t *k, int64_t *precalc_k,
                       int64_t C_array[4], int *has_ssup) {
    if (!f) return fail("NULL handle");
    if (n_nodes) *n_nodes = f->f.n_nodes;
    if (n_kmers) *n_kmers = f->f.n_kmers;
    if (k) *k = f->f.k;
    if (precalc_k) *precalc_k = f->f.precalc_k;
    if (C_array) for (int i = 0; i < 4; i++) C_array[i] = f->f.C[(size_t)i];
    if (has_ssup) *has_ssup = f->f.suffix_group_starts.size() > 0;
    return 0;
}
const uint64_t *sbwthost_file_words(const sbwthost_file *f, int which) {
    if (!f) return nullptr;
    switch (which) {
        case 0: return f->f.A_bits.data();
        case 1: return f->f.C_bits.data();
        case 2: return f->f.G_bits.data();
        case 3: return f->f.T_bits.data();
        case 4: return f->f.suffix_group_starts.size() ? f->f.suffix_group_starts.data() : nullptr;
        default: return nullptr;
    }
}
const int64_t *sbwthost_file_precalc(const sbwthost_file *f) {
    return (f && !f->f.kmer_prefix_precalc.empty()) ? (const int64_t *)f->f.kmer_prefix_precalc.data() : nullptr;
}

int sbwthost_read_sequences(const char *path, char **bases, int64_t **read_off, int64_t *n_reads) {
    if (!path || !bases || !read_off || !n_reads) return fail("invalid argument");
    try {
        sbwt::seq_io::Reader reader(path);
        std::vector<char> b;
        std::vector<int64_t> off{0};
        for (;;) {
            int64_t len = reader.get_next_read_to_buffer();
            if (len == 0) break;
            b.insert(b.end(), reader.read_buf, reader.read_buf + len);
            off.push_back((int64_t)b.size());
        }
        *bases = (char *)malloc(b.size() ? b.size() : 1);
        *read_off = (int64_t *)malloc(off.size() * 8);
        if (!*bases || !*read_off) return fail("out of memory");
        if (!b.empty()) memcpy(*bases, b.data(), b.size());
        memcpy(*read_off, off.data(), off.size() * 8);
        *n_reads = (int64_t)off.size() - 1;
        return 0;
    } catch (const std::exception &e) {
        return fail("%s", e.what());
    }
}
int sbwthost_read_sequences_chunked(const char *path, int64_t chunk_bytes, int n_threads, char **bases, int64_t **read_off,
                                    int64_t *n_reads) {
    if (!path || !bases || !read_off || !n_reads) return fail("invalid argument");
    try {
        int64_t size = 0;
        if (!sbwt::seq_io::chunkable_file(path, &size)) return 1;
        std::vector<char> b;
        std::vector<int64_t> off{0};
        sbwt::seq_io::read_file_chunked(path, size, chunk_bytes, n_threads,
                                        [&](std::vector<char> &&pb, std::vector<int64_t> &&po, bool) {
                                            const int64_t base = (int64_t)b.size();
                                            b.insert(b.end(), pb.begin(), pb.end());
                                            for (size_t r = 1; r < po.size(); r++) off.push_back(base + po[r]);
                                        });
        *bases = (char *)malloc(b.size() ? b.size() : 1);
        *read_off = (int64_t *)malloc(off.size() * 8);
        if (!*bases || !*read_off) return fail("out of memory");
        if (!b.empty()) memcpy(*bases, b.data(), b.size());
        memcpy(*read_off, off.data(), off.size() * 8);
        *n_reads = (int64_t)off.size() - 1;
        return 0;
    } catch (const std::exception &e) {
        return fail("%s", e.what());
    }
}
void sbwthost_free(void *p) { free(p); }

int sbwthost_write_file(const char *path, const char *data, int64_t n, int gzip_output, int n_threads) {
    if (!path || n < 0 || (n > 0 && !data)) return fail("invalid argument");
    try {
        sbwt::seq_io::Buffered_ofstream out(path, gzip_output != 0, n_threads);
        for (int64_t pos = 0; pos < n; pos += 3000000) out.write(data + pos, std::min<int64_t>(3000000, n - pos));
        out.close();
        return 0;
    } catch (const std::exception &e) {
        return fail("%s", e.what());
    }
}

int sbwthost_rank_batch(const uint64_t *bits, int64_t n_bits, const int64_t *pos, int64_t n, int64_t *out) {
    if (n_bits < 0 || n < 0 || (n_bits > 0 && !bits) || (n > 0 && (!pos || !out))) return fail("invalid argument");
    try {
        const sbwt::bit_vector v(bits, n_bits);
        sbwt::rank_support_v5_blob rs;
        rs.build(v);
        for (int64_t i = 0; i < n; i++) {
            if (pos[i] < 0 || pos[i] > n_bits) return fail("position %lld outside [0, %lld]", (long long)pos[i], (long long)n_bits);
            out[i] = rs.rank(v, pos[i]);
        }
        return 0;
    } catch (const std::exception &e) {
        return fail("%s", e.what());
    }
}

}  // extern "C"
