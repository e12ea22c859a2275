// index_builder.hh -- sort-based in-memory construction of the plain-matrix SBWT columns.
//
// The search path needs an index to run on; the reference builds it with KMC + external sorting
// (out of scope here).  This builder produces the SAME bit vectors as
// the reference's constructors (pinned through the test_CLI.hh:90 known answers and differentially
// against the oracle's literal restatement of NodeBOSSInMemoryConstructor.hh) with an algorithm
// of its own: pack every k-mer into an integer whose natural order is the colexicographic order
// (Kmer.hh:108-123), sort, and derive predecessors, edges, dummy prefixes and suffix-group marks
// by merge-joins over the sorted array.  Node/edge rules follow NodeBOSSInMemoryConstructor.hh:98-154
// (edges only on suffix-group starts; k-mers without a predecessor get all their proper prefixes
// as dummy nodes; the empty root always exists) and :174-185 (suffix_group_starts).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace sbwt {

struct PlainMatrixBits {
    int64_t n_nodes = 0, n_kmers = 0, k = 0;
    std::vector<uint64_t> A, C, G, T, ssup;   // ssup empty when streaming support is not built
};

namespace builder_detail {

// Fixed-width unsigned integer of W 64-bit words (little endian) with just the operations the builder
// needs, for k-mers longer than 64 bases (the reference supports k <= 255, CMakeLists.txt:70-78).
template <int W>
struct BigUInt {
    uint64_t w[W];
    BigUInt() { for (int i = 0; i < W; i++) w[i] = 0; }
    BigUInt(uint64_t v) { w[0] = v; for (int i = 1; i < W; i++) w[i] = 0; }   // NOLINT (implicit on purpose)
    BigUInt(int v) : BigUInt((uint64_t)v) {}                                   // NOLINT
    explicit operator int() const { return (int)w[0]; }
    bool operator==(const BigUInt &o) const { for (int i = 0; i < W; i++) if (w[i] != o.w[i]) return false; return true; }
    bool operator!=(const BigUInt &o) const { return !(*this == o); }
    bool operator<(const BigUInt &o) const {
        for (int i = W - 1; i >= 0; i--) { if (w[i] != o.w[i]) return w[i] < o.w[i]; }
        return false;
    }
    bool operator<=(const BigUInt &o) const { return !(o < *this); }
    BigUInt operator|(const BigUInt &o) const { BigUInt r; for (int i = 0; i < W; i++) r.w[i] = w[i] | o.w[i]; return r; }
    BigUInt operator&(const BigUInt &o) const { BigUInt r; for (int i = 0; i < W; i++) r.w[i] = w[i] & o.w[i]; return r; }
    BigUInt operator<<(int n) const {
        BigUInt r;
        const int ws = n / 64, bs = n % 64;
        for (int i = W - 1; i >= ws; i--) {
            uint64_t v = w[i - ws] << bs;
            if (bs && i - ws - 1 >= 0) v |= w[i - ws - 1] >> (64 - bs);
            r.w[i] = v;
        }
        return r;
    }
    BigUInt operator<<(const BigUInt &n) const { return *this << (int)n.w[0]; }
    BigUInt operator>>(int n) const {
        BigUInt r;
        const int ws = n / 64, bs = n % 64;
        for (int i = 0; i + ws < W; i++) {
            uint64_t v = w[i + ws] >> bs;
            if (bs && i + ws + 1 < W) v |= w[i + ws + 1] << (64 - bs);
            r.w[i] = v;
        }
        return r;
    }
    BigUInt operator>>(const BigUInt &n) const { return *this >> (int)n.w[0]; }
    BigUInt operator-(const BigUInt &o) const {
        BigUInt r;
        unsigned __int128 borrow = 0;
        for (int i = 0; i < W; i++) {
            unsigned __int128 d = (unsigned __int128)w[i] - o.w[i] - borrow;
            r.w[i] = (uint64_t)d;
            borrow = (d >> 64) & 1;
        }
        return r;
    }
};

inline int code_of(unsigned char ch) {
    switch (ch) {
        case 'A': return 0;
        case 'C': return 1;
        case 'G': return 2;
        case 'T': return 3;
        default: return -1;
    }
}
inline unsigned char rc_of(unsigned char ch) {   // globals.hh:19-35
    switch (ch) {
        case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
        case 'a': return 't'; case 'c': return 'g'; case 'g': return 'c'; case 't': return 'a';
        default: return ch;
    }
}

// parallel sort: chunks sorted by threads, then pairwise in-place merges
template <typename It>
void parallel_sort(It first, It last, int n_threads) {
    size_t n = (size_t)(last - first);
    if (n_threads < 2 || n < (1u << 16)) {
        std::sort(first, last);
        return;
    }
    int parts = 1;
    while (parts * 2 <= n_threads) parts *= 2;
    std::vector<size_t> cut((size_t)parts + 1);
    for (int i = 0; i <= parts; i++) cut[(size_t)i] = n * (size_t)i / (size_t)parts;
    {
        std::vector<std::thread> th;
        for (int i = 0; i < parts; i++)
            th.emplace_back([&, i] { std::sort(first + cut[(size_t)i], first + cut[(size_t)i + 1]); });
        for (auto &t : th) t.join();
    }
    for (int width = 1; width < parts; width *= 2) {
        std::vector<std::thread> th;
        for (int i = 0; i + width < parts; i += 2 * width) {
            size_t a = cut[(size_t)i], b = cut[(size_t)(i + width)];
            size_t c = cut[(size_t)std::min(i + 2 * width, parts)];
            th.emplace_back([=] { std::inplace_merge(first + a, first + b, first + c); });
        }
        for (auto &t : th) t.join();
    }
}

template <typename Key>
struct Dummy {
    Key data;          // label top-aligned in 2k bits (last char of the label in the top 2 bits)
    uint8_t len;
    uint8_t edges;
    bool operator<(const Dummy &o) const {
        if (data != o.data) return data < o.data;
        return len < o.len;
    }
};

template <typename Key>
PlainMatrixBits build(const std::vector<std::string> &seqs, int k, bool add_revcomp, bool build_ssup, int n_threads) {
    const int kbits = 2 * k;
    const Key top_shift = (Key)(kbits - 2);
    // ---- 1. all k-mers, packed so that integer order == colex order: char i of the k-mer at bits 2i ----
    std::vector<Key> kmers;
    {
        size_t cap = 0;
        for (const auto &s : seqs)
            if ((int64_t)s.size() >= k) cap += s.size() - (size_t)k + 1;
        kmers.reserve(add_revcomp ? 2 * cap : cap);
    }
    auto scan = [&](const std::string &s, bool rc) {
        const int64_t L = (int64_t)s.size();
        Key key = 0;
        int run = 0;
        for (int64_t i = 0; i < L; i++) {
            unsigned char ch = rc ? rc_of((unsigned char)s[(size_t)(L - 1 - i)]) : (unsigned char)s[(size_t)i];
            int c = code_of(ch);
            if (c < 0) { run = 0; key = 0; continue; }
            key = (key >> 2) | ((Key)c << top_shift);
            if (++run >= k) kmers.push_back(key);
        }
    };
    for (const auto &s : seqs) scan(s, false);
    if (add_revcomp)
        for (const auto &s : seqs) scan(s, true);
    parallel_sort(kmers.begin(), kmers.end(), n_threads);
    kmers.erase(std::unique(kmers.begin(), kmers.end()), kmers.end());
    const int64_t nk = (int64_t)kmers.size();

    // ---- 2. suffix groups: consecutive k-mers sharing x[1..k-1] (= key >> 2) ----
    std::vector<uint8_t> edges((size_t)nk, 0);   // only set on group starts
    std::vector<int64_t> gstart;                 // index of the first k-mer of every group
    std::vector<Key> gsuf;                       // its (k-1)-suffix, ascending
    for (int64_t i = 0; i < nk; i++)
        if (i == 0 || (kmers[(size_t)i] >> 2) != (kmers[(size_t)i - 1] >> 2)) {
            gstart.push_back(i);
            gsuf.push_back(kmers[(size_t)i] >> 2);
        }
    // ---- 3. per last-char range: merge-join prefixes z[0..k-2] with the group suffixes ----
    const Key pmask = (k > 1) ? (((Key)1 << (kbits - 2)) - 1) : (Key)0;
    std::vector<Dummy<Key>> dummies;
    dummies.push_back(Dummy<Key>{0, 0, 0});      // the root
    int64_t lo = 0;
    for (int c = 0; c < 4; c++) {
        int64_t hi = lo;
        while (hi < nk && (int)(kmers[(size_t)hi] >> top_shift) == c) hi++;
        size_t g = 0;
        for (int64_t z = lo; z < hi; z++) {
            Key pre = kmers[(size_t)z] & pmask;
            while (g < gsuf.size() && gsuf[g] < pre) g++;
            if (g < gsuf.size() && gsuf[g] == pre) {
                edges[(size_t)gstart[g]] |= (uint8_t)(1u << c);     // group g --c--> z
            } else {
                // z has no predecessor: every proper prefix becomes a dummy node (add_prefixes :70-79)
                Key zk = kmers[(size_t)z];
                for (int j = 0; j < k; j++) {
                    Key label = (j == 0) ? (Key)0 : (Key)((zk & (((Key)1 << (2 * j)) - 1)) << (kbits - 2 * j));
                    int e = (int)((zk >> (2 * j)) & 3);
                    dummies.push_back(Dummy<Key>{label, (uint8_t)j, (uint8_t)(1u << e)});
                }
            }
        }
        lo = hi;
    }
    parallel_sort(dummies.begin(), dummies.end(), n_threads);
    {
        size_t w = 0;
        for (size_t i = 0; i < dummies.size(); i++) {
            if (w > 0 && dummies[w - 1].data == dummies[i].data && dummies[w - 1].len == dummies[i].len)
                dummies[w - 1].edges |= dummies[i].edges;
            else
                dummies[w++] = dummies[i];
        }
        dummies.resize(w);
    }
    // ---- 4. merge dummies and k-mers in Kmer::operator< order and emit the columns ----
    PlainMatrixBits out;
    out.k = k;
    out.n_kmers = nk;
    out.n_nodes = nk + (int64_t)dummies.size();
    const size_t nw = (size_t)((out.n_nodes + 63) / 64);
    out.A.assign(nw, 0); out.C.assign(nw, 0); out.G.assign(nw, 0); out.T.assign(nw, 0);
    if (build_ssup) out.ssup.assign(nw, 0);
    std::vector<uint64_t> *cols[4] = {&out.A, &out.C, &out.G, &out.T};
    size_t di = 0;
    int64_t ki = 0, col = 0;
    auto put = [&](uint8_t e, bool start) {
        for (int c = 0; c < 4; c++)
            if (e & (1u << c)) (*cols[c])[(size_t)(col >> 6)] |= 1ull << (col & 63);
        if (build_ssup && start) out.ssup[(size_t)(col >> 6)] |= 1ull << (col & 63);
        col++;
    };
    while (di < dummies.size() || ki < nk) {
        bool take_dummy;
        if (di == dummies.size()) take_dummy = false;
        else if (ki == nk) take_dummy = true;
        else take_dummy = dummies[di].data <= kmers[(size_t)ki];   // equal data: the shorter (dummy) first
        if (take_dummy) {
            put(dummies[di].edges, true);                           // a dummy is always its own group
            di++;
        } else {
            bool start = (ki == 0) || ((kmers[(size_t)ki] >> 2) != (kmers[(size_t)ki - 1] >> 2));
            put(edges[(size_t)ki], start);
            ki++;
        }
    }
    return out;
}

}  // namespace builder_detail

// Builds A/C/G/T (+ suffix_group_starts) for the distinct k-mers of `seqs` (k-mers containing
// anything but upper-case ACGT are skipped, NodeBOSSInMemoryConstructor.hh:156-159).
inline PlainMatrixBits build_plain_matrix_bits(const std::vector<std::string> &seqs, int k, bool add_revcomp,
                                               bool build_streaming_support, int n_threads = 1) {
    if (k < 2 || k > 255) throw std::runtime_error("Error: this builder supports 2 <= k <= 255");
    if (k <= 32) return builder_detail::build<uint64_t>(seqs, k, add_revcomp, build_streaming_support, n_threads);
    if (k <= 64) return builder_detail::build<unsigned __int128>(seqs, k, add_revcomp, build_streaming_support, n_threads);
    if (k <= 128) return builder_detail::build<builder_detail::BigUInt<4>>(seqs, k, add_revcomp, build_streaming_support, n_threads);
    return builder_detail::build<builder_detail::BigUInt<8>>(seqs, k, add_revcomp, build_streaming_support, n_threads);
}

}  // namespace sbwt
