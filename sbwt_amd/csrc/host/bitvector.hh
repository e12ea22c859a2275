// bitvector.hh -- minimal stand-ins for sdsl::bit_vector and sdsl::rank_support_v5<> as far as the
// plain-matrix index FILE FORMAT needs them (the sdsl-lite submodule is absent from
// the reference checkout, so the byte layout below is [UPSTREAM-KNOWLEDGE] and file-format parity
// is unpinned), and as far as the SCALAR members of the API need them ("scalar rank() runs
// on host"): rank_support_v5_blob::rank answers SubsetMatrixRank::rank(pos, c) for ONE position from
// the directory it would serialize -- a GPU launch per rank is 10^4 x the work.  Batches go to the GPU.
#pragma once
#include <cstdint>
#include <istream>
#include <ostream>
#include <stdexcept>
#include <vector>

namespace sbwt {

class bit_vector {
public:
    bit_vector() = default;
    explicit bit_vector(int64_t n_bits, bool value = false) { resize(n_bits, value); }
    bit_vector(const uint64_t *w, int64_t n_bits) : n_(n_bits), words_(w, w + (n_bits + 63) / 64) { clear_tail(); }

    void resize(int64_t n_bits, bool value = false) {
        n_ = n_bits;
        words_.assign((size_t)((n_bits + 63) / 64), value ? ~0ull : 0ull);
        clear_tail();
    }
    int64_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    bool operator[](int64_t i) const { return (words_[(size_t)(i >> 6)] >> (i & 63)) & 1ull; }
    void set(int64_t i, bool v) {
        uint64_t m = 1ull << (i & 63);
        if (v) words_[(size_t)(i >> 6)] |= m; else words_[(size_t)(i >> 6)] &= ~m;
    }
    const uint64_t *data() const { return words_.data(); }
    uint64_t *data() { return words_.data(); }
    int64_t n_words() const { return (int64_t)words_.size(); }
    bool operator==(const bit_vector &o) const { return n_ == o.n_ && words_ == o.words_; }
    bool operator!=(const bit_vector &o) const { return !(*this == o); }

    // sdsl::int_vector<1>::serialize: uint64 size in bits, then ceil(size/64) little-endian words
    int64_t serialize(std::ostream &os) const {
        uint64_t n = (uint64_t)n_;
        os.write((const char *)&n, 8);
        os.write((const char *)words_.data(), (std::streamsize)(words_.size() * 8));
        return 8 + (int64_t)words_.size() * 8;
    }
    void load(std::istream &is) {
        uint64_t n = 0;
        is.read((char *)&n, 8);
        if (!is.good() || n > ((uint64_t)1 << 48)) throw std::runtime_error("Error: corrupt bit vector in index file");
        n_ = (int64_t)n;
        words_.assign((size_t)((n_ + 63) / 64), 0);
        is.read((char *)words_.data(), (std::streamsize)(words_.size() * 8));
        if (!is.good() && !words_.empty()) throw std::runtime_error("Error: truncated bit vector in index file");
        clear_tail();
    }

private:
    void clear_tail() {
        if (n_ & 63) words_.back() &= (~0ull) >> (64 - (n_ & 63));
    }
    int64_t n_ = 0;
    std::vector<uint64_t> words_;
};

// The directory sdsl::rank_support_v5<1> serializes next to its bit vector: an int_vector<64> with
// two words per 2048-bit superblock -- the number of ones before the superblock, and five 12-bit
// fields (shifts 48,36,24,12,0) with the ones in its first 6,12,18,24,30 words.  Only built so that
// files written here carry the blobs upstream `sbwt` expects; the loader skips them.
class rank_support_v5_blob {
public:
    std::vector<uint64_t> bb;

    void build(const bit_vector &v) {
        const int64_t cap_words = v.n_words();
        bb.assign((size_t)((((cap_words * 64) >> 11) + 1) << 1), 0);
        if (cap_words == 0) return;
        const uint64_t *data = v.data();
        size_t j = 0;
        uint64_t sum = (uint64_t)__builtin_popcountll(data[0]);
        uint64_t second = 0, cnt_words = 1;
        for (int64_t i = 1; i < cap_words; ++i, ++cnt_words) {
            if (cnt_words == 32) {
                j += 2;
                bb[j - 1] = second;
                bb[j] = bb[j - 2] + sum;
                second = sum = cnt_words = 0;
            } else if (cnt_words % 6 == 0) {
                second |= sum << (60 - 12 * (cnt_words / 6));
            }
            sum += (uint64_t)__builtin_popcountll(data[i]);
        }
        if (cnt_words % 6 == 0) second |= sum << (60 - 12 * (cnt_words / 6));
        if (cnt_words == 32) {
            j += 2;
            bb[j - 1] = second;
            bb[j] = bb[j - 2] + sum;
            bb[j + 1] = 0;
        } else {
            bb[j + 1] = second;
        }
    }
    // rank(idx) = number of ones in v[0, idx), idx in [0, v.size()] (sdsl::rank_support_v5<>::rank [UPSTREAM-KNOWLEDGE],
    // SURVEY App. A): the superblock's absolute count, the 12-bit count of its 384-bit blocks before idx, the whole words
    // between that block's start and idx's word, and the masked word.  v must be the vector build() was called with.
    int64_t rank(const bit_vector &v, int64_t idx) const {
        const uint64_t *p = &bb[(size_t)((idx >> 11) << 1)];
        const int64_t in_sb = idx & 0x7FF, blk = in_sb / 384;
        uint64_t r = p[0] + ((p[1] >> (60 - 12 * blk)) & 0x7FFull);
        const uint64_t *w = v.data();
        const int64_t wi = idx >> 6, w0 = ((idx >> 11) << 5) + 6 * blk;        // idx's word; first word of its 384-bit block
        for (int64_t q = w0; q < wi; q++) r += (uint64_t)__builtin_popcountll(w[q]);
        if (idx & 63) r += (uint64_t)__builtin_popcountll(w[wi] & ((1ull << (idx & 63)) - 1ull));
        return (int64_t)r;
    }
    // sdsl::int_vector<64>::serialize: uint64 size in bits, then the words
    int64_t serialize(std::ostream &os) const {
        uint64_t bits = (uint64_t)bb.size() * 64;
        os.write((const char *)&bits, 8);
        os.write((const char *)bb.data(), (std::streamsize)(bb.size() * 8));
        return 8 + (int64_t)bb.size() * 8;
    }
    // skip over a serialized directory (ranks depend only on the bits)
    static void skip(std::istream &is) {
        uint64_t bits = 0;
        is.read((char *)&bits, 8);
        if (!is.good() || (bits & 63) || bits > ((uint64_t)1 << 48))
            throw std::runtime_error("Error: corrupt rank support in index file");
        is.seekg((std::streamoff)(bits / 8), std::ios::cur);
        if (!is.good()) throw std::runtime_error("Error: truncated rank support in index file");
    }
};

}  // namespace sbwt
