// index_file.hh -- the reference's on-disk format of a plain-matrix index:
//   serialize_string(variant) (sbwt_build.cpp:142)  ||  SBWT::serialize (SBWT.hh:462-491)
// GPU-free (only bytes in, bytes out) so that the format is testable on any host.
#pragma once
#include <cstdint>
#include <istream>
#include <ostream>
#include <stdexcept>
#include <utility>
#include <vector>

#include "bitvector.hh"
#include "globals.hh"

namespace sbwt {

const std::string SBWT_VERSION = "v0.1";   // SBWT.hh:28

struct IndexFileData {
    bit_vector A_bits, C_bits, G_bits, T_bits, suffix_group_starts;
    std::vector<int64_t> C;                                        // 4 entries
    std::vector<std::pair<int64_t, int64_t>> kmer_prefix_precalc;  // 4^precalc_k entries
    int64_t precalc_k = 0, n_nodes = 0, n_kmers = 0, k = 0;

    // SBWT::serialize (SBWT.hh:462-491) incl. SubsetMatrixRank::serialize (SubsetMatrixRank.hh:86-100)
    int64_t serialize(std::ostream &os) const {
        static_assert(sizeof(std::pair<int64_t, int64_t>) == 16, "pair<int64,int64> must be 16 bytes");
        int64_t written = 0;
        written += serialize_string(SBWT_VERSION, os);
        int64_t matrix = 0;
        for (const bit_vector *v : {&A_bits, &C_bits, &G_bits, &T_bits}) matrix += v->serialize(os);
        for (const bit_vector *v : {&A_bits, &C_bits, &G_bits, &T_bits}) {
            rank_support_v5_blob rs;
            rs.build(*v);
            matrix += rs.serialize(os);
        }
        write_log("MatrixRank bit vectors total " + std::to_string((double)matrix / (double)A_bits.size() * 8) +
                      " bits total per node",
                  LogLevel::MINOR);
        written += matrix;
        written += suffix_group_starts.serialize(os);
        int64_t nb = (int64_t)C.size() * 8;
        os.write((const char *)&nb, 8);
        os.write((const char *)C.data(), nb);
        written += 8 + nb;
        nb = (int64_t)kmer_prefix_precalc.size() * 16;
        os.write((const char *)&nb, 8);
        os.write((const char *)kmer_prefix_precalc.data(), nb);
        written += 8 + nb;
        os.write((const char *)&precalc_k, 8);
        os.write((const char *)&n_nodes, 8);
        os.write((const char *)&n_kmers, 8);
        os.write((const char *)&k, 8);
        written += 32;
        return written;
    }

    // SBWT::load (SBWT.hh:500-516) incl. SubsetMatrixRank::load (SubsetMatrixRank.hh:102-125); the four
    // rank_support_v5 blobs are skipped -- ranks depend only on the bits.
    void load(std::istream &is) {
        std::string version = load_string(is);
        if (version != SBWT_VERSION)
            throw std::runtime_error(
                "Error: Corrupt index file, or the index was constructed with an incompatible version of SBWT.");
        A_bits.load(is);
        C_bits.load(is);
        G_bits.load(is);
        T_bits.load(is);
        for (int i = 0; i < 4; i++) rank_support_v5_blob::skip(is);
        suffix_group_starts.load(is);
        C = load_vec<int64_t>(is);
        kmer_prefix_precalc = load_vec<std::pair<int64_t, int64_t>>(is);
        is.read((char *)&precalc_k, 8);
        is.read((char *)&n_nodes, 8);
        is.read((char *)&n_kmers, 8);
        is.read((char *)&k, 8);
        if (!is.good() || C.size() != 4 || n_nodes != A_bits.size() || C_bits.size() != n_nodes ||
            G_bits.size() != n_nodes || T_bits.size() != n_nodes ||
            (suffix_group_starts.size() != 0 && suffix_group_starts.size() != n_nodes) || precalc_k < 0 ||
            precalc_k > 20 || kmer_prefix_precalc.size() != (precalc_k ? ((size_t)1 << (2 * precalc_k)) : 0))
            throw std::runtime_error("Error: Corrupt index file");
    }

private:
    template <typename T>
    static std::vector<T> load_vec(std::istream &is) {   // SBWT.hh:451-459
        int64_t nb = 0;
        is.read((char *)&nb, 8);
        if (!is.good() || nb < 0 || nb % (int64_t)sizeof(T) != 0 || nb > ((int64_t)1 << 45))
            throw std::runtime_error("Error: Corrupt index file");
        std::vector<T> v((size_t)nb / sizeof(T));
        is.read((char *)v.data(), nb);
        return v;
    }
};

}  // namespace sbwt
