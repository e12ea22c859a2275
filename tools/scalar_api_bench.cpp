// scalar_api_bench.cpp -- what ONE call of the reference's scalar API costs through the C++ mirror (host/SBWT.hh):
// SBWT::search of one k-mer and SubsetMatrixRank::rank of one position (both on the host), the same two as GPU
// batches of one (what they were before round 5), and streaming_search of one 150-base read (a GPU batch of one).
// usage: scalar_api_bench <index.sbwt> <read of at least 150 bases>      prints one JSON line.
// Built on the fly by bench.py's end-to-end leg (g++ against the host headers and libsbwtgpu.so).
#include <chrono>
#include <cstdio>
#include <string>
#include "SBWT.hh"

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s index.sbwt READ\n", argv[0]); return 2; }
    try {
        sbwt::set_log_level(sbwt::LogLevel::OFF);
        std::ifstream in(argv[1], std::ios::binary);
        if (!in.good()) throw std::runtime_error("cannot open the index");
        const std::string variant = sbwt::load_string(in);
        sbwt::plain_matrix_sbwt_t idx;
        idx.load(in);
        const std::string read = argv[2];
        const int64_t k = idx.get_k(), n = idx.number_of_subsets();
        if ((int64_t)read.size() < k) throw std::runtime_error("read shorter than k");
        const sbwt::SubsetMatrixRank &mr = idx.get_subset_rank_structure();
        int64_t acc = 0;
        auto timeit = [&](auto &&f, int reps) {
            const auto t0 = std::chrono::steady_clock::now();
            for (int t = 0; t < reps; t++) acc += f(t);
            return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        };
        const char *kmer = read.c_str();
        const double search_host = timeit([&](int t) { return idx.search(kmer + t % (read.size() - k + 1)); }, 200000);
        const double rank_host = timeit([&](int t) { return mr.rank((int64_t)((uint64_t)t * 2654435761ull % (uint64_t)(n + 1)), "ACGT"[t & 3]); }, 2000000);
        const double search_dev = timeit([&](int t) { return idx.search_on_device(kmer + t % (read.size() - k + 1)); }, 2000);
        const double rank_dev = timeit([&](int t) { return mr.rank_on_device(t % (n + 1), 'C'); }, 2000);
        const double stream_dev = idx.has_streaming_query_support()
                                      ? timeit([&](int) { return idx.streaming_search(read.c_str(), (int64_t)read.size())[0]; }, 2000) : -1.0;
        printf("{\"search_one_kmer_us\": %.4f, \"rank_us\": %.4f, \"search_one_kmer_as_gpu_batch_of_one_us\": %.2f, "
               "\"rank_as_gpu_batch_of_one_us\": %.2f, \"streaming_search_one_read_us\": %.2f, \"read_bases\": %zu, \"k\": %lld, "
               "\"columns\": %lld, \"checksum\": %lld}\n",
               search_host, rank_host, search_dev, rank_dev, stream_dev, read.size(), (long long)k, (long long)n, (long long)acc);
    } catch (const std::exception &e) {
        fprintf(stderr, "scalar_api_bench: %s\n", e.what());
        return 1;
    }
    return 0;
}
