// test_api.cpp -- the reference's own API tests for the search path, restated against the host
// mirror (sbwt_amd/csrc/host/SBWT.hh).  Mirrors:
//   api_examples/api_example.cpp:13-29            (search / streaming_search of the example)
//   tests/test_small.hh:24-63                     (check_all_queries, check_streaming_queries)
//   tests/test_small.hh:101-126                   (partial_search)
//   tests/test_small.hh:281-290                   (redundant_dummies: 9 subsets)
//   tests/test_small.hh:324-428                   (serialize -> load -> queries, precalc 2, input with NN)
//   tests/test_large.hh:104-115, 126-170          (streaming == search; forward consistency)
// Prints "OK <n checks>" and exits 0, or prints the failed check and exits 1.
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <set>
#include <sstream>
#include <string>
#include <vector>

#include <chrono>
#include "SBWT.hh"

using namespace sbwt;
using namespace std;

static long n_checks = 0;
#define CHECK(cond)                                                                     \
    do {                                                                                \
        n_checks++;                                                                     \
        if (!(cond)) { cerr << "CHECK FAILED line " << __LINE__ << ": " #cond << endl; exit(1); } \
    } while (0)

static set<string> get_all_kmers(const vector<string>& seqs, int64_t k) {
    set<string> s;
    for (const string& x : seqs)
        for (int64_t i = 0; i + k <= (int64_t)x.size(); i++) {
            string w = x.substr(i, k);
            if (w.find_first_not_of("ACGT") == string::npos) s.insert(w);
        }
    return s;
}

static plain_matrix_sbwt_t build(const vector<string>& seqs, int k, bool streaming, int precalc) {
    PlainMatrixBits b = build_plain_matrix_bits(seqs, k, false, streaming, 1);
    return plain_matrix_sbwt_t(b, precalc);
}

static void check_all_queries(const plain_matrix_sbwt_t& index, const set<string>& truth) {   // test_small.hh:24-43
    int64_t k = index.get_k();
    for (uint64_t mask = 0; mask < (1ull << (2 * k)); mask++) {
        string kmer;
        for (int64_t i = 0; i < k; i++) kmer += "ACGT"[(mask >> (2 * i)) & 3];
        int64_t column = index.search(kmer);
        if (truth.count(kmer)) CHECK(column >= 0); else CHECK(column == -1);
    }
    CHECK(index.search(string(k, 'N')) == -1);
}

static void check_streaming_queries(const plain_matrix_sbwt_t& index, const set<string>& truth, const string& input) {
    vector<int64_t> result = index.streaming_search(input.c_str(), input.size());          // test_small.hh:46-63
    for (size_t i = 0; i < result.size(); i++) {
        bool found = truth.count(input.substr(i, index.get_k()));
        if (found) CHECK(result[i] >= 0); else CHECK(result[i] == -1);
    }
    for (int64_t x : index.streaming_search(string(100, 'N'))) CHECK(x == -1);
}

int main() {
    // --- api_example.cpp ---
    vector<string> ex = {"CCCGTGATGGCTA", "TAATGCTGTAGC", "TGGCTCGTGTAGTCGA"};
    plain_matrix_sbwt_t sbwt = build(ex, 6, true, 4);
    set<string> truth6 = get_all_kmers(ex, 6);
    CHECK(sbwt.search("GATGGC") >= 0);
    vector<int64_t> st = sbwt.streaming_search("TAATGCTGTAGC");
    CHECK(st.size() == 7);
    for (size_t i = 0; i < st.size(); i++) CHECK(st[i] == sbwt.search(string("TAATGCTGTAGC").substr(i, 6)));
    CHECK(sbwt.get_C_array().size() == 4 && sbwt.get_C_array()[0] == 1);
    CHECK((int64_t)sbwt.get_precalc().size() == 256 && sbwt.get_precalc_k() == 4);

    // --- partial_search, test_small.hh:101-126 ---
    {
        plain_matrix_sbwt_t ps = build(ex, 6, false, 0);
        auto res = ps.partial_search("GCAAAA");
        CHECK(res.second == 2);
        // every column in the interval, and only those, is reached by update_sbwt_interval("GC")
        auto I = ps.update_sbwt_interval("GC", {0, ps.number_of_subsets() - 1});
        CHECK(I == res.first);
    }
    // --- redundant_dummies, test_small.hh:281-290 ---
    {
        vector<string> s = {"AAAA", "ACCC", "ACCG", "CCCG", "TTTT"};
        plain_matrix_sbwt_t X = build(s, 4, false, 0);
        CHECK(X.number_of_subsets() == 9);
        check_all_queries(X, get_all_kmers(s, 4));
    }
    // --- small cases, test_small.hh:250-322 ---
    check_all_queries(build(ex, 4, false, 0), get_all_kmers(ex, 4));
    check_all_queries(build({"AAAA", "ACCC", "ACCG", "CCCG"}, 3, false, 2), get_all_kmers({"AAAA", "ACCC", "ACCG", "CCCG"}, 3));
    check_all_queries(build({"ACGTACGTACGT"}, 3, true, 0), get_all_kmers({"ACGTACGTACGT"}, 3));
    // --- serialization, test_small.hh:324-428 ---
    {
        vector<string> s = {"CCCGTGATGGCTA", "TAATGCTGTAGC", "TGGCTCGTGTAGTCGA", "NNAAAAAAAAAAAA"};
        set<string> truth = get_all_kmers(s, 4);
        plain_matrix_sbwt_t v1 = build(s, 4, true, 0);
        v1.do_kmer_prefix_precalc(2);
        stringstream file;
        v1.serialize(file);
        plain_matrix_sbwt_t v2;
        v2.load(file);
        CHECK(v2.get_k() == 4 && v2.get_precalc_k() == 2 && v2.number_of_subsets() == v1.number_of_subsets());
        CHECK(v2.get_subset_rank_structure().A_bits == v1.get_subset_rank_structure().A_bits);
        CHECK(v2.get_streaming_support() == v1.get_streaming_support());
        CHECK(v2.get_precalc() == v1.get_precalc() && v2.get_C_array() == v1.get_C_array());
        check_all_queries(v2, truth);
        for (const string& q : s) check_streaming_queries(v2, truth, q);
        plain_matrix_sbwt_t v3 = v2;   // copies share the immutable device image
        check_streaming_queries(v3, truth, "ACGTTGCATGCATGCCCGTGATGG");
    }
    // --- errors (SBWT.hh:546-547, 370-371, 619-624, 503-505) ---
    {
        plain_matrix_sbwt_t ns = build(ex, 6, false, 0);
        bool thrown = false;
        try { ns.streaming_search("ACGTACGTACGT"); } catch (const runtime_error& e) { thrown = string(e.what()) == "Error: streaming search support not built"; }
        CHECK(thrown);
        thrown = false;
        try { ns.forward(0, 'A'); } catch (const runtime_error& e) { thrown = string(e.what()) == "Error: Streaming support required for SBWT::forward"; }
        CHECK(thrown);
        thrown = false;
        try { ns.do_kmer_prefix_precalc(7); } catch (const runtime_error& e) { thrown = string(e.what()) == "Error: Precalc length is longer than k (7 > 6)"; }
        CHECK(thrown);
        thrown = false;
        try { ns.do_kmer_prefix_precalc(21); } catch (const runtime_error& e) { thrown = string(e.what()).find("Can't precalc longer than 20-mers") != string::npos; }
        CHECK(thrown);
        thrown = false;
        stringstream bad;
        serialize_string("v9.9", bad);
        try { plain_matrix_sbwt_t x; x.load(bad); } catch (const runtime_error& e) { thrown = string(e.what()).find("incompatible version") != string::npos; }
        CHECK(thrown);
    }
    // --- streaming == search and forward consistency on a larger random input (test_large.hh) ---
    {
        srand(247829347);
        string g;
        for (int i = 0; i < 30000; i++) g += "ACGT"[rand() % 4];
        plain_matrix_sbwt_t big = build({g}, 30, true, 8);
        set<string> truth = get_all_kmers({g}, 30);
        string q = g.substr(1000, 400);
        q[100] = 'N'; q[200] = (q[200] == 'A') ? 'C' : 'A';
        vector<int64_t> r = big.streaming_search(q);
        for (size_t i = 0; i < r.size(); i++) CHECK(r[i] == big.search(q.c_str() + i));
        for (int t = 0; t < 200; t++) {
            string kmer = g.substr(rand() % (g.size() - 31), 30);
            int64_t colex = big.search(kmer);
            CHECK(colex >= 0);
            for (char c : string("ACGT")) {
                string next = kmer.substr(1) + c;
                int64_t fwd = big.forward(colex, c);
                if (truth.count(next)) CHECK(fwd == big.search(next)); else CHECK(fwd == -1);
            }
            CHECK(big.get_subset_rank_structure().rank(colex + 1, 'A') - big.get_subset_rank_structure().rank(colex, 'A') ==
                  (big.get_subset_rank_structure().contains(colex, 'A') ? 1 : 0));
        }
        // get_kmer / get_kmer_fast / select (SBWT.hh:700-746, SubsetMatrixSelectSupport.hh; usage as in tests/test_large.hh:52-72)
        SubsetMatrixSelectSupport ss(big.get_subset_rank_structure());
        vector<char> buf(31, 0), buf2(31, 0);
        for (int t = 0; t < 50; t++) {
            string kmer = g.substr(rand() % (g.size() - 31), 30);
            int64_t colex = big.search(kmer);
            big.get_kmer(colex, buf.data());
            big.get_kmer_fast(colex, buf2.data(), ss);
            CHECK(string(buf.data(), 30) == kmer);
            CHECK(string(buf2.data(), 30) == kmer);
            // select is the inverse of rank on set bits
            for (char c : string("ACGT")) {
                if (!big.get_subset_rank_structure().contains(colex, c)) continue;
                int64_t j = big.get_subset_rank_structure().rank(colex, c) + 1;
                CHECK(ss.select(j, c) == colex);
            }
        }
        big.get_kmer(0, buf.data());
        CHECK(string(buf.data(), 30) == string(30, '$'));
        // partial_search: a full k-mer matches k chars; a k-mer with a foreign tail stops early (SBWT.hh:525-537)
        {
            string kmer = g.substr(777, 30);
            auto ps = big.partial_search(kmer);
            CHECK(ps.second == 30 && ps.first.first == ps.first.second && ps.first.first == big.search(kmer));
            auto lower = big.partial_search(string("acgt"));
            auto upper = big.partial_search(string("ACGT"));
            CHECK(lower == upper);                        // every char is upper-cased first (:529)
            auto withn = big.partial_search(string("ACNGT"));
            CHECK(withn.second == 2 && withn.first == big.partial_search(string("AC")).first);
        }
        // the scalar members answer on the HOST (SubsetMatrixRank.hh:31-48, SBWT.hh:389-437): the same values as
        // the GPU gives for a batch of one -- every position of the index incl. pos == n_nodes, every symbol incl. non-ACGT;
        // k-mers that are there, absent ones, lower case (Q1), N inside and outside the prefix table's window (Q3)
        {
            const SubsetMatrixRank& mr = big.get_subset_rank_structure();
            const int64_t n = big.number_of_subsets();
            vector<int64_t> pos;
            vector<char> sym;
            for (int64_t p = 0; p <= n; p += (p < 5000 || p > n - 5000) ? 1 : 37)
                for (char c : string("ACGTNa$")) { pos.push_back(p); sym.push_back(c); }
            vector<int64_t> dev(pos.size());
            mr.rank_batch(pos.data(), sym.data(), (int64_t)pos.size(), dev.data());
            for (size_t q = 0; q < pos.size(); q++) CHECK(mr.rank(pos[q], sym[q]) == dev[q]);
            CHECK(mr.rank(n, 'A') + mr.rank(n, 'C') + mr.rank(n, 'G') + mr.rank(n, 'T') == n - 1);
            for (int t = 0; t < 300; t++) {
                string kmer = g.substr(rand() % (g.size() - 31), 30);
                if (t % 3 == 1) kmer[rand() % 30] = "ACGT"[rand() % 4];
                if (t % 7 == 2) kmer[rand() % 30] = 'N';
                if (t % 7 == 3) kmer[rand() % 8] = 'N';
                if (t % 11 == 4) kmer[rand() % 30] = (char)tolower(kmer[5]);
                CHECK(big.search(kmer) == big.search_on_device(kmer.c_str()));
                const int64_t cut = rand() % 31;
                auto I = big.update_sbwt_interval(kmer.c_str(), cut, {0, n - 1});
                CHECK(I == big.update_sbwt_interval_on_device(kmer.c_str(), cut, {0, n - 1}));
                CHECK(big.update_sbwt_interval(kmer.c_str() + cut, 30 - cut, I) ==
                      big.update_sbwt_interval_on_device(kmer.c_str() + cut, 30 - cut, I));
            }
        }
        // cost of the scalar API: on the host (rank, search of one k-mer) and as batches of one through the GPU (what they were
        // before round 5; streaming_search of one read still is): printed, not asserted
        {
            string kmer = g.substr(4242, 30);
            const int reps = 2000;
            int64_t acc = 0;
            auto timeit = [&](auto&& f, int n_rep) {
                auto t0 = std::chrono::steady_clock::now();
                for (int t = 0; t < n_rep; t++) acc += f(t);
                return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n_rep;
            };
            const SubsetMatrixRank& mr = big.get_subset_rank_structure();
            const double us_search = timeit([&](int) { return big.search(kmer.c_str()); }, 100 * reps);
            const double us_rank = timeit([&](int t) { return mr.rank(1000 + t % 20000, 'C'); }, 100 * reps);
            const double us_search_dev = timeit([&](int) { return big.search_on_device(kmer.c_str()); }, reps);
            const double us_rank_dev = timeit([&](int t) { return mr.rank_on_device(1000 + t, 'C'); }, reps);
            string read = g.substr(5000, 150);
            const double us_stream = timeit([&](int) { return big.streaming_search(read.c_str(), 150)[0]; }, reps);
            printf("scalar search(): %.3f us/call on the host (%.1f us as a GPU batch of one), scalar rank(): %.3f us/call "
                   "(%.1f us), streaming_search of one 150-base read (GPU batch of one): %.1f us (checksum %ld)\n",
                   us_search, us_search_dev, us_rank, us_rank_dev, us_stream, (long)acc);
        }
        // batch API == scalar API
        vector<int64_t> off = {0, 150, 150, 400}, ooff = {0, 121, 121, 121 + 221}, out(342, -7);
        big.streaming_search_batch(q.c_str(), off.data(), 3, out.data(), ooff.data());
        vector<int64_t> a = big.streaming_search(q.c_str(), 150), b = big.streaming_search(q.c_str() + 150, 250);
        for (size_t i = 0; i < a.size(); i++) CHECK(out[i] == a[i]);
        for (size_t i = 0; i < b.size(); i++) CHECK(out[121 + i] == b[i]);
    }
    printf("OK %ld checks\n", n_checks);
    return 0;
}
