// SBWT.hh -- host-side C++ mirror of the reference's plain-matrix index interface
//   sbwt::SubsetMatrixRank   (reference include/sbwt/SubsetMatrixRank.hh:13-127)
//   sbwt::SBWT               (reference include/sbwt/SBWT.hh:31-332)
//   sbwt::plain_matrix_sbwt_t (reference include/sbwt/variants.hh:19)
// with the same method names, argument meaning and error behaviour, so that code written against
// the reference's search API compiles against this header.  Every BATCH is executed by the HIP
// library through the C ABI of include/sbwtgpu.h and nothing else: use the *_batch members for
// throughput; without a GPU no index can be made at all.  The SCALAR members that the reference's own
// code calls once per step -- SubsetMatrixRank::rank / contains (SBWT.hh:347-349,430-431,572-573),
// SBWT::search of ONE k-mer and update_sbwt_interval (api_examples/api_example.cpp:26-29) -- run on
// the host from the rank directory this class keeps beside the bit vectors ("scalar rank()
// runs on host, batched rank goes to the GPU"; round 5): a kernel launch per rank cost 30-60 us, the
// directory answers in under 0.1 us.  tests/cpp/test_api.cpp holds the two paths against each other.
#pragma once
#include <cstdint>
#include <algorithm>
#include <cstring>
#include <fstream>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>
#include <chrono>
#include <iostream>

#include "../../../include/sbwtgpu.h"
#include "bitvector.hh"
#include "globals.hh"
#include "index_builder.hh"
#include "index_file.hh"
#include "seqio.hh"

namespace sbwt {

namespace detail {
inline int &default_device() { static int d = 0; return d; }
inline void gpu_check(int rc) {
    if (rc != SBWTGPU_OK) {
        const char *m = sbwtgpu_last_error();
        // reference messages already start with "Error:"; pass them through unchanged
        throw std::runtime_error(m && *m ? std::string(m) : "sbwtgpu error " + std::to_string(rc));
    }
}
struct DeviceIndex {
    sbwtgpu_index *h = nullptr;
    ~DeviceIndex() { sbwtgpu_index_destroy(h); }
};
}  // namespace detail

// Selects the HIP device new indexes are placed on (default 0).
inline void set_default_device(int device) { detail::default_device() = device; }

// The plain-matrix columns of a set of sequences: on the GPU for k <= 64 (sbwtgpu_build_plain_matrix: radix sort of the
// packed k-mers and searches in the sorted array), with the multi-threaded host builder (index_builder.hh) for longer
// k-mers.  Both give the reference constructors' bits (NodeBOSSInMemoryConstructor.hh:98-213).
inline PlainMatrixBits build_plain_matrix_bits_any(const std::vector<std::string> &seqs, int k, bool add_revcomp,
                                                   bool build_streaming_support, int n_threads) {
    if (k < 2 || k > 64) return build_plain_matrix_bits(seqs, k, add_revcomp, build_streaming_support, n_threads);
    std::vector<const char *> ptr(seqs.size());
    std::vector<int64_t> len(seqs.size());
    for (size_t i = 0; i < seqs.size(); i++) { ptr[i] = seqs[i].data(); len[i] = (int64_t)seqs[i].size(); }
    sbwtgpu_plain_matrix_bits b;
    const int rc = sbwtgpu_build_plain_matrix(ptr.data(), len.data(), (int64_t)seqs.size(), k, add_revcomp ? 1 : 0,
                                              build_streaming_support ? 1 : 0, detail::default_device(), &b);
    // the device builder holds the text, its packed form, two key arrays and the sort's temporary at once; an input that
    // does not fit (SBWTGPU_ERR_OOM: its fit limit and its dummy-record limit both map to it) is built by the host
    // builder, and the log says so.  Any other failure (a kernel fault, a launch error) is an error, not a reason to
    // fall back silently.  (This is index CONSTRUCTION; queries have no host path.)
    if (rc == SBWTGPU_ERR_OOM) {
        const char *m = sbwtgpu_last_error();
        write_log(std::string("Device builder: ") + (m && *m ? m : "out of device memory") + "; building the columns on the host",
                  LogLevel::MAJOR);
        return build_plain_matrix_bits(seqs, k, add_revcomp, build_streaming_support, n_threads);
    }
    detail::gpu_check(rc);
    PlainMatrixBits out;
    out.n_nodes = b.n_nodes; out.n_kmers = b.n_kmers; out.k = b.k;
    const size_t nw = (size_t)((b.n_nodes + 63) / 64);
    out.A.assign(b.A_bits, b.A_bits + nw); out.C.assign(b.C_bits, b.C_bits + nw);
    out.G.assign(b.G_bits, b.G_bits + nw); out.T.assign(b.T_bits, b.T_bits + nw);
    if (b.suffix_group_starts) out.ssup.assign(b.suffix_group_starts, b.suffix_group_starts + nw);
    sbwtgpu_free_plain_matrix(&b);
    return out;
}

class SubsetMatrixRank {
public:
    // public like the reference (SubsetMatrixRank.hh:19-28): the four rows and their rank directories (the host's, for
    // scalar calls and for serialize(); the GPU image has its own block counts)
    bit_vector A_bits, C_bits, G_bits, T_bits;
    rank_support_v5_blob A_bits_rs, C_bits_rs, G_bits_rs, T_bits_rs;

    SubsetMatrixRank() {}
    SubsetMatrixRank(const bit_vector &A, const bit_vector &C, const bit_vector &G, const bit_vector &T)
        : A_bits(A), C_bits(C), G_bits(G), T_bits(T) { init_supports(); }

    // Count of character c in subsets up to pos, not including pos (SubsetMatrixRank.hh:31-37): one position, on the host
    int64_t rank(int64_t pos, char c) const {
        switch (c) {
            case 'A': return A_bits_rs.rank(A_bits, pos);
            case 'C': return C_bits_rs.rank(C_bits, pos);
            case 'G': return G_bits_rs.rank(G_bits, pos);
            case 'T': return T_bits_rs.rank(T_bits, pos);
            default: return 0;
        }
    }
    // the same through the GPU as a batch of one (what rank() was before round 5; tests compare the two)
    int64_t rank_on_device(int64_t pos, char c) const {
        int64_t out = 0;
        detail::gpu_check(sbwtgpu_rank_batch(device().h, &pos, &c, 1, &out));
        return out;
    }
    void rank_batch(const int64_t *pos, const char *sym, int64_t n, int64_t *out) const {
        detail::gpu_check(sbwtgpu_rank_batch(device().h, pos, sym, n, out));
    }
    bool contains(int64_t pos, char c) const {   // SubsetMatrixRank.hh:39-48 (plain bit access)
        switch (c) {
            case 'A': return A_bits[pos];
            case 'C': return C_bits[pos];
            case 'G': return G_bits[pos];
            case 'T': return T_bits[pos];
            default: return false;
        }
    }
    int64_t serialize(std::ostream &os) const {   // SubsetMatrixRank.hh:86-100
        int64_t written = 0;
        written += A_bits.serialize(os);
        written += C_bits.serialize(os);
        written += G_bits.serialize(os);
        written += T_bits.serialize(os);
        for (const rank_support_v5_blob *rs : {&A_bits_rs, &C_bits_rs, &G_bits_rs, &T_bits_rs}) written += rs->serialize(os);
        write_log("MatrixRank bit vectors total " + std::to_string((double)written / (double)A_bits.size() * 8) +
                      " bits total per node",
                  LogLevel::MINOR);
        return written;
    }
    void load(std::istream &is) {                  // SubsetMatrixRank.hh:102-125
        A_bits.load(is);
        C_bits.load(is);
        G_bits.load(is);
        T_bits.load(is);
        for (int i = 0; i < 4; i++) rank_support_v5_blob::skip(is);      // rebuilt from the bits: ranks depend on nothing else
        init_supports();
        dev_.reset();
    }
    // used by SBWT to share its (full) device image instead of building a second one
    void attach(const std::shared_ptr<detail::DeviceIndex> &d) const { dev_ = d; }
    // the device image serving this structure (SubsetMatrixSelectSupport shares it)
    std::shared_ptr<detail::DeviceIndex> device_image() const { device(); return dev_; }

private:
    void init_supports() {                         // SubsetMatrixRank.hh:52-58 (sdsl::util::init_support)
        A_bits_rs.build(A_bits); C_bits_rs.build(C_bits); G_bits_rs.build(G_bits); T_bits_rs.build(T_bits);
    }
    const detail::DeviceIndex &device() const {
        if (!dev_) {
            sbwtgpu_index_desc d;
            memset(&d, 0, sizeof(d));
            d.n_nodes = A_bits.size();
            d.A_bits = A_bits.data(); d.C_bits = C_bits.data(); d.G_bits = G_bits.data(); d.T_bits = T_bits.data();
            d.k = 1;
            auto p = std::make_shared<detail::DeviceIndex>();
            detail::gpu_check(sbwtgpu_index_create(&d, detail::default_device(), &p->h));
            dev_ = p;
        }
        return *dev_;
    }
    mutable std::shared_ptr<detail::DeviceIndex> dev_;
};

// SubsetMatrixSelectSupport.hh:16-33: select on the four rows.  The reference builds sdsl select supports that
// point into the SubsetMatrixRank; here select is answered from the same device image as rank (the block counts
// are searched, sbwt_api_kernels.hip k_select), so this object only shares that image.
class SubsetMatrixSelectSupport {
public:
    SubsetMatrixSelectSupport() {}
    explicit SubsetMatrixSelectSupport(const SubsetMatrixRank &mr) : dev_(mr.device_image()) {}
    int64_t select(int64_t pos, char c) const {             // SubsetMatrixSelectSupport.hh:27-33
        if (!dev_) throw std::runtime_error("Error: empty select support");
        int64_t out = 0;
        detail::gpu_check(sbwtgpu_select_batch(dev_->h, &pos, &c, 1, &out));
        return out;
    }
    void select_batch(const int64_t *pos, const char *sym, int64_t n, int64_t *out) const {
        if (!dev_) throw std::runtime_error("Error: empty select support");
        detail::gpu_check(sbwtgpu_select_batch(dev_->h, pos, sym, n, out));
    }
private:
    std::shared_ptr<detail::DeviceIndex> dev_;
};

class SBWT {
public:
    struct BuildConfig {                           // SBWT.hh:61-71
        std::vector<std::string> input_files;
        int k = 30;
        bool build_streaming_support = true;
        int n_threads = 1;
        int min_abundance = 1;                     // only 1 is supported by the in-memory builder
        int max_abundance = 1000000000;
        int ram_gigas = 2;
        int precalc_k = 0;
        std::string temp_dir = ".";
    };

    SBWT() : n_nodes(0), n_kmers(0), k(0) {}

    // SBWT.hh:335-353
    SBWT(const bit_vector &A_bits, const bit_vector &C_bits, const bit_vector &G_bits, const bit_vector &T_bits,
         const bit_vector &streaming_support, int64_t k, int64_t number_of_kmers, int64_t precalc_k)
        : subset_rank(A_bits, C_bits, G_bits, T_bits), suffix_group_starts(streaming_support),
          n_nodes(A_bits.size()), n_kmers(number_of_kmers), k(k) {
        check_precalc(precalc_k);
        this->precalc_k = precalc_k;
        make_device(nullptr);
    }

    // SBWT.hh:355-366 -- construction from sequence files.  The reference runs KMC + external sorting
    // (out of scope); this uses the in-memory sort-based builder of index_builder.hh.
    explicit SBWT(const BuildConfig &config) : n_nodes(0), n_kmers(0), k(0) {
        if (config.min_abundance != 1 || config.max_abundance < 1000000000)
            throw std::runtime_error("Error: abundance filtering is not supported by the in-memory builder");
        std::vector<std::string> seqs;
        for (const std::string &f : config.input_files) {
            seq_io::Reader reader(f);
            for (;;) {
                int64_t len = reader.get_next_read_to_buffer();
                if (len == 0) break;
                seqs.emplace_back(reader.read_buf, (size_t)len);
            }
        }
        PlainMatrixBits b = build_plain_matrix_bits_any(seqs, config.k, false, config.build_streaming_support, config.n_threads);
        adopt_bits(b, config.precalc_k);
    }

    // from the builder's output
    SBWT(const PlainMatrixBits &b, int64_t precalc_k) : n_nodes(0), n_kmers(0), k(0) { adopt_bits(b, precalc_k); }

    // ---- accessors (SBWT.hh:111-157,253) ----
    const SubsetMatrixRank &get_subset_rank_structure() const { return subset_rank; }
    const bit_vector &get_streaming_support() const { return suffix_group_starts; }
    const std::vector<int64_t> &get_C_array() const { return C; }
    const std::vector<std::pair<int64_t, int64_t>> &get_precalc() const { return kmer_prefix_precalc; }
    int64_t get_precalc_k() const { return precalc_k; }
    int64_t number_of_subsets() const { return n_nodes; }
    int64_t number_of_kmers() const { return n_kmers; }
    int64_t get_k() const { return k; }
    bool has_streaming_query_support() const { return suffix_group_starts.size() > 0; }
    const sbwtgpu_index *device_handle() const { return dev_ ? dev_->h : nullptr; }

    // ---- queries ----
    int64_t search(const std::string &kmer) const { return search(kmer.c_str()); }   // SBWT.hh:383-387
    // One k-mer (reads exactly k bytes): on the host, the reference's own steps (SBWT.hh:389-415) over the host rank
    // directory -- the file's prefix table, then k - p interval updates
    int64_t search(const char *kmer) const {
        need_device();
        std::pair<int64_t, int64_t> I;
        if (precalc_k > 0) {                               // SBWT.hh:394-405: raw chars, first char in the low bits
            uint64_t precalc_idx = 0;
            for (int64_t i = 0; i < precalc_k; i++) {
                const int char_idx = DNA_to_char_idx(kmer[i]);
                if (char_idx == -1) return -1;
                precalc_idx |= (uint64_t)char_idx << (2 * i);
            }
            I = host_update_interval(kmer + precalc_k, k - precalc_k, kmer_prefix_precalc[(size_t)precalc_idx]);
        } else {
            I = host_update_interval(kmer, k, {0, n_nodes - 1});
        }
        if (I.first == -1) return -1;
        if (I.first != I.second) bug_exit(SBWTGPU_ERR_NOT_SINGLETON);                 // SBWT.hh:410-413
        return I.first;
    }
    // the same through the GPU as a batch of one (what search() was before round 5; tests compare the two)
    int64_t search_on_device(const char *kmer) const {
        int64_t off[2] = {0, k}, ooff[2] = {0, 1}, out = -1;
        int rc = sbwtgpu_search_batch(need_device(), kmer, off, 1, &out, ooff);
        bug_exit(rc);
        detail::gpu_check(rc);
        return out;
    }
    std::vector<int64_t> streaming_search(const std::string &input) const {          // SBWT.hh:583-586
        return streaming_search(input.c_str(), (int64_t)input.size());
    }
    std::vector<int64_t> streaming_search(const char *input, int64_t len) const {    // SBWT.hh:544-581
        if (suffix_group_starts.size() == 0) throw std::runtime_error("Error: streaming search support not built");
        std::vector<int64_t> ans;
        if (len < k) return ans;
        ans.resize((size_t)(len - k + 1));
        int64_t off[2] = {0, len}, ooff[2] = {0, len - k + 1};
        int rc = sbwtgpu_streaming_search_batch(need_device(), input, off, 1, ans.data(), ooff);
        bug_exit(rc);
        detail::gpu_check(rc);
        return ans;
    }
    // Batched forms (new): read r = bases[read_off[r]..read_off[r+1]), results at out[out_off[r]..].
    // With replicas (use_devices) the reads are split into contiguous ranges balanced by bases, one
    // host thread per GPU; results land in place, so the output order never changes.
    void streaming_search_batch(const char *bases, const int64_t *read_off, int64_t n_reads, int64_t *out,
                                const int64_t *out_off) const {
        if (suffix_group_starts.size() == 0) throw std::runtime_error("Error: streaming search support not built");
        run_sharded(true, bases, read_off, n_reads, out, out_off);
    }
    void search_batch(const char *bases, const int64_t *read_off, int64_t n_reads, int64_t *out,
                      const int64_t *out_off) const {
        run_sharded(false, bases, read_off, n_reads, out, out_off);
    }
    // The same with int32 results (for consumers that only print or store the ranks, like print_vector,
    // sbwt_search.cpp:21-43): narrowed on the device, half the bytes over PCIe.  One device; fewer than 2^31 columns.
    void search_batch_i32(bool streaming, const char *bases, const int64_t *read_off, int64_t n_reads, int32_t *out,
                          const int64_t *out_off) const {
        if (streaming && suffix_group_starts.size() == 0) throw std::runtime_error("Error: streaming search support not built");
        const int rc = streaming ? sbwtgpu_streaming_search_batch_i32(need_device(), bases, read_off, n_reads, out, out_off)
                                 : sbwtgpu_search_batch_i32(need_device(), bases, read_off, n_reads, out, out_off);
        bug_exit(rc);
        detail::gpu_check(rc);
    }
    // The whole inner loop of `sbwt search` for one batch: searches every read (streaming_search when
    // the index has streaming support, else the per-k-mer search loop) and returns the output text in
    // the reference's format (print_vector, sbwt_search.cpp:21-43), formatted on the GPU.  One text
    // piece per GPU shard, in read order.  Returns the number of k-mers searched.
    struct TextPiece {
        char *data = nullptr;
        int64_t size = 0;
        TextPiece() = default;
        TextPiece(const TextPiece &) = delete;
        TextPiece &operator=(const TextPiece &) = delete;
        TextPiece(TextPiece &&o) noexcept : data(o.data), size(o.size) { o.data = nullptr; o.size = 0; }
        ~TextPiece() { sbwtgpu_free_host(data); }
    };
    int64_t search_text_batch(const char *bases, const int64_t *read_off, int64_t n_reads,
                              std::vector<TextPiece> &pieces) const {
        const sbwtgpu_index *root = need_device();
        const bool streaming = has_streaming_query_support();
        const int G = (n_reads < 2 * number_of_devices()) ? 1 : number_of_devices();
        std::vector<int64_t> cut = shard_cuts(read_off, n_reads, G);
        pieces.clear();
        pieces.resize((size_t)G);
        std::vector<int> rcs((size_t)G, SBWTGPU_OK);
        std::vector<std::string> errs((size_t)G);
        std::vector<int64_t> nq((size_t)G, 0);
        auto work = [&](int g) {
            const int64_t lo = cut[(size_t)g], hi = cut[(size_t)g + 1];
            const sbwtgpu_index *h = (G == 1) ? root : replicas_[(size_t)g].h;
            rcs[(size_t)g] = sbwtgpu_search_text_batch(h, bases, read_off + lo, hi - lo, streaming ? 1 : 0,
                                                      &pieces[(size_t)g].data, &pieces[(size_t)g].size, &nq[(size_t)g]);
            if (rcs[(size_t)g] != SBWTGPU_OK) errs[(size_t)g] = sbwtgpu_last_error();
        };
        if (G == 1) {
            work(0);
        } else {
            std::vector<std::thread> th;
            for (int g = 0; g < G; g++) th.emplace_back(work, g);
            for (auto &t : th) t.join();
        }
        int64_t total = 0;
        for (int g = 0; g < G; g++) {
            bug_exit(rcs[(size_t)g]);
            if (rcs[(size_t)g] != SBWTGPU_OK)
                throw std::runtime_error(errs[(size_t)g].empty() ? "sbwtgpu error " + std::to_string(rcs[(size_t)g])
                                                                 : errs[(size_t)g]);
            total += nq[(size_t)g];
        }
        return total;
    }
    // The same on one device, streamed: `sink(text, bytes)` receives consecutive pieces of the output text in order, each
    // straight out of a pinned staging buffer that is valid only during the call.  Returns the number of k-mers searched.
    template <typename Sink>
    int64_t search_text_stream(const char *bases, const int64_t *read_off, int64_t n_reads, Sink &&sink) const {
        const sbwtgpu_index *root = need_device();
        struct Ctx { Sink *s; std::exception_ptr err; } ctx{&sink, nullptr};
        int64_t nq = 0;
        const int rc = sbwtgpu_search_text_stream(
            root, bases, read_off, n_reads, has_streaming_query_support() ? 1 : 0,
            [](void *c, const char *text, int64_t bytes) -> int {
                Ctx *x = static_cast<Ctx *>(c);
                try { (*x->s)(text, bytes); } catch (...) { x->err = std::current_exception(); return 1; }
                return 0;
            },
            &ctx, &nq);
        if (ctx.err) std::rethrow_exception(ctx.err);
        bug_exit(rc);
        detail::gpu_check(rc);
        return nq;
    }
    int number_of_devices_in_use() const { return number_of_devices(); }
    // Replicates the device image onto the listed HIP devices (one RCCL broadcast over xGMI);
    // the first entry should be the device the index was created on.  Listing a device twice is allowed
    // (two host threads share it) and is how the sharding is tested on a single-GPU box.
    void use_devices(const std::vector<int> &devices) {
        need_device();
        replicas_.clear();
        if (devices.size() <= 1) return;
        std::vector<sbwtgpu_index *> hs(devices.size(), nullptr);
        detail::gpu_check(sbwtgpu_index_bcast(dev_->h, (int)devices.size(), devices.data(), hs.data()));
        // one owner per distinct handle: duplicates of a device share its handle (and its owner), and the root's
        // entries hold dev_ itself, so no handle can be freed while a replica entry still points at it
        std::vector<std::pair<sbwtgpu_index *, std::shared_ptr<detail::DeviceIndex>>> owners;
        for (sbwtgpu_index *h : hs) {
            std::shared_ptr<detail::DeviceIndex> p;
            if (h == dev_->h) p = dev_;
            for (auto &o : owners)
                if (o.first == h) p = o.second;
            if (!p) {
                p = std::make_shared<detail::DeviceIndex>();
                p->h = h;
                owners.emplace_back(h, p);
            }
            replicas_.push_back({p, h});
        }
    }
    int number_of_devices() const { return replicas_.empty() ? 1 : (int)replicas_.size(); }
    std::pair<int64_t, int64_t> update_sbwt_interval(const std::string &S, std::pair<int64_t, int64_t> I) const {
        return update_sbwt_interval(S.c_str(), (int64_t)S.size(), I);
    }
    std::pair<int64_t, int64_t> update_sbwt_interval(const char *S, int64_t S_length,
                                                     std::pair<int64_t, int64_t> I) const {   // SBWT.hh:422-437: on the host
        need_device();
        return host_update_interval(S, S_length, I);
    }
    // the same through the GPU as a batch of one
    std::pair<int64_t, int64_t> update_sbwt_interval_on_device(const char *S, int64_t S_length,
                                                               std::pair<int64_t, int64_t> I) const {
        int64_t off[2] = {0, S_length};
        detail::gpu_check(sbwtgpu_update_interval_batch(need_device(), S, off, 1, &I.first, &I.second));
        return I;
    }
    // SBWT.hh:525-542 -- the whole loop is one kernel launch (one lane walks the query)
    std::pair<std::pair<int64_t, int64_t>, int64_t> partial_search(const char *input, int64_t len) const {
        int64_t off[2] = {0, len}, l = 0, r = 0, matched = 0;
        detail::gpu_check(sbwtgpu_partial_search_batch(need_device(), input, off, 1, &l, &r, &matched));
        return {{l, r}, matched};
    }
    // batch form: query q = bases[off[q] .. off[q+1])
    void partial_search_batch(const char *bases, const int64_t *off, int64_t n, int64_t *first, int64_t *second,
                              int64_t *matched) const {
        detail::gpu_check(sbwtgpu_partial_search_batch(need_device(), bases, off, n, first, second, matched));
    }
    // SBWT.hh:700-725 / :727-746: the k-mer of a column into buf (k chars, '$' for dummy positions, no NUL).  Both run
    // the same device kernel (select inside the block counts); `ss` is accepted for source compatibility.
    void get_kmer(int64_t colex_rank, char *buf) const {
        detail::gpu_check(sbwtgpu_get_kmer_batch(need_device(), &colex_rank, 1, buf));
    }
    template <typename subset_select_support_t>
    void get_kmer_fast(int64_t colex_rank, char *buf, const subset_select_support_t &) const {
        get_kmer(colex_rank, buf);
    }
    void get_kmer_batch(const int64_t *colex_ranks, int64_t n, char *out) const {
        detail::gpu_check(sbwtgpu_get_kmer_batch(need_device(), colex_ranks, n, out));
    }
    std::pair<std::pair<int64_t, int64_t>, int64_t> partial_search(const std::string &input) const {
        return partial_search(input.c_str(), (int64_t)input.size());
    }
    int64_t forward(int64_t node, char c) const {                                    // SBWT.hh:368-381
        if (!has_streaming_query_support()) throw std::runtime_error("Error: Streaming support required for SBWT::forward");
        int64_t out = -1;
        detail::gpu_check(sbwtgpu_forward_batch(need_device(), &node, &c, 1, &out));
        return out;
    }

    // SBWT.hh:616-645: (re)computes the prefix table on the device and mirrors it on the host
    void do_kmer_prefix_precalc(int64_t prefix_length) {
        if (prefix_length == 0) return;
        check_precalc(prefix_length);
        precalc_k = prefix_length;
        make_device(nullptr);
    }

    // ---- (de)serialisation, reference file format (SBWT.hh:462-522) ----
    int64_t serialize(std::ostream &os) const {
        IndexFileData f;
        f.A_bits = subset_rank.A_bits; f.C_bits = subset_rank.C_bits;
        f.G_bits = subset_rank.G_bits; f.T_bits = subset_rank.T_bits;
        f.suffix_group_starts = suffix_group_starts;
        f.C = C;
        f.kmer_prefix_precalc = kmer_prefix_precalc;
        f.precalc_k = precalc_k; f.n_nodes = n_nodes; f.n_kmers = n_kmers; f.k = k;
        return f.serialize(os);
    }
    int64_t serialize(const std::string &filename) const {
        std::ofstream out(filename, std::ios::binary);
        if (!out.good()) throw std::runtime_error("Error opening file: " + filename);
        return serialize(out);
    }
    void load(std::istream &is) {
        IndexFileData f;
        const auto t_file0 = std::chrono::steady_clock::now();
        f.load(is);
        if (getenv("SBWT_CLI_TIMING"))
            std::cerr << "timing: index file read "
                      << std::chrono::duration<double>(std::chrono::steady_clock::now() - t_file0).count() << " s" << std::endl;
        subset_rank = SubsetMatrixRank(f.A_bits, f.C_bits, f.G_bits, f.T_bits);
        suffix_group_starts = f.suffix_group_starts;
        precalc_k = f.precalc_k; n_nodes = f.n_nodes; n_kmers = f.n_kmers; k = f.k;
        // the file's prefix table is uploaded as is; C comes back from the device image
        make_device(precalc_k ? (const int64_t *)f.kmer_prefix_precalc.data() : nullptr);
    }
    void load(const std::string &filename) {
        std::ifstream in(filename, std::ios::binary);
        if (!in.good()) throw std::runtime_error("Error opening file: " + filename);
        load(in);
    }

private:
    SubsetMatrixRank subset_rank;
    bit_vector suffix_group_starts;
    std::vector<int64_t> C;
    std::vector<std::pair<int64_t, int64_t>> kmer_prefix_precalc;
    int64_t precalc_k = 0;
    int64_t n_nodes, n_kmers, k;
    std::shared_ptr<detail::DeviceIndex> dev_;
    struct Replica { std::shared_ptr<detail::DeviceIndex> owner; sbwtgpu_index *h; };
    std::vector<Replica> replicas_;

    // contiguous shards balanced by bases: shard g = reads [cut[g], cut[g+1])
    static std::vector<int64_t> shard_cuts(const int64_t *read_off, int64_t n_reads, int G) {
        std::vector<int64_t> cut((size_t)G + 1, 0);
        const int64_t total = n_reads ? read_off[n_reads] - read_off[0] : 0;
        for (int g = 1; g < G; g++) {
            const int64_t target = read_off[0] + total * g / G;
            cut[(size_t)g] = std::lower_bound(read_off, read_off + n_reads + 1, target) - read_off;
            if (cut[(size_t)g] < cut[(size_t)g - 1]) cut[(size_t)g] = cut[(size_t)g - 1];
            if (cut[(size_t)g] > n_reads) cut[(size_t)g] = n_reads;
        }
        cut[(size_t)G] = n_reads;
        return cut;
    }
    void run_one(bool streaming, const sbwtgpu_index *h, const char *bases, const int64_t *read_off, int64_t n_reads,
                 int64_t *out, const int64_t *out_off, int *rc_out, std::string *err) const {
        int rc = streaming ? sbwtgpu_streaming_search_batch(h, bases, read_off, n_reads, out, out_off)
                           : sbwtgpu_search_batch(h, bases, read_off, n_reads, out, out_off);
        *rc_out = rc;
        if (rc != SBWTGPU_OK) *err = sbwtgpu_last_error();   // thread-local: capture it on this thread
    }
    void run_sharded(bool streaming, const char *bases, const int64_t *read_off, int64_t n_reads, int64_t *out,
                     const int64_t *out_off) const {
        const sbwtgpu_index *root = need_device();
        const int G = number_of_devices();
        std::vector<int> rcs((size_t)G, SBWTGPU_OK);
        std::vector<std::string> errs((size_t)G);
        if (G == 1 || n_reads < 2 * G) {
            run_one(streaming, root, bases, read_off, n_reads, out, out_off, &rcs[0], &errs[0]);
        } else {
            std::vector<int64_t> cut = shard_cuts(read_off, n_reads, G);
            std::vector<std::thread> th;
            for (int g = 0; g < G; g++) {
                const int64_t lo = cut[(size_t)g], hi = cut[(size_t)g + 1];
                if (hi == lo) continue;
                th.emplace_back([&, g, lo, hi] {
                    run_one(streaming, replicas_[(size_t)g].h, bases, read_off + lo, hi - lo, out, out_off + lo,
                            &rcs[(size_t)g], &errs[(size_t)g]);
                });
            }
            for (auto &t : th) t.join();
        }
        for (int g = 0; g < G; g++) {
            bug_exit(rcs[(size_t)g]);
            if (rcs[(size_t)g] != SBWTGPU_OK)
                throw std::runtime_error(errs[(size_t)g].empty() ? "sbwtgpu error " + std::to_string(rcs[(size_t)g])
                                                                 : errs[(size_t)g]);
        }
    }

    void check_precalc(int64_t p) const {                 // SBWT.hh:619-624
        if (p > 20)
            throw std::runtime_error("Error: Can't precalc longer than 20-mers (would take over 4^20 = 2^40 bytes");
        if (p > k)
            throw std::runtime_error("Error: Precalc length is longer than k (" + std::to_string(p) + " > " +
                                     std::to_string(k) + ")");
    }
    void adopt_bits(const PlainMatrixBits &b, int64_t precalc) {
        subset_rank = SubsetMatrixRank(bit_vector(b.A.data(), b.n_nodes), bit_vector(b.C.data(), b.n_nodes),
                                       bit_vector(b.G.data(), b.n_nodes), bit_vector(b.T.data(), b.n_nodes));
        suffix_group_starts = b.ssup.empty() ? bit_vector() : bit_vector(b.ssup.data(), b.n_nodes);
        n_nodes = b.n_nodes;
        n_kmers = b.n_kmers;
        k = b.k;
        check_precalc(precalc);
        precalc_k = precalc;
        make_device(nullptr);
    }
    // Builds the device image; C array and prefix table come back from the device.
    void make_device(const int64_t *file_precalc) {
        sbwtgpu_index_desc d;
        memset(&d, 0, sizeof(d));
        d.n_nodes = n_nodes;
        d.A_bits = subset_rank.A_bits.data(); d.C_bits = subset_rank.C_bits.data();
        d.G_bits = subset_rank.G_bits.data(); d.T_bits = subset_rank.T_bits.data();
        d.suffix_group_starts = suffix_group_starts.size() ? suffix_group_starts.data() : nullptr;
        d.k = k;
        d.n_kmers = n_kmers;
        d.precalc_k = precalc_k;
        d.precalc = file_precalc;
        // a new image invalidates the replicas (they hold the old one): back to one device until use_devices() is
        // called again
        replicas_.clear();
        auto p = std::make_shared<detail::DeviceIndex>();
        detail::gpu_check(sbwtgpu_index_create(&d, detail::default_device(), &p->h));
        sbwtgpu_index_info info;
        detail::gpu_check(sbwtgpu_index_get_info(p->h, &info));
        if (info.image_level > 0)       // the library steps down when device memory (or "max_image_bytes") does not hold the full image
            write_log("Device image at level " + std::to_string(info.image_level) +
                          (info.image_level == 1 ? " (no path order / transition table" : " (blocks and dense prefix table only") +
                          "): searches run on the blocks-only kernel, three to four times slower than on the full image",
                      LogLevel::MAJOR);
        C.assign(info.C, info.C + 4);
        kmer_prefix_precalc.assign(precalc_k ? ((size_t)1 << (2 * precalc_k)) : 0, {0, 0});
        static_assert(sizeof(std::pair<int64_t, int64_t>) == 16, "pair<int64,int64> must be 16 bytes");
        if (precalc_k) detail::gpu_check(sbwtgpu_index_get_precalc(p->h, (int64_t *)kmer_prefix_precalc.data()));
        dev_ = p;
        subset_rank.attach(dev_);
    }
    const sbwtgpu_index *need_device() const {
        if (!dev_) throw std::runtime_error("Error: the index is empty");
        return dev_->h;
    }
    // SBWT.hh:422-437 on the host rank directory: validates the RAW char, ranks with the upper-cased one
    std::pair<int64_t, int64_t> host_update_interval(const char *S, int64_t S_length, std::pair<int64_t, int64_t> I) const {
        if (I.first == -1) return I;
        for (int64_t i = 0; i < S_length; i++) {
            const char c = (char)toupper((unsigned char)S[i]);
            const int char_idx = DNA_to_char_idx(S[i]);
            if (char_idx == -1) return {-1, -1};
            I.first = C[(size_t)char_idx] + subset_rank.rank(I.first, c);
            I.second = C[(size_t)char_idx] + subset_rank.rank(I.second + 1, c) - 1;
            if (I.first > I.second) return {-1, -1};
        }
        return I;
    }
    static void bug_exit(int rc) {                        // SBWT.hh:410-413
        if (rc == SBWTGPU_ERR_NOT_SINGLETON) {
            std::cerr << "Bug: k-mer search did not give a singleton interval" << std::endl;
            exit(1);
        }
    }
};

typedef SBWT plain_matrix_sbwt_t;                         // variants.hh:19

}  // namespace sbwt
