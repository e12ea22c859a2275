// sbwt_cli.cpp -- the `sbwt` command: `search` with the reference's flags, output format and log
// lines (reference src/CLI/sbwt_search.cpp, src/CLI/sbwt.cpp), running the queries on the GPU in
// batches; plus a minimal in-memory `build` so that indexes can be produced without KMC.
//
//   sbwt search -o <out> -i <index> -q <query> [-z]        (sbwt_search.cpp:151-157)
//   sbwt build  -i <in> -o <index> -k <k> [-p <precalc>] [--add-reverse-complements]
//               [--no-streaming-support] [-t <threads>]      (subset of sbwt_build.cpp:40-55)
// Extra, GPU-only flags of `search` (defaults keep the reference behaviour and output):
//   --gpu <id>            HIP device to use (default 0)
//   --gpus <n>            shard every batch over HIP devices 0..n-1 (index replicated by one RCCL broadcast)
//   --gpu-list a,b,...    the same with an explicit device list (a device may be listed twice)
//   --batch-bases <n>     bases sent to the GPU per batch (default 64 Mi); the next batch is parsed and the
//                         previous one written by their own threads meanwhile
//   --host-format         print_vector on the CPU (default: formatted on the GPU, pipelined over PCIe)
#include <algorithm>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <iostream>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <unistd.h>
#include <functional>
#include <vector>

#include "SBWT.hh"

using namespace sbwt;
using std::string;
using std::vector;

namespace {

// what cxxopts throws for bad command lines is a std::exception that is not a runtime_error, so
// main() prints it with the "Error: " prefix (sbwt.cpp:53-56)
struct option_error : public std::exception {
    string msg;
    explicit option_error(string m) : msg(std::move(m)) {}
    const char *what() const noexcept override { return msg.c_str(); }
};

struct OptSpec { string longname; char shortname; bool takes_value; string help; string def; };

class Options {
public:
    explicit Options(vector<OptSpec> specs) : specs_(std::move(specs)) {}
    void parse(int argc, char **argv) {
        for (int i = 1; i < argc; i++) {
            string a = argv[i];
            const OptSpec *s = nullptr;
            string val;
            bool has_val = false;
            if (a.rfind("--", 0) == 0) {
                string name = a.substr(2);
                size_t eq = name.find('=');
                if (eq != string::npos) { val = name.substr(eq + 1); name = name.substr(0, eq); has_val = true; }
                for (auto &x : specs_) if (x.longname == name) s = &x;
                if (!s) throw option_error("Option ‘" + name + "’ does not exist");
            } else if (a.size() == 2 && a[0] == '-') {
                for (auto &x : specs_) if (x.shortname == a[1]) s = &x;
                if (!s) throw option_error(string("Option ‘") + a[1] + "’ does not exist");
            } else {
                continue;   // positional arguments are ignored like in the reference
            }
            if (s->takes_value) {
                if (!has_val) {
                    if (i + 1 >= argc) throw option_error("Option ‘" + s->longname + "’ is missing an argument");
                    val = argv[++i];
                }
                values_[s->longname] = val;
            } else {
                values_[s->longname] = "true";
            }
        }
    }
    bool count(const string &name) const { return values_.count(name) > 0; }
    string get(const string &name) const {
        auto it = values_.find(name);
        if (it != values_.end()) return it->second;
        for (auto &x : specs_)
            if (x.longname == name && !x.def.empty()) return x.def;
        throw option_error("Option ‘" + name + "’ has no value");
    }
    string help(const string &prog, const string &desc) const {
        string h = desc + "\nUsage:\n  " + prog + " [OPTION...]\n\n";
        for (auto &x : specs_) {
            string l = "  ";
            if (x.shortname) { l += "-"; l += x.shortname; l += ", "; } else l += "    ";
            l += "--" + x.longname + (x.takes_value ? " arg" : "");
            while (l.size() < 28) l += " ";
            h += l + " " + x.help + (x.def.empty() ? "" : " (default: " + x.def + ")") + "\n";
        }
        return h;
    }

private:
    vector<OptSpec> specs_;
    std::map<string, string> values_;
};

// print_vector, sbwt_search.cpp:21-43: each value followed by one space, '\n' per read; -1 special
// cased; 0 prints as an empty token (kept: that is what the reference emits)
inline void print_vector(const int64_t *v, int64_t n, string &out) {
    char buffer[32];
    for (int64_t t = 0; t < n; t++) {
        int64_t x = v[t];
        int i = 0;
        if (x == -1) { buffer[0] = '1'; buffer[1] = '-'; i = 2; }
        else while (x > 0) { buffer[i++] = (char)('0' + (x % 10)); x /= 10; }
        while (i > 0) out.push_back(buffer[--i]);
        out.push_back(' ');
    }
    out.push_back('\n');
}

struct QueryStats { int64_t queries = 0; int64_t micros = 0; };

// A bounded hand-off between two threads (reader -> search -> writer).
template <typename T>
class Channel {
public:
    explicit Channel(size_t cap) : cap_(cap) {}
    void push(T &&v) {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return q_.size() < cap_; });
        q_.push_back(std::move(v));
        cv_.notify_all();
    }
    bool pop(T &out) {   // false when closed and drained
        std::unique_lock<std::mutex> lk(m_);
        waiting_ = true;
        cv_.notify_all();
        cv_.wait(lk, [&] { return !q_.empty() || closed_; });
        waiting_ = false;
        if (q_.empty()) return false;
        out = std::move(q_.front());
        q_.pop_front();
        cv_.notify_all();
        return true;
    }
    void close() {
        std::lock_guard<std::mutex> lk(m_);
        closed_ = true;
        cv_.notify_all();
    }
    // blocks until every item pushed so far has been popped AND its consumer has come back for the next one
    void drain() {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return q_.empty() && waiting_; });
    }

private:
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<T> q_;
    size_t cap_;
    bool closed_ = false, waiting_ = false;
};

struct ReadBatch {
    vector<char> bases;
    vector<int64_t> read_off{0}, out_off{0};
};
struct TextBatch {
    std::vector<plain_matrix_sbwt_t::TextPiece> pieces;   // GPU-formatted output
    string host_text;                                      // --host-format output
};

// run_file + run_queries_streaming / run_queries_not_streaming (sbwt_search.cpp:46-105), batched and
// pipelined: a reader thread parses the next batch and a writer thread writes the previous one while
// the GPU searches the current one.  The output is written strictly in input order.
// (get_index: the index, waited for at the first use -- while it loads, the reader thread already parses the input)
static int64_t g_t0 = 0;             // process time at the start of search_main (stage marks under SBWT_CLI_TIMING)
static void timing_mark(const char *what) {
    if (getenv("SBWT_CLI_TIMING")) std::cerr << "timing: t+" << (cur_time_micros() - g_t0) / 1e6 << " s " << what << std::endl;
}

QueryStats run_file(const string &infile, const string &outfile, const std::function<const plain_matrix_sbwt_t &()> &get_index,
                    bool gzip_output, int64_t batch_bases, bool host_format) {
    // SBWT_CLI_PARSER_THREADS=n (off by default): a plain regular file is parsed by n threads, piece by piece (seqio.hh
    // read_file_chunked: the parser is the same, the pieces are cut at record starts and handed on in file order).  Measured on
    // 10 M reads: the run is bound by the one thread at a time that writes the text (0.74-0.94 s of a 0.9-1.2 s loop), so
    // 1 / 2 / 3 parser threads give 1.40-1.46 / 1.36-1.44 / 1.50 s against 1.42-1.55 s with the one reader: nothing that would
    // pay for cutting FASTQ files by a heuristic.  (SBWT_CLI_CHUNK_MIN_BYTES: the smallest file read that way, default 32 MiB.)
    int64_t in_size = 0;
    const char *cm = getenv("SBWT_CLI_CHUNK_MIN_BYTES"), *pt = getenv("SBWT_CLI_PARSER_THREADS");
    const int parser_threads = pt ? atoi(pt) : 0;
    const bool chunked = parser_threads > 0 && seq_io::chunkable_file(infile, &in_size) &&
                         in_size >= (cm ? atoll(cm) : ((int64_t)32 << 20));
    std::unique_ptr<seq_io::Reader> reader_p;
    if (!chunked) reader_p.reset(new seq_io::Reader(infile));
    seq_io::Buffered_ofstream writer(outfile, gzip_output);
    QueryStats st;
    Channel<ReadBatch> to_search(2);
    Channel<TextBatch> to_write(2);
    std::exception_ptr reader_err, writer_err;

    const bool timing = getenv("SBWT_CLI_TIMING") != nullptr;       // stage times on stderr (tools/e2e_bench.py)
    int64_t t_parse = 0, t_write = 0;
    std::thread reader_thread([&] {
        try {
            if (chunked) {
                // a piece of the file holds about batch_bases bases: 2.1 bytes of a FASTQ file per base, 1.03 of a FASTA file
                const bool fq = seq_io::figure_out_file_format(infile).format == seq_io::FASTQ;
                const int64_t piece = (int64_t)((double)batch_bases * (fq ? 2.1 : 1.03)) + 4096;
                const int64_t p0 = cur_time_micros();
                int64_t t_push = 0;
                seq_io::read_file_chunked(infile, in_size, piece, parser_threads,
                                          [&](std::vector<char> &&b, std::vector<int64_t> &&off, bool) {
                                              if (off.size() <= 1) return;
                                              ReadBatch rb;
                                              rb.bases = std::move(b);
                                              rb.read_off = std::move(off);
                                              const int64_t q0 = cur_time_micros();
                                              to_search.push(std::move(rb));
                                              t_push += cur_time_micros() - q0;
                                          });
                t_parse += cur_time_micros() - p0 - t_push;      // (wall time of the parsers, waiting for the search excluded)
            }
            bool more = !chunked;
            while (more) {
                ReadBatch rb;
                rb.bases.reserve((size_t)batch_bases + 4096);
                const int64_t p0 = cur_time_micros();
                more = reader_p->read_batch(rb.bases, rb.read_off, batch_bases);
                t_parse += cur_time_micros() - p0;
                if (rb.read_off.size() > 1) to_search.push(std::move(rb));
            }
        } catch (...) {
            reader_err = std::current_exception();
        }
        to_search.close();
    });
    std::thread writer_thread([&] {
        try {
            TextBatch tb;
            while (to_write.pop(tb)) {
                for (const auto &pc : tb.pieces)
                    for (int64_t pos = 0; pos < pc.size; pos += (int64_t)1 << 30)
                        writer.write(pc.data + pos, std::min<int64_t>((int64_t)1 << 30, pc.size - pos));
                writer.write(tb.host_text.data(), (int64_t)tb.host_text.size());
                tb = TextBatch();
            }
        } catch (...) {
            writer_err = std::current_exception();
            TextBatch drop;
            while (to_write.pop(drop)) {}
        }
    });

    std::exception_ptr search_err;
    try {
        const plain_matrix_sbwt_t &index = get_index();
        timing_mark("index ready");
        const bool streaming = index.has_streaming_query_support();
        write_log(string("Running ") + (streaming ? "streaming" : "non-streaming") + " queries from input file " + infile +
                      " to output file " + outfile,
                  LogLevel::MAJOR);
        const int64_t k = index.get_k();
        if (!host_format && index.number_of_devices_in_use() <= 1 && !gzip_output) {
            // The default: search + print_vector on the GPU, pipelined over PCIe; the text goes from the pinned
            // staging buffers straight into the output file (one copy, into the page cache).  TWO threads take the batches
            // in turn: while one copies its batch's text into the file -- the longest stage, one thread's write(2) at the
            // page cache's rate -- the other's batch is already on the GPU and its first pieces of text are waiting; its
            // sink blocks until the batch before it is written.  The output is in input order, piece by piece.
            std::mutex pop_m, turn_m;
            std::condition_variable turn_cv;
            int64_t next_seq = 0, turn = 0;
            bool failed = false;
            std::exception_ptr worker_err[2];
            auto worker = [&](int w) {
                try {
                    for (;;) {
                        ReadBatch rb;
                        int64_t seq;
                        {
                            std::lock_guard<std::mutex> lk(pop_m);
                            if (!to_search.pop(rb)) break;
                            seq = next_seq++;
                        }
                        const int64_t n_reads = (int64_t)rb.read_off.size() - 1;
                        bool mine = false;
                        int64_t t_sink = 0, t_wait = 0, nq = 0;
                        auto take_turn = [&] {
                            if (mine) return;
                            const int64_t w0 = cur_time_micros();
                            std::unique_lock<std::mutex> lk(turn_m);
                            turn_cv.wait(lk, [&] { return turn == seq || failed; });
                            if (failed && turn != seq) throw std::runtime_error("search stopped: another batch failed");
                            mine = true;
                            t_wait += cur_time_micros() - w0;
                        };
                        const int64_t t0 = cur_time_micros();
                        try {
                            nq = index.search_text_stream(rb.bases.data(), rb.read_off.data(), n_reads,
                                                          [&](const char *text, int64_t bytes) {
                                                              take_turn();
                                                              const int64_t w0 = cur_time_micros();
                                                              writer.write(text, bytes);
                                                              t_sink += cur_time_micros() - w0;
                                                          });
                            take_turn();                       // (a batch without any text still waits for its turn)
                        } catch (...) {
                            std::lock_guard<std::mutex> lk(turn_m);
                            failed = true;
                            turn_cv.notify_all();
                            throw;
                        }
                        {
                            std::lock_guard<std::mutex> lk(turn_m);
                            st.queries += nq;
                            st.micros += cur_time_micros() - t0 - t_sink - t_wait;
                            t_write += t_sink;
                            turn = seq + 1;
                        }
                        turn_cv.notify_all();
                        timing_mark("a batch searched and written");
                    }
                } catch (...) {
                    worker_err[w] = std::current_exception();
                }
            };
            // SBWT_CLI_SEARCH_THREADS=1: one worker (three staging slots instead of six: pinned and device memory are tight)
            const char *st_env = getenv("SBWT_CLI_SEARCH_THREADS");
            const bool one_worker = st_env && atoi(st_env) == 1;
            std::thread second;
            if (!one_worker) second = std::thread([&] { worker(1); });
            worker(0);
            if (second.joinable()) second.join();
            for (auto &e : worker_err)
                if (e) std::rethrow_exception(e);
        }
        ReadBatch rb;
        vector<int64_t> out;
        while (to_search.pop(rb)) {
            const int64_t n_reads = (int64_t)rb.read_off.size() - 1;
            rb.out_off.resize(rb.read_off.size());
            for (size_t r = 1; r < rb.read_off.size(); r++)
                rb.out_off[r] = rb.out_off[r - 1] + std::max<int64_t>(0, rb.read_off[r] - rb.read_off[r - 1] - k + 1);
            TextBatch tb;
            if (host_format) {
                // reference-style: raw ranks back to the host, print_vector on the CPU
                out.resize((size_t)rb.out_off.back());
                int64_t t0 = cur_time_micros();
                if (streaming) index.streaming_search_batch(rb.bases.data(), rb.read_off.data(), n_reads, out.data(), rb.out_off.data());
                else index.search_batch(rb.bases.data(), rb.read_off.data(), n_reads, out.data(), rb.out_off.data());
                st.micros += cur_time_micros() - t0;
                st.queries += rb.out_off.back();
                for (int64_t r = 0; r < n_reads; r++)
                    print_vector(out.data() + rb.out_off[(size_t)r], rb.out_off[(size_t)r + 1] - rb.out_off[(size_t)r], tb.host_text);
            } else {
                // several devices (every device's text is a piece of its own), or compressed output (a writer thread)
                int64_t t0 = cur_time_micros();
                st.queries += index.search_text_batch(rb.bases.data(), rb.read_off.data(), n_reads, tb.pieces);
                st.micros += cur_time_micros() - t0;
            }
            to_write.push(std::move(tb));
        }
    } catch (...) {
        search_err = std::current_exception();
        ReadBatch drop;
        while (to_search.pop(drop)) {}
    }
    timing_mark("search loop left");
    to_write.close();
    reader_thread.join();
    timing_mark("reader joined");
    writer_thread.join();
    timing_mark("writer joined");
    if (search_err) std::rethrow_exception(search_err);
    if (reader_err) std::rethrow_exception(reader_err);
    if (writer_err) std::rethrow_exception(writer_err);
    write_log("us/query: " + std::to_string((double)st.micros / (double)st.queries) + " (excluding I/O etc)",
              LogLevel::MAJOR);
    writer.close();
    timing_mark("output closed");
    if (timing)
        std::cerr << "timing: parse " << t_parse / 1e6 << " s, search (GPU + PCIe) " << st.micros / 1e6 << " s, write "
                  << t_write / 1e6 << " s" << std::endl;
    return st;
}

int search_main(int argc, char **argv) {
    int64_t micros_start = cur_time_micros();
    g_t0 = micros_start;
    set_log_level(LogLevel::MINOR);
    Options opts({
        {"out-file", 'o', true, "Output filename.", ""},
        {"index-file", 'i', true, "Index input file.", ""},
        {"query-file", 'q', true,
         "The query in FASTA or FASTQ format, possibly gzipped. Multi-line FASTQ is not supported. If the file "
         "extension is .txt, this is interpreted as a list of query files, one per line. In this case, --out-file is "
         "also interpreted as a list of output files in the same manner, one line for each input file.", ""},
        {"gzip-output", 'z', false,
         "Writes output in gzipped form. This can shrink the output files by an order of magnitude.", ""},
        {"gpu", 0, true, "HIP device to run on.", "0"},
        {"gpus", 0, true, "Shard every batch over HIP devices 0..n-1.", "1"},
        {"gpu-list", 0, true, "Explicit comma separated device list to shard over.", "-"},
        {"batch-bases", 0, true, "Bases sent to the GPU per batch.", "67108864"},
        {"host-format", 0, false, "Format the output on the CPU (print_vector) instead of on the GPU.", ""},
        {"help", 'h', false, "Print usage", ""},
    });
    opts.parse(argc, argv);
    if (argc == 1 || opts.count("help")) {
        std::cerr << opts.help(argv[0], "Query all k-mers of all input reads.") << std::endl;
        exit(1);
    }
    string indexfile = opts.get("index-file");
    check_readable(indexfile);

    string queryfile = opts.get("query-file");
    vector<string> input_files;
    bool multi_file = queryfile.size() >= 4 && queryfile.substr(queryfile.size() - 4) == ".txt";
    if (multi_file) input_files = readlines(queryfile);
    else input_files = {queryfile};
    for (const string &file : input_files) check_readable(file);

    string outfile = opts.get("out-file");
    bool gzip_output = opts.count("gzip-output");
    vector<string> output_files;
    if (multi_file) output_files = readlines(outfile);
    else output_files = {outfile};
    for (const string &file : output_files) check_writable(file);

    vector<int> devices;
    if (opts.get("gpu-list") != "-") {
        string l = opts.get("gpu-list");
        for (size_t pos = 0; pos <= l.size();) {
            size_t e = l.find(',', pos);
            if (e == string::npos) e = l.size();
            devices.push_back(atoi(l.substr(pos, e - pos).c_str()));
            pos = e + 1;
        }
    } else {
        int n = atoi(opts.get("gpus").c_str());
        if (n > 1) for (int g = 0; g < n; g++) devices.push_back(g);
        else devices.push_back(atoi(opts.get("gpu").c_str()));
    }
    set_default_device(devices[0]);
    int64_t batch_bases = atoll(opts.get("batch-bases").c_str());
    if (batch_bases < 1) batch_bases = 1;

    std::ifstream in(indexfile, std::ios::binary);
    if (!in.good()) throw std::runtime_error("Error opening file: " + indexfile);
    string variant = load_string(in);
    static const char *variants[] = {"plain-matrix", "rrr-matrix", "mef-matrix", "plain-split", "rrr-split",
                                     "mef-split", "plain-concat", "mef-concat", "plain-subsetwt", "rrr-subsetwt"};
    if (std::find(std::begin(variants), std::end(variants), variant) == std::end(variants)) {
        std::cerr << "Error loading index from file: unrecognized variant specified in the file" << std::endl;
        return 1;
    }
    write_log("Loading the index variant " + variant, LogLevel::MAJOR);
    if (variant != "plain-matrix")
        throw std::runtime_error("Error: only the plain-matrix variant is supported by the GPU search path (got " +
                                 variant + ")");
    if (input_files.size() != output_files.size())   // run_queries, sbwt_search.cpp:111-115
        throw std::runtime_error("Number of input and output files does not match (" +
                                 std::to_string(input_files.size()) + " vs " + std::to_string(output_files.size()) + ")");
    // The index loads (file, HIP start-up, device image: a few tenths of a second) on a thread of its own while the first
    // input file is already being parsed; the search waits for it.
    plain_matrix_sbwt_t index;
    std::exception_ptr load_err;
    std::thread loader([&] {
        try {
            const int64_t load0 = cur_time_micros();
            timing_mark("loader thread started");
            index.load(in);
            if (getenv("SBWT_CLI_TIMING"))
                std::cerr << "timing: index load + device image " << (cur_time_micros() - load0) / 1e6 << " s" << std::endl;
            if (devices.size() > 1) {
                index.use_devices(devices);
                write_log("Index replicated onto " + std::to_string(devices.size()) + " GPU contexts", LogLevel::MAJOR);
            }
        } catch (...) {
            load_err = std::current_exception();
        }
    });
    bool joined = false;
    std::mutex join_mutex;
    const std::function<const plain_matrix_sbwt_t &()> get_index = [&]() -> const plain_matrix_sbwt_t & {
        std::lock_guard<std::mutex> lk(join_mutex);
        if (!joined) { loader.join(); joined = true; }
        if (load_err) std::rethrow_exception(load_err);
        return index;
    };
    int64_t number_of_queries = 0;
    try {
        for (size_t i = 0; i < input_files.size(); i++)
            number_of_queries += run_file(input_files[i], output_files[i], get_index, gzip_output, batch_bases,
                                          opts.count("host-format")).queries;
    } catch (...) {
        if (!joined) { loader.join(); joined = true; }
        throw;
    }
    if (!joined) { loader.join(); joined = true; }

    timing_mark("all files done");
    int64_t total_micros = cur_time_micros() - micros_start;
    write_log("us/query end-to-end: " + std::to_string((double)total_micros / (double)number_of_queries), LogLevel::MAJOR);
    return 0;
}

int build_main(int argc, char **argv) {
    set_log_level(LogLevel::MAJOR);
    Options opts({
        {"in-file", 'i', true,
         "The input sequences as a FASTA or FASTQ file, possibly gzipped. If the file extension is .txt, the file is "
         "interpreted as a list of input files, one file on each line.", ""},
        {"out-file", 'o', true, "Output file for the constructed index.", ""},
        {"kmer-length", 'k', true, "The k-mer length.", ""},
        {"precalc-length", 'p', true, "Precalculate SBWT intervals of strings of this length.", "8"},
        {"variant", 0, true, "The SBWT variant to build (only plain-matrix here).", "plain-matrix"},
        {"add-reverse-complements", 0, false, "Also add the reverse complement of every k-mer to the index.", ""},
        {"no-streaming-support", 0, false, "Do not build the streaming query support bit vector.", ""},
        {"n-threads", 't', true, "Number of parallel threads.", "1"},
        {"temp-dir", 'd', true, "Ignored (construction is in memory).", "."},
        {"gpu", 0, true, "HIP device used for the prefix table.", "0"},
        {"help", 'h', false, "Print usage", ""},
    });
    opts.parse(argc, argv);
    if (argc == 1 || opts.count("help")) {
        std::cerr << opts.help(argv[0], "Construct a plain-matrix SBWT (in memory).") << std::endl;
        exit(1);
    }
    string variant = opts.get("variant");
    if (variant != "plain-matrix") {
        std::cerr << "Error: unknown variant: " << variant << std::endl;
        return 1;
    }
    string out_file = opts.get("out-file");
    check_writable(out_file);
    string in_file = opts.get("in-file");
    vector<string> input_files;
    if (in_file.size() >= 4 && in_file.substr(in_file.size() - 4) == ".txt") input_files = readlines(in_file);
    else input_files = {in_file};
    for (const string &f : input_files) check_readable(f);
    int64_t k = atoll(opts.get("kmer-length").c_str());
    int64_t precalc = atoll(opts.get("precalc-length").c_str());
    if (precalc > k) {   // sbwt_build.cpp:101-105
        write_log("Warning: precalc length " + std::to_string(precalc) + " is longer than k = " + std::to_string(k), LogLevel::MAJOR);
        write_log("Setting precalc length to " + std::to_string(k), LogLevel::MAJOR);
        precalc = k;
    }
    set_default_device(atoi(opts.get("gpu").c_str()));
    vector<string> seqs;
    for (const string &f : input_files) {
        seq_io::Reader reader(f);
        for (;;) {
            int64_t len = reader.get_next_read_to_buffer();
            if (len == 0) break;
            seqs.emplace_back(reader.read_buf, (size_t)len);
        }
    }
    write_log("Building SBWT subset sequence in memory", LogLevel::MAJOR);
    PlainMatrixBits bits = build_plain_matrix_bits_any(seqs, (int)k, opts.count("add-reverse-complements"),
                                                   !opts.count("no-streaming-support"), atoi(opts.get("n-threads").c_str()));
    plain_matrix_sbwt_t index(bits, 0);
    write_log("Build SBWT for " + std::to_string(index.number_of_kmers()) + " distinct k-mers", LogLevel::MAJOR);
    write_log("SBWT has " + std::to_string(index.number_of_subsets()) + " subsets", LogLevel::MAJOR);
    std::ofstream out(out_file, std::ios::binary);
    if (!out.good()) throw std::runtime_error("Error opening file: " + out_file);
    serialize_string(variant, out);                       // sbwt_build.cpp:142
    index.do_kmer_prefix_precalc(precalc);                // sbwt_build.cpp:157
    int64_t bytes_written = index.serialize(out);
    write_log("Built variant " + variant + " to file " + out_file, LogLevel::MAJOR);
    write_log("Space on disk: " + std::to_string(bytes_written * 8.0 / (double)index.number_of_subsets()) +
                  " bits per column, " + std::to_string(bytes_written * 8.0 / (double)index.number_of_kmers()) +
                  " bits per k-mer",
              LogLevel::MAJOR);
    return 0;
}

const vector<string> commands = {"build", "search"};

void print_help(char **argv) {
    std::cerr << "Available commands: " << std::endl;
    for (const string &S : commands) std::cerr << "   " << argv[0] << " " << S << std::endl;
    std::cerr << "Running a command without arguments prints the usage instructions for the command." << std::endl;
}

}  // namespace

int main(int argc, char **argv) {   // sbwt.cpp:19-57
    // the reference logs its compile-time MAX_KMER_LENGTH here (sbwt.cpp:25); this build handles any k <= 255
    write_log("Maximum k-mer length is set to 255", LogLevel::MAJOR);
    if (argc == 1) { print_help(argv); return 0; }
    string command = argv[1];
    if (command == "--help" || command == "-h") { print_help(argv); return 0; }
    for (int i = 1; i < argc; i++) argv[i - 1] = argv[i];
    argc--;
    try {
        if (command == "build") return build_main(argc, argv);
        else if (command == "search") {
            const int rc = search_main(argc, argv);
            // every output file is closed: leave without the HIP runtime's teardown and the freeing of the index (a tenth
            // of a second of a sub-second run)
            std::cout.flush();
            std::cerr.flush();
            fflush(nullptr);
            _exit(rc);
        }
        else throw std::runtime_error("Invalid command: " + command);
    } catch (const std::runtime_error &e) {
        std::cerr << "Runtime error: " << e.what() << '\n';
        return 1;
    } catch (const std::exception &e) {
        std::cerr << "Error: " << e.what() << '\n';
        return 1;
    }
}
