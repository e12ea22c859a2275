// seqio.hh -- FASTA/FASTQ(.gz) reader and buffered (gz) writer with the observable behaviour the
// reference gets from its SeqIO submodule (absent from the checkout): file format by
// extension, optional .gz, multi-line FASTA, 4-line FASTQ, sequences upper-cased on read
// [UPSTREAM-KNOWLEDGE], get_next_read_to_buffer() returning 0 at end of file.  I/O plumbing around
// the GPU path, not part of it.
#pragma once
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <cerrno>
#include <cctype>
#include <cstdint>
#include <cstring>
#include <algorithm>
#include <cstdio>
#include <condition_variable>
#include <deque>
#include <exception>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace sbwt {
namespace seq_io {

enum Format { FASTA, FASTQ };
struct FileFormat {
    Format format;
    bool gzipped;
    std::string extension;   // with the leading dot, e.g. ".fna.gz"
};

inline FileFormat figure_out_file_format(std::string filename) {
    bool gz = false;
    std::string ext_gz;
    if (filename.size() >= 3 && filename.substr(filename.size() - 3) == ".gz") {
        filename = filename.substr(0, filename.size() - 3);
        gz = true;
        ext_gz = ".gz";
    }
    static const char *fasta[] = {".fasta", ".fna", ".ffn", ".faa", ".frn", ".fa"};
    static const char *fastq[] = {".fastq", ".fq"};
    for (int i = (int)filename.size() - 1; i >= 0; i--) {
        if (filename[(size_t)i] == '.') {
            std::string end = filename.substr((size_t)i);
            for (const char *e : fasta)
                if (end == e) return FileFormat{FASTA, gz, end + ext_gz};
            for (const char *e : fastq)
                if (end == e) return FileFormat{FASTQ, gz, end + ext_gz};
            throw std::runtime_error("Unknown file format: " + filename);
        }
    }
    throw std::runtime_error("Unknown file format: " + filename);
}

// gzread() passes plain files through unchanged, so one reader serves both.
class Reader {
public:
    char *read_buf = nullptr;       // the current read, NUL-terminated (like SeqIO's read_buf)

    explicit Reader(const std::string &filename) : filename_(filename) {
        const FileFormat ff = figure_out_file_format(filename);
        fmt_ = ff.format;
        if (!ff.gzipped) {
            // (a file that is gzip data under a plain name is still read through zlib, like gzread would.)  The probe
            // reads the first two bytes with read(2) and KEEPS them as the start of the parse buffer -- no rewind: the
            // input may be a pipe or a FIFO (`-q <(zcat reads.fq.gz)`, /dev/stdin), where seeking back is impossible.
            // A regular file that turns out to be gzip data is reopened through zlib; gzip data arriving on a pipe
            // under a plain name is rejected loudly (the two bytes are gone) instead of being parsed as text.
            const int fd = ::open(filename.c_str(), O_RDONLY);
            if (fd < 0) throw std::runtime_error("Error opening file: " + filename);
            unsigned char magic[2] = {0, 0};
            size_t got = 0;
            while (got < 2) {
                const ssize_t r = ::read(fd, magic + got, 2 - got);
                if (r < 0 && errno == EINTR) continue;
                if (r < 0) { ::close(fd); throw std::runtime_error("Error reading file: " + filename); }
                if (r == 0) break;
                got += (size_t)r;
            }
            if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
                struct stat sb;
                const bool regular = ::fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode);
                ::close(fd);
                if (!regular)
                    throw std::runtime_error("Error: gzip data on a non-seekable input needs a .gz file name: " + filename);
            } else {
                plain_fd_ = fd;
                buf_.resize((size_t)8 << 20);
                memcpy(buf_.data(), magic, got);     // the probed bytes are the first bytes of the parse buffer
                pos_ = 0;
                end_ = got;
            }
        }
        if (plain_fd_ < 0) {
            f_ = gzopen(filename.c_str(), "rb");
            if (!f_) throw std::runtime_error("Error opening file: " + filename);
            gzbuffer(f_, 1 << 20);
            buf_.resize((size_t)1 << 20);
        }
        seq_.reserve(1 << 10);
    }
    // A byte range [begin, end) of a plain (not gzipped) regular file that starts at a record's first byte and ends at one:
    // the same parser on a piece of the file (read_file_chunked below); bytes are fetched with pread.
    Reader(const std::string &filename, int64_t begin, int64_t end) : filename_(filename) {
        fmt_ = figure_out_file_format(filename).format;
        plain_fd_ = ::open(filename.c_str(), O_RDONLY);
        if (plain_fd_ < 0) throw std::runtime_error("Error opening file: " + filename);
        ranged_ = true;
        range_pos_ = begin;
        range_end_ = end;
        buf_.resize((size_t)8 << 20);
        seq_.reserve(1 << 10);
    }
    ~Reader() {
        if (f_) gzclose(f_);
        if (plain_fd_ >= 0) ::close(plain_fd_);
    }
    // read_batch() / get_next_read_to_buffer() ended at a record without bases (which ends the stream like EOF does), not at
    // the end of the input
    bool stopped_at_empty_record() const { return empty_record_; }
    Reader(const Reader &) = delete;
    Reader &operator=(const Reader &) = delete;

    // Length of the next read (0 = end of file); the read is left in read_buf.
    int64_t get_next_read_to_buffer() {
        seq_.clear();
        if (fmt_ == FASTA) {
            int c = getc_();
            if (c == -1) return finish(0);
            if (c != '>') throw std::runtime_error("Error: FASTA header does not start with '>' in " + filename_);
            skip_line();
            for (;;) {
                c = peekc_();
                if (c == -1 || c == '>') break;
                read_line_append();
            }
        } else {
            int c = getc_();
            if (c == -1) return finish(0);
            if (c != '@') throw std::runtime_error("Error: FASTQ header does not start with '@' in " + filename_);
            skip_line();
            read_line_append();
            c = getc_();
            if (c != '+') throw std::runtime_error("Error: FASTQ separator line missing in " + filename_);
            skip_line();
            skip_line();            // quality line
        }
        for (char &ch : seq_) ch = (char)std::toupper((unsigned char)ch);
        return finish((int64_t)seq_.size());
    }

    // Appends reads to (bases, read_off) until `bases` holds max_bases or the file ends; returns false at end of file.
    // Same reads as get_next_read_to_buffer() one by one, but whole lines are copied once and upper-cased in bulk
    // (the CLI's reader thread: 600 MB of FASTQ per second were the slowest stage of `sbwt search`).
    bool read_batch(std::vector<char> &bases, std::vector<int64_t> &read_off, int64_t max_bases) {
        while ((int64_t)bases.size() < max_bases) {
            const size_t before = bases.size();
            if (fmt_ == FASTQ) {
                int c = getc_();
                if (c == -1) return false;
                if (c != '@') throw std::runtime_error("Error: FASTQ header does not start with '@' in " + filename_);
                skip_line();
                read_line_append_to(bases);
                c = getc_();
                if (c != '+') throw std::runtime_error("Error: FASTQ separator line missing in " + filename_);
                skip_line();
                skip_line();            // quality line
            } else {
                int c = getc_();
                if (c == -1) return false;
                if (c != '>') throw std::runtime_error("Error: FASTA header does not start with '>' in " + filename_);
                skip_line();
                for (;;) {
                    c = peekc_();
                    if (c == -1 || c == '>') break;
                    read_line_append_to(bases);
                }
            }
            if (bases.size() == before) {      // a record without bases ends the stream like a zero-length read does
                empty_record_ = true;           // (get_next_read_to_buffer() returns 0 for it, which callers take as EOF)
                return false;
            }
            char *p = bases.data() + before;
            const size_t n = bases.size() - before;
            for (size_t i = 0; i < n; i++) {    // toupper of ASCII letters (vectorises)
                const unsigned char ch = (unsigned char)p[i];
                p[i] = (char)(ch - (((unsigned)(ch - 'a') < 26u) ? 32 : 0));
            }
            read_off.push_back((int64_t)bases.size());
        }
        return true;
    }

    std::string get_next_read() {   // "" at end of file (tests/test_large.hh:39-45 usage)
        int64_t len = get_next_read_to_buffer();
        return std::string(read_buf, (size_t)len);
    }

private:
    int64_t finish(int64_t len) {
        seq_.push_back('\0');
        read_buf = seq_.data();
        seq_.pop_back();
        return len;
    }
    bool fill() {
        int n;
        if (ranged_) {               // a piece of a plain regular file
            const int64_t want = std::min<int64_t>((int64_t)buf_.size(), range_end_ - range_pos_);
            if (want <= 0) { pos_ = end_ = 0; return false; }
            do {
                n = (int)::pread(plain_fd_, buf_.data(), (size_t)want, (off_t)range_pos_);
            } while (n < 0 && errno == EINTR);
            if (n > 0) range_pos_ += n;
        } else if (plain_fd_ >= 0) {        // not compressed: straight read(2) into the parse buffer (gzread copies twice)
            do {
                n = (int)::read(plain_fd_, buf_.data(), std::min(buf_.size(), (size_t)1 << 30));
            } while (n < 0 && errno == EINTR);
        } else {
            n = gzread(f_, buf_.data(), (unsigned)buf_.size());
        }
        if (n < 0) throw std::runtime_error("Error reading file: " + filename_);
        pos_ = 0;
        end_ = (size_t)n;
        return n > 0;
    }
    int getc_() {
        if (pos_ == end_ && !fill()) return -1;
        return (unsigned char)buf_[pos_++];
    }
    int peekc_() {
        if (pos_ == end_ && !fill()) return -1;
        return (unsigned char)buf_[pos_];
    }
    void skip_line() {
        for (;;) {
            if (pos_ == end_ && !fill()) return;
            char *nl = (char *)memchr(buf_.data() + pos_, '\n', end_ - pos_);
            if (nl) { pos_ = (size_t)(nl - buf_.data()) + 1; return; }
            pos_ = end_;
        }
    }
    void read_line_append_to(std::vector<char> &dst) {
        for (;;) {
            if (pos_ == end_ && !fill()) return;
            char *start = buf_.data() + pos_;
            char *nl = (char *)memchr(start, '\n', end_ - pos_);
            size_t n = nl ? (size_t)(nl - start) : end_ - pos_;
            size_t keep = n;
            if (keep && start[keep - 1] == '\r') keep--;
            dst.insert(dst.end(), start, start + keep);
            pos_ += n + (nl ? 1 : 0);
            if (nl) return;
        }
    }
    void read_line_append() {
        for (;;) {
            if (pos_ == end_ && !fill()) return;
            char *start = buf_.data() + pos_;
            char *nl = (char *)memchr(start, '\n', end_ - pos_);
            size_t n = nl ? (size_t)(nl - start) : end_ - pos_;
            size_t keep = n;
            if (keep && start[keep - 1] == '\r') keep--;
            seq_.insert(seq_.end(), start, start + keep);
            pos_ += n + (nl ? 1 : 0);
            if (nl) return;
        }
    }
    std::string filename_;
    Format fmt_;
    gzFile f_ = nullptr;
    int plain_fd_ = -1;             // uncompressed input: read(2) directly (regular file, pipe or FIFO)
    std::vector<char> buf_;
    size_t pos_ = 0, end_ = 0;
    std::vector<char> seq_;
    bool ranged_ = false;           // a byte range of a regular file (pread)
    int64_t range_pos_ = 0, range_end_ = 0;
    bool empty_record_ = false;
};

// ---- a plain regular file parsed by several threads, piece by piece --------------------------------------------------
// The file is cut at record starts about chunk_bytes apart; every piece is parsed by Reader's own code (range mode), the
// pieces' reads are handed to `emit(bases, read_off, last)` in file order, one piece at a time.  What ends the stream for
// the sequential reader ends it here: a record without bases in piece i means pieces i+1 .. are dropped.
//
// Where a record starts, seen from the middle of a file: FASTA -- a line that begins with '>' (sequence lines cannot hold
// one).  FASTQ (four lines per record, as the sequential reader requires) -- a line that begins with '@' whose next line but
// one begins with '+': a quality line may begin with '@' too, but the line two below a quality line is a sequence line, and
// those begin with a base.
inline bool chunkable_file(const std::string &filename, int64_t *size) {
    FileFormat ff;
    try { ff = figure_out_file_format(filename); } catch (...) { return false; }
    if (ff.gzipped) return false;
    struct stat sb;
    if (::stat(filename.c_str(), &sb) != 0 || !S_ISREG(sb.st_mode)) return false;
    const int fd = ::open(filename.c_str(), O_RDONLY);
    if (fd < 0) return false;
    unsigned char magic[2] = {0, 0};
    const ssize_t got = ::pread(fd, magic, 2, 0);
    ::close(fd);
    if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) return false;       // gzip data under a plain name
    *size = (int64_t)sb.st_size;
    return true;
}
// first record start at or after `from` (file_size if there is none; -1 if it cannot be told within 64 MB)
inline int64_t find_record_start(int fd, Format fmt, int64_t from, int64_t file_size) {
    if (from <= 0) return 0;
    if (from >= file_size) return file_size;
    std::vector<char> w;
    for (size_t span = (size_t)1 << 20; span <= ((size_t)64 << 20); span <<= 2) {
        const int64_t lo = from - 1;                                        // (the byte before: is `from` itself a line start?)
        const int64_t len = std::min<int64_t>((int64_t)span, file_size - lo);
        w.resize((size_t)len);
        int64_t got = 0;
        while (got < len) {
            const ssize_t r = ::pread(fd, w.data() + got, (size_t)(len - got), (off_t)(lo + got));
            if (r < 0 && errno == EINTR) continue;
            if (r <= 0) break;
            got += r;
        }
        if (got < len) return -1;
        const bool whole = lo + len == file_size;                            // the window reaches the end of the file
        // line starts inside the window: index i such that w[i - 1] == '\n'
        size_t i = 0;
        const char *nl0 = (const char *)memchr(w.data(), '\n', (size_t)len);
        if (!nl0) { if (whole) return file_size; continue; }
        i = (size_t)(nl0 - w.data()) + 1;
        while (i < (size_t)len) {
            if (fmt == FASTA) {
                if (w[i] == '>') return lo + (int64_t)i;
            } else if (w[i] == '@') {
                const char *a = (const char *)memchr(w.data() + i, '\n', (size_t)len - i);
                const char *b = a ? (const char *)memchr(a + 1, '\n', (size_t)(w.data() + len - (a + 1))) : nullptr;
                if (b && b + 1 < w.data() + len) {
                    if (b[1] == '+') return lo + (int64_t)i;
                } else if (!whole) {
                    break;                                                   // the next lines are beyond the window: a wider one
                }
            }
            const char *nl = (const char *)memchr(w.data() + i, '\n', (size_t)len - i);
            if (!nl) { i = (size_t)len; break; }
            i = (size_t)(nl - w.data()) + 1;
        }
        if (i >= (size_t)len && whole) return file_size;
        if (whole) return file_size;
    }
    return -1;
}
template <typename Emit>
inline void read_file_chunked(const std::string &filename, int64_t file_size, int64_t chunk_bytes, int n_threads, Emit &&emit) {
    const Format fmt = figure_out_file_format(filename).format;
    const int fd = ::open(filename.c_str(), O_RDONLY);
    if (fd < 0) throw std::runtime_error("Error opening file: " + filename);
    if (chunk_bytes < 4096) chunk_bytes = 4096;
    if (n_threads < 1) n_threads = 1;
    struct Piece { std::vector<char> bases; std::vector<int64_t> read_off{0}; bool ready = false, stop = false; std::exception_ptr err; };
    // pieces are planned one after the other (a cut needs the cut before it), parsed by whoever is free, emitted in order
    std::mutex m;
    std::condition_variable cv;
    int64_t next_begin = 0, planned = 0, emitted = 0;
    bool plan_done = false, stopped = false, emitting = false, planning = false;
    std::deque<Piece> window;                       // pieces emitted .. planned - 1
    auto work = [&] {
        for (;;) {
            int64_t idx, b, e;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return stopped || plan_done || (!planning && planned - emitted < (int64_t)n_threads + 1); });
                if (stopped || plan_done) return;
                b = next_begin;
                // the cut is looked for WITHOUT the lock (find_record_start reads and scans 1 .. 64 MB windows: the other
                // workers go on storing and emitting their pieces meanwhile); `planning` keeps the plans one after the other
                planning = true;
                lk.unlock();
                e = b + chunk_bytes >= file_size ? file_size : find_record_start(fd, fmt, b + chunk_bytes, file_size);
                if (e < 0 || e <= b) e = file_size;       // no cut to be found: the rest is one piece
                lk.lock();
                planning = false;
                next_begin = e;
                if (e >= file_size) plan_done = true;
                idx = planned++;
                window.emplace_back();
                cv.notify_all();
            }
            Piece pc;
            try {
                Reader r(filename, b, e);
                pc.bases.reserve((size_t)((e - b) / (fmt == FASTQ ? 2 : 1)) + 4096);     // (no regrowing: bases are < half of a FASTQ piece)
                r.read_batch(pc.bases, pc.read_off, INT64_MAX);
                pc.stop = r.stopped_at_empty_record();
            } catch (...) {
                pc.err = std::current_exception();
            }
            pc.ready = true;
            std::unique_lock<std::mutex> lk(m);
            window[(size_t)(idx - emitted)] = std::move(pc);
            // whoever completes the piece that is next in line emits it and what is ready behind it
            while (!emitting && !stopped && !window.empty() && window.front().ready) {
                emitting = true;                    // (one emitter at a time: the front stays in place until it is handed over)
                Piece out = std::move(window.front());
                const bool last = out.stop || out.err || (plan_done && emitted + 1 == planned);
                if (out.stop || out.err) stopped = true;
                lk.unlock();
                // (emitting outside the lock: the other threads go on parsing; order is kept because only the front is taken,
                // and the front is not popped until it has been handed over)
                if (out.err) std::rethrow_exception(out.err);           // (the thread's handler sets `stopped`)
                emit(std::move(out.bases), std::move(out.read_off), last);
                lk.lock();
                window.pop_front();
                emitted++;
                emitting = false;
                cv.notify_all();
            }
            cv.notify_all();
        }
    };
    std::vector<std::thread> th;
    std::vector<std::exception_ptr> errs((size_t)n_threads);
    for (int t = 0; t < n_threads; t++)
        th.emplace_back([&, t] {
            try { work(); } catch (...) {
                errs[(size_t)t] = std::current_exception();
                std::lock_guard<std::mutex> lk(m);
                stopped = true;
                cv.notify_all();
            }
        });
    for (auto &t : th) t.join();
    ::close(fd);
    for (auto &e : errs)
        if (e) std::rethrow_exception(e);
}

// Buffered output, optionally gzip-compressed (-z of `sbwt search`, sbwt_search.cpp:120,126-137).
// Compression runs on several threads: the data is cut into 1 MiB blocks, every block becomes a gzip
// member of its own, and the members are written in order -- a multi-member gzip file, which zcat,
// zlib's gzread and Python's gzip module all read as one stream.
class Buffered_ofstream {
public:
    Buffered_ofstream(const std::string &filename, bool gzip, int n_threads = 0) : filename_(filename), gzip_(gzip) {
        // No O_TRUNC: truncating a file to nothing marks it for ext4's "replace via truncate" heuristic (auto_da_alloc), which
        // allocates and starts writing back ALL of its dirty pages inside close() -- 0.77 s for the 8 GB a 10 M-read search
        // writes, a third of the whole run, and it strikes every run because check_writable() has just created the file
        // (truncating an EMPTY file counts).  An existing regular file is overwritten in place from offset 0 and cut to
        // what was written when it is closed (ftruncate to a length > 0 does not set the mark); the result is the same file
        // the reference's ofstream would leave, same inode, same contents.
        fd_ = ::open(filename.c_str(), O_WRONLY | O_CREAT, 0666);
        if (fd_ < 0) throw std::runtime_error("Error opening file: " + filename);
        struct stat sb;
        if (::fstat(fd_, &sb) == 0 && S_ISREG(sb.st_mode)) {
            regular_ = true;
            old_size_ = (int64_t)sb.st_size;
            // An existing file with contents is cut to ONE byte at once (a length > 0 does not set the ext4 mark): a process
            // that dies before close() -- SIGKILL, the OOM killer, exit() from an error path -- then leaves a visibly short
            // file like the reference's ofstream would, not the new text followed by the old file's tail.
            if (old_size_ > 1) {
                if (::ftruncate(fd_, 1) != 0) { ::close(fd_); fd_ = -1; throw std::runtime_error("Error opening file: " + filename); }
                old_size_ = 1;
            }
        }
        if (n_threads <= 0) {
            n_threads = (int)std::thread::hardware_concurrency();
            if (n_threads > 16) n_threads = 16;
            if (n_threads < 1) n_threads = 1;
        }
        n_threads_ = n_threads;
    }
    ~Buffered_ofstream() {
        try { close(); } catch (...) {}
    }
    Buffered_ofstream(const Buffered_ofstream &) = delete;
    Buffered_ofstream &operator=(const Buffered_ofstream &) = delete;
    void write(const char *data, int64_t n) {
        if (n <= 0) return;
        if (!gzip_) {
            // small pieces are gathered; large ones (the search output arrives in pieces of tens of MB) go straight to the
            // file.  (Several threads writing different ranges do not help: buffered writes to one file take its inode
            // lock -- measured, 1.65 GB in 0.17 s either way.)
            if (n < (int64_t)(1 << 20)) {
                small_.insert(small_.end(), data, data + n);
                if (small_.size() >= ((size_t)1 << 20)) flush_small();
                return;
            }
            flush_small();
            put(data, (size_t)n);
            return;
        }
        pending_.insert(pending_.end(), data, data + n);
        if (pending_.size() >= (size_t)n_threads_ * BLOCK) compress_pending(false);
    }
    void close() {
        if (fd_ < 0) return;
        if (gzip_) {
            compress_pending(true);
            if (!wrote_member_) {   // an empty gzip file still needs one (empty) member
                std::vector<char> dummy;
                std::vector<unsigned char> out;
                deflate_block(dummy.data(), 0, out);
                put(out.data(), out.size());
            }
        }
        flush_small();
        const int f = fd_;
        fd_ = -1;
        if (regular_ && old_size_ > written_ && ::ftruncate(f, (off_t)written_) != 0) {
            ::close(f);
            throw std::runtime_error("Error writing to file " + filename_);
        }
        if (::close(f) != 0) throw std::runtime_error("Error writing to file " + filename_);
    }

private:
    static constexpr size_t BLOCK = 1 << 20;
    static void deflate_block(const char *src, size_t n, std::vector<unsigned char> &out) {
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (deflateInit2(&zs, Z_DEFAULT_COMPRESSION, Z_DEFLATED, 15 + 16, 8, Z_DEFAULT_STRATEGY) != Z_OK)
            throw std::runtime_error("deflateInit2 failed");
        out.resize(deflateBound(&zs, (uLong)n) + 64);
        zs.next_in = (Bytef *)src;
        zs.avail_in = (uInt)n;
        zs.next_out = out.data();
        zs.avail_out = (uInt)out.size();
        int rc = deflate(&zs, Z_FINISH);
        size_t produced = out.size() - zs.avail_out;
        deflateEnd(&zs);
        if (rc != Z_STREAM_END) throw std::runtime_error("deflate failed");
        out.resize(produced);
    }
    void compress_pending(bool all) {
        size_t n_blocks = all ? (pending_.size() + BLOCK - 1) / BLOCK : pending_.size() / BLOCK;
        if (n_blocks == 0) return;
        std::vector<std::vector<unsigned char>> outs(n_blocks);
        std::vector<std::string> errs((size_t)n_threads_);
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads_; t++)
            th.emplace_back([&, t] {
                try {
                    for (size_t b = (size_t)t; b < n_blocks; b += (size_t)n_threads_) {
                        size_t lo = b * BLOCK, hi = std::min(pending_.size(), lo + BLOCK);
                        deflate_block(pending_.data() + lo, hi - lo, outs[b]);
                    }
                } catch (const std::exception &e) {
                    errs[(size_t)t] = e.what();
                }
            });
        for (auto &t : th) t.join();
        for (auto &e : errs) if (!e.empty()) throw std::runtime_error("Error compressing " + filename_ + ": " + e);
        for (auto &o : outs) {
            put(o.data(), o.size());
            wrote_member_ = true;
        }
        size_t used = std::min(pending_.size(), n_blocks * BLOCK);
        pending_.erase(pending_.begin(), pending_.begin() + (long)used);
    }
    // sequential append with write(2): works on pipes, FIFOs and /dev/stdout as well as on files (the reference writes
    // through an ofstream); EINTR and short writes are retried
    void put(const void *data, size_t n) {
        const char *p = (const char *)data;
        while (n > 0) {
            const ssize_t w = ::write(fd_, p, std::min(n, (size_t)1 << 30));
            if (w < 0 && errno == EINTR) continue;
            if (w <= 0) throw std::runtime_error("Error writing to file " + filename_);
            p += w;
            n -= (size_t)w;
            written_ += (int64_t)w;
        }
    }
    void flush_small() {
        if (small_.empty()) return;
        put(small_.data(), small_.size());
        small_.clear();
    }
    std::string filename_;
    bool gzip_;
    int n_threads_ = 1;
    int fd_ = -1;
    std::vector<char> small_;
    std::vector<char> pending_;
    bool wrote_member_ = false;
    bool regular_ = false;          // a regular file: cut to written_ bytes at close if it was longer before
    int64_t old_size_ = 0, written_ = 0;
};

// FASTA writer used by tests and `sbwt build --add-reverse-complements`
inline void write_fasta(const std::string &filename, const std::vector<std::string> &seqs, bool gzip) {
    Buffered_ofstream out(filename, gzip);
    for (const auto &s : seqs) {
        out.write(">\n", 2);
        out.write(s.data(), (int64_t)s.size());
        out.write("\n", 1);
    }
}

}  // namespace seq_io
}  // namespace sbwt
