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
    ~Reader() {
        if (f_) gzclose(f_);
        if (plain_fd_ >= 0) ::close(plain_fd_);
    }
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
                return false;                   // (get_next_read_to_buffer() returns 0 for it, which callers take as EOF)
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
        if (plain_fd_ >= 0) {        // not compressed: straight read(2) into the parse buffer (gzread copies twice)
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
};

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
