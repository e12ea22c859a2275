// globals.hh -- the few helpers of the reference's include/sbwt/globals.hh + src/globals.cpp that the
// search path touches: log lines (same text; timestamps are not parity relevant), timers, string
// (de)serialisation of the index file framing, readable/writable checks, readlines.
#pragma once
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cstdint>
#include <ctime>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <istream>
#include <mutex>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

namespace sbwt {

enum LogLevel { OFF = 0, MAJOR = 1, MINOR = 2, DEBUG = 3 };   // globals.hh:61

inline long long cur_time_micros() {                            // globals.cpp:68-70
    return std::chrono::duration_cast<std::chrono::microseconds>(
               std::chrono::high_resolution_clock::now().time_since_epoch())
        .count();
}

namespace detail {
inline LogLevel &loglevel() { static LogLevel l = MAJOR; return l; }
inline long long &start_micros() { static long long t = cur_time_micros(); return t; }
inline std::mutex &log_mutex() { static std::mutex m; return m; }
}  // namespace detail

inline void set_log_level(LogLevel level) { detail::loglevel() = level; }
inline LogLevel get_log_level() { return detail::loglevel(); }

// globals.cpp:94-105 (the "seconds" are really milliseconds there too: micros / 1000)
inline void write_log(const std::string &message, LogLevel level) {
    if (level <= detail::loglevel()) {
        std::lock_guard<std::mutex> lock(detail::log_mutex());
        std::time_t now = std::time(nullptr);
        std::string ts = std::asctime(std::localtime(&now));
        if (!ts.empty()) ts.pop_back();
        std::cerr << std::setprecision(4) << std::fixed << (cur_time_micros() - detail::start_micros()) / 1000.0 << " "
                  << ts << " " << message << std::endl;
    }
}

// globals.cpp:49-62
inline int64_t serialize_string(const std::string &S, std::ostream &out) {
    int64_t size = (int64_t)S.size();
    out.write((const char *)&size, sizeof(size));
    out.write(S.data(), size);
    return (int64_t)sizeof(size) + size;
}
inline std::string load_string(std::istream &in) {
    int64_t size = 0;
    in.read((char *)&size, sizeof(size));
    if (!in.good() || size < 0 || size > (1 << 20)) throw std::runtime_error("Error: corrupt string in index file");
    std::string S((size_t)size, '\0');
    in.read(&S[0], size);
    return S;
}

// throwing_streams.hh:32-35 message
// (a FIFO, a pipe behind /dev/fd/N or /dev/stdout is not opened for the check: opening and closing a FIFO would cut its
// writer off before the real reader arrives; the permission bits answer instead)
// (only FIFOs, character devices and sockets take the shortcut: a directory passes access(W_OK) and must fail here like the
// reference's stream probe does, not later in the writer)
inline bool is_regular_or_missing(const std::string &filename) {
    struct stat sb;
    if (::stat(filename.c_str(), &sb) != 0) return true;
    return !(S_ISFIFO(sb.st_mode) || S_ISCHR(sb.st_mode) || S_ISSOCK(sb.st_mode));
}
inline void check_readable(const std::string &filename) {       // globals.cpp:38-40
    if (!is_regular_or_missing(filename)) {
        if (::access(filename.c_str(), R_OK) != 0) throw std::runtime_error("Error opening file: " + filename);
        return;
    }
    std::ifstream f(filename);
    if (!f.good()) throw std::runtime_error("Error opening file: " + filename);
}
inline void check_writable(const std::string &filename) {       // globals.cpp:43-46 (out|app: does not truncate)
    if (!is_regular_or_missing(filename)) {
        if (::access(filename.c_str(), W_OK) != 0) throw std::runtime_error("Error opening file: " + filename);
        return;
    }
    std::ofstream f(filename, std::ofstream::out | std::ofstream::app);
    if (!f.good()) throw std::runtime_error("Error opening file: " + filename);
}
// ACGT -> 0123, anything else (lower case included) -> -1 (globals.hh:38-47, SBWT.hh:49-57)
inline int DNA_to_char_idx(char c) {
    switch (c) {
        case 'A': return 0;
        case 'C': return 1;
        case 'G': return 2;
        case 'T': return 3;
        default: return -1;
    }
}
inline std::vector<std::string> readlines(const std::string &filename) {   // globals.cpp:20-31
    std::ifstream in(filename);
    if (!in.good()) throw std::runtime_error("Error opening file: " + filename);
    std::vector<std::string> lines;
    std::string line;
    while (std::getline(in, line)) lines.push_back(line);
    return lines;
}

}  // namespace sbwt
