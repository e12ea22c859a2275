// How fast can 3 GiB of text reach the page cache of ONE file: write(2) from one thread (buffered writes take the inode lock)
// against pieces mapped with mmap and filled by T threads (ftruncate or posix_fallocate first).  usage: mmap_write <path> <threads>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const char *path = argv[1];
    const int T = atoi(argv[2]);
    const size_t piece = (size_t)32 << 20, total = (size_t)3 << 30;
    std::vector<char> src(piece, 'x');
    for (int mode = 0; mode < 5; mode++) {        // 0 write(), 1 mmap + T threads, 2 fallocate + mmap + T threads, 3 fallocate(all) + write(), 4 write() of 4 MiB pieces
        unlink(path);
        int fd = open(path, O_RDWR | O_CREAT, 0666);
        double t0 = now();
        size_t pos = 0;
        if (mode == 3 && posix_fallocate(fd, 0, (off_t)total) != 0) { perror("fallocate"); return 1; }
        while (pos < total) {
            if (mode == 4) {
                for (size_t o = 0; o < piece; o += (size_t)4 << 20) {
                    size_t done = 0, n = (size_t)4 << 20;
                    while (done < n) done += (size_t)write(fd, src.data() + o + done, n - done);
                }
            } else if (mode == 0 || mode == 3) {
                size_t done = 0;
                while (done < piece) done += (size_t)write(fd, src.data() + done, piece - done);
            } else {
                if (mode == 2) { if (posix_fallocate(fd, (off_t)pos, (off_t)piece) != 0) { perror("fallocate"); return 1; } }
                else if (ftruncate(fd, (off_t)(pos + piece)) != 0) { perror("ftruncate"); return 1; }
                const size_t a = pos & ~(size_t)4095;
                char *m = (char *)mmap(nullptr, pos + piece - a, PROT_READ | PROT_WRITE, MAP_SHARED, fd, (off_t)a);
                if (m == MAP_FAILED) { perror("mmap"); return 1; }
                char *dst = m + (pos - a);
                std::vector<std::thread> th;
                const size_t per = (piece + T - 1) / T;
                for (int t = 0; t < T; t++)
                    th.emplace_back([=] { size_t lo = t * per, hi = lo + per < piece ? lo + per : piece; if (lo < hi) memcpy(dst + lo, src.data() + lo, hi - lo); });
                for (auto &x : th) x.join();
                munmap(m, pos + piece - a);
            }
            pos += piece;
        }
        double t1 = now();
        close(fd);
        double t2 = now();
        printf("mode %d (%s): write %.2f s (%.1f GB/s) close %.2f s\n", mode, mode == 0 ? "write()" : mode == 1 ? "ftruncate + mmap" : mode == 2 ? "fallocate + mmap" : mode == 3 ? "fallocate(all) + write()" : "write() of 4 MiB pieces", t1 - t0, total / (t1 - t0) / 1e9, t2 - t1);
    }
    unlink(path);
    return 0;
}
