// ThreadSanitizer harness of seqio.hh read_file_chunked (tools/run_sanitizers.sh): a FASTQ file of 60 000 reads parsed in pieces of
// 200 KB by 1, 2, 4 and 8 threads -- the same reads and bases every time, no report.
#include "seqio.hh"
#include <cstdio>
#include <random>
int main() {
    const char *path = "/tmp/tsan_chunk.fastq";
    {
        FILE *f = fopen(path, "wb");
        std::mt19937 rng(3);
        for (int i = 0; i < 60000; i++) {
            int len = 1 + rng() % 300;
            std::string s(len, 'A'), q(len, '@');
            for (auto &c : s) c = "ACGTn"[rng() % 5];
            fprintf(f, "@r%d\n%s\n+\n%s\n", i, s.c_str(), q.c_str());
        }
        fclose(f);
    }
    int64_t size = 0;
    if (!sbwt::seq_io::chunkable_file(path, &size)) return 2;
    for (int threads : {1, 2, 4, 8}) {
        int64_t n = 0, bases = 0;
        sbwt::seq_io::read_file_chunked(path, size, 200000, threads, [&](std::vector<char> &&b, std::vector<int64_t> &&off, bool) {
            n += (int64_t)off.size() - 1; bases += (int64_t)b.size();
        });
        printf("threads %d: %lld reads %lld bases\n", threads, (long long)n, (long long)bases);
    }
    return 0;
}
