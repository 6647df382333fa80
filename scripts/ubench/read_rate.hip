// How fast do T threads read files from the page cache into (a) pinned (hipHostMalloc) and (b) pageable buffers, reused or fresh?
// Build: hipcc -O2 -o read_rate read_rate.hip -lpthread ; run: ./read_rate <dir with f0.fasta .. f9.fasta> (scripts/e2e_full.py writes such files with KEEP=1)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>
#include <atomic>
#include <fcntl.h>
#include <unistd.h>
#include <sys/stat.h>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const std::string dir = argc > 1 ? argv[1] : ".";
    const int nfiles = 10, reps = argc > 2 ? atoi(argv[2]) : 3;
    struct stat st; std::string f0 = dir + "/f0.fasta";
    if (stat(f0.c_str(), &st) != 0) { printf("no files in %s\n", dir.c_str()); return 1; }
    const size_t fsize = (size_t)st.st_size;
    for (int pinned = 0; pinned <= 1; pinned++)
        for (int T : {4, 8, 16, 32, 64}) {
            std::vector<char *> buf(T);
            const double ta = now();
            for (int t = 0; t < T; t++) { if (pinned) { if (hipHostMalloc((void **)&buf[t], fsize, hipHostMallocDefault) != hipSuccess) { printf("alloc failed\n"); return 1; } } else buf[t] = (char *)malloc(fsize); }
            const double tb = now();
            std::atomic<int> next(0);
            const int total = T * reps;
            const double t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; t++) th.emplace_back([&, t] {
                for (;;) {
                    const int i = next.fetch_add(1); if (i >= total) break;
                    const std::string fn = dir + "/f" + std::to_string(i % nfiles) + ".fasta";
                    const int fd = open(fn.c_str(), O_RDONLY); size_t got = 0;
                    while (got < fsize) { const ssize_t r = pread(fd, buf[t] + got, std::min<size_t>(fsize - got, (size_t)64 << 20), (off_t)got); if (r <= 0) break; got += (size_t)r; }
                    close(fd);
                }
            });
            for (auto &x : th) x.join();
            const double dt = now() - t0;
            printf("%s buffers, %2d threads: alloc %.2f s (%.1f GB/s), read %.1f GB in %.2f s = %.1f GB/s\n", pinned ? "pinned  " : "pageable", T, tb - ta, T * fsize / 1e9 / (tb - ta), total * fsize / 1e9, dt, total * fsize / 1e9 / dt);
            for (int t = 0; t < T; t++) { if (pinned) (void)hipHostFree(buf[t]); else free(buf[t]); }
        }
    return 0;
}
