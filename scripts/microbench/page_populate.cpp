#include <sys/mman.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t n = 345u << 20;
    for (int huge = 0; huge < 2; ++huge)
        for (int nt : {1, 2, 4, 8, 16}) {
            char* p = (char*)mmap(nullptr, n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (huge) madvise(p, n, MADV_HUGEPAGE);
            double t0 = now();
            std::vector<std::thread> th;
            for (int t = 0; t < nt; ++t)
                th.emplace_back([=] {
                    size_t lo = n * t / nt / 4096 * 4096, hi = n * (t + 1) / nt / 4096 * 4096;
                    if (madvise(p + lo, hi - lo, MADV_POPULATE_WRITE) != 0) { for (size_t q = lo; q < hi; q += 4096) p[q] = 0; }
                });
            for (auto& t : th) t.join();
            double t1 = now();
            printf("huge=%d threads=%2d populate 345 MB: %.1f ms (%.1f GB/s)\n", huge, nt, (t1 - t0) * 1e3, n / (t1 - t0) / 1e9);
            munmap(p, n);
        }
    FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
    if (f) { char b[128] = {0}; fgets(b, 127, f); printf("THP enabled: %s", b); fclose(f); }
    return 0;
}
