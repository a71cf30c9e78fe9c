#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
int main() {
    const size_t MB = 1 << 20, n = 512 * MB;
    void* d; CK(hipMalloc(&d, n));
    double t0 = now();
    void* hp; CK(hipHostMalloc(&hp, n, hipHostMallocDefault));
    double t1 = now();
    printf("hipHostMalloc 512 MB: %.1f ms\n", (t1 - t0) * 1e3);
    t0 = now(); memset(hp, 1, n); t1 = now();
    printf("first touch memset pinned: %.1f ms\n", (t1 - t0) * 1e3);
    char* pg = (char*)malloc(n);
    t0 = now(); memset(pg, 1, n); t1 = now();
    printf("first touch memset pageable: %.1f ms\n", (t1 - t0) * 1e3);
    t0 = now(); CK(hipHostRegister(pg, n, hipHostRegisterDefault)); t1 = now();
    printf("hipHostRegister 512 MB: %.1f ms\n", (t1 - t0) * 1e3);
    t0 = now(); CK(hipHostUnregister(pg)); t1 = now();
    printf("hipHostUnregister: %.1f ms\n", (t1 - t0) * 1e3);
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int rep = 0; rep < 2; ++rep) {
        t0 = now(); CK(hipMemcpyAsync(d, hp, n, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); t1 = now();
        printf("H2D pinned: %.1f GB/s\n", n / (t1 - t0) / 1e9);
        t0 = now(); CK(hipMemcpyAsync(hp, d, n, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); t1 = now();
        printf("D2H pinned: %.1f GB/s\n", n / (t1 - t0) / 1e9);
        t0 = now(); CK(hipMemcpyAsync(d, pg, n, hipMemcpyHostToDevice, s)); CK(hipStreamSynchronize(s)); t1 = now();
        printf("H2D pageable: %.1f GB/s\n", n / (t1 - t0) / 1e9);
        t0 = now(); CK(hipMemcpyAsync(pg, d, n, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s)); t1 = now();
        printf("D2H pageable: %.1f GB/s\n", n / (t1 - t0) / 1e9);
    }
    // duplex pinned
    void* hp2; CK(hipHostMalloc(&hp2, n, hipHostMallocDefault)); memset(hp2, 0, n);
    void* d2; CK(hipMalloc(&d2, n));
    hipStream_t s2; CK(hipStreamCreate(&s2));
    t0 = now(); CK(hipMemcpyAsync(d, hp, n, hipMemcpyHostToDevice, s)); CK(hipMemcpyAsync(hp2, d2, n, hipMemcpyDeviceToHost, s2)); CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2)); t1 = now();
    printf("duplex pinned: %.1f GB/s each way\n", n / (t1 - t0) / 1e9);
    t0 = now(); memcpy(pg, hp, n); t1 = now();
    printf("CPU memcpy pinned->pageable 1 thread: %.1f GB/s\n", n / (t1 - t0) / 1e9);
    return 0;
}
