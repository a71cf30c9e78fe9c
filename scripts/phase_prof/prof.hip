// Profiling aid (not product, not a test): per-phase cycle stamps of one wave of the band kernel.  The kernel
// source (upx_core.h) runs with an executor that records s_memtime after every phase; see run.sh / ana.py.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../upmix_amd/csrc/upx_core.h"

template <bool WAVE_SYNC, int P>
struct ProfExec {
    upx::ThreadT<P> st;
    unsigned long long* buf;
    int idx = 0, cap = 0;
    bool rec = false;
    __device__ __forceinline__ void stamp() {
        if (rec) {
            unsigned long long t = __builtin_amdgcn_s_memtime();
            if (idx < cap && (threadIdx.x & 63) == 0) buf[idx] = t;
            ++idx;
        }
    }
    __device__ __forceinline__ void sync() {
        if constexpr (WAVE_SYNC) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        }
    }
    __device__ __forceinline__ void wg_barrier() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        stamp();
    }
    template <class F>
    __device__ __forceinline__ void each(F&& f) {
        f((int)threadIdx.x, st);
        stamp();
        sync();
        if constexpr (!WAVE_SYNC) stamp();
    }
    template <class F, class G>
    __device__ __forceinline__ void each2(F&& f, G&& g) {
        f((int)threadIdx.x, st);
        stamp();
        if constexpr (!WAVE_SYNC) { sync(); stamp(); }
        g((int)threadIdx.x, st);
        stamp();
        sync();
        if constexpr (!WAVE_SYNC) stamp();
    }
};

template <class C, class LV>
__global__ __launch_bounds__(C::WG, 2) void prof_kernel(upx::BandArgs a, unsigned long long* buf, int cap, int wave) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    ProfExec<C::WAVE_SYNC || C::WIDE, C::P> ex;
    ex.buf = buf; ex.cap = cap;
    // a workgroup in the middle of the launch: it runs band_program's interior flavour (rotated loop)
    ex.rec = blockIdx.x == gridDim.x / 2 && (int)(threadIdx.x / 64) == wave;
    // single band (MERGED = false) with the live-slot flavour the plan would pick, like the product launch
    upx::band_program_auto<C, decltype(ex), false, LV>(ex, a, reinterpret_cast<upx::cf*>(smem), (int)blockIdx.x);
}

static void turn_trig(double frac, double& c, double& s) { c = std::cos(2 * M_PI * frac); s = std::sin(2 * M_PI * frac); }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <class C, class LV = upx::LiveAll>
int run(int F, int n_wg, int wave, int n_gain) {
    const int N = C::N, HOP = C::HOP;
    const long long n_streams = (long long)n_wg * C::G;
    const long long blocks = n_streams * F;
    const long long T = blocks * HOP;
    std::vector<float> in(2 * T), wa(N), ws(N), gain((size_t)n_gain * (N / 2 + 1));
    for (auto& v : in) v = (float)rand() / RAND_MAX - 0.5f;
    for (int i = 0; i < N; ++i) { wa[i] = 0.5f - 0.5f * cosf(2 * M_PI * i / N); ws[i] = wa[i] / N; }
    for (auto& g : gain) g = 0.5f;
    std::vector<upx::cf> tw(C::TW_CF);
    upx::fill_tables<C>(tw.data(), turn_trig);
    float *d_in, *d_c, *d_l, *d_r, *d_wa, *d_ws, *d_gain, *d_seam; upx::cf* d_tw; unsigned long long* d_buf;
    const int cap = 1 << 16;
    CK(hipMalloc(&d_in, in.size() * 4)); CK(hipMalloc(&d_c, T * 4)); CK(hipMalloc(&d_l, T * 4)); CK(hipMalloc(&d_r, T * 4));
    CK(hipMalloc(&d_wa, N * 4)); CK(hipMalloc(&d_ws, N * 4)); CK(hipMalloc(&d_gain, gain.size() * 4)); CK(hipMalloc(&d_tw, tw.size() * 8));
    CK(hipMalloc(&d_seam, n_streams * 3 * (C::P - C::HS) * C::LANES * 4)); CK(hipMalloc(&d_buf, cap * 8));
    CK(hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_wa, wa.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_ws, ws.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_gain, gain.data(), gain.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_tw, tw.data(), tw.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(d_buf, 0, cap * 8));
    upx::BandArgs a{};
    a.in = (const upx::cf*)d_in; a.out_c = d_c; a.out_l = d_l; a.out_r = d_r; a.w_a = d_wa; a.w_s = d_ws; a.gain = d_gain; a.tw = d_tw;
    a.t_in = (int)T; a.t_out = (int)T; a.j_lo = 0; a.j_hi = (int)blocks; a.m_lo = 0; a.m_hi = (int)blocks;
    a.blocks_per_stream = F; a.n_gain = n_gain; a.gain_stride = N / 2 + 1; a.accumulate = 0; a.seam = d_seam;
    const int lds = C::LDS_CF * 8;
    const void* kfn = reinterpret_cast<const void*>(&prof_kernel<C, LV>);
    CK(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((prof_kernel<C, LV>), dim3(n_wg), dim3(C::WG), lds, 0, a, d_buf, cap, wave);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("N=%d wide=%d F=%d wgs=%d lds=%d: %.3f ms  (%.1f Msamples/s)\n", N, (int)C::WIDE, F, n_wg, lds, ms, T / ms / 1e3);
    }
    std::vector<unsigned long long> buf(cap);
    CK(hipMemcpy(buf.data(), d_buf, cap * 8, hipMemcpyDeviceToHost));
    int n = 0; while (n < cap && buf[n]) ++n;
    printf("events %d\n", n);
    if (const char* path = getenv("PROF_OUT")) {
        FILE* f = fopen(path, "wb");
        fwrite(buf.data(), 8, n, f);
        fclose(f);
    }
    return 0;
}

int main(int argc, char** argv) {
    const int log2n = argc > 1 ? atoi(argv[1]) : 13;
    const int wide = argc > 2 ? atoi(argv[2]) : 1;
    const int F = argc > 3 ? atoi(argv[3]) : 56;
    const int wgs = argc > 4 ? atoi(argv[4]) : 256;
    const int wave = argc > 5 ? atoi(argv[5]) : 0;
    const int ng = argc > 6 ? atoi(argv[6]) : 1;
    if (log2n == 13 && wide) return run<upx::WideCfg<13, 4>>(F, wgs, wave, ng);
    if (log2n == 13) return run<upx::Cfg<13, 4, 16>>(F, wgs, wave, ng);
    if (log2n == 12 && wide) return run<upx::WideCfg<12, 4>>(F, wgs, wave, ng);
    if (log2n == 12) return run<upx::Cfg<12, 4, 16>>(F, wgs, wave, ng);
    if (log2n == 10) return run<upx::Cfg<10, 4, 16>, upx::Live<0, 4>>(F, wgs, wave, ng);
    if (log2n == 8) return run<upx::Cfg<8, 4, 16>>(F, wgs, wave, ng);
    return 1;
}
