// Profiling aid (not product, not a test): when does each workgroup of a fused band launch start and end?
// Every workgroup stamps s_memrealtime (100 MHz, the same clock on every CU) before and after band_program_auto; the host
// prints the spread: if the first / last workgroups (signal-edge flavour) end late, the launch waits for stragglers.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -w -o scripts/phase_prof/wgtime.exe scripts/phase_prof/wgtime.hip
//   wgtime.exe log2N F workgroups live_s1 accumulate prio_split
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../upmix_amd/csrc/upx_kernels.h"

template <class C, class LV>
__global__ __launch_bounds__(C::WG, 2) void wg_kernel(upx::BandArgs a, unsigned long long* t) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    upxk::DevExec<C::WAVE_SYNC || C::WIDE, C::P> ex;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    upx::band_program_auto<C, decltype(ex), false, LV>(ex, a, reinterpret_cast<upx::cf*>(smem), (int)blockIdx.x);
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { t[2 * blockIdx.x] = t0; t[2 * blockIdx.x + 1] = t1; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <class C, class LV>
int run(int F, int n_wg, int live_s1, int accumulate, int prio_split, int prio_young) {
    const int N = C::N, HOP = C::HOP;
    const long long n_streams = (long long)n_wg * C::G;
    const long long blocks = n_streams * F - 1;            // streams start one frame early (frame -1): the library's geometry
    const long long T = blocks * HOP;
    std::vector<float> in(2 * T), wa(N), ws(N), gain(N / 2 + 1, 0.f);
    for (auto& v : in) v = (float)rand() / RAND_MAX - 0.5f;
    for (int i = 0; i < N; ++i) { wa[i] = 0.5f - 0.5f * cosf(2 * M_PI * i / N); ws[i] = wa[i] / N; }
    for (int k = 31; k < std::min(live_s1 * C::LANES, N / 2 + 1); ++k) gain[k] = 0.5f;
    if (live_s1 >= 8) gain[N / 2] = 0.5f;
    std::vector<upx::cf> tw(C::TW_CF);
    upx::fill_tables<C>(tw.data(), upxk::turn_trig);
    float *d_in, *d_c, *d_l, *d_r, *d_wa, *d_ws, *d_gain, *d_seam; upx::cf* d_tw; unsigned long long* d_t;
    CK(hipMalloc(&d_in, in.size() * 4)); CK(hipMalloc(&d_c, T * 4)); CK(hipMalloc(&d_l, T * 4)); CK(hipMalloc(&d_r, T * 4));
    CK(hipMalloc(&d_wa, N * 4)); CK(hipMalloc(&d_ws, N * 4)); CK(hipMalloc(&d_gain, gain.size() * 4)); CK(hipMalloc(&d_tw, tw.size() * 8));
    CK(hipMalloc(&d_seam, n_streams * 3 * (C::P - C::HS) * C::LANES * 4)); CK(hipMalloc(&d_t, n_wg * 16));
    CK(hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_wa, wa.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_ws, ws.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_gain, gain.data(), gain.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_tw, tw.data(), tw.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(d_c, 0, T * 4)); CK(hipMemset(d_l, 0, T * 4)); CK(hipMemset(d_r, 0, T * 4));
    upx::BandArgs a{};
    a.in = (const upx::cf*)d_in; a.out_c = d_c; a.out_l = d_l; a.out_r = d_r; a.w_a = d_wa; a.w_s = d_ws; a.gain = d_gain; a.tw = d_tw;
    a.t_in = (int)T; a.t_out = (int)T; a.j_lo = 0; a.j_hi = (int)blocks; a.m_lo = 0; a.m_hi = (int)blocks;
    a.blocks_per_stream = F; a.n_gain = 1; a.gain_stride = N / 2 + 1; a.accumulate = accumulate; a.seam = d_seam; a.prio_split = prio_split; a.prio_young = prio_young;
    const int lds = C::LDS_CF * 8;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wg_kernel<C, LV>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<unsigned long long> t(2 * n_wg);
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((wg_kernel<C, LV>), dim3(n_wg), dim3(C::WG), lds, 0, a, d_t);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(t.data(), d_t, n_wg * 16, hipMemcpyDeviceToHost));
        unsigned long long s0 = ~0ull, s1 = 0, e_max = 0;
        std::vector<double> dur(n_wg), end(n_wg);
        for (int i = 0; i < n_wg; ++i) { s0 = std::min(s0, t[2 * i]); s1 = std::max(s1, t[2 * i]); e_max = std::max(e_max, t[2 * i + 1]); }
        for (int i = 0; i < n_wg; ++i) { dur[i] = (t[2 * i + 1] - t[2 * i]) * 0.01; end[i] = (t[2 * i + 1] - s0) * 0.01; }
        std::vector<double> se(end); std::sort(se.begin(), se.end());
        std::vector<double> sd(dur); std::sort(sd.begin(), sd.end());
        printf("N=%d F=%d wgs=%d: %.3f ms by events; starts within %.1f us; end (us since first start) median %.1f  p90 %.1f  p99 %.1f  max %.1f; "
               "duration median %.1f max %.1f\n", N, F, n_wg, ms, (s1 - s0) * 0.01, se[n_wg / 2], se[n_wg * 9 / 10], se[n_wg * 99 / 100],
               se[n_wg - 1], sd[n_wg / 2], sd[n_wg - 1]);
        if (rep == 3) {
            printf("  first workgroups end at:");
            for (int i = 0; i < 4; ++i) printf(" wg%d %.1f", i, end[i]);
            printf("   last:");
            for (int i = n_wg - 4; i < n_wg; ++i) printf(" wg%d %.1f", i, end[i]);
            int worst = (int)(std::max_element(end.begin(), end.end()) - end.begin());
            printf("   latest: wg%d\n", worst);
            const int nb = 32, per = n_wg / nb;
            printf("  mean end per %d-workgroup index bin:", per);
            for (int b = 0; b < nb; ++b) {
                double m = 0;
                for (int i = b * per; i < (b + 1) * per; ++i) m += end[i];
                printf(" %.0f", m / per);
            }
            printf("\n");
        }
    }
    return 0;
}

int main(int argc, char** argv) {
    const int log2n = argc > 1 ? atoi(argv[1]) : 10;
    const int F = argc > 2 ? atoi(argv[2]) : 56;
    const int wgs = argc > 3 ? atoi(argv[3]) : 2048;
    const int s1 = argc > 4 ? atoi(argv[4]) : 8;
    const int acc = argc > 5 ? atoi(argv[5]) : 1;
    const int ps = argc > 6 ? atoi(argv[6]) : 0;
    const int py = argc > 7 ? atoi(argv[7]) : 2;
    if (log2n == 10 && s1 == 4) return run<upx::Cfg<10, 4, 16>, upx::Live<0, 4>>(F, wgs, s1, acc, ps, py);
    if (log2n == 10) return run<upx::Cfg<10, 4, 16>, upx::LiveAll>(F, wgs, 8, acc, ps, py);
    if (log2n == 8) return run<upx::Cfg<8, 4, 16>, upx::LiveAll>(F, wgs, 8, acc, ps, py);
    if (log2n == 11 && s1 == 2) return run<upx::Cfg<11, 4, 16>, upx::Live<0, 2>>(F, wgs, s1, acc, ps, py);
    if (log2n == 9) return run<upx::Cfg<9, 4, 16>, upx::LiveAll>(F, wgs, 8, acc, ps, py);
    return 1;
}
