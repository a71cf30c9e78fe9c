// Profiling aid (not product, not a test): when does each workgroup of the band-limited kernels start and end?
// Like wgtime.hip for the fused kernel: every workgroup stamps s_memrealtime (100 MHz) around its program; the host prints
// the spread per dispatch-order decile, so late starters / stragglers and the launch's idle tail can be read off.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -w -o scripts/phase_prof/zwgtime.exe scripts/phase_prof/zwgtime.hip
//   zwgtime.exe log2N log2P RGa RGs F frames n_gain prio Fc
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../upmix_amd/csrc/upx_kernels.h"

template <class Z>
__global__ __launch_bounds__(Z::WG, Z::WPE_A) void zana(upx::ZoomArgs a, unsigned long long* t) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    upxk::DevExec<true, 16> ex;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    upx::zoom_analysis_program<Z>(ex, a, reinterpret_cast<upx::cf*>(smem), (int)blockIdx.x);
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { t[2 * blockIdx.x] = t0; t[2 * blockIdx.x + 1] = t1; }
}
template <class Z>
__global__ __launch_bounds__(Z::WG, Z::WPE_S) void zsyn(upx::ZoomArgs a, unsigned long long* t, int n_groups) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    upxk::DevExec<true, 16> ex;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    int b = (int)blockIdx.x;   // the product kernel's order (upx_kernels.h): Ls/Rs streams, then the centre streams
    const int n0 = a.ns_lr * n_groups;
    if (b < n0) upx::zoom_synthesis_program<Z>(ex, a, reinterpret_cast<upx::cf*>(smem), b % a.ns_lr, b / a.ns_lr, 0);
    else upx::zoom_synthesis_program<Z>(ex, a, reinterpret_cast<upx::cf*>(smem), (b - n0) % a.ns_c, (b - n0) / a.ns_c, 1);
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    const size_t i = blockIdx.x;
    if (threadIdx.x == 0) { t[2 * i] = t0; t[2 * i + 1] = t1; }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static void report(const char* what, const std::vector<unsigned long long>& t, size_t n, float ms, size_t split) {
    unsigned long long lo = ~0ull, hi = 0;
    for (size_t i = 0; i < n; ++i) { lo = std::min(lo, t[2 * i]); hi = std::max(hi, t[2 * i + 1]); }
    printf("%s: %zu workgroups, event time %.1f us, first start .. last end %.1f us\n", what, n, ms * 1e3, (hi - lo) / 100.0);
    const int parts = 16;
    printf("  dispatch-order part: start(min/mean)  end(mean/max)  duration(mean) [us]\n");
    for (int p = 0; p < parts; ++p) {
        const size_t a = n * p / parts, b = n * (p + 1) / parts;
        if (b <= a) continue;
        double s_min = 1e30, s_mean = 0, e_mean = 0, e_max = 0, d_mean = 0;
        for (size_t i = a; i < b; ++i) {
            const double s = (t[2 * i] - lo) / 100.0, e = (t[2 * i + 1] - lo) / 100.0;
            s_min = std::min(s_min, s); s_mean += s; e_mean += e; e_max = std::max(e_max, e); d_mean += e - s;
        }
        const double c = (double)(b - a);
        printf("  %6zu..%-6zu%s  %7.1f %7.1f   %7.1f %7.1f   %7.1f\n", a, b - 1, (split && a >= split) ? " C" : "  ", s_min, s_mean / c,
               e_mean / c, e_max, d_mean / c);
    }
    // how much of the launch has every slot busy: histogram of end times
    std::vector<double> ends(n);
    for (size_t i = 0; i < n; ++i) ends[i] = (t[2 * i + 1] - lo) / 100.0;
    std::sort(ends.begin(), ends.end());
    printf("  end-time quantiles [us]: 10%% %.1f  50%% %.1f  90%% %.1f  99%% %.1f  100%% %.1f\n", ends[n / 10], ends[n / 2], ends[n * 9 / 10],
           ends[n * 99 / 100], ends[n - 1]);
}

template <class ZA, class ZS>
int run(int log2n, int F, int frames, int n_gain, int prio, int c_split) {
    const int N = 1 << log2n, P = ZS::P, D = N / P, hop = N / ZS::K;
    const long long n_streams = frames / F;
    frames = (int)(n_streams * F);
    const long long T = (long long)frames * hop;
    std::vector<float> in(2 * T), wa(N), ws(N), gain((size_t)n_gain * (N / 2 + 1), 0.f);
    for (auto& v : in) v = (float)rand() / RAND_MAX - 0.5f;
    for (int i = 0; i < N; ++i) { wa[i] = 0.5f - 0.5f * cosf(2 * M_PI * i / N); ws[i] = wa[i] / N; }
    for (int q = 0; q < n_gain; ++q)
        for (int k = P / 16; k < P / 2 - P / 10; ++k) gain[(size_t)q * (N / 2 + 1) + k] = 0.5f;
    std::vector<upx::cf> tw(ZS::TW_CF), ramp(upx::zoom_ramp_count(N, P));
    upx::fill_twiddles<typename ZS::Sub>(tw.data(), upxk::turn_trig);
    upx::fill_zoom_ramp(ramp.data(), N, P, upxk::turn_trig);
    float *d_in, *d_c, *d_l, *d_r, *d_wa, *d_ws, *d_gain, *d_seam; upx::cf *d_tw, *d_ramp, *d_y; unsigned long long* d_t;
    CK(hipMalloc(&d_in, in.size() * 4)); CK(hipMalloc(&d_c, T * 4)); CK(hipMalloc(&d_l, T * 4)); CK(hipMalloc(&d_r, T * 4));
    CK(hipMalloc(&d_wa, N * 4)); CK(hipMalloc(&d_ws, N * 4)); CK(hipMalloc(&d_gain, gain.size() * 4));
    CK(hipMalloc(&d_tw, tw.size() * 8)); CK(hipMalloc(&d_ramp, ramp.size() * 8));
    CK(hipMalloc(&d_y, (size_t)frames * P * 12));
    const int Fc = c_split;   // (argument: frames per centre stream; 0 = the Ls/Rs streams)
    const long long n_streams_c = Fc > 0 ? (frames + Fc - 1) / Fc : n_streams;
    CK(hipMalloc(&d_seam, (n_streams * 2 + n_streams_c) * (ZS::K - 1) * hop * 4));
    CK(hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_wa, wa.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_ws, ws.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_gain, gain.data(), gain.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tw, tw.data(), tw.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_ramp, ramp.data(), ramp.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(d_c, 0, T * 4)); CK(hipMemset(d_l, 0, T * 4)); CK(hipMemset(d_r, 0, T * 4));
    upx::ZoomArgs a{};
    a.in = (const upx::cf*)d_in; a.out_c = d_c; a.out_l = d_l; a.out_r = d_r; a.w_a = d_wa; a.w_s = d_ws; a.gain = d_gain;
    a.tw = d_tw; a.ramp = d_ramp; a.y = d_y; a.yc = d_y + (size_t)frames * P; a.seam = d_seam;
    a.n = N; a.d = D; a.hop = hop; a.t_in = (int)T; a.t_out = (int)T; a.j_lo = 0; a.j_hi = frames; a.m_lo = 0; a.m_hi = frames - 1;
    a.blocks_per_stream = F; a.n_gain = n_gain; a.gain_stride = N / 2 + 1; a.accumulate = 1; a.f0 = -1;
    a.pair0 = 0; a.pair_end = frames / 2; a.stream0 = 0; a.stream0_c = 0; a.ns_lr = (int)n_streams; a.ns_c = (int)n_streams_c;
    a.blocks_per_stream_c = Fc; a.seam_c = d_seam + n_streams * 2 * (ZS::K - 1) * hop;
    const int lds = ZS::LDS_S_CF * 8, lds_a = ZA::LDS_A_CF * 8;
    int resident = (ZA::WPE_A * 256) / ZA::WG; if (resident > (160 * 1024) / lds_a) resident = (160 * 1024) / lds_a;
    int resident_s = (ZS::WPE_S * 256) / ZS::WG; if (resident_s > (160 * 1024) / lds) resident_s = (160 * 1024) / lds;
    const long long slots = 256LL * resident;
    a.pairs_per_wg = (int)((slots + 7) / 8);
    a.prio_split = prio ? 256 : 0; a.prio_rounds = resident;
    a.prio_split_s = prio > 1 ? 256 : 0; a.prio_rounds_s = resident_s;   // (prio 2: the synthesis rotates as well)
    const void* ka = reinterpret_cast<const void*>(&zana<ZA>);
    const void* ks = reinterpret_cast<const void*>(&zsyn<ZS>);
    CK(hipFuncSetAttribute(ka, hipFuncAttributeMaxDynamicSharedMemorySize, lds_a));
    CK(hipFuncSetAttribute(ks, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    const int n_awg = 8 * a.pairs_per_wg;
    const size_t n_swg = (size_t)(n_streams + n_streams_c) * (D / ZS::RG);
    unsigned long long *d_ta, *d_ts;
    CK(hipMalloc(&d_ta, n_awg * 16)); CK(hipMalloc(&d_ts, n_swg * 16));
    std::vector<unsigned long long> ta(2 * n_awg), ts(2 * n_swg);
    printf("N=%d P=%d D=%d RGa=%d RGs=%d F=%d frames=%d n_gain=%d prio=%d Fc=%d: analysis resident %d/CU, synthesis resident %d/CU (%zu workgroups = %.2f x slots)\n",
           N, P, D, ZA::RG, ZS::RG, F, frames, n_gain, prio, c_split, resident, resident_s, n_swg, n_swg / (256.0 * resident_s));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(zana<ZA>, dim3(n_awg), dim3(ZA::WG), lds_a, 0, a, d_ta);
        CK(hipEventRecord(e1));
        hipLaunchKernelGGL(zsyn<ZS>, dim3((unsigned)n_swg), dim3(ZS::WG), lds, 0, a, d_ts, D / ZS::RG);
        CK(hipEventRecord(e2)); CK(hipEventSynchronize(e2));
        float ma, ms; CK(hipEventElapsedTime(&ma, e0, e1)); CK(hipEventElapsedTime(&ms, e1, e2));
        if (rep < 2) continue;
        CK(hipMemcpy(ta.data(), d_ta, n_awg * 16, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ts.data(), d_ts, n_swg * 16, hipMemcpyDeviceToHost));
        report("analysis", ta, n_awg, ma, 0);
        report("synthesis (C = centre role)", ts, n_swg, ms, (size_t)n_streams * (D / ZS::RG));
    }
    return 0;
}

int main(int argc, char** argv) {
    const int log2n = argc > 1 ? atoi(argv[1]) : 12;
    const int log2p = argc > 2 ? atoi(argv[2]) : 9;
    const int rga = argc > 3 ? atoi(argv[3]) : 8;
    const int rgs = argc > 4 ? atoi(argv[4]) : 8;
    const int F = argc > 5 ? atoi(argv[5]) : 48;
    const int frames = argc > 6 ? atoi(argv[6]) : 28128;
    const int n_gain = argc > 7 ? atoi(argv[7]) : 1;
    const int prio = argc > 8 ? atoi(argv[8]) : 1;
    const int c_split = argc > 9 ? atoi(argv[9]) : 1;
    if (log2p == 9 && rga == 8 && rgs == 8) return run<upx::ZoomCfg<9, 8, 4>, upx::ZoomCfg<9, 8, 4>>(log2n, F, frames, n_gain, prio, c_split);
    if (log2p == 9 && rga == 8 && rgs == 16) return run<upx::ZoomCfg<9, 8, 4>, upx::ZoomCfg<9, 16, 4>>(log2n, F, frames, n_gain, prio, c_split);
    if (log2p == 8 && rga == 16 && rgs == 16) return run<upx::ZoomCfg<8, 16, 4>, upx::ZoomCfg<8, 16, 4>>(log2n, F, frames, n_gain, prio, c_split);
    return 1;
}
