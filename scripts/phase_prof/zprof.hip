// Profiling aid (not product, not a test): per-phase cycle stamps of one wave of the band-limited ("zoom") kernels
// (upx_zoom.h) on a full grid of synthetic data.  The kernel source runs with an executor that records s_memtime after
// every phase of wave `wave` of workgroup 0.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -w -o scripts/phase_prof/zprof.exe scripts/phase_prof/zprof.hip
//   zprof.exe log2N log2P kernel(0 = analysis, 1 = synthesis Ls/Rs, 2 = synthesis C) period [F] [frames]
// prints the average cycles per stamp position (position = stamp index mod period) over the steady state.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../upmix_amd/csrc/upx_core.h"
#include "../../upmix_amd/csrc/upx_zoom.h"

struct ProfExec {
    upx::ThreadT<16> st;
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
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
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
    }
    template <class F, class G>
    __device__ __forceinline__ void each2(F&& f, G&& g) {
        f((int)threadIdx.x, st);
        stamp();
        g((int)threadIdx.x, st);
        stamp();
        sync();
    }
};

template <class Z>
__global__ __launch_bounds__(Z::WG, Z::WPE_A) void zana(upx::ZoomArgs a, unsigned long long* buf, int cap, int wave) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    ProfExec ex;
    ex.buf = buf; ex.cap = cap;
    ex.rec = blockIdx.x == 0 && (int)(threadIdx.x / 64) == wave;
    upx::zoom_analysis_program<Z>(ex, a, reinterpret_cast<upx::cf*>(smem), (int)blockIdx.x);
}
template <class Z>
__global__ __launch_bounds__(Z::WG, Z::WPE_S) void zsyn(upx::ZoomArgs a, unsigned long long* buf, int cap, int wave, int rec_role) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    ProfExec ex;
    ex.buf = buf; ex.cap = cap;
    // a stream in the middle of the signal (interior flavour); ZPROF_RAW: stamps at entry and exit as well
    ex.rec = blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && (int)blockIdx.z == rec_role && (int)(threadIdx.x / 64) == wave;
    if (a.prio_rounds == -1) ex.stamp();
    upx::zoom_synthesis_program<Z>(ex, a, reinterpret_cast<upx::cf*>(smem), (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
    if (a.prio_rounds == -1) { __builtin_amdgcn_s_waitcnt(0); ex.stamp(); }
}

static void turn_trig(double frac, double& c, double& s) { c = std::cos(2 * M_PI * frac); s = std::sin(2 * M_PI * frac); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <class Z>
int run(int log2n, int kernel, int period, int F, int frames) {
    const int N = 1 << log2n, P = Z::P, D = N / P, hop = N / Z::K;
    const long long n_streams = frames / F;
    frames = (int)(n_streams * F);
    const long long T = (long long)frames * hop;
    std::vector<float> in(2 * T), wa(N), ws(N), gain(N / 2 + 1, 0.f);
    for (auto& v : in) v = (float)rand() / RAND_MAX - 0.5f;
    for (int i = 0; i < N; ++i) { wa[i] = 0.5f - 0.5f * cosf(2 * M_PI * i / N); ws[i] = wa[i] / N; }
    for (int k = P / 16; k < P / 2 - P / 10; ++k) gain[k] = 0.5f;   // like the reference's bands: bins 31..205 of P = 512
    std::vector<upx::cf> tw(Z::TW_CF), ramp(upx::zoom_ramp_count(N, P));
    upx::fill_twiddles<typename Z::Sub>(tw.data(), turn_trig);
    upx::fill_zoom_ramp(ramp.data(), N, P, turn_trig);
    float *d_in, *d_c, *d_l, *d_r, *d_wa, *d_ws, *d_gain, *d_seam; upx::cf *d_tw, *d_ramp, *d_y; unsigned long long* d_buf;
    const int cap = 1 << 16;
    CK(hipMalloc(&d_in, in.size() * 4)); CK(hipMalloc(&d_c, T * 4)); CK(hipMalloc(&d_l, T * 4)); CK(hipMalloc(&d_r, T * 4));
    CK(hipMalloc(&d_wa, N * 4)); CK(hipMalloc(&d_ws, N * 4)); CK(hipMalloc(&d_gain, gain.size() * 4));
    CK(hipMalloc(&d_tw, tw.size() * 8)); CK(hipMalloc(&d_ramp, ramp.size() * 8));
    CK(hipMalloc(&d_y, (size_t)frames * P * 12));
    CK(hipMalloc(&d_seam, n_streams * 3 * (Z::K - 1) * hop * 4)); /* [streams][2][tail] Ls/Rs, then [streams][tail] centre */ CK(hipMalloc(&d_buf, cap * 8));
    CK(hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_wa, wa.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_ws, ws.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_gain, gain.data(), gain.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tw, tw.data(), tw.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(d_ramp, ramp.data(), ramp.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(d_buf, 0, cap * 8));
    upx::ZoomArgs a{};
    a.in = (const upx::cf*)d_in; a.out_c = d_c; a.out_l = d_l; a.out_r = d_r; a.w_a = d_wa; a.w_s = d_ws; a.gain = d_gain;
    a.tw = d_tw; a.ramp = d_ramp; a.y = d_y; a.yc = d_y + (size_t)frames * P; a.seam = d_seam; a.seam_c = d_seam + n_streams * 2 * (Z::K - 1) * hop;
    a.n = N; a.d = D; a.hop = hop; a.t_in = (int)T; a.t_out = (int)T; a.j_lo = 0; a.j_hi = frames; a.m_lo = 0; a.m_hi = frames - 1;
    a.blocks_per_stream = F; a.n_gain = 1; a.gain_stride = N / 2 + 1; a.accumulate = 1; a.f0 = -1;
    a.pair0 = 0; a.pair_end = frames / 2; a.stream0 = 0;
    if (getenv("ZPROF_RAW")) a.prio_rounds = -1;   // (prio_split = 0: unused by the kernels; here: stamp entry and exit)
    const int lds = Z::LDS_S_CF * 8, lds_a = Z::LDS_A_CF * 8;
    int resident = (Z::WPE_A * 256) / Z::WG; if (resident > (160 * 1024) / lds_a) resident = (160 * 1024) / lds_a;
    const long long slots = 256LL * resident;
    a.pairs_per_wg = (int)((slots + 7) / 8);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&zana<Z>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&zsyn<Z>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
    const int n_awg = 8 * a.pairs_per_wg;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(zana<Z>, dim3(n_awg), dim3(Z::WG), lds_a, 0, a, d_buf, kernel == 0 ? cap : 0, 0);
        CK(hipEventRecord(e1));
        hipLaunchKernelGGL(zsyn<Z>, dim3((unsigned)n_streams, D / Z::RG, 2), dim3(Z::WG), lds, 0, a, d_buf, kernel ? cap : 0, 0, kernel == 2 ? 1 : 0);
        CK(hipEventRecord(e2)); CK(hipEventSynchronize(e2));
        float ma, ms; CK(hipEventElapsedTime(&ma, e0, e1)); CK(hipEventElapsedTime(&ms, e1, e2));
        printf("N=%d P=%d D=%d RG=%d F=%d frames=%d lds=%d resident=%d: analysis %.3f ms (%d wgs, %d per xcd)  synthesis %.3f ms (%lld streams)\n",
               N, P, D, Z::RG, F, frames, lds, resident, ma, n_awg, a.pairs_per_wg, ms, n_streams);
    }
    std::vector<unsigned long long> buf(cap);
    CK(hipMemcpy(buf.data(), d_buf, cap * 8, hipMemcpyDeviceToHost));
    int n = 0; while (n < cap && buf[n]) ++n;
    printf("events %d\n", n);
    if (getenv("ZPROF_RAW")) {
        printf("raw stamp deltas (cycles), entry first:");
        for (int k = 1; k < n; ++k) printf("%s%llu", (k - 1) % 16 == 0 ? "\n  " : " ", buf[k] - buf[k - 1]);
        printf("\n  total %llu cycles\n", buf[n - 1] - buf[0]);
    }
    if (n > 3 * period) {
        std::vector<double> acc(period, 0.0); int cnt = 0;
        // skip the first period and the tail; the kernel's prologue stamps shift the phase: print from stamp `off`
        const int off = n % period;
        for (int base = off + period; base + period < n - period; base += period, ++cnt)
            for (int k = 0; k < period; ++k) acc[k] += (double)(buf[base + k] - buf[base + k - 1]);
        double tot = 0;
        for (int k = 0; k < period; ++k) { acc[k] /= cnt; tot += acc[k]; }
        printf("period %d (%d periods averaged), cycles per period %.0f\n", period, cnt, tot);
        for (int k = 0; k < period; ++k) printf("  pos %2d  %8.0f  %5.1f%%\n", k, acc[k], 100 * acc[k] / tot);
    }
    return 0;
}

int main(int argc, char** argv) {
    const int log2n = argc > 1 ? atoi(argv[1]) : 12;
    const int log2p = argc > 2 ? atoi(argv[2]) : 9;
    const int kernel = argc > 3 ? atoi(argv[3]) : 0;
    const int period = argc > 4 ? atoi(argv[4]) : 8;
    const int F = argc > 5 ? atoi(argv[5]) : 24;
    const int frames = argc > 6 ? atoi(argv[6]) : 28128;
    const int d = (1 << log2n) >> log2p;
    const int rg = d >= 16 ? 16 : d;
#define CASE(LP, RG) if (log2p == LP && rg == RG) return run<upx::ZoomCfg<LP, RG, 4>>(log2n, kernel, period, F, frames);
    CASE(8, 8) CASE(8, 16) CASE(9, 4) CASE(9, 8) CASE(9, 16) CASE(10, 8)
    return 1;
}
