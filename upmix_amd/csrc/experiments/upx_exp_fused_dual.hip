// experiments/upx_exp_fused_dual.hip - EXPERIMENT (round 5): the fused streaming kernel with TWO stream sets per wave.
// A lane hosts V = 2 virtual threads (tid, tid + real workgroup size), each with its own register state and LDS buffer,
// and a phase runs them one after the other inside ONE basic block, so the compiler may interleave the two independent
// instruction streams; one wave per SIMD (launch bounds 1: 512 registers per lane, what does not fit the 256 architectural
// VGPRs is parked in AccVGPRs by the register allocator).  Selected by UPX_DUAL=1 for single-band launches at hop N/4
// (N = 256: general flavour, N = 1024: Live<0, 4>).  -DUPX_DUAL_NO_FENCE: without the scheduling fences inside the phases
// (they stop the scheduler from mixing unrolled iterations - and here the two streams - across them).
#if defined(UPX_DUAL_NO_FENCE)
#define UPX_NO_SCHED_FENCE 1
#endif
#if !defined(UPX_EXPERIMENTS)
#error "experiment kernels: build with -DUPX_EXPERIMENTS (__graft_entry__.build_hip(extra_flags=[\"-DUPX_EXPERIMENTS\"], lib=...)); not part of libupmix_hip.so"
#endif
#include "../upx_kernels.h"

namespace upxk {

// the configuration band_program sees: a workgroup of V x C::WG virtual threads carrying V x C::G streams
template <class C, int V>
struct MultiCfg : C {
    static constexpr int WG = V * C::WG;
    static constexpr int G = V * C::G;
    static constexpr int LDS_CF = G * C::PITCH + C::TW_CF;
};

template <int P, int V>
struct MultiExec {
    upx::ThreadT<P> st[V];
    __device__ __forceinline__ void sync() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    }
    __device__ __forceinline__ void wg_barrier() { sync(); }
    template <class F>
    __device__ __forceinline__ void each(F&& f) {
#pragma unroll
        for (int v = 0; v < V; ++v) f((int)threadIdx.x + v * (int)blockDim.x, st[v]);
        sync();
    }
    template <class F, class G>
    __device__ __forceinline__ void each2(F&& f, G&& g) {
#pragma unroll
        for (int v = 0; v < V; ++v) f((int)threadIdx.x + v * (int)blockDim.x, st[v]);
#pragma unroll
        for (int v = 0; v < V; ++v) g((int)threadIdx.x + v * (int)blockDim.x, st[v]);
        sync();
    }
};

template <class C, int V, class LV>
__global__ __launch_bounds__(C::WG, 1) void upx_band_dual_kernel(upx::BandArgs a) {
    static_assert(C::WAVE_SYNC && C::WG == 64, "one-wave workgroups");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using Ex = MultiExec<C::P, V>;
    Ex ex;
    upx::band_program_auto<MultiCfg<C, V>, Ex, false, LV>(ex, a, reinterpret_cast<upx::cf*>(smem), (int)blockIdx.x);
    if (a.pair_cnt) seam_epilogue<MultiCfg<C, V>>(a, (int)blockIdx.x, (int)gridDim.x);
}

template <class C, int V, class LV>
struct DualEntry {
    using M = MultiCfg<C, V>;
    static constexpr int kLds = M::LDS_CF * (int)sizeof(upx::cf);
    static void launch(const upx::BandArgs& a, int n_wg, hipStream_t st) {
        hipLaunchKernelGGL((upx_band_dual_kernel<C, V, LV>), dim3(n_wg), dim3(C::WG), kLds, st, a);
    }
    static int prepare() {
        return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&upx_band_dual_kernel<C, V, LV>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    }
    static void fill(upx::cf* tw) { upx::fill_tables<C>(tw, turn_trig); }
    static KernelEntry get(const char* name) {
        return KernelEntry{&launch, &prepare, C::WG, M::G, kLds, C::TW_CF, C::LANES, 1, &fill, name, &upx::gain_bin<C>, C::LANES};
    }
};

// (log2 N, live slots S1 or 0 = general) -> dual-stream kernel for a single band at hop N/4
const KernelEntry* find_kernel_dual(int log2n, int live_s1) {
    static const std::map<std::pair<int, int>, KernelEntry> table = [] {
        std::map<std::pair<int, int>, KernelEntry> t;
        t[{8, 0}] = DualEntry<upx::Cfg<8, 4, 16>, 2, upx::LiveAll>::get("upx_band_dual_kernel<upx::Cfg<8, 4, 16>, 2, upx::Live<0, 1048576>>");
        t[{10, 0}] = DualEntry<upx::Cfg<10, 4, 16>, 2, upx::LiveAll>::get("upx_band_dual_kernel<upx::Cfg<10, 4, 16>, 2, upx::Live<0, 1048576>>");
        t[{10, 4}] = DualEntry<upx::Cfg<10, 4, 16>, 2, upx::Live<0, 4>>::get("upx_band_dual_kernel<upx::Cfg<10, 4, 16>, 2, upx::Live<0, 4>>");
        return t;
    }();
    auto it = table.find({log2n, live_s1});
    return it == table.end() ? nullptr : &it->second;
}
}   // namespace upxk
