// upx_kernels.h - the gfx950 kernels of libupmix_hip.so as templates, and the tables that map a plan's geometry to
// an instantiation.  The instantiations are spread over several translation units (upx_reg_*.hip) so that they
// compile side by side; upx_lib.hip (host logic, C ABI) sees only the tables' look-up functions.
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <map>
#include <tuple>

#include "upx_core.h"
#include "upx_big.h"
#include "upx_zoom.h"

namespace upxk {

// `each` = the calling thread runs the phase, then synchronises with the other lanes of its
// stream: a workgroup barrier, or - when a stream lives inside one wave - only a
// wavefront-scope fence (LDS operations of one wave execute in order).
template <bool WAVE_SYNC, int P>
struct DevExec {
    upx::ThreadT<P> st;
    // Threads only communicate through LDS, so the fences order LDS accesses only ("local"): global loads and
    // stores (audio, windows, gains, output planes) may stay in flight across a phase boundary.
    __device__ __forceinline__ void sync() {
#if defined(UPX_FULL_FENCES)
        if constexpr (WAVE_SYNC) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        } else {
            __syncthreads();
        }
#else
        if constexpr (WAVE_SYNC) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        }
#endif
    }
    // Wide streams run their phases with wave-level ordering (WAVE_SYNC) and meet here.
    __device__ __forceinline__ void wg_barrier() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    }
    template <class F>
    __device__ __forceinline__ void each(F&& f) {
        f((int)threadIdx.x, st);
        sync();
    }
    // f reads the stream's LDS buffer, g scatters into it.  Across waves that needs a barrier in between;
    // inside one wave the LDS executes the wave's operations in order, so f and g run back to back.
    template <class F, class G>
    __device__ __forceinline__ void each2(F&& f, G&& g) {
        f((int)threadIdx.x, st);
        if constexpr (!WAVE_SYNC) sync();
        g((int)threadIdx.x, st);
        sync();
    }
};

#if defined(UPX_EXPERIMENTS)
// Stream seams inside the launch (BandArgs::pair_cnt).  Runs after band_program: every stream of the workgroup has written
// its hops and its tail.  Seams between two streams of ONE workgroup are added right away; for the seam to either
// neighbour workgroup, the two workgroups count up pair_cnt[w] and the second arriver adds.  Memory: the counter update is
// an agent-scope acquire-release atomic behind a workgroup barrier (the workgroup's stores are then in the L2 of its XCD; the
// release writes that L2 back, the acquire invalidates the reader's), which is what makes the other workgroup's tail and
// head visible across the chip's eight L2s.
template <class C>
__device__ __forceinline__ void seam_epilogue(const upx::BandArgs& a, int wg, int n_wg) {
    constexpr int TAIL = (C::K - 1) * C::HOP;
    __shared__ int second[2];
    auto add = [&](int sid) {   // tail of stream sid onto the first blocks of stream sid + 1
        for (int i = (int)threadIdx.x; i < TAIL; i += (int)blockDim.x)
            upx::stream_seam_add(a, a.n_streams, TAIL, C::HOP, (long long)sid * TAIL + i);
    };
    __syncthreads();
#pragma unroll 1
    for (int g = 0; g + 1 < C::G; ++g) add(wg * C::G + g);
    if (threadIdx.x == 0) {
        second[0] = wg > 0 ? (__hip_atomic_fetch_add(a.pair_cnt + wg - 1, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) & 1) : 0;
        second[1] = wg + 1 < n_wg ? (__hip_atomic_fetch_add(a.pair_cnt + wg, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) & 1) : 0;
    }
    __syncthreads();
    const bool s0 = second[0] != 0, s1 = second[1] != 0;
    if (s0 || s1) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (s0) add(wg * C::G - 1);
        if (s1) add(wg * C::G + C::G - 1);
    }
}
#endif

// WPE = waves per SIMD the register allocator must leave room for (2 -> 256 VGPRs, 3 -> 168).
// MERGED = false: the launch carries one band (one gain slot per bin): the flavour single bands get.
// LV = upx::Live<S0, S1>: single-band flavour specialised for the own-bin slots that carry gain (upx_core.h).
template <class C, int WPE, bool MERGED = true, class LV = upx::LiveAll>
__global__ __launch_bounds__(C::WG, WPE) void upx_band_kernel(upx::BandArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using Ex = DevExec<C::WAVE_SYNC || C::WIDE, C::P>;
    Ex ex;
    // the interior flavour (two more instantiations of the program) for the kernels plans select; the 8-points-per-lane
    // and plain-schedule alternates of experiment builds (csrc/experiments/) keep the one general body
    if constexpr (C::P == 16 && (C::WIDE || C::LOG2N <= 11))
        upx::band_program_auto<C, Ex, MERGED, LV>(ex, a, reinterpret_cast<upx::cf*>(smem), (int)blockIdx.x);
    else
        upx::band_program<C, Ex, MERGED>(ex, a, reinterpret_cast<upx::cf*>(smem), (int)blockIdx.x);
#if defined(UPX_EXPERIMENTS)
    if (a.pair_cnt) seam_epilogue<C>(a, (int)blockIdx.x, (int)gridDim.x);
#endif
}

// ---- unfused path: STFT sizes 16384..65536 and arbitrary hops (upx_big.h) ---------------------
template <class B>
__global__ __launch_bounds__(256) void upx_big_step1_audio_kernel(upx::BigArgs a) {
    upx::big_step1_audio<B>(a, (long long)blockIdx.x * 256 + threadIdx.x);
}
template <class B>
__global__ __launch_bounds__(256) void upx_big_step2_inv_kernel(upx::cf* buf, const upx::cf* tw_n, int frames) {
    upx::big_step2_inv<B>(buf, tw_n, frames, (long long)blockIdx.x * 256 + threadIdx.x);
}
template <class B>
__global__ __launch_bounds__(B::Row::WG) void upx_big_rows_kernel(upx::cf* buf, const upx::cf* tw_rows, int n_rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevExec<B::Row::WAVE_SYNC, B::Row::P> ex;
    upx::big_rows_program<B>(ex, buf, tw_rows, reinterpret_cast<upx::cf*>(smem), (int)blockIdx.x, n_rows);
}
template <class B>
__global__ __launch_bounds__(B::Row::WG) void upx_big_frame_kernel(upx::BigArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevExec<B::Row::WAVE_SYNC, B::Row::P> ex;
    upx::big_frame_program<B>(ex, a, reinterpret_cast<upx::cf*>(smem), (int)blockIdx.x);
}
// rows -> mask -> rows (N1 == 16) of one frame pair: ROWS = 2, one mirror pair of rows per workgroup (two row
// streams); ROWS = 16 (N = 16 384), the whole frame per workgroup with steps 1 and 2 in registers
template <class B, int ROWS>
__global__ __launch_bounds__(ROWS * B::Row::LANES) void upx_big_mid_kernel(upx::BigArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if constexpr (B::N1 == 16 && (ROWS == 2 || ROWS * B::Row::LANES == B::N2)) {
        DevExec<false, B::Row::P> ex;
        upx::big_mid_program<B, ROWS>(ex, a, reinterpret_cast<upx::cf*>(smem), (int)blockIdx.x);
    }
}
template <class B>
__global__ __launch_bounds__(256) void upx_big_mask_kernel(upx::BigArgs a) {
    upx::big_mask<B>(a, (long long)blockIdx.x * 256 + threadIdx.x);
}
template <class B>
__global__ __launch_bounds__(256) void upx_big_ola_kernel(upx::BigArgs a) {
    upx::big_ola<B>(a, (long long)blockIdx.x * 256 + threadIdx.x);
}
// step 2 + overlap-add in one pass (hop = N/K, K = 2, 4, 8; frames of sixteen rows that pass through the scratch)
template <class B, int K>
__global__ __launch_bounds__(256, 2) void upx_big_tail_kernel(upx::BigArgs a, int rb) {
    if constexpr (B::N1 == 16) upx::big_tail<B, K>(a, rb, (long long)blockIdx.x * 256 + threadIdx.x);
}

// ---- band-limited bands: pruned analysis / residue-stream synthesis (upx_zoom.h) --------------------------
// Z::WPE_A / WPE_S = waves per SIMD the register allocator leaves room for (4 -> 128 VGPRs, 3 -> 168): what the LDS
// footprint of the configuration admits.
template <class Z>
__global__ __launch_bounds__(Z::WG, Z::WPE_A) void upx_zoom_analysis_kernel(upx::ZoomArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevExec<true, 16> ex;
    upx::zoom_analysis_program<Z>(ex, a, reinterpret_cast<upx::cf*>(smem), (int)blockIdx.x);
}
// grid = (ns_lr + ns_c) x residue groups workgroups: the Ls/Rs streams first (stream fastest, then residue group), the
// centre streams - half the work each, or less (ZoomArgs::blocks_per_stream_c) - behind them: longest first
template <class Z>
__global__ __launch_bounds__(Z::WG, Z::WPE_S) void upx_zoom_synthesis_kernel(upx::ZoomArgs a, int n_groups) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DevExec<true, 16> ex;
    int b = (int)blockIdx.x;
    const int n0 = a.ns_lr * n_groups;
    if (b < n0) {
        upx::zoom_synthesis_program<Z>(ex, a, reinterpret_cast<upx::cf*>(smem), b % a.ns_lr, b / a.ns_lr, 0);
    } else {
        b -= n0;
        upx::zoom_synthesis_program<Z>(ex, a, reinterpret_cast<upx::cf*>(smem), b % a.ns_c, b / a.ns_c, 1);
    }
}

struct ZoomEntry {
    void (*analysis)(const upx::ZoomArgs&, int n_wg, hipStream_t);
    void (*synthesis)(const upx::ZoomArgs&, int n_groups, hipStream_t);   // streams: ZoomArgs::ns_lr, ns_c
    int (*prepare)();
    void (*fill_tw)(upx::cf*);
    int p, rg, k, wg, lds_bytes, tw_cf, wpe;   // lds_bytes / wpe: synthesis
    int lds_bytes_a, wpe_a;                    // analysis
    const char* name_analysis;
    const char* name_synthesis;
};

struct BigEntry {
    int (*gain_bin)(int);   // order of the per-bin gain rows as the kernels read them
    int n, n1, row_wg, row_lds, row_tw_cf;
    void (*chunk)(const upx::BigArgs&, hipStream_t);
    int (*prepare)();
    void (*fill_tw_n)(upx::cf*);
    void (*fill_tw_rows)(upx::cf*);
};

struct KernelEntry {
    void (*launch)(const upx::BandArgs&, int n_wg, hipStream_t);
    int (*prepare)();
    int wg, g, lds_bytes, tw_cf, lanes, wpe;
    void (*fill_tw)(upx::cf*);
    const char* name;
    int (*gain_bin)(int);   // order of the per-bin gain rows as the kernel reads them
    int layout;             // distinguishes twiddle table layouts of one STFT size
};

inline void turn_trig(double frac, double& c, double& s) {
    const double a = 2.0 * M_PI * frac;
    c = std::cos(a);
    s = std::sin(a);
}

template <class B>
struct BigImpl {
    using Row = typename B::Row;
    static constexpr int kRowLds = Row::LDS_CF * (int)sizeof(upx::cf);
    // N = 16 384: the sixteen 1024-point rows of a frame fit one workgroup's LDS (149 KB): whole-frame variant
    static constexpr int kMidRows = (B::N1 == 16 && 16 * Row::LANES == B::N2 && (16 * Row::PITCH + Row::TW_CF) * 8 <= 160 * 1024) ? 16 : 2;
    static constexpr int kMidLds = (kMidRows * Row::PITCH + Row::TW_CF) * (int)sizeof(upx::cf);
    static constexpr int kTailBlocks = 8;     // emitted blocks per thread of upx_big_tail_kernel
    static unsigned blocks(long long n) { return (unsigned)((n + 255) / 256); }
    static unsigned row_wgs(int rows) { return (unsigned)((rows + Row::G - 1) / Row::G); }
    static void rows(upx::cf* buf, const upx::cf* tw, int n_rows, hipStream_t st) {
        hipLaunchKernelGGL(upx_big_rows_kernel<B>, dim3(row_wgs(n_rows)), dim3(Row::WG), kRowLds, st, buf, tw, n_rows);
    }
    // all launches of one chunk, in stream order
    static void chunk(const upx::BigArgs& a, hipStream_t st) {
        const int ch = a.ch;
        // y (ch frames) and yc (ch/2 frames) are adjacent in the scratch: one launch covers both
        const int inv_frames = ch + ch / 2;
        if (B::N1 == 16 && kMidRows == 16) {
            // whole frame per workgroup: audio in, time-domain y / yc out
            hipLaunchKernelGGL((upx_big_mid_kernel<B, kMidRows>), dim3((unsigned)(ch / 2)), dim3(kMidRows * Row::LANES), kMidLds, st, a);
        } else if (B::N1 == 16) {
            hipLaunchKernelGGL(upx_big_step1_audio_kernel<B>, dim3(blocks((long long)ch * B::N2)), dim3(256), 0, st, a);
            // row transforms, mask and inverse row transforms fused: one workgroup per (frame pair, mirror pair of rows)
            hipLaunchKernelGGL((upx_big_mid_kernel<B, kMidRows>), dim3((unsigned)(ch / 2) * 8), dim3(kMidRows * Row::LANES), kMidLds, st, a);
        } else {
            hipLaunchKernelGGL(upx_big_frame_kernel<B>, dim3(row_wgs(ch)), dim3(Row::WG), kRowLds, st, a);
            hipLaunchKernelGGL(upx_big_mask_kernel<B>, dim3(blocks(upx::big_mask_threads<B>(ch / 2))), dim3(256), 0, st, a);
            rows(a.y, a.tw_rows, inv_frames * B::N1, st);
        }
        if constexpr (B::N1 == 16 && kMidRows == 2) {
            if (a.tail && a.hop * a.kf == B::N && (a.kf == 2 || a.kf == 4 || a.kf == 8)) {
                // ranges of kTailBlocks emitted blocks: 2^24 / N frames per chunk make ~32 ranges x N2 columns (>= 8 waves
                // per CU); the K - 1 frames in front of a range are read by two ranges
                const int rb = a.tail > 1 ? a.tail : kTailBlocks;    // (UPX_BIG_TAIL = n > 1: n blocks per range, for sweeps)
                const long long threads = (long long)((a.m1 - a.m0 + rb - 1) / rb) * B::N2;
                if (a.kf == 2) hipLaunchKernelGGL((upx_big_tail_kernel<B, 2>), dim3(blocks(threads)), dim3(256), 0, st, a, rb);
                else if (a.kf == 4) hipLaunchKernelGGL((upx_big_tail_kernel<B, 4>), dim3(blocks(threads)), dim3(256), 0, st, a, rb);
                else hipLaunchKernelGGL((upx_big_tail_kernel<B, 8>), dim3(blocks(threads)), dim3(256), 0, st, a, rb);
                return;
            }
        }
        if (B::N1 == 16 && kMidRows == 2)
            hipLaunchKernelGGL(upx_big_step2_inv_kernel<B>, dim3(blocks((long long)inv_frames * B::N2)), dim3(256), 0, st, a.y, a.tw_n, inv_frames);
        hipLaunchKernelGGL(upx_big_ola_kernel<B>, dim3(blocks((long long)(a.m1 - a.m0) * a.hop)), dim3(256), 0, st, a);
    }
    static int prepare() {
        int e = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&upx_big_rows_kernel<B>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, kRowLds);
        if (!e && B::N1 == 1)
            e = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&upx_big_frame_kernel<B>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, kRowLds);
        if (!e && B::N1 == 16)
            e = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&upx_big_mid_kernel<B, kMidRows>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, kMidLds);
        return e;
    }
    static void fill_n(upx::cf* tw) { if (B::N1 == 16) upx::fill_big_twiddles<B>(tw, turn_trig); }
    static void fill_rows(upx::cf* tw) { upx::fill_twiddles<Row>(tw, turn_trig); }
    static BigEntry get() {
        return BigEntry{&upx::big_gain_bin<B>, B::N, B::N1, Row::WG, kRowLds, Row::TW_CF, &chunk, &prepare, &fill_n, &fill_rows};
    }
};

template <class Z>
struct ZoomImpl {
    static constexpr int kLdsA = Z::LDS_A_CF * (int)sizeof(upx::cf), kLdsS = Z::LDS_S_CF * (int)sizeof(upx::cf);
    static void analysis(const upx::ZoomArgs& a, int n_wg, hipStream_t st) {
        hipLaunchKernelGGL((upx_zoom_analysis_kernel<Z>), dim3(n_wg), dim3(Z::WG), kLdsA, st, a);
    }
    static void synthesis(const upx::ZoomArgs& a, int n_groups, hipStream_t st) {
        hipLaunchKernelGGL((upx_zoom_synthesis_kernel<Z>), dim3((a.ns_lr + a.ns_c) * n_groups), dim3(Z::WG), kLdsS, st, a, n_groups);
    }
    static int prepare() {
        int e = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&upx_zoom_analysis_kernel<Z>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, kLdsA);
        if (!e)
            e = (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&upx_zoom_synthesis_kernel<Z>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, kLdsS);
        return e;
    }
    static void fill(upx::cf* tw) { upx::fill_twiddles<typename Z::Sub>(tw, turn_trig); }
    static ZoomEntry get(const char* na, const char* ns) {
        return ZoomEntry{&analysis, &synthesis, &prepare, &fill, Z::P, Z::RG, Z::K, Z::WG, kLdsS, Z::TW_CF, Z::WPE_S,
                         kLdsA, Z::WPE_A, na, ns};
    }
};

template <class C, int WPE, bool MERGED = true, class LV = upx::LiveAll>
struct Entry {
    static constexpr int kLds = C::LDS_CF * (int)sizeof(upx::cf);
    static void launch(const upx::BandArgs& a, int n_wg, hipStream_t st) {
        hipLaunchKernelGGL((upx_band_kernel<C, WPE, MERGED, LV>), dim3(n_wg), dim3(C::WG), kLds, st, a);
    }
    static int prepare() {
        return (int)hipFuncSetAttribute(reinterpret_cast<const void*>(&upx_band_kernel<C, WPE, MERGED, LV>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    }
    static void fill(upx::cf* tw) { upx::fill_tables<C>(tw, turn_trig); }
    static KernelEntry get(const char* name) {
        return KernelEntry{&launch, &prepare, C::WG, C::G, kLds, C::TW_CF, C::LANES, WPE, &fill, name,
                           &upx::gain_bin<C>, C::WIDE ? 1000 : C::LANES};
    }
};

// ---- look-up functions, one translation unit each (upx_reg_*.hip) ----------------------------------------------------
// (log2 N, K = N / hop, variant) -> fused streaming kernel
//   variant 0: 16 points per lane, register budget for 2 waves/SIMD (256 VGPRs); N = 4096 / 8192 as wide streams
//   variant 10: variant 0 for launches that carry a single band; 100 + 10 S0 + S1: ... specialised for the live
//               own-bin slots [S0, S1) (upx::Live)
const KernelEntry* find_kernel_default(int log2n, int k, int variant);   // variants 0, 10, 100+
#if defined(UPX_EXPERIMENTS)
// experiment builds only (csrc/experiments/upx_exp_*.hip; none of them is in libupmix_hip.so):
//   variant 1:  8 points per lane, register budget for 4 waves/SIMD (128 VGPRs)
//   variant 2: 16 points per lane, plain Stockham schedule for every size
const KernelEntry* find_kernel_dual(int log2n, int live_s1);             // two stream sets per wave
const KernelEntry* find_kernel_p8(int log2n, int k);                     // variant 1
const KernelEntry* find_kernel_plain(int log2n, int k);                  // variant 2
#endif
// (log2 P, residues per workgroup, K) -> kernels of the band-limited path
const ZoomEntry* find_zoom_p256(int rg, int k);
const ZoomEntry* find_zoom_p512(int rg, int k);
const ZoomEntry* find_zoom_p1024(int rg, int k);
// log2 N -> unfused pipeline
const BigEntry* find_big(int log2n);

}   // namespace upxk
