// upx_lib.hip - libupmix_hip.so: C ABI (include/upmix_hip.h) + gfx950 kernels.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared (see __graft_entry__.build()).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types only; the library is dlopen'ed on first use of upx_comm_*

#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <map>
#include <mutex>
#include <thread>
#include <cstdlib>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/upmix_hip.h"
#include "upx_kernels.h"
#include "upx_pipeline.h"

using namespace upxk;

// ---------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------
namespace {
thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(e_ == hipErrorOutOfMemory ? UPX_ERR_NOMEM : UPX_ERR_HIP, "%s: %s", #expr,   \
                        hipGetErrorString(e_));                                                    \
    } while (0)

// ---------------------------------------------------------------------------
// small kernels (the transform kernels and their tables: upx_kernels.h, upx_reg_*.hip)
// ---------------------------------------------------------------------------
__global__ void upx_stream_seam_add_kernel(upx::BandArgs a, int n_streams, int tail, int hop) {
    const long long total = (long long)n_streams * tail;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x)
        upx::stream_seam_add(a, n_streams, tail, hop, g);
}

__global__ void upx_zoom_seam_add_kernel(upx::ZoomArgs a, int n_lr, int n_c, int tail) {
    const long long total = ((long long)n_lr + n_c) * tail;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x)
        upx::zoom_seam_add(a, n_lr, n_c, tail, g);
}

// The same two passes with 16-byte accesses: one grid row per stream (blockIdx.y; no 64-bit division per element), four
// consecutive samples per thread.  The host takes these when hop % 4 == 0 and the planes and the seam buffer are 16-byte
// aligned (a tail row is then aligned as well: tail = (K-1) hop); results are those of the scalar passes, sample for sample.
__device__ __forceinline__ void upx_add4(float* plane, long long n, const float* row, int i, long long t_out) {
    if (n + 3 < t_out) {
        float4 v = *reinterpret_cast<float4*>(plane + n);
        const float4 s = *reinterpret_cast<const float4*>(row + i);
        v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w;
        *reinterpret_cast<float4*>(plane + n) = v;
    } else {
        for (int q = 0; q < 4 && n + q < t_out; ++q) plane[n + q] += row[i + q];
    }
}
__global__ void upx_stream_seam_add4_kernel(upx::BandArgs a, int n_streams, int tail, int hop) {
    const int sid = blockIdx.y;
    if (sid >= n_streams - 1) return;                       // the last stream's tail lies beyond the emitted range
    const long long m = upx::stream_first(a, sid + 1);
    if (m >= a.m_hi) return;
    const float* row = a.seam + (size_t)sid * 3 * tail;
    for (int i = 4 * (blockIdx.x * blockDim.x + threadIdx.x); i < tail; i += 4 * gridDim.x * blockDim.x) {
        const long long n = m * hop + i;
        if (n < 0 || n >= a.t_out || m + i / hop >= a.m_hi) continue;   // (hop % 4 == 0: the four samples share a block)
        upx_add4(a.out_c, n, row, i, a.t_out);
        upx_add4(a.out_l, n, row + tail, i, a.t_out);
        upx_add4(a.out_r, n, row + 2 * (size_t)tail, i, a.t_out);
    }
}
__global__ void upx_zoom_seam_add4_kernel(upx::ZoomArgs a, int n_lr, int n_c, int tail) {
    const int role = (int)blockIdx.y >= n_lr ? 1 : 0;
    const int sid = role ? (int)blockIdx.y - n_lr : (int)blockIdx.y;
    if (sid >= (role ? n_c : n_lr) - 1) return;
    const long long m = upx::zoom_stream_first(a, role, sid + 1);
    if (m >= a.m_hi) return;
    for (int i = 4 * (blockIdx.x * blockDim.x + threadIdx.x); i < tail; i += 4 * gridDim.x * blockDim.x) {
        const long long n = m * a.hop + i;
        if (n < 0 || n >= a.t_out || m + i / a.hop >= a.m_hi) continue;
        if (role) {
            upx_add4(a.out_c, n, a.seam_c + (size_t)sid * tail, i, a.t_out);
        } else {
            const float* row = a.seam + (size_t)sid * 2 * tail;
            upx_add4(a.out_l, n, row, i, a.t_out);
            upx_add4(a.out_r, n, row + tail, i, a.t_out);
        }
    }
}

// max |x| as a bit pattern: non-negative floats order like their bit patterns, and a NaN (sign cleared) lies above
// every number, so a NaN anywhere gives NaN - what np.max(np.abs(.)) gives main.py:53, :85-88.
// One atomic per WORKGROUP (round 3 issued one per wave: 8192 atomics on one address serialise in the L2 - 100 us for a
// pass that reads 64 MB - which the chunked WAV pipeline of round 4 pays twelve times per file), 16-byte loads.
__device__ __forceinline__ void upx_absmax_body(const float* x, long long n, unsigned int* result);
__global__ void upx_absmax_kernel(const float* x, long long n, unsigned int* result) { upx_absmax_body(x, n, result); }
// the three planes of a chunk in one launch (blockIdx.y = plane; results in result[0..2])
__global__ void upx_absmax3_kernel(const float* c, const float* l, const float* r, long long n, unsigned int* result) {
    upx_absmax_body(blockIdx.y == 0 ? c : (blockIdx.y == 1 ? l : r), n, result + blockIdx.y);
}
__device__ __forceinline__ void upx_absmax_body(const float* x, long long n, unsigned int* result) {
    __shared__ unsigned int part[4];
    unsigned int m = 0u;
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
    // head up to the first 16-byte boundary, body in float4, tail
    long long head = (long long)(((16u - (unsigned)((size_t)x & 15u)) & 15u) / 4u);
    if (head > n) head = n;
    const long long n4 = (n - head) / 4;
    const uint4* x4 = reinterpret_cast<const uint4*>(x + head);
    for (long long i = tid; i < n4; i += stride) {
        const uint4 v = x4[i];
        const unsigned int a = v.x & 0x7fffffffu, b = v.y & 0x7fffffffu, c = v.z & 0x7fffffffu, d = v.w & 0x7fffffffu;
        const unsigned int ab = a > b ? a : b, cd = c > d ? c : d, q = ab > cd ? ab : cd;
        m = q > m ? q : m;
    }
    for (long long i = tid; i < head + (n - head - 4 * n4); i += stride) {
        const long long j = i < head ? i : head + 4 * n4 + (i - head);
        const unsigned int b = __float_as_uint(x[j]) & 0x7fffffffu;
        m = b > m ? b : m;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned int q = (unsigned int)__shfl_xor((int)m, o);
        m = q > m ? q : m;
    }
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = part[w] > m ? part[w] : m;
        if (m) atomicMax(result, m);
    }
}

// max |x| of 32-bit PCM as an integer (|INT_MIN| = 2^31 fits the unsigned word).  main.py:53 takes the input peak over the
// float64 samples soundfile.read returns (x / 2^31): for 16- and 24-bit files the float32 copy the kernels transform holds
// those exactly and upx_absmax_kernel on it is the same number; for 32-bit files it does not - the peak is taken on the
// integers and divided on the host, so that the one global scale of main.py:85-97 is the reference's to the last bit.
__global__ void upx_absmax_i32_kernel(const int* x, long long n, unsigned int* result) {
    __shared__ unsigned int part[4];
    unsigned int m = 0u;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int v = x[i];
        const unsigned int a = v < 0 ? 0u - (unsigned int)v : (unsigned int)v;
        m = a > m ? a : m;
    }
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned int q = (unsigned int)__shfl_xor((int)m, o);
        m = q > m ? q : m;
    }
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = part[w] > m ? part[w] : m;
        if (m) atomicMax(result, m);
    }
}

__global__ void upx_scale_kernel(float* x, long long n, float s) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        x[i] *= s;
}

// upx_process_lr: a work item's samples as the caller holds them -> the interleaved float32 [n][2] the kernels read.
// raw = interleaved [n][2] (planar == 0) or left[n] | right[n] (planar == 1) of float32 / float64 (f64 != 0).  The cast is
// the one the host entry used to make (np.asarray(x, float32): round to nearest even, infinity beyond the float32 range).
__global__ void upx_samples_to_stereo_kernel(const void* raw, int f64, int planar, long long n, float2* stereo) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float l, r;
        if (f64) {
            const double* d = static_cast<const double*>(raw);
            if (planar) { l = (float)d[i]; r = (float)d[n + i]; }
            else { const double2 v = reinterpret_cast<const double2*>(d)[i]; l = (float)v.x; r = (float)v.y; }
        } else {
            const float* f = static_cast<const float*>(raw);
            if (planar) { l = f[i]; r = f[n + i]; }
            else { const float2 v = reinterpret_cast<const float2*>(f)[i]; l = v.x; r = v.y; }
        }
        stereo[i] = make_float2(l, r);
    }
}

// ---- device-side WAV codec + export layouts (main.py:43-55, 85-97, 110-157) -------------------
// what soundfile.read returns for sample i (float64, int / 2^(bits-1))
__device__ __forceinline__ double upx_decode_sample_f64(const unsigned char* p, long long i, int fmt) {
    if (fmt == UPX_PCM16) return (double)reinterpret_cast<const short*>(p)[i] * (1.0 / 32768.0);
    if (fmt == UPX_PCM32) return (double)reinterpret_cast<const int*>(p)[i] * (1.0 / 2147483648.0);
    if (fmt == UPX_F32) return (double)reinterpret_cast<const float*>(p)[i];
    const unsigned char* b = p + 3 * i;   // 24-bit little endian, sign extended
    int v = (int)b[0] | ((int)b[1] << 8) | ((int)(signed char)b[2] << 16);
    return (double)v * (1.0 / 8388608.0);
}
// the float32 signal the kernels transform (center_extraction.py hands float64 to a float64 FFT; here the
// cast happens once, at decode)
__device__ __forceinline__ float upx_decode_sample(const unsigned char* p, long long i, int fmt) {
    return (float)upx_decode_sample_f64(p, i, fmt);
}
__global__ void upx_decode_kernel(const unsigned char* pcm, int fmt, int channels, long long n, float* stereo) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float l = upx_decode_sample(pcm, channels == 2 ? 2 * i : i, fmt);
        const float r = channels == 2 ? upx_decode_sample(pcm, 2 * i + 1, fmt) : l;
        stereo[2 * i] = l;
        stereo[2 * i + 1] = r;
    }
}
// x is a double because main.py hands float64 arrays to the writer whenever one column is float64 (AB: L + R of
// the decoded file, main.py:112-114); float32 columns arrive here exactly (float -> double is lossless)
__device__ __forceinline__ void upx_encode_sample(unsigned char* p, long long i, int fmt, double x) {
    if (fmt == UPX_F32) { reinterpret_cast<float*>(p)[i] = (float)x; return; }
    const int bits = fmt;
    const double full = (double)((1ll << (bits - 1)) - 1);
    double q = rint(x * full);
    q = q > full ? full : (q < -full - 1.0 ? -full - 1.0 : q);
    const long long v = (long long)q;
    if (fmt == UPX_PCM16) reinterpret_cast<short*>(p)[i] = (short)v;
    else if (fmt == UPX_PCM32) reinterpret_cast<int*>(p)[i] = (int)v;
    else { unsigned char* b = p + 3 * i; b[0] = (unsigned char)(v & 0xff); b[1] = (unsigned char)((v >> 8) & 0xff); b[2] = (unsigned char)((v >> 16) & 0xff); }
}
// planes are scaled like `final_x *= scale_factor` (float32 array times a float64 scalar: product in double,
// rounded to float32), then combined in float32 exactly as main.py does
__global__ void upx_export_kernel(const float* c, const float* l, const float* r, const unsigned char* pcm, int in_fmt,
                                  int channels, long long n, double scale, int mode, int fmt, unsigned char* o0,
                                  unsigned char* o1, unsigned char* o2) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float fc = (float)((double)c[i] * scale), fl = (float)((double)l[i] * scale), fr = (float)((double)r[i] * scale);
        if (mode == UPX_EXPORT_STEREO_SUM) {
            upx_encode_sample(o0, 2 * i, fmt, fl + 0.5f * fc);
            upx_encode_sample(o0, 2 * i + 1, fmt, fr + 0.5f * fc);
        } else if (mode == UPX_EXPORT_SPLIT) {
            upx_encode_sample(o0, 2 * i, fmt, fl); upx_encode_sample(o0, 2 * i + 1, fmt, 0.f);
            upx_encode_sample(o1, 2 * i, fmt, fc); upx_encode_sample(o1, 2 * i + 1, fmt, fc);
            upx_encode_sample(o2, 2 * i, fmt, 0.f); upx_encode_sample(o2, 2 * i + 1, fmt, fr);
        } else {
            // main.py:112-113: upmix_sum in float32, orig_sum = L + R in float64 (the decoded file)
            upx_encode_sample(o0, 2 * i, fmt, (fl + fc) + fr);
            const double dl = upx_decode_sample_f64(pcm, channels == 2 ? 2 * i : i, in_fmt);
            const double dr = channels == 2 ? upx_decode_sample_f64(pcm, 2 * i + 1, in_fmt) : dl;
            upx_encode_sample(o0, 2 * i + 1, fmt, dl + dr);
        }
    }
}

// seam[row][plane][spill] <- planes[own_len + i]; other rows zero (done by memset)
// (`count` <= spill samples exist behind own_len; the row pitch stays `spill`)
__global__ void upx_seam_pack_kernel(float* seam_row, const float* c, const float* l, const float* r,
                                     long long own_len, long long spill, long long count) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x) {
        seam_row[i] = c[own_len + i];
        seam_row[spill + i] = l[own_len + i];
        seam_row[2 * spill + i] = r[own_len + i];
    }
}
// head of the next shard += spill of the previous one
__global__ void upx_seam_add_kernel(float* c, float* l, float* r, const float* pc, const float* pl, const float* pr,
                                    long long spill) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < spill; i += (long long)gridDim.x * blockDim.x) {
        c[i] += pc[i];
        l[i] += pl[i];
        r[i] += pr[i];
    }
}

// streaming overlap-add of process_stereo_chunk (center_extraction.py:392-407) on a ring that stays on the device:
// ring[(pos + i) % n] += rec[i]; the first `hop` samples behind `pos` are emitted and cleared (the reference's shift by
// `hop` and zero tail is the advance of `pos`).  One thread per (plane, i): the same float32 additions in the same order.
__global__ void upx_chunk_ola_kernel(const float* rec_c, const float* rec_l, const float* rec_r, float* ring, int n, int hop,
                                     int pos, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int at = (pos + i) % n;
    const float* rec[3] = {rec_c, rec_l, rec_r};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float v = ring[(size_t)k * n + at] + rec[k][i];
        if (i < hop) {
            out[(size_t)k * hop + i] = v;
            ring[(size_t)k * n + at] = 0.f;
        } else {
            ring[(size_t)k * n + at] = v;
        }
    }
}
// ring -> natural order (flush_final, :411-424), optionally clearing it
__global__ void upx_chunk_unroll_kernel(float* ring, int n, int pos, float* out, int clear) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int at = (pos + i) % n;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        out[(size_t)k * n + i] = ring[(size_t)k * n + at];
        if (clear) ring[(size_t)k * n + at] = 0.f;
    }
}

constexpr int kMidEvents = 15;           // up to 8 launch pairs of the band-limited path are timed per phase
constexpr int kTimingSlots = 64;          // recent upx_process_device calls whose per-band events are kept
constexpr int kMaxFramesPerSample = 64;   // unfused path: ceil(N / hop) frames overlap one sample

// Tuning and test knobs.  The library reads UPX_* variables at plan creation ONLY when the process opts in with
// UPX_TUNING=1 (the test suite, scripts/, the A/B harness); without it the environment cannot change which kernels a plan
// selects, how a launch is cut into streams or where chunk seams fall - a caller's results do not depend on what a shell
// happened to export (round-5 review).  Operational settings stay unconditional: UPX_COMM_TIMEOUT / UPX_RDZV_TIMEOUT
// (upx_comm_create), and on the Python side UPX_PINNED_POOL_MB and the UPX_RDZV_* rendezvous variables.
const char* knob(const char* name) {
    const char* on = std::getenv("UPX_TUNING");
    if (!on || std::strcmp(on, "1") != 0) return nullptr;
    return std::getenv(name);
}

double wall_ms() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
// UPX_PLAN_TIMING=1 (with UPX_TUNING=1): upx_plan_create prints where its time goes (scripts/plan_create_breakdown.py)
struct PlanClock {
    bool on;
    double t0, last;
    PlanClock() : on(knob("UPX_PLAN_TIMING") != nullptr), t0(wall_ms()), last(t0) {}
    void lap(const char* what) {
        if (!on) return;
        const double now = wall_ms();
        std::fprintf(stderr, "[upx_plan_create] %-34s %8.3f ms  (at %8.3f)\n", what, now - last, now - t0);
        last = now;
    }
};

// (log2 N, K, variant) -> fused kernel; the tables live in upx_reg_fused*.hip.
const KernelEntry* find_kernel(int log2n, int k, int variant) {
#if defined(UPX_EXPERIMENTS)
    if (variant == 1) return find_kernel_p8(log2n, k);
    if (variant == 2) return find_kernel_plain(log2n, k);
#endif
    const KernelEntry* e = find_kernel_default(log2n, k, variant);
    if (!e && variant >= 100) return nullptr;   // live-slot flavours: the caller tries the next wider one
    if (!e && variant != 0) e = find_kernel_default(log2n, k, 0);
    return e;
}
const ZoomEntry* find_zoom(int log2p, int rg, int k) {
    return log2p == 8 ? find_zoom_p256(rg, k) : log2p == 9 ? find_zoom_p512(rg, k) : log2p == 10 ? find_zoom_p1024(rg, k) : nullptr;
}

// 0 in the product library; experiment builds (-DUPX_EXPERIMENTS) take UPX_KERNEL_VARIANT = 1 (8 points per lane) / 2 (plain
// Stockham schedule) from csrc/experiments/
int default_variant() {
#if defined(UPX_EXPERIMENTS)
    const char* v = knob("UPX_KERNEL_VARIANT");
    return v ? std::atoi(v) : 0;
#else
    return 0;
#endif
}

int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

struct BandState {
    int n = 0, hop = 0, k = 0, log2n = 0;
    const KernelEntry* kern = nullptr;
    const BigEntry* big = nullptr;      // STFT > 8192: four-step path
    const ZoomEntry* zoom = nullptr;    // band-limited group: pruned analysis + residue-stream synthesis (upx_zoom.h)
    const ZoomEntry* zoom_a = nullptr;  // the entry whose ANALYSIS kernel runs (may hold fewer residues per workgroup)
    int zoom_p = 0, zoom_d = 0;         // decimated frame length P, decimation D = N / P
    upx::cf* d_ramp = nullptr;          // ramp seeds [D][P/16 + 4] (upx::zoom_ramp)
    int kmax = 0;                       // highest bin with non-zero gain (leader: over the whole group)
    upx::cf* d_tw_n = nullptr;          // W_N^(k1 n2) for the big path
    int chunk_frames = 0;
    float* d_wa = nullptr;
    float* d_ws = nullptr;     // synthesis window / N
    float* d_gain = nullptr;   // 0.5 * gain
    upx::cf* d_tw = nullptr;   // shared per N (owned by plan->tw)
    int blocks_override = 0;
    // band-limited analysis: pairs dealt by workgroup age (ZoomArgs::deal_rows), cached per launch geometry
    struct Deal {
        long long key = -1;
        int rows = 0;
        int* d_tab = nullptr;
    };
    std::vector<Deal> deals;
    // streams of unequal length (BandArgs::stream_m0): the table of the last geometry, on the host and on the device
    std::vector<int> h_m0, h_m0_sent;
    int* d_m0 = nullptr;
    size_t m0_cap = 0;
    int group_leader = 0;               // index of the band whose launch carries this band
    int group_size = 1;                 // leader: bands merged into its launch; merged members: 0
    int n_gain = 1;                     // gain slots per bin (merged bands overlap at crossovers)
    int last_wg = 0, last_f = 0;
    int timed_wg = 0;                   // last_wg of the last TIMED call (ev0 / ev1 belong to it)
    int last_streams = 0;               // fused launch of the last call: its streams, and whether h_m0 holds their first frames
    bool last_uneven = false;
    // launch geometry of the last call against the chip: workgroups of the (first) launch of the main kernel / of the
    // band-limited analysis and the workgroup slots the chip holds of them at once (upx_plan_band_fill)
    int fill_wg = 0, fill_slots = 0, fill_wg_a = 0, fill_slots_a = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // current slot of the rings below
    std::vector<hipEvent_t> ring0, ring1;       // kTimingSlots event pairs: one per recent upx_process_device call
    std::vector<char> ring_used;
    // band-limited path: events between its launches (analysis | synthesis | analysis | ... ), kMidEvents per slot
    std::vector<hipEvent_t> ring_mid;
    std::vector<int> ring_mid_n;                // events recorded in each slot (0: phases not split)
};
}   // namespace

struct upx_plan {
    int device = 0;
    int n_cu = 256;
    hipStream_t stream = nullptr;
    std::vector<BandState> bands;
    std::map<int, upx::cf*> tw;   // log2n -> device twiddles
    bool timing = false;
    bool warmed = false;            // upx_plan_reserve has run its tiny warm-up call through every kernel
    bool dry_run = false;           // upx_plan_reserve: upx_process_device prepares (geometry, tables, buffers) and launches nothing
    bool timed_once = false;
    long long timed_calls = 0;      // timed upx_process_device calls since timing was enabled
    unsigned int* d_scalar = nullptr;
    float pipe_ms[3] = {0.f, 0.f, 0.f};
    // tuning knobs, read from the environment ONCE when the plan is created (buffers are sized from them)
    long long knob_zoom_scratch_mb = 192;   // UPX_ZOOM_SCRATCH_MB: spectra between the two band-limited kernels
    double knob_zoom_fill = 2.0;            // UPX_ZOOM_FILL: synthesis streams per resident workgroup slot
    long long knob_zoom_f = 0;              // UPX_ZOOM_F: frames per synthesis stream (0 = automatic)
    long long knob_stream_chunk = 1LL << 22;   // UPX_STREAM_CHUNK: owned samples per chunk of a streamed host call
    int knob_edge_percent = 88;             // UPX_EDGE_PERCENT: stream length of a fused launch's edge workgroups (100 = uniform)
    int knob_zoom_once = 1;                 // UPX_ZOOM_ONCE: band-limited synthesis fills every workgroup slot once (0: UPX_ZOOM_FILL x slots)
    int knob_zoom_a_age = 8;                // UPX_ZOOM_A_AGE: % by which each later dispatch round of the band-limited analysis runs slower (0 = equal shares)
    int knob_zoom_edge_percent = 76;        // UPX_ZOOM_EDGE_PERCENT: length of the first / last synthesis stream of a signal (100 = like the others)
    double knob_zoom_c_cost = 0.55;         // UPX_ZOOM_C_COST: what a frame costs a centre stream, in Ls/Rs frames (stream length ratio)
    int knob_big_tail = 1;                  // UPX_BIG_TAIL: unfused frames of sixteen rows (N = 32 768, 65 536) at hop N/2, N/4, N/8 run step 2 + overlap-add as ONE pass (0: big_step2_inv + big_ola, the route of rounds 1-5)
    int knob_seam_vec = 1;                  // UPX_SEAM_VEC: stream-seam passes with 16-byte accesses where alignment allows (0: scalar passes)
#if defined(UPX_EXPERIMENTS)
    // experiment builds only (round 4 / 5 A/Bs, all rejected: docs/LOG.md).  The product library launches the groups in list
    // order - the reference's float32 band sum ((0 + b0) + b1) + ... (center_extraction.py:508-511) - and nothing can change that.
    int knob_first_band = -1;               // UPX_FIRST_BAND: launch this band's group first
    int knob_seam_inkernel = 0;             // UPX_SEAM_INKERNEL: stream seams of fused launches inside the launch (0 never, 1 always, 2 when the launch does not fill the chip)
    int* d_pair_cnt = nullptr;              // counters of seam_epilogue, one per neighbouring workgroup pair; allocated when the knob is on
    int pair_cnt_n = 1 << 16;
    int knob_band_rotate = 0;               // UPX_BAND_ROTATE: the launch groups start this many groups into the list (another sum association)
#endif
    int knob_min_stream_frames = 4;         // UPX_MIN_STREAM_FRAMES: shortest stream of a fused launch that does not fill the chip (>= K, even)
    int knob_prio_young = 3;                // UPX_PRIO_YOUNG: frame pairs of 4 in which the younger half of a launch leads (0 = off)
    float* d_seam = nullptr;        // stream tails of the fused kernel: [streams][3][(K-1) hop]
    size_t seam_floats = 0;
    upx::cf* d_scratch = nullptr;   // z | y | yc of the big path (shared by all big bands)
    size_t scratch_cf = 0;
    upx::cf* d_zoom = nullptr;      // y | yc spectra between the two kernels of the band-limited path
    size_t zoom_cf = 0;
    // streamed host calls (upx_process on long signals): copy streams, events and two rotating buffer sets
    hipStream_t s_h2d = nullptr, s_d2h = nullptr;
    hipEvent_t ev_h2d[2] = {nullptr, nullptr}, ev_comp[2] = {nullptr, nullptr};
    float* d_pipe_in[2] = {nullptr, nullptr};
    float* d_pipe_out[2] = {nullptr, nullptr};
    size_t pipe_in_floats = 0, pipe_out_floats = 0;
    void* d_pipe_raw[2] = {nullptr, nullptr};   // upx_process_lr: a work item's samples as the caller holds them
    size_t pipe_raw_bytes = 0;
    // device buffers of upx_wav_pipeline, kept between calls (grow only): pcm in, stereo, planes, 3 outputs
    void* d_wav[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t wav_cap[6] = {0, 0, 0, 0, 0, 0};
    // the shard between upx_wav_shard_begin and upx_wav_shard_finish
    bool wav_open = false;
    int64_t wav_tin = 0, wav_own = 0, wav_tout = 0;
    int wav_fmt = 0, wav_ch = 0;
    hipEvent_t wav_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // chunked WAV pipeline: owned frames per chunk (UPX_WAV_CHUNK, 0 = one chunk), per-chunk events, the spill of a chunk
    // parked while the next chunk's kernels overwrite that range, peaks (input, C, Ls, Rs) as bit patterns
    long long knob_wav_chunk = 1LL << 22;
    double knob_wav_kernel_rate = 21.0;     // UPX_WAV_KERNEL_RATE: M frames per ms the plan's kernels are assumed to sustain (chunk schedule)
    int knob_wav_uniform = 0;               // UPX_WAV_UNIFORM=1: chunks of equal length (tests: many seams; A/B of the geometry)
    std::vector<hipEvent_t> wav_piece_ev, wav_down_ev;
    float* d_wav_side[2] = {nullptr, nullptr};
    size_t wav_side_floats = 0;
    unsigned int* d_wav_peaks = nullptr;
    double wav_in_peak = 0.0;               // the input peak the last seal reported (32-bit PCM: taken on the integers)
    bool wav_peaks_pending_head = false;    // multi-rank: chunk 0's plane peaks wait for the RCCL seam
    int64_t wav_head_own = 0;
    // between upx_wav_shard_open and _seal: the chunk list, frames fed / decoded, the next chunk to run
    bool wav_feeding = false;
    std::vector<upx::WavChunkRec> wav_chunk_list;
    int64_t wav_fed = 0, wav_decoded = 0, wav_spill = 0, wav_cspill = 0;
    int wav_next_chunk = 0, wav_n_feed = 0;
    upx_comm* wav_comm = nullptr;
    double wav_t0 = 0.0;
    // upx_wav_shard_finish_async: pieces on their way down
    int wav_pieces = 0;
    // streaming state of a one-band plan (upx_stream_chunk): block in, one frame's three reconstructions, the
    // overlap-add ring [3][N] with its read position, the emitted hops
    float* d_chunk = nullptr;        // [2 N] block | [3 N] rec | [3 N] ring | [3 hop] out
    float* h_chunk = nullptr;        // page-locked: [2 N] block | [3 N] results
    int chunk_pos = 0;
};

struct upx_comm {
    upx_plan* plan = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, n_ranks = 1;
    float* d_seam = nullptr;
    long long seam_floats = 0;
    hipEvent_t ev_start = nullptr;  // in front of the last seam exchange's all-reduce on the plan's stream ...
    hipEvent_t ev_done = nullptr;   // ... and behind the exchange (upx_comm_wait: the peers' time limit starts at ev_start)
    bool pending = false;           // an exchange has been queued and not yet waited for
    bool aborted = false;           // ncclCommAbort has run: the communicator is gone
    double timeout_s = 600.0;       // UPX_COMM_TIMEOUT, else UPX_RDZV_TIMEOUT, else 600 s
};

namespace {
// ---------------------------------------------------------------------------
// RCCL through dlopen (so single-GPU use has no RCCL dependency)
// ---------------------------------------------------------------------------
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;                        // optional
    ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t*) = nullptr;   // optional
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;

int load_rccl() {
    if (g_rccl.h) return UPX_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* h = nullptr;
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return fail(UPX_ERR_RCCL, "cannot dlopen librccl: %s", dlerror());
    Rccl r;
    r.h = h;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.CommAbort = (decltype(r.CommAbort))dlsym(h, "ncclCommAbort");
    r.CommGetAsyncError = (decltype(r.CommGetAsyncError))dlsym(h, "ncclCommGetAsyncError");
    r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.GetErrorString)
        return fail(UPX_ERR_RCCL, "librccl lacks a required symbol");
    g_rccl = r;
    return UPX_OK;
}

#define NCCL_TRY(expr)                                                                              \
    do {                                                                                            \
        ncclResult_t r_ = (expr);                                                                   \
        if (r_ != ncclSuccess) return fail(UPX_ERR_RCCL, "%s: %s", #expr, g_rccl.GetErrorString(r_)); \
    } while (0)

// ---- launch geometry of the band-limited path ------------------------------------------------------------------
// frames of spectra the scratch between analysis and synthesis holds (a launch pair covers at most this many):
// 128 MB of y + 64 MB of yc (inside the 256 MB Infinity Cache; 10 min at 48 kHz is one launch pair for every band
// with D >= 16, and short launch pairs cost more in tails than a smaller scratch saves)
int zoom_frames_cap(const upx_plan* p, int zp) {
    const long long mb = p->knob_zoom_scratch_mb;
    long long frames = mb * (1 << 20) / ((long long)zp * 12);   // P complex per frame + P/2 per frame for the pairs
    if (frames < 64) frames = 64;
    return (int)(frames & ~1LL);
}
// workgroups of the band-limited kernels one CU holds (waves per SIMD by registers; LDS: 160 KB)
int zoom_resident(const BandState& s, bool analysis = false) {
    const ZoomEntry* z = analysis ? s.zoom_a : s.zoom;
    const int wpe = analysis ? z->wpe_a : z->wpe, lds = analysis ? z->lds_bytes_a : z->lds_bytes;
    int by_waves = (wpe * 256) / z->wg;
    const int by_lds = (160 * 1024) / lds;
    if (by_waves > by_lds) by_waves = by_lds;
    return by_waves < 1 ? 1 : by_waves;
}
// streams the synthesis aims for: about UPX_ZOOM_FILL x the chip's workgroup slots, an Ls/Rs workgroup counting
// 1 and a centre workgroup 1/2 per (stream, residue group)
long long zoom_streams_wanted(const upx_plan* p, const BandState& s) {
    const double fill = p->knob_zoom_fill;
    const int groups = s.zoom_d / s.zoom->rg;
    long long want = (long long)(fill * p->n_cu * zoom_resident(s) / (1.5 * groups));
    return want < 1 ? 1 : want;
}

// streams the automatic launch geometry of a fused band aims for: every resident workgroup slot of the chip once
long long max_auto_streams(const upx_plan* p, const BandState& s) {
    int resident = (s.kern->wpe * 256) / s.kern->wg;              // workgroups per CU by registers
    const int by_lds = (160 * 1024) / s.kern->lds_bytes;          // ... and by LDS
    if (resident > by_lds) resident = by_lds;
    if (resident < 1) resident = 1;
    return (long long)p->n_cu * resident * s.kern->g;
}

// a reduction to one word: enough blocks to fill the chip twice, one atomic each
int grid_reduce(long long n) {
    long long g = (n / 4 + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 1024 ? 1024 : g));
}
// the 16-byte seam passes apply: hop a multiple of four samples, planes and seam buffer 16-byte aligned (UPX_SEAM_VEC=0: never)
bool seam_vec_ok(const upx_plan* p, const float* c, const float* l, const float* r, int hop) {
    const size_t bits = (size_t)c | (size_t)l | (size_t)r | (size_t)p->d_seam;
    return p->knob_seam_vec != 0 && hop % 4 == 0 && (bits & 15u) == 0;
}
int grid_for(long long n) {
    long long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 2048 ? 2048 : g));
}
}   // namespace

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" {

int upx_abi_version(void) { return 1; }

const char* upx_last_error(void) { return g_err.c_str(); }

int upx_device_count(int* count) {
    if (!count) return fail(UPX_ERR_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(UPX_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return UPX_OK;
}

int upx_supported(int32_t block_size, int32_t hop) {
    if (block_size < 1 || hop < 1 || hop > block_size) return 0;
    const int l = ilog2_exact(block_size);
    if (l < 0 || !find_big(l)) return 0;
    return (block_size + hop - 1) / hop <= kMaxFramesPerSample ? 1 : 0;
}

// ---------------------------------------------------------------------------
// upx_plan_create = selection (host arithmetic: launch groups, kernel family and flavour per group, per-bin gain tables;
// plan_select - no device is touched, upx_plan_kernel_names runs it alone) + upload (device tables and buffers; plan_upload)
// ---------------------------------------------------------------------------
namespace {
int check_bands(const char* who, int n_bands, const int32_t* block_size, const int32_t* hop, const float* w_analysis,
                const float* w_synthesis, const float* gain) {
    if (n_bands < 1 || !block_size || !hop || !w_analysis || !w_synthesis || !gain)
        return fail(UPX_ERR_INVALID, "%s: NULL argument or n_bands < 1", who);
    for (int b = 0; b < n_bands; ++b) {
        if (hop[b] < 1) return fail(UPX_ERR_INVALID, "Overlap too large; hop size < 1 is not allowed.");
        if (!upx_supported(block_size[b], hop[b]))
            return fail(UPX_ERR_UNSUPPORTED,
                        "band %d: STFT size %d with hop %d is not covered by the gfx950 kernels "
                        "(power-of-two sizes 64..65536, at most 64 frames overlapping a sample)",
                        b, block_size[b], hop[b]);
    }
    return UPX_OK;
}

void read_knobs(upx_plan* p) {
    if (const char* e = knob("UPX_ZOOM_SCRATCH_MB")) p->knob_zoom_scratch_mb = std::atoll(e);
    if (const char* e = knob("UPX_ZOOM_FILL")) p->knob_zoom_fill = std::atof(e);
    if (const char* e = knob("UPX_ZOOM_F")) p->knob_zoom_f = std::atoll(e);
    if (const char* e = knob("UPX_STREAM_CHUNK")) p->knob_stream_chunk = std::atoll(e);
    if (const char* e = knob("UPX_EDGE_PERCENT")) p->knob_edge_percent = std::atoi(e);
    if (const char* e = knob("UPX_PRIO_YOUNG")) p->knob_prio_young = std::atoi(e);
    if (const char* e = knob("UPX_MIN_STREAM_FRAMES")) p->knob_min_stream_frames = std::atoi(e);
    if (const char* e = knob("UPX_WAV_CHUNK")) p->knob_wav_chunk = std::atoll(e);
    if (const char* e = knob("UPX_SEAM_VEC")) p->knob_seam_vec = std::atoi(e);
    if (const char* e = knob("UPX_BIG_TAIL")) p->knob_big_tail = std::atoi(e);
    if (const char* e = knob("UPX_WAV_UNIFORM")) p->knob_wav_uniform = std::atoi(e);
    if (const char* e = knob("UPX_WAV_KERNEL_RATE")) p->knob_wav_kernel_rate = std::atof(e);
    if (const char* e = knob("UPX_ZOOM_ONCE")) p->knob_zoom_once = std::atoi(e);
    if (const char* e = knob("UPX_ZOOM_C_COST")) p->knob_zoom_c_cost = std::atof(e);
    if (const char* e = knob("UPX_ZOOM_EDGE_PERCENT")) p->knob_zoom_edge_percent = std::atoi(e);
    if (const char* e = knob("UPX_ZOOM_A_AGE")) p->knob_zoom_a_age = std::atoi(e);
#if defined(UPX_EXPERIMENTS)
    // UPX_N_CU: the launch geometry is cut for this many compute units, e.g. half the chip for each of two plans whose
    // launches are meant to be resident side by side (scripts/r5_two_tiles.py)
    if (const char* e = knob("UPX_N_CU"))
        if (std::atoi(e) >= 8) p->n_cu = std::atoi(e);
    if (const char* e = knob("UPX_FIRST_BAND")) p->knob_first_band = std::atoi(e);
    if (const char* e = knob("UPX_BAND_ROTATE")) p->knob_band_rotate = std::atoi(e);
    if (const char* e = knob("UPX_SEAM_INKERNEL")) p->knob_seam_inkernel = std::atoi(e);
#endif
}

// Selection: p->bands gets geometry, launch groups, pass band, kernel family and flavour; gain_tables[b] the per-bin gain
// list of the launch band b leads, in the order its kernel reads it.  Host arithmetic only.
int plan_select(upx_plan* p, int n_bands, const int32_t* block_size, const int32_t* hop, const float* w_analysis,
                const float* w_synthesis, const float* gain, std::vector<size_t>& band_win_off,
                std::vector<std::vector<float>>& gain_tables) {
    p->bands.resize(n_bands);
    std::vector<size_t> band_gain_off;
    // pass 1: geometry, merged groups, pass band
    {
        size_t off_w = 0, off_g = 0;
        for (int b = 0; b < n_bands; ++b) {
            BandState& s = p->bands[b];
            s.n = block_size[b];
            s.hop = hop[b];
            s.k = (s.n + s.hop - 1) / s.hop;   // frames covering one sample
            s.log2n = ilog2_exact(s.n);
            const int nb = s.n / 2 + 1;
            // Merge with the previous band when it has the same STFT size, hop and (bit-identical) windows:
            // the transforms are then the same linear operators and only gain -> mask runs per band.
            s.group_leader = b;
            if (b > 0 && !knob("UPX_NO_BAND_MERGE")) {
                const BandState& q = p->bands[b - 1];
                if (q.n == s.n && q.hop == s.hop &&
                    !std::memcmp(w_analysis + off_w, w_analysis + off_w - s.n, s.n * sizeof(float)) &&
                    !std::memcmp(w_synthesis + off_w, w_synthesis + off_w - s.n, s.n * sizeof(float)))
                    s.group_leader = q.group_leader;
            }
            if (s.group_leader != b) {
                s.group_size = 0;
                p->bands[s.group_leader].group_size += 1;
            }
            s.kmax = 0;
            for (int k = 0; k < nb; ++k)
                if (gain[off_g + k] != 0.f) s.kmax = k;
            BandState& lead = p->bands[s.group_leader];
            if (s.kmax > lead.kmax) lead.kmax = s.kmax;
            band_win_off.push_back(off_w);
            band_gain_off.push_back(off_g);
            off_w += s.n;
            off_g += nb;
        }
    }
    // pass 2: kernel family per group.
    //   band-limited path (upx_zoom.h): hop = N/2, N/4 or N/8, every non-zero gain below bin P/2 for a P = 256, 512 or
    //     1024 with D = N / P >= 8: the reference's planner makes every band with a large STFT such a band;
    //   fused streaming kernel (upx_core.h): the same hops, N <= 8192, any pass band;
    //   unfused pipeline (upx_big.h): everything else.
    const bool force_unfused = knob("UPX_FORCE_UNFUSED") != nullptr;
    // UPX_ZOOM = smallest decimation D = N / P (>= 8) for which the band-limited path is taken (0: never).  Measured on
    // the MI355X: from D = 8 on it beats the fused kernel (N = 4096: 0.46 vs 0.54 ms); at D = 4 the fused
    // kernel's single launch wins.
    const char* zoom_env = knob("UPX_ZOOM");
    const int zoom_min_d = zoom_env ? std::atoi(zoom_env) : 8;
    for (int b = 0; b < n_bands; ++b) {
        BandState& s = p->bands[b];
        if (s.group_size == 0) continue;
        const bool std_hop = s.n % s.hop == 0 && (s.k == 2 || s.k == 4 || s.k == 8);
        if (std_hop && !force_unfused && zoom_min_d > 0) {
            int zp = 256;
            while (zp < 2 * (s.kmax + 1)) zp *= 2;
            const int d = s.n / zp;
            if (zp <= 1024 && d >= 8 && d >= zoom_min_d) {
                s.zoom = find_zoom(ilog2_exact(zp), d >= 16 ? 16 : d, s.k);
                if (s.zoom) {
                    s.zoom_p = zp;
                    s.zoom_d = d;
                    // The spectra between the two kernels do not depend on how many residues a workgroup holds, so the
                    // analysis need not use the synthesis' 16.  At P >= 512 sixteen residues are a workgroup of 8 or 16 waves
                    // at 128 VGPRs without the input prefetch; eight residues (4 or 8 waves, 168 VGPRs, prefetching, two
                    // groups per frame) measured 0.664 -> 0.593 ms on C4's main group (N = 8192, P = 512) and -5 % on the
                    // default plan's 65 536 / 16 384 bands; at P = 256 sixteen stay (+3 % with eight).  UPX_ZOOM_A_RG overrides.
                    s.zoom_a = s.zoom;
                    const char* e = knob("UPX_ZOOM_A_RG");
                    const int rg_a = e ? std::atoi(e) : (zp >= 512 ? 8 : s.zoom->rg);
                    if (rg_a != s.zoom->rg) {
                        const ZoomEntry* alt = find_zoom(ilog2_exact(zp), rg_a, s.k);
                        if (alt && d % alt->rg == 0) s.zoom_a = alt;
                    }
                }
            }
        }
        if (!s.zoom && std_hop && !force_unfused) s.kern = find_kernel(s.log2n, s.n / s.hop, default_variant());
        if (!s.zoom && !s.kern) s.big = find_big(s.log2n);
    }
    // pass 3: per-bin gain lists of the (possibly merged) launch: gain[q][k] = q-th non-zero half-gain of bin k, band
    // order - and, from them, the flavour of a fused launch that carries one band
    gain_tables.assign(n_bands, {});
    for (int b = 0; b < n_bands; ++b) {
        BandState& s = p->bands[b];
        if (s.group_size == 0) continue;
        const int nb = s.n / 2 + 1;
        std::vector<int> count(nb, 0);
        int slots = 1;
        for (int m = b; m < b + s.group_size; ++m)
            for (int k = 0; k < nb; ++k)
                if (gain[band_gain_off[m] + k] != 0.f && ++count[k] > slots) slots = count[k];
        std::vector<float>& table = gain_tables[b];
        table.assign((size_t)slots * nb, 0.f);
        std::fill(count.begin(), count.end(), 0);
        for (int m = b; m < b + s.group_size; ++m)
            for (int k = 0; k < nb; ++k) {
                const float g = gain[band_gain_off[m] + k];
                if (g != 0.f) table[(size_t)count[k]++ * nb + k] = 0.5f * g;
            }
        if (!s.zoom) {
            // rows in the order the kernel's threads read them (natural for plain streams, whole-frame rows and
            // the band-limited path)
            int (*order)(int) = s.kern ? s.kern->gain_bin : s.big->gain_bin;
            std::vector<float> natural(table);
            for (int q = 0; q < slots; ++q)
                for (int i = 0; i < nb; ++i) table[(size_t)q * nb + i] = natural[(size_t)q * nb + order(i)];
        }
        s.n_gain = slots;
        if (s.kern && slots == 1 && s.log2n != 12 && default_variant() == 0 && !knob("UPX_NO_SINGLE_FLAVOUR")) {
            // one gain slot per bin: the flavour without the second slot's registers (same twiddle / gain layout).
            // Measured (MI355X, C4 plan): N = 2048 1.52 -> 1.43 ms (its ten spills are gone), N = 512 1.43 -> 1.39;
            // N = 4096 loses 3 % (its few spills sit on the signal-edge path only), so it keeps the general flavour.
            const KernelEntry* one = find_kernel(s.log2n, s.n / s.hop, 10);
            if (one && one->layout == s.kern->layout) {
                // ... and, where one is built, the flavour specialised for the own-bin slots that carry gain: slot s of
                // a lane holds bin lane + s lanes (s < 8); the smallest instantiated range [S0, S1) that covers them,
                // S1 = 8 whenever the Nyquist bin carries gain
                if (!knob("UPX_NO_LIVE_FLAVOUR") && s.k == 4) {
                    int lo = 8, hi = 0;
                    for (int sl = 0; sl < 8; ++sl)
                        for (int i = sl * one->lanes; i < (sl + 1) * one->lanes; ++i)
                            if (table[(size_t)i] != 0.f) { lo = sl < lo ? sl : lo; hi = sl + 1; break; }
                    if (table[(size_t)nb - 1] != 0.f) hi = 8;
                    const KernelEntry* live = nullptr;
                    for (int b1 = hi; b1 <= 8 && !live && lo < hi; ++b1)
                        for (int a0 = lo > 1 ? 1 : lo; a0 >= 0 && !live; --a0)
                            if (a0 > 0 || b1 < 8) live = find_kernel(s.log2n, 4, 100 + 10 * a0 + b1);
                    if (live && live->layout == s.kern->layout) one = live;
#if defined(UPX_EXPERIMENTS)
                    // UPX_DUAL = 1: two stream sets per wave, one wave per SIMD (experiments/upx_exp_fused_dual.hip);
                    // = 2: the N = 256 launch only, = 3: the N = 1024 launch only
                    if (const char* e = knob("UPX_DUAL")) {
                        const int mode = std::atoi(e);
                        const bool want = mode == 1 || (mode == 2 && s.log2n == 8) || (mode == 3 && s.log2n == 10);
                        const KernelEntry* dual = want ? find_kernel_dual(s.log2n, live == one ? hi : 0) : nullptr;
                        if (want && !dual) dual = find_kernel_dual(s.log2n, 0);
                        if (dual && dual->layout == s.kern->layout) one = dual;
                    }
#endif
                }
                s.kern = one;
            }
        }
        for (int m = b + 1; m < b + s.group_size; ++m) {   // members: only for reporting
            p->bands[m].zoom = s.zoom; p->bands[m].zoom_a = s.zoom_a; p->bands[m].kern = s.kern; p->bands[m].big = s.big;
            p->bands[m].zoom_p = s.zoom_p; p->bands[m].zoom_d = s.zoom_d;
        }
    }
    return UPX_OK;
}

// Upload: what the selected kernels read on the device - windows, twiddles, ramp seeds, gain tables - and the plan's
// buffers (timing events, scratch of the unfused and band-limited paths, stream seam buffer)
int plan_upload(upx_plan* p, const float* w_analysis, const float* w_synthesis, const std::vector<size_t>& band_win_off,
                const std::vector<std::vector<float>>& gain_tables, PlanClock* clock) {
    const int n_bands = (int)p->bands.size();
    double t_events = 0.0, t_prepare = 0.0, t_tables = 0.0, t_mark = wall_ms();
    auto mark = [&t_mark](double& acc) {
        const double now = wall_ms();
        acc += now - t_mark;
        t_mark = now;
    };
    for (int b = 0; b < n_bands; ++b) {
        BandState& s = p->bands[b];
        const size_t off_w = band_win_off[b];
        s.ring0.assign(kTimingSlots, nullptr);
        s.ring1.assign(kTimingSlots, nullptr);
        s.ring_used.assign(kTimingSlots, 0);
        for (int i = 0; i < kTimingSlots; ++i) {
            HIP_TRY(hipEventCreate(&s.ring0[i]));
            HIP_TRY(hipEventCreate(&s.ring1[i]));
        }
        s.ev0 = s.ring0[0];
        s.ev1 = s.ring1[0];
        mark(t_events);
        if (s.group_size == 0) continue;   // carried by its group leader's launch
        // the dynamic LDS of the group's kernels
        if (s.zoom && s.zoom_a != s.zoom)
            if (int e = s.zoom_a->prepare())
                return fail(UPX_ERR_HIP, "hipFuncSetAttribute(STFT %d): %s", s.n, hipGetErrorString((hipError_t)e));
        if (int e = s.zoom ? s.zoom->prepare() : (s.kern ? s.kern->prepare() : s.big->prepare()))
            return fail(UPX_ERR_HIP, "hipFuncSetAttribute(STFT %d): %s", s.n, hipGetErrorString((hipError_t)e));
        mark(t_prepare);
        if (s.zoom) {
            s.ring_mid.assign((size_t)kTimingSlots * kMidEvents, nullptr);
            s.ring_mid_n.assign(kTimingSlots, 0);
            for (auto& e : s.ring_mid) HIP_TRY(hipEventCreate(&e));
        }
        mark(t_events);
        std::vector<float> ws(s.n);
        for (int i = 0; i < s.n; ++i) ws[i] = w_synthesis[off_w + i] / (float)s.n;   // exact: N is a power of two
        HIP_TRY(hipMalloc(&s.d_wa, s.n * sizeof(float)));
        HIP_TRY(hipMalloc(&s.d_ws, s.n * sizeof(float)));
        HIP_TRY(hipMemcpy(s.d_wa, w_analysis + off_w, s.n * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(s.d_ws, ws.data(), s.n * sizeof(float), hipMemcpyHostToDevice));
        // twiddle tables are shared by the bands that use the same kernel layout
        const int tw_key = s.zoom ? 9000000 + ilog2_exact(s.zoom_p)
                                  : s.log2n * 100000 + (s.kern ? s.kern->layout : 0);
        auto it = p->tw.find(tw_key);
        if (it == p->tw.end()) {
            const size_t cnt = (size_t)(s.zoom ? s.zoom->tw_cf : (s.kern ? s.kern->tw_cf : s.big->row_tw_cf));
            std::vector<upx::cf> host(cnt);
            if (s.zoom) s.zoom->fill_tw(host.data());
            else if (s.kern) s.kern->fill_tw(host.data());
            else s.big->fill_tw_rows(host.data());
            upx::cf* d = nullptr;
            HIP_TRY(hipMalloc(&d, cnt * sizeof(upx::cf)));
            it = p->tw.emplace(tw_key, d).first;
            HIP_TRY(hipMemcpy(d, host.data(), cnt * sizeof(upx::cf), hipMemcpyHostToDevice));
        }
        s.d_tw = it->second;
        if (s.zoom) {
            std::vector<upx::cf> host(upx::zoom_ramp_count(s.n, s.zoom_p));
            upx::fill_zoom_ramp(host.data(), s.n, s.zoom_p, turn_trig);
            HIP_TRY(hipMalloc(&s.d_ramp, host.size() * sizeof(upx::cf)));
            HIP_TRY(hipMemcpy(s.d_ramp, host.data(), host.size() * sizeof(upx::cf), hipMemcpyHostToDevice));
            // spectra between the two kernels: kZoomFrames frames of P complex (+ half as many pairs)
            const size_t need = (size_t)zoom_frames_cap(p, s.zoom_p) * s.zoom_p * 3 / 2;
            if (need > p->zoom_cf) p->zoom_cf = need;
        }
        if (s.big) {
            // frames per chunk: 2^24 complex per scratch buffer (128 MB; z + y + yc = 320 MB).  Measured on the
            // default plan: 2^21 10.2 ms, 2^22 7.9, 2^23 7.0, 2^24 6.7, 2^25 8.3 - below, the 7 launches per chunk
            // (10-19 us each) are too short; above, the scratch falls out of the Infinity Cache altogether.
            const char* cl = knob("UPX_BIG_CHUNK_LOG2");
            s.chunk_frames = (1 << (cl ? std::atoi(cl) : 24)) / s.n;
            if (s.chunk_frames < 2 * s.k + 4) s.chunk_frames = 2 * s.k + 4;
            s.chunk_frames += s.chunk_frames & 1;
            std::vector<upx::cf> host((size_t)s.n);
            s.big->fill_tw_n(host.data());
            HIP_TRY(hipMalloc(&s.d_tw_n, (size_t)s.n * sizeof(upx::cf)));
            HIP_TRY(hipMemcpy(s.d_tw_n, host.data(), (size_t)s.n * sizeof(upx::cf), hipMemcpyHostToDevice));
            const size_t need = (size_t)s.chunk_frames * s.n * 5 / 2;   // z + y + yc/2
            if (need > p->scratch_cf) p->scratch_cf = need;
        }
        const std::vector<float>& table = gain_tables[b];
        HIP_TRY(hipMalloc(&s.d_gain, table.size() * sizeof(float)));
        HIP_TRY(hipMemcpy(s.d_gain, table.data(), table.size() * sizeof(float), hipMemcpyHostToDevice));
        mark(t_tables);
    }
    if (clock && clock->on) {
        std::fprintf(stderr, "[upx_plan_create]   events %.3f ms, hipFuncSetAttribute (code objects) %.3f ms, tables (malloc + memcpy) %.3f ms\n",
                     t_events, t_prepare, t_tables);
        clock->lap("plan_upload: per band");
    }
    if (p->scratch_cf) HIP_TRY(hipMalloc(&p->d_scratch, p->scratch_cf * sizeof(upx::cf)));
    if (p->zoom_cf) HIP_TRY(hipMalloc(&p->d_zoom, p->zoom_cf * sizeof(upx::cf)));
    // stream seam buffer, sized for the largest automatic launch (no allocation, and no stream synchronisation,
    // inside upx_process_device unless upx_plan_set_blocks_per_stream or a very long signal asks for more streams)
    for (const auto& s : p->bands) {
        if (s.group_size == 0 || s.big) continue;
        const size_t streams = s.zoom ? (size_t)zoom_streams_wanted(p, s) + 1 : (size_t)(max_auto_streams(p, s) + s.kern->g);
        const size_t need = streams * 3 * (size_t)(s.k - 1) * s.hop;
        if (need > p->seam_floats) p->seam_floats = need;
    }
    if (p->seam_floats) HIP_TRY(hipMalloc(&p->d_seam, p->seam_floats * sizeof(float)));
    return UPX_OK;
}

// "analysis|synthesis" for a band-limited group, the one kernel otherwise
void group_kernel_names(const BandState& s, std::string& out) {
    if (s.zoom) {
        out += std::string(s.zoom_a->name_analysis) + "|" + s.zoom->name_synthesis;
    } else if (s.kern) {
        out += s.kern->name;
    } else {
        char buf[96];
        std::snprintf(buf, sizeof buf, "upx_big pipeline, STFT %d (upx_big.h)", s.n);
        out += buf;
    }
}
}   // namespace

int upx_plan_create(upx_plan** out, int device, int n_bands, const int32_t* block_size, const int32_t* hop,
                    const float* w_analysis, const float* w_synthesis, const float* gain) {
    if (!out) return fail(UPX_ERR_INVALID, "upx_plan_create: NULL argument or n_bands < 1");
    if (int rc = check_bands("upx_plan_create", n_bands, block_size, hop, w_analysis, w_synthesis, gain)) return rc;
    PlanClock clock;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) return fail(UPX_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= n_dev) return fail(UPX_ERR_INVALID, "device %d out of range (0..%d)", device, n_dev - 1);
    clock.lap("hipGetDeviceCount");
    HIP_TRY(hipSetDevice(device));
    clock.lap("hipSetDevice");
    // every early return below releases what has been created so far (stream, events, device memory)
    struct Guard {
        upx_plan* p;
        ~Guard() { if (p) upx_plan_destroy(p); }
    } guard{new upx_plan()};
    upx_plan* p = guard.p;
    p->device = device;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0)
        p->n_cu = prop.multiProcessorCount;
    read_knobs(p);
    clock.lap("device properties + knobs");
    HIP_TRY(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
    clock.lap("hipStreamCreate");
    HIP_TRY(hipMalloc(&p->d_scalar, sizeof(unsigned int)));
    clock.lap("first hipMalloc");
#if defined(UPX_EXPERIMENTS)
    if (p->knob_seam_inkernel != 0) {
        HIP_TRY(hipMalloc(&p->d_pair_cnt, (size_t)p->pair_cnt_n * sizeof(int)));
        HIP_TRY(hipMemset(p->d_pair_cnt, 0, (size_t)p->pair_cnt_n * sizeof(int)));
    }
#endif
    std::vector<size_t> band_win_off;
    std::vector<std::vector<float>> gain_tables;
    if (int rc = plan_select(p, n_bands, block_size, hop, w_analysis, w_synthesis, gain, band_win_off, gain_tables)) return rc;
    clock.lap("plan_select (host)");
    if (int rc = plan_upload(p, w_analysis, w_synthesis, band_win_off, gain_tables, &clock)) return rc;
    clock.lap("plan_upload: rest");
    guard.p = nullptr;
    *out = p;
    return UPX_OK;
}

// The selection alone, without a device: for every band the kernel(s) of the launch that carries it - "analysis|synthesis"
// for a band-limited group - one line per band, in `names` (NUL-terminated; UPX_ERR_INVALID when n is too small).  What
// upx_plan_create would select in THIS process (the same UPX_TUNING gate); CPU tests pin the selection with it.
int upx_plan_kernel_names(int n_bands, const int32_t* block_size, const int32_t* hop, const float* w_analysis,
                          const float* w_synthesis, const float* gain, char* names, size_t n) {
    if (!names || n == 0) return fail(UPX_ERR_INVALID, "upx_plan_kernel_names: NULL argument");
    if (int rc = check_bands("upx_plan_kernel_names", n_bands, block_size, hop, w_analysis, w_synthesis, gain)) return rc;
    upx_plan plan;          // never reaches the device: no stream, no buffers
    read_knobs(&plan);
    std::vector<size_t> band_win_off;
    std::vector<std::vector<float>> gain_tables;
    if (int rc = plan_select(&plan, n_bands, block_size, hop, w_analysis, w_synthesis, gain, band_win_off, gain_tables)) return rc;
    std::string text;
    for (const auto& s : plan.bands) {
        group_kernel_names(plan.bands[(size_t)s.group_leader], text);
        text += "\n";
    }
    if (text.size() + 1 > n) return fail(UPX_ERR_INVALID, "upx_plan_kernel_names: %zu bytes needed", text.size() + 1);
    std::memcpy(names, text.c_str(), text.size() + 1);
    return UPX_OK;
}

void upx_plan_destroy(upx_plan* p) {
    if (!p) return;
    (void)hipSetDevice(p->device);
    for (auto& s : p->bands) {
        if (s.d_wa) (void)hipFree(s.d_wa);
        if (s.d_ws) (void)hipFree(s.d_ws);
        if (s.d_gain) (void)hipFree(s.d_gain);
        if (s.d_tw_n) (void)hipFree(s.d_tw_n);
        if (s.d_ramp) (void)hipFree(s.d_ramp);
        if (s.d_m0) (void)hipFree(s.d_m0);
        for (auto& d : s.deals) (void)hipFree(d.d_tab);
        for (auto e : s.ring0) if (e) (void)hipEventDestroy(e);
        for (auto e : s.ring1) if (e) (void)hipEventDestroy(e);
        for (auto e : s.ring_mid) if (e) (void)hipEventDestroy(e);
    }
    for (auto& kv : p->tw) (void)hipFree(kv.second);
    if (p->d_scalar) (void)hipFree(p->d_scalar);
#if defined(UPX_EXPERIMENTS)
    if (p->d_pair_cnt) (void)hipFree(p->d_pair_cnt);
#endif
    if (p->d_scratch) (void)hipFree(p->d_scratch);
    if (p->d_zoom) (void)hipFree(p->d_zoom);
    if (p->d_seam) (void)hipFree(p->d_seam);
    for (int i = 0; i < 2; ++i) {
        if (p->d_pipe_in[i]) (void)hipFree(p->d_pipe_in[i]);
        if (p->d_pipe_out[i]) (void)hipFree(p->d_pipe_out[i]);
        if (p->d_pipe_raw[i]) (void)hipFree(p->d_pipe_raw[i]);
        if (p->ev_h2d[i]) (void)hipEventDestroy(p->ev_h2d[i]);
        if (p->ev_comp[i]) (void)hipEventDestroy(p->ev_comp[i]);
    }
    for (auto* q : p->d_wav) if (q) (void)hipFree(q);
    for (auto e : p->wav_ev) if (e) (void)hipEventDestroy(e);
    for (auto e : p->wav_piece_ev) if (e) (void)hipEventDestroy(e);
    for (auto e : p->wav_down_ev) if (e) (void)hipEventDestroy(e);
    for (auto* q : p->d_wav_side) if (q) (void)hipFree(q);
    if (p->d_wav_peaks) (void)hipFree(p->d_wav_peaks);
    if (p->d_chunk) (void)hipFree(p->d_chunk);
    if (p->h_chunk) (void)hipHostFree(p->h_chunk);
    if (p->s_h2d) (void)hipStreamDestroy(p->s_h2d);
    if (p->s_d2h) (void)hipStreamDestroy(p->s_d2h);
    if (p->stream) (void)hipStreamDestroy(p->stream);
    delete p;
}

int upx_plan_set_blocks_per_stream(upx_plan* p, int band, int blocks) {
    if (!p || blocks < 0 || band < -1 || band >= (int)p->bands.size())
        return fail(UPX_ERR_INVALID, "upx_plan_set_blocks_per_stream: bad argument");
    for (int b = 0; b < (int)p->bands.size(); ++b)
        if (band < 0 || band == b) p->bands[b].blocks_override = blocks;
    return UPX_OK;
}

int upx_dev_alloc(upx_plan* p, void** ptr, size_t bytes) {
    if (!p || !ptr) return fail(UPX_ERR_INVALID, "upx_dev_alloc: NULL argument");
    HIP_TRY(hipSetDevice(p->device));
    hipError_t e = hipMalloc(ptr, bytes ? bytes : 4);
    if (e != hipSuccess) return fail(UPX_ERR_NOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return UPX_OK;
}
int upx_dev_free(upx_plan* p, void* ptr) {
    if (!p) return fail(UPX_ERR_INVALID, "upx_dev_free: NULL plan");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipFree(ptr));
    return UPX_OK;
}
// page-locked host memory: copies to and from it go over the link without a staging copy and without page faults
int upx_host_alloc(upx_plan* p, void** ptr, size_t bytes) {
    if (!p || !ptr) return fail(UPX_ERR_INVALID, "upx_host_alloc: NULL argument");
    HIP_TRY(hipSetDevice(p->device));
    hipError_t e = hipHostMalloc(ptr, bytes ? bytes : 4, hipHostMallocPortable);
    if (e != hipSuccess) return fail(UPX_ERR_NOMEM, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return UPX_OK;
}
int upx_host_free(upx_plan* p, void* ptr) {
    // plan == NULL: a block that outlived every plan of the process (hostmem.py); page-locked memory is not tied to a device
    if (p) HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipHostFree(ptr));
    return UPX_OK;
}
int upx_dev_memset(upx_plan* p, void* ptr, int value, size_t bytes) {
    if (!p) return fail(UPX_ERR_INVALID, "upx_dev_memset: NULL plan");
    HIP_TRY(hipMemsetAsync(ptr, value, bytes, p->stream));
    return UPX_OK;
}
int upx_copy_h2d(upx_plan* p, void* dst, const void* src, size_t bytes) {
    if (!p) return fail(UPX_ERR_INVALID, "upx_copy_h2d: NULL plan");
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return UPX_OK;
}
int upx_copy_d2h(upx_plan* p, void* dst, const void* src, size_t bytes) {
    if (!p) return fail(UPX_ERR_INVALID, "upx_copy_d2h: NULL plan");
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return UPX_OK;
}
int upx_sync(upx_plan* p) {
    if (!p) return fail(UPX_ERR_INVALID, "upx_sync: NULL plan");
    HIP_TRY(hipStreamSynchronize(p->stream));
    return UPX_OK;
}

// ---------------------------------------------------------------------------
// upx_process_device: one call = for every launch group, in list order,
//   (1) geometry  - how the group's frames are cut into streams / launch pairs / chunks: host arithmetic only
//                   (zoom_geometry, fused_geometry; the unfused path's is three lines inside launch_unfused),
//   (2) tables    - what that geometry needs on the device: stream tables, dealing tables, seam buffer
//                   (zoom_tables, zoom_deal_table, fused_tables; allocations and uploads, cached per geometry -
//                   a repeated call sends nothing; upx_plan_reserve runs (1) + (2) ahead of the first call),
//   (3) launch    - kernel arguments, events, launches (launch_unfused, zoom_launch, fused_launch; skipped on a dry run).
// ---------------------------------------------------------------------------
namespace {
// one upx_process_device call as every launch group sees it
struct Call {
    const float* d_stereo;
    float *d_c, *d_l, *d_r;
    int64_t t_in, own_len, t_out;
    bool dry;      // upx_plan_reserve: geometry + tables, nothing is launched and no buffer is touched
    bool timing;   // HIP events around the launches
    int slot;      // this call's slot of the event rings
};
// what one launch group covers: frames j < j_hi exist, hop-blocks m < m_hi are emitted; the first group of a call WRITES the
// planes, every later one reads, adds and writes them back - list order = the reference's float32 band sum
// ((0 + b0) + b1) + ... (center_extraction.py:508-511)
struct Range {
    long long j_hi, m_hi;
    bool first;
};

// the plan's stream-seam buffer holds at least `floats` (grows only; growing waits for the stream: older launches read it)
int ensure_seam(upx_plan* p, size_t floats) {
    if (floats <= p->seam_floats) return UPX_OK;
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (p->d_seam) HIP_TRY(hipFree(p->d_seam));
    p->d_seam = nullptr;
    p->seam_floats = 0;
    HIP_TRY(hipMalloc(&p->d_seam, floats * sizeof(float)));
    p->seam_floats = floats;
    return UPX_OK;
}
// room for s.h_m0 on the device (grows only)
int ensure_stream_table(upx_plan* p, BandState& s) {
    if (s.h_m0.size() <= s.m0_cap) return UPX_OK;
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (s.d_m0) HIP_TRY(hipFree(s.d_m0));
    s.d_m0 = nullptr;
    s.m0_cap = 0;
    HIP_TRY(hipMalloc(&s.d_m0, s.h_m0.size() * sizeof(int)));
    s.m0_cap = s.h_m0.size();
    s.h_m0_sent.clear();
    return UPX_OK;
}

// ---- unfused pipeline (upx_big.h): chunks of frames through the scratch ------------------------------------------------
int launch_unfused(upx_plan* p, BandState& s, const Call& c, const Range& r) {
    // a chunk transforms frames j0 .. j0+ch-1 (j0 odd): `halo` frames that only feed the first emitted blocks, `emit`
    // emitted blocks, one trailing pair-partner frame
    const int halo = (s.k - 1) | 1;
    int emit = (s.chunk_frames - halo - 1) & ~1;
    if ((long long)emit > r.m_hi) emit = (int)(r.m_hi + (r.m_hi & 1));
    if (emit < 2) emit = 2;
    const int ch = emit + halo + 1;
    upx::BigArgs a;
    a.in = reinterpret_cast<const upx::cf*>(c.d_stereo);
    a.out_c = c.d_c; a.out_l = c.d_l; a.out_r = c.d_r;
    a.w_a = s.d_wa; a.w_s = s.d_ws; a.gain = s.d_gain; a.tw_n = s.d_tw_n; a.tw_rows = s.d_tw;
    a.n_gain = s.n_gain; a.gain_stride = s.n / 2 + 1;
    a.z = p->d_scratch;
    a.y = a.z + (size_t)ch * s.n;
    a.yc = a.y + (size_t)ch * s.n;
    a.t_in = (int)c.t_in; a.t_out = (int)c.t_out;
    a.hop = s.hop; a.kf = s.k;
    a.j_lo = 0; a.j_hi = (int)r.j_hi; a.ch = ch;
    a.accumulate = r.first ? 0 : 1;
    a.tail = p->knob_big_tail;
    if (c.timing) HIP_TRY(hipEventRecord(s.ev0, p->stream));
    int n_chunks = 0;
    for (long long m0 = 0; m0 < r.m_hi; m0 += emit, ++n_chunks) {
        a.j0 = (int)m0 - halo;
        a.m0 = (int)m0;
        a.m1 = (int)(m0 + emit < r.m_hi ? m0 + emit : r.m_hi);
        if (!c.dry) s.big->chunk(a, p->stream);
    }
    if (c.timing) HIP_TRY(hipEventRecord(s.ev1, p->stream));
    s.last_wg = n_chunks;
    s.last_f = emit;
    return UPX_OK;
}

// ---- band-limited path (upx_zoom.h) ---------------------------------------------------------------------------------
// Launch pairs (analysis, synthesis) over consecutive runs of frame pairs (what the scratch holds), and the streams of
// each; the tables hold the first frame of every Ls/Rs stream and of every centre stream, launch after launch, + the end
// frame.  Streams cover frames -1 .. m_hi-1; stream tails go through the seam buffer.
struct ZoomLaunch {
    long long pair0, pair_end;
    int s0_lr, ns_lr, s0_c, ns_c;
    bool once;
};
struct ZoomGeometry {
    std::vector<ZoomLaunch> launches;
    std::vector<int> tab_lr, tab_c;
    long long n_lr = 0, n_c = 0;   // streams of either role over all launch pairs
    long long tail = 0;            // (K - 1) hop samples a stream leaves for its successor
    int groups = 1, res_s = 1;     // residue groups of a frame; synthesis workgroups a CU holds
};

void zoom_geometry(const upx_plan* p, const BandState& s, long long m_hi, ZoomGeometry& g) {
    const long long frames = m_hi + 1;
    const long long pairs_total = (frames + 1) / 2;
    const int cap = zoom_frames_cap(p, s.zoom_p);
    g.groups = s.zoom_d / s.zoom->rg;
    g.res_s = zoom_resident(s);
    g.tail = (long long)(s.k - 1) * s.hop;
    // n streams over n_pairs frame pairs, lengths within one pair of each other - except the stream that holds the
    // signal's first frame and the one that holds its last: they run the synthesis' signal-edge flavour (per-sample
    // checks on every transform: 15-25 % slower, scripts/phase_prof/zwgtime.hip) and get `edge` of the others' length
    const double edge = p->knob_zoom_edge_percent >= 50 && p->knob_zoom_edge_percent < 100 ? p->knob_zoom_edge_percent / 100.0 : 1.0;
    auto deal = [edge](std::vector<int>& tab, long long pair0, long long n_pairs, long long n, bool first, bool last,
                       long long min_pairs) {
        const double w_first = first && n >= 8 ? edge : 1.0, w_last = last && n >= 8 ? edge : 1.0;
        const double total = (double)(n - 2) + w_first + w_last;
        long long prev = 0;
        for (long long i = 0; i < n; ++i) {
            const double before = i == 0 ? 0.0 : w_first + (double)(i - 1);
            long long q = n >= 8 ? (long long)((double)n_pairs * before / total) : n_pairs * i / n;
            if (i > 0 && q < prev + min_pairs) q = prev + min_pairs;   // (rounding; a stream holds a tail)
            tab.push_back((int)(-1 + 2 * (pair0 + q)));
            prev = q;
        }
    };
    const bool once = p->knob_zoom_once != 0 && s.blocks_override <= 0 && p->knob_zoom_f <= 0;
    if (once) {
        // Every resident workgroup slot once (ZoomArgs): W workgroups per residue group, centre streams longer
        // than the Ls/Rs ones by the ratio of what a frame costs either role, so that all end together.
        const long long n_launch = (2 * pairs_total + cap - 1) / cap;
        const long long W = (long long)p->n_cu * g.res_s / g.groups;
        const double ratio = p->knob_zoom_c_cost > 0.05 && p->knob_zoom_c_cost < 1.0 ? p->knob_zoom_c_cost : 0.55;
        const long long min_pairs = (s.k + 1) / 2 > 8 ? (s.k + 1) / 2 : 8;   // a stream holds a tail; short ones are all prologue
        for (long long L = 0; L < n_launch; ++L) {
            const long long q0 = pairs_total * L / n_launch, q1 = pairs_total * (L + 1) / n_launch;
            const long long np = q1 - q0;
            long long n_c = (long long)std::llround((double)W * ratio / (1.0 + ratio)), n_lr = W - n_c;
            if (n_lr > np / min_pairs) n_lr = np / min_pairs;
            if (n_c > np / (2 * min_pairs)) n_c = np / (2 * min_pairs);
            if (n_lr < 1) n_lr = 1;
            if (n_c < 1) n_c = 1;
            g.launches.push_back(ZoomLaunch{q0, q1, (int)g.tab_lr.size(), (int)n_lr, (int)g.tab_c.size(), (int)n_c,
                                            (n_lr + n_c) * g.groups > (long long)p->n_cu * (g.res_s - 1)});
            deal(g.tab_lr, q0, np, n_lr, L == 0, L == n_launch - 1, min_pairs);
            deal(g.tab_c, q0, np, n_c, L == 0, L == n_launch - 1, min_pairs);
        }
    } else {
        // streams of F frames for both roles, about UPX_ZOOM_FILL x the slots of them (the scheme of rounds 1-2;
        // upx_plan_set_blocks_per_stream and UPX_ZOOM_F choose F)
        const long long want = zoom_streams_wanted(p, s);
        long long f = s.blocks_override > 0 ? s.blocks_override : (frames + want - 1) / want;
        if (s.blocks_override <= 0) {
            if (p->knob_zoom_f > 0) f = p->knob_zoom_f;
            else if (f < 16) f = 16;
            if (f > 48 && p->knob_zoom_f <= 0) f = 48;
        }
        if (f < s.k) f = s.k;
        f += f & 1;
        if (f > cap) f = cap;
        const long long n_streams = (frames + f - 1) / f, per_launch = cap / f;
        for (long long s0 = 0; s0 < n_streams; s0 += per_launch) {
            const long long ns = n_streams - s0 < per_launch ? n_streams - s0 : per_launch;
            g.launches.push_back(ZoomLaunch{s0 * f / 2, (s0 + ns) * f / 2, (int)s0, (int)ns, (int)s0, (int)ns, false});
            for (long long i = 0; i < ns; ++i) g.tab_lr.push_back((int)(-1 + (s0 + i) * f));
        }
        g.tab_c = g.tab_lr;
    }
    const int end_frame = (int)(-1 + 2 * g.launches.back().pair_end);
    g.tab_lr.push_back(end_frame);
    g.tab_c.push_back(end_frame);
    g.n_lr = (long long)g.tab_lr.size() - 1;
    g.n_c = (long long)g.tab_c.size() - 1;
}

// the two stream tables next to each other on the device, the seam buffer large enough
int zoom_tables(upx_plan* p, BandState& s, const ZoomGeometry& g) {
    s.h_m0 = g.tab_lr;
    s.h_m0.insert(s.h_m0.end(), g.tab_c.begin(), g.tab_c.end());
    if (int rc = ensure_stream_table(p, s)) return rc;
    if (s.h_m0 != s.h_m0_sent) {
        // in stream order behind the previous call's kernels, which read the old table; the source is pageable, so the
        // call returns when it has been read (tracks of unequal length change the geometry call after call: no
        // synchronisation here, the uploads and downloads of the neighbouring tracks keep running)
        HIP_TRY(hipMemcpyAsync(s.d_m0, s.h_m0.data(), s.h_m0.size() * sizeof(int), hipMemcpyHostToDevice, p->stream));
        s.h_m0_sent = s.h_m0;
    }
    return ensure_seam(p, (size_t)(2 * g.n_lr + g.n_c) * g.tail);
}

// The analysis fills every resident slot once and its older workgroups, which still end first, take more pairs
// (ZoomArgs::deal_rows): pair after pair to the workgroup that would end first, round r of the `res` dispatch rounds
// running at 1 - r age of the first round's speed; ties go to the older workgroup, so the counts never increase with l.
// Cached per (pairs of an XCD's share, workgroups per XCD, rounds).
int zoom_deal_table(upx_plan* p, BandState& s, long long np_xcd, long long per_xcd, int res, const BandState::Deal** out) {
    const long long key = np_xcd * 4096 + per_xcd * 8 + res;
    for (const auto& d : s.deals)
        if (d.key == key) {
            *out = &d;
            return UPX_OK;
        }
    const int n_l = (int)per_xcd, per_round = n_l / res;
    std::vector<int> count(n_l, 0);
    std::vector<double> cost(n_l);
    for (int l = 0; l < n_l; ++l) {
        const int r = l / per_round < res ? l / per_round : res - 1;
        cost[l] = 1.0 / (1.0 - r * p->knob_zoom_a_age / 100.0);
    }
    for (long long k = 0; k < np_xcd; ++k) {
        int best = 0;
        double t_best = 1e300;
        for (int l = 0; l < n_l; ++l) {
            const double t = (count[l] + 1) * cost[l];
            if (t < t_best - 1e-9) { t_best = t; best = l; }
        }
        ++count[best];
    }
    BandState::Deal d;
    d.key = key;
    d.rows = count[n_l - 1];
    std::vector<int> tab(2 * (size_t)n_l);
    int behind = 0;
    for (int l = 0; l < n_l; ++l) {
        tab[2 * l] = behind;
        tab[2 * l + 1] = count[l] - d.rows;
        behind += tab[2 * l + 1];
    }
    if (s.deals.size() >= 8) {   // (geometries come and go: start over)
        HIP_TRY(hipStreamSynchronize(p->stream));
        for (auto& old : s.deals) (void)hipFree(old.d_tab);
        s.deals.clear();
    }
    HIP_TRY(hipMalloc(&d.d_tab, tab.size() * sizeof(int)));
    HIP_TRY(hipMemcpy(d.d_tab, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
    s.deals.push_back(d);
    *out = &s.deals.back();
    return UPX_OK;
}

// kernel arguments of launch pair L of a group (everything both kernels read)
int zoom_args(upx_plan* p, BandState& s, const Call& c, const Range& r, const ZoomGeometry& g, const ZoomLaunch& L,
              upx::ZoomArgs& a, long long* per_xcd_out) {
    std::memset(&a, 0, sizeof a);
    a.in = reinterpret_cast<const upx::cf*>(c.d_stereo);
    a.out_c = c.d_c; a.out_l = c.d_l; a.out_r = c.d_r;
    a.w_a = s.d_wa; a.w_s = s.d_ws; a.gain = s.d_gain; a.tw = s.d_tw; a.ramp = s.d_ramp;
    a.seam = p->d_seam;
    a.seam_c = p->d_seam + (size_t)2 * g.n_lr * g.tail;
    a.stream_m0 = s.d_m0;
    a.stream_m0_c = s.d_m0 + g.n_lr + 1;
    a.n = s.n; a.d = s.zoom_d; a.hop = s.hop;
    a.t_in = (int)c.t_in; a.t_out = (int)c.t_out;
    a.j_lo = 0; a.j_hi = (int)r.j_hi; a.m_lo = 0; a.m_hi = (int)r.m_hi;
    a.blocks_per_stream = g.tab_lr[1] - g.tab_lr[0];
    a.n_gain = s.n_gain; a.gain_stride = s.n / 2 + 1;
    a.accumulate = r.first ? 0 : 1;
    const int res_a = zoom_resident(s, true);
    const long long slots = (long long)p->n_cu * res_a;
    a.stream0 = L.s0_lr; a.ns_lr = L.ns_lr;
    a.stream0_c = L.s0_c; a.ns_c = L.ns_c;
    a.f0 = (int)(-1 + 2 * L.pair0);
    a.pair0 = (int)L.pair0;
    a.pair_end = (int)L.pair_end;
    const long long pairs = L.pair_end - L.pair0;
    a.y = p->d_zoom;
    a.yc = a.y + (size_t)(2 * pairs) * s.zoom_p;
    // analysis grid: every resident slot once, 8 x (workgroups per XCD label), see zoom_analysis_program
    long long per_xcd = (slots + 7) / 8;
    if (per_xcd > (pairs + 7) / 8) per_xcd = (pairs + 7) / 8;
    if (per_xcd < 1) per_xcd = 1;
    a.pairs_per_wg = (int)per_xcd;
    // its workgroups take the top priority in turn (ZoomArgs) ...
    a.prio_split = (p->knob_prio_young > 0 && res_a >= 2 && res_a <= 4 && 8 * per_xcd >= (long long)p->n_cu * res_a)
                       ? p->n_cu : 0;
    a.prio_rounds = res_a;
    // ... and the older ones take more pairs
    a.deal_rows = 0;
    const long long np_xcd = (pairs + 7) / 8;      // pairs of an XCD's share (zoom_analysis_program)
    if (a.prio_split > 0 && p->knob_zoom_a_age > 0 && p->knob_zoom_a_age < 40 && np_xcd >= 4 * per_xcd) {
        const BandState::Deal* deal = nullptr;
        if (int rc = zoom_deal_table(p, s, np_xcd, per_xcd, res_a, &deal)) return rc;
        a.deal_rows = deal->rows;
        a.deal_tab = deal->d_tab;
    }
    // ... and so does the synthesis when its streams were cut for that
    a.prio_split_s = (p->knob_prio_young > 0 && L.once && g.res_s >= 2 && g.res_s <= 4) ? p->n_cu : 0;
    a.prio_rounds_s = g.res_s;
    *per_xcd_out = per_xcd;
    return UPX_OK;
}

int zoom_launch(upx_plan* p, BandState& s, const Call& c, const Range& r, const ZoomGeometry& g) {
    upx::ZoomArgs a;
    if (c.timing) HIP_TRY(hipEventRecord(s.ev0, p->stream));
    const bool split = c.timing && 2 * (long long)g.launches.size() - 1 <= kMidEvents;
    int n_mid = 0;
    hipEvent_t* mid = s.ring_mid.data() + (size_t)c.slot * kMidEvents;
    int n_launches = 0;
    for (const ZoomLaunch& L : g.launches) {
        long long per_xcd = 1;
        if (int rc = zoom_args(p, s, c, r, g, L, a, &per_xcd)) return rc;
        if (n_launches == 0) {
            s.fill_wg = (int)((L.ns_lr + L.ns_c) * g.groups);
            s.fill_slots = p->n_cu * g.res_s;
            s.fill_wg_a = (int)(8 * per_xcd);
            s.fill_slots_a = p->n_cu * zoom_resident(s, true);
        }
        if (split && n_launches > 0) HIP_TRY(hipEventRecord(mid[n_mid++], p->stream));
        if (!c.dry) s.zoom_a->analysis(a, (int)(8 * per_xcd), p->stream);
        if (split) HIP_TRY(hipEventRecord(mid[n_mid++], p->stream));
        if (!c.dry) s.zoom->synthesis(a, g.groups, p->stream);
        ++n_launches;
    }
    if (c.timing) s.ring_mid_n[c.slot] = n_mid;
    if (!c.dry && (g.n_lr > 1 || g.n_c > 1)) {
        if (seam_vec_ok(p, c.d_c, c.d_l, c.d_r, s.hop) && g.n_lr + g.n_c <= 65535)
            hipLaunchKernelGGL(upx_zoom_seam_add4_kernel, dim3((unsigned)((g.tail / 4 + 255) / 256), (unsigned)(g.n_lr + g.n_c)),
                               dim3(256), 0, p->stream, a, (int)g.n_lr, (int)g.n_c, (int)g.tail);
        else
            hipLaunchKernelGGL(upx_zoom_seam_add_kernel, dim3(grid_for((g.n_lr + g.n_c) * g.tail)), dim3(256), 0,
                               p->stream, a, (int)g.n_lr, (int)g.n_c, (int)g.tail);
    }
    if (c.timing) HIP_TRY(hipEventRecord(s.ev1, p->stream));
    s.last_wg = (int)((g.n_lr + g.n_c) * g.groups);
    s.last_f = g.tab_lr[1] - g.tab_lr[0];
    return UPX_OK;
}

// ---- fused streaming kernel (upx_core.h) ------------------------------------------------------------------------------
struct FusedGeometry {
    long long f = 0;          // frames per stream (of the interior streams when `uneven`)
    long long n_streams = 0, n_wg = 0;
    long long slots = 0;      // workgroups resident at once
    long long tail = 0;
    bool uneven = false;      // streams of unequal length: m0 holds every stream's first frame (+ the end)
    std::vector<int> m0;
};

// blocks per stream: fill every resident workgroup slot once; even (whole frame pairs), at least UPX_MIN_STREAM_FRAMES and
// at least K (a stream's tail must end inside the next stream).  Streams cover frames m_lo-1 .. m_hi-1.
void fused_geometry(const upx_plan* p, const BandState& s, long long m_hi, FusedGeometry& g) {
    const long long target_streams = max_auto_streams(p, s);
    long long f = s.blocks_override > 0 ? s.blocks_override : (m_hi + 1 + target_streams - 1) / target_streams;
    if (s.blocks_override <= 0 && f < p->knob_min_stream_frames) f = p->knob_min_stream_frames;
    if (f < s.k) f = s.k;
    f += f & 1;
    g.n_streams = (m_hi + 1 + f - 1) / f;
    g.n_wg = (g.n_streams + s.kern->g - 1) / s.kern->g;
    g.tail = (long long)(s.k - 1) * s.hop;
    // A launch that fills the machine: the first and the last workgroups run the signal-edge flavour (~12 % slower per
    // frame; measured per workgroup, scripts/phase_prof/wgtime.hip) and the launch used to wait for them.  They get
    // streams of edge_percent of the others' length; the others grow by what that frees (stream table, BandArgs).
    const int G = s.kern->g;
    g.slots = target_streams / G;
    const long long slots = g.slots, total = m_hi + 1;         // frames -1 .. m_hi-1
    if (s.blocks_override <= 0 && p->knob_edge_percent < 100 && p->knob_edge_percent >= 50 && slots >= 8 &&
        total >= slots * G * 16) {
        const double share = p->knob_edge_percent / 100.0;
        long long fu = (long long)std::ceil((double)total / ((double)G * ((double)slots - 2.0 * (1.0 - share))));
        fu += fu & 1;
        for (int attempt = 0; attempt < 8 && !g.uneven; ++attempt, fu += 2) {
            long long fe = (long long)(share * (double)fu);
            fe -= fe & 1;
            if (fe < s.k + (s.k & 1) || fe < 2) break;
            std::vector<long long> len{fe};
            long long remaining = total - G * fe;
            while (remaining > G * fe) {
                len.push_back(fu);
                remaining -= G * fu;
            }
            if (remaining > 0) {
                long long last = (remaining + G - 1) / G;
                last += last & 1;
                if (last < s.k + (s.k & 1)) last = s.k + (s.k & 1);
                len.push_back(last);
            }
            if ((long long)len.size() > slots) continue;        // one workgroup too many: longer streams
            g.m0.assign(len.size() * G + 1, 0);
            g.m0[0] = -1;
            for (size_t w = 0; w < len.size(); ++w)
                for (int q = 0; q < G; ++q) g.m0[w * G + q + 1] = g.m0[w * G + q] + (int)len[w];
            g.n_wg = (long long)len.size();
            g.n_streams = g.n_wg * G;
            f = fu;
            g.uneven = true;
        }
    }
    g.f = f;
}

int fused_tables(upx_plan* p, BandState& s, const FusedGeometry& g) {
    if (g.uneven) {
        // the table goes to the device on the plan's stream, ordered with the launch (the host copy stays in the plan)
        s.h_m0 = g.m0;
        if (int rc = ensure_stream_table(p, s)) return rc;
        if (s.h_m0 != s.h_m0_sent) {   // (a repeated call on the same geometry - every step of a benchmark - sends nothing)
            HIP_TRY(hipMemcpyAsync(s.d_m0, s.h_m0.data(), s.h_m0.size() * sizeof(int), hipMemcpyHostToDevice, p->stream));
            HIP_TRY(hipStreamSynchronize(p->stream));   // the host copy may change on the next call
            s.h_m0_sent = s.h_m0;
        }
    }
    return ensure_seam(p, (size_t)g.n_wg * s.kern->g * 3 * g.tail);
}

int fused_launch(upx_plan* p, BandState& s, const Call& c, const Range& r, const FusedGeometry& g) {
    upx::BandArgs a;
    std::memset(&a, 0, sizeof a);
    a.in = reinterpret_cast<const upx::cf*>(c.d_stereo);
    a.out_c = c.d_c; a.out_l = c.d_l; a.out_r = c.d_r;
    a.w_a = s.d_wa; a.w_s = s.d_ws; a.gain = s.d_gain; a.tw = s.d_tw;
    a.n_gain = s.n_gain; a.gain_stride = s.n / 2 + 1;
    a.t_in = (int)c.t_in; a.t_out = (int)c.t_out;
    a.j_lo = 0; a.j_hi = (int)r.j_hi; a.m_lo = 0; a.m_hi = (int)r.m_hi;
    a.blocks_per_stream = (int)g.f;
    a.stream_m0 = g.uneven ? s.d_m0 : nullptr;
    a.accumulate = r.first ? 0 : 1;
    a.seam = p->d_seam;
    // one-wave workgroups that fill the machine: the first `SIMDs` of them are the older wave of their SIMD
    // (BandArgs::prio_split) ... and two-wave workgroups (N = 2048: four per CU) run at four speeds by dispatch round; the
    // rounds take the top priority in turn (prio_split < 0: workgroups per round)
    a.prio_split = 0;
    if (p->knob_prio_young > 0 && p->knob_prio_young < 4) {
        if (s.kern->wg == 64 && g.n_wg > 4LL * p->n_cu) a.prio_split = 4 * p->n_cu;
        else if (s.kern->wg == 128 && g.n_wg > 2LL * p->n_cu) a.prio_split = -p->n_cu;
    }
    a.prio_young = p->knob_prio_young;
    bool seam_inside = false;
#if defined(UPX_EXPERIMENTS)
    // stream seams inside the launch (seam_epilogue; rejected in round 5: the agent-scope release / acquire it needs costs
    // more than the launch it saves): UPX_SEAM_INKERNEL = 1 always, 2 when the launch does not fill the chip, 0 never
    seam_inside = g.n_streams > 1 && p->d_pair_cnt && g.n_wg <= (long long)p->pair_cnt_n &&
                  (p->knob_seam_inkernel == 1 || (p->knob_seam_inkernel == 2 && g.n_wg <= g.slots));
    a.pair_cnt = seam_inside ? p->d_pair_cnt : nullptr;
    a.n_streams = (int)g.n_streams;
#endif
    s.last_wg = (int)g.n_wg;
    s.last_f = (int)g.f;
    s.last_streams = (int)g.n_streams;
    s.last_uneven = g.uneven;
    s.fill_wg = (int)g.n_wg;
    s.fill_slots = (int)g.slots;
    s.fill_wg_a = s.fill_slots_a = 0;
    if (c.timing) HIP_TRY(hipEventRecord(s.ev0, p->stream));
    if (!c.dry) s.kern->launch(a, (int)g.n_wg, p->stream);
    if (!c.dry && g.n_streams > 1 && !seam_inside) {
        if (seam_vec_ok(p, c.d_c, c.d_l, c.d_r, s.hop) && g.n_streams <= 65535)
            hipLaunchKernelGGL(upx_stream_seam_add4_kernel, dim3((unsigned)((g.tail / 4 + 255) / 256), (unsigned)g.n_streams), dim3(256), 0,
                               p->stream, a, (int)g.n_streams, (int)g.tail, s.hop);
        else
            hipLaunchKernelGGL(upx_stream_seam_add_kernel, dim3(grid_for(g.n_streams * g.tail)), dim3(256), 0, p->stream, a,
                               (int)g.n_streams, (int)g.tail, s.hop);
    }
    if (c.timing) HIP_TRY(hipEventRecord(s.ev1, p->stream));
    return UPX_OK;
}

// what the band / launch reports read (upx_plan_band_info, _band_fill): kept across calls that must not show in them -
// upx_plan_reserve's dry run and its warm-up call on a few frames of silence
struct Reports {
    std::vector<std::tuple<int, int, int, int, int, int>> v;
    void take(const upx_plan* p) {
        for (const auto& s : p->bands) v.emplace_back(s.last_wg, s.last_f, s.fill_wg, s.fill_slots, s.fill_wg_a, s.fill_slots_a);
    }
    void restore(upx_plan* p) const {
        for (size_t i = 0; i < p->bands.size() && i < v.size(); ++i)
            std::tie(p->bands[i].last_wg, p->bands[i].last_f, p->bands[i].fill_wg, p->bands[i].fill_slots, p->bands[i].fill_wg_a,
                     p->bands[i].fill_slots_a) = v[i];
    }
};

// the launch groups of a plan in the order they run
void launch_order(const upx_plan* p, std::vector<size_t>& order) {
    for (size_t b = 0; b < p->bands.size(); ++b)
        if (p->bands[b].group_size > 0) order.push_back(b);      // (members are carried by their group leader's launch)
#if defined(UPX_EXPERIMENTS)
    // UPX_FIRST_BAND = k launches the group that carries band k first; UPX_BAND_ROTATE starts that many groups into the
    // list: the same sum in another float32 association (round-4 / round-5 A/Bs; the product library cannot do this)
    const int fb = p->knob_first_band;
    if (fb >= 0 && fb < (int)p->bands.size()) {
        const size_t lead = (size_t)p->bands[(size_t)fb].group_leader;
        auto it = std::find(order.begin(), order.end(), lead);
        if (it != order.end()) std::rotate(order.begin(), it, it + 1);
    }
    if (p->knob_band_rotate > 0 && !order.empty())
        std::rotate(order.begin(), order.begin() + (p->knob_band_rotate % (int)order.size()), order.end());
#endif
}
}   // namespace

int upx_process_device(upx_plan* p, const float* d_stereo, int64_t t_in, int64_t own_len, float* d_c, float* d_l,
                       float* d_r, int64_t t_out) {
    if (!p || t_in < 0 || own_len < 0 || t_out < 0) return fail(UPX_ERR_INVALID, "upx_process_device: bad argument");
    if (t_out == 0) return UPX_OK;
    if (t_in >= (1LL << 29) || t_out >= (1LL << 29))
        return fail(UPX_ERR_INVALID, "upx_process_device: at most 2^29-1 samples per call (shard longer signals)");
    // upx_plan_reserve: everything this call would do on the host - launch geometry, stream tables, seam / scratch sizes and
    // their allocations and uploads - without the launches (no buffer is touched, the pointers may be NULL)
    const bool dry = p->dry_run;
    if (!dry && (!d_stereo || !d_c || !d_l || !d_r)) return fail(UPX_ERR_INVALID, "upx_process_device: NULL buffer");
    HIP_TRY(hipSetDevice(p->device));
    if (dry && (t_in == 0 || own_len == 0)) return UPX_OK;
    if (t_in == 0 || own_len == 0) {   // nothing to transform: the result is silence
        HIP_TRY(hipMemsetAsync(d_c, 0, (size_t)t_out * sizeof(float), p->stream));
        HIP_TRY(hipMemsetAsync(d_l, 0, (size_t)t_out * sizeof(float), p->stream));
        HIP_TRY(hipMemsetAsync(d_r, 0, (size_t)t_out * sizeof(float), p->stream));
        for (auto& s : p->bands) s.last_wg = 0;
        return UPX_OK;
    }
    const Call c{d_stereo, d_c, d_l, d_r, t_in, own_len, t_out, dry, p->timing && !dry, (int)(p->timed_calls % kTimingSlots)};
    // a dry run reports nothing: the band / launch reports (last_wg, fill_*) keep describing the last REAL call
    Reports kept;
    if (dry) kept.take(p);
    for (auto& s : p->bands) {
        s.last_wg = 0;
        if (!c.timing) continue;     // (ev0 / ev1 stay the events of the last TIMED call: upx_plan_band_times_ms reads them)
        s.ev0 = s.ring0[c.slot];
        s.ev1 = s.ring1[c.slot];
        s.ring_used[c.slot] = 0;
    }
    std::vector<size_t> order;
    launch_order(p, order);
    int rc = UPX_OK;
    for (size_t b : order) {
        BandState& s = p->bands[b];
        Range r;
        r.first = b == order.front();
        r.j_hi = (own_len + s.hop - 1) / s.hop;                          // frames with j*hop < own_len
        const long long m_all = (t_out + s.hop - 1) / s.hop;            // hop-blocks that intersect [0, t_out)
        r.m_hi = r.j_hi + s.k - 1 < m_all ? r.j_hi + s.k - 1 : m_all;
        if (r.first) r.m_hi = m_all;                                    // the first launch initialises every output sample
        if (r.j_hi > 0x7fffffffLL || m_all > 0x7fffffffLL) { rc = fail(UPX_ERR_INVALID, "signal too long for int32 frame index"); break; }
        if (r.m_hi <= 0) continue;
        if (s.big) {
            rc = launch_unfused(p, s, c, r);
        } else if (s.zoom) {
            ZoomGeometry g;
            zoom_geometry(p, s, r.m_hi, g);
            rc = zoom_tables(p, s, g);
            if (rc == UPX_OK) rc = zoom_launch(p, s, c, r, g);
        } else {
            FusedGeometry g;
            fused_geometry(p, s, r.m_hi, g);
            rc = fused_tables(p, s, g);
            if (rc == UPX_OK) rc = fused_launch(p, s, c, r, g);
        }
        if (rc != UPX_OK) break;
    }
    if (dry) {
        kept.restore(p);
        return rc;
    }
    if (rc != UPX_OK) return rc;
    HIP_TRY(hipGetLastError());
    if (c.timing) {
        p->timed_once = true;
        for (auto& s : p->bands) {
            s.ring_used[c.slot] = s.last_wg > 0;
            s.timed_wg = s.last_wg;
        }
        p->timed_calls += 1;
    }
    return UPX_OK;
}

int upx_plan_reserve(upx_plan* p, int64_t t_in, int64_t own_len, int64_t t_out) {
    if (!p || t_in < 0 || own_len < 0 || t_out < 0) return fail(UPX_ERR_INVALID, "upx_plan_reserve: bad argument");
    if (t_out == 0) return UPX_OK;
    HIP_TRY(hipSetDevice(p->device));
    if (!p->warmed) {
        // One tiny REAL call through every kernel of the plan, on silence: the runtime loads a kernel's code object and sets
        // the function up on its first launch (tens of microseconds each, ten kernels in a six-band plan) - paid here, not by
        // the caller's first step.  Then the dry run below prepares the tables and buffers of the shape asked for.
        int n_max = 0;
        for (const auto& s : p->bands) n_max = s.n > n_max ? s.n : n_max;
        const int64_t n = 8LL * n_max;
        float* tmp = nullptr;
        HIP_TRY(hipMalloc(&tmp, (size_t)n * 5 * sizeof(float)));
        hipError_t e = hipMemsetAsync(tmp, 0, (size_t)n * 2 * sizeof(float), p->stream);
        const bool timing = p->timing;
        Reports kept;                    // the warm-up is not a call of the caller's: it leaves no trace in the reports ...
        kept.take(p);
        p->timing = false;               // ... and none in the event rings (timed_once / timed_calls are untouched by untimed calls)
        int rc = e == hipSuccess ? upx_process_device(p, tmp, n, n, tmp + 2 * n, tmp + 3 * n, tmp + 4 * n, n) : UPX_ERR_HIP;
        p->timing = timing;
        kept.restore(p);
        (void)hipStreamSynchronize(p->stream);
        (void)hipFree(tmp);
        if (rc != UPX_OK) return rc;
        p->warmed = true;
    }
    p->dry_run = true;
    const int rc = upx_process_device(p, nullptr, t_in, own_len, nullptr, nullptr, nullptr, t_out);
    p->dry_run = false;
    if (rc != UPX_OK) return rc;
    HIP_TRY(hipStreamSynchronize(p->stream));   // the table uploads have landed: the first real call finds nothing left to do
    return UPX_OK;
}

// Time-shard geometry shared by all bands: cuts sit on multiples of 2 hop_max (so local frame
// parity equals global frame parity in every band), a shard reads `spill` samples past its end and its output
// spills the same number of samples into the next shard.  Returns false when the hops do not share a grid.
static bool shard_geometry(const upx_plan* p, int64_t* grid, int64_t* spill) {
    int64_t hop_max = 0, sp = 0;
    for (const auto& s : p->bands) {
        if (s.hop > hop_max) hop_max = s.hop;
        if (s.n - s.hop > sp) sp = s.n - s.hop;
    }
    for (const auto& s : p->bands)
        if (hop_max % s.hop) return false;
    *grid = 2 * hop_max;
    *spill = sp;
    return true;
}

namespace {
// One work item of a streamed host call: a chunk of a track (or a whole short track).
struct PipeItem {
    const float* src = nullptr;          // host stereo, first sample of the item
    // upx_process_lr: the caller's own arrays go up as they are and are cast / interleaved on the device.  fmt =
    // UPX_SAMPLE_F32 / _F64; src_r == nullptr: `src` is interleaved [t_in][2] of fmt; else `src` / `src_r` are the
    // contiguous left / right arrays.  (plain float32 interleaved input: fmt = F32, src_r = nullptr, no convert pass)
    const void* src_r = nullptr;
    int fmt = UPX_SAMPLE_F32;
    int64_t t_in = 0, own = 0, t_out = 0;   // samples uploaded, samples whose frames are owned, output plane length
    float* dst[3] = {nullptr, nullptr, nullptr};   // host planes at the item's first owned sample
    int64_t seam = 0;                    // > 0: add this many spill samples of the previous item (same track) ...
    int64_t prev_own = 0;                // ... found at this offset of the previous item's planes
};

// Cuts one track into items exactly as upx_process does on its own (so a batch equals per-track calls bit for
// bit): chunks of `chunk` owned samples on the shard grid when the track is long enough, else one item.
int items_of_track(const upx_plan* p, const float* stereo, int64_t n, float* out_c, float* out_l, float* out_r,
                   int64_t chunk, bool force_chunks, std::vector<PipeItem>& items, const void* right = nullptr,
                   int fmt = UPX_SAMPLE_F32) {
    if (n == 0) return UPX_OK;
    // bytes from one sample of `stereo` (and of `right`) to the next
    const size_t elem = fmt == UPX_SAMPLE_F64 ? 8 : 4, step = right ? elem : 2 * elem;
    const size_t first_item = items.size();
    int64_t grid = 0, spill = 0;
    const bool gridded = shard_geometry(p, &grid, &spill);
    bool cut = chunk > 0 && gridded && (force_chunks || (n >= 2 * chunk && chunk >= 4 * spill));
    if (!cut && n >= (1LL << 29)) {
        if (!gridded) return fail(UPX_ERR_INVALID, "signals of 2^29 samples or more need hops that share a shard grid");
        chunk = 1LL << 26;
        cut = true;
    }
    if (!cut) {
        PipeItem it;
        it.src = stereo; it.t_in = n; it.own = n; it.t_out = n;
        it.src_r = right; it.fmt = fmt;
        it.dst[0] = out_c; it.dst[1] = out_l; it.dst[2] = out_r;
        items.push_back(it);
        return UPX_OK;
    }
    // owned samples per chunk: a multiple of the grid, at least the spill (a seam then ends inside the next chunk)
    int64_t own = (chunk + grid - 1) / grid * grid;
    if (own < spill) own = (spill + grid - 1) / grid * grid;
    if (own + spill >= (1LL << 29)) return fail(UPX_ERR_INVALID, "chunk too long (2^29 samples per launch)");
    const int64_t n_chunks = (n + own - 1) / own;
    for (int64_t i = 0; i < n_chunks; ++i) {
        const int64_t start = i * own;
        const bool last = i == n_chunks - 1;
        PipeItem it;
        it.src = reinterpret_cast<const float*>(reinterpret_cast<const char*>(stereo) + (size_t)start * step);
        it.src_r = right ? reinterpret_cast<const char*>(right) + (size_t)start * step : nullptr;
        it.fmt = fmt;
        it.own = last ? n - start : own;
        it.t_in = n - start < it.own + spill ? n - start : it.own + spill;
        it.t_out = last ? it.own : it.own + spill;
        it.dst[0] = out_c + start; it.dst[1] = out_l + start; it.dst[2] = out_r + start;
        if (i > 0) {
            it.seam = spill < it.t_out ? spill : it.t_out;
            it.prev_own = own;
        }
        items.push_back(it);
    }
    (void)first_item;
    return UPX_OK;
}

// Runs the items through the device: the upload of item i+1, the kernels of item i and the download of item i-1
// overlap (two copy streams + the completer thread of upx::run_pipeline).
int run_items(upx_plan* p, const std::vector<PipeItem>& items) {
    if (items.empty()) return UPX_OK;
    HIP_TRY(hipSetDevice(p->device));
    if (!p->s_h2d) {
        HIP_TRY(hipStreamCreateWithFlags(&p->s_h2d, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&p->s_d2h, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            HIP_TRY(hipEventCreateWithFlags(&p->ev_h2d[i], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&p->ev_comp[i], hipEventDisableTiming));
        }
    }
    int64_t cap_in = 0, cap_out = 0;
    for (const auto& it : items) {
        if (it.t_in > cap_in) cap_in = it.t_in;
        if (it.t_out > cap_out) cap_out = it.t_out;
    }
    const size_t in_floats = (size_t)cap_in * 2, out_floats = (size_t)cap_out * 3;
    // items whose samples are cast / interleaved on the device land in a raw buffer first (two sets like the others)
    size_t raw_bytes = 0;
    for (const auto& it : items)
        if (it.src_r || it.fmt != UPX_SAMPLE_F32) {
            const size_t b = (size_t)it.t_in * 2 * (it.fmt == UPX_SAMPLE_F64 ? 8 : 4);
            if (b > raw_bytes) raw_bytes = b;
        }
    if (raw_bytes > p->pipe_raw_bytes) {
        HIP_TRY(hipDeviceSynchronize());
        for (int i = 0; i < 2; ++i) {
            if (p->d_pipe_raw[i]) HIP_TRY(hipFree(p->d_pipe_raw[i]));
            p->d_pipe_raw[i] = nullptr;
        }
        p->pipe_raw_bytes = 0;
        for (int i = 0; i < (items.size() > 1 ? 2 : 1); ++i) {
            hipError_t e = hipMalloc(&p->d_pipe_raw[i], raw_bytes);
            if (e != hipSuccess) return fail(UPX_ERR_NOMEM, "hipMalloc(%zu) for a work item's raw samples: %s", raw_bytes, hipGetErrorString(e));
        }
        p->pipe_raw_bytes = raw_bytes;
    } else if (raw_bytes > 0 && items.size() > 1 && !p->d_pipe_raw[1]) {
        hipError_t e = hipMalloc(&p->d_pipe_raw[1], p->pipe_raw_bytes);
        if (e != hipSuccess) return fail(UPX_ERR_NOMEM, "hipMalloc(%zu) for a work item's raw samples: %s", p->pipe_raw_bytes, hipGetErrorString(e));
    }
    // two rotating buffer sets, grown on demand and never shrunk; a call with ONE work item (a short signal, or
    // UPX_STREAM_CHUNK=0) only touches - and only ever allocates - the first.  Both sets share one plane pitch.
    const int n_sets = items.size() > 1 ? 2 : 1;
    bool grow = in_floats > p->pipe_in_floats || out_floats > p->pipe_out_floats;
    for (int i = 0; i < n_sets; ++i) grow |= !p->d_pipe_in[i] || !p->d_pipe_out[i];
    if (grow) {
        HIP_TRY(hipDeviceSynchronize());
        const size_t want_in = in_floats > p->pipe_in_floats ? in_floats : p->pipe_in_floats;
        const size_t want_out = out_floats > p->pipe_out_floats ? out_floats : p->pipe_out_floats;
        const bool resize = want_in != p->pipe_in_floats || want_out != p->pipe_out_floats;
        for (int i = 0; i < 2; ++i) {
            const bool have = p->d_pipe_in[i] && p->d_pipe_out[i];
            if (have && !resize) continue;
            if (p->d_pipe_in[i]) HIP_TRY(hipFree(p->d_pipe_in[i]));
            if (p->d_pipe_out[i]) HIP_TRY(hipFree(p->d_pipe_out[i]));
            p->d_pipe_in[i] = p->d_pipe_out[i] = nullptr;
            if (i >= n_sets && !have) continue;   // a second set that has never been needed stays unallocated
            hipError_t e = hipMalloc(&p->d_pipe_in[i], (want_in ? want_in : 2) * sizeof(float));
            if (e == hipSuccess) e = hipMalloc(&p->d_pipe_out[i], (want_out ? want_out : 3) * sizeof(float));
            if (e != hipSuccess) {
                p->pipe_in_floats = p->pipe_out_floats = 0;
                return fail(UPX_ERR_NOMEM, "hipMalloc for a %lld-sample work item: %s", (long long)cap_out, hipGetErrorString(e));
            }
        }
        p->pipe_in_floats = want_in;
        p->pipe_out_floats = want_out;
    }
    const size_t plane = p->pipe_out_floats / 3;
#if defined(UPX_TEST_HOOKS)   // never in the product library: a test build fails the download of this item
    const char* inject = knob("UPX_TEST_FAIL_DOWNLOAD");
    const int64_t inject_at = inject ? std::atoll(inject) : -1;
#else
    const int64_t inject_at = -1;
#endif

    auto submit = [&](int64_t i, std::string& msg) -> int {
        const PipeItem& it = items[(size_t)i];
        const int b = (int)(i & 1);
        const bool raw = it.src_r || it.fmt != UPX_SAMPLE_F32;
        const size_t elem = it.fmt == UPX_SAMPLE_F64 ? 8 : 4;
        hipError_t e;
        if (!raw) {
            e = hipMemcpyAsync(p->d_pipe_in[b], it.src, (size_t)it.t_in * 2 * sizeof(float), hipMemcpyHostToDevice, p->s_h2d);
        } else if (!it.src_r) {
            e = hipMemcpyAsync(p->d_pipe_raw[b], it.src, (size_t)it.t_in * 2 * elem, hipMemcpyHostToDevice, p->s_h2d);
        } else {
            char* d = static_cast<char*>(p->d_pipe_raw[b]);
            e = hipMemcpyAsync(d, it.src, (size_t)it.t_in * elem, hipMemcpyHostToDevice, p->s_h2d);
            if (e == hipSuccess) e = hipMemcpyAsync(d + (size_t)it.t_in * elem, it.src_r, (size_t)it.t_in * elem, hipMemcpyHostToDevice, p->s_h2d);
        }
        if (e == hipSuccess) e = hipEventRecord(p->ev_h2d[b], p->s_h2d);
        if (e == hipSuccess) e = hipStreamWaitEvent(p->stream, p->ev_h2d[b], 0);
        if (e == hipSuccess && raw) {
            // the cast is the host's: round to nearest even, beyond the float32 range to infinity (numpy's astype)
            hipLaunchKernelGGL(upx_samples_to_stereo_kernel, dim3(grid_for(it.t_in)), dim3(256), 0, p->stream, p->d_pipe_raw[b],
                               it.fmt == UPX_SAMPLE_F64 ? 1 : 0, it.src_r ? 1 : 0, (long long)it.t_in,
                               reinterpret_cast<float2*>(p->d_pipe_in[b]));
            e = hipGetLastError();
        }
        if (e != hipSuccess) {
            msg = std::string("upload of a work item failed: ") + hipGetErrorString(e);
            return UPX_ERR_HIP;
        }
        float* o = p->d_pipe_out[b];
        int rc = upx_process_device(p, p->d_pipe_in[b], it.t_in, it.own, o, o + plane, o + 2 * plane, it.t_out);
        if (rc == UPX_OK && it.seam > 0) {
            const float* q = p->d_pipe_out[b ^ 1];
            rc = upx_seam_add_local(p, q, q + plane, q + 2 * plane, it.prev_own, o, o + plane, o + 2 * plane, it.seam);
        }
        if (rc != UPX_OK) {
            msg = g_err;
            return rc;
        }
        e = hipEventRecord(p->ev_comp[b], p->stream);
        if (e != hipSuccess) {
            msg = std::string("hipEventRecord: ") + hipGetErrorString(e);
            return UPX_ERR_HIP;
        }
        return UPX_OK;
    };
    auto complete = [&](int64_t i, std::string& msg) -> int {
        const PipeItem& it = items[(size_t)i];
        const int b = (int)(i & 1);
        if (i == 0) (void)hipSetDevice(p->device);   // first call on the completer thread
        hipError_t e = hipEventSynchronize(p->ev_comp[b]);
        for (int k = 0; k < 3 && e == hipSuccess; ++k)
            e = hipMemcpyAsync(it.dst[k], p->d_pipe_out[b] + k * plane, (size_t)it.own * sizeof(float), hipMemcpyDeviceToHost, p->s_d2h);
        if (e == hipSuccess) e = hipStreamSynchronize(p->s_d2h);
        if (e == hipSuccess && i == inject_at) e = hipErrorUnknown;
        if (e != hipSuccess) {
            msg = std::string("download of a work item failed: ") + hipGetErrorString(e);
            return UPX_ERR_HIP;
        }
        return UPX_OK;
    };
    std::string err;
    const int rc = upx::run_pipeline((int64_t)items.size(), submit, complete, err);
    (void)hipStreamSynchronize(p->stream);
    (void)hipStreamSynchronize(p->s_d2h);
    if (rc != UPX_OK) return fail(rc, "%s", err.c_str());
    return UPX_OK;
}

// owned samples per chunk of a long signal (UPX_STREAM_CHUNK when the plan was created), 0 = never cut
int64_t stream_chunk_default(const upx_plan* p) { return p->knob_stream_chunk; }
}   // namespace

int upx_process_chunked(upx_plan* p, const float* stereo, int64_t n, float* out_c, float* out_l, float* out_r,
                        int64_t chunk) {
    if (!p || n < 0 || chunk < 1) return fail(UPX_ERR_INVALID, "upx_process_chunked: bad argument");
    if (n == 0) return UPX_OK;
    if (!stereo || !out_c || !out_l || !out_r) return fail(UPX_ERR_INVALID, "upx_process_chunked: NULL buffer");
    int64_t grid = 0, spill = 0;
    if (!shard_geometry(p, &grid, &spill))
        return fail(UPX_ERR_UNSUPPORTED, "the bands' hops do not share a shard grid: the signal cannot be cut into chunks");
    std::vector<PipeItem> items;
    if (int rc = items_of_track(p, stereo, n, out_c, out_l, out_r, chunk, true, items)) return rc;
    return run_items(p, items);
}

int upx_process(upx_plan* p, const float* stereo, int64_t n, float* out_c, float* out_l, float* out_r) {
    if (!p || n < 0) return fail(UPX_ERR_INVALID, "upx_process: bad argument");
    if (n == 0) return UPX_OK;
    if (!stereo || !out_c || !out_l || !out_r) return fail(UPX_ERR_INVALID, "upx_process: NULL buffer");
    // long signals stream through the device in chunks (uploads, kernels and downloads overlap; no limit on the
    // length); short ones are a single work item of the same pipeline
    std::vector<PipeItem> items;
    if (int rc = items_of_track(p, stereo, n, out_c, out_l, out_r, stream_chunk_default(p), false, items)) return rc;
    return run_items(p, items);
}

int upx_process_lr(upx_plan* p, const void* left, const void* right, int sample_format, int64_t stride, int64_t n,
                   float* out_c, float* out_l, float* out_r) {
    if (!p || n < 0) return fail(UPX_ERR_INVALID, "upx_process_lr: bad argument");
    if (sample_format != UPX_SAMPLE_F32 && sample_format != UPX_SAMPLE_F64)
        return fail(UPX_ERR_INVALID, "upx_process_lr: sample_format %d (UPX_SAMPLE_F32 or UPX_SAMPLE_F64)", sample_format);
    if (n == 0) return UPX_OK;
    if (!left || !right || !out_c || !out_l || !out_r) return fail(UPX_ERR_INVALID, "upx_process_lr: NULL buffer");
    const size_t elem = sample_format == UPX_SAMPLE_F64 ? 8 : 4;
    const bool interleaved = stride == 2 && static_cast<const char*>(right) == static_cast<const char*>(left) + elem;
    if (!interleaved && stride != 1)
        return fail(UPX_ERR_UNSUPPORTED, "upx_process_lr: stride %lld (1 = two contiguous arrays; 2 with right == left + one "
                                         "element = the columns of one interleaved [T][2] array)", (long long)stride);
    std::vector<PipeItem> items;
    if (int rc = items_of_track(p, static_cast<const float*>(left), n, out_c, out_l, out_r, stream_chunk_default(p), false,
                                items, interleaved ? nullptr : right, sample_format))
        return rc;
    return run_items(p, items);
}

int upx_process_tracks(upx_plan* p, int32_t n_tracks, const float* const* stereo, const int64_t* n,
                       float* const* out_c, float* const* out_l, float* const* out_r) {
    if (!p || n_tracks < 0) return fail(UPX_ERR_INVALID, "upx_process_tracks: bad argument");
    if (n_tracks == 0) return UPX_OK;
    if (!stereo || !n || !out_c || !out_l || !out_r) return fail(UPX_ERR_INVALID, "upx_process_tracks: NULL argument");
    std::vector<PipeItem> items;
    const int64_t chunk = stream_chunk_default(p);
    for (int32_t t = 0; t < n_tracks; ++t) {
        if (n[t] < 0) return fail(UPX_ERR_INVALID, "upx_process_tracks: track %d has a negative length", t);
        if (n[t] == 0) continue;
        if (!stereo[t] || !out_c[t] || !out_l[t] || !out_r[t])
            return fail(UPX_ERR_INVALID, "upx_process_tracks: track %d has a NULL buffer", t);
        if (int rc = items_of_track(p, stereo[t], n[t], out_c[t], out_l[t], out_r[t], chunk, false, items)) return rc;
    }
    return run_items(p, items);
}

int upx_plan_enable_timing(upx_plan* p, int enable) {
    if (!p) return fail(UPX_ERR_INVALID, "upx_plan_enable_timing: NULL plan");
    if (enable == 2 || enable == 3) {   // resume / pause: the calls recorded so far stay (a loop that times every n-th call)
        p->timing = enable == 2;
        return UPX_OK;
    }
    p->timing = enable != 0;
    p->timed_calls = 0;
    if (!p->timing) p->timed_once = false;
    return UPX_OK;
}

int upx_plan_band_times_ms(upx_plan* p, float* ms, int n_bands) {
    if (!p || !ms || n_bands != (int)p->bands.size()) return fail(UPX_ERR_INVALID, "upx_plan_band_times_ms: bad argument");
    if (!p->timed_once) return fail(UPX_ERR_INVALID, "no timed upx_process_device call recorded");
    HIP_TRY(hipStreamSynchronize(p->stream));
    for (int b = 0; b < n_bands; ++b) {
        ms[b] = 0.f;
        if (p->bands[b].timed_wg > 0) HIP_TRY(hipEventElapsedTime(&ms[b], p->bands[b].ev0, p->bands[b].ev1));
    }
    return UPX_OK;
}

int upx_plan_band_times_sum_ms(upx_plan* p, float* ms, int n_bands, int n_calls) {
    if (!p || !ms || n_bands != (int)p->bands.size() || n_calls < 1 || n_calls > kTimingSlots)
        return fail(UPX_ERR_INVALID, "upx_plan_band_times_sum_ms: bad argument (at most %d calls are kept)", kTimingSlots);
    if ((long long)n_calls > p->timed_calls) return fail(UPX_ERR_INVALID, "only %lld timed calls recorded", p->timed_calls);
    HIP_TRY(hipStreamSynchronize(p->stream));
    for (int b = 0; b < n_bands; ++b) {
        double sum = 0.0;
        for (long long c = p->timed_calls - n_calls; c < p->timed_calls; ++c) {
            const int slot = (int)(c % kTimingSlots);
            if (!p->bands[b].ring_used[slot]) continue;
            float t = 0.f;
            HIP_TRY(hipEventElapsedTime(&t, p->bands[b].ring0[slot], p->bands[b].ring1[slot]));
            sum += t;
        }
        ms[b] = (float)sum;
    }
    return UPX_OK;
}

int upx_plan_band_times_calls_ms(upx_plan* p, float* ms, int n_bands, int n_calls) {
    if (!p || !ms || n_bands != (int)p->bands.size() || n_calls < 1 || n_calls > kTimingSlots)
        return fail(UPX_ERR_INVALID, "upx_plan_band_times_calls_ms: bad argument (at most %d calls are kept)", kTimingSlots);
    if ((long long)n_calls > p->timed_calls) return fail(UPX_ERR_INVALID, "only %lld timed calls recorded", p->timed_calls);
    HIP_TRY(hipStreamSynchronize(p->stream));
    for (int c = 0; c < n_calls; ++c) {
        const int slot = (int)((p->timed_calls - n_calls + c) % kTimingSlots);
        for (int b = 0; b < n_bands; ++b) {
            float t = 0.f;
            if (p->bands[b].ring_used[slot])
                HIP_TRY(hipEventElapsedTime(&t, p->bands[b].ring0[slot], p->bands[b].ring1[slot]));
            ms[(size_t)c * n_bands + b] = t;
        }
    }
    return UPX_OK;
}

int upx_plan_band_phase_times_sum_ms(upx_plan* p, float* ms_analysis, float* ms_synthesis, int n_bands, int n_calls) {
    if (!p || !ms_analysis || !ms_synthesis || n_bands != (int)p->bands.size() || n_calls < 1 || n_calls > kTimingSlots)
        return fail(UPX_ERR_INVALID, "upx_plan_band_phase_times_sum_ms: bad argument");
    if ((long long)n_calls > p->timed_calls) return fail(UPX_ERR_INVALID, "only %lld timed calls recorded", p->timed_calls);
    HIP_TRY(hipStreamSynchronize(p->stream));
    for (int b = 0; b < n_bands; ++b) {
        const BandState& s = p->bands[b];
        double sa = 0.0, ss = 0.0;
        for (long long c = p->timed_calls - n_calls; c < p->timed_calls; ++c) {
            const int slot = (int)(c % kTimingSlots);
            if (!s.ring_used[slot]) continue;
            const int n_mid = s.ring_mid_n.empty() ? 0 : s.ring_mid_n[slot];
            if (n_mid == 0) {   // one kernel (or unsplit): everything counts as the band's single phase
                float t = 0.f;
                HIP_TRY(hipEventElapsedTime(&t, s.ring0[slot], s.ring1[slot]));
                ss += t;
                continue;
            }
            // ev0 | analysis | mid0 | synthesis | mid1 | analysis | mid2 | ... | synthesis (+ seam add) | ev1
            const hipEvent_t* mid = s.ring_mid.data() + (size_t)slot * kMidEvents;
            hipEvent_t prev = s.ring0[slot];
            for (int i = 0; i <= n_mid; ++i) {
                const hipEvent_t next = i < n_mid ? mid[i] : s.ring1[slot];
                float t = 0.f;
                HIP_TRY(hipEventElapsedTime(&t, prev, next));
                if (i % 2 == 0) sa += t;
                else ss += t;
                prev = next;
            }
        }
        ms_analysis[b] = (float)sa;
        ms_synthesis[b] = (float)ss;
    }
    return UPX_OK;
}

int upx_plan_band_phase_kernel_name(upx_plan* p, int band, int phase, char* name, size_t n) {
    if (!p || !name || n == 0 || band < 0 || band >= (int)p->bands.size() || phase < 0 || phase > 1)
        return fail(UPX_ERR_INVALID, "upx_plan_band_phase_kernel_name: bad argument");
    const BandState& s = p->bands[p->bands[band].group_leader];
    if (s.zoom) std::snprintf(name, n, "%s", phase == 0 ? s.zoom_a->name_analysis : s.zoom->name_synthesis);
    else if (phase == 0) name[0] = 0;   // single-kernel bands have no separate analysis phase
    else if (s.kern) std::snprintf(name, n, "%s", s.kern->name);
    else std::snprintf(name, n, "unfused<%d>", s.n);
    return UPX_OK;
}

int upx_plan_band_info(upx_plan* p, int band, int32_t* workgroups, int32_t* threads, int32_t* lds_bytes,
                       int32_t* blocks_per_stream) {
    if (!p || band < 0 || band >= (int)p->bands.size()) return fail(UPX_ERR_INVALID, "upx_plan_band_info: bad argument");
    const BandState& s = p->bands[band];
    if (workgroups) *workgroups = s.last_wg;
    if (threads) *threads = s.zoom ? s.zoom->wg : (s.kern ? s.kern->wg : s.big->row_wg);
    if (lds_bytes) *lds_bytes = s.zoom ? s.zoom->lds_bytes : (s.kern ? s.kern->lds_bytes : s.big->row_lds);
    if (blocks_per_stream) *blocks_per_stream = s.last_f;
    return UPX_OK;
}

int upx_plan_band_fill(upx_plan* p, int band, int32_t* workgroups, int32_t* slots, int32_t* workgroups_analysis,
                       int32_t* slots_analysis) {
    if (!p || band < 0 || band >= (int)p->bands.size()) return fail(UPX_ERR_INVALID, "upx_plan_band_fill: bad argument");
    const BandState& s = p->bands[p->bands[band].group_leader];
    if (workgroups) *workgroups = s.fill_wg;
    if (slots) *slots = s.fill_slots;
    if (workgroups_analysis) *workgroups_analysis = s.fill_wg_a;
    if (slots_analysis) *slots_analysis = s.fill_slots_a;
    return UPX_OK;
}

int upx_plan_band_stream_starts(upx_plan* p, int band, int32_t* starts, int32_t cap, int32_t* n_out) {
    if (!p || !n_out || cap < 0 || (cap > 0 && !starts) || band < 0 || band >= (int)p->bands.size())
        return fail(UPX_ERR_INVALID, "upx_plan_band_stream_starts: bad argument");
    const BandState& s = p->bands[p->bands[band].group_leader];
    std::vector<int> v;
    if (s.last_wg <= 0) {
        // (the band's group launched nothing in the last call)
    } else if (s.zoom || (s.kern && s.last_uneven)) {
        v = s.h_m0;                                   // band-limited: Ls/Rs table (+ end) then centre table (+ end)
    } else if (s.kern) {
        for (int i = 0; i <= s.last_streams; ++i) v.push_back(-1 + i * s.last_f);
    } else {
        for (int i = 0; i <= s.last_wg; ++i) v.push_back(i * s.last_f);     // unfused: chunks of last_f emitted blocks
    }
    *n_out = (int32_t)v.size();
    for (int i = 0; i < (int)v.size() && i < cap; ++i) starts[i] = v[(size_t)i];
    return UPX_OK;
}

int upx_plan_band_kernel_name(upx_plan* p, int band, char* name, size_t n) {
    if (!p || !name || n == 0 || band < 0 || band >= (int)p->bands.size())
        return fail(UPX_ERR_INVALID, "upx_plan_band_kernel_name: bad argument");
    const BandState& s = p->bands[p->bands[band].group_leader];
    if (s.zoom) std::snprintf(name, n, "%s", s.zoom->name_synthesis);
    else if (s.kern) std::snprintf(name, n, "%s", s.kern->name);
    else std::snprintf(name, n, "unfused<%d>", s.n);
    return UPX_OK;
}

int upx_plan_band_group(upx_plan* p, int band, int32_t* leader, int32_t* size) {
    if (!p || band < 0 || band >= (int)p->bands.size()) return fail(UPX_ERR_INVALID, "upx_plan_band_group: bad argument");
    const BandState& s = p->bands[band];
    if (leader) *leader = s.group_leader;
    if (size) *size = p->bands[s.group_leader].group_size;
    return UPX_OK;
}

int upx_absmax(upx_plan* p, const float* d_x, int64_t n, float* result) {
    if (!p || !result || n < 0) return fail(UPX_ERR_INVALID, "upx_absmax: bad argument");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipMemsetAsync(p->d_scalar, 0, sizeof(unsigned int), p->stream));
    if (n > 0) hipLaunchKernelGGL(upx_absmax_kernel, dim3(grid_reduce(n)), dim3(256), 0, p->stream, d_x, (long long)n, p->d_scalar);
    unsigned int bits = 0;
    HIP_TRY(hipMemcpyAsync(&bits, p->d_scalar, sizeof bits, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    std::memcpy(result, &bits, sizeof bits);
    return UPX_OK;
}

int upx_scale(upx_plan* p, float* d_x, int64_t n, float scale) {
    if (!p || n < 0) return fail(UPX_ERR_INVALID, "upx_scale: bad argument");
    HIP_TRY(hipSetDevice(p->device));
    if (n > 0) hipLaunchKernelGGL(upx_scale_kernel, dim3(grid_for(n)), dim3(256), 0, p->stream, d_x, (long long)n, scale);
    HIP_TRY(hipGetLastError());
    return UPX_OK;
}

namespace {
int wav_bytes_of(int fmt) { return fmt == UPX_F32 ? 4 : fmt / 8; }
bool wav_format_ok(int fmt) { return fmt == UPX_PCM16 || fmt == UPX_PCM24 || fmt == UPX_PCM32 || fmt == UPX_F32; }

// device buffers of the WAV pipeline live in the plan and only grow: allocating and freeing ~0.8 GB per call costs
// more than the kernels
hipError_t wav_ensure(upx_plan* p, int slot, size_t bytes) {
    if (bytes <= p->wav_cap[slot]) return hipSuccess;
    if (p->d_wav[slot]) {
        hipError_t e = hipFree(p->d_wav[slot]);
        p->d_wav[slot] = nullptr;
        p->wav_cap[slot] = 0;
        if (e != hipSuccess) return e;
    }
    hipError_t e = hipMalloc(&p->d_wav[slot], bytes ? bytes : 4);
    if (e == hipSuccess) p->wav_cap[slot] = bytes;
    return e;
}
int ensure_copy_streams(upx_plan* p) {
    if (p->s_h2d) return UPX_OK;
    HIP_TRY(hipStreamCreateWithFlags(&p->s_h2d, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&p->s_d2h, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(hipEventCreateWithFlags(&p->ev_h2d[i], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&p->ev_comp[i], hipEventDisableTiming));
    }
    return UPX_OK;
}
}   // namespace

namespace {
// One chunk of the WAV pipeline: owned frames [start, start + own) of the shard; its kernels read t_in frames from
// `start` (own range + right halo) and write t_out (own + spill; the shard's last chunk: what is left of the planes).
using WavChunk = upx::WavChunkRec;

// the chunk schedule of a shard (upx::wav_schedule, upx_pipeline.h) for this plan's geometry and knobs
void wav_chunks(const upx_plan* p, int64_t t_in, int64_t own_len, int64_t t_out, int bytes_per_frame, std::vector<WavChunk>& out) {
    int64_t grid = 0, spill = 0;
    if (!shard_geometry(p, &grid, &spill)) grid = 0;
    upx::wav_schedule(t_in, own_len, t_out, grid, spill, p->knob_wav_chunk, p->knob_wav_uniform != 0, bytes_per_frame,
                      p->knob_wav_kernel_rate, out);
}

}   // namespace

int upx_wav_shard_open(upx_plan* p, upx_comm* comm, int in_format, int channels, int64_t t_in, int64_t own_len,
                       int64_t t_out, int64_t spill) {
    if (!p || t_in < 1 || own_len < 1 || own_len > t_in || t_out < own_len || spill < 0 || (channels != 1 && channels != 2))
        return fail(UPX_ERR_INVALID, "upx_wav_shard_open: bad argument");
    if (!wav_format_ok(in_format)) return fail(UPX_ERR_INVALID, "upx_wav_shard_open: unknown sample format");
    if (comm && comm->plan != p) return fail(UPX_ERR_INVALID, "upx_wav_shard_open: the communicator belongs to another plan");
    // a launch indexes at most 2^29 - 1 samples: longer shards must be cut into chunks (every plan whose hops share a grid)
    wav_chunks(p, t_in, own_len, t_out, channels * wav_bytes_of(in_format), p->wav_chunk_list);
    for (const auto& w : p->wav_chunk_list)
        if (w.t_in >= (1LL << 29) || w.t_out >= (1LL << 29))
            return fail(UPX_ERR_INVALID, "upx_wav_shard_open: a shard of 2^29 frames or more needs UPX_WAV_CHUNK > 0 and hops that "
                                         "share a shard grid");
    const bool exchange = comm && comm->n_ranks > 1 && spill > 0;
    if (exchange && comm->rank + 1 < comm->n_ranks && t_out < own_len + spill)
        return fail(UPX_ERR_INVALID, "upx_wav_shard_open: planes of a shard with a successor need own_len + spill samples");
    HIP_TRY(hipSetDevice(p->device));
    if (exchange)
        if (int rc = upx_comm_reserve(comm, spill)) return rc;   // (not inside the exchange: that is a synchronising call)
    p->wav_t0 = wall_ms();
    p->wav_open = false;
    p->wav_feeding = false;
    const int width = wav_bytes_of(in_format);
    HIP_TRY(wav_ensure(p, 0, (size_t)t_in * channels * width));
    HIP_TRY(wav_ensure(p, 1, (size_t)t_in * 2 * sizeof(float)));
    HIP_TRY(wav_ensure(p, 2, (size_t)t_out * 3 * sizeof(float)));
    if (int rc = ensure_copy_streams(p)) return rc;
    for (auto& e : p->wav_ev)
        if (!e) HIP_TRY(hipEventCreate(&e));
    if (!p->d_wav_peaks) HIP_TRY(hipMalloc(&p->d_wav_peaks, 4 * sizeof(unsigned int)));
    // The shard runs chunk by chunk: the samples of chunk c + 1 come up (copy stream) while chunk c is decoded and runs
    // through every band (plan's stream); a chunk's kernels write own + spill samples of the planes, the spill is parked
    // in a side row while the next chunk's kernels write that range anew, then added onto it - the chunk seam of
    // upx_process (run_items) on planes that stay resident, because the one scale of main.py:85-97 needs every peak
    // before anything can be exported.  The peaks are folded into the same stream (max of bit patterns, one
    // download of four words at the end).
    p->wav_cspill = 0;
    int64_t cgrid = 0;
    if (p->wav_chunk_list.size() > 1) (void)shard_geometry(p, &cgrid, &p->wav_cspill);
    if (p->wav_chunk_list.size() > 1 && (size_t)(3 * p->wav_cspill) > p->wav_side_floats) {
        HIP_TRY(hipStreamSynchronize(p->stream));
        for (auto*& q : p->d_wav_side) {
            if (q) HIP_TRY(hipFree(q));
            q = nullptr;
        }
        p->wav_side_floats = 0;
        for (auto*& q : p->d_wav_side) HIP_TRY(hipMalloc(&q, (size_t)3 * p->wav_cspill * sizeof(float)));
        p->wav_side_floats = (size_t)3 * p->wav_cspill;
    }
    hipStream_t st = p->stream;
    HIP_TRY(hipEventRecord(p->wav_ev[0], st));
    HIP_TRY(hipStreamWaitEvent(p->s_h2d, p->wav_ev[0], 0));   // the previous call's kernels have left the buffers
    HIP_TRY(hipMemsetAsync(p->d_wav_peaks, 0, 4 * sizeof(unsigned int), st));
    p->wav_tin = t_in; p->wav_own = own_len; p->wav_tout = t_out;
    p->wav_fmt = in_format; p->wav_ch = channels;
    p->wav_comm = exchange ? comm : nullptr;
    p->wav_spill = spill;
    p->wav_fed = p->wav_decoded = 0;
    p->wav_next_chunk = 0;
    p->wav_n_feed = 0;
    p->wav_peaks_pending_head = false;
    p->wav_feeding = true;
    return UPX_OK;
}

int upx_wav_shard_feed(upx_plan* p, const void* pcm, int64_t n_frames) {
    if (!p || !p->wav_feeding) return fail(UPX_ERR_INVALID, "upx_wav_shard_feed: no shard is being fed (call upx_wav_shard_open)");
    if (!pcm || n_frames < 1 || p->wav_fed + n_frames > p->wav_tin)
        return fail(UPX_ERR_INVALID, "upx_wav_shard_feed: %lld frames after %lld of %lld", (long long)n_frames,
                    (long long)p->wav_fed, (long long)p->wav_tin);
    HIP_TRY(hipSetDevice(p->device));
    const int width = wav_bytes_of(p->wav_fmt), channels = p->wav_ch;
    unsigned char* d_pcm = (unsigned char*)p->d_wav[0];
    float* d_st = (float*)p->d_wav[1];
    float* d_pl = (float*)p->d_wav[2];
    const int64_t t_out = p->wav_tout, own_len = p->wav_own, cspill = p->wav_cspill;
    float* d_plane[3] = {d_pl, d_pl + t_out, d_pl + 2 * t_out};
    hipStream_t st = p->stream;
    // the piece goes up on the copy stream ...
    const size_t off = (size_t)p->wav_fed * channels * width;
    HIP_TRY(hipMemcpyAsync(d_pcm + off, pcm, (size_t)n_frames * channels * width, hipMemcpyHostToDevice, p->s_h2d));
    if ((size_t)p->wav_n_feed >= p->wav_piece_ev.size()) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        p->wav_piece_ev.push_back(e);
    }
    hipEvent_t landed = p->wav_piece_ev[(size_t)p->wav_n_feed++];
    HIP_TRY(hipEventRecord(landed, p->s_h2d));
    p->wav_fed += n_frames;
    // ... and every chunk whose input is now complete is queued behind it on the plan's stream
    bool waited = false;
    const std::vector<WavChunk>& chunks = p->wav_chunk_list;
    while ((size_t)p->wav_next_chunk < chunks.size()) {
        const size_t c = (size_t)p->wav_next_chunk;
        const WavChunk& w = chunks[c];
        const int64_t end = w.start + w.t_in;
        if (end > p->wav_fed) break;
        if (!waited) HIP_TRY(hipStreamWaitEvent(st, landed, 0));
        waited = true;
        const int64_t up = p->wav_decoded;
        if (end > up) {
            const size_t o = (size_t)up * channels * width;
            hipLaunchKernelGGL(upx_decode_kernel, dim3(grid_for(end - up)), dim3(256), 0, st, d_pcm + o, p->wav_fmt, channels,
                               (long long)(end - up), d_st + 2 * up);
            const int64_t owned_end = end < own_len ? end : own_len;   // input peak: owned frames only (main.py:53)
            if (owned_end > up && p->wav_fmt == UPX_PCM32)      // (exact: on the integers, see upx_absmax_i32_kernel)
                hipLaunchKernelGGL(upx_absmax_i32_kernel, dim3(grid_reduce(channels * (owned_end - up))), dim3(256), 0, st,
                                   reinterpret_cast<const int*>(d_pcm + o), (long long)(channels * (owned_end - up)), p->d_wav_peaks);
            else if (owned_end > up)
                hipLaunchKernelGGL(upx_absmax_kernel, dim3(grid_reduce(2 * (owned_end - up))), dim3(256), 0, st, d_st + 2 * up,
                                   (long long)(2 * (owned_end - up)), p->d_wav_peaks);
            p->wav_decoded = end;
        }
        if (c + 1 == chunks.size()) HIP_TRY(hipEventRecord(p->wav_ev[1], st));
        if (int rc = upx_process_device(p, d_st + 2 * w.start, w.t_in, w.own, d_plane[0] + w.start, d_plane[1] + w.start,
                                        d_plane[2] + w.start, w.t_out))
            return rc;
        if (c > 0) {
            const float* side = p->d_wav_side[(c - 1) & 1];
            const int64_t parked = chunks[c - 1].t_out - chunks[c - 1].own;     // <= cspill
            const int64_t add = parked < w.t_out ? parked : w.t_out;
            hipLaunchKernelGGL(upx_seam_add_kernel, dim3(grid_for(add)), dim3(256), 0, st, d_plane[0] + w.start,
                               d_plane[1] + w.start, d_plane[2] + w.start, side, side + cspill, side + 2 * cspill, (long long)add);
        }
        if (c + 1 < chunks.size())
            hipLaunchKernelGGL(upx_seam_pack_kernel, dim3(grid_for(cspill)), dim3(256), 0, st, p->d_wav_side[c & 1],
                               d_plane[0] + w.start, d_plane[1] + w.start, d_plane[2] + w.start, (long long)w.own, (long long)cspill,
                               (long long)(w.t_out - w.own));
        // plane peaks of the chunk's owned range, final now - except the shard's head under a multi-rank seam
        if (p->wav_comm && c == 0) {
            p->wav_peaks_pending_head = true;
            p->wav_head_own = w.own;
        } else {
            hipLaunchKernelGGL(upx_absmax3_kernel, dim3(grid_reduce(w.own), 3), dim3(256), 0, st, d_plane[0] + w.start,
                               d_plane[1] + w.start, d_plane[2] + w.start, (long long)w.own, p->d_wav_peaks + 1);
        }
        p->wav_next_chunk += 1;
    }
    HIP_TRY(hipGetLastError());
    return UPX_OK;
}

int upx_wav_shard_seal(upx_plan* p, double* peaks) {
    if (!p || !peaks || !p->wav_feeding) return fail(UPX_ERR_INVALID, "upx_wav_shard_seal: no shard is being fed");
    if (p->wav_fed != p->wav_tin || (size_t)p->wav_next_chunk != p->wav_chunk_list.size())
        return fail(UPX_ERR_INVALID, "upx_wav_shard_seal: %lld of %lld frames fed", (long long)p->wav_fed, (long long)p->wav_tin);
    HIP_TRY(hipSetDevice(p->device));
    hipStream_t st = p->stream;
    float* d_pl = (float*)p->d_wav[2];
    float* d_plane[3] = {d_pl, d_pl + p->wav_tout, d_pl + 2 * p->wav_tout};
    p->wav_feeding = false;
    if (p->wav_comm) {
        if (int rc = upx_comm_seam_exchange(p->wav_comm, d_plane[0], d_plane[1], d_plane[2], p->wav_own, p->wav_spill)) return rc;
        // a peer that never enters the all-reduce must not hold this rank in the synchronisation below for ever
        if (int rc = upx_comm_wait(p->wav_comm, -1.0)) return rc;
    }
    if (p->wav_peaks_pending_head) {
        hipLaunchKernelGGL(upx_absmax3_kernel, dim3(grid_reduce(p->wav_head_own), 3), dim3(256), 0, st, d_plane[0], d_plane[1],
                           d_plane[2], (long long)p->wav_head_own, p->d_wav_peaks + 1);
        p->wav_peaks_pending_head = false;
    }
    p->wav_open = true;
    unsigned int bits[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(bits, p->d_wav_peaks, sizeof bits, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    float pk[4];
    std::memcpy(pk, bits, sizeof pk);
    auto fmax_nan = [](float a, float b) { return (a != a || b != b) ? (a != a ? a : b) : (a > b ? a : b); };
    peaks[0] = p->wav_fmt == UPX_PCM32 ? (double)bits[0] / 2147483648.0 : (double)pk[0];
    p->wav_in_peak = peaks[0];
    peaks[1] = (double)fmax_nan(fmax_nan(pk[1], pk[2]), pk[3]);
    // what open .. seal cost on the host's clock, and how much of it came after the last sample had landed on the device
    float landed_ms = 0.f;
    (void)hipEventElapsedTime(&landed_ms, p->wav_ev[0], p->wav_ev[1]);
    p->pipe_ms[0] = (float)(wall_ms() - p->wav_t0);
    p->pipe_ms[1] = p->pipe_ms[0] - landed_ms > 0.f ? p->pipe_ms[0] - landed_ms : 0.f;
    return UPX_OK;
}

int upx_wav_shard_begin(upx_plan* p, upx_comm* comm, const void* pcm_in, int in_format, int channels, int64_t t_in,
                        int64_t own_len, int64_t t_out, int64_t spill, double* peaks) {
    if (!p || !pcm_in || !peaks) return fail(UPX_ERR_INVALID, "upx_wav_shard_begin: bad argument");
    if (int rc = upx_wav_shard_open(p, comm, in_format, channels, t_in, own_len, t_out, spill)) return rc;
    // one piece per chunk: what chunk c needs beyond chunk c - 1's input (all uploads are queued at once; the kernels of
    // chunk c wait for piece c only)
    const size_t frame = (size_t)channels * wav_bytes_of(in_format);
    int64_t up = 0;
    const std::vector<WavChunk> chunks = p->wav_chunk_list;
    for (const WavChunk& w : chunks) {
        const int64_t end = w.start + w.t_in;
        if (end <= up) continue;
        if (int rc = upx_wav_shard_feed(p, (const unsigned char*)pcm_in + (size_t)up * frame, end - up)) return rc;
        up = end;
    }
    return upx_wav_shard_seal(p, peaks);
}

int upx_wav_shard_peaks(upx_plan* p, double* peaks) {
    if (!p || !peaks || !p->wav_open) return fail(UPX_ERR_INVALID, "upx_wav_shard_peaks: no shard is open");
    HIP_TRY(hipSetDevice(p->device));
    // peaks of the owned range: the input (both channels) and the three planes (main.py:53-55, :85-88)
    const float* d_st = (const float*)p->d_wav[1];
    const float* d_pl = (const float*)p->d_wav[2];
    float pk[4] = {0.f, 0.f, 0.f, 0.f};
    if (int rc = upx_absmax(p, d_st, 2 * p->wav_own, &pk[0])) return rc;
    for (int i = 0; i < 3; ++i)
        if (int rc = upx_absmax(p, d_pl + (size_t)i * p->wav_tout, p->wav_own, &pk[1 + i])) return rc;
    auto fmax_nan = [](float a, float b) { return (a != a || b != b) ? (a != a ? a : b) : (a > b ? a : b); };
    peaks[0] = p->wav_fmt == UPX_PCM32 ? p->wav_in_peak : (double)pk[0];   // (32-bit files: the integer peak of the seal)
    peaks[1] = (double)fmax_nan(fmax_nan(pk[1], pk[2]), pk[3]);
    return UPX_OK;
}

int upx_wav_shard_planes(upx_plan* p, float** d_c, float** d_l, float** d_r, int64_t* own_len, int64_t* t_out) {
    if (!p || !p->wav_open) return fail(UPX_ERR_INVALID, "upx_wav_shard_planes: no shard is open");
    float* d_pl = (float*)p->d_wav[2];
    if (d_c) *d_c = d_pl;
    if (d_l) *d_l = d_pl + p->wav_tout;
    if (d_r) *d_r = d_pl + 2 * p->wav_tout;
    if (own_len) *own_len = p->wav_own;
    if (t_out) *t_out = p->wav_tout;
    return UPX_OK;
}

int upx_wav_shard_finish_async(upx_plan* p, double scale, int mode, int out_format, void* out0, void* out1, void* out2,
                               int64_t piece_frames, int32_t* n_pieces) {
    if (!p || !p->wav_open) return fail(UPX_ERR_INVALID, "upx_wav_shard_finish: no shard is open (call upx_wav_shard_begin)");
    if (!wav_format_ok(out_format)) return fail(UPX_ERR_INVALID, "upx_wav_shard_finish: unknown sample format");
    if (mode != UPX_EXPORT_STEREO_SUM && mode != UPX_EXPORT_SPLIT && mode != UPX_EXPORT_AB)
        return fail(UPX_ERR_INVALID, "upx_wav_shard_finish: unknown export mode");
    const int n_out = mode == UPX_EXPORT_SPLIT ? 3 : 1;
    void* outs[3] = {out0, out1, out2};
    for (int i = 0; i < n_out; ++i)
        if (!outs[i]) return fail(UPX_ERR_INVALID, "upx_wav_shard_finish: output buffer %d is NULL", i);
    HIP_TRY(hipSetDevice(p->device));
    const int64_t n = p->wav_own, t_out = p->wav_tout;
    const size_t out_bytes = (size_t)n * 2 * wav_bytes_of(out_format);
    for (int i = 0; i < n_out; ++i) HIP_TRY(wav_ensure(p, 3 + i, out_bytes));
    unsigned char* d_o[3] = {(unsigned char*)p->d_wav[3], (unsigned char*)p->d_wav[4], (unsigned char*)p->d_wav[5]};
    const float* d_pl = (const float*)p->d_wav[2];
    hipStream_t st = p->stream;
    p->wav_t0 = wall_ms();
    if (int rc = ensure_copy_streams(p)) return rc;
    // (a caller that has not waited for every piece of the previous shard: its last download still reads the buffers the
    // export kernels below write)
    if (p->wav_pieces > 0) HIP_TRY(hipStreamWaitEvent(st, p->wav_down_ev[(size_t)p->wav_pieces - 1], 0));
    // export layout + quantisation piece by piece; a piece goes down (copy stream) while the next one is exported
    int64_t piece = piece_frames > 0 ? piece_frames : (p->knob_wav_chunk > 0 ? p->knob_wav_chunk : n);
    if (piece > n) piece = n;
    const int64_t pieces = (n + piece - 1) / piece;
    const size_t frame_bytes = (size_t)2 * wav_bytes_of(out_format);
    const size_t in_frame_bytes = (size_t)p->wav_ch * wav_bytes_of(p->wav_fmt);
    while ((int64_t)p->wav_piece_ev.size() < pieces || (int64_t)p->wav_down_ev.size() < pieces) {
        hipEvent_t e = nullptr;
        HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ((int64_t)p->wav_piece_ev.size() < pieces ? p->wav_piece_ev : p->wav_down_ev).push_back(e);
    }
    int k = 0;
    for (int64_t f0 = 0; f0 < n; f0 += piece, ++k) {
        const int64_t nf = n - f0 < piece ? n - f0 : piece;
        hipLaunchKernelGGL(upx_export_kernel, dim3(grid_for(nf)), dim3(256), 0, st, d_pl + f0, d_pl + t_out + f0,
                           d_pl + 2 * t_out + f0, (const unsigned char*)p->d_wav[0] + (size_t)f0 * in_frame_bytes, p->wav_fmt,
                           p->wav_ch, (long long)nf, scale, mode, out_format, d_o[0] + (size_t)f0 * frame_bytes,
                           d_o[1] ? d_o[1] + (size_t)f0 * frame_bytes : nullptr, d_o[2] ? d_o[2] + (size_t)f0 * frame_bytes : nullptr);
        HIP_TRY(hipEventRecord(p->wav_piece_ev[k], st));
        HIP_TRY(hipStreamWaitEvent(p->s_d2h, p->wav_piece_ev[k], 0));
        for (int i = 0; i < n_out; ++i)
            HIP_TRY(hipMemcpyAsync((unsigned char*)outs[i] + (size_t)f0 * frame_bytes, d_o[i] + (size_t)f0 * frame_bytes,
                                   (size_t)nf * frame_bytes, hipMemcpyDeviceToHost, p->s_d2h));
        HIP_TRY(hipEventRecord(p->wav_down_ev[k], p->s_d2h));
    }
    HIP_TRY(hipGetLastError());
    p->wav_pieces = (int)pieces;
    p->wav_open = false;
    if (n_pieces) *n_pieces = (int32_t)pieces;
    return UPX_OK;
}

int upx_wav_shard_wait_piece(upx_plan* p, int32_t piece) {
    if (!p || piece < 0 || piece >= p->wav_pieces) return fail(UPX_ERR_INVALID, "upx_wav_shard_wait_piece: no such piece");
    HIP_TRY(hipEventSynchronize(p->wav_down_ev[(size_t)piece]));
    if (piece == p->wav_pieces - 1) p->pipe_ms[2] = (float)(wall_ms() - p->wav_t0);
    return UPX_OK;
}

int upx_wav_shard_finish(upx_plan* p, double scale, int mode, int out_format, void* out0, void* out1, void* out2) {
    int32_t pieces = 0;
    if (int rc = upx_wav_shard_finish_async(p, scale, mode, out_format, out0, out1, out2, 0, &pieces)) return rc;
    return upx_wav_shard_wait_piece(p, pieces - 1);    // (the pieces come down in order on one stream)
}

int upx_wav_pipeline(upx_plan* p, const void* pcm_in, int in_format, int channels, int64_t n, int mode, int out_format,
                     void* out0, void* out1, void* out2, double* stats) {
    if (!p || !pcm_in || n < 0 || (channels != 1 && channels != 2) || !stats)
        return fail(UPX_ERR_INVALID, "upx_wav_pipeline: bad argument");
    if (!wav_format_ok(in_format) || !wav_format_ok(out_format))
        return fail(UPX_ERR_INVALID, "upx_wav_pipeline: unknown sample format");
    if (mode != UPX_EXPORT_STEREO_SUM && mode != UPX_EXPORT_SPLIT && mode != UPX_EXPORT_AB)
        return fail(UPX_ERR_INVALID, "upx_wav_pipeline: unknown export mode");
    const int n_out = mode == UPX_EXPORT_SPLIT ? 3 : 1;
    void* outs[3] = {out0, out1, out2};
    for (int i = 0; i < n_out; ++i)
        if (!outs[i]) return fail(UPX_ERR_INVALID, "upx_wav_pipeline: output buffer %d is NULL", i);
    stats[0] = 1e-9; stats[1] = 1e-9; stats[2] = 1.0;
    if (n == 0) return UPX_OK;
    // one shard that is the whole file: the two halves of the sharded pipeline with the scale of this file alone
    double peaks[2];
    if (int rc = upx_wav_shard_begin(p, nullptr, pcm_in, in_format, channels, n, n, n, 0, peaks)) return rc;
    const double pin = peaks[0] <= 0.0 ? 1e-9 : peaks[0];                  // main.py:53-55 (a NaN peak stays NaN)
    const double overall = peaks[1] > 1e-9 || peaks[1] != peaks[1] ? peaks[1] : 1e-9;   // main.py:88: max(.., 1e-9)
    const double scale = pin / overall;                                   // main.py:90
    stats[0] = pin; stats[1] = overall; stats[2] = scale;
    return upx_wav_shard_finish(p, scale, mode, out_format, out0, out1, out2);
}

int upx_wav_pipeline_times_ms(upx_plan* p, float* ms3) {
    if (!p || !ms3) return fail(UPX_ERR_INVALID, "upx_wav_pipeline_times_ms: bad argument");
    for (int i = 0; i < 3; ++i) ms3[i] = p->pipe_ms[i];
    return UPX_OK;
}

// ---- block-at-a-time streaming (MultiBandExtractorAccu.process_stereo_chunk) ------------------------------------
namespace {
int chunk_state(upx_plan* p, int* n_out, int* hop_out) {
    if (p->bands.size() != 1) return fail(UPX_ERR_INVALID, "upx_stream_*: the plan must hold exactly one band");
    const int n = p->bands[0].n, hop = p->bands[0].hop;
    if (!p->d_chunk) {
        HIP_TRY(hipSetDevice(p->device));
        HIP_TRY(hipMalloc(&p->d_chunk, ((size_t)8 * n + 3 * hop) * sizeof(float)));
        HIP_TRY(hipHostMalloc(&p->h_chunk, (size_t)5 * n * sizeof(float), hipHostMallocPortable));
        HIP_TRY(hipMemsetAsync(p->d_chunk + (size_t)5 * n, 0, (size_t)3 * n * sizeof(float), p->stream));
        p->chunk_pos = 0;
    }
    *n_out = n;
    *hop_out = hop;
    return UPX_OK;
}
}   // namespace

int upx_stream_chunk(upx_plan* p, const float* block_l, int32_t n_l, const float* block_r, int32_t n_r, float* out_c,
                     float* out_l, float* out_r) {
    if (!p || !out_c || !out_l || !out_r || n_l < 0 || n_r < 0 || (n_l > 0 && !block_l) || (n_r > 0 && !block_r))
        return fail(UPX_ERR_INVALID, "upx_stream_chunk: bad argument");
    int n = 0, hop = 0;
    if (int rc = chunk_state(p, &n, &hop)) return rc;
    if (n_l > n || n_r > n) return fail(UPX_ERR_INVALID, "upx_stream_chunk: a block holds at most block_size samples");
    HIP_TRY(hipSetDevice(p->device));
    // interleave into the page-locked staging row (short blocks are zero-extended, center_extraction.py:437-455)
    float* h = p->h_chunk;
    for (int i = 0; i < n; ++i) {
        h[2 * i] = i < n_l ? block_l[i] : 0.f;
        h[2 * i + 1] = i < n_r ? block_r[i] : 0.f;
    }
    float* d_blk = p->d_chunk;
    float* d_rec = d_blk + (size_t)2 * n;
    float* d_ring = d_rec + (size_t)3 * n;
    float* d_out = d_ring + (size_t)3 * n;
    hipStream_t st = p->stream;
    HIP_TRY(hipMemcpyAsync(d_blk, h, (size_t)2 * n * sizeof(float), hipMemcpyHostToDevice, st));
    // own_len = 1: only the frame that starts at sample 0 exists; its three reconstructions fill [0, N)
    if (int rc = upx_process_device(p, d_blk, n, 1, d_rec, d_rec + n, d_rec + 2 * n, n)) return rc;
    hipLaunchKernelGGL(upx_chunk_ola_kernel, dim3((n + 255) / 256), dim3(256), 0, st, d_rec, d_rec + n, d_rec + 2 * n, d_ring,
                       n, hop, p->chunk_pos, d_out);
    HIP_TRY(hipGetLastError());
    float* h_out = h + (size_t)2 * n;
    HIP_TRY(hipMemcpyAsync(h_out, d_out, (size_t)3 * hop * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    p->chunk_pos = (p->chunk_pos + hop) % n;
    std::memcpy(out_c, h_out, (size_t)hop * sizeof(float));
    std::memcpy(out_l, h_out + hop, (size_t)hop * sizeof(float));
    std::memcpy(out_r, h_out + 2 * hop, (size_t)hop * sizeof(float));
    return UPX_OK;
}

int upx_stream_state(upx_plan* p, float* acc_c, float* acc_l, float* acc_r, int clear) {
    if (!p || !acc_c || !acc_l || !acc_r) return fail(UPX_ERR_INVALID, "upx_stream_state: bad argument");
    int n = 0, hop = 0;
    if (int rc = chunk_state(p, &n, &hop)) return rc;
    HIP_TRY(hipSetDevice(p->device));
    float* d_rec = p->d_chunk + (size_t)2 * n;          // (free between calls: the unrolled ring goes through it)
    float* d_ring = d_rec + (size_t)3 * n;
    hipLaunchKernelGGL(upx_chunk_unroll_kernel, dim3((n + 255) / 256), dim3(256), 0, p->stream, d_ring, n, p->chunk_pos, d_rec,
                       clear ? 1 : 0);
    HIP_TRY(hipGetLastError());
    float* h = p->h_chunk + (size_t)2 * n;
    HIP_TRY(hipMemcpyAsync(h, d_rec, (size_t)3 * n * sizeof(float), hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    std::memcpy(acc_c, h, (size_t)n * sizeof(float));
    std::memcpy(acc_l, h + n, (size_t)n * sizeof(float));
    std::memcpy(acc_r, h + 2 * n, (size_t)n * sizeof(float));
    return UPX_OK;
}

int upx_stream_set_state(upx_plan* p, const float* acc_c, const float* acc_l, const float* acc_r) {
    if (!p || !acc_c || !acc_l || !acc_r) return fail(UPX_ERR_INVALID, "upx_stream_set_state: bad argument");
    int n = 0, hop = 0;
    if (int rc = chunk_state(p, &n, &hop)) return rc;
    HIP_TRY(hipSetDevice(p->device));
    float* h = p->h_chunk + (size_t)2 * n;
    std::memcpy(h, acc_c, (size_t)n * sizeof(float));
    std::memcpy(h + n, acc_l, (size_t)n * sizeof(float));
    std::memcpy(h + 2 * n, acc_r, (size_t)n * sizeof(float));
    float* d_ring = p->d_chunk + (size_t)5 * n;
    HIP_TRY(hipMemcpyAsync(d_ring, h, (size_t)3 * n * sizeof(float), hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    p->chunk_pos = 0;
    return UPX_OK;
}

// ---- RCCL seam ------------------------------------------------------------
int upx_comm_unique_id(char* id_out) {
    if (!id_out) return fail(UPX_ERR_INVALID, "upx_comm_unique_id: NULL");
    if (int rc = load_rccl()) return rc;
    ncclUniqueId id;
    NCCL_TRY(g_rccl.GetUniqueId(&id));
    static_assert(sizeof id == UPX_UNIQUE_ID_BYTES, "ncclUniqueId size");
    std::memcpy(id_out, &id, sizeof id);
    return UPX_OK;
}

int upx_comm_create(upx_comm** out, upx_plan* plan, int rank, int n_ranks, const char* id_bytes) {
    if (!out || !plan || !id_bytes || n_ranks < 1 || rank < 0 || rank >= n_ranks)
        return fail(UPX_ERR_INVALID, "upx_comm_create: bad argument");
    if (int rc = load_rccl()) return rc;
    HIP_TRY(hipSetDevice(plan->device));
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof id);
    upx_comm* c = new upx_comm();
    c->plan = plan;
    c->rank = rank;
    c->n_ranks = n_ranks;
    // UPX_COMM_TIMEOUT, else UPX_RDZV_TIMEOUT, else 600 s - parsed exactly like sharding.comm_timeout (a value that is not a
    // positive number falls through to the next variable), so that the Python watchdog and this wait agree
    c->timeout_s = 600.0;
    for (const char* name : {"UPX_COMM_TIMEOUT", "UPX_RDZV_TIMEOUT"}) {
        const char* e = std::getenv(name);
        if (!e || !*e) continue;
        char* end = nullptr;
        const double v = std::strtod(e, &end);
        while (end && (*end == ' ' || *end == '\t' || *end == '\n')) ++end;
        if (end && end != e && *end == 0 && v > 0.0 && std::isfinite(v)) {
            c->timeout_s = v;
            break;
        }
    }
    // (blocking: every rank must arrive - the callers vote over their process group BEFORE this call and keep a watchdog
    // on it, sharding.RcclSeam)
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, n_ranks, id, rank);
    if (r != ncclSuccess) {
        delete c;
        return fail(UPX_ERR_RCCL, "ncclCommInitRank(rank %d of %d): %s", rank, n_ranks, g_rccl.GetErrorString(r));
    }
    // upx_comm_wait polls these two; without them the wait would be an unbounded hipStreamSynchronize again, so a
    // communicator that cannot have them is not handed out (one retry: event creation fails transiently at most)
    for (hipEvent_t* ev : {&c->ev_start, &c->ev_done}) {
        hipError_t e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
        if (e != hipSuccess) e = hipEventCreateWithFlags(ev, hipEventDisableTiming);
        if (e != hipSuccess) {
            *ev = nullptr;
            const int rc = fail(UPX_ERR_HIP, "upx_comm_create: hipEventCreate: %s", hipGetErrorString(e));
            const std::string msg = g_err;
            if (g_rccl.CommAbort) (void)g_rccl.CommAbort(c->comm);   // (a destroy could wait for peers that go on to use theirs)
            c->comm = nullptr;
            upx_comm_destroy(c);
            g_err = msg;
            return rc;
        }
    }
    *out = c;
    return UPX_OK;
}

// The one way out of a collective whose peer never arrives: ncclCommAbort makes the communicator's kernels leave and frees
// it.  The planes behind it hold garbage; the caller reports failure and the process ends (a fresh process is the only
// restart).  Idempotent.
int upx_comm_abort(upx_comm* c) {
    if (!c) return fail(UPX_ERR_INVALID, "upx_comm_abort: NULL");
    if (c->aborted || !c->comm) return UPX_OK;
    c->aborted = true;
    c->pending = false;
    ncclComm_t comm = c->comm;
    c->comm = nullptr;
    if (!g_rccl.CommAbort) return fail(UPX_ERR_RCCL, "this librccl has no ncclCommAbort");
    NCCL_TRY(g_rccl.CommAbort(comm));
    return UPX_OK;
}

// Waits until the last queued seam exchange has run on the plan's stream.  timeout_s < 0: the communicator's default
// (UPX_COMM_TIMEOUT / UPX_RDZV_TIMEOUT / 600 s).  The limit applies to the PEERS: its clock starts when the stream has
// reached the all-reduce (ev_start), not at the entry of this call - the shard's own kernels queued ahead of the exchange get
// the same limit of their own.  If either runs out - a peer that never entered the all-reduce - or RCCL reports an
// asynchronous error, the communicator is ABORTED and an error returned, instead of a hipStreamSynchronize that never comes
// back.  (Multi-rank abort has only ever run on a one-rank communicator: one-GPU boxes.)
int upx_comm_wait(upx_comm* c, double timeout_s) {
    if (!c) return fail(UPX_ERR_INVALID, "upx_comm_wait: NULL");
    if (c->aborted) return fail(UPX_ERR_RCCL, "the communicator has been aborted");
    if (!c->pending) return UPX_OK;
    HIP_TRY(hipSetDevice(c->plan->device));
    const double limit = timeout_s >= 0.0 ? timeout_s : c->timeout_s;
    double t0 = wall_ms();
    bool reached = false;            // the stream has reached the all-reduce
    long long spins = 0;
    for (;;) {
        const hipError_t q = hipEventQuery(c->ev_done);
        if (q == hipSuccess) {
            c->pending = false;
            return UPX_OK;
        }
        if (q != hipErrorNotReady) return fail(UPX_ERR_HIP, "hipEventQuery: %s", hipGetErrorString(q));
        if (!reached && hipEventQuery(c->ev_start) == hipSuccess) {
            reached = true;
            t0 = wall_ms();
        }
        const double waited = (wall_ms() - t0) * 1e-3;
        if ((++spins & 255) == 0 && g_rccl.CommGetAsyncError && c->comm) {
            ncclResult_t async = ncclSuccess;
            if (g_rccl.CommGetAsyncError(c->comm, &async) == ncclSuccess && async != ncclSuccess && async != ncclInProgress) {
                const char* what = g_rccl.GetErrorString(async);
                (void)upx_comm_abort(c);
                return fail(UPX_ERR_RCCL, "rank %d: the seam all-reduce failed (%s); communicator aborted", c->rank, what);
            }
        }
        if (waited > limit) {
            (void)upx_comm_abort(c);
            if (reached)
                return fail(UPX_ERR_RCCL, "rank %d: the seam all-reduce did not finish within %.1f s (a peer never arrived); "
                                          "communicator aborted", c->rank, limit);
            return fail(UPX_ERR_RCCL, "rank %d: the stream did not finish within %.1f s what was queued ahead of the seam all-reduce; "
                                      "communicator aborted", c->rank, limit);
        }
        if (waited < 2e-3) continue;                       // a healthy exchange takes tens of microseconds: spin first
        std::this_thread::sleep_for(std::chrono::microseconds(waited < 0.1 ? 50 : 1000));
    }
}

void upx_comm_destroy(upx_comm* c) {
    if (!c) return;
    (void)hipSetDevice(c->plan->device);
    if (c->d_seam) (void)hipFree(c->d_seam);
    if (c->ev_start) (void)hipEventDestroy(c->ev_start);
    if (c->ev_done) (void)hipEventDestroy(c->ev_done);
    if (c->comm && !c->aborted && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
}

namespace {
// pack my spill into row `my_row` of seam[n_rows][3][spill], all-reduce, add row `add_row` onto my head
// `add_len` <= spill: samples of the predecessor's spill that land inside this rank's planes (the LAST shard's planes end at
// own_len, which ShardGeometry.plan allows to be shorter than the spill; what lies beyond is past the signal's end)
int seam_exchange_impl(upx_comm* c, float* d_c, float* d_l, float* d_r, int64_t own_len, int64_t spill, int n_rows,
                       int my_row, bool pack, int add_row, int64_t add_len) {
    upx_plan* p = c->plan;
    if (c->aborted || !c->comm) return fail(UPX_ERR_RCCL, "the communicator has been aborted");
    HIP_TRY(hipSetDevice(p->device));
    const long long row = 3 * (long long)spill, total = row * n_rows;
    if (c->seam_floats < total) {
        if (c->d_seam) HIP_TRY(hipFree(c->d_seam));
        c->d_seam = nullptr;
        HIP_TRY(hipMalloc(&c->d_seam, (size_t)total * sizeof(float)));
        c->seam_floats = total;
    }
    HIP_TRY(hipMemsetAsync(c->d_seam, 0, (size_t)total * sizeof(float), p->stream));
    if (pack)
        hipLaunchKernelGGL(upx_seam_pack_kernel, dim3(grid_for(spill)), dim3(256), 0, p->stream,
                           c->d_seam + row * my_row, d_c, d_l, d_r, (long long)own_len, (long long)spill, (long long)spill);
    HIP_TRY(hipEventRecord(c->ev_start, p->stream));
    NCCL_TRY(g_rccl.AllReduce(c->d_seam, c->d_seam, (size_t)total, ncclFloat32, ncclSum, c->comm, p->stream));
    if (add_row >= 0 && add_len > 0) {
        const float* prev = c->d_seam + row * add_row;
        hipLaunchKernelGGL(upx_seam_add_kernel, dim3(grid_for(add_len)), dim3(256), 0, p->stream, d_c, d_l, d_r, prev,
                           prev + spill, prev + 2 * spill, (long long)add_len);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(c->ev_done, p->stream));
    c->pending = true;
    return UPX_OK;
}
}   // namespace

// The seam buffer of `n_ranks` rows for this spill, up front (upx_comm_seam_exchange otherwise allocates it - a
// synchronising call - inside the first exchange).
int upx_comm_reserve(upx_comm* c, int64_t spill) {
    if (!c || spill < 0) return fail(UPX_ERR_INVALID, "upx_comm_reserve: bad argument");
    const long long total = 3LL * spill * c->n_ranks;
    if (c->seam_floats >= total) return UPX_OK;
    HIP_TRY(hipSetDevice(c->plan->device));
    HIP_TRY(hipStreamSynchronize(c->plan->stream));
    if (c->d_seam) HIP_TRY(hipFree(c->d_seam));
    c->d_seam = nullptr;
    c->seam_floats = 0;
    HIP_TRY(hipMalloc(&c->d_seam, (size_t)total * sizeof(float)));
    c->seam_floats = total;
    return UPX_OK;
}

int64_t upx_comm_seam_add_len(int rank, int n_ranks, int64_t own_len, int64_t spill) {
    if (rank <= 0 || rank >= n_ranks || own_len <= 0 || spill <= 0) return 0;
    return rank + 1 == n_ranks && own_len < spill ? own_len : spill;
}

int upx_comm_seam_exchange(upx_comm* c, float* d_c, float* d_l, float* d_r, int64_t own_len, int64_t spill) {
    if (!c || !d_c || !d_l || !d_r || own_len <= 0 || spill < 0) return fail(UPX_ERR_INVALID, "upx_comm_seam_exchange: bad argument");
    if (spill == 0 || c->n_ranks == 1) return UPX_OK;
    // the last rank's spill lies beyond the signal; rank 0 has no predecessor.  A rank with a successor holds planes of
    // own_len + spill samples (>= spill); the last rank's planes may end at own_len < spill (upx_wav_shard_begin)
    const bool last = c->rank + 1 == c->n_ranks;
    return seam_exchange_impl(c, d_c, d_l, d_r, own_len, spill, c->n_ranks, c->rank, !last, c->rank > 0 ? c->rank - 1 : -1,
                              upx_comm_seam_add_len(c->rank, c->n_ranks, own_len, spill));
}

int upx_comm_seam_selftest(upx_comm* c, float* d_c, float* d_l, float* d_r, int64_t own_len, int64_t spill, int n_rows,
                           int my_row) {
    if (!c || !d_c || !d_l || !d_r || own_len <= 0 || spill <= 0 || n_rows < 1 || my_row < 0 || my_row >= n_rows)
        return fail(UPX_ERR_INVALID, "upx_comm_seam_selftest: bad argument");
    return seam_exchange_impl(c, d_c, d_l, d_r, own_len, spill, n_rows, my_row, true, my_row, spill);
}

int upx_seam_add_local(upx_plan* p, const float* pc, const float* pl, const float* pr, int64_t prev_own_len,
                       float* nc, float* nl, float* nr, int64_t spill) {
    if (!p || !pc || !pl || !pr || !nc || !nl || !nr || prev_own_len < 0 || spill < 0)
        return fail(UPX_ERR_INVALID, "upx_seam_add_local: bad argument");
    if (spill == 0) return UPX_OK;
    HIP_TRY(hipSetDevice(p->device));
    hipLaunchKernelGGL(upx_seam_add_kernel, dim3(grid_for(spill)), dim3(256), 0, p->stream, nc, nl, nr,
                       pc + prev_own_len, pl + prev_own_len, pr + prev_own_len, (long long)spill);
    HIP_TRY(hipGetLastError());
    return UPX_OK;
}

}   // extern "C"
