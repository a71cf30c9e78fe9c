// TEST INFRASTRUCTURE ONLY.  Host build (g++) of upmix_amd/csrc/upx_core.h with a
// sequential executor: every `each` phase runs for all threads of the workgroup
// before the next phase starts, which is exactly the barrier semantics of the
// gfx950 kernel.  Lets tests/test_emulator.py check the kernel's indexing, LDS
// layout and framing logic against the oracle without a GPU.  The product
// (upmix_amd) never loads this library.
#include <cmath>
#include <type_traits>
#include <cstring>
#include <functional>
#include <vector>

#include "../../upmix_amd/csrc/upx_core.h"
#include "../../upmix_amd/csrc/upx_big.h"
#include "../../upmix_amd/csrc/upx_pipeline.h"
#include "../../upmix_amd/csrc/upx_zoom.h"

#include <atomic>
#include <chrono>

namespace {
template <int PTS>
struct SeqExec {
    std::vector<upx::ThreadT<PTS>> st;
    template <class F>
    void each(F&& f) {
        for (size_t t = 0; t < st.size(); ++t) f((int)t, st[t]);
    }
    // f reads, g scatters: all reads happen before any write (barrier, or in-order LDS issue of one wave)
    template <class F, class G>
    void each2(F&& f, G&& g) {
        each(f);
        each(g);
    }
    void wg_barrier() {}
};

// Wide streams: `each` orders ONE wave only, waves meet at wg_barrier().  Between two barriers the waves are run one
// after the other, each through all of its phases (alternately first-to-last and last-to-first wave): a legal
// schedule on the GPU, and the one that exposes a missing barrier as a read of poisoned or stale LDS.
template <int PTS>
struct WaveExec {
    using Thread = upx::ThreadT<PTS>;
    std::vector<Thread> st;
    std::vector<std::function<void(int, Thread&)>> pending;
    bool flip = false;
    template <class F>
    void each(F&& f) { pending.emplace_back(f); }
    template <class F, class G>
    void each2(F&& f, G&& g) {
        pending.emplace_back(f);
        pending.emplace_back(g);
    }
    void wg_barrier() {
        const int n_waves = (int)st.size() / 64;
        for (int i = 0; i < n_waves; ++i) {
            const int w = flip ? n_waves - 1 - i : i;
            for (auto& ph : pending)
                for (int t = w * 64; t < w * 64 + 64; ++t) ph(t, st[t]);
        }
        pending.clear();
        flip = !flip;
    }
};

void turn_trig(double frac, double& c, double& s) {
    const double a = 2.0 * M_PI * frac;
    c = std::cos(a);
    s = std::sin(a);
}

long long g_interior_wgs = 0;   // workgroups of the last emu_band() call that took band_program's interior flavour

// wg_frames (optional): frames per stream of every workgroup, for streams of unequal length (BandArgs::stream_m0)
template <class C, class LV = upx::LiveAll, bool MERGED = true>
int run(upx::BandArgs a, const std::vector<int>* wg_frames = nullptr) {
    g_interior_wgs = 0;
    std::vector<upx::cf> tw((size_t)C::TW_CF);
    upx::fill_tables<C>(tw.data(), turn_trig);
    a.tw = tw.data();
    // the gain rows in the order the kernel reads them (the library does the same at plan creation)
    std::vector<float> gain((size_t)a.n_gain * a.gain_stride);
    for (int q = 0; q < a.n_gain; ++q)
        for (int i = 0; i <= C::N / 2; ++i) gain[(size_t)q * a.gain_stride + i] = a.gain[(size_t)q * a.gain_stride + upx::gain_bin<C>(i)];
    a.gain = gain.data();
    a.blocks_per_stream += a.blocks_per_stream & 1;              // the kernel needs an even F (the library rounds up too)
    const long long n_blocks = (long long)a.m_hi - a.m_lo + 1;   // streams start one frame early (frame m_lo - 1)
    if (a.m_hi <= a.m_lo) return 0;
    long long n_streams = (n_blocks + a.blocks_per_stream - 1) / a.blocks_per_stream;
    long long n_wg = (n_streams + C::G - 1) / C::G;
    std::vector<int> m0;
    if (wg_frames) {
        n_wg = (long long)wg_frames->size();
        n_streams = n_wg * C::G;
        m0.assign((size_t)n_streams + 1, a.m_lo - 1);
        for (long long w = 0; w < n_wg; ++w)
            for (int g = 0; g < C::G; ++g) m0[(size_t)(w * C::G + g) + 1] = m0[(size_t)(w * C::G + g)] + (*wg_frames)[(size_t)w];
        a.stream_m0 = m0.data();
    }
    const int tail = (C::P - C::HS) * C::LANES;
    std::vector<float> seam((size_t)n_wg * C::G * 3 * tail, NAN);
    a.seam = seam.data();
    std::vector<upx::cf> lds((size_t)C::LDS_CF);
    for (long long wg = 0; wg < n_wg; ++wg) {
        typename std::conditional<C::WIDE, WaveExec<C::P>, SeqExec<C::P>>::type ex;
        ex.st.resize(C::WG);
        // poison LDS so that reads of never-written cells are visible
        for (auto& v : lds) v = upx::mk(NAN, NAN);
        g_interior_wgs += upx::band_interior<C>(a, (int)wg) ? 1 : 0;
        upx::band_program_auto<C, decltype(ex), MERGED, LV>(ex, a, lds.data(), (int)wg);
    }
    for (long long g = 0; g < n_streams * tail; ++g) upx::stream_seam_add(a, (int)n_streams, tail, C::HOP, g);
    return 0;
}
}   // namespace

extern "C" long long emu_last_interior_wgs() { return g_interior_wgs; }

extern "C" int emu_band(int log2n, int k_overlap, int pts, const float* in, long long t_in, float* out_c, float* out_l,
                        float* out_r, long long t_out, const float* w_a, const float* w_s_scaled,
                        const float* gain_half, int j_lo, int j_hi, int m_lo, int m_hi, int blocks_per_stream,
                        int accumulate, int n_gain) {
    upx::BandArgs a;
    std::memset(&a, 0, sizeof a);
    a.in = reinterpret_cast<const upx::cf*>(in);
    a.out_c = out_c; a.out_l = out_l; a.out_r = out_r;
    a.w_a = w_a; a.w_s = w_s_scaled; a.gain = gain_half;
    a.t_in = (int)t_in; a.t_out = (int)t_out;
    a.j_lo = j_lo; a.j_hi = j_hi; a.m_lo = m_lo; a.m_hi = m_hi;
    a.blocks_per_stream = blocks_per_stream; a.accumulate = accumulate;
    a.n_gain = n_gain; a.gain_stride = (1 << log2n) / 2 + 1;
#define UPX_CASE(L, K, PP) if (log2n == L && k_overlap == K && pts == PP) return run<upx::Cfg<L, K, PP>>(a);
    UPX_CASE(8, 4, 16) UPX_CASE(9, 4, 16) UPX_CASE(10, 4, 16) UPX_CASE(11, 4, 16) UPX_CASE(12, 4, 16) UPX_CASE(13, 4, 16)
    UPX_CASE(8, 2, 16) UPX_CASE(10, 2, 16) UPX_CASE(10, 8, 16) UPX_CASE(12, 8, 16) UPX_CASE(13, 2, 16)
    UPX_CASE(8, 4, 8) UPX_CASE(9, 4, 8) UPX_CASE(10, 4, 8) UPX_CASE(11, 4, 8) UPX_CASE(12, 4, 8) UPX_CASE(13, 4, 8)
    UPX_CASE(8, 2, 8) UPX_CASE(10, 2, 8) UPX_CASE(10, 8, 8) UPX_CASE(12, 8, 8) UPX_CASE(13, 2, 8)
#undef UPX_CASE
    // pts == 0: wide streams (N = 4096, 8192)
#define UPX_WIDE(L, K) if (log2n == L && k_overlap == K && pts == 0) return run<upx::WideCfg<L, K>>(a);
    UPX_WIDE(12, 4) UPX_WIDE(13, 4) UPX_WIDE(12, 2) UPX_WIDE(12, 8) UPX_WIDE(13, 2) UPX_WIDE(13, 8)
#undef UPX_WIDE
    return -1;
}

// Streams of unequal length: workgroup w's streams transform wg_frames[w] frames each (even, >= K; the caller makes the sum
// cover the signal).  Same arguments as emu_band otherwise (16 points per lane, general flavour).
extern "C" int emu_band_uneven(int log2n, int k_overlap, const int* wg_frames, int n_wg, const float* in, long long t_in,
                               float* out_c, float* out_l, float* out_r, long long t_out, const float* w_a,
                               const float* w_s_scaled, const float* gain_half, int j_lo, int j_hi, int m_lo, int m_hi,
                               int accumulate) {
    upx::BandArgs a;
    std::memset(&a, 0, sizeof a);
    a.in = reinterpret_cast<const upx::cf*>(in);
    a.out_c = out_c; a.out_l = out_l; a.out_r = out_r;
    a.w_a = w_a; a.w_s = w_s_scaled; a.gain = gain_half;
    a.t_in = (int)t_in; a.t_out = (int)t_out;
    a.j_lo = j_lo; a.j_hi = j_hi; a.m_lo = m_lo; a.m_hi = m_hi;
    a.blocks_per_stream = 2; a.accumulate = accumulate;
    a.n_gain = 1; a.gain_stride = (1 << log2n) / 2 + 1;
    const std::vector<int> frames(wg_frames, wg_frames + n_wg);
#define UPX_UNEVEN(L, K) if (log2n == L && k_overlap == K) return run<upx::Cfg<L, K, 16>>(a, &frames);
    UPX_UNEVEN(8, 4) UPX_UNEVEN(10, 4) UPX_UNEVEN(11, 4) UPX_UNEVEN(9, 2)
#undef UPX_UNEVEN
    return -1;
}

// The single-band flavour specialised for the live own-bin slots [s0, s1) (upx::Live<s0, s1>): same arguments as
// emu_band with n_gain == 1; the caller guarantees that the gain vector is zero outside those slots.
extern "C" int emu_band_live(int log2n, int k_overlap, int s0, int s1, const float* in, long long t_in, float* out_c,
                             float* out_l, float* out_r, long long t_out, const float* w_a, const float* w_s_scaled,
                             const float* gain_half, int j_lo, int j_hi, int m_lo, int m_hi, int blocks_per_stream,
                             int accumulate) {
    upx::BandArgs a;
    std::memset(&a, 0, sizeof a);
    a.in = reinterpret_cast<const upx::cf*>(in);
    a.out_c = out_c; a.out_l = out_l; a.out_r = out_r;
    a.w_a = w_a; a.w_s = w_s_scaled; a.gain = gain_half;
    a.t_in = (int)t_in; a.t_out = (int)t_out;
    a.j_lo = j_lo; a.j_hi = j_hi; a.m_lo = m_lo; a.m_hi = m_hi;
    a.blocks_per_stream = blocks_per_stream; a.accumulate = accumulate;
    a.n_gain = 1; a.gain_stride = (1 << log2n) / 2 + 1;
#define UPX_LIVE(L, K, A, B) \
    if (log2n == L && k_overlap == K && s0 == A && s1 == B) return run<upx::Cfg<L, K, 16>, upx::Live<A, B>, false>(a);
    UPX_LIVE(10, 4, 0, 2) UPX_LIVE(10, 4, 0, 3) UPX_LIVE(10, 4, 0, 4) UPX_LIVE(11, 4, 0, 2) UPX_LIVE(11, 4, 0, 3)
    UPX_LIVE(11, 4, 0, 4) UPX_LIVE(8, 4, 1, 8) UPX_LIVE(9, 4, 1, 8) UPX_LIVE(10, 4, 1, 8) UPX_LIVE(10, 4, 0, 8)
    UPX_LIVE(8, 4, 0, 5) UPX_LIVE(10, 2, 1, 6) UPX_LIVE(10, 8, 0, 1) UPX_LIVE(11, 4, 1, 2)
#undef UPX_LIVE
    return -1;
}

// ---- unfused path (large STFT sizes and arbitrary hops), same chunk loop as the library ----------
namespace {
template <class B>
int run_big(upx::BigArgs a, int m_lo, int m_hi, int ch_max) {
    using Row = typename B::Row;
    std::vector<upx::cf> tw_n((size_t)B::N), tw_rows((size_t)Row::TW_CF);
    if (B::N1 == 16) upx::fill_big_twiddles<B>(tw_n.data(), turn_trig);
    upx::fill_twiddles<Row>(tw_rows.data(), turn_trig);
    // chunk geometry: halo frames (odd count so that every chunk starts on an odd frame), emitted blocks (even),
    // plus one trailing pair-partner frame
    const int halo = (a.kf - 1) | 1;
    int emit = (ch_max - halo - 1) & ~1;
    if (emit < 2) emit = 2;
    const int ch = emit + halo + 1;
    std::vector<upx::cf> z((size_t)ch * B::N), y((size_t)ch * B::N), yc((size_t)(ch / 2) * B::N);
    std::vector<upx::cf> lds((size_t)Row::LDS_CF);
    a.tw_n = tw_n.data(); a.tw_rows = tw_rows.data();
    a.z = z.data(); a.y = y.data(); a.yc = yc.data();
    // the gain rows in the order the kernels read them (the library does the same at plan creation)
    std::vector<float> gain_perm((size_t)a.n_gain * a.gain_stride);
    for (int q = 0; q < a.n_gain; ++q)
        for (int i = 0; i <= B::N / 2; ++i)
            gain_perm[(size_t)q * a.gain_stride + i] = a.gain[(size_t)q * a.gain_stride + upx::big_gain_bin<B>(i)];
    a.gain = gain_perm.data();
    a.ch = ch;
    auto rows = [&](upx::cf* buf, int n_rows) {
        for (int wg = 0; wg < (n_rows + Row::G - 1) / Row::G; ++wg) {
            SeqExec<Row::P> ex;
            ex.st.resize(Row::WG);
            for (auto& v : lds) v = upx::mk(NAN, NAN);
            upx::big_rows_program<B>(ex, buf, a.tw_rows, lds.data(), wg, n_rows);
        }
    };
    for (int m0 = m_lo; m0 < m_hi; m0 += emit) {
        a.j0 = m0 - halo;
        a.m0 = m0;
        a.m1 = m0 + emit < m_hi ? m0 + emit : m_hi;
        for (auto& v : z) v = upx::mk(NAN, NAN);
        // the library's choice: N = 16 384 holds the whole frame in one workgroup (steps 1 and 2 inside the fused
        // kernel), 32 768 / 65 536 take one mirror pair of rows per workgroup between separate step kernels
        constexpr int kMidRows = (B::N1 == 16 && 16 * Row::LANES == B::N2 && (16 * Row::PITCH + Row::TW_CF) * 8 <= 160 * 1024) ? 16 : 2;
        if constexpr (B::N1 == 16) {
            if (kMidRows == 2)
                for (long long g = 0; g < (long long)ch * B::N2; ++g) upx::big_step1_audio<B>(a, g);
            std::vector<upx::cf> lds2((size_t)(kMidRows * Row::PITCH + Row::TW_CF));
            for (int wg = 0; wg < (ch / 2) * (kMidRows == 2 ? 8 : 1); ++wg) {
                SeqExec<Row::P> ex;
                ex.st.resize(kMidRows * Row::LANES);
                for (auto& v : lds2) v = upx::mk(NAN, NAN);
                upx::big_mid_program<B, kMidRows>(ex, a, lds2.data(), wg);
            }
            if (kMidRows == 2) {
                for (long long g = 0; g < (long long)ch * B::N2; ++g) upx::big_step2_inv<B>(a.y, a.tw_n, ch, g);
                for (long long g = 0; g < (long long)(ch / 2) * B::N2; ++g) upx::big_step2_inv<B>(a.yc, a.tw_n, ch / 2, g);
            }
        } else {
            for (int wg = 0; wg < (ch + Row::G - 1) / Row::G; ++wg) {
                SeqExec<Row::P> ex;
                ex.st.resize(Row::WG);
                for (auto& v : lds) v = upx::mk(NAN, NAN);
                upx::big_frame_program<B>(ex, a, lds.data(), wg);
            }
            for (long long g = 0; g < upx::big_mask_threads<B>(ch / 2); ++g) upx::big_mask<B>(a, g);
            rows(a.y, ch * B::N1);
            rows(a.yc, (ch / 2) * B::N1);
        }
        for (long long g = 0; g < (long long)(a.m1 - a.m0) * a.hop; ++g) upx::big_ola<B>(a, g);
    }
    return 0;
}
}   // namespace

extern "C" int emu_big_band(int log2n, int hop, const float* in, long long t_in, float* out_c, float* out_l,
                            float* out_r, long long t_out, const float* w_a, const float* w_s_scaled,
                            const float* gain_half, int j_lo, int j_hi, int m_lo, int m_hi, int chunk_frames,
                            int accumulate, int n_gain) {
    upx::BigArgs a;
    std::memset(&a, 0, sizeof a);
    a.in = reinterpret_cast<const upx::cf*>(in);
    a.out_c = out_c; a.out_l = out_l; a.out_r = out_r;
    a.w_a = w_a; a.w_s = w_s_scaled; a.gain = gain_half;
    a.t_in = (int)t_in; a.t_out = (int)t_out;
    a.hop = hop; a.kf = ((1 << log2n) + hop - 1) / hop;
    a.j_lo = j_lo; a.j_hi = j_hi; a.accumulate = accumulate;
    a.n_gain = n_gain; a.gain_stride = (1 << log2n) / 2 + 1;
#define UPX_BIG(L) if (log2n == L) return run_big<upx::BigCfg<L>>(a, m_lo, m_hi, chunk_frames);
    UPX_BIG(6) UPX_BIG(7) UPX_BIG(8) UPX_BIG(9) UPX_BIG(10) UPX_BIG(11) UPX_BIG(12) UPX_BIG(13) UPX_BIG(14) UPX_BIG(15) UPX_BIG(16)
#undef UPX_BIG
    return -1;
}

// ---- hand-over logic of the streamed host calls (upx_pipeline.h) with injected failures ----------
// submit / complete sleep `*_us` microseconds; item `fail_submit` / `fail_complete` (-1: none) returns -3.
// Reports how many items each side finished.  A regression of the round-1 hang shows up as a test timeout.
// upx::wav_schedule (upx_pipeline.h): -> number of chunks; the first `cap` of them as rows (start, own, t_in, t_out)
extern "C" int emu_wav_schedule(long long t_in, long long own_len, long long t_out, long long grid, long long spill, long long chunk,
                                int uniform, int bytes_per_frame, double rate, long long* rows, int cap) {
    std::vector<upx::WavChunkRec> v;
    upx::wav_schedule(t_in, own_len, t_out, grid, spill, chunk, uniform != 0, bytes_per_frame, rate, v);
    for (size_t i = 0; i < v.size() && (int)i < cap; ++i) {
        rows[4 * i] = v[i].start; rows[4 * i + 1] = v[i].own; rows[4 * i + 2] = v[i].t_in; rows[4 * i + 3] = v[i].t_out;
    }
    return (int)v.size();
}

extern "C" int emu_pipeline(long long n_items, long long fail_submit, long long fail_complete, int submit_us,
                            int complete_us, long long* n_submitted, long long* n_completed, char* msg, int msg_len) {
    std::atomic<long long> subs{0}, comps{0};
    std::atomic<bool> order_ok{true};
    auto submit = [&](int64_t i, std::string& m) -> int {
        // buffer set i % 2 is free: item i-2 is complete
        if (i >= 2 && comps.load() < i - 1) order_ok = false;
        std::this_thread::sleep_for(std::chrono::microseconds(submit_us));
        if (i == fail_submit) { m = "injected submit failure"; return -3; }
        subs = i + 1;
        return 0;
    };
    auto complete = [&](int64_t i, std::string& m) -> int {
        if (subs.load() < i + 1) order_ok = false;   // never complete what has not been submitted
        std::this_thread::sleep_for(std::chrono::microseconds(complete_us));
        if (i == fail_complete) { m = "injected download failure"; return -3; }
        comps = i + 1;
        return 0;
    };
    std::string err;
    const int rc = upx::run_pipeline((int64_t)n_items, submit, complete, err);
    *n_submitted = subs;
    *n_completed = comps;
    if (msg && msg_len > 0) {
        std::strncpy(msg, err.c_str(), (size_t)msg_len - 1);
        msg[msg_len - 1] = 0;
    }
    return order_ok ? rc : -100;
}

// ---- band-limited ("zoom") path: analysis kernel, synthesis kernel, stream seams (upx_zoom.h) ------------
namespace {
// pairs of the analysis dealt by age (ZoomArgs::deal_rows), set by emu_zoom_set_deal for the calls that follow
int g_deal_rows = 0;
std::vector<int> g_deal_tab;

// streams: uniform (blocks_per_stream for both roles, or F / c_split for the centre), or - tab_lr != nullptr - the two
// tables of first frames (n + 1 entries each; ZoomArgs::stream_m0 / stream_m0_c)
template <class Z>
int run_zoom(upx::ZoomArgs a, int pairs_per_wg, int c_split, const int* tab_lr = nullptr, int n_lr = 0,
             const int* tab_c = nullptr, int n_c = 0) {
    using Sub = typename Z::Sub;
    std::vector<upx::cf> tw((size_t)Z::TW_CF), ramp(upx::zoom_ramp_count(a.n, Z::P));
    upx::fill_twiddles<Sub>(tw.data(), turn_trig);
    upx::fill_zoom_ramp(ramp.data(), a.n, Z::P, turn_trig);
    a.tw = tw.data();
    a.ramp = ramp.data();
    a.d = a.n / Z::P;
    a.blocks_per_stream += a.blocks_per_stream & 1;
    if (a.m_hi <= a.m_lo) return 0;
    const int groups = a.d / Z::RG;
    const size_t tail = (size_t)(Z::K - 1) * a.hop;
    long long n_streams, n_streams_c;
    int n_frames;
    if (tab_lr) {
        if (tab_lr[0] != a.m_lo - 1 || tab_c[0] != tab_lr[0] || tab_c[n_c] != tab_lr[n_lr]) return -3;
        for (int i = 0; i < n_lr; ++i)
            if ((tab_lr[i + 1] - tab_lr[i]) % 2 != 0 || tab_lr[i + 1] - tab_lr[i] < Z::K) return -3;
        for (int i = 0; i < n_c; ++i)
            if ((tab_c[i + 1] - tab_c[i]) % 2 != 0 || tab_c[i + 1] - tab_c[i] < Z::K) return -3;
        if (tab_lr[n_lr] < a.m_hi) return -3;   // the streams cover frames m_lo - 1 .. m_hi - 1
        n_streams = n_lr; n_streams_c = n_c;
        n_frames = tab_lr[n_lr] - tab_lr[0];
        a.stream_m0 = tab_lr; a.stream_m0_c = tab_c;
    } else {
        const int F = a.blocks_per_stream;
        n_streams = ((long long)a.m_hi - a.m_lo + 1 + F - 1) / F;
        n_frames = (int)(n_streams * F);
        // centre streams of their own length (ZoomArgs::blocks_per_stream_c): c_split per Ls/Rs stream
        if (c_split > 1 && (F % (2 * c_split) != 0 || F / c_split < Z::K)) return -2;
        const int Fc = F / c_split;
        n_streams_c = ((long long)a.m_hi - a.m_lo + 1 + Fc - 1) / Fc;
        if (c_split > 1) a.blocks_per_stream_c = Fc;
    }
    a.f0 = a.m_lo - 1;
    std::vector<upx::cf> y((size_t)n_frames * Z::P, upx::mk(NAN, NAN)), yc((size_t)(n_frames / 2) * Z::P, upx::mk(NAN, NAN));
    a.y = y.data();
    a.yc = yc.data();
    std::vector<float> seam((size_t)n_streams * 2 * tail, NAN), seam_c((size_t)n_streams_c * tail, NAN);
    a.seam = seam.data();
    a.seam_c = seam_c.data();
    std::vector<upx::cf> lds((size_t)Z::LDS_CF);
    // analysis: pairs q0 .. q0 + n_frames/2 - 1
    a.pair0 = a.m_lo / 2;
    a.pair_end = a.pair0 + n_frames / 2;
    a.pairs_per_wg = pairs_per_wg;   // workgroups per XCD label: the grid is 8 x this
    a.deal_rows = (int)g_deal_tab.size() == 2 * pairs_per_wg ? g_deal_rows : 0;
    a.deal_tab = g_deal_tab.data();
    for (int wg = 0; wg < 8 * pairs_per_wg; ++wg) {
        WaveExec<16> ex;
        ex.st.resize(Z::WG);
        for (auto& v : lds) v = upx::mk(NAN, NAN);
        upx::zoom_analysis_program<Z>(ex, a, lds.data(), wg);
    }
    a.stream0 = 0; a.stream0_c = 0;
    a.ns_lr = (int)n_streams; a.ns_c = (int)n_streams_c;
    for (int role = 0; role < 2; ++role)
        for (long long sid = 0; sid < (role ? n_streams_c : n_streams); ++sid)
            for (int grp = 0; grp < groups; ++grp) {
                WaveExec<16> ex;
                ex.st.resize(Z::WG);
                for (auto& v : lds) v = upx::mk(NAN, NAN);
                upx::zoom_synthesis_program<Z>(ex, a, lds.data(), (int)sid, grp, role);
            }
    for (long long g = 0; g < (n_streams + n_streams_c) * (long long)tail; ++g)
        upx::zoom_seam_add(a, (int)n_streams, (int)n_streams_c, (int)tail, g);
    return 0;
}
}   // namespace

extern "C" void emu_zoom_set_deal(int rows, const int* tab, int n_l) {   // tab: [n_l][2] (ZoomArgs::deal_tab)
    g_deal_rows = rows;
    g_deal_tab.assign(tab, tab + (rows > 0 ? 2 * n_l : 0));
}

extern "C" int emu_zoom_band_streams(int log2n, int k_overlap, int log2p, const float* in, long long t_in, float* out_c,
                                     float* out_l, float* out_r, long long t_out, const float* w_a, const float* w_s_scaled,
                                     const float* gain_half, int j_lo, int j_hi, int m_lo, int m_hi, int blocks_per_stream,
                                     int accumulate, int n_gain, int pairs_per_wg, int c_split, const int* tab_lr, int n_lr,
                                     const int* tab_c, int n_c) {
    upx::ZoomArgs a;
    std::memset(&a, 0, sizeof a);
    a.in = reinterpret_cast<const upx::cf*>(in);
    a.out_c = out_c; a.out_l = out_l; a.out_r = out_r;
    a.w_a = w_a; a.w_s = w_s_scaled; a.gain = gain_half;
    a.n = 1 << log2n; a.hop = a.n / k_overlap;
    a.t_in = (int)t_in; a.t_out = (int)t_out;
    a.j_lo = j_lo; a.j_hi = j_hi; a.m_lo = m_lo; a.m_hi = m_hi;
    a.blocks_per_stream = blocks_per_stream; a.accumulate = accumulate;
    a.n_gain = n_gain; a.gain_stride = a.n / 2 + 1;
    const int d = a.n >> log2p;
    const int rg = d >= 16 ? 16 : d;
#define UPX_ZOOM(LP, RG, K) if (log2p == LP && rg == RG && k_overlap == K) return run_zoom<upx::ZoomCfg<LP, RG, K>>(a, pairs_per_wg, c_split, tab_lr, n_lr, tab_c, n_c);
#define UPX_ZOOM_K(LP, RG) UPX_ZOOM(LP, RG, 2) UPX_ZOOM(LP, RG, 4) UPX_ZOOM(LP, RG, 8)
    UPX_ZOOM_K(8, 4) UPX_ZOOM_K(8, 8) UPX_ZOOM_K(8, 16) UPX_ZOOM_K(9, 4) UPX_ZOOM_K(9, 8) UPX_ZOOM_K(9, 16)
    UPX_ZOOM_K(10, 4) UPX_ZOOM_K(10, 8) UPX_ZOOM_K(10, 16)
#undef UPX_ZOOM_K
#undef UPX_ZOOM
    return -1;
}

extern "C" int emu_zoom_band(int log2n, int k_overlap, int log2p, const float* in, long long t_in, float* out_c,
                             float* out_l, float* out_r, long long t_out, const float* w_a, const float* w_s_scaled,
                             const float* gain_half, int j_lo, int j_hi, int m_lo, int m_hi, int blocks_per_stream,
                             int accumulate, int n_gain, int pairs_per_wg) {
    return emu_zoom_band_streams(log2n, k_overlap, log2p, in, t_in, out_c, out_l, out_r, t_out, w_a, w_s_scaled, gain_half, j_lo,
                                 j_hi, m_lo, m_hi, blocks_per_stream, accumulate, n_gain, pairs_per_wg, 1, nullptr, 0, nullptr, 0);
}
