// upx_zoom.h - band-limited bands: pruned ("zoom") STFT analysis and synthesis for gfx950.
//
// The reference's planner couples the STFT size to the band's low edge (center_extraction.py:173-197:
// N = nextpow2(32 sr / f_low)), so whatever N is, the pass band of a band - the bins where its _band_limit gain
// (:282-332) is non-zero - ends near bin 200: 206 of 4097 bins at N = 8192, 206 of 32 769 at N = 65 536.  Bins
// with zero gain contribute exactly nothing (L = R = 0 there, :373-384 give C = Ls = Rs = 0), so only
//     Z[k], |k| <= kmax      (Z = FFT_N of (L + iR) w_A; both signs because L and R are separated by symmetry)
// has to be computed, and the inverse transforms have non-zero input only there.  With P = a power of two
// > 2 kmax and D = N / P, split n = D m + r (residue r < D, decimated index m < P):
//     analysis    Z[k]      = sum_r W_N^(r k) F_r[k mod P],      F_r = FFT_P over m of (x w_A)[D m + r]
//     synthesis   y[D m + r] = IFFT_P over k of ( Y[k] W_N^(-r k) ) [m]
// i.e. D transforms of P points plus one "ramp" multiply each way instead of a transform of N points: log2 P
// levels instead of log2 N (9 of 13 at N = 8192, 9 of 16 at 65 536), a mask over P/2 bins instead of N/2, and - the
// larger gain - spectra small enough (P complex per frame) to pass through memory, so the path is TWO kernels
// with no shared register state:
//     zoom_analysis   frame -> Z -> L/R split, gains, mask -> swap(Ls + i Rs)[P], swap(Ca + i Cb)[P] per pair
//     zoom_synthesis  residue streams: for residue r the samples D m + r form a decimated signal whose frames
//                     are P long with hop P/K, so one stream = P/16 lanes walking frames with the overlap-add state
//                     in registers, exactly like the fused kernel's inverse half (upx_core.h), one per residue.
// Neither kernel holds what the other needs (no overlap-add state in the analysis, no forward / mask state in
// the synthesis; Ls/Rs and C streams are separate workgroups), so both run at 3 or 4 waves per SIMD (168 / 128
// VGPRs: whatever the LDS footprint of the configuration admits, ZoomCfg::WPE_A / WPE_S) where the fused kernel runs 2,
// with room to request every global access well ahead of its use.  It also lifts the size limit: a frame never has
// to fit LDS (N = 16 384 .. 65 536 run here without the scratch passes of upx_big.h whenever the band is
// band-limited, which the reference's planner makes them).
//
// Three rules shaped the loops (DESIGN.md 5c has the measurements):
//   * vector-memory loads return in issue order: a load needed soon is never issued behind one that may miss to HBM
//     (short loads first, prefetches last, each prefetch waited for a whole unit / transform later);
//   * the backend's s_waitcnt bookkeeping is exact only when every path between a load and its use issues the same
//     number of loads and stores: the steady state of the synthesis (interior streams) runs a body without branches
//     around memory operations, instantiated for accumulate / not;
//   * LDS stores and paired loads are banked over 32 dwords in groups of 16 contiguous lanes: the residue buffers'
//     pitch interleaves the residues of such a group (ZoomCfg::BUF).
//
// Layouts.  A workgroup holds RG = min(D, 16) residues as RG sub-FFT buffers of P points (Stream<Cfg<log2 P>>):
//     wave-local  thread t = (g = t / SL, sl = t % SL), SL = P/16: sub-FFT g lives in SL lanes of one wave
//                 (Stockham passes through that wave's LDS region with wave-level ordering only);
//     coalesced   thread t = (rho = t % RG, sl = t / RG): slot s is sample D (sl + SL s) + rho of the frame, so
//                 consecutive lanes touch consecutive samples (global loads / stores of whole cache lines).
// The first pass of a forward sub-FFT and the last pass of an inverse one are radix-16 butterflies over the 16
// slots of ONE thread in either layout, so they run in the coalesced layout next to the global access and the
// exchange with the wave-local passes is the ordinary Stockham scatter / read through LDS, across waves: one
// workgroup barrier each way, as in the wide streams of upx_core.h.
//
// Conventions are those of upx_core.h: frame pairs (odd j, j+1) share one centre transform, inverse transforms
// by re/im swap, overlap-add in increasing j from 0.0f, stream tails through the seam buffer
// (stream_seam_add), bands summed in list order.
#pragma once
#include "upx_core.h"

namespace upx {

struct ZoomArgs {
    const cf* in;          // interleaved stereo, local sample 0
    float* out_c;
    float* out_l;
    float* out_r;
    const float* w_a;      // analysis window [N]
    const float* w_s;      // synthesis window / N [N]
    const float* gain;     // 0.5 * band-limit gain, [n_gain][gain_stride], natural bin order (merged bands: BandArgs)
    const cf* tw;          // compact twiddle table of the P-point sub-FFT
    const cf* ramp;        // ramp seeds [D][P/16 + 4] (zoom_ramp): W_N^(r ks), ks = k < P/2 ? k : k - P, on the fly
    cf* y;                 // [frames][P]: swap(Ls + i Rs) of frame j at (j - f0) P
    cf* yc;                // [pairs][P]:  swap(Ca + i Cb) of pair q = (2q-1, 2q) at (q - (f0+1)/2) P
    float* seam;           // [streams][3][(K-1) hop] stream tails (BandArgs::seam)
    int n, d, hop;         // STFT size N, decimation D = N / P, hop = N / K
    int t_in, t_out;
    int j_lo, j_hi;        // frames that exist
    int m_lo, m_hi;        // hop-blocks to emit
    int blocks_per_stream; // F (even)
    int n_gain, gain_stride;
    int accumulate;
    int f0;                // first frame held in y (odd)
    int pair0, pair_end;   // analysis launch: pairs [pair0, pair_end)
    int pairs_per_wg;      // analysis launch: workgroups per XCD (grid = 8 x this; see zoom_analysis_program)
    int stream0;           // synthesis launch: first stream
    // Streams of the synthesis.  Ls/Rs stream sid transforms the frames [first, first + frames): uniform, first =
    // m_lo - 1 + sid F with F = blocks_per_stream, or - stream_m0 != nullptr - [stream_m0[sid], stream_m0[sid + 1]) (odd
    // starts, even lengths >= K).  The centre streams are cut independently (blocks_per_stream_c / stream_m0_c; Fc = 0 and
    // no table: like the Ls/Rs streams): a centre transform serves two frames and costs about as much as an Ls/Rs one,
    // so a centre stream of the same running time is about twice as long - which lets a launch fill every resident
    // workgroup slot exactly once with workgroups that end together (upx_process_device), instead of dispatching a
    // second and a third helping into slots that run dry one by one at the end (measured per workgroup,
    // scripts/phase_prof/zwgtime.hip: 12 % of the synthesis' slot time).
    // Tails: seam[sid][2][(K-1) hop] (Ls, Rs) and seam_c[sid][(K-1) hop].
    int blocks_per_stream_c;   // Fc (even, >= K; 0: F)
    const int* stream_m0;      // [Ls/Rs streams + 1]
    const int* stream_m0_c;    // [centre streams + 1]
    float* seam_c;
    int stream0_c;             // synthesis launch: first centre stream
    int ns_lr, ns_c;           // synthesis launch: streams of either role (grid = (ns_lr + ns_c) x residue groups)
    // synthesis launch that fills every slot once: like the analysis below, a transform pair each (0: off)
    int prio_split_s, prio_rounds_s;
    // analysis launch (every resident slot once, a static share of the pairs each): the workgroups that share a CU were
    // dispatched in `prio_rounds` rounds of `prio_split` and the hardware favours the older ones (BandArgs::prio_split);
    // they take the top priority in turn, a pair each.  0: off.
    int prio_split, prio_rounds;
    // ... and with the turns the older workgroups still end first.  deal_rows > 0: the XCD's l-th workgroup takes pair l
    // of each of the first deal_rows rows of n_l pairs, then deal_tab[l][1] consecutive pairs that start deal_tab[l][0]
    // pairs behind those rows: low l = dispatched first = older = faster, so they are dealt more and all end together
    // (upx_process_device deals by age).  0: rows of n_l pairs until the XCD's share ends.
    int deal_rows;
    const int* deal_tab;   // [n_l][2]
};

// first frame / frames of synthesis stream sid of a role (0: Ls/Rs, 1: centre)
UPX_HD int zoom_stream_first(const ZoomArgs& a, int role, int sid) {
    const int* tab = role ? a.stream_m0_c : a.stream_m0;
    if (tab) {
#if defined(__HIP_DEVICE_COMPILE__)
        return ((const UPX_GLOBAL int*)tab)[sid];
#else
        return tab[sid];
#endif
    }
    return a.m_lo - 1 + sid * (role && a.blocks_per_stream_c > 0 ? a.blocks_per_stream_c : a.blocks_per_stream);
}
UPX_HD int zoom_stream_frames(const ZoomArgs& a, int role, int sid) {
    if (role ? a.stream_m0_c != nullptr : a.stream_m0 != nullptr)
        return zoom_stream_first(a, role, sid + 1) - zoom_stream_first(a, role, sid);
    return role && a.blocks_per_stream_c > 0 ? a.blocks_per_stream_c : a.blocks_per_stream;
}

template <int LOG2P_, int RG_, int K_>
struct ZoomCfg {
    static constexpr int LOG2P = LOG2P_;
    static constexpr int P = 1 << LOG2P_;     // decimated frame length
    static constexpr int RG = RG_;            // residues per workgroup
    static constexpr int K = K_;              // frames overlapping a sample
    static constexpr int PTS = 16;
    static constexpr int HS = 16 / K_;        // register slots per (decimated) hop
    using Sub = Cfg<LOG2P_, K_, 16>;
    static constexpr int SL = Sub::LANES;     // lanes per sub-FFT
    static constexpr int WG = RG_ * SL;
    // Distance between the residues' buffers.  LDS stores and paired loads (ds_write_b64, ds_read2_b64) are served in
    // groups of 16 contiguous lanes over 32 dword banks, i.e. a group is conflict-free when its 16 complex indices
    // differ mod 16; plain ds_read_b64 in groups of 32 lanes over 64 banks (MI355X_MICROARCH.md, LDS).  In the
    // coalesced layout 16 contiguous lanes are RG residues x 16/RG consecutive lanes of each, whose Stockham patterns
    // (17 sl + r, padp(sl) + s SPITCH) advance by 1 mod 16 per lane: a pitch = 16/RG (mod 16) interleaves the
    // residues into the gaps, and the extra 16 moves the second sub-FFT of a 32-lane group to the other 32 banks.
    // (Sub::PITCH carries a spare row the fused kernel's mirror addressing needs; P + P/16 is enough here.)
    static constexpr int BUF = (P + P / 16 + 31) / 32 * 32 + 16 + 16 / RG_;
    static constexpr int SP = Sub::SPITCH;
    static constexpr int TW_CF = Sub::TW_CF;
    static constexpr int LDS_A_CF = RG_ * BUF + TW_CF;     // analysis: sub-FFT buffers, twiddles
    // synthesis: + the staging row of the next spectrum, + (workgroups of up to 4 waves) the ramp seeds of its
    // residues, so that the only vector-memory loads of a transform are the prefetches (SEEDS_LDS)
    static constexpr bool SEEDS_LDS = WG <= 256;
    static constexpr int LDS_S_CF = LDS_A_CF + P + (SEEDS_LDS ? RG_ * (SL + 4) : 0);
    static constexpr int LDS_CF = LDS_S_CF;
    // Waves per SIMD the 160 KB of LDS admit (at most 4); the register allocator is given exactly that many
    // (fewer waves -> more registers, never the other way round).
    static constexpr int waves_by_lds(int lds_cf) { return (160 * 1024 / (lds_cf * 8)) * (WG / 64) / 4; }
    // The analysis requests a unit's input one unit ahead when its workgroup is small enough for three of them per
    // CU at 168 VGPRs (the 16 + 16 prefetched slots live next to the working set).  Workgroups of 8 or 16 waves
    // (RG = 16 at P >= 512) would drop to ONE per CU that way: they stay at 128 VGPRs and load at the top of the unit.
    static constexpr bool PREFETCH_A = WG <= 256;
    static constexpr int WPE_A_MAX = PREFETCH_A ? 3 : 4;
    static constexpr int WPE_A = waves_by_lds(LDS_A_CF) >= WPE_A_MAX ? WPE_A_MAX
                                                                      : (waves_by_lds(LDS_A_CF) < 1 ? 1 : waves_by_lds(LDS_A_CF));
    static constexpr int WPE_S = waves_by_lds(LDS_S_CF) >= 4 ? 4 : (waves_by_lds(LDS_S_CF) < 1 ? 1 : waves_by_lds(LDS_S_CF));
    static constexpr int BPT = (P / 2 + WG - 1) / WG;   // bins per thread in the mask phase
    static_assert(RG_ == 4 || RG_ == 8 || RG_ == 16, "residues per workgroup");
    static_assert(WG % 64 == 0 && WG <= 1024, "whole waves");
    static_assert(SL <= 64 && 64 % SL == 0, "a sub-FFT lives inside one wave");
};

// The ramp W_N^(r ks), ks = k < P/2 ? k : k - P, of the 16 bins k = sl + SL s a thread holds for residue r, built from
// five table values instead of sixteen loads: ramp_s = b0 q^s with b0 = W_N^(r sl), q = W_N^(r SL); for s >= 8 the
// signed wrap adds W_N^(-r P), folded into the seed of bit 3.  Seeds per residue: [SL] b0, then q, q^2, q^4,
// q^8 W_N^(-r P) (row pitch SL + 4); every ramp value is a product of at most five table values.
template <int SL>
UPX_HD void zoom_ramp(const UPX_GLOBAL cf* seeds, int r, int sl, cf* rv) {
    const UPX_GLOBAL cf* row = seeds + (size_t)r * (SL + 4);
    const cf b0 = row[sl], q1 = row[SL], q2 = row[SL + 1], q4 = row[SL + 2], q8 = row[SL + 3];
    rv[0] = b0;
    rv[1] = cmul(b0, q1);
    rv[2] = cmul(b0, q2);
    rv[3] = cmul(rv[2], q1);
    rv[4] = cmul(b0, q4);
    rv[5] = cmul(rv[4], q1);
    rv[6] = cmul(rv[4], q2);
    rv[7] = cmul(rv[6], q1);
#pragma unroll
    for (int s = 8; s < 16; ++s) rv[s] = cmul(rv[s - 8], q8);
}
// x[s] = v(s) * ramp_s without holding the sixteen ramp values at once (v(s) is read when its product is formed)
template <int SL, class V>
UPX_HD void zoom_ramp_mul(const UPX_GLOBAL cf* seeds, int r, int sl, cf* x, V v) {
    // (32-bit lane offsets against the uniform table base: no 64-bit vector address to keep or spill)
    const unsigned ro = (unsigned)(r * (SL + 4));
    const cf b0 = gat(seeds, ro + (unsigned)sl, 0), q1 = gat(seeds, ro, SL), q2 = gat(seeds, ro, SL + 1),
             q4 = gat(seeds, ro, SL + 2), q8 = gat(seeds, ro, SL + 3);
    auto put = [&](int s, cf rs) {
        x[s] = cmul(v(s), rs);
        x[s + 8] = cmul(v(s + 8), cmul(rs, q8));
    };
    put(0, b0);
    put(1, cmul(b0, q1));
    const cf r2 = cmul(b0, q2);
    put(2, r2);
    put(3, cmul(r2, q1));
    const cf r4 = cmul(b0, q4);
    put(4, r4);
    put(5, cmul(r4, q1));
    const cf r6 = cmul(r4, q2);
    put(6, r6);
    put(7, cmul(r6, q1));
}

// `left` samples remain from a (uniform) position to the end of a buffer: the lanes' own offsets o < 2^29 are inside iff
// o < zoom_limit(left) - a 32-bit compare against a uniform, where (long long)o < left keeps a 64-bit copy of o alive
// through the whole transform loop (it was the band-limited synthesis' only spill: 2-5 dwords on the signal-edge body)
UPX_HD unsigned zoom_limit(long long left) {
    return left <= 0 ? 0u : (left > 0x7fffffffLL ? 0x7fffffffu : (unsigned)left);
}

struct ZoomYes { static constexpr bool value = true; };
struct ZoomNo { static constexpr bool value = false; };

// the same from seeds already in registers: sd[0] = b0, sd[1..4] = q, q^2, q^4, q^8 W_N^(-r P)
template <class V>
UPX_HD void zoom_ramp_mul_seeds(const cf* sd, cf* x, V v) {
    const cf b0 = sd[0], q1 = sd[1], q2 = sd[2], q4 = sd[3], q8 = sd[4];
    auto put = [&](int s, cf rs) {
        x[s] = cmul(v(s), rs);
        x[s + 8] = cmul(v(s + 8), cmul(rs, q8));
    };
    put(0, b0);
    put(1, cmul(b0, q1));
    const cf r2 = cmul(b0, q2);
    put(2, r2);
    put(3, cmul(r2, q1));
    const cf r4 = cmul(b0, q4);
    put(4, r4);
    put(5, cmul(r4, q1));
    const cf r6 = cmul(r4, q2);
    put(6, r6);
    put(7, cmul(r6, q1));
}

// EAGER: all LDS reads of a radix-16 pass (inputs and twiddles, 62 registers) issued before the first multiply
// SWAP (wave-local layout on both sides, Sub::SWAP_LAST): the exchange before the last pass stays in registers
// (Stream::row_exchange); the caller ends the transform with S::last_compute<true>
template <class Z, int PI, bool EAGER, bool SWAP = false, class Ex>
UPX_HD void zoom_mid_passes(Ex& ex, cf* lds_all, const cf* tw) {
    using S = Stream<typename Z::Sub>;
    using Thread = ThreadT<16>;
    static_assert(!SWAP || Z::Sub::SWAP_LAST, "row exchange needs sub-FFTs of 2 or 4 rows");
    if constexpr (SWAP && PI == Z::Sub::PS::n - 2) {
        ex.each([lds_all, tw](int tid, Thread& th) {
            S::template read_compute<PI, EAGER>(th, lds_all + (tid / Z::SL) * Z::BUF, tw, tid % Z::SL);
            S::row_exchange(th);
        });
    } else if constexpr (PI < Z::Sub::PS::n - 1) {
        ex.each2(
            [lds_all, tw](int tid, Thread& th) {
                S::template read_compute<PI, EAGER>(th, lds_all + (tid / Z::SL) * Z::BUF, tw, tid % Z::SL);
            },
            [lds_all](int tid, Thread& th) { S::template pass_write<PI>(th, lds_all + (tid / Z::SL) * Z::BUF, tid % Z::SL); });
        zoom_mid_passes<Z, PI + 1, EAGER, SWAP>(ex, lds_all, tw);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Analysis.  A *unit* = (frame, residue group): RG sub-FFTs of one frame.  Per unit:
//   registers of the previous unit's prefetch x window -> pass 0 (coalesced), scatter; the unit's small loads (ramp
//   seeds, gains) and THEN the next unit's input + window are requested | B | passes 1.. (wave-local), ramp -> own
//   LDS cells | B | sum over the group's residues into registers | B |; after a frame's last group: L/R split,
//   gains, mask -> global.
// Vector memory loads return in the order they were issued, so a load that is needed soon must never be queued
// behind one that may miss to HBM: every load of a unit is issued at ONE point, the short ones first, and the long
// ones (the next unit's 16 + 16 slots) are only waited for a whole unit later.
// ---------------------------------------------------------------------------------------------------------------
template <class Z, class Ex>
UPX_HD void zoom_analysis_program(Ex& ex, const ZoomArgs& a, cf* lds_all, int wg_index) {
    using SC = typename Z::Sub;
    using S = Stream<SC>;
    using PS = typename SC::PS;
    using Thread = ThreadT<16>;
    constexpr int P = Z::P, SL = Z::SL, RG = Z::RG, BUF = Z::BUF, LAST = PS::n - 1, BPT = Z::BPT;
    cf* const tw = lds_all + RG * BUF;
    const int D = a.d;
    const int n_groups = D / RG;

    ex.each([&](int tid, Thread&) {
        const UPX_GLOBAL cf* src = opaque(a.tw);
        for (int i = tid; i < Z::TW_CF; i += Z::WG) tw[i] = src[i];
    });
    ex.wg_barrier();

    // Which pairs this workgroup transforms.  Blocks are dealt round-robin over the 8 XCDs (blocks b and b + 8 share
    // one and its 4 MB L2) and frames overlap K-fold: XCD x takes the x-th eighth of the launch's pairs and its
    // workgroups walk it side by side (pair l, l + n_l, l + 2 n_l, ... for its l-th workgroup), so the XCD's
    // workgroups read neighbouring frames at the same time: an input line is fetched from HBM once per XCD and the
    // K-1 other reads are L2 hits.  (A speed choice only: any placement computes the same pairs.)
    const int n_pairs = a.pair_end - a.pair0;
    const int per_xcd = (n_pairs + 7) / 8;
    const int xcd = wg_index % 8, l = wg_index / 8, n_l = a.pairs_per_wg;   // pairs_per_wg: workgroups per XCD here
    int q_stop = a.pair0 + (xcd + 1) * per_xcd;
    if (q_stop > a.pair_end) q_stop = a.pair_end;

    // the pair after q for this workgroup (>= q_stop: none); see ZoomArgs::deal_rows
    const int q_base = a.pair0 + xcd * per_xcd;
    int extra_first = 0, extra_end = 0;   // this workgroup's consecutive pairs behind the rows, relative to q_base
    const int rect = a.deal_rows * n_l;
    if (a.deal_rows > 0) {
#if defined(__HIP_DEVICE_COMPILE__)
        const UPX_GLOBAL int* tab = (const UPX_GLOBAL int*)a.deal_tab;
#else
        const int* tab = a.deal_tab;
#endif
        extra_first = rect + uniform_int(tab[2 * l]);       // (vector loads of a uniform address: back to scalar registers)
        extra_end = extra_first + uniform_int(tab[2 * l + 1]);
    }
    auto next_q = [&](int q) {
        if (a.deal_rows <= 0) return q + n_l;
        const int loc = q - q_base;
        if (loc + n_l < rect) return q + n_l;                                             // the next row
        if (loc < rect) return extra_first < extra_end ? q_base + extra_first : q_stop;   // the last row: on to the extras
        return loc + 1 < extra_end ? q + 1 : q_stop;
    };
    struct Unit {
        int q, half, grp;
    };
    auto frame_of = [](const Unit& u) { return 2 * u.q - 1 + u.half; };
    auto exists = [&](int j) { return j >= a.j_lo && j < a.j_hi; };
    // the unit after u among the frames that exist (q = q_stop: none)
    auto advance = [&](Unit u) {
        for (;;) {
            if (u.q >= q_stop) return u;
            if (exists(frame_of(u)) && u.grp + 1 < n_groups) {
                ++u.grp;
                return u;
            }
            u.grp = -1;   // next frame, group 0 (if the frame exists)
            if (u.half == 0) u.half = 1;
            else { u.half = 0; u.q = next_q(u.q); }
            if (u.q >= q_stop) return u;
            if (exists(frame_of(u))) {
                u.grp = 0;
                return u;
            }
        }
    };
    // input and window of unit u -> th.pre / th.acc_c (coalesced layout: slot s = sample D (sl + SL s) + r of the
    // frame).  One code path: samples past the signal are read from an in-range address and meet a zero window
    // (zero extension of center_extraction.py:437-455); a unit past the end re-reads the current one (never used).
    auto request = [&](int tid, Thread& th, Unit u) {
        const int rho = tid % RG, sl = tid / RG;
        const unsigned o_lane = (unsigned)(D * sl + u.grp * RG + rho);
        const long long stride = (long long)D * SL;                       // samples between slots (uniform)
        const long long base = (long long)frame_of(u) * a.hop;            // first sample of the frame (uniform)
        const UPX_GLOBAL cf* in = opaque(a.in);
        const UPX_GLOBAL float* w_a = opaque(a.w_a);
        if (base + a.n <= a.t_in) {   // (uniform; both sides issue the same 32 loads)
            const unsigned o = pin_lane(o_lane);   // (pinned inside the block that uses it)
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                th.pre[s] = gat_u(in, base + s * stride, o);
                th.acc_c[s] = gat_u(w_a, s * stride, o);
            }
        } else {
            const unsigned o = pin_lane(o_lane);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const bool inside = o < zoom_limit((long long)a.t_in - (base + s * stride));   // (uniform limit)
                // (the uniform part stays uniform: clipped to the signal, which changes nothing for a lane that is inside)
                const long long ub = base + s * stride < (long long)a.t_in ? base + s * stride : (long long)a.t_in - 1;
                th.pre[s] = gat_u(in, ub, inside ? o : 0u);
                const float w = gat_u(w_a, s * stride, o);
                th.acc_c[s] = inside ? w : 0.f;
            }
        }
    };

    Unit cur{q_base + l, 0, -1};
    if (cur.q < q_stop && exists(frame_of(cur))) cur.grp = 0;
    else cur = advance(cur);
    if (Z::PREFETCH_A && cur.q < q_stop) ex.each([&, cur](int tid, Thread& th) { request(tid, th, cur); });

    // frames are visited pair by pair; a pair whose frames do not exist still writes its (zero) centre spectrum
    int turn = 0;
    for (int q = q_base + l; q < q_stop; q = next_q(q), ++turn) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (a.prio_split > 0) {
            const int r = (wg_index / a.prio_split + turn) % a.prio_rounds;
            if (r == 0) __builtin_amdgcn_s_setprio(3);
            else if (r == 1) __builtin_amdgcn_s_setprio(2);
            else if (r == 2) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
#endif
        for (int half = 0; half < 2; ++half) {
            const int j = 2 * q - 1 + half;
            const bool ex_j = exists(j);
            if (ex_j) {
                for (int grp = 0; grp < n_groups; ++grp) {
                    const Unit nxt = advance(Unit{q, half, grp});
                    const bool last_group = grp == n_groups - 1;
                    ex.each([&, q, half, grp, nxt, last_group](int tid, Thread& th) {
                        const int rho = tid % RG, sl = tid / RG;
                        if constexpr (Z::PREFETCH_A) {
                            // window and the first butterflies of pass 0 in one (the prefetch below reuses th.pre / th.acc_c)
#pragma unroll
                            for (int s = 0; s < 16; ++s) th.x[s] = UPX_FMA_BUTTERFLY ? th.pre[s] : scale(th.pre[s], th.acc_c[s]);
                            if constexpr (UPX_FMA_BUTTERFLY != 0) Dft<16>::first_sc(th.x, th.acc_c);
                        }
                        // short loads of this unit first ...
                        if constexpr (Z::PREFETCH_A) {
                            const int g = tid / SL, wsl = tid % SL;          // wave-local layout of the later phases
                            const UPX_GLOBAL cf* seeds = opaque(a.ramp);
                            const unsigned ro = (unsigned)((grp * RG + g) * (SL + 4));
                            th.cs[0] = gat(seeds, ro + (unsigned)wsl, 0);
#pragma unroll
                            for (int i = 0; i < 4; ++i) th.cs[1 + i] = gat(seeds, ro, SL + i);
                        }
                        if (Z::PREFETCH_A && last_group) {
                            const UPX_GLOBAL float* gain = opaque(a.gain);
#pragma unroll
                            for (int i = 0; i < BPT; ++i) {
                                const int k = tid + i * Z::WG;
                                th.g0[i] = k < P / 2 ? gain[k] : 0.f;
                                th.g1[i] = (k < P / 2 && a.n_gain > 1) ? gain[a.gain_stride + k] : 0.f;
                            }
                        }
                        UPX_SCHED_FENCE();
                        if constexpr (Z::PREFETCH_A) {
                            // ... then the long ones: the next unit (the last unit re-requests itself: no branch, so
                            // the wait for the short loads above can leave exactly these in flight)
                            request(tid, th, nxt.q < q_stop ? nxt : Unit{q, half, grp});
                        } else {
                            request(tid, th, Unit{q, half, grp});
#pragma unroll
                            for (int s = 0; s < 16; ++s) th.x[s] = UPX_FMA_BUTTERFLY ? th.pre[s] : scale(th.pre[s], th.acc_c[s]);
                            if constexpr (UPX_FMA_BUTTERFLY != 0) Dft<16>::first_sc(th.x, th.acc_c);
                        }
                        UPX_SCHED_FENCE();
                        static_assert(PS::r[0] == 16, "pass 0 is one radix-16 butterfly per thread");
                        if constexpr (UPX_FMA_BUTTERFLY != 0) Dft<16>::second_inplace(th.x);
                        else S::template pass_compute<0>(th, tw, sl);
                        S::template pass_write<0>(th, lds_all + rho * BUF, sl);
                    });
                    ex.wg_barrier();
                    // wave-local layout: sub-FFT g
                    zoom_mid_passes<Z, 1, true, Z::Sub::SWAP_LAST>(ex, lds_all, tw);
                    ex.each([&, grp](int tid, Thread& th) {
                        const int g = tid / SL, sl = tid % SL;
                        S::template last_compute<Z::Sub::SWAP_LAST, Z::PREFETCH_A>(th, lds_all + g * BUF, tw, sl);
                        // slot s holds F_r[k = sl + SL s]: times the ramp, back into the cells this thread has just read
                        cf zv[16];
                        if constexpr (Z::PREFETCH_A) zoom_ramp_mul_seeds(th.cs, zv, [&](int s) { return th.x[s]; });
                        else zoom_ramp_mul<SL>(opaque(a.ramp), grp * RG + g, sl, zv, [&](int s) { return th.x[s]; });
                        cf* b = lds_all + g * BUF + padp<16>(sl);
#pragma unroll
                        for (int s = 0; s < 16; ++s) b[s * Z::SP] = zv[s];
                    });
                    ex.wg_barrier();
                    // sum over the group's residues; the sum over the groups stays in registers (bin k = tid + i WG
                    // and its mirror P - k: two complex per bin)
                    ex.each([&, grp](int tid, Thread& th) {
#pragma unroll
                        for (int i = 0; i < BPT; ++i) {
                            const int k = tid + i * Z::WG;
                            if (k >= P / 2) break;
                            const cf* pa = lds_all + padp<16>(k);
                            const cf* pb = lds_all + padp<16>(k == 0 ? 0 : P - k);
                            cf za = lds_load(pa), zb = lds_load(pb);
#pragma unroll
                            for (int g = 1; g < RG; ++g) {
                                za = za + lds_load(pa + g * BUF);
                                zb = zb + lds_load(pb + g * BUF);
                            }
                            th.part[2 * i] = grp == 0 ? za : th.part[2 * i] + za;
                            th.part[2 * i + 1] = grp == 0 ? zb : th.part[2 * i + 1] + zb;
                        }
                    });
                    ex.wg_barrier();   // the buffers are free for the next scatter
                }
            }
            // L/R split, gains, mask of the bins this thread has summed (no LDS access: no barrier around it)
            ex.each([&, j, q, half, ex_j](int tid, Thread& th) {
                const UPX_GLOBAL float* gain = opaque(a.gain);
                UPX_GLOBAL cf* y = opaque(a.y) + (size_t)(j - a.f0) * P;
                UPX_GLOBAL cf* yc = opaque(a.yc) + (size_t)(q - (a.f0 + 1) / 2) * P;
#pragma unroll
                for (int i = 0; i < BPT; ++i) {
                    const int k = tid + i * Z::WG;
                    if (k >= P / 2) break;
                    cf c = mk(0.f, 0.f), ls = c, rs = c;
                    if (ex_j) {
                        const cf za = th.part[2 * i], zb = th.part[2 * i + 1];
                        const cf l0 = add_conj(za, zb);      // Z[k] + conj Z[-k]        (x gain/2 = L)
                        const cf r0 = mi_sub_conj(za, zb);   // (Z[k] - conj Z[-k]) / i  (x gain/2 = R)
                        for (int qg = 0; qg < a.n_gain; ++qg) {
                            // slots 0 and 1 were requested with the frame's last unit; more (three or more merged
                            // bands overlapping in one bin) are rare
                            const float g2 = (Z::PREFETCH_A && qg < 2) ? (qg == 0 ? th.g0[i] : th.g1[i])
                                                                       : gain[qg * a.gain_stride + k];
                            if (g2 != 0.f) {
                                cf l = scale(l0, g2), r = scale(r0, g2), cq, lq, rq;
                                mask_bin(l, r, cq, lq, rq);
                                c = c + cq; ls = ls + lq; rs = rs + rq;
                            }
                        }
                        // inverse-by-swap input: Y[k] = Ls + i Rs, Y[-k] = conj(Ls) + i conj(Rs), kept re/im swapped
                        y[k] = swap_add_i(ls, rs);
                        if (k == 0) y[P / 2] = mk(0.f, 0.f);
                        else y[P - k] = swap_conj_add_i(ls, rs);
                    }
                    if (half == 0) {
                        th.cs[5 + i] = c;
                    } else {
                        const cf ca = th.cs[5 + i];
                        yc[k] = swap_add_i(ca, c);
                        if (k == 0) yc[P / 2] = mk(0.f, 0.f);
                        else yc[P - k] = swap_conj_add_i(ca, c);
                    }
                }
            });
        }
    }
    ex.wg_barrier();
}

// ---------------------------------------------------------------------------------------------------------------
// Synthesis: one workgroup = RG residue streams of one stream slot (frames m0 .. m0+F-1), role 0 = Ls/Rs, 1 = C.
// Per transform:  staged spectrum x ramp, pass 0 (wave-local, registers only) | B | next spectrum -> stage, scatter, passes 1..n-2 | B |
//                 last pass (coalesced), window, overlap-add, emit hop.
// ---------------------------------------------------------------------------------------------------------------
template <class Z, int ROLE, class Ex>
UPX_HD void zoom_synthesis_role(Ex& ex, const ZoomArgs& a, cf* lds_all, int stream_index, int grp) {
    constexpr int role = ROLE;   // 0: Ls/Rs streams, 1: centre streams (separate instantiations: separate register sets)
    using SC = typename Z::Sub;
    using S = Stream<SC>;
    using PS = typename SC::PS;
    using Thread = ThreadT<16>;
    constexpr int P = Z::P, SL = Z::SL, RG = Z::RG, BUF = Z::BUF, LAST = PS::n - 1, HS = Z::HS, WG = Z::WG;
    constexpr int NPT = P / WG;   // spectrum values a thread moves from global memory to the staging row
    cf* const tw = lds_all + RG * BUF;
    cf* const stage = tw + Z::TW_CF;   // the spectrum of the transform about to start, shared by the RG residues
    cf* const seeds_lds = stage + P;   // SEEDS_LDS: ramp seeds of this workgroup's residues, [RG][SL + 4]
    // AHEAD_OLD: the old plane values of a transform's hop (HBM misses of bands >= 1) are requested in the LAST phase
    // of the previous transform, before its stores; possible where nothing short has to be loaded in between
    // (SEEDS_LDS), because loads return in order
    constexpr bool AHEAD_OLD = Z::SEEDS_LDS;
    const int D = a.d;
    const int sid = (role ? a.stream0_c : a.stream0) + stream_index;
    const int m0 = uniform_int(zoom_stream_first(a, role, sid));   // (table entries: vector loads of a uniform address)
    const int F = uniform_int(zoom_stream_frames(a, role, sid));
    const int stride = D * SL;   // samples between slots
    // position in the launch's dispatch order (upx_zoom_synthesis_kernel): Ls/Rs workgroups, then the centre ones
    const int lin = role == 0 ? stream_index + a.ns_lr * grp : a.ns_lr * (D / RG) + stream_index + a.ns_c * grp;
    (void)lin;
    const int n_tr = role == 0 ? F : F / 2;   // transforms of this stream: one per frame (Ls/Rs) or per pair (C)

    // Spectrum of transform t (nullptr: all zero).  Ls/Rs: frame m0 + t.  C: pair (m0 + 2t, m0 + 2t + 1), m0 odd.
    auto spec_of = [&](int t) -> const cf* {
        if (t >= n_tr) return nullptr;
        if (role == 0) {
            const int j = m0 + t;
            return j >= a.j_lo && j < a.j_hi ? a.y + (size_t)(j - a.f0) * P : nullptr;
        }
        const int ja = m0 + 2 * t;
        const bool ex2 = (ja >= a.j_lo && ja < a.j_hi) || (ja + 1 >= a.j_lo && ja + 1 < a.j_hi);
        return ex2 ? a.yc + (size_t)((ja + 1) / 2 - (a.f0 + 1) / 2) * P : nullptr;
    };
    // One emitted hop of a plane: slot s of a thread is sample `base + s stride` (uniform) + `o` (the thread's own).
    struct Hop {
        long long base;   // first sample of the hop
        unsigned o;
        bool emit;        // this stream emits the hop
        bool fast;        // ... and it lies inside the planes: no per-sample checks
    };
    auto hop_of = [&](int tid, int j) {
        const int rho = tid % RG, sl = tid / RG;
        Hop h;
        h.emit = j >= a.m_lo && j < a.m_hi && j < m0 + F;
        h.base = (long long)j * a.hop;
        h.o = pin_lane((unsigned)(D * sl + grp * RG + rho));
        h.fast = h.emit && h.base + a.hop <= a.t_out;
        return h;
    };
    // old values of an emitted hop (band sum in list order)
    auto load_old = [&](UPX_GLOBAL float* plane, const Hop& h, float* old) {
#pragma unroll
        for (int s = 0; s < HS; ++s) old[s] = 0.f;
        if (!a.accumulate) return;
        if (h.fast) {
#pragma unroll
            for (int s = 0; s < HS; ++s) old[s] = load_nt(gat_u(plane, h.base + s * (long long)stride, h.o));
        } else if (h.emit) {
#pragma unroll
            for (int s = 0; s < HS; ++s) {
                if (h.o < zoom_limit((long long)a.t_out - (h.base + s * (long long)stride)))
                    old[s] = gat_u(plane, h.base + s * (long long)stride, h.o);
            }
        }
    };
    auto emit = [&](UPX_GLOBAL float* plane, const Hop& h, int s, float v) {
        if (h.fast) {
            // A plain store, not store_nt: with more residues than a workgroup holds (D > RG: bands 0-2 of C3, the default
            // plan's 65 536 bands) a store instruction writes 64-byte segments, half a cache line each, and the streaming hint
            // costs the write-only synthesis 4-7 % (A/B round 4: 0.181-0.190 vs 0.194-0.198 ms; contiguous 256-byte rows -
            // D = RG, band 3 - do not care).  (Until round 4 the hint was silently lost here anyway: the optimiser merged the
            // fast and the checked store paths and dropped the metadata.)
#if defined(UPX_ZOOM_STORE_NT)
            store_nt(gat_u(plane, h.base + s * (long long)stride, h.o), v);
#else
            gat_u(plane, h.base + s * (long long)stride, h.o) = v;
#endif
        } else if (h.emit) {
            if (h.o < zoom_limit((long long)a.t_out - (h.base + s * (long long)stride)))
                gat_u(plane, h.base + s * (long long)stride, h.o) = v;
        }
    };
    // old plane values of the hop(s) transform t emits -> o8[] (Ls/Rs: (old_l, old_r) per slot; C: (old of a, old of b))
    auto fetch_old = [&](int tid, int t, cf* o8) {
        float u[HS], v[HS];
        if (role == 0) {
            const Hop h = hop_of(tid, m0 + t);
            load_old(opaque(a.out_l), h, u);
            load_old(opaque(a.out_r), h, v);
        } else {
            load_old(opaque(a.out_c), hop_of(tid, m0 + 2 * t), u);
            load_old(opaque(a.out_c), hop_of(tid, m0 + 2 * t + 1), v);
        }
#pragma unroll
        for (int s = 0; s < HS; ++s) o8[s] = mk(u[s], v[s]);
    };
    // this thread's part of a spectrum, global memory -> registers (th.part: consumed one transform later)
    auto fetch_spec = [&](int tid, Thread& th, const cf* spec) {
        if (spec) {
            const UPX_GLOBAL cf* sp = opaque(spec);
#pragma unroll
            for (int u = 0; u < NPT; ++u) th.part[u] = gat(sp, (unsigned)tid, u * WG);
        } else {
#pragma unroll
            for (int u = 0; u < NPT; ++u) th.part[u] = mk(0.f, 0.f);
        }
    };
    auto put_stage = [&](int tid, Thread& th) {
#pragma unroll
        for (int u = 0; u < NPT; ++u) stage[tid + u * WG] = th.part[u];
    };

    // prologue: twiddles, spectrum 0 staged, spectrum 1 and the first old values in flight
    ex.each([&](int tid, Thread& th) {
        const UPX_GLOBAL cf* src = opaque(a.tw);
        for (int i = tid; i < Z::TW_CF; i += WG) tw[i] = src[i];
        if constexpr (Z::SEEDS_LDS) {
            const UPX_GLOBAL cf* sd = opaque(a.ramp) + (size_t)grp * RG * (SL + 4);
            for (int i = tid; i < RG * (SL + 4); i += WG) seeds_lds[i] = sd[i];
        }
        fetch_spec(tid, th, spec_of(0));
        put_stage(tid, th);
        fetch_spec(tid, th, spec_of(1));
        if constexpr (AHEAD_OLD) fetch_old(tid, 0, th.pre);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            if (role == 0) th.acc_rl[s] = mk(0.f, 0.f);
            else th.acc_c[s] = 0.f;
        }
    });
    ex.wg_barrier();

    // Per transform:  staged spectrum x ramp, pass 0 (wave-local, registers) | B1 | next spectrum -> stage, the one
    // after it -> registers, scatter, passes 1..n-2 | B2 | last pass (coalesced), window, overlap-add, emit.
    // The old plane values of the hop a transform emits are HBM misses.  Where the LDS footprint leaves registers
    // to spare (3 waves per SIMD: 168 VGPRs; or the centre role) they are requested at the top of the transform,
    // BEHIND the ramp seeds (loads return in order), and waited for at its end; otherwise at the top of the last phase.
    constexpr bool EARLY_OLD = !AHEAD_OLD && (Z::WPE_S <= 3 || ROLE == 1);
    constexpr bool EARLY_WINDOW = Z::WPE_S <= 3;
#if !defined(UPX_ZOOM_SYN_EAGER)
#define UPX_ZOOM_SYN_EAGER 1
#endif
    constexpr bool EAGER_PASSES = UPX_ZOOM_SYN_EAGER != 0 && Z::WPE_S <= 3;
    // INTERIOR streams - every frame exists, every hop is emitted and lies inside the planes: all streams but the
    // first and the last few of a signal - run a loop body whose vector-memory operations are unconditional (same
    // count on every path).  The backend's s_waitcnt bookkeeping is exact only then: with a branch that issues a
    // varying number of loads or stores between a load and its use it falls back to vmcnt(0), i.e. it waits for
    // the youngest prefetch too.
    const bool interior = m0 >= a.j_lo && m0 + F <= a.j_hi && m0 >= a.m_lo && m0 + F <= a.m_hi &&
                          (long long)(m0 + F) * a.hop <= a.t_out;
    auto spec_in = [&](int t) -> const cf* {   // interior: spectrum of transform min(t, last) (re-read past the end: unused)
        const int tt = t < n_tr ? t : n_tr - 1;
        return role == 0 ? a.y + (size_t)(m0 + tt - a.f0) * P : a.yc + (size_t)((m0 + 2 * tt + 1) / 2 - (a.f0 + 1) / 2) * P;
    };
    auto fetch_old_in = [&](int tid, int t, cf* o8) {   // interior + accumulating: 2 HS unconditional loads
        const int tt = t < n_tr ? t : n_tr - 1;
        const unsigned o = pin_lane((unsigned)(D * (tid / RG) + grp * RG + tid % RG));
        UPX_GLOBAL float* p0 = role == 0 ? opaque(a.out_l) : opaque(a.out_c);
        UPX_GLOBAL float* p1 = role == 0 ? opaque(a.out_r) : opaque(a.out_c);
        const long long b0 = (long long)(role == 0 ? m0 + tt : m0 + 2 * tt) * a.hop;
        const long long b1 = role == 0 ? b0 : b0 + a.hop;
#pragma unroll
        for (int s = 0; s < HS; ++s)
            o8[s] = mk(load_nt(gat_u(p0, b0 + s * (long long)stride, o)), load_nt(gat_u(p1, b1 + s * (long long)stride, o)));
    };
    auto run = [&](auto in_tag, auto acc_tag) {
    constexpr bool IN = decltype(in_tag)::value;      // interior stream: unconditional loads / stores
    constexpr bool ACC = decltype(acc_tag)::value;    // (interior only) the band accumulates onto the planes
    for (int t = 0; t < n_tr; ++t) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (a.prio_split_s > 0 && (t & 1) == 0) {
            const int r = (lin / a.prio_split_s + (t >> 1)) % a.prio_rounds_s;
            if (r == 0) __builtin_amdgcn_s_setprio(3);
            else if (r == 1) __builtin_amdgcn_s_setprio(2);
            else if (r == 2) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
#endif
        const bool nonzero = IN || spec_of(t) != nullptr;
        ex.each([&, t, nonzero](int tid, Thread& th) {
            if (nonzero) {
                const int g = tid / SL, sl = tid % SL;
                const cf* sp = stage + sl;
                if constexpr (Z::SEEDS_LDS) {
                    const cf* row = seeds_lds + g * (SL + 4);
                    const cf sd[5] = {lds_load(row + sl), lds_load(row + SL), lds_load(row + SL + 1), lds_load(row + SL + 2),
                                      lds_load(row + SL + 3)};
                    zoom_ramp_mul_seeds(sd, th.x, [&](int s) { return lds_load(sp + s * SL); });
                } else {
                    zoom_ramp_mul<SL>(opaque(a.ramp), grp * RG + g, sl, th.x, [&](int s) { return lds_load(sp + s * SL); });
                }
            }
            if constexpr (EARLY_OLD) {
                UPX_SCHED_FENCE();
                if constexpr (IN && ACC) fetch_old_in(tid, t, th.pre);
                else if constexpr (!IN) fetch_old(tid, t, th.pre);
            }
            if (nonzero) S::template pass_compute<0>(th, tw, tid % SL);
        });
        ex.wg_barrier();   // B1: the stage row and (previous transform) the sub-FFT buffers have been read by every wave
        ex.each([&, t, nonzero](int tid, Thread& th) {
            put_stage(tid, th);                       // spectrum t+1 (in flight since the previous transform)
            if constexpr (EARLY_WINDOW) {
                // the window of the last phase BEFORE the next long request (loads return in order: queued behind
                // the spectrum prefetch the window would arrive with it); coalesced layout of that phase
                const unsigned o = pin_lane((unsigned)(D * (tid / RG) + grp * RG + tid % RG));
                const UPX_GLOBAL float* w_s = opaque(a.w_s);
#pragma unroll
                for (int s = 0; s < 16; ++s) th.g0w[s] = gat_u(w_s, s * (long long)stride, o);
                UPX_SCHED_FENCE();
                fetch_spec(tid, th, IN ? spec_in(t + 2) : spec_of(t + 2));
            }
            if (nonzero) S::template pass_write<0>(th, lds_all + (tid / SL) * BUF, tid % SL);
        });
        if (nonzero) {
            // (eager passes - all LDS reads of a radix-16 pass before its first multiply - only where the LDS footprint
            // leaves 168 VGPRs: next to the overlap-add state their 30 twiddle registers do not fit 128)
            zoom_mid_passes<Z, 1, EAGER_PASSES>(ex, lds_all, tw);
        }
        ex.wg_barrier();   // B2: sub-FFT buffers and the stage row complete
        // last pass in the coalesced layout; slot s = sample D (sl + SL s) + r of the frame
        ex.each([&, t, nonzero](int tid, Thread& th) {
            const int rho = tid % RG, sl = tid / RG;
            const unsigned o = pin_lane((unsigned)(D * sl + grp * RG + rho));
            // loads first, stores last: the old plane values of the hop(s) about to be emitted (band sum in list
            // order; HBM misses that the last pass and the overlap-add below cover in part) and the window
            cf old[HS];
            if constexpr (EARLY_OLD || AHEAD_OLD) {
#pragma unroll
                for (int s = 0; s < HS; ++s) old[s] = (!IN || ACC) ? th.pre[s] : mk(0.f, 0.f);
            } else {
                if constexpr (IN && ACC) {
                    fetch_old_in(tid, t, old);
                } else if constexpr (IN) {
#pragma unroll
                    for (int s = 0; s < HS; ++s) old[s] = mk(0.f, 0.f);
                } else {
                    fetch_old(tid, t, old);
                }
            }
            if constexpr (AHEAD_OLD) {
                // fold the old values into the slots about to be emitted, then request the next transform's
                // (loads first, stores last; same registers)
#pragma unroll
                for (int s = 0; s < HS; ++s) {
                    if (role == 0) {
                        th.acc_rl[s] = th.acc_rl[s] + mk(old[s].y, old[s].x);   // acc_rl = (Rs, Ls), old = (old_l, old_r)
                    } else {
                        th.acc_c[s] += old[s].x;
                        th.acc_c[s + HS] += old[s].y;
                    }
                    old[s] = mk(0.f, 0.f);
                }
                if constexpr (IN && ACC) fetch_old_in(tid, t + 1, th.pre);
                else if constexpr (!IN) fetch_old(tid, t + 1, th.pre);
            }
            UPX_SCHED_FENCE();   // issued before anything below
            if (nonzero) {
                S::template read_compute<LAST, EAGER_PASSES>(th, lds_all + rho * BUF, tw, sl);
            } else {
#pragma unroll
                for (int s = 0; s < 16; ++s) th.x[s] = mk(0.f, 0.f);
            }
            UPX_SCHED_FENCE();
            float w[16];
            if constexpr (EARLY_WINDOW) {
#pragma unroll
                for (int s = 0; s < 16; ++s) w[s] = th.g0w[s];
            } else {
                // (the window only now: sixteen more live registers during the earlier phases do not fit next to
                // the overlap-add state at 128 VGPRs; and the next long request only behind it)
                const UPX_GLOBAL float* w_s = opaque(a.w_s);
#pragma unroll
                for (int s = 0; s < 16; ++s) w[s] = gat_u(w_s, s * (long long)stride, o);
                UPX_SCHED_FENCE();
                fetch_spec(tid, th, IN ? spec_in(t + 2) : spec_of(t + 2));
            }
            UPX_SCHED_FENCE();
            if (role == 0) {
                Hop h = hop_of(tid, m0 + t);
                if constexpr (IN) h.fast = true;
                UPX_GLOBAL float* out_l = opaque(a.out_l);
                UPX_GLOBAL float* out_r = opaque(a.out_r);
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    th.acc_rl[s] = th.acc_rl[s] + scale(th.x[s], w[s]);   // swapped output: Ls = x.y, Rs = x.x
#pragma unroll
                for (int s = 0; s < HS; ++s) {
                    emit(out_l, h, s, old[s].x + th.acc_rl[s].y);
                    emit(out_r, h, s, old[s].y + th.acc_rl[s].x);
                }
#pragma unroll
                for (int s = 0; s < 16; ++s) th.acc_rl[s] = s + HS < 16 ? th.acc_rl[s + HS] : mk(0.f, 0.f);
            } else {
                UPX_GLOBAL float* out_c = opaque(a.out_c);
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    Hop h = hop_of(tid, m0 + 2 * t + half);
                    if constexpr (IN) h.fast = true;
#pragma unroll
                    for (int s = 0; s < 16; ++s)   // swapped output: c_a = x.y, c_b = x.x
                        th.acc_c[s] += (half == 0 ? th.x[s].y : th.x[s].x) * w[s];
#pragma unroll
                    for (int s = 0; s < HS; ++s) emit(out_c, h, s, (half == 0 ? old[s].x : old[s].y) + th.acc_c[s]);
#pragma unroll
                    for (int s = 0; s < 16; ++s) th.acc_c[s] = s + HS < 16 ? th.acc_c[s + HS] : 0.f;
                }
            }
        });
    }
    };   // run
    if (interior && a.accumulate) run(ZoomYes{}, ZoomYes{});
    else if (interior) run(ZoomYes{}, ZoomNo{});
    else run(ZoomNo{}, ZoomNo{});
    // what is left in the accumulators belongs to the K-1 blocks after this stream
    ex.each([&](int tid, Thread& th) {
        const int rho = tid % RG, sl = tid / RG;
        const size_t tail = (size_t)(Z::K - 1) * a.hop;
        UPX_GLOBAL float* seam = (role ? opaque(a.seam_c) + (size_t)sid * tail : opaque(a.seam) + (size_t)sid * 2 * tail) +
                                 (D * sl + grp * RG + rho);
#pragma unroll
        for (int s = 0; s < 16 - HS; ++s) {
            if (role == 0) {
                seam[(size_t)s * stride] = th.acc_rl[s].y;
                seam[tail + (size_t)s * stride] = th.acc_rl[s].x;
            } else {
                seam[(size_t)s * stride] = th.acc_c[s];
            }
        }
    });
    ex.wg_barrier();
}

template <class Z, class Ex>
UPX_HD void zoom_synthesis_program(Ex& ex, const ZoomArgs& a, cf* lds_all, int stream_index, int grp, int role) {
    if (role == 0) zoom_synthesis_role<Z, 0>(ex, a, lds_all, stream_index, grp);
    else zoom_synthesis_role<Z, 1>(ex, a, lds_all, stream_index, grp);
}

// The tails of the band-limited streams onto the first blocks of their successors (stream_seam_add of upx_core.h for
// the two stream lists): gid < n_lr tail: Ls / Rs of stream gid / tail, above that the centre streams.
UPX_HD void zoom_seam_add(const ZoomArgs& a, int n_lr, int n_c, int tail, long long gid) {
    const int role = gid >= (long long)n_lr * tail ? 1 : 0;
    if (role) gid -= (long long)n_lr * tail;
    const int sid = (int)(gid / tail), i = (int)(gid % tail);
    if (sid >= (role ? n_c : n_lr) - 1) return;             // the last stream's tail lies beyond the emitted range
    const long long m = zoom_stream_first(a, role, sid + 1);
    const long long n = m * a.hop + i;
    if (m >= a.m_hi || n < 0 || n >= a.t_out) return;
    if (m + i / a.hop >= a.m_hi) return;                    // blocks past the emitted range stay untouched
    if (role) {
        a.out_c[n] += a.seam_c[(size_t)sid * tail + i];
    } else {
        const float* row = a.seam + (size_t)sid * 2 * tail;
        a.out_l[n] += row[i];
        a.out_r[n] += row[tail + i];
    }
}

// Host-side helper: the ramp seeds of zoom_ramp, [D][P/16 + 4] (double precision -> float).
template <class TrigFn>
inline void fill_zoom_ramp(cf* seeds, int n, int p, TrigFn trig) {
    const int d = n / p, sl_n = p / 16;
    auto w = [&](long long num) {   // W_N^num
        num %= n;
        if (num < 0) num += n;
        double c, s;
        trig((double)num / (double)n, c, s);
        return mk((float)c, (float)-s);
    };
    for (int r = 0; r < d; ++r) {
        cf* row = seeds + (size_t)r * (sl_n + 4);
        for (int sl = 0; sl < sl_n; ++sl) row[sl] = w((long long)r * sl);
        row[sl_n] = w((long long)r * sl_n);
        row[sl_n + 1] = w((long long)r * sl_n * 2);
        row[sl_n + 2] = w((long long)r * sl_n * 4);
        row[sl_n + 3] = w((long long)r * sl_n * 8 - (long long)r * p);
    }
}
inline size_t zoom_ramp_count(int n, int p) { return (size_t)(n / p) * (p / 16 + 4); }

}   // namespace upx
