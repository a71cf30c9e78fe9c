// upx_big.h - STFT sizes 16 384 .. 65 536 (the reference's default plan,
// center_extraction.py:173 max_block_size = 2**16, chain_bands :555).
//
// A frame no longer fits one workgroup's LDS (65 536 complex = 512 KB), so the
// per-frame transform is a four-step FFT through HBM/L2 scratch:
//     N = 16 * N2,  n = n1 N2 + n2,  k = k1 + 16 k2
//     step 1  (global)  A[k1][n2] = W_N^(k1 n2) * sum_n1 x[n1 N2 + n2] W_16^(n1 k1)
//     step 2  (LDS)     X[k1 + 16 k2] = FFT_N2 over n2 of A[k1][.]     (row k1, reuses Stream<>)
// so bin k lives at scratch offset (k & 15) N2 + (k >> 4).  The inverse transforms run the
// transposed (decimation-in-time) order on that same layout,
//     rows (FFT_N2 over k2 of row k1)  ->  x[n1 N2 + n2] = sum_k1 W_16^(k1 n1) W_N^(k1 n2) row_k1[n2]
// and therefore end in NATURAL time order: every kernel reads and writes coalesced.
// Frames are processed in chunks of CH frames:
//     step1(audio*w_A) -> [rows -> mask -> rows of Ls+iRs and Ca+iCb] -> step2 -> overlap-add
// (the bracket is one kernel, big_mid_program, when N1 = 16; three kernels when the frame is a single row)
// with the same conventions as the fused kernel (upx_core.h): frame pairs
// (odd j, j+1) share one centre transform, inverse by re/im swap, contributions
// added in increasing j in float32, bands summed in list order.
// The same unfused pipeline (with N1 = 1, i.e. the whole frame transform inside one LDS row) also
// serves every STFT size when the hop is NOT N/2, N/4 or N/8 (e.g. overlap 0.6 -> hop = int(0.4 N),
// center_extraction.py:252): hop and the number of frames covering a sample are run-time values here.
#pragma once
#include "upx_core.h"

namespace upx {

struct BigArgs {
    const cf* in;          // interleaved stereo, local sample 0
    float* out_c;
    float* out_l;
    float* out_r;
    const float* w_a;      // analysis window [N]
    const float* w_s;      // synthesis window / N [N]
    const float* gain;     // 0.5 * band-limit gain, [n_gain][gain_stride] (merged bands, see BandArgs)
    int n_gain, gain_stride;
    const cf* tw_n;        // W_N^(k1 n2), [16][N2]
    const cf* tw_rows;     // compact twiddle table of the N2-point row transform
    cf* z;                 // scratch [CH][N]: forward spectra (bin k at (k&15) N2 + (k>>4))
    cf* y;                 // scratch [CH][N]: Ls + i Rs spectra (same layout as z), then their time signals (natural n)
    cf* yc;                // scratch [CH/2][N]: Ca + i Cb spectra, then time signals
    int t_in, t_out;
    int hop;               // frame advance in samples (any value in [1, N])
    int kf;                // frames that cover one sample: ceil(N / hop)
    int j_lo, j_hi;        // frames that exist
    int j0;                // first frame of the chunk (odd): chunk frames j0 .. j0+ch-1
    int ch;                // frames in the chunk (even)
    int m0, m1;            // hop-blocks this chunk emits: [m0, m1)
    int accumulate;
    int tail;              // 1: big_tail (step 2 + overlap-add in one pass) where it applies; 0: big_step2_inv + big_ola
};

template <int LOG2N>
struct BigCfg {
    static constexpr int N = 1 << LOG2N;
    static constexpr int N1 = LOG2N >= 14 ? 16 : 1;      // radix of the global step (1: the frame fits one LDS row)
    static constexpr int N2 = N / N1;
    static constexpr int LOG2N2 = LOG2N >= 14 ? LOG2N - 4 : LOG2N;
    using Row = Cfg<LOG2N2, 4, (LOG2N2 >= 8 ? 16 : 8)>;   // the N2-point row transform (its K is unused); 8 points per lane below 256
    // bin k -> scratch offset after the forward transform
    UPX_HD static int scr(int i) { return N1 == 1 ? i : (i & 15) * N2 + (i >> 4); }
};

// ---- step 1 on audio: window, radix-16 over n1, twiddle; one thread per (frame, n2) ----
template <class B>
UPX_HD void big_step1_audio(const BigArgs& a, long long gid) {
    constexpr int N2 = B::N2, N = B::N;
    const int jj = (int)(gid / N2), n2 = (int)(gid % N2);
    if (jj >= a.ch) return;
    const int j = a.j0 + jj;
    const bool exists = j >= a.j_lo && j < a.j_hi;
    cf v[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
        const int o = n1 * N2 + n2;
        const long long n = (long long)j * a.hop + o;
        v[n1] = mk(0.f, 0.f);
        if (exists && n >= 0 && n < a.t_in) {
            const cf s = a.in[n];
            const float w = a.w_a[o];
            v[n1] = mk(s.x * w, s.y * w);
        }
    }
    Dft<16>::run(v);
    cf* dst = a.z + (size_t)jj * N;
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) dst[k1 * N2 + n2] = k1 == 0 ? v[0] : cmul(v[k1], a.tw_n[k1 * N2 + n2]);
}

// ---- last step of the inverse transforms: twiddle, radix-16 over k1; natural order out, in place ----
template <class B>
UPX_HD void big_step2_inv(cf* buf, const cf* tw_n, int frames, long long gid) {
    constexpr int N2 = B::N2, N = B::N;
    const int jj = (int)(gid / N2), n2 = (int)(gid % N2);
    if (jj >= frames) return;
    cf* p = buf + (size_t)jj * N;
    cf v[16];
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) v[k1] = k1 == 0 ? p[n2] : cmul(p[k1 * N2 + n2], tw_n[k1 * N2 + n2]);
    Dft<16>::run(v);
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) p[n1 * N2 + n2] = v[n1];
}

// ---- step 2: N2-point transforms of the rows, in LDS; one workgroup per row ----------------
// rows are contiguous [N2] complex; row r of the buffer starts at r * N2.
// A workgroup carries G = WG/LANES rows (more than one only when a row needs less than a wave).
template <class B, class Ex>
UPX_HD void big_rows_program(Ex& ex, cf* buf, const cf* tw_global, cf* lds_all, int wg_index, int n_rows) {
    using C = typename B::Row;
    using S = Stream<C>;
    using Thread = ThreadT<C::P>;
    constexpr int LANES = C::LANES, P = C::P;
    cf* const tw = lds_all + C::G * C::PITCH;
    ex.each([&](int tid, Thread& th) {
        for (int i = tid; i < C::TW_CF; i += C::WG) tw[i] = tw_global[i];
        const int row = wg_index * C::G + tid / LANES, lane = tid % LANES;
        const cf* data = buf + (size_t)row * C::N;
#pragma unroll
        for (int s = 0; s < P; ++s) th.x[s] = row < n_rows ? data[lane + s * LANES] : mk(0.f, 0.f);
        S::template pass_compute<0>(th, tw, lane);
    });
    ex.each([&](int tid, Thread& th) { S::template pass_write<0>(th, lds_all + (tid / LANES) * C::PITCH, tid % LANES); });
    S::template mid_passes<1>(ex, lds_all, tw);
    ex.each([&](int tid, Thread& th) {
        const int row = wg_index * C::G + tid / LANES, lane = tid % LANES;
        cf* data = buf + (size_t)row * C::N;
        S::read_all(th, lds_all + (tid / LANES) * C::PITCH, lane);
        S::template pass_compute<C::PS::n - 1>(th, tw, lane);
        if (row < n_rows) {
#pragma unroll
            for (int s = 0; s < P; ++s) data[lane + s * LANES] = th.x[s];
        }
    });
}

// ---- N1 == 1: window + whole-frame transform in one LDS row; G frames per workgroup ----------------
template <class B, class Ex>
UPX_HD void big_frame_program(Ex& ex, const BigArgs& a, cf* lds_all, int wg_index) {
    using C = typename B::Row;
    using S = Stream<C>;
    using Thread = ThreadT<C::P>;
    constexpr int LANES = C::LANES, P = C::P;
    cf* const tw = lds_all + C::G * C::PITCH;
    ex.each([&](int tid, Thread& th) {
        for (int i = tid; i < C::TW_CF; i += C::WG) tw[i] = a.tw_rows[i];
        const int jj = wg_index * C::G + tid / LANES, lane = tid % LANES;
        const int j = a.j0 + jj;
        const bool exists = jj < a.ch && j >= a.j_lo && j < a.j_hi;
#pragma unroll
        for (int s = 0; s < P; ++s) {
            const int o = lane + s * LANES;
            const long long n = (long long)j * a.hop + o;
            th.x[s] = mk(0.f, 0.f);
            if (exists && n >= 0 && n < a.t_in) {
                const cf v = a.in[n];
                const float w = a.w_a[o];
                th.x[s] = mk(v.x * w, v.y * w);
            }
        }
        S::template pass_compute<0>(th, tw, lane);
    });
    ex.each([&](int tid, Thread& th) { S::template pass_write<0>(th, lds_all + (tid / LANES) * C::PITCH, tid % LANES); });
    S::template mid_passes<1>(ex, lds_all, tw);
    ex.each([&](int tid, Thread& th) {
        const int jj = wg_index * C::G + tid / LANES, lane = tid % LANES;
        cf* data = a.z + (size_t)jj * C::N;
        S::read_all(th, lds_all + (tid / LANES) * C::PITCH, lane);
        S::template pass_compute<C::PS::n - 1>(th, tw, lane);
        if (jj < a.ch) {
#pragma unroll
            for (int s = 0; s < P; ++s) data[lane + s * LANES] = th.x[s];
        }
    });
}

// ---- mask: one thread per (frame pair, k1, k2 <= N2/2), i.e. in the storage order of z ----------
// bin k = k1 + 16 k2 sits at k1 N2 + k2; its partner N-k at ((16-k1)&15) N2 + (N2 - k2 - (k1 != 0)) & (N2-1):
// consecutive threads read and write consecutive addresses (the partner side in reverse).
template <class B>
UPX_HD long long big_mask_threads(int pairs) {
    return B::N1 == 1 ? (long long)pairs * (B::N / 2 + 1) : (long long)pairs * 16 * (B::N2 / 2 + 1);
}
template <class B>
UPX_HD void big_mask(const BigArgs& a, long long gid) {
    constexpr int N = B::N, N2 = B::N2;
    int pp, k;
    if (B::N1 == 1) {                            // natural layout: one thread per bin
        pp = (int)(gid / (N / 2 + 1));
        k = (int)(gid % (N / 2 + 1));
    } else {
        constexpr int HALF = N2 / 2 + 1;
        const int k2 = (int)(gid % HALF);
        const int k1 = (int)((gid / HALF) % 16);
        pp = (int)(gid / (16 * HALF));
        k = k1 + 16 * k2;
    }
    if (pp >= a.ch / 2) return;
    if (k > N / 2) return;                       // the upper half is written by the partners
    const int km = (N - k) & (N - 1);           // partner bin; k = 0 and k = N/2 pair with themselves
    const bool self = km == k;
    const int ok = B::scr(k), om = B::scr(km);
    cf c2[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const size_t f = (size_t)(2 * pp + half) * N;
        const cf za = a.z[f + ok];
        const cf zb = self ? za : a.z[f + om];
        const cf l0 = mk(za.x + zb.x, za.y - zb.y), r0 = mk(za.y + zb.y, zb.x - za.x);
        cf c = mk(0.f, 0.f), ls = c, rs = c;
        for (int q = 0; q < a.n_gain; ++q) {
            const float g2 = a.gain[q * a.gain_stride + k];
            if (g2 != 0.f) {
                cf l = mk(g2 * l0.x, g2 * l0.y), r = mk(g2 * r0.x, g2 * r0.y), cq, lq, rq;
                mask_bin(l, r, cq, lq, rq);
                c = c + cq; ls = ls + lq; rs = rs + rq;
            }
        }
        const cf yk = mk(ls.x - rs.y, ls.y + rs.x);
        const cf ym = mk(ls.x + rs.y, rs.x - ls.y);
        a.y[f + ok] = cswap(yk);
        if (!self) a.y[f + om] = cswap(ym);
        c2[half] = c;
    }
    const cf ca = c2[0], cb = c2[1];
    const cf ck = mk(ca.x - cb.y, ca.y + cb.x);
    const cf cm = mk(ca.x + cb.y, cb.x - ca.y);
    const size_t fc = (size_t)pp * N;
    a.yc[fc + ok] = cswap(ck);
    if (!self) a.yc[fc + om] = cswap(cm);
}

// ---- N1 == 16: rows -> mask -> rows in ONE kernel --------------------------------------------------
// A workgroup takes one frame pair and one mirror pair of rows, k1 and 16 - k1 (0 and 8 for unit 0): two row
// streams of N2/16 lanes each.  Bin k = k1 + 16 k2 and its mirror N - k = (16 - k1) + 16 (N2 - 1 - k2) (k1 = 0:
// (0, N2 - k2)) then sit in the two LDS row buffers of the same workgroup, so the forward row transforms, the L/R
// split + mask and the inverse row transforms of Ls + i Rs (both frames) and Ca + i Cb (the pair) run back to back
// without the spectra leaving the CU: z is read once, y and yc are written once (2.5 of the 7.5 passes over the
// scratch that separate rows / mask / rows kernels make).  The layout inside a row stream is the one of the wide
// streams of upx_core.h (own bins in slots s < 8, mirrors in the partner's upper slots, rewritten in place).
// The per-bin gain rows are read in the order (row r = 2 unit + stream, slot s, lane): big_gain_bin().
template <class B>
inline int big_gain_bin(int i) {
    constexpr int L = B::Row::LANES;
    if (B::N1 == 1 || i >= B::N / 2) return i;
    const int sl = i % L, s = (i / L) % 8, r = i / (8 * L);
    return wide_k1_of_sub(r) + 16 * (sl + L * s);
}

// ROWS = 2: the workgroup described above (rows come from / go to the scratch).
// ROWS = 16 (N = 16 384 only, where a frame's sixteen 1024-point rows fit one workgroup's LDS): the workgroup holds
// the WHOLE frame, so step 1 (window, radix-16 over n1, twiddle, transposed write into the row buffers) and step 2
// (column read, twiddle, radix-16 over k1) run in registers at its two ends, exactly like the cross-wave phases of a
// wide stream: audio in, time-domain y / yc out, no z at all (3 passes over the scratch instead of 8).
template <class B, int ROWS, class Ex>
UPX_HD void big_mid_program(Ex& ex, const BigArgs& a, cf* lds_all, int wg_index) {
    using C = typename B::Row;
    using S = Stream<C>;
    using PS = typename C::PS;
    using Thread = ThreadT<C::P>;
    constexpr int L = C::LANES, P = C::P, H = P / 2, SP = C::SPITCH, BUF = C::PITCH, N = B::N, N2 = B::N2;
    constexpr int LAST = PS::n - 1;
    constexpr int THREADS = ROWS * L;
    constexpr bool WHOLE = ROWS == 16;
    static_assert(P == 16 && B::N1 == 16, "fused rows/mask/rows needs 16 points per lane and the 16 x N2 layout");
    static_assert(ROWS == 2 || (ROWS == 16 && THREADS == N2), "two rows, or the whole frame with one lane per n2");
    cf* const tw = lds_all + ROWS * BUF;
    const int pp = WHOLE ? wg_index : wg_index / 8;   // frame pair of the chunk
    const int unit = WHOLE ? 0 : wg_index % 8;        // ROWS == 2: which mirror pair of rows
    ex.each([&](int tid, Thread&) {
        for (int i = tid; i < C::TW_CF; i += THREADS) tw[i] = a.tw_rows[i];
    });
    // per thread: row stream g of the workgroup = row index r of the frame (in mirror-pair order), lane sl, k1
    auto r_of = [&](int tid) { return 2 * unit + tid / L; };
    auto row_of = [&](int tid) { return wide_k1_of_sub(r_of(tid)); };
    auto partner = [&](int tid) {   // the partner cells, see mirror_of in upx_core.h
        const int g = tid / L, sl = tid % L;
        const int gp = (r_of(tid) >> 1) ? (g ^ 1) : g;
        return lds_all + gp * BUF + padp<P>((H + 1) * L - (row_of(tid) ? 1 : 0) - sl);
    };
    auto gains = [&](int tid, Thread& th) {
        // the gain rows of the own bins: issued early, their latency hides behind the row transform
        const int lane = r_of(tid) * 8 * L + tid % L;
#pragma unroll
        for (int s = 0; s < H; ++s) {
            th.g0[s] = a.gain[lane + s * L];
            th.g1[s] = a.n_gain > 1 ? a.gain[a.gain_stride + lane + s * L] : 0.f;
        }
        th.gn[0] = a.gain[N / 2];
        th.gn[1] = a.n_gain > 1 ? a.gain[a.gain_stride + N / 2] : 0.f;
    };
    auto scatter0 = [&](int tid, Thread& th) { S::template pass_write<0>(th, lds_all + (tid / L) * BUF, tid % L); };
    auto mids = [&]() { S::template mid_passes<1>(ex, lds_all, tw); };
    // (the whole-frame variant runs 16 waves per CU = 128 VGPRs: no eager LDS reads, gains and partners read at use)
    auto last_pass = [&](int tid, Thread& th) { S::template read_compute<LAST, !WHOLE>(th, lds_all + (tid / L) * BUF, tw, tid % L); };
    // forward row transforms up to the scatter of pass 0
    auto forward_in = [&](int frame_in_chunk) {
        if constexpr (WHOLE) {
            ex.each([&](int tid, Thread& th) {   // step 1 in registers (tid = n2), transposed into the row buffers
                // (addresses are rebuilt from laundered SGPR bases per phase: hoisted out of the frame loop they
                //  would cost more registers than this 128-VGPR kernel has)
                const int j = a.j0 + frame_in_chunk;
                const bool exists = j >= a.j_lo && j < a.j_hi;
                const long long n0 = (long long)j * a.hop;            // frame start (may be negative: halo frames)
                const UPX_GLOBAL cf* in = opaque(a.in);
                const UPX_GLOBAL float* w_a = opaque(a.w_a);
                const UPX_GLOBAL cf* tw_n = opaque(a.tw_n);
#pragma unroll
                for (int n1 = 0; n1 < 16; ++n1) {
                    const long long n = n0 + n1 * N2 + tid;
                    th.x[n1] = mk(0.f, 0.f);
                    if (exists && n >= 0 && n < a.t_in) th.x[n1] = scale(in[n], gat(w_a, (unsigned)tid, n1 * N2));
                }
                Dft<16>::run(th.x);
                cf* b = lds_all + padp<P>(tid);
#pragma unroll
                for (int k1 = 0; k1 < 16; ++k1)
                    b[wide_sub_of_k1(k1) * BUF] = k1 == 0 ? th.x[0] : cmul(th.x[k1], gat(tw_n, (unsigned)tid, k1 * N2));
            });
            ex.each2(
                [&](int tid, Thread& th) { S::template read_compute<0, false>(th, lds_all + (tid / L) * BUF, tw, tid % L); },
                scatter0);
        } else {
            const cf* zf = a.z + (size_t)frame_in_chunk * N;
            ex.each2(
                [&](int tid, Thread& th) {
                    gains(tid, th);
                    const cf* data = zf + (size_t)row_of(tid) * N2 + tid % L;
#pragma unroll
                    for (int s = 0; s < P; ++s) th.x[s] = data[s * L];
                    S::template pass_compute<0>(th, tw, tid % L);
                },
                scatter0);
        }
    };
    // inverse row transform of what the mask / stage step left (own slots in registers, upper slots in LDS), then
    // ROWS == 2: the rows go to the scratch (step 2 is a separate kernel); WHOLE: step 2 here, natural order out
    auto inverse_to = [&](cf* frame) {
        ex.each2(
            [&](int tid, Thread& th) {
                const cf* b = lds_all + (tid / L) * BUF + padp<P>(tid % L);
#pragma unroll
                for (int s = H; s < P; ++s) th.x[s] = lds_load(b + s * SP);
                S::template pass_compute<0>(th, tw, tid % L);
            },
            scatter0);
        mids();
        if constexpr (WHOLE) {
            ex.each2(last_pass, [&](int tid, Thread& th) {
                cf* b = lds_all + (tid / L) * BUF + padp<P>(tid % L);
#pragma unroll
                for (int s = 0; s < P; ++s) b[s * SP] = th.x[s];
            });
            ex.each([&](int tid, Thread& th) {   // tid = n2: column of the sixteen rows, twiddle, radix-16 over k1
                const cf* b = lds_all + padp<P>(tid);
                const UPX_GLOBAL cf* tw_n = opaque(a.tw_n);
                UPX_GLOBAL cf* out = opaque(frame);
#pragma unroll
                for (int k1 = 0; k1 < 16; ++k1) {
                    const cf v = lds_load(b + wide_sub_of_k1(k1) * BUF);
                    th.x[k1] = k1 == 0 ? v : cmul(v, gat(tw_n, (unsigned)tid, k1 * N2));
                }
                Dft<16>::run(th.x);
#pragma unroll
                for (int n1 = 0; n1 < 16; ++n1) gat(out, (unsigned)tid, n1 * N2) = th.x[n1];
            });
        } else {
            ex.each([&](int tid, Thread& th) {
                last_pass(tid, th);
                cf* data = frame + (size_t)row_of(tid) * N2 + tid % L;
#pragma unroll
                for (int s = 0; s < P; ++s) data[s * L] = th.x[s];
            });
        }
    };
    const bool pair_ok = pp < a.ch / 2;   // (grid is exact; kept for safety with padded grids)
    for (int half = 0; half < 2 && pair_ok; ++half) {
        forward_in(2 * pp + half);
        mids();
        ex.each2(last_pass, [&](int tid, Thread& th) {
            cf* b = lds_all + (tid / L) * BUF + padp<P>(tid % L);
#pragma unroll
            for (int s = H; s < P; ++s) b[s * SP] = th.x[s];   // park the upper slots for the partners
        });
        ex.each([&](int tid, Thread& th) {
            const bool first = r_of(tid) == 0 && tid % L == 0;   // holds DC (slot 0) and Nyquist (slot H) of the frame
            cf* const part = partner(tid);
            cf* const nyq = lds_all + H * SP;
            const int lane = r_of(tid) * 8 * L + tid % L;
            cf nyq_y = mk(0.f, 0.f);
            float nyq_c = 0.f;
            if (first) {
                const cf z = th.x[H];
                cf cn = mk(0.f, 0.f), lsn = cn, rsn = cn;
                for (int q = 0; q < a.n_gain; ++q) {
                    const float g2 = (q < 2 && !WHOLE) ? th.gn[q] : a.gain[q * a.gain_stride + N / 2];
                    if (g2 != 0.f) {
                        cf l = mk(g2 * (z.x + z.x), 0.f), r = mk(g2 * (z.y + z.y), 0.f), c, ls, rs;
                        mask_bin(l, r, c, ls, rs);
                        cn = cn + c; lsn = lsn + ls; rsn = rsn + rs;
                    }
                }
                nyq_y = mk(lsn.x, rsn.x);
                nyq_c = cn.x;
            }
            // WHOLE: the centre spectrum of the pair waits in the (still unused) yc frame of the pair instead of in
            // registers, lane-private cells [slot][tid]: 32 VGPRs less across the inverse passes
            UPX_GLOBAL cf* const cstage = opaque(a.yc + (size_t)pp * N);
            cf ca_in[H];
            if constexpr (WHOLE) {
                if (half != 0) {
#pragma unroll
                    for (int s = 0; s < H; ++s) ca_in[s] = gat(cstage, (unsigned)tid, s * THREADS);
                }
            }
            cf zpart[H];
            if constexpr (!WHOLE) {
#pragma unroll
                for (int s = 0; s < H; ++s) zpart[s] = lds_load(part + (H - 1 - s) * SP);
            }
#pragma unroll
            for (int s = 0; s < H; ++s) {
                const bool dc = s == 0 && first;
                const cf za = th.x[s];
                if constexpr (WHOLE) zpart[s] = lds_load(part + (H - 1 - s) * SP);
                const cf zb = dc ? za : zpart[s];
                const cf l0 = add_conj(za, zb), r0 = mi_sub_conj(za, zb);
                cf c = mk(0.f, 0.f), ls = c, rs = c;
                auto add_band = [&](float g2) {
                    if (g2 != 0.f) {
                        cf l = scale(l0, g2), r = scale(r0, g2), cq, lq, rq;
                        mask_bin(l, r, cq, lq, rq);
                        c = c + cq; ls = ls + lq; rs = rs + rq;
                    }
                };
                if constexpr (WHOLE) {
                    for (int q = 0; q < a.n_gain; ++q) add_band(gat(opaque(a.gain + q * a.gain_stride), (unsigned)lane, s * L));
                } else {
                    add_band(th.g0[s]);
                    if (a.n_gain > 1) {
                        add_band(th.g1[s]);
                        for (int q = 2; q < a.n_gain; ++q) add_band(a.gain[q * a.gain_stride + lane + s * L]);
                    }
                }
                th.x[s] = swap_add_i(ls, rs);
                const cf ym = swap_conj_add_i(ls, rs);
                if (s == 0) {
                    cf* dst = first ? nyq : part + (H - 1) * SP;
                    *dst = first ? cswap(nyq_y) : ym;
                } else {
                    part[(H - 1 - s) * SP] = ym;
                }
                const cf cv = dc ? mk(c.x, nyq_c) : c;
                if (half == 0) {
                    if constexpr (WHOLE) gat(cstage, (unsigned)tid, s * THREADS) = cv;
                    else th.cs[s] = cv;
                } else {
                    cf ca, cb = cv;
                    if constexpr (WHOLE) ca = ca_in[s];
                    else ca = th.cs[s];
                    cf ck = swap_add_i(ca, cb), cm = swap_conj_add_i(ca, cb);
                    if (dc) {
                        ck = mk(cb.x, ca.x);
                        cm = mk(cb.y, ca.y);
                    }
                    if constexpr (WHOLE) {
                        gat(cstage, (unsigned)tid, s * THREADS) = ck;
                        gat(cstage, (unsigned)tid, (H + s) * THREADS) = cm;
                    } else {
                        th.cs[s] = ck;
                        th.part[s] = cm;
                    }
                }
            }
        });
        inverse_to(a.y + (size_t)(2 * pp + half) * N);
    }
    if (pair_ok) {
        ex.each([&](int tid, Thread& th) {
            const bool first = r_of(tid) == 0 && tid % L == 0;
            cf* const part = partner(tid);
            const UPX_GLOBAL cf* const cstage = opaque(a.yc + (size_t)pp * N);
#pragma unroll
            for (int s = 0; s < H; ++s) {
                cf own, mir;
                if constexpr (WHOLE) {
                    own = gat(cstage, (unsigned)tid, s * THREADS);
                    mir = gat(cstage, (unsigned)tid, (H + s) * THREADS);
                } else {
                    own = th.cs[s];
                    mir = th.part[s];
                }
                th.x[s] = own;
                if (s == 0) {
                    cf* dst = first ? lds_all + H * SP : part + (H - 1) * SP;
                    *dst = mir;
                } else {
                    part[(H - 1 - s) * SP] = mir;
                }
            }
        });
        inverse_to(a.yc + (size_t)pp * N);
    }
}

// ---- overlap-add of the chunk's frames into the output planes; one thread per sample -------
template <class B>
UPX_HD void big_ola(const BigArgs& a, long long gid) {
    constexpr int N = B::N;
    const long long n = (long long)a.m0 * a.hop + gid;
    if (n >= (long long)a.m1 * a.hop || n >= a.t_out) return;
    // frames covering n: j*hop <= n < j*hop + N, increasing j (the reference's accumulation order)
    long long j_first = (n - N + a.hop) / a.hop;     // ceil((n - N + 1) / hop) for n - N + 1 > 0
    if (n - N + 1 <= 0) j_first = 0;
    if (j_first < a.j_lo) j_first = a.j_lo;
    long long j_last = n / a.hop;
    if (j_last >= a.j_hi) j_last = a.j_hi - 1;
    float acc_c = 0.f, acc_l = 0.f, acc_r = 0.f;
    for (long long j = j_first; j <= j_last; ++j) {
        const int jj = (int)(j - a.j0);                  // inside the chunk for every emitted block
        const int idx = (int)(n - j * a.hop);
        const float w = a.w_s[idx];
        const cf lr = a.y[(size_t)jj * N + idx];        // inverse transforms end in natural order
        const cf cc = a.yc[(size_t)(jj >> 1) * N + idx];
        acc_l += lr.y * w;                              // swapped outputs: Re = .y, Im = .x
        acc_r += lr.x * w;
        acc_c += ((jj & 1) == 0 ? cc.y : cc.x) * w;
    }
    if (a.accumulate) {
        a.out_c[n] += acc_c;
        a.out_l[n] += acc_l;
        a.out_r[n] += acc_r;
    } else {
        a.out_c[n] = acc_c;
        a.out_l[n] = acc_l;
        a.out_r[n] = acc_r;
    }
}

// ---- step 2 and the overlap-add in ONE pass (round 6) ------------------------------------------------------------
// For hop = N/2, N/4, N/8 a hop is a whole number HS = 16 / K of N2-sample slots, so the last step of the inverse
// transforms (twiddle, radix-16 over k1: sample n1 N2 + n2 of a frame comes out of column n2) can keep the running
// overlap-add sum of its column in registers exactly like a fused stream does: thread = (range of RB emitted blocks,
// column n2) walks the frames that feed its blocks in increasing j - slot s of the three 16-slot accumulators holds the
// sum at sample (j HS + s) N2 + n2 -, emits the HS finished slots of block j and shifts.  y and yc are read ONCE (+ the
// K - 1 frames in front of a range) and never written back: big_step2_inv's read + write of both and big_ola's gather
// read are gone - 3 of the 5.3 passes over the scratch per chunk.  Same float32 additions in the same order as big_ola
// (from 0.0f, increasing j).  Consecutive n2 read and write consecutive addresses.
template <class B, int K>
UPX_HD void big_tail(const BigArgs& a, int rb, long long gid) {
    constexpr int N = B::N, N2 = B::N2, HS = 16 / K;
    static_assert(B::N1 == 16 && (K == 2 || K == 4 || K == 8), "hop = N/2, N/4, N/8 of a four-step frame");
    const int n2 = (int)(gid % N2);
    const long long m_a = (long long)a.m0 + (gid / N2) * rb;
    if (m_a >= a.m1) return;
    const long long m_b = m_a + rb < a.m1 ? m_a + rb : a.m1;
    float acc_c[16], acc_l[16], acc_r[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) acc_c[s] = acc_l[s] = acc_r[s] = 0.f;
    cf cc[16];                       // the pair's centre signals: frame jj even -> .y, odd -> .x (swapped outputs)
    int pair_have = -1;
    long long j = m_a - (K - 1);
    if (j < a.j_lo) j = a.j_lo;      // (frames in front of the first one do not exist: their slots stay 0)
    // accumulator slot 0 stands for block j of this first frame; earlier blocks of the range (m_a < j never happens:
    // j <= m_a) need no skipping
    for (; j < m_b; ++j) {
        if (j < a.j_hi) {
            const int jj = (int)(j - a.j0);
            const cf* p = a.y + (size_t)jj * N;
            // the twiddle column and the window column are the same for every frame: fetched per frame all the same (L2
            // hits) - hoisted out of the loop they are 46 registers and the kernel spills at two waves per SIMD
            const auto* tw = opaque(a.tw_n);
            const auto* ws = opaque(a.w_s);
            cf v[16];
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) v[k1] = k1 == 0 ? p[n2] : cmul(p[k1 * N2 + n2], tw[k1 * N2 + n2]);
            Dft<16>::run(v);
            if ((jj >> 1) != pair_have) {
                pair_have = jj >> 1;
                const cf* q = a.yc + (size_t)pair_have * N;
#pragma unroll
                for (int k1 = 0; k1 < 16; ++k1) cc[k1] = k1 == 0 ? q[n2] : cmul(q[k1 * N2 + n2], tw[k1 * N2 + n2]);
                Dft<16>::run(cc);
            }
            const bool even = (jj & 1) == 0;
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                const float w = ws[n1 * N2 + n2];
                acc_l[n1] += v[n1].y * w;                 // swapped outputs: Re = .y, Im = .x
                acc_r[n1] += v[n1].x * w;
                acc_c[n1] += (even ? cc[n1].y : cc[n1].x) * w;
            }
        }
        if (j >= m_a) {
#pragma unroll
            for (int s = 0; s < HS; ++s) {
                const long long n = (j * HS + s) * N2 + n2;
                if (n < a.t_out) {
                    if (a.accumulate) {
                        a.out_c[n] += acc_c[s];
                        a.out_l[n] += acc_l[s];
                        a.out_r[n] += acc_r[s];
                    } else {
                        a.out_c[n] = acc_c[s];
                        a.out_l[n] = acc_l[s];
                        a.out_r[n] = acc_r[s];
                    }
                }
            }
        }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            acc_c[s] = s + HS < 16 ? acc_c[s + HS] : 0.f;
            acc_l[s] = s + HS < 16 ? acc_l[s + HS] : 0.f;
            acc_r[s] = s + HS < 16 ? acc_r[s + HS] : 0.f;
        }
    }
}

// host: W_N^(k1 n2) table, [16][N2]
template <class B, class TrigFn>
inline void fill_big_twiddles(cf* tw, TrigFn trig) {
    for (int k1 = 0; k1 < 16; ++k1)
        for (int n2 = 0; n2 < B::N2; ++n2) {
            double c, s;
            trig((double)k1 * (double)n2 / (double)B::N, c, s);
            tw[k1 * B::N2 + n2] = mk((float)c, (float)-s);
        }
}

}   // namespace upx
