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
// Frames are processed in chunks of CH frames (scratch stays L2/MALL resident):
//     step1(audio*w_A) -> rows -> mask -> rows,step2 on Ls+iRs -> rows,step2 on Ca+iCb -> overlap-add
// with the same conventions as the fused kernel (upx_core.h): frame pairs
// (odd j, j+1) share one centre transform, inverse by re/im swap, contributions
// added in increasing j in float32, bands summed in list order.
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
    int j_lo, j_hi;        // frames that exist
    int j0;                // first frame of the chunk (odd): chunk frames j0 .. j0+ch-1
    int ch;                // frames in the chunk (even)
    int m0, m1;            // hop-blocks this chunk emits: [m0, m1)
    int accumulate;
};

template <int LOG2N, int K>
struct BigCfg {
    static constexpr int N = 1 << LOG2N;
    static constexpr int N2 = N / 16;
    static constexpr int HOP = N / K;
    using Row = Cfg<LOG2N - 4, 4, 16>;   // the N2-point row transform (K unused there)
    UPX_HD static int scr(int i) { return (i & 15) * N2 + (i >> 4); }   // index -> scratch offset after a four-step transform
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
        const long long n = (long long)j * B::HOP + o;
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
template <class B, class Ex>
UPX_HD void big_rows_program(Ex& ex, cf* buf, const cf* tw_global, cf* lds_all, int row) {
    using C = typename B::Row;
    using S = Stream<C>;
    using Thread = ThreadT<C::P>;
    constexpr int LANES = C::LANES, P = C::P;
    cf* const tw = lds_all + C::G * C::PITCH;
    cf* const data = buf + (size_t)row * C::N;
    ex.each([&](int tid, Thread& th) {
        for (int i = tid; i < C::TW_CF; i += C::WG) tw[i] = tw_global[i];
#pragma unroll
        for (int s = 0; s < P; ++s) th.x[s] = data[tid + s * LANES];
        S::template pass_compute<0>(th, tw, tid);
    });
    ex.each([&](int tid, Thread& th) { S::template pass_write<0>(th, lds_all, tid); });
    S::template mid_passes<1>(ex, lds_all, tw);
    ex.each([&](int tid, Thread& th) {
        S::read_all(th, lds_all, tid);
        S::template pass_compute<C::PS::n - 1>(th, tw, tid);
#pragma unroll
        for (int s = 0; s < P; ++s) data[tid + s * LANES] = th.x[s];
    });
}

// ---- mask: one thread per (frame pair, k1, k2 <= N2/2), i.e. in the storage order of z ----------
// bin k = k1 + 16 k2 sits at k1 N2 + k2; its partner N-k at ((16-k1)&15) N2 + (N2 - k2 - (k1 != 0)) & (N2-1):
// consecutive threads read and write consecutive addresses (the partner side in reverse).
template <class B>
UPX_HD void big_mask(const BigArgs& a, long long gid) {
    constexpr int N = B::N, N2 = B::N2, HALF = N2 / 2 + 1;
    const int k2 = (int)(gid % HALF);
    const int k1 = (int)((gid / HALF) % 16);
    const int pp = (int)(gid / (16 * HALF));
    if (pp >= a.ch / 2) return;
    const int k = k1 + 16 * k2;
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

// ---- overlap-add of the chunk's frames into the output planes; one thread per sample -------
template <class B, int K>
UPX_HD void big_ola(const BigArgs& a, long long gid) {
    constexpr int N = B::N, HOP = B::HOP;
    const long long n = (long long)a.m0 * HOP + gid;
    if (n >= (long long)a.m1 * HOP || n >= a.t_out) return;
    const int m = (int)(n / HOP);
    float acc_c = 0.f, acc_l = 0.f, acc_r = 0.f;
#pragma unroll
    for (int d = K - 1; d >= 0; --d) {       // frames m-K+1 .. m, increasing
        const int j = m - d;
        if (j < a.j_lo || j >= a.j_hi) continue;       // frames that do not exist contribute exactly 0
        const int jj = j - a.j0;                        // always inside the chunk for emitted blocks
        const int idx = (int)(n - (long long)j * HOP);
        const float w = a.w_s[idx];
        const int o = idx;                              // inverse transforms end in natural order
        const cf lr = a.y[(size_t)jj * N + o];
        const cf cc = a.yc[(size_t)(jj >> 1) * N + o];
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
