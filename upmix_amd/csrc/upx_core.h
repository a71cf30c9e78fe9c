// upx_core.h - per-band WOLA STFT centre-extraction stream for gfx950 (CDNA4).
//
// One *stream* = N/16 lanes that walk consecutive STFT frames of one band and
// keep the overlap-add state of Ls/C/Rs in registers.  Per frame:
//   load interleaved stereo (L + iR is the complex FFT input) * w_A
//   -> complex FFT_N (Stockham, radix-16 passes through LDS, 16 points/lane)
//   -> split L/R spectra, band gain, per-bin coherence/balance mask
//   -> iFFT_N of (Ls + i Rs); iFFT_N of (C_a + i C_b) once per frame PAIR
//   -> * w_S, overlap-add in registers, emit one hop, shift.
// That is 5 complex FFTs per 2 frames for the reference's 10 real transforms
// (center_extraction.py:366-367 forward, :387-389 inverse), the mask of
// :373-384, the band limiter of :334-351 as a precomputed gain vector, and
// the overlap-add of :392-407 / framing of :426-472 in closed form
// (SURVEY.md section 3.3).
//
// Streams of N <= 2048 run a plain Stockham schedule (Cfg); N = 4096 and 8192 span
// several waves and are routed so that all exchanges but one per transform stay
// inside a wave (WideCfg, below).
//
// The body is written against an executor `Ex` whose `each(f)` runs `f` for
// every thread of the workgroup and then synchronises: a wave-level fence when
// the data exchanged stays inside a wave, s_barrier otherwise (`wg_barrier()`
// where a wide stream's waves meet).  tests/emu instantiates the SAME code with
// sequential executors on the host to check indexing and barrier placement
// without a GPU (test infrastructure only - nothing in the product path runs on
// the CPU).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define UPX_HD __host__ __device__ __forceinline__
#else
#define UPX_HD inline
#endif

namespace upx {

// One complex float.  Under clang (device AND host pass of hipcc, so that kernel signatures agree) it is a
// 2-vector: a complex add is then ONE v_pk_add_f32 and the products below ONE or TWO packed instructions.  A wave
// issues a packed f32 instruction about as fast as a scalar one, and at two waves per SIMD the kernel is bound by
// each wave's own issue stream, so halving the instruction count of the butterflies is what pays.  Other
// compilers (the g++ build of the host emulator) see a plain struct with the same layout and scalar arithmetic.
#if defined(__clang__)
typedef float cf __attribute__((ext_vector_type(2)));
#else
struct cf {
    float x, y;
};
UPX_HD cf operator+(cf a, cf b) { cf r; r.x = a.x + b.x; r.y = a.y + b.y; return r; }
UPX_HD cf operator-(cf a, cf b) { cf r; r.x = a.x - b.x; r.y = a.y - b.y; return r; }
#endif

UPX_HD cf mk(float a, float b) { cf r; r.x = a; r.y = b; return r; }
UPX_HD cf cswap(cf a) { return mk(a.y, a.x); }
UPX_HD cf scale(cf a, float s) {
#if defined(__clang__)
    return a * s;
#else
    return mk(a.x * s, a.y * s);
#endif
}
UPX_HD cf mul_mi(cf a) { return mk(a.y, -a.x); }   // a * (-i)

// The sign / swap patterns of complex arithmetic as single packed instructions (VOP3P op_sel picks the source
// half per result half, neg_lo / neg_hi flip a source per result half); the host forms define the meaning.
#if defined(__HIP_DEVICE_COMPILE__)
#define UPX_PK2(name, text)                                       \
    UPX_HD cf name(cf a, cf b) {                                  \
        cf t;                                                     \
        asm(text : "=v"(t) : "v"(a), "v"(b));                     \
        return t;                                                 \
    }
// a b: (ax bx - ay by, ax by + ay bx)
UPX_HD cf cmul(cf a, cf b) {
    cf t;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=&v"(t)
        : "v"(a), "v"(b));
    return t;
}
UPX_PK2(add_mi, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]")        // a - i b
UPX_PK2(sub_mi, "v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]")        // a + i b
UPX_PK2(add_conj, "v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,1] neg_hi:[0,1]")                                   // a + conj b
UPX_PK2(mi_sub_conj, "v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0] neg_hi:[1,0]")   // -i (a - conj b)
UPX_PK2(swap_add_i, "v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_hi:[0,1]")    // swap(a + i b)
UPX_PK2(swap_conj_add_i, "v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1] neg_lo:[1,0]")   // swap(conj a + i conj b)
#undef UPX_PK2
// g.x z - u with the halves of z (and of u) exchanged: the two spectra of mask() in one instruction each
UPX_HD cf fma_swap_sub(cf z, cf g, cf u) {    // swap(g.x z - u) = (g.x z.y - u.y, g.x z.x - u.x)
    cf t;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[0,0,0] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(t) : "v"(z), "v"(g), "v"(u));
    return t;
}
UPX_HD cf fma_swapz_sub(cf z, cf g, cf u) {   // swap(g.x z) - u = (g.x z.y - u.x, g.x z.x - u.y)
    cf t;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(t) : "v"(z), "v"(g), "v"(u));
    return t;
}
// c + a b and 2 p - t: a twiddled butterfly (p + w q, p - w q) is cfma and twice_minus, three instructions for the pair
// where multiply, add and subtract take four (the difference comes back from the sum: p - w q = 2 p - (p + w q))
UPX_HD cf cfma(cf a, cf b, cf c) {
    cf t;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=&v"(t)
        : "v"(a), "v"(b), "v"(c));
    return t;
}
UPX_HD cf twice_minus(cf p, cf t) {   // (the inline constant is the low half: both result halves select it)
    cf r;
    asm("v_pk_fma_f32 %0, %1, 2.0, %2 op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(p), "v"(t));
    return r;
}
// The pair in ONE asm statement (UPX_BFLY_ASM): sum = c + a b, diff = 2 c - sum.  The backend treats every inline asm that
// defines a VGPR as a possible partial-register writer ("assume inline asm has dst forwarding hazard") and puts an
// `s_nop 0` in front of a consumer that follows it directly - 59 of the 88 s_nop of the N = 256 interior loop sat between
// cfma and its twice_minus (round 6 hazard audit, profiles/r06_hazard_audit.txt).  Inside one statement nothing is inserted;
// v_pk_fma_f32 writes whole registers, and cfma's own second instruction already consumes the first one's result back to back.
#if !defined(UPX_BFLY_ASM)
#define UPX_BFLY_ASM 0
#endif
UPX_HD void bfly(cf a, cf b, cf c, cf& sum, cf& diff) {
#if UPX_BFLY_ASM
    cf t, d;
    asm("v_pk_fma_f32 %0, %2, %3, %4 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
        "v_pk_fma_f32 %1, %4, 2.0, %0 op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]"
        : "=&v"(t), "=&v"(d)
        : "v"(a), "v"(b), "v"(c));
    sum = t;
    diff = d;
#else
    sum = cfma(a, b, c);
    diff = twice_minus(c, sum);
#endif
}
// the same with a compile-time constant b: it lives in a scalar register pair (one scalar source per instruction),
// not in two vector registers that the allocator would have to keep or rebuild
UPX_HD cf cmul_k(cf a, cf b) {
    cf t;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=&v"(t)
        : "v"(a), "s"(b));
    return t;
}
UPX_HD cf cfma_k(cf a, cf b, cf c) {
    cf t;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=&v"(t)
        : "v"(a), "s"(b), "v"(c));
    return t;
}
UPX_HD void bfly_k(cf a, cf b, cf c, cf& sum, cf& diff) {
#if UPX_BFLY_ASM
    cf t, d;
    asm("v_pk_fma_f32 %0, %2, %3, %4 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]\n\t"
        "v_pk_fma_f32 %1, %4, 2.0, %0 op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]"
        : "=&v"(t), "=&v"(d)
        : "v"(a), "s"(b), "v"(c));
    sum = t;
    diff = d;
#else
    sum = cfma_k(a, b, c);
    diff = twice_minus(c, sum);
#endif
}
#else
UPX_HD void bfly(cf a, cf b, cf c, cf& sum, cf& diff);
UPX_HD void bfly_k(cf a, cf b, cf c, cf& sum, cf& diff);
UPX_HD cf cmul_k(cf a, cf b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
UPX_HD cf cfma_k(cf a, cf b, cf c) {
    return mk(__builtin_fmaf(-a.y, b.y, __builtin_fmaf(a.x, b.x, c.x)), __builtin_fmaf(a.y, b.x, __builtin_fmaf(a.x, b.y, c.y)));
}
UPX_HD cf cfma(cf a, cf b, cf c) {
    return mk(__builtin_fmaf(-a.y, b.y, __builtin_fmaf(a.x, b.x, c.x)), __builtin_fmaf(a.y, b.x, __builtin_fmaf(a.x, b.y, c.y)));
}
UPX_HD cf twice_minus(cf p, cf t) { return mk(__builtin_fmaf(2.0f, p.x, -t.x), __builtin_fmaf(2.0f, p.y, -t.y)); }
UPX_HD void bfly(cf a, cf b, cf c, cf& sum, cf& diff) { sum = cfma(a, b, c); diff = twice_minus(c, sum); }
UPX_HD void bfly_k(cf a, cf b, cf c, cf& sum, cf& diff) { sum = cfma_k(a, b, c); diff = twice_minus(c, sum); }
UPX_HD cf cmul(cf a, cf b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
UPX_HD cf add_mi(cf a, cf b) { return mk(a.x + b.y, a.y - b.x); }
UPX_HD cf sub_mi(cf a, cf b) { return mk(a.x - b.y, a.y + b.x); }
UPX_HD cf add_conj(cf a, cf b) { return mk(a.x + b.x, a.y - b.y); }
UPX_HD cf mi_sub_conj(cf a, cf b) { return mk(a.y + b.y, b.x - a.x); }
UPX_HD cf swap_add_i(cf a, cf b) { return mk(a.y + b.x, a.x - b.y); }
UPX_HD cf swap_conj_add_i(cf a, cf b) { return mk(b.x - a.y, a.x + b.y); }
UPX_HD cf fma_swap_sub(cf z, cf g, cf u) { return mk(g.x * z.y - u.y, g.x * z.x - u.x); }
UPX_HD cf fma_swapz_sub(cf z, cf g, cf u) { return mk(g.x * z.y - u.x, g.x * z.x - u.y); }
#endif

UPX_HD float fast_rcp(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(v);
#else
    return 1.0f / v;
#endif
}
UPX_HD float fast_sqrt(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(v);
#else
    return __builtin_sqrtf(v);
#endif
}

// Per-thread table values (twiddles, windows, gains) are the same every frame, so
// the optimiser would hoist ~120 registers of loads out of the frame loop.  Passing
// the table pointer through an empty asm per phase keeps the loads inside the loop
// (they hit L1/L2) and the state in registers small.
#if defined(__HIP_DEVICE_COMPILE__)
#define UPX_GLOBAL __attribute__((address_space(1)))
#else
#define UPX_GLOBAL
#endif
template <class T>
UPX_HD const UPX_GLOBAL T* opaque(const T* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(p));
    return (const UPX_GLOBAL T*)p;   // tables live in global memory: global_load, not flat_load
#else
    return p;
#endif
}
// LDS read of one complex.  (Tried: forcing single ds_read_b64 with volatile address_space(3) loads,
// because the backend pairs neighbours into ds_read2_b64 at half the bytes per clock; the ordering
// constraints of volatile cost more than the pairing - 3.9 ms vs 2.75 ms for C3 - so plain loads stay.)
UPX_HD cf lds_load(const cf* p) { return *p; }

#if !defined(UPX_MASK_ALGEBRA)
#define UPX_MASK_ALGEBRA 1   // single-band launches: mask() through mask_weight (0: the general per-band sum)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define UPX_DEVICE_PASS true
#else
#define UPX_DEVICE_PASS false
#endif
// Stops the instruction scheduler from interleaving independent unrolled
// iterations across this point (it otherwise trades ~130 extra VGPRs for ILP).
#if defined(__HIP_DEVICE_COMPILE__) && defined(UPX_NO_SCHED_FENCE)
#define UPX_SCHED_FENCE() ((void)0)
#define UPX_ALL(c) (__builtin_amdgcn_ballot_w64(!(c)) == 0ull)
#elif defined(__HIP_DEVICE_COMPILE__)
#define UPX_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define UPX_ALL(c) (__builtin_amdgcn_ballot_w64(!(c)) == 0ull)   // true on every active lane of the wave
#else
#define UPX_SCHED_FENCE() ((void)0)
#define UPX_ALL(c) (c)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
template <class T>
UPX_HD const UPX_GLOBAL T* opaque(const UPX_GLOBAL T* p) {
    asm volatile("" : "+s"(p));
    return p;
}
template <class T>
UPX_HD UPX_GLOBAL T* opaque(UPX_GLOBAL T* p) {
    asm volatile("" : "+s"(p));
    return p;
}
#endif
template <class T>
UPX_HD UPX_GLOBAL T* opaque(T* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(p));
#endif
    return (UPX_GLOBAL T*)p;
}

// Element `voff + c` (in elements; c a compile-time constant) of a global array whose base is wave-uniform.
// Written so that the backend emits `global_load/store v, v_off, s[base:base+1] offset:imm`: the base (plus
// the part of c beyond the 12-bit immediate) stays in SGPRs, the only VGPR is the 32-bit byte offset, shared
// by every slot of the array.  Byte offsets fit 32 bits because a launch covers < 2^29 samples.
template <class T>
UPX_HD UPX_GLOBAL T& gat(UPX_GLOBAL T* base, unsigned voff, int c) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int cb = c * (int)sizeof(T);
    UPX_GLOBAL char* b = (UPX_GLOBAL char*)base + (cb & ~4095);
    return *(UPX_GLOBAL T*)(b + ((size_t)(voff * (unsigned)sizeof(T)) + (unsigned)(cb & 4095)));
#else
    return base[(size_t)voff + c];
#endif
}
template <class T>
UPX_HD const UPX_GLOBAL T& gat(const UPX_GLOBAL T* base, unsigned voff, int c) {
    return gat(const_cast<UPX_GLOBAL T*>(base), voff, c);
}

// Plane values are read once and written once per band: non-temporal accesses (`nt`) keep them from displacing the
// input lines that the next three frames read again from L2.
template <class T>
UPX_HD T load_nt(const UPX_GLOBAL T& ref) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_nontemporal_load(&ref);
#else
    return ref;
#endif
}
template <class T>
UPX_HD void store_nt(UPX_GLOBAL T& ref, T v) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_nontemporal_store(v, &ref);
#else
    ref = v;
#endif
}

// A value every lane of the wave holds alike, moved to a scalar register (a table entry fetched with a vector load is
// uniform by value but lives in a VGPR, and everything computed from it - frame numbers, sample offsets, pointers - would
// be vector arithmetic).
UPX_HD int uniform_int(int v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_readfirstlane(v);
#else
    return v;
#endif
}

// The lane's own element offset of gat_u, pinned where it is used: hoisted out of a loop, its zero-extension becomes a
// 64-bit loop-invariant VGPR pair in another basic block, where instruction selection no longer sees `sgpr base + zext(vgpr)`
// and falls back to a 64-bit vector add per access instead of `global_load/store v_off, s[base:base+1]`.  Call it once per
// phase (one v_mov), not per access.
UPX_HD unsigned pin_lane(unsigned voff) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(UPX_GAT_U_PLAIN)
    asm volatile("" : "+v"(voff));
#endif
    return voff;
}

// Element `uniform + voff` of a global array: `uniform` (elements) is the same for every lane and need not be a
// compile-time constant (it joins the base in SGPRs), `voff` is the lane's own 32-bit element offset:
// `global_load/store v, v_off, s[base:base+1]` without 64-bit vector arithmetic.
template <class T>
UPX_HD UPX_GLOBAL T& gat_u(UPX_GLOBAL T* base, long long uniform, unsigned voff) {
#if defined(__HIP_DEVICE_COMPILE__)
    UPX_GLOBAL char* b = (UPX_GLOBAL char*)(base + uniform);
#if !defined(UPX_GAT_U_PLAIN)
    // The uniform pointer is pinned to a scalar register pair HERE.  Without this the optimiser re-associates the
    // address as (base + lane offset) + uniform and strength-reduces the slots of an array into a chain of 64-bit VECTOR
    // adds (v_lshl_add_u64, one per access, each waiting for the previous one): 25-35 per transform in the band-limited
    // synthesis' interior loop, 40 per unit in the analysis - a tenth of their vector instructions (round 4, ISA).
    asm volatile("" : "+s"(b));
    // (the lane's offset must come from pin_lane() in the same basic block, see there)
    return *(UPX_GLOBAL T*)(b + (size_t)(voff * (unsigned)sizeof(T)));
#else
    return *(UPX_GLOBAL T*)(b + (size_t)(voff * (unsigned)sizeof(T)));
#endif
#else
    return base[uniform + (long long)voff];
#endif
}
template <class T>
UPX_HD const UPX_GLOBAL T& gat_u(const UPX_GLOBAL T* base, long long uniform, unsigned voff) {
    return gat_u(const_cast<UPX_GLOBAL T*>(base), uniform, voff);
}

// ---------------------------------------------------------------------------
// Register DFTs, forward sign exp(-2 pi i nk/R), natural-order in and out.
// ---------------------------------------------------------------------------
constexpr float kSqrtHalf = 0.70710678118654752440f;
constexpr float kC16 = 0.92387953251128675613f;   // cos(pi/8)
constexpr float kS16 = 0.38268343236508977173f;   // sin(pi/8)

UPX_HD void dft2(cf& a, cf& b) {
    cf t = a - b;
    a = a + b;
    b = t;
}

UPX_HD void dft4(cf& a0, cf& a1, cf& a2, cf& a3) {
    cf t0 = a0 + a2, t1 = a0 - a2, t2 = a1 + a3, d = a1 - a3;
    a0 = t0 + t2;
    a2 = t0 - t2;
    a1 = add_mi(t1, d);
    a3 = sub_mi(t1, d);
}
// the same with input a2 still to be multiplied by -i
UPX_HD void dft4_mi2(cf& a0, cf& a1, cf& a2, cf& a3) {
    cf t0 = add_mi(a0, a2), t1 = sub_mi(a0, a2), t2 = a1 + a3, d = a1 - a3;
    a0 = t0 + t2;
    a2 = t0 - t2;
    a1 = add_mi(t1, d);
    a3 = sub_mi(t1, d);
}

// UPX_FMA_BUTTERFLY: twiddled butterflies in the multiply-add form (cfma / twice_minus)
#if !defined(UPX_FMA_BUTTERFLY)
#define UPX_FMA_BUTTERFLY 1
#endif
// dft4 of (a0 [w0], a1 w1, a2 w2, a3 w3); W0: a0 carries a twiddle as well.  MI2: w2 = -i (not passed)
// KONST: the twiddles are compile-time constants (scalar registers)
template <bool W0, bool MI2 = false, bool KONST = false>
UPX_HD void dft4_tw(cf& a0, cf& a1, cf& a2, cf& a3, cf w0, cf w1, cf w2, cf w3) {
#if UPX_FMA_BUTTERFLY
    const cf p0 = W0 ? (KONST ? cmul_k(a0, w0) : cmul(a0, w0)) : a0;
    cf t0, t1;
    if constexpr (MI2) {
        t0 = add_mi(p0, a2);
        t1 = sub_mi(p0, a2);
    } else {
        if constexpr (KONST) bfly_k(a2, w2, p0, t0, t1);
        else bfly(a2, w2, p0, t0, t1);
    }
    const cf p1 = KONST ? cmul_k(a1, w1) : cmul(a1, w1);
    cf t2, d;
    if constexpr (KONST) bfly_k(a3, w3, p1, t2, d);
    else bfly(a3, w3, p1, t2, d);
    a0 = t0 + t2;
    a2 = t0 - t2;
    a1 = add_mi(t1, d);
    a3 = sub_mi(t1, d);
#else
    if (W0) a0 = cmul(a0, w0);
    a1 = cmul(a1, w1);
    if (!MI2) a2 = cmul(a2, w2);
    a3 = cmul(a3, w3);
    if (MI2) dft4_mi2(a0, a1, a2, a3);
    else dft4(a0, a1, a2, a3);
#endif
}

// dft4 of (s0 a0, s1 a1, s2 a2, s3 a3) with real factors (a window): the products ride on the first butterflies
UPX_HD void dft4_sc(cf& a0, cf& a1, cf& a2, cf& a3, float s0, float s1, float s2, float s3) {
#if UPX_FMA_BUTTERFLY
    const cf p0 = scale(a0, s0), p1 = scale(a1, s1);
    const cf t0 = scale(a2, s2) + p0, t2 = scale(a3, s3) + p1;   // (contracted: one packed multiply-add each)
    const cf t1 = twice_minus(p0, t0), d = twice_minus(p1, t2);
    a0 = t0 + t2;
    a2 = t0 - t2;
    a1 = add_mi(t1, d);
    a3 = sub_mi(t1, d);
#else
    a0 = scale(a0, s0); a1 = scale(a1, s1); a2 = scale(a2, s2); a3 = scale(a3, s3);
    dft4(a0, a1, a2, a3);
#endif
}

// multiply by W8^1 = (1 - i)/sqrt2 and W8^3 = (-1 - i)/sqrt2
UPX_HD cf mul_w8_1(cf a) { return scale(add_mi(a, a), kSqrtHalf); }    // (x + y, y - x) / sqrt2
UPX_HD cf mul_w8_3(cf a) { return scale(sub_mi(a, a), -kSqrtHalf); }   // -(x - y, y + x) / sqrt2

// ---------------------------------------------------------------------------
// Pruned butterflies.  A band's gain vector (center_extraction.py:334-351) is zero outside its pass band, so the
// bins there carry nothing: the forward transform's last pass need not produce them and the inverse transform's
// first pass sees zeros there.  Which register slots those are is known when the plan is created (Live<> below),
// so the butterflies take two compile-time masks: IM bit i = input i may be non-zero, OM bit i = output i is
// used.  An unused output's variable is left untouched; a dead input's variable is never read.  With both masks
// full every helper is the plain butterfly above (same instructions, same order).
// ---------------------------------------------------------------------------
template <int I>
struct IC {
    static constexpr int value = I;
};
template <int B, int E, class F>
UPX_HD void static_for(F&& f) {
    if constexpr (B < E) {
        f(IC<B>{});
        static_for<B + 1, E>(f);
    }
}
UPX_HD cf cneg(cf a) { return mk(-a.x, -a.y); }
UPX_HD cf mul_pi(cf a) { return mk(-a.y, a.x); }   // a * (+i)
template <bool LA, bool LB>
UPX_HD cf padd(cf a, cf b) {   // a + b
    if constexpr (LA && LB) return a + b;
    else if constexpr (LA) return a;
    else if constexpr (LB) return b;
    else return mk(0.f, 0.f);
}
template <bool LA, bool LB>
UPX_HD cf psub(cf a, cf b) {   // a - b
    if constexpr (LA && LB) return a - b;
    else if constexpr (LA) return a;
    else if constexpr (LB) return cneg(b);
    else return mk(0.f, 0.f);
}
template <bool LA, bool LB>
UPX_HD cf padd_mi(cf a, cf b) {   // a - i b
    if constexpr (LA && LB) return add_mi(a, b);
    else if constexpr (LA) return a;
    else if constexpr (LB) return mul_mi(b);
    else return mk(0.f, 0.f);
}
template <bool LA, bool LB>
UPX_HD cf psub_mi(cf a, cf b) {   // a + i b
    if constexpr (LA && LB) return sub_mi(a, b);
    else if constexpr (LA) return a;
    else if constexpr (LB) return mul_pi(b);
    else return mk(0.f, 0.f);
}
constexpr unsigned bit(unsigned m, int i) { return (m >> i) & 1u; }
// MI2: input a2 is still to be multiplied by -i (dft4_mi2)
template <unsigned IM, unsigned OM, bool MI2 = false>
UPX_HD void dft4_p(cf& a0, cf& a1, cf& a2, cf& a3) {
    if constexpr ((IM & 15u) == 15u && (OM & 15u) == 15u) {
        if constexpr (MI2) dft4_mi2(a0, a1, a2, a3);
        else dft4(a0, a1, a2, a3);
    } else {
        constexpr bool l0 = bit(IM, 0), l1 = bit(IM, 1), l2 = bit(IM, 2), l3 = bit(IM, 3);
        constexpr bool le = l0 || l2, lo = l1 || l3;
        cf t0 = mk(0.f, 0.f), t1 = t0, t2 = t0, d = t0;
        if constexpr ((OM & 5u) != 0) {   // outputs 0, 2
            t0 = MI2 ? padd_mi<l0, l2>(a0, a2) : padd<l0, l2>(a0, a2);
            t2 = padd<l1, l3>(a1, a3);
        }
        if constexpr ((OM & 10u) != 0) {   // outputs 1, 3
            t1 = MI2 ? psub_mi<l0, l2>(a0, a2) : psub<l0, l2>(a0, a2);
            d = psub<l1, l3>(a1, a3);
        }
        if constexpr (bit(OM, 0)) a0 = padd<le, lo>(t0, t2);
        if constexpr (bit(OM, 2)) a2 = psub<le, lo>(t0, t2);
        if constexpr (bit(OM, 1)) a1 = padd_mi<le, lo>(t1, d);
        if constexpr (bit(OM, 3)) a3 = psub_mi<le, lo>(t1, d);
    }
}

template <int R>
struct Dft;

// run_tw(v, w): the DFT of (v[0], v[1] w[1], ..., v[R-1] w[R-1]) - a pass's twiddles folded into the first butterflies
template <>
struct Dft<2> {
    static UPX_HD void run(cf* v) { dft2(v[0], v[1]); }
    static UPX_HD void run_tw(cf* v, const cf* w) {
#if UPX_FMA_BUTTERFLY
        cf t, d;
        bfly(v[1], w[1], v[0], t, d);
        v[1] = d;
        v[0] = t;
#else
        v[1] = cmul(v[1], w[1]);
        run(v);
#endif
    }
    template <unsigned IM, unsigned OM>
    static UPX_HD void run_p(cf* v) {
        constexpr bool l0 = bit(IM, 0), l1 = bit(IM, 1);
        const cf a = v[0], b = v[1];
        if constexpr (bit(OM, 0)) v[0] = padd<l0, l1>(a, b);
        if constexpr (bit(OM, 1)) v[1] = psub<l0, l1>(a, b);
    }
};
template <>
struct Dft<4> {
    static UPX_HD void run(cf* v) { dft4(v[0], v[1], v[2], v[3]); }
    static UPX_HD void run_tw(cf* v, const cf* w) { dft4_tw<false>(v[0], v[1], v[2], v[3], w[0], w[1], w[2], w[3]); }
    template <unsigned IM, unsigned OM>
    static UPX_HD void run_p(cf* v) { dft4_p<IM, OM>(v[0], v[1], v[2], v[3]); }
};
template <>
struct Dft<8> {
    static UPX_HD void run_tw(cf* v, const cf* w) {
        // first stage: the two DFT4 over a take inputs (0, 2, 4, 6) and (1, 3, 5, 7)
        cf e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
        cf o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
        dft4_tw<false>(e0, e1, e2, e3, w[0], w[2], w[4], w[6]);
        dft4_tw<true>(o0, o1, o2, o3, w[1], w[3], w[5], w[7]);
        second(v, e0, e1, e2, e3, o0, o1, o2, o3);
    }
    static UPX_HD void second(cf* v, cf e0, cf e1, cf e2, cf e3, cf o0, cf o1, cf o2, cf o3) {
        v[0] = e0 + o0; v[4] = e0 - o0;
        v[2] = add_mi(e2, o2); v[6] = sub_mi(e2, o2);   // o2 * (-i)
#if UPX_FMA_BUTTERFLY
        bfly_k(o1, mk(kSqrtHalf, -kSqrtHalf), e1, v[1], v[5]);
        bfly_k(o3, mk(-kSqrtHalf, -kSqrtHalf), e3, v[3], v[7]);
#else
        o1 = mul_w8_1(o1);
        o3 = mul_w8_3(o3);
        v[1] = e1 + o1; v[5] = e1 - o1;
        v[3] = e3 + o3; v[7] = e3 - o3;
#endif
    }
    // n = 2a + b: two DFT4 over a, twiddle W8^(b k1), DFT2 over b
    static UPX_HD void run(cf* v) {
        cf e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
        cf o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
        dft4(e0, e1, e2, e3);
        dft4(o0, o1, o2, o3);
        second(v, e0, e1, e2, e3, o0, o1, o2, o3);
    }
    template <unsigned IM, unsigned OM>
    static UPX_HD void run_p(cf* v) {
        if constexpr ((IM & 255u) == 255u && (OM & 255u) == 255u) {
            run(v);
        } else {
            constexpr unsigned ime = bit(IM, 0) | bit(IM, 2) << 1 | bit(IM, 4) << 2 | bit(IM, 6) << 3;
            constexpr unsigned imo = bit(IM, 1) | bit(IM, 3) << 1 | bit(IM, 5) << 2 | bit(IM, 7) << 3;
            constexpr unsigned om4 = (OM | OM >> 4) & 15u;   // e_k and o_k feed outputs k and k + 4
            constexpr bool le = ime != 0, lo = imo != 0;
            cf e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6];
            cf o0 = v[1], o1 = v[3], o2 = v[5], o3 = v[7];
            dft4_p<ime, om4>(e0, e1, e2, e3);
            dft4_p<imo, om4>(o0, o1, o2, o3);
            if constexpr (lo && bit(om4, 1)) o1 = mul_w8_1(o1);
            if constexpr (lo && bit(om4, 3)) o3 = mul_w8_3(o3);
            if constexpr (bit(OM, 0)) v[0] = padd<le, lo>(e0, o0);
            if constexpr (bit(OM, 4)) v[4] = psub<le, lo>(e0, o0);
            if constexpr (bit(OM, 1)) v[1] = padd<le, lo>(e1, o1);
            if constexpr (bit(OM, 5)) v[5] = psub<le, lo>(e1, o1);
            if constexpr (bit(OM, 2)) v[2] = padd_mi<le, lo>(e2, o2);
            if constexpr (bit(OM, 6)) v[6] = psub_mi<le, lo>(e2, o2);
            if constexpr (bit(OM, 3)) v[3] = padd<le, lo>(e3, o3);
            if constexpr (bit(OM, 7)) v[7] = psub<le, lo>(e3, o3);
        }
    }
};
template <>
struct Dft<16> {
    // the second stage: DFT4 over b of W16^(b k1) y[b][k1]; the constants ride on the butterflies (dft4_tw)
    static UPX_HD void second(cf* v, cf (&y)[4][4]) {
        const cf w1 = mk(kC16, -kS16), w2 = mk(kSqrtHalf, -kSqrtHalf), w3 = mk(kS16, -kC16);
        const cf w6 = mk(-kSqrtHalf, -kSqrtHalf), w9 = mk(-kC16, kS16), one = mk(1.f, 0.f);
        dft4(y[0][0], y[1][0], y[2][0], y[3][0]);
        dft4_tw<false, false, true>(y[0][1], y[1][1], y[2][1], y[3][1], one, w1, w2, w3);
        dft4_tw<false, true, true>(y[0][2], y[1][2], y[2][2], y[3][2], one, w2, one, w6);   // W16^4 = -i inside
        dft4_tw<false, false, true>(y[0][3], y[1][3], y[2][3], y[3][3], one, w3, w6, w9);
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) {
            v[k1] = y[0][k1]; v[k1 + 4] = y[1][k1]; v[k1 + 8] = y[2][k1]; v[k1 + 12] = y[3][k1];
        }
    }
    // the DFT of (s[0] v[0], ..., s[15] v[15]), s real, in two steps and in place: first_sc leaves y[b][k1] in v[4 k1 + b],
    // second_inplace finishes
    static UPX_HD void first_sc(cf* v, const float* s) {
#pragma unroll
        for (int b = 0; b < 4; ++b) dft4_sc(v[b], v[4 + b], v[8 + b], v[12 + b], s[b], s[4 + b], s[8 + b], s[12 + b]);
    }
    static UPX_HD void second_inplace(cf* v) {
        cf y[4][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            y[b][0] = v[b]; y[b][1] = v[4 + b]; y[b][2] = v[8 + b]; y[b][3] = v[12 + b];
        }
        second(v, y);
    }
    static UPX_HD void run_tw(cf* v, const cf* w) {
        cf y[4][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            y[b][0] = v[b]; y[b][1] = v[4 + b]; y[b][2] = v[8 + b]; y[b][3] = v[12 + b];
        }
        dft4_tw<false>(y[0][0], y[0][1], y[0][2], y[0][3], w[0], w[4], w[8], w[12]);
        dft4_tw<true>(y[1][0], y[1][1], y[1][2], y[1][3], w[1], w[5], w[9], w[13]);
        dft4_tw<true>(y[2][0], y[2][1], y[2][2], y[2][3], w[2], w[6], w[10], w[14]);
        dft4_tw<true>(y[3][0], y[3][1], y[3][2], y[3][3], w[3], w[7], w[11], w[15]);
        second(v, y);
    }
    // n = 4a + b, k = k1 + 4 k2:  X[k1+4k2] = DFT4_b( W16^(b k1) * DFT4_a(v[4a+b])[k1] )[k2]
    static UPX_HD void run(cf* v) {
        cf y[4][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            y[b][0] = v[b]; y[b][1] = v[4 + b]; y[b][2] = v[8 + b]; y[b][3] = v[12 + b];
            dft4(y[b][0], y[b][1], y[b][2], y[b][3]);
        }
#if UPX_FMA_BUTTERFLY
        second(v, y);
#else
        const cf w1 = mk(kC16, -kS16), w3 = mk(kS16, -kC16);
        y[1][1] = cmul(y[1][1], w1);
        y[1][2] = mul_w8_1(y[1][2]);
        y[1][3] = cmul(y[1][3], w3);
        y[2][1] = mul_w8_1(y[2][1]);
        y[2][3] = mul_w8_3(y[2][3]);   // y[2][2] * (-i) happens inside dft4_mi2
        y[3][1] = cmul(y[3][1], w3);
        y[3][2] = mul_w8_3(y[3][2]);
        y[3][3] = cmul(y[3][3], mk(-kC16, kS16));   // W16^9 = -W16^1
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) {
            if (k1 == 2) dft4_mi2(y[0][k1], y[1][k1], y[2][k1], y[3][k1]);
            else dft4(y[0][k1], y[1][k1], y[2][k1], y[3][k1]);
            v[k1] = y[0][k1]; v[k1 + 4] = y[1][k1]; v[k1 + 8] = y[2][k1]; v[k1 + 12] = y[3][k1];
        }
#endif
    }
    // inputs 4a + b with mask bit set may be non-zero, outputs k1 + 4 k2 with mask bit set are used
    static constexpr unsigned col4(unsigned m, int b) { return bit(m, b) | bit(m, 4 + b) << 1 | bit(m, 8 + b) << 2 | bit(m, 12 + b) << 3; }
    template <unsigned IM, unsigned OM>
    static UPX_HD void run_p(cf* v) {
        if constexpr ((IM & 0xFFFFu) == 0xFFFFu && (OM & 0xFFFFu) == 0xFFFFu) {
            run(v);
        } else {
            // first stage: column b is live if any of its inputs is; its output k1 is used if any output k1 + 4 k2 is
            constexpr unsigned om1 = (col4(OM, 0) ? 1u : 0u) | (col4(OM, 1) ? 2u : 0u) | (col4(OM, 2) ? 4u : 0u) | (col4(OM, 3) ? 8u : 0u);
            constexpr unsigned ly = (col4(IM, 0) ? 1u : 0u) | (col4(IM, 1) ? 2u : 0u) | (col4(IM, 2) ? 4u : 0u) | (col4(IM, 3) ? 8u : 0u);
            cf y[4][4] = {};   // (dead columns stay zero and are never read)
            static_for<0, 4>([&](auto ib) {
                constexpr int b = decltype(ib)::value;
                if constexpr (bit(ly, b)) {
                    y[b][0] = v[b]; y[b][1] = v[4 + b]; y[b][2] = v[8 + b]; y[b][3] = v[12 + b];
                    dft4_p<col4(IM, b), om1>(y[b][0], y[b][1], y[b][2], y[b][3]);
                }
            });
            const cf w1 = mk(kC16, -kS16), w3 = mk(kS16, -kC16);
            if constexpr (bit(ly, 1) && bit(om1, 1)) y[1][1] = cmul(y[1][1], w1);
            if constexpr (bit(ly, 1) && bit(om1, 2)) y[1][2] = mul_w8_1(y[1][2]);
            if constexpr (bit(ly, 1) && bit(om1, 3)) y[1][3] = cmul(y[1][3], w3);
            if constexpr (bit(ly, 2) && bit(om1, 1)) y[2][1] = mul_w8_1(y[2][1]);
            if constexpr (bit(ly, 2) && bit(om1, 3)) y[2][3] = mul_w8_3(y[2][3]);   // y[2][2] * (-i): dft4_p<.., MI2>
            if constexpr (bit(ly, 3) && bit(om1, 1)) y[3][1] = cmul(y[3][1], w3);
            if constexpr (bit(ly, 3) && bit(om1, 2)) y[3][2] = mul_w8_3(y[3][2]);
            if constexpr (bit(ly, 3) && bit(om1, 3)) y[3][3] = cmul(y[3][3], mk(-kC16, kS16));
            static_for<0, 4>([&](auto ik) {
                constexpr int k1 = decltype(ik)::value;
                constexpr unsigned om2 = bit(OM, k1) | bit(OM, k1 + 4) << 1 | bit(OM, k1 + 8) << 2 | bit(OM, k1 + 12) << 3;
                if constexpr (om2 != 0) {
                    dft4_p<ly, om2, k1 == 2>(y[0][k1], y[1][k1], y[2][k1], y[3][k1]);
                    if constexpr (bit(om2, 0)) v[k1] = y[0][k1];
                    if constexpr (bit(om2, 1)) v[k1 + 4] = y[1][k1];
                    if constexpr (bit(om2, 2)) v[k1 + 8] = y[2][k1];
                    if constexpr (bit(om2, 3)) v[k1 + 12] = y[3][k1];
                }
            });
        }
    }
};

// ---------------------------------------------------------------------------
// Static configuration per (log2 N, K = frames overlapping one sample)
// ---------------------------------------------------------------------------
// Radix schedule per (log2 N, P = complex points per lane).  Passes that go through
// LDS are radix-P (one butterfly per lane); only the final pass may be smaller.
template <int LOG2N, int P>
struct Passes;
template <> struct Passes<8, 16>  { static constexpr int n = 2; static constexpr int r[6] = {16, 16, 1, 1, 1, 1}; };
template <> struct Passes<9, 16>  { static constexpr int n = 3; static constexpr int r[6] = {16, 16, 2, 1, 1, 1}; };
template <> struct Passes<10, 16> { static constexpr int n = 3; static constexpr int r[6] = {16, 16, 4, 1, 1, 1}; };
template <> struct Passes<11, 16> { static constexpr int n = 3; static constexpr int r[6] = {16, 16, 8, 1, 1, 1}; };
template <> struct Passes<12, 16> { static constexpr int n = 3; static constexpr int r[6] = {16, 16, 16, 1, 1, 1}; };
template <> struct Passes<13, 16> { static constexpr int n = 4; static constexpr int r[6] = {16, 16, 16, 2, 1, 1}; };
template <> struct Passes<6, 8>   { static constexpr int n = 2; static constexpr int r[6] = {8, 8, 1, 1, 1, 1}; };
template <> struct Passes<7, 8>   { static constexpr int n = 3; static constexpr int r[6] = {8, 8, 2, 1, 1, 1}; };
template <> struct Passes<8, 8>   { static constexpr int n = 3; static constexpr int r[6] = {8, 8, 4, 1, 1, 1}; };
template <> struct Passes<9, 8>   { static constexpr int n = 3; static constexpr int r[6] = {8, 8, 8, 1, 1, 1}; };
template <> struct Passes<10, 8>  { static constexpr int n = 4; static constexpr int r[6] = {8, 8, 8, 2, 1, 1}; };
template <> struct Passes<11, 8>  { static constexpr int n = 4; static constexpr int r[6] = {8, 8, 8, 4, 1, 1}; };
template <> struct Passes<12, 8>  { static constexpr int n = 4; static constexpr int r[6] = {8, 8, 8, 8, 1, 1}; };
template <> struct Passes<13, 8>  { static constexpr int n = 5; static constexpr int r[6] = {8, 8, 8, 8, 2, 1}; };

constexpr int pass_ns(const int* r, int p) { return p == 0 ? 1 : r[p - 1] * pass_ns(r, p - 1); }
// offset (in complex) of pass p's rows in the compact twiddle table: passes 1..p-1 hold (R-1) rows of NS entries
constexpr int tw_offset(const int* r, int p) {
    return p <= 1 ? 0 : (r[p - 1] - 1) * pass_ns(r, p - 1) + tw_offset(r, p - 1);
}

template <int LOG2N_, int K_, int P_ = 16>
struct Cfg {
    static constexpr int LOG2N = LOG2N_;
    static constexpr int N = 1 << LOG2N_;
    static constexpr int K = K_;                         // hop = N / K
    static constexpr int P = P_;                         // complex points per lane
    static constexpr int HOP = N / K_;
    static constexpr int LANES = N / P_;                 // lanes per stream
    static constexpr int WG = LANES < 64 ? 64 : LANES;   // threads per workgroup (>= one wave)
    static constexpr int G = WG / LANES;                 // streams per workgroup
    static constexpr int HS = P_ / K_;                   // register slots per hop
    static constexpr int SPITCH = LANES + LANES / P_;    // padded distance of one slot step
    static constexpr int PITCH = N + N / P_ + P_;        // padded complex per stream buffer (+1 row: index N is addressable)
    using PS = Passes<LOG2N_, P_>;
    static constexpr int TW_CF = tw_offset(PS::r, PS::n) > 0 ? tw_offset(PS::r, PS::n) : 1;   // twiddle table entries
    static constexpr int LDS_CF = G * PITCH + TW_CF;     // stream buffers + twiddle table
    static constexpr bool WAVE_SYNC = LANES <= 64;       // a stream lives inside one wave: no workgroup barrier needed
    static constexpr bool WIDE = false;
    // band_program may issue short loads early where registers allow (single-wave streams, hop >= N/4): the analysis
    // window of a head between the preceding tail's arithmetic and its stores, the mask's gains in front of the prefetch
    static constexpr bool EARLY_LOADS = LANES <= 64 && K_ <= 4 && P_ == 16;
    // The exchange in front of the last pass moves values between the 16-lane rows of a wave only when a stream is 2 or
    // 4 rows wide and the last pass has that radix (N = 512: 16.16.2, N = 1024: 16.16.4): on the device it is done in
    // registers (Stream::row_exchange) instead of through LDS.  The host emulator keeps the LDS route (same values).
    static constexpr bool SWAP_LAST = UPX_DEVICE_PASS && P_ == 16 && PS::n == 3 && PS::r[1] == 16 &&
                                      ((LANES == 64 && PS::r[2] == 4) || (LANES == 32 && PS::r[2] == 2));
    using Sub = Cfg;                                     // the FFT that runs through LDS is the whole frame
    static_assert(K_ == 2 || K_ == 4 || K_ == 8 || K_ == 16, "hop must be N/2, N/4, N/8 or N/16");
    static_assert(P_ % K_ == 0 && P_ >= K_, "hop must be a whole number of register slots");
    static_assert(LANES % P_ == 0 && WG <= 1024, "unsupported size");
};

// ---------------------------------------------------------------------------
// Wide streams (N = 4096, 8192): the frame spans 4 / 8 waves.  A plain Stockham schedule makes every pass a
// workgroup-wide exchange (two s_barriers each), and the waves of the only resident workgroup then move in
// lock-step: LDS phases and butterfly phases never overlap.  Here the same radix schedule is routed so that all
// but ONE exchange per transform stay inside a wave:
//   N = 16 * N2:  radix-16 over n1 in registers (lane n2 holds x[n2 + N2 n1]), * W_N^(k1 n2),
//                 transposed write -> | s_barrier | -> sixteen N2-point sub-FFTs over n2, one per k1,
//                 each living in N2/16 lanes of ONE wave (Stockham through that wave's LDS region, wave-level
//                 ordering only) -> X[k1 + 16 k2].
// Sub-FFT g holds k1 = (g even ? g/2 : 16 - g/2) (g = 0, 1 hold k1 = 0, 8), so a bin and its mirror N-k sit in
// sub-FFTs g and g^1 of the SAME wave: the L/R split, the mask and the construction of the inverse input need no
// barrier either.  The inverse runs the mirror image (sub-FFTs, * W_N^(k1 n2), | s_barrier |, radix-16 over k1),
// which lands sample n2 + N2 n1 in slot n1 of lane n2 - the overlap-add layout.  Between the two barriers of a
// transform pair each wave runs ~10 phases on its own, so the two waves of a SIMD drift apart and one computes
// while the other exchanges.  Cross-wave phases touch lane-private cells (g(r), n2 = lane), r = 0..15.
// ---------------------------------------------------------------------------
constexpr int wide_sub_of_k1(int k1) { return k1 == 0 ? 0 : k1 == 8 ? 1 : k1 < 8 ? 2 * k1 : 2 * (16 - k1) + 1; }
constexpr int wide_k1_of_sub(int g) { return (g & 1) ? ((g >> 1) ? 16 - (g >> 1) : 8) : (g >> 1); }

template <int LOG2N_, int K_>
struct WideCfg {
    static constexpr int LOG2N = LOG2N_;
    static constexpr int N = 1 << LOG2N_;
    static constexpr int K = K_;
    static constexpr int P = 16;
    static constexpr int HOP = N / K_;
    static constexpr int LANES = N / 16;                 // = N2; lane n2 owns samples n2 + N2 n1
    static constexpr int WG = LANES;
    static constexpr int G = 1;
    static constexpr int HS = 16 / K_;
    using Sub = Cfg<LOG2N_ - 4, K_, 16>;                 // the N2-point sub-FFT (K is not used by it)
    static constexpr int SPITCH = Sub::SPITCH;           // slot step inside a sub-FFT buffer
    static constexpr int PITCH = 16 * Sub::PITCH;        // sixteen sub-FFT buffers
    static constexpr int BT_ROW = LANES + 1;             // row pitch of W_N^(k1 n2), k1 = 0..15 (odd: the backend cannot
                                                         // pair two rows into a half-rate ds_read2_b64 / ds_read2st64_b64)
    static constexpr int BT_CF = 16 * BT_ROW;
    static constexpr int TW_CF = Sub::TW_CF + BT_CF;     // sub-FFT twiddles, then the big table
    static constexpr int LDS_CF = PITCH + TW_CF;
    static constexpr bool WAVE_SYNC = false;
    static constexpr bool WIDE = true;
    static constexpr bool EARLY_LOADS = false;
    static_assert(LOG2N_ == 12 || LOG2N_ == 13, "wide streams cover N = 4096 and 8192");
    static_assert(64 % Sub::LANES == 0 && (64 / Sub::LANES) % 2 == 0, "a wave holds whole (g, g^1) pairs of sub-FFTs");
};

// LDS index padding: one spare complex after every P, so that the stride-P
// scatter of the first Stockham pass (lane j writes P j + r) spreads over all
// banks.  padp(a + b) == padp(a) + b + b/P whenever b is a multiple of P.
template <int P>
UPX_HD int padp(int i) { return i + i / P; }

// ---------------------------------------------------------------------------
// Kernel arguments (one band, one launch).  Sample counts per launch are
// limited to 2^29 so that byte offsets fit 32 bits; the host shards longer
// signals (the same seam arithmetic as the multi-GPU path).
// ---------------------------------------------------------------------------
struct BandArgs {
    const cf* in;          // interleaved stereo, local sample 0; (L,R) is read as L + iR
    float* out_c;          // planar outputs, local sample 0
    float* out_l;
    float* out_r;
    const float* w_a;      // analysis window [N]
    const float* w_s;      // synthesis window / N [N]
    const float* gain;     // 0.5 * band-limit gain, [n_gain][gain_stride] (see below)
    const cf* tw;          // inter-pass twiddles, [row][LANES]
    int t_in;              // valid input samples  [0, t_in)
    int t_out;             // valid output samples [0, t_out)
    int j_lo, j_hi;        // frames that exist: j in [j_lo, j_hi), frame j starts at j*hop
    int m_lo, m_hi;        // hop-blocks to emit: m in [m_lo, m_hi)
    int blocks_per_stream; // F (frames of every stream when stream_m0 == nullptr)
    // Streams of unequal length: stream sid transforms the frames [stream_m0[sid], stream_m0[sid + 1]) (n_streams + 1
    // entries, even lengths, equal inside a workgroup).  The first and the last workgroups of a launch run the slower
    // signal-edge flavour: with streams as long as everybody's they end ~12 % late and the launch waits for them, so the
    // host hands them shorter streams (upx_process_device).  nullptr: stream sid starts at m_lo - 1 + sid F.
    const int* stream_m0;
    // Adjacent bands that share N, hop and windows are MERGED into one launch: their transforms are
    // the same linear operators, so sum_b OLA(iFFT(Y_b)) = OLA(iFFT(sum_b Y_b)) and only the per-bin
    // gain -> mask step runs per band.  gain[q][k], q < n_gain, lists the non-zero band gains of bin k
    // in band order (zero padded); n_gain = 1 for an unmerged band.
    int n_gain;
    int gain_stride;
    int accumulate;        // 0: out = band, 1: out += band (band sum in list order)
    float* seam;           // [streams][3][(K-1) hop]: what a stream's last frames add to the NEXT stream's first blocks
    // Two waves share a SIMD and the hardware arbitrates their instruction issue by priority, then AGE: the workgroups of
    // the first half of a machine-filling launch (dispatched first, one per SIMD) ran at full speed and ended 20-25 %
    // before their younger partners, which then finished alone on their SIMDs (measured per workgroup:
    // scripts/phase_prof/wgtime.hip).  prio_split > 0: workgroups with index >= prio_split raise their priority on every
    // other frame pair, the others on the pairs in between: both halves advance at the same average rate and end together.
    // prio_split < 0 (two-wave workgroups, four per CU): -prio_split workgroups per dispatch round, the four rounds of a CU
    // ran at four speeds (233 / 283 / 283 / 298 us for the same work) and take the top priority in turn, a frame pair each.
    int prio_split;
    int prio_young;        // prio_split > 0: frame pairs out of 4 in which the younger half leads (3 balances; 2 = even turns)
#if defined(UPX_EXPERIMENTS)
    // EXPERIMENT builds only (-DUPX_EXPERIMENTS; rejected in round 5, docs/LOG.md): stream seams inside the launch
    // (upxk::seam_epilogue).  pair_cnt != nullptr: the workgroups w and w + 1 each add 1 to pair_cnt[w] when they are done;
    // whoever finds the count odd came SECOND and adds the tail of w's last stream onto the first blocks of w + 1's first
    // stream.  The product library has neither the fields nor the epilogue: its kernels take exactly the arguments above.
    int* pair_cnt;
    int n_streams;         // streams of the launch (stream_seam_add's guard)
#endif
};

constexpr float kEps = 1e-12f;   // center_extraction.py:36

// first frame of stream sid / frames of the streams of workgroup `wg` (G streams per workgroup)
UPX_HD int stream_first(const BandArgs& a, int sid) {
    if (a.stream_m0) {
#if defined(__HIP_DEVICE_COMPILE__)
        return ((const UPX_GLOBAL int*)a.stream_m0)[sid];
#else
        return a.stream_m0[sid];
#endif
    }
    return a.m_lo - 1 + sid * a.blocks_per_stream;
}
UPX_HD int stream_frames(const BandArgs& a, int sid) {
    return a.stream_m0 ? stream_first(a, sid + 1) - stream_first(a, sid) : a.blocks_per_stream;
}

// coherence * (1 - |balance|) mask of center_extraction.py:373-384 on one bin.
// |L conj(R)| is evaluated as |L||R| (identical in exact arithmetic).
UPX_HD void mask_bin(cf l, cf r, cf& c, cf& ls, cf& rs) {
    float ml = fast_sqrt(l.x * l.x + l.y * l.y);
    float mr = fast_sqrt(r.x * r.x + r.y * r.y);
    float p = ml * mr;
    float coh = p * fast_rcp(p + kEps);
    float bal = (ml - mr) * fast_rcp(ml + mr + kEps);
    float h = 0.5f * coh * (1.0f - __builtin_fabsf(bal));
    c = scale(l + r, h);
    ls = l - c;
    rs = r - c;
}

// The same mask as one weight for a launch that carries ONE band: with l = g2 l0, r = g2 r0 (g2 = gain / 2 >= 0)
//   C = w (l0 + r0),  w = g2 * 0.5 coh (1 - |bal|),   Ls + i Rs = 2 g2 Z[k] - (1 + i) C   (because l0 + i r0 = 2 Z[k]),
// so the two inverse-transform inputs of a bin pair need C only (band_program's mask()).  coh (1 - |bal|) =
// p (S - |D|) / ((p + eps) S) with p = |l||r|, S = |l| + |r| + eps, D = |l| - |r|: one reciprocal.  g2 = 0 gives w = 0.
UPX_HD float mask_weight(cf l0, cf r0, float g2) {
    const float ml = g2 * fast_sqrt(l0.x * l0.x + l0.y * l0.y);
    const float mr = g2 * fast_sqrt(r0.x * r0.x + r0.y * r0.y);
    const float p = ml * mr;
    const float sum = ml + mr + kEps;
    const float num = (0.5f * g2) * p * (sum - __builtin_fabsf(ml - mr));
    return num * fast_rcp((p + kEps) * sum);
}

// Per-thread state kept in registers across frames.
template <int P>
struct ThreadT {
    cf x[P];          // FFT working set
    cf acc_rl[P];     // overlap-add state of (Rs, Ls), slot s <-> sample lane + s*LANES of the current frame
    float acc_c[P];
    cf cs[P / 2];     // centre spectrum of the pair: C_a, then Yc[k]
    cf part[P / 2];   // Yc[N-k] of the pair
    cf pre[P];        // (first P/K used) raw samples of the NEXT frame's new hop, fetched one frame ahead
    float g0[P / 2];  // first gain slot of the own bins, fetched one phase ahead of the mask
    float g1[P / 2];  // second slot (merged bands)
    float gn[2];      // first two gain slots of the Nyquist bin
    float g0w[P];     // (upx_zoom.h) synthesis window of the pending last phase, fetched ahead
    int m0;           // first frame of this thread's stream (stream_first)
};

template <class C>
struct Stream {
    static constexpr int N = C::N, LANES = C::LANES, P = C::P;
    using PS = typename C::PS;
    using Thread = ThreadT<C::P>;

    // --- one Stockham pass ---------------------------------------------------
    // inputs of butterfly q are slots q + r*(P/R); pass PI multiplies input r by
    // W_(NS*R)^(r*k), k = j mod NS (j = lane + q LANES), then a radix-R DFT in place;
    // output r belongs at (j - k) R + k + r NS.  Twiddles come from the LDS table
    // tw[row(PI, r)][k] (compact: NS entries per row).
    // IM / OM: bit s = slot s may be non-zero on entry / is used on exit (pruned butterflies; full masks = plain pass)
    static constexpr unsigned FULL = (1u << P) - 1u;
    template <int PI, unsigned IM = FULL, unsigned OM = FULL>
    static UPX_HD void pass_compute(Thread& th, const cf* tw, int lane) {
        constexpr int R = PS::r[PI];
        constexpr int NS = pass_ns(PS::r, PI);
        constexpr int NB = P / R;
        constexpr int OFF = tw_offset(PS::r, PI);
        if constexpr ((IM & FULL) == FULL && (OM & FULL) == FULL) {
#pragma unroll
            for (int q = 0; q < NB; ++q) {
                cf v[R];
#pragma unroll
                for (int r = 0; r < R; ++r) v[r] = th.x[q + r * NB];
                if (NS > 1) {
                    const cf* row = tw + OFF + ((lane + q * LANES) & (NS - 1));
                    cf w[R];
                    w[0] = mk(1.f, 0.f);
#pragma unroll
                    for (int r = 1; r < R; ++r) w[r] = lds_load(row + (r - 1) * NS);
                    Dft<R>::run_tw(v, w);
                } else {
                    Dft<R>::run(v);
                }
#pragma unroll
                for (int r = 0; r < R; ++r) th.x[q + r * NB] = v[r];
            }
        } else {
            static_for<0, NB>([&](auto iq) {
                constexpr int q = decltype(iq)::value;
                constexpr unsigned im = gather_mask(IM, q, NB, R), om = gather_mask(OM, q, NB, R);
                if constexpr (om != 0) {
                    cf v[R] = {}, w[R] = {};
                    const cf* row = tw + OFF + ((lane + q * LANES) & (NS - 1));
                    static_for<1, R>([&](auto ir) {   // every twiddle read before the first multiply (read_compute)
                        constexpr int r = decltype(ir)::value;
                        if constexpr (NS > 1 && bit(im, r)) w[r] = lds_load(row + (r - 1) * NS);
                    });
                    static_for<0, R>([&](auto ir) {
                        constexpr int r = decltype(ir)::value;
                        if constexpr (bit(im, r)) {
                            v[r] = th.x[q + r * NB];
                            if constexpr (NS > 1 && r > 0) v[r] = cmul(v[r], w[r]);
                        }
                    });
                    Dft<R>::template run_p<im, om>(v);
                    static_for<0, R>([&](auto ir) {
                        constexpr int r = decltype(ir)::value;
                        if constexpr (bit(om, r)) th.x[q + r * NB] = v[r];
                    });
                }
            });
        }
    }
    // bits q + r NB (r < R) of a slot mask, as the mask of butterfly q
    static constexpr unsigned gather_mask(unsigned m, int q, int nb, int r_n) {
        unsigned o = 0;
        for (int r = 0; r < r_n; ++r) o |= bit(m, q + r * nb) << r;
        return o;
    }
    // scatter of a radix-P pass (one butterfly per lane): slot r -> (lane - k) P + k + r NS
    template <int PI>
    static UPX_HD void pass_write(const Thread& th, cf* lds, int lane) {
        constexpr int R = PS::r[PI];
        constexpr int NS = pass_ns(PS::r, PI);
        static_assert(R == P, "passes that go through LDS are radix-P (one butterfly per lane)");
        static_assert(NS == 1 || NS % P == 0, "padding algebra needs NS multiple of P");
        const int k = lane & (NS - 1);
        // NS == 1: P lane + r -> (P+1) lane + r ; else padp(base) + r (NS + NS/P)
        cf* b = lds + (NS == 1 ? lane * (R + 1) : padp<P>((lane - k) * R + k));
        constexpr int STEP = NS == 1 ? 1 : NS + NS / P;
#pragma unroll
        for (int r = 0; r < R; ++r) b[r * STEP] = th.x[r];
    }
    template <unsigned IM = FULL>
    static UPX_HD void read_all(Thread& th, const cf* lds, int lane) {
        const cf* b = lds + padp<P>(lane);
#pragma unroll
        for (int s = 0; s < P; ++s)
            if (bit(IM, s)) th.x[s] = lds_load(b + s * C::SPITCH);
    }
    // Register form of pass_write<n-2> + read_all for streams of ROWS = LANES/16 rows (Cfg::SWAP_LAST): output r = ROWS h + c
    // of the lane in row a belongs in slot ROWS^2... = (16/ROWS) a + h of the same column in row c - per h a ROWS x ROWS
    // transpose between the row index and c.  v_permlane32_swap exchanges the upper half of its first operand with the
    // lower half of the second, v_permlane16_swap the odd rows of the first with the even rows of the second.
    static UPX_HD void row_exchange(Thread& th) {
#if defined(__HIP_DEVICE_COMPILE__)
        constexpr int ROWS = LANES / 16, NH = P / ROWS;
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        cf y[P];
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            unsigned v[ROWS][2];
#pragma unroll
            for (int c = 0; c < ROWS; ++c) {
                // (through scalars: __builtin_bit_cast of a vector element `.y` reads element 0 with this hipcc)
                const float re = th.x[ROWS * h + c].x, im = th.x[ROWS * h + c].y;
                v[c][0] = __builtin_bit_cast(unsigned, re);
                v[c][1] = __builtin_bit_cast(unsigned, im);
            }
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                if constexpr (ROWS == 4) {
                    u2 a = __builtin_amdgcn_permlane32_swap(v[0][d], v[2][d], false, false);
                    u2 b = __builtin_amdgcn_permlane32_swap(v[1][d], v[3][d], false, false);
                    u2 e = __builtin_amdgcn_permlane16_swap(a.x, b.x, false, false);
                    u2 f = __builtin_amdgcn_permlane16_swap(a.y, b.y, false, false);
                    v[0][d] = e.x; v[1][d] = e.y; v[2][d] = f.x; v[3][d] = f.y;
                } else {
                    u2 e = __builtin_amdgcn_permlane16_swap(v[0][d], v[1][d], false, false);
                    v[0][d] = e.x; v[1][d] = e.y;
                }
            }
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
                y[NH * j + h] = mk(__builtin_bit_cast(float, v[j][0]), __builtin_bit_cast(float, v[j][1]));
        }
#pragma unroll
        for (int s = 0; s < P; ++s) th.x[s] = y[s];
#else
        (void)th;
#endif
    }
    // read_all + pass_compute<PI>.  For a radix-P pass all 2P-1 LDS reads (inputs and twiddles) are issued
    // before the first multiply, in the order the butterfly consumes them (its first radix-4 takes inputs 0, 4, 8,
    // 12): left to itself the scheduler splits them into three batches and waits for each to come back in
    // full - three exposed LDS round trips per pass instead of one.
    // (EAGER = false where registers are scarce: the tail phases hold windows, old samples and the next frame.)
    template <int PI, bool EAGER = true, unsigned IM = FULL, unsigned OM = FULL>
    static UPX_HD void read_compute(Thread& th, const cf* lds, const cf* tw, int lane) {
        constexpr int R = PS::r[PI];
        constexpr int NS = pass_ns(PS::r, PI);
        if constexpr ((IM & FULL) != FULL || (OM & FULL) != FULL) {
            read_all<IM>(th, lds, lane);
            pass_compute<PI, IM, OM>(th, tw, lane);
        } else if constexpr (R == P && P == 16 && EAGER) {
            constexpr int OFF = tw_offset(PS::r, PI);
            const cf* b = lds + padp<P>(lane);
            const cf* row = tw + OFF + (lane & (NS - 1));
            cf w[P];
#pragma unroll
            for (int i = 0; i < P; ++i) {
                const int r = (i & 3) * 4 + (i >> 2);   // 0 4 8 12 1 5 9 13 ...
                th.x[r] = lds_load(b + r * C::SPITCH);
                if (NS > 1 && r > 0) w[r] = lds_load(row + (r - 1) * NS);
            }
            UPX_SCHED_FENCE();
            if (NS > 1) {
                w[0] = mk(1.f, 0.f);
                Dft<P>::run_tw(th.x, w);
            } else {
                Dft<P>::run(th.x);
            }
        } else {
            read_all(th, lds, lane);
            pass_compute<PI>(th, tw, lane);
        }
    }

    // passes PI..n-2:  [read, transform] | [scatter] |   ('|' = barrier)
    // the last pass after mid_passes<1, SWAP>
    template <bool SWAP, bool EAGER = true, unsigned OM = FULL>
    static UPX_HD void last_compute(Thread& th, const cf* lds, const cf* tw, int lane) {
        if constexpr (SWAP) pass_compute<PS::n - 1, FULL, OM>(th, tw, lane);   // the inputs are in the registers already
        else read_compute<PS::n - 1, EAGER, FULL, OM>(th, lds, tw, lane);
    }
    // SWAP (only with C::SWAP_LAST): the exchange in front of the last pass stays in registers; the caller ends the
    // transform with last_compute<true>
    template <int PI, bool SWAP = false, class Ex>
    static UPX_HD void mid_passes(Ex& ex, cf* lds_all, const cf* tw) {
        static_assert(!SWAP || C::SWAP_LAST, "row exchange needs a stream of 2 or 4 rows");
        if constexpr (SWAP && PI == PS::n - 2) {
            ex.each([lds_all, tw](int tid, Thread& th) {
                read_compute<PI>(th, lds_all + (tid / LANES) * C::PITCH, tw, tid % LANES);
                row_exchange(th);
            });
        } else if constexpr (PI < PS::n - 1) {
            // `each2(f, g)`: g scatters into the buffer f has read.  Multi-wave streams need a barrier in
            // between; a stream inside ONE wave does not (LDS operations of a wave execute in order and every
            // scattered value depends on all 16 values read), so its executor runs f and g back to back.
            // (captured by value: an executor may run the phases after this function has returned)
            ex.each2(
                [lds_all, tw](int tid, Thread& th) { read_compute<PI>(th, lds_all + (tid / LANES) * C::PITCH, tw, tid % LANES); },
                [lds_all](int tid, Thread& th) { pass_write<PI>(th, lds_all + (tid / LANES) * C::PITCH, tid % LANES); });
            mid_passes<PI + 1, SWAP>(ex, lds_all, tw);
        }
    }
};

// ---------------------------------------------------------------------------
// The band stream program.  `ex.each(f)` = run f on every thread, then barrier
// (a wave-level fence when a stream fits in one wave).  One LDS buffer per
// stream: a phase that scatters into it is separated by a barrier from the phase
// that last read it.  The inter-pass twiddles live in LDS (copied once per
// workgroup), so the per-frame global traffic is the audio itself plus the
// window / gain vectors.
//
// Per frame pair (a, b) = (odd j, j+1), n = passes per FFT:
//   head(a)   load, window, pass 0                      | scatter |
//   mid       (read, pass p | scatter |) x (n-2)
//   zsplit    read, final pass | park Z[N/2..N) |
//   mask(a)   partners from the upper region, L/R split, gain, mask; Y mirror -> lower region |
//   inv0      read upper slots, pass 0 | scatter |  mid ...
//   tailLR(a) read, final pass, window, overlap-add, emit hop; then head(b) ...
//   tailLR(b), then stage the centre pair | mirror -> lower region | inv0 ... mid
//   tailC is merged into the next iteration's head(a).
// Wide streams (C::WIDE): head = load, window, radix-16 over n1, * W_N^(k1 n2), transposed write | B1 |
//   sub-FFT passes, mask and inverse sub-FFT passes with wave-level ordering only, * W_N^(k1 n2) | B2 |
//   tail = column read, radix-16 over k1, window, overlap-add;  | B3 | before the centre pair is staged.
// Loads are placed for latency, not where they are used: gains one phase before the mask, the frame's
// re-read samples at the top of the phase that ends the previous inverse, the new hop one frame ahead.
// ---------------------------------------------------------------------------
// MERGED = false: the launch carries ONE band (a.n_gain == 1, the host guarantees it): the second gain slot's
// registers and loads disappear, which is what keeps the N = 2048 / 4096 kernels free of spills.
//
// IN = true ("interior" workgroup, band_interior() below): every frame of its streams exists and lies inside the
// signal, every hop is emitted and lies inside the planes, and a.accumulate == ACC.  All but the first and the last few
// workgroups of a launch are interior.  Their loop body has no branch: every load and store is unconditional, so
// the backend counts outstanding vector-memory operations exactly (`s_waitcnt vmcnt(n)` leaves the younger prefetches
// and stores in flight); where paths with different numbers of loads or stores merge it has to wait with vmcnt(0).
//
// LV = Live<S0, S1> (single-band launches of plain streams): only the own-bin slots S0 <= s < S1 - bins lane + s LANES -
// carry gain anywhere in the launch (the host reads that off the gain vector when the plan is created; S1 >= P/2 also
// means "the Nyquist bin may carry gain").  The forward transform's last pass then produces only those slots and
// their mirror partners, the L/R split, mask, partner parking and mirror writes run over them alone, and the first
// pass of the inverse transforms knows the other inputs to be zero (pruned butterflies, Dft<>::run_p).  S0 <= 1.
template <int S0_, int S1_>
struct Live {
    static constexpr int S0 = S0_, S1 = S1_;
};
using LiveAll = Live<0, 1 << 20>;

template <class C, class Ex, bool MERGED = true, bool IN = false, bool ACC = false, class LV = LiveAll>
UPX_HD void band_program(Ex& ex, const BandArgs& a, cf* lds_all, int wg_index) {
    using SC = typename C::Sub;     // the FFT that goes through LDS: the frame itself, or a wide stream's sub-FFT
    using S = Stream<SC>;
    using PS = typename SC::PS;
    using Thread = ThreadT<C::P>;
    constexpr int N = C::N, LANES = C::LANES, P = C::P, HS = C::HS, HOP = C::HOP, K = C::K;
    constexpr int H = P / 2;        // slots holding own bins k < N/2
    constexpr int LAST = PS::n - 1; // final pass index
    constexpr int SP = C::SPITCH;
    constexpr int SL = SC::LANES;   // lanes per LDS buffer (= LANES unless wide)
    constexpr int BUF = SC::PITCH;  // complex per LDS buffer
    constexpr bool WIDE = C::WIDE;
    constexpr bool GAINS_EARLY = C::EARLY_LOADS && !MERGED;   // (the merged flavour also holds the second gain slot)
    constexpr bool SWAP = SC::SWAP_LAST;   // the exchange before the last pass stays in registers (device, 2- or 4-row streams)
    // live own-bin slots [S0, S1); slots LO .. HI-1 are the ones whose mirror cells the mask (re)writes: one dead slot on
    // either side, because lane 0's mirror of slot s lives in upper slot P - s and every other lane's in P - 1 - s
    constexpr int S0 = LV::S0 < H ? LV::S0 : H, S1 = LV::S1 < H ? LV::S1 : H;
    constexpr bool PRUNED = S0 > 0 || S1 < H;
    constexpr bool NYQ = S1 == H;                                  // the Nyquist bin may carry gain
    constexpr int LO = S0 > 0 ? S0 - 1 : 0, HI = S1 < H ? S1 + 1 : H;
    constexpr int ULO = P - S1, UHI = P - S0 < P - 1 ? P - S0 : P - 1;   // upper slots that may be non-zero: [ULO, UHI]
    constexpr unsigned FULLM = (1u << P) - 1u;
    constexpr unsigned OWN_M = ((1u << S1) - 1u) & ~((1u << S0) - 1u);
    constexpr unsigned UP_M = ((UHI >= P - 1 ? FULLM : ((1u << (UHI + 1)) - 1u)) & ~((1u << ULO) - 1u)) & FULLM;
    constexpr unsigned LIVE_M = PRUNED ? (OWN_M | UP_M) : FULLM;
    static_assert(!PRUNED || (!WIDE && !MERGED && S0 <= 1 && S0 < S1), "pruned flavours: plain streams, one band, S0 <= 1");
    auto own_live = [](int s) { return s >= S0 && s < S1; };

    // A stream transforms the F frames [m0, m0+F), m0 = m_lo - 1 + stream * F, and nothing else: no halo
    // frames are recomputed.  Frames come in pairs (a, b) = (odd j, j+1) so that one inverse FFT returns the
    // centre signal of both; with m_lo and F even every stream starts on an 'a' frame and the pairing is the
    // same for every partition.  Block m = sum of frames m-K+1..m: the K-1 blocks after m0+F-1 are only
    // partially known here (the "tail") and go to a.seam; upx_stream_seam_add() adds each tail onto the first
    // blocks of the next stream afterwards (same float32 association as the multi-GPU seam).
    const int F = stream_frames(a, wg_index * C::G);   // even, the same for the streams of a workgroup (host guarantees)
    const int n_iter = F / 2;
    int it = 0;   // the frame-pair counter of the main loop (declared here: the pieces below read it; it outlives the
                  // loop because the host emulator of wide streams runs recorded phases at the next barrier)

    cf* const tw = lds_all + C::G * C::PITCH;   // LDS twiddle table, after the stream buffers
    const cf* const bigtw = tw + SC::TW_CF;     // wide: W_N^(k1 n2), row k1 at k1 * BT_ROW

    // ---- copy the twiddle table into LDS ------------------------------------
    ex.each([&](int tid, Thread& th) {
        const UPX_GLOBAL cf* src = opaque(a.tw);
        for (int i = tid; i < C::TW_CF; i += C::WG) tw[i] = src[i];
        th.m0 = stream_first(a, wg_index * C::G + tid / LANES);
    });
    if constexpr (WIDE) ex.wg_barrier();

    // ---- pieces (per thread) ------------------------------------------------
    // Frame j of a stream covers samples j HOP + lane + s LANES.  Slots s < P-HS were part of the
    // previous frame (L2 hits); the HS "new hop" slots are HBM misses, so they are fetched one frame
    // ahead into th.pre and only consumed here.
    // (a thread's stream start lives in th.m0: set once, below)
    auto frame_of = [&](int m0, int it, int half, bool& exists) {
        const int j = m0 + 2 * it + half;
        exists = IN || (j >= a.j_lo && j < a.j_hi && j < m0 + F);
        return j;
    };
    // Interior frames - all but the last few of a signal - take a branch-free path: one base address per array
    // and immediate offsets per slot.  UPX_ALL makes the choice per wave, so the fast path has no per-slot
    // clamps, compares or exec masks (they were a third of the instructions of these phases).
    auto prefetch = [&](int tid, Thread& th, int it, int half) {
        bool exists;
        const int j = frame_of(th.m0, it, half, exists);
        const int e = exists ? j * HOP + tid % LANES : 0;
        const UPX_GLOBAL cf* in = opaque(a.in);
        const int last = a.t_in - 1;
        if (IN || UPX_ALL(e + (P - 1) * LANES <= last)) {
#pragma unroll
            for (int s = P - HS; s < P; ++s) th.pre[s - (P - HS)] = gat(in, (unsigned)e, s * LANES);
        } else {
#pragma unroll
            for (int s = P - HS; s < P; ++s) {
                const int n = e + s * LANES;
                th.pre[s - (P - HS)] = in[n < last ? n : last];
            }
        }
    };
    // The samples a frame shares with its predecessor (L2 hits) and the analysis window.  Issued at the TOP of the
    // phase that ends the previous inverse transform: behind that phase's stores the compiler could not hoist
    // them itself (it cannot tell the planes from the input), and a whole L2 round trip would be exposed.
    struct HeadRegs {
        cf v[P];      // first P-HS used
        float wa[P];  // analysis window, if have_wa
        bool fast;    // the whole wave loads a frame that lies inside the signal
        bool have_wa = false;
    };
    // The analysis window of the head that follows a tail in the same phase: issued between the tail's arithmetic and
    // its stores (registers are free there).  Loads return in issue order: behind the stores they would wait for the
    // stores' acknowledgements before the head can start.
    auto window_fetch = [&](int tid, HeadRegs& hr) {
        if constexpr (!C::EARLY_LOADS) return;
        const UPX_GLOBAL float* w_a = opaque(a.w_a);
#pragma unroll
        for (int s = 0; s < P; ++s) hr.wa[s] = gat(w_a, (unsigned)(tid % LANES), s * LANES);
        hr.have_wa = true;
    };
    auto head_fetch = [&](int tid, const Thread& th, int it, int half, HeadRegs& hr) {
        const int lane = tid % LANES;
        bool exists;
        const int j = frame_of(th.m0, it, half, exists);
        const int e = exists ? j * HOP + lane : 0;
        const UPX_GLOBAL cf* in = opaque(a.in);
        const int last = a.t_in - 1;   // host guarantees t_in >= 1
        hr.fast = IN || UPX_ALL(exists && e + (P - 1) * LANES <= last);
        constexpr int NFETCH = P - HS;
        if (hr.fast) {
#pragma unroll
            for (int s = 0; s < NFETCH; ++s) hr.v[s] = gat(in, (unsigned)e, s * LANES);
        } else {
#pragma unroll
            for (int s = 0; s < NFETCH; ++s) {
                // always load an in-range sample; head() zeroes what lies past the signal or in a frame this
                // stream does not own (zero-extension of center_extraction.py:437-455)
                const int n = e + s * LANES;
                hr.v[s] = in[n < last ? n : last];
            }
        }
        UPX_SCHED_FENCE();
    };
    auto head = [&](int tid, Thread& th, int it, int half, const HeadRegs& hr) {
        const int lane = tid % LANES;
        float wa[P];   // L1 / L2 hits
        if (hr.have_wa) {
#pragma unroll
            for (int s = 0; s < P; ++s) wa[s] = hr.wa[s];
        } else {
            const UPX_GLOBAL float* w_a = opaque(a.w_a);
#pragma unroll
            for (int s = 0; s < P; ++s) wa[s] = gat(w_a, (unsigned)lane, s * LANES);
        }
        // (the window folded into pass 0's first butterflies - Dft<16>::first_sc, as the band-limited analysis does - made
        // these kernels 0-2 % slower: not used here)
        if (hr.fast) {
#pragma unroll
            for (int s = 0; s < P; ++s) th.x[s] = scale(s < P - HS ? hr.v[s] : th.pre[s < P - HS ? 0 : s - (P - HS)], wa[s]);
        } else {
            bool exists;
            const int j = frame_of(th.m0, it, half, exists);
            const int e = exists ? j * HOP + lane : 0;
            const int last = a.t_in - 1;
#pragma unroll
            for (int s = 0; s < P; ++s) {
                const int n = e + s * LANES;
                const cf v = s < P - HS ? hr.v[s] : th.pre[s < P - HS ? 0 : s - (P - HS)];
                const float w = (exists && n <= last) ? wa[s] : 0.f;
                th.x[s] = scale(v, w);
            }
        }
        if constexpr (WIDE) {
            // radix-16 over n1 in registers, then W_N^(k1 n2) (n2 = tid)
            Dft<16>::run(th.x);
            const cf* bt = bigtw + tid;
#pragma unroll
            for (int r = 1; r < P; ++r) th.x[r] = cmul(th.x[r], lds_load(bt + r * C::BT_ROW));
        } else {
            S::template pass_compute<0>(th, tw, lane);
        }
        UPX_SCHED_FENCE();   // keep the prefetch behind the loads this frame waits for (vmcnt retires in order)
        if constexpr (GAINS_EARLY) {
            // first gain slot of the mask three phases on: in front of the prefetch (an HBM miss), not behind it
            const UPX_GLOBAL float* gain = opaque(a.gain);
#pragma unroll
            for (int s = 0; s < H; ++s)
                if (own_live(s)) th.g0[s] = gat(gain, (unsigned)lane, s * LANES);
            if constexpr (NYQ) th.gn[0] = gain[N / 2];
        }
        prefetch(tid, th, it + (half == 1 ? 1 : 0), half == 1 ? 0 : 1);
    };
    // last step of an inverse transform: time samples lane + s LANES land in slot s
    auto final_pass = [&](int tid, Thread& th) {
        if constexpr (WIDE) {
            // column n2 = tid of the sixteen sub-FFT buffers, radix-16 over k1
            const cf* b = lds_all + padp<P>(tid);
#pragma unroll
            for (int r = 0; r < P; ++r) th.x[r] = lds_load(b + wide_sub_of_k1(r) * BUF);
            Dft<16>::run(th.x);
        } else {
            S::template last_compute<SWAP, false>(th, lds_all + (tid / SL) * BUF, tw, tid % SL);
        }
    };
    // first exchange of a forward transform
    auto head_write = [&](int tid, Thread& th) {
        if constexpr (WIDE) {
            // transposed: value k1 of lane n2 goes to element n2 of sub-FFT g(k1) - the same cells final_pass reads
            cf* b = lds_all + padp<P>(tid);
#pragma unroll
            for (int r = 0; r < P; ++r) b[wide_sub_of_k1(r) * BUF] = th.x[r];
        } else {
            S::template pass_write<0>(th, lds_all + (tid / SL) * BUF, tid % SL);
        }
    };
    // One emitted hop of a plane: the old values (band sum in list order) are fetched BEFORE the final pass so
    // that their latency overlaps the butterflies; so is the synthesis window.
    struct Hop {
        int e;        // first sample of this lane's part of the hop (0 if nothing is emitted)
        bool emit;    // this stream emits the hop
        bool fast;    // the whole wave emits a hop that lies inside the planes
    };
    auto hop_of = [&](int m0, int tid, int j) {
        Hop h;
        h.emit = IN || (j >= a.m_lo && j < a.m_hi && j < m0 + F);
        h.e = h.emit ? j * HOP + tid % LANES : 0;
        h.fast = IN || UPX_ALL(h.emit && h.e + (HS - 1) * LANES <= a.t_out - 1);
        return h;
    };
    auto load_old = [&](UPX_GLOBAL float* plane, const Hop& h, float* old) {
        if (!(IN ? ACC : a.accumulate != 0)) {
#pragma unroll
            for (int s = 0; s < HS; ++s) old[s] = 0.f;
        } else if (h.fast) {
#pragma unroll
            for (int s = 0; s < HS; ++s) old[s] = load_nt(gat(plane, (unsigned)h.e, s * LANES));
        } else {
            const int last = a.t_out - 1;
#pragma unroll
            for (int s = 0; s < HS; ++s) {
                const int n = h.e + s * LANES;
                old[s] = plane[n < last ? n : last];
            }
        }
    };
    auto no_hook = []() {};
    auto tail_lr = [&](int tid, Thread& th, int it, int half, auto&& before_stores) {
        const int lane = tid % LANES;
        const int m0 = th.m0;
        const Hop h = hop_of(m0, tid, m0 + 2 * it + half);
        UPX_GLOBAL float* out_l = opaque(a.out_l);
        UPX_GLOBAL float* out_r = opaque(a.out_r);
        float old_l[HS], old_r[HS];
        load_old(out_l, h, old_l);
        load_old(out_r, h, old_r);
        const UPX_GLOBAL float* w_s = opaque(a.w_s);
        float w[P];
#pragma unroll
        for (int s = 0; s < P; ++s) w[s] = gat(w_s, (unsigned)lane, s * LANES);
        final_pass(tid, th);
#pragma unroll
        for (int s = 0; s < P; ++s)
            th.acc_rl[s] = th.acc_rl[s] + scale(th.x[s], w[s]);   // swapped output: Ls = Re y = x.y, Rs = Im y = x.x
        UPX_SCHED_FENCE();
        before_stores();
        UPX_SCHED_FENCE();
        if (h.fast) {
#pragma unroll
            for (int s = 0; s < HS; ++s) {
                store_nt(gat(out_l, (unsigned)h.e, s * LANES), old_l[s] + th.acc_rl[s].y);
                store_nt(gat(out_r, (unsigned)h.e, s * LANES), old_r[s] + th.acc_rl[s].x);
            }
        } else {
            const int last = a.t_out - 1;
#pragma unroll
            for (int s = 0; s < HS; ++s) {
                const int n = h.e + s * LANES;
                if (h.emit && n <= last) {
                    out_l[n] = old_l[s] + th.acc_rl[s].y;
                    out_r[n] = old_r[s] + th.acc_rl[s].x;
                }
            }
        }
#pragma unroll
        for (int s = 0; s < P; ++s) th.acc_rl[s] = s + HS < P ? th.acc_rl[s + HS] : mk(0.f, 0.f);
    };
    auto tail_c = [&](int tid, Thread& th, int it, auto&& before_stores) {
        const int lane = tid % LANES;
        const int m0 = th.m0;
        UPX_GLOBAL float* out_c = opaque(a.out_c);
        Hop h[2];
        float old_c[2][HS];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            h[half] = hop_of(m0, tid, m0 + 2 * it + half);
            load_old(out_c, h[half], old_c[half]);
        }
        const UPX_GLOBAL float* w_s = opaque(a.w_s);
        float w[P];
#pragma unroll
        for (int s = 0; s < P; ++s) w[s] = gat(w_s, (unsigned)lane, s * LANES);
        final_pass(tid, th);
        float emit_c[2][HS];   // old + new of the two hops; stored after the hook
#pragma unroll
        for (int half = 0; half < 2; ++half) {
#pragma unroll
            for (int s = 0; s < P; ++s)   // swapped output: c_a = Re y = x.y ; c_b = Im y = x.x
                th.acc_c[s] += (half == 0 ? th.x[s].y : th.x[s].x) * w[s];
#pragma unroll
            for (int s = 0; s < HS; ++s) emit_c[half][s] = old_c[half][s] + th.acc_c[s];
#pragma unroll
            for (int s = 0; s < P; ++s) th.acc_c[s] = s + HS < P ? th.acc_c[s + HS] : 0.f;
        }
        UPX_SCHED_FENCE();
        before_stores();
        UPX_SCHED_FENCE();
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (h[half].fast) {
#pragma unroll
                for (int s = 0; s < HS; ++s) store_nt(gat(out_c, (unsigned)h[half].e, s * LANES), emit_c[half][s]);
            } else {
                const int last = a.t_out - 1;
#pragma unroll
                for (int s = 0; s < HS; ++s) {
                    const int n = h[half].e + s * LANES;
                    if (h[half].emit && n <= last) out_c[n] = emit_c[half][s];
                }
            }
        }
    };
    // Where the mirror bins live.  Own bins k (slots s < H) stay in registers; the partner Z[N-k_s] is read
    // from `zpart + (H-1-s) SP` and the mirrored output Y[N-k_s] is written to `ymir + (H-1-s) SP`, where the
    // first inverse pass picks it up as an upper slot.
    //   plain: partners are parked in the upper region [N/2, N); mirrors go to the (free) lower region at
    //          position (N-k) - N/2, position 0 = Nyquist (lane 0).
    //   wide:  bin k = k1 + 16 k2 of sub-FFT g pairs with k2' = N2 - (k1 != 0) - k2 of sub-FFT g^1 (g itself for
    //          k1 = 0, 8): an upper slot of a lane in the same wave, rewritten in place.
    struct Mirror {
        cf* zpart;
        cf* ymir;
        cf* nyq;      // where the Nyquist bin of the inverse input goes
        bool first;   // this lane holds DC (slot 0) and Nyquist (slot H)
    };
    auto mirror_of = [&](int tid) {
        Mirror m;
        const int sl = tid % SL;
        if constexpr (WIDE) {
            const int g = tid / SL;
            const int gp = (g >> 1) ? (g ^ 1) : g;
            const int z = wide_k1_of_sub(g) ? 1 : 0;
            m.zpart = m.ymir = lds_all + gp * BUF + padp<P>((H + 1) * SL - z - sl);
            m.nyq = lds_all + H * SP;
            m.first = tid == 0;
        } else {
            cf* lds = lds_all + (tid / SL) * BUF;
            m.zpart = lds + padp<P>(N - sl - (H - 1) * LANES);
            m.ymir = lds + padp<P>(N / 2 - sl - (H - 1) * LANES);
            m.nyq = lds;
            m.first = sl == 0;
        }
        return m;
    };
    // split L/R, gain, mask, build the iFFT input
    auto mask = [&](int tid, Thread& th, int half) {
        const int lane = tid % LANES;
        const Mirror mir = mirror_of(tid);
        const UPX_GLOBAL float* gain = opaque(a.gain);
        cf nyq_y = mk(0.f, 0.f);
        float nyq_c = 0.f;
        const int n_gain = MERGED ? a.n_gain : 1, gstride = a.gain_stride;
        if (NYQ && mir.first) {
            // Nyquist bin is real: L = Re Z[N/2], R = Im Z[N/2]
            const cf z = th.x[H];
            cf cn = mk(0.f, 0.f), lsn = cn, rsn = cn;
            for (int q = 0; q < n_gain; ++q) {
                const float g2 = q < 2 ? th.gn[q] : gain[q * gstride + N / 2];
                if (g2 != 0.f) {
                    cf l = mk(g2 * (z.x + z.x), 0.f), r = mk(g2 * (z.y + z.y), 0.f), c, ls, rs;
                    mask_bin(l, r, c, ls, rs);
                    cn = cn + c; lsn = lsn + ls; rsn = rsn + rs;
                }
            }
            nyq_y = mk(lsn.x, rsn.x);
            nyq_c = cn.x;
        }
        // all partners first: one LDS round trip instead of one per bin (the mirror writes below may alias them
        // as far as the compiler can tell, so it would not hoist the reads itself)
        cf zpart[H];
#pragma unroll
        for (int s = 0; s < H; ++s)
            if (own_live(s)) zpart[s] = lds_load(mir.zpart + (H - 1 - s) * SP);   // k == 0: spare row, unused
#pragma unroll
        for (int s = 0; s < H; ++s) {
            if (s < LO || s >= HI) continue;       // (pruned flavours) nothing of this slot is read again
            const bool live = own_live(s);         // a dead slot next to the live ones: zeros into its mirror cell
            const bool dc = s == 0 && mir.first;   // k == 0
            cf c = mk(0.f, 0.f), ls = c, rs = c;
            cf yk = c, ym = c;                     // swap(Y[k]), swap(Y[N-k]): the inverse transform's input (iFFT by swap)
            if (live) {
            const cf za = th.x[s];
            const cf zb = dc ? za : zpart[s];        // DC pairs with itself
            const cf l0 = add_conj(za, zb);      // Z[k] + conj Z[N-k]        (x gain/2 = L)
            const cf r0 = mi_sub_conj(za, zb);   // (Z[k] - conj Z[N-k]) / i  (x gain/2 = R)
            if constexpr (!MERGED && UPX_MASK_ALGEBRA) {
                // one band: Y[k] = G Z[k] - (1 + i) C, Y[N-k] = G Z[N-k] - (1 + i) conj C, G = 2 g2 (mask_weight);
                // no branch on the gain: a zero gain gives exact zeros, and the bins' latency chains interleave
                const float g2 = th.g0[s];
                c = scale(l0 + r0, mask_weight(l0, r0, g2));
                const cf u = sub_mi(c, c);           // (1 + i) C; its swap is (1 + i) conj C
                const cf gg = mk(g2 + g2, g2 + g2);
                yk = fma_swap_sub(za, gg, u);
                ym = fma_swapz_sub(zb, gg, u);
            } else {
            auto add_band = [&](float g2) {
                if (g2 != 0.f) {   // whole waves lie outside the band: the branch skips them
                    cf l = scale(l0, g2), r = scale(r0, g2), cq, lq, rq;
                    mask_bin(l, r, cq, lq, rq);
                    c = c + cq; ls = ls + lq; rs = rs + rq;
                }
            };
            // Gain slot 0 was fetched during the previous phase, slot 1 at the top of this one: a load issued
            // here exposes one L2 round trip per bin, eight in a row.  Further slots (three or more merged
            // bands overlapping in one bin) are rare and take that path.
            // plain: bin lane + s LANES.  wide: the table is stored in the kernel's bin order (gain_bin)
            add_band(th.g0[s]);
            if constexpr (MERGED) {
                if (n_gain > 1) {
                    add_band(th.g1[s]);
                    for (int q = 2; q < n_gain; ++q) add_band(gat(gain + q * gstride, (unsigned)lane, s * LANES));
                }
            }
            // Y[k] = Ls + i Rs, Y[N-k] = conj(Ls) + i conj(Rs); kept re/im swapped (iFFT by swap)
            yk = swap_add_i(ls, rs);
            ym = swap_conj_add_i(ls, rs);
            }
            }
            if (live) th.x[s] = yk;
            if (s == 0) {
                cf* dst = mir.first ? mir.nyq : mir.ymir + (H - 1) * SP;
                *dst = mir.first ? cswap(nyq_y) : ym;
            } else {
                mir.ymir[(H - 1 - s) * SP] = ym;
            }
            // centre spectrum; the first lane's slot 0 packs the two real bins (DC, Nyquist)
            if (!live && s > 0) continue;          // (a dead slot 0 still carries the Nyquist bin of the centre)
            const cf cv = dc ? mk(c.x, nyq_c) : c;
            if (half == 0) {
                th.cs[s] = cv;
            } else {
                const cf ca = th.cs[s], cb = cv;
                // Yc[k] = Ca + i Cb ; Yc[N-k] = conj(Ca) + i conj(Cb), kept swapped
                cf ck = swap_add_i(ca, cb);
                cf cm = swap_conj_add_i(ca, cb);
                if (dc) {
                    ck = mk(cb.x, ca.x);   // swap(Yc[0])
                    cm = mk(cb.y, ca.y);   // swap(Yc[N/2])
                }
                th.cs[s] = ck;
                th.part[s] = cm;
            }
        }
    };
    // first inverse pass: slots >= H hold the mirrored bins (plain: parked at position idx - N/2; wide: in place)
    auto inv0 = [&](int tid, Thread& th) {
        const int sl = tid % SL;
        const cf* b = lds_all + (tid / SL) * BUF + padp<P>(sl);
#pragma unroll
        for (int s = H; s < P; ++s)
            if (bit(LIVE_M, s)) th.x[s] = lds_load(b + (WIDE ? s : s - H) * SP);
        S::template pass_compute<0, LIVE_M>(th, tw, sl);
    };
    auto scatter0 = [&](int tid, Thread& th) { S::template pass_write<0>(th, lds_all + (tid / SL) * BUF, tid % SL); };
    auto mids = [&]() { S::template mid_passes<1, SWAP>(ex, lds_all, tw); };
    auto zsplit_compute = [&](int tid, Thread& th) {
        const UPX_GLOBAL float* gain = opaque(a.gain);   // (slot 0: head())
        if constexpr (!GAINS_EARLY) {
#pragma unroll
            for (int s = 0; s < H; ++s)
                if (own_live(s)) th.g0[s] = gat(gain, (unsigned)(tid % LANES), s * LANES);
            if constexpr (NYQ) th.gn[0] = gain[N / 2];
        }
        if constexpr (MERGED) {
            th.gn[1] = a.n_gain > 1 ? gain[a.gain_stride + N / 2] : 0.f;
            // (always written: a value kept from the previous frame would be live through the whole loop)
#pragma unroll
            for (int s = 0; s < H; ++s) th.g1[s] = 0.f;
            if (a.n_gain > 1) {
#pragma unroll
                for (int s = 0; s < H; ++s) th.g1[s] = gat(gain + a.gain_stride, (unsigned)(tid % LANES), s * LANES);
            }
        }
        S::template last_compute<SWAP, true, LIVE_M>(th, lds_all + (tid / SL) * BUF, tw, tid % SL);
    };
    auto zsplit_write = [&](int tid, Thread& th) {
        cf* b = lds_all + (tid / SL) * BUF + padp<P>(tid % SL);
#pragma unroll
        for (int s = H; s < P; ++s)
            if (bit(LIVE_M, s)) b[s * SP] = th.x[s];
    };
    auto stage_c = [&](int tid, Thread& th) {
        const Mirror mir = mirror_of(tid);
#pragma unroll
        for (int s = 0; s < H; ++s) {
            if (s < LO || s >= HI) continue;
            const bool live = own_live(s);
            if (live) th.x[s] = th.cs[s];
            if (s == 0) {
                cf* dst = mir.first ? mir.nyq : mir.ymir + (H - 1) * SP;
                *dst = th.part[0];
            } else {
                mir.ymir[(H - 1 - s) * SP] = live ? th.part[s] : mk(0.f, 0.f);
            }
        }
    };
    // wide only: first pass of a sub-FFT, and the end of an inverse sub-FFT (last pass, * W_N^(k1 n2), back
    // into the own buffer, from where final_pass reads columns after the barrier)
    auto sub_first = [&](int tid, Thread& th) {
        S::template read_compute<0>(th, lds_all + (tid / SL) * BUF, tw, tid % SL);
    };
    auto sub_last_inv = [&](int tid, Thread& th) {
        if constexpr (WIDE) {
            const int g = tid / SL, sl = tid % SL;
            cf* b = lds_all + g * BUF + padp<P>(sl);
            const cf* bt = bigtw + wide_k1_of_sub(g) * C::BT_ROW + sl;   // row 0 is all ones
            cf bw[P];
#pragma unroll
            for (int s = 0; s < P; ++s) bw[s] = lds_load(bt + s * SL);
            S::template last_compute<SWAP>(th, lds_all + g * BUF, tw, sl);
#pragma unroll
            for (int s = 0; s < P; ++s) b[s * SP] = cmul(th.x[s], bw[s]);
        }
    };
    // inverse transform from the staged input (own slots in registers, mirrors in LDS) up to the final pass
    auto inverse_body = [&]() {
        ex.each2(inv0, scatter0);
        mids();
        if constexpr (WIDE) {
            ex.each(sub_last_inv);
            ex.wg_barrier();   // columns complete: final_pass may read across waves
        }
    };
    // forward transform from the first exchange to the iFFT input, then the inverse passes
    auto frame_body = [&](int half) {
        if constexpr (WIDE) {
            ex.wg_barrier();   // transposed writes of head_write complete
            ex.each2(sub_first, scatter0);
        }
        mids();
        ex.each2(zsplit_compute, zsplit_write);
        ex.each([&](int tid, Thread& th) { mask(tid, th, half); });
        inverse_body();
    };

    ex.each([&](int tid, Thread& th) {
#pragma unroll
        for (int s = 0; s < P; ++s) {
            th.acc_rl[s] = mk(0.f, 0.f);
            th.acc_c[s] = 0.f;
        }
        prefetch(tid, th, 0, 0);
    });

    // (see BandArgs::prio_split; a scalar instruction per frame pair, nothing on the host emulator)
    auto issue_priority = [&](int pair) {
#if defined(__HIP_DEVICE_COMPILE__)
        if (a.prio_split < 0) {
            // multi-wave workgroups, -prio_split of them per dispatch round (one per CU): the four rounds that share a CU
            // take the top priority in turn, a frame pair each
            const int r = (wg_index / (-a.prio_split) + pair) & 3;
            if (r == 0) __builtin_amdgcn_s_setprio(3);
            else if (r == 1) __builtin_amdgcn_s_setprio(2);
            else if (r == 2) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        } else if (a.prio_split > 0) {
            // of every 4 frame pairs the younger half leads in prio_young, the older half in the others
            const bool young = wg_index >= a.prio_split;
            const bool lead = ((pair & 3) < a.prio_young) == young;
            if (lead) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
#else
        (void)pair;
#endif
    };

    // One extra trip runs only the centre tail of the last pair.  tail_c is instantiated ONCE
    // (inside this loop) on purpose: two inlined copies may contract multiply-adds differently,
    // and which copy a frame meets would then depend on how the signal is cut into streams.
    // Wide streams: `each` orders one wave only; the cross-wave phases (final_pass reads, head_write writes)
    // touch lane-private cells, so they need barriers only against the wave-local phases around them.
    if constexpr (IN) {
        // the same phases, rotated so that the body is straight-line code: the centre tail of pair `it` runs with the
        // head of pair it+1 at the END of trip `it`; the last trip heads a frame of the next stream (inside the signal
        // by the definition of interior) and nothing uses it.  (`it` by value: an executor may run a phase after ++it)
        ex.each2(
            [&, it](int tid, Thread& th) {
                HeadRegs hr;
                head_fetch(tid, th, 0, 0, hr);
                head(tid, th, 0, 0, hr);
            },
            head_write);
        for (; it < n_iter; ++it) {
            issue_priority(it);
            frame_body(0);
            ex.each2(
                [&, it](int tid, Thread& th) {
                    HeadRegs hr;
                    head_fetch(tid, th, it, 1, hr);
                    tail_lr(tid, th, it, 0, [&]() { window_fetch(tid, hr); });
                    head(tid, th, it, 1, hr);
                },
                head_write);
            frame_body(1);
            if constexpr (WIDE) {
                ex.each([&, it](int tid, Thread& th) { tail_lr(tid, th, it, 1, no_hook); });
                ex.wg_barrier();
                ex.each(stage_c);
            } else {
                ex.each2([&, it](int tid, Thread& th) { tail_lr(tid, th, it, 1, no_hook); }, stage_c);
            }
            inverse_body();
            ex.each2(
                [&, it](int tid, Thread& th) {
                    HeadRegs hr;
                    head_fetch(tid, th, it + 1, 0, hr);
                    tail_c(tid, th, it, [&]() { window_fetch(tid, hr); });
                    head(tid, th, it + 1, 0, hr);
                },
                head_write);
        }
    } else
    for (; it <= n_iter; ++it) {
        issue_priority(it);
        ex.each2(
            [&](int tid, Thread& th) {
                HeadRegs hr;
                if (it < n_iter) head_fetch(tid, th, it, 0, hr);
                if (it > 0) tail_c(tid, th, it - 1, no_hook);
                if (it < n_iter) head(tid, th, it, 0, hr);
            },
            [&](int tid, Thread& th) {
                if (it < n_iter) head_write(tid, th);
            });
        if (it == n_iter) break;
        frame_body(0);
        ex.each2(
            [&](int tid, Thread& th) {
                HeadRegs hr;
                head_fetch(tid, th, it, 1, hr);
                tail_lr(tid, th, it, 0, [&]() { window_fetch(tid, hr); });
                head(tid, th, it, 1, hr);
            },
            head_write);
        frame_body(1);
        if constexpr (WIDE) {
            ex.each([&](int tid, Thread& th) { tail_lr(tid, th, it, 1, no_hook); });
            ex.wg_barrier();   // every wave has read its columns before the centre pair is staged over them
            ex.each(stage_c);
        } else {
            ex.each2([&](int tid, Thread& th) { tail_lr(tid, th, it, 1, no_hook); }, stage_c);
        }
        inverse_body();
    }
    // what is left in the accumulators belongs to the K-1 blocks after this stream
    ex.each([&](int tid, Thread& th) {
        const int lane = tid % LANES;
        const int sid = wg_index * C::G + tid / LANES;
        UPX_GLOBAL float* seam = opaque(a.seam) + (size_t)sid * 3 * (P - HS) * LANES + lane;
#pragma unroll
        for (int s = 0; s < P - HS; ++s) {
            seam[(0 * (P - HS) + s) * LANES] = th.acc_c[s];
            seam[(1 * (P - HS) + s) * LANES] = th.acc_rl[s].y;
            seam[(2 * (P - HS) + s) * LANES] = th.acc_rl[s].x;
        }
    });
    if constexpr (WIDE) ex.wg_barrier();
}

// Workgroup `wg_index` is interior (band_program<.., IN = true>): its streams' frames m0 .. m0+F-1 all exist, are emitted
// and lie inside the signal and the planes, and so do the two frames after them (the rotated loop heads and prefetches
// them without using them).
template <class C>
UPX_HD bool band_interior(const BandArgs& a, int wg_index) {
    const long long F = stream_frames(a, wg_index * C::G);
    const long long first = stream_first(a, wg_index * C::G);                         // first frame of the first stream
    const long long end = first + C::G * F;                                           // one past the last stream's frames
    const long long lo = a.j_lo > a.m_lo ? a.j_lo : a.m_lo;
    const long long hi = a.j_hi < a.m_hi ? a.j_hi : a.m_hi;
    return first >= (lo > 0 ? lo : 0) && end <= hi && end * C::HOP <= a.t_out && (end + 1) * C::HOP + C::N <= a.t_in;
}
// band_program with the interior flavour where it applies (the choice is uniform over the workgroup)
template <class C, class Ex, bool MERGED = true, class LV = LiveAll>
UPX_HD void band_program_auto(Ex& ex, const BandArgs& a, cf* lds_all, int wg_index) {
    if (band_interior<C>(a, wg_index)) {
        if (a.accumulate) band_program<C, Ex, MERGED, true, true, LV>(ex, a, lds_all, wg_index);
        else band_program<C, Ex, MERGED, true, false, LV>(ex, a, lds_all, wg_index);
    } else {
        band_program<C, Ex, MERGED, false, false, LV>(ex, a, lds_all, wg_index);
    }
}

// Adds the tail of stream `sid` onto the first blocks of stream sid+1; one call per (sid, i), i < (K-1) hop.
// (host / a trivial kernel; see BandArgs::seam)
UPX_HD void stream_seam_add(const BandArgs& a, int n_streams, int tail, int hop, long long gid) {
    const int sid = (int)(gid / tail), i = (int)(gid % tail);
    if (sid >= n_streams - 1) return;                       // the last stream's tail lies beyond the emitted range
    const long long m = stream_first(a, sid + 1);
    const long long n = m * hop + i;
    if (m >= a.m_hi || n < 0 || n >= a.t_out) return;
    if (m + i / hop >= a.m_hi) return;                      // blocks past the emitted range stay untouched
    const float* row = a.seam + (size_t)sid * 3 * tail;
    a.out_c[n] += row[i];
    a.out_l[n] += row[tail + i];
    a.out_r[n] += row[2 * tail + i];
}

// Host-side helper: fill the compact twiddle table for Cfg (double precision -> float).
// Row (pass p, input r >= 1) holds W_(NS*R)^(r*k) for k = 0..NS-1 at tw_offset(p) + (r-1)*NS + k.
template <class C, class TrigFn>
inline void fill_twiddles(cf* tw, TrigFn trig) {
    using PS = typename C::PS;
    for (int p = 1; p < PS::n; ++p) {
        const int R = PS::r[p], NS = pass_ns(PS::r, p), off = tw_offset(PS::r, p);
        for (int r = 1; r < R; ++r)
            for (int k = 0; k < NS; ++k) {
                double c, s;
                trig((double)r * (double)k / ((double)NS * (double)R), c, s);   // fraction of a turn
                tw[off + (r - 1) * NS + k] = mk((float)c, (float)-s);
            }
    }
}

// Twiddle table of a kernel configuration: the (sub-)FFT passes, then for wide streams W_N^(k1 n2).
template <class C, class TrigFn>
inline void fill_tables(cf* tw, TrigFn trig) {
    fill_twiddles<typename C::Sub>(tw, trig);
    if constexpr (C::WIDE) {
        cf* bt = tw + C::Sub::TW_CF;
        for (int k1 = 0; k1 < 16; ++k1)
            for (int n2 = 0; n2 < C::BT_ROW; ++n2) {
                double c, s;
                trig((double)k1 * (double)n2 / (double)C::N, c, s);
                bt[k1 * C::BT_ROW + n2] = mk((float)c, (float)-s);
            }
    }
}

// Order of the per-bin gain rows as the kernel reads them: entry i = s * LANES + tid (s < P/2) is the bin that
// slot s of thread tid owns, entry N/2 the Nyquist bin.  Plain streams: the natural order.
template <class C>
inline int gain_bin(int i) {
    if constexpr (C::WIDE) {
        if (i >= C::N / 2) return i;
        const int s = i / C::LANES, tid = i % C::LANES;
        const int g = tid / C::Sub::LANES, sl = tid % C::Sub::LANES;
        return wide_k1_of_sub(g) + 16 * (sl + C::Sub::LANES * s);
    } else {
        return i;
    }
}

}   // namespace upx
