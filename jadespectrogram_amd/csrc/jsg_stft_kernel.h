// The STFT -> dB kernel of libjsg.so and everything that instantiates it (lane tables, launch helpers).  Included by the
// translation units that hold the kernels: jsg_stft_a.hip (512 / 1024 / 2048 / 8192 points) and jsg_stft_b.hip (4096 points) --
// two units because the two groups want different instruction schedulers (see jadespectrogram_amd/_build.py) -- and by
// jsg_kernels.hip, which owns the plan, the launcher and the C-ABI and calls the per-plan entry points declared at the end.
//
//   stft_db_kernel   L lanes transform one real frame (64 = one wavefront per frame; 32: two frames per wave at
//                    N = 512 and in the two-stage N = 2048 plan; 128 / 256: two / four wavefronts per frame at
//                    N = 4096 / 8192):
//                    lane tables (window, twiddles) staged once per workgroup into LDS, issued ahead of the frame
//                    loads so that the one workgroup barrier completes while those are in flight; coalesced 8-byte
//                    loads of the frame straight from the audio stream in HBM, software-prefetched (the 50..87.5 %
//                    overlap of neighbouring frames is served by L1/L2: neighbouring frames sit in one workgroup);
//                    window multiply; N/2-point complex FFT as three register-resident radix stages with two
//                    bank-conflict-free LDS exchanges (wave-private and barrier-free for L <= 64), or two radix-32
//                    stages with one exchange (Cfg2048B); paired real-split post pass (X[k] and X[N/2-k] from one
//                    butterfly, both |X|^2 formed side by side in packed math); channel mix in registers (pair
//                    accumulators); 10*log10 on the hardware log unit; non-temporal 256-byte coalesced ring stores.
//                    No MFMA: the path is bandwidth / latency / VALU-issue bound, not a contraction.
//                    Replaces Spectrogram.cpp:50-119 + :137-145 + spectrum::power (call site :144) of the reference.
//   colormap_kernel  dB ring columns -> ARGB image rows (transpose through LDS so both sides are coalesced),
//                    CColorPalette::getRGBColor inlined.  Replaces Spectrogram.cpp:632-648 / :673-680 / :693-700.
//
// The index algebra, the twiddle tables and the LDS layouts are modelled and checked in tools/fft_model.py.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/jsg.h"
#include "jsg_internal.h"
#include "jsg_exact_math.h"

namespace jsg {

// ------------------------------------------------------------------------------------------------------------
// complex arithmetic on packed pairs: cf = (re, im) in one aligned 64-bit VGPR pair.
//
// A wave64 VALU instruction occupies its SIMD for ~4 cycles unless four or more waves of that SIMD have VALU work
// ready at the same time (measured: tools/probes/valu_rate.hip; this kernel sits at ~4.1 cycles per instruction, PMC
// SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU), and a packed v_pk_{add,mul,fma}_f32 performs two float operations in such a
// slot.  A complex add is one packed add, a complex multiply is a packed multiply plus a packed fma, and a
// multiplication by -i costs nothing: it is folded into the operand-select / negate modifiers of the consuming
// instruction.  The compiler only derives the broadcast and whole-vector-negate forms of those modifiers from vector
// code, so the swizzled forms are written out as inline assembly.
// ------------------------------------------------------------------------------------------------------------
typedef float cf __attribute__((ext_vector_type(2)));

// a + (-i) b = (a.x + b.y, a.y - b.x)
__device__ __forceinline__ cf add_mi(cf a, cf b) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a - (-i) b = (a.x - b.y, a.y + b.x)
__device__ __forceinline__ cf sub_mi(cf a, cf b) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a + conj(b) = (a.x + b.x, a.y - b.y)   and   a - conj(b) = (a.x - b.x, a.y + b.y)
__device__ __forceinline__ cf add_conj(cf a, cf b) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ cf sub_conj(cf a, cf b) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// a * w (complex): t = (a.x w.x, a.y w.x);  r = (t.x - a.y w.y, t.y + a.x w.y)
__device__ __forceinline__ cf cmul(cf a, cf w) {
    cf r;   // one asm statement: between two of them the hazard recognizer pads an s_nop
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]"
        : "=&v"(r) : "v"(a), "v"(w));
    return r;
}

// (a.x + b.x, a.x - b.x) and (a.y + b.y, a.y - b.y): the real parts, and the imaginary parts, of a + b and a - b side by side
__device__ __forceinline__ cf addsub_re(cf a, cf b) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ cf addsub_im(cf a, cf b) {
    cf r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// cos/sin(2*pi*i/32), i = 0..7 (compile-time twiddles of the in-register radix butterflies, first quadrant)
__device__ constexpr float kCos32[8] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                                        0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f,
                                        0.19509032201612826785f};
__device__ constexpr float kSin32[8] = {0.0f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f,
                                        0.70710678118654752440f, 0.83146961230254523708f, 0.92387953251128675613f,
                                        0.98078528040323044913f};

// cos/sin(2*pi*i/64), i = 0..15: the uniform factors W_64^rho of the factorised post-pass twiddles (Cfg::TWF)
__device__ constexpr float kCos64[16] = {1.00000000000000000000f, 0.99518472667219692873f, 0.98078528040323043058f, 0.95694033573220882438f, 0.92387953251128673848f, 0.88192126434835504956f, 0.83146961230254523567f, 0.77301045336273699338f, 0.70710678118654757274f, 0.63439328416364548779f, 0.55557023301960228867f, 0.47139673682599780857f, 0.38268343236508983729f, 0.29028467725446233105f, 0.19509032201612833135f, 0.09801714032956077016f};
__device__ constexpr float kSin64[16] = {0.00000000000000000000f, 0.09801714032956060363f, 0.19509032201612824808f, 0.29028467725446233105f, 0.38268343236508978178f, 0.47139673682599764204f, 0.55557023301960217765f, 0.63439328416364548779f, 0.70710678118654746172f, 0.77301045336273699338f, 0.83146961230254523567f, 0.88192126434835493853f, 0.92387953251128673848f, 0.95694033573220893540f, 0.98078528040323043058f, 0.99518472667219681771f};

// a * w with a wave-uniform w (scalar registers; same arithmetic as cmul)
__device__ __forceinline__ cf cmul_s(cf a, cf w) {
    cf r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]"
        : "=&v"(r) : "v"(a), "s"(w));
    return r;
}

// v * exp(-2*pi*i*Q/R) for a first-quadrant exponent (Q < R/4); exponents R/4 <= J < R/2 are this times a pending -i
template <int Q, int R>
__device__ __forceinline__ cf mul_w_q1(cf v) {
    constexpr int idx = Q * (32 / R);
    static_assert(idx >= 0 && idx < 8, "first quadrant");
    if constexpr (idx == 0) {
        return v;
    } else if constexpr (idx == 4) {   // (1 - i)/sqrt(2) * v = (v + (-i) v)/sqrt(2)
        return add_mi(v, v) * 0.70710678118654752440f;
    } else {                           // (c - i s)(x + i y) = (c x + s y) + i (c y - s x)
        constexpr float c = kCos32[idx], sn = kSin32[idx];
        const cf t = v * c;
        const cf sv = {sn, sn};
        cf r;
        // The sine as a SCALAR operand (unit A: the radix-32 plans hold seven of these constants as register pairs otherwise -- Cfg2048B 250
        // -> 238 VGPRs, the pair plan 256 + a spill -> 244); unit B keeps them in vector registers (JSG_TWIDDLE_CONST_VGPR: with scalar
        // operands the default scheduler of the 4096-point kernels pads 20 more s_nop per FFT round).  Same instruction, same bits.
#ifdef JSG_TWIDDLE_CONST_VGPR
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(v), "v"(sv), "v"(t));
#else
        asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(v), "s"(sv), "v"(t));
#endif
        return r;
    }
}

// In-register decimation-in-frequency DFT of R points, natural order in and out (the bit reversal is a compile-time
// renaming of registers).  UR: the upper half of the inputs carries a pending factor -i (the twiddles W_R^J with
// J >= R/4 of the layer before); it is absorbed by this layer's butterflies.  R = 8: 28 packed VALU ops, R = 16: 84.
template <int R, bool UR>
__device__ __forceinline__ void dft(cf (&x)[R]);

template <int R, bool UR, int J>
struct DifLayer {
    static __device__ __forceinline__ void run(const cf (&x)[R], cf (&a)[R / 2], cf (&b)[R / 2]) {
        const cf lo = x[J], hi = x[J + R / 2];
        a[J] = UR ? add_mi(lo, hi) : lo + hi;
        const cf d = UR ? sub_mi(lo, hi) : lo - hi;
        constexpr int Q = J % (R / 4);   // W_R^J = (-i) W_R^(J - R/4) for J >= R/4: the -i stays pending (dft<R/2, true>)
        b[J] = mul_w_q1<Q, R>(d);
        if constexpr (J + 1 < R / 2) DifLayer<R, UR, J + 1>::run(x, a, b);
    }
};

template <int R, bool UR = false>
__device__ __forceinline__ void dft(cf (&x)[R]) {
    if constexpr (R == 2) {
        const cf a = x[0], b = x[1];
        x[0] = UR ? add_mi(a, b) : a + b;
        x[1] = UR ? sub_mi(a, b) : a - b;
    } else if constexpr (R > 2) {
        cf a[R / 2], b[R / 2];
        DifLayer<R, UR, 0>::run(x, a, b);
        dft<R / 2, false>(a);
        dft<R / 2, true>(b);   // b[J], J >= R/4, still lacks its factor -i
#pragma unroll
        for (int q = 0; q < R / 2; ++q) {
            x[2 * q] = a[q];
            x[2 * q + 1] = b[q];
        }
    }
}

// The first radix stage with the window multiply folded into its first butterfly layer: that layer forms lo + hi and lo - hi of two
// WINDOWED samples; the upper one's product is taken on its own (one packed multiply), the lower one's is fused into the add and the
// subtract (one packed fma each) -- a = fma(x_lo, w_lo, P_hi), d = fma(x_lo, w_lo, -P_hi) -- which saves P/2 packed multiplies per frame.
// (The optimiser contracted the separate statements into exactly this form before; it is written out so that the rounding is a property
// of the source, not of an optimisation: oracle/jsg_mirror.c restates it, tests/test_gpu_mirror.py holds the two together bit for bit.)
// x[J], J < R/2: raw samples with their window values w[J]; x[J + R/2]: already multiplied by their window values.
template <int R, int J>
struct DifLayerWin {
    static __device__ __forceinline__ void run(const cf (&x)[R], const cf (&w)[R / 2], cf (&a)[R / 2], cf (&b)[R / 2]) {
        const cf hi = x[J + R / 2];
        a[J] = __builtin_elementwise_fma(x[J], w[J], hi);
        const cf d = __builtin_elementwise_fma(x[J], w[J], -hi);
        b[J] = mul_w_q1<J % (R / 4), R>(d);
        if constexpr (J + 1 < R / 2) DifLayerWin<R, J + 1>::run(x, w, a, b);
    }
};
template <int R>
__device__ __forceinline__ void dft_win(cf (&x)[R], const cf (&w)[R / 2]) {
    static_assert(R >= 4, "radix");
    cf a[R / 2], b[R / 2];
    DifLayerWin<R, 0>::run(x, w, a, b);
    dft<R / 2, false>(a);
    dft<R / 2, true>(b);
#pragma unroll
    for (int q = 0; q < R / 2; ++q) {
        x[2 * q] = a[q];
        x[2 * q + 1] = b[q];
    }
}

// everything of dft_win<R> behind the sums and differences of its first layer (a[J], d[J] as DifLayerWin forms them): natural order out
template <int R, int J>
struct DifTwiddle {
    static __device__ __forceinline__ void run(const cf (&d)[R / 2], cf (&b)[R / 2]) {
        b[J] = mul_w_q1<J % (R / 4), R>(d[J]);
        if constexpr (J + 1 < R / 2) DifTwiddle<R, J + 1>::run(d, b);
    }
};
template <int R>
__device__ __forceinline__ void dft_rest(cf (&a)[R / 2], const cf (&d)[R / 2], cf (&x)[R]) {
    static_assert(R >= 4, "radix");
    cf b[R / 2];
    DifTwiddle<R, 0>::run(d, b);
    dft<R / 2, false>(a);
    dft<R / 2, true>(b);
#pragma unroll
    for (int q = 0; q < R / 2; ++q) {
        x[2 * q] = a[q];
        x[2 * q + 1] = b[q];
    }
}

// ------------------------------------------------------------------------------------------------------------
// Split-radix register DFTs for R = 16 and 32 (round 5).  The radix-2 recursion above spends 10 (R = 16) / 34
// (R = 32) non-trivial twiddle multiplies per transform, the split-radix one 8 / 26: X[2k] comes from the half-size transform of
// x[n] + x[n + R/2] WITHOUT a twiddle, and the odd outputs from two quarter-size transforms,
//     X[4k+1] = DFT_{R/4}( (d[n] - i d[n + R/4]) W_R^n  ),   X[4k+3] = DFT_{R/4}( (d[n] + i d[n + R/4]) W_R^{3n} ),   d[n] = x[n] - x[n + R/2],
// where the factors -i / +i are absorbed by the packed add's operand selects (add_mi / sub_mi) and n = 0 needs no multiply.  Same number of
// additions (5 R/2 log2 R ... the L-shaped butterfly is two radix-2 layers), 2 x 8 = 16 fewer packed instructions per 32-point transform
// (212 instead of 228), 4 fewer per 16-point one (80 instead of 84).  Transforms of 8 points and fewer are the radix-2 code above (the
// counts are equal there), so the 512- and 1024-point plans keep their bits.
// ------------------------------------------------------------------------------------------------------------
// v * exp(-2 pi i E / 32) for E = 1 .. 31, E % 8 != 0: the first-quadrant constants (cos, sin)(2 pi (E % 8) / 32) in ONE scalar register
// pair, the quadrant E / 8 -- a factor (-i)^(E/8) -- in the operand selects and negate modifiers of the two instructions:
//   q = 0: (c x + s y,  c y - s x)    q = 1: (c y - s x, -c x - s y)    q = 2: (-c x - s y, -c y + s x)    q = 3: (s x - c y,  c x + s y)
// each as t = (+-c v_a, +-c v_b) (packed multiply) and r = t + (+-s v_c, +-s v_d) (packed fma): the roundings of cmul.
template <int E>
__device__ __forceinline__ cf mul_w32(cf v) {
    static_assert(E > 0 && E < 32 && E % 8 != 0, "non-trivial twiddle");
    constexpr int e = E % 8, q = E / 8;
    const cf cs = {kCos32[e], kSin32[e]};
    cf r;
#ifdef JSG_TWIDDLE_CONST_VGPR
#define JSG_CS_CONSTRAINT "v"
#else
#define JSG_CS_CONSTRAINT "s"
#endif
    if constexpr (q == 0)
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]\n\t"
            "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=&v"(r) : "v"(v), JSG_CS_CONSTRAINT(cs));
    else if constexpr (q == 1)
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,0] neg_hi:[1,0]\n\t"
            "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]" : "=&v"(r) : "v"(v), JSG_CS_CONSTRAINT(cs));
    else if constexpr (q == 2)
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]\n\t"
            "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=&v"(r) : "v"(v), JSG_CS_CONSTRAINT(cs));
    else
        asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,0] neg_lo:[1,0]\n\t"
            "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=&v"(r) : "v"(v), JSG_CS_CONSTRAINT(cs));
#undef JSG_CS_CONSTRAINT
    return r;
}

template <int R>
__device__ __forceinline__ void sr_dft(cf (&x)[R]);

// the odd half of one split-radix level: z1[n] = (d[n] - i d[n + R/4]) W_R^n, z3[n] = (d[n] + i d[n + R/4]) W_R^{3n}
template <int R, int N0>
struct SrOdd {
    static __device__ __forceinline__ void run(const cf (&d)[R / 2], cf (&z1)[R / 4], cf (&z3)[R / 4]) {
        const cf p = d[N0], q = d[N0 + R / 4];
        const cf a = add_mi(p, q), b = sub_mi(p, q);
        if constexpr (N0 == 0) {
            z1[0] = a;
            z3[0] = b;
        } else {
            z1[N0] = mul_w32<N0 * (32 / R)>(a);
            z3[N0] = mul_w32<(3 * N0 * (32 / R)) % 32>(b);
        }
        if constexpr (N0 + 1 < R / 4) SrOdd<R, N0 + 1>::run(d, z1, z3);
    }
};
// everything behind the first layer (a[n] = x[n] + x[n + R/2], d[n] = x[n] - x[n + R/2]); natural order out
template <int R>
__device__ __forceinline__ void sr_rest(cf (&a)[R / 2], const cf (&d)[R / 2], cf (&x)[R]) {
    cf z1[R / 4], z3[R / 4];
    SrOdd<R, 0>::run(d, z1, z3);
    sr_dft<R / 2>(a);
    sr_dft<R / 4>(z1);
    sr_dft<R / 4>(z3);
#pragma unroll
    for (int k = 0; k < R / 2; ++k) x[2 * k] = a[k];
#pragma unroll
    for (int k = 0; k < R / 4; ++k) {
        x[4 * k + 1] = z1[k];
        x[4 * k + 3] = z3[k];
    }
}
template <int R>
__device__ __forceinline__ void sr_dft(cf (&x)[R]) {
    if constexpr (R <= 8) dft<R, false>(x);
    else {
        cf a[R / 2], d[R / 2];
#pragma unroll
        for (int J = 0; J < R / 2; ++J) {
            a[J] = x[J] + x[J + R / 2];
            d[J] = x[J] - x[J + R / 2];
        }
        sr_rest<R>(a, d, x);
    }
}
// ... with the window multiply folded into the first layer, as dft_win: x[J], J < R/2, raw with window values w[J]; x[J + R/2] windowed
template <int R>
__device__ __forceinline__ void sr_dft_win(cf (&x)[R], const cf (&w)[R / 2]) {
    if constexpr (R <= 8) dft_win<R>(x, w);
    else {
        cf a[R / 2], d[R / 2];
#pragma unroll
        for (int J = 0; J < R / 2; ++J) {
            const cf hi = x[J + R / 2];
            a[J] = __builtin_elementwise_fma(x[J], w[J], hi);
            d[J] = __builtin_elementwise_fma(x[J], w[J], -hi);
        }
        sr_rest<R>(a, d, x);
    }
}

// Pair plan (Cfg::PAIR): the samples are complex (z = x_c + i x_(c+1)) and the window is real, one float per sample, held as PAIRS of
// consecutive values (w[m], w[m+1]) in one aligned register pair: the multiply broadcasts one of the two floats through the packed
// instruction's operand selects (written out: the compiler derives the broadcast form for some uses and copies the float into a fresh
// register pair for others -- 2 x v_mov each, 30 per round in the ISA of the first version).
template <int HI>
__device__ __forceinline__ cf mul_bc(cf v, cf wp) {   // v * wp[HI] (both components)
    cf r;
    if constexpr (HI) asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(v), "v"(wp));
    else asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(r) : "v"(v), "v"(wp));
    return r;
}
template <int HI, bool NEG>
__device__ __forceinline__ cf fma_bc(cf v, cf wp, cf c) {   // v * wp[HI] + c   (NEG: - c)
    cf r;
    if constexpr (HI && NEG) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(v), "v"(wp), "v"(c));
    else if constexpr (HI) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(v), "v"(wp), "v"(c));
    else if constexpr (NEG) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(v), "v"(wp), "v"(c));
    else asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(v), "v"(wp), "v"(c));
    return r;
}
// The first butterfly layer of the pair plan's stage 1 with the window folded in, as DifLayerWin: hi = z[J + R/2] w[J + R/2] (one packed
// multiply), a = fma(z[J], w[J], hi), d = fma(z[J], w[J], -hi), b = W_R^J d.  wp[i] = (w[2 i], w[2 i + 1]).
template <int R, int J>
struct PairFirstLayer {
    static __device__ __forceinline__ void run(const cf (&z)[R], const cf (&wp)[R / 2], cf (&a)[R / 2], cf (&b)[R / 2]) {
        const cf hi = mul_bc<(J + R / 2) % 2>(z[J + R / 2], wp[(J + R / 2) / 2]);
        a[J] = fma_bc<J % 2, false>(z[J], wp[J / 2], hi);
        const cf d = fma_bc<J % 2, true>(z[J], wp[J / 2], hi);
        b[J] = mul_w_q1<J % (R / 4), R>(d);
        if constexpr (J + 1 < R / 2) PairFirstLayer<R, J + 1>::run(z, wp, a, b);
    }
};

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>), written out (the rounds of a run, stft_db_kernel STREAM == 2)
template <class Fn, int... Ts>
__device__ __forceinline__ void for_each_t(Fn&& f, std::integer_sequence<int, Ts...>) {
    (f(std::integral_constant<int, Ts>{}), ...);
}

// Lanes of one wavefront run in lock-step, so a wave-private LDS exchange needs no s_barrier; what it does need is
// that the COMPILER keeps the stores ahead of the loads that other lanes of the same wave perform.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// two floats that are only 4-byte aligned (odd hop sizes such as the reference's perc10 hop of 205 samples)
struct __attribute__((packed, aligned(4))) f2u {
    float x, y;
};
__device__ __forceinline__ cf to_cf(const f2u& v) { return cf{v.x, v.y}; }

// ------------------------------------------------------------------------------------------------------------
// per-size configuration (mirrors tools/fft_model.py CONFIGS; layouts found by its conflict search)
// ------------------------------------------------------------------------------------------------------------
// TLOC: where the lane tables (window pairs, stage-1/2 twiddles, post-pass twiddles; layout: Cfg::TAB_*) live:
//   0 = read from global memory (L1/L2) at every use, 1 = brought into LDS once per workgroup and read from there.
// FPW:  frames a wavefront transforms at the same time, as one interleaved instruction stream (L = 64 plans).  All plans use 1: two
//   interleaved frames per wavefront gave bit-identical results but were slower (tried and measured); the loops over F stay.
// (Round 4: the development ablation / stamp modes that used to live in this kernel -- ABL, JSG_X_* switches, in-kernel time stamps --
//   are gone from the product source; they are in the history, rounds 2-3, with the measurements they produced in DESIGN.md section 6.)
// PAIR (round 5): the plan transforms a PAIR of channels as ONE complex sequence z = x1 + i x2 of N points (sum-type mixes over an even
//   channel count): |X1[k]|^2 + |X2[k]|^2 = (|Z[k]|^2 + |Z[N-k]|^2) / 2, so no real-split post pass and no mirror exchange per channel
//   frame -- one fold per COLUMN instead.  The N-point transform is split by decimation in time: the two half-waves of a wavefront (L = 32)
//   transform the even and the odd samples (E, O: two N/2-point FFTs with the two-stage engine of Cfg2048B), a v_permlane32_swap brings
//   E[k] and O[k] into one lane, and Z[k] = E[k] + W_N^k O[k], Z[k + N/2] = E[k] - W_N^k O[k].  See "pair plan" in the kernel.
#ifndef JSG_X_RUNLEN
#define JSG_X_RUNLEN 2       // (variant builds sweep this: 2 / 4 / 8 -- measured 2 >= 4 > none > 8)
#endif
constexpr int kRunLen = JSG_X_RUNLEN;   // STREAM == 2 ("runs"): consecutive columns a wavefront transforms one after the other (see stft_db_kernel)

template <int N_, int R1_, int R2_, int R3_, int L_, int S1_, int AX_, int AY_, int AZ_, int WPB_, int TLOC_, int WPS_,
          int FPW_ = 1, int EARLY1_ = 0, int TWF_ = 0, int PAIR_ = 0, int RTAB_ = 0>
struct Cfg {
    static constexpr bool PAIR = PAIR_ != 0;
    // RTAB (round 5): the one-channel dB kernels of the plan keep their stage-1 and post-pass twiddles (R1 - 1 + P/2 complex values per lane) in
    // registers for the whole kernel instead of reading them from the LDS tables in every FFT round.  A timing experiment with ALL lane-table
    // reads removed put their cost at 3-4 % of the C2 dispatch (5-7 % at C3, where no register is free); Cfg1024's one-channel kernel has 66 of
    // the 96 VGPRs that five waves per SIMD -- all its LDS allows -- may use: 6 of its 14 table reads per frame go, 86 VGPRs, +3..5 %.
    static constexpr bool RTAB = RTAB_ != 0;
    // EARLY1 (round 5): the first butterfly layer of stage 1 -- the one that carries the window -- runs AHEAD of the next round's frame loads.
    // It consumes every raw sample, so the loads land in the registers they leave; otherwise the raw lower inputs outlive the loads and the
    // compiler copies them out of the way first (Cfg4096B: 32 v_mov per FFT round in the ISA).  Same operations, same bits.
    static constexpr bool EARLY1 = EARLY1_ != 0;
    // TWF: factorised twiddle tables (for plans whose full lane tables do not fit beside the exchange buffers).  The stage-2
    // twiddle W_M^(n3 (k1 + R1 k2)) is read as B[n3][k2] = W_(M/R1)^(n3 k2) (one row per n3, shared by the lanes) times the
    // lane's constant A[v] = W_M^(n3 k1); the post-pass twiddle -i W_N^(ll + L rho) as the lane's constant C = -i W_N^ll times
    // the wave-uniform W_(N/L)^rho (compile-time constants, N / L = 64).  One more complex multiply per value, 24 KB less LDS.
    static constexpr bool TWF = TWF_ != 0;
    static constexpr int N = N_, M = N_ / 2, R1 = R1_, R2 = R2_, R3 = R3_, L = L_;
    static constexpr int P = M / L;                  // complex values per lane
    static constexpr int U1 = P / R1, U2 = P / R2, U3 = P / R3;
    static constexpr int S1 = S1_, AX = AX_, AY = AY_, AZ = AZ_;
    static constexpr int SUB = L < 64 ? 64 / L : 1;  // frames side by side in one wavefront (L = 32: two)
    static constexpr int WPF = L > 64 ? L / 64 : 1;  // wavefronts per frame (L = 128, 256: the exchanges use s_barrier)
    static constexpr int FPW = FPW_;                 // frames interleaved in one wavefront's instruction stream
    static constexpr int TL = L;                     // entries per lane-table row (L = 32: both half-waves read the same entries)
    static constexpr bool TWO_STAGE = R3_ == 1;      // R1 * R2 = M: one exchange, the second radix stage leaves bin ll + L*k2 in register k2
    static constexpr int WPB = WPB_;                 // wavefronts per workgroup
    static constexpr int SLOTS = WPB * 64 / L * FPW; // exchange regions (sub-transforms in flight) per workgroup
    static constexpr int TPB = PAIR ? WPB : SLOTS;   // frames (columns) per workgroup per iteration; PAIR: one column per wavefront
    static constexpr int TLOC = TLOC_;
    static constexpr int WPS = WPS_;                 // waves per SIMD the register allocator is asked to allow
    // Lane tables (float2 elements): window pairs [P][TL]; stage-1 twiddles W_{R1 R2}^{n2 k1}, which depend on the lane
    // only through n2 = t1 / R3, stored once per n2 as [R2][TS1]; stage-2 twiddles [P][TL]; post-pass twiddles of the
    // lower half of the bins [P/2][TL].
    // They are read two values (16 bytes) at a time: ds_read_b128 moves 1 KiB per wave-instruction at the full LDS rate
    // even with one or two waves per SIMD, 8-byte reads need about four (MI355X_MICROARCH.md, LDS), and every table
    // value is used by exactly one instruction -- so value j of a lane sits next to value j+1: element (j, e) of a
    // [J][TL] table is stored at ((j / 2) * TL + e) * 2 + j % 2.  Stage-1 rows hold k1 = 1.. at column k1 - 1, row stride
    // R1 + 2 (16-byte aligned rows whose 16-byte chunks fall into different banks for the n2 groups of a wave).
    static constexpr int TS1 = R1_ + 2;
    static constexpr int tab_idx(int j, int e) { return ((j / 2) * L_ + e) * 2 + j % 2; }
    static constexpr int TSB = R2_ + 2;              // TWF: row stride of B (16-byte aligned rows in different banks)
    static constexpr int TAB_WIN = 0, TAB_TW1 = P * TL, TAB_TW2 = TAB_TW1 + R2_ * TS1;
    static constexpr int TAB_A = TAB_TW2 + R3_ * TSB;                                   // TWF: [U2][TL] behind B
    static constexpr int TAB_POST = TWF ? TAB_A + U2 * TL : TAB_TW2 + (TWO_STAGE ? 0 : P * TL);   // TWF: C[TL]
    static constexpr int TAB_ELEMS = TAB_POST + (TWF ? TL : (P / 2) * TL);
    static constexpr int e1max = (R1 - 1) * S1 + M / R1;
    static constexpr int e2max = TWO_STAGE ? 0 : (R1 - 1) * AX + (R2 - 1) * AY + (R3 - 1) * AZ + 1;
    static constexpr int raw = e1max > e2max ? (e1max > M + 1 ? e1max : M + 1) : (e2max > M + 1 ? e2max : M + 1);
    // float2 elements per frame slot, 16-byte multiple.  L = 16 (four frames side by side in a wavefront): a multiple of 32 elements, i.e. the
    // regions of neighbouring frames start at the same bank.  A ds_read_b128 is served in lane groups {0-3, 12-15, 20-27}, ... (MI355X_MICROARCH.md,
    // LDS): lanes 0-3 and 12-15 of one frame together with lanes 4-11 of the NEXT -- with equal region phases the sixteen 16-byte pieces of the
    // exchange read (lane stride S1 = 34 elements = 272 bytes) are the sixteen distinct pieces of one 256-byte row.  (First version: regions 128
    // bytes out of phase, chosen for the 8-byte accesses -- 12.5 % of the kernel's LDS cycles were bank conflicts, all in these reads.)  The
    // 8-byte reads of the post pass, which ARE served two frames at a time, get their 128 bytes of phase from a skew of that pass alone (PSKEW).
    static constexpr int LDS_ELEMS = L_ == 16 ? (raw + 31) / 32 * 32 : (raw + 1) & ~1;
    static constexpr int PSKEW = L_ == 16 ? 16 : 0;   // elements by which the odd frames of a wavefront shift their post-pass exchange
    static constexpr int LDS_BYTES = LDS_ELEMS * SLOTS * 8;                        // dynamic: exchange buffers
    static constexpr int LDS_TOTAL = LDS_BYTES + (TLOC == 1 ? TAB_ELEMS : 2) * 8;  // + static: lane tables
    static_assert(R1 * R2 * R3 == M, "radices");
    static_assert(P % R1 == 0 && P % R2 == 0 && P % R3 == 0, "each lane owns whole butterflies");
    static_assert(WPB % WPF == 0, "a workgroup holds whole frames");
    static_assert(FPW == 1 || L == 64, "interleaved frames: one-wavefront-per-frame plans only");
    static_assert(TLOC == 0 || TLOC == 1, "lane tables in registers: removed (DESIGN.md, tried and measured)");
    static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget of one CU");
    static_assert(!TWF || ((N_ / L_ == 64 || N_ / L_ == 32) && L_ % R3_ == 0 && !TWO_STAGE && TLOC_ == 1), "factorised tables: 4096-point plans");
    // PAIR: the window table holds P x 64 FLOATS (w[64 m + 2 ll + h], four values of a lane side by side: [P/4][64][4]) in the P*TL float2
    // elements of TAB_WIN, and TAB_POST holds the last stage's twiddles W_N^(lane + 64 j), j < P/4, as [P/4][64] pairs of j (tab_idx with 64
    // entries per row) -- the same element counts as the tables of Cfg2048B, so the LDS budget is that plan's
    static_assert(!PAIR || (TWO_STAGE && L_ == 32 && FPW_ == 1 && TLOC_ == 1 && !TWF && P % 4 == 0 && (P / 2) * TL == (P / 4) * 64), "pair plan: two-stage, 32 lanes per half");
    static constexpr int pair_tw_idx(int j, int lane) { return ((j / 2) * 64 + lane) * 2 + j % 2; }
};

using Cfg512 = Cfg<512, 8, 8, 4, 32, 36, 4, 33, 1, 8, 1, 2>;   // (a 80-VGPR budget = 3 workgroups per CU measured no faster)
// <= 80 VGPRs for EVERY instantiation (six waves per SIMD allowed).  Left at 2 the ILP-first scheduler takes 88 for the mixing
// instantiations: stereo -2..-4 % with the cap.
// Workgroups of FOUR waves (four frames per step; five workgroups = 20 waves per CU, 29.5 KB of LDS each) since the end of round 4, where
// rounds 1-4 had eight (three workgroups = 24 waves per CU): measured with tools/pool_probe.py / strided_probe.py on six boxes, variant
// builds of the same source -- the C2 dispatch +2.3..+2.7 % on five boxes (0.582-0.585 -> 0.594-0.601 of 8 TB/s) and -1.9 % on one
// (0.648 -> 0.636), the C4 shard (8 channels, one column each) +5.9 %, 75 % overlap +11 %, single launches of <= 1024 frames +14 %, two
// channels mixed level, eight channels mixed -3.5 %.  Workgroups of 2 waves: +1 %; of 3, 5, 6 waves: -4..-5 %.
using Cfg1024 = Cfg<1024, 8, 8, 8, 64, 72, 9, 72, 2, 4, 1, 6, 1, 0, 0, 0, 1>;   // (RTAB: the one-channel dB kernels hold two of the four tables in registers)
// ... and the eight-wave form for the single-kernel display path (OUTK == 2), whose store phase is laid out for eight columns per step;
// same tables, same arithmetic per frame (bit-identical columns); instantiated for that path only (image_only)
using Cfg1024I = Cfg<1024, 8, 8, 8, 64, 72, 9, 72, 2, 8, 1, 6>;
// 1024 points as TWO stages with ONE exchange (round 6): 512 complex points = split-radix 16 x 32, the engine of Cfg2048B.
// 16 lanes per frame, 32 complex values per lane, FOUR frames side by side in a wavefront, one 8-wave workgroup (32 frames per step) per CU:
// 32 x 4.4 KB of exchange + 10.5 KB of tables.  Against the three-stage plan per frame: one exchange of 32 writes + 16 16-byte reads per four
// frames instead of two of 8 + 8 per frame, the lane tables read once per four frames, 2 x 80 + 212 packed instructions of butterflies per four
// frames... (tools/kernel_regs.py and the PMC figures in DESIGN.md section 6).  A 4 x 4 transposition across the quarter-waves
// (v_permlane32_swap + v_permlane16_swap) turns the registers into runs of 64 consecutive bins of ONE frame before the column stores.
// dB / power columns only (sum-type and one-channel mixes); opt-in or automatic: see wants_plan_1024b in jsg_kernels.hip.
using Cfg1024B = Cfg<1024, 16, 32, 1, 16, 34, 0, 0, 0, 8, 1, 1>;
// factorised stage-2 / post tables (Cfg::TWF): 11.3 instead of 21.2 KB of tables, so that THREE 4-wave workgroups fit a CU (12 waves
// instead of 8): stereo launches -9..-13 %, mono -1..-5 %
using Cfg2048 = Cfg<2048, 16, 8, 8, 64, 72, 65, 16, 2, 4, 1, 3, 1, 1, 1>;   // (6-, 8-, 12-wave workgroups: no faster)
// 2048 points as TWO radix-32 stages with ONE exchange: 32 lanes per frame, 32 complex values per lane, two frames side by side
// in a wavefront.  Against the three-stage plan (16*8*8, 64 lanes): the same butterfly count, but 40 % fewer LDS
// instructions per frame (one exchange of 16 + 16 instead of two of 32 + 32), which is what capped C3 (VALU and LDS
// each about half busy at two waves per SIMD).  One 8-wave workgroup per CU: 16 frames * 8.7 KB of exchange + 21 KB of tables.
// It needs 210-252 VGPRs (two waves per SIMD) and 156 KB of LDS, so a CU holds exactly one such workgroup, which is launched
// once per CU and loops over its frames.  For launches that fill the GPU in whole rounds of 256 workgroups it is ahead of the
// three-stage plan, the more channels are mixed into a column the more (8 ch -10..-13 %, 4 ch -3..-8 %, 1-2 ch level);
// for anything smaller the 4-frame workgroups of the three-stage plan use more CUs (tools/abbench --cfg x2048 / mid).
// The launcher picks by channel count and by how well the launch fills its rounds (stft_launch_impl).
using Cfg2048B = Cfg<2048, 32, 32, 1, 32, 34, 0, 0, 0, 8, 1, 1>;
// The pair plan (round 5): the same two radix-32 stages and the same single exchange per 1024-point sub-transform, but a
// wavefront transforms the channel pair (c, c + 1) of ONE column as z = x_c + i x_(c+1): half-wave 0 the even samples, half-wave 1 the odd
// ones, one radix-2 stage across the half-waves (v_permlane32_swap + one twiddle), |Z|^2 accumulated over the pairs of the column in
// registers (bins 0 .. 2047 of the complex spectrum), and ONE fold per column: out[k] = (acc[k] + acc[2048 - k]) / 2 / channels.
// Against Cfg2048B per two channel FFTs: no paired post pass (2 x 72 vector instructions) and no mirror exchange (2 x 33 LDS instructions),
// instead 32 swaps + 32 twiddle + 32 butterfly + 32 power instructions and, per column, a 16-value mirror exchange.
using Cfg2048P = Cfg<2048, 32, 32, 1, 32, 34, 0, 0, 0, 8, 1, 1, 1, 0, 0, 1>;
constexpr int k2048B_min_channels = 2;   // channels mixed into one column from which the two-stage plan is the faster one (round 3, ILP-first
                                         // scheduler, 32 768 FFTs per launch: stereo 42.9-52.8 vs 48.1-55.9 us, 4 ch 41.3-49.1 vs 45.1-52.0, 8 ch
                                         // 39.6-46.5 vs 44.8-50.6; mono level: 51.2-62.7 vs 52.1-61.2)
constexpr int k4096B_min_channels = 1;   // ... the one-wavefront-per-frame 4096-point plan is ahead at every channel count once the rounds are
                                         // full (round 3, 16 384 FFTs, inputs and rings rotating over 1 GB: mono 66.5 vs 73.4 us, stereo 56.7 vs 60.8,
                                         // 4 ch 51.6 vs 57.6, 8 ch 50.3 vs 54.7); launches that do not fill their rounds keep the two-wave plan
// The "B" plans run ONE 8-wave workgroup per CU (16 / 8 frames at a time), i.e. a launch proceeds in rounds of <CU count> workgroups:
// a launch that fills its last round badly leaves CUs idle where the small workgroups of the other plan would fill them
// (1024 stereo 2048-point frames: 7.2 vs 13.0 us).  At full rounds "B" is about 13 % faster, so it is used when the rounds of
// the launch are at least 87 % full (224..256 workgroups on 256 CUs, 446..512, ..., everything from 7 rounds on).  Sub-launches of one
// stream that fall on different sides of that rule therefore agree within the float32 bound, not bit for bit; everything
// with one or two channels per column, and the engine's per-block launches, always take the small-workgroup plan.
constexpr double kB_min_round_fill = 0.87;
using Cfg4096 = Cfg<4096, 16, 8, 16, 128, 144, 1, 272, 17, 4, 1, 1, 1, 1>;
// 4096 points with ONE wavefront per frame: 8*16*16, 32 complex values per lane, both exchanges wave-private (no workgroup
// barrier at all), one 8-wave workgroup per CU.  8 x 17.6 KB of exchange leave 22 KB of LDS for tables, so only the window
// and the stage-1 rows are kept whole and the other two tables are factorised (Cfg::TWF: one more complex multiply per
// value).  Like Cfg2048B it trades waves per SIMD for independence of the waves: measured (abbench --cfg x4096, 16 384 FFTs,
// us three-stage two-wave plan -> this one) 8 ch 56.4 -> 48.4, 4 ch 57.2 -> 52.5, 2 ch 59.4 -> 60.6 (61.4 -> 67.4 at 50 %
// overlap), 1 ch 66.4 -> 79.1: it is the plan of the launches that mix >= 3 channels into a column.
using Cfg4096B = Cfg<4096, 8, 16, 16, 64, 272, 276, 17, 1, 8, 1, 1, 1, 1, 1>;    // two wavefronts per frame, two frames per workgroup
// four wavefronts per frame, one frame per workgroup.  Its full lane tables (82 KB) do not fit beside the exchange buffer,
// and reading them from L2 at every use cost 12 %: the factorised set (Cfg::TWF, 40.7 KB) lives in LDS like everywhere else.
using Cfg8192 = Cfg<8192, 16, 16, 16, 256, 272, 1, 272, 17, 4, 1, 1, 1, 0, 1>;
// (frames of more than one wavefront exchange through the workgroup barrier, so every further frame in the workgroup joins
// five barriers per FFT: 8-wave workgroups were 9-15 % slower, 12-wave ones 30 %)

struct StftKArgs {
    const float* in;
    long long in_pitch;
    int hop, feedblocks;
    int regular;         // hop * feedblocks == N: frame j starts at j*hop (no division in the kernel)
    int c_begin, c_end;  // channel range that is combined into one column (mixed modes)
    int per_channel;     // one column per (channel, frame): blockIdx.y is the channel
    int linear;          // store linear power instead of dB
    int exact_log;       // dB by jsg_exact_db (the XLOG instantiations) instead of v_log_f32
    float scale;         // AbsMean: 1/C (exact for power-of-two C); others: 1
    float divisor;       // AbsMean: float(C)
    int exact_div;       // C is not a power of two: divide (IEEE) instead of scaling
    unsigned first_frame, n_frames;
    float* out;
    long long out_pitch, out_cpitch;
    float* tail;         // jsg_stft_args.out_tail: bin N/2 of ring column col of row r (batch x channel plane) goes to tail[r * ring_w + col]
                         // instead of out[.. + N/2] -- the 4-byte piece that would otherwise open a 17th 128-byte line per column
    int ring_w, ring_pos;
    int iters;
    const float2* tab;   // lane tables (Cfg::TAB_* layout): window pairs, stage-1 / stage-2 twiddles, post-pass twiddles
    int xcd_remap;                // 1: XCD-aware block remap (default); 0: identity (development A/B)
    int chunked;                  // 0: grid-stride traversal (default); 1: one contiguous chunk per workgroup
    // OUTK == 1 (fused display path): the column leaves as 8-bit palette indices instead of dB floats
    unsigned char* idx;
    long long idx_pitch;          // bytes between index columns
    float vmin, vmax, top, mult;  // CColorPalette::setValueRange / getRGBColor (CColorpalette.cpp:39-54, CColorpalette.h:34-45)
    int n_colors;
    int cmap_fast;                // 1: for this range the index is min(u32((v - vmin) * mult), n_colors - 1) for EVERY v (see color_index2_fast)
    // OUTK == 2 (single-kernel display path): the workgroup turns the columns of an iteration into ARGB image rows itself
    unsigned* argb;               // image [height][argb_pitch], pixel (x, height - 1 - bin)
    long long argb_pitch;
    const int* lut;               // n_colors entries 0x00RRGGBB
    int x_first, x_wrap;          // column i of the launch lands at x = (x_first + i) % x_wrap
    // OUTK == 2, several images of one geometry in one launch (jsg_stft_image_launch_strided): the launch's workgroup iterations
    // ("groups" of TPB columns) are numbered through the images, n_groups = n_images * img_gpi; n_frames stays the columns of ONE image
    unsigned n_groups;            // groups of the whole launch
    unsigned img_gpi;             // groups per image = ceil(n_frames / TPB)
    unsigned long long img_magic; // ceil(2^40 / img_gpi): group / img_gpi == (group * img_magic) >> 40 exactly for group, img_gpi <= 2^20
    long long in_image_stride;    // floats between the inputs of consecutive images
    long long argb_image_stride;  // pixels between consecutive images
    // STREAM != 0 (jsg_stft_db_launch_strided): the same numbering through the ROWS of the launch -- a row is one batch, or one
    // (batch, channel) pair in per-channel mode: row = batch * bat_cpb + channel.  n_groups / img_gpi / img_magic / in_image_stride
    // (floats between the inputs of consecutive batches) are shared with the image form above.
    unsigned bat_cpb;             // rows per batch: channels in per-channel mode, else 1
    unsigned long long bat_magic; // ceil(2^40 / bat_cpb)
    long long out_batch_stride;   // floats between the rings of consecutive batches
};

// dB = 10*log10(p + 1e-11f) -- reference Spectrogram.cpp:107 with g_minValForLogSpectrogram (:36).
// v_log_f32 (log2, 1 ulp) times float(10*log10(2)); inputs are >= 1e-11, never denormal.  The constant's rounding
// (<= 6e-8 relative) stays below half an ulp of the dB value, so no hi/lo split is spent on it.
__device__ __forceinline__ float to_db(float p) {
    return __builtin_amdgcn_logf(p + 1e-11f) * 3.0102999566398120f;
}

// CColorPalette::getRGBColor's index (reference CColorpalette.h:34-45), float32 arithmetic, truncation.
__device__ __forceinline__ int color_index(float v, float vmin, float vmax, float top, float mult, int n_colors) {
    if (v >= vmax) v = top;          // value = m_Max*0.9999f
    if (v < vmin) v = vmin;
    int idx = (int)((v - vmin) * mult);
    return idx < n_colors ? idx : n_colors - 1;
}

// The same for the two values of a packed pair: the subtraction and the multiplication as one packed instruction each (same
// roundings), the lower clamp as v_max_f32 (a NaN ends at index 0 either way: v_cvt_i32_f32(NaN) = 0).  12 instead of 16
// instructions per pair; the display kernels spend them on every bin.
__device__ __forceinline__ void color_index2(cf v, float vmin, float vmax, float top, float mult, int n_colors, int& ix, int& iy) {
    v.x = v.x >= vmax ? top : v.x;
    v.y = v.y >= vmax ? top : v.y;
    // (written out: for __builtin_fmaxf the compiler first quiets a possible signalling NaN with a second v_max_f32 per value)
    asm("v_max_f32 %0, %1, %2" : "=v"(v.x) : "v"(v.x), "s"(vmin));
    asm("v_max_f32 %0, %1, %2" : "=v"(v.y) : "v"(v.y), "s"(vmin));
    const cf t = (v - cf{vmin, vmin}) * cf{mult, mult};
    ix = (int)t.x;
    iy = (int)t.y;
    ix = ix < n_colors ? ix : n_colors - 1;
    iy = iy < n_colors ? iy : n_colors - 1;
}

// The same index without the two selects, for value ranges where they cannot matter (round 5; the launcher decides: cmap_is_fast).  The
// reference replaces v >= max by max * 0.9999 and v < min by min before it scales.  The lower clamp is what v_cvt_u32_f32 does to a negative
// product (it saturates at 0; NaN gives 0 too, as above).  The upper replacement is visible only if the index of max * 0.9999 is NOT the
// last colour -- for the plugin's ranges it is (-50 .. +50 dB, 256 colours: 255.987 -> 255) -- or if a value between max and max * 0.9999
// (a negative max) could land below it; with mult > 0 the scaling is monotone, so checking those two indices on the host covers every v.
// 5 instead of 12 instructions per pair: the display kernels spend them on every bin.
__device__ __forceinline__ void color_index2_fast(cf v, float vmin, float mult, int n_colors, int& ix, int& iy) {
    const cf t = (v - cf{vmin, vmin}) * cf{mult, mult};
    unsigned ux, uy;
    asm("v_cvt_u32_f32 %0, %1" : "=v"(ux) : "v"(t.x));
    asm("v_cvt_u32_f32 %0, %1" : "=v"(uy) : "v"(t.y));
    const unsigned last = (unsigned)(n_colors - 1);
    ix = (int)(ux < last ? ux : last);
    iy = (int)(uy < last ? uy : last);
}
// host side of the above: true if color_index2_fast equals color_index for every float v of this range (float32 arithmetic as on the device)
inline bool cmap_is_fast(float vmin, float vmax, float top, float mult, int n_colors) {
    if (!(mult > 0.0f) || !(vmax > vmin) || n_colors < 1) return false;
    auto raw = [&](float v) -> long long {   // u32((v - vmin) * mult), saturating like v_cvt_u32_f32
        volatile float d = v - vmin;
        volatile float t = d * mult;
        const float tt = t;
        if (!(tt > 0.0f)) return 0;
        return tt >= 4294967296.0f ? 4294967295ll : (long long)tt;
    };
    const long long last = n_colors - 1;
    // v >= vmax is replaced by top (then clamped from below): its index must be the last colour, and so must the unreplaced one of every
    // v >= vmax (the smallest is vmax itself; monotone from there)
    const float topc = top < vmin ? vmin : top;
    return raw(topc) >= last && raw(vmax) >= last;
}

template <int MIXOP>
__device__ __forceinline__ float mix_combine(float acc, float pw) {
    if constexpr (MIXOP == 0) return acc + pw;                  // AbsMean / Sum: m_powerfinal += m_power[cc]  (:72)
    else if constexpr (MIXOP == 1) return pw > acc ? pw : acc;  // Max (:81)
    else if constexpr (MIXOP == 2) return pw < acc ? pw : acc;  // Min (:89)
    else return pw;                                             // one channel per column: 0 + pw == pw exactly
}

// the same on a pair (bin k, bin M - k)
template <int MIXOP>
__device__ __forceinline__ cf mix_combine2(cf acc, cf pw) {
    if constexpr (MIXOP == 0) return acc + pw;
    else if constexpr (MIXOP == 1) return cf{pw.x > acc.x ? pw.x : acc.x, pw.y > acc.y ? pw.y : acc.y};
    else if constexpr (MIXOP == 2) return cf{pw.x < acc.x ? pw.x : acc.x, pw.y < acc.y ? pw.y : acc.y};
    else return pw;
}

#define GETREG_HW_ID ((32 - 1) << 11 | 4)   // s_getreg_b32 hwreg(HW_REG_HW_ID, 0, 32)

// MIXOP: 0 sum over a channel range (AbsMean, Sum), 1 max, 2 min, 3 exactly one channel per column (mono, Left, Right,
// per-channel): the channel bookkeeping folds away and every iteration ends in the store epilogue, which makes the
// number of vector-memory instructions per iteration a compile-time fact.  (Rounds 2-5 stated here that the s_waitcnt in
// front of the next frame's data then counts past the younger column stores.  It does not: the loop header merges the
// prologue's edge -- loads with no younger stores -- and the steady-state wait is vmcnt(0) in every instantiation.  With
// the first round peeled it becomes a counted wait, and the dispatch takes the same time: the stores' acknowledgements are
// back from L2 by then -- tools/experiments/r06_peel_first_round.txt.)
//
// JSG_NO_LDS_MERGE: the backend's load/store optimizer fuses pairs of 8-byte LDS accesses into ds_read2_b64 /
// ds_read2st64_b64.  On gfx950 a ds_read_b64 is serviced as 2 x 32 lanes (2 LDS cycles, 256 B/clk) but a ds_read2_b64 as
// two accesses of 4 x 16 lanes (8 cycles, 128 B/clk) -- MI355X_MICROARCH.md, LDS table -- so the fused form halves the
// read bandwidth of the exchange reads, on the busiest shared pipe of this kernel.  The pass is switched off for this
// kernel only.
#if defined(__HIP_DEVICE_COMPILE__)
#define JSG_NO_LDS_MERGE __attribute__((target("no-load-store-opt")))
#else
#define JSG_NO_LDS_MERGE
#endif
// OUTK: 0 = the column is stored as floats (dB, or linear power), 1 = as 8-bit palette indices (fused display path:
// the dB value never goes to memory; reference Spectrogram.cpp:632-648 consumes the column it has just produced),
// 2 = the workgroup colours its columns itself (one-wavefront-per-frame plans): the palette indices of the TPB consecutive
// columns of an iteration are parked in the (then idle) exchange regions, and after a workgroup barrier all waves write
// ARGB image rows -- runs of TPB pixels (32 bytes for the eight-frame workgroups) per row.  Neither the dB column nor an index
// column goes to memory: per C5 column 4096 B in + 8196 B out, the algorithmic bytes.
// Progress marks of the display kernels (OUTK == 2), eight per FFT round: a wave lowers its issue priority as it advances (3, 2, 1, 0,
// cyclically), so of the two waves that share a SIMD the one that is BEHIND is served first.  Without them the arbiter prefers the older
// wave for the whole iteration, and that wave then idles at the workgroup barrier of the store phase (in-kernel stamps, tools/abbench
// AB_IMGSTAMPS: the median wave waited there for 20 % of a one-image launch, 7 % with the marks; C5 image in order 21.4 vs 22.0 us,
// 30 000 columns in one launch 226 vs 235 us; the strided batches of the bench are level, 14.3 us per image either way).
#define JSG_MARK(k) do { if constexpr (OUTK == 2) __builtin_amdgcn_s_setprio(3 - ((k) & 3)); } while (0)
// STREAM (jsg_stft_db_launch_strided: K independent batches of one geometry in ONE launch, OUTK == 0): 0 = one batch per launch;
// 1 = the workgroups walk through the groups (TPB consecutive columns) of all batches, everything else as for one batch: tables loaded
// once per workgroup and launch, the prefetch pipeline alive across batches, no ramp-up and drain per 4096 frames.
// (Round 4 also built STREAM == 2, "staged": one persistent 16-wave workgroup per CU whose input spans travel through LDS by LDS-DMA,
// every sample once.  Bit-identical, and level with or behind this form everywhere it was measured -- its sixteen lock-stepped waves
// per CU transform at most 1.0e9 frames/s where the 24 independent ones of this form reach 1.6e9; tools/experiments/r04_staged_span_kernel.patch,
// DESIGN.md section 6.)
// XLOG (jsg_stft_args.exact_log): 0 = 10*log10 on the hardware log unit (v_log_f32: 1 ulp, not specified bit for bit); 1 = by jsg_exact_db of
// jsg_exact_math.h (plain float32 arithmetic shared with the CPU mirror: the dB values, the palette indices and the ARGB pixels are then
// reproducible bit for bit -- reference Spectrogram.cpp:107 feeding CColorpalette.h:32-47).  A separate INSTANTIATION, not a run-time
// branch: as a branch (round 4) the routine moved the register allocation of every kernel (+3 VGPRs, spills in the index-out forms), so
// the default path does not know it exists.  Round 4 ran it as a second elementwise pass over the columns of a launch (one more read and
// write of the output, and no display path); since round 5 it sits in the epilogue of every output form.
template <class C, int MIXOP, int OUTK = 0, int STREAM = 0, int XLOG = 0>
// (RTAB instantiations: five waves per SIMD -- what the plan's LDS allows -- instead of the plan's C::WPS, see Cfg::RTAB.  Strided dispatches
// only: a single launch of one 4096-frame batch gives every wave ONE frame, and reading the tables into registers first costs it 1-1.5 %
// -- 5.33 vs 5.25 us, tools/single_launch_probe.py.)
#define JSG_RTAB_OF(C, MIXOP, OUTK) (C::RTAB && MIXOP == 3 && OUTK == 0 && STREAM != 0)
__global__ __launch_bounds__(C::WPB * 64, (JSG_RTAB_OF(C, MIXOP, OUTK) ? 5 : C::WPS)) JSG_NO_LDS_MERGE void stft_db_kernel(
    // Everything the table and frame loads depend on sits in the first 16 dwords of the kernel arguments: with
    // -amdgpu-kernarg-preload-count=16 those arrive in SGPRs with the wave, so the loads are issued without waiting for a
    // scalar-memory round trip to the kernarg segment (4096 waves starting at once queue up on the scalar cache:
    // measured 0.47 us median / 1.1 us p90 from wave start to the first frame load before this ordering, tools/abbench
    // --stamps).  The rest (output geometry, mix scale) is fetched by s_load while the frame is in flight.
    const float* __restrict__ k_in, const long long k_in_pitch, const float2* __restrict__ k_tab, const unsigned k_n_frames,
    const unsigned k_first_frame, const int k_hop, const int k_flags, const unsigned k_c_range, const int k_iters,
    const unsigned k_nblk, const int k_feedblocks, const StftKArgs a_rest) {   // 14 dwords are preloaded; k_c_range = c_begin | c_end << 16
    StftKArgs a = a_rest;
    a.in = k_in; a.in_pitch = k_in_pitch; a.n_frames = k_n_frames; a.first_frame = k_first_frame; a.hop = k_hop;
    a.iters = k_iters; a.regular = k_flags & 1; a.per_channel = (k_flags >> 1) & 1; a.c_begin = int(k_c_range & 0xffffu); a.c_end = int(k_c_range >> 16);
    a.xcd_remap = (k_flags >> 2) & 1; a.chunked = (k_flags >> 3) & 1; a.tab = k_tab; a.feedblocks = k_feedblocks;
    // Two distinct LDS objects on purpose: the (read-only) lane tables and the exchange buffers.  With one object
    // the compiler must assume that a table read may alias an exchange store and serialises them.
    __shared__ __attribute__((aligned(16))) cf s_tab[C::TLOC == 1 ? C::TAB_ELEMS : 2];
    __shared__ int s_lut[OUTK == 2 ? 256 : 1];
    static_assert(OUTK != 2 || (C::L == 64 && C::FPW == 1 && C::LDS_TOTAL + 1024 <= 160 * 1024), "single-kernel display path: one wavefront per frame");
    constexpr bool BAT = STREAM != 0;   // rows (batches, or batch x channel) numbered through the launch
    // STREAM == 2 (round 6, "runs"; 50 % overlap, one channel per column, one wavefront per frame): a wavefront transforms kRunLen CONSECUTIVE
    // columns of a row one after the other, and the upper half of a frame's raw samples -- which IS the lower half of the next frame -- stays
    // in its registers: from the second column of a run on, only the new hop is loaded (P/2 loads instead of P; the other half used to come
    // from L1 / L2: every sample was loaded twice).  A workgroup step is a super-group of TPB * kRunLen consecutive columns (wave w: columns
    // w * kRunLen + t, t = 0 .. kRunLen - 1).  The two halves trade roles from round to round (the registers of the consumed lower half take
    // the next round's upper half), so the round loop is unrolled by two and nothing is copied.  Same arithmetic, same bits.
    constexpr bool RUNS = STREAM == 2;
    constexpr int RL = kRunLen;
    static_assert(STREAM == 0 || STREAM == 1 || STREAM == 2, "see above");
    static_assert(!RUNS || (MIXOP == 3 && OUTK == 0 && C::L == 64 && C::FPW == 1 && !C::PAIR && (RL & (RL - 1)) == 0 && RL % 2 == 0), "runs: one-channel dB kernels of the 64-lane plans");
    static_assert(!BAT || OUTK == 0, "strided multi-batch launches write dB / power columns");
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int L = C::L, P = C::P, M = C::M, R1 = C::R1, R2 = C::R2, R3 = C::R3, F = C::FPW;
    constexpr int U1 = C::U1, U2 = C::U2, U3 = C::U3;
    typedef float v4f __attribute__((ext_vector_type(4)));

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: keeps task math scalar
    // frame slot inside the workgroup and lane index inside the frame (L lanes cooperate on one frame); a wavefront of an
    // L <= 64 plan owns the SUB * F consecutive slots from slot0 on: slot0 + sub * F + f
    const int ll = L <= 64 ? lane % L : (wave % C::WPF) * 64 + lane;
    const int sub = L <= 64 ? lane / L : 0;
    // (pair plan: a wavefront owns ONE column -- task slot = wave -- and both of its half-waves' exchange regions, 2 wave + sub)
    const int slot0 = C::PAIR ? wave : (L <= 64 ? wave * C::SUB * F : wave / C::WPF);   // wave-uniform part of the slot
    const int tsub = C::PAIR ? 0 : sub;                                                   // the lane's frame among the wave's tasks
    cf* const lds0 = reinterpret_cast<cf*>(smem_raw) + (C::PAIR ? wave * 2 + sub : slot0 + sub * F) * C::LDS_ELEMS;   // frame f: lds0 + f * LDS_ELEMS
    const int tl = ll;                                            // index into a lane-table row (TL = L entries)

    // XCD-aware block remap (bijective): blocks b, b+8, b+16.. share an XCD (and its L2); neighbouring workgroups are given to the
    // same XCD so that the overlapped halves of neighbouring frames hit the same L2 -- in CHUNKS of 32 workgroups that alternate between
    // the XCDs (logical blocks [256 s + 32 x, +32) belong to XCD x), so that the eight XCDs together still sweep one contiguous window
    // of the stream.  (Rounds 2-4 gave every XCD one contiguous EIGHTH of the grid, i.e. eight sweeps 1/8 of the launch apart: the C2
    // dispatch 0.599 -> 0.606-0.610 of 8 TB/s with chunks of 1..64, 0.602 with 128; a grid of 256 -- the "B" plans -- is the same either
    // way.)  The blocks behind the last whole 256 (all of them in a grid below 256, e.g. the 253 workgroups of a C5 dispatch) share
    // one contiguous range per XCD among themselves, as every grid did before: left to their own numbers, neighbouring workgroups land on
    // different XCDs and the 32-byte pieces of an image row line no longer meet in one L2 (C5: traffic 1.10 -> 1.31 x, 584 -> 646 us).
    const unsigned nblk = k_nblk, b = blockIdx.x;   // == gridDim.x (a hidden kernel argument: would cost an s_load)
    const unsigned xcd = b & 7, jb = b >> 3, full = nblk & ~255u;
    unsigned lb = b;
    if (a.xcd_remap) {
#ifndef JSG_X_XCD_CHUNK_LOG2
#define JSG_X_XCD_CHUNK_LOG2 5   // (variant builds sweep this: tools/README.md)
#endif
        constexpr unsigned CL = JSG_X_XCD_CHUNK_LOG2;   // chunks of 2^CL workgroups per XCD; a "row" of eight chunks = 8 << CL logical blocks
        if (b < full && b < (nblk & ~((8u << CL) - 1u))) lb = ((jb >> CL) << (CL + 3)) + (xcd << CL) + (jb & ((1u << CL) - 1u));
        else if (b < full) lb = b;   // (between the last whole row of chunks and the last whole 256: identity)
        else {
            const unsigned t = nblk - full, q = t >> 3, r = t & 7;   // (full is a multiple of 8: block b - full sits on XCD b & 7 too)
            lb = full + (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + ((b - full) >> 3);
        }
    }

    // ---- task bookkeeping (32-bit, scalar): task t of this launch is frame t; the channel range is fixed ----
    // Grid-stride traversal: in iteration `it` the workgroups cover one contiguous window of gridDim.x*TPB frames that
    // sweeps through the stream (neighbouring workgroups touch neighbouring DRAM pages at the same time), instead
    // of every workgroup streaming through a private region (thousands of concurrent streams: 10-15 % less HBM
    // bandwidth, see tools/copy_width_probe.py).  variant builds: JSG_TRAVERSAL=chunk selects the old order.
    const unsigned task_stride = a.chunked ? C::TPB : nblk * C::TPB;
    const unsigned task0 = (a.chunked ? lb * (unsigned)a.iters * C::TPB : lb * C::TPB) + slot0;   // first task of this wave
    // (STREAM: the channel of a per-channel row comes from the row number, see row_of; c0 / c1 then only give nc = 1)
    const int c0 = a.per_channel ? (BAT ? 0 : (int)blockIdx.y) : a.c_begin;
    const int c1 = a.per_channel ? (BAT ? 1 : (int)blockIdx.y + 1) : a.c_end;
    constexpr bool ONE = MIXOP == 3;
    // (pair plan: one FFT round transforms the channel PAIR c, c + 1; the launcher selects it for an even channel count only)
    const int nc = ONE ? 1 : (C::PAIR ? (c1 - c0) / 2 : c1 - c0);   // FFT rounds per column
    constexpr int CSTEP = C::PAIR ? 2 : 1;                            // channels per round
    __builtin_assume(a.iters >= 1 && nc >= 1);   // (the launcher guarantees it) keeps the first frame's loads unconditional
    const int n_fft = a.iters * nc;   // FFT rounds this wave performs (F frames each), s = it*nc + (c - c0) / CSTEP
    // task (= frame of the launch) of frame f of this lane in iteration `it`; tasks past the end are given the last frame
    // again: they hold the same bits as that frame's own lanes and store them to the same column, so nothing is masked
    // OUTK == 2: group (= workgroup iteration) `it` of this workgroup -> image and first column inside that image (all scalar).  An
    // iteration past the last group (last round of the launch) repeats the last group and is not stored (`live` in the store phase).
    // STREAM: the same with rows in the place of images.
    auto group_of = [&](unsigned it, unsigned& image, unsigned& col0) -> bool {
        unsigned g = (task0 - slot0) / C::TPB + it * (task_stride / C::TPB);
        const bool inside = g < a.n_groups;
        // steps past the end (the last round of a launch whose step count is not a multiple of the grid): the display form repeats the
        // last group and does not store it; the strided dB form starts over with the first groups -- recomputed and stored a second
        // time with the same bits, but every surplus workgroup at a DIFFERENT place (all of them on the last group: hundreds of
        // workgroups storing into the same eight columns, measured 0.44 instead of 0.58 of 8 TB/s with 9 workgroups per CU)
        if (!inside) g = BAT ? g - a.n_groups : a.n_groups - 1;
        image = (unsigned)(((unsigned long long)g * a.img_magic) >> 40);
        col0 = (g - image * a.img_gpi) * C::TPB;
        return inside;
    };
    // row -> offsets of its input and of its ring: batch * stride (+ channel * pitch in per-channel mode)
    auto row_of = [&](unsigned row, long long& in_off, long long& out_off) {
        unsigned batch = row, ch = 0;
        if (a.per_channel) {
            batch = (unsigned)(((unsigned long long)row * a.bat_magic) >> 40);
            ch = row - batch * a.bat_cpb;
        }
        in_off = (long long)batch * a.in_image_stride + (long long)ch * a.in_pitch;
        out_off = (long long)batch * a.out_batch_stride + (long long)ch * a.out_cpitch;
    };
    // CARRY (round 6; strided one-channel kernels): a round's row, first column and ring offset are computed ONCE, when its loads are issued
    // (frame_src, one round ahead), and carried in scalar registers to its epilogue a round later -- which used to recompute group_of / row_of
    // (two 64-bit magic multiplies, the batch and channel offsets): 156 scalar instructions per frame in the loop of the C2 kernel against 67 in
    // the single-batch kernel.  Same values, same bits.
#ifdef JSG_X_NO_CARRY   // (variant builds: the rounds 4-5 form, for A/B)
    constexpr bool CARRY = false;
#else
    constexpr bool CARRY = BAT && MIXOP == 3 && !C::PAIR && F == 1;
#endif
    unsigned nx_row = 0, nx_col0 = 0, cu_row = 0, cu_col0 = 0;   // nx_*: of the round whose loads were issued last; cu_*: of the round being transformed
    long long nx_out = 0, cu_out = 0;
    auto task_of = [&](unsigned it, int f) -> unsigned {
        unsigned t;
        if constexpr (OUTK == 2 || BAT) {
            unsigned image, col0;
            group_of(it, image, col0);
            t = col0 + slot0 + tsub * F + f;
        } else t = task0 + it * task_stride + tsub * F + f;
        return t < a.n_frames ? t : a.n_frames - 1;
    };
    auto frame_src = [&](int s, int f) -> const f2u* {
        const unsigned it = (nc == 1) ? (unsigned)s : (unsigned)s / (unsigned)nc;
        const int c = c0 + CSTEP * (s - (int)it * nc);
        const unsigned j = a.first_frame + task_of(it, f);
        long long start;
        if (a.regular) {
            start = (long long)j * a.hop;
        } else {   // the reference's perc10: every fft-size block restarts at offset 0 (Spectrogram.cpp:50-55,216)
            const unsigned blk = j / (unsigned)a.feedblocks;
            start = (long long)blk * C::N + (long long)(j - blk * a.feedblocks) * a.hop;
        }
        if constexpr (OUTK == 2) {
            unsigned image, col0;
            group_of(it, image, col0);
            start += (long long)image * a.in_image_stride;
        }
        if constexpr (BAT) {
            unsigned row, col0;
            long long in_off, out_off;
            group_of(it, row, col0);
            row_of(row, in_off, out_off);
            start += in_off;
            if constexpr (CARRY) { nx_row = row; nx_col0 = col0; nx_out = out_off; }
        }
        if constexpr (C::PAIR)   // the lane's samples of BOTH channels: x_c[start + 64 m + 2 ll + h] (see frame_load)
            return reinterpret_cast<const f2u*>(a.in + (long long)c * a.in_pitch + start + (2 * ll + sub));
        else
        return reinterpret_cast<const f2u*>(a.in + (long long)c * a.in_pitch + start) + ll;
    };
    // the P loads of one frame (pair plan: 2 P dword loads -- the sample of channel c into .x, of channel c + 1 into .y: the lane holds
    // z[n] = x_c[n] + i x_(c+1)[n] for n = 2 (ll + 32 m) + h, i.e. half-wave h the samples of parity h; a wave instruction reads 64
    // consecutive floats)
    // part: 0 = all of them, 1 / 2 = the first / second half of the pair plan's loads (see the prefetch in process())
    auto frame_load = [&](f2u (&dst)[P], const f2u* src, int part = 0) {
        if constexpr (C::PAIR) {
            const float* p1 = reinterpret_cast<const float*>(src);
            const float* p2 = p1 + a.in_pitch;
#pragma unroll
            for (int m = 0; m < P; ++m) {
                if (part == 1 && m >= P / 2) continue;
                if (part == 2 && m < P / 2) continue;
                dst[m].x = p1[64 * m];
                dst[m].y = p2[64 * m];
            }
        } else {
#pragma unroll
            for (int m = 0; m < P; ++m) dst[m] = src[L * m];
        }
    };
    // ---- runs (STREAM == 2): the wave's run of RL consecutive columns of one row ----
    f2u rw[2][RUNS ? P / 2 : 1];          // the two halves of the raw frame; in round t of a run rw[t & 1] is the lower half, rw[(t & 1) ^ 1] the upper
    unsigned rn_col0 = 0;                 // first column (inside its row) of this wave's current run
    long long rn_in = 0;                  // input offset of the run's row
    bool nx_live = true, cu_live = true;  // the round's column exists (columns past the end of a row are transformed on whatever the clamped
                                          // loads bring and NOT stored; the other kernels repeat the row's last column instead)
    // a new run: super-group `sg` of this workgroup -> row, first column of this wave's run, the row's input / ring offsets (scalar, once per RL rounds)
    auto runs_group = [&](unsigned sg) {
        unsigned g = lb + sg * nblk;
        if (g >= a.n_groups) g -= a.n_groups;                     // surplus steps of the last round start over with the first groups
        const unsigned row = (unsigned)(((unsigned long long)g * a.img_magic) >> 40);
        rn_col0 = (g - row * a.img_gpi) * (unsigned)(C::TPB * RL) + (unsigned)(slot0 * RL);
        row_of(row, rn_in, nx_out);
        nx_row = row;
    };
    // where column tu of the run's row starts (a column past the row's end: its last column, so that every load stays inside the row)
    auto runs_src = [&](unsigned tu) -> const f2u* {
        nx_live = tu < a.n_frames;
        nx_col0 = nx_live ? tu : a.n_frames - 1;
        const long long start = (long long)(a.first_frame + nx_col0) * a.hop + rn_in + (long long)c0 * a.in_pitch;
        return reinterpret_cast<const f2u*>(a.in + start) + ll;
    };
    // ---- lane tables first: two 16-byte LDS-DMA pieces per thread bring the tables from L2 straight into LDS (wave-uniform
    //      base + lane * 16 bytes, no VGPR round trip).  They are issued AHEAD of the frame loads: every wave's FFT start is
    //      gated by the workgroup barrier behind the tables, so they are the latency-critical load (frame loads first, or
    //      half of them first, measured 0.3-0.5 us slower per C2 launch) ----
    constexpr int NTL = C::TLOC == 1 ? (C::TAB_ELEMS / 2 + C::WPB * 64 - 1) / (C::WPB * 64) : 1;   // 16-byte pieces per thread
    constexpr bool TAB_EVEN = (C::TAB_ELEMS / 2) % (C::WPB * 64) == 0;
    if constexpr (C::TLOC == 1) {
        const v4f* g4 = reinterpret_cast<const v4f*>(a.tab) + threadIdx.x;
        char* sbase = reinterpret_cast<char*>(s_tab) + wave * 1024;
#pragma unroll
        for (int i = 0; i < NTL; ++i)
            if (TAB_EVEN || i + 1 < NTL || (int)threadIdx.x + i * C::WPB * 64 < C::TAB_ELEMS / 2)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g4 + i * C::WPB * 64),
                                                 (__attribute__((address_space(3))) void*)(sbase + i * C::WPB * 1024), 16, 0, 0);
        __builtin_amdgcn_sched_barrier(0);   // pin the order of the table pieces against the frame loads (in-order vmcnt)
    }
    // ---- issue the loads of the first FFT round (software pipeline, one round ahead) ----
    f2u raw[F][P];
    if constexpr (RUNS) {   // round 0: the whole frame, lower half into rw[0], upper half into rw[1] (P loads, as the other kernels: the counted wait below)
        runs_group(0u);
        const f2u* src = runs_src(rn_col0);
#pragma unroll
        for (int m = 0; m < P / 2; ++m) rw[0][m] = src[L * m];
#pragma unroll
        for (int m = 0; m < P / 2; ++m) rw[1][m] = src[L * (m + P / 2)];
        __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
        for (int f = 0; f < F; ++f) frame_load(raw[f], frame_src(0, f));
        __builtin_amdgcn_sched_barrier(0);
    }
    const cf* tBase;   // start of the tables (LDS copy, or global memory for TLOC == 0)
    if constexpr (C::TLOC == 1) {
        // the table pieces are older than the F * P frame loads of this wave: a counted vmcnt retires them and leaves the
        // frames in flight (an LDS-DMA is a pending LDS write on the VM counter; __syncthreads() would drain everything)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // (compiler only) the frame loads stay above the counted wait
        // (pair plan: 2 P = 64 frame loads, one more than the 6-bit counter can name: the wait then also retires the oldest frame load)
        constexpr int NFL = (C::PAIR ? 2 : 1) * F * P;
        asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(NFL > 63 ? 63 : NFL) : "memory");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        tBase = s_tab;
    } else {
        tBase = reinterpret_cast<const cf*>(a.tab);
    }
    if constexpr (OUTK == 2) {   // colour table into LDS (n_colors <= 256); it is read after the first workgroup barrier of the store phase
        if (threadIdx.x < 256) s_lut[threadIdx.x] = a.lut[(int)threadIdx.x < a.n_colors ? (int)threadIdx.x : a.n_colors - 1] | (int)0xFF000000u;   // opaque: pixel = LUT | 0xFF000000
    }
    cf twA[C::TWF ? U2 : 1], twC = {0.f, 0.f};           // TWF: this lane's constant factors (see Cfg)
    int twBrow = 0;
    if constexpr (C::TWF) {
#pragma unroll
        for (int v = 0; v < U2; ++v) twA[v] = tBase[C::TAB_A + v * C::TL + tl];
        twC = tBase[C::TAB_POST + tl];
        twBrow = C::TAB_TW2 + (ll % R3) * C::TSB;
    }
    const cf* const tTw1 = tBase + C::TAB_TW1;       // compact: [n2][TS1], not indexed by lane
    int tw1row[U1];                                   // row of this lane's butterfly u in the stage-1 table
#pragma unroll
    for (int u = 0; u < U1; ++u) tw1row[u] = ((ll + L * u) / R3) * C::TS1;
    // RTAB: stage-1 row and post-pass twiddles of this lane, read once (the tables are in LDS behind the barrier above)
    constexpr bool RTAB = JSG_RTAB_OF(C, MIXOP, OUTK);
    cf rt1[RTAB ? R1 : 1], rpost[RTAB ? P / 2 : 1];
    if constexpr (RTAB) {
        static_assert(!RTAB || (U1 == 1 && !C::TWF && !C::PAIR && C::TLOC == 1), "register tables: one stage-1 butterfly per lane, whole tables in LDS");
#pragma unroll
        for (int k1 = 1; k1 < R1; k1 += 2) {
            const v4f q4 = *reinterpret_cast<const v4f*>(tTw1 + tw1row[0] + k1 - 1);
            rt1[k1] = cf{q4.x, q4.y};
            if (k1 + 1 < R1) rt1[k1 + 1] = cf{q4.z, q4.w};
        }
#pragma unroll
        for (int rho = 0; rho < P / 2; rho += 2) {
            const v4f q4 = *reinterpret_cast<const v4f*>(tBase + C::TAB_POST + C::tab_idx(rho, tl));
            rpost[rho] = cf{q4.x, q4.y};
            rpost[rho + 1] = cf{q4.z, q4.w};
        }
    }
    // values j and j + 1 (j even) of this lane from the [J][TL] table at `off`: one 16-byte read (pair layout)
    auto tab2 = [&](int off, int j, cf& a0, cf& a1) {
        const v4f q4 = *reinterpret_cast<const v4f*>(tBase + off + C::tab_idx(j, tl));
        a0 = cf{q4.x, q4.y};
        a1 = cf{q4.z, q4.w};
    };
    // exchange synchronisation: lock-step lanes of one wave need only a compiler fence; frames that span several
    // waves (L > 64) need the workgroup barrier (every wave of the workgroup runs the same number of them)
    auto frame_sync = [&]() {
        if constexpr (L > 64) __syncthreads();
        else wave_sync();
    };

    // per-lane LDS element offsets of the two exchanges
    constexpr bool TWO = C::TWO_STAGE;
    int e1r[U2], e2w[TWO ? 1 : U2], e2r[TWO ? 1 : U3];
#pragma unroll
    for (int v = 0; v < U2; ++v) {
        const int t2 = ll + L * v;
        e1r[v] = (t2 / R3) * C::S1 + (t2 % R3);
        if constexpr (!TWO) e2w[v] = (t2 / R3) * C::AX + (t2 % R3) * C::AZ;
    }
    if constexpr (!TWO) {
#pragma unroll
        for (int w = 0; w < U3; ++w) {
            const int t3 = ll + L * w;
            e2r[w] = (t3 % R1) * C::AX + (t3 / R1) * C::AY;
        }
    } else {
        e2w[0] = e2r[0] = 0;
    }

    // pair plan: (sum of Re^2, sum of Im^2) over the channel pairs of the column, of Z[k] (accU) and Z[k + M] (accV), k = lane + 64 j
    cf accU[C::PAIR ? P / 2 : 1], accV[C::PAIR ? P / 2 : 1];
    if constexpr (C::PAIR) {
#pragma unroll
        for (int j = 0; j < P / 2; ++j) accU[j] = accV[j] = cf{0.f, 0.f};
    }
    cf acc[F][P / 2];   // .x: bin k = ll + L rho of the lower half, .y: its mirror M - k (filled as pairs by the post pass)
    float accNy[F];
    constexpr float init = (MIXOP == 2) ? 1000000.0f : 0.0f;   // reference Spectrogram.cpp:69,78,86
#pragma unroll
    for (int f = 0; f < F; ++f) {
#pragma unroll
        for (int m = 0; m < P / 2; ++m) acc[f][m] = cf{init, init};
        accNy[f] = init;
    }

    // One FFT round of the sequence (F frames of one channel): consumes `raw` (loaded one round ago), re-issues it for
    // round s+1, transforms, accumulates |X|^2 into acc, and after the last channel of a column runs the mix epilogue + ring
    // store.  `last_tag` (std::true_type): the peeled final round of the wave, which prefetches nothing.  Peeling keeps
    // the steady-state loop free of the "did the prefetch happen" merge (register moves and a full vmcnt(0)).
    // Every stage loops over the F frames INSIDE the stage, in one basic block: the scheduler interleaves the frames'
    // independent chains, and each table value is fetched once for all of them.
    auto process = [&](int s, auto last_tag, auto t_tag) __attribute__((always_inline)) {
        constexpr bool LAST = decltype(last_tag)::value;
        constexpr int T = decltype(t_tag)::value;       // runs: the round's place in its run (0 otherwise)
        constexpr int PAR = T & 1;                       // runs: which of rw[0] / rw[1] holds the lower half of this round's frame
        if constexpr (CARRY) { cu_row = nx_row; cu_col0 = nx_col0; cu_out = nx_out; }   // (before the prefetch below overwrites nx_*)
        if constexpr (RUNS) cu_live = nx_live;
        cf x[F][P];
        JSG_MARK(0);
        // ---- window multiply: register m = u + U1 n1 is input n1 of stage-1 butterfly u.  The upper inputs (n1 >= R1 / 2) are multiplied
        //      here; the lower ones stay raw and keep their window values, their products are fused into the butterfly (dft_win) ----
        cf wlo[P / 2];   // window values of the lower inputs, index (m % U1) + U1 * n1 == m for m < P / 2
        if constexpr (C::PAIR) {
            // one real window value per complex sample z = x_c + i x_(c+1): four values of this lane per 16-byte read ([P/4][64][4] floats)
            static_assert(F == 1, "pair plan");
            // (the values are used as pairs (w[m], w[m + 1]): see PairFirstLayer, which does the multiplies)
        } else {
#pragma unroll
        for (int m = 0; m < P; m += 2) {
            cf w0, w1;
            tab2(C::TAB_WIN, m, w0, w1);
            static_assert((P / 2) % 2 == 0 || P == 2, "the window pairs do not straddle the halves");
#pragma unroll
            for (int f = 0; f < F; ++f) {
                if constexpr (RUNS) {
                    if (m < P / 2) {
                        x[f][m] = to_cf(rw[PAR][m]);
                        x[f][m + 1] = to_cf(rw[PAR][m + 1]);
                    } else {
                        x[f][m] = to_cf(rw[PAR ^ 1][m - P / 2]) * w0;
                        x[f][m + 1] = to_cf(rw[PAR ^ 1][m + 1 - P / 2]) * w1;
                    }
                } else
                if (m < P / 2) {     // (m < P/2 <=> n1 < R1/2: P/2 = U1 * R1/2)
                    x[f][m] = to_cf(raw[f][m]);
                    x[f][m + 1] = to_cf(raw[f][m + 1]);
                } else {
                    x[f][m] = to_cf(raw[f][m]) * w0;
                    x[f][m + 1] = to_cf(raw[f][m + 1]) * w1;
                }
            }
            if (m < P / 2) { wlo[m] = w0; wlo[m + 1] = w1; }
        }
        }
        // pair plan: the first butterfly layer of stage 1 runs HERE, ahead of the prefetch: it consumes every raw value, so the loads below
        // can land in the registers they leave.  (Left to the scheduler, the raw lower inputs -- which the other plans' register allocation
        // keeps through the prefetch for free -- were copied first: 38 v_mov per round in the ISA.)
        // EARLY1: sums and differences of the first layer for every butterfly of the lane, register m = u + U1 J (J < R1 / 2), ahead of the loads
        cf fa[C::EARLY1 ? F : 1][C::EARLY1 ? P / 2 : 1], fd[C::EARLY1 ? F : 1][C::EARLY1 ? P / 2 : 1];
        if constexpr (C::EARLY1) {
            static_assert(!C::EARLY1 || !C::PAIR, "the pair plan has its own early first layer");
#pragma unroll
            for (int f = 0; f < F; ++f)
#pragma unroll
                for (int m = 0; m < P / 2; ++m) {
                    const cf hi = x[f][m + P / 2];
                    fa[f][m] = __builtin_elementwise_fma(x[f][m], wlo[m], hi);
                    fd[f][m] = __builtin_elementwise_fma(x[f][m], wlo[m], -hi);
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        cf pa[C::PAIR ? R1 / 2 : 1], pb[C::PAIR ? R1 / 2 : 1];
        if constexpr (C::PAIR) {
            static_assert(!C::PAIR || U1 == 1, "pair plan: one radix-R1 butterfly per lane");
            const float* wtab = reinterpret_cast<const float*>(tBase + C::TAB_WIN) + lane * 4;
            cf wp[P / 2], zr[P];
#pragma unroll
            for (int m = 0; m < P; m += 4) {
                const v4f w4 = *reinterpret_cast<const v4f*>(wtab + (m / 4) * 256);
                wp[m / 2] = cf{w4.x, w4.y};
                wp[m / 2 + 1] = cf{w4.z, w4.w};
            }
#pragma unroll
            for (int m = 0; m < P; ++m) zr[m] = to_cf(raw[0][m]);
            PairFirstLayer<R1, 0>::run(zr, wp, pa, pb);
            __builtin_amdgcn_sched_barrier(0);
        }
        // (pair plan: 2 P = 64 dword loads per round, and a wave can have 63 vector-memory instructions outstanding -- the 64th would hold
        // the wave's issue until the first has returned, a full memory round trip per FFT round.  The first half goes out here, the second
        // behind stage 1, when part of the first has arrived.)
        const f2u* next_src = nullptr;
        if constexpr (RUNS) {
            // the next round.  Inside a run only the NEW hop -- the upper half of the next frame -- is loaded, into the registers of this round's
            // lower half (consumed by the time the data arrives); this round's upper half stays where it is and IS the next round's lower half.
            // The last round of a run loads the whole first frame of the workgroup's next super-group.  Which of the two is a compile-time fact.
            if constexpr (T + 1 < RL) {
                const f2u* src = runs_src(rn_col0 + (unsigned)(T + 1));
#pragma unroll
                for (int m = 0; m < P / 2; ++m) rw[PAR][m] = src[L * (m + P / 2)];
            } else if (!LAST) {
                static_assert(!RUNS || RL % 2 == 0, "a run starts with rw[0] as the lower half");
                runs_group((unsigned)s / (unsigned)RL + 1u);
                const f2u* src = runs_src(rn_col0);
#pragma unroll
                for (int m = 0; m < P / 2; ++m) rw[PAR ^ 1][m] = src[L * m];
#pragma unroll
                for (int m = 0; m < P / 2; ++m) rw[PAR][m] = src[L * (m + P / 2)];
            }
        } else
        if (!LAST) {   // the next round's frames travel while this one is transformed
            if constexpr (C::PAIR) {
                next_src = frame_src(s + 1, 0);
                frame_load(raw[0], next_src, 1);
            } else {
#pragma unroll
            for (int f = 0; f < F; ++f) frame_load(raw[f], frame_src(s + 1, f));
            }
        }
        JSG_MARK(1);
        // ---- stage 1: radix-R1 over n1, twiddle W_{R1R2}^{n2 k1}, exchange 1 ----
#pragma unroll
        for (int u = 0; u < U1; ++u) {
            if (u == U1 / 2) JSG_MARK(2);
            cf t[F][R1];
#pragma unroll
            for (int f = 0; f < F; ++f) {
#pragma unroll
                for (int n1 = 0; n1 < R1; ++n1) t[f][n1] = x[f][u + U1 * n1];
                if constexpr (C::PAIR) {   // the rest of dft_win<R1> (its first layer ran above)
                    dft<R1 / 2, false>(pa);
                    dft<R1 / 2, true>(pb);
#pragma unroll
                    for (int q = 0; q < R1 / 2; ++q) {
                        t[f][2 * q] = pa[q];
                        t[f][2 * q + 1] = pb[q];
                    }
                } else if constexpr (C::EARLY1) {
                    cf ea[R1 / 2], ed[R1 / 2];
#pragma unroll
                    for (int J = 0; J < R1 / 2; ++J) {
                        ea[J] = fa[f][u + U1 * J];
                        ed[J] = fd[f][u + U1 * J];
                    }
                    if constexpr (R1 <= 8) dft_rest<R1>(ea, ed, t[f]);
                    else sr_rest<R1>(ea, ed, t[f]);
                } else {
                cf wl[R1 / 2];
#pragma unroll
                for (int n1 = 0; n1 < R1 / 2; ++n1) wl[n1] = wlo[u + U1 * n1];
                sr_dft_win<R1>(t[f], wl);
                }
                lds0[f * C::LDS_ELEMS + ll + L * u] = t[f][0];
            }
#pragma unroll
            for (int k1 = 1; k1 < R1; k1 += 2) {   // row: k1 = 1, 2 | 3, 4 | ... | R1 - 1, (pad)
                v4f q4;
                if constexpr (RTAB) q4 = v4f{rt1[k1].x, rt1[k1].y, rt1[k1 + 1 < R1 ? k1 + 1 : k1].x, rt1[k1 + 1 < R1 ? k1 + 1 : k1].y};
                else q4 = *reinterpret_cast<const v4f*>(tTw1 + tw1row[u] + k1 - 1);
#pragma unroll
                for (int f = 0; f < F; ++f) {
                    lds0[f * C::LDS_ELEMS + k1 * C::S1 + ll + L * u] = cmul(t[f][k1], cf{q4.x, q4.y});
                    if (k1 + 1 < R1) lds0[f * C::LDS_ELEMS + (k1 + 1) * C::S1 + ll + L * u] = cmul(t[f][k1 + 1], cf{q4.z, q4.w});
                }
            }
        }
        JSG_MARK(3);
        if constexpr (C::PAIR) {
            if (!LAST) {
                __builtin_amdgcn_sched_barrier(0);
                frame_load(raw[0], next_src, 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        frame_sync();
#pragma unroll
        for (int f = 0; f < F; ++f)
#pragma unroll
            for (int v = 0; v < U2; ++v) {
                if constexpr (TWO) {   // R3 == 1: the R2 values of a lane are neighbours in its row (S1 even: 16-byte aligned)
#pragma unroll
                    for (int n2 = 0; n2 < R2; n2 += 2) {
                        const v4f q4 = *reinterpret_cast<const v4f*>(lds0 + f * C::LDS_ELEMS + e1r[v] + n2);
                        x[f][v * R2 + n2] = cf{q4.x, q4.y};
                        x[f][v * R2 + n2 + 1] = cf{q4.z, q4.w};
                    }
                } else {
#pragma unroll
                    for (int n2 = 0; n2 < R2; ++n2) x[f][v * R2 + n2] = lds0[f * C::LDS_ELEMS + e1r[v] + n2 * R3];
                }
            }
        frame_sync();
        JSG_MARK(4);
        // ---- stage 2: radix-R2 over n2, twiddle W_M^{n3 (k1 + R1 k2)}, exchange 2 ----
#pragma unroll
        for (int v = 0; v < U2; ++v) {
            cf t[F][R2];
#pragma unroll
            for (int f = 0; f < F; ++f) {
#pragma unroll
                for (int n2 = 0; n2 < R2; ++n2) t[f][n2] = x[f][v * R2 + n2];
                if constexpr (C::PAIR) dft<R2>(t[f]);   // (the pair plan keeps the radix-2 transforms it was measured with: 244 VGPRs, no scratch)
                else sr_dft<R2>(t[f]);
            }
            if constexpr (TWO) {   // n3 = 0: the stage-2 twiddle is 1, and Z[ll + L k2] already sits in register k2 of lane ll
#pragma unroll
                for (int f = 0; f < F; ++f)
#pragma unroll
                    for (int k2 = 0; k2 < R2; ++k2) x[f][v * R2 + k2] = t[f][k2];
            } else {
#pragma unroll
                for (int k2 = 0; k2 < R2; k2 += 2) {
                    cf w0, w1;
                    if constexpr (C::TWF) {
                        const v4f q4 = *reinterpret_cast<const v4f*>(tBase + twBrow + k2);
                        w0 = cmul(cf{q4.x, q4.y}, twA[v]);
                        w1 = cmul(cf{q4.z, q4.w}, twA[v]);
                    } else tab2(C::TAB_TW2, v * R2 + k2, w0, w1);
#pragma unroll
                    for (int f = 0; f < F; ++f) {
                        lds0[f * C::LDS_ELEMS + e2w[v] + k2 * C::AY] = cmul(t[f][k2], w0);
                        lds0[f * C::LDS_ELEMS + e2w[v] + (k2 + 1) * C::AY] = cmul(t[f][k2 + 1], w1);
                    }
                }
            }
        }
        if constexpr (!TWO) {
        frame_sync();
        JSG_MARK(5);
        // ---- stage 3: radix-R3 over n3; Z[k], k = t3 + R1 R2 k3 ----
#pragma unroll
        for (int f = 0; f < F; ++f)
#pragma unroll
            for (int w = 0; w < U3; ++w) {
                if constexpr (C::AZ == 1) {
                    // AZ == 1 (512 points, 4096 "B"): the n3 values of a lane are neighbours, and the IR load/store vectorizer
                    // (not the backend pass that JSG_NO_LDS_MERGE switches off) pairs them into 16-byte loads of 8-byte
                    // alignment = ds_read2_b64: 8 LDS cycles instead of 2 x 2, banked differently from what the layout was
                    // searched for (profile of round 3: 15.7 % of the LDS cycles of the 4096 "B" kernel were bank conflicts).
                    // The odd n3 are therefore read through a second pointer whose relation to the first the optimizer cannot see.
                    typedef const cf __attribute__((address_space(3))) * lds_cf_ptr;   // (an LDS pointer: 32 bits, stays a ds_read)
                    const lds_cf_ptr pe = (lds_cf_ptr)(lds0 + f * C::LDS_ELEMS + e2r[w]);
                    lds_cf_ptr po = pe + 1;
                    asm("" : "+v"(po));
#pragma unroll
                    for (int n3 = 0; n3 < R3; n3 += 2) {
                        x[f][w * R3 + n3] = pe[n3];
                        x[f][w * R3 + n3 + 1] = po[n3];
                    }
                } else {
#pragma unroll
                    for (int n3 = 0; n3 < R3; ++n3) x[f][w * R3 + n3] = lds0[f * C::LDS_ELEMS + e2r[w] + n3 * C::AZ];
                }
            }
        JSG_MARK(6);
        frame_sync();
#pragma unroll
        for (int f = 0; f < F; ++f)
#pragma unroll
            for (int w = 0; w < U3; ++w) {
                cf t[R3];
#pragma unroll
                for (int n3 = 0; n3 < R3; ++n3) t[n3] = x[f][w * R3 + n3];
                sr_dft<R3>(t);
#pragma unroll
                for (int k3 = 0; k3 < R3; ++k3) x[f][w * R3 + k3] = t[k3];
            }
        } else {
            frame_sync();   // the post pass reuses the exchange buffer: keep its stores behind the exchange-1 loads
        }
        if constexpr (C::PAIR) {
            // ---- pair plan, last stage (radix 2 across the half-waves).  Register k2 of half-wave h holds Y_h[ll + 32 k2], the 1024-point
            // transforms of the even (h = 0: E) and odd (h = 1: O) samples of z.  v_permlane32_swap trades the upper half of register 2 j
            // against the lower half of register 2 j + 1: afterwards register 2 j holds E[k] and register 2 j + 1 holds O[k], k = lane + 64 j,
            // over all 64 lanes.  Z[k] = E[k] + W_N^k O[k], Z[k + M] = E[k] - W_N^k O[k]; the table holds W_N^k for j < P/4 and
            // W_N^(k + M/2) = -i W_N^k serves the rest (add_mi / sub_mi: the -i costs nothing).  The powers are accumulated as
            // (sum Re^2, sum Im^2): one packed fma per spectrum value and channel pair.
            constexpr int Q4 = P / 4;
#pragma unroll
            for (int jp = 0; jp < Q4; jp += 2) {
                const v4f q4 = *reinterpret_cast<const v4f*>(tBase + C::TAB_POST + C::pair_tw_idx(jp, lane));
                const cf tw0 = {q4.x, q4.y}, tw1 = {q4.z, q4.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int j = jp + (i & 1) + (i >> 1) * Q4;
                    const auto re = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[0][2 * j].x), __float_as_uint(x[0][2 * j + 1].x), false, false);
                    const auto im = __builtin_amdgcn_permlane32_swap(__float_as_uint(x[0][2 * j].y), __float_as_uint(x[0][2 * j + 1].y), false, false);
                    const cf E = {__uint_as_float(re[0]), __uint_as_float(im[0])};
                    const cf O = cmul(cf{__uint_as_float(re[1]), __uint_as_float(im[1])}, (i & 1) ? tw1 : tw0);
                    const cf Z = (i >> 1) ? add_mi(E, O) : E + O;
                    const cf Zm = (i >> 1) ? sub_mi(E, O) : E - O;
                    accU[j] = __builtin_elementwise_fma(Z, Z, accU[j]);
                    accV[j] = __builtin_elementwise_fma(Zm, Zm, accV[j]);
                }
            }
        } else {
            JSG_MARK(7);
            // ---- paired real-split post pass.  Bin k = ll + L*rho (rho = w + U3*k3).  A lane owns the pairs of its
            // lower registers rho < P/2: (k, M-k); Z[M-k] is the upper register P-1-rho of lane L-ll (lane 0: its own
            // register P-rho, and Z[M] = Z[0]).  With the window pre-scaled by 1/2 and T = (-i W_N^k) (Z[k] - conj Z[M-k]):
            //     X[k] = S + T,   X[M-k] = conj(S - T),   S = Z[k] + conj Z[M-k]
            // The self-paired bin M/2 (lane 0, register rho = P/2) is conj Z[M/2].  acc[j] = (|X[ll + L j]|^2,
            // |X[M - ll - L j]|^2) (j < P/2), accNy = |X[M/2]|^2.  The two powers of a pair are formed side by side: real parts
            // (S.x + T.x, S.x - T.x) and imaginary parts of X[k] and X[M-k] in one packed add each, then one packed multiply
            // and one packed fma give both |.|^2, and the mix accumulates the pair with one packed add.
            auto reg_of = [](int rho) { return (rho % U3) * R3 + rho / U3; };
            cf* const ldsP = lds0 + (C::PSKEW ? (sub & 1) * C::PSKEW : 0);   // (L = 16: see Cfg::PSKEW)
            static_assert(M + 1 + C::PSKEW <= C::LDS_ELEMS, "the skewed post-pass exchange stays inside the frame's region");
#pragma unroll
            for (int f = 0; f < F; ++f) {
#pragma unroll
                for (int rho = P / 2; rho < P; ++rho) ldsP[f * C::LDS_ELEMS + ll + L * rho] = x[f][reg_of(rho)];
                if (ll == 0) ldsP[f * C::LDS_ELEMS + M] = x[f][0];   // Z[M] := Z[0]
            }
            frame_sync();
            cf zq[F][P / 2];
#pragma unroll
            for (int f = 0; f < F; ++f)
#pragma unroll
                for (int rho = 0; rho < P / 2; ++rho) zq[f][rho] = ldsP[f * C::LDS_ELEMS + M - (ll + L * rho)];
            frame_sync();   // the next round's exchange stores must stay behind these loads
            cf wpost[P / 2];
#pragma unroll
            for (int rho = 0; rho < P / 2; rho += 2) {
                if constexpr (C::TWF) {
                    constexpr int Q = 64 / (C::N / L);   // W_(N/L)^rho as a multiple of 2 pi / 64
                    wpost[rho] = cmul_s(twC, cf{kCos64[Q * rho], -kSin64[Q * rho]});
                    wpost[rho + 1] = cmul_s(twC, cf{kCos64[Q * (rho + 1)], -kSin64[Q * (rho + 1)]});
                } else if constexpr (RTAB) { wpost[rho] = rpost[rho]; wpost[rho + 1] = rpost[rho + 1]; }
                else tab2(C::TAB_POST, rho, wpost[rho], wpost[rho + 1]);
            }
#pragma unroll
            for (int f = 0; f < F; ++f) {
#pragma unroll
                for (int rho = 0; rho < P / 2; ++rho) {
                    const cf z = x[f][reg_of(rho)], p = zq[f][rho];
                    const cf S = add_conj(z, p);
                    const cf T = cmul(sub_conj(z, p), wpost[rho]);
                    const cf re = addsub_re(S, T), im = addsub_im(S, T);   // (Re X[k], Re X[M-k]), (Im X[k], -Im X[M-k])
                    cf pw = re * re;
                    pw = __builtin_elementwise_fma(im, im, pw);
                    acc[f][rho] = mix_combine2<MIXOP>(acc[f][rho], pw);
                }
                {   // bin M/2 (held by the frame's lane 0): the window carries 1/2, so |X|^2 = 4 |Z'|^2.  For single-wave
                    // frames the value is broadcast so that every lane can take part in an unmasked store below.
                    cf z = x[f][reg_of(P / 2)];
                    if constexpr (L == 64) {
                        z.x = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(z.x)));
                        z.y = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(z.y)));
                    } else if constexpr (L == 32) {
                        const int ax = __builtin_amdgcn_readlane(__float_as_int(z.x), 0), bx = __builtin_amdgcn_readlane(__float_as_int(z.x), 32);
                        const int ay = __builtin_amdgcn_readlane(__float_as_int(z.y), 0), by = __builtin_amdgcn_readlane(__float_as_int(z.y), 32);
                        z.x = __int_as_float(sub ? bx : ax);
                        z.y = __int_as_float(sub ? by : ay);
                    } else if constexpr (L == 16) {   // lane 0 of every row of 16 lanes to its row: DPP row_newbcast:0
                        z.x = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(z.x), 0x150, 0xf, 0xf, false));
                        z.y = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(z.y), 0x150, 0xf, 0xf, false));
                    }
                    // (the fused form is written out: which of the two products the compiler contracts is its choice otherwise, and
                    // oracle/jsg_mirror.c restates this kernel operation by operation)
                    accNy[f] = mix_combine<MIXOP>(accNy[f], 4.0f * __builtin_fmaf(z.x, z.x, z.y * z.y));
                }
            }
        }

        // ---- last channel of this column: mix epilogue + dB + ring store ----
        const unsigned it = (nc == 1) ? (unsigned)s : (unsigned)s / (unsigned)nc;
        if constexpr (C::PAIR) {
            // ---- pair plan: ONE fold per column.  acc[k] = sum over the pairs of |Z[k]|^2 for the N bins of the complex spectrum: U[j] =
            // acc[k], V[j] = acc[k + M], k = lane + 64 j.  The column is out[k] = (acc[k] + acc[N - k]) / 2 = (U[k] + V[M - k]) / 2 for
            // 0 < k < M, out[0] = U[0], out[M] = V[0] (sum over the channels; AbsMean: / channels).  V[M - k] lives in lane 64 - lane,
            // register P/2 - 1 - j (lane 0: its own register P/2 - j): one mirror exchange of P/2 floats per lane through the wave's
            // exchange region, per COLUMN (the paired post pass of the other plans exchanges P/2 complex values per channel frame).
            static_assert(!C::PAIR || (MIXOP == 0 && OUTK == 0), "pair plan: sum-type mixes, float columns");
            if (s - (int)it * nc == nc - 1) {
                constexpr int J = P / 2;
                unsigned col = a.ring_pos + task_of(it, 0);
                if (col >= (unsigned)a.ring_w) col -= a.ring_w;
                float* const ex = reinterpret_cast<float*>(reinterpret_cast<cf*>(smem_raw) + wave * 2 * C::LDS_ELEMS);   // wave-private
                float U[J], V[J];
#pragma unroll
                for (int j = 0; j < J; ++j) {
                    U[j] = accU[j].x + accU[j].y;
                    V[j] = accV[j].x + accV[j].y;
                    accU[j] = accV[j] = cf{0.f, 0.f};
                }
#pragma unroll
                for (int j = 0; j < J; ++j) ex[lane + 64 * j] = V[j];
                if (lane == 0) ex[M] = U[0];                 // "V[M]" := U[0]: out[0] = (U[0] + U[0]) / 2 exactly
                wave_sync();
                float o[J];
#pragma unroll
                for (int j = 0; j < J; ++j) o[j] = U[j] + ex[M - (lane + 64 * j)];
                wave_sync();                                 // the next round's exchange stores stay behind these loads
                float oM = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(V[0])));   // bin M = N/2 (uniform: unmasked store below)
                oM = oM + oM;
                if (a.exact_div) {   // (. / 2) / channels: the halving is exact, the division IEEE (Spectrogram.cpp:74)
#pragma unroll
                    for (int j = 0; j < J; ++j) o[j] = (o[j] * 0.5f) / a.divisor;
                    oM = (oM * 0.5f) / a.divisor;
                } else {             // power-of-two channel count (or the plain sum): 1/2 and 1/channels as one exact scaling
                    const float hs = 0.5f * a.scale;
#pragma unroll
                    for (int j = 0; j < J; ++j) o[j] *= hs;
                    oM *= hs;
                }
                if (!a.linear) {
#pragma unroll
                    for (int j = 0; j < J; ++j) o[j] = XLOG ? jsg_exact_db(o[j]) : to_db(o[j]);
                    oM = XLOG ? jsg_exact_db(oM) : to_db(oM);
                }
                long long cofs = 0ll;
                unsigned trow = 0u;
                if constexpr (BAT) {
                    unsigned col0_;
                    long long in_off;
                    group_of(it, trow, col0_);
                    row_of(trow, in_off, cofs);
                }
                float* dst = a.out + (long long)col * a.out_pitch + cofs;
#pragma unroll
                for (int j = 0; j < J; ++j) __builtin_nontemporal_store(o[j], &dst[lane + 64 * j]);   // 256 contiguous bytes per instruction
                if (a.tail) a.tail[(long long)trow * a.ring_w + col] = oM;   // (plain store: see the tail plane of the other plans)
                else __builtin_nontemporal_store(oM, &dst[M]);
            }
        } else
        if (ONE || s - (int)it * nc == nc - 1) {
            // The eight waves of a "B" workgroup run independently (no barrier in the FFT), and over the 100 and more rounds of a persistent
            // workgroup they drift apart by whole columns: a wave then fetches the samples it shares with its neighbours (3/4 of a frame at 75 %
            // overlap) long after those have left the L2 -- the C3 dispatch read 1.38 x its samples from memory (round 4-5 counters).  With >= 4
            // channels per column the waves therefore meet at a column end about every 8 FFT rounds (round 5; C3: every column): reads 1.38 ->
            // 0.98 x (1.21 x with a period of 16), C3 +1..2 %, 8 channels at 4096 points +4..5 %.  With one or two channels per column the
            // meeting costs more than the traffic (stereo at 4096 points -4 % with a period of 4 or 8 rounds, level with 16; mono -1..-3 %):
            // not done there.  Every wave of the workgroup runs the same rounds, so the condition is uniform.
            if constexpr ((C::TWO_STAGE || (C::N == 4096 && L == 64)) && OUTK != 2 && !C::PAIR && !ONE) {
                if (nc >= 4 && (unsigned)(s + 1) / 8u != (unsigned)(s + 1 - nc) / 8u) __syncthreads();
            }
#pragma unroll
            for (int f = 0; f < F; ++f) {
                unsigned tk;
                if constexpr (RUNS) tk = cu_col0;        // (the column itself, clamped: runs_next)
                else if constexpr (CARRY) {
                    tk = cu_col0 + slot0 + tsub * F + f;
                    tk = tk < a.n_frames ? tk : a.n_frames - 1;
                } else tk = task_of(it, f);
                unsigned col = a.ring_pos + tk;                    // n_frames <= ring_w (checked by the launcher)
                if (col >= (unsigned)a.ring_w) col -= a.ring_w;
                const bool fuse_scale = !ONE && !XLOG && !a.linear && !a.exact_div;   // (uniform) v_log path of a mixing kernel: see below
                if constexpr (ONE) {
                    // one channel: the launcher selects this instantiation only when the mix scale is exactly 1
                } else
                if (a.exact_div) {   // m_powerfinal[kk] /= m_channels (Spectrogram.cpp:74), IEEE division
#pragma unroll
                    for (int m = 0; m < P / 2; ++m) acc[f][m] = cf{acc[f][m].x / a.divisor, acc[f][m].y / a.divisor};
                    accNy[f] = accNy[f] / a.divisor;
                } else if (a.scale != 1.0f && !fuse_scale) {   // power-of-two channel count: the same division as an exact scaling
#pragma unroll
                    for (int m = 0; m < P / 2; ++m) acc[f][m] *= a.scale;
                    accNy[f] *= a.scale;
                }
                if (!a.linear) {
                    if constexpr (XLOG) {   // the shared float32 routine, value by value (16 full-rate instructions each)
#pragma unroll
                        for (int m = 0; m < P / 2; ++m) acc[f][m] = cf{jsg_exact_db(acc[f][m].x), jsg_exact_db(acc[f][m].y)};
                        accNy[f] = jsg_exact_db(accNy[f]);
                    } else {
                    // (mixing kernels: the exact scaling by 1 / channels rides on the floor add as one packed fma -- the product is exact, so
                    // fma(p, 1/C, 1e-11) == p / C + 1e-11 bit for bit, and with 1/C == 1 it IS the add; one packed multiply less per pair)
                    const float sc = fuse_scale ? a.scale : 1.0f;
#pragma unroll
                    for (int m = 0; m < P / 2; ++m) {   // to_db() of both values of a pair: packed add and multiply around the two v_log
                        cf t;
                        if constexpr (ONE) t = acc[f][m] + cf{1e-11f, 1e-11f};
                        else t = __builtin_elementwise_fma(acc[f][m], cf{sc, sc}, cf{1e-11f, 1e-11f});
                        t = cf{__builtin_amdgcn_logf(t.x), __builtin_amdgcn_logf(t.y)};
                        acc[f][m] = t * cf{3.0102999566398120f, 3.0102999566398120f};
                    }
                    if constexpr (ONE) accNy[f] = to_db(accNy[f]);
                    else accNy[f] = __builtin_amdgcn_logf(__builtin_fmaf(accNy[f], sc, 1e-11f)) * 3.0102999566398120f;
                    }
                }
                if constexpr (L == 32) {
                    // Two frames sit side by side in the wavefront (lanes 0-31 | 32-63), and register rho of a lane is bin
                    // ll + 32 rho of ITS frame: stored as they lie, one instruction would write two 128-byte runs, in columns
                    // whose starts are not 128-byte aligned.  v_permlane32_swap trades the upper half of register rho
                    // (rho even) against the lower half of register rho + 1: afterwards register rho holds bins
                    // 32 rho + lane of the lower frame over all 64 lanes and register rho + 1 the same bins of the upper
                    // frame -- 256 contiguous bytes per store instruction, as in the 64-lane plans.
#pragma unroll
                    for (int m = 0; m < P / 2; m += 2) {
                        const auto lo = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[f][m].x), __float_as_uint(acc[f][m + 1].x), false, false);
                        const auto hi = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[f][m].y), __float_as_uint(acc[f][m + 1].y), false, false);
                        acc[f][m] = cf{__uint_as_float(lo[0]), __uint_as_float(hi[0])};
                        acc[f][m + 1] = cf{__uint_as_float(lo[1]), __uint_as_float(hi[1])};
                    }
                }
                if constexpr (L == 16) {
                    // FOUR frames side by side (quarter-wave s = lanes 16 s .. 16 s + 15), register rho of a lane = bin ll + 16 rho of ITS frame.
                    // A 4 x 4 transposition of (register within a group of four) x (quarter-wave): v_permlane32_swap trades the upper half of
                    // register a against the lower half of register c (and b against d), v_permlane16_swap then the odd quarters of a against
                    // the even quarters of b (and c against d) -- afterwards register 4 q + j holds bins 64 q + lane of frame j over all 64
                    // lanes (.x; .y: the mirrored bins M - those): 256 contiguous bytes of one column per store instruction.
                    static_assert(L != 16 || (P / 2) % 4 == 0, "groups of four registers");
                    auto tr4 = [](float& r0, float& r1, float& r2, float& r3) {
                        const auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(r0), __float_as_uint(r2), false, false);
                        const auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(r1), __float_as_uint(r3), false, false);
                        const auto t01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
                        const auto t23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
                        r0 = __uint_as_float(t01[0]); r1 = __uint_as_float(t01[1]); r2 = __uint_as_float(t23[0]); r3 = __uint_as_float(t23[1]);
                    };
#pragma unroll
                    for (int m = 0; m < P / 2; m += 4) {
                        float lo0 = acc[f][m].x, lo1 = acc[f][m + 1].x, lo2 = acc[f][m + 2].x, lo3 = acc[f][m + 3].x;
                        float up0 = acc[f][m].y, up1 = acc[f][m + 1].y, up2 = acc[f][m + 2].y, up3 = acc[f][m + 3].y;
                        tr4(lo0, lo1, lo2, lo3);
                        tr4(up0, up1, up2, up3);
                        acc[f][m] = cf{lo0, up0}; acc[f][m + 1] = cf{lo1, up1}; acc[f][m + 2] = cf{lo2, up2}; acc[f][m + 3] = cf{lo3, up3};
                    }
                }
                // column of the frame whose bins register rho holds after the swap (L == 32), else this lane's own
                const unsigned colA = L == 32 ? (unsigned)__builtin_amdgcn_readlane((int)col, 0) : col;
                const unsigned colB = L == 32 ? (unsigned)__builtin_amdgcn_readlane((int)col, 32) : col;
                constexpr int LW = L == 32 ? 64 : L;         // lanes that share one store instruction's run
                const int lw = L == 32 ? lane : ll;
                if constexpr (OUTK == 2) {
                    // palette indices of the column, four to a dword, into this wave's exchange region: dword (r4, ll) of the lower
                    // half holds bins ll + 64 (4 r4 + j), j = 0..3, the upper half M - those; the region is skewed per wave so that
                    // the store phase below reads without bank conflicts (eight columns x eight consecutive ll per instruction)
                    constexpr int D = C::LDS_ELEMS * 2;                       // dwords between the waves' regions
                    constexpr int SK = ((4 - D % 32) + 32) % 32;              // skew per wave: (D + SK) == 4 (mod 32)
                    unsigned* ix = reinterpret_cast<unsigned*>(lds0 + f * C::LDS_ELEMS) + (SK * wave) % 32;
                    auto park = [&](auto fast_tag) {
                        constexpr bool FAST = decltype(fast_tag)::value;
#pragma unroll
                        for (int r4 = 0; r4 < P / 8; ++r4) {
                            unsigned wx = 0, wy = 0;
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                int ix_, iy_;
                                if constexpr (FAST) color_index2_fast(acc[f][4 * r4 + j], a.vmin, a.mult, a.n_colors, ix_, iy_);
                                else color_index2(acc[f][4 * r4 + j], a.vmin, a.vmax, a.top, a.mult, a.n_colors, ix_, iy_);
                                wx |= (unsigned)ix_ << (8 * j);
                                wy |= (unsigned)iy_ << (8 * j);
                            }
                            ix[r4 * 64 + ll] = wx;
                            ix[(P / 8 + r4) * 64 + ll] = wy;
                        }
                    };
                    if (a.cmap_fast) park(std::true_type{});   // (uniform)
                    else park(std::false_type{});
                    if (ll == 0) ix[(P / 4) * 64] = (unsigned)color_index(accNy[f], a.vmin, a.vmax, a.top, a.mult, a.n_colors);
                    __syncthreads();
                    {
                        // ---- store phase: every wave-instruction covers 8 consecutive values of ll (rows) x the 8 columns of the
                        //      iteration: lane = 8 * dl + c reads dword (hr, 8 wave + dl) of column c and writes four pixels.  Wave w takes the
                        //      items q = w + 8 i (i < 2 P/8), i.e. group g = w of every (half, r4) = hr = i: the LDS reads of a wave are
                        //      compile-time offsets from one address and its rows step by 64 from value to value, so a pixel costs one
                        //      32-bit add (byte offsets inside the image: the launcher takes this path for images below 4 GiB only), one
                        //      palette read and one store.  (Round 3 first had a rolled loop with a 64-bit multiply per pixel and three
                        //      serial LDS round trips per item: 30 000 columns in one launch 238 us, this form 221-223 us, no store phase
                        //      at all 174 us -- tools/abbench --cfg c5wide.) ----
                        const int c = lane & 7, dl = lane >> 3;
                        static_assert(C::TPB == 8 && C::WPB == 8, "the store phase is laid out for eight columns per iteration");
                        unsigned image, col0;
                        const bool inside = group_of(it, image, col0);
                        const unsigned tcol = col0 + c;                               // column c of this iteration, inside its image
                        const bool live = inside && tcol < a.n_frames;
                        const unsigned k0 = 8u * (unsigned)wave + (unsigned)dl;       // this lane's bin of the first value: k = k0 + 64 m
                        const unsigned* cx = reinterpret_cast<const unsigned*>(smem_raw) + c * D + (SK * c) % 32 + k0;
                        constexpr int NI = 2 * (P / 8);
                        unsigned w4[NI];
#pragma unroll
                        for (int i = 0; i < NI; ++i) w4[i] = cx[i * 64];
                        const unsigned wNy = cx[(P / 4) * 64 - k0];                   // bin M / 2 of column c (stored by one wave below)
                        // Only these reads sit between the two workgroup barriers: the indices now live in registers, the exchange regions
                        // are free again, and the palette reads and the stores below run on the wave's own time, not inside the barrier pair
                        // (Also tried: the waves 0-3 deferring their pixels into the middle of their next FFT round so that on every SIMD one
                        // wave's stores lie under the other's arithmetic: 229 vs 221 us)
                        __syncthreads();
                        unsigned x = (unsigned)a.x_first + tcol;                      // x_first < x_wrap, tcol < n_frames <= x_wrap (launcher)
                        if (x >= (unsigned)a.x_wrap) x -= (unsigned)a.x_wrap;
                        const unsigned pitch4 = (unsigned)a.argb_pitch * 4u, step = 64u * pitch4;
                        char* const img = reinterpret_cast<char*>(a.argb + (long long)image * a.argb_image_stride);
                        // y = H - 1 - bin (Spectrogram.cpp:642): the lower half (bin k) walks up the image from row M - k0, the upper half
                        // (bin M - k) down from row k0.  Plain stores on purpose: a wave-instruction writes 32-byte pieces of eight
                        // image rows, and the neighbouring workgroups (same XCD: the block remap gives an XCD one contiguous column
                        // range) write the rest of those 128-byte lines; with the default policy the pieces meet in L2 and leave as
                        // whole lines, streamed out (non-temporal) every piece went to memory alone: 28.6 vs 22.3 us per C5 image
                        unsigned oLo = ((unsigned)M - k0) * pitch4 + x * 4u, oUp = k0 * pitch4 + x * 4u;
                        if (live) {
                            // All palette reads of the lane first, then the stores (round 5).  Written pixel by pixel the compiler issued read,
                            // s_waitcnt lgkmcnt(0), store -- 33 LDS round trips in a row per wave, at the one moment when both waves of
                            // a SIMD are in this phase and cannot cover for each other.  The FFT's registers are dead here: 33 more cost nothing.
                            unsigned rgb[NI * 4];
#pragma unroll
                            for (int i = 0; i < NI; ++i)
#pragma unroll
                                for (int j = 0; j < 4; ++j) rgb[4 * i + j] = (unsigned)s_lut[(w4[i] >> (8 * j)) & 0xffu];
                            const unsigned rgbNy = (unsigned)s_lut[wNy & 0xffu];
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int i = 0; i < NI; ++i) {
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    if (i < P / 8) { *reinterpret_cast<unsigned*>(img + oLo) = rgb[4 * i + j]; oLo -= step; }
                                    else { *reinterpret_cast<unsigned*>(img + oUp) = rgb[4 * i + j]; oUp += step; }
                                }
                            }
                            if (wave == C::WPB - 1 && dl == 0)     // (row M - M / 2)
                                *reinterpret_cast<unsigned*>(img + ((unsigned)(M / 2) * pitch4 + x * 4u)) = rgbNy;
                        }
                    }
                } else if constexpr (OUTK == 1) {
                    // palette index of every bin; 64 consecutive bytes of the index column per store instruction
                    unsigned char* icA = a.idx + (long long)colA * a.idx_pitch;
                    unsigned char* icB = a.idx + (long long)colB * a.idx_pitch;
                    unsigned char* ic = a.idx + (long long)col * a.idx_pitch;
                    // The lane offset is made opaque HERE: left visible, the 64-bit per-lane store addresses are loop invariants which the
                    // compiler computes once in the prologue and keeps for the whole kernel -- in Cfg4096B's index-out instantiation (254
                    // VGPRs) they were what it spilled (4 VGPRs, 20 bytes of scratch, rounds 3-4).  Recomputing them per column is a
                    // handful of integer adds.
                    int lwo = lw;
                    asm volatile("" : "+v"(lwo));
                    auto put = [&](auto fast_tag) {
                        constexpr bool FAST = decltype(fast_tag)::value;
#pragma unroll
                        for (int rho = 0; rho < P / 2; ++rho) {
                            unsigned char* d = (L == 32 && (rho & 1)) ? icB : icA;
                            const int k = lwo + LW * (L == 32 ? rho / 2 : rho);
                            int ix_, iy_;
                            if constexpr (FAST) color_index2_fast(acc[f][rho], a.vmin, a.mult, a.n_colors, ix_, iy_);
                            else color_index2(acc[f][rho], a.vmin, a.vmax, a.top, a.mult, a.n_colors, ix_, iy_);
                            d[k] = (unsigned char)ix_;
                            d[M - k] = (unsigned char)iy_;
                        }
                    };
                    if (a.cmap_fast) put(std::true_type{});   // (uniform)
                    else put(std::false_type{});
                    if (L <= 64 || ll == 0) ic[M / 2] = (unsigned char)color_index(accNy[f], a.vmin, a.vmax, a.top, a.mult, a.n_colors);
                } else {
                    // non-temporal dword stores, 256 contiguous bytes of the column per instruction (streaming the columns out
                    // instead of leaving them dirty in L2 removed the end-of-kernel write-back)
                    long long cofs = a.per_channel ? (long long)c0 * a.out_cpitch : 0ll;
                    if constexpr (CARRY) cofs = cu_out;
                    else if constexpr (BAT) {   // the ring of this row (batch, or batch and channel)
                        unsigned row, col0_;
                        long long in_off;
                        group_of(it, row, col0_);
                        row_of(row, in_off, cofs);
                    }
                    float* dstA = a.out + (long long)colA * a.out_pitch + cofs;
                    float* dstB = a.out + (long long)colB * a.out_pitch + cofs;
                    float* dst = a.out + (long long)col * a.out_pitch + cofs;
                    // tail plane (jsg_stft_args.out_tail): bin N/2 -- the one value of a column that lies behind its sixteen (N = 1024)
                    // whole 128-byte lines -- goes to a dense plane of one float per column instead; row = batch x channel plane
                    float* tl_row = nullptr;
                    if (a.tail) {
                        unsigned trow = a.per_channel ? (unsigned)c0 : 0u;
                        if constexpr (CARRY) trow = cu_row;
                        else if constexpr (BAT) {
                            unsigned col0_;
                            group_of(it, trow, col0_);
                        }
                        tl_row = a.tail + (long long)trow * a.ring_w;
                    }
                    if constexpr (L == 16) {
                        // as the L <= 64 form below with four frames: register 4 q + j = window q (bins 64 q + lane and their mirror) of frame j
                        static_assert(L != 16 || OUTK == 0, "the 16-lane plan writes float columns");
                        float* dJ[4];
                        unsigned cJ[4];
                        float nyJ[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            cJ[j] = (unsigned)__builtin_amdgcn_readlane((int)col, 16 * j);
                            dJ[j] = a.out + (long long)cJ[j] * a.out_pitch + cofs;
                            nyJ[j] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(accNy[f]), 16 * j));
                        }
                        // (the mirrored half is addressed from its LOWEST window upwards: one non-negative lane offset per column and
                        // non-negative immediates -- written as M - (..) the compiler formed a 64-bit address per store)
                        constexpr int NQ = P / 8;                           // windows of 64 bins per half column
                        static_assert((M - 64 * NQ) % 64 == 0 && M - 64 * NQ >= 0, "the lane offset of the mirrored half is an OR");
                        const int upo = (M - 64 * NQ) | ((64 - lane) & 63);   // = M - 64 (NQ - 1) - (lane == 0 ? 64 : lane), with every bit known
#pragma unroll
                        for (int rho = 0; rho < P / 2; ++rho) {
                            const int j = rho % 4, q = rho / 4;
                            float* d = dJ[j];
                            __builtin_nontemporal_store(acc[f][rho].x, &d[lane + 64 * q]);
                            const float nxt = rho + 4 < P / 2 ? acc[f][rho + 4].y : nyJ[j];
                            const float up = lane == 0 ? nxt : acc[f][rho].y;
                            __builtin_nontemporal_store(up, &d[upo + 64 * (NQ - 1 - q)]);
                        }
                        if (lane == 0) {                                    // bin M of the four frames: into the tail plane, or behind the column
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                float* pm = tl_row ? tl_row + cJ[j] : dJ[j] + M;
                                *pm = acc[f][j].y;
                            }
                        }
                    } else
                    if constexpr (L <= 64) {
                        // Whole 128-byte lines.  The lower half of the column (bins k = lane + 64 r) starts every instruction on a line.  The
                        // upper half is the mirror, bins M - k: as the registers hold them an instruction would cover bins M - 64 r - 63 ..
                        // M - 64 r -- 252 bytes of two lines plus FOUR bytes (lane 0: bin M - 64 r) of a third, which the previous
                        // instruction fills: every such line reached memory as a 124-byte and a 4-byte piece (round 5 counters: writes
                        // 1.023 x the column bytes with or without the tail plane).  So lane 0 takes the value of the NEXT register -- bin
                        // M - 64 (r + 1), the low end of this instruction's two lines -- and the last instruction's lane 0 the self-paired
                        // bin M/2 (accNy, held by every lane): bins M - 64 r - 64 .. M - 64 r - 1 per instruction, and no separate store
                        // for bin M/2.  Bin M itself (lane 0 of the first register) goes to the tail plane, or alone into the column's
                        // last line.  Which lane stores a value changes, no value does.
                        constexpr int RS = L == 32 ? 2 : 1;                 // registers between consecutive windows of one frame
                        float nyA = accNy[f], nyB = accNy[f];               // bin M/2 of the frame(s) of this wavefront
                        if constexpr (L == 32) {
                            nyA = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(accNy[f]), 0));
                            nyB = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(accNy[f]), 32));
                        }
                        const int lwu = lw == 0 ? 64 : lw;                  // lane 0 stores 64 bins further down (one select, not one per register)
                        if (!RUNS || cu_live)                               // (runs: a column past the end of its row is not stored; uniform)
#pragma unroll
                        for (int rho = 0; rho < P / 2; ++rho) {
                            const bool fb = L == 32 && (rho & 1);           // the upper frame of a two-frame wavefront
                            float* d = fb ? dstB : dstA;
                            const int k = lw + LW * (rho / RS);
                            __builtin_nontemporal_store(acc[f][rho].x, &d[k]);
                            const float nxt = rho + RS < P / 2 ? acc[f][rho + RS].y : (fb ? nyB : nyA);
                            const float up = lw == 0 ? nxt : acc[f][rho].y;
                            __builtin_nontemporal_store(up, &d[M - (lwu + LW * (rho / RS))]);
                            if (rho / RS == 0 && lw == 0) {                 // bin M
                                if (tl_row) tl_row[fb ? colB : colA] = acc[f][rho].y;   // (plain store: the pieces of 32 columns meet in L2)
                                else __builtin_nontemporal_store(acc[f][rho].y, &d[M]);
                            }
                        }
                    } else {
#pragma unroll
                    for (int rho = 0; rho < P / 2; ++rho) {
                        float* d = dstA;
                        const int k = lw + LW * rho;
                        __builtin_nontemporal_store(acc[f][rho].x, &d[k]);
                        if (rho == 0 && tl_row) {   // this instruction's lane k == 0 holds bin M = N/2
                            if (k == 0) tl_row[colA] = acc[f][rho].y;
                            else __builtin_nontemporal_store(acc[f][rho].y, &d[M - k]);
                        } else
                        __builtin_nontemporal_store(acc[f][rho].y, &d[M - k]);
                    }
                    if (ll == 0) __builtin_nontemporal_store(accNy[f], &dst[M / 2]);
                    }
                }
#pragma unroll
                for (int m = 0; m < P / 2; ++m) {
                    // (sum mix: one 64-bit move per pair, written out -- the compiler zeroes the pairs with two 32-bit moves each)
                    if constexpr (MIXOP == 0) asm volatile("v_mov_b64 %0, 0" : "=v"(acc[f][m]));
                    else acc[f][m] = cf{init, init};
                }
                accNy[f] = init;
            }
        }
    };

    using t0 = std::integral_constant<int, 0>;
    if constexpr (RUNS) {   // a.iters runs of RL rounds each, every run written out (a round's place in its run is a compile-time fact);
                            // the last round of the last run prefetches nothing
        auto run = [&](int s0, auto last_tag) __attribute__((always_inline)) {
            for_each_t([&](auto t_tag) __attribute__((always_inline)) {
                constexpr int T = decltype(t_tag)::value;
                if constexpr (T + 1 < RL) process(s0 + T, std::false_type{}, t_tag);
                else process(s0 + T, last_tag, t_tag);
            }, std::make_integer_sequence<int, RL>{});
        };
        int s = 0;
        for (int r = 0; r + 1 < a.iters; ++r, s += RL) run(s, std::false_type{});
        run(s, std::true_type{});
    } else {
        for (int s = 0; s + 1 < n_fft; ++s) process(s, std::false_type{}, t0{});
        process(n_fft - 1, std::true_type{}, t0{});
    }
}

// ------------------------------------------------------------------------------------------------------------
// plan: window + twiddle tables, laid out [table][register j][lane]
// ------------------------------------------------------------------------------------------------------------
template <class C>
void fill_tables(std::vector<float2>& t, const float* window, double amp) {
    constexpr int L = C::L, P = C::P, R1 = C::R1, R2 = C::R2, R3 = C::R3, M = C::M, N = C::N, TL = C::TL;
    t.assign(size_t(C::TAB_ELEMS), make_float2(0.f, 0.f));
    const double two_pi = 6.283185307179586476925286766559;
    for (int n2 = 0; n2 < R2; ++n2)   // stage-1 twiddles, one row per n2 (shared by the lanes with t1 / R3 == n2)
        for (int k1 = 0; k1 < R1; ++k1) {
            const double ang = -two_pi * double((long long)n2 * k1 % (R1 * R2)) / double(R1 * R2);
            if (k1 == 0) continue;   // the twiddle of k1 = 0 is 1 and never read; k1 sits at column k1 - 1
            t[C::TAB_TW1 + n2 * C::TS1 + k1 - 1] = make_float2(float(std::cos(ang)), float(std::sin(ang)));
        }
    if constexpr (C::TWF)
        for (int n3 = 0; n3 < R3; ++n3)
            for (int k2 = 0; k2 < R2; ++k2) {
                const double ang = -two_pi * double((long long)n3 * k2 % (M / R1)) / double(M / R1);
                t[C::TAB_TW2 + n3 * C::TSB + k2] = make_float2(float(std::cos(ang)), float(std::sin(ang)));
            }
    if constexpr (C::PAIR) {
        // window: P x 64 floats, value m of wavefront lane (ll, h) = w[64 m + 2 ll + h] * amp at [m / 4][lane][m % 4] (no 1/2: there is no
        // paired post pass); last stage: W_N^(lane + 64 j), j < P/4, pairs of j side by side (Cfg::pair_tw_idx)
        float* wf = reinterpret_cast<float*>(t.data() + C::TAB_WIN);
        for (int lane = 0; lane < 64; ++lane) {
            const int q = 2 * (lane % L) + lane / L;
            for (int m = 0; m < P; ++m) wf[(m / 4) * 256 + lane * 4 + m % 4] = float(double(window[64 * m + q]) * amp);
            for (int j = 0; j < P / 4; ++j) {
                const double ang = -two_pi * double(lane + 64 * j) / double(N);
                t[C::TAB_POST + C::pair_tw_idx(j, lane)] = make_float2(float(std::cos(ang)), float(std::sin(ang)));
            }
        }
        return;
    }
    for (int e = 0; e < TL; ++e) {   // entry e of a table row belongs to lane-in-frame ll (L = 32: both half-waves)
        const int ll = e % L;
        for (int m = 0; m < P; ++m) {   // window pairs: samples 2n, 2n+1 with n = ll + L m
            const int n = ll + L * m;
            const double a2 = 0.5 * amp;   // the paired post pass expects Z/2
            t[C::TAB_WIN + C::tab_idx(m, e)] = make_float2(float(double(window[2 * n]) * a2), float(double(window[2 * n + 1]) * a2));
        }
        if constexpr (C::TWF) {   // lane constants A[v] = W_M^(n3 k1), C = -i W_N^ll (the shared B rows are filled above)
            for (int v = 0; v < C::U2; ++v) {
                const int t2 = ll + L * v, k1 = t2 / R3, n3 = t2 % R3;
                const double ang = -two_pi * double((long long)n3 * k1 % M) / double(M);
                t[C::TAB_A + v * TL + e] = make_float2(float(std::cos(ang)), float(std::sin(ang)));
            }
            const double angc = -two_pi * double(ll) / double(N);
            t[C::TAB_POST + e] = make_float2(float(std::sin(angc)), float(-std::cos(angc)));
            continue;
        }
        for (int v = 0; v < (C::TWO_STAGE ? 0 : C::U2); ++v)
            for (int k2 = 0; k2 < R2; ++k2) {
                const int t2 = ll + L * v, k1 = t2 / R3, n3 = t2 % R3;
                const double ang = -two_pi * double((long long)n3 * (k1 + R1 * k2) % M) / double(M);
                t[C::TAB_TW2 + C::tab_idx(v * R2 + k2, e)] = make_float2(float(std::cos(ang)), float(std::sin(ang)));
            }
        for (int rho = 0; rho < P / 2; ++rho) {   // bins k = ll + L rho of the lower half: -i exp(i ang) = sin(ang) - i cos(ang)
            const double ang = -two_pi * double(ll + L * rho) / double(N);
            t[C::TAB_POST + C::tab_idx(rho, e)] = make_float2(float(std::sin(ang)), float(-std::cos(ang)));
        }
    }
}

// Kernels that need more than 48 KB of dynamic LDS must be told so once per device.  jsg_plan_create does it for every
// instantiation of the plan's size (so that the first launch may already sit inside a stream capture); the launch path
// repeats the check for plans that are used on a device other than the one they were created on.
template <class C, int MIXOP, int OUTK, int STREAM = 0, int XLOG = 0>
hipError_t ensure_lds_attr() {
    static std::atomic<bool> attr_done[64];   // set once per device; setting it twice from two threads is harmless
    constexpr int bytes = C::LDS_BYTES;
    if (bytes <= 48 * 1024) return hipSuccess;
    int dev = 0;
    hipError_t err = hipGetDevice(&dev);
    if (err != hipSuccess) return err;
    if (dev >= 0 && dev < 64 && !attr_done[dev].load(std::memory_order_acquire)) {
        err = hipFuncSetAttribute(reinterpret_cast<const void*>(&stft_db_kernel<C, MIXOP, OUTK, STREAM, XLOG>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (err != hipSuccess) return err;
        attr_done[dev].store(true, std::memory_order_release);
    }
    return hipSuccess;
}

// plans that can colour their own columns (stft_db_kernel, OUTK == 2): one wavefront per frame, eight frames per workgroup
template <class C>
constexpr bool image_ok = C::L == 64 && C::FPW == 1 && C::TPB == 8 && C::WPB == 8 && C::LDS_TOTAL + 1024 <= 160 * 1024;
// configurations that exist for the single-kernel display path alone (nothing else is instantiated for them)
template <class C>
constexpr bool image_only = std::is_same<C, Cfg1024I>::value;

// plans that exist for float columns (dB / power) and the sum-type and one-channel mixes only: the 16-lane two-stage plan
template <class C>
constexpr bool db_only = C::L == 16;

// every instantiation of a plan that can be launched: (MIXOP, OUTK, STREAM) x the two logarithms
template <class C, int XLOG>
hipError_t ensure_lds_attrs_of_plan_x() {
    if constexpr (db_only<C>) {
        hipError_t e = ensure_lds_attr<C, 0, 0, 0, XLOG>();
        if (e == hipSuccess) e = ensure_lds_attr<C, 3, 0, 0, XLOG>();
        if (e == hipSuccess) e = ensure_lds_attr<C, 0, 0, 1, XLOG>();
        if (e == hipSuccess) e = ensure_lds_attr<C, 3, 0, 1, XLOG>();
        return e;
    } else
    if constexpr (C::PAIR) {   // the pair plan exists for the sum-type mixes and float columns only
        hipError_t e = ensure_lds_attr<C, 0, 0, 0, XLOG>();
        if (e == hipSuccess) e = ensure_lds_attr<C, 0, 0, 1, XLOG>();
        return e;
    } else
    if constexpr (image_only<C>) {
        hipError_t e = ensure_lds_attr<C, 0, 2, 0, XLOG>();
        if (e == hipSuccess) e = ensure_lds_attr<C, 3, 2, 0, XLOG>();
        return e;
    } else {
    hipError_t e = ensure_lds_attr<C, 0, 0, 0, XLOG>();
    if (e == hipSuccess) e = ensure_lds_attr<C, 1, 0, 0, XLOG>();
    if (e == hipSuccess) e = ensure_lds_attr<C, 2, 0, 0, XLOG>();
    if (e == hipSuccess) e = ensure_lds_attr<C, 3, 0, 0, XLOG>();
    if (e == hipSuccess) e = ensure_lds_attr<C, 0, 1, 0, XLOG>();
    if (e == hipSuccess) e = ensure_lds_attr<C, 3, 1, 0, XLOG>();
    if (e == hipSuccess) e = ensure_lds_attr<C, 0, 0, 1, XLOG>();
    if (e == hipSuccess) e = ensure_lds_attr<C, 3, 0, 1, XLOG>();
    if constexpr (image_ok<C>) {
        if (e == hipSuccess) e = ensure_lds_attr<C, 0, 2, 0, XLOG>();
        if (e == hipSuccess) e = ensure_lds_attr<C, 3, 2, 0, XLOG>();
    }
    return e;
    }
}
template <class C>
hipError_t ensure_lds_attrs_of_plan() {
    hipError_t e = ensure_lds_attrs_of_plan_x<C, 0>();
    if (e == hipSuccess) e = ensure_lds_attrs_of_plan_x<C, 1>();
    return e;
}

template <class C, int MIXOP, int OUTK = 0, int STREAM = 0>
hipError_t launch_stft_mix(const StftKArgs& ka, dim3 grid, hipStream_t s) {
    const bool xlog = ka.exact_log && !ka.linear;   // (linear power has no logarithm: one instantiation serves both)
    hipError_t err = xlog ? ensure_lds_attr<C, MIXOP, OUTK, STREAM, 1>() : ensure_lds_attr<C, MIXOP, OUTK, STREAM, 0>();
    if (err != hipSuccess) return err;
    const int flags = (ka.regular ? 1 : 0) | (ka.per_channel ? 2 : 0) | (ka.xcd_remap ? 4 : 0) | (ka.chunked ? 8 : 0);
    constexpr int lds = C::LDS_BYTES;
    if (xlog)
        hipLaunchKernelGGL((stft_db_kernel<C, MIXOP, OUTK, STREAM, 1>), grid, dim3(C::WPB * 64), lds, s, ka.in, ka.in_pitch, ka.tab, ka.n_frames,
                           ka.first_frame, ka.hop, flags, unsigned(ka.c_begin) | unsigned(ka.c_end) << 16, ka.iters, grid.x, ka.feedblocks, ka);
    else
        hipLaunchKernelGGL((stft_db_kernel<C, MIXOP, OUTK, STREAM, 0>), grid, dim3(C::WPB * 64), lds, s, ka.in, ka.in_pitch, ka.tab, ka.n_frames,
                           ka.first_frame, ka.hop, flags, unsigned(ka.c_begin) | unsigned(ka.c_end) << 16, ka.iters, grid.x, ka.feedblocks, ka);
    return hipGetLastError();
}

template <class C>
hipError_t launch_stft(const StftKArgs& ka, int mixop, dim3 grid, hipStream_t s) {
    if constexpr (db_only<C>) {
        if (ka.argb || ka.idx || (mixop != 0 && mixop != 3)) return hipErrorInvalidValue;
        return mixop == 3 ? launch_stft_mix<C, 3>(ka, grid, s) : launch_stft_mix<C, 0>(ka, grid, s);
    } else
    if constexpr (C::PAIR) {
        if (ka.argb || ka.idx || mixop != 0) return hipErrorInvalidValue;
        return launch_stft_mix<C, 0>(ka, grid, s);
    } else {
    if (ka.argb) {  // single-kernel display path
        if constexpr (image_ok<C>) {
            if (mixop == 3) return launch_stft_mix<C, 3, 2>(ka, grid, s);
            if (mixop == 0) return launch_stft_mix<C, 0, 2>(ka, grid, s);
        }
        return hipErrorInvalidValue;
    }
    if constexpr (image_only<C>) return hipErrorInvalidValue;
    else {
    if (ka.idx) {   // fused display path: the mixed (AbsMean / Sum) and the one-channel instantiations only
        if (mixop == 3) return launch_stft_mix<C, 3, 1>(ka, grid, s);
        if (mixop == 0) return launch_stft_mix<C, 0, 1>(ka, grid, s);
        return hipErrorInvalidValue;
    }
    switch (mixop) {
        case 1: return launch_stft_mix<C, 1>(ka, grid, s);
        case 2: return launch_stft_mix<C, 2>(ka, grid, s);
        case 3: return launch_stft_mix<C, 3>(ka, grid, s);
        default: return launch_stft_mix<C, 0>(ka, grid, s);
    }
    }
    }
}

// strided multi-batch launches (STREAM == 1): the sum-mixed and the one-channel instantiations
template <class C, int STREAM>
hipError_t launch_stft_strided(const StftKArgs& ka, int mixop, dim3 grid, hipStream_t s) {
    if constexpr (image_only<C>) return hipErrorInvalidValue;
    else {
        if (ka.argb || ka.idx) return hipErrorInvalidValue;
        if constexpr (C::PAIR) return mixop == 0 ? launch_stft_mix<C, 0, 0, STREAM>(ka, grid, s) : hipErrorInvalidValue;
        else
        if (mixop == 3) return launch_stft_mix<C, 3, 0, STREAM>(ka, grid, s);
        if constexpr (STREAM == 2) return hipErrorInvalidValue;   // (runs: the one-channel kernels only)
        else
        if (mixop == 0) return launch_stft_mix<C, 0, 0, STREAM>(ka, grid, s);
        return hipErrorInvalidValue;
    }
}


// ------------------------------------------------------------------------------------------------------------
// Per-plan entry points.  The kernels of a plan are instantiated in exactly one translation unit (jsg_stft_a.hip or
// jsg_stft_b.hip: JSG_STFT_PLANS below); jsg_kernels.hip reaches them through these plain functions only, so no kernel is
// ever compiled twice and hipFuncSetAttribute always addresses the one copy that is launched.
// ------------------------------------------------------------------------------------------------------------
#define JSG_FOR_EACH_PLAN(X) X(Cfg512) X(Cfg1024) X(Cfg1024I) X(Cfg1024B) X(Cfg2048) X(Cfg2048B) X(Cfg2048P) X(Cfg4096) X(Cfg4096B) X(Cfg8192)
#define JSG_DECLARE_PLAN(C)                                                                   \
    hipError_t launch_##C(const StftKArgs& ka, int mixop, dim3 grid, hipStream_t s);          \
    hipError_t launch_strided_##C(const StftKArgs& ka, int mixop, dim3 grid, hipStream_t s);  \
    hipError_t ensure_attrs_##C();
JSG_FOR_EACH_PLAN(JSG_DECLARE_PLAN)
#undef JSG_DECLARE_PLAN
#define JSG_DEFINE_PLAN(C)                                                                                                  \
    hipError_t launch_##C(const StftKArgs& ka, int mixop, dim3 grid, hipStream_t s) { return launch_stft<C>(ka, mixop, grid, s); } \
    hipError_t launch_strided_##C(const StftKArgs& ka, int mixop, dim3 grid, hipStream_t s) { return launch_stft_strided<C, 1>(ka, mixop, grid, s); } \
    hipError_t ensure_attrs_##C() { return ensure_lds_attrs_of_plan<C>(); }
// Loads the unit's code object onto the current device (the runtime loads lazily, 2.5 ms on first use: done when a plan is
// created, not inside the audio thread's first jsg_process_block).
hipError_t touch_module_a();
hipError_t touch_module_b();
// STREAM == 2 ("runs": consecutive columns per wavefront, the overlapped half of a frame kept in registers) exists for the 1024-point plan
hipError_t launch_runs_Cfg1024(const StftKArgs& ka, int mixop, dim3 grid, hipStream_t s);

}  // namespace jsg
