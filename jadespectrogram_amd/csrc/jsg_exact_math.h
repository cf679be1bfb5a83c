// Arithmetic that is shared, source line by source line, between the device code of libjsg.so and the CPU mirror of the kernel
// (oracle/jsg_mirror.c, test infrastructure): plain float32 operations -- add, multiply, fused multiply-add, integer bit
// manipulation -- which IEEE 754 defines bit for bit, so both sides produce the same results.
//
//   jsg_exact_db(p) = 10 * log10(p + 1e-11f)     reference: Spectrogram.cpp:107 with g_minValForLogSpectrogram (:36)
//
// The default kernels take this logarithm on the hardware unit (v_log_f32: one quarter-rate instruction, accurate to 1 ulp but not
// specified bit for bit, hence not reproducible on a CPU).  With jsg_stft_args.exact_log / jsg_set_exact_log the epilogue calls this
// routine instead: exponent extraction, a degree-9 polynomial for log2 of the mantissa (|relative error| < 4.2e-9 before rounding) and a
// split constant for the final scaling -- 16 full-rate vector instructions per value instead of 3.  The dB
// values, and with them the palette indices (CColorPalette::getRGBColor, CColorpalette.h:32-47), are then bit-identical to the mirror's.
// Against the reference's double log10 rounded to float: within 2 ulp of the dB value over [1e-11, 1e12] (tests/test_oracle_golden.py).
// Finite, non-negative inputs only (what |X|^2 produces); no NaN / infinity handling.
#pragma once

#if defined(__HIPCC__)
#define JSG_EXACT_HD __host__ __device__ __forceinline__
#else
#define JSG_EXACT_HD static inline
#include <math.h>
#include <string.h>
#endif

JSG_EXACT_HD unsigned jsg_exact_bits(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(v);
#else
    unsigned b;
    memcpy(&b, &v, 4);
    return b;
#endif
}
JSG_EXACT_HD float jsg_exact_float(unsigned b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(b);
#else
    float v;
    memcpy(&v, &b, 4);
    return v;
#endif
}
JSG_EXACT_HD float jsg_exact_fma(float a, float b, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fmaf(a, b, c);
#else
    return fmaf(a, b, c);
#endif
}

JSG_EXACT_HD float jsg_exact_db(float p) {
    const float y = p + 1e-11f;                       // >= 1e-11: a normal float
    const unsigned b = jsg_exact_bits(y);
    int e = (int)(b >> 23) - 127;
    unsigned mb = (b & 0x007fffffu) | 0x3f800000u;    // mantissa as a float in [1, 2)
    if (mb > 0x3fb504f3u) {                           // > sqrt(2): halve it, so that m lies in (sqrt(1/2), sqrt(2)]
        mb -= 0x00800000u;
        e += 1;
    }
    const float f = jsg_exact_float(mb) - 1.0f;       // exact
    // log2(1 + f) = f * q(f), q fitted on [sqrt(1/2) - 1, sqrt(2) - 1] (Chebyshev nodes, degree 9)
    float q = -0.11020158976316452f;
    q = jsg_exact_fma(q, f, 0.18631209433078766f);
    q = jsg_exact_fma(q, f, -0.19102497398853302f);
    q = jsg_exact_fma(q, f, 0.2045752853155136f);
    q = jsg_exact_fma(q, f, -0.23961904644966125f);
    q = jsg_exact_fma(q, f, 0.2885688841342926f);
    q = jsg_exact_fma(q, f, -0.3606966435909271f);
    q = jsg_exact_fma(q, f, 0.4808982014656067f);
    q = jsg_exact_fma(q, f, -0.7213473320007324f);
    q = jsg_exact_fma(q, f, 1.4426950216293335f);
    const float t = f * q;                            // log2 of the mantissa
    // 10 log10(2) = 3.0102999566398120 = K_hi + K_lo (K_hi = the nearest float)
    const float K_hi = 3.01029992103576660156f, K_lo = 3.56040453e-08f;
    const float ef = (float)e;
    const float r = jsg_exact_fma(ef, K_lo, t * K_hi);
    return jsg_exact_fma(ef, K_hi, r);
}
