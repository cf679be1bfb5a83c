// Host-side precompute of the hot path: ring geometry, analysis windows, colour-map tables.
// These are the one-off set-up steps that feed the HIP kernels;
// they run on the CPU in the reference too and are restated here from its behaviour.
//
//   geometry  : Spectrogram::buildmem            reference Spectrogram.cpp:213-218
//   windows   : Spectrogram::setWindowFkt        reference Spectrogram.cpp:239-293
//   colours   : CColorPalette::ComputeColors     reference CColorpalette.cpp:106-339
//   range     : CColorPalette::setValueRange     reference CColorpalette.cpp:39-54
//
// The arithmetic types (float vs double, where truncation happens) are chosen so that the results are
// bit-identical to the reference built with g++ on x86-64; tests/test_host_math.py checks that against the
// golden vectors produced by the reference's own CColorpalette.cpp and against SURVEY's known answers.
#include <cmath>
#include <cstdint>
#include <cstddef>
#include <vector>

#include "../../include/jsg.h"
#include "jsg_colormap_tables.inc"

namespace {

const double kPi = 3.14159265358979323846;

inline int32_t pack_rgb(int r, int g, int b) {
    // the reference shifts plain ints and ORs them without masking (iRed<<16 | iGreen<<8 | iBlue)
    return int32_t((uint32_t(r) << 16) | (uint32_t(g) << 8) | uint32_t(b));
}

}  // namespace

extern "C" {

int jsg_abi_version(void) { return JSG_ABI_VERSION; }

int jsg_feed_samples(float feed_percent, int fftsize) {
    if (fftsize <= 0) return JSG_ERR_INVALID;
    // float * double * size_t -> double, + 0.5, truncate
    return int(double(feed_percent) * 0.01 * double(fftsize) + 0.5);
}

int jsg_memsize_blocks(float memsize_s, float fs, int feed_samples) {
    if (feed_samples <= 0) return JSG_ERR_INVALID;
    const float prod = memsize_s * fs;            // float * float
    const float q = prod / float(feed_samples);   // float / int -> float
    return int(double(q) + 0.5);
}

int jsg_next_power_of_2(float fftsize_ms, float fs) {
    const float first = float(double(fftsize_ms) * 0.001 * double(fs));
    if (!(first > 0.f)) return JSG_ERR_INVALID;
    const int e = int(std::log(double(first)) / double(std::log(2.f))) + 1;
    return int(std::pow(2.f, float(e)));
}

int jsg_window_build(int window, int n, float* out) {
    if (n <= 0 || out == nullptr) return JSG_ERR_INVALID;
    if (window < JSG_WIN_RECT || window > JSG_WIN_HANNPOISSON) return JSG_ERR_INVALID;
    const size_t N = size_t(n);
    float norm = 0.f;
    for (size_t k = 0; k < N; ++k) {
        const double c1 = std::cos(2.0 * kPi * double(k) / double(N));
        float v = 0.f;
        switch (window) {
            case JSG_WIN_RECT: v = 1.f; break;
            case JSG_WIN_HANN: v = float(0.5 * (1.0 - c1)); break;
            case JSG_WIN_HAMMING: v = float(25.0 / 46.0 - (1.0 - 25.0 / 46.0) * c1); break;
            case JSG_WIN_BLACKMANHARRIS: {
                const float a0 = 0.35875f, a1 = 0.48829f, a2 = 0.14128f, a3 = 0.01168f;
                v = float(double(a0) - double(a1) * c1 + double(a2) * std::cos(4.0 * kPi * double(k) / double(N)) -
                          double(a3) * std::cos(6.0 * kPi * double(k) / double(N)));
                break;
            }
            case JSG_WIN_FLATTOP: {
                const float a0 = 0.21557895f, a1 = 0.41663158f, a2 = 0.277263158f, a3 = 0.083578947f,
                            a4 = 0.006947368f;
                v = float(double(a0) - double(a1) * c1 + double(a2) * std::cos(4.0 * kPi * double(k) / double(N)) -
                          double(a3) * std::cos(6.0 * kPi * double(k) / double(N)) +
                          double(a4) * std::cos(8.0 * kPi * double(k) / double(N)));
                break;
            }
            case JSG_WIN_HANNPOISSON: {
                // the reference evaluates fabs(N - 2k) in unsigned arithmetic: for 2k > N the difference
                // wraps to ~1.8e19 and the exponential underflows to exactly 0 (upper half of the window).
                const size_t d = N - 2 * k;
                const double e = std::exp(-2.0 * double(d) / double(N));
                v = float(0.5 * (1.0 - c1) * e);
                break;
            }
        }
        out[k] = v;
        norm += v * v;  // float accumulator, ascending k
    }
    norm /= float(N);
    norm = std::sqrt(norm);
    for (size_t k = 0; k < N; ++k) out[k] /= norm;
    return JSG_OK;
}

int jsg_colormap_build(int n_colors, int scheme, int32_t* lut_out) {
    if (n_colors <= 0 || lut_out == nullptr) return JSG_ERR_INVALID;
    if (scheme < JSG_CM_MONO || scheme > JSG_CM_JADE) return JSG_ERR_INVALID;
    const int N = n_colors;
    const int half = N / 2;
    for (int kk = 0; kk < N; ++kk) {
        const float fk = float(kk);
        int r = 0, g = 0, b = 0;
        switch (scheme) {
            case JSG_CM_MONO:
                r = g = b = (kk <= half) ? 0 : 255;
                break;
            case JSG_CM_BW:
                r = g = b = int(255.f * fk / float(N));
                break;
            case JSG_CM_RAINBOW: {
                const float slope = 4.f / float(N);
                if (kk < N / 8) {
                    b = int(double(255.f) * (double(fk * slope) + 0.5));
                } else if (kk < 3 * N / 8) {
                    b = 255;
                    g = int(255.f * float(kk - N / 8) * slope);
                } else if (kk < 5 * N / 8) {
                    b = int(255.f * float(1.f - float(kk - 3 * N / 8) * slope));
                    g = 255;
                    r = int(255.f * float(kk - 3 * N / 8) * slope);
                } else if (kk < 7 * N / 8) {
                    g = int(255.f * float(1.f - float(kk - 5 * N / 8) * slope));
                    r = 255;
                } else {
                    r = int(255.f * float(1.f - float(kk - 7 * N / 8) * slope));
                }
                break;
            }
            case JSG_CM_HOT: {
                const float s1 = 8.f / float(3 * N);
                const float s2 = 8.f / float(2 * N);
                if (kk < 3 * N / 8) {
                    r = int(255.f * (fk * s1));
                } else if (kk < 6 * N / 8) {
                    g = int(255.f * float(kk - 3 * N / 8) * s1);
                    r = 255;
                } else {
                    b = int(255.f * float(kk - 6 * N / 8) * s2);
                    g = 255;
                    r = 255;
                }
                break;
            }
            case JSG_CM_VIRIDIS:
            case JSG_CM_PLASMA: {
                const int index = int(fk / float(N) * float(256));
                const uint32_t c = (scheme == JSG_CM_VIRIDIS ? kViridisRgb24 : kPlasmaRgb24)[index & 255];
                r = int(c >> 16) & 255;
                g = int(c >> 8) & 255;
                b = int(c) & 255;
                break;
            }
            case JSG_CM_JADE: {
                const float rs = 0.3529f, rm = 0.89019f, re = 0.95f;
                const float gs = 0.372549f, gm = 0.023529f, ge = 0.95f;
                const float bs = 0.33725f, bm = 0.074509f, be = 0.95f;
                const int mix = 2 * N / 4;
                if (kk < mix) {
                    const float t = fk / float(mix);
                    b = int(255.f * (t * (bm - bs) + bs));
                    g = int(255.f * (t * (gm - gs) + gs));
                    r = int(255.f * (t * (rm - rs) + rs));
                } else {
                    const float t = float(kk - mix) / float(mix);
                    b = int(255.f * (t * (be - bm) + bm));
                    g = int(255.f * (t * (ge - gm) + gm));
                    r = int(255.f * (t * (re - rm) + rm));
                }
                break;
            }
        }
        lut_out[kk] = pack_rgb(r, g, b);
    }
    return JSG_OK;
}

int jsg_colormap_range(int n_colors, float lo, float hi, float* vmin, float* vmax, float* access_mult) {
    if (n_colors <= 0) return JSG_ERR_INVALID;
    float mn, mx;
    if (hi >= lo) { mn = lo; mx = hi; } else { mn = hi; mx = lo; }
    if (mx == mn) mn = float(0.99 * double(mx));
    if (vmin) *vmin = mn;
    if (vmax) *vmax = mx;
    if (access_mult) *access_mult = float(n_colors) / (mx - mn);
    return JSG_OK;
}

int jsg_display_freq_rows(float fs, int height, float min_freq, float max_freq, int* start_pixel, int* end_pixel,
                          int* height_interval, int* h_start) {
    if (!(fs > 0.f) || height <= 0) return JSG_ERR_INVALID;
    float mn = min_freq, mx = max_freq;
    if (double(mn) >= double(fs) * 0.5) mn = float(0.9 * double(fs) * 0.5);   // Spectrogram.cpp:444-445
    if (double(mx) >= double(fs) * 0.5) mx = float(double(fs) * 0.5);         // :446-447
    if (mn >= mx) mn = float(0.9 * double(mx));                               // :449-453
    const double lo = 2.0 * double(mn) / double(fs) * double(height);
    const double hi = 2.0 * double(mx) / double(fs) * double(height);
    const int start = int(lo + 0.5), end = int(hi + 0.5);                     // :455-456
    if (start_pixel) *start_pixel = start;
    if (end_pixel) *end_pixel = end;
    if (height_interval) *height_interval = int(hi - lo + 0.5);               // :457
    if (h_start) *h_start = height - end;                                     // :459
    return JSG_OK;
}

}  // extern "C"
