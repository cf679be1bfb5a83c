// CPU-only: every host-math entry point of the C-ABI (include/jsg.h, section 1 + jsg_display_freq_rows) over its whole
// parameter range, built together with csrc/jsg_host_math.cpp under -fsanitize=address,undefined.
// Prints an FNV-1a hash of everything it computed; tests/test_host_cpp.py compares it with the same walk through
// libjsg.so, so the sanitised build is also checked for equal results.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/jsg.h"

static uint64_t h = 1469598103934665603ull;
static void mix(const void* p, size_t n) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
}
template <class T> static void mix(T v) { mix(&v, sizeof v); }

int main() {
    const int sizes[] = {512, 1024, 2048, 4096, 8192};
    for (int n : sizes) {
        for (int w = -1; w <= 6; ++w) {   // -1 and 6 are invalid on purpose
            std::vector<float> win(size_t(n), -1.f);
            const int rc = jsg_window_build(w, n, win.data());
            mix(rc);
            if (rc == 0) mix(win.data(), win.size() * 4);
        }
        for (float pct : {100.f, 50.f, 25.f, 10.f, 12.5f, 0.01f, 0.f, -5.f}) {
            const int hop = jsg_feed_samples(pct, n);
            mix(hop);
            for (float fs : {8000.f, 44100.f, 48000.f, 96000.f, 192000.f})
                for (float sec : {0.001f, 1.f, 10.f}) mix(jsg_memsize_blocks(sec, fs, hop));
        }
    }
    mix(jsg_feed_samples(50.f, 0));
    mix(jsg_memsize_blocks(1.f, 48000.f, 0));
    for (float ms : {0.1f, 1.f, 5.f, 21.3f, 42.7f, 100.f, 1000.f})
        for (float fs : {8000.f, 48000.f, 96000.f}) mix(jsg_next_power_of_2(ms, fs));
    const int ncols[] = {1, 2, 3, 4, 7, 255, 256, 257, 1024, 4096, 65535};
    for (int nc : ncols)
        for (int scheme = -1; scheme <= 7; ++scheme) {
            std::vector<int32_t> lut(size_t(nc), -1);
            const int rc = jsg_colormap_build(nc, scheme, lut.data());
            mix(rc);
            if (rc == 0) mix(lut.data(), lut.size() * 4);
        }
    mix(jsg_colormap_build(0, 6, nullptr));
    const float ranges[][2] = {{-50.f, 50.f}, {50.f, -50.f}, {0.f, 0.f}, {-120.f, -120.f}, {1e-30f, 1e30f}, {-3.4e38f, 3.4e38f}, {7.f, 7.f}};
    for (auto& r : ranges)
        for (int nc : {1, 256, 65535}) {
            float a = 0, b = 0, m = 0;
            mix(jsg_colormap_range(nc, r[0], r[1], &a, &b, &m));
            mix(a); mix(b); mix(m);
        }
    for (int height : {1, 2, 257, 513, 1025, 2049, 4097})
        for (float fs : {8000.f, 48000.f, 96000.f})
            for (float lo : {0.f, 1.f, 100.f, 5000.f, 30000.f})
                for (float hi : {0.f, 500.f, 20000.f, 48000.f, 1e9f}) {
                    int s = 0, e = 0, hi_ = 0, hs = 0;
                    mix(jsg_display_freq_rows(fs, height, lo, hi, &s, &e, &hi_, &hs));
                    mix(s); mix(e); mix(hi_); mix(hs);
                }
    std::printf("%016llx\n", (unsigned long long)h);
    return 0;
}
