// The wait-free producer of the real engine under a geometry storm (C-ABI, GPU box): an audio thread pushes blocks with
// jsg_process_block_n as fast as the queue takes them, sized for whatever geometry it was last TOLD about (a host class learns a new
// FFT size only when its combo-box callback has run, reference Spectrogram.cpp:760-767), while a message thread keeps changing the FFT
// size and the channel count and reads the ring in between.  Nothing may crash, fail or read past a block: blocks of a stale
// geometry, blocks that arrive during a change and blocks that find the ring full are dropped and counted, everything else is
// processed.  Afterwards the storm stops, the engine is given a known configuration and a known signal, and its ring must equal that
// of a fresh engine fed the same signal in one batch.
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/jsg.h"

#define CK(call)                                                                         \
    do {                                                                                 \
        int rc_ = (call);                                                                \
        if (rc_ < 0) {                                                                   \
            std::fprintf(stderr, "%s -> %d: %s\n", #call, rc_, jsg_last_error(nullptr)); \
            return 2;                                                                    \
        }                                                                                \
    } while (0)

int main(int argc, char** argv) {
    const int changes = argc > 1 ? std::atoi(argv[1]) : 60;
    jsg_engine* e = nullptr;
    CK(jsg_create(&e, 2));
    CK(jsg_set_samplerate(e, 48000.f));
    CK(jsg_set_memory_time_s(e, 2.f));
    CK(jsg_set_fft_size(e, 1024));
    CK(jsg_set_feed_percent(e, JSG_FEED_50));
    std::atomic<int> told_n{1024}, told_c{2};      // what the audio thread believes
    std::atomic<bool> stop{false};
    std::atomic<long> pushed{0}, queued{0}, dropped_rc{0}, errors{0};
    std::vector<float> buf(8 * 8192, 0.25f);
    std::thread audio([&] {
        const float* ptrs[8];
        while (!stop.load()) {
            const int n = told_n.load(), c = told_c.load();
            for (int i = 0; i < c; ++i) ptrs[i] = buf.data() + size_t(i) * 8192;
            const int rc = jsg_process_block_n(e, ptrs, c, n);
            ++pushed;
            if (rc == 0) ++queued;
            else if (rc == 1) ++dropped_rc;
            else ++errors;
            if ((pushed.load() & 63) == 0) std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    });
    const int sizes[] = {1024, 2048, 512, 4096, 1024, 8192};
    std::vector<float> mem;
    long reads = 0;
    for (int k = 0; k < changes; ++k) {
        const int n = sizes[k % 6], c = (k % 5 == 4) ? 1 : 2;
        CK(jsg_set_fft_size(e, n));
        if (c != jsg_get_channels(e)) CK(jsg_set_channels(e, c));
        // the "combo-box callback": the audio thread learns the new geometry a little later than the engine has it
        std::this_thread::sleep_for(std::chrono::microseconds(300));
        told_n.store(n);
        told_c.store(c);
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
        const int W = jsg_get_memory_size(e), H = jsg_get_spectrum_size(e);
        mem.resize(size_t(W) * H);
        int pos = 0;
        CK(jsg_peek_mem(e, mem.data(), W, &pos));
        ++reads;
    }
    stop = true;
    audio.join();
    CK(jsg_sync(e));
    const long long dropped = jsg_get_dropped_blocks(e);
    // a known configuration and signal: the engine must behave like a fresh one
    CK(jsg_set_channels(e, 2));
    CK(jsg_set_fft_size(e, 1024));
    jsg_engine* ref = nullptr;
    CK(jsg_create(&ref, 2));
    CK(jsg_set_samplerate(ref, 48000.f));
    CK(jsg_set_memory_time_s(ref, 2.f));
    CK(jsg_set_fft_size(ref, 1024));
    CK(jsg_set_feed_percent(ref, JSG_FEED_50));
    const int K = 40, N = 1024;
    std::vector<float> x(size_t(2) * K * N);
    for (size_t i = 0; i < x.size(); ++i) x[i] = 0.4f * std::sin(0.013f * float(i % 7919)) + 0.05f * float(int(i * 2654435761u >> 20) % 17 - 8);
    for (int b = 0; b < K; ++b) {
        const float* ptrs[2] = {x.data() + size_t(b) * N, x.data() + size_t(K) * N + size_t(b) * N};
        int rc;
        while ((rc = jsg_process_block_n(e, ptrs, 2, N)) == 1) std::this_thread::sleep_for(std::chrono::microseconds(100));   // (ring full: try again)
        if (rc < 0) { std::fprintf(stderr, "process_block: %s\n", jsg_last_error(e)); return 2; }
    }
    CK(jsg_process_blocks(ref, x.data(), int64_t(K) * N, K));
    const int W = jsg_get_memory_size(e), H = jsg_get_spectrum_size(e);
    std::vector<float> a(size_t(W) * H), r(size_t(W) * H);
    int pa = -1, pr = -1;
    CK(jsg_peek_mem(e, a.data(), W, &pa));
    CK(jsg_peek_mem(ref, r.data(), W, &pr));
    size_t diff = 0;
    for (size_t i = 0; i < a.size(); ++i) diff += std::memcmp(&a[i], &r[i], 4) != 0;
    std::printf("{\"geometry_changes\": %d, \"pushed\": %ld, \"queued\": %ld, \"dropped_by_return_code\": %ld, \"dropped_blocks_counter\": %lld, \"errors\": %ld, "
                "\"reads\": %ld, \"pos\": %d, \"pos_ref\": %d, \"differing_floats_after_the_storm\": %zu}\n",
                changes, pushed.load(), queued.load(), dropped_rc.load(), dropped, errors.load(), reads, pa, pr, diff);
    jsg_destroy(e);
    jsg_destroy(ref);
    return (errors.load() == 0 && diff == 0 && pa == pr && dropped == dropped_rc.load() && queued.load() > 0) ? 0 : 1;
}
