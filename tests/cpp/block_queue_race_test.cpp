// CPU-only, built with -fsanitize=thread (and once more with address,undefined): the engine's block queue (csrc/jsg_block_queue.h) with
// its three parties -- the audio thread pushing blocks as fast as it can, the worker thread consuming them in order, and the message
// thread changing the slot geometry (channel count / FFT size) every now and then -- exactly the code the engine runs, minus HIP
// (plain memory instead of page-locked memory, a checksum instead of the H2D copy).
// Checks: no data race (the sanitizer), every consumed block is intact (all its samples carry the sequence number it was pushed with,
// in the geometry it was pushed with), the consumed sequence is strictly increasing (FIFO, nothing twice), pushed = consumed + dropped.
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "../../jadespectrogram_amd/csrc/jsg_block_queue.h"

int main(int argc, char** argv) {
    const long total = argc > 1 ? std::atol(argv[1]) : 200000;
    jsg::BlockQueue q;
    std::vector<float> storage;
    auto configure = [&](int ch, int n) {   // (between begin_ and end_geometry_change, or before the threads start)
        storage.assign(size_t(jsg::BlockQueue::kSlots) * size_t(ch) * size_t(n), -1.f);
        q.mem = storage.data();
        q.slot_floats = size_t(ch) * size_t(n);
        q.channels = ch;
        q.n = n;
    };
    configure(2, 256);
    std::atomic<bool> producer_done{false}, stop_gui{false};
    std::atomic<long> consumed{0}, bad{0}, queued{0};
    long last_seq = -1;

    std::thread worker([&] {
        for (;;) {
            unsigned long long t;
            const float* block;
            bool current;
            if (!q.front(t, block, current)) {
                if (producer_done.load() && q.tail.load() >= q.head.load()) return;
                std::this_thread::yield();
                continue;
            }
            if (current) {
                const long seq = long(block[0]);
                bool ok = seq > last_seq;
                for (size_t i = 0; i < q.slot_floats; ++i) ok = ok && block[i] == block[0];
                if (!ok) ++bad;
                last_seq = seq;
                ++consumed;
            } else ++bad;   // the setters drain the queue before they change the geometry: a stale block must never be seen
            q.pop(t);
        }
    });
    std::thread gui([&] {
        const int geo[][2] = {{2, 256}, {1, 512}, {3, 128}, {2, 1024}, {8, 64}};
        int k = 0;
        while (!stop_gui.load()) {
            std::this_thread::sleep_for(std::chrono::microseconds(300));
            q.begin_geometry_change();
            ++k;
            configure(geo[k % 5][0], geo[k % 5][1]);
            q.end_geometry_change();
        }
    });
    // the audio thread: sizes its block for the geometry it last saw (like a host class would) and says so
    std::vector<float> buf(8 * 1024);
    const float* ptrs[8];
    long pushed = 0, seq = 0;
    while (pushed < total) {
        // a real host knows its geometry from its own configuration; here every geometry is tried until one is taken
        static const int geo[][2] = {{2, 256}, {1, 512}, {3, 128}, {2, 1024}, {8, 64}};
        for (int g = 0; g < 5 && pushed < total; ++g) {
            const int c = geo[g][0], m = geo[g][1];
            for (int i = 0; i < c * m; ++i) buf[size_t(i)] = float(seq);
            for (int i = 0; i < c; ++i) ptrs[i] = buf.data() + size_t(i) * size_t(m);
            // every third attempt takes the lossless route (round 5: jsg_process_block_wait): ask first (can_push, what the all-or-nothing
            // sharded push does), then retry a FULL ring a few times without counting a drop; whatever is not taken in the end is counted
            int rc;
            if (pushed % 3 == 2) {
                (void)q.can_push(c, m);
                rc = q.try_push(ptrs, c, m, false);
                for (int spin = 0; rc == 1 && spin < 50; ++spin) {
                    std::this_thread::yield();
                    rc = q.try_push(ptrs, c, m, false);
                }
                if (rc > 0) { q.dropped.fetch_add(1); rc = 1; }
            } else rc = q.push(ptrs, c, m);
            ++pushed;
            if (rc == 0) { ++queued; ++seq; }
            else if (rc != 1) ++bad;
        }
    }
    producer_done = true;
    stop_gui = true;
    gui.join();
    worker.join();
    const long dropped = long(q.dropped.load());
    std::printf("{\"pushed\": %ld, \"queued\": %ld, \"consumed\": %ld, \"dropped\": %ld, \"bad\": %ld}\n", pushed, queued.load(), consumed.load(), dropped, bad.load());
    return (bad.load() == 0 && queued.load() == consumed.load() && queued.load() + dropped == pushed && consumed.load() > total / 50) ? 0 : 1;
}
