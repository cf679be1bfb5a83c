// The audio thread's block queue of the engine (jsg_engine.cpp, "Producer"): a single-producer / single-consumer ring of
// [channels][n] float slots with a wait-free push, and an epoch -- instead of a lock on the producer's side -- for the rare
// moments in which the slot geometry changes (channel count, FFT size).  No HIP in here: the engine allocates the slots in
// page-locked memory and its worker thread is the consumer; tests/cpp/block_queue_race_test.cpp runs the same code under
// ThreadSanitizer with plain memory.
//
//   producer (audio thread)     push()                       wait-free: 2 atomic RMWs, channels x memcpy, 1 atomic store
//   consumer (worker thread)    front() ... pop()            in order
//   anyone but the producer     wait_drained()               until everything pushed before the call has been popped
//   message thread              begin_geometry_change() ... end_geometry_change()
//
// Reference call path of the producer: PluginProcessor.cpp:145-150 -> Spectrogram::processSynchronBlock, Spectrogram.cpp:37-48.
#pragma once
#include <atomic>
#include <chrono>
#include <cstddef>
#include <cstring>
#include <thread>

namespace jsg {

struct BlockQueue {
    static const int kSlots = 64;
    // slot storage and geometry: written only between begin_ and end_geometry_change (no producer inside, queue drained)
    float* mem = nullptr;
    size_t slot_floats = 0;
    int channels = 0, n = 0;
    unsigned geom_gen = 0;                       // the (even) cfg_gen the geometry belongs to
    unsigned slot_gen[kSlots] = {};              // cfg_gen under which a slot was filled (ordered by `head`)
    std::atomic<unsigned long long> head{0};     // blocks published by the producer
    std::atomic<unsigned long long> tail{0};     // blocks the consumer is done with
    std::atomic<unsigned> cfg_gen{0};            // odd while the geometry changes
    std::atomic<int> inflight{0};                // producer calls inside push()
    std::atomic<unsigned long long> dropped{0};

    // 0: queued; 1: not taken, ring full; 2: not taken, geometry change in progress or the caller's geometry is not the queue's (also a
    // queue whose last geometry change failed: no slots); -2: null pointer.  ch_check / n_check > 0: the geometry the caller's pointers
    // were sized for.  `count_drop`: count a block that was not taken in `dropped` (the lossless caller retries a full ring instead).
    int try_push(const float* const* planar, int ch_check, int n_check, bool count_drop) {
        inflight.fetch_add(1);                   // (sequentially consistent with cfg_gen: Dekker-style hand-shake with the setter)
        const unsigned g = cfg_gen.load();
        int rc = 0;
        if (g & 1u) rc = 2;
        else if (!mem || channels <= 0 || n <= 0) rc = 2;   // the last geometry change failed: closed until a setter succeeds
        else if ((ch_check > 0 && ch_check != channels) || (n_check > 0 && n_check != n)) rc = 2;
        else {
            const unsigned long long h = head.load(std::memory_order_relaxed);
            if (h - tail.load(std::memory_order_acquire) >= (unsigned long long)kSlots) rc = 1;
            else {
                const unsigned slot = unsigned(h % kSlots);
                float* dst = mem + size_t(slot) * slot_floats;
                for (int c = 0; c < channels && rc == 0; ++c) {
                    if (!planar[c]) rc = -2;
                    else std::memcpy(dst + size_t(c) * size_t(n), planar[c], size_t(n) * sizeof(float));
                }
                if (rc == 0) {
                    slot_gen[slot] = g;
                    head.store(h + 1, std::memory_order_release);
                }
            }
        }
        if (rc > 0 && count_drop) dropped.fetch_add(1, std::memory_order_relaxed);
        inflight.fetch_sub(1);
        return rc;
    }
    // the wait-free producer call: 0 queued, 1 dropped (and counted), -2 null pointer
    int push(const float* const* planar, int ch_check, int n_check) {
        const int rc = try_push(planar, ch_check, n_check, true);
        return rc > 0 ? 1 : rc;
    }
    // would a push of this geometry be taken right now?  (single producer: free slots can only grow until its next push)
    // (the same hand-shake as try_push: the geometry fields may only be read between inflight++ and inflight-- with an even cfg_gen seen in
    // between -- the setter waits for inflight == 0 behind making cfg_gen odd; ThreadSanitizer found the unprotected first version)
    bool can_push(int ch_check, int n_check) {
        inflight.fetch_add(1);
        bool ok = false;
        if (!(cfg_gen.load() & 1u) && mem && channels > 0 && n > 0 &&
            !((ch_check > 0 && ch_check != channels) || (n_check > 0 && n_check != n)))
            ok = head.load(std::memory_order_relaxed) - tail.load(std::memory_order_acquire) < (unsigned long long)kSlots;
        inflight.fetch_sub(1);
        return ok;
    }

    // consumer: the oldest block, if any (`current`: it was pushed under the geometry that is in force)
    bool front(unsigned long long& t, const float*& data, bool& current) const {
        t = tail.load(std::memory_order_relaxed);
        if (t >= head.load(std::memory_order_acquire)) return false;
        const unsigned slot = unsigned(t % kSlots);
        data = mem + size_t(slot) * slot_floats;
        current = slot_gen[slot] == geom_gen;
        return true;
    }
    void pop(unsigned long long t) { tail.store(t + 1, std::memory_order_release); }

    void wait_drained() const {
        const unsigned long long target = head.load(std::memory_order_acquire);
        int spins = 0;
        while (tail.load(std::memory_order_acquire) < target) {
            if (++spins < 200) std::this_thread::yield();
            else std::this_thread::sleep_for(std::chrono::microseconds(50));   // (the engine's worker polls every 250 us)
        }
    }

    // The caller (message thread) then owns mem / slot_floats / channels / n until end_geometry_change.
    void begin_geometry_change() {
        cfg_gen.fetch_add(1);                                        // odd: producers drop from here on
        while (inflight.load() != 0) std::this_thread::yield();      // at most one call, a memcpy long
        wait_drained();                                              // what was pushed before is consumed under the old geometry
    }
    void end_geometry_change() {
        geom_gen = cfg_gen.load() + 1;
        cfg_gen.fetch_add(1);                                        // even again: the new geometry is visible to the producer
    }
};

}  // namespace jsg
