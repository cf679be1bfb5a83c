// Audio-thread latency of jsg_process_block (wait-free since round 4: a lock-free ring + a worker thread of the engine) while a GUI
// thread reads as hard as it can (C-ABI, GPU box).
//   usage: producer_latency_test [blocks = 300] [pace_us = 500] [period = 256]      (the signal repeats after `period` blocks)
//   C5 geometry: stereo 96 kHz, 4096-point FFT, hop 512 (87.5 % overlap), 10 s memory -> ring 1875 x 2049, 15 MB image.
// The producer pushes `blocks` fft-size blocks at a real-time-like pace and records how long every call takes; a consumer
// thread alternates jsg_get_mem (up to 15 MB device-to-host) and jsg_display_update (colour kernel + 15 MB image copy)
// without pause.  Afterwards the ring must be bit-identical to a second engine that was fed the same samples in one
// batch, undisturbed.  Prints one JSON line; the pytest wrapper asserts the bounds.
#include <pthread.h>
#include <sched.h>
#include <sys/resource.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/jsg.h"

#define CK(call)                                                                     \
    do {                                                                             \
        int rc_ = (call);                                                            \
        if (rc_ < 0) {                                                               \
            std::fprintf(stderr, "%s -> %d: %s\n", #call, rc_, jsg_last_error(nullptr)); \
            return 2;                                                                \
        }                                                                            \
    } while (0)

static int configure(jsg_engine* e) {
    CK(jsg_set_samplerate(e, 96000.f));
    CK(jsg_set_memory_time_s(e, 10.f));
    CK(jsg_set_fft_size(e, 4096));
    CK(jsg_set_feed_percent_ext(e, 12.5f));
    return 0;
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? std::atoi(argv[1]) : 300;
    const int pace_us = argc > 2 ? std::atoi(argv[2]) : 500;   // pause between blocks (a 4096-sample block at 96 kHz lasts 42.7 ms)
    const int period = std::max(1, std::min(blocks, argc > 3 ? std::atoi(argv[3]) : 256));
    const int C = 2, N = 4096;
    jsg_engine *live = nullptr, *batch = nullptr;
    CK(jsg_create(&live, C));
    CK(jsg_create(&batch, C));
    if (configure(live) || configure(batch)) return 2;
    const int W = jsg_get_memory_size(live), H = jsg_get_spectrum_size(live);
    std::vector<float> x(size_t(C) * size_t(period) * N);
    unsigned s = 777u;
    for (size_t i = 0; i < x.size(); ++i) {
        s = s * 1664525u + 1013904223u;
        x[i] = 0.4f * std::sin(0.01f * float(i % 9973)) + 0.2f * (float(int(s >> 9)) / 4194304.0f - 1.0f);
    }
    const size_t chan_pitch = size_t(period) * N;   // planar: channel c at x[c*chan_pitch ...]

    std::atomic<bool> stop{false};
    std::atomic<long> reads{0}, read_columns{0};
    std::atomic<int> reader_rc{0};
    std::thread consumer([&] {
        std::vector<float> mem(size_t(W) * H);
        std::vector<uint32_t> img(size_t(W) * H);
        int k = 0;
        while (!stop.load()) {
            int pos = 0, nv = 0;
            int rc = (k++ & 1) ? jsg_get_mem(live, mem.data(), W, &pos) : jsg_display_update(live, -50.f, 50.f, img.data(), W, &nv, &pos);
            if (rc < 0) { reader_rc.store(rc); break; }
            if (k & 1) read_columns += rc < 1000000 ? rc : 0;
            ++reads;
            if ((k % 7) == 0) jsg_display_invalidate(live);   // force full recolours as well
        }
    });

    // An audio thread has a core (and usually a real-time priority) of its own; an ordinary user cannot ask for the priority, but it can
    // keep the measurement's own busy threads off the producer's core: the producer takes the last CPU this process may use, the
    // consumer (and whatever else the process starts) the others.
    {
        cpu_set_t all;
        CPU_ZERO(&all);
        if (sched_getaffinity(0, sizeof all, &all) == 0 && CPU_COUNT(&all) >= 4) {
            int last = -1;
            for (int c = 0; c < CPU_SETSIZE; ++c) if (CPU_ISSET(c, &all)) last = c;
            cpu_set_t mine, rest = all;
            CPU_ZERO(&mine);
            CPU_SET(last, &mine);
            CPU_CLR(last, &rest);
            (void)pthread_setaffinity_np(consumer.native_handle(), sizeof rest, &rest);
            (void)pthread_setaffinity_np(pthread_self(), sizeof mine, &mine);
        }
    }
    // Attribution of the tail: beside the wall-clock time of every call, the CPU time THIS THREAD spent in it
    // (CLOCK_THREAD_CPUTIME_ID: stands still while the thread is descheduled) and the thread's involuntary context switches
    // (getrusage(RUSAGE_THREAD).ru_nivcsw) around it.  A call that is long on the wall clock but short in thread-CPU time was taken off its
    // core by the host's scheduler; a call that is long in thread-CPU time would be the library's own doing.
    auto thread_cpu_us = [] {
        timespec ts;
        clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
        return double(ts.tv_sec) * 1e6 + double(ts.tv_nsec) * 1e-3;
    };
    auto nivcsw = [] {
        rusage ru;
        getrusage(RUSAGE_THREAD, &ru);
        return ru.ru_nivcsw;
    };
    // ... and a CONTROL on the same thread, same core, same moment: right after every real call the same number of bytes is copied into an
    // ordinary private buffer and one atomic is touched -- no library, no page-locked memory, no other thread -- and timed the same way.  A
    // tail that the control shows too belongs to the box (interrupt handlers and hypervisor time are charged to whatever thread is running
    // and are NOT context switches), not to jsg_process_block.
    // Round 6: the ORDER of call and control alternates from block to block.  Whatever runs FIRST after the pacing sleep
    // meets the wake-up's leftovers (the timer interrupt's soft-irq work, a core coming out of an idle state, cold caches and TLB); whatever
    // runs second runs warm.  A control that always ran second could never show that tail.  With the order alternating, each of the two is
    // "first after the sleep" in half of the blocks, and the claim "the tail belongs to the box, not to jsg_process_block" becomes a relation
    // that can be asserted: per position, the call's count of long executions is of the order of the control's -- else the call is at fault.
    std::vector<float> ctrl_dst(size_t(C) * N);
    std::atomic<unsigned long long> ctrl_word{0};
    size_t ctrl_over50 = 0, ctrl_cpu_over50 = 0;
    double ctrl_max = 0.0;
    size_t pos_over50[2][2] = {{0, 0}, {0, 0}}, pos_cpu_over50[2][2] = {{0, 0}, {0, 0}}, pos_n[2][2] = {{0, 0}, {0, 0}};   // [what: 0 call, 1 control][position: 0 first, 1 second]
    double pos_max[2][2] = {{0, 0}, {0, 0}};
    std::vector<double> ctrl_lat;
    ctrl_lat.reserve(size_t(blocks));
    // interrupts served by the producer's CPU during the run (/proc/interrupts, the column of that CPU)
    auto irqs_on_cpu = [](int cpu) -> long long {
        FILE* f = std::fopen("/proc/interrupts", "r");
        if (!f) return -1;
        long long total = 0;
        char line[16384];
        if (!std::fgets(line, sizeof line, f)) { std::fclose(f); return -1; }      // header: CPU0 CPU1 ...
        int col = -1, k = 0;
        for (char* tok = std::strtok(line, " \t\n"); tok; tok = std::strtok(nullptr, " \t\n"), ++k)
            if (std::strncmp(tok, "CPU", 3) == 0 && std::atoi(tok + 3) == cpu) col = k;
        while (col >= 0 && std::fgets(line, sizeof line, f)) {
            char* tok = std::strtok(line, " \t\n");                                 // "NN:" label
            for (int c = 0; tok && c <= col; ++c) tok = std::strtok(nullptr, " \t\n");
            if (tok && tok[0] >= '0' && tok[0] <= '9') total += std::atoll(tok);
        }
        std::fclose(f);
        return total;
    };
    const int my_cpu = sched_getcpu();
    const long long irq_before = irqs_on_cpu(my_cpu);
    std::vector<double> lat, cpu;
    std::vector<unsigned char> switched;
    lat.reserve(size_t(blocks));
    cpu.reserve(size_t(blocks));
    switched.reserve(size_t(blocks));
    const long nivcsw_before = nivcsw();
    for (int b = 0; b < blocks; ++b) {
        const float* ptrs[2] = {x.data() + size_t(b % period) * N, x.data() + chan_pitch + size_t(b % period) * N};
        const int control_first = b & 1;
        for (int step = 0; step < 2; ++step) {
            const bool is_control = (step == 0) == (control_first != 0);
            if (!is_control) {
                const long sw0 = nivcsw();
                const double c0 = thread_cpu_us();
                const auto t0 = std::chrono::steady_clock::now();
                const int rc = jsg_process_block(live, ptrs);
                const auto t1 = std::chrono::steady_clock::now();
                const double c1 = thread_cpu_us();
                const long sw1 = nivcsw();
                if (rc < 0) { std::fprintf(stderr, "process_block: %s\n", jsg_last_error(live)); stop = true; consumer.join(); return 2; }
                const double w = std::chrono::duration<double, std::micro>(t1 - t0).count();
                lat.push_back(w);
                cpu.push_back(c1 - c0);
                switched.push_back(sw1 != sw0);
                if (b > 0) {
                    ++pos_n[0][step];
                    pos_over50[0][step] += w > 50.0;
                    pos_cpu_over50[0][step] += (c1 - c0) > 50.0;
                    pos_max[0][step] = std::max(pos_max[0][step], w);
                }
            } else {   // the control: the same bytes into private memory, timed the same way
                const double k0 = thread_cpu_us();
                const auto u0 = std::chrono::steady_clock::now();
                ctrl_word.fetch_add(1);
                for (int c = 0; c < C; ++c) std::memcpy(ctrl_dst.data() + size_t(c) * N, ptrs[c], size_t(N) * sizeof(float));
                ctrl_word.fetch_add(1);
                const auto u1 = std::chrono::steady_clock::now();
                const double k1 = thread_cpu_us();
                const double w = std::chrono::duration<double, std::micro>(u1 - u0).count();
                ctrl_lat.push_back(w);
                if (b > 0) {
                    ctrl_over50 += w > 50.0;
                    ctrl_cpu_over50 += (k1 - k0) > 50.0;
                    ctrl_max = std::max(ctrl_max, w);
                    ++pos_n[1][step];
                    pos_over50[1][step] += w > 50.0;
                    pos_cpu_over50[1][step] += (k1 - k0) > 50.0;
                    pos_max[1][step] = std::max(pos_max[1][step], w);
                }
            }
        }
        if (pace_us > 0) std::this_thread::sleep_for(std::chrono::microseconds(pace_us));
    }
    const long nivcsw_total = nivcsw() - nivcsw_before;
    const long long irq_after = irqs_on_cpu(my_cpu);
    if (ctrl_dst[7] == 12345.678f) std::fprintf(stderr, "(keep the control copy)\n");
    // long calls (wall clock > 50 us): how many coincide with an involuntary switch, how many are long in thread-CPU time as well
    size_t long_calls = 0, long_with_switch = 0, long_cpu_over_50 = 0, cpu_over_50 = 0;
    double cpu_max = 0.0, cpu_max_of_long = 0.0;
    for (size_t i = 1; i < lat.size(); ++i) {   // (the first call pays the page faults of the fresh ring: reported on its own)
        cpu_max = std::max(cpu_max, cpu[i]);
        cpu_over_50 += cpu[i] > 50.0;
        if (lat[i] > 50.0) {
            ++long_calls;
            long_with_switch += switched[i];
            long_cpu_over_50 += cpu[i] > 50.0;
            cpu_max_of_long = std::max(cpu_max_of_long, cpu[i]);
        }
    }
    std::sort(ctrl_lat.begin(), ctrl_lat.end());
    const double ctrl_p50 = ctrl_lat.empty() ? 0.0 : ctrl_lat[ctrl_lat.size() / 2];
    std::vector<double> cpu_sorted(cpu);
    std::sort(cpu_sorted.begin(), cpu_sorted.end());
    stop = true;
    consumer.join();
    if (reader_rc.load() < 0) { std::fprintf(stderr, "reader failed: %s\n", jsg_last_error(live)); return 2; }
    CK(jsg_sync(live));

    const long long dropped = jsg_get_dropped_blocks(live);
    // reference run: the same samples in batches of one period, nobody reading
    for (int b = 0; b < blocks; b += period) CK(jsg_process_blocks(batch, x.data(), int64_t(chan_pitch), std::min(period, blocks - b)));
    // both rings, column for column, without touching the new-column counters
    std::vector<float> a(size_t(W) * H), r(size_t(W) * H);
    int pa = -1, pr = -1;
    CK(jsg_peek_mem(live, a.data(), W, &pa));
    CK(jsg_peek_mem(batch, r.data(), W, &pr));
    size_t diff_floats = 0;
    for (size_t i = 0; i < a.size(); ++i) diff_floats += std::memcmp(&a[i], &r[i], 4) != 0;
    // and the images of a full recolour
    std::vector<uint32_t> ia(size_t(W) * H), ib(size_t(W) * H);
    int nva = 0, nvb = 0, qa = 0, qb = 0;
    CK(jsg_display_invalidate(live));
    CK(jsg_display_invalidate(batch));
    CK(jsg_display_update(live, -50.f, 50.f, ia.data(), W, &nva, &qa));
    CK(jsg_display_update(batch, -50.f, 50.f, ib.data(), W, &nvb, &qb));
    size_t diff_px = 0;
    for (size_t i = 0; i < ia.size(); ++i) diff_px += ia[i] != ib[i];
    const double first_us = lat.empty() ? 0.0 : lat[0];
    size_t worst_at = 0;
    for (size_t i = 1; i < lat.size(); ++i) if (lat[i] > lat[worst_at]) worst_at = i;
    const double max_after_first = lat.size() > 1 ? *std::max_element(lat.begin() + 1, lat.end()) : 0.0;
    std::sort(lat.begin(), lat.end());
    auto pct = [&](double q) { return lat[size_t(q * double(lat.size() - 1))]; };
    size_t over50 = 0;
    for (double v : lat) over50 += v > 50.0;
    std::printf("{\"blocks\": %d, \"W\": %d, \"H\": %d, \"reads\": %ld, \"p50_us\": %.2f, \"p99_us\": %.2f, \"p9999_us\": %.2f, \"max_us\": %.1f, \"first_call_us\": %.1f, "
                "\"max_after_first_us\": %.1f, \"worst_call\": %zu, \"calls_over_50us\": %zu, \"dropped_blocks\": %lld, "
                "\"pos_live\": %d, \"pos_batch\": %d, \"differing_floats\": %zu, \"differing_pixels\": %zu, "
                "\"thread_cpu_p50_us\": %.2f, \"thread_cpu_p9999_us\": %.2f, \"thread_cpu_max_after_first_us\": %.1f, \"thread_cpu_calls_over_50us\": %zu, "
                "\"long_wall_calls_after_first\": %zu, \"long_wall_calls_with_involuntary_switch\": %zu, \"long_wall_calls_long_in_thread_cpu_too\": %zu, "
                "\"thread_cpu_max_of_long_wall_calls_us\": %.1f, \"involuntary_switches_total\": %ld, "
                "\"control_copy_calls_over_50us\": %zu, \"control_copy_thread_cpu_over_50us\": %zu, \"control_copy_max_us\": %.1f, "
                "\"control_copy_p50_us\": %.2f, "
                "\"first_after_sleep\": {\"call_n\": %zu, \"call_over_50us\": %zu, \"call_thread_cpu_over_50us\": %zu, \"call_max_us\": %.1f, "
                "\"control_n\": %zu, \"control_over_50us\": %zu, \"control_thread_cpu_over_50us\": %zu, \"control_max_us\": %.1f}, "
                "\"second_after_sleep\": {\"call_n\": %zu, \"call_over_50us\": %zu, \"call_thread_cpu_over_50us\": %zu, \"call_max_us\": %.1f, "
                "\"control_n\": %zu, \"control_over_50us\": %zu, \"control_thread_cpu_over_50us\": %zu, \"control_max_us\": %.1f}, "
                "\"producer_cpu\": %d, \"interrupts_on_producer_cpu_during_run\": %lld}\n",
                blocks, W, H, reads.load(), pct(0.5), pct(0.99), pct(0.9999), lat.back(), first_us, max_after_first, worst_at, over50, dropped, pa, pr, diff_floats, diff_px,
                cpu_sorted[size_t(0.5 * double(cpu_sorted.size() - 1))], cpu_sorted[size_t(0.9999 * double(cpu_sorted.size() - 1))], cpu_max, cpu_over_50,
                long_calls, long_with_switch, long_cpu_over_50, cpu_max_of_long, nivcsw_total,
                ctrl_over50, ctrl_cpu_over50, ctrl_max, ctrl_p50,
                pos_n[0][0], pos_over50[0][0], pos_cpu_over50[0][0], pos_max[0][0], pos_n[1][0], pos_over50[1][0], pos_cpu_over50[1][0], pos_max[1][0],
                pos_n[0][1], pos_over50[0][1], pos_cpu_over50[0][1], pos_max[0][1], pos_n[1][1], pos_over50[1][1], pos_cpu_over50[1][1], pos_max[1][1],
                my_cpu, (irq_before >= 0 && irq_after >= 0) ? irq_after - irq_before : -1LL);
    jsg_destroy(live);
    jsg_destroy(batch);
    return 0;
}
