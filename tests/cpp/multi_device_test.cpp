// Several GPUs from ONE host process through the C-ABI ("Several GPUs"): one jsg_engine per device,
// channels sharded contiguously, every engine fed from its own host thread, outputs stay on their device -- the path
// shards by independent channels, so there is no collective.  On a box with one GPU the shards share it.
//   usage: multi_device_test [n_shards]      (default: one shard per visible device, at least 2)
// Check: every channel's spectrogram from the sharded run equals the same channel of ONE engine that holds all channels
// (per-channel mode), bit for bit.  Prints one JSON line.
#include <algorithm>
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
            std::exit(2);                                                                \
        }                                                                                \
    } while (0)

static void configure(jsg_engine* e) {
    CK(jsg_set_samplerate(e, 48000.f));
    CK(jsg_set_memory_time_s(e, 1.f));
    CK(jsg_set_fft_size(e, 1024));
    CK(jsg_set_feed_percent(e, JSG_FEED_50));
    CK(jsg_set_mix_mode(e, JSG_MIX_PER_CHANNEL));
}

int main(int argc, char** argv) {
    const int ndev = jsg_device_count();
    if (ndev <= 0) { std::fprintf(stderr, "no device\n"); return 2; }
    const int shards = argc > 1 ? std::atoi(argv[1]) : std::max(2, ndev);
    const int C = 8, N = 1024, blocks = 12;
    std::vector<float> x(size_t(C) * blocks * N);
    for (int c = 0; c < C; ++c)
        for (int i = 0; i < blocks * N; ++i)
            x[size_t(c) * blocks * N + i] = 0.5f * std::sin(0.002f * float(c + 3) * float(i)) + 0.001f * float((i * 7919 + c * 104729) % 1000 - 500);

    // reference: one engine with all 8 channels
    jsg_engine* whole = nullptr;
    CK(jsg_create_on_device(&whole, C, 0));
    configure(whole);
    CK(jsg_process_blocks(whole, x.data(), int64_t(blocks) * N, blocks));
    const int W = jsg_get_memory_size(whole), H = jsg_get_spectrum_size(whole);
    std::vector<float> ref(size_t(C) * W * H);
    int pos_ref = 0;
    CK(jsg_peek_mem(whole, ref.data(), C * W, &pos_ref));

    // sharded: engine s owns channels [first[s], first[s] + count[s]) on device s % ndev
    std::vector<jsg_engine*> eng(size_t(shards), static_cast<jsg_engine*>(nullptr));
    const size_t ns = size_t(shards);
    std::vector<int> first(ns, 0), count(ns, 0);
    for (int s = 0; s < shards; ++s) {
        const int base = C / shards, extra = C % shards;
        first[size_t(s)] = s * base + std::min(s, extra);
        count[size_t(s)] = base + (s < extra ? 1 : 0);
        if (count[size_t(s)] == 0) continue;
        CK(jsg_create_on_device(&eng[size_t(s)], count[size_t(s)], s % ndev));
        if (jsg_get_device(eng[size_t(s)]) != s % ndev) return 3;
        configure(eng[size_t(s)]);
    }
    std::vector<std::thread> feeders;   // one host thread per engine: block by block, like audio callbacks
    for (int s = 0; s < shards; ++s) {
        if (!eng[size_t(s)]) continue;
        feeders.emplace_back([&, s] {
            std::vector<const float*> ptrs(size_t(count[size_t(s)]));
            for (int b = 0; b < blocks; ++b) {
                for (int c = 0; c < count[size_t(s)]; ++c) ptrs[size_t(c)] = x.data() + size_t(first[size_t(s)] + c) * blocks * N + size_t(b) * N;
                CK(jsg_process_block(eng[size_t(s)], ptrs.data()));
            }
        });
    }
    for (auto& t : feeders) t.join();
    size_t diff = 0;
    int pos_bad = 0;
    for (int s = 0; s < shards; ++s) {
        if (!eng[size_t(s)]) continue;
        std::vector<float> got(size_t(count[size_t(s)]) * W * H);
        int pos = -1;
        CK(jsg_peek_mem(eng[size_t(s)], got.data(), count[size_t(s)] * W, &pos));
        pos_bad += pos != pos_ref;
        diff += std::memcmp(got.data(), ref.data() + size_t(first[size_t(s)]) * W * H, got.size() * sizeof(float)) != 0;
        diff += jsg_get_dropped_blocks(eng[size_t(s)]) != 0;   // (a return value of 1 is not an error for CK: 12 blocks fit the 64-slot queue, none may be dropped)
        CK(jsg_destroy(eng[size_t(s)]));
    }
    // the same through the convenience entry points (jsg_create_sharded): one call creates the set, one call per block feeds it
    std::vector<jsg_engine*> set(ns, static_cast<jsg_engine*>(nullptr));
    std::vector<int> sfirst(ns, 0), scount(ns, 0), devs(ns, 0);
    for (int s = 0; s < shards; ++s) devs[size_t(s)] = s % ndev;
    CK(jsg_create_sharded(set.data(), sfirst.data(), scount.data(), devs.data(), shards, C));
    size_t api_diff = 0;
    for (int s = 0; s < shards; ++s) {
        api_diff += sfirst[size_t(s)] != first[size_t(s)] || scount[size_t(s)] != count[size_t(s)];
        if (set[size_t(s)]) configure(set[size_t(s)]);
    }
    std::vector<const float*> all(static_cast<size_t>(C));
    for (int b = 0; b < blocks; ++b) {
        for (int c = 0; c < C; ++c) all[size_t(c)] = x.data() + size_t(c) * blocks * N + size_t(b) * N;
        CK(jsg_process_block_sharded(set.data(), sfirst.data(), shards, all.data()));
    }
    for (int s = 0; s < shards; ++s) {
        if (!set[size_t(s)]) continue;
        std::vector<float> got(size_t(scount[size_t(s)]) * W * H);
        int pos = -1;
        CK(jsg_peek_mem(set[size_t(s)], got.data(), scount[size_t(s)] * W, &pos));
        api_diff += pos != pos_ref;
        api_diff += std::memcmp(got.data(), ref.data() + size_t(sfirst[size_t(s)]) * W * H, got.size() * sizeof(float)) != 0;
    }
    // A FULL ring: push far more blocks than the queues have slots, without a pause.  jsg_process_block_sharded is all or
    // nothing -- a block is taken by every engine or by none -- so after any number of drops all shards have taken the same blocks: same
    // dropped count, same ring position, and every shard's columns are those of its channels in ONE engine fed the accepted blocks only.
    long long storm_dropped = 0;
    size_t storm_diff = 0;
    int storm_blocks = 0;
    {
        // at least 600 blocks, and on until the all-or-nothing branch HAS been taken (a drop seen) -- at most 4000 (the branch may
        // never run on a box whose worker keeps up; it cannot keep up with pushes that cost a microsecond each, but the test now says so)
        const int storm_min = 600, storm_max = 4000;
        int storm = 0;
        std::vector<unsigned char> taken(static_cast<size_t>(storm_max), 0);
        std::vector<long long> before(ns, 0);
        for (int s = 0; s < shards; ++s)
            if (set[size_t(s)]) before[size_t(s)] = jsg_get_dropped_blocks(set[size_t(s)]);
        for (int b = 0; b < storm_max && (b < storm_min || storm_dropped == 0); ++b) {
            for (int c = 0; c < C; ++c) all[size_t(c)] = x.data() + size_t(c) * blocks * N + size_t(b % blocks) * N;
            const int rc = jsg_process_block_sharded(set.data(), sfirst.data(), shards, all.data());
            if (rc < 0) { std::fprintf(stderr, "storm push -> %d\n", rc); return 2; }
            taken[size_t(b)] = rc == 0;
            storm_dropped += rc == 1;
            storm = b + 1;
        }
        // the reference run: one engine, only the accepted blocks, in order, behind the 12 blocks it already holds
        std::vector<float> acc;
        int n_acc = 0;
        for (int b = 0; b < storm; ++b) n_acc += taken[size_t(b)];
        acc.resize(size_t(C) * size_t(std::max(1, n_acc)) * N);
        for (int c = 0; c < C; ++c) {
            int k = 0;
            for (int b = 0; b < storm; ++b)
                if (taken[size_t(b)]) std::memcpy(&acc[(size_t(c) * n_acc + k++) * N], x.data() + size_t(c) * blocks * N + size_t(b % blocks) * N, N * sizeof(float));
        }
        if (n_acc > 0) CK(jsg_process_blocks(whole, acc.data(), int64_t(n_acc) * N, n_acc));
        CK(jsg_peek_mem(whole, ref.data(), C * W, &pos_ref));
        for (int s = 0; s < shards; ++s) {
            if (!set[size_t(s)]) continue;
            storm_diff += jsg_get_dropped_blocks(set[size_t(s)]) - before[size_t(s)] != storm_dropped;
            std::vector<float> got(size_t(scount[size_t(s)]) * W * H);
            int pos = -1;
            CK(jsg_peek_mem(set[size_t(s)], got.data(), scount[size_t(s)] * W, &pos));
            storm_diff += pos != pos_ref;
            storm_diff += std::memcmp(got.data(), ref.data() + size_t(sfirst[size_t(s)]) * W * H, got.size() * sizeof(float)) != 0;
        }
        storm_blocks = storm;
    }
    CK(jsg_destroy_sharded(set.data(), shards));
    CK(jsg_destroy(whole));
    api_diff += storm_diff;
    std::printf("{\"storm_blocks\": %d, \"storm_dropped_on_every_shard\": %lld, \"storm_shards_out_of_step\": %zu}\n", storm_blocks, storm_dropped, storm_diff);
    std::printf("{\"devices\": %d, \"shards\": %d, \"channels\": %d, \"columns\": %d, \"shards_differing\": %zu, \"pos_mismatch\": %d, \"sharded_api_differing\": %zu}\n", ndev,
                shards, C, 2 * blocks, diff, pos_bad, api_diff);
    return diff == 0 && pos_bad == 0 && api_diff == 0 ? 0 : 1;
}
