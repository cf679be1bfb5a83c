// CPU-only test of the SynchronBlockProcessor stand-in: arbitrary host block sizes in, fixed blocks out, order kept.
#include <cstdio>
#include <vector>

#include "../../jadespectrogram_amd/host/SynchronBlockProcessor.h"

struct Probe : SynchronBlockProcessor {
    std::vector<float> seen[2];
    int calls = 0;
    size_t want = 0;
    int processSynchronBlock(std::vector<std::vector<float>>& data, juce::MidiBuffer&) override {
        ++calls;
        if (data.size() != 2 || data[0].size() != want) return -1;
        for (int c = 0; c < 2; ++c) seen[c].insert(seen[c].end(), data[size_t(c)].begin(), data[size_t(c)].end());
        return 0;
    }
};

int main() {
    Probe p;
    p.preparetoProcess(2, 512);
    p.setDesiredBlockSizeSamples(256);
    p.want = 256;
    juce::MidiBuffer midi;
    const int sizes[] = {1, 255, 256, 300, 7, 1000, 64, 165};   // 2048 samples in total
    float v = 0.f;
    for (int n : sizes) {
        juce::AudioBuffer<float> buf(2, n);
        for (int i = 0; i < n; ++i, v += 1.f) {
            buf.getWritePointer(0)[i] = v;
            buf.getWritePointer(1)[i] = -v;
        }
        p.processBlock(buf, midi);
    }
    if (p.calls != 8 || p.seen[0].size() != 2048) return 1;
    for (size_t i = 0; i < 2048; ++i)
        if (p.seen[0][i] != float(i) || p.seen[1][i] != -float(i)) return 2;
    // a mono host buffer on a stereo processor: the missing channel is zero-filled
    Probe q;
    q.preparetoProcess(2, 64);
    q.setDesiredBlockSizeSamples(64);
    q.want = 64;
    juce::AudioBuffer<float> mono(1, 64);
    for (int i = 0; i < 64; ++i) mono.getWritePointer(0)[i] = 1.f;
    q.processBlock(mono, midi);
    if (q.calls != 1 || q.seen[1][10] != 0.f || q.seen[0][10] != 1.f) return 3;
    std::puts("reblocker ok");
    return 0;
}
