// CPU-only, built with -fsanitize=thread (and once more with address,undefined): the audio thread runs processBlock while
// the message thread changes the block size (the FFT-size combo box, reference Spectrogram.cpp:760-767).  The reference
// guards both with m_protect; the stand-in guards the FIFO with an epoch (no lock on the audio thread): it must be free of data
// races and must never hand out a block of a stale size.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

#include "../../jadespectrogram_amd/host/SynchronBlockProcessor.h"

struct Probe : SynchronBlockProcessor {
    std::atomic<long> calls{0}, bad{0};
    int processSynchronBlock(std::vector<std::vector<float>>& data, juce::MidiBuffer&) override {
        ++calls;
        // called inside the audio thread's epoch section: the size it was filled for is the size that is current
        if (data.size() != 2 || data[0].size() != getDesiredBlockSizeSamples() || data[1].size() != data[0].size()) ++bad;
        return 0;
    }
};

int main() {
    Probe p;
    p.preparetoProcess(2, 512);
    p.setDesiredBlockSizeSamples(1024);
    std::atomic<bool> stop{false};
    std::thread gui([&] {
        const size_t sizes[] = {512, 1024, 2048, 4096, 8192};
        for (int i = 0; i < 400; ++i) {
            p.setDesiredBlockSizeSamples(sizes[i % 5]);
            if (i % 50 == 0) p.preparetoProcess(2, 480);
            std::this_thread::sleep_for(std::chrono::microseconds(200));   // (the audio thread gets whole blocks through between resizes)
        }
        stop = true;
    });
    juce::MidiBuffer midi;
    juce::AudioBuffer<float> buf(2, 480);
    long host_blocks = 0;
    while (!stop.load()) {
        p.processBlock(buf, midi);
        ++host_blocks;
    }
    gui.join();
    std::printf("host blocks %ld (skipped during resizes: %llu), synchron blocks %ld, bad %ld\n", host_blocks, p.droppedHostBlocks(), p.calls.load(), p.bad.load());
    return (p.bad.load() == 0 && p.calls.load() > 0) ? 0 : 1;
}
