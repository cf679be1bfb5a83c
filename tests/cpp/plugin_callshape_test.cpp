// The plugin's call shapes against the drop-in headers (no JUCE in this image: juce_shim.h supplies the few types).
// A miniature processor/editor pair written for this test makes the same calls, in the same order and with the same
// argument types, as the reference does at
//   PluginProcessor.cpp:20-28    constructor: parameter tree, then m_spectrogram.prepareParameter(m_parameterVTS)
//   PluginProcessor.cpp:102-114  prepareToPlay: preparetoProcess / setSamplerate / setmemoryTime_s / setFFTSize / setfeed_percent
//   PluginProcessor.cpp:145-150  processBlock: m_spectrogram.processBlock(buffer, midiMessages)
//   Spectrogram.cpp:592-608      timerCallback: getMemorySize / getSpectrumSize / (re)size m_displaymem / getMem(m_displaymem, pos)
//   Spectrogram.cpp:409,735,767  combo boxes and pause button: setWindow / setPauseMode / setFFTSize
// Runs on the GPU box (it creates an engine); prints one JSON line that tests/test_host_cpp.py checks.
#include <cmath>
#include <cstdio>
#include <memory>
#include <vector>

#include "../../jadespectrogram_amd/host/Spectrogram.h"

using juce::AudioProcessorValueTreeState;

class MiniProcessor {
public:
    MiniProcessor() {
        m_parameterVTS = std::make_unique<AudioProcessorValueTreeState>();
#if !JSG_HAVE_JUCE
        m_parameterVTS->addRawParameter("MinFreq", std::log(1.f));
        m_parameterVTS->addRawParameter("MaxFreq", std::log(20000.f));
        m_parameterVTS->addRawParameter("MinColor", -50.f);
        m_parameterVTS->addRawParameter("MaxColor", 50.f);
#endif
        m_spectrogram.prepareParameter(m_parameterVTS);
    }
    void prepareToPlay(double sampleRate, int samplesPerBlock) {
        m_fs = sampleRate;
        m_spectrogram.preparetoProcess(2, samplesPerBlock);
        m_spectrogram.setSamplerate(float(sampleRate));
        m_spectrogram.setmemoryTime_s(10.0);
        m_spectrogram.setFFTSize(m_fftsize);
        m_spectrogram.setfeed_percent(Spectrogram::FeedPercentage::perc50);
    }
    void processBlock(juce::AudioBuffer<float>& buffer, juce::MidiBuffer& midiMessages) { m_spectrogram.processBlock(buffer, midiMessages); }
    Spectrogram m_spectrogram;
    std::unique_ptr<AudioProcessorValueTreeState> m_parameterVTS;
    double m_fs = 48000.0;
    size_t m_fftsize = 2048;
};

class MiniComponent {   // the engine-facing part of SpectrogramComponent::timerCallback
public:
    explicit MiniComponent(Spectrogram& s) : m_spectrogram(s) {}
    int timerCallback() {
        int memsize = m_spectrogram.getMemorySize();
        int freqsize = m_spectrogram.getSpectrumSize();
        if (memsize != m_memsize || freqsize != m_freqsize) {
            m_memsize = memsize;
            m_freqsize = freqsize;
            m_displaymem.resize(size_t(m_memsize));
            for (auto& v : m_displaymem) {
                v.resize(size_t(m_freqsize));
                std::fill(v.begin(), v.end(), -120.f);
            }
        }
        int pos = 0;
        int newVals = m_spectrogram.getMem(m_displaymem, pos);
        m_pos = pos;
        return newVals;
    }
    Spectrogram& m_spectrogram;
    std::vector<std::vector<float>> m_displaymem;
    int m_memsize = 0, m_freqsize = 0, m_pos = 0;
};

int main() {
    MiniProcessor proc;
    const auto& dp = proc.m_spectrogram.displayParameters();
    if (!dp.m_DisplayMinFreq || !dp.m_DisplayMaxColor || dp.m_DisplayMaxColor->load() != 50.f) return 3;
    proc.prepareToPlay(48000.0, 480);
    MiniComponent gui(proc.m_spectrogram);
    juce::MidiBuffer midi;
    // 1 kHz full-scale sine on both channels, host blocks of 480 samples
    long t = 0;
    int total_new = 0, ticks = 0;
    for (int blk = 0; blk < 64; ++blk) {
        juce::AudioBuffer<float> buf(2, 480);
        for (int i = 0; i < 480; ++i, ++t) {
            const float v = std::sin(2.0 * M_PI * 1000.0 * double(t) / 48000.0);
            buf.getWritePointer(0)[i] = v;
            buf.getWritePointer(1)[i] = v;
        }
        proc.processBlock(buf, midi);
        if (blk % 16 == 15) {   // a 25 Hz timer beside a 100 Hz audio callback
            const int nv = gui.timerCallback();
            if (ticks++ > 0) total_new += nv;   // the first call reports the reference's start-up sentinel
        }
    }
    // FFT-size combo box while audio runs (reference Spectrogram.cpp:760-767), then more audio
    proc.m_spectrogram.setFFTSize(1024);
    proc.m_spectrogram.setWindow(static_cast<Spectrogram::Windows>(1));
    proc.m_spectrogram.setPauseMode(false);
    for (int blk = 0; blk < 16; ++blk) {
        juce::AudioBuffer<float> buf(2, 480);
        for (int i = 0; i < 480; ++i, ++t) buf.getWritePointer(0)[i] = buf.getWritePointer(1)[i] = std::sin(2.0 * M_PI * 1000.0 * double(t) / 48000.0);
        proc.processBlock(buf, midi);
    }
    const int nv_after = gui.timerCallback();
    // strongest bin of the newest column: 1 kHz at 48 kHz / 1024 -> bin 21
    const int W = gui.m_memsize, H = gui.m_freqsize;
    const int newest = (gui.m_pos - 1 + W) % W;
    int arg = 0;
    for (int k = 1; k < H; ++k)
        if (gui.m_displaymem[size_t(newest)][size_t(k)] > gui.m_displaymem[size_t(newest)][size_t(arg)]) arg = k;
    if (proc.m_spectrogram.getSamplerate() != 48000.f) return 4;   // SpectrogramComponent::paint, reference Spectrogram.cpp:439
    // an unsupported size must not throw out of a GUI callback
    proc.m_spectrogram.setFFTSize(1000);
    std::printf("{\"W\": %d, \"H\": %d, \"new_between_ticks\": %d, \"new_after_resize\": %d, \"peak_bin\": %d, \"peak_db\": %.4f, "
                "\"fft_after_bad_setter\": %d, \"last_error_set\": %d}\n",
                W, H, total_new, nv_after, arg, double(gui.m_displaymem[size_t(newest)][size_t(arg)]),
                int(proc.m_spectrogram.getDesiredBlockSizeSamples()), int(!proc.m_spectrogram.lastError().empty()));
    return 0;
}
