// Drop-in replacement for the engine half of the reference's `class Spectrogram` (reference Spectrogram.h:81-169,
// Spectrogram.cpp:16-331).  Same public method names, argument meaning, return values and enumerator order; the
// body is a thin veneer over the C-ABI of libjsg.so, i.e. all signal processing runs in hand-written HIP kernels
// on an MI355X.  Header-only: add this directory to the include path and link libjsg.so.
//
//   PluginProcessor.cpp:28        m_spectrogram.prepareParameter(m_parameterVTS)                 -> unchanged
//   PluginProcessor.cpp:102-114   prepareToPlay: preparetoProcess / setSamplerate / setmemoryTime_s / setFFTSize /
//                                 setfeed_percent            -> unchanged call sites
//   PluginProcessor.cpp:148       m_spectrogram.processBlock(buffer, midi)   -> unchanged
//   Spectrogram.cpp:592-608       getMemorySize / getSpectrumSize / getMem   -> unchanged
//   Spectrogram.cpp:623-724       the colour loops -> SpectrogramGpuDisplay::update (one call instead of the loops)
//
// Differences from the reference that a maintainer should know:
//   * the channel count comes from preparetoProcess()/setchannels(); the reference keeps its ctor default of 2
//     unless the (external) base class sets it;
//   * there is no CPU fallback: if no MI355X is usable the constructor throws std::runtime_error;
//   * the setters never throw (they run inside JUCE callbacks): a failed setter leaves the engine as it was, prints the
//     reason to stderr and keeps it in lastError();
//   * processSynchronBlock only enqueues work and never waits for getMem / the display (jsg.h, "threading").
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/jsg.h"
#include "CColorpalette.h"
#include "SynchronBlockProcessor.h"

class Spectrogram : public SynchronBlockProcessor {
public:
    enum class ChannelMixMode { AbsMean, Max, Min, Left, Right };
    enum class Windows { Rect, Hann, Hamming, BlackmanHarris, FlatTop, HannPoisson };
    enum class FeedPercentage { perc100, perc50, perc25, perc10 };

    Spectrogram() : SynchronBlockProcessor() {
        jsg_engine* e = nullptr;
        const int rc = jsg_create(&e, 2);   // m_channels(2), Spectrogram.cpp:17
        if (rc != JSG_OK) throw std::runtime_error(std::string("Spectrogram: ") + jsg_last_error(nullptr));
        m_engine.reset(e);
        m_ptrs.resize(64);   // (grown by setchannels for wider buses)
        setDesiredBlockSizeSamples(size_t(jsg_get_fft_size(e)));
    }

    virtual int processSynchronBlock(std::vector<std::vector<float>>& data, juce::MidiBuffer& midiMessages) override {
        juce::ignoreUnused(midiMessages);
        // Wait-free: the block goes into the engine's lock-free ring together with the geometry it was sized for; the engine drops it
        // (returns 1) when that is not its current geometry -- the FFT-size combo box has just been moved and the re-blocker has not
        // been resized yet (setFFTSize below), or the ring is full -- instead of reading past the rows.
        const size_t ch = data.size();
        if (ch == 0 || ch > m_ptrs.size()) return JSG_ERR_SIZE_MISMATCH;   // (m_ptrs is sized on the message thread: channelsPrepared)
        for (size_t c = 0; c < ch; ++c) m_ptrs[c] = data[c].data();
        // An offline bounce (setNonRealtime(true)) may run faster than the GPU takes blocks out: there the call waits for a free slot
        // instead of dropping -- the reference never loses a block (Spectrogram.cpp:37-135).
        if (m_nonRealtime.load(std::memory_order_acquire))   // (acquire: pairs with setNonRealtime's release, the timeout below is the one stored before it)
            return jsg_process_block_wait(m_engine.get(), m_ptrs.data(), int(ch), int(data[0].size()), m_nonRealtimeTimeoutMs.load(std::memory_order_relaxed));
        return jsg_process_block_n(m_engine.get(), m_ptrs.data(), int(ch), int(data[0].size()));
    }
    long long droppedBlocks() const { return jsg_get_dropped_blocks(m_engine.get()); }   // blocks the engine did not take (see above)
    // Mirror of juce::AudioProcessor::setNonRealtime: the processor forwards its own flag (prepareToPlay / processBlock:
    // m_spectrogram.setNonRealtime(isNonRealtime())).  false (default): wait-free and lossy when the ring is full (a live host);
    // true: lossless, the call may wait up to timeoutMs for the engine's worker (default 5 s: a worker that stalls -- a hung GPU -- must not
    // hold the host's render thread forever; the block then counts as dropped and 1 is returned; < 0: no limit, the caller's choice).
    void setNonRealtime(bool nonRealtime, int timeoutMs = 5000) {
        m_nonRealtimeTimeoutMs.store(timeoutMs, std::memory_order_relaxed);
        m_nonRealtime.store(nonRealtime, std::memory_order_release);
    }
    // Spectrogram::prepareParameter (reference Spectrogram.cpp:25-35): remember where the four display parameters live.
    // Nothing on the GPU depends on them; the IDs and defaults are those of reference Spectrogram.h:22-58
    // (frequencies are log-Hz, colour limits g_minColorVal / g_maxColorVal = -50 / +50 dB, PlugInGUISettings.h:37-38).
    void prepareParameter(std::unique_ptr<juce::AudioProcessorValueTreeState>& vts) {
        m_SpecParameter.m_DisplayMinFreq = vts->getRawParameterValue("MinFreq");
        m_SpecParameter.m_DisplayMinFreqOld = std::log(1.f);
        m_SpecParameter.m_DisplayMaxFreq = vts->getRawParameterValue("MaxFreq");
        m_SpecParameter.m_DisplayMaxFreqOld = std::log(20000.f);
        m_SpecParameter.m_DisplayMinColor = vts->getRawParameterValue("MinColor");
        m_SpecParameter.m_DisplayMinColorOld = -50.f;
        m_SpecParameter.m_DisplayMaxColor = vts->getRawParameterValue("MaxColor");
        m_SpecParameter.m_DisplayMaxColorOld = 50.f;
    }
    struct DisplayParameterRefs {   // the members of the reference's SpectrogramParameter (Spectrogram.h:66-75)
        std::atomic<float>* m_DisplayMinFreq = nullptr;
        float m_DisplayMinFreqOld = 0.f;
        std::atomic<float>* m_DisplayMaxFreq = nullptr;
        float m_DisplayMaxFreqOld = 0.f;
        std::atomic<float>* m_DisplayMinColor = nullptr;
        float m_DisplayMinColorOld = 0.f;
        std::atomic<float>* m_DisplayMaxColor = nullptr;
        float m_DisplayMaxColorOld = 0.f;
    };
    const DisplayParameterRefs& displayParameters() const { return m_SpecParameter; }

    // setter
    void setSamplerate(float samplerate) { check(jsg_set_samplerate(m_engine.get(), samplerate)); }
    void setchannels(size_t newchannels) {
        std::lock_guard<std::recursive_mutex> lk(configLock());
        // the pointer table grows on the message thread while the audio thread is kept out; the audio thread never allocates
        if (m_ptrs.size() < newchannels) whileAudioThreadIsOut([&] { m_ptrs.resize(newchannels); });
        check(jsg_set_channels(m_engine.get(), int(newchannels)));
    }
    void setFFTSize(size_t newFFTSize) {
        // engine first, re-blocker second (the reference does both under m_protect, Spectrogram.cpp:162-167).  In between the audio
        // thread may still deliver blocks of the old size: they carry their size and the engine drops them.
        std::lock_guard<std::recursive_mutex> lk(configLock());
        if (check(jsg_set_fft_size(m_engine.get(), int(newFFTSize)))) setDesiredBlockSizeSamples(newFFTSize);
    }
    void setclosestFFTSize_ms(float fftsize_ms) { setFFTSize(getnextpowerof2(fftsize_ms)); }
    void setmemoryTime_s(float memsize_s) { check(jsg_set_memory_time_s(m_engine.get(), memsize_s)); }
    void setfeed_percent(FeedPercentage feed) { check(jsg_set_feed_percent(m_engine.get(), int(feed))); }
    void setPauseMode(bool mode) { check(jsg_set_pause_mode(m_engine.get(), mode ? 1 : 0)); }
    void setWindow(Spectrogram::Windows win) { check(jsg_set_window(m_engine.get(), int(win))); }

    size_t getnextpowerof2(float fftsize_ms) {
        return size_t(jsg_next_power_of_2(fftsize_ms, jsg_get_samplerate(m_engine.get())));
    }

    int getSpectrumSize() { return jsg_get_spectrum_size(m_engine.get()); }
    int getMemorySize() { return jsg_get_memory_size(m_engine.get()); }
    int getMem(std::vector<std::vector<float>>& mem, int& pos) {
        const int W = getMemorySize(), H = getSpectrumSize();
        if (int(mem.size()) != W) return -1;   // Spectrogram.cpp:297-298
        // the engine writes the new columns straight into the caller's rows (rows of another length are skipped)
        m_rows.resize(size_t(W));
        for (int c = 0; c < W; ++c) m_rows[size_t(c)] = int(mem[size_t(c)].size()) == H ? mem[size_t(c)].data() : nullptr;
        return jsg_get_mem_rows(m_engine.get(), m_rows.data(), W, H, &pos);
    }
    float getSamplerate() { return jsg_get_samplerate(m_engine.get()); }

    // extensions (not in the reference)
    void setMixMode(ChannelMixMode m) { check(jsg_set_mix_mode(m_engine.get(), int(m))); }
    void setPowerScale(float s) { check(jsg_set_power_scale(m_engine.get(), s)); }
    jsg_engine* engine() { return m_engine.get(); }
    const std::string& lastError() const { return m_lastError; }   // text of the last failed setter ("" if none)

protected:
    void channelsPrepared(size_t channels) override {
        if (m_engine && int(channels) != jsg_get_channels(m_engine.get())) setchannels(channels);
    }

private:
    struct Deleter {
        void operator()(jsg_engine* e) const { jsg_destroy(e); }
    };
    bool check(int rc) {   // setters run inside JUCE callbacks: report, never throw
        if (rc >= 0) return true;
        m_lastError = jsg_last_error(m_engine.get());
        std::fprintf(stderr, "Spectrogram (libjsg): %s\n", m_lastError.c_str());
        return false;
    }
    std::unique_ptr<jsg_engine, Deleter> m_engine;
    std::atomic<bool> m_nonRealtime{false};
    std::atomic<int> m_nonRealtimeTimeoutMs{5000};
    std::vector<const float*> m_ptrs;
    std::vector<float*> m_rows;
    std::string m_lastError;
    DisplayParameterRefs m_SpecParameter;
};

// The colour half of SpectrogramComponent::timerCallback (reference Spectrogram.cpp:590-731) as one call:
//     gpuDisplay.update(minColorSlider, maxColorSlider, destData.data, destData.lineStride/4, newVals, pos);
// replaces getMem + the pixel loops; the image is [getSpectrumSize()][getMemorySize()] 32-bit ARGB
// (juce::Colour(uint32) order), low frequencies at the bottom, time running left to right.
class SpectrogramGpuDisplay {
public:
    explicit SpectrogramGpuDisplay(Spectrogram& s, int nrOfColors = 256, int scheme = CColorPalette::kJade) : m_spec(s) {
        setColorSceme(scheme, nrOfColors);   // m_colorpalette(256,6), reference Spectrogram.cpp:337
    }
    void setColorSceme(int scheme, int nrOfColors = 256) { jsg_display_set_colormap(m_spec.engine(), nrOfColors, scheme); }
    void setRunningDisplay(bool running) { jsg_display_set_running(m_spec.engine(), running ? 1 : 0); }
    void recomputeAll() { jsg_display_invalidate(m_spec.engine()); }
    int update(float minColor, float maxColor, uint32_t* argb, int64_t pitchPixels, int& newVals, int& pos) {
        return jsg_display_update(m_spec.engine(), minColor, maxColor, argb, pitchPixels, &newVals, &pos);
    }
    // Incremental variant for hosts that keep the reference's scroll (m_internalImg.moveImageSection, Spectrogram.cpp:665):
    // fills `tile` ([height][pitch], oldest column first) with the newVals newest columns; returns 1 when the whole
    // image has to be redrawn through update() instead (colour range / scheme changed, or more than maxCols are new).
    int updateTile(float minColor, float maxColor, uint32_t* tile, int64_t pitchPixels, int maxCols, int& newVals, int& pos) {
        return jsg_display_update_tile(m_spec.engine(), minColor, maxColor, tile, pitchPixels, maxCols, &newVals, &pos);
    }
    // the rows paint() blits for [minFreq, maxFreq] (reference Spectrogram.cpp:441-459)
    static void freqRows(float fs, int height, float minFreq, float maxFreq, int& startPixel, int& endPixel,
                         int& heightInterval, int& hStart) {
        jsg_display_freq_rows(fs, height, minFreq, maxFreq, &startPixel, &endPixel, &heightInterval, &hStart);
    }

private:
    Spectrogram& m_spec;
};
