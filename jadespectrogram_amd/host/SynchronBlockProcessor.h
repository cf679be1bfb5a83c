// Stand-in for the author's external TGM-library base class `SynchronBlockProcessor` (the reference includes it at
// Spectrogram.h:15 but does not ship it).  Only the surface the plugin uses is provided:
//   preparetoProcess(channels, maxBlockSize)      reference PluginProcessor.cpp:108
//   setDesiredBlockSizeSamples(n)                 reference Spectrogram.cpp:164,180
//   processBlock(AudioBuffer<float>&, MidiBuffer&) reference PluginProcessor.cpp:148
//   virtual processSynchronBlock(vector<vector<float>>&, MidiBuffer&)   reference Spectrogram.h:112
// It turns host blocks of arbitrary length into fixed blocks of the desired size (the FFT size) and hands each to
// processSynchronBlock.  The audio itself passes through untouched (the spectrogram only analyses).
#pragma once
#include <algorithm>
#include <cstddef>
#include <vector>

#include "juce_shim.h"

class SynchronBlockProcessor {
public:
    SynchronBlockProcessor() = default;
    virtual ~SynchronBlockProcessor() = default;

    void preparetoProcess(int channels, int maxBlockSize) {
        juce::ignoreUnused(maxBlockSize);
        m_syncChannels = channels > 0 ? size_t(channels) : 1;
        resetFifo();
        channelsPrepared(m_syncChannels);
    }
    void setDesiredBlockSizeSamples(size_t n) {
        m_syncBlock = n;
        resetFifo();
    }
    size_t getDesiredBlockSizeSamples() const { return m_syncBlock; }

    void processBlock(juce::AudioBuffer<float>& buffer, juce::MidiBuffer& midi) {
        const size_t ch = std::min(m_syncChannels, size_t(buffer.getNumChannels()));
        const size_t n = size_t(buffer.getNumSamples());
        size_t done = 0;
        while (done < n) {
            const size_t take = std::min(n - done, m_syncBlock - m_fill);
            for (size_t c = 0; c < m_syncChannels; ++c) {
                if (c < ch) {
                    const float* src = buffer.getReadPointer(int(c)) + done;
                    std::copy(src, src + take, m_fifo[c].begin() + long(m_fill));
                } else {
                    std::fill(m_fifo[c].begin() + long(m_fill), m_fifo[c].begin() + long(m_fill + take), 0.f);
                }
            }
            m_fill += take;
            done += take;
            if (m_fill == m_syncBlock) {
                processSynchronBlock(m_fifo, midi);
                m_fill = 0;
            }
        }
    }

    virtual int processSynchronBlock(std::vector<std::vector<float>>& data, juce::MidiBuffer& midi) = 0;

protected:
    virtual void channelsPrepared(size_t /*channels*/) {}

private:
    void resetFifo() {
        m_fifo.assign(m_syncChannels, std::vector<float>(m_syncBlock, 0.f));
        m_fill = 0;
    }
    size_t m_syncChannels = 2;
    size_t m_syncBlock = 1024;
    size_t m_fill = 0;
    std::vector<std::vector<float>> m_fifo;
};
