// Stand-in for the author's external TGM-library base class `SynchronBlockProcessor` (the reference includes it at
// Spectrogram.h:15 but does not ship it).  Only the surface the plugin uses is provided:
//   preparetoProcess(channels, maxBlockSize)      reference PluginProcessor.cpp:108
//   setDesiredBlockSizeSamples(n)                 reference Spectrogram.cpp:164,180
//   processBlock(AudioBuffer<float>&, MidiBuffer&) reference PluginProcessor.cpp:148
//   virtual processSynchronBlock(vector<vector<float>>&, MidiBuffer&)   reference Spectrogram.h:112
// It turns host blocks of arbitrary length into fixed blocks of the desired size (the FFT size) and hands each to
// processSynchronBlock.  The audio itself passes through untouched (the spectrogram only analyses).
//
// Threads: processBlock runs on the audio thread, setDesiredBlockSizeSamples / preparetoProcess on the message thread
// (the FFT-size combo box, reference Spectrogram.cpp:760-767).  The FIFO is guarded by one lock, like the reference's
// m_protect around processSynchronBlock and setFFTSize (Spectrogram.cpp:40,132,162,167) -- but the audio thread only
// TRIES it: while a resize is in progress the host block is skipped instead of making the audio thread wait
// (a resize wipes the spectrogram's history anyway, reference buildmem, Spectrogram.cpp:213-238).
#pragma once
#include <algorithm>
#include <atomic>
#include <cstddef>
#include <mutex>
#include <vector>

#include "juce_shim.h"

class SynchronBlockProcessor {
public:
    SynchronBlockProcessor() = default;
    virtual ~SynchronBlockProcessor() = default;

    void preparetoProcess(int channels, int maxBlockSize) {
        juce::ignoreUnused(maxBlockSize);
        std::lock_guard<std::recursive_mutex> lk(m_syncLock);
        m_syncChannels = channels > 0 ? size_t(channels) : 1;
        resetFifo();
        channelsPrepared(m_syncChannels);
    }
    void setDesiredBlockSizeSamples(size_t n) {
        std::lock_guard<std::recursive_mutex> lk(m_syncLock);
        m_syncBlock.store(n > 0 ? n : 1, std::memory_order_release);
        resetFifo();
    }
    // lock-free: a getter on the message thread must never make the audio thread's try_lock fail (that would drop a host block)
    size_t getDesiredBlockSizeSamples() const { return m_syncBlock.load(std::memory_order_acquire); }

    void processBlock(juce::AudioBuffer<float>& buffer, juce::MidiBuffer& midi) {
        std::unique_lock<std::recursive_mutex> lk(m_syncLock, std::try_to_lock);
        if (!lk.owns_lock()) return;   // a resize is running on the message thread: skip this host block
        const size_t ch = std::min(m_syncChannels, size_t(buffer.getNumChannels()));
        const size_t n = size_t(buffer.getNumSamples());
        const size_t block = m_syncBlock.load(std::memory_order_relaxed);   // stable: writers hold m_syncLock
        size_t done = 0;
        while (done < n) {
            const size_t take = std::min(n - done, block - m_fill);
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
            if (m_fill == block) {
                processSynchronBlock(m_fifo, midi);
                m_fill = 0;
            }
        }
    }

    virtual int processSynchronBlock(std::vector<std::vector<float>>& data, juce::MidiBuffer& midi) = 0;

protected:
    virtual void channelsPrepared(size_t /*channels*/) {}
    // for derived classes that change the block size together with their own state (Spectrogram::setFFTSize)
    std::recursive_mutex& syncLock() { return m_syncLock; }

private:
    void resetFifo() {
        m_fifo.assign(m_syncChannels, std::vector<float>(m_syncBlock.load(std::memory_order_relaxed), 0.f));
        m_fill = 0;
    }
    mutable std::recursive_mutex m_syncLock;
    size_t m_syncChannels = 2;
    std::atomic<size_t> m_syncBlock{1024};   // written under m_syncLock only; read lock-free by the getter
    size_t m_fill = 0;
    std::vector<std::vector<float>> m_fifo;
};
