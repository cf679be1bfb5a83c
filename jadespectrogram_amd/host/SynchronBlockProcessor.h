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
// (the FFT-size combo box, reference Spectrogram.cpp:760-767).  The audio thread takes NO lock (round 4): the FIFO's geometry is
// guarded by an epoch -- m_gen is odd while a resize is in progress, the resizing thread waits until the one processBlock call that
// may be inside has left (m_inflight) before it touches the FIFO -- so a host block is skipped only while a resize is really
// happening (it wipes the spectrogram's history anyway, reference buildmem, Spectrogram.cpp:213-238), never because some other
// thread merely held a lock.  Skipped host blocks are counted (droppedHostBlocks()).  Message-thread callers serialise among
// themselves with m_cfgLock, which the audio thread never touches.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstddef>
#include <mutex>
#include <thread>
#include <vector>

#include "juce_shim.h"

class SynchronBlockProcessor {
public:
    SynchronBlockProcessor() = default;
    virtual ~SynchronBlockProcessor() = default;

    void preparetoProcess(int channels, int maxBlockSize) {
        juce::ignoreUnused(maxBlockSize);
        std::lock_guard<std::recursive_mutex> lk(m_cfgLock);
        Resize r(*this);
        m_syncChannels = channels > 0 ? size_t(channels) : 1;
        resetFifo();
        channelsPrepared(m_syncChannels);
    }
    void setDesiredBlockSizeSamples(size_t n) {
        std::lock_guard<std::recursive_mutex> lk(m_cfgLock);
        Resize r(*this);
        m_syncBlock.store(n > 0 ? n : 1, std::memory_order_release);
        resetFifo();
    }
    size_t getDesiredBlockSizeSamples() const { return m_syncBlock.load(std::memory_order_acquire); }
    unsigned long long droppedHostBlocks() const { return m_dropped.load(std::memory_order_relaxed); }

    void processBlock(juce::AudioBuffer<float>& buffer, juce::MidiBuffer& midi) {
        m_inflight.fetch_add(1);                       // (sequentially consistent with m_gen: hand-shake with Resize)
        if (m_gen.load() & 1u) {                       // a resize is running on the message thread: skip this host block
            m_dropped.fetch_add(1, std::memory_order_relaxed);
            m_inflight.fetch_sub(1);
            return;
        }
        const size_t ch = std::min(m_syncChannels, size_t(buffer.getNumChannels()));
        const size_t n = size_t(buffer.getNumSamples());
        const size_t block = m_syncBlock.load(std::memory_order_relaxed);   // stable: written inside a Resize only
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
        m_inflight.fetch_sub(1);
    }

    virtual int processSynchronBlock(std::vector<std::vector<float>>& data, juce::MidiBuffer& midi) = 0;

protected:
    virtual void channelsPrepared(size_t /*channels*/) {}
    // for derived classes whose setters change their own state together with the block size (Spectrogram::setFFTSize): serialises
    // the message-thread callers; the audio thread never takes it
    std::recursive_mutex& configLock() { return m_cfgLock; }
    // runs f on the calling (message) thread while the audio thread is guaranteed to be outside processBlock and skipping
    template <class F>
    void whileAudioThreadIsOut(F f) {
        std::lock_guard<std::recursive_mutex> lk(m_cfgLock);
        Resize r(*this);
        f();
    }

private:
    struct Resize {   // the FIFO may be touched between construction and destruction (re-entrant on the message thread)
        SynchronBlockProcessor& p;
        bool outer;
        explicit Resize(SynchronBlockProcessor& sp) : p(sp), outer(sp.m_resizeDepth++ == 0) {
            if (!outer) return;
            p.m_gen.fetch_add(1);                                            // odd: the audio thread skips from here on
            while (p.m_inflight.load() != 0) std::this_thread::yield();      // the one call that may be inside leaves
        }
        ~Resize() {
            --p.m_resizeDepth;
            if (outer) p.m_gen.fetch_add(1);                                 // even: the new geometry is visible
        }
    };
    void resetFifo() {
        m_fifo.assign(m_syncChannels, std::vector<float>(m_syncBlock.load(std::memory_order_relaxed), 0.f));
        m_fill = 0;
    }
    std::recursive_mutex m_cfgLock;          // message-thread callers only
    int m_resizeDepth = 0;                   // (under m_cfgLock)
    std::atomic<unsigned> m_gen{0};
    std::atomic<int> m_inflight{0};
    std::atomic<unsigned long long> m_dropped{0};
    size_t m_syncChannels = 2;
    std::atomic<size_t> m_syncBlock{1024};   // written inside a Resize only
    size_t m_fill = 0;
    std::vector<std::vector<float>> m_fifo;
};
