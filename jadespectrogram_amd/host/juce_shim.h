// Minimal stand-ins for the few JUCE types the Spectrogram call shape touches, used ONLY when JUCE is not
// available (this build image has no JUCE).  With a real JUCE tree on the include path the real headers win.
#pragma once
#if __has_include(<juce_audio_processors/juce_audio_processors.h>)
#include <juce_audio_processors/juce_audio_processors.h>
#define JSG_HAVE_JUCE 1
#else
#define JSG_HAVE_JUCE 0
#include <atomic>
#include <cstddef>
#include <map>
#include <memory>
#include <string>
#include <vector>
namespace juce {
class MidiBuffer {};
template <typename... Ts>
inline void ignoreUnused(Ts&&...) {}
// planar float buffer with the accessor subset used by processBlock()
template <typename T>
class AudioBuffer {
public:
    AudioBuffer() = default;
    AudioBuffer(int channels, int samples) { setSize(channels, samples); }
    void setSize(int channels, int samples) {
        m_channels = channels;
        m_samples = samples;
        m_data.assign(size_t(channels) * size_t(samples), T(0));
    }
    int getNumChannels() const { return m_channels; }
    int getNumSamples() const { return m_samples; }
    const T* getReadPointer(int ch) const { return m_data.data() + size_t(ch) * size_t(m_samples); }
    T* getWritePointer(int ch) { return m_data.data() + size_t(ch) * size_t(m_samples); }
private:
    int m_channels = 0, m_samples = 0;
    std::vector<T> m_data;
};
// Holder of raw parameter values with the one accessor Spectrogram::prepareParameter uses (reference
// Spectrogram.cpp:25-35: vts->getRawParameterValue(ID)).  The real class owns a parameter tree; this one owns a map
// id -> value so that the call at PluginProcessor.cpp:28 compiles and can be exercised without JUCE.
class AudioProcessorValueTreeState {
public:
    void addRawParameter(const std::string& id, float value) {
        auto& p = m_values[id];
        if (!p) p = std::make_unique<std::atomic<float>>(value);
        else p->store(value);
    }
    std::atomic<float>* getRawParameterValue(const std::string& id) const {
        auto it = m_values.find(id);
        return it == m_values.end() ? nullptr : it->second.get();
    }
private:
    std::map<std::string, std::unique_ptr<std::atomic<float>>> m_values;
};
}  // namespace juce
#endif
