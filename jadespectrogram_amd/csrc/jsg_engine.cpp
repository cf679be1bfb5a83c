// Engine: the state of the reference's class Spectrogram (Spectrogram.h:81-169) kept resident on one MI355X,
// plus the colour half of SpectrogramComponent::timerCallback (Spectrogram.cpp:590-731).
//
// Data layout in HBM (all owned by the engine):
//   d_in    [channels][in_pitch]   audio stream window: [ tail: the previous fft-size samples | new blocks ... ]
//                                  (the reference's 2N ring m_indatamem, Spectrogram.cpp:41-48,121-131, stretched to
//                                  hold a whole batch so that every HBM sample is read by the kernel exactly once
//                                  per overlapping frame and never copied again except the N-sample tail).
//   d_ring  [W][pitch]             dB columns, column-contiguous bins like m_mem[col][bin] (Spectrogram.h:144);
//                                  pitch = n/2+1 rounded up to 32 floats so every column starts on a 128-byte line.
//                                  Per-channel mode: [channels][W][pitch].
//   d_img   [H][img_pitch] ARGB    display image in ring order (x = ring column); the running-mode rotation is
//                                  applied while copying out, so nothing is ever scrolled on the device.  Rows are
//                                  padded to 32 pixels: with 128-byte-aligned rows the 64-pixel tiles of a full
//                                  recolour write whole cache lines (C5 image: 8.2 instead of 10.2 us).
//   d_lut   [n_colors] int32       CColorPalette table.
//   h_pin   (host, page-locked)    8 slots of [channels][n] floats: staging ring of processSynchronBlock.
//
// There is no CPU compute path here: without a usable HIP device every entry point reports an error.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "jsg_internal.h"

namespace jsg {

std::string& tls_error() {
    static thread_local std::string s;
    return s;
}
int jsg_fail(int code, const char* what) {
    tls_error() = what ? what : "";
    return code;
}
int jsg_fail_hip(hipError_t err, const char* where) {
    tls_error() = std::string(where ? where : "hip") + ": " + hipGetErrorString(err);
    return (err == hipErrorNoDevice || err == hipErrorInvalidDevice) ? JSG_ERR_NO_DEVICE : JSG_ERR_HIP;
}

}  // namespace jsg

using namespace jsg;

namespace {
const int kNewEntrySentinel = 1215752192;   // int(100000000000), reference Spectrogram.cpp:18,168,236
const float kRingFillDb = -120.0f;          // reference Spectrogram.cpp:223
const uint32_t kJuceRed = 0xFFFF0000u;      // juce::Colours::red (Spectrogram.cpp:654,717)

inline int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }
}  // namespace

struct jsg_engine {
    std::mutex mu;
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    // configuration (reference ctor defaults, Spectrogram.cpp:16-24)
    float fs = 48000.f;
    int channels = 2;
    float feed_percent = 100.f;
    int feedblocks = 1;
    float memsize_s = 1.f;
    int n = 1024;
    int mix = JSG_MIX_ABSMEAN;
    int window_choice = JSG_WIN_HANN;
    bool window_custom = false;
    bool pause = false;
    float power_scale = 1.f;
    std::vector<float> window;

    // derived geometry
    int hop = 1024, W = 0, H = 513;
    int64_t pitch = 0;

    // device state
    jsg_plan* plan = nullptr;
    float* d_ring = nullptr;
    size_t ring_floats = 0;
    float* d_in = nullptr;
    int64_t in_pitch = 0;
    int in_cap_blocks = 0;
    int mem_counter = 0;
    long long new_entry = kNewEntrySentinel;

    // pinned staging ring of processSynchronBlock (audio thread): the block is copied into page-locked memory and
    // leaves for the GPU with an asynchronous DMA, so the caller never waits for the device
    static const int kPinSlots = 8;
    float* h_pin = nullptr;
    size_t pin_slot_floats = 0;
    hipEvent_t pin_done[kPinSlots] = {};
    unsigned pin_next = 0;

    // display
    int n_colors = 256, scheme = JSG_CM_JADE;   // CColorPalette(256,6), Spectrogram.cpp:337
    std::vector<int32_t> lut;
    int32_t* d_lut = nullptr;
    uint32_t* d_img = nullptr;
    int img_w = 0, img_h = 0;
    int64_t img_pitch = 0;   // pixels per device image row (img_w rounded up to 32)
    bool recompute_all = true;
    bool running = true;

    int fail(int code, const std::string& what) {
        err = what;
        tls_error() = what;
        return code;
    }
    int fail_hip(hipError_t e, const char* where) {
        const int code = jsg_fail_hip(e, where);
        err = tls_error();
        return code;
    }
    int planes() const { return mix == JSG_MIX_PER_CHANNEL ? channels : 1; }
};

#define JSG_HIP(e, call)                                   \
    do {                                                   \
        hipError_t _err = (call);                          \
        if (_err != hipSuccess) return (e)->fail_hip(_err, #call); \
    } while (0)

namespace {

int rebuild_plan(jsg_engine* e) {
    if (e->plan) {
        jsg_plan_destroy(e->plan);
        e->plan = nullptr;
    }
    const int rc = jsg_plan_create(&e->plan, e->n, e->window.data(), e->power_scale);
    if (rc != JSG_OK) e->err = tls_error();
    return rc;
}

int build_window(jsg_engine* e) {   // Spectrogram::setWindowFkt
    if (e->window_custom && int(e->window.size()) == e->n) return rebuild_plan(e);
    e->window_custom = false;
    e->window.assign(size_t(e->n), 0.f);
    const int rc = jsg_window_build(e->window_choice, e->n, e->window.data());
    if (rc != JSG_OK) return e->fail(rc, "jsg_window_build failed");
    return rebuild_plan(e);
}

int ensure_input_capacity(jsg_engine* e, int blocks, bool keep_tail) {
    if (e->d_in && blocks <= e->in_cap_blocks) return JSG_OK;
    const int cap = std::max(blocks, std::max(1, e->in_cap_blocks));
    const int64_t new_pitch = round_up(int64_t(cap + 1) * e->n, 64);
    float* d_new = nullptr;
    JSG_HIP(e, hipMalloc(reinterpret_cast<void**>(&d_new), size_t(new_pitch) * e->channels * sizeof(float)));
    JSG_HIP(e, hipMemsetAsync(d_new, 0, size_t(new_pitch) * e->channels * sizeof(float), e->stream));
    if (e->d_in && keep_tail) {
        JSG_HIP(e, hipMemcpy2DAsync(d_new, size_t(new_pitch) * 4, e->d_in, size_t(e->in_pitch) * 4, size_t(e->n) * 4,
                                    size_t(e->channels), hipMemcpyDeviceToDevice, e->stream));
    }
    JSG_HIP(e, hipStreamSynchronize(e->stream));
    if (e->d_in) (void)hipFree(e->d_in);
    e->d_in = d_new;
    e->in_pitch = new_pitch;
    e->in_cap_blocks = cap;
    return JSG_OK;
}

// Spectrogram::buildmem (Spectrogram.cpp:213-238): geometry, -120 dB ring, zeroed input ring, counters.
int buildmem(jsg_engine* e) {
    JSG_HIP(e, hipSetDevice(e->device));
    e->hop = jsg_feed_samples(e->feed_percent, e->n);
    if (e->hop <= 0) return e->fail(JSG_ERR_INVALID, "feed percentage gives a hop of 0 samples");
    e->W = jsg_memsize_blocks(e->memsize_s, e->fs, e->hop);
    if (e->W <= 0) return e->fail(JSG_ERR_INVALID, "memory time gives an empty ring");
    e->H = e->n / 2 + 1;
    e->pitch = round_up(e->H, 32);
    const size_t need = size_t(e->W) * size_t(e->pitch) * size_t(e->planes());
    JSG_HIP(e, hipStreamSynchronize(e->stream));
    if (need != e->ring_floats) {
        if (e->d_ring) (void)hipFree(e->d_ring);
        e->d_ring = nullptr;
        e->ring_floats = 0;
        JSG_HIP(e, hipMalloc(reinterpret_cast<void**>(&e->d_ring), need * sizeof(float)));
        e->ring_floats = need;
    }
    uint32_t bits;
    std::memcpy(&bits, &kRingFillDb, 4);
    JSG_HIP(e, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(e->d_ring), int(bits), need, e->stream));
    // input ring: drop and re-create zeroed (channel count or fft size may have changed)
    if (e->d_in) (void)hipFree(e->d_in);
    e->d_in = nullptr;
    e->in_cap_blocks = 0;
    int rc = ensure_input_capacity(e, 1, false);
    if (rc != JSG_OK) return rc;
    const size_t slot = size_t(e->channels) * size_t(e->n);
    if (slot != e->pin_slot_floats) {
        if (e->h_pin) (void)hipHostFree(e->h_pin);
        e->h_pin = nullptr;
        e->pin_slot_floats = 0;
        JSG_HIP(e, hipHostMalloc(reinterpret_cast<void**>(&e->h_pin), slot * jsg_engine::kPinSlots * sizeof(float), hipHostMallocDefault));
        e->pin_slot_floats = slot;
    }
    for (int i = 0; i < jsg_engine::kPinSlots; ++i)
        if (!e->pin_done[i]) JSG_HIP(e, hipEventCreateWithFlags(&e->pin_done[i], hipEventDisableTiming));
    e->mem_counter = 0;
    e->new_entry = kNewEntrySentinel;
    return JSG_OK;
}

int upload_lut(jsg_engine* e) {
    e->lut.assign(size_t(e->n_colors), 0);
    int rc = jsg_colormap_build(e->n_colors, e->scheme, e->lut.data());
    if (rc != JSG_OK) return e->fail(rc, "jsg_colormap_build failed");
    JSG_HIP(e, hipStreamSynchronize(e->stream));
    if (e->d_lut) (void)hipFree(e->d_lut);
    e->d_lut = nullptr;
    JSG_HIP(e, hipMalloc(reinterpret_cast<void**>(&e->d_lut), size_t(e->n_colors) * 4));
    JSG_HIP(e, hipMemcpy(e->d_lut, e->lut.data(), size_t(e->n_colors) * 4, hipMemcpyHostToDevice));
    return JSG_OK;
}

// frames of `blocks` new blocks whose samples already sit behind the tail in d_in
int run_blocks(jsg_engine* e, int blocks) {
    const long long frames = (long long)blocks * e->feedblocks;
    if (!e->pause && frames > 0) {   // paused: the reference computes and drops the columns (Spectrogram.cpp:111)
        // only the newest W columns survive in the ring; never let one launch write a column twice
        const long long skip = frames > e->W ? frames - e->W : 0;
        jsg_stft_args a{};
        a.in = e->d_in;
        a.in_pitch = e->in_pitch;
        a.channels = e->channels;
        a.hop = e->hop;
        a.feedblocks = e->feedblocks;
        a.mix_mode = e->mix;
        a.first_frame = skip;
        a.n_frames = frames - skip;
        a.out_db = e->d_ring;
        a.out_pitch = e->pitch;
        a.out_channel_pitch = int64_t(e->W) * e->pitch;
        a.ring_width = e->W;
        a.ring_pos = int((e->mem_counter + skip) % e->W);
        const int rc = jsg_stft_db_launch(e->plan, &a, e->stream);
        if (rc != JSG_OK) {
            e->err = tls_error();
            return rc;
        }
        e->mem_counter = int((e->mem_counter + frames) % e->W);
        e->new_entry += frames;
        if (e->new_entry > 2000000000ll) e->new_entry = 2000000000ll;   // the reference's int would overflow
    }
    // keep the last block as the next call's history (reference Spectrogram.cpp:121-131)
    JSG_HIP(e, hipMemcpy2DAsync(e->d_in, size_t(e->in_pitch) * 4, e->d_in + size_t(blocks) * e->n, size_t(e->in_pitch) * 4,
                                size_t(e->n) * 4, size_t(e->channels), hipMemcpyDeviceToDevice, e->stream));
    return JSG_OK;
}

}  // namespace

extern "C" {

const char* jsg_last_error(const jsg_engine* e) { return e ? e->err.c_str() : tls_error().c_str(); }

int jsg_create(jsg_engine** out, int channels) {
    if (!out || channels <= 0) return jsg_fail(JSG_ERR_INVALID, "jsg_create: bad argument");
    *out = nullptr;
    int ndev = 0;
    hipError_t herr = hipGetDeviceCount(&ndev);
    if (herr != hipSuccess || ndev <= 0)
        return jsg_fail(JSG_ERR_NO_DEVICE, "jsg_create: no HIP device available (this engine has no CPU fallback)");
    jsg_engine* e = new (std::nothrow) jsg_engine();
    if (!e) return jsg_fail(JSG_ERR_NOMEM, "jsg_create: out of memory");
    e->channels = channels;
    herr = hipGetDevice(&e->device);
    if (herr == hipSuccess) herr = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
    if (herr != hipSuccess) {
        const int code = jsg_fail_hip(herr, "jsg_create");
        delete e;
        return code;
    }
    int rc = buildmem(e);
    if (rc == JSG_OK) rc = build_window(e);   // (the reference leaves m_window empty until setFFTSize/setWindow)
    if (rc == JSG_OK) rc = upload_lut(e);
    if (rc != JSG_OK) {
        tls_error() = e->err;
        jsg_destroy(e);
        return rc;
    }
    *out = e;
    return JSG_OK;
}

int jsg_destroy(jsg_engine* e) {
    if (!e) return JSG_OK;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->plan) jsg_plan_destroy(e->plan);
    if (e->d_ring) (void)hipFree(e->d_ring);
    if (e->d_in) (void)hipFree(e->d_in);
    if (e->d_lut) (void)hipFree(e->d_lut);
    if (e->d_img) (void)hipFree(e->d_img);
    if (e->h_pin) (void)hipHostFree(e->h_pin);
    for (int i = 0; i < jsg_engine::kPinSlots; ++i)
        if (e->pin_done[i]) (void)hipEventDestroy(e->pin_done[i]);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
    return JSG_OK;
}

#define JSG_LOCK(e)                                                          \
    if (!(e)) return jsg_fail(JSG_ERR_INVALID, "null engine");               \
    std::lock_guard<std::mutex> _lk((e)->mu);                                \
    do {                                                                     \
        hipError_t _e = hipSetDevice((e)->device);                           \
        if (_e != hipSuccess) return (e)->fail_hip(_e, "hipSetDevice");      \
    } while (0)

int jsg_set_samplerate(jsg_engine* e, float fs) {
    JSG_LOCK(e);
    if (!(fs > 0.f)) return e->fail(JSG_ERR_INVALID, "sample rate must be positive");
    e->fs = fs;
    return buildmem(e);
}

int jsg_set_channels(jsg_engine* e, int channels) {
    JSG_LOCK(e);
    if (channels <= 0) return e->fail(JSG_ERR_INVALID, "channel count must be positive");
    e->channels = channels;
    return buildmem(e);
}

int jsg_set_fft_size(jsg_engine* e, int n) {
    JSG_LOCK(e);
    if (n != 512 && n != 1024 && n != 2048 && n != 4096 && n != 8192)
        return e->fail(JSG_ERR_UNSUPPORTED, "FFT size must be 512, 1024, 2048, 4096 or 8192");
    e->n = n;
    e->window_custom = false;
    int rc = buildmem(e);
    if (rc == JSG_OK) rc = build_window(e);
    e->new_entry = kNewEntrySentinel;
    return rc;
}

int jsg_set_closest_fft_size_ms(jsg_engine* e, float ms) {
    if (!e) return jsg_fail(JSG_ERR_INVALID, "null engine");
    const int n = jsg_next_power_of_2(ms, e->fs);
    return jsg_set_fft_size(e, n);
}

int jsg_set_memory_time_s(jsg_engine* e, float seconds) {
    JSG_LOCK(e);
    if (!(seconds > 0.f)) return e->fail(JSG_ERR_INVALID, "memory time must be positive");
    e->memsize_s = seconds;
    return buildmem(e);
}

int jsg_set_feed_percent(jsg_engine* e, int feed) {
    JSG_LOCK(e);
    switch (feed) {   // Spectrogram.cpp:191-209
        case JSG_FEED_100: e->feed_percent = 100.f; e->feedblocks = 1; break;
        case JSG_FEED_50: e->feed_percent = 50.f; e->feedblocks = 2; break;
        case JSG_FEED_25: e->feed_percent = 25.f; e->feedblocks = 4; break;
        case JSG_FEED_10: e->feed_percent = 10.f; e->feedblocks = 10; break;
        default: return e->fail(JSG_ERR_INVALID, "unknown feed percentage");
    }
    return buildmem(e);
}

int jsg_set_feed_percent_ext(jsg_engine* e, float percent) {
    JSG_LOCK(e);
    const int hop = jsg_feed_samples(percent, e->n);
    if (hop <= 0 || hop > e->n) return e->fail(JSG_ERR_INVALID, "feed percentage out of range");
    e->feed_percent = percent;
    e->feedblocks = e->n / hop;
    return buildmem(e);
}

int jsg_set_pause_mode(jsg_engine* e, int paused) {
    JSG_LOCK(e);
    e->pause = paused != 0;
    return JSG_OK;
}

int jsg_set_window(jsg_engine* e, int window) {
    JSG_LOCK(e);
    if (window < JSG_WIN_RECT || window > JSG_WIN_HANNPOISSON) return e->fail(JSG_ERR_INVALID, "unknown window");
    e->window_choice = window;
    e->window_custom = false;
    JSG_HIP(e, hipStreamSynchronize(e->stream));
    return build_window(e);
}

int jsg_set_window_table(jsg_engine* e, const float* w, int n) {
    JSG_LOCK(e);
    if (!w || n != e->n) return e->fail(JSG_ERR_SIZE_MISMATCH, "window table must have fft-size entries");
    e->window.assign(w, w + n);
    e->window_custom = true;
    JSG_HIP(e, hipStreamSynchronize(e->stream));
    return rebuild_plan(e);
}

int jsg_set_mix_mode(jsg_engine* e, int mode) {
    JSG_LOCK(e);
    const bool known = (mode >= JSG_MIX_ABSMEAN && mode <= JSG_MIX_RIGHT) || mode == JSG_MIX_PER_CHANNEL;
    if (!known) return e->fail(JSG_ERR_INVALID, "unknown mix mode");
    if (mode == JSG_MIX_RIGHT && e->channels < 2) return e->fail(JSG_ERR_INVALID, "JSG_MIX_RIGHT needs two channels");
    const bool replane = (mode == JSG_MIX_PER_CHANNEL) != (e->mix == JSG_MIX_PER_CHANNEL);
    e->mix = mode;
    return replane ? buildmem(e) : JSG_OK;
}

int jsg_set_power_scale(jsg_engine* e, float scale) {
    JSG_LOCK(e);
    if (!(scale > 0.f)) return e->fail(JSG_ERR_INVALID, "power scale must be positive");
    e->power_scale = scale;
    JSG_HIP(e, hipStreamSynchronize(e->stream));
    return rebuild_plan(e);
}

int jsg_get_spectrum_size(const jsg_engine* e) { return e ? e->H : JSG_ERR_INVALID; }
int jsg_get_memory_size(const jsg_engine* e) { return e ? e->W : JSG_ERR_INVALID; }
float jsg_get_samplerate(const jsg_engine* e) { return e ? e->fs : 0.f; }
int jsg_get_fft_size(const jsg_engine* e) { return e ? e->n : JSG_ERR_INVALID; }
int jsg_get_feed_samples(const jsg_engine* e) { return e ? e->hop : JSG_ERR_INVALID; }
int jsg_get_feedblocks(const jsg_engine* e) { return e ? e->feedblocks : JSG_ERR_INVALID; }
int jsg_get_channels(const jsg_engine* e) { return e ? e->channels : JSG_ERR_INVALID; }

int jsg_get_window(const jsg_engine* e, float* out, int n) {
    if (!e || !out) return jsg_fail(JSG_ERR_INVALID, "null argument");
    if (n != int(e->window.size())) return jsg_fail(JSG_ERR_SIZE_MISMATCH, "window size mismatch");
    std::memcpy(out, e->window.data(), size_t(n) * sizeof(float));
    return JSG_OK;
}

int jsg_process_block(jsg_engine* e, const float* const* planar) {
    JSG_LOCK(e);
    if (!planar) return e->fail(JSG_ERR_INVALID, "null block");
    int rc = ensure_input_capacity(e, 1, true);
    if (rc != JSG_OK) return rc;
    // copy-in (reference Spectrogram.cpp:41-48) through the pinned ring: host memcpy now, DMA later
    const unsigned slot = e->pin_next++ % jsg_engine::kPinSlots;
    JSG_HIP(e, hipEventSynchronize(e->pin_done[slot]));   // only waits if the GPU is 8 blocks behind
    float* stage = e->h_pin + size_t(slot) * e->pin_slot_floats;
    for (int c = 0; c < e->channels; ++c) {
        if (!planar[c]) return e->fail(JSG_ERR_INVALID, "null channel pointer");
        std::memcpy(stage + size_t(c) * e->n, planar[c], size_t(e->n) * sizeof(float));
    }
    JSG_HIP(e, hipMemcpy2DAsync(e->d_in + e->n, size_t(e->in_pitch) * 4, stage, size_t(e->n) * 4, size_t(e->n) * 4,
                                size_t(e->channels), hipMemcpyHostToDevice, e->stream));
    JSG_HIP(e, hipEventRecord(e->pin_done[slot], e->stream));
    return run_blocks(e, 1);
}

int jsg_process_blocks(jsg_engine* e, const float* samples, int64_t pitch, int n_blocks) {
    JSG_LOCK(e);
    if (!samples || n_blocks < 0 || pitch < int64_t(n_blocks) * e->n) return e->fail(JSG_ERR_INVALID, "bad batch");
    if (n_blocks == 0) return JSG_OK;
    int rc = ensure_input_capacity(e, n_blocks, true);
    if (rc != JSG_OK) return rc;
    JSG_HIP(e, hipMemcpy2DAsync(e->d_in + e->n, size_t(e->in_pitch) * 4, samples, size_t(pitch) * 4,
                                size_t(n_blocks) * e->n * 4, size_t(e->channels), hipMemcpyHostToDevice, e->stream));
    return run_blocks(e, n_blocks);
}

int jsg_process_blocks_device(jsg_engine* e, const float* d_samples, int64_t pitch, int n_blocks) {
    JSG_LOCK(e);
    if (!d_samples || n_blocks < 0 || pitch < int64_t(n_blocks) * e->n) return e->fail(JSG_ERR_INVALID, "bad batch");
    if (n_blocks == 0) return JSG_OK;
    int rc = ensure_input_capacity(e, n_blocks, true);
    if (rc != JSG_OK) return rc;
    JSG_HIP(e, hipMemcpy2DAsync(e->d_in + e->n, size_t(e->in_pitch) * 4, d_samples, size_t(pitch) * 4,
                                size_t(n_blocks) * e->n * 4, size_t(e->channels), hipMemcpyDeviceToDevice, e->stream));
    return run_blocks(e, n_blocks);
}

int jsg_get_mem(jsg_engine* e, float* dst, int dst_columns, int* pos) {
    JSG_LOCK(e);
    const int planes = e->planes();
    if (!dst || dst_columns != e->W * planes) return e->fail(JSG_ERR_SIZE_MISMATCH, "getMem: buffer size mismatch");
    const int W = e->W, H = e->H;
    auto copy_cols = [&](int first, int count) -> hipError_t {
        if (count <= 0) return hipSuccess;
        for (int p = 0; p < planes; ++p) {
            hipError_t r = hipMemcpy2DAsync(dst + (size_t(p) * W + first) * H, size_t(H) * 4,
                                            e->d_ring + (size_t(p) * W + first) * e->pitch, size_t(e->pitch) * 4,
                                            size_t(H) * 4, size_t(count), hipMemcpyDeviceToHost, e->stream);
            if (r != hipSuccess) return r;
        }
        return hipSuccess;
    };
    const long long nec = e->new_entry;
    if (nec >= W) {                                   // Spectrogram.cpp:300-304
        JSG_HIP(e, copy_cols(0, W));
    } else {
        const int start = e->mem_counter - int(nec);  // :307
        if (start >= 0) {
            JSG_HIP(e, copy_cols(start, int(nec)));   // :310-311
        } else {
            JSG_HIP(e, copy_cols(0, e->mem_counter));          // :315-316
            JSG_HIP(e, copy_cols(W + start, -start));          // :318-319
        }
    }
    JSG_HIP(e, hipStreamSynchronize(e->stream));
    e->new_entry = 0;                                 // :328
    if (pos) *pos = e->mem_counter;                   // :329
    return int(nec);
}

int jsg_ring_device(jsg_engine* e, float** d_ring, int64_t* pitch, int* width, int* pos) {
    JSG_LOCK(e);
    if (d_ring) *d_ring = e->d_ring;
    if (pitch) *pitch = e->pitch;
    if (width) *width = e->W;
    if (pos) *pos = e->mem_counter;
    return JSG_OK;
}

int jsg_sync(jsg_engine* e) {
    JSG_LOCK(e);
    JSG_HIP(e, hipStreamSynchronize(e->stream));
    return JSG_OK;
}

void* jsg_stream(jsg_engine* e) { return e ? reinterpret_cast<void*>(e->stream) : nullptr; }

int jsg_display_set_colormap(jsg_engine* e, int n_colors, int scheme) {
    JSG_LOCK(e);
    if (n_colors <= 0 || n_colors > 65535 || scheme < JSG_CM_MONO || scheme > JSG_CM_JADE)
        return e->fail(JSG_ERR_INVALID, "bad colour map");
    e->n_colors = n_colors;
    e->scheme = scheme;
    e->recompute_all = true;   // Spectrogram.cpp:400
    return upload_lut(e);
}

int jsg_display_set_running(jsg_engine* e, int running) {
    JSG_LOCK(e);
    e->running = running != 0;
    return JSG_OK;
}

int jsg_display_invalidate(jsg_engine* e) {
    JSG_LOCK(e);
    e->recompute_all = true;
    return JSG_OK;
}

int jsg_display_update(jsg_engine* e, float min_color, float max_color, uint32_t* argb, int64_t pitch, int* new_vals,
                       int* pos_out) {
    JSG_LOCK(e);
    if (e->mix == JSG_MIX_PER_CHANNEL) return e->fail(JSG_ERR_UNSUPPORTED, "display needs a mixed (single) spectrogram");
    const int W = e->W, H = e->H;
    if (!argb || pitch < W) return e->fail(JSG_ERR_INVALID, "bad image buffer");
    if (W != e->img_w || H != e->img_h) {             // Spectrogram.cpp:595-605
        JSG_HIP(e, hipStreamSynchronize(e->stream));
        if (e->d_img) (void)hipFree(e->d_img);
        e->d_img = nullptr;
        e->img_pitch = round_up(W, 32);
        JSG_HIP(e, hipMalloc(reinterpret_cast<void**>(&e->d_img), size_t(e->img_pitch) * H * 4));
        e->img_w = W;
        e->img_h = H;
        e->recompute_all = true;
    }
    const long long nec = e->new_entry;               // getMem, Spectrogram.cpp:607-608
    e->new_entry = 0;
    const int pos = e->mem_counter;
    if (nec > W) e->recompute_all = true;             // :610-613
    jsg_colormap_args a{};
    a.db = e->d_ring;
    a.db_pitch = e->pitch;
    a.ring_width = W;
    a.height = H;
    a.x_wrap = W;
    a.lut = e->d_lut;
    a.n_colors = e->n_colors;
    jsg_colormap_range(e->n_colors, min_color, max_color, &a.vmin, &a.vmax, &a.access_mult);   // :617
    a.argb_out = e->d_img;
    a.argb_pitch = e->img_pitch;
    const bool all = e->recompute_all;
    if (all) {                                        // :623-657
        e->recompute_all = false;
        a.col_first = 0;
        a.n_cols = W;
    } else {                                          // :658-724 (only the new columns)
        a.n_cols = int(std::min<long long>(nec, W));
        a.col_first = ((pos - a.n_cols) % W + W) % W;
    }
    a.x_first = a.col_first;
    int rc = jsg_colormap_launch(&a, e->stream);
    if (rc != JSG_OK) {
        e->err = tls_error();
        return rc;
    }
    if (e->running) {
        // x = (col + W - pos) mod W  (Spectrogram.cpp:626-631): columns [pos,W) first, then [0,pos)
        JSG_HIP(e, hipMemcpy2DAsync(argb, size_t(pitch) * 4, e->d_img + pos, size_t(e->img_pitch) * 4, size_t(W - pos) * 4, size_t(H),
                                    hipMemcpyDeviceToHost, e->stream));
        if (pos > 0)
            JSG_HIP(e, hipMemcpy2DAsync(argb + (W - pos), size_t(pitch) * 4, e->d_img, size_t(e->img_pitch) * 4, size_t(pos) * 4,
                                        size_t(H), hipMemcpyDeviceToHost, e->stream));
        JSG_HIP(e, hipStreamSynchronize(e->stream));
    } else {
        JSG_HIP(e, hipMemcpy2DAsync(argb, size_t(pitch) * 4, e->d_img, size_t(e->img_pitch) * 4, size_t(W) * 4, size_t(H),
                                    hipMemcpyDeviceToHost, e->stream));
        JSG_HIP(e, hipStreamSynchronize(e->stream));
        // red cursor (Spectrogram.cpp:650-656 one column after a full recolour, :703-720 otherwise)
        int drawwidth = 1;
        if (!all) {
            if (H < 2048) drawwidth++;
            if (H < 1024) drawwidth += 2;
        }
        for (int dd = 0; dd < drawwidth; ++dd) {
            int drawpos = pos + dd;
            if (drawpos == W) drawpos -= W;
            if (drawpos >= W) continue;   // the reference would write outside the image here
            for (int y = 0; y < H; ++y) argb[size_t(y) * pitch + drawpos] = kJuceRed;
        }
    }
    if (new_vals) *new_vals = int(nec);
    if (pos_out) *pos_out = pos;
    return JSG_OK;
}

int jsg_display_update_tile(jsg_engine* e, float min_color, float max_color, uint32_t* tile, int64_t tile_pitch,
                            int max_cols, int* new_vals, int* pos_out) {
    JSG_LOCK(e);
    if (e->mix == JSG_MIX_PER_CHANNEL) return e->fail(JSG_ERR_UNSUPPORTED, "display needs a mixed (single) spectrogram");
    const int W = e->W, H = e->H;
    if (!tile || max_cols <= 0 || tile_pitch < max_cols) return e->fail(JSG_ERR_INVALID, "bad tile buffer");
    const long long nec = e->new_entry;
    if (e->recompute_all || W != e->img_w || H != e->img_h || nec > W || nec > max_cols) {
        if (new_vals) *new_vals = int(std::min<long long>(nec, 2000000000ll));
        if (pos_out) *pos_out = e->mem_counter;
        return 1;   // the caller needs the whole image: jsg_display_update
    }
    e->new_entry = 0;
    const int pos = e->mem_counter;
    const int nv = int(nec);
    if (nv > 0) {
        jsg_colormap_args a{};
        a.db = e->d_ring;
        a.db_pitch = e->pitch;
        a.ring_width = W;
        a.height = H;
        a.x_wrap = W;
        a.lut = e->d_lut;
        a.n_colors = e->n_colors;
        jsg_colormap_range(e->n_colors, min_color, max_color, &a.vmin, &a.vmax, &a.access_mult);
        a.argb_out = e->d_img;
        a.argb_pitch = e->img_pitch;
        a.n_cols = nv;
        a.col_first = ((pos - nv) % W + W) % W;
        a.x_first = a.col_first;
        const int rc = jsg_colormap_launch(&a, e->stream);
        if (rc != JSG_OK) {
            e->err = tls_error();
            return rc;
        }
        const int first = a.col_first;
        const int n1 = std::min(nv, W - first);   // columns before the ring wraps
        JSG_HIP(e, hipMemcpy2DAsync(tile, size_t(tile_pitch) * 4, e->d_img + first, size_t(e->img_pitch) * 4, size_t(n1) * 4, size_t(H),
                                    hipMemcpyDeviceToHost, e->stream));
        if (nv > n1)
            JSG_HIP(e, hipMemcpy2DAsync(tile + n1, size_t(tile_pitch) * 4, e->d_img, size_t(e->img_pitch) * 4, size_t(nv - n1) * 4,
                                        size_t(H), hipMemcpyDeviceToHost, e->stream));
        JSG_HIP(e, hipStreamSynchronize(e->stream));
    }
    if (new_vals) *new_vals = nv;
    if (pos_out) *pos_out = pos;
    return JSG_OK;
}

}  // extern "C"
