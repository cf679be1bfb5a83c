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
//   h_q     (host, page-locked)    64 slots of [channels][n] floats: the audio thread's block queue (see "Producer" below).
//   h_mem   (host, page-locked)    [planes][W][H] floats: landing area of getMem's device-to-host copies.
//
// Threads (reference: audio thread = producer, message thread = consumer and setters):
//   mu      state lock.  Held only while counters are read/updated and work is ENQUEUED -- never across a stream
//           synchronisation or a blocking copy, except by the setters (reconfiguration quiesces both streams; the
//           reference holds m_protect across setFFTSize too, Spectrogram.cpp:162-167).
//   rd_mu   serialises the readers (getMem / display ticks) with each other and with the setters; a reader keeps it
//           while it waits for its copies, the producer never takes it.  Lock order: rd_mu, then mu.
//   stream  compute stream: H2D of the block, stft_db_kernel, tail copy.
//   rstream read-out stream: copies of ring columns, colour kernel, image copies.  A reader records an event on
//           `stream`, lets `rstream` wait for it (so it sees every column produced before the call), enqueues its ring
//           reads, and makes `stream` wait for the event behind those reads: later kernels start after the snapshot
//           was taken -- a dependency on the GPU, not on the audio thread.
//
// Producer (round 4; reference call path PluginProcessor.cpp:145-150 -> Spectrogram::processSynchronBlock, Spectrogram.cpp:37-48):
//   jsg_process_block is WAIT-FREE: it copies the block into a slot of a single-producer / single-consumer ring of page-locked
//   memory and publishes it with one atomic store -- no mutex, no HIP call, no system call, no allocation, never a wait.  A worker
//   thread that the engine owns takes the blocks out in order and does every HIP call (H2D copy, launch) under `mu`, exactly
//   what the audio thread did itself until round 3.  A full ring (the GPU more than 64 blocks behind) or an engine in the
//   middle of a geometry change DROPS the block and counts it (return value 1, jsg_get_dropped_blocks) instead of blocking.
//   Geometry changes (channel count, FFT size: the slot size changes) use an epoch instead of a lock on the producer's side:
//   cfg_gen goes odd, the setter waits until the one producer call that may be in flight has left (prod_inflight), drains the
//   queue, rebuilds, and makes cfg_gen even again; a producer that sees an odd value drops its block (history is wiped by that
//   setter anyway, reference buildmem, Spectrogram.cpp:213-238).  Readers, the other setters, jsg_sync and the batch entry
//   points first wait until the worker has taken every block that was pushed before they were called (flush_queue): a block whose
//   jsg_process_block has returned is part of everything that is read afterwards, as before.
//
// There is no CPU compute path here: without a usable HIP device every entry point reports an error.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "jsg_block_queue.h"
#include "jsg_internal.h"

// (Rounds 2-3 had a library constructor here that set GPU_MAX_HW_QUEUES for the process.  It is gone: a plugin is loaded into a
// multi-threaded host, setenv is not thread-safe there, and the variable changes every other HIP user of the process.  The
// multi-stream launch pool -- jsg_stft_db_launch_batches -- documents the variable in INTEGRATION.md instead, and the headline path
// no longer needs it: jsg_stft_db_launch_strided is one launch on one stream.)
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
    std::mutex mu;      // state + enqueue sections
    std::mutex rd_mu;   // readers and setters (taken before mu)
    std::mutex err_mu;  // the error text only
    int device = 0;
    hipStream_t stream = nullptr;    // compute
    hipStream_t rstream = nullptr;   // read-out
    hipEvent_t ev_ring = nullptr;    // "every column produced so far" (recorded on stream, awaited by rstream)
    hipEvent_t ev_read = nullptr;    // "the ring has been read" (recorded on rstream, awaited by stream)
    hipEvent_t ev_host = nullptr;    // "getMem's columns have arrived in h_mem" (recorded on rstream, awaited by the reader)
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
    int exact_log = 0;       // jsg_set_exact_log: dB by the shared float32 routine (bit-reproducible on a CPU) instead of v_log_f32
    std::vector<float> window;

    // derived geometry; the atomics mirror W, H, n and the channel count for the lock-free getters
    int hop = 1024, W = 0, H = 513;
    int64_t pitch = 0;
    std::atomic<int> a_W{0}, a_H{513}, a_n{1024}, a_channels{2}, a_hop{1024}, a_feedblocks{1};
    std::atomic<float> a_fs{48000.f};

    // device state
    jsg_plan* plan = nullptr;
    float* d_ring = nullptr;
    size_t ring_floats = 0;
    float* d_in = nullptr;
    int64_t in_pitch = 0;
    int in_cap_blocks = 0;
    int mem_counter = 0;
    long long new_entry = kNewEntrySentinel;
    unsigned long long generation = 0;   // bumped by every reconfiguration (buildmem)

    // the audio thread's block queue (jsg_process_block; jsg_block_queue.h): single producer, single consumer = the worker thread.
    // Its slots live in page-locked memory that the geometry setters (and creation) allocate: 64 x [channels][n] floats.
    BlockQueue q;
    std::atomic<int> async_rc{0};                  // first error the worker met (text in `err`); reported by the calls that follow
    hipEvent_t ev_q = nullptr;                     // "the H2D copy has read the slot"
    std::thread worker;
    std::atomic<bool> stop{false};

    // getMem (readers only, under rd_mu): d_snap takes a device-side copy of the wanted ring columns (microseconds; the
    // only part later kernels have to wait for), h_mem is the page-locked landing area of the slow copy across PCIe
    float* d_snap = nullptr;
    size_t d_snap_floats = 0;
    float* h_mem = nullptr;
    size_t h_mem_floats = 0;

    // display (d_img and the flags below belong to the readers: rd_mu; the setters hold it too)
    int n_colors = 256, scheme = JSG_CM_JADE;   // CColorPalette(256,6), Spectrogram.cpp:337
    std::vector<int32_t> lut;
    int32_t* d_lut = nullptr;
    uint32_t* d_img = nullptr;
    int img_w = 0, img_h = 0;
    int64_t img_pitch = 0;   // pixels per device image row (img_w rounded up to 32)
    bool recompute_all = true;
    bool running = true;
    // what the caller's image holds after the last jsg_display_update (fixed display: only changed columns are copied)
    const uint32_t* host_img = nullptr;
    int64_t host_pitch = 0;
    bool host_fixed_valid = false;
    int host_cursor_pos = 0, host_cursor_w = 0;

    int fail(int code, const std::string& what) {
        {
            std::lock_guard<std::mutex> lk(err_mu);
            err = what;
        }
        tls_error() = what;
        return code;
    }
    int fail_hip(hipError_t e, const char* where) {
        const int code = jsg_fail_hip(e, where);
        std::lock_guard<std::mutex> lk(err_mu);
        err = tls_error();
        return code;
    }
    int fail_tls(int code) {   // a stateless entry point failed and left its text in tls_error()
        std::lock_guard<std::mutex> lk(err_mu);
        err = tls_error();
        return code;
    }
    int planes() const { return mix == JSG_MIX_PER_CHANNEL ? channels : 1; }
    void publish() {
        a_W.store(W); a_H.store(H); a_n.store(n); a_channels.store(channels); a_hop.store(hop);
        a_feedblocks.store(feedblocks); a_fs.store(fs);
    }
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
    if (rc != JSG_OK) return e->fail_tls(rc);
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

// both streams idle (setters; they hold rd_mu and mu)
int quiesce(jsg_engine* e) {
    JSG_HIP(e, hipStreamSynchronize(e->stream));
    JSG_HIP(e, hipStreamSynchronize(e->rstream));
    return JSG_OK;
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
// Called with rd_mu and mu held.
int buildmem(jsg_engine* e) {
    // A changed slot geometry (channel count, FFT size: only the geometry setters and creation come here with one, inside a GeomEpoch
    // -- no producer call is inside, the worker has drained the queue) CLOSES the producer's queue first: whatever fails below, the
    // audio thread finds a queue without slots and drops its blocks (try_push) instead of copying into slots of the old size that the
    // worker would then read with the new one.  The queue opens again at the end of a build that succeeded.
    const bool slot_geometry_changed = e->q.channels != e->channels || e->q.n != e->n;
    if (slot_geometry_changed) {
        e->q.channels = 0;
        e->q.n = 0;
    }
    JSG_HIP(e, hipSetDevice(e->device));
    e->hop = jsg_feed_samples(e->feed_percent, e->n);
    if (e->hop <= 0) return e->fail(JSG_ERR_INVALID, "feed percentage gives a hop of 0 samples");
    e->W = jsg_memsize_blocks(e->memsize_s, e->fs, e->hop);
    if (e->W <= 0) return e->fail(JSG_ERR_INVALID, "memory time gives an empty ring");
    e->H = e->n / 2 + 1;
    e->pitch = round_up(e->H, 32);
    const size_t need = size_t(e->W) * size_t(e->pitch) * size_t(e->planes());
    int rc = quiesce(e);
    if (rc != JSG_OK) return rc;
    ++e->generation;
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
    // input ring: drop and re-create zeroed (channel count or fft size may have changed); one block of capacity is
    // always there, so jsg_process_block never allocates
    if (e->d_in) (void)hipFree(e->d_in);
    e->d_in = nullptr;
    e->in_cap_blocks = 0;
    rc = ensure_input_capacity(e, 1, false);
    if (rc != JSG_OK) return rc;
    // the producer's queue: only the geometry setters (and creation) come here with a changed slot size; they have made sure that
    // no producer call is inside and that the worker has drained the queue (GeomEpoch)
    const size_t slot = size_t(e->channels) * size_t(e->n);
    if (slot != e->q.slot_floats) {
        if (e->q.mem) (void)hipHostFree(e->q.mem);
        e->q.mem = nullptr;
        e->q.slot_floats = 0;
        JSG_HIP(e, hipHostMalloc(reinterpret_cast<void**>(&e->q.mem), slot * BlockQueue::kSlots * sizeof(float), hipHostMallocDefault));
        e->q.slot_floats = slot;
    }
    if (!e->ev_q) JSG_HIP(e, hipEventCreateWithFlags(&e->ev_q, hipEventDisableTiming));
    e->q.channels = e->channels;   // the queue's geometry is published only together with slots of that size
    e->q.n = e->n;
    e->mem_counter = 0;
    e->new_entry = kNewEntrySentinel;
    e->host_fixed_valid = false;
    e->publish();
    return JSG_OK;
}

int upload_lut(jsg_engine* e) {
    e->lut.assign(size_t(e->n_colors), 0);
    int rc = jsg_colormap_build(e->n_colors, e->scheme, e->lut.data());
    if (rc != JSG_OK) return e->fail(rc, "jsg_colormap_build failed");
    rc = quiesce(e);
    if (rc != JSG_OK) return rc;
    if (e->d_lut) (void)hipFree(e->d_lut);
    e->d_lut = nullptr;
    JSG_HIP(e, hipMalloc(reinterpret_cast<void**>(&e->d_lut), size_t(e->n_colors) * 4));
    JSG_HIP(e, hipMemcpy(e->d_lut, e->lut.data(), size_t(e->n_colors) * 4, hipMemcpyHostToDevice));
    return JSG_OK;
}

// frames of `blocks` new blocks whose samples already sit behind the tail in d_in (mu held; enqueues only)
int run_blocks(jsg_engine* e, int blocks) {
    const long long frames = (long long)blocks * e->feedblocks;
    if (!e->pause && frames > 0) {   // paused: the reference computes and drops the columns (Spectrogram.cpp:111)
        // only the newest W columns survive in the ring; never let one launch write a column twice
        const long long skip = frames > e->W ? frames - e->W : 0;
        jsg_stft_args a{};
        a.in = e->d_in;
        a.in_pitch = e->in_pitch;
        a.in_samples = int64_t(blocks + 1) * e->n;
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
        a.plan_select = 1;   // the ring must not depend on how the host cut the stream into calls: always the same kernel
        a.exact_log = e->exact_log;
        const int rc = jsg_stft_db_launch(e->plan, &a, e->stream);
        if (rc != JSG_OK) return e->fail_tls(rc);
        e->mem_counter = int((e->mem_counter + frames) % e->W);
        e->new_entry += frames;
        if (e->new_entry > 2000000000ll) e->new_entry = 2000000000ll;   // the reference's int would overflow
    }
    // keep the last block as the next call's history (reference Spectrogram.cpp:121-131)
    JSG_HIP(e, hipMemcpy2DAsync(e->d_in, size_t(e->in_pitch) * 4, e->d_in + size_t(blocks) * e->n, size_t(e->in_pitch) * 4,
                                size_t(e->n) * 4, size_t(e->channels), hipMemcpyDeviceToDevice, e->stream));
    return JSG_OK;
}

// Reader protocol, part 1 (mu held): everything produced so far becomes visible to rstream.
int reader_begin(jsg_engine* e) {
    JSG_HIP(e, hipEventRecord(e->ev_ring, e->stream));
    JSG_HIP(e, hipStreamWaitEvent(e->rstream, e->ev_ring, 0));
    return JSG_OK;
}
// Reader protocol, part 2 (mu held, after the ring reads were enqueued on rstream): later kernels wait for them.
int reader_end(jsg_engine* e) {
    JSG_HIP(e, hipEventRecord(e->ev_read, e->rstream));
    JSG_HIP(e, hipStreamWaitEvent(e->stream, e->ev_read, 0));
    return JSG_OK;
}

// ---- the worker thread: takes the audio thread's blocks out of the queue in order and does what processSynchronBlock does ----
void drain_queue(jsg_engine* e) {
    unsigned long long t;
    const float* block;
    bool current;
    while (e->q.front(t, block, current)) {
        bool copied = false;
        {
            std::lock_guard<std::mutex> lk(e->mu);
            // a block that was pushed under another geometry than the current one is discarded (cannot happen while the geometry
            // setters drain the queue before they rebuild; kept as a guard)
            if (current && e->async_rc.load(std::memory_order_relaxed) == 0) {
                int rc = JSG_OK;
                hipError_t herr = hipMemcpy2DAsync(e->d_in + e->n, size_t(e->in_pitch) * 4, block, size_t(e->n) * 4, size_t(e->n) * 4,
                                                   size_t(e->channels), hipMemcpyHostToDevice, e->stream);
                if (herr == hipSuccess) herr = hipEventRecord(e->ev_q, e->stream);
                if (herr != hipSuccess) rc = e->fail_hip(herr, "worker: H2D copy of a block");
                else {
                    copied = true;
                    rc = run_blocks(e, 1);
                }
                if (rc != JSG_OK) e->async_rc.store(rc, std::memory_order_release);
            }
        }
        if (copied) (void)hipEventSynchronize(e->ev_q);   // the DMA has read the slot (outside the state lock)
        else e->q.dropped.fetch_add(1, std::memory_order_relaxed);   // discarded (an earlier error of the worker, or a stale geometry): not in the ring
        e->q.pop(t);
    }
}

// The worker POLLS the queue instead of being woken by the producer: a wake-up (sem_post / futex) is a system call on the audio thread,
// and the scheduler may run the woken thread in the caller's place.  250 us between polls while blocks keep coming (a block of the
// plugin lasts 5-90 ms), 1 ms after 10 ms of silence: about a thousand short wake-ups per second and engine when idle.
void worker_main(jsg_engine* e) {
    (void)hipSetDevice(e->device);
    int idle = 0;
    while (!e->stop.load(std::memory_order_acquire)) {
        unsigned long long t;
        const float* block;
        bool current;
        if (e->q.front(t, block, current)) {
            drain_queue(e);
            idle = 0;
        } else {
            std::this_thread::sleep_for(std::chrono::microseconds(idle < 40 ? 250 : 1000));
            if (idle < 1000000) ++idle;
        }
    }
}

// Every block whose jsg_process_block returned before this call is on the stream afterwards.  Never called with mu or rd_mu held
// (the worker needs mu), and never from the audio thread.
void flush_queue(jsg_engine* e) { e->q.wait_drained(); }

// Geometry setters (channel count, FFT size): see "Producer" in the header comment.  Constructed BEFORE the setter takes its locks.
struct GeomEpoch {
    jsg_engine* e;
    explicit GeomEpoch(jsg_engine* eng) : e(eng) { if (e) e->q.begin_geometry_change(); }
    ~GeomEpoch() { if (e) e->q.end_geometry_change(); }
};

int create_on(jsg_engine** out, int channels, int device) {
    if (!out || channels <= 0) return jsg_fail(JSG_ERR_INVALID, "jsg_create: bad argument");
    *out = nullptr;
    int ndev = 0;
    hipError_t herr = hipGetDeviceCount(&ndev);
    if (herr != hipSuccess || ndev <= 0)
        return jsg_fail(JSG_ERR_NO_DEVICE, "jsg_create: no HIP device available (this engine has no CPU fallback)");
    if (device >= ndev) return jsg_fail(JSG_ERR_NO_DEVICE, "jsg_create_on_device: no such device");
    jsg_engine* e = new (std::nothrow) jsg_engine();
    if (!e) return jsg_fail(JSG_ERR_NOMEM, "jsg_create: out of memory");
    e->channels = channels;
    if (device < 0) herr = hipGetDevice(&e->device);
    else {
        e->device = device;
        herr = hipSetDevice(device);
    }
    if (herr == hipSuccess) herr = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
    if (herr == hipSuccess) herr = hipStreamCreateWithFlags(&e->rstream, hipStreamNonBlocking);
    if (herr == hipSuccess) herr = hipEventCreateWithFlags(&e->ev_ring, hipEventDisableTiming);
    if (herr == hipSuccess) herr = hipEventCreateWithFlags(&e->ev_read, hipEventDisableTiming);
    if (herr == hipSuccess) herr = hipEventCreateWithFlags(&e->ev_host, hipEventDisableTiming);
    if (herr != hipSuccess) {
        const int code = jsg_fail_hip(herr, "jsg_create");
        jsg_destroy(e);
        return code;
    }
    int rc;
    {
        std::lock_guard<std::mutex> rlk(e->rd_mu);
        std::lock_guard<std::mutex> lk(e->mu);
        rc = buildmem(e);
        if (rc == JSG_OK) rc = build_window(e);   // (the reference leaves m_window empty until setFFTSize/setWindow)
        if (rc == JSG_OK) rc = upload_lut(e);
    }
    if (rc == JSG_OK) {
        try {
            e->worker = std::thread(worker_main, e);
        } catch (...) {
            rc = e->fail(JSG_ERR_NOMEM, "jsg_create: the worker thread could not be started");
        }
    }
    if (rc != JSG_OK) {
        tls_error() = e->err;
        jsg_destroy(e);
        return rc;
    }
    *out = e;
    return JSG_OK;
}

}  // namespace

extern "C" {

const char* jsg_last_error(const jsg_engine* e) {
    if (!e) return tls_error().c_str();
    static thread_local std::string copy;   // the engine's text may be rewritten by another thread: hand out a private copy
    jsg_engine* m = const_cast<jsg_engine*>(e);
    std::lock_guard<std::mutex> lk(m->err_mu);
    copy = m->err;
    return copy.c_str();
}

int jsg_create(jsg_engine** out, int channels) { return create_on(out, channels, -1); }

int jsg_create_on_device(jsg_engine** out, int channels, int device) {
    if (device < 0) return jsg_fail(JSG_ERR_INVALID, "jsg_create_on_device: negative device index");
    return create_on(out, channels, device);
}

int jsg_get_device(const jsg_engine* e) { return e ? e->device : JSG_ERR_INVALID; }

// Several GPUs from one host process (SURVEY 8b "jsg_create_sharded(devices[], n)", 8e): one engine per entry of `devices`, the
// `channels` channels of one stream dealt out in contiguous runs whose sizes differ by at most one (BASELINE configs[3]: 64
// channels -> 8 per GPU).  Entry i gets channels [first_channel[i], first_channel[i] + channel_count[i]); an entry that gets no
// channel gets no engine (out[i] = NULL).  The engines share nothing: outputs stay on their device, there is no collective.
int jsg_create_sharded(jsg_engine** out, int* first_channel, int* channel_count, const int* devices, int n_devices, int channels) {
    if (!out || !first_channel || !channel_count || !devices || n_devices <= 0 || channels <= 0)
        return jsg_fail(JSG_ERR_INVALID, "jsg_create_sharded: bad argument");
    for (int i = 0; i < n_devices; ++i) out[i] = nullptr;
    const int base = channels / n_devices, extra = channels % n_devices;
    for (int i = 0; i < n_devices; ++i) {
        first_channel[i] = i * base + (i < extra ? i : extra);
        channel_count[i] = base + (i < extra ? 1 : 0);
        if (channel_count[i] == 0) continue;
        const int rc = devices[i] < 0 ? jsg_fail(JSG_ERR_INVALID, "jsg_create_sharded: negative device index")
                                      : create_on(&out[i], channel_count[i], devices[i]);
        if (rc != JSG_OK) {
            const std::string keep = tls_error();
            for (int k = 0; k < i; ++k) { (void)jsg_destroy(out[k]); out[k] = nullptr; }
            tls_error() = keep;
            return rc;
        }
    }
    return JSG_OK;
}

// processSynchronBlock for the sharded set: every engine takes its own run of the planar channel pointers.  ALL OR NOTHING: the rings
// of the shards must advance together (a cross-GPU AbsMean, or per-channel columns shown side by side, rely on the same ring position on
// every shard), so the call first asks every engine whether it would take a block now -- queue not full, no geometry change in progress,
// no earlier worker error -- and pushes only if all of them would.  Otherwise NO engine gets the block, every engine counts one dropped
// block, and 1 is returned (< 0: the first engine's stored error).  The check stays true until the push for everything the producer
// thread can cause (single producer: slots only get freed in between); the one thing that can still come between is a geometry setter of
// ONE engine from the message thread -- then the engines that took the block are ahead by one, the return value is that engine's 1, and
// the caller has to resynchronise the set, which a geometry change of a sharded set needs anyway (every setter wipes the history:
// buildmem).  Enqueue only, wait-free like jsg_process_block.
int jsg_process_block_sharded(jsg_engine* const* engines, const int* first_channel, int n_devices, const float* const* planar) {
    if (!engines || !first_channel || n_devices <= 0 || !planar) return jsg_fail(JSG_ERR_INVALID, "jsg_process_block_sharded: bad argument");
    bool all = true;
    for (int i = 0; i < n_devices; ++i) {
        if (!engines[i]) continue;
        const int arc = engines[i]->async_rc.load(std::memory_order_acquire);
        if (arc != 0) {   // an engine with a stored worker error takes nothing any more: no shard gets the block, every shard counts it
            for (int k = 0; k < n_devices; ++k)
                if (engines[k]) engines[k]->q.dropped.fetch_add(1, std::memory_order_relaxed);
            return arc;
        }
        if (!engines[i]->q.can_push(0, 0)) all = false;
    }
    if (!all) {
        for (int i = 0; i < n_devices; ++i)
            if (engines[i]) engines[i]->q.dropped.fetch_add(1, std::memory_order_relaxed);
        return 1;
    }
    int worst = JSG_OK;
    for (int i = 0; i < n_devices; ++i) {
        if (!engines[i]) continue;
        const int rc = jsg_process_block(engines[i], planar + first_channel[i]);
        // (a worker error that appears between the scan above and this push: the shards before i have taken the block and cannot give it
        //  back; the error is final for that engine -- every later call returns it before anything is pushed -- so the set is out of step by at
        //  most this one block, and the caller is told with the error code)
        if (rc < 0) return rc;
        if (rc > worst) worst = rc;
    }
    return worst;
}

int jsg_destroy_sharded(jsg_engine** engines, int n_devices) {
    if (!engines) return JSG_OK;
    for (int i = 0; i < n_devices; ++i) {
        (void)jsg_destroy(engines[i]);
        engines[i] = nullptr;
    }
    return JSG_OK;
}

int jsg_destroy(jsg_engine* e) {
    if (!e) return JSG_OK;
    if (e->worker.joinable()) {
        e->stop.store(true, std::memory_order_release);
        e->worker.join();
    }
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    if (e->rstream) (void)hipStreamSynchronize(e->rstream);
    if (e->plan) jsg_plan_destroy(e->plan);
    if (e->d_ring) (void)hipFree(e->d_ring);
    if (e->d_in) (void)hipFree(e->d_in);
    if (e->d_lut) (void)hipFree(e->d_lut);
    if (e->d_img) (void)hipFree(e->d_img);
    if (e->q.mem) (void)hipHostFree(e->q.mem);
    if (e->h_mem) (void)hipHostFree(e->h_mem);
    if (e->d_snap) (void)hipFree(e->d_snap);
    if (e->ev_host) (void)hipEventDestroy(e->ev_host);
    if (e->ev_q) (void)hipEventDestroy(e->ev_q);
    if (e->ev_ring) (void)hipEventDestroy(e->ev_ring);
    if (e->ev_read) (void)hipEventDestroy(e->ev_read);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    if (e->rstream) (void)hipStreamDestroy(e->rstream);
    delete e;
    return JSG_OK;
}

// producer / short sections: the state lock only
#define JSG_LOCK(e)                                                          \
    if (!(e)) return jsg_fail(JSG_ERR_INVALID, "null engine");               \
    flush_queue(e);                                                          \
    std::lock_guard<std::mutex> _lk((e)->mu);                                \
    do {                                                                     \
        hipError_t _e = hipSetDevice((e)->device);                           \
        if (_e != hipSuccess) return (e)->fail_hip(_e, "hipSetDevice");      \
    } while (0)

// setters: exclude the readers as well (they may free what a reader is still copying from)
#define JSG_LOCK_CONFIG(e)                                                   \
    if (!(e)) return jsg_fail(JSG_ERR_INVALID, "null engine");               \
    flush_queue(e);                                                          \
    std::lock_guard<std::mutex> _rlk((e)->rd_mu);                            \
    std::lock_guard<std::mutex> _lk((e)->mu);                                \
    do {                                                                     \
        hipError_t _e = hipSetDevice((e)->device);                           \
        if (_e != hipSuccess) return (e)->fail_hip(_e, "hipSetDevice");      \
    } while (0)

int jsg_set_samplerate(jsg_engine* e, float fs) {
    GeomEpoch epoch(e);   // every rebuild is bracketed: a queue that an earlier failed geometry change left closed may be re-opened here
    JSG_LOCK_CONFIG(e);
    if (!(fs > 0.f)) return e->fail(JSG_ERR_INVALID, "sample rate must be positive");
    e->fs = fs;
    return buildmem(e);
}

int jsg_set_channels(jsg_engine* e, int channels) {
    GeomEpoch epoch(e);
    JSG_LOCK_CONFIG(e);
    if (channels <= 0) return e->fail(JSG_ERR_INVALID, "channel count must be positive");
    e->channels = channels;
    return buildmem(e);
}

int jsg_set_fft_size(jsg_engine* e, int n) {
    GeomEpoch epoch(e);
    JSG_LOCK_CONFIG(e);
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
    const int n = jsg_next_power_of_2(ms, e->a_fs.load());
    return jsg_set_fft_size(e, n);
}

int jsg_set_memory_time_s(jsg_engine* e, float seconds) {
    GeomEpoch epoch(e);   // every rebuild is bracketed: a queue that an earlier failed geometry change left closed may be re-opened here
    JSG_LOCK_CONFIG(e);
    if (!(seconds > 0.f)) return e->fail(JSG_ERR_INVALID, "memory time must be positive");
    e->memsize_s = seconds;
    return buildmem(e);
}

int jsg_set_feed_percent(jsg_engine* e, int feed) {
    GeomEpoch epoch(e);   // every rebuild is bracketed: a queue that an earlier failed geometry change left closed may be re-opened here
    JSG_LOCK_CONFIG(e);
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
    GeomEpoch epoch(e);   // every rebuild is bracketed: a queue that an earlier failed geometry change left closed may be re-opened here
    JSG_LOCK_CONFIG(e);
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
    JSG_LOCK_CONFIG(e);
    if (window < JSG_WIN_RECT || window > JSG_WIN_HANNPOISSON) return e->fail(JSG_ERR_INVALID, "unknown window");
    e->window_choice = window;
    e->window_custom = false;
    JSG_HIP(e, hipStreamSynchronize(e->stream));
    return build_window(e);
}

int jsg_set_window_table(jsg_engine* e, const float* w, int n) {
    JSG_LOCK_CONFIG(e);
    if (!w || n != e->n) return e->fail(JSG_ERR_SIZE_MISMATCH, "window table must have fft-size entries");
    e->window.assign(w, w + n);
    e->window_custom = true;
    JSG_HIP(e, hipStreamSynchronize(e->stream));
    return rebuild_plan(e);
}

int jsg_set_mix_mode(jsg_engine* e, int mode) {
    GeomEpoch epoch(e);   // every rebuild is bracketed: a queue that an earlier failed geometry change left closed may be re-opened here
    JSG_LOCK_CONFIG(e);
    const bool known = (mode >= JSG_MIX_ABSMEAN && mode <= JSG_MIX_RIGHT) || mode == JSG_MIX_PER_CHANNEL;
    if (!known) return e->fail(JSG_ERR_INVALID, "unknown mix mode");
    if (mode == JSG_MIX_RIGHT && e->channels < 2) return e->fail(JSG_ERR_INVALID, "JSG_MIX_RIGHT needs two channels");
    const bool replane = (mode == JSG_MIX_PER_CHANNEL) != (e->mix == JSG_MIX_PER_CHANNEL);
    e->mix = mode;
    return replane ? buildmem(e) : JSG_OK;
}

int jsg_set_power_scale(jsg_engine* e, float scale) {
    JSG_LOCK_CONFIG(e);
    if (!(scale > 0.f)) return e->fail(JSG_ERR_INVALID, "power scale must be positive");
    e->power_scale = scale;
    JSG_HIP(e, hipStreamSynchronize(e->stream));
    return rebuild_plan(e);
}

int jsg_set_exact_log(jsg_engine* e, int on) {
    JSG_LOCK_CONFIG(e);
    e->exact_log = on ? 1 : 0;   // (columns already in the ring keep the logarithm they were written with)
    return JSG_OK;
}

// lock-free getters (the GUI polls them while the audio thread runs)
int jsg_get_spectrum_size(const jsg_engine* e) { return e ? e->a_H.load() : JSG_ERR_INVALID; }
int jsg_get_memory_size(const jsg_engine* e) { return e ? e->a_W.load() : JSG_ERR_INVALID; }
float jsg_get_samplerate(const jsg_engine* e) { return e ? e->a_fs.load() : 0.f; }
int jsg_get_fft_size(const jsg_engine* e) { return e ? e->a_n.load() : JSG_ERR_INVALID; }
int jsg_get_feed_samples(const jsg_engine* e) { return e ? e->a_hop.load() : JSG_ERR_INVALID; }
int jsg_get_feedblocks(const jsg_engine* e) { return e ? e->a_feedblocks.load() : JSG_ERR_INVALID; }
int jsg_get_channels(const jsg_engine* e) { return e ? e->a_channels.load() : JSG_ERR_INVALID; }

int jsg_get_window(const jsg_engine* ce, float* out, int n) {
    jsg_engine* e = const_cast<jsg_engine*>(ce);
    if (!e || !out) return jsg_fail(JSG_ERR_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(e->mu);
    if (n != int(e->window.size())) return jsg_fail(JSG_ERR_SIZE_MISMATCH, "window size mismatch");
    std::memcpy(out, e->window.data(), size_t(n) * sizeof(float));
    return JSG_OK;
}

// Spectrogram::processSynchronBlock (reference Spectrogram.cpp:37-48: the copy-in; the frame loop runs on the worker thread).
// WAIT-FREE on the caller's thread: two atomic read-modify-writes, channels x memcpy into page-locked memory, one atomic store; no
// system call (the worker polls).  channels / n (> 0): the geometry the caller's pointers were sized for; a block that does not fit the engine's current
// geometry is dropped (a host class whose FFT-size combo box has just been moved, reference Spectrogram.cpp:760-767).
// Returns 0 (queued), 1 (dropped: ring full, geometry change in progress, or geometry mismatch), < 0 (error; also an error the worker
// thread met earlier).
int jsg_process_block_n(jsg_engine* e, const float* const* planar, int channels, int n) {
    if (!e) return jsg_fail(JSG_ERR_INVALID, "null engine");
    if (!planar) return e->fail(JSG_ERR_INVALID, "null block");
    const int arc = e->async_rc.load(std::memory_order_acquire);
    if (arc != 0) {
        e->q.dropped.fetch_add(1, std::memory_order_relaxed);   // not in the ring either
        return arc;
    }
    const int rc = e->q.push(planar, channels, n);
    if (rc < 0) return e->fail(JSG_ERR_INVALID, "null channel pointer");
    return rc;
}

int jsg_process_block(jsg_engine* e, const float* const* planar) { return jsg_process_block_n(e, planar, 0, 0); }

// The LOSSLESS form for callers that are not bound to real time (a DAW's offline bounce -- juce::AudioProcessor::isNonRealtime() --,
// file converters, test loops): where jsg_process_block would drop a block because the ring is full (the caller runs faster than the GPU
// takes blocks out) this call WAITS -- yields, then sleeps 100 us at a time -- until a slot is free, for at most timeout_ms milliseconds
// (< 0: no limit).  The reference never drops a block (processSynchronBlock computes in place, Spectrogram.cpp:37-135); this is the
// entry point that keeps that property.  Returns 0 (queued), 1 (dropped and counted: geometry change in progress / geometry mismatch,
// or still full at the timeout), < 0 (error).  NOT for the audio thread of a live host.
int jsg_process_block_wait(jsg_engine* e, const float* const* planar, int channels, int n, int timeout_ms) {
    if (!e) return jsg_fail(JSG_ERR_INVALID, "null engine");
    if (!planar) return e->fail(JSG_ERR_INVALID, "null block");
    const auto t0 = std::chrono::steady_clock::now();
    for (int spins = 0;; ++spins) {
        const int arc = e->async_rc.load(std::memory_order_acquire);
        if (arc != 0) {
            e->q.dropped.fetch_add(1, std::memory_order_relaxed);
            return arc;
        }
        const int rc = e->q.try_push(planar, channels, n, false);
        if (rc == 0) return 0;
        if (rc < 0) return e->fail(JSG_ERR_INVALID, "null channel pointer");
        const bool timed_out = timeout_ms >= 0 && std::chrono::steady_clock::now() - t0 >= std::chrono::milliseconds(timeout_ms);
        if (rc == 2 || timed_out) {   // a geometry change wipes the history anyway: nothing to wait for
            e->q.dropped.fetch_add(1, std::memory_order_relaxed);
            return 1;
        }
        if (spins < 64) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
}

long long jsg_get_dropped_blocks(const jsg_engine* e) { return e ? (long long)e->q.dropped.load(std::memory_order_relaxed) : JSG_ERR_INVALID; }

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

namespace {

// Spectrogram::getMem (Spectrogram.cpp:295-331).  The new columns (or all of them) travel ring -> page-locked h_mem on
// rstream; the caller's buffer is filled from h_mem after the copy has completed, outside the state lock.
// dst_dense: [planes*W][H] floats, or rows: planes*W pointers to H floats each (vector<vector<float>> of the host class).
int get_mem_impl(jsg_engine* e, float* dst_dense, float* const* rows, int dst_columns, int* pos, bool peek = false) {
    if (!e) return jsg_fail(JSG_ERR_INVALID, "null engine");
    flush_queue(e);   // every block pushed before this call is on the stream
    std::lock_guard<std::mutex> rlk(e->rd_mu);
    // Geometry cannot change while rd_mu is held (every setter takes it first), so the reader's own buffers are
    // (re)allocated here, OUTSIDE the state lock: page-locking 15 MB takes milliseconds and must not stall the producer.
    hipError_t herr = hipSetDevice(e->device);
    if (herr != hipSuccess) return e->fail_hip(herr, "hipSetDevice");
    const int planes = e->planes();
    const int W = e->W, H = e->H;
    int mem_counter;
    long long nec;
    struct Span { int first, count; } spans[2] = {{0, 0}, {0, 0}};
    if ((!dst_dense && !rows) || dst_columns != W * planes) return e->fail(JSG_ERR_SIZE_MISMATCH, "getMem: buffer size mismatch");
    const size_t need = size_t(W) * size_t(H) * size_t(planes);
    if (need > e->h_mem_floats) {   // first read after a (re)configuration
        if (e->h_mem) (void)hipHostFree(e->h_mem);
        e->h_mem = nullptr;
        e->h_mem_floats = 0;
        JSG_HIP(e, hipHostMalloc(reinterpret_cast<void**>(&e->h_mem), need * sizeof(float), hipHostMallocDefault));
        e->h_mem_floats = need;
    }
    if (e->ring_floats > e->d_snap_floats) {
        if (e->d_snap) (void)hipFree(e->d_snap);
        e->d_snap = nullptr;
        e->d_snap_floats = 0;
        JSG_HIP(e, hipMalloc(reinterpret_cast<void**>(&e->d_snap), e->ring_floats * sizeof(float)));
        e->d_snap_floats = e->ring_floats;
    }
    {
        std::lock_guard<std::mutex> lk(e->mu);
        nec = e->new_entry;
        mem_counter = e->mem_counter;
        if (peek || nec >= W) {                           // Spectrogram.cpp:300-304
            spans[0] = {0, W};
        } else {
            const int start = mem_counter - int(nec);     // :307
            if (start >= 0) {
                spans[0] = {start, int(nec)};             // :310-311
            } else {
                spans[0] = {0, mem_counter};              // :315-316
                spans[1] = {W + start, -start};           // :318-319
            }
        }
        int rc = reader_begin(e);
        if (rc != JSG_OK) return rc;
        // 1. ring -> d_snap on the device (whole columns incl. padding: contiguous, HBM speed).  Only this holds up later kernels.
        for (const Span& s : spans) {
            if (s.count <= 0) continue;
            for (int p = 0; p < planes; ++p) {
                const size_t off = (size_t(p) * W + s.first) * size_t(e->pitch);
                JSG_HIP(e, hipMemcpyAsync(e->d_snap + off, e->d_ring + off, size_t(s.count) * e->pitch * sizeof(float),
                                          hipMemcpyDeviceToDevice, e->rstream));
            }
        }
        rc = reader_end(e);
        if (rc != JSG_OK) return rc;
        // 2. d_snap -> page-locked host memory, dense [W][H]
        for (const Span& s : spans) {
            if (s.count <= 0) continue;
            for (int p = 0; p < planes; ++p)
                JSG_HIP(e, hipMemcpy2DAsync(e->h_mem + (size_t(p) * W + s.first) * H, size_t(H) * 4,
                                            e->d_snap + (size_t(p) * W + s.first) * e->pitch, size_t(e->pitch) * 4, size_t(H) * 4,
                                            size_t(s.count), hipMemcpyDeviceToHost, e->rstream));
        }
        JSG_HIP(e, hipEventRecord(e->ev_host, e->rstream));
        if (!peek) e->new_entry = 0;                      // :328
    }
    JSG_HIP(e, hipEventSynchronize(e->ev_host));           // the producer is not held up by this wait
    for (const Span& s : spans) {
        if (s.count <= 0) continue;
        for (int p = 0; p < planes; ++p) {
            const float* src = e->h_mem + (size_t(p) * W + s.first) * H;
            if (dst_dense) {
                std::memcpy(dst_dense + (size_t(p) * W + s.first) * H, src, size_t(s.count) * H * sizeof(float));
            } else {
                for (int c = 0; c < s.count; ++c) {
                    float* row = rows[size_t(p) * W + s.first + c];
                    if (row) std::memcpy(row, src + size_t(c) * H, size_t(H) * sizeof(float));
                }
            }
        }
    }
    if (pos) *pos = mem_counter;                          // :329
    return int(nec);
}

}  // namespace

int jsg_get_mem(jsg_engine* e, float* dst, int dst_columns, int* pos) {
    if (e && !dst) return e->fail(JSG_ERR_SIZE_MISMATCH, "getMem: buffer size mismatch");
    return get_mem_impl(e, dst, nullptr, dst_columns, pos);
}

int jsg_peek_mem(jsg_engine* e, float* dst, int dst_columns, int* pos) {
    if (e && !dst) return e->fail(JSG_ERR_SIZE_MISMATCH, "peekMem: buffer size mismatch");
    const int rc = get_mem_impl(e, dst, nullptr, dst_columns, pos, true);
    return rc < 0 ? rc : int(std::min<long long>(rc, 2000000000ll));
}

int jsg_get_mem_rows(jsg_engine* e, float* const* rows, int n_rows, int row_len, int* pos) {
    if (!e) return jsg_fail(JSG_ERR_INVALID, "null engine");
    if (!rows || row_len != e->a_H.load()) return e->fail(JSG_ERR_SIZE_MISMATCH, "getMem: row length mismatch");
    return get_mem_impl(e, nullptr, rows, n_rows, pos);
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
    if (!e) return jsg_fail(JSG_ERR_INVALID, "null engine");
    flush_queue(e);
    const int arc = e->async_rc.load(std::memory_order_acquire);
    if (arc != 0) return arc;
    hipError_t herr = hipSetDevice(e->device);
    if (herr == hipSuccess) herr = hipStreamSynchronize(e->stream);   // no lock: the stream handle lives as long as the engine
    if (herr != hipSuccess) return e->fail_hip(herr, "jsg_sync");
    return JSG_OK;
}

void* jsg_stream(jsg_engine* e) { return e ? reinterpret_cast<void*>(e->stream) : nullptr; }

int jsg_display_set_colormap(jsg_engine* e, int n_colors, int scheme) {
    JSG_LOCK_CONFIG(e);
    if (n_colors <= 0 || n_colors > 65535 || scheme < JSG_CM_MONO || scheme > JSG_CM_JADE)
        return e->fail(JSG_ERR_INVALID, "bad colour map");
    e->n_colors = n_colors;
    e->scheme = scheme;
    e->recompute_all = true;   // Spectrogram.cpp:400
    return upload_lut(e);
}

int jsg_display_set_running(jsg_engine* e, int running) {
    JSG_LOCK_CONFIG(e);
    if (e->running != (running != 0)) e->host_fixed_valid = false;
    e->running = running != 0;
    return JSG_OK;
}

int jsg_display_invalidate(jsg_engine* e) {
    JSG_LOCK_CONFIG(e);
    e->recompute_all = true;
    return JSG_OK;
}

namespace {

struct Tick {   // what a display tick decided under the state lock
    int W = 0, H = 0, pos = 0, n_cols = 0, col_first = 0;
    long long nec = 0;
    bool all = false;
};

// (rd_mu held by the caller) snapshot + colour kernel on rstream; returns with the image columns of this tick coloured in
// d_img as far as the GPU queue is concerned.  The state lock is released before anything is copied to the host.
int display_enqueue(jsg_engine* e, float min_color, float max_color, bool tile_mode, int max_cols, Tick* t, int* need_full) {
    hipError_t herr = hipSetDevice(e->device);
    if (herr != hipSuccess) return e->fail_hip(herr, "hipSetDevice");
    if (!tile_mode && (e->W != e->img_w || e->H != e->img_h)) {   // Spectrogram.cpp:595-605; geometry is stable under rd_mu,
        JSG_HIP(e, hipStreamSynchronize(e->rstream));             // and the image belongs to the readers: no state lock here
        if (e->d_img) (void)hipFree(e->d_img);
        e->d_img = nullptr;
        e->img_pitch = round_up(e->W, 32);
        JSG_HIP(e, hipMalloc(reinterpret_cast<void**>(&e->d_img), size_t(e->img_pitch) * e->H * 4));
        e->img_w = e->W;
        e->img_h = e->H;
        e->recompute_all = true;
    }
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->mix == JSG_MIX_PER_CHANNEL) return e->fail(JSG_ERR_UNSUPPORTED, "display needs a mixed (single) spectrogram");
    const int W = e->W, H = e->H;
    t->W = W;
    t->H = H;
    const long long nec = e->new_entry;                   // getMem, Spectrogram.cpp:607-608
    if (tile_mode) {
        if (e->recompute_all || W != e->img_w || H != e->img_h || nec > W || nec > max_cols) {
            t->nec = std::min<long long>(nec, 2000000000ll);
            t->pos = e->mem_counter;
            *need_full = 1;
            return JSG_OK;
        }
    }
    e->new_entry = 0;
    t->nec = nec;
    t->pos = e->mem_counter;
    if (nec > W) e->recompute_all = true;                 // :610-613
    t->all = !tile_mode && e->recompute_all;
    if (t->all) {                                         // :623-657
        e->recompute_all = false;
        t->col_first = 0;
        t->n_cols = W;
    } else {                                              // :658-724 (only the new columns)
        t->n_cols = int(std::min<long long>(nec, W));
        t->col_first = ((t->pos - t->n_cols) % W + W) % W;
    }
    if (t->n_cols > 0) {
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
        a.col_first = t->col_first;
        a.n_cols = t->n_cols;
        a.x_first = t->col_first;
        int rc = reader_begin(e);
        if (rc != JSG_OK) return rc;
        rc = jsg_colormap_launch(&a, e->rstream);
        if (rc != JSG_OK) return e->fail_tls(rc);
        rc = reader_end(e);                               // the ring is free again once the colour kernel has run
        if (rc != JSG_OK) return rc;
    }
    return JSG_OK;
}

// copy image columns [first, first+count) (ring order, no wrap) to host x = x0.. of `argb`
hipError_t copy_cols(jsg_engine* e, uint32_t* argb, int64_t pitch, int x0, int first, int count) {
    if (count <= 0) return hipSuccess;
    return hipMemcpy2DAsync(argb + x0, size_t(pitch) * 4, e->d_img + first, size_t(e->img_pitch) * 4, size_t(count) * 4, size_t(e->img_h),
                            hipMemcpyDeviceToHost, e->rstream);
}

}  // namespace

int jsg_display_update(jsg_engine* e, float min_color, float max_color, uint32_t* argb, int64_t pitch, int* new_vals,
                       int* pos_out) {
    if (!e) return jsg_fail(JSG_ERR_INVALID, "null engine");
    flush_queue(e);
    std::lock_guard<std::mutex> rlk(e->rd_mu);
    if (!argb || pitch < e->a_W.load()) return e->fail(JSG_ERR_INVALID, "bad image buffer");
    Tick t;
    int dummy = 0;
    int rc = display_enqueue(e, min_color, max_color, false, 0, &t, &dummy);
    if (rc != JSG_OK) return rc;
    const int W = t.W, H = t.H, pos = t.pos;
    if (pitch < W) return e->fail(JSG_ERR_INVALID, "bad image buffer");
    if (e->running) {
        // x = (col + W - pos) mod W  (Spectrogram.cpp:626-631): columns [pos,W) first, then [0,pos).  Every pixel moves
        // with every new column, so the whole image is copied (hosts that scroll their own image: jsg_display_update_tile).
        JSG_HIP(e, copy_cols(e, argb, pitch, 0, pos, W - pos));
        JSG_HIP(e, copy_cols(e, argb, pitch, W - pos, 0, pos));
        JSG_HIP(e, hipStreamSynchronize(e->rstream));
        e->host_fixed_valid = false;
    } else {
        // fixed display: x = column.  When the caller's image still holds the previous tick, only the columns that changed
        // are copied: the new ones and those under the previous cursor.
        const bool incremental = !t.all && e->host_fixed_valid && e->host_img == argb && e->host_pitch == pitch;
        if (!incremental) {
            JSG_HIP(e, copy_cols(e, argb, pitch, 0, 0, W));
        } else {
            auto copy_range = [&](int first, int count) -> hipError_t {   // ring columns [first, first+count) with wrap
                if (count <= 0) return hipSuccess;
                if (count >= W) return copy_cols(e, argb, pitch, 0, 0, W);
                first = (first % W + W) % W;
                const int n1 = std::min(count, W - first);
                hipError_t r = copy_cols(e, argb, pitch, first, first, n1);
                if (r == hipSuccess && count > n1) r = copy_cols(e, argb, pitch, 0, 0, count - n1);
                return r;
            };
            JSG_HIP(e, copy_range(t.col_first, t.n_cols));
            JSG_HIP(e, copy_range(e->host_cursor_pos, e->host_cursor_w));
        }
        JSG_HIP(e, hipStreamSynchronize(e->rstream));
        // red cursor (Spectrogram.cpp:650-656 one column after a full recolour, :703-720 otherwise)
        int drawwidth = 1;
        if (!t.all) {
            if (H < 2048) drawwidth++;
            if (H < 1024) drawwidth += 2;
        }
        for (int dd = 0; dd < drawwidth; ++dd) {
            int drawpos = pos + dd;
            if (drawpos == W) drawpos -= W;
            if (drawpos >= W) continue;   // the reference would write outside the image here
            for (int y = 0; y < H; ++y) argb[size_t(y) * pitch + drawpos] = kJuceRed;
        }
        e->host_img = argb;
        e->host_pitch = pitch;
        e->host_fixed_valid = true;
        e->host_cursor_pos = pos;
        e->host_cursor_w = drawwidth;
    }
    if (new_vals) *new_vals = int(t.nec);
    if (pos_out) *pos_out = pos;
    return JSG_OK;
}

int jsg_display_update_tile(jsg_engine* e, float min_color, float max_color, uint32_t* tile, int64_t tile_pitch,
                            int max_cols, int* new_vals, int* pos_out) {
    if (!e) return jsg_fail(JSG_ERR_INVALID, "null engine");
    flush_queue(e);
    std::lock_guard<std::mutex> rlk(e->rd_mu);
    if (!tile || max_cols <= 0 || tile_pitch < max_cols) return e->fail(JSG_ERR_INVALID, "bad tile buffer");
    Tick t;
    int need_full = 0;
    int rc = display_enqueue(e, min_color, max_color, true, max_cols, &t, &need_full);
    if (rc != JSG_OK) return rc;
    if (new_vals) *new_vals = int(t.nec);
    if (pos_out) *pos_out = t.pos;
    if (need_full) return 1;   // the caller needs the whole image: jsg_display_update
    const int nv = t.n_cols, W = t.W;
    if (nv > 0) {
        const int first = t.col_first;
        const int n1 = std::min(nv, W - first);   // columns before the ring wraps
        JSG_HIP(e, copy_cols(e, tile, tile_pitch, 0, first, n1));
        JSG_HIP(e, copy_cols(e, tile, tile_pitch, n1, 0, nv - n1));
        JSG_HIP(e, hipStreamSynchronize(e->rstream));
    }
    e->host_fixed_valid = false;
    return JSG_OK;
}

}  // extern "C"
