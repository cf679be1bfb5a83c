/*
 * jsg.h -- C-ABI of the MI355X-native (gfx950) STFT spectrogram engine.
 *
 * This is the drop-in boundary for the hot path of JoergBitzer/JadeSpectrogram
 * (sliding-window FFT -> |X|^2 -> channel mix -> 10*log10 -> colour-map LUT -> ARGB).
 * Host code (C++/JUCE: jadespectrogram_amd/host/Spectrogram.h, or any FFI) calls HIP
 * only through these entry points: plain pointers and sizes, no C++/torch types.
 *
 * Every entry point cites the reference interface it replaces (paths are relative to the
 * reference tree, i.e. JadeSpectrogram/<file>:<line>).
 *
 * Conventions
 *   - return value: JSG_OK (0) or a negative jsg_status, except where the reference returns a
 *     count (jsg_get_mem: number of new columns, -1 on size mismatch, like Spectrogram.cpp:297-298).
 *   - nothing throws across this boundary; jsg_last_error() gives the text of the last failure.
 *   - an engine is bound to the HIP device that is current when it is created; one engine per GPU,
 *     one process per GPU for multi-GPU use (channels/streams are sharded, no collective needed).
 *   - threading: one producer thread (jsg_process_block) and one consumer thread (get_mem / display_* / setters).
 *     jsg_process_block is wait-free (a lock-free ring of page-locked memory; a worker thread of the engine makes the HIP
 *     calls); readers take their snapshot of the ring on a second HIP stream and hold the state lock only while they
 *     enqueue; what follows them waits on the GPU.  Setters quiesce the engine (the reference's m_protect around
 *     setFFTSize, Spectrogram.cpp:162).
 *   - there is NO CPU fallback: every compute entry point fails with JSG_ERR_HIP / JSG_ERR_NO_DEVICE
 *     when no gfx950 device is usable.
 */
#ifndef JSG_H_
#define JSG_H_

#include <stddef.h>
#include <stdint.h>

/* libjsg.so is built with -fvisibility=hidden: only the entry points declared here are exported. */
#if defined(__GNUC__)
#define JSG_API __attribute__((visibility("default")))
#else
#define JSG_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define JSG_ABI_VERSION 6   /* 3: + kernel-name query, launch pool, image scratch query, sharded set; 4: + strided image batches (round 3);
                               5: + strided dB batches, exact-log mode, producer ring statistics (round 4);
                               6: + lossless producer call, all-or-nothing sharded push, exact-log display path, tail plane, pair plan (round 5) */

typedef enum jsg_status {
    JSG_OK = 0,
    JSG_ERR_SIZE_MISMATCH = -1, /* the reference's "-1" (Spectrogram.cpp:297-298) */
    JSG_ERR_INVALID = -2,       /* bad argument */
    JSG_ERR_UNSUPPORTED = -3,   /* e.g. FFT size outside 512..8192 or not a power of two */
    JSG_ERR_HIP = -4,           /* a HIP call failed; see jsg_last_error */
    JSG_ERR_NO_DEVICE = -5,
    JSG_ERR_NOMEM = -6
} jsg_status;

/* Spectrogram::ChannelMixMode, Spectrogram.h:84-91 (same order). PER_CHANNEL is an extension:
 * no mix, one spectrogram per channel (what channel-sharded multi-GPU runs produce). */
typedef enum jsg_mix_mode {
    JSG_MIX_ABSMEAN = 0, JSG_MIX_MAX = 1, JSG_MIX_MIN = 2, JSG_MIX_LEFT = 3, JSG_MIX_RIGHT = 4,
    JSG_MIX_PER_CHANNEL = 100,
    JSG_MIX_SUM = 101           /* extension: plain sum over the local channels, no divide (partial of a
                                   cross-GPU AbsMean; finished by jsg_db_from_power_launch after the reduce) */
} jsg_mix_mode;

/* Spectrogram::Windows, Spectrogram.h:92-100 (same order; the GUI casts combo indices to it,
 * Spectrogram.cpp:409). */
typedef enum jsg_window {
    JSG_WIN_RECT = 0, JSG_WIN_HANN = 1, JSG_WIN_HAMMING = 2, JSG_WIN_BLACKMANHARRIS = 3,
    JSG_WIN_FLATTOP = 4, JSG_WIN_HANNPOISSON = 5
} jsg_window;

/* Spectrogram::FeedPercentage, Spectrogram.h:101-107. */
typedef enum jsg_feed { JSG_FEED_100 = 0, JSG_FEED_50 = 1, JSG_FEED_25 = 2, JSG_FEED_10 = 3 } jsg_feed;

/* CColorPalette scheme ids, CColorpalette.h:9-18. */
typedef enum jsg_colorscheme {
    JSG_CM_MONO = 0, JSG_CM_BW = 1, JSG_CM_HOT = 2, JSG_CM_RAINBOW = 3, JSG_CM_VIRIDIS = 4,
    JSG_CM_PLASMA = 5, JSG_CM_JADE = 6
} jsg_colorscheme;

JSG_API int jsg_abi_version(void);
/* Number of usable gfx950 devices (0 when there is none; never initialises a context). */
JSG_API int jsg_device_count(void);

/* ------------------------------------------------------------------------------------------------
 * 1. Host-side precompute that feeds the kernels (pure integer / double arithmetic, no GPU).
 * ------------------------------------------------------------------------------------------------ */

/* m_feed_samples = int(m_feed_percent*0.01*m_fftsize+0.5)           -- Spectrogram.cpp:216 */
JSG_API int jsg_feed_samples(float feed_percent, int fftsize);
/* m_memsize_blocks = int(m_memsize_s*m_fs/m_feed_samples + 0.5)     -- Spectrogram.cpp:217 */
JSG_API int jsg_memsize_blocks(float memsize_s, float fs, int feed_samples);
/* Spectrogram::getnextpowerof2                                       -- Spectrogram.cpp:171-176 */
JSG_API int jsg_next_power_of_2(float fftsize_ms, float fs);
/* Spectrogram::setWindowFkt: n RMS-normalised window samples        -- Spectrogram.cpp:239-293 */
JSG_API int jsg_window_build(int window, int n, float* out);
/* CColorPalette::ComputeColors: n_colors ints 0x00RRGGBB             -- CColorpalette.cpp:106-339 */
JSG_API int jsg_colormap_build(int n_colors, int scheme, int32_t* lut_out);
/* CColorPalette::setValueRange: resolves (lo,hi) -> m_Min, m_Max, m_AccessMult -- CColorpalette.cpp:39-54 */
JSG_API int jsg_colormap_range(int n_colors, float lo, float hi, float* vmin, float* vmax, float* access_mult);

/* ------------------------------------------------------------------------------------------------
 * 2. Stateless device operations.  Every data pointer is a DEVICE pointer on the current device;
 *    `stream` is a hipStream_t (NULL = default stream).  Asynchronous: they only enqueue.
 * ------------------------------------------------------------------------------------------------ */

/* FFT plan: twiddle / window tables resident in HBM for one (fft size, window).  Replaces the
 * `spectrum m_fft` member + m_window (Spectrogram.h:157-159; Spectrogram.cpp:215, :239-293).
 * `window` = n host floats; it is multiplied by sqrt(power_scale) when the tables are built. */
typedef struct jsg_plan jsg_plan;
JSG_API int jsg_plan_create(jsg_plan** out, int n, const float* window, float power_scale);
JSG_API int jsg_plan_destroy(jsg_plan* plan);
JSG_API int jsg_plan_fft_size(const jsg_plan* plan);

/* One launch of the fused kernel: framing + window + real FFT + |X|^2 + channel mix + 10*log10
 * + ring store.  Replaces the body of Spectrogram::processSynchronBlock's frame loop
 * (Spectrogram.cpp:50-119) incl. computePowerSpectrum (:137-145) and spectrum::power (:144).
 *
 * Frame j (j = first_frame .. first_frame+n_frames-1) of channel c reads n samples starting at
 *     in[c*in_pitch + (j / feedblocks)*n + (j % feedblocks)*hop]
 * (for a regular hop this is j*hop; the split reproduces the reference's irregular `perc10` hop).
 * The caller provides the stream WITH the reference's n leading zeros if it wants the reference's
 * time line; the engine API below does that itself.
 * Column i of this launch is written to ring column (ring_pos + i) % ring_width:
 *     out_db[col*out_pitch + bin],  bin = 0..n/2          (per-channel mode: + c*out_channel_pitch)
 */
typedef struct jsg_stft_args {
    const float* in;
    int64_t in_pitch;        /* floats between channel rows */
    int32_t channels;
    int32_t hop;
    int32_t feedblocks;
    int32_t mix_mode;        /* jsg_mix_mode */
    int64_t first_frame;
    int64_t n_frames;
    float* out_db;
    int64_t out_pitch;       /* floats between ring columns (>= n/2+1) */
    int64_t out_channel_pitch;
    int32_t ring_width;
    int32_t ring_pos;
    int32_t linear_out;      /* 0: dB = 10*log10(p + 1e-11f) (the reference's column); 1: mixed linear power p */
    int32_t blocks_per_cu;   /* 0: default (up to 8 workgroups per CU -- 16 or 32 in strided multi-batch launches of the small-workgroup plans --
                                the rest of the frames is looped over); smaller values make
                                fewer, longer-lived workgroups that prefetch their next frame -- better when several
                                launches run concurrently, worse for one launch alone */
    int64_t in_samples;      /* floats of every channel row that may be read; the launch is refused (JSG_ERR_INVALID) when
                                its last frame would read past them.  0: unknown, not checked */
    int32_t plan_select;     /* 1024 points (round 6): 0 automatic -- the two-stage kernel "Cfg1024B" (split-radix 16 x 32, four frames
                                per wavefront, one 8-wave workgroup of 32 columns per CU) where >= 4 channels are mixed into one column
                                (AbsMean / Sum) and the launch fills its rounds, the three-stage kernel "Cfg1024" otherwise; 1: always
                                "Cfg1024" (the engine pins this); 2: "Cfg1024B" wherever it exists (float columns, not Max / Min, not the
                                display launches).  The two agree inside the float32 bound, not bit for bit.
                                2048 / 4096 points have two kernels each.  0: automatic -- the large-workgroup "B" kernel (one 8-wave
                                workgroup of 16 / 8 columns per CU; faster when it can fill the GPU) for launches that fill their
                                rounds of <CU count> workgroups to at least 87 % (e.g. 3584..4096 columns of 2048 points or
                                1784..2048 of 4096 points on 256 CUs, or any launch of 7 rounds and more) AND, at 2048 points, mix
                                >= 2 channels per column (at 4096 points "B" is ahead at every channel count); the small-workgroup
                                kernel otherwise.  1: always the small-workgroup kernel, 2: always "B".  The two round differently in
                                the last bits (both inside the parity bound): callers that cut one stream into launches of
                                very different sizes and need bit-identical columns pin one of them (the engine pins 1).
                                jsg_stft_kernel_name() tells which one a launch takes.  3 (2048 points, round 5): the PAIR plan "Cfg2048P"
                                where it applies -- AbsMean / Sum over an even channel count, dB / power launches: two channels as one
                                complex transform z = x_c + i x_(c+1), sum |X_c|^2 = (|Z[k]|^2 + |Z[N-k]|^2) / 2 (a reassociation of the
                                mix of Spectrogram.cpp:64-76 inside the parity bound); as 0 where it does not.  Never chosen
                                automatically: on an MI355X it is slower than "B" (DESIGN.md section 6).  Other sizes: ignored */
    int32_t exact_log;       /* 0: 10*log10 on the hardware log unit (1 ulp, not specified bit for bit); 1: by the library's own float32
                                routine (jadespectrogram_amd/csrc/jsg_exact_math.h: exponent + degree-9 polynomial, within 2 ulp of the
                                reference's double log10): every dB value -- and so every palette index and ARGB pixel -- is then
                                reproducible bit for bit on a CPU (oracle/jsg_mirror.c does).  A separate instantiation of the kernel
                                whose epilogue calls the routine (16 instead of 3 vector instructions per value): every launch form takes
                                it -- jsg_stft_db_launch, _strided, _batches, jsg_stft_image_launch(_strided) in both of its forms, the
                                engine via jsg_set_exact_log.  (round 4: a second elementwise pass, dB launches only) */
    int32_t reserved0;       /* 0 */
    float* out_tail;         /* NULL: the reference's column layout, bin n/2 at out_db[col*out_pitch + n/2].  Otherwise: a dense plane of
                                one float per ring column; bin n/2 of ring column col is written to out_tail[r*ring_width + col] (r = 0, or
                                the channel in per-channel mode, or batch * planes + channel in a strided launch: rows x ring_width floats)
                                and NOT into the column, whose pitch may then be n/2 floats (>= n/2): a column is exactly n/2 * 4 bytes of
                                whole 128-byte lines and the 4-byte piece that opened one more line per column is gone (N = 1024: 16
                                lines instead of 16 + 1/32).  Values are bit-identical to the reference layout.  dB launches only
                                (jsg_stft_db_launch, _strided, _batches); must be NULL for the image launches */
} jsg_stft_args;
JSG_API int jsg_stft_db_launch(const jsg_plan* plan, const jsg_stft_args* args, void* stream);
/* The kernel configuration jsg_stft_db_launch picks for these arguments on the current device, as text ("Cfg1024", "Cfg2048",
 * "Cfg2048B", "Cfg4096B", ...; out_len >= 24): lets a benchmark or a test name -- and pin -- the kernel it times or checks. */
JSG_API int jsg_stft_kernel_name(const jsg_plan* plan, const jsg_stft_args* args, char* out, int out_len);

/* `count` independent launches issued from one call, launch i on streams[i % n_streams] (hipStream_t handles; NULL or
 * n_streams == 0: the default stream).  For batches that do not depend on each other (distinct input and output
 * buffers): takes the per-launch FFI cost out of the caller's loop and, with more than one stream, lets the tail of
 * one launch overlap the ramp-up of the next.  Ordering between the streams is the caller's business. */
JSG_API int jsg_stft_db_launch_many(const jsg_plan* plan, const jsg_stft_args* args, int count, void* const* streams, int n_streams);
/* The same issued by n_threads host threads (stream k belongs to thread k % n_threads, so the order inside a stream
 * is kept): for callers whose single issuing thread (~3.5 us per launch) is slower than the GPU. */
JSG_API int jsg_stft_db_launch_many_threads(const jsg_plan* plan, const jsg_stft_args* args, int count, void* const* streams,
                                    int n_streams, int n_threads);

/* Independent launches that CANNOT be laid out at a stride (otherwise: jsg_stft_db_launch_strided below, one kernel launch for all of
 * them): `count` launches (no two of them write the same ring columns), stream-ordered with respect to `stream` like one launch -- they
 * start after everything that was enqueued on `stream` before the call, and work enqueued on `stream` after the call sees all their
 * results -- but spread over FOUR working streams: the caller's own plus three that the library owns (created once per device, in
 * jsg_plan_create), issued by two host threads, so that the ramp-up and drain of consecutive launches overlap.  Two working streams for
 * the one-workgroup-per-CU kernels of 2048 / 4096 points.  The caller neither creates streams nor knows the good stream count; it must
 * not enqueue on `stream` from another thread during the call.
 * Hardware queues: the GPU's compute front end serves four hardware queues at a time.  A FIFTH busy queue makes the command processor
 * time-slice them (measured at C2: 1.08e9 frames/s with four busy queues, 0.35e9 with five) -- which is why the caller's stream is one of
 * the four working streams and not a fifth beside them -- and two busy streams that the HIP runtime maps onto ONE hardware queue
 * serialise.  The runtime multiplexes all streams of the process onto GPU_MAX_HW_QUEUES queues (default 4) in creation order (measured
 * with nine streams in the process: 0.79 / 0.41 / 0.84 / 1.08 / 1.08e9 frames/s at 4 / 6 / 8 / 12 / 16 queues): a host that wants this
 * entry point at its full rate exports GPU_MAX_HW_QUEUES=16 itself before its first HIP call.  The library does not touch the
 * environment (rounds 2-3 set the variable from a constructor: not thread-safe inside a multi-threaded host, and it changed every
 * other HIP user of the process).  Calls for one device are serialised on the host.  Inside a stream capture (hipGraph) the launches are
 * issued by the calling thread and become parallel branches of the graph; every forked stream is joined even when a launch fails. */
JSG_API int jsg_stft_db_launch_batches(const jsg_plan* plan, const jsg_stft_args* args, int count, void* stream);

/* `n_batches` independent batches of ONE geometry in ONE kernel launch on ONE stream (the frame loop of
 * Spectrogram::processSynchronBlock, Spectrogram.cpp:50-119, over K streams' worth of blocks): `args` describes batch 0; batch b reads
 * args->in + b * in_batch_stride (floats; 0 = the same input) and writes its own ring at args->out_db + b * out_batch_stride (floats,
 * at least one ring: the rings must not overlap).  Everything else in `args` (frame count, hop, ring_pos, mix, in_samples -- which then
 * holds for the rows of EVERY batch) is the same for all batches.  The workgroups of the one launch walk through the columns of all
 * batches: lane tables loaded once per workgroup instead of once per step of four to sixteen frames, no ramp-up and drain per batch, the next columns in
 * flight while the current ones are transformed -- the rate of back-to-back launches without extra streams, hardware queues or issuing
 * threads (bench.py's default C2 step).  args->blocks_per_cu: workgroups per CU of the grid (0 = the library's choice).
 *   2048 / 4096 points: the plan rule of plan_select looks at the frames of the WHOLE launch; columns are those of single launches
 *   with plan_select pinned to that plan.  All other sizes: bit-identical to single launches.
 * Max / Min mixes have no strided kernel: they are launched batch by batch in stream order.  More than 2^20 workgroup steps go out as
 * several launches.  jsg_stft_db_strided_kernel_name tells the kernel (as jsg_stft_kernel_name, for the total size). */
JSG_API int jsg_stft_db_launch_strided(const jsg_plan* plan, const jsg_stft_args* args, int n_batches, int64_t in_batch_stride,
                               int64_t out_batch_stride, void* stream);
JSG_API int jsg_stft_db_strided_kernel_name(const jsg_plan* plan, const jsg_stft_args* args, int n_batches, int64_t in_batch_stride, char* out,
                                    int out_len);

/* out[i] = 10*log10(power[i]/divisor + 1e-11f), i < count: the tail of the mix (reference Spectrogram.cpp:74,107)
 * for partial sums that were reduced across GPUs (JSG_MIX_SUM).  In place (out == power) is allowed. */
JSG_API int jsg_db_from_power_launch(const float* power, float* out, int64_t count, float divisor, void* stream);
/* ... with the logarithm of jsg_stft_args.exact_log (1) instead of the hardware unit (0) */
JSG_API int jsg_db_from_power_launch_ex(const float* power, float* out, int64_t count, float divisor, int exact_log, void* stream);

/* The reference's dense column shape out of the tail-plane layout (jsg_stft_args.out_tail): dst[col*dst_pitch + bin] for bin < height - 1 from
 * db[col*db_pitch + bin], and dst[col*dst_pitch + height - 1] from tail[col], col < n_columns (one ring / one row of the plane; height = n/2 + 1).
 * What Spectrogram::getMem hands out (m_mem[col][bin], Spectrogram.h:144, Spectrogram.cpp:295-331) for callers that computed in whole-line
 * columns.  Device pointers; db_pitch >= height - 1, dst_pitch >= height. */
JSG_API int jsg_columns_from_tail_layout_launch(const float* db, int64_t db_pitch, const float* tail, int n_columns, int height, float* dst,
                                        int64_t dst_pitch, void* stream);

/* Roofline calibration (no reference counterpart): a tuned float4 streaming copy of `bytes` bytes (multiple of 16, 16-byte aligned
 * device pointers), non-temporal loads and stores, one thread per 16 bytes.  bench.py times it on buffers that rotate over > 1 GB to
 * report what the HBM of THIS box gives a balanced read + write stream (`peak_copy_GBps`), the yardstick beside the 8 TB/s spec. */
JSG_API int jsg_calib_copy_launch(const void* src, void* dst, int64_t bytes, void* stream);

/* Colour loop: dB ring columns -> ARGB pixels (and/or 8-bit palette indices).
 * Replaces the pixel loops of SpectrogramComponent::timerCallback (Spectrogram.cpp:632-648,
 * :673-680, :693-700) with CColorPalette::getRGBColor inlined (CColorpalette.h:32-47):
 *     pixel(x, height-1-bin) = lut[index(db[col][bin])] | 0xFF000000
 * for i in [0,n_cols): col = (col_first+i) % ring_width, x = (x_first+i) % x_wrap.
 * Image layout is row-major [height][img_pitch] (what juce::Image::BitmapData exposes). */
typedef struct jsg_colormap_args {
    const float* db;
    int64_t db_pitch;
    int32_t ring_width;
    int32_t height;          /* bins = n/2+1 */
    int32_t col_first;
    int32_t n_cols;
    int32_t x_first;
    int32_t x_wrap;          /* image width */
    const int32_t* lut;      /* device, n_colors entries 0x00RRGGBB */
    int32_t n_colors;
    float vmin, vmax, access_mult;   /* from jsg_colormap_range */
    uint32_t* argb_out;      /* may be NULL */
    int64_t argb_pitch;      /* pixels between image rows */
    uint8_t* index_out;      /* may be NULL; requires n_colors <= 256 */
    int64_t index_pitch;
} jsg_colormap_args;
JSG_API int jsg_colormap_launch(const jsg_colormap_args* args, void* stream);

/* Fused display path: STFT -> palette index -> ARGB without the dB column ever going to memory (reference
 * Spectrogram.cpp:632-648: the colour loop consumes the column the engine has just produced).  The image is bit-identical to
 * jsg_stft_db_launch (same plan_select) followed by jsg_colormap_launch.
 *   ONE kernel where a workgroup of the plan holds eight whole columns -- 1024 points, and 4096 points when the launch takes the
 *   one-wavefront-per-frame kernel (automatic rule of jsg_stft_args.plan_select, or plan_select = 2; e.g. a 10-second stereo image
 *   at 96 kHz = 1875 columns): the workgroup parks the palette indices (CColorPalette::getRGBColor's index, CColorpalette.h:34-45)
 *   of its columns in LDS and writes the ARGB rows itself; only the input is read, only the image is written, `index_scratch`
 *   is not touched and may be NULL.  Needs colour.argb_out, no colour.index_out, n_colors <= 256, n_cols <= x_wrap.
 *   TWO kernels otherwise: the STFT kernel writes 1 byte per bin into `index_scratch`, the colour kernel reads those bytes.
 * jsg_stft_image_needs_scratch() tells which (1: index_scratch is required, 0: it is not used).
 * stft.out_db may be NULL (it is not written); colour.db is ignored; colour must cover exactly the columns of the launch
 * (n_cols == n_frames, col_first == ring_pos, ring_width equal, height n/2+1); n_colors <= 256; mixes: AbsMean / Sum / Left / Right. */
typedef struct jsg_stft_image_args {
    jsg_stft_args stft;
    jsg_colormap_args colour;
    uint8_t* index_scratch;        /* device: ring_width columns of index_scratch_pitch bytes each (two-kernel form only) */
    int64_t index_scratch_pitch;   /* >= n/2+1; a multiple of 64 keeps the columns line-aligned */
} jsg_stft_image_args;
JSG_API int jsg_stft_image_launch(const jsg_plan* plan, const jsg_stft_image_args* args, void* stream);
JSG_API int jsg_stft_image_needs_scratch(const jsg_plan* plan, const jsg_stft_image_args* args);
/* `n_images` images of ONE geometry from one call (a batch of independent streams, or the pages of a long recording): `args`
 * describes one image; image i reads args->stft.in + i * in_image_stride (floats, >= 0) and writes args->colour.argb_out +
 * i * argb_image_stride (pixels, >= height * argb_pitch).  Where the single-kernel form applies to a launch of the TOTAL size
 * (1024 points; 4096 points when all images together fill the one-workgroup-per-CU kernel's rounds, or plan_select = 2) the images
 * share ONE kernel launch whose workgroups walk through the columns of all of them: tables loaded once, the next columns in flight
 * while the current ones are transformed, no idle workgroup slots at the end of every image (C5, 1875 columns = 235 groups of
 * eight on 256 CUs: see DESIGN.md 4.4 for the measured gain).  Pixels: those of n_images jsg_stft_image_launch calls with plan_select pinned to the
 * plan the whole launch takes (the plan rule looks at the total column count).  Otherwise: n_images launches in stream order,
 * which need `index_scratch` like a single one (jsg_stft_image_strided_needs_scratch tells).  colour.index_out must be NULL. */
JSG_API int jsg_stft_image_launch_strided(const jsg_plan* plan, const jsg_stft_image_args* args, int n_images, int64_t in_image_stride,
                                  int64_t argb_image_stride, void* stream);
JSG_API int jsg_stft_image_strided_needs_scratch(const jsg_plan* plan, const jsg_stft_image_args* args, int n_images);

/* ------------------------------------------------------------------------------------------------
 * 3. Engine: the state of class Spectrogram (Spectrogram.h:81-169) living on the GPU.
 * ------------------------------------------------------------------------------------------------ */
typedef struct jsg_engine jsg_engine;

/* Spectrogram::Spectrogram() defaults (Spectrogram.cpp:16-24): fs 48000, n 1024, feed 100 %, 1 s memory,
 * Hann, AbsMean.  `channels` is explicit (the plugin never calls setchannels). */
JSG_API int jsg_create(jsg_engine** out, int channels);
/* The same on an explicit HIP device (0 .. jsg_device_count()-1) instead of the calling thread's current one: what a
 * single-process host that drives several GPUs uses, one engine per device ("Several GPUs"). */
JSG_API int jsg_create_on_device(jsg_engine** out, int channels, int device);
JSG_API int jsg_get_device(const jsg_engine* e);
/* One engine per entry of `devices` (entries may repeat), the `channels` channels of one stream dealt out in contiguous runs
 * whose sizes differ by at most one: entry i owns [first_channel[i], first_channel[i] + channel_count[i]) and gets no engine
 * (out[i] = NULL) when that run is empty.  The engines share nothing -- the path shards by independent channels, there is no
 * collective (a cross-GPU AbsMean is the one exchange: INTEGRATION.md C).  Configure every engine with the usual setters.
 * jsg_process_block_sharded hands every engine its run of the planar pointers (enqueue only, wait-free) -- ALL OR NOTHING: if any
 * engine would not take the block now (its queue is full, it is in a geometry change, or its worker met an error) NO engine gets it,
 * every engine counts one dropped block and 1 is returned, so the rings of the shards keep the same position.  (A geometry setter of one
 * engine racing with the call is the one case that can still leave the set uneven: the caller changes the geometry of ALL engines and
 * every setter wipes the history, so it resynchronises there.)  jsg_destroy_sharded frees the set. */
JSG_API int jsg_create_sharded(jsg_engine** out, int* first_channel, int* channel_count, const int* devices, int n_devices, int channels);
JSG_API int jsg_process_block_sharded(jsg_engine* const* engines, const int* first_channel, int n_devices, const float* const* planar);
JSG_API int jsg_destroy_sharded(jsg_engine** engines, int n_devices);
JSG_API int jsg_destroy(jsg_engine* e);
JSG_API const char* jsg_last_error(const jsg_engine* e);   /* e may be NULL: last error of the calling thread */

/* setters: each rebuilds the memory like Spectrogram::buildmem (Spectrogram.cpp:213-238) */
JSG_API int jsg_set_samplerate(jsg_engine* e, float fs);                 /* Spectrogram.cpp:148-152 */
JSG_API int jsg_set_channels(jsg_engine* e, int channels);               /* :153-157 */
JSG_API int jsg_set_fft_size(jsg_engine* e, int n);                      /* :160-170 */
JSG_API int jsg_set_closest_fft_size_ms(jsg_engine* e, float ms);        /* :177-183 */
JSG_API int jsg_set_memory_time_s(jsg_engine* e, float seconds);         /* :184-188 */
JSG_API int jsg_set_feed_percent(jsg_engine* e, int feed);               /* :189-211, jsg_feed */
JSG_API int jsg_set_feed_percent_ext(jsg_engine* e, float percent);      /* extension: any overlap, e.g. 12.5 */
JSG_API int jsg_set_pause_mode(jsg_engine* e, int paused);               /* Spectrogram.h:122 */
JSG_API int jsg_set_window(jsg_engine* e, int window);                   /* Spectrogram.h:123 */
JSG_API int jsg_set_window_table(jsg_engine* e, const float* w, int n);  /* extension: caller-supplied window */
JSG_API int jsg_set_mix_mode(jsg_engine* e, int mode);                   /* m_mode has no setter in the reference (:21) */
JSG_API int jsg_set_power_scale(jsg_engine* e, float scale);             /* normalisation of spectrum::power, default 1 */
JSG_API int jsg_set_exact_log(jsg_engine* e, int on);                    /* extension: jsg_stft_args.exact_log for the engine's launches (default 0);
                                                                    keeps the ring (no buildmem) */

JSG_API int jsg_get_spectrum_size(const jsg_engine* e);                  /* Spectrogram.h:127 */
JSG_API int jsg_get_memory_size(const jsg_engine* e);                    /* Spectrogram.h:128 */
JSG_API float jsg_get_samplerate(const jsg_engine* e);                   /* Spectrogram.h:130 */
JSG_API int jsg_get_fft_size(const jsg_engine* e);
JSG_API int jsg_get_feed_samples(const jsg_engine* e);
JSG_API int jsg_get_feedblocks(const jsg_engine* e);
JSG_API int jsg_get_channels(const jsg_engine* e);
JSG_API int jsg_get_window(const jsg_engine* e, float* out, int n);      /* copy of m_window */

/* Spectrogram::processSynchronBlock (Spectrogram.cpp:37-135): `planar` = channels host pointers to fft-size samples each.
 * WAIT-FREE on the caller's (audio) thread: the block is copied into a page-locked single-producer ring and published with one
 * atomic store; a worker thread owned by the engine does the H2D copy and the launch.  No mutex, no HIP call, no allocation, never
 * a wait: when the ring is full (the GPU more than 64 blocks behind) or the engine is in the middle of a channel-count / FFT-size
 * change the block is DROPPED and counted.  Returns 0 (queued), 1 (dropped), < 0 (error -- also an error the worker thread met
 * earlier, text in jsg_last_error).  Readers, setters and jsg_sync see every block whose call returned before theirs began. */
JSG_API int jsg_process_block(jsg_engine* e, const float* const* planar);
/* The same for callers that state the geometry their pointers were sized for (host classes whose re-blocker runs unlocked beside
 * the FFT-size combo box): a block of another channel count or length than the engine's current one is dropped (returns 1)
 * instead of being read past its end.  0 = do not check that value. */
JSG_API int jsg_process_block_n(jsg_engine* e, const float* const* planar, int channels, int n);
/* The LOSSLESS form for callers that are not bound to real time (a DAW's offline bounce -- juce::AudioProcessor::isNonRealtime() --,
 * converters, test loops that push faster than the GPU takes blocks out): where jsg_process_block would drop a block because the ring is
 * full, this call waits (yield, then 100 us sleeps) until a slot is free, at most timeout_ms milliseconds (< 0: no limit).  The reference
 * never drops a block (Spectrogram.cpp:37-135 computes in place); this entry point keeps that property.  channels / n as in
 * jsg_process_block_n (0 = not checked).  Returns 0 (queued), 1 (dropped and counted: geometry change in progress, geometry mismatch, or
 * still full at the timeout), < 0 (error).  NOT for the audio thread of a live host. */
JSG_API int jsg_process_block_wait(jsg_engine* e, const float* const* planar, int channels, int n, int timeout_ms);
/* blocks that did not reach the ring since the engine was created: dropped by jsg_process_block(_n/_wait/_sharded), refused because of an
 * earlier worker error, or discarded by the worker */
JSG_API long long jsg_get_dropped_blocks(const jsg_engine* e);
/* The same for n_blocks consecutive blocks in one launch: samples[c*pitch + i], i < n_blocks*n. */
JSG_API int jsg_process_blocks(jsg_engine* e, const float* samples, int64_t pitch, int n_blocks);
/* The same with the samples already in HBM (device pointer, same layout); no host copy. */
JSG_API int jsg_process_blocks_device(jsg_engine* e, const float* d_samples, int64_t pitch, int n_blocks);

/* Spectrogram::getMem (Spectrogram.cpp:295-331): dst is the caller's dense [dst_columns][n/2+1]
 * buffer; copies all columns when at least a ring-full is new, else only the new ones (in place,
 * wrap-aware); returns the new-column count and zeroes it, -1 on size mismatch. */
JSG_API int jsg_get_mem(jsg_engine* e, float* dst, int dst_columns, int* pos);
/* The same into the reference's own container shape, vector<vector<float>> mem[W][H] (Spectrogram.h:144): rows[c]
 * points to the row_len = n/2+1 floats of column c (a NULL row is skipped); only the new columns are touched. */
JSG_API int jsg_get_mem_rows(jsg_engine* e, float* const* rows, int n_rows, int row_len, int* pos);
/* Extension: all columns of the ring as they are now, without consuming the new-column counter (returns the counter). */
JSG_API int jsg_peek_mem(jsg_engine* e, float* dst, int dst_columns, int* pos);
/* Device pointer / geometry of the dB ring (stays valid until the next setter). */
JSG_API int jsg_ring_device(jsg_engine* e, float** d_ring, int64_t* pitch, int* width, int* pos);
JSG_API int jsg_sync(jsg_engine* e);
JSG_API void* jsg_stream(jsg_engine* e);                                  /* the engine's hipStream_t */

/* ------------------------------------------------------------------------------------------------
 * 4. Display: the colour half of SpectrogramComponent::timerCallback (Spectrogram.cpp:590-731).
 * ------------------------------------------------------------------------------------------------ */
/* CColorPalette(n_colors, scheme) / setColorSceme (Spectrogram.cpp:337, :400); forces a full recolour */
JSG_API int jsg_display_set_colormap(jsg_engine* e, int n_colors, int scheme);
/* m_isRunningDisplay (Spectrogram.cpp:745-759) */
JSG_API int jsg_display_set_running(jsg_engine* e, int running);
/* m_recomputeAll = true (colour sliders, Spectrogram.cpp:370,379) */
JSG_API int jsg_display_invalidate(jsg_engine* e);
/* One timer tick: consumes the new columns (like getMem), colours them (all of them when a recolour is
 * pending) and writes the [height][width] ARGB image into host memory `argb` (`pitch` pixels per row).
 * min_color/max_color are the slider values (Spectrogram.cpp:614-617). */
JSG_API int jsg_display_update(jsg_engine* e, float min_color, float max_color, uint32_t* argb, int64_t pitch,
                       int* new_vals, int* pos);

/* Incremental tick for hosts that scroll their own image like the reference (moveImageSection + fill of the
 * right-most columns, Spectrogram.cpp:663-683): colours only the new columns and copies just them into `tile`
 * ([height][tile_pitch] pixels, oldest column first).  Returns 0 and *new_vals (<= max_cols) columns; returns 1
 * (nothing consumed) when a full recolour is pending or more than max_cols columns are new -- then call
 * jsg_display_update for the whole image. */
JSG_API int jsg_display_update_tile(jsg_engine* e, float min_color, float max_color, uint32_t* tile, int64_t tile_pitch,
                            int max_cols, int* new_vals, int* pos);

/* The frequency window of SpectrogramComponent::paint (Spectrogram.cpp:441-459): which image rows show
 * [min_freq, max_freq] Hz.  Pure host arithmetic (same clamps and roundings); outputs displayStartPixel,
 * displayEndPixel, heightInterval and hStart (= height - displayEndPixel, the first image row to blit). */
JSG_API int jsg_display_freq_rows(float fs, int height, float min_freq, float max_freq, int* start_pixel, int* end_pixel,
                          int* height_interval, int* h_start);

#ifdef __cplusplus
}
#endif
#endif /* JSG_H_ */
