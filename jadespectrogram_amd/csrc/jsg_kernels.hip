// libjsg.so, device side: the colour loop kernel, the plans (lane tables), the launcher and the stateless C-ABI entry points.
//
//   stft_db_kernel   (jsg_stft_kernel.h; instantiated in jsg_stft_a.hip for 512 / 1024 / 2048 / 8192 points and in jsg_stft_b.hip for
//                    4096 points -- two units because the two groups want different machine schedulers) framing + window + real FFT +
//                    |X|^2 + channel mix + dB + ring store, or palette indices, or the ARGB rows of its own eight columns.
//                    Replaces Spectrogram.cpp:50-119 + :137-145 + spectrum::power (call site :144) of the reference.
//   colormap_kernel  (here) dB ring columns -> ARGB image rows (transpose through LDS so both sides are coalesced),
//                    CColorPalette::getRGBColor inlined.  Replaces Spectrogram.cpp:632-648 / :673-680 / :693-700.
//   launcher         stft_launch_impl: which kernel a launch takes (wants_plan_b: launch fill, channel count, CU count of the device),
//                    grid and loop counts; jsg_stft_db_launch_batches: the library's launch pool (caller's stream + three).
//
// The index algebra, the twiddle tables and the LDS layouts are modelled and checked in tools/fft_model.py.
#include "jsg_stft_kernel.h"

namespace jsg {

// Development knobs (A/B switches read from the environment once per process) exist in VARIANT builds only (-DJSG_DEV_KNOBS:
// jadespectrogram_amd/_build.py build_variant, tools/README.md).  The product library never reads the environment: inside a DAW process
// an inherited variable would silently change which plan runs, and with it the last bits of the columns (jsg.h: plan_select).
#ifdef JSG_DEV_KNOBS
static int dev_knob_int(const char* name) { const char* e = getenv(name); return e ? atoi(e) : 0; }
static bool dev_knob_set(const char* name) { return getenv(name) != nullptr; }
static bool dev_knob_is(const char* name, char first) { const char* e = getenv(name); return e && e[0] == first; }
#else
static constexpr int dev_knob_int(const char*) { return 0; }
static constexpr bool dev_knob_set(const char*) { return false; }
static constexpr bool dev_knob_is(const char*, char) { return false; }
#endif

static bool b_plan_fills_its_rounds(long long n_frames, int frames_per_workgroup, int n_cu) {
    const long long want = (n_frames + frames_per_workgroup - 1) / frames_per_workgroup;
    const long long rounds = (want + n_cu - 1) / n_cu;
    return rounds > 0 && double(want) >= kB_min_round_fill * double(rounds * n_cu);
}
// compute units of the current device (256 on an MI355X in SPX mode, fewer when it is partitioned: CPX / DPX): read once per
// device from the runtime.  The "one workgroup per CU" grids and the round rule above are sized with it.
static int cu_count_of_device(int dev) {
    static std::atomic<int> cached[64];
    if (dev >= 0 && dev < 64) {
        const int c = cached[dev].load(std::memory_order_acquire);
        if (c > 0) return c;
    }
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    if (dev >= 0 && dev < 64) cached[dev].store(n, std::memory_order_release);
    return n;
}


// ------------------------------------------------------------------------------------------------------------
// colour loop
// ------------------------------------------------------------------------------------------------------------
struct CmapKArgs {
    const float* db;
    long long db_pitch;
    int ring_w, height, col_first, n_cols, x_first, x_wrap;
    const int* lut;
    int n_colors;
    float vmin, vmax, top, mult;
    unsigned* argb;
    long long argb_pitch;
    unsigned char* index;
    long long index_pitch;
    const unsigned char* idx_in;   // fused display path: palette indices per column (1 byte per bin) instead of dB
    long long idx_in_pitch;
};


constexpr int CM_TILE = 64;   // 64 columns x 64 bins per workgroup

__global__ __launch_bounds__(256) void colormap_kernel(const CmapKArgs a) {
    __shared__ int s_lut[1024];
    // [bin][col], one dword per pixel and a row stride of 65 dwords: the column-wise writes of the read phase and the row-wise reads
    // of the write phase both touch 32 different banks per half-wave (as 16-bit pairs, round 2, a third of this kernel's LDS
    // cycles were bank conflicts: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.33)
    __shared__ int s_idx[CM_TILE][CM_TILE + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool lut_in_lds = a.n_colors <= 1024;

    const int col0 = blockIdx.x * CM_TILE;   // first column (relative to col_first) of this tile
    const int bin0 = blockIdx.y * CM_TILE;
    // read phase: each wave takes 16 columns; lane = bin (256 B coalesced per column).  All 16 loads are issued
    // before the first use (one HBM round trip per tile instead of sixteen).
    const int bin = bin0 + lane;
    if (a.idx_in) {   // the STFT kernel has already quantised the column (wave-uniform branch)
        // one 16-byte load per thread: thread t takes bins 16 (t % 4) .. + 15 of column t / 4 of the tile
        const int tc = tid >> 2, tb = (tid & 3) * 16;
        const int i = col0 + tc;
        int col = a.col_first + i;
        if (col >= a.ring_w) col -= a.ring_w;
        const unsigned char* src = a.idx_in + (long long)col * a.idx_in_pitch + bin0 + tb;
        typedef unsigned cu4 __attribute__((ext_vector_type(4)));
        cu4 w = {0u, 0u, 0u, 0u};
        const bool whole = i < a.n_cols && bin0 + tb + 16 <= a.height && ((a.idx_in_pitch | reinterpret_cast<unsigned long long>(a.idx_in)) & 15) == 0;
        if (whole) w = *reinterpret_cast<const cu4*>(src);
        else if (i < a.n_cols) {   // ragged edge of the image, or an unaligned scratch: byte by byte
            unsigned char bb[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) bb[k] = bin0 + tb + k < a.height ? src[k] : (unsigned char)0;
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = bb[4 * k] | bb[4 * k + 1] << 8 | bb[4 * k + 2] << 16 | (unsigned)bb[4 * k + 3] << 24;
        }
        if (lut_in_lds)
            for (int k = tid; k < a.n_colors; k += 256) s_lut[k] = a.lut[k];
#pragma unroll
        for (int k = 0; k < 16; ++k) s_idx[tb + k][tc] = (int)((w[k >> 2] >> (8 * (k & 3))) & 0xffu);
    } else {
        float v[CM_TILE / 4];
#pragma unroll
        for (int q = 0; q < CM_TILE / 4; ++q) {
            const int i = col0 + wave + 4 * q;
            int col = a.col_first + i;               // col_first < ring_w and i < n_cols <= ring_w
            if (col >= a.ring_w) col -= a.ring_w;
            v[q] = (i < a.n_cols && bin < a.height) ? a.db[(long long)col * a.db_pitch + bin] : 0.f;
        }
        if (lut_in_lds)   // table load behind the tile loads: one latency, not two
            for (int i = tid; i < a.n_colors; i += 256) s_lut[i] = a.lut[i];
#pragma unroll
        for (int q = 0; q < CM_TILE / 4; ++q)
            s_idx[lane][wave + 4 * q] = color_index(v[q], a.vmin, a.vmax, a.top, a.mult, a.n_colors);
    }
    __syncthreads();
    // write phase: each wave takes 16 image rows; lane = column (256 B coalesced per row when x does not wrap)
    const int i = col0 + lane;
    int x = a.x_first + i;                       // x_first < x_wrap; wraps at most once when n_cols <= x_wrap
    x %= a.x_wrap;
    if (i < a.n_cols) {
#pragma unroll
        for (int q = 0; q < CM_TILE / 4; ++q) {
            const int rr = wave + 4 * q;
            const int bb = bin0 + rr;
            if (bb < a.height) {
                const int idx = s_idx[rr][lane];
                const long long y = a.height - 1 - bb;          // low frequencies at the bottom (Spectrogram.cpp:642)
                if (a.argb) {
                    const int rgb = lut_in_lds ? s_lut[idx] : a.lut[idx];
                    __builtin_nontemporal_store((unsigned)rgb | 0xFF000000u, &a.argb[y * a.argb_pitch + x]);   // Spectrogram.cpp:637
                }
                if (a.index) a.index[y * a.index_pitch + x] = (unsigned char)idx;
            }
        }
    }
}

__global__ __launch_bounds__(256) void db_from_power_kernel(const float* p, float* out,
                                                            long long count, float divisor, int exact_log) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        float v = p[i];
        if (divisor != 1.0f) v = v / divisor;
        out[i] = exact_log ? jsg_exact_db(v) : to_db(v);
    }
}

// jsg_columns_from_tail_layout_launch: the reference's dense [W][n/2+1] shape (m_mem[col][bin], Spectrogram.h:144) out of the tail-plane
// layout of jsg_stft_args.out_tail -- columns of n/2 floats + a plane of bin n/2.  One workgroup per column, one float per thread and step
// (dense destination columns of n/2+1 floats start 4 bytes further off a 16-byte boundary from column to column: no wider store fits them all).
__global__ __launch_bounds__(256) void tail_merge_kernel(const float* __restrict__ db, long long db_pitch, const float* __restrict__ tail, int half,
                                                         float* __restrict__ dst, long long dst_pitch) {
    const long long col = blockIdx.x;
    const float* src = db + col * db_pitch;
    float* out = dst + col * dst_pitch;
    for (int k = threadIdx.x; k < half; k += 256) out[k] = src[k];
    if (threadIdx.x == 0) out[half] = tail[col];
}

// roofline calibration (jsg_calib_copy_launch): the float4 streaming copy that reaches the most on an MI355X -- one thread per 16 bytes,
// non-looping grid (the dispatcher hands the workgroups out in address order: the access front stays tight), non-temporal loads and
// stores (tools/copy_roof.hip: 6.4-6.5 TB/s read + written at >= 0.5 GB, 6.3 at 134 MB; grid-stride forms 4.5-6.3)
__global__ __launch_bounds__(256) void calib_copy_kernel(const float* __restrict__ src, float* __restrict__ dst, long long n4) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) __builtin_nontemporal_store(__builtin_nontemporal_load(reinterpret_cast<const v4f*>(src) + i), reinterpret_cast<v4f*>(dst) + i);
}

}  // namespace jsg

using namespace jsg;

struct jsg_plan {
    int n = 0;
    int device = -1;
    float2* d_tab = nullptr;
    size_t tab_elems = 0;
    float2* d_tab_b = nullptr;   // 1024 / 2048 / 4096 points: lane tables of the second plan (Cfg1024B / Cfg2048B / Cfg4096B), behind d_tab in the same allocation
    float2* d_tab_p = nullptr;   // 2048 points: lane tables of the pair plan (Cfg2048P), behind those
};

extern "C" {


int jsg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static void pool_prepare_for_device(int dev);

int jsg_plan_create(jsg_plan** out, int n, const float* window, float power_scale) {
    if (!out || !window) return jsg_fail(JSG_ERR_INVALID, "jsg_plan_create: null argument");
    *out = nullptr;
    if (!(power_scale > 0.f)) return jsg_fail(JSG_ERR_INVALID, "jsg_plan_create: power_scale must be > 0");
    std::vector<float2> t;
    size_t tab_b_at = 0, tab_p_at = 0;
    const double amp = std::sqrt(double(power_scale));   // |FFT(a w x)|^2 = a^2 |FFT(w x)|^2
    switch (n) {
        case 512: fill_tables<Cfg512>(t, window, amp); break;
        case 1024: {   // the three-stage plan and, behind it, the two-stage one (Cfg1024B)
            fill_tables<Cfg1024>(t, window, amp);
            std::vector<float2> tb;
            fill_tables<Cfg1024B>(tb, window, amp);
            t.resize((t.size() + 31) / 32 * 32, make_float2(0.f, 0.f));
            tab_b_at = t.size();
            t.insert(t.end(), tb.begin(), tb.end());
            break;
        }
        case 2048: {   // both 2048-point plans (the launcher picks per launch); the second table set starts 256-byte aligned
            fill_tables<Cfg2048>(t, window, amp);
            std::vector<float2> tb;
            fill_tables<Cfg2048B>(tb, window, amp);
            t.resize((t.size() + 31) / 32 * 32, make_float2(0.f, 0.f));
            tab_b_at = t.size();
            t.insert(t.end(), tb.begin(), tb.end());
            fill_tables<Cfg2048P>(tb, window, amp);   // ... and the pair plan's
            t.resize((t.size() + 31) / 32 * 32, make_float2(0.f, 0.f));
            tab_p_at = t.size();
            t.insert(t.end(), tb.begin(), tb.end());
            break;
        }
        case 4096: {   // likewise the two 4096-point plans
            fill_tables<Cfg4096>(t, window, amp);
            std::vector<float2> tb;
            fill_tables<Cfg4096B>(tb, window, amp);
            t.resize((t.size() + 31) / 32 * 32, make_float2(0.f, 0.f));
            tab_b_at = t.size();
            t.insert(t.end(), tb.begin(), tb.end());
            break;
        }
        case 8192: fill_tables<Cfg8192>(t, window, amp); break;
        default:
            return jsg_fail(JSG_ERR_UNSUPPORTED, "jsg_plan_create: FFT size must be 512, 1024, 2048, 4096 or 8192");
    }
    jsg_plan* p = new (std::nothrow) jsg_plan();
    if (!p) return jsg_fail(JSG_ERR_NOMEM, "jsg_plan_create: out of host memory");
    p->n = n;
    p->tab_elems = t.size();
    if (hipGetDevice(&p->device) != hipSuccess) {
        delete p;
        return jsg_fail(JSG_ERR_NO_DEVICE, "jsg_plan_create: no HIP device (the engine has no CPU fallback)");
    }
    // Force the (lazily loaded) code object onto the device now, on the thread that configures, not inside the audio
    // thread's first jsg_process_block: querying any kernel of the module loads all of them (measured: 2.5 ms otherwise).
    {
        hipFuncAttributes fa;
        (void)touch_module_a();
        (void)touch_module_b();
        (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&colormap_kernel));
    }
    hipError_t err = hipSuccess;
    switch (n) {
        case 512: err = ensure_attrs_Cfg512(); break;
        case 1024: err = ensure_attrs_Cfg1024(); if (err == hipSuccess) err = ensure_attrs_Cfg1024I(); if (err == hipSuccess) err = ensure_attrs_Cfg1024B(); break;
        case 2048: err = ensure_attrs_Cfg2048(); if (err == hipSuccess) err = ensure_attrs_Cfg2048B(); if (err == hipSuccess) err = ensure_attrs_Cfg2048P(); break;
        case 4096: err = ensure_attrs_Cfg4096(); if (err == hipSuccess) err = ensure_attrs_Cfg4096B(); break;
        case 8192: err = ensure_attrs_Cfg8192(); break;
    }
    if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void**>(&p->d_tab), t.size() * sizeof(float2));
    if (err == hipSuccess) err = hipMemcpy(p->d_tab, t.data(), t.size() * sizeof(float2), hipMemcpyHostToDevice);
    if (err != hipSuccess) {
        if (p->d_tab) (void)hipFree(p->d_tab);
        delete p;
        return jsg_fail_hip(err, "jsg_plan_create");
    }
    if (tab_b_at) p->d_tab_b = p->d_tab + tab_b_at;
    if (tab_p_at) p->d_tab_p = p->d_tab + tab_p_at;
    pool_prepare_for_device(p->device);
    *out = p;
    return JSG_OK;
}

int jsg_plan_destroy(jsg_plan* plan) {
    if (!plan) return JSG_OK;
    if (plan->d_tab) (void)hipFree(plan->d_tab);
    delete plan;
    return JSG_OK;
}

int jsg_plan_fft_size(const jsg_plan* plan) { return plan ? plan->n : JSG_ERR_INVALID; }

}   // extern "C"

namespace {
struct IndexOut {   // fused display path: where and how the palette indices of the columns are written ...
    unsigned char* idx;
    long long pitch;
    float vmin, vmax, mult;
    int n_colors;
    // ... or (single-kernel path, idx == nullptr) where the ARGB pixels go
    unsigned* argb;
    long long argb_pitch;
    const int* lut;
    int x_first, x_wrap;
    // ... of `n_images` images of this geometry (single-kernel path; 1: the plain jsg_stft_image_launch)
    int n_images = 1;
    long long in_image_stride = 0, argb_image_stride = 0;
};
}  // namespace

// 2048 points: does this launch take the pair plan (Cfg2048P: a channel PAIR as one complex transform)?  Sum-type mixes (AbsMean, Sum)
// over an EVEN number of channels, float columns (not the display launches), and -- like the "B" kernels, it runs one 8-wave workgroup
// per CU, here of eight columns -- launches that fill their rounds.  plan_select = 3 pins it where it applies (else: as 0).
constexpr bool kPairPlanByDefault = false;  // opt-in only (plan_select = 3): measured 20 % BEHIND Cfg2048B on the C3 dispatch (537 vs 446 us)
static bool wants_plan_pair(int n, const jsg_stft_args* g, int n_cu, bool display, long long frames_of_launch = -1) {
    if (n != 2048 || display) return false;
    if (g->mix_mode != JSG_MIX_ABSMEAN && g->mix_mode != JSG_MIX_SUM) return false;
    if (g->channels < 2 || (g->channels & 1)) return false;
    if (g->plan_select == 3) return true;
    if (g->plan_select != 0 || !kPairPlanByDefault) return false;
    static const int forced2048 = dev_knob_int("JSG_2048_PLAN");   // variant builds only: 4 = pair plan, 2 / 3 = not
    if (forced2048) return forced2048 == 4;
    return b_plan_fills_its_rounds(frames_of_launch >= 0 ? frames_of_launch : g->n_frames, Cfg2048P::TPB, n_cu);
}

// 1024 points (round 6): the two-stage plan Cfg1024B -- float columns, sum-type and one-channel mixes.  plan_select = 2 pins it where it
// applies, 1 pins the three-stage plan.  Automatic rule = what the interleaved A/B decided (tools/plan1024b_ab.py):
// one channel per column (C2, the C4 shard) it is 3-7 % BEHIND the three-stage plan (two waves per SIMD instead of five; its core clock under
// the 1400 W cap is 2.0-2.1 instead of 1.6 GHz and it is still slower), two channels mixed +1.5 %, four +3.3 %, eight +4.1..+4.8 %: taken
// from four channels per column on, for launches that fill their rounds of one workgroup per CU.
constexpr int k1024B_min_channels = 4;
// 1024 / 2048 / 4096 points: does this launch take the "B" kernel?  (nc: channels combined into one column)
static bool wants_plan_b(int n, const jsg_stft_args* g, int nc, int n_cu, long long frames_of_launch = -1) {   // (-1: g->n_frames)
    if (n == 1024) {
        if (g->mix_mode == JSG_MIX_MAX || g->mix_mode == JSG_MIX_MIN) return false;
        if (g->plan_select == 2) return true;
        if (g->plan_select != 0 || nc < k1024B_min_channels) return false;
        return b_plan_fills_its_rounds(frames_of_launch >= 0 ? frames_of_launch : g->n_frames, Cfg1024B::TPB, n_cu);
    }
    if (n != 2048 && n != 4096) return false;
    static const int forced2048 = dev_knob_int("JSG_2048_PLAN");   // variant builds only (JSG_DEV_KNOBS): 2 = "B" | 3
    static const int forced4096 = dev_knob_int("JSG_4096_PLAN");
    const int forced = g->plan_select == 1 ? 3 : g->plan_select == 2 ? 2 : (n == 2048 ? (forced2048 == 4 ? 0 : forced2048) : forced4096);
    const int tpb_b = n == 2048 ? Cfg2048B::TPB : Cfg4096B::TPB;
    return forced == 2 || (forced != 3 && nc >= (n == 2048 ? k2048B_min_channels : k4096B_min_channels) &&
                           b_plan_fills_its_rounds(frames_of_launch >= 0 ? frames_of_launch : g->n_frames, tpb_b, n_cu));
}

namespace {
struct BatchSpec {   // jsg_stft_db_launch_strided: `n` batches of the geometry in jsg_stft_args, batch b at in + b * in_stride, out_db + b * out_stride
    int n;
    long long in_stride, out_stride;
};
}  // namespace

static int stft_launch_impl(const jsg_plan* plan, const jsg_stft_args* g, const IndexOut* io, void* stream, const BatchSpec* bs = nullptr) {
    if (!plan || !g) return jsg_fail(JSG_ERR_INVALID, "jsg_stft_db_launch: null argument");
    if (g->n_frames == 0) return JSG_OK;
    const int H = plan->n / 2 + 1;
    if (!g->in || (!io && !g->out_db) || g->channels <= 0 || g->hop <= 0 || g->feedblocks <= 0 || g->n_frames < 0 ||
        g->first_frame < 0 || g->ring_width <= 0 || g->ring_pos < 0 || g->ring_pos >= g->ring_width ||
        (!io && g->out_pitch < H - (g->out_tail ? 1 : 0)))   // (with a tail plane a column is n/2 floats)
        return jsg_fail(JSG_ERR_INVALID, "jsg_stft_db_launch: bad geometry");
    if (io && g->out_tail) return jsg_fail(JSG_ERR_INVALID, "jsg_stft_image_launch: out_tail belongs to the dB launches");
    if (io && !io->argb && (!io->idx || io->pitch < H || io->n_colors <= 0 || io->n_colors > 256 || g->linear_out))
        return jsg_fail(JSG_ERR_INVALID, "jsg_stft_image_launch: bad index scratch (needs n_colors <= 256, pitch >= n/2+1, dB mode)");
    if (io && io->argb && (!io->lut || io->n_colors <= 0 || io->n_colors > 256 || io->x_wrap <= 0 || io->x_first < 0 || g->linear_out ||
                           (plan->n != 1024 && plan->n != 4096) || io->n_images < 1))
        return jsg_fail(JSG_ERR_INVALID, "jsg_stft_image_launch: bad image geometry for the single-kernel path");
    if (g->n_frames > g->ring_width)
        return jsg_fail(JSG_ERR_INVALID, "jsg_stft_db_launch: more frames than ring columns in one launch (columns would race)");
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev != plan->device)
        return jsg_fail(JSG_ERR_INVALID, "jsg_stft_db_launch: the plan was created on another device");
    const int n_cu = cu_count_of_device(dev);
    if (g->channels > 65535) return jsg_fail(JSG_ERR_UNSUPPORTED, "jsg_stft_db_launch: more than 65535 channels");
    if (g->in_samples != 0) {   // the caller told us how long the channel rows are: refuse to read past them
        const long long j = g->first_frame + g->n_frames - 1;
        const long long start = ((long long)g->hop * g->feedblocks == plan->n)
                                    ? j * g->hop : (j / g->feedblocks) * plan->n + (j % g->feedblocks) * g->hop;
        if (g->in_samples < 0 || start + plan->n > g->in_samples || (g->channels > 1 && g->in_pitch < g->in_samples))
            return jsg_fail(JSG_ERR_INVALID, "jsg_stft_db_launch: the last frame would read past the end of the input rows");
    }
    if (g->n_frames >= (1ll << 31) || g->first_frame + g->n_frames >= (1ll << 31))
        return jsg_fail(JSG_ERR_UNSUPPORTED, "jsg_stft_db_launch: frame index does not fit 31 bits");
    StftKArgs ka{};
    ka.in = g->in;
    ka.in_pitch = g->in_pitch;
    ka.hop = g->hop;
    ka.feedblocks = g->feedblocks;
    ka.regular = ((long long)g->hop * g->feedblocks == plan->n) ? 1 : 0;
    ka.first_frame = (unsigned)g->first_frame;
    ka.n_frames = (unsigned)g->n_frames;
    ka.out = g->out_db;
    ka.out_pitch = g->out_pitch;
    ka.out_cpitch = g->out_channel_pitch;
    ka.ring_w = g->ring_width;
    ka.ring_pos = g->ring_pos;
    ka.tail = io ? nullptr : g->out_tail;
    ka.exact_log = g->exact_log ? 1 : 0;
    ka.tab = plan->d_tab;
    if (io) {
        ka.argb = io->argb;
        ka.argb_pitch = io->argb_pitch;
        ka.lut = io->lut;
        ka.x_first = io->x_first;
        ka.x_wrap = io->x_wrap;
        ka.in_image_stride = io->in_image_stride;
        ka.argb_image_stride = io->argb_image_stride;
        ka.idx = io->idx;
        ka.idx_pitch = io->pitch;
        ka.vmin = io->vmin;
        ka.vmax = io->vmax;
        ka.top = io->vmax * 0.9999f;
        ka.mult = io->mult;
        ka.n_colors = io->n_colors;
        ka.cmap_fast = jsg::cmap_is_fast(ka.vmin, ka.vmax, ka.top, ka.mult, ka.n_colors) ? 1 : 0;
    }
    static const int xcd_remap = dev_knob_set("JSG_NO_XCD_REMAP") ? 0 : 1;   // (variant builds only, as every dev_knob_*)
    ka.xcd_remap = xcd_remap;
    static const int chunked = dev_knob_is("JSG_TRAVERSAL", 'c') ? 1 : 0;
    ka.chunked = chunked;
    ka.per_channel = 0;
    ka.c_begin = 0;
    ka.c_end = g->channels;
    ka.scale = 1.0f;
    ka.divisor = 1.0f;
    ka.exact_div = 0;
    int mixop = 0;
    switch (g->mix_mode) {
        case JSG_MIX_ABSMEAN:
            // m_powerfinal[kk] /= m_channels (Spectrogram.cpp:74).  For a power-of-two channel count the division
            // is an exact scaling and is done as one multiply; otherwise the kernel performs the IEEE division.
            ka.divisor = float(g->channels);
            ka.scale = 1.0f / float(g->channels);
            ka.exact_div = (g->channels & (g->channels - 1)) != 0;
            break;
        case JSG_MIX_MAX: mixop = 1; break;
        case JSG_MIX_MIN: mixop = 2; break;
        case JSG_MIX_LEFT: ka.c_end = 1; break;
        case JSG_MIX_RIGHT:
            // reference Spectrogram.cpp:97-105 reads m_power[1] whenever m_channels > 0; with one channel that is
            // out of bounds there, so it is rejected here
            if (g->channels < 2) return jsg_fail(JSG_ERR_INVALID, "JSG_MIX_RIGHT needs at least two channels");
            ka.c_begin = 1;
            ka.c_end = 2;
            break;
        case JSG_MIX_PER_CHANNEL: ka.per_channel = 1; break;
        case JSG_MIX_SUM: break;
        default: return jsg_fail(JSG_ERR_INVALID, "jsg_stft_db_launch: unknown mix mode");
    }
    ka.linear = g->linear_out ? 1 : 0;
    // one channel per column and nothing to scale: the specialised instantiation (see stft_db_kernel)
    if (mixop == 0 && (ka.per_channel || ka.c_end - ka.c_begin == 1) && ka.scale == 1.0f && !ka.exact_div) mixop = 3;
    if (io && (ka.per_channel || (mixop != 0 && mixop != 3)))
        return jsg_fail(JSG_ERR_UNSUPPORTED, "jsg_stft_image_launch: AbsMean / Sum / Left / Right mixes only");
    // 2048 / 4096 points: the "B" plan (two-stage / one wavefront per frame) where several channels are mixed into one column and the
    // launch fills its rounds, else the other one (see Cfg2048B, Cfg4096B, kB_min_round_fill).  The choice depends on the launch
    // geometry (channels per column, frames, CU count) only; sub-launches of one stream that fall on different sides of the rule
    // agree within the float32 bound, not bit for bit (jsg.h: plan_select pins one plan).
    // (the single-kernel display path exists for the one-wavefront-per-frame plans: at 4096 points that is "B")
    const long long rows = bs ? (long long)bs->n * (ka.per_channel ? g->channels : 1) : 1;
    // (a strided launch is judged by the frames of ALL its rows: the "B" kernels then fill their rounds)
    const bool plan_p = wants_plan_pair(plan->n, g, n_cu, io != nullptr, bs ? rows * g->n_frames : -1) && mixop == 0 && !ka.per_channel &&
                        ((ka.c_end - ka.c_begin) & 1) == 0;
    const bool plan_b = !plan_p && !(plan->n == 1024 && io) &&   // (1024 points: the display launches keep the three-stage plans)
                        ((io && io->argb && plan->n == 4096) ||
                         wants_plan_b(plan->n, g, ka.per_channel ? 1 : ka.c_end - ka.c_begin, n_cu, bs ? rows * g->n_frames : -1));
    if (plan_b) ka.tab = plan->d_tab_b;
    if (plan_p) ka.tab = plan->d_tab_p;
    // "runs" (STREAM == 2, round 6): strided one-channel dB launches of the 1024-point three-stage plan at exactly 50 % overlap -- a wavefront
    // transforms kRunLen consecutive columns and keeps the overlapped half of the raw frame in registers (see stft_db_kernel)
    static const bool no_runs = dev_knob_set("JSG_NO_RUNS");   // (variant builds only)
    const bool runs = bs && plan->n == 1024 && !plan_b && !io && mixop == 3 && ka.regular && 2 * g->hop == plan->n && !ka.chunked && !no_runs;
    int tpb = 0;
    switch (plan->n) {
        case 512: tpb = Cfg512::TPB; break;
        case 1024: tpb = (io && io->argb) ? Cfg1024I::TPB : plan_b ? Cfg1024B::TPB : runs ? Cfg1024::TPB * jsg::kRunLen : Cfg1024::TPB; break;   // (the display path's eight-column workgroups)
        case 2048: tpb = plan_p ? Cfg2048P::TPB : plan_b ? Cfg2048B::TPB : Cfg2048::TPB; break;
        case 4096: tpb = plan_b ? Cfg4096B::TPB : Cfg4096::TPB; break;
        case 8192: tpb = Cfg8192::TPB; break;
    }
    long long want = (g->n_frames + tpb - 1) / tpb;   // workgroup iterations ("groups" of tpb frames) of the launch
    if (io && io->argb) {   // single-kernel display path: the groups are numbered through the images of the launch
        const long long gpi = want;
        want = gpi * io->n_images;
        if (gpi > (1ll << 20) || want > (1ll << 20))
            return jsg_fail(JSG_ERR_UNSUPPORTED, "jsg_stft_image_launch: more than 2^20 groups of eight columns in one launch");
        ka.img_gpi = unsigned(gpi);
        ka.n_groups = unsigned(want);
        ka.img_magic = ((1ull << 40) + (unsigned long long)gpi - 1) / (unsigned long long)gpi;
    }
    if (bs) {   // strided multi-batch launch: the groups are numbered through the rows (batches, or batch x channel)
        const long long gpr = want;
        want = gpr * rows;
        if (gpr > (1ll << 20) || want > (1ll << 20))
            return jsg_fail(JSG_ERR_UNSUPPORTED, "jsg_stft_db_launch_strided: more than 2^20 workgroup steps in one launch");
        ka.img_gpi = unsigned(gpr);
        ka.n_groups = unsigned(want);
        ka.img_magic = ((1ull << 40) + (unsigned long long)gpr - 1) / (unsigned long long)gpr;
        ka.bat_cpb = ka.per_channel ? unsigned(g->channels) : 1u;
        ka.bat_magic = ((1ull << 40) + (unsigned long long)ka.bat_cpb - 1) / (unsigned long long)ka.bat_cpb;
        ka.in_image_stride = bs->in_stride;
        ka.out_batch_stride = bs->out_stride;
    }
    const int ny = (ka.per_channel && !bs) ? g->channels : 1;
    if (ny > 65535) return jsg_fail(JSG_ERR_UNSUPPORTED, "jsg_stft_db_launch: more than 65535 channels in per-channel mode");
    static const int blocks_per_cu_env = std::max(0, dev_knob_int("JSG_STFT_BLOCKS_PER_CU"));
    // workgroups per CU of the grid; the rest of the frames is looped over.  The "B" plans hold one workgroup per CU: a grid of
    // exactly that many keeps the tables and the prefetch pipeline alive across a workgroup's frames (16 384 mono 4096-point
    // frames: 65 vs 79 us with a grid of 8 per CU); the other plans do best with more workgroups than are resident.
    // Strided multi-batch launches of the small-workgroup plans do better with MORE, shorter-lived workgroups than fit a CU at once (five
    // four-wave workgroups of the 1024-point plan, three of the others): the hardware dispatcher hands the later ones out as the earlier ones
    // finish -- closer to address order, and the end of the dispatch is balanced -- where a grid of 8 per CU runs as one full round of resident
    // workgroups followed by a 60 %-full one.  Measured round 5 (tools/tail_probe.py TP_BPC, interleaved, us per dispatch 8 -> 16 -> 32 per
    // CU): C2 212.7 -> 204.0 -> 205.2 (reference layout 209.4 -> 204.0 -> 204.2), 512 points hop 512 143.2 -> 139.2 -> 135.6, hop 256 103.0
    // -> 98.2 -> 98.3, 2048 points mono 100.8 -> 97.2 -> 101.6, 1024 points 8 channels mixed 161.1 -> 159.8 -> 164.6, 8192 points 94.6 ->
    // 100.9.  (Round 4 had found 32 per CU behind 8 with the eight-wave workgroups and the old remap.)
    int bpc_default = 8;
    if (bs) {
        const int ncol = ka.per_channel ? 1 : ka.c_end - ka.c_begin;
        if (plan->n == 512) bpc_default = 32;
        else if (plan->n == 1024) bpc_default = ncol == 1 ? 32 : 16;
        else if (plan->n == 2048) bpc_default = 16;
    }
    const int bpc = g->blocks_per_cu > 0 ? g->blocks_per_cu : (blocks_per_cu_env > 0 ? blocks_per_cu_env : ((plan_b || plan_p) ? 1 : bpc_default));
    long long max_blocks = (long long)n_cu * bpc / ny;     // resident workgroups; the rest is looped over
    if (max_blocks < 64) max_blocks = 64;
    static const int max_blocks_env = dev_knob_int("JSG_STFT_MAX_BLOCKS");
    if (max_blocks_env > 0) max_blocks = max_blocks_env;
    // steps per workgroup first, then the smallest grid that covers the launch with that many: at most iters - 1 surplus steps in the
    // last round (a grid of max_blocks would leave up to max_blocks - 1 of them: 9 workgroups per CU for 30 720 steps measured
    // 0.54 against 0.58 of 8 TB/s at 8 per CU)
    ka.iters = int((want + max_blocks - 1) / max_blocks);
    const int nblk = int((want + ka.iters - 1) / ka.iters);
    const dim3 grid(nblk, ny);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipError_t err = hipSuccess;
    if (bs) {
        switch (plan->n) {
            case 512: err = launch_strided_Cfg512(ka, mixop, grid, s); break;
            case 1024: err = plan_b ? launch_strided_Cfg1024B(ka, mixop, grid, s) : runs ? launch_runs_Cfg1024(ka, mixop, grid, s) : launch_strided_Cfg1024(ka, mixop, grid, s); break;
            case 2048: err = plan_p ? launch_strided_Cfg2048P(ka, mixop, grid, s) : plan_b ? launch_strided_Cfg2048B(ka, mixop, grid, s) : launch_strided_Cfg2048(ka, mixop, grid, s); break;
            case 4096: err = plan_b ? launch_strided_Cfg4096B(ka, mixop, grid, s) : launch_strided_Cfg4096(ka, mixop, grid, s); break;
            case 8192: err = launch_strided_Cfg8192(ka, mixop, grid, s); break;
        }
        if (err != hipSuccess) return jsg_fail_hip(err, "jsg_stft_db_launch_strided");
        return JSG_OK;
    }
    switch (plan->n) {
        case 512: err = launch_Cfg512(ka, mixop, grid, s); break;
        case 1024: err = (io && io->argb) ? launch_Cfg1024I(ka, mixop, grid, s) : plan_b ? launch_Cfg1024B(ka, mixop, grid, s) : launch_Cfg1024(ka, mixop, grid, s); break;
        case 2048: err = plan_p ? launch_Cfg2048P(ka, mixop, grid, s) : plan_b ? launch_Cfg2048B(ka, mixop, grid, s) : launch_Cfg2048(ka, mixop, grid, s); break;
        case 4096: err = plan_b ? launch_Cfg4096B(ka, mixop, grid, s) : launch_Cfg4096(ka, mixop, grid, s); break;
        case 8192: err = launch_Cfg8192(ka, mixop, grid, s); break;
    }
    if (err != hipSuccess) return jsg_fail_hip(err, "jsg_stft_db_launch");
    return JSG_OK;
}

extern "C" {

int jsg_stft_db_launch(const jsg_plan* plan, const jsg_stft_args* g, void* stream) { return stft_launch_impl(plan, g, nullptr, stream); }

// Which kernel configuration jsg_stft_db_launch runs for these arguments on the current device (benchmarks and tests name the
// kernel they time / check with it; the rule itself: wants_plan_b).
int jsg_stft_kernel_name(const jsg_plan* plan, const jsg_stft_args* g, char* out, int out_len) {
    if (!plan || !g || !out || out_len < 24) return jsg_fail(JSG_ERR_INVALID, "jsg_stft_kernel_name: bad argument");
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return jsg_fail(JSG_ERR_NO_DEVICE, "jsg_stft_kernel_name: no device");
    int nc = g->channels;
    if (g->mix_mode == JSG_MIX_LEFT || g->mix_mode == JSG_MIX_RIGHT || g->mix_mode == JSG_MIX_PER_CHANNEL) nc = 1;
    const bool pp = wants_plan_pair(plan->n, g, cu_count_of_device(dev), false);
    const bool b = !pp && wants_plan_b(plan->n, g, nc, cu_count_of_device(dev));
    std::snprintf(out, size_t(out_len), "Cfg%d%s", plan->n, pp ? "P" : b ? "B" : "");
    return JSG_OK;
}

static int colormap_launch_impl(const jsg_colormap_args* g, const unsigned char* idx_in, long long idx_in_pitch, void* stream);

// Fused display path (reference Spectrogram.cpp:632-648: the colour loop consumes the column it has just been given):
// the STFT kernel's epilogue maps every bin to its palette index (CColorPalette::getRGBColor) and writes 1 byte per bin
// into `index_scratch` -- the dB column never goes to memory -- and the colour kernel turns those columns into ARGB
// image rows through its LDS transpose tiles.  Per column of C5: 4096 B in + 2049 B + 2049 B + 8196 B instead of
// 4096 + 8196 + 8196 + 8196 B.  The image is bit-identical to jsg_stft_db_launch + jsg_colormap_launch.
// Does jsg_stft_image_launch run as ONE kernel for these arguments?  Where the plan's workgroups hold eight whole columns: 1024
// points, and 4096 points when the launcher's choice for the launch is the one-wavefront-per-frame kernel ("B": automatic rule or
// plan_select = 2) -- so the image is always that of jsg_stft_db_launch (same plan_select) + jsg_colormap_launch, bit for bit.
static bool image_takes_one_kernel(const jsg_plan* plan, const jsg_stft_image_args* g, int n_images = 1) {
    static const int two_kernels = dev_knob_set("JSG_IMAGE_TWO_KERNELS") ? 1 : 0;
    const jsg_colormap_args& c = g->colour;
    if (two_kernels || !(plan->n == 1024 || plan->n == 4096) || !c.argb_out || c.index_out || c.n_colors <= 0 || c.n_colors > 256 ||
        !c.lut || c.x_wrap <= 0 || c.x_first < 0 || c.n_cols > c.x_wrap ||
        (long long)c.height * c.argb_pitch * 4 >= (1ll << 32))   // (the kernel addresses pixels by 32-bit byte offsets from the image's start)
        return false;
    const int mm = g->stft.mix_mode;
    if (mm == JSG_MIX_PER_CHANNEL || mm == JSG_MIX_MAX || mm == JSG_MIX_MIN) return false;
    if (plan->n == 4096) {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        const int nc = (mm == JSG_MIX_LEFT || mm == JSG_MIX_RIGHT) ? 1 : g->stft.channels;
        const long long gpi = (g->stft.n_frames + Cfg4096B::TPB - 1) / Cfg4096B::TPB;
        return wants_plan_b(4096, &g->stft, nc, cu_count_of_device(dev), gpi * Cfg4096B::TPB * n_images);   // (the fill of the whole launch)
    }
    return true;
}

int jsg_stft_image_needs_scratch(const jsg_plan* plan, const jsg_stft_image_args* g) {
    if (!plan || !g) return jsg_fail(JSG_ERR_INVALID, "jsg_stft_image_needs_scratch: null argument");
    return image_takes_one_kernel(plan, g) ? 0 : 1;
}

int jsg_stft_image_strided_needs_scratch(const jsg_plan* plan, const jsg_stft_image_args* g, int n_images) {
    if (!plan || !g || n_images < 1) return jsg_fail(JSG_ERR_INVALID, "jsg_stft_image_strided_needs_scratch: bad argument");
    return image_takes_one_kernel(plan, g, n_images) ? 0 : 1;
}

int jsg_stft_image_launch(const jsg_plan* plan, const jsg_stft_image_args* g, void* stream) {
    if (!plan || !g) return jsg_fail(JSG_ERR_INVALID, "jsg_stft_image_launch: null argument");
    if (g->stft.n_frames == 0) return JSG_OK;
    const jsg_colormap_args& c = g->colour;
    if (c.ring_width <= 0 || c.col_first < 0)   // (before the modulo below: nothing fatal may cross the C boundary)
        return jsg_fail(JSG_ERR_INVALID, "jsg_stft_image_launch: bad colour geometry (ring_width <= 0 or col_first < 0)");
    if (c.n_cols != g->stft.n_frames || c.ring_width != g->stft.ring_width || c.height != plan->n / 2 + 1 ||
        (c.col_first % c.ring_width) != g->stft.ring_pos)
        return jsg_fail(JSG_ERR_INVALID, "jsg_stft_image_launch: the colour loop must cover exactly the columns of the launch");
    if (c.argb_out && (c.x_wrap <= 0 || c.argb_pitch < c.x_wrap))   // (a zero, negative or too small pitch would be out-of-bounds device writes)
        return jsg_fail(JSG_ERR_INVALID, "jsg_stft_image_launch: argb_pitch must be at least x_wrap pixels");
    // Single kernel where the plan's workgroups hold eight whole columns (1024 points; 4096 points: the one-wavefront-per-frame
    // kernel): the workgroup colours its columns itself and writes ARGB rows -- nothing but the input is read, nothing but the
    // image is written.  Everything else: index columns through `index_scratch` + the colour kernel.
    const bool single = image_takes_one_kernel(plan, g);
    if (!single && !g->index_scratch)
        return jsg_fail(JSG_ERR_INVALID, "jsg_stft_image_launch: this launch needs index_scratch (jsg_stft_image_needs_scratch)");
    if (single) {
        IndexOut io{nullptr, 0, c.vmin, c.vmax, c.access_mult, c.n_colors, c.argb_out, (long long)c.argb_pitch, c.lut,
                    c.x_first % c.x_wrap, c.x_wrap};
        return stft_launch_impl(plan, &g->stft, &io, stream);
    }
    IndexOut io{g->index_scratch, (long long)g->index_scratch_pitch, c.vmin, c.vmax, c.access_mult, c.n_colors, nullptr, 0, nullptr, 0, 1};
    int rc = stft_launch_impl(plan, &g->stft, &io, stream);
    if (rc != JSG_OK) return rc;
    return colormap_launch_impl(&c, g->index_scratch, (long long)g->index_scratch_pitch, stream);
}

// `n_images` images of ONE geometry in one call: image i reads stft.in + i * in_image_stride (floats) and writes
// colour.argb_out + i * argb_image_stride (pixels); everything else in `g` describes one image and holds for all of them.
// Where jsg_stft_image_launch would take the single-kernel form for a launch of this total size, the images share ONE kernel launch:
// its workgroups walk through the columns of all images, so the tables are loaded once, the next columns travel while the current
// ones are transformed, and no workgroup slot idles at the end of an image (C5: 235 eight-column groups per image on 256 CUs).
// Otherwise the images are launched one after the other on `stream` (two kernels each; `index_scratch` is reused in stream order).
int jsg_stft_image_launch_strided(const jsg_plan* plan, const jsg_stft_image_args* g, int n_images, int64_t in_image_stride,
                                  int64_t argb_image_stride, void* stream) {
    if (!plan || !g) return jsg_fail(JSG_ERR_INVALID, "jsg_stft_image_launch_strided: null argument");
    if (n_images < 0) return jsg_fail(JSG_ERR_INVALID, "jsg_stft_image_launch_strided: negative image count");
    if (n_images == 0 || g->stft.n_frames == 0) return JSG_OK;
    if (n_images == 1) return jsg_stft_image_launch(plan, g, stream);
    const jsg_colormap_args& c = g->colour;
    if (c.ring_width <= 0 || c.col_first < 0 || c.height <= 0 || c.argb_pitch <= 0)
        return jsg_fail(JSG_ERR_INVALID, "jsg_stft_image_launch_strided: bad colour geometry");
    if (in_image_stride < 0 || (c.argb_out && argb_image_stride < (long long)c.height * c.argb_pitch))
        return jsg_fail(JSG_ERR_INVALID, "jsg_stft_image_launch_strided: the images would overlap (argb_image_stride < height * argb_pitch) or a stride is negative");
    if (c.argb_out && c.argb_pitch < c.x_wrap)
        return jsg_fail(JSG_ERR_INVALID, "jsg_stft_image_launch_strided: argb_pitch must be at least x_wrap pixels");
    // in_samples describes the rows of EVERY image: with a stride the rows of one image must end before the next image begins (a stride
    // of 0 = the same input for every image)
    if (g->stft.in_samples != 0 && in_image_stride != 0 &&
        in_image_stride < (long long)(g->stft.channels - 1) * g->stft.in_pitch + g->stft.in_samples)
        return jsg_fail(JSG_ERR_INVALID, "jsg_stft_image_launch_strided: in_image_stride is smaller than one image's input rows "
                                         "((channels - 1) * in_pitch + in_samples); 0 = the same input for all images");
    if (c.n_cols != g->stft.n_frames || c.ring_width != g->stft.ring_width || c.height != plan->n / 2 + 1 ||
        (c.col_first % c.ring_width) != g->stft.ring_pos)
        return jsg_fail(JSG_ERR_INVALID, "jsg_stft_image_launch_strided: the colour loop must cover exactly the columns of the launch");
    if (image_takes_one_kernel(plan, g, n_images)) {
        IndexOut io{nullptr, 0, c.vmin, c.vmax, c.access_mult, c.n_colors, c.argb_out, (long long)c.argb_pitch, c.lut,
                    c.x_first % c.x_wrap, c.x_wrap, n_images, (long long)in_image_stride, (long long)argb_image_stride};
        return stft_launch_impl(plan, &g->stft, &io, stream);
    }
    for (int i = 0; i < n_images; ++i) {
        jsg_stft_image_args one = *g;
        one.stft.in = g->stft.in + (long long)i * in_image_stride;
        if (one.colour.argb_out) one.colour.argb_out = g->colour.argb_out + (long long)i * argb_image_stride;
        if (one.colour.index_out) return jsg_fail(JSG_ERR_UNSUPPORTED, "jsg_stft_image_launch_strided: index_out planes are not strided (launch the images one by one)");
        const int rc = jsg_stft_image_launch(plan, &one, stream);
        if (rc != JSG_OK) return rc;
    }
    return JSG_OK;
}

// K independent batches of ONE geometry in ONE kernel launch on ONE stream (reference loop: Spectrogram.cpp:50-119 over K streams'
// worth of blocks): batch b reads args->in + b * in_batch_stride and writes its own ring at args->out_db + b * out_batch_stride.
// The workgroups of the launch walk through the columns of all batches: tables loaded once per workgroup, no ramp-up and drain per
// batch, no dependence on extra streams, hardware queues or issuing threads.
static int strided_checks(const jsg_plan* plan, const jsg_stft_args* g, int n_batches, int64_t in_batch_stride, int64_t out_batch_stride, const char* who) {
    if (!plan || !g) return jsg_fail(JSG_ERR_INVALID, (std::string(who) + ": null argument").c_str());
    if (n_batches < 0) return jsg_fail(JSG_ERR_INVALID, (std::string(who) + ": negative batch count").c_str());
    if (in_batch_stride < 0 || out_batch_stride < 0) return jsg_fail(JSG_ERR_INVALID, (std::string(who) + ": negative stride").c_str());
    if (n_batches > 1 && g->n_frames > 0) {
        if (g->channels <= 0 || g->ring_width <= 0 || g->out_pitch <= 0) return jsg_fail(JSG_ERR_INVALID, (std::string(who) + ": bad geometry").c_str());
        const long long ring_extent = (g->mix_mode == JSG_MIX_PER_CHANNEL ? (long long)(g->channels - 1) * g->out_channel_pitch : 0ll) +
                                      (long long)(g->ring_width - 1) * g->out_pitch + plan->n / 2 + (g->out_tail ? 0 : 1);
        if (out_batch_stride < ring_extent)
            return jsg_fail(JSG_ERR_INVALID, (std::string(who) + ": the rings of consecutive batches would overlap (out_batch_stride too small)").c_str());
        // the caller told us how long the channel rows are: consecutive batches may share samples (stride < row length: a long stream
        // cut along time), but a batch must not reach past what the stride + in_samples describe for the LAST batch -- checked per
        // batch by the launcher -- and a stride of 0 means "the same input for every batch"
        if (g->in_samples != 0 && in_batch_stride != 0 && g->channels > 1 && g->in_pitch < g->in_samples)
            return jsg_fail(JSG_ERR_INVALID, (std::string(who) + ": in_pitch < in_samples").c_str());
    }
    return JSG_OK;
}

int jsg_stft_db_launch_strided(const jsg_plan* plan, const jsg_stft_args* g, int n_batches, int64_t in_batch_stride, int64_t out_batch_stride,
                               void* stream) {
    int rc = strided_checks(plan, g, n_batches, in_batch_stride, out_batch_stride, "jsg_stft_db_launch_strided");
    if (rc != JSG_OK) return rc;
    if (n_batches == 0 || g->n_frames == 0) return JSG_OK;
    if (n_batches == 1) return jsg_stft_db_launch(plan, g, stream);
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return jsg_fail(JSG_ERR_NO_DEVICE, "jsg_stft_db_launch_strided: no device");
    const bool one_by_one = g->mix_mode == JSG_MIX_MAX || g->mix_mode == JSG_MIX_MIN;   // (no strided instantiation: rare modes)
    if (g->n_frames < 0 || g->channels <= 0) return jsg_fail(JSG_ERR_INVALID, "jsg_stft_db_launch_strided: bad geometry");
    const long long rows_per_batch = g->mix_mode == JSG_MIX_PER_CHANNEL ? g->channels : 1;
    // at most 2^20 workgroup steps per launch (the kernel's group -> row arithmetic): longer jobs go out in several launches
    // (counted with the smallest workgroup step of the plan's kernels: 16 / 8 / 4 / 2 / 1 columns at 512 ... 8192 points)
    const int tpb_min = plan->n == 512 ? Cfg512::TPB : plan->n == 1024 ? Cfg1024::TPB : plan->n == 2048 ? Cfg2048::TPB : plan->n == 4096 ? Cfg4096::TPB : Cfg8192::TPB;
    const long long steps_per_batch = rows_per_batch * ((g->n_frames + tpb_min - 1) / tpb_min);
    const long long per_launch = one_by_one ? 1 : std::max(1ll, (1ll << 20) / std::max(1ll, steps_per_batch));
    for (long long b0 = 0; b0 < n_batches; b0 += per_launch) {
        const int nb = int(std::min<long long>(per_launch, n_batches - b0));
        jsg_stft_args one = *g;
        one.in = g->in + b0 * in_batch_stride;
        one.out_db = g->out_db ? g->out_db + b0 * out_batch_stride : nullptr;
        if (g->out_tail) one.out_tail = g->out_tail + b0 * rows_per_batch * g->ring_width;   // (a dense plane: rows x ring_width)
        if (nb == 1) rc = jsg_stft_db_launch(plan, &one, stream);
        else {
            BatchSpec bs{nb, (long long)in_batch_stride, (long long)out_batch_stride};
            rc = stft_launch_impl(plan, &one, nullptr, stream, &bs);
        }
        if (rc != JSG_OK) return rc;
    }
    return JSG_OK;
}

// The kernel a strided launch takes, as text (as jsg_stft_kernel_name, judged by the frames of the whole launch).
int jsg_stft_db_strided_kernel_name(const jsg_plan* plan, const jsg_stft_args* g, int n_batches, int64_t in_batch_stride, char* out, int out_len) {
    if (!plan || !g || !out || out_len < 24 || n_batches < 1) return jsg_fail(JSG_ERR_INVALID, "jsg_stft_db_strided_kernel_name: bad argument");
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) return jsg_fail(JSG_ERR_NO_DEVICE, "jsg_stft_db_strided_kernel_name: no device");
    if (n_batches == 1) return jsg_stft_kernel_name(plan, g, out, out_len);
    const int n_cu = cu_count_of_device(dev);
    (void)in_batch_stride;
    int nc = g->channels;
    if (g->mix_mode == JSG_MIX_LEFT || g->mix_mode == JSG_MIX_RIGHT || g->mix_mode == JSG_MIX_PER_CHANNEL) nc = 1;
    const long long rows = (long long)n_batches * (g->mix_mode == JSG_MIX_PER_CHANNEL ? g->channels : 1);
    const bool pp = wants_plan_pair(plan->n, g, n_cu, false, rows * g->n_frames);
    const bool b = !pp && wants_plan_b(plan->n, g, nc, n_cu, rows * g->n_frames);
    std::snprintf(out, size_t(out_len), "Cfg%d%s", plan->n, pp ? "P" : b ? "B" : "");
    return JSG_OK;
}

int jsg_stft_db_launch_many(const jsg_plan* plan, const jsg_stft_args* args, int count, void* const* streams, int n_streams) {
    if (!plan || (!args && count > 0) || count < 0 || n_streams < 0 || (n_streams > 0 && !streams))
        return jsg_fail(JSG_ERR_INVALID, "jsg_stft_db_launch_many: bad argument");
    for (int i = 0; i < count; ++i) {
        void* st = n_streams > 0 ? streams[i % n_streams] : nullptr;
        const int rc = jsg_stft_db_launch(plan, &args[i], st);
        if (rc != JSG_OK) return rc;
    }
    return JSG_OK;
}

// The same with `n_threads` host threads issuing: stream k is served by thread k % n_threads only, so the order inside every
// stream is the order of `args`.  One thread issues a launch every ~3.5 us, which is about what eight overlapped
// 4096-frame launches take on the GPU; a second thread takes the host out of the picture (tools/abbench --threads).
int jsg_stft_db_launch_many_threads(const jsg_plan* plan, const jsg_stft_args* args, int count, void* const* streams, int n_streams,
                                    int n_threads) {
    if (n_threads <= 1 || n_streams <= 1) return jsg_stft_db_launch_many(plan, args, count, streams, n_streams);
    if (!plan || (!args && count > 0) || count < 0 || !streams) return jsg_fail(JSG_ERR_INVALID, "jsg_stft_db_launch_many_threads: bad argument");
    if (n_threads > n_streams) n_threads = n_streams;
    if (n_threads > 16) n_threads = 16;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return jsg_fail(JSG_ERR_NO_DEVICE, "jsg_stft_db_launch_many_threads: no device");
    std::vector<int> rcs(size_t(n_threads), JSG_OK);
    std::vector<std::string> errs{size_t(n_threads)};
    auto body = [&](int t) {
        if (hipSetDevice(dev) != hipSuccess) { rcs[size_t(t)] = JSG_ERR_HIP; return; }
        for (int i = 0; i < count; ++i) {
            if ((i % n_streams) % n_threads != t) continue;
            const int rc = jsg_stft_db_launch(plan, &args[i], streams[i % n_streams]);
            if (rc != JSG_OK) { rcs[size_t(t)] = rc; errs[size_t(t)] = tls_error(); return; }
        }
    };
    std::vector<std::thread> th;
    int spawn_failed_from = -1;   // std::thread's constructor can throw (std::system_error): nothing may escape the C boundary
    try {
        th.reserve(size_t(n_threads));
        for (int t = 1; t < n_threads; ++t) th.emplace_back(body, t);
    } catch (...) {
        spawn_failed_from = int(th.size()) + 1;
    }
    body(0);
    if (spawn_failed_from > 0)   // the streams of the threads that could not be started are served by this one
        for (int t = spawn_failed_from; t < n_threads; ++t) body(t);
    for (auto& x : th) x.join();
    for (int t = 0; t < n_threads; ++t)
        if (rcs[size_t(t)] != JSG_OK) { tls_error() = errs[size_t(t)]; return rcs[size_t(t)]; }
    return JSG_OK;
}

// ---- the library's own launch pool: the caller's stream plus three streams of the library, fork / join against the caller's stream ----
// The compute front end of the GPU runs four hardware queues at a time: with a fifth BUSY queue the command processor time-slices them
// and the rate collapses (measured, C2: 1.08e9 frames/s with four busy queues, 0.35e9 with five).  The caller's own stream is therefore one
// of the four working streams -- it would be busy anyway with the fork and join events -- and not a fifth one beside them.
namespace {
struct LaunchPool {
    std::mutex mu;                 // one call at a time per device (the fork / join events are reused)
    bool ready = false;
    hipStream_t s[3] = {nullptr, nullptr, nullptr};
    hipEvent_t fork = nullptr, join[3] = {nullptr, nullptr, nullptr};
};
LaunchPool g_pools[64];
constexpr int kPoolStreams = 4;    // working streams (the caller's + 3): 3 streams 1.25e9, 4 streams 1.32e9, 5 and more 0.5-0.8e9 frames/s at C2 (round 2)
constexpr int kPoolThreads = 2;    // one host thread issues a launch every ~3.5 us, the GPU finishes one every ~3.1-3.9 us

// streams and events of a device's pool, all or nothing (p.mu held)
int pool_ensure(LaunchPool& p) {
    if (p.ready) return JSG_OK;
    hipError_t err = hipSuccess;
    for (int i = 0; i < 3 && err == hipSuccess; ++i) {
        err = hipStreamCreateWithFlags(&p.s[i], hipStreamNonBlocking);
        if (err == hipSuccess) err = hipEventCreateWithFlags(&p.join[i], hipEventDisableTiming);
    }
    if (err == hipSuccess) err = hipEventCreateWithFlags(&p.fork, hipEventDisableTiming);
    if (err != hipSuccess) {   // give back what was created: the next call starts from nothing again
        for (int i = 0; i < 3; ++i) {
            if (p.join[i]) (void)hipEventDestroy(p.join[i]);
            if (p.s[i]) (void)hipStreamDestroy(p.s[i]);
            p.join[i] = nullptr;
            p.s[i] = nullptr;
        }
        if (p.fork) (void)hipEventDestroy(p.fork);
        p.fork = nullptr;
        return jsg_fail_hip(err, "jsg_stft_db_launch_batches: creating the launch pool");
    }
    p.ready = true;
    return JSG_OK;
}
}  // namespace

// called by jsg_plan_create: the pool of the plan's device exists before any launch, so the first jsg_stft_db_launch_batches may
// already sit inside a stream capture
static void pool_prepare_for_device(int dev) {
    if (dev < 0 || dev >= 64) return;
    LaunchPool& p = g_pools[dev];
    std::lock_guard<std::mutex> lk(p.mu);
    (void)pool_ensure(p);
}

int jsg_stft_db_launch_batches(const jsg_plan* plan, const jsg_stft_args* args, int count, void* stream) {
    if (!plan || (!args && count > 0) || count < 0) return jsg_fail(JSG_ERR_INVALID, "jsg_stft_db_launch_batches: bad argument");
    if (count == 0) return JSG_OK;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return jsg_fail(JSG_ERR_NO_DEVICE, "jsg_stft_db_launch_batches: no device");
    hipStream_t user = reinterpret_cast<hipStream_t>(stream);
    if (count == 1) return jsg_stft_db_launch(plan, &args[0], stream);
    LaunchPool& p = g_pools[dev];
    std::lock_guard<std::mutex> lk(p.mu);
    hipError_t err = hipSuccess;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(user, &cap);
    if (!p.ready) {
        // (jsg_plan_create has normally done this already: pool_ensure; a stream cannot be created in the middle of a capture)
        if (cap == hipStreamCaptureStatusActive)
            return jsg_fail(JSG_ERR_INVALID, "jsg_stft_db_launch_batches: the launch pool of this device does not exist yet and cannot be created "
                                             "inside a stream capture (create the plan on this device first, or call once outside the capture)");
        const int rc = pool_ensure(p);
        if (rc != JSG_OK) return rc;
    }
    // launches of the one-workgroup-per-CU kernels (2048 / 4096 "B") fill the GPU by themselves: two streams are enough to hide
    // the ramp-up and drain (measured at C3, round 2), four make them queue behind each other
    int nc0 = args[0].channels;
    if (args[0].mix_mode == JSG_MIX_LEFT || args[0].mix_mode == JSG_MIX_RIGHT || args[0].mix_mode == JSG_MIX_PER_CHANNEL) nc0 = 1;
    const int want_s = wants_plan_b(plan->n, &args[0], nc0, cu_count_of_device(dev)) ? 2 : kPoolStreams;
    const int n_s = count < want_s ? count : want_s;     // working streams: sv[0] = the caller's, sv[1..] = the library's
    // fork: the library's streams start behind everything already enqueued on the caller's stream
    err = hipEventRecord(p.fork, user);
    int forked = 0;      // library streams that wait for the fork event: every one of them is joined below, whatever happens in between
    for (int i = 0; i + 1 < n_s && err == hipSuccess; ++i) {
        err = hipStreamWaitEvent(p.s[i], p.fork, 0);
        if (err == hipSuccess) ++forked;
    }
    void* sv[kPoolStreams] = {user, p.s[0], p.s[1], p.s[2]};
    // (the default stream cannot be named by a null handle in a round-robin table: launch_many treats NULL as "default stream" too)
    int rc = JSG_OK;
    if (err == hipSuccess)
        rc = jsg_stft_db_launch_many_threads(plan, args, count, sv, n_s, cap == hipStreamCaptureStatusActive ? 1 : kPoolThreads);
    // join (also after a failed fork or launch: what was enqueued must still be ordered before the caller's later work, and a capture
    // must not be left with branches that never come back)
    hipError_t jerr = hipSuccess;
    for (int i = 0; i < forked; ++i) {
        hipError_t e1 = hipEventRecord(p.join[i], p.s[i]);
        if (e1 == hipSuccess) e1 = hipStreamWaitEvent(user, p.join[i], 0);
        if (e1 != hipSuccess && jerr == hipSuccess) jerr = e1;
    }
    if (err != hipSuccess) return jsg_fail_hip(err, "jsg_stft_db_launch_batches: fork");
    if (rc != JSG_OK) return rc;
    if (jerr != hipSuccess) return jsg_fail_hip(jerr, "jsg_stft_db_launch_batches: join");
    return JSG_OK;
}

int jsg_db_from_power_launch(const float* power, float* out, int64_t count, float divisor, void* stream) {
    return jsg_db_from_power_launch_ex(power, out, count, divisor, 0, stream);
}

int jsg_db_from_power_launch_ex(const float* power, float* out, int64_t count, float divisor, int exact_log, void* stream) {
    if (count == 0) return JSG_OK;
    if (!power || !out || count < 0 || !(divisor > 0.f)) return jsg_fail(JSG_ERR_INVALID, "jsg_db_from_power_launch: bad argument");
    const long long blocks = (count + 255) / 256;
    hipLaunchKernelGGL(db_from_power_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0,
                       reinterpret_cast<hipStream_t>(stream), power, out, (long long)count, divisor, exact_log ? 1 : 0);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return jsg_fail_hip(err, "jsg_db_from_power_launch");
    return JSG_OK;
}

int jsg_columns_from_tail_layout_launch(const float* db, int64_t db_pitch, const float* tail, int n_columns, int height, float* dst, int64_t dst_pitch,
                                        void* stream) {
    if (n_columns == 0) return JSG_OK;
    if (!db || !tail || !dst || n_columns < 0 || height < 2 || db_pitch < height - 1 || dst_pitch < height)
        return jsg_fail(JSG_ERR_INVALID, "jsg_columns_from_tail_layout_launch: bad argument (db_pitch >= height - 1, dst_pitch >= height)");
    hipLaunchKernelGGL(tail_merge_kernel, dim3((unsigned)n_columns), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), db, (long long)db_pitch, tail,
                       height - 1, dst, (long long)dst_pitch);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return jsg_fail_hip(err, "jsg_columns_from_tail_layout_launch");
    return JSG_OK;
}

int jsg_calib_copy_launch(const void* src, void* dst, int64_t bytes, void* stream) {
    if (bytes == 0) return JSG_OK;
    if (!src || !dst || bytes < 0 || (bytes & 15) || ((reinterpret_cast<unsigned long long>(src) | reinterpret_cast<unsigned long long>(dst)) & 15))
        return jsg_fail(JSG_ERR_INVALID, "jsg_calib_copy_launch: needs 16-byte aligned pointers and a byte count that is a multiple of 16");
    const long long n4 = bytes / 16, blocks = (n4 + 255) / 256;
    if (blocks >= (1ll << 31)) return jsg_fail(JSG_ERR_UNSUPPORTED, "jsg_calib_copy_launch: more than 2^31 workgroups");
    hipLaunchKernelGGL(calib_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), static_cast<const float*>(src),
                       static_cast<float*>(dst), n4);
    const hipError_t err = hipGetLastError();
    if (err != hipSuccess) return jsg_fail_hip(err, "jsg_calib_copy_launch");
    return JSG_OK;
}

int jsg_colormap_launch(const jsg_colormap_args* g, void* stream) { return colormap_launch_impl(g, nullptr, 0, stream); }

static int colormap_launch_impl(const jsg_colormap_args* g, const unsigned char* idx_in, long long idx_in_pitch, void* stream) {
    if (!g) return jsg_fail(JSG_ERR_INVALID, "jsg_colormap_launch: null argument");
    if (g->n_cols == 0) return JSG_OK;
    if ((!g->db && !idx_in) || !g->lut || g->height <= 0 || g->ring_width <= 0 || g->n_cols < 0 || g->x_wrap <= 0 ||
        g->n_colors <= 0 || g->col_first < 0 || g->x_first < 0 || (!g->argb_out && !g->index_out))
        return jsg_fail(JSG_ERR_INVALID, "jsg_colormap_launch: bad geometry");
    if (g->n_cols > g->ring_width)
        return jsg_fail(JSG_ERR_INVALID, "jsg_colormap_launch: more columns than the ring holds");
    if (g->index_out && g->n_colors > 256)
        return jsg_fail(JSG_ERR_INVALID, "jsg_colormap_launch: the 8-bit index plane needs n_colors <= 256");
    if ((g->argb_out && g->argb_pitch < g->x_wrap) || (g->index_out && g->index_pitch < g->x_wrap))
        return jsg_fail(JSG_ERR_INVALID, "jsg_colormap_launch: image pitch smaller than x_wrap (rows would overlap / leave the image)");
    if (g->n_colors > 65535) return jsg_fail(JSG_ERR_UNSUPPORTED, "jsg_colormap_launch: n_colors > 65535");
    CmapKArgs ka{};
    ka.db = g->db;
    ka.db_pitch = g->db_pitch;
    ka.ring_w = g->ring_width;
    ka.height = g->height;
    ka.col_first = g->col_first % g->ring_width;
    ka.n_cols = g->n_cols;
    ka.x_first = g->x_first % g->x_wrap;
    ka.x_wrap = g->x_wrap;
    ka.lut = g->lut;
    ka.n_colors = g->n_colors;
    ka.vmin = g->vmin;
    ka.vmax = g->vmax;
    ka.top = g->vmax * 0.9999f;
    ka.mult = g->access_mult;
    ka.argb = g->argb_out;
    ka.argb_pitch = g->argb_pitch;
    ka.index = g->index_out;
    ka.index_pitch = g->index_pitch;
    ka.idx_in = idx_in;
    ka.idx_in_pitch = idx_in_pitch;
    dim3 grid((g->n_cols + CM_TILE - 1) / CM_TILE, (g->height + CM_TILE - 1) / CM_TILE);
    hipLaunchKernelGGL(colormap_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), ka);
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) return jsg_fail_hip(err, "jsg_colormap_launch");
    return JSG_OK;
}

}  // extern "C"
