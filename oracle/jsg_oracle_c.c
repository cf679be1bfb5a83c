/* CPU ORACLE (plain C) -- TEST INFRASTRUCTURE ONLY, never linked into the product.
 *
 * A scalar float32 restatement of the reference's frame loop, used (a) by tests as a second, independent
 * checker next to oracle/jsg_oracle.py and (b) by bench.py's cpu_baseline leg ("kind": "port"): it is what
 * the reference's Spectrogram::processSynchronBlock does on the host -- gather frame, window, FFT power, channel
 * mix, 10*log10, ring store -- single-threaded like the plugin's audio thread, or OpenMP over frames.
 *
 *   frame loop + gather        reference Spectrogram.cpp:50-59
 *   window multiply            reference Spectrogram.cpp:137-141
 *   spectrum::power            external in the reference (call site Spectrogram.cpp:144); here: float32
 *                              radix-2 FFT of the N/2-point packed sequence + real split, |X|^2 un-normalised.
 *                              PARITY UNPINNED for its absolute scale (see oracle/jsg_oracle.py).
 *   channel mix                reference Spectrogram.cpp:64-106
 *   10*log10(p + 1e-11f)       reference Spectrogram.cpp:107 (double log10, the canonical overload)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef struct {
    int n;        /* real FFT size */
    float* cs;    /* twiddles of the n/2-point complex FFT: cos, -sin interleaved, n/4 entries */
    float* split; /* exp(-2 pi i k / n), k < n/4+1 ... stored for k <= n/2: cos, sin */
    int* rev;     /* bit reversal of n/2 points */
} rfft_plan;

static rfft_plan* plan_create(int n) {
    rfft_plan* p = (rfft_plan*)malloc(sizeof(rfft_plan));
    const int m = n / 2;
    p->n = n;
    p->cs = (float*)malloc(sizeof(float) * 2 * (m / 2 > 0 ? m / 2 : 1));
    p->split = (float*)malloc(sizeof(float) * 2 * (m + 1));
    p->rev = (int*)malloc(sizeof(int) * m);
    for (int k = 0; k < m / 2; ++k) {
        p->cs[2 * k] = (float)cos(2.0 * M_PI * k / m);
        p->cs[2 * k + 1] = (float)(-sin(2.0 * M_PI * k / m));
    }
    for (int k = 0; k <= m; ++k) {
        p->split[2 * k] = (float)cos(2.0 * M_PI * k / n);
        p->split[2 * k + 1] = (float)(-sin(2.0 * M_PI * k / n));
    }
    int bits = 0;
    while ((1 << bits) < m) ++bits;
    for (int i = 0; i < m; ++i) {
        int r = 0;
        for (int b = 0; b < bits; ++b)
            if (i & (1 << b)) r |= 1 << (bits - 1 - b);
        p->rev[i] = r;
    }
    return p;
}

static void plan_destroy(rfft_plan* p) {
    free(p->cs); free(p->split); free(p->rev); free(p);
}

/* power[k] = |sum_n x[n] exp(-2 pi i k n / N)|^2, k = 0..N/2; work: 2*(N/2) floats */
static void rfft_power(const rfft_plan* p, const float* x, float* power, float* work) {
    const int n = p->n, m = n / 2;
    float* re = work;
    float* im = work + m;
    for (int i = 0; i < m; ++i) {   /* z[i] = x[2i] + i x[2i+1], bit-reversed order */
        re[p->rev[i]] = x[2 * i];
        im[p->rev[i]] = x[2 * i + 1];
    }
    for (int len = 2; len <= m; len <<= 1) {   /* iterative radix-2 DIT */
        const int half = len / 2, step = m / len;
        for (int s = 0; s < m; s += len)
            for (int k = 0; k < half; ++k) {
                const float wr = p->cs[2 * k * step], wi = p->cs[2 * k * step + 1];
                const int a = s + k, b = a + half;
                const float tr = re[b] * wr - im[b] * wi;
                const float ti = re[b] * wi + im[b] * wr;
                re[b] = re[a] - tr; im[b] = im[a] - ti;
                re[a] += tr; im[a] += ti;
            }
    }
    for (int k = 0; k <= m; ++k) {   /* real split */
        const int kk = k == m ? 0 : k, pk = (m - k) % m;
        const float zr = re[kk], zi = im[kk], pr = re[pk], pi = -im[pk];   /* conj Z[M-k] */
        const float er = 0.5f * (zr + pr), ei = 0.5f * (zi + pi);
        const float dr = 0.5f * (zr - pr), di = 0.5f * (zi - pi);          /* (Z - conj Zp)/2 */
        /* O = -i * d ; X = E + W * O */
        const float or_ = di, oi = -dr;
        const float wr = p->split[2 * k], wi = p->split[2 * k + 1];
        const float xr = er + (wr * or_ - wi * oi);
        const float xi = ei + (wr * oi + wi * or_);
        power[k] = xr * xr + xi * xi;
    }
}

/* The plan of the last size used and one set of scratch buffers per thread live across calls (bench.py calls this in
 * a loop; rebuilding twiddles and malloc'ing per call made the all-cores figure meaningless). */
static rfft_plan* g_plan = 0;
static _Thread_local float* t_buf = 0;      /* frame (n) + work (n) + power (h * channels) */
static _Thread_local size_t t_buf_floats = 0;

static float* thread_buffers(size_t floats) {
    if (t_buf_floats < floats) {
        free(t_buf);
        t_buf = (float*)malloc(sizeof(float) * floats);
        t_buf_floats = t_buf ? floats : 0;
    }
    return t_buf;
}

/* mix modes: Spectrogram::ChannelMixMode order (0 AbsMean, 1 Max, 2 Min, 3 Left, 4 Right) */
int jsg_oracle_stft_db(const float* x, int channels, long pitch, int n, int hop, int feedblocks, long n_frames,
                       const float* win, int mix, float power_scale, float* out_db, int threads) {
    const int h = n / 2 + 1;
    if (n < 4 || (n & (n - 1)) || channels < 1) return -1;
    if (!g_plan || g_plan->n != n) {   /* (single caller at a time: tests and bench.py) */
        if (g_plan) plan_destroy(g_plan);
        g_plan = plan_create(n);
    }
    const rfft_plan* plan = g_plan;
    int failed = 0;
    (void)threads;
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
    {
        float* frame = thread_buffers((size_t)2 * n + (size_t)h * channels);
        if (!frame) {
#pragma omp atomic write
            failed = 1;
        }
        float* work = frame ? frame + n : 0;
        float* pw = frame ? work + n : 0;
#pragma omp for schedule(static)
        for (long j = 0; j < n_frames; ++j) {
            if (!frame) continue;
            const long start = (j / feedblocks) * (long)n + (j % feedblocks) * (long)hop;
            for (int c = 0; c < channels; ++c) {
                const float* src = x + (long)c * pitch + start;
                for (int k = 0; k < n; ++k) frame[k] = src[k];          /* Spectrogram.cpp:54-55 */
                for (int k = 0; k < n; ++k) frame[k] *= win[k];          /* :140-141 */
                rfft_power(plan, frame, pw + (size_t)c * h, work);       /* :144 */
                if (power_scale != 1.0f)
                    for (int k = 0; k < h; ++k) pw[(size_t)c * h + k] *= power_scale;
            }
            float* dst = out_db + (size_t)j * h;
            for (int k = 0; k < h; ++k) {                                /* :64-108 */
                float v;
                switch (mix) {
                    case 0: v = 0.0f; for (int c = 0; c < channels; ++c) v += pw[(size_t)c * h + k]; v /= (float)channels; break;
                    case 1: v = 0.0f; for (int c = 0; c < channels; ++c) if (pw[(size_t)c * h + k] > v) v = pw[(size_t)c * h + k]; break;
                    case 2: v = 1000000.0f; for (int c = 0; c < channels; ++c) if (pw[(size_t)c * h + k] < v) v = pw[(size_t)c * h + k]; break;
                    case 3: v = pw[k]; break;
                    default: v = pw[(size_t)(channels > 1 ? 1 : 0) * h + k]; break;
                }
                dst[k] = (float)(10.0 * log10((double)(v + 0.00000000001f)));
            }
        }
    }
    return failed ? -2 : 0;
}

/* CColorPalette::getRGBColor over a block of dB columns (CColorpalette.h:34-45) with the pixel placement of a full recolour
 * in ring order x = column, y = H-1-bin (Spectrogram.cpp:632-648): the CPU side of the C5 end-to-end baseline. */
int jsg_oracle_colour_columns(const float* db, long n_cols, int h, const int* lut, int n_colors, float vmin, float vmax, float mult,
                              unsigned* argb, long argb_pitch, int threads) {
    if (!db || !lut || !argb || h < 1 || n_colors < 1) return -1;
    (void)threads;
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
    for (long c = 0; c < n_cols; ++c)
        for (int k = 0; k < h; ++k) {
            float v = db[(size_t)c * h + k];
            if (v >= vmax) v = vmax * 0.9999f;
            if (v < vmin) v = vmin;
            int idx = (int)((v - vmin) * mult);
            if (idx >= n_colors) idx = n_colors - 1;
            argb[(size_t)(h - 1 - k) * argb_pitch + c] = (unsigned)lut[idx] | 0xFF000000u;
        }
    return 0;
}
