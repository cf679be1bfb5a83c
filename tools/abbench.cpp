// DEV TOOL: A/B timing of several builds of libjsg.so in ONE process, interleaved, without Python.
//
//   abbench [--cfg c2|c3|c5|c4|big|all] [--reps R] [--rounds K] [--streams S] libA.so [libB.so ...]
//
// For every configuration the same seeded input is run through every library; the first library's output is the
// yardstick (max |difference| of the others is printed: a variant that changes results shows up here).  Timing: HIP
// events around R back-to-back launches on ONE stream (in order, the per-kernel view of bench.py's roofline leg),
// rotating over enough distinct input/output batches to defeat the 256 MiB Infinity Cache; K rounds, libraries
// interleaved; median and minimum are printed.  With --streams S > 1 a second figure gives the time per launch when
// the launches are spread over S streams (the throughput view).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../include/jsg.h"

#define CK(call)                                                                                  \
    do {                                                                                          \
        hipError_t e_ = (call);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            std::exit(2);                                                                         \
        }                                                                                         \
    } while (0)

// launch floor: a kernel with the STFT kernel's geometry (512 threads, 36 KB dynamic + 10 KB static LDS) that does nothing,
// and a tuned streaming copy of the same byte count (16-byte loads, non-temporal 16-byte stores)
__global__ __launch_bounds__(512) void ab_null_kernel(float* out, int never) {
    __shared__ float s_tab[2720];
    extern __shared__ float s_dyn[];
    if (never) {   // keeps both LDS objects allocated
        s_tab[threadIdx.x] = 1.f;
        s_dyn[threadIdx.x] = 2.f;
        __syncthreads();
        out[threadIdx.x] = s_tab[threadIdx.x ^ 1] + s_dyn[threadIdx.x ^ 1];
    }
}
typedef float ab_v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void ab_copy_kernel(const ab_v4f* __restrict__ src, ab_v4f* __restrict__ dst, long long n4) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const ab_v4f a = src[i], b = src[i + stride], c = src[i + 2 * stride], e = src[i + 3 * stride];
        __builtin_nontemporal_store(a, &dst[i]);
        __builtin_nontemporal_store(b, &dst[i + stride]);
        __builtin_nontemporal_store(c, &dst[i + 2 * stride]);
        __builtin_nontemporal_store(e, &dst[i + 3 * stride]);
    }
    for (; i < n4; i += stride) __builtin_nontemporal_store(src[i], &dst[i]);
}

template <int THREADS, int STATIC_FLOATS>
__global__ __launch_bounds__(THREADS) void ab_null_geo(float* out, int never) {
    __shared__ float s_tab[STATIC_FLOATS];
    extern __shared__ float s_dyn[];
    if (never) {
        s_tab[threadIdx.x % STATIC_FLOATS] = 1.f;
        s_dyn[threadIdx.x] = 2.f;
        __syncthreads();
        out[threadIdx.x] = s_tab[(threadIdx.x ^ 1) % STATIC_FLOATS] + s_dyn[threadIdx.x ^ 1];
    }
}

struct Lib {
    std::string path;
    void* h = nullptr;
    int (*plan_create)(jsg_plan**, int, const float*, float) = nullptr;
    int (*plan_destroy)(jsg_plan*) = nullptr;
    int (*stft)(const jsg_plan*, const jsg_stft_args*, void*) = nullptr;
    int (*window_build)(int, int, float*) = nullptr;
    int (*cmap)(const jsg_colormap_args*, void*) = nullptr;
    int (*cmap_build)(int, int, int32_t*) = nullptr;
    int (*cmap_range)(int, float, float, float*, float*, float*) = nullptr;
    const char* (*last_error)(const void*) = nullptr;
    // optional (newer builds): fused STFT -> colour launch
    int (*stft_image)(const jsg_plan*, const jsg_stft_image_args*, void*) = nullptr;
    void (*set_stamps)(void*) = nullptr;   // development builds (-DJSG_DEV_VARIANTS -DJSG_X_ABL=3)
};

struct Config {
    const char* name;
    int n, hop, channels, frames, mix;
    bool colour;   // also run the colour loop after the STFT (C5)
};

static const Config kConfigs[] = {
    {"c2", 1024, 512, 1, 4096, JSG_MIX_ABSMEAN, false},    // BASELINE configs[1]
    {"c3", 2048, 512, 8, 4096, JSG_MIX_ABSMEAN, false},    // configs[2]: 8 ch, 75 % overlap, one mixed column per frame
    {"c4", 1024, 512, 8, 4096, JSG_MIX_PER_CHANNEL, false},// configs[3] shard: 8 ch per GPU, per-channel columns
    {"c5", 4096, 512, 2, 1875, JSG_MIX_ABSMEAN, true},     // configs[4]: stereo, 87.5 % overlap, 10 s ring -> ARGB
    {"big", 1024, 512, 1, 65536, JSG_MIX_ABSMEAN, false},  // asymptotic rate of the 1024-point plan
    {"c5wide", 4096, 512, 2, 30000, JSG_MIX_ABSMEAN, true},  // sixteen C5 images' worth of columns in one image: what the one-round launch of C5 costs
    {"c5wide", 4096, 512, 2, 7500, JSG_MIX_ABSMEAN, true},
    {"s1024", 1024, 512, 2, 4096, JSG_MIX_ABSMEAN, false},  // the plugin's stereo bus at 1024 points (mixed kernel instantiation)
    {"s1024", 1024, 512, 2, 32768, JSG_MIX_ABSMEAN, false},
    {"s1024", 1024, 512, 8, 8192, JSG_MIX_ABSMEAN, false},
    {"c3big", 2048, 512, 8, 16384, JSG_MIX_ABSMEAN, false},
    {"n512", 512, 256, 1, 131072, JSG_MIX_ABSMEAN, false},   // single plans (counter passes: tools/pmc_ab.sh r02_n512 n512 ...)
    {"n8192", 8192, 4096, 1, 8192, JSG_MIX_ABSMEAN, false},
    // --cfg sizes: every plan at 50 % and 75 % overlap, mono, launches of 64 Mi samples' worth of frames
    {"sizes", 512, 256, 1, 131072, JSG_MIX_ABSMEAN, false},
    {"sizes", 512, 128, 1, 131072, JSG_MIX_ABSMEAN, false},
    {"sizes", 1024, 512, 1, 65536, JSG_MIX_ABSMEAN, false},
    {"sizes", 1024, 256, 1, 65536, JSG_MIX_ABSMEAN, false},
    {"sizes", 2048, 1024, 1, 32768, JSG_MIX_ABSMEAN, false},
    {"sizes", 2048, 512, 1, 32768, JSG_MIX_ABSMEAN, false},
    {"sizes", 4096, 2048, 1, 16384, JSG_MIX_ABSMEAN, false},
    {"sizes", 4096, 1024, 1, 16384, JSG_MIX_ABSMEAN, false},
    {"sizes", 8192, 4096, 1, 8192, JSG_MIX_ABSMEAN, false},
    {"sizes", 8192, 2048, 1, 8192, JSG_MIX_ABSMEAN, false},
    // --cfg x2048: 32768 FFTs of 2048 points at every overlap / channel count (which of the two 2048-point plans, JSG_2048_PLAN)
    {"x2048", 2048, 1024, 1, 32768, JSG_MIX_ABSMEAN, false},
    {"x2048", 2048, 512, 1, 32768, JSG_MIX_ABSMEAN, false},
    {"x2048", 2048, 256, 1, 32768, JSG_MIX_ABSMEAN, false},
    {"x2048", 2048, 128, 1, 32768, JSG_MIX_ABSMEAN, false},
    {"x2048", 2048, 1024, 2, 16384, JSG_MIX_ABSMEAN, false},
    {"x2048", 2048, 512, 2, 16384, JSG_MIX_ABSMEAN, false},
    {"x2048", 2048, 256, 2, 16384, JSG_MIX_ABSMEAN, false},
    {"x2048", 2048, 128, 2, 16384, JSG_MIX_ABSMEAN, false},
    {"x2048", 2048, 1024, 4, 8192, JSG_MIX_ABSMEAN, false},
    {"x2048", 2048, 512, 4, 8192, JSG_MIX_ABSMEAN, false},
    {"x2048", 2048, 256, 4, 8192, JSG_MIX_ABSMEAN, false},
    {"x2048", 2048, 128, 4, 8192, JSG_MIX_ABSMEAN, false},
    {"x2048", 2048, 1024, 8, 4096, JSG_MIX_ABSMEAN, false},
    {"x2048", 2048, 512, 8, 4096, JSG_MIX_ABSMEAN, false},
    {"x2048", 2048, 256, 8, 4096, JSG_MIX_ABSMEAN, false},
    {"x2048", 2048, 128, 8, 4096, JSG_MIX_ABSMEAN, false},
    // --cfg mid: launches that do not fill the GPU with 8- / 16-frame workgroups (which plan for few-second images?)
    {"mid", 4096, 512, 2, 512, JSG_MIX_ABSMEAN, false},
    {"mid", 4096, 512, 2, 1024, JSG_MIX_ABSMEAN, false},
    {"mid", 4096, 512, 1, 1024, JSG_MIX_ABSMEAN, false},
    {"mid", 2048, 512, 2, 1024, JSG_MIX_ABSMEAN, false},
    {"mid", 2048, 512, 2, 2048, JSG_MIX_ABSMEAN, false},
    {"mid", 2048, 512, 1, 2048, JSG_MIX_ABSMEAN, false},
    // --cfg x4096: 16384 FFTs of 4096 points (which 4096-point plan)
    {"x4096", 4096, 2048, 1, 16384, JSG_MIX_ABSMEAN, false},
    {"x4096", 4096, 512, 1, 16384, JSG_MIX_ABSMEAN, false},
    {"x4096", 4096, 2048, 2, 8192, JSG_MIX_ABSMEAN, false},
    {"x4096", 4096, 512, 2, 8192, JSG_MIX_ABSMEAN, false},
    {"x4096", 4096, 2048, 4, 4096, JSG_MIX_ABSMEAN, false},
    {"x4096", 4096, 512, 4, 4096, JSG_MIX_ABSMEAN, false},
    {"x4096", 4096, 2048, 8, 2048, JSG_MIX_ABSMEAN, false},
    {"x4096", 4096, 512, 8, 2048, JSG_MIX_ABSMEAN, false},
    // --cfg c4sweep: the C4 shard (8 channels per GPU, per-channel columns) at other launch sizes
    {"c4sweep", 1024, 512, 8, 1024, JSG_MIX_PER_CHANNEL, false},
    {"c4sweep", 1024, 512, 8, 16384, JSG_MIX_PER_CHANNEL, false},
};

// GPU-side time per launch with the host taken out: `issue(count)` enqueues `count` launches on `st`; they are captured
// into one hipGraph, which is replayed (one warm replay, then one timed with events).
template <class F>
static double graph_us_per_launch(hipStream_t st, hipEvent_t e0, hipEvent_t e1, int count, F issue, hipGraphExec_t* cache) {
    if (!*cache) {
        hipGraph_t g;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        issue(count);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(cache, g, nullptr, nullptr, 0));
        CK(hipGraphDestroy(g));
    }
    CK(hipGraphLaunch(*cache, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    CK(hipGraphLaunch(*cache, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return double(ms) * 1e3 / count;
}

static double median(std::vector<double> v) {
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

int main(int argc, char** argv) {
    std::string cfg = "c2";
    int reps = 400, rounds = 5, nstreams = 1;
    bool stamps = false;
    int nthreads = 1;
    std::vector<Lib> libs;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if (a == "--cfg" && i + 1 < argc) cfg = argv[++i];
        else if (a == "--reps" && i + 1 < argc) reps = std::atoi(argv[++i]);
        else if (a == "--rounds" && i + 1 < argc) rounds = std::atoi(argv[++i]);
        else if (a == "--streams" && i + 1 < argc) nstreams = std::atoi(argv[++i]);
        else if (a == "--stamps") stamps = true;
        else if (a == "--threads" && i + 1 < argc) nthreads = std::atoi(argv[++i]);
        else {
            Lib l;
            l.path = a;
            libs.push_back(l);
        }
    }
    if (libs.empty()) {
        std::fprintf(stderr, "usage: abbench [--cfg c2|c3|c4|c5|big|c3big|all] [--reps R] [--rounds K] [--streams S] lib.so ...\n");
        return 1;
    }
    for (Lib& l : libs) {
        l.h = dlopen(l.path.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!l.h) {
            std::fprintf(stderr, "dlopen %s: %s\n", l.path.c_str(), dlerror());
            return 1;
        }
#define SYM(field, name) l.field = reinterpret_cast<decltype(l.field)>(dlsym(l.h, name))
        SYM(plan_create, "jsg_plan_create");
        SYM(plan_destroy, "jsg_plan_destroy");
        SYM(stft, "jsg_stft_db_launch");
        SYM(window_build, "jsg_window_build");
        SYM(cmap, "jsg_colormap_launch");
        SYM(cmap_build, "jsg_colormap_build");
        SYM(cmap_range, "jsg_colormap_range");
        SYM(last_error, "jsg_last_error");
        SYM(stft_image, "jsg_stft_image_launch");
        SYM(set_stamps, "jsg_dev_set_stamp_buffer");
#undef SYM
        if (!l.plan_create || !l.stft || !l.window_build || !l.cmap) {
            std::fprintf(stderr, "%s: missing C-ABI symbols\n", l.path.c_str());
            return 1;
        }
    }
    CK(hipSetDevice(0));
    hipStream_t one;
    CK(hipStreamCreateWithFlags(&one, hipStreamNonBlocking));
    std::vector<hipStream_t> streams(size_t(std::max(1, nstreams)));
    for (auto& s : streams) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));

    for (const Config& c : kConfigs) {
        if (cfg != "all" && cfg != c.name) continue;
        const int H = c.n / 2 + 1;
        const int64_t pitch = (H + 31) / 32 * 32;
        const int64_t n_samples = int64_t(c.frames) * c.hop + (c.n - c.hop);
        const int64_t in_pitch = (n_samples + 63) / 64 * 64;
        const int planes = c.mix == JSG_MIX_PER_CHANNEL ? c.channels : 1;
        const size_t in_bytes = size_t(in_pitch) * c.channels * 4, out_bytes = size_t(c.frames) * pitch * planes * 4;
        const int64_t img_pitch = (c.frames + 31) / 32 * 32;
        const size_t img_bytes = c.colour ? size_t(img_pitch) * H * 4 : 0;
        // rotation: distinct batches worth AB_ROT_MB (default 1000 MB, about 4x the 256 MiB Infinity Cache -- with the 300 MB of
        // round 2 a good part of the reads still hit it: bench.py --nbuf sweep, 0.69 -> 0.57 of 8 TB/s at C2)
        const double rot_bytes = (std::getenv("AB_ROT_MB") ? std::atof(std::getenv("AB_ROT_MB")) : 1000.0) * 1e6;
        int nbuf = int(rot_bytes / double(in_bytes + out_bytes + img_bytes)) + 1;
        nbuf = std::max(2, std::min(nbuf, 256));
        while (nstreams > 1 && nbuf % nstreams) ++nbuf;   // a batch always lands on the same stream
        // seeded input: sine + noise per channel (the shape of SURVEY 8d's signal; exact values do not matter here)
        std::vector<float> hx(size_t(in_pitch) * c.channels);
        uint32_t s = 12345u;
        for (int ch = 0; ch < c.channels; ++ch)
            for (int64_t i = 0; i < in_pitch; ++i) {
                s = s * 1664525u + 1013904223u;
                const float u = float(int32_t(s) >> 8) * (1.0f / 8388608.0f);
                hx[size_t(ch) * in_pitch + i] = 0.5f * std::sin(0.0288f * float(ch + 1) * float(i % 100000)) + 0.1f * u;
            }
        std::vector<float*> d_in(nbuf), d_out(nbuf);
        std::vector<uint32_t*> d_img(nbuf, nullptr);
        std::vector<uint8_t*> d_idx(nbuf, nullptr);
        const int64_t idx_pitch = (H + 63) / 64 * 64;
        for (int b = 0; b < nbuf; ++b) {
            CK(hipMalloc(reinterpret_cast<void**>(&d_in[b]), in_bytes));
            CK(hipMalloc(reinterpret_cast<void**>(&d_out[b]), out_bytes));
            CK(hipMemcpy(d_in[b], hx.data(), in_bytes, hipMemcpyHostToDevice));
            if (c.colour) CK(hipMalloc(reinterpret_cast<void**>(&d_img[b]), img_bytes));
            if (c.colour) CK(hipMalloc(reinterpret_cast<void**>(&d_idx[b]), size_t(c.frames) * idx_pitch));
        }
        int32_t* d_lut = nullptr;
        CK(hipMalloc(reinterpret_cast<void**>(&d_lut), 256 * 4));
        const int64_t ffts = int64_t(c.frames) * c.channels;
        const double algo = double(4ll * c.hop * c.channels + 4ll * H * planes + (c.colour ? 4ll * H : 0)) * c.frames;
        std::printf("== %s: n=%d hop=%d ch=%d frames=%d mix=%d nbuf=%d  algorithmic bytes/launch=%.0f%s\n", c.name, c.n, c.hop,
                    c.channels, c.frames, c.mix, nbuf, algo, c.colour ? " (in + dB + ARGB)" : "");

        if (std::getenv("AB_FLOOR")) {   // what this launch size costs with no work / as a plain copy (same buffers, same rotation)
            const long long n4 = (long long)(std::min(in_bytes, out_bytes) / 16);
            const int nwg = std::getenv("AB_FLOOR_WG") ? std::atoi(std::getenv("AB_FLOOR_WG")) : (c.frames + 7) / 8;
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ab_null_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 37 * 1024);
            for (int mode = 0; mode < 2; ++mode) {
                std::vector<double> t;
                hipGraphExec_t ge = nullptr;
                auto issue = [&](int count) {
                    for (int i = 0; i < count; ++i) {
                        const int b = i % nbuf;
                        if (mode == 0) hipLaunchKernelGGL(ab_null_kernel, dim3(nwg), dim3(512), 37 * 1024, one, d_out[b], 0);
                        else hipLaunchKernelGGL(ab_copy_kernel, dim3(2048), dim3(256), 0, one, reinterpret_cast<const ab_v4f*>(d_in[b]),
                                                reinterpret_cast<ab_v4f*>(d_out[b]), n4);
                    }
                };
                for (int r = 0; r < rounds; ++r) t.push_back(graph_us_per_launch(one, e0, e1, reps, issue, &ge));
                CK(hipGraphExecDestroy(ge));
                std::printf("   floor: %-32s graph %7.2f us/launch (min %7.2f)\n", mode == 0 ? "empty kernel, same geometry" : "streaming copy, same bytes",
                            median(t), *std::min_element(t.begin(), t.end()));
                if (nstreams > 1) {   // the same, host-issued round-robin over the streams (what overlapped launches can reach)
                    std::vector<double> tm;
                    for (int r = 0; r < rounds; ++r) {
                        auto issue_ms = [&](int count) {
                            auto body = [&](int t) {
                                (void)hipSetDevice(0);
                                for (int i = t; i < count; i += nthreads) {
                                    const int b = i % nbuf;
                                    hipStream_t st = streams[size_t(b) % streams.size()];
                                    if (mode == 0) hipLaunchKernelGGL(ab_null_kernel, dim3(nwg), dim3(512), 37 * 1024, st, d_out[b], 0);
                                    else hipLaunchKernelGGL(ab_copy_kernel, dim3(2048), dim3(256), 0, st, reinterpret_cast<const ab_v4f*>(d_in[b]),
                                                            reinterpret_cast<ab_v4f*>(d_out[b]), n4);
                                }
                            };
                            if (nthreads <= 1) { body(0); return; }
                            std::vector<std::thread> th;
                            for (int t = 0; t < nthreads; ++t) th.emplace_back(body, t);
                            for (auto& x : th) x.join();
                        };
                        issue_ms(std::min(reps, 200));
                        CK(hipDeviceSynchronize());
                        const auto t0 = std::chrono::steady_clock::now();
                        issue_ms(reps);
                        CK(hipDeviceSynchronize());
                        tm.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps);
                    }
                    std::printf("   floor: %-32s %d streams %7.2f us/launch (min %7.2f)\n", mode == 0 ? "empty kernel, same geometry" : "streaming copy, same bytes",
                                nstreams, median(tm), *std::min_element(tm.begin(), tm.end()));
                }
            }
        }
        if (std::getenv("AB_GEO")) {   // empty kernels: what do workgroup size, workgroup count and LDS allocation cost?
            struct G { const char* what; int blocks, threads, dyn; int which; };
            const G geos[] = {{"512 x 512 thr, 47 KB LDS", 512, 512, 37 * 1024, 0}, {"512 x 512 thr, no LDS", 512, 512, 0, 1},
                              {"2048 x 256 thr, no LDS", 2048, 256, 0, 2},        {"256 x 1024 thr, no LDS", 256, 1024, 0, 3},
                              {"256 x 256 thr, no LDS", 256, 256, 0, 2},          {"1024 x 256 thr, 23 KB LDS", 1024, 256, 18 * 1024, 4},
                              {"1 x 64 thr", 1, 64, 0, 5}};
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ab_null_geo<512, 2720>), hipFuncAttributeMaxDynamicSharedMemorySize, 37 * 1024);
            for (const G& g : geos) {
                std::vector<double> t;
                hipGraphExec_t ge = nullptr;
                auto issue = [&](int count) {
                    for (int i = 0; i < count; ++i) {
                        float* o = d_out[i % nbuf];
                        switch (g.which) {
                            case 0: hipLaunchKernelGGL((ab_null_geo<512, 2720>), dim3(g.blocks), dim3(g.threads), g.dyn, one, o, 0); break;
                            case 1: hipLaunchKernelGGL((ab_null_geo<512, 1>), dim3(g.blocks), dim3(g.threads), g.dyn, one, o, 0); break;
                            case 2: hipLaunchKernelGGL((ab_null_geo<256, 1>), dim3(g.blocks), dim3(g.threads), g.dyn, one, o, 0); break;
                            case 3: hipLaunchKernelGGL((ab_null_geo<1024, 1>), dim3(g.blocks), dim3(g.threads), g.dyn, one, o, 0); break;
                            case 4: hipLaunchKernelGGL((ab_null_geo<256, 1360>), dim3(g.blocks), dim3(g.threads), g.dyn, one, o, 0); break;
                            default: hipLaunchKernelGGL((ab_null_geo<64, 1>), dim3(g.blocks), dim3(g.threads), g.dyn, one, o, 0); break;
                        }
                    }
                };
                for (int r = 0; r < rounds; ++r) t.push_back(graph_us_per_launch(one, e0, e1, reps, issue, &ge));
                CK(hipGraphExecDestroy(ge));
                std::printf("   empty kernel %-28s graph %7.2f us/launch (min %7.2f)\n", g.what, median(t), *std::min_element(t.begin(), t.end()));
            }
        }
        std::vector<jsg_plan*> plans(libs.size(), nullptr);
        std::vector<std::vector<double>> t_inorder(libs.size()), t_multi(libs.size()), t_fused(libs.size());
        std::vector<float> ref_out, got(size_t(c.frames) * pitch * planes);
        std::vector<uint32_t> ref_img, got_img(c.colour ? size_t(img_pitch) * H : 0);
        auto args_for = [&](int b) {
            jsg_stft_args a{};
            a.in = d_in[b];
            a.in_pitch = in_pitch;
            a.channels = c.channels;
            a.hop = c.hop;
            a.feedblocks = c.n / c.hop;
            a.mix_mode = c.mix;
            a.first_frame = 0;
            a.n_frames = c.frames;
            a.out_db = d_out[b];
            a.out_pitch = pitch;
            a.out_channel_pitch = int64_t(c.frames) * pitch;
            a.ring_width = c.frames;
            a.ring_pos = 0;
            return a;
        };
        jsg_colormap_args ca{};
        for (size_t li = 0; li < libs.size(); ++li) {
            Lib& l = libs[li];
            std::vector<float> win(size_t(c.n));
            l.window_build(JSG_WIN_HANN, c.n, win.data());
            if (l.plan_create(&plans[li], c.n, win.data(), 1.0f) != 0) {
                std::fprintf(stderr, "%s: plan_create failed: %s\n", l.path.c_str(), l.last_error ? l.last_error(nullptr) : "?");
                return 1;
            }
            if (c.colour && li == 0) {
                std::vector<int32_t> lut(256);
                l.cmap_build(256, JSG_CM_JADE, lut.data());
                CK(hipMemcpy(d_lut, lut.data(), 256 * 4, hipMemcpyHostToDevice));
                ca.db_pitch = pitch;
                ca.ring_width = c.frames;
                ca.height = H;
                ca.col_first = 0;
                ca.n_cols = c.frames;
                ca.x_first = 0;
                ca.x_wrap = c.frames;
                ca.lut = d_lut;
                ca.n_colors = 256;
                l.cmap_range(256, -50.f, 50.f, &ca.vmin, &ca.vmax, &ca.access_mult);
                ca.argb_pitch = img_pitch;
            }
            // correctness against the first library
            for (int b = 0; b < 1; ++b) {
                CK(hipMemsetAsync(d_out[b], 0, out_bytes, one));   // on the launch stream (`one` does not synchronise with the null stream)
                jsg_stft_args a = args_for(b);
                if (l.stft(plans[li], &a, one) != 0) {
                    std::fprintf(stderr, "%s: launch failed: %s\n", l.path.c_str(), l.last_error ? l.last_error(nullptr) : "?");
                    return 1;
                }
                if (c.colour) {
                    jsg_colormap_args cc = ca;
                    cc.db = d_out[b];
                    cc.argb_out = d_img[b];
                    l.cmap(&cc, one);
                }
                CK(hipStreamSynchronize(one));
                CK(hipMemcpy(got.data(), d_out[b], out_bytes, hipMemcpyDeviceToHost));
                if (c.colour) CK(hipMemcpy(got_img.data(), d_img[b], img_bytes, hipMemcpyDeviceToHost));
            }
            if (c.colour && l.stft_image) {   // fused image of this library against its own two-kernel image
                std::vector<uint32_t> fimg(got_img.size());
                CK(hipMemsetAsync(d_img[1], 0, img_bytes, one));
                jsg_stft_image_args fa{};
                fa.stft = args_for(1);
                fa.stft.in = d_in[0];
                fa.stft.out_db = nullptr;
                fa.colour = ca;
                fa.colour.argb_out = d_img[1];
                fa.index_scratch = d_idx[1];
                fa.index_scratch_pitch = idx_pitch;
                if (l.stft_image(plans[li], &fa, one) != 0) std::fprintf(stderr, "fused launch failed: %s\n", l.last_error(nullptr));
                CK(hipStreamSynchronize(one));
                CK(hipMemcpy(fimg.data(), d_img[1], img_bytes, hipMemcpyDeviceToHost));
                size_t px = 0;
                for (int y = 0; y < H; ++y)
                    for (int x = 0; x < c.frames; ++x) px += fimg[size_t(y) * img_pitch + x] != got_img[size_t(y) * img_pitch + x];
                std::printf("   %-40s fused image vs two-kernel image: %zu differing pixels of %zu\n", l.path.c_str(), px, size_t(H) * c.frames);
            }
            if (li == 0) {
                ref_out = got;
                ref_img = got_img;
            } else {
                double worst = 0;
                size_t nbad = 0;
                for (int p = 0; p < planes; ++p)
                    for (int f = 0; f < c.frames; ++f)
                        for (int k = 0; k < H; ++k) {
                            const size_t i = (size_t(p) * c.frames + f) * pitch + k;
                            const double d = std::fabs(double(got[i]) - double(ref_out[i]));
                            if (!(d <= worst)) worst = d;
                            if (!(d <= 1e-3)) ++nbad;
                        }
                size_t px = 0;
                for (size_t i = 0; i < got_img.size(); ++i) px += got_img[i] != ref_img[i];
                std::printf("   %-40s vs first: max |dB diff| = %.3g, elements > 1e-3 dB: %zu, differing pixels: %zu\n",
                            l.path.c_str(), worst, nbad, px);
            }
        }
        if (stamps) {
            // in-kernel s_memtime stamps of one launch in the middle of a replayed chain (10 values per wave, see stft_db_kernel)
            for (size_t li = 0; li < libs.size(); ++li) {
                Lib& l = libs[li];
                if (!l.set_stamps) continue;
                const size_t nw = size_t((c.frames + 7) / 8) * 8;
                unsigned long long* d_st = nullptr;
                CK(hipMalloc(reinterpret_cast<void**>(&d_st), nw * 10 * 8));
                CK(hipMemset(d_st, 0, nw * 10 * 8));
                l.set_stamps(d_st);
                hipGraphExec_t ge = nullptr;
                graph_us_per_launch(one, e0, e1, 40, [&](int n) { for (int i = 0; i < n; ++i) { jsg_stft_args a = args_for(i % nbuf); l.stft(plans[li], &a, one); } }, &ge);
                CK(hipGraphExecDestroy(ge));
                std::vector<unsigned long long> st(nw * 10);
                CK(hipMemcpy(st.data(), d_st, nw * 10 * 8, hipMemcpyDeviceToHost));
                l.set_stamps(nullptr);
                CK(hipFree(d_st));
                // absolute time line from s_memrealtime (100 MHz); per-wave cycle deltas from s_memtime
                unsigned long long r_first = ~0ull;
                for (size_t w = 0; w < nw; ++w) if (st[w * 10 + 3]) r_first = std::min(r_first, st[w * 10 + 4]);
                std::vector<double> ratio;
                for (size_t w = 0; w < nw; ++w) {
                    const unsigned long long* d = &st[w * 10];
                    if (d[3] && d[5] > d[4] + 100) ratio.push_back(double(d[3] - d[0]) / (double(d[5] - d[4]) / 100.0));
                }
                const double cyc_per_us = ratio.empty() ? 2100.0 : median(ratio);
                const char* names[8] = {"wave start", "before table loads", "before frame loads", "tables in LDS", "after barrier", "frame data arrived", "FFT+dB done", "stores drained"};
                const int idx[8] = {0, 9, 6, 7, 8, 1, 2, 3};
                std::printf("   stamps %s: shader clock ~%.0f cycles/us; times in us since the first wave of the launch started\n", l.path.c_str(), cyc_per_us);
                for (int k = 0; k < 8; ++k) {
                    std::vector<double> v;
                    for (size_t w = 0; w < nw; ++w) {
                        const unsigned long long* d = &st[w * 10];
                        if (!d[3]) continue;
                        const double start = double(d[4] - r_first) / 100.0;
                        v.push_back(start + double(d[idx[k]] - d[0]) / cyc_per_us);
                    }
                    if (v.empty()) continue;
                    std::sort(v.begin(), v.end());
                    auto pc = [&](double q) { return v[size_t(q * double(v.size() - 1))]; };
                    std::printf("      %-20s min %5.2f  p10 %5.2f  median %5.2f  p90 %5.2f  max %5.2f\n", names[k], v.front(), pc(0.1), pc(0.5), pc(0.9), v.back());
                }
                // per-wave phase durations
                const char* pn[4] = {"prologue -> loads issued", "loads issued -> data", "data -> FFT+dB done", "store + drain"};
                const int pa[4] = {0, 6, 1, 2}, pb[4] = {6, 1, 2, 3};
                for (int k = 0; k < 4; ++k) {
                    std::vector<double> v;
                    for (size_t w = 0; w < nw; ++w) {
                        const unsigned long long* d = &st[w * 10];
                        if (d[3]) v.push_back(double(d[pb[k]] - d[pa[k]]) / cyc_per_us);
                    }
                    if (v.empty()) continue;
                    std::sort(v.begin(), v.end());
                    std::printf("      %-26s min %5.2f  median %5.2f  p90 %5.2f  max %5.2f us\n", pn[k], v.front(), v[v.size() / 2], v[size_t(0.9 * double(v.size() - 1))], v.back());
                }
            }
        }
        if (std::getenv("AB_IMGSTAMPS") && c.colour) {
            // builds with -DJSG_DEV_VARIANTS -DJSG_X_IMGSTAMP: per-wave cycle sums of the phases of the one-kernel display path
            for (size_t li = 0; li < libs.size(); ++li) {
                Lib& l = libs[li];
                if (!l.set_stamps || !l.stft_image) continue;
                const size_t nw = 256 * 8;
                unsigned long long* d_st = nullptr;
                CK(hipMalloc(reinterpret_cast<void**>(&d_st), nw * 10 * 8));
                CK(hipMemset(d_st, 0, nw * 10 * 8));
                for (int rep = 0; rep < 3; ++rep) {
                    if (rep == 2) l.set_stamps(d_st);
                    jsg_stft_image_args fa{};
                    fa.stft = args_for(rep % nbuf);
                    fa.stft.out_db = nullptr;
                    fa.colour = ca;
                    fa.colour.db = nullptr;
                    fa.colour.argb_out = d_img[rep % nbuf];
                    l.stft_image(plans[li], &fa, one);
                    CK(hipStreamSynchronize(one));
                }
                l.set_stamps(nullptr);
                std::vector<unsigned long long> st(nw * 10);
                CK(hipMemcpy(st.data(), d_st, nw * 10 * 8, hipMemcpyDeviceToHost));
                CK(hipFree(d_st));
                const char* nm[7] = {"colour index + LDS writes", "wait at barrier 1", "index reads + barrier 2", "palette reads + stores issued", "epilogues", "kernel (cycles)", "between epilogues (FFT rounds)"};
                std::printf("   image stamps %s (cycles per wave, summed over its epilogues; median / p90 over waves)\n", l.path.c_str());
                for (int k = 0; k < 7; ++k) {
                    std::vector<double> v;
                    for (size_t w = 0; w < nw; ++w) if (st[w * 10 + 4]) v.push_back(double(st[w * 10 + k]));
                    if (v.empty()) continue;
                    std::sort(v.begin(), v.end());
                    std::printf("      %-32s median %10.0f  p90 %10.0f  (waves %zu)\n", nm[k], v[v.size() / 2], v[size_t(0.9 * double(v.size() - 1))], v.size());
                }
            }
        }
        auto time_inorder = [&](size_t li, int count, bool fused) {
            Lib& l = libs[li];
            for (int i = 0; i < count; ++i) {
                const int b = i % nbuf;
                jsg_stft_args a = args_for(b);
                if (fused) {
                    jsg_stft_image_args fa{};
                    fa.stft = a;
                    fa.stft.out_db = nullptr;
                    fa.colour = ca;
                    fa.colour.db = nullptr;
                    fa.colour.argb_out = d_img[b];
                    fa.index_scratch = d_idx[b];
                    fa.index_scratch_pitch = idx_pitch;
                    l.stft_image(plans[li], &fa, one);
                } else {
                    l.stft(plans[li], &a, one);
                    if (c.colour) {
                        jsg_colormap_args cc = ca;
                        cc.db = d_out[b];
                        cc.argb_out = d_img[b];
                        l.cmap(&cc, one);
                    }
                }
            }
        };
        std::vector<hipGraphExec_t> gex(libs.size(), nullptr), gex_fused(libs.size(), nullptr);
        const bool eager = std::getenv("AB_EAGER") != nullptr;   // host-issued launches instead of graph replays
        for (int r = 0; r < rounds; ++r) {
            for (size_t li = 0; li < libs.size(); ++li) {
                if (eager) {
                    time_inorder(li, std::min(reps, 100), false);   // warm-up (also settles the clock)
                    CK(hipStreamSynchronize(one));
                    CK(hipEventRecord(e0, one));
                    time_inorder(li, reps, false);
                    CK(hipEventRecord(e1, one));
                    CK(hipEventSynchronize(e1));
                    float ms = 0;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    t_inorder[li].push_back(double(ms) * 1e3 / reps);
                } else {
                    t_inorder[li].push_back(graph_us_per_launch(one, e0, e1, reps, [&](int n) { time_inorder(li, n, false); }, &gex[li]));
                }
                if (c.colour && libs[li].stft_image)
                    t_fused[li].push_back(graph_us_per_launch(one, e0, e1, reps, [&](int n) { time_inorder(li, n, true); }, &gex_fused[li]));
                if (nstreams > 1) {
                    Lib& l = libs[li];
                    auto issue = [&](int count) {   // `nthreads` host threads, thread t issues the launches with i % nthreads == t
                        auto body = [&](int t) {
                            (void)hipSetDevice(0);
                            for (int i = t; i < count; i += nthreads) {
                                const int b = i % nbuf;
                                jsg_stft_args a = args_for(b);
                                a.blocks_per_cu = std::getenv("AB_BPC") ? std::atoi(std::getenv("AB_BPC")) : 1;
                                hipStream_t st = streams[size_t(b) % streams.size()];
                                l.stft(plans[li], &a, st);
                            }
                        };
                        if (nthreads <= 1) { body(0); return; }
                        std::vector<std::thread> th;
                        for (int t = 0; t < nthreads; ++t) th.emplace_back(body, t);
                        for (auto& x : th) x.join();
                    };
                    issue(std::min(reps, 200));
                    CK(hipDeviceSynchronize());
                    const auto t0 = std::chrono::steady_clock::now();
                    issue(reps);
                    CK(hipDeviceSynchronize());
                    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                    t_multi[li].push_back(us / reps);
                }
            }
        }
        for (size_t li = 0; li < libs.size(); ++li) {
            const double med = median(t_inorder[li]), mn = *std::min_element(t_inorder[li].begin(), t_inorder[li].end());
            std::printf("   %-40s %7.2f us/launch (min %7.2f)  %8.1f MFFT/s  %7.1f GB/s  frac(8TB/s) %.3f", libs[li].path.c_str(),
                        med, mn, double(ffts) / med, algo / med / 1e3, algo / med / 1e3 / 8000.0);
            if (!t_fused[li].empty()) {
                const double fm = median(t_fused[li]);
                std::printf("  | fused image %7.2f us", fm);
            }
            if (!t_multi[li].empty()) std::printf("  | %d streams %7.2f us/launch", nstreams, median(t_multi[li]));
            std::printf("\n");
        }
        for (size_t li = 0; li < libs.size(); ++li) {
            if (gex[li]) CK(hipGraphExecDestroy(gex[li]));
            if (gex_fused[li]) CK(hipGraphExecDestroy(gex_fused[li]));
            if (libs[li].plan_destroy) libs[li].plan_destroy(plans[li]);
        }
        for (int b = 0; b < nbuf; ++b) {
            CK(hipFree(d_in[b]));
            CK(hipFree(d_out[b]));
            if (d_img[b]) CK(hipFree(d_img[b]));
            if (d_idx[b]) CK(hipFree(d_idx[b]));
        }
        CK(hipFree(d_lut));
    }
    return 0;
}
