// DEV TOOL (standalone, no libjsg): what the HBM of THIS box delivers to a streaming kernel, swept over the knobs a
// kernel author controls -- bytes per lane, grid shape, unroll, cache policy of loads and stores, launch size.
//
//   hipcc -O3 --offload-arch=gfx950 tools/copy_roof.hip -o tools/variants/copy_roof && tools/variants/copy_roof [quick]
//
// It answers the round-3 verdict's question "the builder's tuned float4 copy reaches 0.63 of 8 TB/s on 268 MB while
// MI355X_MICROARCH.md records 6.29 TB/s (0.79) for a float4 copy -- why?" and gives bench.py's `peak_copy_GBps` a
// measured yardstick.  Every figure is (bytes read + bytes written) / time, the buffers rotate over >= 2 GB so that the
// 256 MiB Infinity Cache does not serve them, timing = HIP events around R back-to-back launches on one stream.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(call)                                                                                      \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            std::exit(2);                                                                             \
        }                                                                                             \
    } while (0)

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

// POL: 0 = plain loads + plain stores, 1 = plain loads + nt stores, 2 = nt loads + nt stores, 3 = nt loads + plain stores
template <class T, int U, int POL>
__global__ __launch_bounds__(256) void copy_gs(const T* __restrict__ src, T* __restrict__ dst, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = (POL >= 2) ? __builtin_nontemporal_load(&src[i + u * stride]) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (POL == 1 || POL == 2) __builtin_nontemporal_store(v[u], &dst[i + u * stride]);
            else dst[i + u * stride] = v[u];
        }
    }
    for (; i < n; i += stride) dst[i] = src[i];
}

// every workgroup streams through ONE contiguous chunk (what "a wavefront takes consecutive frames" does to the DRAM pattern)
template <class T, int U, int POL>
__global__ __launch_bounds__(256) void copy_chunk(const T* __restrict__ src, T* __restrict__ dst, long long n) {
    const long long per = (n + gridDim.x - 1) / gridDim.x;
    const long long b = (long long)blockIdx.x * per, e = b + per < n ? b + per : n;
    long long i = b + threadIdx.x;
    for (; i + (U - 1) * 256 < e; i += U * 256) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = (POL >= 2) ? __builtin_nontemporal_load(&src[i + u * 256]) : src[i + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (POL == 1 || POL == 2) __builtin_nontemporal_store(v[u], &dst[i + u * 256]);
            else dst[i + u * 256] = v[u];
        }
    }
    for (; i < e; i += 256) dst[i] = src[i];
}

template <class T, int U>
__global__ __launch_bounds__(256) void read_gs(const T* __restrict__ src, float* __restrict__ sink, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    float acc = 0.f;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x;
    }
    if (acc == 12345.678f) sink[0] = acc;
}

template <class T, int U, int NT>
__global__ __launch_bounds__(256) void fill_gs(T* __restrict__ dst, long long n, float val) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    T v;
    for (int k = 0; k < (int)(sizeof(T) / 4); ++k) v[k] = val;
    for (; i < n; i += stride) {
        if (NT) __builtin_nontemporal_store(v, &dst[i]);
        else dst[i] = v;
    }
}

// the STFT kernel's own shape, without arithmetic: a wave reads one 4 KB frame as 8 x 512 contiguous bytes (8 B per lane) -- of
// which the first half overlaps the previous frame -- and writes one 513-float column as 2 x 8 runs of 256 bytes (4 B per lane) plus
// bin 512.  Knobs: OV 1 = re-read the overlapped half as the kernel does, 0 = read once; W16 = 16-byte stores; NTL = non-temporal
// loads; TAIL = store bin N/2 (4 bytes in a 128-byte line of their own); RUN = consecutive frames per wave (1: the workgroup's eight
// waves take eight neighbouring frames; R > 1: every wave walks R neighbouring frames, the workgroup 8 R); LDSKB = dynamic LDS per
// workgroup (occupancy of the real kernel: 47 KB -> three 8-wave workgroups per CU).
template <int OV, int W16, int NTL, int TAIL, int RUN>
__global__ __launch_bounds__(512) void stft_shape(const float* __restrict__ in, float* __restrict__ out, long long n_frames, long long out_pitch) {
    extern __shared__ float s_dyn[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;
    if (n_frames < 0) s_dyn[threadIdx.x] = 1.f;
    for (long long g = blockIdx.x; g * 8 * RUN < n_frames; g += gridDim.x) {
        v2f r[8];
#pragma unroll
        for (int k = 0; k < RUN; ++k) {
            const long long f = g * 8 * RUN + wave * RUN + k;
            if (f >= n_frames) break;
            const v2f* src = reinterpret_cast<const v2f*>(in + f * 512) + lane;
            const bool fresh = OV || k == 0;
            if (!fresh) {
#pragma unroll
                for (int m = 0; m < 4; ++m) r[m] = r[m + 4];
            }
#pragma unroll
            for (int m = 0; m < 8; ++m)
                if (fresh || m >= 4) r[m] = NTL ? __builtin_nontemporal_load(&src[64 * m]) : src[64 * m];
            float* dst = out + f * out_pitch;
            if (W16) {   // 16-byte stores: a lane writes 4 consecutive floats, 1 KB per instruction
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    v4f v = {r[4 * m].x, r[4 * m + 1].x, r[4 * m + 2].y, r[4 * m + 3].y};
                    __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(dst) + lane + 64 * m);
                }
            } else {
#pragma unroll
                for (int m = 0; m < 8; ++m) __builtin_nontemporal_store(m < 4 ? r[m].x : r[m].y, &dst[lane + 64 * m]);
            }
            if (TAIL) __builtin_nontemporal_store(r[0].y, &dst[512]);
            acc += r[1].y;
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

// The same byte movement with the knobs that decide how tight the chip-wide access front stays:
//   WAVES  wavefronts per workgroup (each takes one frame per step: the workgroup WAVES neighbouring frames)
//   DYN    1: a workgroup fetches the index of its next group from a global counter (one step ahead), like the hardware
//             dispatcher hands out workgroups of a non-looping grid; 0: static grid-stride
//   STAGE  1: the input span of a step is brought into LDS once by LDS-DMA (16 bytes per lane, every byte once, two steps
//             in flight), the waves read their frames from LDS (overlap served there) -- the "loader + consumers" shape
template <int WAVES, int DYN, int STAGE>
__global__ __launch_bounds__(WAVES * 64) void stft_shape2(const float* __restrict__ in, float* __restrict__ out, long long n_frames, long long out_pitch,
                                                          unsigned* __restrict__ counter) {
    extern __shared__ __attribute__((aligned(16))) float s_dyn[];
    __shared__ unsigned s_next[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long n_groups = (n_frames + WAVES - 1) / WAVES;
    constexpr int SPAN = (WAVES + 1) * 512;   // floats of one step's input span (WAVES frames of hop 512 + the last one's second half)
    float acc = 0.f;
    auto stage = [&](long long g, int buf) {   // span of group g -> s_dyn[buf * SPAN ..]: SPAN * 4 / 1024 pieces of 1 KB, dealt round-robin to the waves
        const char* gsrc = reinterpret_cast<const char*>(in + g * WAVES * 512);
        char* l = reinterpret_cast<char*>(s_dyn + buf * SPAN);
        for (int p = wave; p < SPAN * 4 / 1024; p += WAVES)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + p * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(l + p * 1024), 16, 0, 0);
    };
    long long g = blockIdx.x, g_next;
    if (DYN) {
        if (threadIdx.x == 0) s_next[0] = atomicAdd(counter, 1u) + gridDim.x;
    }
    g_next = g + gridDim.x;
    if (STAGE && g < n_groups) stage(g, 0);
    for (int it = 0; g < n_groups; ++it) {
        if (DYN) {
            __syncthreads();
            g_next = s_next[it & 1];
            if (threadIdx.x == 0) s_next[(it + 1) & 1] = atomicAdd(counter, 1u) + gridDim.x;
        }
        if (STAGE) {
            if (g_next < n_groups) stage(g_next, (it + 1) & 1);
            // the pieces of THIS step are older than the ones just issued: a counted wait leaves the next step in flight
            // (timing probe: behind them sit the nine column stores of the previous step, then the new pieces -- counted roughly, data is not checked)
            if (g_next < n_groups) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((SPAN * 4 / 1024 + WAVES - 1) / WAVES + 9) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        const long long f = g * WAVES + wave;
        if (f < n_frames) {
            v2f r[8];
            if (STAGE) {
                const v2f* l = reinterpret_cast<const v2f*>(s_dyn + (it & 1) * SPAN + wave * 512) + lane;
#pragma unroll
                for (int m = 0; m < 8; ++m) r[m] = l[64 * m];
            } else {
                const v2f* src = reinterpret_cast<const v2f*>(in + f * 512) + lane;
#pragma unroll
                for (int m = 0; m < 8; ++m) r[m] = src[64 * m];
            }
            float* dst = out + f * out_pitch;
#pragma unroll
            for (int m = 0; m < 8; ++m) __builtin_nontemporal_store(m < 4 ? r[m].x : r[m].y, &dst[lane + 64 * m]);
            __builtin_nontemporal_store(r[0].y, &dst[512]);
            acc += r[1].y;
        }
        if (STAGE) __syncthreads();   // the buffer of this step is refilled two steps later: everyone has read it
        g = g_next;
        if (!DYN) g_next = g + gridDim.x;
    }
    if (acc == 12345.678f) out[0] = acc;
}

struct Timer {
    hipEvent_t a, b;
    Timer() { CK(hipEventCreate(&a)); CK(hipEventCreate(&b)); }
    template <class F>
    double us_per(F&& launch, int reps, int warm = 3) {
        for (int i = 0; i < warm; ++i) launch(i);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(a, 0));
        for (int i = 0; i < reps; ++i) launch(i);
        CK(hipEventRecord(b, 0));
        CK(hipEventSynchronize(b));
        float ms = 0.f;
        CK(hipEventElapsedTime(&ms, a, b));
        return double(ms) * 1e3 / reps;
    }
};

int main(int argc, char** argv) {
    const bool quick = argc > 1 && !std::strcmp(argv[1], "quick");
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    std::printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"mem_clock_mhz\": %d}\n", prop.name, prop.multiProcessorCount,
                prop.clockRate / 1000, prop.memoryClockRate / 1000);
    const long long POOL = 2ll << 30;   // 2 GiB source pool + 2 GiB destination pool
    char *src = nullptr, *dst = nullptr;
    CK(hipMalloc(&src, POOL));
    CK(hipMalloc(&dst, POOL));
    CK(hipMemset(src, 1, POOL));
    CK(hipMemset(dst, 0, POOL));
    Timer T;
    const int cus = prop.multiProcessorCount;

    const bool shapes_only = argc > 1 && !std::strcmp(argv[1], "shapes");
    // ---- 1. float4 grid-stride copy: size x grid x unroll x policy ----
    const long long sizes[] = {8396800ll, 134348800ll, 536870912ll, 1073741824ll, POOL};   // bytes each way (C2 launch: 8.4 MB; 65 536 frames: 134 MB)
    for (long long bytes : sizes) {
        if (shapes_only) break;
        const long long n4 = bytes / 16;
        const int nrot = int(std::max(1ll, POOL / bytes));
        const int reps = int(std::max(4ll, std::min(400ll, (8ll << 30) / bytes)));
        double best = 1e30;
        char bestname[128] = "";
        for (int bpc : {1, 2, 4, 8, 16, 32, 0}) {   // workgroups per CU; 0 = one thread per element
            const long long full = (n4 + 255) / 256;
            const int grid = bpc ? int(std::min<long long>(full, (long long)cus * bpc)) : int(std::min<long long>(full, 1ll << 30));
            for (int pol = 0; pol < 4; ++pol) {
                for (int U : {1, 2, 4, 8}) {
                    if (quick && (U == 2 || pol == 3)) continue;
                    if (!bpc && U != 1) continue;
                    auto launch = [&](int i) {
                        const v4f* s = reinterpret_cast<const v4f*>(src + (long long)(i % nrot) * bytes);
                        v4f* d = reinterpret_cast<v4f*>(dst + (long long)(i % nrot) * bytes);
#define GO(UU, PP) hipLaunchKernelGGL((copy_gs<v4f, UU, PP>), dim3(grid), dim3(256), 0, 0, s, d, n4)
#define GOU(PP) do { if (U == 1) GO(1, PP); else if (U == 2) GO(2, PP); else if (U == 4) GO(4, PP); else GO(8, PP); } while (0)
                        if (pol == 0) GOU(0); else if (pol == 1) GOU(1); else if (pol == 2) GOU(2); else GOU(3);
                    };
                    const double us = T.us_per(launch, reps);
                    const double tbs = 2.0 * bytes / us / 1e6;
                    std::printf("{\"kernel\": \"copy_gs_f4\", \"bytes_each_way\": %lld, \"wg_per_cu\": %d, \"grid\": %d, \"unroll\": %d, \"policy\": %d, \"us\": %.2f, \"TBps\": %.3f}\n",
                                bytes, bpc, grid, U, pol, us, tbs);
                    if (us < best) { best = us; std::snprintf(bestname, sizeof bestname, "wg_per_cu=%d unroll=%d policy=%d", bpc, U, pol); }
                }
            }
        }
        std::printf("{\"best\": \"copy_gs_f4\", \"bytes_each_way\": %lld, \"us\": %.2f, \"TBps\": %.3f, \"frac_of_8\": %.3f, \"how\": \"%s\"}\n", bytes, best,
                    2.0 * bytes / best / 1e6, 2.0 * bytes / best / 8e6, bestname);
        std::fflush(stdout);
    }
    // ---- 2. chunked traversal, narrower accesses, read-only, write-only (1 GiB each way) ----
    if (!shapes_only) {
        const long long bytes = 1ll << 30;
        const int nrot = 2, reps = 8;
        for (int bpc : {4, 8, 16}) {
            const int grid = cus * bpc;
            auto L1 = [&](int i) { hipLaunchKernelGGL((copy_chunk<v4f, 4, 1>), dim3(grid), dim3(256), 0, 0, reinterpret_cast<const v4f*>(src + (long long)(i % nrot) * bytes), reinterpret_cast<v4f*>(dst + (long long)(i % nrot) * bytes), bytes / 16); };
            double us = T.us_per(L1, reps);
            std::printf("{\"kernel\": \"copy_chunk_f4_u4_ntst\", \"wg_per_cu\": %d, \"us\": %.2f, \"TBps\": %.3f}\n", bpc, us, 2.0 * bytes / us / 1e6);
            auto L2 = [&](int i) { hipLaunchKernelGGL((copy_gs<v2f, 4, 1>), dim3(grid), dim3(256), 0, 0, reinterpret_cast<const v2f*>(src + (long long)(i % nrot) * bytes), reinterpret_cast<v2f*>(dst + (long long)(i % nrot) * bytes), bytes / 8); };
            us = T.us_per(L2, reps);
            std::printf("{\"kernel\": \"copy_gs_f2_u4_ntst\", \"wg_per_cu\": %d, \"us\": %.2f, \"TBps\": %.3f}\n", bpc, us, 2.0 * bytes / us / 1e6);
            auto L2b = [&](int i) { hipLaunchKernelGGL((copy_gs<v2f, 8, 1>), dim3(grid), dim3(256), 0, 0, reinterpret_cast<const v2f*>(src + (long long)(i % nrot) * bytes), reinterpret_cast<v2f*>(dst + (long long)(i % nrot) * bytes), bytes / 8); };
            us = T.us_per(L2b, reps);
            std::printf("{\"kernel\": \"copy_gs_f2_u8_ntst\", \"wg_per_cu\": %d, \"us\": %.2f, \"TBps\": %.3f}\n", bpc, us, 2.0 * bytes / us / 1e6);
            auto L3 = [&](int i) { hipLaunchKernelGGL((read_gs<v4f, 4>), dim3(grid), dim3(256), 0, 0, reinterpret_cast<const v4f*>(src + (long long)(i % nrot) * bytes), reinterpret_cast<float*>(dst), bytes / 16); };
            us = T.us_per(L3, reps);
            std::printf("{\"kernel\": \"read_only_f4_u4\", \"wg_per_cu\": %d, \"us\": %.2f, \"TBps\": %.3f}\n", bpc, us, 1.0 * bytes / us / 1e6);
            auto L4 = [&](int i) { hipLaunchKernelGGL((fill_gs<v4f, 1, 1>), dim3(grid), dim3(256), 0, 0, reinterpret_cast<v4f*>(dst + (long long)(i % nrot) * bytes), bytes / 16, 1.f); };
            us = T.us_per(L4, reps);
            std::printf("{\"kernel\": \"write_only_f4_nt\", \"wg_per_cu\": %d, \"us\": %.2f, \"TBps\": %.3f}\n", bpc, us, 1.0 * bytes / us / 1e6);
            auto L5 = [&](int i) { hipLaunchKernelGGL((fill_gs<v4f, 1, 0>), dim3(grid), dim3(256), 0, 0, reinterpret_cast<v4f*>(dst + (long long)(i % nrot) * bytes), bytes / 16, 1.f); };
            us = T.us_per(L5, reps);
            std::printf("{\"kernel\": \"write_only_f4_plain\", \"wg_per_cu\": %d, \"us\": %.2f, \"TBps\": %.3f}\n", bpc, us, 1.0 * bytes / us / 1e6);
        }
        auto M = [&](int i) { CK(hipMemcpyAsync(dst + (long long)(i % nrot) * bytes, src + (long long)(i % nrot) * bytes, bytes, hipMemcpyDeviceToDevice, 0)); };
        double us = T.us_per(M, reps);
        std::printf("{\"kernel\": \"hipMemcpyAsync_d2d\", \"us\": %.2f, \"TBps\": %.3f}\n", us, 2.0 * bytes / us / 1e6);
        std::fflush(stdout);
    }
    // ---- 3. the STFT kernel's own access shape without arithmetic: 245 760 frames (60 batches x 4096), 512-thread workgroups ----
    {
        const long long frames = 245760, pitch = 544;
        const long long in_bytes = (frames * 512 + 512) * 4, out_bytes = frames * pitch * 4;
        const int nrot = int(POOL / std::max(in_bytes, out_bytes));
        const double algo = double(frames) * 4100.0;
#define SHAPE(OV, W16, NTL, TAIL, RUN, LDSKB)                                                                                                  \
        for (int bpc : {1, 2, 3, 8}) {                                                                                                        \
            if (LDSKB * bpc > 160) continue;                                                                                                  \
            const int grid = cus * bpc;                                                                                                       \
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stft_shape<OV, W16, NTL, TAIL, RUN>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)); \
            auto L = [&](int i) { hipLaunchKernelGGL((stft_shape<OV, W16, NTL, TAIL, RUN>), dim3(grid), dim3(512), LDSKB * 1024, 0, reinterpret_cast<const float*>(src + (long long)(i % nrot) * in_bytes), reinterpret_cast<float*>(dst + (long long)(i % nrot) * out_bytes), frames, pitch); }; \
            const double us = T.us_per(L, 12);                                                                                                \
            std::printf("{\"kernel\": \"stft_shape\", \"reread_overlap\": %d, \"store16\": %d, \"nt_loads\": %d, \"tail\": %d, \"run\": %d, \"lds_kb\": %d, \"wg_per_cu\": %d, \"us\": %.2f, \"algorithmic_TBps\": %.3f, \"frac_of_8\": %.3f}\n", \
                        OV, W16, NTL, TAIL, RUN, LDSKB, bpc, us, algo / us / 1e6, algo / us / 8e6);                                            \
        }
        SHAPE(1, 0, 0, 1, 1, 0)    // the kernel as it is
        SHAPE(1, 0, 0, 1, 1, 47)   // ... at its real occupancy
        SHAPE(1, 0, 0, 0, 1, 0)    // without the 4-byte tail line
        SHAPE(1, 0, 1, 1, 1, 0)    // nt loads (the overlap is then read twice from memory?)
        SHAPE(0, 0, 0, 1, 2, 0)    // runs of 2, 4, 8 frames per wave, overlap kept in registers
        SHAPE(0, 0, 0, 1, 4, 0)
        SHAPE(0, 0, 0, 1, 8, 0)
        SHAPE(0, 0, 1, 1, 4, 0)    // ... with nt loads
        SHAPE(0, 0, 1, 1, 8, 0)
        SHAPE(0, 0, 1, 1, 8, 47)
        SHAPE(0, 1, 1, 1, 8, 0)    // ... and 16-byte stores
        SHAPE(0, 1, 1, 0, 8, 0)    // ... and no tail
        SHAPE(0, 0, 1, 0, 8, 0)
        std::fflush(stdout);
    }
    // ---- 4. how tight the access front is: waves per workgroup x workgroups per CU x dynamic group hand-out x LDS staging ----
    {
        const long long frames = 245760, pitch = 544;
        const long long in_bytes = (frames * 512 + 512) * 4, out_bytes = frames * pitch * 4;
        const int nrot = int(POOL / std::max(in_bytes, out_bytes));
        const double algo = double(frames) * 4100.0;
        unsigned* counter = nullptr;
        CK(hipMalloc(&counter, 4));
#define SHAPE2(WAVES, DYN, STAGE)                                                                                                             \
        for (int bpc : {1, 2, 3, 4, 6}) {                                                                                                     \
            if (bpc * WAVES > 32) continue;                                                                                                   \
            const int grid = cus * bpc;                                                                                                       \
            const int lds = STAGE ? 2 * (WAVES + 1) * 2048 : 0;                                                                               \
            CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stft_shape2<WAVES, DYN, STAGE>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024)); \
            auto L = [&](int i) {                                                                                                             \
                if (DYN) CK(hipMemsetAsync(counter, 0, 4, 0));                                                                                \
                hipLaunchKernelGGL((stft_shape2<WAVES, DYN, STAGE>), dim3(grid), dim3(WAVES * 64), lds, 0, reinterpret_cast<const float*>(src + (long long)(i % nrot) * in_bytes), reinterpret_cast<float*>(dst + (long long)(i % nrot) * out_bytes), frames, pitch, counter); }; \
            const double us = T.us_per(L, 12);                                                                                                \
            std::printf("{\"kernel\": \"stft_shape2\", \"waves_per_wg\": %d, \"dynamic\": %d, \"lds_staged\": %d, \"wg_per_cu\": %d, \"waves_per_cu\": %d, \"us\": %.2f, \"algorithmic_TBps\": %.3f, \"frac_of_8\": %.3f}\n", \
                        WAVES, DYN, STAGE, bpc, bpc * WAVES, us, algo / us / 1e6, algo / us / 8e6);                                           \
        }
        SHAPE2(4, 0, 0)
        SHAPE2(8, 0, 0)
        SHAPE2(16, 0, 0)
        SHAPE2(8, 1, 0)
        SHAPE2(16, 1, 0)
        SHAPE2(4, 0, 1)
        SHAPE2(8, 0, 1)
        SHAPE2(16, 0, 1)
        SHAPE2(8, 1, 1)
        SHAPE2(16, 1, 1)
        std::fflush(stdout);
    }
    CK(hipFree(src));
    CK(hipFree(dst));
    return 0;
}
