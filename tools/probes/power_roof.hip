// DEV PROBE (not part of the product): how many packed-f32 wave-instructions per second does the card sustain at its power cap?
// Synthetic instruction streams -- pure v_pk_fma_f32, pure v_pk_add_f32, the add / mul / fma mix of the FFT kernels, and that mix with the
// kernels' share of LDS traffic (one 16-byte LDS read or 8-byte write per six vector instructions) -- kept running for some seconds each
// while tools/power_roof.py samples socket power and core clock (hwmon).  The STFT kernels of C3 / C5 run 3.08e11 vector instructions/s at
// 1400 W whatever their occupancy; this probe says where that number sits against what the silicon gives a stream without any dependency.
// build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/probes/power_roof.hip -o tools/variants/libpower_roof.so
#include <hip/hip_runtime.h>
#include <chrono>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// one "block" = 48 vector instructions (mode-dependent) on 16 independent packed accumulators (mode 3: + 9 LDS instructions and the 6 packed adds that consume the reads)
template <int MODE>
__global__ __launch_bounds__(256) __attribute__((target("no-load-store-opt"))) void roof_kernel(float* out, int iters, const v4f* __restrict__ stream, unsigned stream_mask) {
    __shared__ __attribute__((aligned(16))) v2f buf[256 * 6];
    v2f p[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) p[i] = v2f{float(threadIdx.x + i), float(i) * 0.5f};
    const v2f c = {1.0001f, 0.9999f}, d = {0.5f, 0.25f};
    v2f* mine = buf + threadIdx.x;                                   // 8-byte stride per lane: no bank conflicts
    const v4f* mine4 = reinterpret_cast<const v4f*>(buf) + threadIdx.x;   // 16-byte stride per lane
    for (int it = 0; it < iters; ++it) {
        v4f g = {0.f, 0.f, 0.f, 0.f};
        if constexpr (MODE == 4)   // one 16-byte load per lane and block: C3's 8 KB of samples per 354 vector instructions
            g = stream[((unsigned)it * gridDim.x * 256u + blockIdx.x * 256u + threadIdx.x) & stream_mask];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if constexpr (MODE == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(c), "v"(d));
                else if constexpr (MODE == 1) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(p[i]) : "v"(d));
                else {   // 54 % add, 19 % mul, 19 % fma (+ 8 % that the kernels spend on other things: counted as adds here)
                    const int q = (r * 16 + i) % 16;
                    if (q < 10) asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "+v"(p[i]) : "v"(p[(i + 5) & 15]));
                    else if (q < 13) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c));
                    else asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(c), "v"(p[(i + 3) & 15]));
                }
            }
            if constexpr (MODE >= 3) {   // 3 LDS instructions (two 8-byte writes, one 16-byte read) per 18 vector instructions: the kernels' 60 per 354
                mine[(2 * r) * 256] = p[r];
                mine[(2 * r + 1) * 256] = p[r + 3];
                const v4f a = mine4[(r & 1) * 256];
                p[4 + 2 * r] += v2f{a.x, a.y};
                p[5 + 2 * r] += v2f{a.z, a.w};
            }
        }
        if constexpr (MODE == 4) { p[14] += v2f{g.x, g.y}; p[15] += v2f{g.z, g.w}; }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float* g_out = nullptr;
static v4f* g_stream = nullptr;
constexpr unsigned kStreamElems = 1u << 26;   // 1 GiB of 16-byte elements

// keeps the card busy with `mode` at `waves_per_simd` for about `seconds`; returns vector wave-instructions per second (whole card; LDS
// instructions are not counted, as in the kernels' counters)
extern "C" double power_roof_run(int mode, int waves_per_simd, double seconds, int cus) {
    const int blocks = cus * waves_per_simd;   // 256 threads = one wave per SIMD
    if (!g_out) { hipMalloc(&g_out, sizeof(float) * 256 * 256 * 64); hipMalloc(&g_stream, size_t(kStreamElems) * 16); hipMemset(g_stream, 0, size_t(kStreamElems) * 16); }
    const int iters = 4000;                    // 4000 * 48 instructions per wave per launch: a few hundred microseconds
    auto launch = [&]() {
        switch (mode) {
            case 0: hipLaunchKernelGGL(roof_kernel<0>, dim3(blocks), dim3(256), 0, 0, g_out, iters, g_stream, kStreamElems - 1); break;
            case 1: hipLaunchKernelGGL(roof_kernel<1>, dim3(blocks), dim3(256), 0, 0, g_out, iters, g_stream, kStreamElems - 1); break;
            case 2: hipLaunchKernelGGL(roof_kernel<2>, dim3(blocks), dim3(256), 0, 0, g_out, iters, g_stream, kStreamElems - 1); break;
            case 3: hipLaunchKernelGGL(roof_kernel<3>, dim3(blocks), dim3(256), 0, 0, g_out, iters, g_stream, kStreamElems - 1); break;
            default: hipLaunchKernelGGL(roof_kernel<4>, dim3(blocks), dim3(256), 0, 0, g_out, iters, g_stream, kStreamElems - 1); break;
        }
    };
    launch();
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    long launches = 0;
    double el = 0;
    do {
        for (int i = 0; i < 50; ++i) launch();
        hipDeviceSynchronize();
        launches += 50;
        el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    } while (el < seconds);
    return double(launches) * blocks * 4.0 * iters * (mode == 4 ? 56.0 : mode == 3 ? 54.0 : 48.0) / el;   // (mode 3: + the 6 adds that consume the LDS reads)
}
