// DEV PROBE: issue rate of v_fma_f32 vs v_pk_fma_f32 (and v_pk_add / v_pk_mul) on gfx950, W waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 tools/probes/valu_rate.hip -o jadespectrogram_amd/build/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned long long* cyc) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float c = 1.0001f, d = 0.5f;
    const v2f pc = {1.0001f, 0.9999f}, pd = {0.5f, 0.25f};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if constexpr (MODE == 0) {   // 8 independent scalar FMAs
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(d));
        } else if constexpr (MODE == 1) {   // 8 independent packed FMAs
            asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                         "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc), "v"(pd));
        } else if constexpr (MODE == 2) {   // packed add with swizzle + negate modifiers
            asm volatile("v_pk_add_f32 %0, %0, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %1, %1, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n"
                         "v_pk_add_f32 %2, %2, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %3, %3, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n"
                         "v_pk_add_f32 %4, %4, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %5, %5, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n"
                         "v_pk_add_f32 %6, %6, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n v_pk_add_f32 %7, %7, %8 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pd));
        } else if constexpr (MODE == 3) {   // scalar add
            asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(d));
        } else {   // packed mul
            asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                         "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pc));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p4.y + p5.x + p5.y + p6.x + p6.y + p7.x + p7.y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int waves_per_simd) {
    float* out; unsigned long long* cyc;
    const int blocks = 256 * waves_per_simd;   // 256 threads = 4 waves = one per SIMD; blocks/256 workgroups per CU
    hipMalloc(&out, sizeof(float) * 256 * blocks); hipMalloc(&cyc, 8);
    const int iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, 100, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c = 0; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double instr_per_simd = double(iters) * 8 * waves_per_simd;
    printf("%-28s waves/SIMD %d: %.2f ms; wave-0 cycles per instruction %.2f; SIMD cycles per instruction at 2.4 GHz %.2f\n", name,
           waves_per_simd, ms, double(c) / (iters * 8.0), ms * 1e-3 * 2.4e9 / instr_per_simd);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", w); run<1>("v_pk_fma_f32", w); run<3>("v_add_f32", w); run<2>("v_pk_add_f32 op_sel+neg", w); run<4>("v_pk_mul_f32", w);
    }
    return 0;
}
