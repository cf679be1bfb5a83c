// DEV TOOL (standalone, no libjsg): how the ORDER in which workgroup steps are handed out shapes the rate of the C2 access pattern.
//
//   hipcc -O3 --offload-arch=gfx950 tools/front_probe.hip -o tools/variants/front_probe && tools/variants/front_probe
//
// The byte movement of the 1024-point kernel without its arithmetic (four-wave workgroups: a step = four neighbouring frames of hop 512,
// each wave reads its 4 KB frame with 8-byte loads one step ahead and writes one 513-float column), 262 144 frames per launch, rotating over
// 2 GiB pools.  Variants of the step -> workgroup assignment:
//   MAP 0  static grid-stride, workgroup b takes steps b, b + grid, ...                       (neighbours on different XCDs)
//   MAP 1  ... with every XCD given one contiguous eighth of the grid                          (the kernel's remap of rounds 2-4)
//   MAP 2  ... with chunks of 32 workgroups alternating between the XCDs                       (the kernel's remap now)
//   (map 4 in the output: MAP 2 with 16-byte loads -- 1 KB per wave-instruction, four per frame -- which the FFT's lane layout does not allow:
//          what the 8-byte loads cost)
//   MAP 3  dynamic: eight ticket counters, one per XCD; a workgroup draws its next step one step ahead; ticket t of XCD x is step
//          256 (t / 32) + 32 x + t % 32 -- the chip-wide front then advances as ONE contiguous window no matter how the workgroups drift
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(call)                                                                                      \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) {                                                                       \
            std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            std::exit(2);                                                                             \
        }                                                                                             \
    } while (0)

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MAP, int W16 = 0>
__global__ __launch_bounds__(256) void shape4(const float* __restrict__ in, float* __restrict__ out, unsigned n_steps, long long out_pitch,
                                              unsigned* __restrict__ tickets) {
    extern __shared__ float s_dyn[];   // (occupancy knob only)
    __shared__ unsigned s_next[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned nblk = gridDim.x, b = blockIdx.x, xcd = b & 7, jb = b >> 3;
    if (out_pitch < 0) s_dyn[threadIdx.x] = 1.f;
    unsigned lb = b;
    if (MAP == 1) {
        const unsigned q = nblk >> 3, r = nblk & 7;
        lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + jb;
    } else if (MAP == 2 && b < (nblk & ~255u)) {
        lb = ((jb >> 5) << 8) + (xcd << 5) + (jb & 31u);
    }
    auto step_of_ticket = [&](unsigned t) { return ((t >> 5) << 8) + (xcd << 5) + (t & 31u); };
    unsigned g, g_next;
    if (MAP == 3) {
        if (threadIdx.x == 0) {
            s_next[0] = step_of_ticket(atomicAdd(&tickets[xcd * 32], 1u));   // (counters 128 bytes apart)
            s_next[1] = step_of_ticket(atomicAdd(&tickets[xcd * 32], 1u));
        }
        __syncthreads();
        g = s_next[0];
        g_next = s_next[1];
    } else {
        g = lb;
        g_next = lb + nblk;
    }
    float acc = 0.f;
    v2f r[8];
    if (g < n_steps) {
        const v2f* src = reinterpret_cast<const v2f*>(in + ((long long)g * 4 + wave) * 512) + lane;
        if (W16) {   // 16-byte loads: 1 KB per wave-instruction, four per frame
            const v4f* s4 = reinterpret_cast<const v4f*>(in + ((long long)g * 4 + wave) * 512) + lane;
#pragma unroll
            for (int m = 0; m < 4; ++m) { const v4f q = s4[64 * m]; r[2 * m] = v2f{q.x, q.y}; r[2 * m + 1] = v2f{q.z, q.w}; }
        } else {
#pragma unroll
            for (int m = 0; m < 8; ++m) r[m] = src[64 * m];
        }
    }
    for (int it = 0; g < n_steps; ++it) {
        unsigned g_nn = g_next + nblk;
        if (MAP == 3) {
            __syncthreads();                      // s_next[it & 1] has been consumed by everyone (it was read into g / g_next one step ago)
            if (threadIdx.x == 0) s_next[it & 1] = step_of_ticket(atomicAdd(&tickets[xcd * 32], 1u));
        }
        v2f cur[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) cur[m] = r[m];
        if (g_next < n_steps) {                   // next step's frame travels while this one is "transformed"
            const v2f* src = reinterpret_cast<const v2f*>(in + ((long long)g_next * 4 + wave) * 512) + lane;
            if (W16) {
                const v4f* s4 = reinterpret_cast<const v4f*>(in + ((long long)g_next * 4 + wave) * 512) + lane;
#pragma unroll
                for (int m = 0; m < 4; ++m) { const v4f q = s4[64 * m]; r[2 * m] = v2f{q.x, q.y}; r[2 * m + 1] = v2f{q.z, q.w}; }
            } else {
#pragma unroll
                for (int m = 0; m < 8; ++m) r[m] = src[64 * m];
            }
        }
        float* dst = out + ((long long)g * 4 + wave) * out_pitch;
#pragma unroll
        for (int m = 0; m < 8; ++m) __builtin_nontemporal_store(m < 4 ? cur[m].x : cur[m].y, &dst[lane + 64 * m]);
        __builtin_nontemporal_store(cur[0].y, &dst[512]);
        acc += cur[1].y;
        if (MAP == 3) {
            __syncthreads();
            g_nn = s_next[it & 1];
        }
        g = g_next;
        g_next = g_nn;
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const long long POOL = 2ll << 30;
    char *src = nullptr, *dst = nullptr;
    unsigned* tickets = nullptr;
    CK(hipMalloc(&src, POOL));
    CK(hipMalloc(&dst, POOL));
    CK(hipMalloc(&tickets, 8 * 128));
    CK(hipMemset(src, 1, POOL));
    CK(hipMemset(dst, 0, POOL));
    const long long frames = 262144, pitch = 544;
    const unsigned n_steps = unsigned(frames / 4);
    const long long in_bytes = (frames * 512 + 512) * 4, out_bytes = frames * pitch * 4;
    const int nrot = int(POOL / (in_bytes > out_bytes ? in_bytes : out_bytes));
    const double algo = double(frames) * 4100.0;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int rounds = argc > 1 ? std::atoi(argv[1]) : 3;
    for (int round = 0; round < rounds; ++round)
        for (int ldskb : {29, 38}) {              // 5 / 4 workgroups per CU
            for (int map = 0; map < 5; ++map)
                for (int bpc : {4, 5, 8}) {
                    if (map == 3 && bpc == 8) continue;          // dynamic: resident workgroups only
                    if (bpc * ldskb > 160 && map == 3) continue;
                    const int grid = cus * bpc;
                    auto launch = [&](int i) {
                        const float* s = reinterpret_cast<const float*>(src + (long long)(i % nrot) * in_bytes);
                        float* d = reinterpret_cast<float*>(dst + (long long)(i % nrot) * out_bytes);
                        if (map == 3) CK(hipMemsetAsync(tickets, 0, 8 * 128, 0));
                        switch (map) {
                            case 0: hipLaunchKernelGGL(shape4<0>, dim3(grid), dim3(256), ldskb * 1024, 0, s, d, n_steps, pitch, tickets); break;
                            case 1: hipLaunchKernelGGL(shape4<1>, dim3(grid), dim3(256), ldskb * 1024, 0, s, d, n_steps, pitch, tickets); break;
                            case 2: hipLaunchKernelGGL(shape4<2>, dim3(grid), dim3(256), ldskb * 1024, 0, s, d, n_steps, pitch, tickets); break;
                            case 4: hipLaunchKernelGGL((shape4<2, 1>), dim3(grid), dim3(256), ldskb * 1024, 0, s, d, n_steps, pitch, tickets); break;
                            default: hipLaunchKernelGGL(shape4<3>, dim3(grid), dim3(256), ldskb * 1024, 0, s, d, n_steps, pitch, tickets); break;
                        }
                    };
                    for (int i = 0; i < 3; ++i) launch(i);
                    CK(hipDeviceSynchronize());
                    const int reps = 10;
                    CK(hipEventRecord(e0, 0));
                    for (int i = 0; i < reps; ++i) launch(i);
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    float ms = 0.f;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    const double us = double(ms) * 1e3 / reps;
                    std::printf("{\"round\": %d, \"map\": %d, \"lds_kb\": %d, \"wg_per_cu\": %d, \"us\": %.2f, \"frac_of_8\": %.4f}\n", round, map, ldskb, bpc, us, algo / us / 8e6);
                    std::fflush(stdout);
                }
        }
    return 0;
}
