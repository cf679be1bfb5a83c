// K independent mono streams -> K dB rings from a C++ host through the C-ABI in ONE kernel launch ("Many independent dB
// batches of one geometry"): the frame loop of Spectrogram::processSynchronBlock (reference Spectrogram.cpp:50-119) over K streams'
// worth of blocks.  48 kHz, 1024-point FFT, hop 512, Hann, `F` frames per stream (BASELINE configs[1]: 4096).
//     one call    jsg_stft_db_launch_strided(plan, &args, K, floats_between_inputs, floats_between_rings, stream)
//     reference   K calls of jsg_stft_db_launch on the same buffers, and the same columns with exact_log (bit-reproducible dB)
// Check: every column identical, the padding behind the columns untouched.  Prints one JSON line with the time per batch of both forms.
//   usage: offline_db_batches_example [K] [frames]
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/jsg.h"

#define CKJ(call)                                                                        \
    do {                                                                                 \
        int rc_ = (call);                                                                \
        if (rc_ < 0) {                                                                   \
            std::fprintf(stderr, "%s -> %d: %s\n", #call, rc_, jsg_last_error(nullptr)); \
            std::exit(2);                                                                \
        }                                                                                \
    } while (0)
#define CKH(call)                                                           \
    do {                                                                    \
        hipError_t e_ = (call);                                             \
        if (e_ != hipSuccess) {                                             \
            std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); \
            std::exit(2);                                                   \
        }                                                                   \
    } while (0)

int main(int argc, char** argv) {
    if (jsg_device_count() < 1) {
        std::printf("{\"skipped\": \"no device\"}\n");
        return 0;
    }
    const int K = argc > 1 ? std::atoi(argv[1]) : 32, F = argc > 2 ? std::atoi(argv[2]) : 4096;
    const int N = 1024, hop = 512, H = N / 2 + 1, pitch = (H + 31) / 32 * 32;
    const int64_t n_samples = int64_t(F - 1) * hop + N, in_stride = (n_samples + 63) / 64 * 64, ring_stride = int64_t(F) * pitch;
    std::vector<float> x(size_t(in_stride) * K, 0.f);
    uint32_t s = 88172645u;
    for (int k = 0; k < K; ++k)
        for (int64_t i = 0; i < n_samples; ++i) {
            s ^= s << 13; s ^= s >> 17; s ^= s << 5;
            x[size_t(k) * in_stride + i] = 0.5f * std::sin(0.0288f * float(k % 5 + 1) * float(i % 100000)) + 0.1f * (float(s >> 8) * (2.0f / 16777216.0f) - 1.0f);
        }
    std::vector<float> win(N);
    CKJ(jsg_window_build(JSG_WIN_HANN, N, win.data()));
    float *d_in = nullptr, *d_one = nullptr, *d_all = nullptr;
    const size_t ring_bytes = size_t(ring_stride) * K * sizeof(float);
    CKH(hipMalloc(reinterpret_cast<void**>(&d_in), x.size() * sizeof(float)));
    CKH(hipMalloc(reinterpret_cast<void**>(&d_one), ring_bytes));
    CKH(hipMalloc(reinterpret_cast<void**>(&d_all), ring_bytes));
    CKH(hipMemcpy(d_in, x.data(), x.size() * sizeof(float), hipMemcpyHostToDevice));
    hipStream_t st;
    CKH(hipStreamCreate(&st));
    jsg_plan* plan = nullptr;
    CKJ(jsg_plan_create(&plan, N, win.data(), 1.0f));

    jsg_stft_args a;
    std::memset(&a, 0, sizeof a);
    a.in = d_in;
    a.in_pitch = in_stride;          // (one channel: the pitch is not used)
    a.in_samples = n_samples;
    a.channels = 1;
    a.hop = hop;
    a.feedblocks = N / hop;
    a.mix_mode = JSG_MIX_ABSMEAN;
    a.n_frames = F;
    a.out_pitch = pitch;
    a.ring_width = F;

    std::vector<float> one(size_t(ring_stride) * K), all(one.size());
    double us_one = 0.0, us_all = 0.0;
    size_t differing[2] = {0, 0}, padding_touched = 0;
    for (int exact = 0; exact < 2; ++exact) {
        a.exact_log = exact;
        CKH(hipMemset(d_one, 0x7f, ring_bytes));   // (0x7f7f7f7f: a pattern no column holds)
        CKH(hipMemset(d_all, 0x7f, ring_bytes));
        for (int rep = 0; rep < 3; ++rep) {        // the third repetition is the timed one
            CKH(hipStreamSynchronize(st));
            auto t0 = std::chrono::steady_clock::now();
            for (int k = 0; k < K; ++k) {
                jsg_stft_args b = a;
                b.in = d_in + size_t(k) * in_stride;
                b.out_db = d_one + size_t(k) * ring_stride;
                CKJ(jsg_stft_db_launch(plan, &b, st));
            }
            CKH(hipStreamSynchronize(st));
            auto t1 = std::chrono::steady_clock::now();
            a.out_db = d_all;
            CKJ(jsg_stft_db_launch_strided(plan, &a, K, in_stride, ring_stride, st));
            CKH(hipStreamSynchronize(st));
            auto t2 = std::chrono::steady_clock::now();
            if (!exact) {
                us_one = std::chrono::duration<double, std::micro>(t1 - t0).count() / K;
                us_all = std::chrono::duration<double, std::micro>(t2 - t1).count() / K;
            }
        }
        CKH(hipMemcpy(one.data(), d_one, ring_bytes, hipMemcpyDeviceToHost));
        CKH(hipMemcpy(all.data(), d_all, ring_bytes, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < one.size(); ++i) differing[exact] += std::memcmp(&one[i], &all[i], 4) != 0;
        uint32_t fill = 0x7f7f7f7fu;
        for (int k = 0; k < K; ++k)
            for (int c = 0; c < F; ++c)
                for (int p = H; p < pitch; ++p) padding_touched += std::memcmp(&all[size_t(k) * ring_stride + size_t(c) * pitch + p], &fill, 4) != 0;
    }
    char name[32] = "";
    a.exact_log = 0;
    CKJ(jsg_stft_db_strided_kernel_name(plan, &a, K, in_stride, name, sizeof name));
    const double bytes = double(F) * 4100.0;
    std::printf("{\"batches\": %d, \"frames_per_batch\": %d, \"kernel\": \"%s\", \"us_per_batch_one_launch_each\": %.2f, \"us_per_batch_one_strided_launch\": %.2f, "
                "\"frac_of_8TBps_strided\": %.3f, \"columns_differing\": %zu, \"columns_differing_exact_log\": %zu, \"padding_floats_touched\": %zu}\n",
                K, F, name, us_one, us_all, bytes / us_all / 8e6, differing[0], differing[1], padding_touched);
    CKJ(jsg_plan_destroy(plan));
    CKH(hipFree(d_in)); CKH(hipFree(d_one)); CKH(hipFree(d_all));
    CKH(hipStreamDestroy(st));
    return (differing[0] || differing[1] || padding_touched) ? 1 : 0;
}
