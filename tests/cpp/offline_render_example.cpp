// Offline rendering of K spectrogram images of one geometry from a C++ host through the C-ABI ("Offline rendering of
// many images"): K stereo streams at 96 kHz, 4096-point FFT, 87.5 % overlap, Jade palette -50..50 dB -- the pixel loop of
// SpectrogramComponent::timerCallback (reference Spectrogram.cpp:632-648: pixel(x, H-1-bin) = getRGBColor(mem[col][bin]) | 0xFF000000)
// applied to columns that never leave the GPU as dB values.
//     one call    jsg_stft_image_launch_strided(plan, &args, K, samples_between_streams, pixels_between_images, stream)
//     reference   K calls of jsg_stft_image_launch on the same buffers (plan pinned to the kernel the batch takes)
// Check: every pixel identical, nothing outside the images' columns touched.  Prints one JSON line (with the time per image of both).
//   usage: offline_render_example [K] [columns]
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
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
    const int K = argc > 1 ? std::atoi(argv[1]) : 12, F = argc > 2 ? std::atoi(argv[2]) : 1875;
    const int C = 2, N = 4096, hop = 512, H = N / 2 + 1;
    const int64_t n_samples = int64_t(F - 1) * hop + N, in_pitch = (n_samples + 63) / 64 * 64, stream_stride = in_pitch * C;
    const int64_t img_pitch = (F + 31) / 32 * 32, img_stride = img_pitch * H;
    std::vector<float> x(size_t(stream_stride) * K, 0.f);
    uint32_t s = 2463534242u;
    for (int k = 0; k < K; ++k)
        for (int c = 0; c < C; ++c)
            for (int64_t i = 0; i < n_samples; ++i) {
                s ^= s << 13; s ^= s >> 17; s ^= s << 5;
                x[size_t(k) * stream_stride + size_t(c) * in_pitch + i] =
                    0.5f * std::sin(0.0144f * float(k % 7 + c + 1) * float(i % 100000)) + 0.1f * (float(s >> 8) * (2.0f / 16777216.0f) - 1.0f);
            }
    std::vector<float> win(N);
    CKJ(jsg_window_build(JSG_WIN_HANN, N, win.data()));
    std::vector<int32_t> lut(256);
    CKJ(jsg_colormap_build(256, JSG_CM_JADE, lut.data()));
    float *d_in = nullptr;
    uint32_t *d_img = nullptr, *d_ref = nullptr;
    int32_t* d_lut = nullptr;
    hipStream_t st = nullptr;
    CKH(hipSetDevice(0));
    jsg_plan* plan = nullptr;
    CKJ(jsg_plan_create(&plan, N, win.data(), 1.0f));
    CKH(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    CKH(hipMalloc(reinterpret_cast<void**>(&d_in), x.size() * 4));
    CKH(hipMalloc(reinterpret_cast<void**>(&d_img), size_t(img_stride) * K * 4));
    CKH(hipMalloc(reinterpret_cast<void**>(&d_ref), size_t(img_stride) * K * 4));
    CKH(hipMalloc(reinterpret_cast<void**>(&d_lut), 256 * 4));
    CKH(hipMemcpy(d_in, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    CKH(hipMemcpy(d_lut, lut.data(), 256 * 4, hipMemcpyHostToDevice));
    CKH(hipMemset(d_img, 0x5a, size_t(img_stride) * K * 4));
    CKH(hipMemset(d_ref, 0x5a, size_t(img_stride) * K * 4));

    jsg_stft_image_args a{};
    a.stft.in = d_in;
    a.stft.in_pitch = in_pitch;
    a.stft.in_samples = n_samples;
    a.stft.channels = C;
    a.stft.hop = hop;
    a.stft.feedblocks = N / hop;
    a.stft.mix_mode = JSG_MIX_ABSMEAN;
    a.stft.n_frames = F;
    a.stft.ring_width = F;
    a.stft.plan_select = 2;              // pinned: the batch and the single launches run the same kernel, so the pixels are comparable bit for bit
    a.colour.ring_width = F;
    a.colour.height = H;
    a.colour.n_cols = F;
    a.colour.x_wrap = int32_t(img_pitch);
    a.colour.lut = d_lut;
    a.colour.n_colors = 256;
    CKJ(jsg_colormap_range(256, -50.0f, 50.0f, &a.colour.vmin, &a.colour.vmax, &a.colour.access_mult));
    a.colour.argb_out = d_img;
    a.colour.argb_pitch = img_pitch;
    const int needs_scratch = jsg_stft_image_strided_needs_scratch(plan, &a, K);
    CKJ(needs_scratch);

    auto now = [] { return std::chrono::steady_clock::now(); };
    auto strided = [&] { CKJ(jsg_stft_image_launch_strided(plan, &a, K, stream_stride, img_stride, st)); };
    auto singles = [&] {
        for (int k = 0; k < K; ++k) {
            jsg_stft_image_args one = a;
            one.stft.in = d_in + int64_t(k) * stream_stride;
            one.colour.argb_out = d_ref + int64_t(k) * img_stride;
            CKJ(jsg_stft_image_launch(plan, &one, st));
        }
    };
    strided(); singles();
    CKH(hipStreamSynchronize(st));
    const int reps = std::max(4, 600 / K);                  // ~10 ms per leg; the same again before the clock is read (GPU clocks settle)
    for (int r = 0; r < reps; ++r) { strided(); singles(); }
    CKH(hipStreamSynchronize(st));
    auto t0 = now();
    for (int r = 0; r < reps; ++r) strided();
    CKH(hipStreamSynchronize(st));
    auto t1 = now();
    for (int r = 0; r < reps; ++r) singles();
    CKH(hipStreamSynchronize(st));
    auto t2 = now();

    std::vector<uint32_t> img(size_t(img_stride) * K), ref(size_t(img_stride) * K);
    CKH(hipMemcpy(img.data(), d_img, img.size() * 4, hipMemcpyDeviceToHost));
    CKH(hipMemcpy(ref.data(), d_ref, ref.size() * 4, hipMemcpyDeviceToHost));
    size_t differing = 0, untouched_wrong = 0, opaque = 0;
    for (int k = 0; k < K; ++k)
        for (int y = 0; y < H; ++y)
            for (int64_t xx = 0; xx < img_pitch; ++xx) {
                const size_t i = size_t(k) * img_stride + size_t(y) * img_pitch + xx;
                differing += img[i] != ref[i];
                if (xx >= F) untouched_wrong += img[i] != 0x5a5a5a5au;
                else opaque += (img[i] >> 24) == 0xffu;
            }
    const double us = 1e6 / double(reps * K);
    std::printf("{\"images\": %d, \"columns\": %d, \"one_kernel_for_the_batch\": %s, \"pixels_differing\": %zu, \"padding_pixels_touched\": %zu, "
                "\"opaque_pixels\": %zu, \"pixels\": %zu, \"us_per_image_strided\": %.2f, \"us_per_image_single_launches\": %.2f}\n",
                K, F, needs_scratch == 0 ? "true" : "false", differing, untouched_wrong, opaque, size_t(K) * H * F,
                std::chrono::duration<double>(t1 - t0).count() * us, std::chrono::duration<double>(t2 - t1).count() * us);
    CKH(hipFree(d_in)); CKH(hipFree(d_img)); CKH(hipFree(d_ref)); CKH(hipFree(d_lut));
    CKJ(jsg_plan_destroy(plan));
    CKH(hipStreamDestroy(st));
    return differing == 0 && untouched_wrong == 0 && opaque == size_t(K) * H * F ? 0 : 1;
}
