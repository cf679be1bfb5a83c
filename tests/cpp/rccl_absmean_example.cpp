// Cross-GPU AbsMean from ONE host process through the C-ABI + RCCL ("Several GPUs": the one case where the
// sharded path has a real exchange step).  Reference semantics: Spectrogram.cpp:68-76 -- the column is the float sum of the
// channels' power spectra divided by the channel count, then 10*log10 (:107).  With the channels of one stream sharded over
// several GPUs:
//     every device   jsg_stft_db_launch(mix = JSG_MIX_SUM, linear_out = 1)   partial sums of linear power   [frames][pitch]
//     all devices    ncclAllReduce(sum, float) over xGMI                      (in place; one call per device inside a group)
//     every device   jsg_db_from_power_launch(divisor = total channels)       -> the dB columns of the mixed stream
// One communicator per device from ncclCommInitAll, one stream per device, no host synchronisation between the three steps.
//   usage: rccl_absmean_example            every visible device is one rank.  ONE visible device (the test box): a one-rank communicator --
//                                          ncclCommInitAll / ncclAllReduce / the finish kernel execute the same call sequence; the device then
//                                          holds all channels, its JSG_MIX_SUM partial sums ARE the full sums, and the all-reduce must leave
//                                          them bit for bit as they were (checked).  No device: {"skipped": ...}, exit 0.
// Check: the columns equal those of ONE device that holds all channels (fused AbsMean kernel) within float32 reassociation
// (the partial sums are added in a different order), and they are identical on every device.  Prints one JSON line.
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cmath>
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
#define CKH(call)                                                                             \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));                   \
            std::exit(2);                                                                     \
        }                                                                                     \
    } while (0)
#define CKN(call)                                                                             \
    do {                                                                                      \
        ncclResult_t r_ = (call);                                                             \
        if (r_ != ncclSuccess) {                                                              \
            std::fprintf(stderr, "%s: %s\n", #call, ncclGetErrorString(r_));                  \
            std::exit(2);                                                                     \
        }                                                                                     \
    } while (0)

int main() {
    const int ndev = jsg_device_count();
    if (ndev < 1) {
        std::printf("{\"skipped\": \"no device\"}\n");
        return 0;
    }
    const int C = 64, N = 1024, hop = 512, F = 4096;               // BASELINE configs[3]: 64 channels sharded over the GPUs of one node
    const int H = N / 2 + 1, pitch = (H + 31) / 32 * 32;
    const int64_t n_samples = int64_t(F) * hop + (N - hop), in_pitch = (n_samples + 63) / 64 * 64;
    std::vector<float> x(size_t(C) * in_pitch, 0.f);
    uint32_t s = 2463534242u;
    for (int c = 0; c < C; ++c)
        for (int64_t i = 0; i < n_samples; ++i) {
            s ^= s << 13; s ^= s >> 17; s ^= s << 5;
            x[size_t(c) * in_pitch + i] = 0.5f * std::sin(0.0288f * float(c % 12 + 1) * float(i % 100000)) + 0.1f * (float(s >> 8) * (2.0f / 16777216.0f) - 1.0f);
        }
    std::vector<float> win(N);
    CKJ(jsg_window_build(JSG_WIN_HANN, N, win.data()));

    std::vector<int> devs(size_t(ndev), 0), first(size_t(ndev), 0), count(size_t(ndev), 0);
    for (int d = 0; d < ndev; ++d) {
        devs[size_t(d)] = d;
        first[size_t(d)] = d * (C / ndev) + std::min(d, C % ndev);       // contiguous shards, sizes differ by at most one
        count[size_t(d)] = C / ndev + (d < C % ndev ? 1 : 0);
    }
    std::vector<ncclComm_t> comm(size_t(ndev), nullptr);
    CKN(ncclCommInitAll(comm.data(), ndev, devs.data()));

    std::vector<jsg_plan*> plan(size_t(ndev), nullptr);
    std::vector<hipStream_t> st(size_t(ndev), nullptr);
    std::vector<float*> d_in(size_t(ndev), nullptr), d_pow(size_t(ndev), nullptr);
    const size_t out_bytes = size_t(F) * pitch * sizeof(float);
    for (int d = 0; d < ndev; ++d) {
        CKH(hipSetDevice(d));
        CKJ(jsg_plan_create(&plan[size_t(d)], N, win.data(), 1.0f));
        CKH(hipStreamCreateWithFlags(&st[size_t(d)], hipStreamNonBlocking));
        const int cd = count[size_t(d)] > 0 ? count[size_t(d)] : 1;
        CKH(hipMalloc(reinterpret_cast<void**>(&d_in[size_t(d)]), size_t(cd) * in_pitch * sizeof(float)));
        CKH(hipMalloc(reinterpret_cast<void**>(&d_pow[size_t(d)]), out_bytes));
        CKH(hipMemsetAsync(d_pow[size_t(d)], 0, out_bytes, st[size_t(d)]));            // a device without channels contributes zeros
        if (count[size_t(d)] > 0)
            CKH(hipMemcpyAsync(d_in[size_t(d)], x.data() + size_t(first[size_t(d)]) * in_pitch, size_t(count[size_t(d)]) * in_pitch * sizeof(float),
                               hipMemcpyHostToDevice, st[size_t(d)]));
    }
    // 1. partial sums of linear power on every device
    for (int d = 0; d < ndev; ++d) {
        if (count[size_t(d)] == 0) continue;
        CKH(hipSetDevice(d));
        jsg_stft_args a{};
        a.in = d_in[size_t(d)]; a.in_pitch = in_pitch; a.in_samples = n_samples; a.channels = count[size_t(d)];
        a.hop = hop; a.feedblocks = N / hop; a.mix_mode = JSG_MIX_SUM; a.linear_out = 1;
        a.n_frames = F; a.out_db = d_pow[size_t(d)]; a.out_pitch = pitch; a.ring_width = F;
        CKJ(jsg_stft_db_launch(plan[size_t(d)], &a, st[size_t(d)]));
    }
    // (one rank: keep the partial sums as they are before the collective -- with a single rank the all-reduce must be the identity)
    std::vector<float> before;
    if (ndev == 1) {
        before.resize(size_t(F) * pitch);
        CKH(hipMemcpyAsync(before.data(), d_pow[0], out_bytes, hipMemcpyDeviceToHost, st[0]));
        CKH(hipStreamSynchronize(st[0]));
    }
    // 2. one all-reduce of [F][pitch] floats (8.9 MB) over xGMI, in place, stream-ordered behind the kernels
    CKN(ncclGroupStart());
    for (int d = 0; d < ndev; ++d)
        CKN(ncclAllReduce(d_pow[size_t(d)], d_pow[size_t(d)], size_t(F) * pitch, ncclFloat, ncclSum, comm[size_t(d)], st[size_t(d)]));
    CKN(ncclGroupEnd());
    size_t allreduce_changed = 0;
    if (ndev == 1) {
        std::vector<float> after(size_t(F) * pitch);
        CKH(hipMemcpyAsync(after.data(), d_pow[0], out_bytes, hipMemcpyDeviceToHost, st[0]));
        CKH(hipStreamSynchronize(st[0]));
        for (int f = 0; f < F; ++f)
            for (int k = 0; k < H; ++k)
                allreduce_changed += after[size_t(f) * pitch + k] != before[size_t(f) * pitch + k];
    }
    // 3. divide by the channel count and take the log, on every device (in place)
    for (int d = 0; d < ndev; ++d) {
        CKH(hipSetDevice(d));
        CKJ(jsg_db_from_power_launch(d_pow[size_t(d)], d_pow[size_t(d)], int64_t(F) * pitch, float(C), st[size_t(d)]));
    }
    std::vector<std::vector<float>> got(size_t(ndev), std::vector<float>(size_t(F) * pitch));
    for (int d = 0; d < ndev; ++d) {
        CKH(hipSetDevice(d));
        CKH(hipMemcpyAsync(got[size_t(d)].data(), d_pow[size_t(d)], out_bytes, hipMemcpyDeviceToHost, st[size_t(d)]));
        CKH(hipStreamSynchronize(st[size_t(d)]));
    }
    // reference: all 64 channels on device 0, fused AbsMean
    CKH(hipSetDevice(0));
    float *d_all = nullptr, *d_ref = nullptr;
    CKH(hipMalloc(reinterpret_cast<void**>(&d_all), size_t(C) * in_pitch * sizeof(float)));
    CKH(hipMalloc(reinterpret_cast<void**>(&d_ref), out_bytes));
    CKH(hipMemcpy(d_all, x.data(), size_t(C) * in_pitch * sizeof(float), hipMemcpyHostToDevice));
    jsg_stft_args a{};
    a.in = d_all; a.in_pitch = in_pitch; a.in_samples = n_samples; a.channels = C; a.hop = hop; a.feedblocks = N / hop;
    a.mix_mode = JSG_MIX_ABSMEAN; a.n_frames = F; a.out_db = d_ref; a.out_pitch = pitch; a.ring_width = F;
    CKJ(jsg_stft_db_launch(plan[0], &a, st[0]));
    std::vector<float> ref(size_t(F) * pitch);
    CKH(hipMemcpyAsync(ref.data(), d_ref, out_bytes, hipMemcpyDeviceToHost, st[0]));
    CKH(hipStreamSynchronize(st[0]));
    double worst = 0.0;
    size_t devices_differing = 0;
    for (int f = 0; f < F; ++f)
        for (int k = 0; k < H; ++k) {
            const size_t i = size_t(f) * pitch + k;
            worst = std::max(worst, std::fabs(double(got[0][i]) - double(ref[i])));
        }
    for (int d = 1; d < ndev; ++d) {
        bool same = true;
        for (int f = 0; f < F && same; ++f)
            for (int k = 0; k < H; ++k)
                if (got[size_t(d)][size_t(f) * pitch + k] != got[0][size_t(f) * pitch + k]) { same = false; break; }
        devices_differing += same ? 0 : 1;
    }
    std::printf("{\"devices\": %d, \"ranks\": %d, \"channels\": %d, \"columns\": %d, \"max_abs_db_diff_vs_one_device\": %.3g, \"devices_differing_from_device0\": %zu, "
                "\"one_rank_allreduce_changed_values\": %zu, \"rccl_calls_executed\": \"ncclCommInitAll, ncclGroupStart, ncclAllReduce x %d, ncclGroupEnd, ncclCommDestroy\"}\n",
                ndev, ndev, C, F, worst, devices_differing, allreduce_changed, ndev);
    for (int d = 0; d < ndev; ++d) {
        CKH(hipSetDevice(d));
        CKN(ncclCommDestroy(comm[size_t(d)]));
        (void)hipFree(d_in[size_t(d)]); (void)hipFree(d_pow[size_t(d)]);
        (void)hipStreamDestroy(st[size_t(d)]);
        CKJ(jsg_plan_destroy(plan[size_t(d)]));
    }
    (void)hipFree(d_all); (void)hipFree(d_ref);
    return (worst < 1e-3 && devices_differing == 0 && allreduce_changed == 0) ? 0 : 1;   // float32 reassociation of 64 addends: ~1e-5 dB on ordinary bins
}
