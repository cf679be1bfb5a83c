// Translation unit A of the STFT kernels: the 512-, 1024- (three kernels), 2048- (three kernels) and 8192-point plans.
// Compiled with -mllvm -amdgpu-sched-strategy=max-ilp (jadespectrogram_amd/_build.py): the ILP-first machine scheduler keeps
// dependent packed-math instructions apart (a consumer directly behind its v_pk_*_f32 producer is given an s_nop by the hazard
// recognizer: 264 of them in the two-stage 2048-point kernel, 69 with this scheduler) and needs fewer s_waitcnt.  Measured
// (round 3, abbench, same box): C3 45.7 -> 40.6 us per launch (-11 %), C2 5.71 -> 5.56 us in order, 512 points -2 %, 8192 -1 %;
// the 4096-point kernels lose 1-4 % with it and live in jsg_stft_b.hip.
#include "jsg_stft_kernel.h"

namespace jsg {
JSG_DEFINE_PLAN(Cfg512)
JSG_DEFINE_PLAN(Cfg1024)
JSG_DEFINE_PLAN(Cfg1024I)
JSG_DEFINE_PLAN(Cfg1024B)
JSG_DEFINE_PLAN(Cfg2048)
JSG_DEFINE_PLAN(Cfg2048B)
JSG_DEFINE_PLAN(Cfg2048P)
JSG_DEFINE_PLAN(Cfg8192)
hipError_t launch_runs_Cfg1024(const StftKArgs& ka, int mixop, dim3 grid, hipStream_t s) { return launch_stft_strided<Cfg1024, 2>(ka, mixop, grid, s); }
hipError_t touch_module_a() {
    hipFuncAttributes fa;
    return hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&stft_db_kernel<Cfg1024, 3>));
}
}  // namespace jsg
