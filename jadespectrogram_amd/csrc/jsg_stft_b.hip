// Translation unit B of the STFT kernels: the two 4096-point plans (default machine scheduler: see jsg_stft_a.hip).
#ifndef JSG_X_B_SCALAR_TWIDDLE   // (variant builds: A/B of the scalar form in this unit)
#define JSG_TWIDDLE_CONST_VGPR 1   // see mul_w_q1
#endif
#include "jsg_stft_kernel.h"

namespace jsg {
JSG_DEFINE_PLAN(Cfg4096)
JSG_DEFINE_PLAN(Cfg4096B)
hipError_t touch_module_b() {
    hipFuncAttributes fa;
    return hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&stft_db_kernel<Cfg4096, 3>));
}
}  // namespace jsg
