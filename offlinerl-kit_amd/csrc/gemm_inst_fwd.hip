// gemm_inst_fwd.hip — explicit instantiations of the tiled GEMM (csrc/gemm_kernel.h); one translation unit per group so they build in parallel.
#include "gemm_kernel.h"

namespace orl {
template hipError_t launch_gemm<PA_PLAIN, PB_PLAIN, E_BIAS_RELU>(int, const GemmP&, int, hipStream_t, bool, bool, int);
template hipError_t launch_gemm<PA_PLAIN, PB_PLAIN, E_BIAS>(int, const GemmP&, int, hipStream_t, bool, bool, int);
}  // namespace orl
