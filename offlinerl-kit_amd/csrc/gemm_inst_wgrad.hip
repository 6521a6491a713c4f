// gemm_inst_wgrad.hip — explicit instantiations of the tiled GEMM (csrc/gemm_kernel.h); one translation unit per group so they build in parallel.
#include "gemm_kernel.h"

namespace orl {
template hipError_t launch_gemm<PA_RANK1, PB_PLAIN, E_WGRAD>(int, const GemmP&, int, hipStream_t, bool, bool, int);
template hipError_t launch_gemm<PA_PLAIN, PB_PLAIN, E_WGRAD>(int, const GemmP&, int, hipStream_t, bool, bool, int);
}  // namespace orl
