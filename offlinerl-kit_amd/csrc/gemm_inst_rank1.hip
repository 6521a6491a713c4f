// gemm_inst_rank1.hip — explicit instantiations of the tiled GEMM (csrc/gemm_kernel.h); one translation unit per group so they build in parallel.
#include "gemm_kernel.h"

namespace orl {
template hipError_t launch_gemm<PA_RANK1, PB_PLAIN, E_MASK>(int, const GemmP&, int, hipStream_t, bool, bool, int);
template hipError_t launch_gemm<PA_RANK1, PB_PLAIN, E_PLAIN>(int, const GemmP&, int, hipStream_t, bool, bool, int);
template hipError_t launch_gemm_rank1_bits<E_MASK>(int, const GemmP&, int, hipStream_t, int);
}  // namespace orl
