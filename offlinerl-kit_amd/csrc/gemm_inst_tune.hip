// gemm_inst_tune.hip — the tuning tap behind orl_debug_gemm_time (csrc/gemm_kernel.h, launch_tune).
#define ORL_GEMM_TUNE_TU 1
#include "gemm_kernel.h"
