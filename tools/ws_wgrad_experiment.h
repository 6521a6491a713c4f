// The output-stationary weight-gradient kernel graduated into csrc/ws_gemm.h (ws_wgrad_kernel); this header only keeps the lab building.
#pragma once
#include "../offlinerl-kit_amd/csrc/ws_gemm.h"
