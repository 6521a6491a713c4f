// ws_device.h — device helpers shared by the weight-stationary kernels (ws_fwd.hip, ws_dgrad.hip, ws_wgrad.hip).
#pragma once
#include "ws_gemm.h"

namespace orl {

__device__ inline void ws_split8(const f32x4& a, const f32x4& b, hx8& h, hx8& l) {
  hx4 ha, la, hb, lb;
  orl_split4(a, ha, la);
  orl_split4(b, hb, lb);
  h = __builtin_shufflevector(ha, hb, 0, 1, 2, 3, 4, 5, 6, 7);
  l = __builtin_shufflevector(la, lb, 0, 1, 2, 3, 4, 5, 6, 7);
}

__device__ inline void ws_split8x3(const f32x4& a, const f32x4& b, hx8& h, hx8& m, hx8& l) {
  hx4 ha, ma, la, hb, mb, lb;
  orl_split4x3(a, ha, ma, la);
  orl_split4x3(b, hb, mb, lb);
  h = __builtin_shufflevector(ha, hb, 0, 1, 2, 3, 4, 5, 6, 7);
  m = __builtin_shufflevector(ma, mb, 0, 1, 2, 3, 4, 5, 6, 7);
  l = __builtin_shufflevector(la, lb, 0, 1, 2, 3, 4, 5, 6, 7);
}

}  // namespace orl
