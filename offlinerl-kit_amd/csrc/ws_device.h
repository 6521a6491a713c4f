// ws_device.h — device helpers shared by the weight-stationary kernels (ws_fwd.hip, ws_dgrad.hip, ws_wgrad.hip).
#pragma once
#include "ws_gemm.h"

namespace orl {

typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ inline void ws_split8(const f32x4& a, const f32x4& b, bf16x8& h, bf16x8& l) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const __bf16 ha = (__bf16)a[j]; h[j] = ha; l[j] = (__bf16)(a[j] - (float)ha);
    const __bf16 hb = (__bf16)b[j]; h[4 + j] = hb; l[4 + j] = (__bf16)(b[j] - (float)hb);
  }
}

}  // namespace orl
