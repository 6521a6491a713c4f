// What sign does a NaN born inside the split-precision pipeline carry?  ("range and NaN safety")
// An activation beyond fp16's range splits into hi = +inf, lo = -inf; the three MFMA products then add +inf and -inf.  The integer-view
// ReLU (gemm.h: orl_relu_mask4) keeps a NaN whose sign bit is clear and maps one whose sign bit is set to +0.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned* o, float big, float wsign) {
  const int lane = threadIdx.x;
  // A (weights) = wsign * 0.5 everywhere; B (activations): hi = half(big) = inf, lo = half(big - inf) = -inf in k element 0 of row 0
  h8 wa, xh, xl;
  for (int j = 0; j < 8; ++j) { wa[j] = (_Float16)(wsign * 0.5f); xh[j] = (_Float16)0.25f; xl[j] = (_Float16)0.0f; }
  if (lane == 0) { const _Float16 hi = (_Float16)big; xh[0] = hi; xl[0] = (_Float16)(big - (float)hi); }
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, xl, acc, 0, 0, 0);   // w_hi * x_lo
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa, xh, acc, 0, 0, 0);   // w_hi * x_hi
  float z = acc[0] * (1.0f / 64.0f) + 0.125f;                             // epilogue: inverse operand scale, bias
  int zi = __float_as_int(z);
  int relu = zi > 0 ? zi : 0;                                              // integer-view ReLU
  o[lane * 4 + 0] = __float_as_uint(acc[0]);
  o[lane * 4 + 1] = __float_as_uint(z);
  o[lane * 4 + 2] = (unsigned)relu;
  o[lane * 4 + 3] = __float_as_uint(fmaxf(z, 0.f));
}
int main() {
  unsigned *d, h[256];
  hipMalloc(&d, sizeof(h));
  for (float ws : {1.0f, -1.0f}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 70000.0f, ws);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // column 0 of the output (lanes 0, 16, 32, 48 hold rows 0..3 / 4..7 / ...) sees the overflowed element
    printf("weights %+.1f: acc %08x  bias-fma %08x  int-view relu %08x  v_max_f32(z, 0) %08x   (an untouched column: acc %08x)\n", ws, h[0], h[1], h[2], h[3], h[4 * 1]);
  }
  float a = __builtin_inff(), r = a - a; unsigned rb; memcpy(&rb, &r, 4);
  printf("host inf - inf = %08x\n", rb);
  return 0;
}
