#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
__global__ void k(float* o, const float* x) {
  float a = x[threadIdx.x], b = x[threadIdx.x + 64];
  h2 h; h[0] = (_Float16)a; h[1] = (_Float16)b;
  unsigned hb = *(unsigned*)&h;
  float la, lb;
  asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(la) : "v"(hb), "v"(a));
  asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lb) : "v"(hb), "v"(b));
  o[threadIdx.x] = la - (a - (float)h[0]);
  o[threadIdx.x + 64] = lb - (b - (float)h[1]);
}
int main() {
  float h[128], *d, *o; for (int i = 0; i < 128; ++i) h[i] = 0.37f * (i - 60) + 1e-3f * i * i;
  hipMalloc(&d, 512); hipMalloc(&o, 512); hipMemcpy(d, h, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, d);
  hipMemcpy(h, o, 512, hipMemcpyDeviceToHost);
  float m = 0; for (int i = 0; i < 128; ++i) m = fmaxf(m, fabsf(h[i]));
  printf("v_fma_mix_f32 lo plane vs cvt + sub: max abs difference %g (0 = identical)\n", m);
  return 0;
}
