// gfx950 probe: v_permlane16_swap_b32 / v_permlane32_swap_b32 as a cross-row sum (lanes l, l ^ 16, l ^ 32, l ^ 48) on the vector ALU --
// what two dependent ds_bpermute round trips (__shfl_xor 16, 32) do through the LDS crossbar.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/permlane_swap_check.hip -o /tmp/plc && /tmp/plc
#include <hip/hip_runtime.h>
__global__ void k(float* out, const float* in) {
  float part = in[threadIdx.x];
  {   // (the builtin's two results come back in ONE register with this compiler: v_add_f32 v1, v1, v1 -- inline asm instead; the s_nop covers
      // the VALU-write -> permlane-swap hazard the compiler would otherwise pad)
    float a = part, b = part;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    part = a + b;
  }
  {
    float a = part, b = part;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    part = a + b;
  }
  out[threadIdx.x] = part;
}
int main() {
  float *in, *out; hipMalloc(&in, 256); hipMalloc(&out, 256);
  float h[64]; for (int i = 0; i < 64; ++i) h[i] = (float)(1 << (i / 16)) * (1 + (i % 16) * 0.001f);
  hipMemcpy(in, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, in);
  float o[64]; hipMemcpy(o, out, 256, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 64; ++i) { float e = (h[i % 16] + h[16 + i % 16]) + (h[32 + i % 16] + h[48 + i % 16]); /* (rows 0 + 1) + (rows 2 + 3) */ if (o[i] != e) { ++bad; if (bad < 4) printf("lane %d got %g expected %g\n", i, o[i], e); } }
  printf("permlane swap row sums: %s\n", bad ? "MISMATCH" : "ok");
  return bad != 0;
}
