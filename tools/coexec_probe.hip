// Micro-probe (run on the GPU box): do VALU and MFMA instructions of the same SIMD execute concurrently on gfx950?
//   hipcc -O3 --offload-arch=gfx950 tools/coexec_probe.hip -o gpurun_out/coexec_probe && gpurun_out/coexec_probe
// One workgroup per CU, 8 waves (2 per SIMD).  Modes: every wave MFMA only / VALU only / waves 0-3 MFMA + waves 4-7 VALU (the two
// waves of a SIMD run different pipes) / every wave an interleaved MFMA + VALU stream.  Reports cycles per loop trip.
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifndef VKIND
#define VKIND 0      // 0: compiler fmaf (becomes v_pk_fma_f32), 1: v_fma_f32, 2: v_add_u32, 3: v_cvt_pk_bf16_f32, 4: v_max_f32
#endif
__device__ __forceinline__ void valu_op(float& x, float s) {
#if VKIND == 0
  x = __builtin_fmaf(x, s, 1.0f);
#elif VKIND == 1
  asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(x) : "v"(s));
#elif VKIND == 2
  asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(s));
#elif VKIND == 3
  asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(s));
#elif VKIND == 4
  asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(s));
#elif VKIND == 5
  asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x) : "v"(s));
#elif VKIND == 6
  { unsigned long long t; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(t) : "v"(x), "v"(s) : "vcc"); x = __uint_as_float((unsigned)t); }
#elif VKIND == 7
  { unsigned long long t = __float_as_uint(x); asm volatile("v_lshl_add_u64 %0, %0, 2, %0" : "+v"(t)); x = __uint_as_float((unsigned)t); }
#elif VKIND == 8
  asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(x) : "v"(s));
#elif VKIND == 9
  asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(s));
#elif VKIND == 10
  asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, 0, %0, vcc" : "+v"(x) : "v"(s) : "vcc");
#elif VKIND == 11
  asm volatile("v_med3_i32 %0, %0, 0, 1" : "+v"(x));
#elif VKIND == 13      // x - float(half in the low / high 16 bits of s): the lo plane of the fp16 split without a separate v_cvt_f32_f16
  asm volatile("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel_hi:[1,0,0]" : "+v"(x) : "v"(s));
#elif VKIND == 14
  asm volatile("v_fma_mix_f32 %0, %1, -1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(x) : "v"(s));
#elif VKIND == 15
  asm volatile("v_cvt_f32_f16 %0, %1" : "+v"(x) : "v"(s));
#elif VKIND == 16
  asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x) : "v"(s));
#else
  asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(s));
#endif
}
template <int NM, int NV>      // MFMAs and VALU ops per trip (each on independent chains)
__device__ __forceinline__ void body(f32x4 (&acc)[4], float (&v)[8], const bf16x8& a, const bf16x8& b, float s) {
#pragma unroll
  for (int i = 0; i < (NM > NV / 4 ? NM : NV / 4); ++i) {
    if (i < NM) acc[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i & 3], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (4 * i + j < NV) valu_op(v[(4 * i + j) & 7], s);
  }
}

template <int MODE>
__global__ __launch_bounds__(512) void probe(float* out, long long* cycles, int trips, float s) {
  const int wave = threadIdx.x >> 6;
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float v[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (i + 1)); }
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int t = 0; t < trips; ++t) {
    if (MODE == 0) body<16, 0>(acc, v, a, b, s);                 // 16 MFMAs
    else if (MODE == 1) body<0, 64>(acc, v, a, b, s);            // 64 VALU
    else if (MODE == 2) { if (wave < 4) body<16, 0>(acc, v, a, b, s); else body<0, 64>(acc, v, a, b, s); }
    else if (MODE == 3) body<16, 64>(acc, v, a, b, s);           // both, interleaved 1 : 4, every wave
    else if (MODE == 4) body<16, 32>(acc, v, a, b, s);           // 1 : 2
    else if (MODE == 5) { if (wave < 4) body<16, 0>(acc, v, a, b, s); }     // one MFMA wave per SIMD only
    else if (MODE == 6) { if (wave >= 4) body<0, 64>(acc, v, a, b, s); }    // one VALU wave per SIMD only
  }
  const long long t1 = __builtin_readcyclecounter();
  float r = 0.f;
  for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) r += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char* what, int blocks) {
  float* out; long long* cyc;
  hipMalloc(&out, sizeof(float) * 512 * blocks);
  hipMalloc(&cyc, sizeof(long long) * blocks);
  const int trips = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(512), 0, 0, out, cyc, 100, 0.999f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(512), 0, 0, out, cyc, trips, 0.999f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c0; hipMemcpy(&c0, cyc, sizeof(c0), hipMemcpyDeviceToHost);
  printf("%-64s blocks %3d: %8.3f ms, %7.1f ns per trip, s_memtime ticks per trip %.1f\n", what, blocks, ms, ms * 1e6 / trips, (double)c0 / trips);
  hipFree(out); hipFree(cyc);
}

int main() {
  printf("VKIND %d\n", VKIND);
  for (int blocks : {1}) {
#if VKIND < 5
    run<0>("every wave: 16 MFMA (16x16x32 bf16) per trip", blocks);
#endif
    run<1>("every wave: 64 VALU ops per trip", blocks);
#if VKIND >= 5
    continue;
#endif
    run<5>("waves 0-3 only: 16 MFMA per trip (one wave per SIMD)", blocks);
    run<6>("waves 4-7 only: 64 VALU per trip (one wave per SIMD)", blocks);
    run<2>("waves 0-3: 16 MFMA, waves 4-7: 64 VALU (different pipes per wave)", blocks);
    run<3>("every wave: 16 MFMA + 64 VALU interleaved", blocks);
    run<4>("every wave: 16 MFMA + 32 VALU interleaved", blocks);
  }
  return 0;
}
