// Micro-probe (run on the GPU box): how many vector instructions ride along with a bf16 MFMA for free, by MFMA shape?
//   hipcc -O3 --offload-arch=gfx950 tools/coexec_probe32.hip -o gpurun_out/coexec_probe32 && gpurun_out/coexec_probe32
// Same arrangement as tools/coexec_probe.hip (one workgroup of 8 waves = two waves per SIMD, every wave runs the same stream);
// a trip holds the SAME matrix work in both shapes -- 16 x v_mfma_f32_16x16x32_bf16 or 8 x v_mfma_f32_32x32x16_bf16 (= 131072 MACs
// per wave) -- plus NV independent vector instructions spread evenly between the MFMAs.  /opt/skills/guides/MI355X_MICROARCH.md
// (cycle table) says an MFMA blocks the SIMD's vector issue for 8 of its 16 cycles (16x16x32) but 8 of its 32 (32x32x16).
// Operands are hashed pseudo-random bf16 values so that the 256-workgroup rows see the clock the chip holds under load.
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifndef VKIND
#define VKIND 1      // 1: v_fma_f32 (full rate), 4: v_max_f32 (half rate), 3: v_cvt_pk_bf16_f32
#endif
__device__ __forceinline__ void valu_op(float& x, float s) {
#if VKIND == 1
  asm volatile("v_fma_f32 %0, %0, %1, 1.0" : "+v"(x) : "v"(s));
#elif VKIND == 3
  asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(s));
#elif VKIND == 4
  asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(s));
#else
  asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(s));
#endif
}

// SHAPE 16: 16 MFMAs 16x16x32 on four accumulators; SHAPE 32: 8 MFMAs 32x32x16 on two accumulators.  NV vector ops per trip.
template <int SHAPE, int NV>
__device__ __forceinline__ void body(f32x4 (&acc4)[4], f32x16 (&acc16)[2], float (&v)[8], const bf16x8& a, const bf16x8& b, float s) {
  constexpr int NM = SHAPE == 16 ? 16 : 8;
  constexpr int PER = NV / NM;          // vector ops behind each MFMA
#pragma unroll
  for (int i = 0; i < NM; ++i) {
    if (SHAPE == 16) acc4[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4[i & 3], 0, 0, 0);
    else acc16[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc16[i & 1], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < PER; ++j) valu_op(v[(i * PER + j) & 7], s);
  }
}

template <int SHAPE, int NV, bool PRIO>
__global__ __launch_bounds__(512) void probe(float* out, long long* cycles, int trips, float s) {
  f32x4 acc4[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  f32x16 acc16[2];
  for (int i = 0; i < 16; ++i) { acc16[0][i] = 0.f; acc16[1][i] = 0.f; }
  float v[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  bf16x8 a, b;
  unsigned h = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u;
  for (int i = 0; i < 8; ++i) {
    h = h * 1664525u + 1013904223u; a[i] = (__bf16)(((int)(h >> 8) & 0xffff) * (1.f / 65536.f) - 0.5f);
    h = h * 1664525u + 1013904223u; b[i] = (__bf16)(((int)(h >> 8) & 0xffff) * (1.f / 65536.f) - 0.5f);
  }
  if (PRIO && (threadIdx.x >> 6) >= 4) __builtin_amdgcn_s_setprio(1);
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int t = 0; t < trips; ++t) body<SHAPE, NV>(acc4, acc16, v, a, b, s);
  const long long t1 = __builtin_readcyclecounter();
  float r = 0.f;
  for (int i = 0; i < 4; ++i) r += acc4[i][0] + acc4[i][1] + acc4[i][2] + acc4[i][3];
  for (int i = 0; i < 16; ++i) r += acc16[0][i] + acc16[1][i];
  for (int i = 0; i < 8; ++i) r += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int SHAPE, int NV, bool PRIO>
static double run(int blocks, bool print = true) {
  float* out; long long* cyc;
  hipMalloc(&out, sizeof(float) * 512 * blocks);
  hipMalloc(&cyc, sizeof(long long) * blocks);
  const int trips = blocks == 1 ? 20000 : 200000;      // 256 workgroups: long enough (~50 ms) for the clock to settle
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((probe<SHAPE, NV, PRIO>), dim3(blocks), dim3(512), 0, 0, out, cyc, trips / 10, 0.999f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((probe<SHAPE, NV, PRIO>), dim3(blocks), dim3(512), 0, 0, out, cyc, trips, 0.999f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long c0; hipMemcpy(&c0, cyc, sizeof(c0), hipMemcpyDeviceToHost);
  const double ns = ms * 1e6 / trips;
  if (print)
    printf("%s  %2d MFMA + %3d VALU per trip%s  blocks %3d: %7.1f ns per trip  (%5.2f TFLOP/s per CU, x256 = %6.0f)  s_memtime ticks %.1f\n",
           SHAPE == 16 ? "16x16x32" : "32x32x16", SHAPE == 16 ? 16 : 8, NV, PRIO ? " prio(4-7)=1" : "            ", blocks, ns,
           8 * 2.0 * 131072 / ns * 1e-3, 8 * 2.0 * 131072 / ns * 1e-3 * 256, (double)c0 / trips);
  hipFree(out); hipFree(cyc);
  return ns;
}

template <int NV>
static void pair(int blocks) {
  run<16, NV, false>(blocks);
  run<32, NV, false>(blocks);
}

int main() {
  printf("VKIND %d (1 v_fma_f32, 3 v_cvt_pk_bf16_f32, 4 v_max_f32)\n", VKIND);
  for (int blocks : {1, 256}) {
    pair<0>(blocks); pair<16>(blocks); pair<32>(blocks); pair<48>(blocks); pair<64>(blocks); pair<96>(blocks); pair<128>(blocks);
    run<16, 48, true>(blocks); run<32, 48, true>(blocks); run<16, 64, true>(blocks); run<32, 64, true>(blocks);
  }
  return 0;
}
