// Micro-probe (run on the GPU box): how much does the ORDER of one wave's instruction stream matter for a k step shaped like the
// weight-stationary forward's -- 4 ds_read_b128 fragment reads, 12 dependent-chain MFMAs (16x16x32, four accumulators), ~45 half-rate vector
// instructions of the neighbouring pipeline stages -- with two waves per SIMD running the same program and one barrier per 8 k steps?
//   hipcc -O3 --offload-arch=gfx950 tools/stream_order_probe.hip -o gpurun_out/stream_order_probe && gpurun_out/stream_order_probe
// MODE 0: reads, wait, 12 MFMAs, then the 45 vector instructions (what hipcc emits for ws_fwd's epilogue pieces: k steps 0-3)
// MODE 1: reads, wait, then MFMA + 3-4 vector instructions, twelve times (a hand-interleaved stream)
// MODE 2: as 0, but waves 4-7 (the second wave of every SIMD) run the vector block BEFORE the MFMAs (half a k step out of phase)
// MODE 3: MFMAs only (the floor);  MODE 4: vector instructions only
// MODE 5: SPECIALISED waves -- waves 0-3 (one per SIMD) issue all 24 MFMAs of the SIMD's k step (two column halves off the same four fragment
//         reads) and nothing else; waves 4-7 issue all 2 NV vector instructions.  Same work per SIMD and k step as modes 0-2.
// MODE 6: as 5, and the vector waves also move the accumulators' worth of data through LDS (8 ds_read_b128 per 8 k steps)
// NV = vector instructions per k step (default 45), VKIND as in coexec_probe.hip (4 = v_max_f32, half rate).
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#ifndef NV
#define NV 45
#endif
__device__ __forceinline__ void valu_op(float& x, float s) { asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(s)); }

template <int MODE>
__global__ __launch_bounds__(512) void probe(float* out, int groups, float s) {
  __shared__ __attribute__((aligned(16))) _Float16 img[2 * 32 * 256];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lq = lane >> 4;
  for (int i = tid; i < 2 * 32 * 256; i += 512) img[i] = (_Float16)(((i * 2654435761u) >> 20 & 1023) * (1.f / 1024.f) - 0.5f);
  h8 b[2];
  for (int i = 0; i < 8; ++i) { b[0][i] = (_Float16)(0.01f * (lane + i)); b[1][i] = (_Float16)(0.02f * (i + 1) - 0.001f * lane); }
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float v[8] = {1, 2, 3, 4, 5, 6, 7, 8};
  __syncthreads();
  const bool late = (MODE == 2) && wave >= 4;
  if (MODE == 5 || MODE == 6) {
    f32x4 acc2[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int g = 0; g < groups; ++g) {
      if (wave < 4) {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          h8 fa[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) fa[r] = *(const h8*)&img[((r & 1) * 16 + li) * 256 + (((4 * ks + lq + 8 * (r >> 1)) ^ li) << 3)];
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < 12; ++j) {
            acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j & 1], fa[j & 3], acc[j & 3], 0, 0, 0);
            acc2[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[(j + 1) & 1], fa[j & 3], acc2[j & 3], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
          for (int j = 0; j < 2 * NV; ++j) valu_op(v[j & 7], s);
          if (MODE == 6) {
            const f32x4 t = *(const f32x4*)&img[(ks * 64 + lane) * 8];
            v[ks & 7] += t[0] + t[3];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __syncthreads();
    }
    float r = 0.f;
    for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + acc2[i][0] + acc2[i][3];
    for (int i = 0; i < 8; ++i) r += v[i];
    out[blockIdx.x * 512 + tid] = r;
    return;
  }
  for (int g = 0; g < groups; ++g) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      h8 fa[4];
      if (MODE != 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r) fa[r] = *(const h8*)&img[((r & 1) * 16 + li) * 256 + (((4 * ks + lq + 8 * (r >> 1)) ^ li) << 3)];
      }
      __builtin_amdgcn_sched_barrier(0);
      if (MODE == 0 || MODE == 2 || MODE == 3) {
        if (late) {
#pragma unroll
          for (int j = 0; j < NV; ++j) valu_op(v[j & 7], s);
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int j = 0; j < 12; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j & 1], fa[j & 3], acc[j & 3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (MODE != 3 && !late) {
#pragma unroll
          for (int j = 0; j < NV; ++j) valu_op(v[j & 7], s);
        }
      } else if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 12; ++j) {
          acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[j & 1], fa[j & 3], acc[j & 3], 0, 0, 0);
#pragma unroll
          for (int q = (NV * j) / 12; q < (NV * (j + 1)) / 12; ++q) valu_op(v[q & 7], s);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int j = 0; j < NV; ++j) valu_op(v[j & 7], s);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }
  float r = 0.f;
  for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < 8; ++i) r += v[i];
  out[blockIdx.x * 512 + tid] = r;
}

template <int MODE>
static void run(const char* what, int blocks) {
  float* out; hipMalloc(&out, sizeof(float) * 512 * blocks);
  const int groups = blocks == 1 ? 2000 : 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(512), 0, 0, out, groups / 10, 0.5f);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(512), 0, 0, out, groups, 0.5f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-78s blocks %3d: %8.1f ns per 8-k-step group\n", what, blocks, ms * 1e6 / groups);
  hipFree(out);
}

int main() {
  printf("NV = %d half-rate vector instructions (v_max_f32) per k step, 12 MFMA 16x16x32 f16 + 4 ds_read_b128 per k step, 8 waves per workgroup\n", NV);
  for (int blocks : {1, 256}) {
    run<3>("MFMAs + fragment reads only", blocks);
    run<4>("vector instructions only", blocks);
    run<0>("clustered: 12 MFMAs, then the vector block (compiler's order for the epilogue pieces)", blocks);
    run<1>("interleaved: MFMA + 3-4 vector instructions, twelve times", blocks);
    run<2>("clustered, waves 4-7 run the vector block BEFORE their MFMAs (half a k step out of phase)", blocks);
    run<5>("specialised: waves 0-3 all MFMAs (24 per k step), waves 4-7 all vector instructions", blocks);
    run<6>("specialised, vector waves also read 8 x 1 KB from LDS per group", blocks);
  }
  return 0;
}
