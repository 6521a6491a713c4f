// ws_fwd0_experiment.h — per-wave row-owning first-layer kernel.  EXPERIMENT, not part of the product: correct (tools/ws_lab0.hip)
// but at 88 us for the 16-run CQL shape it is slower than the tiled forward (65-75 us): each wave's load -> MFMA -> store chain is
// latency-bound without a prefetch stage.
#pragma once
#include "../offlinerl-kit_amd/csrc/ws_gemm.h"
namespace orl {
// =====================================================================================================================
// ws_fwd0: first layer of a many-row batch, h0 = relu(X W0^T + b0) with in_dim + 1 <= 32 and 256 outputs.
// The launch is bound by writing h0 (1 KB per row), so the kernel is organised around the stores: every WAVE owns whole
// rows.  It keeps W0 (256 x 32 after padding, bias folded in as column `in0` against a ones column of X) as split-bf16 B
// fragments for ALL 256 columns in 128 VGPRs, reads 16 rows of X straight into its A fragment (no LDS, no barriers),
// and for each 16-column block issues 3 MFMAs, applies the ReLU and stores -- the 16 stores of a row group fill complete
// 1 KB rows.  The packed ReLU mask is assembled with two shuffles per block.
// =====================================================================================================================
struct WsFwd0P {
  const float* X; long x_s0, x_s1; int x_pitch, in0;            // inputs [z][M][x_pitch], columns >= in0 are zero padding
  const float* W; long w_s0, w_s1;                               // W0 (256, in0) row-major
  const float* bias; long b_s0, b_s1;
  float* Y; long y_s0, y_s1; int y_pitch;
  unsigned int* mb; long mb_s0, mb_s1; int mb_g;
  int M, nz1, groups;                                            // groups = M / 16
};
enum { WF0_NT = 256 };

__global__ __launch_bounds__(WF0_NT) void ws_fwd0_kernel(const WsFwd0P p) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lq = lane >> 4;
  const int z = blockIdx.z, z0 = z / p.nz1, z1 = z - z0 * p.nz1;
  const float* __restrict__ Xg = p.X + z0 * p.x_s0 + z1 * p.x_s1;
  const float* __restrict__ Wg = p.W + z0 * p.w_s0 + z1 * p.w_s1;
  const float* __restrict__ bg = p.bias + z0 * p.b_s0 + z1 * p.b_s1;
  float* __restrict__ Yg = p.Y + z0 * p.y_s0 + z1 * p.y_s1;
  unsigned int* __restrict__ mbg = p.mb + z0 * p.mb_s0 + z1 * p.mb_s1;

  // resident B fragments: lane (li, lq) supplies W0'[n = 16 nb + li][k = 8 lq + j], W0'[n][in0] = b0[n], zero beyond
  bf16x8 bh[16], bl[16];
#pragma unroll
  for (int nb = 0; nb < 16; ++nb) {
    const int n = 16 * nb + li;
    f32x4 a, b;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k0 = 8 * lq + j, k1 = k0 + 4;
      a[j] = k0 < p.in0 ? Wg[(long)n * p.in0 + k0] : (k0 == p.in0 ? bg[n] : 0.f);
      b[j] = k1 < p.in0 ? Wg[(long)n * p.in0 + k1] : (k1 == p.in0 ? bg[n] : 0.f);
    }
    ws_split8(a, b, bh[nb], bl[nb]);
  }
  const int wstride = gridDim.x * (WF0_NT / 64);
  for (int g = blockIdx.x * (WF0_NT / 64) + wave; g < p.groups; g += wstride) {
    const long m = (long)g * 16 + li;
    // A fragment: X[m][8 lq .. 8 lq + 7] with the ones column at in0 (columns beyond the row pitch are zero)
    f32x4 xa = (f32x4){0.f, 0.f, 0.f, 0.f}, xb = xa;
    if (8 * lq < p.x_pitch) xa = *(const f32x4*)&Xg[m * p.x_pitch + 8 * lq];
    if (8 * lq + 4 < p.x_pitch) xb = *(const f32x4*)&Xg[m * p.x_pitch + 8 * lq + 4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { if (8 * lq + j == p.in0) xa[j] = 1.f; if (8 * lq + 4 + j == p.in0) xb[j] = 1.f; }
    bf16x8 fah, fal;
    ws_split8(xa, xb, fah, fal);
    unsigned int word = 0;
#pragma unroll
    for (int nb = 0; nb < 16; ++nb) {
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};                        // operands swapped: lane holds C[m = li][n = 16 nb + 4 lq + r]
      v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[nb], fah, v, 0, 0, 0);
      v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[nb], fal, v, 0, 0, 0);
      v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[nb], fah, v, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
      *(f32x4*)&Yg[m * p.y_pitch + 16 * nb + 4 * lq] = v;
      // 4 mask bits per lane -> 16 bits per (row, block) over the four lq lanes -> one word per two blocks
      unsigned int w = ((v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u)) << (4 * lq);
      w |= __shfl_xor(w, 16);
      w |= __shfl_xor(w, 32);
      if (nb & 1) { word |= w << 16; if (lq == 0) mbg[m * p.mb_g + (nb >> 1)] = word; }
      else word = w;
    }
  }
}

static inline bool ws_fwd0_supported(const WsFwd0P& p, int N) {
  if (N != WS_N || p.in0 + 1 > 32 || p.in0 >= p.x_pitch || p.x_pitch > 32 || (p.x_pitch & 3) || (p.M & 15) || !p.mb || p.mb_g != 8) return false;
  if (!aligned16(p.X) || (p.x_s0 & 3) || (p.x_s1 & 3) || !aligned16(p.Y) || (p.y_pitch & 3) || (p.y_s0 & 3) || (p.y_s1 & 3)) return false;
  return true;
}
static inline hipError_t launch_ws_fwd0(WsFwd0P p, int nz, hipStream_t st) {
  p.groups = p.M / 16;
  // two 4-wave workgroups per CU: 512 workgroups spread over the nz problems
  int per_z = (512 + nz - 1) / nz;
  const int maxb = (p.groups + 3) / 4;
  if (per_z > maxb) per_z = maxb;
  if (per_z < 1) per_z = 1;
  hipLaunchKernelGGL(ws_fwd0_kernel, dim3(per_z, 1, nz), dim3(WF0_NT), 0, st, p);
  return hipGetLastError();
}

}  // namespace orl
