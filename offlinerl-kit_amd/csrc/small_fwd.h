// small_fwd.h — one launch for the whole forward pass of a [in0 -> 256 -> 256 -> out] net on FEW batched rows.
//
// The weight-stationary kernels (ws_gemm.h) pay a ~20 us floor per launch (256 KB of resident weights gathered per workgroup) and the
// tiled kernels (gemm_kernel.h) need three launches per pass (layer 0, layer 1, tail) at 7 - 11 us each: with one to eight runs per
// engine a CQL step is mostly such 256-row passes (actor, critic(s, pi(s)), actor on [s; s'], target critics) and each kernel node of a
// graph costs its ~4 us whatever it computes.  Here a workgroup walks ONE 32-row group through all three layers: the rows' activations stay
// in LDS between the layers, the weights stream through LDS in 32-wide k chunks (coalesced 128-byte reads, shared through L2 by the
// workgroups of a net), layer 0 runs as one more chunk with the bias folded in as a ones column, and the tail (1 .. 16 outputs) is a
// vector-ALU dot product of the second layer's accumulators reduced across the eight waves.
//
// Reference: nets/mlp.py:9-33 (MLP.forward), modules/critic_module.py:17-28, modules/actor_module.py:22-27 (backbone + last Linear).
#pragma once
#include <hip/hip_runtime.h>
#include "gemm.h"
#include "sample.h"

namespace orl {

struct SmallFwdP {
  const float* X; long x_s0, x_s1; int x_pitch, in0;           // input rows [M][x_pitch], in0 + 1 <= 32 (column in0 becomes the ones column)
  const float* W0; long w0_s0, w0_s1;                           // (256, in0) row-major
  const float* b0; long b0_s0, b0_s1;
  const float* W1; long w1_s0, w1_s1;                           // (256, 256) row-major: element (n = output unit, k) at n * 256 + k
  const float* b1; long b1_s0, b1_s1;
  const float* Wt; long wt_s0, wt_s1;                           // (out_dim, 256) row-major
  const float* bt; long bt_s0, bt_s1;
  float* H0; long h0_s0, h0_s1;                                 // [M][256] (pitch 256); nullptr: not stored (forward-only pass)
  float* H1; long h1_s0, h1_s1;
  float* OUT; long o_s0, o_s1; int o_pitch, out_dim;            // tail output [M][o_pitch], out_dim <= 16
  int M, nz1;
  int f32;                                                      // exact fp32 MFMA instead of the split 16-bit planes
  // QG mode (G != nullptr; single-output nets, i.e. critics): forward AND backward of the row group for a unit seed in the same launch --
  // OUT = q [M], G[m][a] = dq[m] / dx[m][gc0 + a], a < gn <= 8 (the action columns of the critic input: what the actor loss differentiates,
  // cql.py:93-98).  The hidden activations never leave the workgroup: ReLU masks stay in registers, dz1 = w_tail (.) 1[h1 > 0] goes back into
  // the LDS image the forward used, W1 streams through LDS a second time as 32-row chunks read with transposing LDS loads.
  float* G; long g_s0, g_s1; int g_pitch, gc0, gn;
  // optional epilogue: the tanh-Gaussian sampling jobs that consume this pass's head rows (k_tanh_sample's arithmetic on the rows the
  // workgroup just produced: one kernel node less per actor pass).  nz1 == 1, out_dim == 2 A.
  int njobs, A;
  SampleJob job[3];
  unsigned long long* lab_clk;                                  // lab builds (-DSB_LAB_CLOCK): shader-clock stamps of workgroup (0, 0, 0), else unused
};
enum { SF_ROWS = 32, SF_N = 256, SF_NT = 512, SF_MAXOUT = 16 };

static inline bool small_fwd_supported(const SmallFwdP& p) {
  if (p.M < SF_ROWS || (p.M % SF_ROWS) || p.in0 + 1 > 32 || p.x_pitch > 32 || p.in0 > p.x_pitch || p.out_dim < 1 || p.out_dim > SF_MAXOUT) return false;
  if (!aligned16(p.W1) || (p.w1_s0 & 3) || (p.w1_s1 & 3)) return false;
  if (p.H0 && (!aligned16(p.H0) || (p.h0_s0 & 3) || (p.h0_s1 & 3))) return false;
  if (p.H1 && (!aligned16(p.H1) || (p.h1_s0 & 3) || (p.h1_s1 & 3))) return false;
  if (p.njobs && (p.njobs > 3 || p.nz1 != 1 || p.out_dim != 2 * p.A || p.A > 8 || p.G)) return false;
  for (int i = 0; i < p.njobs; ++i) if (p.job[i].rep < 1 || p.job[i].rep > 16) return false;      // (the epilogue prefetches its noise: rep / 2 <= 8 values per thread and job)
  if (p.G && (p.out_dim != 1 || p.gn < 1 || p.gn > 8 || p.gc0 < 0 || p.gc0 + p.gn > p.in0 || p.g_pitch < p.gn)) return false;
  return true;
}
hipError_t launch_small_fwd(const SmallFwdP& p, int nz, hipStream_t st);      // small_fwd.hip

}  // namespace orl
