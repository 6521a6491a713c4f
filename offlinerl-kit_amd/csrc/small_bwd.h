// small_bwd.h — the SAC-style actor update of FEW batched rows as one launch: actor loss + temperature step + backward of the tanh-Gaussian
// head + backward of the [in0 -> 256 -> 256 -> 2A] actor, one workgroup per 32-row group.
//
// With one to sixteen runs per engine the tiled path spends eight kernel nodes here (k_actor_loss, k_head_bwd, tail wgrad, tail dgrad, wgrad1,
// dgrad1, wgrad0 -- 5 - 17 us each, every one a round trip through HBM for a 32 KB .. 256 KB operand).  Everything between the critics'
// action gradients and the actor's weight gradients is local to a batch row except the sums over the batch, so a workgroup keeps its 32 rows
// on chip: dhead (VALU) -> dz1 = (dhead W_head) (.) 1[h1 > 0] (VALU, K = 2A) -> dz0 = (dz1 W1) (.) 1[h0 > 0] (MFMA, W1 streamed through LDS as
// in small_fwd.h's QG mode) -> dW1 = dz1^T h0 (MFMA over the 32 rows, both operands through transposing LDS reads), dW0 = dz0^T [x | 1],
// dW_head = dhead^T h1 (VALU).  Every workgroup writes ONE split-K slab of every actor tensor (slab index = its row group; k_adam sums them in
// slab order); the loss / entropy sums go through per-workgroup partials and a ticket: the workgroup that arrives last finishes the run
// (metrics, Adam step of log alpha) in a fixed summation order.
//
// Reference: cql.py:92-106 / sac.py:107-124 (actor loss, alpha loss), dist_module.py:17-42, 117-127 (tanh-Gaussian rsample / log_prob, whose
// backward is oracle/nn.py tanh_gauss_bwd), nets/mlp.py:9-33, modules/actor_module.py:22-27.
#pragma once
#include <hip/hip_runtime.h>
#include "gemm.h"

namespace orl {

struct RunScalars;
struct Hyper;

struct SmallABwdP {
  const float* X; long x_s0; int x_pitch, in0;                 // actor input rows [M][x_pitch] (observations), in0 + 1 <= 32
  const float* H0; long h0_s0;                                  // [M][256] hidden activations of the forward pass (post-ReLU)
  const float* H1; long h1_s0;
  const float* head; long head_s0;                              // [M][2A] = [mu | log sigma (unclamped)]
  const float* eps; long eps_s0;                                // [M][A]
  const float* xa; long xa_s0; int xa_pitch, xa_col;            // sampled actions a = tanh(u) at xa[m][xa_col + a]
  const float* logp; long logp_s0;                              // [M]
  const float* qa; long qa_s0, qa_s1;                           // [K][M] critic values Q_k(s, a)
  const float* ga; long ga_s0, ga_s1; int ga_pitch;             // [K][M][ga_pitch] unit-seed gradients dQ_k / da
  int K;                                                        // critics (2)
  const float* W1; long w1_s0;                                  // actor layer 1 (256, 256) row-major
  const float* Wh; long wh_s0;                                  // head (2A, 256) row-major
  // split-K slabs (slab = row group): tensor t of the run at out + z0 * o_s0 + slab * o_ks + off_t
  float* out; long o_s0, o_ks;
  long off_w0, off_b0, off_w1, off_b1, off_wh, off_bh;
  // loss / temperature
  RunScalars* sc; const Hyper* hy;
  int auto_alpha; float fixed_alpha, target_entropy; int clamp_alpha01;
  float b1, b2, adam_eps;
  const unsigned long long* gstep;
  float* metrics_last; float* metrics_sum; int nm;
  int m_actor, m_alpha_loss, m_alpha;
  float* part; unsigned int* ticket;                            // [R][groups][2] partial sums, [R] arrival counters (zero between launches)
  int M, A, f32;
  unsigned long long* lab_clk;                                  // lab builds (-DSB_LAB_CLOCK): shader-clock stamps of workgroup (0, 0), else unused
};
enum { SB_ROWS = 32, SB_N = 256, SB_NT = 512, SB_MAXGROUPS = 64 };

static inline bool small_abwd_supported(const SmallABwdP& p) {
  if (p.M < SB_ROWS || (p.M % SB_ROWS) || p.M / SB_ROWS > SB_MAXGROUPS || p.in0 + 1 > 32 || p.x_pitch > 32 || p.in0 > p.x_pitch) return false;
  if (p.A < 1 || p.A > 8 || p.K != 2 || p.ga_pitch < p.A) return false;
  if (!aligned16(p.W1) || (p.w1_s0 & 3) || !aligned16(p.H0) || (p.h0_s0 & 3) || !aligned16(p.H1) || (p.h1_s0 & 3)) return false;
  if (!aligned16(p.Wh) || (p.wh_s0 & 3)) return false;
  return true;
}
hipError_t launch_small_abwd(const SmallABwdP& p, int runs, hipStream_t st);      // small_bwd.hip

}  // namespace orl
