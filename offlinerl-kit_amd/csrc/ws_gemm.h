// ws_gemm.h — weight-stationary, row-streaming kernels for the 256-wide critic layers of a many-row batch.
//
// The generic tile kernel (gemm.h) spends about half of a workgroup's lifetime in its prologue / epilogue when K is
// only 256 (eight K chunks per tile) and re-reads the weight tile for every row tile.  For the hot shapes of the
// update engine (hidden layers of width 256 evaluated on thousands of rows: CQL's 7936-row critic batch, reference
// cql.py:132-190) the whole weight matrix fits in ONE CU's register file once it is split into 16-bit hi/lo planes
// (256 x 256 x 4 B = 256 KB of the 512 KB VGPR file).  So:
//
//   * a workgroup = 8 waves; wave w owns output columns [32w, 32w+32) and keeps the B fragments of those columns for
//     the complete K = 256 in registers (2 column blocks x 8 k-steps x {hi, lo} x 4 VGPRs = 128 VGPRs), loaded and
//     split ONCE per workgroup;
//   * the workgroup then streams row groups of WS_ROWS = 32 rows: the fp32 rows are fetched with full-row coalesced loads,
//     split into 16-bit hi/lo planes while they are staged into a double-buffered LDS image, and every wave multiplies the
//     shared A fragments against its resident B fragments (16x16x32 16-bit MFMA, lo*hi + hi*lo + hi*hi, fp32
//     accumulation) -- one barrier per row group, no per-tile pipeline fill / drain, no weight traffic in the loop;
//   * the epilogue works on the wave's own 16 x 32 accumulator blocks (bias, ReLU, packed ReLU-mask bits, the fused
//     single-output tail q = h . w_tail + b_tail), so it needs no LDS round trip.
//
// Per row group and CU: 32 KB of HBM reads against 2 x 96 (+ 12 with the fused first layer) MFMAs per SIMD (~3460 matrix-pipe cycles;
// measured ~6200 cycles per group with the vector work), i.e. the kernel sits at the crossover of the HBM and
// matrix-pipe rooflines instead of far below both.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>

#include "gemm.h"

namespace orl {

#ifdef SB_LAB_CLOCK
#define WS_STAMP(i) do { if (p.lab_clk && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.z == 0) p.lab_clk[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WS_STAMP(i) do { } while (0)
#endif

struct WsFwdP {
  const float* X; long x_s0, x_s1; int x_pitch;        // input activations [z][M][K] fp32, K == 256
  const float* W; long w_s0, w_s1;                      // weights of problem z; element (n, k) at W[n * w_sn + k * w_sk]:
  long w_sn, w_sk;                                      //   nn.Linear (out, in): w_sn = 256, w_sk = 1; EnsembleLinear (in, out): w_sn = 1, w_sk = 256
  const float* bias; long b_s0, b_s1;
  float* Y; long y_s0, y_s1; int y_pitch;               // relu(X W^T + b) [z][M][256]
  unsigned int* mb; long mb_s0, mb_s1; int mb_g;        // packed ReLU mask of Y (gemm.h layout) or null
  const float* tw; long tw_s0, tw_s1;                   // fused tail: q[m] = Y[m] . tw + tb  (null = off)
  const float* tb; long tb_s0, tb_s1;
  float* tq; long tq_s0, tq_s1, tq_sm;
  unsigned long long* lab_clk;                          // lab builds (-DSB_LAB_CLOCK): shader-clock stamps of workgroup (0, 0, 0)
  int M, nz1, groups;                                   // groups = ceil(M / WS_ROWS)
  // fused first layer (template L0): X is then PRODUCED here as relu(X0 W0^T + b0) from the narrow input rows X0 (in0 + 1 <= 32 columns
  // incl. the bias as a ones column), stored to `X` for the backward pass, and handed to the second layer through LDS only
  const float* X0; long x0_s0, x0_s1; int x0_pitch, in0;
  int x0_discard;               // fused first layer of a forward-only pass: h0 goes to LDS and its mask bits to HBM, the values are not stored
  const float* W0; long w0_s0, w0_s1, w0_sn, w0_sk;     // element (n, k) at W0[n * w0_sn + k * w0_sk] ((256, in0) row-major: in0, 1)
  const float* b0; long b0_s0, b0_s1;
  unsigned int* mb0; long mb0_s0, mb0_s1; int mb0_g;    // packed ReLU mask of X (= h0)
  // plain dgrad mode (template DG): Y = (X B^T) (.) mask, B given by the strides above (W viewed transposed), no bias / ReLU / mask
  // emission; `dmask` = packed ReLU mask of the activation the gradient flows into
  const unsigned int* dmask; long dm_s0, dm_s1; int dm_g;
  int f32;                                              // exact fp32 arithmetic (v_mfma_f32_16x16x4_f32) instead of the split 16-bit planes
  const float* gscale;                                  // DG, split precision: dynamic power-of-two scale of the gradient rows X, one float per run (z0); null = 1
  // precision 2 (ws_fwd3_kernel: three fp16 planes; fused first layer + fused tail, top activation not stored): a workgroup owns half of the
  // columns, so the tail comes out as two partial sums -- half 0 writes tq (with the tail bias), half 1 tq2[z][m] (k_tail_add folds it in);
  // `dump`: WS_DUMP_SLOTS scratch lines of WS_N floats for the h0 blocks a half computes but does not own
  int np3;
  float* tq2; long tq2_s0, tq2_s1;
  float* dump;
};
enum { WS_DUMP_SLOTS = 8192 };

#ifndef WS_WAVES
#define WS_WAVES 8      // 16 waves (16 columns each) measured slower: the 128-VGPR budget spills
#endif
enum { WS_ROWS = 32, WS_K = 256, WS_N = 256, WS_PITCH = WS_K, WS_NW = WS_WAVES, WS_NT = 64 * WS_NW, WS_CB = WS_N / 16 / WS_NW };   // unpadded rows: 16-byte chunks are XOR-swizzled; WS_CB 16-column blocks per wave
enum { WS_SUB = WS_ROWS / 16, WS_LD = WS_ROWS * WS_K / 4 / WS_NT };
enum { WS_NBP = 68, WS_XLP = 36 };      // byte pitch of a mask-nibble row (64 used) / float pitch of a narrow-input row (32 used): odd multiples of 4 B / 16 B spread the rows over the banks   // 16-row blocks per group; float4 loads per thread per group
// LDS: A image [2 buffers][hi, lo][WS_ROWS][256] bf16 (swizzled) + tail partial sums [2][WS_NW waves][WS_ROWS] floats
//      + ReLU-mask nibbles [2][WS_ROWS][64] bytes
static constexpr size_t ws_fwd_lds_bytes(bool l0 = false) {
  return (size_t)2 * 2 * WS_ROWS * WS_PITCH * 2 + sizeof(float) * 2 * WS_NW * WS_ROWS + 2 * WS_ROWS * WS_NBP +
         (l0 ? sizeof(float) * 2 * WS_ROWS * WS_XLP + 2 * WS_ROWS * WS_NBP : 0) + sizeof(float) * 2 * WS_N;   // + bias / tail weights
}

// host: does the launch qualify?  (K = N = 256, 16-byte aligned operands)
// Launch geometry of one engine's weight-stationary kernels (orl_config::ws_one_round / ws_cus; the ORL_WS_ONE_ROUND / ORL_WS_CUS
// environment variables override them once, at engine creation).  `cus` < 256 leaves room for the launches of other engines' streams to
// run side by side instead of one after the other.
struct WsGeom { int cus = 256; bool one_round = false; };

// Workgroups per problem (net).  One workgroup occupies a CU (512 threads x 256 VGPRs), so a launch runs in ROUNDS of 256 workgroups.
// With `nz` problems of `groups` row groups each, `per_z` workgroups per problem take ceil(nz * per_z / CUs) rounds of
// (prologue + ceil(groups / per_z)) group-times.  The old rule (CUs / nz workgroups, one round) left 64 of the 256 CUs idle at 192 problems
// (96 runs x 2 critics); four workgroups per problem fill three full rounds instead (-16 % by this model: 3 x (10 + 62) against 258).
// `prologue` = the per-workgroup fixed cost in units of one row group (resident-operand load + pipeline fill: ~10 for the forward /
// dgrad kernels; the output-stationary wgrad passes a prohibitive value: its per-workgroup slab write and derived tail gradients cost what
// the idle CUs cost).  Measured, one engine x 96 runs: forward 772 -> 622 us, dgrad 520 -> 439 us, 39.3k -> 41.7k steps/s.
// one_round restores the old rule: with TWO engines per GPU the idle CUs of one engine's launch are where the other engine's kernels
// run, and filling them costs more than it gains (2 x 96 runs: 48.0k one round, 44.3k whole rounds) -- bench.py asks for it then.
static inline int ws_blocks_per_problem(int groups, int nz, int prologue, int cap, const WsGeom& geo) {
  const int cus = geo.cus;
  int base = cus / nz;
  if (base < 1) base = 1;
  if (base > groups) base = groups;
  if (base > cap) base = cap;
  if (geo.one_round) return base;
  int best = base;
  long best_cost = (long)((nz * (long)base + cus - 1) / cus) * (prologue + (groups + base - 1) / base);
  for (int pz = base + 1; pz <= 4 * base + 4 && pz <= groups && pz <= cap; ++pz) {
    const long cost = (long)((nz * (long)pz + cus - 1) / cus) * (prologue + (groups + pz - 1) / pz);
    if (cost < best_cost) { best_cost = cost; best = pz; }
  }
  return best;
}

static inline bool ws_fwd_supported(const WsFwdP& p, int K, int N) {
  if (K != WS_K || N != WS_N || p.M < 256 || (p.M % WS_ROWS) || (!p.mb && !p.dmask)) return false;
  if (!aligned16(p.X) || (p.x_pitch & 3) || (p.x_s0 & 3) || (p.x_s1 & 3)) return false;
  if (p.w_sk == 1 && (!aligned16(p.W) || (p.w_s0 & 3) || (p.w_s1 & 3) || (p.w_sn & 3))) return false;
  if (p.dmask) { if (p.tq || p.X0 || p.dm_g != 8) return false; }
  else if (!aligned16(p.bias) || (p.b_s0 & 3) || (p.b_s1 & 3)) return false;
  if (!p.Y && (!p.tq || !p.mb)) return false;      // the activation may stay unstored only when the tail is folded in
  if (!aligned16(p.Y) || (p.y_pitch & 3) || (p.y_s0 & 3) || (p.y_s1 & 3)) return false;
  if (p.tq && (!aligned16(p.tw) || (p.tw_s0 & 3) || (p.tw_s1 & 3))) return false;
  return true;
}

static inline bool ws_fwd01_supported(const WsFwdP& p) {      // extra conditions of the fused first layer
  if (!p.X0 || !p.mb0 || p.mb0_g != 8 || p.in0 + 1 > 32 || p.in0 >= p.x0_pitch || p.x0_pitch > 32 || WS_ROWS * p.x0_pitch > 2 * WS_NT) return false;
  return true;
}
hipError_t launch_ws_fwd(WsFwdP p, int nz, hipStream_t st, const WsGeom& geo);      // ws_fwd.hip
// precision 2: LDS = A image [2][hi, mid, lo][32][256] + tail partial sums + mask nibbles of h1 / h0 + narrow input rows (three planes) + bias / tail weights
static constexpr size_t ws_fwd3_lds_bytes() {
  return (size_t)2 * 3 * WS_ROWS * WS_PITCH * 2 + sizeof(float) * 2 * WS_NW * WS_ROWS + 2 * WS_ROWS * WS_NBP + sizeof(float) * 2 * WS_ROWS * 52 +
         2 * WS_ROWS * WS_NBP + sizeof(float) * 2 * WS_N;
}
static inline bool ws_fwd3_supported(const WsFwdP& p, int K, int N) {
  if (!ws_fwd_supported(p, K, N)) return false;
  if (p.dmask) return p.Y != nullptr;                                             // plain dgrad mode (ws_fwd_supported: no tail, no fused first layer, 8 mask words per row)
  if (p.tq && !p.tq2) return false;
  const bool tq = p.tq != nullptr, sy = p.Y != nullptr, xs = !p.x0_discard;      // the flavours the engine's passes use (ws_fwd3.hip)
  if (!p.X0) return (tq || sy) && (p.M % WS_ROWS) == 0;                           // input rows from HBM
  if (!ws_fwd01_supported(p) || !p.dump) return false;
  return (tq && !sy) || (tq && sy && xs) || (!tq && sy);
}
hipError_t launch_ws_fwd3(WsFwdP p, int nz, int per_z, hipStream_t st);              // ws_fwd3.hip (grid.y = 2 column halves)

// =====================================================================================================================
// ws_dgrad_w0: backward through the top hidden layer of a single-output net, fused with the layer-0 weight gradient.
//
//   dz0[m][n] = 1[h0[m][n] > 0] * dq[m] * sum_k 1[h1[m][k] > 0] * (w_tail[k] * W1[k][n])          (reference: autograd of
//   dW0[n][c] = sum_m dz0[m][n] * X[m][c],   db0[n] = sum_m dz0[m][n]                               critic_module.py:17-28)
//
// Same weight-stationary structure as ws_fwd: wave w keeps B'[n][k] = w_tail[k] * W1[k][n] for its 32 columns n and all
// 256 k as split-bf16 fragments in registers.  The A operand is the ReLU mask of h1, i.e. exactly 0 / 1 in bf16: it is
// expanded from the packed mask bits straight into the swizzled LDS image (1 KB of HBM per 32 rows instead of 32 KB) and
// needs no lo plane, so a block costs 2 MFMAs instead of 3.  The accumulators come out as D[m][n] with four consecutive rows
// per lane -- which is precisely the B-operand layout of v_mfma_f32_16x16x16_bf16 -- so after the dq scale and the h0 mask
// they are fed, still in registers, into dW0^T[c][n] += X^T[c][m] dz0[m][n] (X^T staged in LDS as split bf16, column
// `in0` = 1 gives db0).  dW0 accumulates in 16 VGPRs over ALL row groups of the workgroup and is written once, as one
// split-K slab per workgroup.  dz0 itself never exists outside registers.
// =====================================================================================================================
struct WsDgradP {
  const unsigned int* abits; long ab_s0, ab_s1; int ab_g;     // mask words of the top hidden activation (K = 256 columns)
  const unsigned int* xbits; long xb_s0, xb_s1; int xb_g;     // mask words of the layer-0 activation (N = 256 columns)
  const float* dq; long dq_s0, dq_s1, dq_sm;                   // dLoss/dq per row
  const float* wt; long wt_s0, wt_s1;                          // w_tail [256]
  const float* W; long w_s0, w_s1, w_sn, w_sk;                 // W1: element (k = output unit, n = input unit) at W[n * w_sn + k * w_sk]
                                                               //   nn.Linear (out, in): w_sn = 1, w_sk = 256; EnsembleLinear (in, out): 256, 1
  const float* X; long x_s0, x_s1; int x_pitch, in0;           // layer-0 input rows [M][x_pitch], in0 + 1 <= 32
  float* w0_out; float* b0_out; long o_s0, o_s1, ob_s1, o_ks; int o_sr, o_sc;   // slab outputs (dW0 element (unit n, input c) at n * o_sr + c * o_sc: nn.Linear (in0, 1), EnsembleLinear (1, 256); db0 [256]); W0 variant
  float* C; long c_s0, c_s1; int c_pitch;                      // dz0 [M][256]; STORE variant
  // PLAIN variant (Z != nullptr; W0 only): the incoming gradient is a materialised matrix dz1 [M][256] (a hidden layer below the top
  // one) instead of (mask bits, dq, w_tail):  dz0 = 1[h0 > 0] * (dz1 W1),  dW0 / db0 as above.  The A image gets a lo plane (three
  // products per block), B' = W1 itself.
  const float* Z; long z_s0, z_s1; int z_pitch;
  int M, nz1, groups;
  int f32;                                                     // exact fp32 arithmetic (ws_dgrad32_w0_kernel) instead of the split 16-bit planes
  const float* gscale;                                         // split precision: dynamic power-of-two scale applied to dq / dz1, one float per run (z0); null = 1
};
enum { WD_XP = WS_ROWS + 4 };                                       // bf16 pitch of an X^T row (72 B: scattered 2-byte stores and 8-byte reads spread over the banks)
static constexpr size_t ws_dgrad_lds_bytes(bool plain = false) {     // mask images (plain: hi + lo planes of dz1) + X^T images + per-group epilogue operands (dq, h0 mask words)
  return (size_t)(plain ? 4 : 2) * WS_ROWS * WS_PITCH * 2 + (size_t)2 * 2 * 32 * WD_XP * 2 + (size_t)2 * (WS_ROWS + WS_NW * WS_ROWS) * 4;
}

static inline bool ws_dgrad_supported(const WsDgradP& p, int K, int N) {
  if (K != WS_K || N != WS_N || p.M < 256 || (p.M % WS_ROWS)) return false;
  if (p.Z) {
    if (!p.w0_out || p.C || !p.xbits || p.xb_g != 8) return false;
    if (!aligned16(p.Z) || (p.z_pitch & 3) || (p.z_s0 & 3) || (p.z_s1 & 3)) return false;
    return !(p.in0 + 1 > 32 || p.in0 >= p.x_pitch || p.x_pitch > 32 || WS_ROWS * p.x_pitch > 2 * WS_NT);
  }
  if (!p.abits || !p.xbits || p.ab_g != 8 || p.xb_g != 8) return false;
  if (p.w0_out && (p.in0 + 1 > 32 || p.in0 >= p.x_pitch || p.x_pitch > 32 || WS_ROWS * p.x_pitch > 2 * WS_NT)) return false;
  if (!p.w0_out && !p.C) return false;
  if (!aligned16(p.wt) || (p.wt_s0 & 3) || (p.wt_s1 & 3)) return false;
  return true;
}
// blocks per problem (= split-K slabs written per problem)
static inline int ws_dgrad_blocks(int M, int nz, int max_slab, const WsGeom& geo, int prologue = 10) {
  return ws_blocks_per_problem(M / WS_ROWS, nz, prologue, max_slab, geo);
}
hipError_t launch_ws_dgrad_w0(WsDgradP p, int nz, int per_z, hipStream_t st);      // ws_dgrad.hip
// precision 2 (three fp16 planes): the W0 flavour from mask bits only; a workgroup owns half of the net's columns (grid.y = 2), so the launch has
// 2 * per_z workgroups per problem and per_z split-K slabs (ws_dgrad3.hip)
static constexpr size_t ws_dgrad3_lds_bytes(bool plain = false) {      // mask image (plain: three planes of dz1) + X^T images (three planes) + epilogue operands
  return (size_t)2 * (plain ? 3 : 1) * WS_ROWS * WS_PITCH * 2 + (size_t)2 * 3 * 32 * WD_XP * 2 + (size_t)2 * (WS_ROWS + WS_NW * WS_ROWS) * 4;
}
static inline bool ws_dgrad3_supported(const WsDgradP& p, int K, int N) {      // from mask bits: the W0 flavour (nothing stored) or the storing one; or the plain W0 flavour
  if (!ws_dgrad_supported(p, K, N)) return false;
  if (p.Z) return true;                                                        // (ws_dgrad_supported checked w0_out / !C / xbits)
  return (p.w0_out && !p.C) || (!p.w0_out && p.C);
}
hipError_t launch_ws_dgrad3_w0(WsDgradP p, int nz, int per_z, hipStream_t st);     // ws_dgrad3.hip

// =====================================================================================================================
// ws_wgrad: weight gradient of the top hidden layer of a single-output net, output-stationary (+ the tail layer's gradients).
//
//   dW1[k][n] = w_tail[k] * sum_m 1[h1[m][k] > 0] * (dq[m] * h0[m][n])        db1[k] = w_tail[k] * sum_m 1[h1[m][k] > 0] * dq[m]
//   TAILS variant: dw_tail[k] = sum_m dq[m] h1[m][k], db_tail = sum_m dq[m] and db1 from h1 streamed through registers (VALU), in
//   the shadow of the MFMAs -- no separate HBM-bound launch for them
//
// The 256 x 256 result stays in registers for the whole launch: wave w owns columns n in [32w, 32w+32) and all 256 rows k
// (16 x 2 blocks of 16 x 16 = 128 accumulator VGPRs) and the workgroup streams 32-row groups of the batch.  Both MFMA
// operands are "transposed" views of row-major data (the reduction runs over the rows m), which is exactly what
// ds_read_b64_tr_b16 delivers from row-major LDS images: A = the 0/1 ReLU mask of h1 expanded from its packed bits (exact in
// bf16, no lo plane), B = G = dq (.) h0 split into bf16 hi/lo while it is staged.  One v_mfma_f32_16x16x32_bf16 covers a whole
// 32-row group (its 8 k-values per lane are two transposed reads; the k order is free as long as A and B agree), two per block;
// dependent MFMAs are kept four instructions apart and every staging register is written to LDS and refilled at the same point
// of each iteration (a full iteration in flight).  One split-K slab per workgroup.
// =====================================================================================================================
struct WsWgradP {
  const unsigned int* abits; long ab_s0, ab_s1; int ab_g;     // mask words of the top hidden activation h1
  const float* dq; long dq_s0, dq_s1, dq_sm;
  const float* H0; long h0_s0, h0_s1; int h0_pitch;            // input of the top hidden layer [M][256]
  const float* wt; long wt_s0, wt_s1;                          // w_tail [256]
  float *dW, *db;                                              // slab outputs; run stride o_s0, member strides below, slab stride o_ks
  long o_s0, o_s1w, o_s1b, o_ks;
  // TAILS variant: h1 itself is streamed too (through registers only) and the launch also produces the tail layer's gradients
  //   dw_tail[k] = sum_m dq[m] h1[m][k],  db_tail = sum_m dq[m];  db1 then comes from the same pass (no MFMA operand for it)
  const float* H1; long h1_s0, h1_s1; int h1_pitch;
  float *dwt, *dbt; long o_s1wt, o_s1bt;
  // DERIVED variant (H1 == nullptr, W1 != nullptr): h1 was never stored.  With G[n][k] = sum_m dq[m] 1[h1[m][n] > 0] h0[m][k] (the
  // accumulators before the w_tail scaling) and g[n] = sum_m dq[m] 1[h1[m][n] > 0], h1 = relu(h0 W1^T + b1) gives
  //   dw_tail[n] = sum_m dq[m] h1[m][n] = sum_k W1[n][k] G[n][k] + b1[n] g[n]      (linear in G, so it holds per slab)
  const float* W1; long w1_s0, w1_s1;                          // [256][256] (out, in) row-major
  const float* b1; long b1_s0, b1_s1;
  // PLAIN variant (dZ != nullptr): the gradient w.r.t. this layer's output is a materialised matrix (a hidden layer below the top one):
  //   dW[k][n] = sum_m dZ[m][k] * H0[m][n],  db[k] = sum_m dZ[m][k]
  // Same output-stationary structure; the A operand has a lo plane now (three products per block instead of two), no mask, no w_tail.
  const float* dZ; long dz_s0, dz_s1; int dz_pitch;
  // RECOMPUTE variant of PLAIN (X0 != nullptr): H0 = relu(X0 W0^T + b0) was NOT stored by the forward pass (a three-layer net's first hidden
  // activation: 2 GB written and 2 GB read per step at 128 runs for a matrix that 12 MFMAs per wave and row group rebuild from a 24-column
  // input).  Every wave reproduces its own 32 columns of the group with the forward's instruction sequence (bit-identical values) straight
  // into the H image.  in0 + 1 <= 32, x0_pitch <= 32.
  const float* X0; long x0_s0, x0_s1; int x0_pitch, in0;
  const float* W0; long w0_s0, w0_s1, w0_sn, w0_sk;
  const float* b0; long b0_s0, b0_s1;
  unsigned long long* lab_clk;                                 // lab builds (-DSB_LAB_CLOCK): shader-clock stamps of workgroup (0, 0, 0)
  int M, nz1, groups;
  int f32;                                                     // exact fp32 arithmetic (ws_wgrad32_kernel) instead of the split 16-bit planes
  int np3;                                                     // precision 2: three fp16 planes of G (ws_wgrad_kernel<5>; derived-tail flavour only), overrides f32
  const float* gscale;                                         // split precision: dynamic power-of-two scale applied to dq / dZ, one float per run (z0); null = 1
};
enum { WW_IMG = WS_ROWS * WS_K };                               // bf16 elements of one [32][256] LDS image
static constexpr size_t ws_wgrad_lds_bytes(bool plain = false, bool recompute = false, bool p3 = false) {   // 2 buffers x {mask, G hi, G lo} (plain: {dZ hi, dZ lo, H hi, H lo}; p3: {mask, G hi, G mid, G lo}) + dq / ones blocks (+ the narrow input rows)
  return (size_t)2 * ((plain || p3) ? 4 : 3) * WW_IMG * 2 + (size_t)2 * (p3 ? 3 : 2) * WS_ROWS * 16 * 2 + (recompute ? sizeof(float) * 2 * WS_ROWS * WS_XLP : 0);
}

static inline bool ws_wgrad_supported(const WsWgradP& p, int K, int N) {
  if (p.dZ) {
    if (K != WS_K || N != WS_N || p.M < 256 || (p.M % WS_ROWS)) return false;
    if (!aligned16(p.dZ) || (p.dz_pitch & 3) || (p.dz_s0 & 3) || (p.dz_s1 & 3)) return false;
    if (p.X0) return p.W0 && p.b0 && p.in0 + 1 <= 32 && p.in0 < p.x0_pitch + 1 && p.x0_pitch <= 32 && p.in0 <= p.x0_pitch && WS_ROWS * p.x0_pitch <= 2 * WS_NT;
    return aligned16(p.H0) && !(p.h0_pitch & 3) && !(p.h0_s0 & 3) && !(p.h0_s1 & 3);
  }
  if (K != WS_K || N != WS_N || p.M < 256 || (p.M % WS_ROWS) || !p.abits || p.ab_g != 8) return false;
  if (p.np3 && (p.H1 || !p.W1)) return false;
  if (!aligned16(p.H0) || (p.h0_pitch & 3) || (p.h0_s0 & 3) || (p.h0_s1 & 3)) return false;
  if (!p.H1 && p.W1 && (!p.b1 || !p.dwt || !p.dbt)) return false;
  if (p.H1 && (!aligned16(p.H1) || (p.h1_pitch & 3) || (p.h1_s0 & 3) || (p.h1_s1 & 3) || !p.dwt || !p.dbt)) return false;
  return aligned16(p.wt) && !(p.wt_s0 & 3) && !(p.wt_s1 & 3);
}
hipError_t launch_ws_wgrad(WsWgradP p, int nz, int per_z, hipStream_t st);      // ws_wgrad.hip
// precision 2, plain (materialised dZ) flavour: three planes of both operands; a workgroup owns half of the output rows (grid.y = 2)
// (ws_wgrad3p.hip).  LDS: 2 buffers x {3 x [32][128] dZ planes, 3 x [32][256] H0 planes} + the ones block
static constexpr size_t ws_wgrad3p_lds_bytes() { return (size_t)2 * (3 * WS_ROWS * 128 + 3 * WS_ROWS * WS_K) * 2 + (size_t)WS_ROWS * 16 * 2; }
static inline bool ws_wgrad3p_supported(const WsWgradP& p, int K, int N) {
  return p.dZ && !p.X0 && ws_wgrad_supported(p, K, N);
}
hipError_t launch_ws_wgrad3p(WsWgradP p, int nz, int per_z, hipStream_t st);     // ws_wgrad3p.hip

}  // namespace orl
