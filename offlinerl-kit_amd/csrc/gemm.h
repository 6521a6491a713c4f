// gemm.h — run-batched MFMA GEMM for the MLP forward / dgrad / wgrad of the update engine.
//
// One template covers every matrix product of the hot path (reference: nets/mlp.py:9-33 forward,
// autograd's mm/addmm/threshold_backward for the backward):
//
//     C[z][m][n] (+slab) = epi( sum_k  pro_a(A[z][m][k]) * pro_b(B[z][n][k]) )
//
// A and B are addressed with arbitrary element strides (row stride, k stride), so the same kernel does
//   forward  Y  = X  W^T      A = X [M x K] k-contiguous,  B = W  [N x K] k-contiguous
//   dgrad    dX = dY W        A = dY,                      B = W viewed as [K' x N'] (row-contiguous)
//   wgrad    dW = dY^T X      A = dY viewed [N' x M],      B = X viewed [K' x M]     (both row-contiguous)
// Tiles are staged global -> registers -> LDS as [row][k] (k contiguous, pitch TK+4), prefetching the next
// K chunk into registers while the current one is multiplied.  Three branch-free loaders, chosen per
// operand on the host from strides / alignment:
//   L_VECK   k-contiguous, 16 B aligned : one global_load_dwordx4 per 4 k, one ds_write_b128
//   L_BLK4   row-contiguous, aligned    : a 4(row) x 4(k) block per slot: four dwordx4 loads along the rows,
//                                         transposed in registers, four ds_write_b128 along k
//   L_VECKU  k-contiguous, unaligned    : same 4-k slots and wide LDS stores, but four dword loads (e.g. a
//                                         first-layer weight matrix whose rows are 23 floats long)
//   L_SCALAR anything else              : one dword per slot, clamped address + select (no divergent branches)
// Rows beyond M/N are read from a clamped (valid) row and discarded by the epilogue; only the K tail is
// zero-filled.  The multiply is v_mfma_f32_16x16x4_f32 (exact fp32, 64 FLOP/clk/SIMD on gfx950): lane
// (i = l&15, q = l>>4) reads four consecutive k of row i with one ds_read_b128 and feeds them to four MFMAs,
// i.e. hardware k-slot q of MFMA s carries logical k = 4q + s for both operands (any permutation of k is a
// valid reduction order).  Wavefront = 64 lanes; a workgroup is WM x WN waves, each wave owns an
// (MA*16) x (NB*16) block of C.
//
// z = blockIdx.z is the run/net batch index, decomposed z = z0 * nz1 + z1 (run, member) with two strides per
// operand, so twin critics / ensembles and all runs of an engine go through one launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <atomic>

namespace orl {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- 16-bit operand type of the split-precision multiply (P_SPLIT) ----
// Default: IEEE half (fp16) hi + lo planes = 22 significand bits per operand on v_mfma_f32_16x16x32_f16 (same cycles as the bf16 form).
// fp16 has only 5 exponent bits, so every operand class carries a power-of-two scale that is folded back into the fp32 accumulators
// (weights: static; gradient matrices: one dynamic scale per run and backward pass, k_grad_scale) -- exact, and it keeps hi in the normal
// range and lo = x - hi on the 2^-24 subnormal grid or better (gfx950's f16 MFMA multiplies subnormals exactly and v_cvt_pk_f16_f32
// rounds onto that grid: tools/fp16_split_probe.hip, profiles/r03_fp16_split_probe.txt).  Measured on 256-deep dot products: fp16
// hi + lo with scales 2 - 3e-7 of the result scale = an fp32 fmaf chain's own error; bf16 hi + lo 4 - 5e-6.
// -DORL_SPLIT_BF16 (build.py --variant) keeps the round-2 bf16 planes (8 + 8 bits, no range limit) for A/B runs.
#ifdef ORL_SPLIT_BF16
typedef __bf16 hx_t;
#define ORL_SPLIT_BITS 16
#define ORL_HX_ONE_BITS 0x3F80u                      // 1.0 as a 16-bit pattern
#define ORL_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#else
typedef _Float16 hx_t;
#define ORL_SPLIT_BITS 22
#define ORL_HX_ONE_BITS 0x3C00u
#define ORL_MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#endif
typedef hx_t hx4 __attribute__((ext_vector_type(4)));
typedef hx_t hx8 __attribute__((ext_vector_type(8)));
// static operand scales of the split multiply (powers of two; 1 in the bf16 build would do as well, they are exact either way)
#define ORL_WSCALE 64.0f             // weight matrices (|w| < 1023): U(+-1/16)-sized weights land at O(1)
#define ORL_WWSCALE 256.0f           // products w_tail[k] * W1[k][n] held by ws_dgrad_w0 (|.| < 255)
// tiled three-plane launches (P_SPLIT3): static factor on top of a gradient matrix's dynamic scale (which puts the seed's largest element at
// 8 .. 16 and leaves 2^12 for growth through the layers): the third plane of an element 2^-9 below the largest one would otherwise sit on
// fp16's 2^-24 grid with two or three bits.  2^5 keeps 2^7 of growth headroom below 65504.
#define ORL_GSCALE3 32.0f

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
#ifdef ORL_SPLIT_BF16
#define ORL_MFMA_16x16x16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0)       // operands as s16x4 bit patterns
// x = hi + lo with hi = bf16(x) (round to nearest even) and lo = bf16(x - hi): twelve VALU instructions for four values (two packed
// conversions per plane, the widening of hi back to fp32 done on the packed words) -- the element-wise form compiled to 20
__device__ __forceinline__ void orl_split4(const f32x4& v, hx4& h, hx4& l) {
  h = __builtin_convertvector(v, hx4);
  const u32x2_t hb = *(const u32x2_t*)&h;
  f32x4 hf;
  hf[0] = __uint_as_float(hb[0] << 16); hf[1] = __uint_as_float(hb[0] & 0xffff0000u);
  hf[2] = __uint_as_float(hb[1] << 16); hf[3] = __uint_as_float(hb[1] & 0xffff0000u);
  l = __builtin_convertvector(v - hf, hx4);
}
#else
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
#define ORL_MFMA_16x16x16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x16f16(*(const f16x4_t*)&(a), *(const f16x4_t*)&(b), c, 0, 0, 0)
// x = hi + lo with hi = fp16(x) (round to nearest even, v_cvt_pk_f16_f32) and lo = fp16(x - hi); the remainder x - hi is exact in fp32.
// The remainder is one v_fma_mix_f32 per element (fma(float(half), -1, x) reading the half straight out of the packed hi word) instead of
// v_cvt_f32_f16 + v_sub_f32: 8 vector instructions for four values, all of the 4.4-cycle class, against 12 (4 of them full rate) --
// measured issue costs tools/coexec_probe.hip VKIND 13-16, profiles/r03_fp16_split_probe.txt: -21 % per split, 5 % below the bf16 split.
// The compiler does not form the mix instruction on its own (it folds the fma back into convert + subtract).
__device__ __forceinline__ void orl_split4(const f32x4& v, hx4& h, hx4& l) {
  h = __builtin_convertvector(v, hx4);
#ifdef ORL_SPLIT_NO_MIX      // A/B build: convert + subtract
  l = __builtin_convertvector(v - __builtin_convertvector(h, f32x4), hx4);
  return;
#endif
  const u32x2_t hb = *(const u32x2_t*)&h;
  f32x4 r;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r[0]) : "v"(hb[0]), "v"(v[0]));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r[1]) : "v"(hb[0]), "v"(v[1]));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r[2]) : "v"(hb[1]), "v"(v[2]));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r[3]) : "v"(hb[1]), "v"(v[3]));
  l = __builtin_convertvector(r, hx4);
}
#endif
__device__ __forceinline__ void orl_split1(float x, hx_t& h, hx_t& l) { h = (hx_t)x; l = (hx_t)(x - (float)h); }
// Three planes (precision 2): x = hi + mid + lo with hi = half(x), mid = half(x - hi), lo = half(x - hi - mid); both remainders are exact in
// fp32, so the three fp16 planes carry 33 significand bits wherever the last plane stays on fp16's 2^-24 grid -- an fp32 operand is
// represented exactly (the operand scales keep it there) and the six products hi*hi, hi*mid, mid*hi, hi*lo, mid*mid, lo*hi with fp32
// accumulation drop only terms below 2^-33 of the product: the arithmetic class of v_mfma_f32_16x16x4_f32 at 2500 / 6 TFLOP/s.
// 14 vector instructions for four values (8 for two planes).
__device__ __forceinline__ void orl_split4x3(const f32x4& v, hx4& h, hx4& m, hx4& l) {
  h = __builtin_convertvector(v, hx4);
#if defined(ORL_SPLIT_BF16) || defined(ORL_SPLIT_NO_MIX)
  const f32x4 r1 = v - __builtin_convertvector(h, f32x4);
  m = __builtin_convertvector(r1, hx4);
  l = __builtin_convertvector(r1 - __builtin_convertvector(m, f32x4), hx4);
#else
  const u32x2_t hb = *(const u32x2_t*)&h;
  f32x4 r1, r2;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r1[0]) : "v"(hb[0]), "v"(v[0]));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1[1]) : "v"(hb[0]), "v"(v[1]));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r1[2]) : "v"(hb[1]), "v"(v[2]));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1[3]) : "v"(hb[1]), "v"(v[3]));
  m = __builtin_convertvector(r1, hx4);
  const u32x2_t mb = *(const u32x2_t*)&m;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r2[0]) : "v"(mb[0]), "v"(r1[0]));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r2[1]) : "v"(mb[0]), "v"(r1[1]));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r2[2]) : "v"(mb[1]), "v"(r1[2]));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r2[3]) : "v"(mb[1]), "v"(r1[3]));
  l = __builtin_convertvector(r2, hx4);
#endif
}
__device__ __forceinline__ void orl_split1x3(float x, hx_t& h, hx_t& m, hx_t& l) {
  h = (hx_t)x;
  const float r1 = x - (float)h;
  m = (hx_t)r1;
  l = (hx_t)(r1 - (float)m);
}
// bit j = (z[j] > 0): on the fp32 bit patterns read as signed integers, clamp(bits, 0, 1) is 1 exactly for positive non-zero values
// (v_med3_i32; -0, +0 and negative values give 0) -- 7 instructions instead of 4 compares + 4 selects + 3 ors
// ReLU in place + the 4 mask bits on the integer view of the floats: max(bits, 0) is ReLU (negative floats, -0 included, are negative
// integers), min(bits, 1u) of the result is the mask bit.  v_max_i32 / v_min_u32 are full-rate vector instructions (2.5 cycles of a
// gfx950 SIMD against 4.4 for v_max_f32 and for the compare + select pairs the float formulation compiles to; nothing overlaps with
// the MFMAs: DESIGN.md §5).
// NaN: torch.relu propagates a NaN; here a NaN whose sign bit is set is a negative integer and becomes +0 (mask bit 0), one with a clear
// sign bit stays a NaN.  A diverged run can therefore have part of its NaNs scrubbed inside the forward instead of surfacing in the losses:
// training health checks must look at the parameters (Adam writes the NaN gradient of the first non-finite step into them), not only at
// the metrics -- tests/test_gpu_training.py does, and bench.py asserts finite metrics AND finite sampled parameters.
__device__ __forceinline__ unsigned int orl_relu_mask4(f32x4& z) {
  unsigned int m = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = max(__float_as_int(z[j]), 0);
    z[j] = __int_as_float(r);
    unsigned int b;
    asm("v_min_u32 %0, 1, %1" : "=v"(b) : "v"(r));      // (written as min(r, 1u) the compiler turns it back into compare + select)
    m |= b << j;
  }
  return m;
}
__device__ __forceinline__ unsigned int orl_mask4(const f32x4& z) {
  unsigned int m = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int b = min(max(__float_as_int(z[j]), 0), 1);
    m |= (unsigned int)b << j;
  }
  return m;
}
#endif

// precision of the multiply: P_F32 = v_mfma_f32_16x16x4_f32 (exact fp32);  P_SPLIT = every operand (times its power-of-two scale) split
// into hi = half(x), lo = half(x - hi) while it is staged into LDS, product = lo*hi + hi*lo + hi*hi on the 16-bit MFMA with fp32
// accumulation (22 significand bits per operand, 3/16 of the fp32 MFMA cycles), the scales divided out of the accumulators
// P_SPLIT3 (precision 2): three planes per operand (hi + mid + lo = an fp32 value exactly), six products -- fp32-class arithmetic
enum { P_F32 = 0, P_SPLIT = 1, P_BF16X3 = P_SPLIT, P_SPLIT3 = 2 };

enum { PA_PLAIN = 0, PA_RANK1 = 1, PA_RANK1B = 2 };                    // prologue on A elements (RANK1B: ReLU mask from packed bits)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
enum { PB_PLAIN = 0, PB_ONES = 1 };                                    // prologue on B elements
enum { E_PLAIN = 0, E_BIAS = 1, E_BIAS_RELU = 2, E_MASK = 3, E_WGRAD = 4 };
enum { L_SCALAR = 0, L_VECK = 1, L_BLK4 = 2, L_VECKU = 3 };            // operand loaders

struct ZPtr {        // base + z0 * s0 + z1 * s1 (element strides)
  const float* p;
  long s0, s1;
  __host__ __device__ const float* at(int z0, int z1) const { return p ? p + z0 * s0 + z1 * s1 : nullptr; }
};

struct GemmP {
  ZPtr A, B;
  float* C;
  long c_s0, c_s1;   // z strides of C
  long c_sr;         // row stride of C (E_WGRAD: row stride of the weight-grad matrix)
  long c_sn;         // column stride of C (1 except for EnsembleLinear (in,out)-major weight gradients)
  long c_ks;         // split-K slab stride of C (elements)
  int M, N, K;
  long a_sr, a_sk;   // A[m*a_sr + k*a_sk]
  long b_sr, b_sk;   // B[n*b_sr + k*b_sk]
  int a_rlim, b_rlim;  // L_BLK4: number of rows that may be read starting at the operand base (multiple of 4)
  int nz1;           // z = z0*nz1 + z1
  int ksplit;        // blockIdx.y in [0, ksplit)
  // PA_RANK1: a(m', n') = A(m', n') > 0 ? rowv[m'] * colv[n'] : 0   (dz_L = dq (x) w_last (.) relu-mask)
  //   a_trans = 0: tile row = m', tile k = n' (dgrad)   a_trans = 1: tile row = n', tile k = m' (wgrad)
  ZPtr rowv, colv;
  int a_trans;
  // PB_ONES: logical B row index == ones_row -> 1.0 (bias-gradient column of wgrad)
  int ones_row;
  ZPtr bias;         // E_BIAS / E_BIAS_RELU: bias[n]
  ZPtr aux;          // E_MASK: C = aux[m*aux_sr + n] > 0 ? acc : 0
  long aux_sr;
  // E_WGRAD: C[m*c_sr + n*c_sn] = dW ; bias_out[m] = sum_k A[m][k] (bias gradient; same slab / z strides as C) or null
  float* bias_out;
  long bo_s0, bo_s1, bo_ks;
  // PA_RANK1 wgrad through the L_BLK4 loader: the kernel already streams h (raw activations) and dq (rowv), so it can
  // also produce the TAIL layer's gradients dw_tail[n'] = sum_m dq[m] h[m][n'] and db_tail = sum_m dq[m] (first column
  // tile only); null = not requested.  Same slab / z strides as C.
  float* tail_w_out; float* tail_b_out;
  long tw_s0, tw_s1, tb_s1;
  // E_MASK with w0_out != null (dgrad of hidden layer 1, tiles whose waves own 32 columns each): the masked tile
  // dz0 = C never has to reach HBM for the weight gradient of layer 0 -- the epilogue also produces this row tile's
  // contribution  dW0[n][c] = sum_r dz0[r][n] * X[r][c],  db0[n] = sum_r dz0[r][n]  (exact fp32 MFMA) into split-K slab
  // `row tile index`; Adam sums the slabs.  C == null then skips the store of dz0 altogether.
  // E_BIAS_RELU with tq_out != null (last hidden layer of a net whose tail is a single output, e.g. a critic): the epilogue
  // also reduces q_part[m] = sum_n relu(..)[m][n] * tq_w[n] over this tile's columns.  Column tile 0 writes
  // tq_out[m * tq_sm] (+ the tail bias), column tile t >= 1 writes tq_part[(t-1) * tq_ts + m]; the host adds the parts.
  ZPtr tq_w, tq_b; float* tq_out; float* tq_part; long tq_s0, tq_s1, tq_sm, tq_ps0, tq_ps1, tq_ts;
  // Packed ReLU masks: one 32-bit word per (row, 32 columns); bit b of word w of a row <-> column 32 w + b is set when the
  // activation is > 0.  `*_g` = words per row; z strides in 32-bit words.
  //   mb_out   (E_BIAS_RELU epilogue through LDS): also emit the mask of the produced activation
  //   aux_bits (E_MASK): read the mask from bits instead of the activation matrix `aux`
  //   a_bits   (PA_RANK1B, L_VECK, a_trans = 0): a(m, k) = bit(m, k) ? rowv[m] * colv[k] : 0 without reading A
  unsigned int* mb_out; long mb_s0, mb_s1; int mb_g;
  const unsigned int* aux_bits; long xb_s0, xb_s1; int xb_g;
  const unsigned int* a_bits; long ab_s0, ab_s1; int ab_g;
  ZPtr w0_x; long w0_xsr; int w0_in;
  float* w0_out; float* w0_bias; long w0_s0, w0_s1, w0_bs1, w0_ks, w0_sr;
  // P_SPLIT operand scales (powers of two; 0 = 1): an element x of A enters the multiply as x * a_scale * (a_dscale ? a_dscale[z0] : 1),
  // likewise B; the accumulators are divided by the product before the epilogue.  a_dscale / b_dscale: one float per run (z0) in device
  // memory -- the scale k_grad_scale chose for the gradient matrices of the backward pass this operand belongs to.
  float a_scale, b_scale;
  const float* a_dscale; const float* b_dscale;
  // z-major workgroup mapping (set by launch_inst for batched problems with few tiles each): gridDim.x = 8 * (tiles * ksplit),
  // gridDim.z = ceil(nz / 8); problem z = 8 * blockIdx.z + (blockIdx.x & 7), item = blockIdx.x >> 3 -- every tile of one problem has
  // the same linear-id residue mod 8, i.e. (round-robin placement) the same XCD and its L2, instead of one XCD per tile
  int zmajor, nz_total;
};

enum { W0_XP = 28 };     // LDS pitch of the X tile staged by the fused layer-0 weight gradient (floats)

template <int WM, int WN, int MA, int NB, int TK>
struct GemmCfg {
  static constexpr int kWM = WM, kWN = WN, kMA = MA, kNB = NB, kTK = TK;
  static constexpr int TM = WM * MA * 16, TN = WN * NB * 16, NT = WM * WN * 64, PITCH = TK + 4;
  static constexpr int PITCH_H = TK + 8;                       // bf16 planes: row stride (TK+8)*2 B, 16-B aligned
  static constexpr int LDS_FLOATS = 2 * (TM + TN) * PITCH;
  static constexpr size_t lds_bytes(int prec) {
    return prec == P_F32 ? sizeof(float) * LDS_FLOATS : (size_t)2 /*buf*/ * (prec == P_SPLIT3 ? 3 : 2) /*planes*/ * (TM + TN) * PITCH_H * 2;
  }
  // LDS the epilogue may use for the staged C tile: the operand buffers, or (when those already limit a CU to one
  // workgroup) the whole 160 KB
  static constexpr size_t epi_lds_limit(int prec) { return lds_bytes(prec) > 80 * 1024 ? (size_t)160 * 1024 : lds_bytes(prec); }
  static constexpr size_t epi_lds_bytes() { return sizeof(float) * (size_t)TM * (TN + 4); }
  static_assert(TK % 16 == 0, "TK multiple of 16");
};

// tile configurations
typedef GemmCfg<2, 4, 2, 4, 32> CfgBig;    // 64 x 256, 8 waves (two per SIMD): big forward / dgrad
typedef GemmCfg<2, 2, 2, 2, 32> CfgMid;    // 64 x 64  : wgrad tiles with split-K
typedef GemmCfg<1, 4, 1, 1, 64> CfgSmall;  // 16 x 64  : batch-sized (256-row) phases, many workgroups
typedef GemmCfg<4, 1, 1, 1, 32> CfgTall;   // 64 x 16  : narrow outputs (heads, action-gradient columns)
typedef GemmCfg<2, 2, 4, 4, 32> CfgSq;     // 128 x 128, 4 waves: many-row dgrad (several runs) and square wgrad tiles
typedef GemmCfg<2, 4, 4, 2, 32> CfgSq8;    // 128 x 128, 8 waves (4 per SIMD with two workgroups per CU): many-row forward
typedef GemmCfg<4, 2, 4, 4, 32> CfgWg;     // 256 x 128, 8 waves: weight gradients of 256-wide layers (each dz column block is read once)
enum { CFG_BIG = 0, CFG_MID = 1, CFG_SMALL = 2, CFG_TALL = 3, CFG_SQ = 4, CFG_SQ8 = 5, CFG_WG = 6, CFG_AUTO = -1 };

// heuristic tile choice when cfg == CFG_AUTO (measured with tools/gemm_sweep.py on MI355X)
static inline int pick_cfg(int M, int N, int K, int nz) {
  if (N <= 16) return CFG_TALL;
  if (M >= 2048 && N >= 128) return ((long)M * nz >= 40000) ? CFG_SQ : CFG_BIG;   // few rows: 8-wave 64x256 fills the CUs
  if (M <= 32) return CFG_SMALL;
  // batch-sized products of MANY batched nets (e.g. the 256-row phases of 64+ runs): enough 128 x 128 / 64 x 64 tiles exist to fill the
  // CUs, and they re-read each operand far less often than the 16 x 64 tiles (256 x 256 x 256, 128 nets: wgrad 108 -> 42 us with two
  // k-ranges, forward 68 -> 38 us, dgrad 96 -> 47 us; tools/small_gemm_sweep.py)
  // narrow products of many nets (the first layer's weight gradient, 17 .. 63 inputs): 64-row tiles read 256-byte pieces of the
  // transposed operand's rows where the 16-row tiles read 64-byte ones
  if (K < 1024 && M >= 64 && N > 16 && N < 64 && (long)nz * ((M + 63) / 64) >= 1024) return CFG_MID;
  if (K < 1024 && M >= 64 && N >= 64) {
    // a few input columns only (EDAC's t_0 = (gamma W_0[action rows]) (.) m_0): the launch is its epilogue -- 128 x 128 tiles move 64 KB each
    if (K <= 32 && M >= 128 && N >= 128 && (long)nz * ((M + 127) / 128) * ((N + 127) / 128) >= 1024) return CFG_SQ;
    if (K >= 256 && M >= 128 && N >= 128 && (long)nz * ((M + 127) / 128) * ((N + 127) / 128) >= 256) return CFG_SQ;
    if ((long)nz * ((M + 63) / 64) * ((N + 63) / 64) >= 1024) return CFG_MID;
  }
  // long reductions (wgrad over thousands of rows): square tiles + split-K; batch-sized products: many small workgroups
  if (K >= 1024) return (M >= 256 && N >= 128) ? CFG_WG : ((M >= 128 && N >= 128) ? CFG_SQ : CFG_MID);
  return CFG_SMALL;
}

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// Fused tail of a forward launch (GemmP::tq_*): supported on the big-tile configurations, whose epilogue goes through LDS
// when the pointers / pitches below are 16-byte aligned.  Returns the number of column tiles (partial sums), 0 if not.
static inline int tq_fused_parts(int cfg, const GemmP& p, const float* tail_w, long tw_s0, long tw_s1) {
  const int TN = cfg == CFG_SQ8 ? CfgSq8::TN : (cfg == CFG_BIG ? CfgBig::TN : 0);
  if (!TN) return 0;
  if ((p.N & 3) || !aligned16(p.C) || (p.c_sr & 3) || (p.c_s0 & 3) || (p.c_s1 & 3) || p.c_sn != 1) return 0;
  if (!aligned16(p.bias.p) || (p.bias.s0 & 3) || (p.bias.s1 & 3)) return 0;
  if (!aligned16(tail_w) || (tw_s0 & 3) || (tw_s1 & 3)) return 0;
  return (p.N + TN - 1) / TN;
}

// Will a forward launch on `cfg` emit packed mask bits (GemmP::mb_out)?  Same conditions as the fused tail.
static inline bool mb_supported(int cfg, const GemmP& p) {
  if (cfg == CFG_TALL || (p.N & 31)) return false;        // every other tile stages its epilogue through LDS with >= 8 lanes per row
  if (!aligned16(p.C) || (p.c_sr & 3) || (p.c_s0 & 3) || (p.c_s1 & 3) || p.c_sn != 1) return false;
  return aligned16(p.bias.p) && !(p.bias.s0 & 3) && !(p.bias.s1 & 3);
}

// Can the E_MASK dgrad launch (M x N x K, nz problems) also produce the layer-0 weight gradient in its epilogue (GemmP::w0_*)?
// Returns the number of split-K slabs it would write (= row tiles), 0 if not.  Mirrors the kernel's conditions.
static inline int w0_fused_slabs(const GemmP& p, int nz, int in0, long x_pitch, const void* x_ptr, long x_s0, long x_s1, int max_slab) {
  const int cfg = pick_cfg(p.M, p.N, p.K, nz);
  const int TM = cfg == CFG_SQ ? CfgSq::TM : (cfg == CFG_BIG ? CfgBig::TM : 0);
  if (!TM) return 0;                                                       // tiles whose waves own 32 columns each
  if (in0 + 1 > W0_XP || in0 >= x_pitch || x_pitch > W0_XP || (x_pitch & 3) || !aligned16(x_ptr) || (x_s0 & 3) || (x_s1 & 3)) return 0;
  if (p.N & 3) return 0;
  if (!p.aux_bits && ((p.aux_sr & 3) || !aligned16(p.aux.p) || (p.aux.s0 & 3) || (p.aux.s1 & 3))) return 0;   // LDS epilogue path
  if (p.C && (!aligned16(p.C) || (p.c_sr & 3) || (p.c_s0 & 3) || (p.c_s1 & 3) || p.c_sn != 1)) return 0;
  const int tiles_m = (p.M + TM - 1) / TM;
  return tiles_m <= max_slab ? tiles_m : 0;
}

// loader choice per operand from strides / alignment.  k_pad_ok: rows are zero-padded up to a multiple of 4 in k
static inline int pick_loader(const ZPtr& z, long sr, long sk, int K, bool k_pad_ok, int rlim) {
  if (sk == 1 && (!aligned16(z.p) || (z.s0 & 3) || (z.s1 & 3) || (sr & 3) || !((K & 3) == 0 || k_pad_ok))) return L_VECKU;
  if (!aligned16(z.p) || (z.s0 & 3) || (z.s1 & 3)) return L_SCALAR;
  if (sk == 1 && (sr & 3) == 0 && ((K & 3) == 0 || k_pad_ok)) return L_VECK;
  if (sr == 1 && (sk & 3) == 0 && rlim >= 4 && (rlim & 3) == 0) return L_BLK4;
  return L_SCALAR;
}

// true when launch_gemm will run a rank-1 wgrad through the (L_BLK4, L_BLK4) loaders, i.e. when the fused
// tail-gradient outputs of GemmP are honoured
static inline bool rank1_wgrad_is_fast(const GemmP& p, bool force_scalar) {
  if (force_scalar || p.a_trans != 1) return false;
  if (pick_loader(p.A, p.a_sr, p.a_sk, p.K, false, p.a_rlim) != L_BLK4) return false;
  if (pick_loader(p.B, p.b_sr, p.b_sk, p.K, false, p.b_rlim) != L_BLK4) return false;
  return aligned16(p.colv.p) && (p.colv.s0 & 3) == 0 && (p.colv.s1 & 3) == 0;
}

// dgrad whose A operand is the rank-1 virtual gradient rebuilt from packed mask bits: only the shapes the engine uses it for
// (k-contiguous A side, big tiles); anything else is refused and the caller keeps the float-mask path.
static inline bool rank1_bits_supported(int cfg, const GemmP& p, bool force_scalar) {
  if (force_scalar || (cfg != CFG_SQ && cfg != CFG_BIG) || p.a_trans != 0 || (p.K & 31)) return false;
  if (!aligned16(p.colv.p) || (p.colv.s0 & 3) || (p.colv.s1 & 3)) return false;
  const int lb = pick_loader(p.B, p.b_sr, p.b_sk, p.K, false, p.b_rlim);
  return lb == L_BLK4 || lb == L_VECK;
}

// ---- launch entry points: defined in gemm_kernel.h, instantiated in gemm_inst_*.hip (one translation unit per group so the
//      several hundred kernel instantiations compile in parallel) ----
template <int PA, int PB, int EPI>
hipError_t launch_gemm(int cfg, const GemmP& p, int nz, hipStream_t st, bool a_kpad = false, bool force_scalar = false, int prec = P_F32);
template <int EPI>
hipError_t launch_gemm_rank1_bits(int cfg, const GemmP& p, int nz, hipStream_t st, int prec);
hipError_t launch_tune(int cfg_in, int kind, const GemmP& p, int nz, hipStream_t st);

}  // namespace orl
