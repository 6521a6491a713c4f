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
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// precision of the multiply: P_F32 = v_mfma_f32_16x16x4_f32 (exact fp32);  P_BF16X3 = every operand split into
// hi = bf16(x), lo = bf16(x - hi) while it is staged into LDS, product = lo*hi + hi*lo + hi*hi on
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation (~16 mantissa bits per operand, 3/16 of the fp32 MFMA cycles)
enum { P_F32 = 0, P_BF16X3 = 1 };

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
};

enum { W0_XP = 28 };     // LDS pitch of the X tile staged by the fused layer-0 weight gradient (floats)

template <int WM, int WN, int MA, int NB, int TK>
struct GemmCfg {
  static constexpr int kWM = WM, kWN = WN, kMA = MA, kNB = NB, kTK = TK;
  static constexpr int TM = WM * MA * 16, TN = WN * NB * 16, NT = WM * WN * 64, PITCH = TK + 4;
  static constexpr int PITCH_H = TK + 8;                       // bf16 planes: row stride (TK+8)*2 B, 16-B aligned
  static constexpr int LDS_FLOATS = 2 * (TM + TN) * PITCH;
  static constexpr size_t lds_bytes(int prec) {
    return prec == P_F32 ? sizeof(float) * LDS_FLOATS : (size_t)2 /*buf*/ * 2 /*hi,lo*/ * (TM + TN) * PITCH_H * 2;
  }
  // LDS the epilogue may use for the staged C tile: the operand buffers, or (when those already limit a CU to one
  // workgroup) the whole 160 KB
  static constexpr size_t epi_lds_limit(int prec) { return lds_bytes(prec) > 80 * 1024 ? (size_t)160 * 1024 : lds_bytes(prec); }
  static constexpr size_t epi_lds_bytes() { return sizeof(float) * (size_t)TM * (TN + 4); }
  static_assert(TK % 16 == 0, "TK multiple of 16");
};

// ------------------------------------------------------------------------------------------------
// operand tile loader: ROWS x TK tile -> registers -> LDS [row][k]
// ------------------------------------------------------------------------------------------------
template <int ROWS, int TK, int NT, int PITCH, int LMODE, bool IS_A, int PRO>
struct TileLoader {
  static constexpr int ELEMS = ROWS * TK;
  static constexpr int SLOT_ELEMS = (LMODE == L_SCALAR) ? 1 : ((LMODE == L_VECK || LMODE == L_VECKU) ? 4 : 16);
  static constexpr int NSLOTS = ELEMS / SLOT_ELEMS;
  static constexpr int PER_THREAD = (NSLOTS + NT - 1) / NT;
  static constexpr bool EXACT = (NSLOTS % NT) == 0;
  float reg[PER_THREAD * SLOT_ELEMS];
  // loop-invariant per-slot state, computed once by init(): the K loop only adds a uniform k offset
  int goff[PER_THREAD];     // element offset of the slot's first element at k0 = 0 (row already clamped)
  int loff[PER_THREAD];     // LDS offset of the slot's first element
  int kin[PER_THREAD];      // k of the slot inside a chunk
  int grow[PER_THREAD];     // clamped global row (rank-1 prologue) 
  int ones[PER_THREAD];     // PB_ONES: bit rr set -> row (r + rr) is the virtual ones row
  long sk_;
  float tacc[(LMODE == L_BLK4 && IS_A && PRO == PA_RANK1) ? PER_THREAD * 4 : 1];   // fused tail-weight gradient partials
  float bacc;

  __device__ static inline void slot_rk(int e, bool k_contig, int& r, int& k) {
    if (LMODE == L_VECK || LMODE == L_VECKU) { constexpr int Q = TK / 4; k = 4 * (e % Q); r = e / Q; }
    else if (LMODE == L_BLK4) { constexpr int Q = ROWS / 4; r = 4 * (e % Q); k = 4 * (e / Q); }
    else if (k_contig) { k = e % TK; r = e / TK; }
    else { r = e % ROWS; k = e / ROWS; }
  }

  __device__ inline void init(const GemmP& p, int row0, int tid) {
    const long sr = IS_A ? p.a_sr : p.b_sr, sk = IS_A ? p.a_sk : p.b_sk;
    sk_ = sk;
    // rows that exist in memory: the wgrad "ones" row (bias-gradient column) is virtual and must never be read
    const int nrows = IS_A ? p.M : ((PRO == PB_ONES && p.ones_row < p.N) ? p.ones_row : p.N);
    const int rlim = IS_A ? p.a_rlim : p.b_rlim;
    const bool k_contig = (sk == 1);
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      const int e = tid + i * NT;
      int r = 0, k = 0;
      slot_rk(EXACT ? e : (e < NSLOTS ? e : 0), k_contig, r, k);
      int gr = row0 + r;
      if (LMODE == L_BLK4) gr = gr <= rlim - 4 ? gr : rlim - 4;
      else gr = gr < nrows ? gr : nrows - 1;
      grow[i] = gr;
      kin[i] = k;
      loff[i] = r * PITCH + k;
      goff[i] = (LMODE == L_BLK4) ? (int)((long)k * sk + gr) : (int)((long)gr * sr + (long)k * sk);
      int om = 0;
      if (PRO == PB_ONES && !IS_A) {
        if (LMODE == L_BLK4) { for (int rr = 0; rr < 4; ++rr) om |= (row0 + r + rr == p.ones_row) ? (1 << rr) : 0; }
        else om = (row0 + r == p.ones_row) ? 1 : 0;
      }
      ones[i] = om;
    }
    bacc = 0.f;
    for (int i = 0; i < (int)(sizeof(tacc) / sizeof(float)); ++i) tacc[i] = 0.f;
  }

  // TAIL = false: the whole chunk [k0, k0+TK) is inside K (no k checks)
  template <bool TAIL>
  __device__ inline void load(const GemmP& p, const float* __restrict__ g, const float* __restrict__ rowv,
                              const float* __restrict__ colv, int k0, int tid) {
    const long sk = sk_;
    const float* __restrict__ gk0 = g + (long)k0 * sk;      // uniform per chunk
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      if (!EXACT && tid + i * NT >= NSLOTS) break;
      const int gk = k0 + kin[i];
      float* o = &reg[i * SLOT_ELEMS];
      if (LMODE == L_SCALAR) {
        const bool kv = !TAIL || gk < p.K;
        float v = kv ? gk0[goff[i]] : g[goff[i] - (long)kin[i] * sk];          // clamped in-bounds address
        if (PRO == PA_RANK1 && IS_A) {
          const int kk = kv ? gk : 0;
          const int mm = p.a_trans ? kk : grow[i], nn = p.a_trans ? grow[i] : kk;
          v = v > 0.f ? rowv[mm] * colv[nn] : 0.f;
        }
        if (PRO == PB_ONES && !IS_A) v = ones[i] ? 1.f : v;
        o[0] = kv ? v : 0.f;
      } else if (LMODE == L_VECK) {
        const bool kv = !TAIL || gk < p.K;               // K % 4 == 0 or zero-padded rows (host guarantees)
        if (PRO == PA_RANK1B && IS_A) {                  // `g` = packed mask words of this z; A itself is never read
          const int kk = kv ? gk : 0;
          const unsigned int w = ((const unsigned int*)g)[(long)grow[i] * p.ab_g + (kk >> 5)] >> (kk & 31);
          const float rv = rowv[grow[i]];
          const f32x4 cv = *(const f32x4*)&colv[kk];
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = (kv && ((w >> j) & 1u)) ? rv * cv[j] : 0.f;
          continue;
        }
        const float* src = kv ? gk0 + goff[i] : g + goff[i] - kin[i];
        f32x4 v = *(const f32x4*)src;
        if (PRO == PA_RANK1 && IS_A) {
          const float rv = rowv[grow[i]];
          const f32x4 cv = *(const f32x4*)&colv[kv ? gk : 0];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? rv * cv[j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = kv ? v[j] : 0.f;
      } else if (LMODE == L_VECKU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool kv = !TAIL || (gk + j) < p.K;
          float v = kv ? gk0[goff[i] + j] : 0.f;
          if (PRO == PA_RANK1 && IS_A) {
            const int kk = kv ? gk + j : 0;
            const int mm = p.a_trans ? kk : grow[i], nn = p.a_trans ? grow[i] : kk;
            v = (kv && v > 0.f) ? rowv[mm] * colv[nn] : 0.f;
          }
          o[j] = v;
        }
      } else {  // L_BLK4
        // all global loads of the slot are issued before anything consumes them (a branch between a load and its use
        // would otherwise serialise the four row fetches)
        f32x4 vv[4];
        float rvv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool kv = !TAIL || (gk + j) < p.K;
          const float* src = kv ? gk0 + goff[i] + (long)j * sk : g + goff[i] - (long)kin[i] * sk;
          vv[j] = *(const f32x4*)src;
          if (PRO == PA_RANK1 && IS_A) rvv[j] = rowv[kv ? gk + j : 0];
        }
        f32x4 cv;
        if (PRO == PA_RANK1 && IS_A) cv = *(const f32x4*)&colv[grow[i]];
        if (PRO == PA_RANK1 && IS_A && p.tail_w_out) {       // uniform branch: raw vv = post-ReLU activation (>= 0)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool kv = !TAIL || (gk + j) < p.K;
            const float rv = kv ? rvv[j] : 0.f;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) tacc[i * 4 + rr] += rv * vv[j][rr];
            if (((tid + i * NT) % (ROWS / 4)) == 0) bacc += rv;
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool kv = !TAIL || (gk + j) < p.K;
          f32x4 v = vv[j];
          if (PRO == PA_RANK1 && IS_A) {
            const float rv = rvv[j];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) v[rr] = v[rr] > 0.f ? rv * cv[rr] : 0.f;
          }
          if (PRO == PB_ONES && !IS_A) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) v[rr] = ((ones[i] >> rr) & 1) ? 1.f : v[rr];
          }
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) o[rr * 4 + j] = kv ? v[rr] : 0.f;    // transpose: o[row][k]
        }
      }
    }
  }

  // split-bf16 staging: hi plane at lds_h, lo plane ROWS*PITCH elements further
  __device__ static inline void split4(const float* o, bf16x4& h, bf16x4& l) {
#ifdef ORL_LAB_FAKE_SPLIT      // tools/gemm_lab.hip only: numerically wrong (raw dword copies), bounds the cost of the conversion VALU work
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 a = {o[0], o[1]}, b = {o[2], o[3]};
    h = *(const bf16x4*)&a; l = *(const bf16x4*)&b;
#else
#pragma unroll
    for (int j = 0; j < 4; ++j) { const __bf16 hh = (__bf16)o[j]; h[j] = hh; l[j] = (__bf16)(o[j] - (float)hh); }
#endif
  }
  __device__ inline void store_split(__bf16* __restrict__ lds_h, int tid) const {
    __bf16* lds_l = lds_h + ROWS * PITCH;
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      if (!EXACT && tid + i * NT >= NSLOTS) break;
      const float* o = &reg[i * SLOT_ELEMS];
      if (LMODE == L_SCALAR) {
        const __bf16 hh = (__bf16)o[0];
        lds_h[loff[i]] = hh; lds_l[loff[i]] = (__bf16)(o[0] - (float)hh);
      } else if (LMODE == L_VECK || LMODE == L_VECKU) {
        bf16x4 h, l; split4(o, h, l);
        *(bf16x4*)(lds_h + loff[i]) = h; *(bf16x4*)(lds_l + loff[i]) = l;
      } else {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          bf16x4 h, l; split4(o + rr * 4, h, l);
          *(bf16x4*)(lds_h + loff[i] + rr * PITCH) = h; *(bf16x4*)(lds_l + loff[i] + rr * PITCH) = l;
        }
      }
    }
  }

  __device__ inline void store(float* __restrict__ lds, int tid) const {
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      if (!EXACT && tid + i * NT >= NSLOTS) break;
      const float* o = &reg[i * SLOT_ELEMS];
      float* d = lds + loff[i];
      if (LMODE == L_SCALAR) d[0] = o[0];
      else if (LMODE == L_VECK || LMODE == L_VECKU) *(f32x4*)d = (f32x4){o[0], o[1], o[2], o[3]};
      else {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) *(f32x4*)(d + rr * PITCH) = (f32x4){o[rr * 4], o[rr * 4 + 1], o[rr * 4 + 2], o[rr * 4 + 3]};
      }
    }
  }
};

#ifdef ORL_LAB_STAMPS   // tools/gemm_lab.hip only: s_memtime stamps of wave 0 of a few blocks
__device__ unsigned long long* g_lab_stamps;
#define LAB_STAMP(i) do { if (lab_on) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); lab_t[(i)] = t_; } } while (0)
#else
#define LAB_STAMP(i) do { } while (0)
#endif

template <class CFG, int LA, int LB, int PA, int PB, int EPI, int PREC>
__global__ __launch_bounds__(CFG::NT) void gemm16_kernel(const GemmP p) {
#ifdef ORL_LAB_STAMPS
  unsigned long long lab_t[16];
  const bool lab_on = (threadIdx.x == 0) && (blockIdx.z == 7) && (blockIdx.x < 128);
  for (int i = 0; i < 16; ++i) lab_t[i] = 0;
  LAB_STAMP(0);
#endif
  constexpr int TM = CFG::TM, TN = CFG::TN, TK = CFG::kTK, NT = CFG::NT;
  constexpr int PITCH = (PREC == P_F32) ? CFG::PITCH : CFG::PITCH_H;   // LDS row pitch in elements of the plane type
  constexpr int MA = CFG::kMA, NB = CFG::kNB;
  static_assert(PREC == P_F32 || TK % 32 == 0, "bf16 MFMA consumes 32 k per instruction");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // fp32: As [2][TM][PITCH] floats, Bs [2][TN][PITCH].   split-bf16: As [2][hi,lo][TM][PITCH] bf16, Bs likewise
  float* As = smem;
  float* Bs = smem + 2 * TM * PITCH;
  __bf16* Ah = (__bf16*)smem;
  __bf16* Bh = Ah + 2 * 2 * TM * PITCH;

  const int tid = threadIdx.x;
  const int z = blockIdx.z;
  const int z0 = z / p.nz1, z1 = z - z0 * p.nz1;
  // XCD-aware tile mapping (MI355X: 8 XCDs with private L2s; workgroups are dealt round-robin over them by linear
  // id, and gridDim.x is padded to a multiple of 8 so the XCD of a block is blockIdx.x % 8 for every z).  Logical work
  // items (tile, k-split) are numbered tile-fastest and each XCD takes a CONTIGUOUS range of them, so the tiles that
  // share an operand panel (all column tiles of one row tile; all tiles of one k-split in wgrad) hit the same L2
  // instead of each fetching the panel from HBM.  Only speed depends on the placement, never correctness.
  const int tiles_n = (p.N + TN - 1) / TN;
  const int tiles = ((p.M + TM - 1) / TM) * tiles_n;
  const int total = tiles * p.ksplit;
  const int per_xcd = gridDim.x >> 3;
  const int item = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (item >= total) return;
  const int ks = item / tiles, tile = item - ks * tiles;
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * TM, n0 = tn * TN;

  const float* __restrict__ Ag = (PA == PA_RANK1B) ? (const float*)(p.a_bits + z0 * p.ab_s0 + z1 * p.ab_s1) : p.A.at(z0, z1);
  const float* __restrict__ Bg = p.B.at(z0, z1);
  const float* __restrict__ rowv = p.rowv.at(z0, z1);
  const float* __restrict__ colv = p.colv.at(z0, z1);

  // K range of this split (chunk aligned); only the globally last chunk can be partial
  const int kchunks = (p.K + TK - 1) / TK;
  const int per = (kchunks + p.ksplit - 1) / p.ksplit;
  const int kc_begin = ks * per;
  int kc_end = kc_begin + per;
  if (kc_end > kchunks) kc_end = kchunks;
  const int kfull = p.K / TK;             // chunks [0, kfull) need no k checks

  TileLoader<TM, TK, NT, PITCH, LA, true, PA> la;
  TileLoader<TN, TK, NT, PITCH, LB, false, PB> lb;
  la.init(p, m0, tid);
  lb.init(p, n0, tid);

  auto load_chunk = [&](int kc) {
    if (kc < kfull) { la.template load<false>(p, Ag, rowv, colv, kc * TK, tid); lb.template load<false>(p, Bg, nullptr, nullptr, kc * TK, tid); }
    else { la.template load<true>(p, Ag, rowv, colv, kc * TK, tid); lb.template load<true>(p, Bg, nullptr, nullptr, kc * TK, tid); }
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave / CFG::kWN, wn = wave - wm * CFG::kWN;
  const int li = lane & 15, lq = lane >> 4;
  const int wrow0 = wm * MA * 16, wcol0 = wn * NB * 16;

  f32x4 acc[MA][NB];
#pragma unroll
  for (int a = 0; a < MA; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // E_WGRAD with bias_out: the bias gradient db[m] = sum_k A[m][k] is accumulated by the first column tile with one
  // extra MFMA per row block whose other operand is all ones (no extra B column, so N stays tile-aligned)
  const bool want_bias = (EPI == E_WGRAD) && (p.bias_out != nullptr) && (tn == 0) && (wn == 0);
  f32x4 accb[MA];
#pragma unroll
  for (int a = 0; a < MA; ++a) accb[a] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto store_chunk = [&](int buf) {
    if (PREC == P_F32) { la.store(As + buf * TM * PITCH, tid); lb.store(Bs + buf * TN * PITCH, tid); }
    else { la.store_split(Ah + buf * 2 * TM * PITCH, tid); lb.store_split(Bh + buf * 2 * TN * PITCH, tid); }
  };
  LAB_STAMP(1);
  if (kc_begin < kc_end) {
    load_chunk(kc_begin);
    store_chunk(0);
  }
  __syncthreads();
  LAB_STAMP(2);
  for (int kc = kc_begin; kc < kc_end; ++kc) {
    const int buf = (kc - kc_begin) & 1;
    const bool more = kc + 1 < kc_end;
    if (more) load_chunk(kc + 1);
    if (kc - kc_begin < 2) LAB_STAMP(3 + 3 * (kc - kc_begin));
    if (PREC == P_F32) {
      const float* as = As + buf * TM * PITCH;
      const float* bs = Bs + buf * TN * PITCH;
#pragma unroll
      for (int kk = 0; kk < TK; kk += 16) {
        f32x4 fa[MA], fb[NB];
#pragma unroll
        for (int a = 0; a < MA; ++a) fa[a] = *(const f32x4*)&as[(wrow0 + a * 16 + li) * PITCH + kk + 4 * lq];
#pragma unroll
        for (int b = 0; b < NB; ++b) fb[b] = *(const f32x4*)&bs[(wcol0 + b * 16 + li) * PITCH + kk + 4 * lq];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int a = 0; a < MA; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[b][s], fa[a][s], acc[a][b], 0, 0, 0);   // transposed tile: D[n][m]
        if (EPI == E_WGRAD && want_bias) {
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int a = 0; a < MA; ++a) accb[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, fa[a][s], accb[a], 0, 0, 0);
        }
      }
    } else {
      // lane (li, lq) supplies 8 consecutive k (8*lq ..) of row li for both operands of v_mfma_f32_16x16x32_bf16
      const __bf16* ah = Ah + buf * 2 * TM * PITCH;
      const __bf16* al = ah + TM * PITCH;
      const __bf16* bh = Bh + buf * 2 * TN * PITCH;
      const __bf16* bl = bh + TN * PITCH;
#pragma unroll
      for (int kk = 0; kk < TK; kk += 32) {
        bf16x8 fah[MA], fal[MA], fbh[NB], fbl[NB];
#pragma unroll
        for (int a = 0; a < MA; ++a) {
          const int o = (wrow0 + a * 16 + li) * PITCH + kk + 8 * lq;
          fah[a] = *(const bf16x8*)&ah[o]; fal[a] = *(const bf16x8*)&al[o];
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const int o = (wcol0 + b * 16 + li) * PITCH + kk + 8 * lq;
          fbh[b] = *(const bf16x8*)&bh[o]; fbl[b] = *(const bf16x8*)&bl[o];
        }
#pragma unroll
        for (int a = 0; a < MA; ++a)
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fbl[b], fah[a], acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fbh[b], fal[a], acc[a][b], 0, 0, 0);
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fbh[b], fah[a], acc[a][b], 0, 0, 0);
          }
        if (EPI == E_WGRAD && want_bias) {
          bf16x8 one;
#pragma unroll
          for (int j = 0; j < 8; ++j) one[j] = (__bf16)1.0f;
#pragma unroll
          for (int a = 0; a < MA; ++a) {
            accb[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(one, fal[a], accb[a], 0, 0, 0);
            accb[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(one, fah[a], accb[a], 0, 0, 0);
          }
        }
      }
    }
    if (kc - kc_begin < 2) LAB_STAMP(3 + 3 * (kc - kc_begin) + 1);
    if (more) store_chunk(buf ^ 1);
    if (kc - kc_begin < 2) LAB_STAMP(3 + 3 * (kc - kc_begin) + 2);
    __syncthreads();
    if (kc - kc_begin < 2) LAB_STAMP(3 + 3 * (kc - kc_begin) + 3);
  }
  LAB_STAMP(12);

  // ---- epilogue.  The MFMA was issued with the operands swapped (W-tile as A, X-tile as B), so the 16x16 block
  // lives transposed in the accumulators: lane (li, lq) holds C[m = li][n = 4*lq + reg], i.e. four CONSECUTIVE
  // output columns of one row -> one 16-byte store (and 16-byte bias / mask loads) per block instead of four
  // dword stores. ----
  float* Cg = p.C + z0 * p.c_s0 + z1 * p.c_s1 + (long)ks * p.c_ks;
  const float* __restrict__ bias = p.bias.at(z0, z1);
  const float* __restrict__ aux = p.aux.at(z0, z1);
  float* bo = (EPI == E_WGRAD && p.bias_out) ? p.bias_out + z0 * p.bo_s0 + z1 * p.bo_s1 + (long)ks * p.bo_ks : nullptr;
  const bool vec_ok = (p.c_sn == 1) && ((p.c_sr & 3) == 0) && ((((uintptr_t)Cg) & 15) == 0) &&
                      (EPI != E_MASK || p.aux_bits != nullptr || (((p.aux_sr & 3) == 0) && ((((uintptr_t)aux) & 15) == 0))) &&
                      ((EPI != E_BIAS && EPI != E_BIAS_RELU) || ((((uintptr_t)bias) & 15) == 0));
  if (EPI == E_WGRAD && PA == PA_RANK1 && LA == L_BLK4) {
    if (p.tail_w_out && tn == 0) {
      // deterministic reduction of the per-thread partials: sred[k-slot][row], summed in k-slot order
      constexpr int KQ = TK / 4, Q = TM / 4;
      float* sred = smem;                                   // the staging tiles are dead after the last barrier
      float* sb = smem + KQ * TM;
#pragma unroll
      for (int i = 0; i < decltype(la)::PER_THREAD; ++i) {
        const int e = tid + i * NT;
        if (e < decltype(la)::NSLOTS) {
          const int r = 4 * (e % Q), kq = e / Q;
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) sred[kq * TM + r + rr] = la.tacc[i * 4 + rr];
        }
      }
      sb[tid] = la.bacc;                                     // zero for threads that own no first-row-block slot
      __syncthreads();
      float* tw = p.tail_w_out + z0 * p.tw_s0 + z1 * p.tw_s1 + (long)ks * p.c_ks;
      for (int r = tid; r < TM; r += NT) {
        const int m = m0 + r;
        if (m < p.M) {
          float sacc = 0.f;
#pragma unroll
          for (int kq = 0; kq < KQ; ++kq) sacc += sred[kq * TM + r];
          tw[m] = sacc;
        }
      }
      if (tm == 0 && tid == 0 && p.tail_b_out) {
        float sacc = 0.f;
        for (int t = 0; t < NT; ++t) sacc += sb[t];
        p.tail_b_out[z0 * p.tw_s0 + z1 * p.tb_s1 + (long)ks * p.c_ks] = sacc;
      }
    }
  }
  if (EPI == E_WGRAD && want_bias && lq == 0) {
#pragma unroll
    for (int a = 0; a < MA; ++a) {
      const int m = m0 + wrow0 + a * 16 + li;
      if (m < p.M) bo[m] = accb[a][0];      // every n-row of the ones-block holds the same sum; lane (li, 0) reg 0 = D[0][m]
    }
  }
  // Coalesced path: the accumulator blocks hold 16 rows x 64 B each, so direct stores would write (and the mask would read)
  // 64-byte pieces of 16 different rows per instruction.  Stage the TM x TN tile through LDS (the operand buffers are dead)
  // and let every wave move whole row segments: 64 lanes x 16 B = two 512-B runs (TN = 128) per instruction.
  constexpr int CP = TN + 4;                                   // LDS pitch of the staged C tile (floats)
  constexpr bool LDS_EPI_FITS = (size_t)TM * CP * sizeof(float) <= CFG::epi_lds_limit(PREC) && (NT % (TN / 4)) == 0;
  constexpr bool W0_CAP = (EPI == E_MASK) && (TN / (NT / 64) == 32) && (TM % 16 == 0);
  const bool w0 = W0_CAP && (p.w0_out != nullptr);             // uniform; the host only asks when the LDS path below is taken
  if (LDS_EPI_FITS && (vec_ok || (w0 && p.C == nullptr)) && (p.N & 3) == 0) {             // uniform per workgroup
    float* cs = smem;
    constexpr int C4 = TN / 4, RPP = NT / C4, NPASS = (TM + RPP - 1) / RPP;   // float4 columns per row, rows per pass
    constexpr bool MB_CAP = (C4 % 8) == 0;                                     // eight lanes of a row own one 32-column mask word
    const int c4 = tid % C4, r0 = tid / C4;
    const int n = n0 + 4 * c4;
    const bool n_ok = n < p.N;
    // the mask tile is fetched first (clamped addresses, no branches) so its latency hides behind the LDS staging
    f32x4 hv[EPI == E_MASK ? NPASS : 1];
    const bool xbits = (EPI == E_MASK) && (p.aux_bits != nullptr);           // uniform: the mask comes as packed bits
    if (EPI == E_MASK) {
      const unsigned int* xb = p.aux_bits + z0 * p.xb_s0 + z1 * p.xb_s1;
#pragma unroll
      for (int i = 0; i < NPASS; ++i) {
        int m = m0 + r0 + i * RPP;
        m = m < p.M ? m : p.M - 1;
        if (xbits) hv[i][0] = __uint_as_float(xb[(long)m * p.xb_g + ((n_ok ? n : 0) >> 5)]);
        else hv[i] = *(const f32x4*)&aux[(long)m * p.aux_sr + (n_ok ? n : 0)];
      }
    }
    // fused layer-0 weight gradient: this thread's pieces of the X tile [TM][W0_XP], fetched now for the same reason
    constexpr int XQ = W0_XP / 4, XPT = W0_CAP ? (TM * XQ + NT - 1) / NT : 1;
    f32x4 xv[XPT];
    if (W0_CAP && w0) {
      const float* __restrict__ xg = p.w0_x.at(z0, z1);
      const int xq = (int)(p.w0_xsr >> 2);                     // float4 per X row in memory (<= XQ)
#pragma unroll
      for (int i = 0; i < XPT; ++i) {
        const int e = tid + i * NT, r = e / XQ, q = e - r * XQ;
        int m = m0 + r; m = m < p.M ? m : p.M - 1;
        xv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (q < xq && e < TM * XQ) xv[i] = *(const f32x4*)&xg[(long)m * p.w0_xsr + 4 * q];
#pragma unroll
        for (int j = 0; j < 4; ++j) if (4 * q + j == p.w0_in) xv[i][j] = 1.0f;     // ones column -> bias gradient
      }
    }
    f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (EPI == E_BIAS || EPI == E_BIAS_RELU) bv = *(const f32x4*)&bias[n_ok ? n : 0];
    const bool tq = (EPI == E_BIAS_RELU) && (p.tq_out != nullptr);       // uniform
    f32x4 tw = (f32x4){0.f, 0.f, 0.f, 0.f};
    float tq_bias = 0.f;
    if (EPI == E_BIAS_RELU && tq) {
      if (n_ok) tw = *(const f32x4*)&(p.tq_w.at(z0, z1)[n]);
      if (tn == 0) tq_bias = p.tq_b.at(z0, z1)[0];
    }
    if (EPI == E_WGRAD && PA == PA_RANK1 && LA == L_BLK4) __syncthreads();   // the tail-gradient reduction above used smem
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
      for (int b = 0; b < NB; ++b)
        *(f32x4*)&cs[(wrow0 + a * 16 + li) * CP + wcol0 + b * 16 + 4 * lq] = acc[a][b];
    if (W0_CAP && w0) {
      float* xs = cs + TM * CP;                                // X tile behind the C tile
#pragma unroll
      for (int i = 0; i < XPT; ++i) {
        const int e = tid + i * NT;
        if (e < TM * XQ) *(f32x4*)&xs[(e / XQ) * W0_XP + 4 * (e % XQ)] = xv[i];
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int r = r0 + i * RPP;
      const int m = m0 + r;
      if (r < TM) {
        f32x4 v = *(const f32x4*)&cs[r * CP + 4 * c4];
        if (EPI == E_BIAS || EPI == E_BIAS_RELU) v += bv;
        if (EPI == E_BIAS_RELU) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
        }
        if (EPI == E_MASK) {
          if (xbits) {
            const unsigned int w = __float_as_uint(hv[i][0]) >> (n & 31);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ((w >> j) & 1u) ? v[j] : 0.f;
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = hv[i][j] > 0.f ? v[j] : 0.f;
          }
        }
        if (EPI == E_BIAS_RELU && MB_CAP && p.mb_out) {        // uniform: emit the packed mask of this tile's rows
          // 4 bits per lane, 8 consecutive lanes of a row make one 32-bit word (OR-reduction over lane bits 0..2)
          unsigned int w = 0;
          if (n_ok) w = ((v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u)) << (4 * (c4 & 7));
          w |= __shfl_xor(w, 1); w |= __shfl_xor(w, 2); w |= __shfl_xor(w, 4);
          if ((c4 & 7) == 0 && n_ok && m < p.M) p.mb_out[z0 * p.mb_s0 + z1 * p.mb_s1 + (long)m * p.mb_g + (n >> 5)] = w;
        }
        if (n_ok && m < p.M && (!W0_CAP || p.C != nullptr)) *(f32x4*)&Cg[(long)m * p.c_sr + n] = v;
        if (EPI == E_BIAS_RELU && tq) {                        // row m of the tile is spread over C4 consecutive lanes
          float pd = (v[0] * tw[0] + v[1] * tw[1]) + (v[2] * tw[2] + v[3] * tw[3]);
#pragma unroll
          for (int o = C4 / 2; o > 0; o >>= 1) pd += __shfl_xor(pd, o);
          if (c4 == 0 && m < p.M) {
            if (tn == 0) p.tq_out[z0 * p.tq_s0 + z1 * p.tq_s1 + (long)m * p.tq_sm] = pd + tq_bias;
            else p.tq_part[z0 * p.tq_ps0 + z1 * p.tq_ps1 + (long)(tn - 1) * p.tq_ts + m] = pd;
          }
        }
        if (W0_CAP && w0) {                                    // masked values (zero beyond M / N) back into the staged tile
          if (!(n_ok && m < p.M)) v = (f32x4){0.f, 0.f, 0.f, 0.f};
          *(f32x4*)&cs[r * CP + 4 * c4] = v;
        }
      }
    }
    if (W0_CAP && w0) {
      const float* xs = cs + TM * CP;
      __syncthreads();
      // wave w owns columns [32w, 32w+32) of the tile: D[n][c] = sum_r cs[r][n] * xs[r][c], 2 x 2 blocks of 16 x 16
      f32x4 d[2][2];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) d[nb][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int ncol0 = 32 * wave;
      const int c1 = (16 + li) < W0_XP ? 16 + li : W0_XP - 1;   // columns >= W0_XP are never stored
#pragma unroll 2
      for (int k0 = 0; k0 < TM; k0 += 16) {
        float av[2][4], bw[2][4];
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
          const int r = k0 + 4 * lq + sidx;
          av[0][sidx] = cs[r * CP + ncol0 + li];
          av[1][sidx] = cs[r * CP + ncol0 + 16 + li];
          bw[0][sidx] = xs[r * W0_XP + li];
          bw[1][sidx] = xs[r * W0_XP + c1];
        }
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
              d[nb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[nb][sidx], bw[cb][sidx], d[nb][cb], 0, 0, 0);
      }
      // lane (li, lq) holds D[n = 4*lq + reg][c = li] of each block
      float* wo = p.w0_out + z0 * p.w0_s0 + z1 * p.w0_s1 + (long)tm * p.w0_ks;
      float* bo0 = p.w0_bias + z0 * p.w0_s0 + z1 * p.w0_bs1 + (long)tm * p.w0_ks;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int nn = n0 + ncol0 + 16 * nb + 4 * lq + rg, c = 16 * cb + li;
            if (nn < p.N) {
              if (c < p.w0_in) wo[(long)nn * p.w0_sr + c] = d[nb][cb][rg];
              else if (c == p.w0_in) bo0[nn] = d[nb][cb][rg];
            }
          }
    }
  } else {
#pragma unroll
    for (int a = 0; a < MA; ++a) {
      const int m = m0 + wrow0 + a * 16 + li;
  #pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int nb = n0 + wcol0 + b * 16 + 4 * lq;
        if (m >= p.M || nb >= p.N) continue;
        f32x4 v = acc[a][b];
        const int nlim = p.N;
        if (vec_ok && nb + 4 <= nlim) {
          if (EPI == E_BIAS || EPI == E_BIAS_RELU) { const f32x4 bv = *(const f32x4*)&bias[nb]; v += bv; }
          if (EPI == E_BIAS_RELU) {
  #pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
          }
          if (EPI == E_MASK) {
            const f32x4 hv = *(const f32x4*)&aux[(long)m * p.aux_sr + nb];
  #pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = hv[r] > 0.f ? v[r] : 0.f;
          }
          *(f32x4*)&Cg[(long)m * p.c_sr + nb] = v;
        } else {
  #pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int n = nb + r;
            if (n >= p.N) continue;
            float x = v[r];
            if (EPI == E_BIAS) x += bias[n];
            if (EPI == E_BIAS_RELU) { x += bias[n]; x = x > 0.f ? x : 0.f; }
            if (EPI == E_MASK) x = aux[(long)m * p.aux_sr + n] > 0.f ? x : 0.f;
            Cg[(long)m * p.c_sr + (long)n * p.c_sn] = x;
          }
        }
      }
    }
}
#ifdef ORL_LAB_STAMPS
  LAB_STAMP(13);
  if (lab_on) for (int i = 0; i < 16; ++i) g_lab_stamps[blockIdx.x * 16 + i] = lab_t[i];
#endif
}

// tile configurations
typedef GemmCfg<2, 4, 2, 4, 32> CfgBig;    // 64 x 256, 8 waves (two per SIMD): big forward / dgrad
typedef GemmCfg<2, 2, 2, 2, 32> CfgMid;    // 64 x 64  : wgrad tiles with split-K
typedef GemmCfg<1, 4, 1, 1, 64> CfgSmall;  // 16 x 64  : batch-sized (256-row) phases, many workgroups
typedef GemmCfg<4, 1, 1, 1, 32> CfgTall;   // 64 x 16  : narrow outputs (heads, action-gradient columns)
typedef GemmCfg<2, 2, 4, 4, 32> CfgSq;     // 128 x 128, 4 waves: many-row dgrad (several runs) and square wgrad tiles
typedef GemmCfg<2, 4, 4, 2, 32> CfgSq8;    // 128 x 128, 8 waves (4 per SIMD with two workgroups per CU): many-row forward
typedef GemmCfg<4, 2, 4, 4, 32> CfgWg;     // 256 x 128, 8 waves: weight gradients of 256-wide layers (each dz column block is read once)
enum { CFG_BIG = 0, CFG_MID = 1, CFG_SMALL = 2, CFG_TALL = 3, CFG_SQ = 4, CFG_SQ8 = 5, CFG_WG = 6, CFG_AUTO = -1 };

template <class CFG, int LA, int LB, int PA, int PB, int EPI, int PREC = P_F32>
static inline hipError_t launch_inst(const GemmP& p, int nz, hipStream_t st) {
  const int tiles = ((p.M + CFG::TM - 1) / CFG::TM) * ((p.N + CFG::TN - 1) / CFG::TN);
  dim3 grid((tiles * p.ksplit + 7) & ~7, 1, nz), block(CFG::NT);     // padded to the 8 XCDs (see the kernel's tile mapping)
  size_t lds = CFG::lds_bytes(PREC);
  if (CFG::epi_lds_bytes() <= CFG::epi_lds_limit(PREC)) lds = std::max(lds, CFG::epi_lds_bytes());   // staged C tile (kernel: LDS_EPI_FITS)
  if (EPI == E_MASK && p.w0_out) lds = std::max(lds, sizeof(float) * ((size_t)CFG::TM * (CFG::TN + 4) + (size_t)CFG::TM * W0_XP));
  auto kern = gemm16_kernel<CFG, LA, LB, PA, PB, EPI, PREC>;
  if (lds > 64 * 1024) {
    static std::atomic<size_t> raised_to{0};   // per instantiation; the request can grow (fused layer-0 gradient), engines may launch from several threads
    if (lds > raised_to.load(std::memory_order_acquire)) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      raised_to.store(lds, std::memory_order_release);
    }
  }
  hipLaunchKernelGGL(kern, grid, block, lds, st, p);
  return hipGetLastError();
}

// heuristic tile choice when cfg == CFG_AUTO (measured with tools/gemm_sweep.py on MI355X)
static inline int pick_cfg(int M, int N, int K, int nz) {
  if (N <= 16) return CFG_TALL;
  if (M >= 2048 && N >= 128) return ((long)M * nz >= 40000) ? CFG_SQ : CFG_BIG;   // few rows: 8-wave 64x256 fills the CUs
  if (M <= 32) return CFG_SMALL;
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

template <class CFG, int PA, int PB, int EPI, int PREC>
static inline hipError_t launch_cfg(const GemmP& p, int la, int lb, int nz, hipStream_t st) {
  // supported loader pairs; anything else falls back to the scalar loaders
  if (la == L_VECK && lb == L_VECK) return launch_inst<CFG, L_VECK, L_VECK, PA, PB, EPI, PREC>(p, nz, st);
  if (la == L_VECK && lb == L_BLK4) return launch_inst<CFG, L_VECK, L_BLK4, PA, PB, EPI, PREC>(p, nz, st);
  if (la == L_BLK4 && lb == L_BLK4) return launch_inst<CFG, L_BLK4, L_BLK4, PA, PB, EPI, PREC>(p, nz, st);
  if (la == L_VECK && lb == L_VECKU) return launch_inst<CFG, L_VECK, L_VECKU, PA, PB, EPI, PREC>(p, nz, st);
  if (la == L_VECK) return launch_inst<CFG, L_VECK, L_SCALAR, PA, PB, EPI, PREC>(p, nz, st);
  return launch_inst<CFG, L_SCALAR, L_SCALAR, PA, PB, EPI, PREC>(p, nz, st);
}
template <class CFG, int PA, int PB, int EPI>
static inline hipError_t launch_cfg_prec(const GemmP& p, int la, int lb, int nz, hipStream_t st, int prec) {
  if (prec == P_BF16X3) return launch_cfg<CFG, PA, PB, EPI, P_BF16X3>(p, la, lb, nz, st);
  return launch_cfg<CFG, PA, PB, EPI, P_F32>(p, la, lb, nz, st);
}

template <int PA, int PB, int EPI>
static inline hipError_t launch_gemm(int cfg, const GemmP& p, int nz, hipStream_t st, bool a_kpad = false, bool force_scalar = false,
                                     int prec = P_F32) {
  if (cfg == CFG_AUTO) cfg = pick_cfg(p.M, p.N, p.K, nz);
  int la = L_SCALAR, lb = L_SCALAR;
  if (!force_scalar) {
    la = pick_loader(p.A, p.a_sr, p.a_sk, p.K, a_kpad, p.a_rlim);
    lb = pick_loader(p.B, p.b_sr, p.b_sk, p.K, false, p.b_rlim);
    if (PA == PA_RANK1) {   // the rank-1 prologue reads colv / rowv with the operand's vector shape
      if (la == L_VECK && (p.a_trans != 0 || !aligned16(p.colv.p) || (p.colv.s0 & 3) || (p.colv.s1 & 3))) la = L_SCALAR;
      if (la == L_BLK4 && (p.a_trans != 1 || !aligned16(p.colv.p) || (p.colv.s0 & 3) || (p.colv.s1 & 3))) la = L_SCALAR;
    }
  }
  switch (cfg) {
    case CFG_BIG: return launch_cfg_prec<CfgBig, PA, PB, EPI>(p, la, lb, nz, st, prec);
    case CFG_MID: return launch_cfg_prec<CfgMid, PA, PB, EPI>(p, la, lb, nz, st, prec);
    case CFG_SMALL: return launch_cfg_prec<CfgSmall, PA, PB, EPI>(p, la, lb, nz, st, prec);
    case CFG_SQ: return launch_cfg_prec<CfgSq, PA, PB, EPI>(p, la, lb, nz, st, prec);
    case CFG_SQ8: return launch_cfg_prec<CfgSq8, PA, PB, EPI>(p, la, lb, nz, st, prec);
    case CFG_WG: return launch_cfg_prec<CfgWg, PA, PB, EPI>(p, la, lb, nz, st, prec);
    default: return launch_cfg_prec<CfgTall, PA, PB, EPI>(p, la, lb, nz, st, prec);
  }
}

// dgrad whose A operand is the rank-1 virtual gradient rebuilt from packed mask bits: only the shapes the engine uses it for
// (k-contiguous A side, big tiles); anything else is refused and the caller keeps the float-mask path.
static inline bool rank1_bits_supported(int cfg, const GemmP& p, bool force_scalar) {
  if (force_scalar || (cfg != CFG_SQ && cfg != CFG_BIG) || p.a_trans != 0 || (p.K & 31)) return false;
  if (!aligned16(p.colv.p) || (p.colv.s0 & 3) || (p.colv.s1 & 3)) return false;
  const int lb = pick_loader(p.B, p.b_sr, p.b_sk, p.K, false, p.b_rlim);
  return lb == L_BLK4 || lb == L_VECK;
}
template <int EPI>
static inline hipError_t launch_gemm_rank1_bits(int cfg, const GemmP& p, int nz, hipStream_t st, int prec) {
  if (!rank1_bits_supported(cfg, p, false)) return hipErrorInvalidValue;
  const int lb = pick_loader(p.B, p.b_sr, p.b_sk, p.K, false, p.b_rlim);
#define ORL_RB(CFG, LB, PREC) launch_inst<CFG, L_VECK, LB, PA_RANK1B, PB_PLAIN, EPI, PREC>(p, nz, st)
  if (cfg == CFG_SQ) {
    if (prec == P_BF16X3) return lb == L_BLK4 ? ORL_RB(CfgSq, L_BLK4, P_BF16X3) : ORL_RB(CfgSq, L_VECK, P_BF16X3);
    return lb == L_BLK4 ? ORL_RB(CfgSq, L_BLK4, P_F32) : ORL_RB(CfgSq, L_VECK, P_F32);
  }
  if (prec == P_BF16X3) return lb == L_BLK4 ? ORL_RB(CfgBig, L_BLK4, P_BF16X3) : ORL_RB(CfgBig, L_VECK, P_BF16X3);
  return lb == L_BLK4 ? ORL_RB(CfgBig, L_BLK4, P_F32) : ORL_RB(CfgBig, L_VECK, P_F32);
#undef ORL_RB
}

// ---- tuning tap (orl_debug_gemm_time): the three hot kernel kinds on the main + two extra tile shapes ----
typedef GemmCfg<1, 4, 4, 4, 32> CfgT7;    // 64 x 256, 4 waves
typedef GemmCfg<2, 2, 2, 4, 32> CfgT11;   // 64 x 128
template <class CFG, int PREC>
static inline hipError_t launch_tune_kind_p(int kind, const GemmP& p, int nz, hipStream_t st) {
  if (kind == 0) return launch_inst<CFG, L_VECK, L_VECK, PA_PLAIN, PB_PLAIN, E_BIAS_RELU, PREC>(p, nz, st);
  if (kind == 1) return launch_inst<CFG, L_VECK, L_BLK4, PA_RANK1, PB_PLAIN, E_MASK, PREC>(p, nz, st);
  return launch_inst<CFG, L_BLK4, L_BLK4, PA_RANK1, PB_PLAIN, E_WGRAD, PREC>(p, nz, st);
}
template <class CFG>
static inline hipError_t launch_tune_kind(int kind, const GemmP& p, int nz, hipStream_t st, int prec) {
  if (prec == P_BF16X3) return launch_tune_kind_p<CFG, P_BF16X3>(kind, p, nz, st);
  return launch_tune_kind_p<CFG, P_F32>(kind, p, nz, st);
}
static inline hipError_t launch_tune(int cfg_in, int kind, const GemmP& p, int nz, hipStream_t st) {
  const int prec = (cfg_in & 32) ? P_BF16X3 : P_F32;      // bit 5 selects the split-bf16 multiply
  switch (cfg_in & 31) {
    case 0: return launch_tune_kind<CfgBig>(kind, p, nz, st, prec);
    case 1: return launch_tune_kind<CfgMid>(kind, p, nz, st, prec);
    case 2: return launch_tune_kind<CfgSmall>(kind, p, nz, st, prec);
    case 3: return launch_tune_kind<CfgTall>(kind, p, nz, st, prec);
    case 4: return launch_tune_kind<CfgSq>(kind, p, nz, st, prec);
    case 7: return launch_tune_kind<CfgT7>(kind, p, nz, st, prec);
    default: return launch_tune_kind<CfgT11>(kind, p, nz, st, prec);
  }
}

}  // namespace orl
