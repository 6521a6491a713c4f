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

namespace orl {

typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { PA_PLAIN = 0, PA_RANK1 = 1 };                                   // prologue on A elements
enum { PB_PLAIN = 0, PB_ONES = 1 };                                    // prologue on B elements
enum { E_PLAIN = 0, E_BIAS = 1, E_BIAS_RELU = 2, E_MASK = 3, E_WGRAD = 4 };
enum { L_SCALAR = 0, L_VECK = 1, L_BLK4 = 2 };                         // operand loaders

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
  // E_WGRAD: n < ones_row -> C[m*c_sr + n*c_sn];  n == ones_row -> bias_out[m]   (same slab / z strides as C)
  float* bias_out;
  long bo_s0, bo_s1, bo_ks;
};

template <int WM, int WN, int MA, int NB, int TK>
struct GemmCfg {
  static constexpr int kWM = WM, kWN = WN, kMA = MA, kNB = NB, kTK = TK;
  static constexpr int TM = WM * MA * 16, TN = WN * NB * 16, NT = WM * WN * 64, PITCH = TK + 4;
  static constexpr int LDS_FLOATS = 2 * (TM + TN) * PITCH;
  static_assert(TK % 16 == 0, "TK multiple of 16");
};

// ------------------------------------------------------------------------------------------------
// operand tile loader: ROWS x TK tile -> registers -> LDS [row][k]
// ------------------------------------------------------------------------------------------------
template <int ROWS, int TK, int NT, int PITCH, int LMODE, bool IS_A, int PRO>
struct TileLoader {
  static constexpr int ELEMS = ROWS * TK;
  static constexpr int SLOT_ELEMS = (LMODE == L_SCALAR) ? 1 : (LMODE == L_VECK ? 4 : 16);
  static constexpr int NSLOTS = ELEMS / SLOT_ELEMS;
  static constexpr int PER_THREAD = (NSLOTS + NT - 1) / NT;
  static constexpr bool EXACT = (NSLOTS % NT) == 0;
  float reg[PER_THREAD * SLOT_ELEMS];

  // slot -> (row, k) of its first element
  __device__ static inline void slot_rk(int e, bool k_contig, int& r, int& k) {
    if (LMODE == L_VECK) { constexpr int Q = TK / 4; k = 4 * (e % Q); r = e / Q; }
    else if (LMODE == L_BLK4) { constexpr int Q = ROWS / 4; r = 4 * (e % Q); k = 4 * (e / Q); }
    else if (k_contig) { k = e % TK; r = e / TK; }
    else { r = e % ROWS; k = e / ROWS; }
  }

  // TAIL = false: the whole chunk [k0, k0+TK) is inside K (no k checks)
  template <bool TAIL>
  __device__ inline void load(const GemmP& p, const float* __restrict__ g, const float* __restrict__ rowv,
                              const float* __restrict__ colv, int row0, int k0, int tid) {
    const long sr = IS_A ? p.a_sr : p.b_sr, sk = IS_A ? p.a_sk : p.b_sk;
    // rows that exist in memory: the wgrad "ones" row (bias-gradient column) is virtual and must never be read
    const int nrows = IS_A ? p.M : ((PRO == PB_ONES && p.ones_row < p.N) ? p.ones_row : p.N);
    const bool k_contig = (sk == 1);
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      const int e = tid + i * NT;
      if (!EXACT && e >= NSLOTS) break;
      int r, k;
      slot_rk(e, k_contig, r, k);
      const int gk = k0 + k;
      float* o = &reg[i * SLOT_ELEMS];
      if (LMODE == L_SCALAR) {
        int gr = row0 + r;
        gr = gr < nrows ? gr : nrows - 1;
        const int kk = TAIL ? (gk < p.K ? gk : p.K - 1) : gk;
        float v = g[(long)gr * sr + (long)kk * sk];
        if (PRO == PA_RANK1 && IS_A) {
          const int mm = p.a_trans ? kk : gr, nn = p.a_trans ? gr : kk;
          v = v > 0.f ? rowv[mm] * colv[nn] : 0.f;
        }
        if (PRO == PB_ONES && !IS_A) v = (row0 + r == p.ones_row) ? 1.f : v;
        if (TAIL) v = gk < p.K ? v : 0.f;
        o[0] = v;
      } else if (LMODE == L_VECK) {
        int gr = row0 + r;
        gr = gr < nrows ? gr : nrows - 1;
        const int kk = TAIL ? (gk < p.K ? gk : 0) : gk;     // K % 4 == 0 or zero-padded rows (host guarantees)
        f32x4 v = *(const f32x4*)&g[(long)gr * sr + kk];
        if (PRO == PA_RANK1 && IS_A) {
          const float rv = rowv[gr];
          const f32x4 cv = *(const f32x4*)&colv[kk];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? rv * cv[j] : 0.f;
        }
        if (TAIL && gk >= p.K) v = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = v[j];
      } else {  // L_BLK4
        const int rlim = IS_A ? p.a_rlim : p.b_rlim;
        int gr = row0 + r;
        gr = gr <= rlim - 4 ? gr : rlim - 4;
        f32x4 cv;
        if (PRO == PA_RANK1 && IS_A) cv = *(const f32x4*)&colv[gr];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int gkj = gk + j;
          const int kk = TAIL ? (gkj < p.K ? gkj : p.K - 1) : gkj;
          f32x4 v = *(const f32x4*)&g[(long)kk * sk + gr];
          if (PRO == PA_RANK1 && IS_A) {
            const float rv = rowv[kk];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) v[rr] = v[rr] > 0.f ? rv * cv[rr] : 0.f;
          }
          if (PRO == PB_ONES && !IS_A) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) v[rr] = (row0 + r + rr == p.ones_row) ? 1.f : v[rr];
          }
          if (TAIL && gkj >= p.K) v = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) o[rr * 4 + j] = v[rr];    // transpose: o[row][k]
        }
      }
    }
  }

  __device__ inline void store(float* __restrict__ lds, bool k_contig, int tid) const {
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      const int e = tid + i * NT;
      if (!EXACT && e >= NSLOTS) break;
      int r, k;
      slot_rk(e, k_contig, r, k);
      const float* o = &reg[i * SLOT_ELEMS];
      if (LMODE == L_SCALAR) lds[r * PITCH + k] = o[0];
      else if (LMODE == L_VECK) *(f32x4*)&lds[r * PITCH + k] = (f32x4){o[0], o[1], o[2], o[3]};
      else {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) *(f32x4*)&lds[(r + rr) * PITCH + k] = (f32x4){o[rr * 4], o[rr * 4 + 1], o[rr * 4 + 2], o[rr * 4 + 3]};
      }
    }
  }
};

template <class CFG, int LA, int LB, int PA, int PB, int EPI>
__global__ __launch_bounds__(CFG::NT) void gemm16_kernel(const GemmP p) {
  constexpr int TM = CFG::TM, TN = CFG::TN, TK = CFG::kTK, NT = CFG::NT, PITCH = CFG::PITCH;
  constexpr int MA = CFG::kMA, NB = CFG::kNB;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                       // [2][TM][PITCH]
  float* Bs = smem + 2 * TM * PITCH;      // [2][TN][PITCH]

  const int tid = threadIdx.x;
  const int z = blockIdx.z;
  const int z0 = z / p.nz1, z1 = z - z0 * p.nz1;
  const int tiles_n = (p.N + TN - 1) / TN;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
  const int m0 = tm * TM, n0 = tn * TN;
  const int ks = blockIdx.y;

  const float* __restrict__ Ag = p.A.at(z0, z1);
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
  const bool a_kc = (p.a_sk == 1), b_kc = (p.b_sk == 1);

  auto load_chunk = [&](int kc) {
    if (kc < kfull) { la.template load<false>(p, Ag, rowv, colv, m0, kc * TK, tid); lb.template load<false>(p, Bg, nullptr, nullptr, n0, kc * TK, tid); }
    else { la.template load<true>(p, Ag, rowv, colv, m0, kc * TK, tid); lb.template load<true>(p, Bg, nullptr, nullptr, n0, kc * TK, tid); }
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

  if (kc_begin < kc_end) {
    load_chunk(kc_begin);
    la.store(As, a_kc, tid);
    lb.store(Bs, b_kc, tid);
  }
  __syncthreads();
  for (int kc = kc_begin; kc < kc_end; ++kc) {
    const int buf = (kc - kc_begin) & 1;
    const bool more = kc + 1 < kc_end;
    if (more) load_chunk(kc + 1);
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
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[a][s], fb[b][s], acc[a][b], 0, 0, 0);
    }
    if (more) { la.store(As + (buf ^ 1) * TM * PITCH, a_kc, tid); lb.store(Bs + (buf ^ 1) * TN * PITCH, b_kc, tid); }
    __syncthreads();
  }

  // ---- epilogue: lane (li, lq) holds C[row = 4*lq + reg][col = li] of each 16x16 block ----
  float* Cg = p.C + z0 * p.c_s0 + z1 * p.c_s1 + (long)ks * p.c_ks;
  const float* __restrict__ bias = p.bias.at(z0, z1);
  const float* __restrict__ aux = p.aux.at(z0, z1);
  float* bo = (EPI == E_WGRAD && p.bias_out) ? p.bias_out + z0 * p.bo_s0 + z1 * p.bo_s1 + (long)ks * p.bo_ks : nullptr;
#pragma unroll
  for (int a = 0; a < MA; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int n = n0 + wcol0 + b * 16 + li;
      const bool n_ok = n < p.N;
      float bv = 0.f;
      if (EPI == E_BIAS || EPI == E_BIAS_RELU) bv = bias[n_ok ? n : 0];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wrow0 + a * 16 + 4 * lq + r;
        if (!(n_ok && m < p.M)) continue;
        float v = acc[a][b][r];
        if (EPI == E_BIAS) v += bv;
        if (EPI == E_BIAS_RELU) { v += bv; v = v > 0.f ? v : 0.f; }
        if (EPI == E_MASK) v = aux[(long)m * p.aux_sr + n] > 0.f ? v : 0.f;
        if (EPI == E_WGRAD) {
          if (n < p.ones_row) Cg[(long)m * p.c_sr + (long)n * p.c_sn] = v;
          else if (n == p.ones_row && bo) bo[m] = v;
        } else {
          Cg[(long)m * p.c_sr + (long)n * p.c_sn] = v;
        }
      }
    }
}

// tile configurations
typedef GemmCfg<1, 4, 4, 4, 32> CfgBig;    // 64 x 256 : big forward / dgrad (full hidden width per workgroup)
typedef GemmCfg<2, 2, 2, 2, 32> CfgMid;    // 64 x 64  : wgrad tiles with split-K
typedef GemmCfg<1, 4, 1, 1, 64> CfgSmall;  // 16 x 64  : batch-sized (256-row) phases, many workgroups
typedef GemmCfg<4, 1, 1, 1, 32> CfgTall;   // 64 x 16  : narrow outputs (heads, action-gradient columns)
enum { CFG_BIG = 0, CFG_MID = 1, CFG_SMALL = 2, CFG_TALL = 3, CFG_AUTO = -1 };

template <class CFG, int LA, int LB, int PA, int PB, int EPI>
static inline hipError_t launch_inst(const GemmP& p, int nz, hipStream_t st) {
  const int tiles = ((p.M + CFG::TM - 1) / CFG::TM) * ((p.N + CFG::TN - 1) / CFG::TN);
  dim3 grid(tiles, p.ksplit, nz), block(CFG::NT);
  const size_t lds = CFG::LDS_FLOATS * sizeof(float);
  auto kern = gemm16_kernel<CFG, LA, LB, PA, PB, EPI>;
  if (lds > 64 * 1024) {
    static bool raised = false;   // one flag per instantiation
    if (!raised) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      raised = true;
    }
  }
  hipLaunchKernelGGL(kern, grid, block, lds, st, p);
  return hipGetLastError();
}

// heuristic tile choice when cfg == CFG_AUTO
static inline int pick_cfg(int M, int N, int K) {
  if (N <= 16) return CFG_TALL;
  if (M >= 2048 && N >= 128) return CFG_BIG;
  if (M <= 32) return CFG_SMALL;
  // long reductions (wgrad over thousands of rows) get 64x64 tiles + split-K; batch-sized products get
  // many small workgroups instead
  return (K >= 1024) ? CFG_MID : CFG_SMALL;
}

static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// loader choice per operand from strides / alignment.  k_pad_ok: rows are zero-padded up to a multiple of 4 in k
static inline int pick_loader(const ZPtr& z, long sr, long sk, int K, bool k_pad_ok, int rlim) {
  if (!aligned16(z.p) || (z.s0 & 3) || (z.s1 & 3)) return L_SCALAR;
  if (sk == 1 && (sr & 3) == 0 && ((K & 3) == 0 || k_pad_ok)) return L_VECK;
  if (sr == 1 && (sk & 3) == 0 && rlim >= 4 && (rlim & 3) == 0) return L_BLK4;
  return L_SCALAR;
}

template <class CFG, int PA, int PB, int EPI>
static inline hipError_t launch_cfg(const GemmP& p, int la, int lb, int nz, hipStream_t st) {
  // supported loader pairs; anything else falls back to the scalar loaders
  if (la == L_VECK && lb == L_VECK) return launch_inst<CFG, L_VECK, L_VECK, PA, PB, EPI>(p, nz, st);
  if (la == L_VECK && lb == L_BLK4) return launch_inst<CFG, L_VECK, L_BLK4, PA, PB, EPI>(p, nz, st);
  if (la == L_BLK4 && lb == L_BLK4) return launch_inst<CFG, L_BLK4, L_BLK4, PA, PB, EPI>(p, nz, st);
  if (la == L_VECK) return launch_inst<CFG, L_VECK, L_SCALAR, PA, PB, EPI>(p, nz, st);
  return launch_inst<CFG, L_SCALAR, L_SCALAR, PA, PB, EPI>(p, nz, st);
}

template <int PA, int PB, int EPI>
static inline hipError_t launch_gemm(int cfg, const GemmP& p, int nz, hipStream_t st, bool a_kpad = false, bool force_scalar = false) {
  if (cfg == CFG_AUTO) cfg = pick_cfg(p.M, p.N, p.K);
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
    case CFG_BIG: return launch_cfg<CfgBig, PA, PB, EPI>(p, la, lb, nz, st);
    case CFG_MID: return launch_cfg<CfgMid, PA, PB, EPI>(p, la, lb, nz, st);
    case CFG_SMALL: return launch_cfg<CfgSmall, PA, PB, EPI>(p, la, lb, nz, st);
    default: return launch_cfg<CfgTall, PA, PB, EPI>(p, la, lb, nz, st);
  }
}

}  // namespace orl
