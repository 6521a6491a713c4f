// gemm.h — run-batched MFMA GEMM for the MLP forward / dgrad / wgrad of the update engine.
//
// One template covers every matrix product of the hot path (reference: nets/mlp.py:9-33 forward,
// autograd's mm/addmm/threshold_backward for the backward):
//
//     C[z][m][n] (+slab) = epi( sum_k  pro_a(A[z][m][k]) * pro_b(B[z][n][k]) )
//
// A and B are addressed with arbitrary element strides (row stride, k stride), so the same kernel does
//   forward  Y  = X  W^T      A = X [M x K] k-contiguous,  B = W  [N x K] k-contiguous
//   dgrad    dX = dY W        A = dY,                      B = W viewed as [K' x N'] (k strided)
//   wgrad    dW = dY^T X      A = dY viewed [N' x M] (k strided),  B = X viewed [K' x M] (k strided)
// Tiles are staged global -> registers -> LDS as [row][k] (k contiguous, pitch TK+4), prefetching the
// next K chunk into registers while the current one is multiplied.  The multiply is
// v_mfma_f32_16x16x4_f32 (exact fp32, 64 FLOP/clk/SIMD on gfx950): lane (i = l&15, q = l>>4) reads four
// consecutive k of row i with one ds_read_b128 and feeds them to four MFMAs, i.e. hardware k-slot q of
// MFMA s carries logical k = 4q + s for both operands (any permutation of k is a valid reduction order).
// Wavefront = 64 lanes; a workgroup is WM x WN waves, each wave owns an (MA*16) x (NB*16) block of C.
//
// z = blockIdx.z is the run/net batch index, decomposed z = z0 * nz1 + z1 (run, net) with two strides per
// operand, so twin critics and all runs of an engine go through one launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace orl {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// prologue applied to A elements while staging
enum { PA_PLAIN = 0, PA_RANK1 = 1 };
// prologue for B
enum { PB_PLAIN = 0, PB_ONES = 1 };
// epilogues
enum { E_PLAIN = 0, E_BIAS = 1, E_BIAS_RELU = 2, E_MASK = 3, E_WGRAD = 4 };

struct ZPtr {        // base + z0 * s0 + z1 * s1 (element strides)
  const float* p;
  long s0, s1;
  __host__ __device__ const float* at(int z0, int z1) const { return p ? p + z0 * s0 + z1 * s1 : nullptr; }
};

struct GemmP {
  ZPtr A, B;
  float* C;
  long c_s0, c_s1;   // z strides of C
  long c_sr;         // row stride of C (E_WGRAD: row stride of the weight-grad matrix = K_in)
  long c_sn;         // column stride of C (1 except for EnsembleLinear (in,out)-major weight gradients)
  long c_ks;         // split-K slab stride of C (elements)
  int M, N, K;
  long a_sr, a_sk;   // A[m*a_sr + k*a_sk]
  long b_sr, b_sk;   // B[n*b_sr + k*b_sk]
  int nz1;           // z = z0*nz1 + z1
  int ksplit;        // blockIdx.y in [0, ksplit)
  // PA_RANK1: a(m', n') = A(m', n') > 0 ? rowv[m'] * colv[n'] : 0   (dz_L = dq (x) w_last (.) relu-mask)
  //   a_trans = 0: tile row = m', tile k = n' (dgrad)   a_trans = 1: tile row = n', tile k = m' (wgrad)
  ZPtr rowv, colv;
  int a_trans;
  // PB_ONES: logical B row index == ones_row -> 1.0 (bias-gradient column of wgrad)
  int ones_row;
  // E_BIAS / E_BIAS_RELU: bias[n]
  ZPtr bias;
  // E_MASK: C = aux[m*aux_sr + n] > 0 ? acc : 0
  ZPtr aux;
  long aux_sr;
  // E_WGRAD: n < ones_row -> C[m*c_sr + n];  n == ones_row -> bias_out[m]   (same slab / z strides as C)
  float* bias_out;
  long bo_s0, bo_s1, bo_ks;
};

template <int WM, int WN, int MA, int NB, int TK>
struct GemmCfg {
  static constexpr int kWM = WM, kWN = WN, kMA = MA, kNB = NB, kTK = TK;
  static constexpr int TM = WM * MA * 16, TN = WN * NB * 16, NT = WM * WN * 64, PITCH = TK + 4;
  static constexpr int A_REGS = TM * TK / NT, B_REGS = TN * TK / NT;
  static constexpr int LDS_FLOATS = 2 * (TM + TN) * PITCH;
  static_assert((TM * TK) % NT == 0 && (TN * TK) % NT == 0, "tile must divide evenly over threads");
  static_assert(TK % 16 == 0, "TK multiple of 16");
};

template <class CFG, int PA, int PB, int EPI>
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

  // K range of this split (chunk aligned)
  const int kchunks = (p.K + TK - 1) / TK;
  const int per = (kchunks + p.ksplit - 1) / p.ksplit;
  const int kc_begin = ks * per;
  int kc_end = kc_begin + per;
  if (kc_end > kchunks) kc_end = kchunks;

  float ra[CFG::A_REGS], rb[CFG::B_REGS];

  auto load_a = [&](int k0) {
#pragma unroll
    for (int i = 0; i < CFG::A_REGS; ++i) {
      const int e = tid + i * NT;
      int r, k;
      if (p.a_sk == 1) { k = e % TK; r = e / TK; } else { r = e % TM; k = e / TM; }
      const int gr = m0 + r, gk = k0 + k;
      float v = 0.f;
      if (gr < p.M && gk < p.K) {
        v = Ag[(long)gr * p.a_sr + (long)gk * p.a_sk];
        if (PA == PA_RANK1) {
          const int mm = p.a_trans ? gk : gr, nn = p.a_trans ? gr : gk;
          v = v > 0.f ? rowv[mm] * colv[nn] : 0.f;
        }
      }
      ra[i] = v;
    }
  };
  auto load_b = [&](int k0) {
#pragma unroll
    for (int i = 0; i < CFG::B_REGS; ++i) {
      const int e = tid + i * NT;
      int r, k;
      if (p.b_sk == 1) { k = e % TK; r = e / TK; } else { r = e % TN; k = e / TN; }
      const int gr = n0 + r, gk = k0 + k;
      float v = 0.f;
      if (gr < p.N && gk < p.K) {
        if (PB == PB_ONES && gr == p.ones_row) v = 1.f;
        else v = Bg[(long)gr * p.b_sr + (long)gk * p.b_sk];
      }
      rb[i] = v;
    }
  };
  auto store_ab = [&](int buf) {
    float* as = As + buf * TM * PITCH;
    float* bs = Bs + buf * TN * PITCH;
#pragma unroll
    for (int i = 0; i < CFG::A_REGS; ++i) {
      const int e = tid + i * NT;
      int r, k;
      if (p.a_sk == 1) { k = e % TK; r = e / TK; } else { r = e % TM; k = e / TM; }
      as[r * PITCH + k] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < CFG::B_REGS; ++i) {
      const int e = tid + i * NT;
      int r, k;
      if (p.b_sk == 1) { k = e % TK; r = e / TK; } else { r = e % TN; k = e / TN; }
      bs[r * PITCH + k] = rb[i];
    }
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
    load_a(kc_begin * TK);
    load_b(kc_begin * TK);
    store_ab(0);
  }
  __syncthreads();
  for (int kc = kc_begin; kc < kc_end; ++kc) {
    const int buf = (kc - kc_begin) & 1;
    const bool more = kc + 1 < kc_end;
    if (more) { load_a((kc + 1) * TK); load_b((kc + 1) * TK); }
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
    if (more) store_ab(buf ^ 1);
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
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wrow0 + a * 16 + 4 * lq + r;
        if (m >= p.M || n >= p.N) continue;
        float v = acc[a][b][r];
        if (EPI == E_BIAS) v += bias[n];
        if (EPI == E_BIAS_RELU) { v += bias[n]; v = v > 0.f ? v : 0.f; }
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
typedef GemmCfg<1, 4, 4, 4, 16> CfgBig;    // 64 x 256 : big forward / dgrad (full hidden width per workgroup)
typedef GemmCfg<2, 2, 2, 2, 16> CfgMid;    // 64 x 64  : wgrad tiles with split-K
typedef GemmCfg<1, 4, 1, 1, 64> CfgSmall;  // 16 x 64  : batch-sized (256-row) phases, many workgroups
typedef GemmCfg<4, 1, 1, 1, 16> CfgTall;   // 64 x 16  : narrow outputs (heads, action-gradient columns)
enum { CFG_BIG = 0, CFG_MID = 1, CFG_SMALL = 2, CFG_TALL = 3, CFG_AUTO = -1 };

template <class CFG, int PA, int PB, int EPI>
static inline hipError_t launch_cfg(const GemmP& p, int nz, hipStream_t st) {
  const int tiles = ((p.M + CFG::TM - 1) / CFG::TM) * ((p.N + CFG::TN - 1) / CFG::TN);
  dim3 grid(tiles, p.ksplit, nz), block(CFG::NT);
  const size_t lds = CFG::LDS_FLOATS * sizeof(float);
  hipLaunchKernelGGL((gemm16_kernel<CFG, PA, PB, EPI>), grid, block, lds, st, p);
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

template <int PA, int PB, int EPI>
static inline hipError_t launch_gemm(int cfg, const GemmP& p, int nz, hipStream_t st) {
  if (cfg == CFG_AUTO) cfg = pick_cfg(p.M, p.N, p.K);
  switch (cfg) {
    case CFG_BIG: return launch_cfg<CfgBig, PA, PB, EPI>(p, nz, st);
    case CFG_MID: return launch_cfg<CfgMid, PA, PB, EPI>(p, nz, st);
    case CFG_SMALL: return launch_cfg<CfgSmall, PA, PB, EPI>(p, nz, st);
    default: return launch_cfg<CfgTall, PA, PB, EPI>(p, nz, st);
  }
}

}  // namespace orl
