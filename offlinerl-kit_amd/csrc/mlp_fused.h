// mlp_fused.h — fused MLP forward: x -> [Linear+ReLU] x L -> tail Linear, one launch per network family.
//
// Reference math: nets/mlp.py:9-33 + modules/critic_module.py:17-28 / actor heads (y = x W^T + b, ReLU between).
// A workgroup owns TM rows and walks the layers: the activation tile lives in LDS (A operand, never re-read from
// HBM), each layer's weights are streamed from L2/HBM in K chunks through a double-buffered LDS tile (B operand),
// the post-ReLU activations are written to HBM once (the backward needs them) and back into the LDS tile for the
// next layer.  The tail (N = out_dim <= 16) runs as one more MFMA "layer" on the first column wave.
// Versus one launch per layer this removes, per hidden layer, one read of the activation matrix, one kernel
// boundary and one pipeline fill — the layer-wise kernels are HBM-bound in the split-bf16 mode.
//
// Geometry: 512 threads = 8 waves as WM x WN, wave tile (MA*16) x (NB*16), WN*NB*16 == H (hidden width).
//   <2,4,2,4>: TM = 64 rows  (thousands of rows)      <1,8,1,2>: TM = 16 rows (batch-sized phases: more workgroups)
#pragma once
#include "gemm.h"

namespace orl {

struct FusedFwdP {
  ZPtr X; long x_sr; int in_dim, in_pad;     // input rows (pitch x_sr >= in_pad = roundup4(in_dim), zero padded)
  int M, L, H, out_dim;
  ZPtr Wt;                                    // parameter block of (run, member); nn.Linear layout
  long w_off[ORL_MAX_HIDDEN + 1], b_off[ORL_MAX_HIDDEN + 1];
  float* hs[ORL_MAX_HIDDEN]; long h_s0[ORL_MAX_HIDDEN], h_s1[ORL_MAX_HIDDEN]; int h_pitch[ORL_MAX_HIDDEN];
  float* out; long o_s0, o_s1; int o_pitch;   // tail output [M x out_dim]
  int nz1;
};

template <int WM, int WN, int MA, int NB, int PREC>
__global__ __launch_bounds__(512) void mlp_fwd_kernel(const FusedFwdP p) {
  constexpr int TM = WM * MA * 16, H = WN * NB * 16, TK = 32, NT = 512;
  constexpr int PW = (PREC == P_F32) ? TK + 4 : TK + 8;        // weight tile pitch (elements)
  constexpr int PA = (PREC == P_F32) ? H + 4 : H + 8;          // activation tile pitch (elements)
  static_assert(WM * WN == 8, "8 waves");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // fp32:  act [TM][PA] floats | wbuf [2][H][PW] floats
  // bf16:  act [hi,lo][TM][PA] bf16 | wbuf [2][hi,lo][H][PW] bf16
  float* act = smem;
  float* wbuf = smem + TM * PA;
  __bf16* acth = (__bf16*)smem;
  __bf16* wbh = acth + 2 * TM * PA;

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int li = lane & 15, lq = lane >> 4;
  const int wrow0 = wm * MA * 16, wcol0 = wn * NB * 16;
  const int z = blockIdx.z, z0 = z / p.nz1, z1 = z - z0 * p.nz1;
  const int m0 = blockIdx.x * TM;
  const float* __restrict__ Xg = p.X.at(z0, z1);
  const float* __restrict__ Wg = p.Wt.at(z0, z1);

  // ---- stage the input rows (zero-filled to a multiple of TK so that stale LDS never meets a zero weight) ----
  {
    const int kpad = ((p.in_pad + TK - 1) / TK) * TK;
    for (int e = tid; e < TM * (kpad / 4); e += NT) {
      const int r = e / (kpad / 4), k = 4 * (e - r * (kpad / 4));
      int gr = m0 + r; gr = gr < p.M ? gr : p.M - 1;
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (k < p.in_pad) v = *(const f32x4*)&Xg[(long)gr * p.x_sr + k];
      if (PREC == P_F32) *(f32x4*)&act[r * PA + k] = v;
      else {
        bf16x4 h, l;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const __bf16 hh = (__bf16)v[j]; h[j] = hh; l[j] = (__bf16)(v[j] - (float)hh); }
        *(bf16x4*)&acth[r * PA + k] = h; *(bf16x4*)&acth[TM * PA + r * PA + k] = l;
      }
    }
  }

  const int nlayers = p.L + 1;      // hidden layers + tail
  for (int l = 0; l < nlayers; ++l) {
    const bool tail = (l == p.L);
    const int K = (l == 0) ? p.in_dim : H;                 // true reduction length (weights' row length)
    const int N = tail ? p.out_dim : H;
    // weight tile loader for this layer: rows n in [0,N), k contiguous (nn.Linear W (out,in))
    GemmP q;
    q.b_sr = K; q.b_sk = 1; q.N = N; q.K = K; q.M = 0; q.a_sr = 0; q.a_sk = 0; q.b_rlim = 0; q.a_rlim = 0; q.ones_row = 1 << 30; q.a_trans = 0;
    const float* __restrict__ Wl = Wg + p.w_off[l];
    const bool vec = ((K & 3) == 0);                       // W rows 16-B aligned (block base is 16-B aligned by layout)
    TileLoader<H, TK, NT, PW, L_VECK, false, PB_PLAIN> lv;
    TileLoader<H, TK, NT, PW, L_VECKU, false, PB_PLAIN> ls;
    if (vec) lv.init(q, 0, tid); else ls.init(q, 0, tid);
    const int kchunks = (K + TK - 1) / TK, kfull = K / TK;
    auto loadw = [&](int kc) {
      if (vec) { if (kc < kfull) lv.template load<false>(q, Wl, nullptr, nullptr, kc * TK, tid); else lv.template load<true>(q, Wl, nullptr, nullptr, kc * TK, tid); }
      else { if (kc < kfull) ls.template load<false>(q, Wl, nullptr, nullptr, kc * TK, tid); else ls.template load<true>(q, Wl, nullptr, nullptr, kc * TK, tid); }
    };
    auto storew = [&](int buf) {
      if (PREC == P_F32) { float* d = wbuf + buf * H * PW; if (vec) lv.store(d, tid); else ls.store(d, tid); }
      else { __bf16* d = wbh + buf * 2 * H * PW; if (vec) lv.store_split(d, tid); else ls.store_split(d, tid); }
    };

    f32x4 acc[MA][NB];
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool active = !tail || wn == 0;      // the tail's single 16-column block is computed by the first column wave(s)

    loadw(0);
    __syncthreads();                           // previous layer: everyone finished reading wbuf / writing act
    storew(0);
    __syncthreads();
    for (int kc = 0; kc < kchunks; ++kc) {
      const int buf = kc & 1;
      const bool more = kc + 1 < kchunks;
      if (more) loadw(kc + 1);
      if (active) {
        const int k0 = kc * TK;
        if (PREC == P_F32) {
          const float* bs = wbuf + buf * H * PW;
#pragma unroll
          for (int kk = 0; kk < TK; kk += 16) {
            f32x4 fa[MA], fb[NB];
#pragma unroll
            for (int a = 0; a < MA; ++a) fa[a] = *(const f32x4*)&act[(wrow0 + a * 16 + li) * PA + k0 + kk + 4 * lq];
#pragma unroll
            for (int b = 0; b < NB; ++b) fb[b] = *(const f32x4*)&bs[((tail ? 0 : wcol0) + b * 16 + li) * PW + kk + 4 * lq];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
              for (int a = 0; a < MA; ++a)
#pragma unroll
                for (int b = 0; b < NB; ++b)
                  if (!tail || b == 0) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[b][s], fa[a][s], acc[a][b], 0, 0, 0);
          }
        } else {
          const __bf16* ah = acth;
          const __bf16* al = acth + TM * PA;
          const __bf16* bh = wbh + buf * 2 * H * PW;
          const __bf16* bl = bh + H * PW;
          bf16x8 fah[MA], fal[MA], fbh[NB], fbl[NB];
#pragma unroll
          for (int a = 0; a < MA; ++a) {
            const int o = (wrow0 + a * 16 + li) * PA + k0 + 8 * lq;
            fah[a] = *(const bf16x8*)&ah[o]; fal[a] = *(const bf16x8*)&al[o];
          }
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            const int o = ((tail ? 0 : wcol0) + b * 16 + li) * PW + 8 * lq;
            fbh[b] = *(const bf16x8*)&bh[o]; fbl[b] = *(const bf16x8*)&bl[o];
          }
#pragma unroll
          for (int a = 0; a < MA; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b)
              if (!tail || b == 0) {
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fbl[b], fah[a], acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fbh[b], fal[a], acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fbh[b], fah[a], acc[a][b], 0, 0, 0);
              }
        }
      }
      if (more) storew(buf ^ 1);
      __syncthreads();
    }

    // ---- epilogue: lane (li, lq) holds out[m = li][n = 4*lq + r] of each block (transposed MFMA tile) ----
    const float* __restrict__ bias = Wg + p.b_off[l];
    if (!tail) {
      float* Hg = p.hs[l] + z0 * p.h_s0[l] + z1 * p.h_s1[l];
      const int hp = p.h_pitch[l];
#pragma unroll
      for (int a = 0; a < MA; ++a) {
        const int r = wrow0 + a * 16 + li, m = m0 + r;
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const int n = wcol0 + b * 16 + 4 * lq;
          f32x4 v = acc[a][b] + *(const f32x4*)&bias[n];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
          if (m < p.M) *(f32x4*)&Hg[(long)m * hp + n] = v;
          // next layer's A operand (all chunk reads of this layer finished at the loop's last barrier)
          if (PREC == P_F32) *(f32x4*)&act[r * PA + n] = v;
          else {
            bf16x4 h, lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) { const __bf16 hh = (__bf16)v[j]; h[j] = hh; lo[j] = (__bf16)(v[j] - (float)hh); }
            *(bf16x4*)&acth[r * PA + n] = h; *(bf16x4*)&acth[TM * PA + r * PA + n] = lo;
          }
        }
      }
    } else if (wn == 0) {
      float* Og = p.out + z0 * p.o_s0 + z1 * p.o_s1;
#pragma unroll
      for (int a = 0; a < MA; ++a) {
        const int m = m0 + wrow0 + a * 16 + li;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = 4 * lq + j;
          if (n < p.out_dim) Og[(long)m * p.o_pitch + n] = acc[a][0][j] + bias[n];
        }
      }
    }
  }
}

template <int WM, int WN, int MA, int NB, int PREC>
static inline hipError_t launch_fused_fwd_inst(const FusedFwdP& p, int nz, hipStream_t st) {
  constexpr int TM = WM * MA * 16, H = WN * NB * 16, TK = 32;
  constexpr size_t lds = (PREC == P_F32) ? sizeof(float) * ((size_t)TM * (H + 4) + 2 * H * (TK + 4))
                                         : 2 * ((size_t)2 * TM * (H + 8) + 2 * 2 * H * (TK + 8));
  auto kern = mlp_fwd_kernel<WM, WN, MA, NB, PREC>;
  static const hipError_t attr_err =      // thread-safe one-time initialisation, one per instantiation
      hipFuncSetAttribute((const void*)mlp_fwd_kernel<WM, WN, MA, NB, PREC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (attr_err != hipSuccess) return attr_err;
  dim3 grid((p.M + TM - 1) / TM, 1, nz);
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, st, p);
  return hipGetLastError();
}

// eligibility + dispatch.  Returns hipErrorNotSupported when the layer-wise path must be used.
static inline hipError_t launch_fused_fwd(const FusedFwdP& p, int nz, int prec, hipStream_t st) {
  if (p.H != 256 || p.out_dim > 16 || p.L < 1 || p.L > ORL_MAX_HIDDEN) return hipErrorNotSupported;
  const bool big = (long)p.M * nz >= 4096;
  if (prec == P_BF16X3) return big ? launch_fused_fwd_inst<2, 4, 2, 4, P_BF16X3>(p, nz, st) : launch_fused_fwd_inst<1, 8, 1, 2, P_BF16X3>(p, nz, st);
  return big ? launch_fused_fwd_inst<2, 4, 2, 4, P_F32>(p, nz, st) : launch_fused_fwd_inst<1, 8, 1, 2, P_F32>(p, nz, st);
}

}  // namespace orl
