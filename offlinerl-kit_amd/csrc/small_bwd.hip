// small_bwd.hip — fused SAC-style actor update for few batched rows (interface and design notes: small_bwd.h).
#include "small_bwd.h"

#include "scalars.h"

namespace orl {

// LDS layout (bytes).  Split planes: 16-bit elements; F32: floats.
//   Z  : dz1, the gradient w.r.t. the second hidden pre-activation of the 32 rows (split: hi + lo planes [32][SB_ZP]; F32: [32][SB_ZF])
//   H  : the first hidden activation h0 (split: hi + lo planes [32][256], 16-byte chunks XOR-swizzled per row for the transposing reads;
//        F32: [32][SB_HF] floats);  M: split only, 1[h0 > 0] as bytes [32][256] (the fp16 hi plane would lose positives below 2^-25)
//   W  : two buffers of one 32-ROW chunk of W1 (rows = output units n = the contraction index of dz0 = dz1 W1): split hi + lo planes
//        [32][256] swizzled, F32 [32][SB_TF].  Buffer 1 first holds the fp32 h1 tile [32][256]; after the last chunk the region holds
//        dz0 [32][SB_DP] fp32 and the input tile [32][SB_XP] fp32 (ones column at in0) for the first layer's weight gradient.
//   DH : dhead [32][SB_DHP];  WH : head weights [16][SB_WHP];  RED : scratch
// The products with a 16- or 32-deep contraction (dz1 = dhead W_head, dW_head = dhead^T h1, dW0 = dz0^T [x | 1]) run on the exact fp32 MFMA
// (v_mfma_f32_16x16x4_f32) in BOTH precisions: 16 - 32 instructions per wave each, operands single floats read from the fp32 tiles.  As
// VALU loops they were the two longest phases of the kernel (17 000 + 12 000 of 100 000 clocks: per-thread dot products fed by LDS
// broadcasts keep the LDS pipe busy for 8 clocks per 16-byte read whatever the lanes share).
enum { SB_ZP = 272, SB_ZF = 260, SB_HF = 272, SB_TF = 260, SB_DP = 272, SB_XP = 80, SB_DHP = 17, SB_WHP = 272 };
template <bool F32> static constexpr size_t sb_z_bytes() { return F32 ? (size_t)SB_ROWS * SB_ZF * 4 : (size_t)2 * SB_ROWS * SB_ZP * 2; }
template <bool F32> static constexpr size_t sb_h_bytes() { return F32 ? (size_t)SB_ROWS * SB_HF * 4 : (size_t)2 * SB_ROWS * SB_N * 2; }
template <bool F32> static constexpr size_t sb_m_bytes() { return F32 ? 0 : (size_t)SB_ROWS * SB_N; }
template <bool F32> static constexpr size_t sb_w_bytes() { return F32 ? (size_t)2 * SB_ROWS * SB_TF * 4 : (size_t)2 * 2 * SB_ROWS * SB_N * 2; }
enum { SB_SMALL_FLOATS = SB_ROWS * SB_DHP + 16 * SB_WHP + 96 };
template <bool F32> static constexpr size_t sb_lds_bytes() {
  return sb_z_bytes<F32>() + sb_h_bytes<F32>() + sb_m_bytes<F32>() + sb_w_bytes<F32>() + (size_t)(((SB_SMALL_FLOATS + 3) / 4) * 4) * 4;
}
static_assert(sb_w_bytes<false>() / 2 >= (size_t)SB_ROWS * SB_N * 4 && sb_w_bytes<true>() / 2 >= (size_t)SB_ROWS * SB_N * 4, "the h1 tile fits one chunk buffer");
static_assert(sb_w_bytes<false>() >= (size_t)SB_ROWS * (SB_DP + SB_XP) * 4 && sb_w_bytes<true>() >= (size_t)SB_ROWS * (SB_DP + SB_XP) * 4, "dz0 + input tile fit the chunk buffers");
static_assert(sb_lds_bytes<false>() <= 163840 && sb_lds_bytes<true>() <= 163840, "LDS budget of one CU");

__device__ __forceinline__ int sb_toff(int r, int chunk, int half) { return r * SB_N + ((chunk ^ (2 * (r & 7))) << 3) + (half << 2); }
// transposing LDS reads (ds_read_b64_tr_b16): lane li of 16-lane group lq receives column col0 + li of rows row0 + 4 lq .. + 3
__device__ __forceinline__ s16x4 sb_tr_sw(const hx_t* img, int row0, int col0, int lane) {          // swizzled [32][256] plane
  const int li = lane & 15, lq = lane >> 4, row = row0 + 4 * lq + (li >> 2), col = col0 + 4 * (li & 3);
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(img + sb_toff(row, col >> 3, (col >> 2) & 1)));
}
__device__ __forceinline__ s16x4 sb_tr_z(const hx_t* img, int row0, int col0, int lane) {           // plain [32][SB_ZP] plane
  const int li = lane & 15, lq = lane >> 4, row = row0 + 4 * lq + (li >> 2), col = col0 + 4 * (li & 3);
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(img + row * SB_ZP + col));
}
__device__ __forceinline__ hx8 sb_cat(s16x4 x, s16x4 y) {
  hx8 r;
  *(s16x4*)&r = x; *((s16x4*)&r + 1) = y;
  return r;
}
// sum over the 16 lanes of a row group (quad pairs, quads, half-row mirror, row mirror: every lane ends with the full sum, fixed order)
__device__ __forceinline__ float sb_row_sum16(float t) {
  t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xB1, 0xF, 0xF, false));
  t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x4E, 0xF, 0xF, false));
  t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x141, 0xF, 0xF, false));
  t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x140, 0xF, 0xF, false));
  return t;
}

#ifdef SB_LAB_CLOCK
#define SB_STAMP(i) do { if (p.lab_clk && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) p.lab_clk[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SB_STAMP(i) do { } while (0)
#endif

template <bool F32>
__global__ __launch_bounds__(SB_NT) void small_abwd_kernel(const SmallABwdP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sb_smem[];
  unsigned char* sZ = sb_smem;
  unsigned char* sH = sZ + sb_z_bytes<F32>();
  unsigned char* sM = sH + sb_h_bytes<F32>();
  unsigned char* sW = sM + sb_m_bytes<F32>();
  float* sDH = (float*)(sW + sb_w_bytes<F32>());
  float* sWh = sDH + SB_ROWS * SB_DHP;
  float* sRed = sWh + 16 * SB_WHP;
  float* sH1 = (float*)(sW + sb_w_bytes<F32>() / 2);                  // the h1 tile lives in chunk buffer 1 until dz1 exists
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lq = lane >> 4;
  const int g = blockIdx.x, z0 = blockIdx.y, ncol0 = 32 * wave;
  const int A = p.A, A2 = 2 * A, M = p.M;
  const long row0 = (long)g * SB_ROWS;
  const float* __restrict__ W1g = p.W1 + z0 * p.w1_s0;
  const float* __restrict__ Whg = p.Wh + z0 * p.wh_s0;
  float* __restrict__ slab = p.out + z0 * p.o_s0 + (long)g * p.o_ks;
  RunScalars& sc = p.sc[z0];

  SB_STAMP(0);
  // ---- every global load of the launch up front: the eight 32-row chunks of W1 (128 VGPRs until the dgrad loop has stored them: a ring
  // of register sets refilled inside that loop made every iteration wait out an L2 / fabric round trip, 3000 - 7000 clocks per chunk
  // against ~1800), the h0 / h1 tiles, the input rows, the head weights ----
  f32x4 wr[8][4];
#pragma unroll
  for (int c = 0; c < 8; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + SB_NT * i;
      wr[c][i] = *(const f32x4*)&W1g[(long)(32 * c + (e >> 6)) * SB_N + 4 * (e & 63)];
    }
  f32x4 hv0[4], hv1[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = tid + SB_NT * i;
    hv0[i] = *(const f32x4*)&(p.H0 + z0 * p.h0_s0)[(row0 + (e >> 6)) * SB_N + 4 * (e & 63)];
    hv1[i] = *(const f32x4*)&(p.H1 + z0 * p.h1_s0)[(row0 + (e >> 6)) * SB_N + 4 * (e & 63)];
  }
  float xs[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + SB_NT * i, r = e >> 5, k = e & 31;
    const float x = (p.X + z0 * p.x_s0)[(row0 + r) * p.x_pitch + (k < p.in0 ? k : p.in0 - 1)];
    xs[i] = (k < p.in0 ? 1.f : 0.f) * x + (k == p.in0 ? 1.0f : 0.f);        // ones column: its weight-gradient column is db0
  }
  const int nwh = A2 * SB_N;
  f32x4 wts[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = 4 * (tid + SB_NT * i);
    wts[i] = *(const f32x4*)&Whg[e < nwh ? e : nwh - 4] * (e < nwh ? 1.f : 0.f);            // rows >= 2A of the LDS copy are zero
  }
  // ---- dhead: thread (row r, column j) of [32][16]; j < A: d/d mu, A <= j < 2A: d/d log sigma (k_head_bwd's arithmetic), else 0 ----
  const float alpha = p.auto_alpha ? sc.alpha : p.fixed_alpha;
  {
    const int r = tid >> 4, j = tid & 15;
    const int a = j < A ? j : (j < A2 ? j - A : 0);
    const long m = row0 + r;
    const float q0 = (p.qa + z0 * p.qa_s0)[m], q1 = (p.qa + z0 * p.qa_s0 + p.qa_s1)[m];
    const float g0 = (p.ga + z0 * p.ga_s0)[m * p.ga_pitch + a], g1 = (p.ga + z0 * p.ga_s0 + p.ga_s1)[m * p.ga_pitch + a];
    const float lsr = (p.head + z0 * p.head_s0)[m * A2 + A + a];
    const float ep = (p.eps + z0 * p.eps_s0)[m * A + a];
    const float act = (p.xa + z0 * p.xa_s0)[m * p.xa_pitch + p.xa_col + a];
    const float lp = (p.logp + z0 * p.logp_s0)[m];
    const float qmin = fminf(q0, q1);
    const float nmin = (float)((q0 == qmin) + (q1 == qmin));
    const float gq = -1.0f / (float)M;
    // dL/da = sum_c dL/dq_c dq_c/da with dL/dq_c = -1/B on the smaller critic (split between equal ones): cql.py:93-98
    const float da = ((q0 == qmin) ? gq / nmin : 0.f) * g0 + ((q1 == qmin) ? gq / nmin : 0.f) * g1;
    const float dlogp = alpha / (float)M;
    const float sg = expf(fminf(fmaxf(lsr, -5.0f), 2.0f));
    const float om = 1.0f - act * act;
    const float t = 2.0f * act * om / (om + 1e-6f);
    const float du = da * om + dlogp * t;
    const float dls = du * sg * ep - dlogp;
    const float dl2 = (lsr >= -5.0f && lsr <= 2.0f) ? dls : 0.f;
    sDH[r * SB_DHP + j] = j < A ? du : (j < A2 ? dl2 : 0.f);
    if (j == 0) { sRed[r] = alpha * lp - qmin; sRed[SB_ROWS + r] = lp; }
  }
  // ---- tiles -> LDS ----
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = tid + SB_NT * i, r = e >> 6, col = 4 * (e & 63);
    *(f32x4*)&sH1[r * SB_N + col] = hv1[i];
    if constexpr (F32) *(f32x4*)((float*)sH + r * SB_HF + col) = hv0[i];
    else {
      hx4 h, l;
      orl_split4(hv0[i], h, l);
      const int o = sb_toff(r, col >> 3, (col >> 2) & 1);
      *(hx4*)((hx_t*)sH + o) = h;
      *(hx4*)((hx_t*)sH + SB_ROWS * SB_N + o) = l;
      unsigned int mb = 0;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) mb |= (hv0[i][jj] > 0.f ? 1u : 0u) << (8 * jj);
      *(unsigned int*)(sM + r * SB_N + col) = mb;
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = 4 * (tid + SB_NT * i);
    *(f32x4*)&sWh[(e >> 8) * SB_WHP + (e & 255)] = wts[i];
  }
  auto store_chunkT = [&](int c) __attribute__((always_inline)) {      // chunk c -> buffer c & 1
    const int buf = c & 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + SB_NT * i, r = e >> 6, col = 4 * (e & 63);
      if constexpr (F32) *(f32x4*)((float*)sW + ((long)buf * SB_ROWS + r) * SB_TF + col) = wr[c][i];
      else {
        hx_t* th = (hx_t*)sW + (long)buf * 2 * SB_ROWS * SB_N;
        hx4 h, l;
        orl_split4(wr[c][i] * ORL_WSCALE, h, l);
        const int o = sb_toff(r, col >> 3, (col >> 2) & 1);
        *(hx4*)(th + o) = h;
        *(hx4*)(th + SB_ROWS * SB_N + o) = l;
      }
    }
  };
  SB_STAMP(1);
  store_chunkT(0);
  __syncthreads();                                                     // B1: dhead, h1 tile, head weights, loss terms, chunk 0 visible
  SB_STAMP(2);

  // loss terms of this row group, summed by lane 0 in a fixed order; published at the very end of the kernel
  float s_loss = 0.f, s_lp = 0.f;
  if (tid == 0) {
#pragma unroll
    for (int r4 = 0; r4 < SB_ROWS; r4 += 4) {                           // (unrolled vector reads: all 16 in flight)
      const f32x4 a = *(const f32x4*)&sRed[r4], b = *(const f32x4*)&sRed[SB_ROWS + r4];
      s_loss += (a[0] + a[1]) + (a[2] + a[3]);
      s_lp += (b[0] + b[1]) + (b[2] + b[3]);
    }
  }
  // ---- dz1[m][n] = 1[h1 > 0] (dhead W_head)[m][n] on the fp32 MFMA (K = 16): lane holds C[m = 16 s + li][n = ncol0 + 16 cb + 4 lq + r] ----
  f32x4 dz[2][2];
  float amax = 0.f;
  {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) dz[s][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float a[2], b[2];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) a[cb] = sWh[(4 * t + lq) * SB_WHP + ncol0 + 16 * cb + li];
#pragma unroll
      for (int s = 0; s < 2; ++s) b[s] = sDH[(16 * s + li) * SB_DHP + 4 * t + lq];
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) dz[s][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cb], b[s], dz[s][cb], 0, 0, 0);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const f32x4 h = *(const f32x4*)&sH1[(16 * s + li) * SB_N + ncol0 + 16 * cb + 4 * lq];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dz[s][cb][r] = h[r] > 0.f ? dz[s][cb][r] : 0.f;
          amax = fmaxf(amax, fabsf(dz[s][cb][r]));
        }
      }
    // db1[n] = sum_m dz1[m][n]: the lane's two row blocks, then the 16 lanes of its row group
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t = sb_row_sum16(dz[0][cb][r] + dz[1][cb][r]);
        if (li == 0) slab[p.off_b1 + ncol0 + 16 * cb + 4 * lq + r] = t;
      }
  }
  // ---- dW_head[j][n] = sum_m dhead[m][j] h1[m][n] (K = 32 rows) and the head bias gradient ----
  {
    f32x4 hacc[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) hacc[cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float a = sDH[(4 * t + lq) * SB_DHP + li];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) hacc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, sH1[(4 * t + lq) * SB_N + ncol0 + 16 * cb + li], hacc[cb], 0, 0, 0);
    }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * lq + r < A2) slab[p.off_wh + (long)(4 * lq + r) * SB_N + ncol0 + 16 * cb + li] = hacc[cb][r];
    if (tid < A2) {
      float sbh = 0.f;
#pragma unroll
      for (int m = 0; m < SB_ROWS; ++m) sbh += sDH[m * SB_DHP + tid];           // (unrolled: 32 reads in flight, not 32 round trips)
      slab[p.off_bh + tid] = sbh;
    }
  }
  // split precision: one power-of-two scale per workgroup for its gradient matrices, chosen from max |dz1| (largest entry lands in [8, 16));
  // the slabs are written unscaled
  float gs = 1.0f;
  if constexpr (!F32) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_down(amax, o, 64));
    if (lane == 0) sRed[64 + wave] = amax;
  }
  SB_STAMP(3);
  __syncthreads();                                                     // B2: every read of the h1 tile is done (chunk buffer 1 is free)
  if constexpr (!F32) {
    float a = sRed[64];
#pragma unroll
    for (int w = 1; w < 8; ++w) a = fmaxf(a, sRed[64 + w]);
    gs = orl_pow2_scale(a);
  }
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int m = 16 * s + li, n = ncol0 + 16 * cb + 4 * lq;
      if constexpr (F32) *(f32x4*)((float*)sZ + m * SB_ZF + n) = dz[s][cb];
      else {
        hx4 h, l;
        orl_split4(dz[s][cb] * gs, h, l);
        *(hx4*)((hx_t*)sZ + m * SB_ZP + n) = h;
        *(hx4*)((hx_t*)sZ + SB_ROWS * SB_ZP + m * SB_ZP + n) = l;
      }
    }
  SB_STAMP(4);
  __syncthreads();                                                     // B3: dz1 image complete
  SB_STAMP(5);

  // ---- dz0[m][j] = sum_n dz1[m][n] W1[n][j] (.) 1[h0 > 0]: chunk c covers n = 32 c .. 32 c + 31; lane holds C[m = 16 s + li][j = ncol0 + 16 cb + 4 lq + r] ----
  f32x4 acc[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) acc[s][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  auto computeT = [&](int c) __attribute__((always_inline)) {
    const int buf = c & 1;
    if constexpr (F32) {
      const float* tf = (const float*)sW + (long)buf * SB_ROWS * SB_TF;
      const float* af = (const float*)sZ + 32 * c;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x4 fa[2];
        float fw[2][4];
#pragma unroll
        for (int s = 0; s < 2; ++s) fa[s] = *(const f32x4*)&af[(16 * s + li) * SB_ZF + 16 * t + 4 * lq];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int e = 0; e < 4; ++e) fw[cb][e] = tf[(16 * t + 4 * lq + e) * SB_TF + ncol0 + 16 * cb + li];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) acc[s][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[cb][e], fa[s][e], acc[s][cb], 0, 0, 0);
      }
    } else {
      const hx_t* th = (const hx_t*)sW + (long)buf * 2 * SB_ROWS * SB_N;
      const hx_t* tl = th + SB_ROWS * SB_N;
      const hx_t* ah = (const hx_t*)sZ + 32 * c;
      const hx_t* al = ah + SB_ROWS * SB_ZP;
      hx8 fah[2], fal[2], fwh[2], fwl[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int o = (16 * s + li) * SB_ZP + 4 * lq;               // n = 32 c + 4 lq + {0..3} and 32 c + 16 + 4 lq + {0..3}: the order of two transposing reads
        fah[s] = __builtin_shufflevector(*(const hx4*)&ah[o], *(const hx4*)&ah[o + 16], 0, 1, 2, 3, 4, 5, 6, 7);
        fal[s] = __builtin_shufflevector(*(const hx4*)&al[o], *(const hx4*)&al[o + 16], 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        fwh[cb] = sb_cat(sb_tr_sw(th, 0, ncol0 + 16 * cb, lane), sb_tr_sw(th, 16, ncol0 + 16 * cb, lane));
        fwl[cb] = sb_cat(sb_tr_sw(tl, 0, ncol0 + 16 * cb, lane), sb_tr_sw(tl, 16, ncol0 + 16 * cb, lane));
      }
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[s][cb] = ORL_MFMA_16x16x32(fwl[cb], fah[s], acc[s][cb]);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[s][cb] = ORL_MFMA_16x16x32(fwh[cb], fal[s], acc[s][cb]);
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) acc[s][cb] = ORL_MFMA_16x16x32(fwh[cb], fah[s], acc[s][cb]);
    }
  };
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if (c == 3) SB_STAMP(12);
    computeT(c);
    if (c == 3) SB_STAMP(13);
    if (c + 1 < 8) store_chunkT(c + 1);
    if (c == 3) SB_STAMP(14);
    __syncthreads();
    if (c == 3) SB_STAMP(15);
  }
  SB_STAMP(6);
  // masked dz0 (unscaled fp32) and the input tile go where the chunks were (dead: barrier above)
  const float inv_d = F32 ? 1.0f : 1.0f / (gs * ORL_WSCALE);
  float* sD0 = (float*)sW;                                             // [32][SB_DP]
  float* sX = sD0 + SB_ROWS * SB_DP;                                   // [32][SB_XP]
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int m = 16 * s + li, j = ncol0 + 16 * cb + 4 * lq;
      f32x4 d = acc[s][cb] * inv_d;
      if constexpr (F32) {
        const f32x4 h = *(const f32x4*)((const float*)sH + m * SB_HF + j);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) d[jj] = h[jj] > 0.f ? d[jj] : 0.f;
      } else {
        const unsigned int mb = *(const unsigned int*)(sM + m * SB_N + j);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) d[jj] = ((mb >> (8 * jj)) & 1u) ? d[jj] : 0.f;
      }
      *(f32x4*)&sD0[m * SB_DP + j] = d;
    }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + SB_NT * i;
    sX[(e >> 5) * SB_XP + (e & 31)] = xs[i];
  }
  SB_STAMP(7);

  // ---- dW1[n][k] = sum_m dz1[m][n] h0[m][k]: the 256 x 256 result of the 32 rows in registers (wave w owns columns k = 32 w .. 32 w + 31),
  // one MFMA per 16 x 16 block and product (K = 32 = the row group), both operands through transposing reads of row-major images ----
  {
    f32x4 wacc[16][2];
#pragma unroll
    for (int nb = 0; nb < 16; ++nb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) wacc[nb][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (F32) {
      const float* zf = (const float*)sZ;
      const float* hf = (const float*)sH;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        float b[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) b[cb] = hf[(4 * t + lq) * SB_HF + ncol0 + 16 * cb + li];
#pragma unroll
        for (int nb = 0; nb < 16; ++nb) {
          const float a = zf[(4 * t + lq) * SB_ZF + 16 * nb + li];
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) wacc[nb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[cb], wacc[nb][cb], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      const hx_t* zh = (const hx_t*)sZ;
      const hx_t* zl = zh + SB_ROWS * SB_ZP;
      const hx_t* hh = (const hx_t*)sH;
      const hx_t* hl = hh + SB_ROWS * SB_N;
      hx8 bh[2], bl[2];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        bh[cb] = sb_cat(sb_tr_sw(hh, 0, ncol0 + 16 * cb, lane), sb_tr_sw(hh, 16, ncol0 + 16 * cb, lane));
        bl[cb] = sb_cat(sb_tr_sw(hl, 0, ncol0 + 16 * cb, lane), sb_tr_sw(hl, 16, ncol0 + 16 * cb, lane));
      }
#pragma unroll
      for (int nb = 0; nb < 16; ++nb) {
        const hx8 ah = sb_cat(sb_tr_z(zh, 0, 16 * nb, lane), sb_tr_z(zh, 16, 16 * nb, lane));
        const hx8 al = sb_cat(sb_tr_z(zl, 0, 16 * nb, lane), sb_tr_z(zl, 16, 16 * nb, lane));
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) wacc[nb][cb] = ORL_MFMA_16x16x32(al, bh[cb], wacc[nb][cb]);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) wacc[nb][cb] = ORL_MFMA_16x16x32(ah, bl[cb], wacc[nb][cb]);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) wacc[nb][cb] = ORL_MFMA_16x16x32(ah, bh[cb], wacc[nb][cb]);
        __builtin_amdgcn_sched_barrier(0);                                // (unfenced, the scheduler hoists all 64 fragment reads and spills the accumulators)
      }
    }
    SB_STAMP(8);
    const float inv_w = F32 ? 1.0f : 1.0f / gs;                       // (h0 enters unscaled)
    float* __restrict__ dW1 = slab + p.off_w1;
#pragma unroll
    for (int nb = 0; nb < 16; ++nb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) dW1[(long)(16 * nb + 4 * lq + r) * SB_N + ncol0 + 16 * cb + li] = wacc[nb][cb][r] * inv_w;
  }
  SB_STAMP(9);
  __syncthreads();                                                     // B4: dz0 and the input tile visible
  SB_STAMP(10);

  // ---- dW0[j][i] = sum_m dz0[m][j] x[m][i], i <= in0 (tile column in0 = ones -> db0), on the fp32 MFMA (K = 32 rows): the lane holds
  // D[i = 16 ib + 4 lq + r][j = ncol0 + 16 cb + li] ----
  {
    f32x4 w0[2][2];
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) w0[ib][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      float a[2], b[2];
#pragma unroll
      for (int ib = 0; ib < 2; ++ib) a[ib] = sX[(4 * t + lq) * SB_XP + 16 * ib + li];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) b[cb] = sD0[(4 * t + lq) * SB_DP + ncol0 + 16 * cb + li];
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) w0[ib][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ib], b[cb], w0[ib][cb], 0, 0, 0);
    }
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * ib + 4 * lq + r, j = ncol0 + 16 * cb + li;
          if (i < p.in0) slab[p.off_w0 + (long)j * p.in0 + i] = w0[ib][cb][r];
          else if (i == p.in0) slab[p.off_b0 + j] = w0[ib][cb][r];
        }
  }
  SB_STAMP(11);
  // ---- loss sums of this row group -> partials; the workgroup that arrives last finishes the run (k_actor_loss's arithmetic, group order).
  // Last in the kernel: the atomics and the scalar Adam step are a chain of dependent latencies that only lane 0 walks ----
  if (tid == 0) {
    const int groups = gridDim.x;
    float* o = p.part + ((long)z0 * SB_MAXGROUPS + g) * 2;
    __hip_atomic_store(o, s_loss, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(o + 1, s_lp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned int t = __hip_atomic_fetch_add(p.ticket + z0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == (unsigned int)groups - 1u) {
      __hip_atomic_store(p.ticket + z0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      float tl = 0.f, tp = 0.f;
      const float* q = p.part + (long)z0 * SB_MAXGROUPS * 2;
      for (int k = 0; k < groups; ++k) {
        tl += __hip_atomic_load(q + 2 * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tp += __hip_atomic_load(q + 2 * k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      float* ml = p.metrics_last + (long)z0 * p.nm;
      float* ms = p.metrics_sum + (long)z0 * p.nm;
      const float loss = tl / (float)M;
      ml[p.m_actor] = loss; ms[p.m_actor] += loss;
      sc.alpha_bwd = alpha;
      if (p.auto_alpha) {
        // alpha_loss = -(log_alpha * (logp + target_entropy)).mean(): every workgroup of the run has read alpha before it took its ticket
        const float mean_t = tp / (float)M + p.target_entropy;
        const float aloss = -(sc.log_alpha * mean_t);
        adam_scalar(sc.log_alpha, sc.la_m, sc.la_v, -mean_t, p.hy->lr[2], p.b1, p.b2, p.adam_eps, *p.gstep + 1ull);
        float na = expf(sc.log_alpha);
        if (p.clamp_alpha01) na = fminf(fmaxf(na, 0.f), 1.f);
        sc.alpha = na;
        ml[p.m_alpha_loss] = aloss; ms[p.m_alpha_loss] += aloss;
        ml[p.m_alpha] = na; ms[p.m_alpha] += na;
      }
    }
  }
}

hipError_t launch_small_abwd(const SmallABwdP& p, int runs, hipStream_t st) {
  static const hipError_t attr_err = [] {
    hipError_t e = hipFuncSetAttribute((const void*)small_abwd_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sb_lds_bytes<false>());
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)small_abwd_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sb_lds_bytes<true>());
    return e;
  }();
  if (attr_err != hipSuccess) return attr_err;
  const dim3 grid(p.M / SB_ROWS, runs), block(SB_NT);
  if (p.f32) hipLaunchKernelGGL(small_abwd_kernel<true>, grid, block, sb_lds_bytes<true>(), st, p);
  else hipLaunchKernelGGL(small_abwd_kernel<false>, grid, block, sb_lds_bytes<false>(), st, p);
  return hipGetLastError();
}

}  // namespace orl
