// ws_wgrad_experiment.h — output-stationary weight-gradient kernel for the top hidden layer (ds_read_b64_tr_b16 operands).
// EXPERIMENT, not part of the product: numerically correct (tools/ws_lab3.hip) but at 165 us for the 16-run CQL shape it does not
// beat the generic 256x128-tile wgrad, because with one 8-wave workgroup per CU its staging and MFMA phases do not overlap.
#pragma once
#include "../offlinerl-kit_amd/csrc/ws_gemm.h"
namespace orl {
// =====================================================================================================================
// ws_wgrad: weight gradient of the top hidden layer of a single-output net, output-stationary, plus the tail gradients.
//
//   dW1[k][n] = w_tail[k] * sum_m 1[h1[m][k] > 0] * (dq[m] * h0[m][n])        db1[k] = w_tail[k] * sum_m 1[h1[m][k] > 0] * dq[m]
//   dw_tail[k] = sum_m dq[m] * h1[m][k]                                         db_tail = sum_m dq[m]
//
// The 256 x 256 result stays in registers for the whole launch: wave w owns columns n in [32w, 32w+32) and all 256 rows k
// (16 x 2 blocks of 16 x 16 = 128 accumulator VGPRs) and the workgroup streams 32-row groups of the batch.  Both MFMA
// operands are "transposed" views of row-major data (the reduction runs over the rows m), which is exactly what
// ds_read_b64_tr_b16 delivers from row-major LDS images: A = the 0/1 ReLU mask of h1 expanded from its packed bits (exact in
// bf16, no lo plane), B = G = dq (.) h0 split into bf16 hi/lo while it is staged.  v_mfma_f32_16x16x16_bf16, two per block
// pair.  h1 itself is only streamed through registers for dw_tail.  One split-K slab per workgroup.
// =====================================================================================================================
struct WsWgradP {
  const unsigned int* abits; long ab_s0, ab_s1; int ab_g;     // mask words of the top hidden activation h1
  const float* dq; long dq_s0, dq_s1, dq_sm;
  const float* H0; long h0_s0, h0_s1; int h0_pitch;            // input of the top hidden layer [M][256]
  const float* H1; long h1_s0, h1_s1; int h1_pitch;            // top hidden activation [M][256] (tail gradient only)
  const float* wt; long wt_s0, wt_s1;                          // w_tail [256]
  float *dW, *db, *dwt, *dbt;                                  // slab outputs; run stride o_s0, member strides below, slab stride o_ks
  long o_s0, o_s1w, o_s1b, o_s1wt, o_s1bt, o_ks;
  int M, nz1, groups;
};
enum { WW_IMG = WS_ROWS * WS_K };                               // bf16 elements of one [32][256] LDS image
static constexpr size_t ws_wgrad_lds_bytes() { return (size_t)2 * 3 * WW_IMG * 2 + (size_t)2 * 2 * WS_ROWS * 16 * 2; }   // 2 buffers x {mask, G hi, G lo} + dq blocks

// bf16 offset of the 8-byte piece (16-byte chunk `chunk`, half `half`) of row r: chunks are XOR-swizzled with 2 (r & 7) so that
// the transposed reads (8 rows x 32 B per 32-lane half) and the 8-byte staging stores are both bank-conflict free
__device__ inline int ww_off(int r, int chunk, int half) { return r * WS_K + ((chunk ^ (2 * (r & 7))) << 3) + (half << 2); }

__device__ inline s16x4 ww_tr(const __bf16* img, int row0, int col0, int lane) {
  // 16-lane group lq reads rows row0 + 4 lq + q (q = li >> 2), columns col0 + 4 (li & 3) ..; lane li receives column col0 + li of
  // rows row0 + 4 lq .. + 3  (= the 16x16x16 MFMA operand layout, for A as the transpose of the image)
  const int li = lane & 15, lq = lane >> 4, row = row0 + 4 * lq + (li >> 2), col = col0 + 4 * (li & 3);
  const __bf16* a = img + ww_off(row, col >> 3, (col >> 2) & 1);
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)a);
}

__global__ __launch_bounds__(WS_NT) void ws_wgrad_kernel(const WsWgradP p) {
  static_assert(WS_NW == 8 && WS_ROWS == 32, "8 waves x 32 columns, 32-row groups");
  extern __shared__ __attribute__((aligned(16))) float ws_smem[];
  __bf16* img = (__bf16*)ws_smem;                                   // [buf][{mask, G hi, G lo}][32][256]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lq = lane >> 4;
  const int z = blockIdx.z, z0 = z / p.nz1, z1 = z - z0 * p.nz1;
  const unsigned int* __restrict__ ab = p.abits + z0 * p.ab_s0 + z1 * p.ab_s1;
  const float* __restrict__ dqg = p.dq + z0 * p.dq_s0 + z1 * p.dq_s1;
  const float* __restrict__ H0g = p.H0 + z0 * p.h0_s0 + z1 * p.h0_s1;
  const float* __restrict__ H1g = p.H1 + z0 * p.h1_s0 + z1 * p.h1_s1;
  const float* __restrict__ wtg = p.wt + z0 * p.wt_s0 + z1 * p.wt_s1;
  const int ncol0 = 32 * wave;

  f32x4 acc[16][2], accb[2];
#pragma unroll
  for (int kb = 0; kb < 16; ++kb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) acc[kb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  accb[0] = accb[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float dqsum = 0.f;

  // ---- staging registers of one row group ----
  f32x4 s0[4];
  float sdq[4];
  unsigned int sm_word;
  __bf16* dqimg = img + 2 * 3 * WW_IMG;                              // [buf][hi, lo][32 rows][16]: column 0 = dq, others 0 (db1 operand)
  auto load_group = [&](int g) __attribute__((always_inline)) {
    sm_word = ab[(long)(g * WS_ROWS + (tid >> 4)) * p.ab_g + ((tid & 15) >> 1)];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + WS_NT * i, r = idx >> 6, kq = idx & 63;
      const long m = (long)g * WS_ROWS + r;
      s0[i] = *(const f32x4*)&H0g[m * p.h0_pitch + 4 * kq];
      sdq[i] = dqg[m * p.dq_sm];
    }
  };
  auto load_piece = [&](int g, int i) __attribute__((always_inline)) {
    const int idx = tid + WS_NT * i, r = idx >> 6, kq = idx & 63;
    const long m = (long)g * WS_ROWS + r;
    s0[i] = *(const f32x4*)&H0g[m * p.h0_pitch + 4 * kq];
    sdq[i] = dqg[m * p.dq_sm];
  };
  auto load_mask = [&](int g) __attribute__((always_inline)) {
    sm_word = ab[(long)(g * WS_ROWS + (tid >> 4)) * p.ab_g + ((tid & 15) >> 1)];
  };
  auto store_mask = [&](int buf) __attribute__((always_inline)) {
    __bf16* mi = img + (long)buf * 3 * WW_IMG;
    const int r = tid >> 4, hw = tid & 15;
    const unsigned int bits = (sm_word >> (16 * (hw & 1))) & 0xFFFFu;
    u32x4 c0, c1;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned int y0 = (bits >> (2 * j)) & 3u, y1 = (bits >> (8 + 2 * j)) & 3u;
      c0[j] = ((y0 & 1u) | ((y0 >> 1) << 16)) * 0x3F80u;
      c1[j] = ((y1 & 1u) | ((y1 >> 1) << 16)) * 0x3F80u;
    }
    *(u32x4*)(mi + ww_off(r, 2 * hw, 0)) = c0;
    *(u32x4*)(mi + ww_off(r, 2 * hw + 1, 0)) = c1;
  };
  auto store_piece = [&](int buf, int i) __attribute__((always_inline)) {
    __bf16* gh = img + (long)buf * 3 * WW_IMG + WW_IMG;
    __bf16* gl = gh + WW_IMG;
    const int idx = tid + WS_NT * i, r = idx >> 6, kq = idx & 63;
    bf16x4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gv = s0[i][j] * sdq[i];
      const __bf16 hh = (__bf16)gv; h[j] = hh; l[j] = (__bf16)(gv - (float)hh);
    }
    const int o = ww_off(r, kq >> 1, kq & 1);
    *(bf16x4*)(gh + o) = h;
    *(bf16x4*)(gl + o) = l;
    if (kq == 0) {                                                   // this row's dq into the bias-gradient operand block
      const __bf16 hh = (__bf16)sdq[i];
      __bf16* dqi = dqimg + (long)buf * 2 * WS_ROWS * 16;
      dqi[r * 16] = hh; dqi[WS_ROWS * 16 + r * 16] = (__bf16)(sdq[i] - (float)hh);
      dqsum += sdq[i];
    }
  };
  auto store_group = [&](int buf) __attribute__((always_inline)) {
    store_mask(buf);
#pragma unroll
    for (int i = 0; i < 4; ++i) store_piece(buf, i);
  };
  for (int e = tid; e < 2 * 2 * WS_ROWS * 16 / 2; e += WS_NT) ((unsigned int*)dqimg)[e] = 0u;   // columns 1..15 stay zero
  __syncthreads();

  const int g0 = blockIdx.x, gs = gridDim.x;
  if (g0 < p.groups) {
    load_group(g0);
    store_group(0);
    if (g0 + gs < p.groups) load_group(g0 + gs);
  }
  __syncthreads();
  int it = 0;
  for (int g = g0; g < p.groups; g += gs, ++it) {
    const int buf = it & 1;
    const bool more = g + gs < p.groups, more2 = g + 2 * gs < p.groups;
    const __bf16* mi = img + (long)buf * 3 * WW_IMG;
    const __bf16* gh = mi + WW_IMG;
    const __bf16* gl = gh + WW_IMG;
    const __bf16* dqi = dqimg + (long)buf * 2 * WS_ROWS * 16;
#pragma unroll
    for (int st = 0; st < 2; ++st) {                                 // two reduction steps of 16 rows
      const int m0 = 16 * st;
      s16x4 bh[2], bl[2];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) { bh[nb] = ww_tr(gh, m0, ncol0 + 16 * nb, lane); bl[nb] = ww_tr(gl, m0, ncol0 + 16 * nb, lane); }
      // B operand of the bias gradient: the [32][16] dq block (column 0 = dq), rows m0 + 4 lq + q, columns 4 (li & 3) ..
      const int dro = (m0 + 4 * lq + (li >> 2)) * 16 + 4 * (li & 3);
      const s16x4 bdh = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(dqi + dro));
      const s16x4 bdl = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(dqi + WS_ROWS * 16 + dro));
#pragma unroll
      for (int kb = 0; kb < 16; ++kb) {
#ifdef WW_LAB_NO_AREAD
        const s16x4 a = bh[kb & 1];
#else
        const s16x4 a = ww_tr(mi, m0, 16 * kb, lane);                // A[i = k][kk = m] = mask[m][k]
#endif
#ifdef WW_LAB_NO_MFMA
        asm volatile("" :: "v"(a), "v"(bl[0]), "v"(bh[0]), "v"(bl[1]), "v"(bh[1]));
#else
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[kb][nb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, bl[nb], acc[kb][nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[kb][nb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, bh[nb], acc[kb][nb], 0, 0, 0);
#endif
        if ((kb >> 1) == wave) {                                     // uniform per wave: this wave's share of db1
          accb[kb & 1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, bdl, accb[kb & 1], 0, 0, 0);
          accb[kb & 1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, bdh, accb[kb & 1], 0, 0, 0);
        }
        // the next group's staging (VALU + LDS stores into the other buffer) is spread between the MFMAs
        // each staging register is written to LDS and refilled at the same point of every iteration: a full iteration in flight
        if (st == 0 && (kb & 3) == 3) {
          if (more) store_piece(buf ^ 1, kb >> 2);
          if (more2) load_piece(g + 2 * gs, kb >> 2);
        }
        if (st == 1 && kb == 3) {
          if (more) store_mask(buf ^ 1);
          if (more2) load_mask(g + 2 * gs);
        }
      }
    }
    __syncthreads();
  }

  // ---- one slab per workgroup ----
  const long so = z0 * p.o_s0 + (long)blockIdx.x * p.o_ks;
  float* dW = p.dW + so + z1 * p.o_s1w;
  float* db = p.db + so + z1 * p.o_s1b;
#pragma unroll
  for (int kb = 0; kb < 16; ++kb) {
    const f32x4 w4 = *(const f32x4*)&wtg[16 * kb + 4 * lq];           // lane holds rows k = 16 kb + 4 lq + r, column n = ncol0 + 16 nb + li
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) dW[(long)(16 * kb + 4 * lq + r) * WS_N + ncol0 + 16 * nb + li] = w4[r] * acc[kb][nb][r];
  }
  if (li == 0) {
#pragma unroll
    for (int x = 0; x < 2; ++x) {
      const int k0 = 16 * (2 * wave + x) + 4 * lq;
#pragma unroll
      for (int r = 0; r < 4; ++r) db[k0 + r] = wtg[k0 + r] * accb[x][r];
    }
  }
  // db_tail = sum of dq: eight row-slice partial sums, fixed order (dw_tail comes from k_tail_wgrad)
  float* red = ws_smem;                                              // the images are dead after the loop's last barrier
  if ((tid & 63) == 0) red[tid >> 6] = dqsum;
  __syncthreads();
  if (tid == 0) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) a += red[w];
    p.dbt[so + z1 * p.o_s1bt] = a;
  }
}

static inline bool ws_wgrad_supported(const WsWgradP& p, int K, int N) {
  if (K != WS_K || N != WS_N || p.M < 1024 || (p.M % WS_ROWS) || !p.abits || p.ab_g != 8) return false;
  if (!aligned16(p.H0) || (p.h0_pitch & 3) || (p.h0_s0 & 3) || (p.h0_s1 & 3)) return false;
  if (!aligned16(p.H1) || (p.h1_pitch & 3) || (p.h1_s0 & 3) || (p.h1_s1 & 3)) return false;
  return aligned16(p.wt) && !(p.wt_s0 & 3) && !(p.wt_s1 & 3);
}
static inline hipError_t launch_ws_wgrad(WsWgradP p, int nz, int per_z, hipStream_t st) {
  p.groups = p.M / WS_ROWS;
  static bool raised = false;
  if (!raised) {
    hipError_t e = hipFuncSetAttribute((const void*)ws_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ws_wgrad_lds_bytes());
    if (e != hipSuccess) return e;
    raised = true;
  }
  hipLaunchKernelGGL(ws_wgrad_kernel, dim3(per_z, 1, nz), dim3(WS_NT), ws_wgrad_lds_bytes(), st, p);
  return hipGetLastError();
}

}  // namespace orl
