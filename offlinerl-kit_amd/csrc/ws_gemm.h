// ws_gemm.h — weight-stationary, row-streaming kernels for the 256-wide critic layers of a many-row batch.
//
// The generic tile kernel (gemm.h) spends about half of a workgroup's lifetime in its prologue / epilogue when K is
// only 256 (eight K chunks per tile) and re-reads the weight tile for every row tile.  For the hot shapes of the
// update engine (hidden layers of width 256 evaluated on thousands of rows: CQL's 7936-row critic batch, reference
// cql.py:132-190) the whole weight matrix fits in ONE CU's register file once it is split into bf16 hi/lo planes
// (256 x 256 x 4 B = 256 KB of the 512 KB VGPR file).  So:
//
//   * a workgroup = 8 waves; wave w owns output columns [32w, 32w+32) and keeps the B fragments of those columns for
//     the complete K = 256 in registers (2 column blocks x 8 k-steps x {hi, lo} x 4 VGPRs = 128 VGPRs), loaded and
//     split ONCE per workgroup;
//   * the workgroup then streams row groups of 64 rows: the fp32 rows are fetched with full-row coalesced loads,
//     split into bf16 hi/lo while they are staged into a double-buffered LDS image, and every wave multiplies the
//     shared A fragments against its resident B fragments (v_mfma_f32_16x16x32_bf16, lo*hi + hi*lo + hi*hi, fp32
//     accumulation) -- one barrier per 64 rows, no per-tile pipeline fill / drain, no weight traffic in the loop;
//   * the epilogue works on the wave's own 16 x 32 accumulator blocks (bias, ReLU, packed ReLU-mask bits, the fused
//     single-output tail q = h . w_tail + b_tail), so it needs no LDS round trip.
//
// Per row group and CU: 64 KB of HBM reads against 2 x 192 MFMAs per SIMD (6144 cycles), i.e. the kernel sits at the
// crossover of the HBM and matrix-pipe rooflines instead of far below both.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>

#include "gemm.h"

namespace orl {

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
  int M, nz1, groups;                                   // groups = ceil(M / WS_ROWS)
  // fused first layer (template L0): X is then PRODUCED here as relu(X0 W0^T + b0) from the narrow input rows X0 (in0 + 1 <= 32 columns
  // incl. the bias as a ones column), stored to `X` for the backward pass, and handed to the second layer through LDS only
  const float* X0; long x0_s0, x0_s1; int x0_pitch, in0;
  const float* W0; long w0_s0, w0_s1, w0_sn, w0_sk;     // element (n, k) at W0[n * w0_sn + k * w0_sk] ((256, in0) row-major: in0, 1)
  const float* b0; long b0_s0, b0_s1;
  unsigned int* mb0; long mb0_s0, mb0_s1; int mb0_g;    // packed ReLU mask of X (= h0)
  // plain dgrad mode (template DG): Y = (X B^T) (.) mask, B given by the strides above (W viewed transposed), no bias / ReLU / mask
  // emission; `dmask` = packed ReLU mask of the activation the gradient flows into
  const unsigned int* dmask; long dm_s0, dm_s1; int dm_g;
};

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

__device__ inline void ws_split8(const f32x4& a, const f32x4& b, bf16x8& h, bf16x8& l) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const __bf16 ha = (__bf16)a[j]; h[j] = ha; l[j] = (__bf16)(a[j] - (float)ha);
    const __bf16 hb = (__bf16)b[j]; h[4 + j] = hb; l[4 + j] = (__bf16)(b[j] - (float)hb);
  }
}

template <bool TQ, bool L0, bool DG = false, bool SY = true>      // SY = false: the activation itself is not stored (TQ only)
__global__ __launch_bounds__(WS_NT) void ws_fwd_kernel(const WsFwdP p) {
  extern __shared__ __attribute__((aligned(16))) float ws_smem[];
  __bf16* Ah = (__bf16*)ws_smem;                                   // [buf][plane][row][WS_PITCH]
  float* qs = ws_smem + (2 * 2 * WS_ROWS * WS_PITCH * 2) / 4;       // [parity][wave][row]
  unsigned char* nbs = (unsigned char*)(qs + 2 * WS_NW * WS_ROWS);       // [parity][row][64]: 4 mask bits per (row, 4 columns)
  float* Xl = (float*)(nbs + 2 * WS_ROWS * WS_NBP);                          // L0: [buf][row][32] narrow input rows (fp32, ones column at in0)
  unsigned char* nbs0 = (unsigned char*)(Xl + 2 * WS_ROWS * WS_XLP);         // L0: mask nibbles of the produced h0
  float* cst = (float*)((char*)ws_smem + ws_fwd_lds_bytes(L0) - sizeof(float) * 2 * WS_N);   // [bias | tail weights]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lq = lane >> 4;
  const int z = blockIdx.z, z0 = z / p.nz1, z1 = z - z0 * p.nz1;
  const float* __restrict__ Xg = p.X + z0 * p.x_s0 + z1 * p.x_s1;
  const float* __restrict__ Wg = p.W + z0 * p.w_s0 + z1 * p.w_s1;
  const float* __restrict__ bg = DG ? nullptr : p.bias + z0 * p.b_s0 + z1 * p.b_s1;
  float* __restrict__ Yg = p.Y + z0 * p.y_s0 + z1 * p.y_s1;
  const int ncol0 = 16 * WS_CB * wave;
  float* __restrict__ Y0g = L0 ? const_cast<float*>(Xg) : nullptr;       // L0: h0 is written where the plain kernel reads it
  const float* __restrict__ X0g = L0 ? p.X0 + z0 * p.x0_s0 + z1 * p.x0_s1 : nullptr;

  // ---- resident B fragments: lane (li, lq) supplies W[n = ncol0 + 16 cb + li][k = 32 ks + 8 lq .. +7] ----
  bf16x8 bh[WS_CB][8], bl[WS_CB][8];
#pragma unroll
  for (int cb = 0; cb < WS_CB; ++cb)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const float* src = Wg + (long)(ncol0 + 16 * cb + li) * p.w_sn + (long)(32 * ks + 8 * lq) * p.w_sk;
      if (p.w_sk == 1) ws_split8(*(const f32x4*)src, *(const f32x4*)(src + 4), bh[cb][ks], bl[cb][ks]);
      else {                                           // (in, out)-major weights: eight strided loads, once per workgroup
        f32x4 a, b;
#pragma unroll
        for (int j = 0; j < 4; ++j) { a[j] = src[(long)j * p.w_sk]; b[j] = src[(long)(4 + j) * p.w_sk]; }
        ws_split8(a, b, bh[cb][ks], bl[cb][ks]);
      }
    }
  // L0: first-layer fragments of the same columns, K = 32: W0'[n][k] = W0[n][k] (k < in0), b0[n] (k == in0), 0 beyond
  bf16x8 b0h[WS_CB], b0l[WS_CB];
  if (L0) {
    const float* __restrict__ W0g = p.W0 + z0 * p.w0_s0 + z1 * p.w0_s1;
    const float* __restrict__ b0g = p.b0 + z0 * p.b0_s0 + z1 * p.b0_s1;
#pragma unroll
    for (int cb = 0; cb < WS_CB; ++cb) {
      const int n = ncol0 + 16 * cb + li;
      f32x4 a, b;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k0 = 8 * lq + j, k1 = k0 + 4;
        a[j] = k0 < p.in0 ? W0g[(long)n * p.w0_sn + (long)k0 * p.w0_sk] : (k0 == p.in0 ? b0g[n] : 0.f);
        b[j] = k1 < p.in0 ? W0g[(long)n * p.w0_sn + (long)k1 * p.w0_sk] : (k1 == p.in0 ? b0g[n] : 0.f);
      }
      ws_split8(a, b, b0h[cb], b0l[cb]);
    }
  }
  // epilogue constants of this lane's columns n = ncol0 + 16 cb + 4 lq + r sit in LDS (not in 16 VGPRs next to the 128 VGPRs of
  // resident B fragments, and not re-read from global memory: vmcnt is in-order, so waiting for such a load in the epilogue would
  // also wait for every activation store issued before it)
  const float* __restrict__ twg = TQ ? p.tw + z0 * p.tw_s0 + z1 * p.tw_s1 : bg;
  if (!DG && tid < WS_N) { cst[tid] = bg[tid]; cst[WS_N + tid] = twg[tid]; }      // visible after the prologue's barriers
  const float tbias = TQ ? (p.tb + z0 * p.tb_s0 + z1 * p.tb_s1)[0] : 0.f;

  // ---- staging of one row group: thread t moves float4 #(t + 512 i), i = 0..7, of the [64][256] tile ----
  // one staging register set: refilled with group g + 2 gs right after group g + gs has been written to LDS
  f32x4 st0[WS_LD];
  auto load_group = [&](int g, f32x4 (&st)[WS_LD]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < WS_LD; ++i) {
      const int idx = tid + WS_NT * i, r = idx >> 6, kq = idx & 63;
      int m = g * WS_ROWS + r; m = m < p.M ? m : p.M - 1;
      st[i] = *(const f32x4*)&Xg[(long)m * p.x_pitch + 4 * kq];
    }
  };
  auto load_piece = [&](int g, f32x4 (&st)[WS_LD], int i) __attribute__((always_inline)) {
    const int idx = tid + WS_NT * i, r = idx >> 6, kq = idx & 63;
    int m = g * WS_ROWS + r; m = m < p.M ? m : p.M - 1;
    st[i] = *(const f32x4*)&Xg[(long)m * p.x_pitch + 4 * kq];
  };
  auto store_group = [&](int buf, const f32x4 (&st)[WS_LD], int i0 = 0, int i1 = WS_LD) __attribute__((always_inline)) {
    __bf16* dh = Ah + (long)buf * 2 * WS_ROWS * WS_PITCH;
    __bf16* dl = dh + WS_ROWS * WS_PITCH;
#pragma unroll
    for (int i = i0; i < i1; ++i) {
      const int idx = tid + WS_NT * i, r = idx >> 6, kq = idx & 63;
      bf16x4 h, l;
#pragma unroll
      for (int j = 0; j < 4; ++j) { const __bf16 hh = (__bf16)st[i][j]; h[j] = hh; l[j] = (__bf16)(st[i][j] - (float)hh); }
      // 16-byte chunk c = k / 8 of row r lives at chunk c ^ (r & 15): ds_read_b128 of a fragment column is then conflict-free
      // for the hardware's 16-lane groups (which mix lanes of two neighbouring chunks), and these 8-byte stores stay so too
      const int o = r * WS_PITCH + ((((kq >> 1) ^ (r & 15)) << 3) | ((kq & 1) << 2));
      *(bf16x4*)(dh + o) = h;
      *(bf16x4*)(dl + o) = l;
    }
  };

  // ---- L0: narrow-input staging (two elements per thread) and the producer of one h0 row group ----
  const int xe = L0 ? WS_ROWS * p.x0_pitch : 0;
  int xr[2], xc[2];
  float sx[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + WS_NT * i;
    xr[i] = L0 ? e / (L0 ? p.x0_pitch : 1) : 0; xc[i] = L0 ? e - xr[i] * p.x0_pitch : 0;
    if (L0 && e >= xe) { xr[i] = 0; xc[i] = 32; }       // never read (rows are consumed as 32 columns of the 36-float pitch)
  }
  auto loadX = [&](int g) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int e = tid + WS_NT * i; sx[i] = X0g[(long)g * xe + (e < xe ? e : xe - 1)]; }   // clamped, not predicated
  };
  auto storeX = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) Xl[(buf * WS_ROWS + xr[i]) * WS_XLP + xc[i]] = (xc[i] == p.in0) ? 1.0f : sx[i];   // surplus threads: pad column 32
  };
  // produce(g): h0 rows of group g for this wave's columns -> global (fp32), the LDS image `buf` (split bf16), mask nibbles
  auto prod_x = [&](int xbuf, int s, bf16x8& xah, bf16x8& xal) __attribute__((always_inline)) {
    const float* xrow = Xl + (xbuf * WS_ROWS + 16 * s + li) * WS_XLP + 8 * lq;
    ws_split8(*(const f32x4*)xrow, *(const f32x4*)(xrow + 4), xah, xal);
  };
  auto prod_block = [&](int g, int buf, int par, int s, int cb, const bf16x8& xah, const bf16x8& xal) __attribute__((always_inline)) {
    __bf16* dh = Ah + (long)buf * 2 * WS_ROWS * WS_PITCH;
    __bf16* dl = dh + WS_ROWS * WS_PITCH;
    const int r = 16 * s + li;
    const long m = (long)g * WS_ROWS + r;
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
    v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0l[cb], xah, v, 0, 0, 0);
    v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0h[cb], xal, v, 0, 0, 0);
    v = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0h[cb], xah, v, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
    const int k = ncol0 + 16 * cb + 4 * lq;                      // h0 columns k .. k + 3 of row r (lane holds C[m = li][n = 4 lq + j])
    *(f32x4*)&Y0g[m * p.x_pitch + k] = v;
    bf16x4 h, l;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const __bf16 hh = (__bf16)v[j]; h[j] = hh; l[j] = (__bf16)(v[j] - (float)hh); }
    const int o = r * WS_PITCH + ((((k >> 3) ^ (r & 15)) << 3) | (((k >> 2) & 1) << 2));
    *(bf16x4*)(dh + o) = h;
    *(bf16x4*)(dl + o) = l;
    nbs0[(par * WS_ROWS + r) * WS_NBP + (k >> 2)] =
        (unsigned char)((v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u));
  };
  auto produce = [&](int g, int buf, int xbuf, int par) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < WS_SUB; ++s) {
      bf16x8 xah, xal;
      prod_x(xbuf, s, xah, xal);
#pragma unroll
      for (int cb = 0; cb < WS_CB; ++cb) prod_block(g, buf, par, s, cb, xah, xal);
    }
  };
  auto finish0 = [&](int g, int par) __attribute__((always_inline)) {  // after the barrier that follows produce(g): pack the h0 mask
    if (tid < WS_ROWS * 8) {
      const int row = tid >> 3, wd = tid & 7;
      const unsigned int* nb = (const unsigned int*)(nbs0 + (par * WS_ROWS + row) * WS_NBP + 8 * wd);
      const unsigned int d0 = nb[0], d1 = nb[1];
      const unsigned int lo16 = (d0 & 0xFu) | ((d0 >> 4) & 0xF0u) | ((d0 >> 8) & 0xF00u) | ((d0 >> 12) & 0xF000u);
      const unsigned int hi16 = (d1 & 0xFu) | ((d1 >> 4) & 0xF0u) | ((d1 >> 8) & 0xF00u) | ((d1 >> 12) & 0xF000u);
      p.mb0[z0 * p.mb0_s0 + z1 * p.mb0_s1 + ((long)g * WS_ROWS + row) * p.mb0_g + wd] = lo16 | (hi16 << 16);
    }
  };

  const int g0 = blockIdx.x, gs = gridDim.x;
  if (g0 >= p.groups) return;
  if (L0) {
    for (int e = tid; e < 2 * WS_ROWS * WS_XLP; e += WS_NT) Xl[e] = 0.f;   // columns >= x0_pitch stay zero
    loadX(g0);
    __syncthreads();
    storeX(0);
    if (g0 + gs < p.groups) loadX(g0 + gs);
    __syncthreads();
    produce(g0, 0, 0, 0);
    if (g0 + gs < p.groups) storeX(1);
    if (g0 + 2 * gs < p.groups) loadX(g0 + 2 * gs);
    __syncthreads();
    finish0(g0, 0);
  } else {
    load_group(g0, st0);
    store_group(0, st0);
    if (g0 + gs < p.groups) load_group(g0 + gs, st0);
    __syncthreads();
  }
  // Software pipeline: iteration `it` multiplies group g out of LDS buffer it & 1 while the epilogue of the PREVIOUS group
  // (accumulators `pacc`) runs in the shadow of those MFMAs -- both are in one basic block (no row guards: M is a multiple of
  // WS_ROWS), so the scheduler can pair every MFMA with the VALU / store work of the other stage.  Then group g + gs is staged
  // from register set (it + 1) & 1.
  constexpr bool storeY = SY;                      // false: a single-output net whose backward needs only the mask bits of this activation
  // one 16 x 16 block of the (non-gradient) epilogue: bias, ReLU, optional store, tail partial sum, 4 mask bits -> LDS
  auto epi_block = [&](const f32x4& a, int g, int par, int s, int cb, float& part) __attribute__((always_inline)) {
    const f32x4 bq = *(const f32x4*)&cst[ncol0 + 16 * cb + 4 * lq], twq = *(const f32x4*)&cst[WS_N + ncol0 + 16 * cb + 4 * lq];
    const int m = g * WS_ROWS + 16 * s + li;
    f32x4 v = a + bq;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
#ifndef WS_LAB_NO_STORE
    if (storeY) *(f32x4*)&Yg[(long)m * p.y_pitch + ncol0 + 16 * cb + 4 * lq] = v;
#else
    if (v[0] == 12345.678f) *(f32x4*)&Yg[(long)m * p.y_pitch + ncol0 + 16 * cb + 4 * lq] = v;
#endif
    part += (v[0] * twq[0] + v[1] * twq[1]) + (v[2] * twq[2] + v[3] * twq[3]);
    // 4 mask bits of (row 16 s + li, columns ncol0 + 16 cb + 4 lq ..) -> LDS, packed into words after the barrier
    nbs[(par * WS_ROWS + 16 * s + li) * WS_NBP + 4 * WS_CB * wave + 4 * cb + lq] =
        (unsigned char)((v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u));
  };
  auto epi_row = [&](int par, int s, float part) __attribute__((always_inline)) {
    if (TQ) {
      part += __shfl_xor(part, 16);
      part += __shfl_xor(part, 32);
      (qs + (par * WS_NW + wave) * WS_ROWS)[16 * s + li] = part;   // all four lq lanes hold the same sum: no divergent branch in this block
    }
  };
  auto epilogue = [&](const f32x4 (&acc)[WS_SUB][WS_CB], int g, int par) __attribute__((always_inline)) {
    if (DG) {                                      // gradient epilogue: ReLU mask of the receiving activation from its packed bits
      static_assert(!DG || WS_CB == 2, "one 32-column mask word per wave");
      const unsigned int* __restrict__ dm = p.dmask + z0 * p.dm_s0 + z1 * p.dm_s1;
#pragma unroll
      for (int s = 0; s < WS_SUB; ++s) {
        const long m = (long)g * WS_ROWS + 16 * s + li;
        const unsigned int w = dm[m * p.dm_g + wave];
#pragma unroll
        for (int cb = 0; cb < WS_CB; ++cb) {
          const unsigned int nib = w >> (16 * cb + 4 * lq);
          f32x4 v = acc[s][cb];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = ((nib >> r) & 1u) ? v[r] : 0.f;
          *(f32x4*)&Yg[m * p.y_pitch + ncol0 + 16 * cb + 4 * lq] = v;
        }
      }
      return;
    }
#pragma unroll
    for (int s = 0; s < WS_SUB; ++s) {
      float part = 0.f;
#pragma unroll
      for (int cb = 0; cb < WS_CB; ++cb) epi_block(acc[s][cb], g, par, s, cb, part);
      epi_row(par, s, part);
    }
  };
  auto finish = [&](int g, int par) __attribute__((always_inline)) {   // after the barrier that follows epilogue(g)
    if (DG) return;
    if (tid < WS_ROWS * 8) {                         // thread (row, word): eight nibbles -> one 32-column mask word
      const int row = tid >> 3, wd = tid & 7, m = g * WS_ROWS + row;
      const unsigned int* nb = (const unsigned int*)(nbs + (par * WS_ROWS + row) * WS_NBP + 8 * wd);
      const unsigned int d0 = nb[0], d1 = nb[1];
      const unsigned int lo16 = (d0 & 0xFu) | ((d0 >> 4) & 0xF0u) | ((d0 >> 8) & 0xF00u) | ((d0 >> 12) & 0xF000u);
      const unsigned int hi16 = (d1 & 0xFu) | ((d1 >> 4) & 0xF0u) | ((d1 >> 8) & 0xF00u) | ((d1 >> 12) & 0xF000u);
      p.mb[z0 * p.mb_s0 + z1 * p.mb_s1 + (long)m * p.mb_g + wd] = lo16 | (hi16 << 16);
    }
    if (TQ && tid < WS_ROWS) {                       // eight column-slice partial sums per row, fixed order
      const int m = g * WS_ROWS + tid;
      const float* q8 = qs + par * WS_NW * WS_ROWS + tid;
      float a = tbias;
#pragma unroll
      for (int w = 0; w < WS_NW; ++w) a += q8[w * WS_ROWS];
      p.tq[z0 * p.tq_s0 + z1 * p.tq_s1 + (long)m * p.tq_sm] = a;
    }
  };

  f32x4 pacc[WS_SUB][WS_CB];
  // steady = true: groups g + gs .. g + 3 gs exist, so the body has no conditionals (one basic block up to the barrier)
  auto iteration = [&](int g, int it, f32x4 (&stn)[WS_LD], bool first, bool steady) __attribute__((always_inline)) {
    const int buf = it & 1;
    const __bf16* ah = Ah + (long)buf * 2 * WS_ROWS * WS_PITCH;
    const __bf16* al = ah + WS_ROWS * WS_PITCH;
    f32x4 acc[WS_SUB][WS_CB];
#pragma unroll
    for (int s = 0; s < WS_SUB; ++s)
#pragma unroll
      for (int cb = 0; cb < WS_CB; ++cb) acc[s][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#ifdef WS_LAB_NO_FINE
    const bool fine = false;
#else
    const bool fine = steady && !DG;
#endif
    float fpart = 0.f;
    bf16x8 fxah, fxal;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#ifndef WS_LAB_NO_INTERLEAVE
      bf16x8 fah2[WS_SUB], fal2[WS_SUB];
#pragma unroll
      for (int s = 0; s < WS_SUB; ++s) {
        const int o = (16 * s + li) * WS_PITCH + (((4 * ks + lq) ^ li) << 3);
        fah2[s] = *(const bf16x8*)&ah[o]; fal2[s] = *(const bf16x8*)&al[o];
      }
#pragma unroll
      for (int s = 0; s < WS_SUB; ++s)
#pragma unroll
        for (int cb = 0; cb < WS_CB; ++cb) acc[s][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[cb][ks], fah2[s], acc[s][cb], 0, 0, 0);
#pragma unroll
      for (int s = 0; s < WS_SUB; ++s)
#pragma unroll
        for (int cb = 0; cb < WS_CB; ++cb) acc[s][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[cb][ks], fal2[s], acc[s][cb], 0, 0, 0);
#pragma unroll
      for (int s = 0; s < WS_SUB; ++s)
#pragma unroll
        for (int cb = 0; cb < WS_CB; ++cb) acc[s][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[cb][ks], fah2[s], acc[s][cb], 0, 0, 0);
      if (false)
#endif
#pragma unroll
      for (int s = 0; s < WS_SUB; ++s) {
        const int o = (16 * s + li) * WS_PITCH + (((4 * ks + lq) ^ li) << 3);
        const bf16x8 fah = *(const bf16x8*)&ah[o], fal = *(const bf16x8*)&al[o];
#pragma unroll
        for (int cb = 0; cb < WS_CB; ++cb) {             // operands swapped: D[n][m], lane holds C[m = li][n = 4 lq + r]
#ifdef WS_LAB_NO_MFMA
          asm volatile("" :: "v"(fah), "v"(fal), "v"(bl[cb][ks]), "v"(bh[cb][ks]));
#else
          acc[s][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[cb][ks], fah, acc[s][cb], 0, 0, 0);
          acc[s][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[cb][ks], fal, acc[s][cb], 0, 0, 0);
          acc[s][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[cb][ks], fah, acc[s][cb], 0, 0, 0);
#endif
        }
      }
      if (fine) {
        // Steady state: the work of the other pipeline stages is cut into eight pieces, one per k step, and fenced together
        // with that step's 12 MFMAs -- the default scheduler otherwise clusters all 96 MFMAs and the matrix pipe idles during
        // the epilogue / staging arithmetic.  k steps 0..3: the four 16 x 16 blocks of the previous group's epilogue;
        // 4..7: the four blocks of the next group's first layer (or the four staging pieces of the plain variant).
        static_assert(WS_SUB == 2 && WS_CB == 2 && WS_LD == 4, "eight pieces");
        const int par = (it - 1) & 1;
        if (ks < 4) {
          const int s = ks >> 1, cb = ks & 1;
          if (cb == 0) fpart = 0.f;
          epi_block(pacc[s][cb], g - gs, par, s, cb, fpart);
          if (cb == 1) epi_row(par, s, fpart);
        } else if (L0) {
          const int s = (ks - 4) >> 1, cb = (ks - 4) & 1;
          if (cb == 0) prod_x((it + 1) & 1, s, fxah, fxal);
          prod_block(g + gs, buf ^ 1, (it + 1) & 1, s, cb, fxah, fxal);
        } else {
          store_group(buf ^ 1, stn, ks - 4, ks - 3);
          load_piece(g + 2 * gs, stn, ks - 4);
        }
#ifdef WS_FINE_SGB
#pragma unroll
        for (int i = 0; i < 12 + ((L0 && ks >= 4) ? 3 : 0); ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, WS_FINE_SGB, 0);
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (fine) {
      if (L0) { storeX(it & 1); loadX(g + 3 * gs); }
    } else {
      if (!first) epilogue(pacc, g - gs, (it - 1) & 1);
      if (L0) {
        // X rows of group g + gs sit in Xl[(it + 1) & 1] (written one iteration ago); rows of g + 2 gs are in registers
        if (steady || g + gs < p.groups) produce(g + gs, buf ^ 1, (it + 1) & 1, (it + 1) & 1);
        if (steady || g + 2 * gs < p.groups) storeX(it & 1);
        if (steady || g + 3 * gs < p.groups) loadX(g + 3 * gs);
      } else {
        if (steady || g + gs < p.groups) store_group(buf ^ 1, stn);
        if (steady || g + 2 * gs < p.groups) load_group(g + 2 * gs, stn);
      }
    }
    __syncthreads();
    if (!first) finish(g - gs, (it - 1) & 1);
    if (L0 && (steady || g + gs < p.groups)) finish0(g + gs, (it + 1) & 1);
#pragma unroll
    for (int s = 0; s < WS_SUB; ++s)
#pragma unroll
      for (int cb = 0; cb < WS_CB; ++cb) pacc[s][cb] = acc[s][cb];
  };
  int g = g0, it = 0;
  iteration(g, it, st0, true, false);
  g += gs; ++it;
  while (g + 3 * gs < p.groups) {
    iteration(g, it, st0, false, true);
    g += gs; ++it;
  }
  while (g < p.groups) {
    iteration(g, it, st0, false, false);
    g += gs; ++it;
  }
  // drain: the last group's epilogue
  epilogue(pacc, g - gs, (it - 1) & 1);
  __syncthreads();
  finish(g - gs, (it - 1) & 1);
}

// host: does the launch qualify?  (split-bf16 precision, K = N = 256, 16-byte aligned operands)
// CUs one weight-stationary launch spreads over (one workgroup per CU).  ORL_WS_CUS < 256 leaves room for the launches of other
// engines' streams to run side by side instead of one after the other.
static inline int ws_cu_budget() {
  static const int v = [] { const char* f = getenv("ORL_WS_CUS"); const int x = f ? atoi(f) : 0; return (x >= 8 && x <= 256) ? x : 256; }();
  return v;
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

static inline hipError_t launch_ws_fwd(WsFwdP p, int nz, hipStream_t st) {
  p.groups = (p.M + WS_ROWS - 1) / WS_ROWS;
  // one workgroup per CU (register-resident weights): spread the 256 CUs over the nz problems, never more workgroups than CUs
  // (a second partial round of workgroups would double the launch time)
  int per_z = ws_cu_budget() / nz;
  if (per_z < 1) per_z = 1;
  if (per_z > p.groups) per_z = p.groups;
  const bool l0 = p.X0 != nullptr;
  const size_t lds = ws_fwd_lds_bytes(l0);
  static const hipError_t attr_err = [] {       // thread-safe one-time initialisation (engines may launch from several host threads)
    const int big = (int)ws_fwd_lds_bytes(true);
    hipError_t e = hipFuncSetAttribute((const void*)ws_fwd_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_fwd_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_fwd_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_fwd_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_fwd_kernel<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_fwd_kernel<true, true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_fwd_kernel<true, false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
    return e;
  }();
  if (attr_err != hipSuccess) return attr_err;
  const dim3 grid(per_z, 1, nz), block(WS_NT);
  if (p.dmask) hipLaunchKernelGGL((ws_fwd_kernel<false, false, true>), grid, block, lds, st, p);
  else if (l0) {
    if (p.tq && !p.Y) hipLaunchKernelGGL((ws_fwd_kernel<true, true, false, false>), grid, block, lds, st, p);
    else if (p.tq) hipLaunchKernelGGL((ws_fwd_kernel<true, true>), grid, block, lds, st, p);
    else hipLaunchKernelGGL((ws_fwd_kernel<false, true>), grid, block, lds, st, p);
  } else {
    if (p.tq && !p.Y) hipLaunchKernelGGL((ws_fwd_kernel<true, false, false, false>), grid, block, lds, st, p);
    else if (p.tq) hipLaunchKernelGGL((ws_fwd_kernel<true, false>), grid, block, lds, st, p);
    else hipLaunchKernelGGL((ws_fwd_kernel<false, false>), grid, block, lds, st, p);
  }
  return hipGetLastError();
}

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
typedef short s16x4 __attribute__((ext_vector_type(4)));

struct WsDgradP {
  const unsigned int* abits; long ab_s0, ab_s1; int ab_g;     // mask words of the top hidden activation (K = 256 columns)
  const unsigned int* xbits; long xb_s0, xb_s1; int xb_g;     // mask words of the layer-0 activation (N = 256 columns)
  const float* dq; long dq_s0, dq_s1, dq_sm;                   // dLoss/dq per row
  const float* wt; long wt_s0, wt_s1;                          // w_tail [256]
  const float* W; long w_s0, w_s1, w_sn, w_sk;                 // W1: element (k = output unit, n = input unit) at W[n * w_sn + k * w_sk]
                                                               //   nn.Linear (out, in): w_sn = 1, w_sk = 256; EnsembleLinear (in, out): 256, 1
  const float* X; long x_s0, x_s1; int x_pitch, in0;           // layer-0 input rows [M][x_pitch], in0 + 1 <= 32
  float* w0_out; float* b0_out; long o_s0, o_s1, ob_s1, o_ks; int o_sr;   // slab outputs (dW0 [256][in0], db0 [256]); W0 variant
  float* C; long c_s0, c_s1; int c_pitch;                      // dz0 [M][256]; STORE variant
  int M, nz1, groups;
};
enum { WD_XP = WS_ROWS + 4 };                                       // bf16 pitch of an X^T row (72 B: scattered 2-byte stores and 8-byte reads spread over the banks)
static constexpr size_t ws_dgrad_lds_bytes() {     // mask images + X^T images + per-group epilogue operands (dq, h0 mask words)
  return (size_t)2 * WS_ROWS * WS_PITCH * 2 + (size_t)2 * 2 * 32 * WD_XP * 2 + (size_t)2 * (WS_ROWS + WS_NW * WS_ROWS) * 4;
}

template <bool W0, bool STORE>
__global__ __launch_bounds__(WS_NT) void ws_dgrad_w0_kernel(const WsDgradP p) {
  static_assert(WS_NW == 8 && WS_ROWS == 32, "one 32-column mask word per wave, 32-row groups");
  extern __shared__ __attribute__((aligned(16))) float ws_smem[];
  __bf16* Ah = (__bf16*)ws_smem;                                   // [buf][row][256] 0/1 mask as bf16, swizzled
  __bf16* XT = Ah + 2 * WS_ROWS * WS_PITCH;                        // [buf][hi, lo][c = 32][WD_XP]: X^T of the row group
  float* EO = (float*)(XT + 2 * 2 * 32 * WD_XP);                   // [buf][dq[32] | h0 mask words [wave = 8][row = 32]]: epilogue operands
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lq = lane >> 4;
  const int z = blockIdx.z, z0 = z / p.nz1, z1 = z - z0 * p.nz1;
  const unsigned int* __restrict__ ab = p.abits + z0 * p.ab_s0 + z1 * p.ab_s1;
  const unsigned int* __restrict__ xb = p.xbits + z0 * p.xb_s0 + z1 * p.xb_s1;
  const float* __restrict__ dqg = p.dq + z0 * p.dq_s0 + z1 * p.dq_s1;
  const float* __restrict__ Wg = p.W + z0 * p.w_s0 + z1 * p.w_s1;
  const float* __restrict__ wtg = p.wt + z0 * p.wt_s0 + z1 * p.wt_s1;
  const float* __restrict__ Xg = p.X + z0 * p.x_s0 + z1 * p.x_s1;
  const int ncol0 = 32 * wave;

  // resident B' fragments: lane (li, lq) supplies B'[k = 32 ks + 8 lq + j][n = ncol0 + 16 cb + li] = w_tail[k] * W1[k][n]
  bf16x8 bh[2][8], bl[2][8];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int n = ncol0 + 16 * cb + li, k0 = 32 * ks + 8 * lq;
      const f32x4 t0 = *(const f32x4*)&wtg[k0], t1 = *(const f32x4*)&wtg[k0 + 4];
      f32x4 a, b;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a[j] = t0[j] * Wg[(long)n * p.w_sn + (long)(k0 + j) * p.w_sk];
        b[j] = t1[j] * Wg[(long)n * p.w_sn + (long)(k0 + 4 + j) * p.w_sk];
      }
      ws_split8(a, b, bh[cb][ks], bl[cb][ks]);
    }
  // zero both X^T images once (rows c >= x_pitch are never written again)
  if (W0) for (int e = tid; e < 2 * 2 * 32 * WD_XP / 2; e += WS_NT) ((unsigned int*)XT)[e] = 0u;
  __syncthreads();

  // ---- staging of one row group: thread (row r = t >> 4, half-word hw = t & 15) expands 16 mask bits; X^T elements ----
  unsigned int sm_word;
  float sx[2];
  const int xe = W0 ? WS_ROWS * p.x_pitch : 0;                       // X elements of a row group (<= 1024)
  float* __restrict__ Cg = STORE ? p.C + z0 * p.c_s0 + z1 * p.c_s1 : nullptr;
  // X element e = tid + 512 i of a row group -> X^T position (column c, row rr); surplus threads use a pad slot that is never read
  // (rows are consumed as 32 of the WD_XP entries); computed once: no division and no predication inside the loop
  int xo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + WS_NT * i;
    int rr = W0 ? e / (W0 ? p.x_pitch : 1) : 0, c = W0 ? e - rr * p.x_pitch : 0;
    if (e >= xe) { rr = 32; c = 0; }
    xo[i] = ((c == p.in0) ? (1 << 16) : 0) | (c * WD_XP + rr);        // bit 16: the ones column (bias gradient)
  }
  float sdq;
  unsigned int sxw;
  auto load_group = [&](int g) __attribute__((always_inline)) {
    sm_word = ab[(long)(g * WS_ROWS + (tid >> 4)) * p.ab_g + ((tid & 15) >> 1)];
    // the epilogue's dq and h0 mask words travel through LDS with the group (fetched a full iteration ahead by the staging threads:
    // the epilogue then has no global loads of its own to wait for)
    sdq = dqg[(long)(g * WS_ROWS + (tid & 31)) * p.dq_sm];
    sxw = xb[(long)(g * WS_ROWS + ((tid >> 3) & 31)) * p.xb_g + (tid & 7)];
    if (W0) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int e = tid + WS_NT * i;
        sx[i] = Xg[(long)g * xe + (e < xe ? e : xe - 1)];              // clamped, not predicated
      }
    }
  };
  auto store_group = [&](int buf) __attribute__((always_inline)) {
    const int r = tid >> 4, hw = tid & 15;
    const unsigned int bits = (sm_word >> (16 * (hw & 1))) & 0xFFFFu;
    u32x4 c0, c1;                                                    // 16 bf16 values: 1.0 = 0x3F80 where the bit is set
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const unsigned int y0 = (bits >> (2 * j)) & 3u, y1 = (bits >> (8 + 2 * j)) & 3u;
      c0[j] = ((y0 & 1u) | ((y0 >> 1) << 16)) * 0x3F80u;
      c1[j] = ((y1 & 1u) | ((y1 >> 1) << 16)) * 0x3F80u;
    }
    __bf16* d = Ah + (long)buf * WS_ROWS * WS_PITCH + r * WS_PITCH;
    *(u32x4*)(d + (((2 * hw) ^ (r & 15)) << 3)) = c0;
    *(u32x4*)(d + (((2 * hw + 1) ^ (r & 15)) << 3)) = c1;
    float* eo = EO + buf * (WS_ROWS + WS_NW * WS_ROWS);
    eo[tid & 31] = sdq;                                              // (replicated writes of identical values)
    ((unsigned int*)eo)[WS_ROWS + (tid & 7) * WS_ROWS + ((tid >> 3) & 31)] = sxw;
    __bf16* xt = XT + (long)buf * 2 * 32 * WD_XP;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (W0) {
        const float x = (xo[i] >> 16) ? 1.0f : sx[i];
        const __bf16 hh = (__bf16)x;
        xt[xo[i] & 0xFFFF] = hh;
        xt[32 * WD_XP + (xo[i] & 0xFFFF)] = (__bf16)(x - (float)hh);
      }
    }
  };

  f32x4 d2[2][2];                                                    // dW0^T blocks [c block][cb], accumulated over all groups
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) d2[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int g0 = blockIdx.x, gs = gridDim.x;
  if (g0 < p.groups) {
    load_group(g0);
    store_group(0);
    if (g0 + gs < p.groups) load_group(g0 + gs);
  }
  __syncthreads();
  // steady = true: groups g + gs and g + 2 gs exist -> the body is one basic block (no conditionals)
  auto iteration = [&](int g, int it, bool steady) __attribute__((always_inline)) {
    const int buf = it & 1;
    // epilogue operands of this group: dq of the lane's 4 rows per 16-row block, and the h0 mask word of those rows
    f32x4 dq4[WS_SUB];
    unsigned int xw[WS_SUB][4];
    const float* eo = EO + buf * (WS_ROWS + WS_NW * WS_ROWS);
#pragma unroll
    for (int s = 0; s < WS_SUB; ++s) {
      dq4[s] = *(const f32x4*)&eo[16 * s + 4 * lq];
      const u32x4 w4 = *(const u32x4*)&((const unsigned int*)eo)[WS_ROWS + wave * WS_ROWS + 16 * s + 4 * lq];
#pragma unroll
      for (int r = 0; r < 4; ++r) xw[s][r] = w4[r];
    }
    const __bf16* ah = Ah + (long)buf * WS_ROWS * WS_PITCH;
    f32x4 acc[WS_SUB][2];
#pragma unroll
    for (int s = 0; s < WS_SUB; ++s)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) acc[s][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
      for (int s = 0; s < WS_SUB; ++s) {
        const bf16x8 fa = *(const bf16x8*)&ah[(16 * s + li) * WS_PITCH + (((4 * ks + lq) ^ li) << 3)];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {               // D[m][n]: lane holds rows 4 lq + r of column li
          acc[s][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, bl[cb][ks], acc[s][cb], 0, 0, 0);
          acc[s][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, bh[cb][ks], acc[s][cb], 0, 0, 0);
        }
      }
    }
    // dz0 block -> (hi, lo) bf16 B operand of the 16x16x16 MFMA; A operand = X^T rows c, columns m = 16 s + 4 lq ..
    const __bf16* xth = XT + (long)buf * 2 * 32 * WD_XP;
    const __bf16* xtl = xth + 32 * WD_XP;
#pragma unroll
    for (int s = 0; s < WS_SUB; ++s) {
      s16x4 xh[2], xl[2];
      if (W0) {
#pragma unroll
        for (int cbk = 0; cbk < 2; ++cbk) {
          xh[cbk] = *(const s16x4*)&xth[(16 * cbk + li) * WD_XP + 16 * s + 4 * lq];
          xl[cbk] = *(const s16x4*)&xtl[(16 * cbk + li) * WD_XP + 16 * s + 4 * lq];
        }
      }
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        bf16x4 zh, zl;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = ((xw[s][r] >> (16 * cb + li)) & 1u) ? acc[s][cb][r] * dq4[s][r] : 0.f;
          if (STORE) Cg[(long)(g * WS_ROWS + 16 * s + 4 * lq + r) * p.c_pitch + ncol0 + 16 * cb + li] = v;
          const __bf16 hh = (__bf16)v;
          zh[r] = hh; zl[r] = (__bf16)(v - (float)hh);
        }
        const s16x4 bzh = *(const s16x4*)&zh, bzl = *(const s16x4*)&zl;
#pragma unroll
        for (int cbk = 0; cbk < 2 && W0; ++cbk) {
          d2[cbk][cb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(xl[cbk], bzh, d2[cbk][cb], 0, 0, 0);
          d2[cbk][cb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(xh[cbk], bzl, d2[cbk][cb], 0, 0, 0);
          d2[cbk][cb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(xh[cbk], bzh, d2[cbk][cb], 0, 0, 0);
        }
      }
    }
    if (steady || g + gs < p.groups) store_group(buf ^ 1);
    if (steady || g + 2 * gs < p.groups) load_group(g + 2 * gs);
    __syncthreads();
  };
  int g = g0, it = 0;
  // (a conditional-free steady-state copy of the body, as in ws_fwd / ws_wgrad, measured 4 % slower here: 688 vs 658 us)
  for (; g < p.groups; g += gs, ++it) iteration(g, it, false);
  if (!W0) return;
  // one slab per workgroup: lane (li, lq) holds dW0^T[c = 16 cbk + 4 lq + r][n = ncol0 + 16 cb + li]
  float* wo = p.w0_out + z0 * p.o_s0 + z1 * p.o_s1 + (long)blockIdx.x * p.o_ks;
  float* bo = p.b0_out + z0 * p.o_s0 + z1 * p.ob_s1 + (long)blockIdx.x * p.o_ks;
#pragma unroll
  for (int cbk = 0; cbk < 2; ++cbk)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 16 * cbk + 4 * lq + r, n = ncol0 + 16 * cb + li;
        if (c < p.in0) wo[(long)n * p.o_sr + c] = d2[cbk][cb][r];
        else if (c == p.in0) bo[n] = d2[cbk][cb][r];
      }
}

static inline bool ws_dgrad_supported(const WsDgradP& p, int K, int N) {
  if (K != WS_K || N != WS_N || p.M < 256 || (p.M % WS_ROWS)) return false;
  if (!p.abits || !p.xbits || p.ab_g != 8 || p.xb_g != 8) return false;
  if (p.w0_out && (p.in0 + 1 > 32 || p.in0 >= p.x_pitch || p.x_pitch > 32 || WS_ROWS * p.x_pitch > 2 * WS_NT)) return false;
  if (!p.w0_out && !p.C) return false;
  if (!aligned16(p.wt) || (p.wt_s0 & 3) || (p.wt_s1 & 3)) return false;
  return true;
}
// blocks per problem (= split-K slabs written per problem)
static inline int ws_dgrad_blocks(int M, int nz, int max_slab) {
  const int groups = M / WS_ROWS;
  int per_z = ws_cu_budget() / nz;
  if (per_z > groups) per_z = groups;
  if (per_z > max_slab) per_z = max_slab;
  return per_z < 1 ? 1 : per_z;
}
static inline hipError_t launch_ws_dgrad_w0(WsDgradP p, int nz, int per_z, hipStream_t st) {
  p.groups = p.M / WS_ROWS;
  if (p.w0_out) hipLaunchKernelGGL((ws_dgrad_w0_kernel<true, false>), dim3(per_z, 1, nz), dim3(WS_NT), ws_dgrad_lds_bytes(), st, p);
  else hipLaunchKernelGGL((ws_dgrad_w0_kernel<false, true>), dim3(per_z, 1, nz), dim3(WS_NT), ws_dgrad_lds_bytes(), st, p);
  return hipGetLastError();
}

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

template <int MODE>      // 0: dW1 / db1 only, 1: h1 streamed for the tail gradients, 2: tail gradients derived from the accumulators
__global__ __launch_bounds__(WS_NT) void ws_wgrad_kernel(const WsWgradP p) {
  static_assert(WS_NW == 8 && WS_ROWS == 32, "8 waves x 32 columns, 32-row groups");
  constexpr bool TAILS = (MODE == 1);
  extern __shared__ __attribute__((aligned(16))) float ws_smem[];
  __bf16* img = (__bf16*)ws_smem;                                   // [buf][{mask, G hi, G lo}][32][256]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lq = lane >> 4;
  const int z = blockIdx.z, z0 = z / p.nz1, z1 = z - z0 * p.nz1;
  const unsigned int* __restrict__ ab = p.abits + z0 * p.ab_s0 + z1 * p.ab_s1;
  const float* __restrict__ dqg = p.dq + z0 * p.dq_s0 + z1 * p.dq_s1;
  const float* __restrict__ H0g = p.H0 + z0 * p.h0_s0 + z1 * p.h0_s1;
  const float* __restrict__ wtg = p.wt + z0 * p.wt_s0 + z1 * p.wt_s1;
  const int ncol0 = 32 * wave;

  f32x4 acc[16][2], accb[2];
#pragma unroll
  for (int kb = 0; kb < 16; ++kb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) acc[kb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  accb[0] = accb[1] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- staging registers of one row group ----
  f32x4 s0[4];
  f32x4 s1[TAILS ? 4 : 1];
  f32x4 tacc = (f32x4){0.f, 0.f, 0.f, 0.f}, bacc = (f32x4){0.f, 0.f, 0.f, 0.f};   // TAILS: dw_tail / db1 partials of columns 4 (tid & 63) ..
  float dqsum = 0.f;
  const float* __restrict__ H1g = TAILS ? p.H1 + z0 * p.h1_s0 + z1 * p.h1_s1 : nullptr;
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
      if (TAILS) s1[i] = *(const f32x4*)&H1g[m * p.h1_pitch + 4 * kq];
      sdq[i] = dqg[m * p.dq_sm];
    }
  };
  auto load_piece = [&](int g, int i) __attribute__((always_inline)) {
    const int idx = tid + WS_NT * i, r = idx >> 6, kq = idx & 63;
    const long m = (long)g * WS_ROWS + r;
    s0[i] = *(const f32x4*)&H0g[m * p.h0_pitch + 4 * kq];
    if (TAILS) s1[i] = *(const f32x4*)&H1g[m * p.h1_pitch + 4 * kq];
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
    if (TAILS) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { tacc[j] += sdq[i] * s1[i][j]; bacc[j] += s1[i][j] > 0.f ? sdq[i] : 0.f; }
      dqsum += sdq[i];
    } else if (kq == 0) {                                            // this row's dq into the bias-gradient operand block
      const __bf16 hh = (__bf16)sdq[i];
      __bf16* dqi = dqimg + (long)buf * 2 * WS_ROWS * 16;
      dqi[r * 16] = hh; dqi[WS_ROWS * 16 + r * 16] = (__bf16)(sdq[i] - (float)hh);
      if (MODE == 2) dqsum += sdq[i];
    }
  };
  auto store_group = [&](int buf) __attribute__((always_inline)) {
    store_mask(buf);
#pragma unroll
    for (int i = 0; i < 4; ++i) store_piece(buf, i);
  };
  if (!TAILS) for (int e = tid; e < 2 * 2 * WS_ROWS * 16 / 2; e += WS_NT) ((unsigned int*)dqimg)[e] = 0u;   // columns 1..15 stay zero
  __syncthreads();

  const int g0 = blockIdx.x, gs = gridDim.x;
  if (g0 < p.groups) {
    load_group(g0);
    store_group(0);
    if (g0 + gs < p.groups) load_group(g0 + gs);
  }
  __syncthreads();
  // steady = true: groups g + gs and g + 2 gs exist -> no conditionals, the whole body up to the barrier is one basic block
  // (LDS reads of the next k step are hoisted over the MFMAs of the current one)
  auto iteration = [&](int g, int it, bool steady) __attribute__((always_inline)) {
    const int buf = it & 1;
    const bool more = steady || g + gs < p.groups, more2 = steady || g + 2 * gs < p.groups;
    const __bf16* mi = img + (long)buf * 3 * WW_IMG;
    const __bf16* gh = mi + WW_IMG;
    const __bf16* gl = gh + WW_IMG;
    const __bf16* dqi = dqimg + (long)buf * 2 * WS_ROWS * 16;
    {
      // one v_mfma_f32_16x16x32_bf16 covers the whole 32-row group: its 8 k-values per lane are the two transposed reads of
      // rows 4 lq .. + 3 and 16 + 4 lq .. + 3 (the k order is free as long as A and B agree)
      auto cat = [](s16x4 x, s16x4 y) __attribute__((always_inline)) {
        bf16x8 r;
        *(s16x4*)&r = x; *((s16x4*)&r + 1) = y;
        return r;
      };
      bf16x8 bh[2], bl[2];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        bh[nb] = cat(ww_tr(gh, 0, ncol0 + 16 * nb, lane), ww_tr(gh, 16, ncol0 + 16 * nb, lane));
        bl[nb] = cat(ww_tr(gl, 0, ncol0 + 16 * nb, lane), ww_tr(gl, 16, ncol0 + 16 * nb, lane));
      }
      const int dro0 = (4 * lq + (li >> 2)) * 16 + 4 * (li & 3), dro1 = dro0 + 16 * 16;
      typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
      bf16x8 bdh, bdl;
      if (!TAILS) {
        bdh = cat(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(dqi + dro0)), __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(dqi + dro1)));
        bdl = cat(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(dqi + WS_ROWS * 16 + dro0)),
                  __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(dqi + WS_ROWS * 16 + dro1)));
      }
#pragma unroll
      for (int kp = 0; kp < 8; ++kp) {                               // two 16-row k blocks per trip: dependent MFMAs are 4 apart
        const int kb0 = 2 * kp, kb1 = kb0 + 1;
        const bf16x8 a0 = cat(ww_tr(mi, 0, 16 * kb0, lane), ww_tr(mi, 16, 16 * kb0, lane));   // A[i = k][kk = m] = mask[m][k]
        const bf16x8 a1 = cat(ww_tr(mi, 0, 16 * kb1, lane), ww_tr(mi, 16, 16 * kb1, lane));
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[kb0][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bl[nb], acc[kb0][nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[kb1][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bl[nb], acc[kb1][nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[kb0][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bh[nb], acc[kb0][nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[kb1][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bh[nb], acc[kb1][nb], 0, 0, 0);
        if (!TAILS && kp == 7) {                                     // this wave's share of db1: k blocks 2 wave, 2 wave + 1 (own reads: no branch)
          const bf16x8 c0 = cat(ww_tr(mi, 0, 32 * wave, lane), ww_tr(mi, 16, 32 * wave, lane));
          const bf16x8 c1 = cat(ww_tr(mi, 0, 32 * wave + 16, lane), ww_tr(mi, 16, 32 * wave + 16, lane));
          accb[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c0, bdl, accb[0], 0, 0, 0);
          accb[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c1, bdl, accb[1], 0, 0, 0);
          accb[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c0, bdh, accb[0], 0, 0, 0);
          accb[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(c1, bdh, accb[1], 0, 0, 0);
        }
        // each staging register is written to LDS and refilled at the same point of every iteration: a full iteration in flight
        if (kp < 4) {
          if (more) store_piece(buf ^ 1, kp);
          if (more2) load_piece(g + 2 * gs, kp);
        }
        if (kp == 5) {
          if (more) store_mask(buf ^ 1);
          if (more2) load_mask(g + 2 * gs);
        }
      }
    }
    __syncthreads();
  };
  int g = g0, it = 0;
  for (; MODE != 1 && g + 2 * gs < p.groups; g += gs, ++it) iteration(g, it, true);    // (the h1-streaming variant measured slower that way)
  for (; g < p.groups; g += gs, ++it) iteration(g, it, false);

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
  if (MODE == 2) {
    // dw_tail partial of this slab: every lane folds its two input columns of each of its 64 output units, the 16 lanes of a
    // group and then the 8 waves (= all 256 input columns) are summed in a fixed order
    const float* __restrict__ W1g = p.W1 + z0 * p.w1_s0 + z1 * p.w1_s1;
    float* red = ws_smem;                                            // the images are dead after the loop's last barrier
#pragma unroll
    for (int kb = 0; kb < 16; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int o = 16 * kb + 4 * lq + r;
        float t = W1g[(long)o * WS_N + ncol0 + li] * acc[kb][0][r] + W1g[(long)o * WS_N + ncol0 + 16 + li] * acc[kb][1][r];
        t += __shfl_xor(t, 1); t += __shfl_xor(t, 2); t += __shfl_xor(t, 4); t += __shfl_xor(t, 8);
        if (li == 0) red[wave * WS_K + o] = t;
      }
    if (li == 0) {
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[8 * WS_K + 16 * (2 * wave + x) + 4 * lq + r] = accb[x][r];
    }
    if (lane == 0) red[9 * WS_K + wave] = dqsum;
    __syncthreads();
    if (tid < WS_K) {
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) a += red[w * WS_K + tid];
      const float gs1 = red[8 * WS_K + tid];
      p.dwt[so + z1 * p.o_s1wt + tid] = a + (p.b1 + z0 * p.b1_s0 + z1 * p.b1_s1)[tid] * gs1;
      db[tid] = wtg[tid] * gs1;
    }
    if (tid == 0) {
      float a = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) a += red[9 * WS_K + w];
      p.dbt[so + z1 * p.o_s1bt] = a;
    }
    return;
  }
  if (!TAILS) {
    if (li == 0) {
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        const int k0 = 16 * (2 * wave + x) + 4 * lq;
#pragma unroll
        for (int r = 0; r < 4; ++r) db[k0 + r] = wtg[k0 + r] * accb[x][r];
      }
    }
    return;
  }
  // TAILS: eight row-slice partial sums per column (threads tid, tid + 64, ...), summed in a fixed order
  float* red = ws_smem;                                              // the images are dead after the loop's last barrier
  *(f32x4*)&red[(tid >> 6) * WS_K + 4 * (tid & 63)] = tacc;
  *(f32x4*)&red[(8 + (tid >> 6)) * WS_K + 4 * (tid & 63)] = bacc;
  if ((tid & 63) == 0) red[16 * WS_K + (tid >> 6)] = dqsum;
  __syncthreads();
  if (tid < WS_K) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) { a += red[w * WS_K + tid]; b += red[(8 + w) * WS_K + tid]; }
    p.dwt[so + z1 * p.o_s1wt + tid] = a;
    db[tid] = wtg[tid] * b;
  }
  if (tid == 0) {
    float a = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) a += red[16 * WS_K + w];
    p.dbt[so + z1 * p.o_s1bt] = a;
  }
}

static inline bool ws_wgrad_supported(const WsWgradP& p, int K, int N) {
  if (K != WS_K || N != WS_N || p.M < 256 || (p.M % WS_ROWS) || !p.abits || p.ab_g != 8) return false;
  if (!aligned16(p.H0) || (p.h0_pitch & 3) || (p.h0_s0 & 3) || (p.h0_s1 & 3)) return false;
  if (!p.H1 && p.W1 && (!p.b1 || !p.dwt || !p.dbt)) return false;
  if (p.H1 && (!aligned16(p.H1) || (p.h1_pitch & 3) || (p.h1_s0 & 3) || (p.h1_s1 & 3) || !p.dwt || !p.dbt)) return false;
  return aligned16(p.wt) && !(p.wt_s0 & 3) && !(p.wt_s1 & 3);
}
static inline hipError_t launch_ws_wgrad(WsWgradP p, int nz, int per_z, hipStream_t st) {
  p.groups = p.M / WS_ROWS;
  static const hipError_t attr_err = [] {
    hipError_t e = hipFuncSetAttribute((const void*)ws_wgrad_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ws_wgrad_lds_bytes());
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_wgrad_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ws_wgrad_lds_bytes());
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_wgrad_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ws_wgrad_lds_bytes());
    return e;
  }();
  if (attr_err != hipSuccess) return attr_err;
  if (p.H1) hipLaunchKernelGGL(ws_wgrad_kernel<1>, dim3(per_z, 1, nz), dim3(WS_NT), ws_wgrad_lds_bytes(), st, p);
  else if (p.W1) hipLaunchKernelGGL(ws_wgrad_kernel<2>, dim3(per_z, 1, nz), dim3(WS_NT), ws_wgrad_lds_bytes(), st, p);
  else hipLaunchKernelGGL(ws_wgrad_kernel<0>, dim3(per_z, 1, nz), dim3(WS_NT), ws_wgrad_lds_bytes(), st, p);
  return hipGetLastError();
}

}  // namespace orl
