// ws_fwd3.hip — precision 2 (three fp16 planes, fp32-class arithmetic) flavour of ws_fwd_kernel<TQ, L0, ., SY = false>: the fused
// first + second layer forward of a single-output net with the tail folded in (interface and design notes: ws_gemm.h; the two-plane
// kernel this follows: ws_fwd.hip).
//
// x = hi + mid + lo in fp16 planes represents an fp32 operand exactly; the six products hi*hi, hi*mid, mid*hi, hi*lo, mid*mid, lo*hi with
// fp32 accumulation drop only terms below 2^-33 of a product.  Three resident planes of W1 for 32 columns would be 192 VGPRs per lane, so a
// workgroup owns HALF of the net's 256 output columns (blockIdx.y = the half; wave w: columns 128 half + 16 w .. + 15 = 96 VGPRs) and two
// workgroups stream the same row groups.  Layer 1 needs all 256 columns of h0, so BOTH halves run the fused first layer for all columns
// (wave w: h0 columns 32 w .. + 31, 24 of a wave's 120 MFMAs per group); a wave stores its h0 block to HBM only in the workgroup whose half
// contains those columns -- the other copy goes to a scratch line through the same store instruction (a wave-uniform branch inside the row
// loop costs the product kernel ~20 %: DESIGN.md section 8).  Each half reduces the tail over its 128 columns: half 0 writes q (with the
// tail bias), half 1 a partial that k_tail_add folds in.  Mask words: a half writes its four words per row of h1; half 0 writes h0's.
// Reference: nets/mlp.py:9-33 (Linear + ReLU stack), modules/critic_module.py:17-28 (the single-output tail), modules/actor_module.py:22-27.
#include "ws_device.h"

namespace orl {

enum { WF3_XLP = 52 };      // float pitch of a narrow-input row: three planes of 32 fp16 slots + pad (208 B: sixteen rows cover the 64 banks once)

// TQ: the single-output tail folded in (two partial sums); SY: the top activation is stored (each half its 128 columns); XS: h0 is stored
// (false: forward-only passes, WsFwdP::x0_discard).  L0 = false: no fused first layer -- the layer's input rows come from HBM (a hidden layer
// above the second one of a deeper net: run_cql.py:31's [256, 256, 256]) and are split into the three planes while both halves stage them.
// DG (with L0 = false, SY): plain dgrad mode -- Y = (X B^T) (.) mask with X a materialised gradient (times the run's dynamic scale while it is
// staged), B the weights viewed transposed, no bias / ReLU / mask emission; the mask of the receiving activation comes from its packed bits.
// All compile-time flavours, as in ws_fwd_kernel.
template <bool TQ, bool SY, bool XS, bool L0 = true, bool DG = false>
__global__ __launch_bounds__(WS_NT) void ws_fwd3_kernel(const WsFwdP p) {
  static_assert(!DG || (!TQ && SY && !L0), "gradient mode: rows from HBM, result stored, no tail");
  static_assert(WS_NW == 8 && WS_ROWS == 32 && WS_SUB == 2, "eight waves, 32-row groups");
  extern __shared__ __attribute__((aligned(16))) float ws_smem[];
  hx_t* Ah = (hx_t*)ws_smem;                                           // [buf][hi, mid, lo][row][WS_PITCH]
  float* qs = ws_smem + (2 * 3 * WS_ROWS * WS_PITCH * 2) / 4;             // [parity][wave][row]
  unsigned char* nbs = (unsigned char*)(qs + 2 * WS_NW * WS_ROWS);         // [parity][row][WS_NBP]: 4 mask bits per (row, 4 columns of this half)
  float* Xl = (float*)(nbs + 2 * WS_ROWS * WS_NBP);                        // [buf][row][WF3_XLP]: narrow input rows, planes in 16-bit slots 0.. / 32.. / 64..
  unsigned char* nbs0 = (unsigned char*)(Xl + 2 * WS_ROWS * WF3_XLP);      // mask nibbles of the produced h0 (all 256 columns)
  float* cst = (float*)(nbs0 + 2 * WS_ROWS * WS_NBP);                      // [bias | tail weights]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lq = lane >> 4;
  const int half = blockIdx.y;
  const int z = blockIdx.z, z0 = z / p.nz1, z1 = z - z0 * p.nz1;
  const float* __restrict__ Wg = p.W + z0 * p.w_s0 + z1 * p.w_s1;
  const float* __restrict__ bg = DG ? nullptr : p.bias + z0 * p.b_s0 + z1 * p.b_s1;
  const unsigned int* __restrict__ dmg = DG ? p.dmask + z0 * p.dm_s0 + z1 * p.dm_s1 : nullptr;
  const float a_sc = (DG && p.gscale) ? p.gscale[z0] : 1.f;          // gradient mode: the staged rows carry the run's dynamic gradient scale
  float* __restrict__ Y0g = const_cast<float*>(p.X) + z0 * p.x_s0 + z1 * p.x_s1;      // h0 is written where the plain kernel reads it
  float* __restrict__ Yg = SY ? p.Y + z0 * p.y_s0 + z1 * p.y_s1 : nullptr;
  const float* __restrict__ X0g = L0 ? p.X0 + z0 * p.x0_s0 + z1 * p.x0_s1 : nullptr;
  const float* __restrict__ Xg = p.X + z0 * p.x_s0 + z1 * p.x_s1;           // L0 = false: the layer's input rows
  const int ncol1 = 128 * half + 16 * wave;                              // layer-1 columns of this wave
  const int ncol0 = 32 * wave;                                           // h0 columns this wave produces
  // the h0 store of a wave whose columns belong to the other half lands in a per-workgroup scratch line (row pitch 0)
  const bool mine = (wave >> 2) == half;
  float* __restrict__ y0b = !XS ? nullptr : (mine ? Y0g : p.dump + (long)(((blockIdx.z * gridDim.x + blockIdx.x) * 2 + half) & (WS_DUMP_SLOTS - 1)) * WS_N);
  const long y0p = mine ? p.x_pitch : 0;

  // ---- resident layer-1 fragments: lane (li, lq) supplies W1[n = ncol1 + li][k = 32 ks + 8 lq .. + 7], times ORL_WSCALE ----
  hx8 bh[8], bm[8], bl[8];
  {
    f32x4 raw[8][2];
    if (p.w_sk == 1) {                                                   // nn.Linear (out, in): k-contiguous rows
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const float* src = Wg + (long)(ncol1 + li) * p.w_sn + (32 * ks + 8 * lq);
        raw[ks][0] = *(const f32x4*)src; raw[ks][1] = *(const f32x4*)(src + 4);
      }
    } else {                                                             // EnsembleLinear (in, out): eight strided loads per fragment, once per workgroup
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const float* src = Wg + (long)(ncol1 + li) * p.w_sn + (long)(32 * ks + 8 * lq) * p.w_sk;
#pragma unroll
        for (int j = 0; j < 4; ++j) { raw[ks][0][j] = src[(long)j * p.w_sk]; raw[ks][1][j] = src[(long)(4 + j) * p.w_sk]; }
      }
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) ws_split8x3(raw[ks][0] * ORL_WSCALE, raw[ks][1] * ORL_WSCALE, bh[ks], bm[ks], bl[ks]);
  }
  // first-layer fragments of columns ncol0 + 16 cb + li, K = 32: W0'[n][k] = W0[n][k] (k < in0), b0[n] (k == in0), 0 beyond (unscaled, as in ws_fwd)
  hx8 b0h[2], b0m[2], b0l[2];
  if constexpr (L0) {
    const float* __restrict__ W0g = p.W0 + z0 * p.w0_s0 + z1 * p.w0_s1;
    const float* __restrict__ b0g = p.b0 + z0 * p.b0_s0 + z1 * p.b0_s1;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int n = ncol0 + 16 * cb + li;
      f32x4 a, b;
      const float bn = b0g[n];
#pragma unroll
      for (int j = 0; j < 4; ++j) {                                      // clamped addresses and 0 / 1 factors, not guarded loads (ws_fwd.hip)
        const int k0 = 8 * lq + j, k1 = k0 + 4;
        const float w0 = W0g[(long)n * p.w0_sn + (long)(k0 < p.in0 ? k0 : p.in0 - 1) * p.w0_sk];
        const float w1 = W0g[(long)n * p.w0_sn + (long)(k1 < p.in0 ? k1 : p.in0 - 1) * p.w0_sk];
        a[j] = (k0 < p.in0 ? 1.f : 0.f) * w0 + (k0 == p.in0 ? 1.f : 0.f) * bn;
        b[j] = (k1 < p.in0 ? 1.f : 0.f) * w1 + (k1 == p.in0 ? 1.f : 0.f) * bn;
      }
      ws_split8x3(a, b, b0h[cb], b0m[cb], b0l[cb]);
    }
  }
  const float* __restrict__ twg = TQ ? p.tw + z0 * p.tw_s0 + z1 * p.tw_s1 : bg;
  if (!DG && tid < WS_N) { cst[tid] = bg[tid]; cst[WS_N + tid] = twg[tid]; }     // visible after the prologue's barriers
  const float tbias = (TQ && half == 0) ? (p.tb + z0 * p.tb_s0 + z1 * p.tb_s1)[0] : 0.f;
  float* __restrict__ tqo = !TQ ? nullptr : (half == 0 ? p.tq + z0 * p.tq_s0 + z1 * p.tq_s1 : p.tq2 + z0 * p.tq2_s0 + z1 * p.tq2_s1);
  const long tqsm = half == 0 ? p.tq_sm : 1;
  const float inv_sc = 1.0f / (ORL_WSCALE * a_sc);

  // ---- narrow-input staging (two elements per thread), split ONCE into three planes ----
  const int xe = L0 ? WS_ROWS * p.x0_pitch : 0;
  int xr[2], xc[2];
  float sx[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + WS_NT * i;
    xr[i] = L0 ? e / (L0 ? p.x0_pitch : 1) : 0; xc[i] = L0 ? e - xr[i] * p.x0_pitch : 0;
    if (L0 && e >= xe) { xr[i] = 0; xc[i] = 32; }                        // surplus threads: pad slots
  }
  // ---- L0 = false: staging of one [32][256] row group from HBM (thread t moves float4 #(t + 512 i), i = 0..3), split into three planes ----
  f32x4 st0[WS_LD];
  auto load_piece = [&](int g, int i) __attribute__((always_inline)) {
    const int idx = tid + WS_NT * i, r = idx >> 6, kq = idx & 63;
    st0[i] = *(const f32x4*)&Xg[((long)g * WS_ROWS + r) * p.x_pitch + 4 * kq];
  };
  auto store_piece = [&](int buf, int i) __attribute__((always_inline)) {
    hx_t* dh = Ah + (long)buf * 3 * WS_ROWS * WS_PITCH;
    const int idx = tid + WS_NT * i, r = idx >> 6, kq = idx & 63;
    hx4 h, mm, l;
    if (DG) orl_split4x3(st0[i] * a_sc, h, mm, l); else orl_split4x3(st0[i], h, mm, l);
    const int o = r * WS_PITCH + ((((kq >> 1) ^ (r & 15)) << 3) | ((kq & 1) << 2));
    *(hx4*)(dh + o) = h;
    *(hx4*)(dh + WS_ROWS * WS_PITCH + o) = mm;
    *(hx4*)(dh + 2 * WS_ROWS * WS_PITCH + o) = l;
  };
  auto loadX = [&](int g) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int e = tid + WS_NT * i; sx[i] = X0g[(long)g * xe + (e < xe ? e : xe - 1)]; }   // clamped, not predicated
  };
  auto storeX = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float x = (xc[i] == p.in0) ? 1.0f : sx[i];
      hx_t* row = (hx_t*)(Xl + (buf * WS_ROWS + xr[i]) * WF3_XLP);
      const bool pad = xc[i] >= 32;
      hx_t hh, mm, ll;
      orl_split1x3(x, hh, mm, ll);
      row[pad ? 96 : xc[i]] = hh;
      row[pad ? 97 : 32 + xc[i]] = mm;
      row[pad ? 98 : 64 + xc[i]] = ll;
    }
  };
  auto prod_x = [&](int xbuf, int s, hx8& xah, hx8& xam, hx8& xal) __attribute__((always_inline)) {
    const hx_t* xrow = (const hx_t*)(Xl + (xbuf * WS_ROWS + 16 * s + li) * WF3_XLP) + 8 * lq;
    xah = *(const hx8*)xrow; xam = *(const hx8*)(xrow + 32); xal = *(const hx8*)(xrow + 64);
  };
  // one 16 x 16 block of h0 (row block s, column block cb of this wave) -> global (fp32), the three planes of LDS image `buf`, mask nibbles
  auto prod_block = [&](int g, int buf, int par, int s, int cb, const hx8& xah, const hx8& xam, const hx8& xal) __attribute__((always_inline)) {
    hx_t* dh = Ah + (long)buf * 3 * WS_ROWS * WS_PITCH;
    const int r = 16 * s + li;
    const long m = (long)g * WS_ROWS + r;
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
    v = ORL_MFMA_16x16x32(b0l[cb], xah, v);
    v = ORL_MFMA_16x16x32(b0h[cb], xal, v);
    v = ORL_MFMA_16x16x32(b0m[cb], xam, v);
    v = ORL_MFMA_16x16x32(b0m[cb], xah, v);
    v = ORL_MFMA_16x16x32(b0h[cb], xam, v);
    v = ORL_MFMA_16x16x32(b0h[cb], xah, v);
    const unsigned int nib0 = orl_relu_mask4(v);
    const int k = ncol0 + 16 * cb + 4 * lq;                              // h0 columns k .. k + 3 of row r
    if constexpr (XS) *(f32x4*)&y0b[m * y0p + k] = v;
    hx4 h, mm, l;
    orl_split4x3(v, h, mm, l);
    const int o = r * WS_PITCH + ((((k >> 3) ^ (r & 15)) << 3) | (((k >> 2) & 1) << 2));
    *(hx4*)(dh + o) = h;
    *(hx4*)(dh + WS_ROWS * WS_PITCH + o) = mm;
    *(hx4*)(dh + 2 * WS_ROWS * WS_PITCH + o) = l;
    nbs0[(par * WS_ROWS + r) * WS_NBP + (k >> 2)] = (unsigned char)nib0;
  };
  auto produce = [&](int g, int buf, int xbuf, int par) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < WS_SUB; ++s) {
      hx8 xah, xam, xal;
      prod_x(xbuf, s, xah, xam, xal);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) prod_block(g, buf, par, s, cb, xah, xam, xal);
    }
  };
  auto finish0 = [&](int g, int par) __attribute__((always_inline)) {    // after the barrier that follows produce(g): pack the h0 mask (half 0 only)
    if (tid < WS_ROWS * 8 && half == 0) {
      const int row = tid >> 3, wd = tid & 7;
      const unsigned int* nb = (const unsigned int*)(nbs0 + (par * WS_ROWS + row) * WS_NBP + 8 * wd);
      const unsigned int d0 = nb[0], d1 = nb[1];
      const unsigned int lo16 = (d0 & 0xFu) | ((d0 >> 4) & 0xF0u) | ((d0 >> 8) & 0xF00u) | ((d0 >> 12) & 0xF000u);
      const unsigned int hi16 = (d1 & 0xFu) | ((d1 >> 4) & 0xF0u) | ((d1 >> 8) & 0xF00u) | ((d1 >> 12) & 0xF000u);
      p.mb0[z0 * p.mb0_s0 + z1 * p.mb0_s1 + ((long)g * WS_ROWS + row) * p.mb0_g + wd] = lo16 | (hi16 << 16);
    }
  };
  // the epilogue of one 16 x 16 block of h1 (row block s): bias, ReLU, tail partial sum, 4 mask bits -> LDS
  auto epi_block = [&](const f32x4& a, int g, int par, int s) __attribute__((always_inline)) {
    if constexpr (DG) {                                                  // gradient epilogue: ReLU mask of the receiving activation from its packed bits
      const long m = (long)g * WS_ROWS + 16 * s + li;
      const unsigned int nib = dmg[m * p.dm_g + (ncol1 >> 5)] >> ((ncol1 & 31) + 4 * lq);
      f32x4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = ((nib >> r) & 1u) ? a[r] * inv_sc : 0.f;
      *(f32x4*)&Yg[m * p.y_pitch + ncol1 + 4 * lq] = v;
      return;
    }
    const f32x4 bq = *(const f32x4*)&cst[ncol1 + 4 * lq], twq = *(const f32x4*)&cst[WS_N + ncol1 + 4 * lq];
    f32x4 v = a * inv_sc + bq;
    const unsigned int nib = orl_relu_mask4(v);
    if constexpr (SY) *(f32x4*)&Yg[((long)g * WS_ROWS + 16 * s + li) * p.y_pitch + ncol1 + 4 * lq] = v;
    nbs[(par * WS_ROWS + 16 * s + li) * WS_NBP + 4 * wave + lq] = (unsigned char)nib;
    if constexpr (TQ) {
      float part = (v[0] * twq[0] + v[1] * twq[1]) + (v[2] * twq[2] + v[3] * twq[3]);
      part += __shfl_xor(part, 16);
      part += __shfl_xor(part, 32);
      (qs + (par * WS_NW + wave) * WS_ROWS)[16 * s + li] = part;         // all four lq lanes hold the same sum
    }
  };
  auto finish = [&](int g, int par) __attribute__((always_inline)) {     // after the barrier that follows the epilogue of group g
    if (!DG && tid < WS_ROWS * 4) {                                      // thread (row, word of this half): eight nibbles -> one 32-column mask word
      const int row = tid >> 2, wd = tid & 3, m = g * WS_ROWS + row;
      const unsigned int* nb = (const unsigned int*)(nbs + (par * WS_ROWS + row) * WS_NBP + 8 * wd);
      const unsigned int d0 = nb[0], d1 = nb[1];
      const unsigned int lo16 = (d0 & 0xFu) | ((d0 >> 4) & 0xF0u) | ((d0 >> 8) & 0xF00u) | ((d0 >> 12) & 0xF000u);
      const unsigned int hi16 = (d1 & 0xFu) | ((d1 >> 4) & 0xF0u) | ((d1 >> 8) & 0xF00u) | ((d1 >> 12) & 0xF000u);
      p.mb[z0 * p.mb_s0 + z1 * p.mb_s1 + (long)m * p.mb_g + 4 * half + wd] = lo16 | (hi16 << 16);
    }
    if (TQ && tid >= WS_NT - WS_ROWS) {                                  // eight column-slice partial sums per row, fixed order
      const int row = tid - (WS_NT - WS_ROWS), m = g * WS_ROWS + row;
      const float* q8 = qs + par * WS_NW * WS_ROWS + row;
      float a = tbias;
#pragma unroll
      for (int w = 0; w < WS_NW; ++w) a += q8[w * WS_ROWS];
      tqo[(long)m * tqsm] = a;
    }
  };

  const int g0 = blockIdx.x, gs = gridDim.x;
  if (g0 >= p.groups) return;
  if constexpr (L0) {
    for (int e = tid; e < 2 * WS_ROWS * WF3_XLP; e += WS_NT) Xl[e] = 0.f;  // columns >= x0_pitch stay zero
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
#pragma unroll
    for (int i = 0; i < WS_LD; ++i) load_piece(g0, i);
#pragma unroll
    for (int i = 0; i < WS_LD; ++i) store_piece(0, i);
    if (g0 + gs < p.groups) {
#pragma unroll
      for (int i = 0; i < WS_LD; ++i) load_piece(g0 + gs, i);
    }
    __syncthreads();
  }

  // Software pipeline as in ws_fwd_kernel: iteration `it` multiplies group g out of LDS buffer it & 1 while the epilogue of the PREVIOUS group and
  // the first layer of the NEXT one run, cut into pieces, in the shadow of its MFMAs.
  f32x4 pacc[WS_SUB];
  auto iteration = [&](int g, int it, bool first, bool steady) __attribute__((always_inline)) {
    const int buf = it & 1;
    const hx_t* ah = Ah + (long)buf * 3 * WS_ROWS * WS_PITCH;
    const hx_t* am = ah + WS_ROWS * WS_PITCH;
    const hx_t* al = am + WS_ROWS * WS_PITCH;
    f32x4 acc[WS_SUB];
#pragma unroll
    for (int s = 0; s < WS_SUB; ++s) acc[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
    hx8 fxah, fxam, fxal;
    // steady state, one piece per k step (fenced with that step's 12 MFMAs): 0, 1 the two epilogue blocks of the previous group; 2 .. 5 the four
    // first-layer blocks of the next group; 6 the narrow rows of the group after next
    auto piece = [&](int ks) __attribute__((always_inline)) {
      const int par = (it - 1) & 1;
      if (ks < 2) epi_block(pacc[ks], g - gs, par, ks);
      else if (ks < 6) {
        if constexpr (L0) {
          const int s = (ks - 2) >> 1, cb = (ks - 2) & 1;
          if (cb == 0) prod_x((it + 1) & 1, s, fxah, fxam, fxal);
          prod_block(g + gs, buf ^ 1, (it + 1) & 1, s, cb, fxah, fxam, fxal);
        } else {                                                         // the staging register is refilled right after it was written to LDS
          store_piece(buf ^ 1, ks - 2);
          load_piece(g + 2 * gs, ks - 2);
        }
      } else if (L0 && ks == 6) { storeX(it & 1); loadX(g + 3 * gs); }
    };
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      hx8 fah[WS_SUB], fam[WS_SUB], fal[WS_SUB];
#pragma unroll
      for (int s = 0; s < WS_SUB; ++s) {
        const int o = (16 * s + li) * WS_PITCH + (((4 * ks + lq) ^ li) << 3);
        fah[s] = *(const hx8*)&ah[o]; fam[s] = *(const hx8*)&am[o]; fal[s] = *(const hx8*)&al[o];
      }
      // smallest terms first; operands swapped: D[n][m], lane holds C[m = li][n = 4 lq + r]
#pragma unroll
      for (int s = 0; s < WS_SUB; ++s) acc[s] = ORL_MFMA_16x16x32(bl[ks], fah[s], acc[s]);
#pragma unroll
      for (int s = 0; s < WS_SUB; ++s) acc[s] = ORL_MFMA_16x16x32(bh[ks], fal[s], acc[s]);
#pragma unroll
      for (int s = 0; s < WS_SUB; ++s) acc[s] = ORL_MFMA_16x16x32(bm[ks], fam[s], acc[s]);
#pragma unroll
      for (int s = 0; s < WS_SUB; ++s) acc[s] = ORL_MFMA_16x16x32(bm[ks], fah[s], acc[s]);
#pragma unroll
      for (int s = 0; s < WS_SUB; ++s) acc[s] = ORL_MFMA_16x16x32(bh[ks], fam[s], acc[s]);
#pragma unroll
      for (int s = 0; s < WS_SUB; ++s) acc[s] = ORL_MFMA_16x16x32(bh[ks], fah[s], acc[s]);
      if (steady) {
        piece(ks);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (!steady) {
      if (!first) {
#pragma unroll
        for (int s = 0; s < WS_SUB; ++s) epi_block(pacc[s], g - gs, (it - 1) & 1, s);
      }
      if constexpr (L0) {
        // X rows of group g + gs sit in Xl[(it + 1) & 1] (written one iteration ago); rows of g + 2 gs are in registers
        if (g + gs < p.groups) produce(g + gs, buf ^ 1, (it + 1) & 1, (it + 1) & 1);
        if (g + 2 * gs < p.groups) storeX(it & 1);
        if (g + 3 * gs < p.groups) loadX(g + 3 * gs);
      } else {
        if (g + gs < p.groups) {
#pragma unroll
          for (int i = 0; i < WS_LD; ++i) store_piece(buf ^ 1, i);
        }
        if (g + 2 * gs < p.groups) {
#pragma unroll
          for (int i = 0; i < WS_LD; ++i) load_piece(g + 2 * gs, i);
        }
      }
    }
    __syncthreads();
    if (!first) finish(g - gs, (it - 1) & 1);
    if (L0 && (steady || g + gs < p.groups)) finish0(g + gs, (it + 1) & 1);
#pragma unroll
    for (int s = 0; s < WS_SUB; ++s) pacc[s] = acc[s];
  };
  int g = g0, it = 0;
  iteration(g, it, true, false);
  g += gs; ++it;
  while (g + 3 * gs < p.groups) {
    iteration(g, it, false, true);
    g += gs; ++it;
  }
  while (g < p.groups) {
    iteration(g, it, false, false);
    g += gs; ++it;
  }
  // drain: the last group's epilogue
#pragma unroll
  for (int s = 0; s < WS_SUB; ++s) epi_block(pacc[s], g - gs, (it - 1) & 1, s);
  __syncthreads();
  finish(g - gs, (it - 1) & 1);
}

hipError_t launch_ws_fwd3(WsFwdP p, int nz, int per_z, hipStream_t st) {
  p.groups = p.M / WS_ROWS;
  static const hipError_t attr_err = [] {
    const int lds = (int)ws_fwd3_lds_bytes();
    hipError_t e = hipFuncSetAttribute((const void*)ws_fwd3_kernel<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_fwd3_kernel<true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_fwd3_kernel<false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_fwd3_kernel<true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_fwd3_kernel<false, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_fwd3_kernel<true, false, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_fwd3_kernel<true, true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_fwd3_kernel<false, true, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_fwd3_kernel<false, true, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    return e;
  }();
  if (attr_err != hipSuccess) return attr_err;
  const dim3 grid(per_z, 2, nz), block(WS_NT);
  const size_t lds = ws_fwd3_lds_bytes();
  const bool tq = p.tq != nullptr, sy = p.Y != nullptr, xs = !p.x0_discard;
  if (p.dmask) {                                    // plain dgrad mode
    hipLaunchKernelGGL((ws_fwd3_kernel<false, true, false, false, true>), grid, block, lds, st, p);
    return hipGetLastError();
  }
  if (!p.X0) {                                      // no fused first layer: the input rows are staged from HBM
    if (tq && !sy) hipLaunchKernelGGL((ws_fwd3_kernel<true, false, false, false>), grid, block, lds, st, p);
    else if (tq && sy) hipLaunchKernelGGL((ws_fwd3_kernel<true, true, false, false>), grid, block, lds, st, p);
    else if (!tq && sy) hipLaunchKernelGGL((ws_fwd3_kernel<false, true, false, false>), grid, block, lds, st, p);
    else return hipErrorInvalidValue;
    return hipGetLastError();
  }
  if (tq && !sy && xs) hipLaunchKernelGGL((ws_fwd3_kernel<true, false, true>), grid, block, lds, st, p);
  else if (tq && sy && xs) hipLaunchKernelGGL((ws_fwd3_kernel<true, true, true>), grid, block, lds, st, p);
  else if (!tq && sy && xs) hipLaunchKernelGGL((ws_fwd3_kernel<false, true, true>), grid, block, lds, st, p);
  else if (tq && !sy && !xs) hipLaunchKernelGGL((ws_fwd3_kernel<true, false, false>), grid, block, lds, st, p);
  else if (!tq && sy && !xs) hipLaunchKernelGGL((ws_fwd3_kernel<false, true, false>), grid, block, lds, st, p);
  else return hipErrorInvalidValue;                 // (ws_fwd3_supported refuses every other combination)
  return hipGetLastError();
}

}  // namespace orl
