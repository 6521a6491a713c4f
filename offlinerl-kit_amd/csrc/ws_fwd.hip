// ws_fwd.hip — weight-stationary forward of the 256-wide layers (+ fused first layer, tail, mask bits; plain-dgrad mode) (interface and design notes: ws_gemm.h).
#ifdef WS_FWD_WAVES16      // lab: sixteen waves (16 columns each, four per SIMD, <= 128 VGPRs) for this translation unit only
#define WS_WAVES 16
#endif
#ifdef WS_FWD_WAVES4       // lab: four waves (64 columns each, ONE per SIMD, up to 512 registers): half the A-fragment LDS traffic, no cross-wave overlap
#define WS_WAVES 4
#endif
#include "ws_device.h"

namespace orl {

// F32 = true: exact fp32 arithmetic (v_mfma_f32_16x16x4_f32) with the same structure: the wave's weight slice is 128 VGPRs either
// way (fp32 [256 x 32] vs bf16 hi + lo), the LDS image holds the fp32 rows themselves ([buf][row][256] floats, 16-byte chunks
// XOR-swizzled with the row like the bf16 planes), and a lane's ds_read_b128 of four consecutive k feeds four MFMAs (the k order
// inside a 16-wide step is free as long as the resident weight fragments use the same one).
// SY = false: the activation itself is not stored (TQ only); XS = false (L0 only): the fused first layer's h0 is not stored either
// (forward-only passes, WsFwdP::x0_discard -- a compile-time flavour: the same test at run time costs the product kernel ~20 %)
template <bool TQ, bool L0, bool DG = false, bool SY = true, bool F32 = false, bool XS = true>
__global__ __launch_bounds__(WS_NT) void ws_fwd_kernel(const WsFwdP p) {
  extern __shared__ __attribute__((aligned(16))) float ws_smem[];
  hx_t* Ah = (hx_t*)ws_smem;                                   // [buf][plane][row][WS_PITCH]
  float* Af = ws_smem;                                             // F32: [buf][row][WS_K]
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

  WS_STAMP(0);
  // ---- resident B fragments: lane (li, lq) supplies W[n = ncol0 + 16 cb + li][k = 32 ks + 8 lq .. +7] ----
  // F32: lane (li, lq) supplies W[n][k = 16 t + 4 lq .. + 3], t = 0..15 (element e of the float4 = MFMA k step e of block t)
  hx8 bh[WS_CB][F32 ? 1 : 8], bl[WS_CB][F32 ? 1 : 8];
  f32x4 bw[WS_CB][F32 ? 16 : 1];
  // The layout test (w_sk == 1) sits OUTSIDE the fragment loops and all loads of a layout are issued before the first conversion: with the
  // test inside, every fragment was its own load -> wait -> split round trip (16 - 32 serialized L2 latencies = the ~20 us floor of a launch)
  if constexpr (F32) {
    if (p.w_sk == 1) {
#pragma unroll
      for (int cb = 0; cb < WS_CB; ++cb)
#pragma unroll
        for (int t = 0; t < 16; ++t) bw[cb][t] = *(const f32x4*)(Wg + (long)(ncol0 + 16 * cb + li) * p.w_sn + (16 * t + 4 * lq));
    } else {
#pragma unroll
      for (int cb = 0; cb < WS_CB; ++cb)
#pragma unroll
        for (int t = 0; t < 16; ++t) {
          const float* src = Wg + (long)(ncol0 + 16 * cb + li) * p.w_sn + (long)(16 * t + 4 * lq) * p.w_sk;
#pragma unroll
          for (int j = 0; j < 4; ++j) bw[cb][t][j] = src[(long)j * p.w_sk];
        }
    }
  } else {
    // the resident weights carry the static scale ORL_WSCALE (divided out in the epilogue): hi stays in fp16's normal range
    f32x4 raw[WS_CB][8][2];
    if (p.w_sk == 1) {
#pragma unroll
      for (int cb = 0; cb < WS_CB; ++cb)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          const float* src = Wg + (long)(ncol0 + 16 * cb + li) * p.w_sn + (32 * ks + 8 * lq);
          raw[cb][ks][0] = *(const f32x4*)src; raw[cb][ks][1] = *(const f32x4*)(src + 4);
        }
    } else {                                           // (in, out)-major weights: eight strided loads per fragment, once per workgroup
#pragma unroll
      for (int cb = 0; cb < WS_CB; ++cb)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          const float* src = Wg + (long)(ncol0 + 16 * cb + li) * p.w_sn + (long)(32 * ks + 8 * lq) * p.w_sk;
#pragma unroll
          for (int j = 0; j < 4; ++j) { raw[cb][ks][0][j] = src[(long)j * p.w_sk]; raw[cb][ks][1][j] = src[(long)(4 + j) * p.w_sk]; }
        }
    }
#pragma unroll
    for (int cb = 0; cb < WS_CB; ++cb)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) ws_split8(raw[cb][ks][0] * ORL_WSCALE, raw[cb][ks][1] * ORL_WSCALE, bh[cb][ks], bl[cb][ks]);
  }
  WS_STAMP(1);
  // L0: first-layer fragments of the same columns, K = 32: W0'[n][k] = W0[n][k] (k < in0), b0[n] (k == in0), 0 beyond
  hx8 b0h[WS_CB], b0l[WS_CB];
  f32x4 b0w[WS_CB][2];                              // F32: k = 16 t + 4 lq + e
  if (L0) {
    const float* __restrict__ W0g = p.W0 + z0 * p.w0_s0 + z1 * p.w0_s1;
    const float* __restrict__ b0g = p.b0 + z0 * p.b0_s0 + z1 * p.b0_s1;
#pragma unroll
    for (int cb = 0; cb < WS_CB; ++cb) {
      const int n = ncol0 + 16 * cb + li;
      f32x4 a, b;
      // clamped addresses and 0 / 1 factors, not guarded loads (each guard is a branch with its own wait: twelve serialized round trips
      // per workgroup and problem).  0 * w is exact for finite weights; a diverged run's Inf / NaN would spread into the zero columns.
      const float bn = b0g[n];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k0 = 8 * lq + j, k1 = k0 + 4;
        const float w0 = W0g[(long)n * p.w0_sn + (long)(k0 < p.in0 ? k0 : p.in0 - 1) * p.w0_sk];
        const float w1 = W0g[(long)n * p.w0_sn + (long)(k1 < p.in0 ? k1 : p.in0 - 1) * p.w0_sk];
        a[j] = (k0 < p.in0 ? 1.f : 0.f) * w0 + (k0 == p.in0 ? 1.f : 0.f) * bn;
        b[j] = (k1 < p.in0 ? 1.f : 0.f) * w1 + (k1 == p.in0 ? 1.f : 0.f) * bn;
      }
      if constexpr (!F32) ws_split8(a, b, b0h[cb], b0l[cb]);
      if constexpr (F32) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int k = 16 * t + 4 * lq + j;
            const float w = W0g[(long)n * p.w0_sn + (long)(k < p.in0 ? k : p.in0 - 1) * p.w0_sk];
            b0w[cb][t][j] = (k < p.in0 ? 1.f : 0.f) * w + (k == p.in0 ? 1.f : 0.f) * bn;
          }
      }
    }
  }
  // epilogue constants of this lane's columns n = ncol0 + 16 cb + 4 lq + r sit in LDS (not in 16 VGPRs next to the 128 VGPRs of
  // resident B fragments, and not re-read from global memory: vmcnt is in-order, so waiting for such a load in the epilogue would
  // also wait for every activation store issued before it)
  const float* __restrict__ twg = TQ ? p.tw + z0 * p.tw_s0 + z1 * p.tw_s1 : bg;
  if (!DG && tid < WS_N) { cst[tid] = bg[tid]; cst[WS_N + tid] = twg[tid]; }      // visible after the prologue's barriers
  const float tbias = TQ ? (p.tb + z0 * p.tb_s0 + z1 * p.tb_s1)[0] : 0.f;
  // split precision: the accumulators carry ORL_WSCALE (weights) and, in gradient mode, the run's dynamic gradient scale (staged rows)
  const float a_sc = (DG && !F32 && p.gscale) ? p.gscale[z0] : 1.f;
  const float inv_sc = F32 ? 1.f : 1.0f / (ORL_WSCALE * a_sc);

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
    hx_t* dh = Ah + (long)buf * 2 * WS_ROWS * WS_PITCH;
    hx_t* dl = dh + WS_ROWS * WS_PITCH;
#pragma unroll
    for (int i = i0; i < i1; ++i) {
      const int idx = tid + WS_NT * i, r = idx >> 6, kq = idx & 63;
      if constexpr (F32) {        // float4 #kq of row r at chunk kq ^ (r & 15)
        *(f32x4*)(Af + (long)buf * WS_ROWS * WS_K + r * WS_K + ((kq ^ (r & 15)) << 2)) = st[i];
        continue;
      }
      hx4 h, l;
      if (DG) orl_split4(st[i] * a_sc, h, l); else orl_split4(st[i], h, l);
      // 16-byte chunk c = k / 8 of row r lives at chunk c ^ (r & 15): ds_read_b128 of a fragment column is then conflict-free
      // for the hardware's 16-lane groups (which mix lanes of two neighbouring chunks), and these 8-byte stores stay so too
      const int o = r * WS_PITCH + ((((kq >> 1) ^ (r & 15)) << 3) | ((kq & 1) << 2));
      *(hx4*)(dh + o) = h;
      *(hx4*)(dl + o) = l;
    }
  };

  // ---- L0: narrow-input staging (two elements per thread) and the producer of one h0 row group ----
  const int xe = L0 ? WS_ROWS * p.x0_pitch : 0;
  constexpr int XI = (WS_ROWS * 32 + WS_NT - 1) / WS_NT;              // narrow-input elements per thread (two at 512 threads)
  int xr[XI], xc[XI];
  float sx[XI];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int e = tid + WS_NT * i;
    xr[i] = L0 ? e / (L0 ? p.x0_pitch : 1) : 0; xc[i] = L0 ? e - xr[i] * p.x0_pitch : 0;
    if (L0 && e >= xe) { xr[i] = 0; xc[i] = 32; }       // never read (rows are consumed as 32 columns of the 36-float pitch)
  }
  auto loadX = [&](int g) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < XI; ++i) { const int e = tid + WS_NT * i; sx[i] = X0g[(long)g * xe + (e < xe ? e : xe - 1)]; }   // clamped, not predicated
  };
  // Split-bf16: the narrow rows are split ONCE here, by the two staging threads of an element, into a hi plane (bf16 slots 0..31 of
  // the 144-byte row) and a lo plane (slots 32..63) -- not by every wave in prod_x (eight times the same 32 vector instructions per
  // group, and vector instructions do not overlap with a SIMD's MFMAs).  Surplus threads write the pad slots 64 / 65.
  auto storeX = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const float x = (xc[i] == p.in0) ? 1.0f : sx[i];
      if constexpr (F32) Xl[(buf * WS_ROWS + xr[i]) * WS_XLP + xc[i]] = x;   // surplus threads: pad column 32
      else {
        hx_t* row = (hx_t*)(Xl + (buf * WS_ROWS + xr[i]) * WS_XLP);
        const bool pad = xc[i] >= 32;
        hx_t hh, ll;
        orl_split1(x, hh, ll);
        row[pad ? 64 : xc[i]] = hh;
        row[pad ? 65 : 32 + xc[i]] = ll;
      }
    }
  };
  // produce(g): h0 rows of group g for this wave's columns -> global (fp32), the LDS image `buf` (split bf16), mask nibbles
  f32x4 fx32[2];                                     // F32: the narrow-input fragments of prod_x (k = 16 t + 4 lq + e)
  auto prod_x = [&](int xbuf, int s, hx8& xah, hx8& xal) __attribute__((always_inline)) {
    if constexpr (F32) {
      const float* xrow = Xl + (xbuf * WS_ROWS + 16 * s + li) * WS_XLP + 4 * lq;
      fx32[0] = *(const f32x4*)xrow; fx32[1] = *(const f32x4*)(xrow + 16);
    } else {
      const hx_t* xrow = (const hx_t*)(Xl + (xbuf * WS_ROWS + 16 * s + li) * WS_XLP) + 8 * lq;
      xah = *(const hx8*)xrow; xal = *(const hx8*)(xrow + 32);
    }
  };
  auto prod_block = [&](int g, int buf, int par, int s, int cb, const hx8& xah, const hx8& xal) __attribute__((always_inline)) {
    hx_t* dh = Ah + (long)buf * 2 * WS_ROWS * WS_PITCH;
    hx_t* dl = dh + WS_ROWS * WS_PITCH;
    const int r = 16 * s + li;
    const long m = (long)g * WS_ROWS + r;
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (F32) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) v = __builtin_amdgcn_mfma_f32_16x16x4f32(b0w[cb][t][e], fx32[t][e], v, 0, 0, 0);
    } else {
      v = ORL_MFMA_16x16x32(b0l[cb], xah, v);
      v = ORL_MFMA_16x16x32(b0h[cb], xal, v);
      v = ORL_MFMA_16x16x32(b0h[cb], xah, v);
    }
    const unsigned int nib0 = orl_relu_mask4(v);
    const int k = ncol0 + 16 * cb + 4 * lq;                      // h0 columns k .. k + 3 of row r (lane holds C[m = li][n = 4 lq + j])
    if constexpr (XS) *(f32x4*)&Y0g[m * p.x_pitch + k] = v;
    if constexpr (F32) {
      *(f32x4*)(Af + (long)buf * WS_ROWS * WS_K + r * WS_K + (((k >> 2) ^ (r & 15)) << 2)) = v;
    } else {
    hx4 h, l;
    orl_split4(v, h, l);
    const int o = r * WS_PITCH + ((((k >> 3) ^ (r & 15)) << 3) | (((k >> 2) & 1) << 2));
    *(hx4*)(dh + o) = h;
    *(hx4*)(dl + o) = l;
    }
    nbs0[(par * WS_ROWS + r) * WS_NBP + (k >> 2)] = (unsigned char)nib0;
  };
  auto produce = [&](int g, int buf, int xbuf, int par) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < WS_SUB; ++s) {
      hx8 xah, xal;
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
  WS_STAMP(2);
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
    f32x4 v = F32 ? a + bq : a * inv_sc + bq;
    const unsigned int nib = orl_relu_mask4(v);
    if (storeY) *(f32x4*)&Yg[(long)m * p.y_pitch + ncol0 + 16 * cb + 4 * lq] = v;
    part += (v[0] * twq[0] + v[1] * twq[1]) + (v[2] * twq[2] + v[3] * twq[3]);
    // 4 mask bits of (row 16 s + li, columns ncol0 + 16 cb + 4 lq ..) -> LDS, packed into words after the barrier
    nbs[(par * WS_ROWS + 16 * s + li) * WS_NBP + 4 * WS_CB * wave + 4 * cb + lq] = (unsigned char)nib;
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
      const unsigned int* __restrict__ dm = p.dmask + z0 * p.dm_s0 + z1 * p.dm_s1;
#pragma unroll
      for (int s = 0; s < WS_SUB; ++s) {
        const long m = (long)g * WS_ROWS + 16 * s + li;
        const unsigned int w = dm[m * p.dm_g + (ncol0 >> 5)];              // the 32-column mask word that holds this wave's columns
#pragma unroll
        for (int cb = 0; cb < WS_CB; ++cb) {
          const unsigned int nib = w >> ((ncol0 & 31) + 16 * cb + 4 * lq);
          f32x4 v = acc[s][cb];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = ((nib >> r) & 1u) ? (F32 ? v[r] : v[r] * inv_sc) : 0.f;
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
    const hx_t* ah = Ah + (long)buf * 2 * WS_ROWS * WS_PITCH;
    const hx_t* al = ah + WS_ROWS * WS_PITCH;
    f32x4 acc[WS_SUB][WS_CB];
#pragma unroll
    for (int s = 0; s < WS_SUB; ++s)
#pragma unroll
      for (int cb = 0; cb < WS_CB; ++cb) acc[s][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bool fine = steady && !DG;
    float fpart = 0.f;
    hx8 fxah, fxal;
    // Steady state: the work of the other pipeline stages is cut into eight pieces, one per k step, and fenced together with that
    // step's 12 MFMAs -- the default scheduler otherwise clusters all 96 MFMAs and the matrix pipe idles during the epilogue /
    // staging arithmetic.  k steps 0..3: the four 16 x 16 blocks of the previous group's epilogue; 4..7: the four blocks of the next
    // group's first layer (or the four staging pieces of the plain variant).
    auto piece = [&](int ks) __attribute__((always_inline)) {
      static_assert(WS_SUB == 2 && ((WS_CB == 2 && WS_LD == 4) || (WS_CB == 1 && WS_LD == 2) || (WS_CB == 4 && WS_LD == 8)), "eight (four, sixteen) pieces");
      const int par = (it - 1) & 1;
      if constexpr (WS_CB == 4) {        // four waves: two pieces per k step -- 0..7 the epilogue blocks, 8..15 the first-layer blocks / staging pieces
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int pc = 2 * ks + h;
          if (pc < 8) {
            const int s = pc >> 2, cb = pc & 3;
            if (cb == 0) fpart = 0.f;
            epi_block(pacc[s][cb], g - gs, par, s, cb, fpart);
            if (cb == 3) epi_row(par, s, fpart);
          } else if (L0) {
            const int s = (pc - 8) >> 2, cb = (pc - 8) & 3;
            if (cb == 0) prod_x((it + 1) & 1, s, fxah, fxal);
            prod_block(g + gs, buf ^ 1, (it + 1) & 1, s, cb, fxah, fxal);
          } else {
            store_group(buf ^ 1, stn, pc - 8, pc - 7);
            load_piece(g + 2 * gs, stn, pc - 8);
          }
        }
        return;
      }
      if constexpr (WS_CB == 1) {        // sixteen waves: two epilogue blocks (k steps 0, 2), two first-layer blocks / staging pieces (4, 6)
        if (ks == 0 || ks == 2) {
          const int s = ks >> 1;
          fpart = 0.f;
          epi_block(pacc[s][0], g - gs, par, s, 0, fpart);
          epi_row(par, s, fpart);
        } else if (ks == 4 || ks == 6) {
          const int s = (ks - 4) >> 1;
          if (L0) {
            prod_x((it + 1) & 1, s, fxah, fxal);
            prod_block(g + gs, buf ^ 1, (it + 1) & 1, s, 0, fxah, fxal);
          } else {
            store_group(buf ^ 1, stn, s, s + 1);
            load_piece(g + 2 * gs, stn, s);
          }
        }
        return;
      }
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
    };
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      // the three products of a block are issued plane by plane (lo*hi, hi*lo, hi*hi over all four blocks) so that dependent MFMAs
      // on one accumulator are four instructions apart; operands swapped: D[n][m], lane holds C[m = li][n = 4 lq + r]
      // (fetching the fragments one k step ahead, behind the MFMAs that free their registers, measured no different: r02 A/B)
      if constexpr (F32) {
        // 32 k per step = two float4 chunks per lane and row block, 8 MFMA k steps x 4 accumulator blocks
        const float* af = Af + (long)buf * WS_ROWS * WS_K;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          f32x4 fa[WS_SUB];
#pragma unroll
          for (int s = 0; s < WS_SUB; ++s) fa[s] = *(const f32x4*)&af[(16 * s + li) * WS_K + (((8 * ks + 4 * tt + lq) ^ li) << 2)];
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int s = 0; s < WS_SUB; ++s)
#pragma unroll
              for (int cb = 0; cb < WS_CB; ++cb)
                acc[s][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[cb][2 * ks + tt][e], fa[s][e], acc[s][cb], 0, 0, 0);
        }
      } else {
      hx8 fah2[WS_SUB], fal2[WS_SUB];
#pragma unroll
      for (int s = 0; s < WS_SUB; ++s) {
        const int o = (16 * s + li) * WS_PITCH + (((4 * ks + lq) ^ li) << 3);
        fah2[s] = *(const hx8*)&ah[o]; fal2[s] = *(const hx8*)&al[o];
      }
#pragma unroll
      for (int s = 0; s < WS_SUB; ++s)
#pragma unroll
        for (int cb = 0; cb < WS_CB; ++cb) acc[s][cb] = ORL_MFMA_16x16x32(bl[cb][ks], fah2[s], acc[s][cb]);
#pragma unroll
      for (int s = 0; s < WS_SUB; ++s)
#pragma unroll
        for (int cb = 0; cb < WS_CB; ++cb) acc[s][cb] = ORL_MFMA_16x16x32(bh[cb][ks], fal2[s], acc[s][cb]);
#pragma unroll
      for (int s = 0; s < WS_SUB; ++s)
#pragma unroll
        for (int cb = 0; cb < WS_CB; ++cb) acc[s][cb] = ORL_MFMA_16x16x32(bh[cb][ks], fah2[s], acc[s][cb]);
      }
      if (fine) {
        piece(ks);
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
  WS_STAMP(3);
  iteration(g, it, st0, true, false);
  WS_STAMP(4);
  g += gs; ++it;
  while (g + 3 * gs < p.groups) {
    iteration(g, it, st0, false, true);
    g += gs; ++it;
  }
  while (g < p.groups) {
    iteration(g, it, st0, false, false);
    g += gs; ++it;
  }
  WS_STAMP(5);
  // drain: the last group's epilogue
  epilogue(pacc, g - gs, (it - 1) & 1);
  __syncthreads();
  finish(g - gs, (it - 1) & 1);
  WS_STAMP(6);
}

template <bool F32>
static hipError_t ws_fwd_attrs() {
  const int big = (int)ws_fwd_lds_bytes(true);
  hipError_t e = hipFuncSetAttribute((const void*)ws_fwd_kernel<true, false, false, true, F32>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_fwd_kernel<false, false, false, true, F32>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_fwd_kernel<true, true, false, true, F32>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_fwd_kernel<false, true, false, true, F32>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_fwd_kernel<false, false, true, true, F32>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_fwd_kernel<true, true, false, false, F32>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_fwd_kernel<true, false, false, false, F32>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_fwd_kernel<true, true, false, false, F32, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_fwd_kernel<false, true, false, true, F32, false>, hipFuncAttributeMaxDynamicSharedMemorySize, big);
  return e;
}

template <bool F32>
static void ws_fwd_dispatch(const WsFwdP& p, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
  const bool l0 = p.X0 != nullptr;
  if (p.dmask) hipLaunchKernelGGL((ws_fwd_kernel<false, false, true, true, F32>), grid, block, lds, st, p);
  else if (l0 && p.x0_discard && !(p.tq && p.Y)) {     // forward-only passes (storing h0 anyway is always correct: any other shape takes the storing flavour)
    if (p.tq) hipLaunchKernelGGL((ws_fwd_kernel<true, true, false, false, F32, false>), grid, block, lds, st, p);
    else hipLaunchKernelGGL((ws_fwd_kernel<false, true, false, true, F32, false>), grid, block, lds, st, p);
  } else if (l0) {
    if (p.tq && !p.Y) hipLaunchKernelGGL((ws_fwd_kernel<true, true, false, false, F32>), grid, block, lds, st, p);
    else if (p.tq) hipLaunchKernelGGL((ws_fwd_kernel<true, true, false, true, F32>), grid, block, lds, st, p);
    else hipLaunchKernelGGL((ws_fwd_kernel<false, true, false, true, F32>), grid, block, lds, st, p);
  } else {
    if (p.tq && !p.Y) hipLaunchKernelGGL((ws_fwd_kernel<true, false, false, false, F32>), grid, block, lds, st, p);
    else if (p.tq) hipLaunchKernelGGL((ws_fwd_kernel<true, false, false, true, F32>), grid, block, lds, st, p);
    else hipLaunchKernelGGL((ws_fwd_kernel<false, false, false, true, F32>), grid, block, lds, st, p);
  }
}

hipError_t launch_ws_fwd(WsFwdP p, int nz, hipStream_t st, const WsGeom& geo) {
  p.groups = (p.M + WS_ROWS - 1) / WS_ROWS;
  // one workgroup per CU (register-resident weights): whole rounds of 256 workgroups over the nz problems (ws_blocks_per_problem)
  const int per_z = ws_blocks_per_problem(p.groups, nz, 10, 1 << 20, geo);
  const size_t lds = ws_fwd_lds_bytes(p.X0 != nullptr);
  static const hipError_t attr_err = [] {       // thread-safe one-time initialisation (engines may launch from several host threads)
    hipError_t e = ws_fwd_attrs<false>();
    return e == hipSuccess ? ws_fwd_attrs<true>() : e;
  }();
  if (attr_err != hipSuccess) return attr_err;
  const dim3 grid(per_z, 1, nz), block(WS_NT);
  if (p.f32) ws_fwd_dispatch<true>(p, grid, block, lds, st);
  else ws_fwd_dispatch<false>(p, grid, block, lds, st);
  return hipGetLastError();
}

}  // namespace orl
