// ws_dgrad.hip — top-layer dgrad from mask bits fused with the layer-0 weight gradient (interface and design notes: ws_gemm.h).
#include "ws_device.h"

namespace orl {

// PLAIN: the incoming gradient is a materialised matrix (WsDgradP::Z) -- A image = its hi and lo planes, B' = W1, no dq in the epilogue
template <bool W0, bool STORE, bool PLAIN = false>
__global__ __launch_bounds__(WS_NT) void ws_dgrad_w0_kernel(const WsDgradP p) {
  static_assert(WS_NW == 8 && WS_ROWS == 32, "one 32-column mask word per wave, 32-row groups");
  static_assert(!PLAIN || (W0 && !STORE), "the plain variant exists for the fused layer-0 gradient only");
  constexpr int APL = PLAIN ? 2 : 1;                                // planes of the A image
  extern __shared__ __attribute__((aligned(16))) float ws_smem[];
  hx_t* Ah = (hx_t*)ws_smem;                                   // [buf][row][256] 0/1 mask as bf16, swizzled   (PLAIN: [buf][hi, lo][row][256])
  hx_t* XT = Ah + 2 * APL * WS_ROWS * WS_PITCH;                  // [buf][hi, lo][c = 32][WD_XP]: X^T of the row group
  float* EO = (float*)(XT + 2 * 2 * 32 * WD_XP);                   // [buf][dq[32] | h0 mask words [wave = 8][row = 32]]: epilogue operands
  __shared__ u32x2_t mlut[16];                                     // 4 mask bits -> 4 bf16 values
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lq = lane >> 4;
  const int z = blockIdx.z, z0 = z / p.nz1, z1 = z - z0 * p.nz1;
  const unsigned int* __restrict__ ab = PLAIN ? nullptr : p.abits + z0 * p.ab_s0 + z1 * p.ab_s1;
  const unsigned int* __restrict__ xb = p.xbits + z0 * p.xb_s0 + z1 * p.xb_s1;
  const float* __restrict__ dqg = PLAIN ? nullptr : p.dq + z0 * p.dq_s0 + z1 * p.dq_s1;
  const float* __restrict__ Wg = p.W + z0 * p.w_s0 + z1 * p.w_s1;
  const float* __restrict__ wtg = PLAIN ? nullptr : p.wt + z0 * p.wt_s0 + z1 * p.wt_s1;
  const float* __restrict__ Xg = p.X + z0 * p.x_s0 + z1 * p.x_s1;
  const float* __restrict__ Zg = PLAIN ? p.Z + z0 * p.z_s0 + z1 * p.z_s1 : nullptr;
  const int ncol0 = 32 * wave;
  // split precision: dq (PLAIN: dz1) enters scaled by the run's dynamic gradient scale (W0 variant only: the stored dz0 of the STORE variant
  // must be the true values), the resident products by ORL_WWSCALE (PLAIN: ORL_WSCALE); dz0 therefore carries gs * that scale into the
  // second MFMA stage
  const float gsc = (W0 && p.gscale) ? p.gscale[z0] : 1.f;
  constexpr float BSC = PLAIN ? ORL_WSCALE : ORL_WWSCALE;
  const float dq_sc = W0 ? gsc : 1.0f / ORL_WWSCALE;                // factor applied to dq when it is staged
  const float out_inv = 1.0f / (gsc * BSC);                          // W0: applied to the dW0 / db0 slab

  // resident B' fragments: lane (li, lq) supplies B'[k = 32 ks + 8 lq + j][n = ncol0 + 16 cb + li] = w_tail[k] * W1[k][n]
  hx8 bh[2][8], bl[2][8];
  {
    // all 128 loads of the lane are issued before the first conversion (one exposed memory latency instead of sixteen)
    f32x4 raw[2][8][2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int n = ncol0 + 16 * cb + li, k0 = 32 * ks + 8 * lq;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          raw[cb][ks][0][j] = Wg[(long)n * p.w_sn + (long)(k0 + j) * p.w_sk];
          raw[cb][ks][1][j] = Wg[(long)n * p.w_sn + (long)(k0 + 4 + j) * p.w_sk];
        }
      }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const int k0 = 32 * ks + 8 * lq;
        f32x4 t0 = (f32x4){1.f, 1.f, 1.f, 1.f}, t1 = t0;
        if (!PLAIN) { t0 = *(const f32x4*)&wtg[k0]; t1 = *(const f32x4*)&wtg[k0 + 4]; }
        // static scale BSC, divided out below
        ws_split8((t0 * BSC) * raw[cb][ks][0], (t1 * BSC) * raw[cb][ks][1], bh[cb][ks], bl[cb][ks]);
      }
  }
  // zero both X^T images once (rows c >= x_pitch are never written again)
  if (W0) for (int e = tid; e < 2 * 2 * 32 * WD_XP / 2; e += WS_NT) ((unsigned int*)XT)[e] = 0u;
  if (tid < 16) mlut[tid] = (u32x2_t){((tid & 1u) | ((tid & 2u) << 15)) * ORL_HX_ONE_BITS, (((tid >> 2) & 1u) | ((tid & 8u) << 13)) * ORL_HX_ONE_BITS};
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
  f32x4 sz[PLAIN ? 4 : 1];                                          // PLAIN: the thread's four pieces of the dz1 row group (row (tid >> 6) + 8 i, columns 4 (tid & 63) ..)
  const unsigned int vo_z = PLAIN ? (unsigned int)((tid >> 6) * p.z_pitch + 4 * (tid & 63)) : 0u;
  // global addresses = scalar row-group base + per-thread offset computed once (no 64-bit vector multiplies in the loop)
  const unsigned int vo_ab = (unsigned int)((tid >> 4) * p.ab_g + ((tid & 15) >> 1)), vo_dq = (unsigned int)((tid & 31) * (int)p.dq_sm);
  const unsigned int vo_xb = (unsigned int)(((tid >> 3) & 31) * p.xb_g + (tid & 7));
  unsigned int vo_x[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { const int e = tid + WS_NT * i; vo_x[i] = (unsigned int)(e < xe ? e : (xe > 0 ? xe - 1 : 0)); }   // clamped, not predicated
  auto load_group = [&](int g) __attribute__((always_inline)) {
    const long row0 = (long)g * WS_ROWS;
    if (PLAIN) {
#pragma unroll
      for (int i = 0; i < 4; ++i) sz[i] = *(const f32x4*)&(Zg + (row0 + 8 * i) * p.z_pitch)[vo_z];
    } else {
      sm_word = (ab + row0 * p.ab_g)[vo_ab];
      // the epilogue's dq and h0 mask words travel through LDS with the group (fetched a full iteration ahead by the staging threads:
      // the epilogue then has no global loads of its own to wait for)
      sdq = (dqg + row0 * p.dq_sm)[vo_dq];
    }
    sxw = (xb + row0 * p.xb_g)[vo_xb];
    if (W0) {
#pragma unroll
      for (int i = 0; i < 2; ++i) sx[i] = (Xg + (long)g * xe)[vo_x[i]];
    }
  };
  auto store_group = [&](int buf) __attribute__((always_inline)) {
    float* eo = EO + buf * (WS_ROWS + WS_NW * WS_ROWS);
    if (PLAIN) {
      hx_t* dh = Ah + (long)buf * 2 * WS_ROWS * WS_PITCH;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = (tid >> 6) + 8 * i, kq = tid & 63;              // columns 4 kq ..: half (kq & 1) of the 16-byte chunk kq >> 1
        hx4 h, l;
        orl_split4(sz[i] * gsc, h, l);
        const int o = r * WS_PITCH + ((((kq >> 1) ^ (r & 15)) << 3) | ((kq & 1) << 2));
        *(hx4*)(dh + o) = h;
        *(hx4*)(dh + WS_ROWS * WS_PITCH + o) = l;
      }
    } else {
    const int r = tid >> 4, hw = tid & 15;
    const unsigned int bits = (sm_word >> (16 * (hw & 1))) & 0xFFFFu;
    u32x4 c0, c1;                                                    // 16 bf16 values: 1.0 = 0x3F80 where the bit is set
    // through a 16-entry LDS table (4 bits -> 4 bf16): four LDS reads instead of ~32 vector instructions (see ws_wgrad_kernel)
    const u32x2_t q0 = mlut[bits & 15u], q1 = mlut[(bits >> 4) & 15u], q2 = mlut[(bits >> 8) & 15u], q3 = mlut[bits >> 12];
    c0 = (u32x4){q0[0], q0[1], q1[0], q1[1]};
    c1 = (u32x4){q2[0], q2[1], q3[0], q3[1]};
    hx_t* d = Ah + (long)buf * WS_ROWS * WS_PITCH + r * WS_PITCH;
    *(u32x4*)(d + (((2 * hw) ^ (r & 15)) << 3)) = c0;
    *(u32x4*)(d + (((2 * hw + 1) ^ (r & 15)) << 3)) = c1;
    eo[tid & 31] = sdq * dq_sc;                                      // (replicated writes of identical values)
    }
    ((unsigned int*)eo)[WS_ROWS + (tid & 7) * WS_ROWS + ((tid >> 3) & 31)] = sxw;
    hx_t* xt = XT + (long)buf * 2 * 32 * WD_XP;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (W0) {
        const float x = (xo[i] >> 16) ? 1.0f : sx[i];
        hx_t hh, ll;
        orl_split1(x, hh, ll);
        xt[xo[i] & 0xFFFF] = hh;
        xt[32 * WD_XP + (xo[i] & 0xFFFF)] = ll;
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
      dq4[s] = PLAIN ? (f32x4){1.f, 1.f, 1.f, 1.f} : *(const f32x4*)&eo[16 * s + 4 * lq];
      const u32x4 w4 = *(const u32x4*)&((const unsigned int*)eo)[WS_ROWS + wave * WS_ROWS + 16 * s + 4 * lq];
#pragma unroll
      for (int r = 0; r < 4; ++r) xw[s][r] = w4[r];
    }
    const hx_t* ah = Ah + (long)buf * APL * WS_ROWS * WS_PITCH;
    f32x4 acc[WS_SUB][2];
#pragma unroll
    for (int s = 0; s < WS_SUB; ++s)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) acc[s][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
      for (int s = 0; s < WS_SUB; ++s) {
        const hx8 fa = *(const hx8*)&ah[(16 * s + li) * WS_PITCH + (((4 * ks + lq) ^ li) << 3)];
        hx8 fl;
        if (PLAIN) fl = *(const hx8*)&ah[WS_ROWS * WS_PITCH + (16 * s + li) * WS_PITCH + (((4 * ks + lq) ^ li) << 3)];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {               // D[m][n]: lane holds rows 4 lq + r of column li
          acc[s][cb] = ORL_MFMA_16x16x32(fa, bl[cb][ks], acc[s][cb]);
          if (PLAIN) acc[s][cb] = ORL_MFMA_16x16x32(fl, bh[cb][ks], acc[s][cb]);
          acc[s][cb] = ORL_MFMA_16x16x32(fa, bh[cb][ks], acc[s][cb]);
        }
      }
    }
    // dz0 block -> (hi, lo) bf16 B operand of the 16x16x16 MFMA; A operand = X^T rows c, columns m = 16 s + 4 lq ..
    const hx_t* xth = XT + (long)buf * 2 * 32 * WD_XP;
    const hx_t* xtl = xth + 32 * WD_XP;
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
        hx4 zh, zl;
        // (element by element: gathering the four values first and splitting them with orl_split4 measured 15 % slower here)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = ((xw[s][r] >> (16 * cb + li)) & 1u) ? (PLAIN ? acc[s][cb][r] : acc[s][cb][r] * dq4[s][r]) : 0.f;
          if (STORE) Cg[(long)(g * WS_ROWS + 16 * s + 4 * lq + r) * p.c_pitch + ncol0 + 16 * cb + li] = v;
          hx_t hh, ll;
          orl_split1(v, hh, ll);
          zh[r] = hh; zl[r] = ll;
        }
        const s16x4 bzh = *(const s16x4*)&zh, bzl = *(const s16x4*)&zl;
#pragma unroll
        for (int cbk = 0; cbk < 2 && W0; ++cbk) {
          d2[cbk][cb] = ORL_MFMA_16x16x16(xl[cbk], bzh, d2[cbk][cb]);
          d2[cbk][cb] = ORL_MFMA_16x16x16(xh[cbk], bzl, d2[cbk][cb]);
          d2[cbk][cb] = ORL_MFMA_16x16x16(xh[cbk], bzh, d2[cbk][cb]);
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
        if (c < p.in0) wo[(long)n * p.o_sr + (long)c * p.o_sc] = d2[cbk][cb][r] * out_inv;
        else if (c == p.in0) bo[n] = d2[cbk][cb][r] * out_inv;
      }
}

// ---- exact-fp32 flavour (precision 0): same structure on v_mfma_f32_16x16x4_f32 ----
// Resident B'[k][n] = w_tail[k] W1[k][n] as fp32 (lane (li, lq): n = ncol0 + 16 cb + li, k = 16 t + 4 lq + e: 128 VGPRs), the 0/1
// mask of h1 expanded from its bits into an fp32 LDS image (16-byte chunks XOR-swizzled with the row, as in ws_fwd's fp32 image;
// a lane's ds_read_b128 = four MFMA k steps), and dW0^T += X^T dz0 on the same instruction with the accumulators as B operand
// (k step r = the lane's row 4 lq + r) and X^T rows read as float4 from an fp32 image.
enum { WD32_XP = WS_ROWS + 4 };                                      // float pitch of an X^T row
static constexpr size_t ws_dgrad32_lds_bytes() {
  return sizeof(float) * ((size_t)2 * WS_ROWS * WS_K + (size_t)2 * 32 * WD32_XP + (size_t)2 * (WS_ROWS + WS_NW * WS_ROWS));
}

template <bool W0, bool STORE, bool PLAIN = false>                   // PLAIN: A image = the materialised dz1 rows (WsDgradP::Z), B' = W1
__global__ __launch_bounds__(WS_NT) void ws_dgrad32_w0_kernel(const WsDgradP p) {
  static_assert(WS_NW == 8 && WS_ROWS == 32, "one 32-column mask word per wave, 32-row groups");
  static_assert(!PLAIN || (W0 && !STORE), "the plain variant exists for the fused layer-0 gradient only");
  extern __shared__ __attribute__((aligned(16))) float ws_smem[];
  float* Am = ws_smem;                                              // [buf][row][256] 0/1 mask as fp32, swizzled   (PLAIN: the dz1 values)
  float* XT = Am + 2 * WS_ROWS * WS_K;                              // [buf][c = 32][WD32_XP]: X^T of the row group
  float* EO = XT + 2 * 32 * WD32_XP;                                // [buf][dq[32] | h0 mask words [wave = 8][row = 32]]
  __shared__ f32x4 mlut32[16];                                      // 4 mask bits -> 4 floats (0.0 / 1.0)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lq = lane >> 4;
  const int z = blockIdx.z, z0 = z / p.nz1, z1 = z - z0 * p.nz1;
  const unsigned int* __restrict__ ab = PLAIN ? nullptr : p.abits + z0 * p.ab_s0 + z1 * p.ab_s1;
  const unsigned int* __restrict__ xb = p.xbits + z0 * p.xb_s0 + z1 * p.xb_s1;
  const float* __restrict__ dqg = PLAIN ? nullptr : p.dq + z0 * p.dq_s0 + z1 * p.dq_s1;
  const float* __restrict__ Wg = p.W + z0 * p.w_s0 + z1 * p.w_s1;
  const float* __restrict__ wtg = PLAIN ? nullptr : p.wt + z0 * p.wt_s0 + z1 * p.wt_s1;
  const float* __restrict__ Xg = p.X + z0 * p.x_s0 + z1 * p.x_s1;
  const float* __restrict__ Zg = PLAIN ? p.Z + z0 * p.z_s0 + z1 * p.z_s1 : nullptr;
  const int ncol0 = 32 * wave;

  f32x4 bw[2][16];
#pragma unroll
  for (int cb = 0; cb < 2; ++cb)                                    // (all loads first, then the w_tail products: see ws_dgrad_w0_kernel)
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int n = ncol0 + 16 * cb + li, k0 = 16 * t + 4 * lq;
#pragma unroll
      for (int j = 0; j < 4; ++j) bw[cb][t][j] = Wg[(long)n * p.w_sn + (long)(k0 + j) * p.w_sk];
    }
  if (!PLAIN) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int t = 0; t < 16; ++t) bw[cb][t] *= *(const f32x4*)&wtg[16 * t + 4 * lq];
  }
  if (W0) for (int e = tid; e < 2 * 32 * WD32_XP; e += WS_NT) XT[e] = 0.f;      // rows c >= x_pitch are never written again
  if (tid < 16) mlut32[tid] = (f32x4){(float)(tid & 1), (float)((tid >> 1) & 1), (float)((tid >> 2) & 1), (float)(tid >> 3)};
  __syncthreads();

  unsigned int sm_word;
  float sx[2];
  const int xe = W0 ? WS_ROWS * p.x_pitch : 0;
  float* __restrict__ Cg = STORE ? p.C + z0 * p.c_s0 + z1 * p.c_s1 : nullptr;
  int xo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + WS_NT * i;
    int rr = W0 ? e / (W0 ? p.x_pitch : 1) : 0, c = W0 ? e - rr * p.x_pitch : 0;
    if (e >= xe) { rr = 32; c = 0; }                                 // pad slot, never read
    xo[i] = ((c == p.in0) ? (1 << 16) : 0) | (c * WD32_XP + rr);
  }
  float sdq;
  unsigned int sxw;
  f32x4 sz[PLAIN ? 4 : 1];                                          // PLAIN: row (tid >> 6) + 8 i, columns 4 (tid & 63) .. of the dz1 row group
  const unsigned int vo_z = PLAIN ? (unsigned int)((tid >> 6) * p.z_pitch + 4 * (tid & 63)) : 0u;
  const unsigned int vo_ab = (unsigned int)((tid >> 4) * p.ab_g + ((tid & 15) >> 1)), vo_dq = (unsigned int)((tid & 31) * (int)p.dq_sm);
  const unsigned int vo_xb = (unsigned int)(((tid >> 3) & 31) * p.xb_g + (tid & 7));
  unsigned int vo_x[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) { const int e = tid + WS_NT * i; vo_x[i] = (unsigned int)(e < xe ? e : (xe > 0 ? xe - 1 : 0)); }
  auto load_group = [&](int g) __attribute__((always_inline)) {      // scalar row-group base + per-thread offset (see ws_dgrad_w0_kernel)
    const long row0 = (long)g * WS_ROWS;
    if (PLAIN) {
#pragma unroll
      for (int i = 0; i < 4; ++i) sz[i] = *(const f32x4*)&(Zg + (row0 + 8 * i) * p.z_pitch)[vo_z];
    } else {
      sm_word = (ab + row0 * p.ab_g)[vo_ab];
      sdq = (dqg + row0 * p.dq_sm)[vo_dq];
    }
    sxw = (xb + row0 * p.xb_g)[vo_xb];
    if (W0) {
#pragma unroll
      for (int i = 0; i < 2; ++i) sx[i] = (Xg + (long)g * xe)[vo_x[i]];
    }
  };
  auto store_group = [&](int buf) __attribute__((always_inline)) {
    float* eo = EO + buf * (WS_ROWS + WS_NW * WS_ROWS);
    if (PLAIN) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = (tid >> 6) + 8 * i, kq = tid & 63;              // chunk kq = columns 4 kq ..
        *(f32x4*)(Am + (long)buf * WS_ROWS * WS_K + r * WS_K + ((kq ^ (r & 15)) << 2)) = sz[i];
      }
    } else {
    const int r = tid >> 4, hw = tid & 15;
    const unsigned int bits = (sm_word >> (16 * (hw & 1))) & 0xFFFFu;
    float* d = Am + (long)buf * WS_ROWS * WS_K + r * WS_K;
#pragma unroll
    for (int j = 0; j < 4; ++j)                                      // chunk 4 hw + j = columns 16 hw + 4 j ..; 4 bits -> 4 floats: LDS table
      *(f32x4*)(d + (((4 * hw + j) ^ (r & 15)) << 2)) = mlut32[(bits >> (4 * j)) & 15u];
    eo[tid & 31] = sdq;
    }
    ((unsigned int*)eo)[WS_ROWS + (tid & 7) * WS_ROWS + ((tid >> 3) & 31)] = sxw;
    float* xt = XT + (long)buf * 32 * WD32_XP;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (W0) xt[xo[i] & 0xFFFF] = (xo[i] >> 16) ? 1.0f : sx[i];
  };

  f32x4 d2[2][2];
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
  auto iteration = [&](int g, int it) __attribute__((always_inline)) {
    const int buf = it & 1;
    const float* am = Am + (long)buf * WS_ROWS * WS_K;
    f32x4 acc[WS_SUB][2];
#pragma unroll
    for (int s = 0; s < WS_SUB; ++s)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) acc[s][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // (fenced per step: the scheduler otherwise hoists all 32 fragment reads of the unrolled loop to its top, 128 more live VGPRs
    // next to the 128 of the resident weights; the SIMD's other wave covers the read latency)
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      f32x4 fa[WS_SUB];
#pragma unroll
      for (int s = 0; s < WS_SUB; ++s) fa[s] = *(const f32x4*)&am[(16 * s + li) * WS_K + (((4 * t + lq) ^ li) << 2)];
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int s = 0; s < WS_SUB; ++s)
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)               // D[m][n]: lane holds rows 4 lq + r of column li
            acc[s][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[s][e], bw[cb][t][e], acc[s][cb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // epilogue operands of this group (staged with it): dq of the lane's 4 rows per 16-row block, the h0 mask word of those rows
    const float* eo = EO + buf * (WS_ROWS + WS_NW * WS_ROWS);
    const float* xt = XT + (long)buf * 32 * WD32_XP;
#pragma unroll
    for (int s = 0; s < WS_SUB; ++s) {
      const f32x4 dq4s = PLAIN ? (f32x4){1.f, 1.f, 1.f, 1.f} : *(const f32x4*)&eo[16 * s + 4 * lq];
      const u32x4 xws = *(const u32x4*)&((const unsigned int*)eo)[WS_ROWS + wave * WS_ROWS + 16 * s + 4 * lq];
      f32x4 xa[2];
      if (W0) {
#pragma unroll
        for (int cbk = 0; cbk < 2; ++cbk) xa[cbk] = *(const f32x4*)&xt[(16 * cbk + li) * WD32_XP + 16 * s + 4 * lq];
      }
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        f32x4 v;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = ((xws[r] >> (16 * cb + li)) & 1u) ? (PLAIN ? acc[s][cb][r] : acc[s][cb][r] * dq4s[r]) : 0.f;
          if (STORE) Cg[(long)(g * WS_ROWS + 16 * s + 4 * lq + r) * p.c_pitch + ncol0 + 16 * cb + li] = v[r];
        }
#pragma unroll
        for (int cbk = 0; cbk < 2 && W0; ++cbk)
#pragma unroll
          for (int r = 0; r < 4; ++r) d2[cbk][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[cbk][r], v[r], d2[cbk][cb], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);             // one block at a time: interleaving the four blocks costs ~50 spilled VGPRs
      }
    }
    if (g + gs < p.groups) store_group(buf ^ 1);
    if (g + 2 * gs < p.groups) load_group(g + 2 * gs);
    __syncthreads();
  };
  int g = g0, it = 0;
  for (; g < p.groups; g += gs, ++it) iteration(g, it);
  if (!W0) return;
  float* wo = p.w0_out + z0 * p.o_s0 + z1 * p.o_s1 + (long)blockIdx.x * p.o_ks;
  float* bo = p.b0_out + z0 * p.o_s0 + z1 * p.ob_s1 + (long)blockIdx.x * p.o_ks;
#pragma unroll
  for (int cbk = 0; cbk < 2; ++cbk)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 16 * cbk + 4 * lq + r, n = ncol0 + 16 * cb + li;
        if (c < p.in0) wo[(long)n * p.o_sr + (long)c * p.o_sc] = d2[cbk][cb][r];
        else if (c == p.in0) bo[n] = d2[cbk][cb][r];
      }
}

hipError_t launch_ws_dgrad_w0(WsDgradP p, int nz, int per_z, hipStream_t st) {
  p.groups = p.M / WS_ROWS;
  const dim3 grid(per_z, 1, nz), block(WS_NT);
  if (p.f32) {
    static const hipError_t attr_err = [] {
      hipError_t e = hipFuncSetAttribute((const void*)ws_dgrad32_w0_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ws_dgrad32_lds_bytes());
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_dgrad32_w0_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ws_dgrad32_lds_bytes());
      if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_dgrad32_w0_kernel<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ws_dgrad32_lds_bytes());
      return e;
    }();
    if (attr_err != hipSuccess) return attr_err;
    if (p.Z) hipLaunchKernelGGL((ws_dgrad32_w0_kernel<true, false, true>), grid, block, ws_dgrad32_lds_bytes(), st, p);
    else if (p.w0_out) hipLaunchKernelGGL((ws_dgrad32_w0_kernel<true, false>), grid, block, ws_dgrad32_lds_bytes(), st, p);
    else hipLaunchKernelGGL((ws_dgrad32_w0_kernel<false, true>), grid, block, ws_dgrad32_lds_bytes(), st, p);
    return hipGetLastError();
  }
  if (p.Z) {
    static const hipError_t attr_err = hipFuncSetAttribute((const void*)ws_dgrad_w0_kernel<true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ws_dgrad_lds_bytes(true));
    if (attr_err != hipSuccess) return attr_err;
    hipLaunchKernelGGL((ws_dgrad_w0_kernel<true, false, true>), grid, block, ws_dgrad_lds_bytes(true), st, p);
    return hipGetLastError();
  }
  if (p.w0_out) hipLaunchKernelGGL((ws_dgrad_w0_kernel<true, false>), grid, block, ws_dgrad_lds_bytes(), st, p);
  else hipLaunchKernelGGL((ws_dgrad_w0_kernel<false, true>), grid, block, ws_dgrad_lds_bytes(), st, p);
  return hipGetLastError();
}

}  // namespace orl
