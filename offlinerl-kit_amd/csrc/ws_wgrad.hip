// ws_wgrad.hip — output-stationary top-layer weight gradient (+ tail-layer gradients) (interface and design notes: ws_gemm.h).
#include "ws_device.h"

namespace orl {

// bf16 offset of the 8-byte piece (16-byte chunk `chunk`, half `half`) of row r: chunks are XOR-swizzled with 2 (r & 7) so that
// the transposed reads (8 rows x 32 B per 32-lane half) and the 8-byte staging stores are both bank-conflict free
// (a padded row pitch of 272 elements instead of the swizzle makes every fragment address lane part + immediate, but measured 2 %
// slower at 128 runs: 581 vs 569 us)
__device__ inline int ww_off(int r, int chunk, int half) { return r * WS_K + ((chunk ^ (2 * (r & 7))) << 3) + (half << 2); }

__device__ inline s16x4 ww_tr(const hx_t* img, int row0, int col0, int lane) {
  // 16-lane group lq reads rows row0 + 4 lq + q (q = li >> 2), columns col0 + 4 (li & 3) ..; lane li receives column col0 + li of
  // rows row0 + 4 lq .. + 3  (= the 16x16x16 MFMA operand layout, for A as the transpose of the image)
  const int li = lane & 15, lq = lane >> 4, row = row0 + 4 * lq + (li >> 2), col = col0 + 4 * (li & 3);
  const hx_t* a = img + ww_off(row, col >> 3, (col >> 2) & 1);
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)a);
}

// Everything after the row loop, shared by the split-bf16 and the exact-fp32 kernel: the workgroup's split-K slab of dW1 / db1 and
// the tail layer's gradients (streamed or derived).  `smem` = the kernel's LDS (the operand images are dead by now).
template <int MODE>
__device__ __forceinline__ void ww_finish(const WsWgradP& p, float* ws_smem, const f32x4 (&acc)[16][2], const f32x4 (&accb)[2], const f32x4& tacc,
                                          const f32x4& bacc, float dqsum, const float* __restrict__ wtg, int z0, int z1, int ncol0, float inv = 1.0f) {
  constexpr bool TAILS = (MODE == 1);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lq = lane >> 4;
  // `inv`: the MFMA accumulators (acc, accb) carry the dynamic gradient scale of the split-precision kernel; tacc / bacc / dqsum do not
  // ---- one slab per workgroup ----
  const long so = z0 * p.o_s0 + (long)blockIdx.x * p.o_ks;
  float* dW = p.dW + so + z1 * p.o_s1w;
  float* db = p.db + so + z1 * p.o_s1b;
  // ---- MODE 2, first: the tail gradient's partial sums (they need the accumulators and W1 only) BEFORE the slab stores -- vmcnt returns
  // in order, so a load issued behind 128 stores waits for all of them.  The W1 elements are fetched 16 at a time (one exposed L2 latency
  // per two k blocks; one load -> wait -> multiply round trip per element cost ~40 us per workgroup), and the 16 lanes of a row group are
  // summed with DPP row operations on the vector ALU instead of four dependent ds_bpermute round trips per element.
  if (MODE == 2) {
    const float* __restrict__ W1g = p.W1 + z0 * p.w1_s0 + z1 * p.w1_s1;
    float* red = ws_smem;                                            // the images are dead after the loop's last barrier
#pragma unroll
    for (int kb2 = 0; kb2 < 16; kb2 += 2) {
      float wa[2][4], wb[2][4];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = 16 * (kb2 + h) + 4 * lq + r;
          wa[h][r] = W1g[(long)o * WS_N + ncol0 + li];
          wb[h][r] = W1g[(long)o * WS_N + ncol0 + 16 + li];
        }
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int o = 16 * (kb2 + h) + 4 * lq + r;
          float t = (wa[h][r] * acc[kb2 + h][0][r] + wb[h][r] * acc[kb2 + h][1][r]) * inv;
          // row sum over the 16 lanes li: quad pairs, quads, half-row mirror, row mirror (every lane ends with the full sum)
          t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0xB1, 0xF, 0xF, false));
          t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x4E, 0xF, 0xF, false));
          t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x141, 0xF, 0xF, false));
          t += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, t), 0x140, 0xF, 0xF, false));
          if (li == 0) red[wave * WS_K + o] = t;
        }
    }
  }
  if (MODE == 3) {                                   // plain dZ: the accumulators are the gradient (no rank-1 factor)
#pragma unroll
    for (int kb = 0; kb < 16; ++kb)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int r = 0; r < 4; ++r) dW[(long)(16 * kb + 4 * lq + r) * WS_N + ncol0 + 16 * nb + li] = acc[kb][nb][r] * inv;
    if (li == 0) {
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        const int k0 = 16 * (2 * wave + x) + 4 * lq;
#pragma unroll
        for (int r = 0; r < 4; ++r) db[k0 + r] = accb[x][r] * inv;
      }
    }
    return;
  }
#pragma unroll
  for (int kb = 0; kb < 16; ++kb) {
    const f32x4 w4 = *(const f32x4*)&wtg[16 * kb + 4 * lq];           // lane holds rows k = 16 kb + 4 lq + r, column n = ncol0 + 16 nb + li
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) dW[(long)(16 * kb + 4 * lq + r) * WS_N + ncol0 + 16 * nb + li] = (w4[r] * inv) * acc[kb][nb][r];
  }
  if (MODE == 2) {
    // dw_tail partial of this slab: every lane folded its two input columns of each of its 64 output units above (before the slab
    // stores); the 8 waves (= all 256 input columns) are summed in a fixed order
    float* red = ws_smem;
    if (li == 0) {
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[8 * WS_K + 16 * (2 * wave + x) + 4 * lq + r] = accb[x][r] * inv;
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
        for (int r = 0; r < 4; ++r) db[k0 + r] = wtg[k0 + r] * (accb[x][r] * inv);
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

// MODE 0: dW1 / db1 only, 1: h1 streamed for the tail gradients, 2: tail gradients derived from the accumulators,
//      3: PLAIN -- a materialised dZ instead of (mask, dq, w_tail): A gets a lo plane (three products per block), B = H0 itself
//      4: PLAIN with H0 = relu(X0 W0^T + b0) recomputed per row group from the narrow input instead of streamed (WsWgradP::X0)
//      5: mode 2 at precision 2 -- G = dq (.) h0 in THREE fp16 planes (exact for an fp32 product on the planes' grid), three products per
//         block (the mask operand is exact in one plane): LDS planes {mask, G hi, G mid, G lo}
template <int MODE>
__global__ __launch_bounds__(WS_NT) void ws_wgrad_kernel(const WsWgradP p) {
  static_assert(WS_NW == 8 && WS_ROWS == 32, "8 waves x 32 columns, 32-row groups");
  constexpr bool TAILS = (MODE == 1), PLAIN = (MODE == 3 || MODE == 4), RECOMP = (MODE == 4), P3 = (MODE == 5);
  constexpr int NPL = (PLAIN || P3) ? 4 : 3;                        // LDS planes per buffer
  constexpr int DQP = P3 ? 3 : 2;                                   // planes of the dq block (bias-gradient operand)
  extern __shared__ __attribute__((aligned(16))) float ws_smem[];
  hx_t* img = (hx_t*)ws_smem;                                   // [buf][{mask, G hi, G lo}][32][256]   (PLAIN: {dZ hi, dZ lo, H hi, H lo})
  __shared__ u32x2_t mlut[16];                                      // 4 mask bits -> 4 bf16 values
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lq = lane >> 4;
  const int z = blockIdx.z, z0 = z / p.nz1, z1 = z - z0 * p.nz1;
  const unsigned int* __restrict__ ab = PLAIN ? nullptr : p.abits + z0 * p.ab_s0 + z1 * p.ab_s1;
  const float* __restrict__ dqg = PLAIN ? nullptr : p.dq + z0 * p.dq_s0 + z1 * p.dq_s1;
  const float* __restrict__ H0g = p.H0 + z0 * p.h0_s0 + z1 * p.h0_s1;
  const float* __restrict__ wtg = PLAIN ? nullptr : p.wt + z0 * p.wt_s0 + z1 * p.wt_s1;
  const int ncol0 = 32 * wave;
  // split precision: G = (dq * gs) (.) h0 (PLAIN: dZ * gs) with the run's dynamic power-of-two gradient scale gs, divided out of the slab in ww_finish
  const float gsc = p.gscale ? p.gscale[z0] : 1.f;

  WS_STAMP(0);
  f32x4 acc[16][2], accb[2];
#pragma unroll
  for (int kb = 0; kb < 16; ++kb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) acc[kb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  accb[0] = accb[1] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- staging registers of one row group ----
  f32x4 s0[4];
  f32x4 s1[(TAILS || PLAIN) ? 4 : 1];
  f32x4 tacc = (f32x4){0.f, 0.f, 0.f, 0.f}, bacc = (f32x4){0.f, 0.f, 0.f, 0.f};   // TAILS: dw_tail / db1 partials of columns 4 (tid & 63) ..
  float dqsum = 0.f;
  // second streamed matrix: TAILS h1 (registers only), PLAIN dZ (the A operand)
  const float* __restrict__ H1g = TAILS ? p.H1 + z0 * p.h1_s0 + z1 * p.h1_s1 : (PLAIN ? p.dZ + z0 * p.dz_s0 + z1 * p.dz_s1 : nullptr);
  const int h1_pitch = PLAIN ? p.dz_pitch : p.h1_pitch;
  float sdq[4];
  unsigned int sm_word;
  hx_t* dqimg = img + 2 * NPL * WW_IMG;                            // [buf][hi, lo][32 rows][16]: column 0 = dq (PLAIN: 1), others 0 (db1 operand)
  // ---- RECOMP: first-layer fragments of this wave's 32 columns (K = 32: W0'[n][k] = W0[n][k] (k < in0), b0[n] (k == in0), 0 beyond) and the
  // staging of the narrow input rows -- ws_fwd_kernel<., L0>'s producer, writing into this kernel's H image ----
  float* Xl = (float*)(dqimg + 2 * DQP * WS_ROWS * 16);             // [buf][32][WS_XLP]: hi plane in 16-bit slots 0..31, lo plane in 32..63 of a row
  const float* __restrict__ X0g = RECOMP ? p.X0 + z0 * p.x0_s0 + z1 * p.x0_s1 : nullptr;
  hx8 b0h[2], b0l[2];
  if (RECOMP) {
    const float* __restrict__ W0g = p.W0 + z0 * p.w0_s0 + z1 * p.w0_s1;
    const float* __restrict__ b0g = p.b0 + z0 * p.b0_s0 + z1 * p.b0_s1;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int n = ncol0 + 16 * cb + li;
      f32x4 a, b;
      const float bn = b0g[n];
#pragma unroll
      for (int j = 0; j < 4; ++j) {                                  // clamped addresses and 0 / 1 factors, not guarded loads (ws_fwd.hip)
        const int k0 = 8 * lq + j, k1 = k0 + 4;
        const float w0 = W0g[(long)n * p.w0_sn + (long)(k0 < p.in0 ? k0 : p.in0 - 1) * p.w0_sk];
        const float w1 = W0g[(long)n * p.w0_sn + (long)(k1 < p.in0 ? k1 : p.in0 - 1) * p.w0_sk];
        a[j] = (k0 < p.in0 ? 1.f : 0.f) * w0 + (k0 == p.in0 ? 1.f : 0.f) * bn;
        b[j] = (k1 < p.in0 ? 1.f : 0.f) * w1 + (k1 == p.in0 ? 1.f : 0.f) * bn;
      }
      ws_split8(a, b, b0h[cb], b0l[cb]);
    }
  }
  const int xe = RECOMP ? WS_ROWS * p.x0_pitch : 0;
  int xr[2], xc[2];
  float sx[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + WS_NT * i;
    xr[i] = RECOMP ? e / (RECOMP ? p.x0_pitch : 1) : 0; xc[i] = RECOMP ? e - xr[i] * p.x0_pitch : 0;
    if (RECOMP && e >= xe) { xr[i] = 0; xc[i] = 32; }               // surplus threads: pad slots
  }
  auto loadX = [&](int g) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int e = tid + WS_NT * i; sx[i] = X0g[(long)g * xe + (e < xe ? e : xe - 1)]; }   // clamped, not predicated
  };
  auto storeX = [&](int xbuf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float x = (xc[i] == p.in0) ? 1.0f : sx[i];
      hx_t* row = (hx_t*)(Xl + (xbuf * WS_ROWS + xr[i]) * WS_XLP);
      const bool pad = xc[i] >= 32;
      hx_t hh, ll;
      orl_split1(x, hh, ll);
      row[pad ? 64 : xc[i]] = hh;
      row[pad ? 65 : 32 + xc[i]] = ll;
    }
  };
  hx8 xah, xal;
  auto prod_x = [&](int xbuf, int s) __attribute__((always_inline)) {
    const hx_t* xrow = (const hx_t*)(Xl + (xbuf * WS_ROWS + 16 * s + li) * WS_XLP) + 8 * lq;
    xah = *(const hx8*)xrow; xal = *(const hx8*)(xrow + 32);
  };
  // one 16 x 16 block of h0 of the group staged in Xl[xbuf] -> the H planes of image `buf` (the forward's three products and ReLU, bit for bit)
  auto prod_block = [&](int buf, int s, int cb) __attribute__((always_inline)) {
    hx_t* gh = img + (long)buf * NPL * WW_IMG + (NPL - 2) * WW_IMG;
    hx_t* gl = gh + WW_IMG;
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
    v = ORL_MFMA_16x16x32(b0l[cb], xah, v);
    v = ORL_MFMA_16x16x32(b0h[cb], xal, v);
    v = ORL_MFMA_16x16x32(b0h[cb], xah, v);
    orl_relu_mask4(v);
    hx4 h, l;
    orl_split4(v, h, l);
    const int k = ncol0 + 16 * cb + 4 * lq;                          // the lane holds C[m = 16 s + li][n = k .. k + 3]
    const int o = ww_off(16 * s + li, k >> 3, (k >> 2) & 1);
    *(hx4*)(gh + o) = h;
    *(hx4*)(gl + o) = l;
  };
  // Global addresses = (uniform part: row group and piece, scalar ALU) + (per-thread part, computed once): the loop carries no vector
  // address arithmetic (64-bit multiplies cost a SIMD 4 - 7 cycles each, and vector instructions do not overlap with its MFMAs).
  const unsigned int vo_h0 = (unsigned int)((tid >> 6) * p.h0_pitch + 4 * (tid & 63));
  const unsigned int vo_h1 = (TAILS || PLAIN) ? (unsigned int)((tid >> 6) * h1_pitch + 4 * (tid & 63)) : 0u;
  const unsigned int vo_dq = (unsigned int)((tid >> 6) * (int)p.dq_sm);
  const unsigned int vo_ab = (unsigned int)((tid >> 4) * p.ab_g + ((tid & 15) >> 1));
  auto load_piece = [&](int g, int i) __attribute__((always_inline)) {
    const long row0 = (long)g * WS_ROWS + 8 * i;                      // uniform: rows row0 + (tid >> 6)
    if (!RECOMP) s0[i] = *(const f32x4*)&(H0g + row0 * p.h0_pitch)[vo_h0];
    if (TAILS || PLAIN) s1[i] = *(const f32x4*)&(H1g + row0 * h1_pitch)[vo_h1];
    if (!PLAIN) sdq[i] = (dqg + row0 * p.dq_sm)[vo_dq];
  };
  auto load_mask = [&](int g) __attribute__((always_inline)) {
    if (!PLAIN) sm_word = (ab + (long)g * WS_ROWS * p.ab_g)[vo_ab];
  };
  auto load_group = [&](int g) __attribute__((always_inline)) {
    load_mask(g);
#pragma unroll
    for (int i = 0; i < 4; ++i) load_piece(g, i);
  };
  auto store_mask = [&](int buf) __attribute__((always_inline)) {
    if (PLAIN) return;
    hx_t* mi = img + (long)buf * NPL * WW_IMG;
    const int r = tid >> 4, hw = tid & 15;
    const unsigned int bits = (sm_word >> (16 * (hw & 1))) & 0xFFFFu;
    u32x4 c0, c1;
    // 4 bits -> 4 bf16 (0.0 / 1.0) through a 16-entry LDS table: four LDS reads instead of ~32 vector instructions per thread and group
    // (vector instructions and MFMAs of a SIMD do not overlap; the LDS pipe has room)
    const u32x2_t q0 = mlut[bits & 15u], q1 = mlut[(bits >> 4) & 15u], q2 = mlut[(bits >> 8) & 15u], q3 = mlut[bits >> 12];
    c0 = (u32x4){q0[0], q0[1], q1[0], q1[1]};
    c1 = (u32x4){q2[0], q2[1], q3[0], q3[1]};
    *(u32x4*)(mi + ww_off(r, 2 * hw, 0)) = c0;
    *(u32x4*)(mi + ww_off(r, 2 * hw + 1, 0)) = c1;
  };
  auto store_piece = [&](int buf, int i) __attribute__((always_inline)) {
    hx_t* gh = img + (long)buf * NPL * WW_IMG + (NPL - 2) * WW_IMG;
    hx_t* gl = gh + WW_IMG;
    const int idx = tid + WS_NT * i, r = idx >> 6, kq = idx & 63;
    hx4 h, l;
    const int o = ww_off(r, kq >> 1, kq & 1);
    if (PLAIN) {
      if (!RECOMP) {
        orl_split4(s0[i], h, l);
        *(hx4*)(gh + o) = h;
        *(hx4*)(gl + o) = l;
      }
      hx_t* zh = img + (long)buf * NPL * WW_IMG;
      orl_split4(s1[i] * gsc, h, l);
      *(hx4*)(zh + o) = h;
      *(hx4*)(zh + WW_IMG + o) = l;
      return;
    }
    const float dqs = sdq[i] * gsc;
    if (P3) {
      hx4 m;
      hx_t* g3 = img + (long)buf * NPL * WW_IMG + WW_IMG;
      orl_split4x3(s0[i] * dqs, h, m, l);
      *(hx4*)(g3 + o) = h;
      *(hx4*)(g3 + WW_IMG + o) = m;
      *(hx4*)(g3 + 2 * WW_IMG + o) = l;
      if (kq == 0) {
        hx_t hh, mm, ll;
        orl_split1x3(dqs, hh, mm, ll);
        hx_t* dqi = dqimg + (long)buf * DQP * WS_ROWS * 16;
        dqi[r * 16] = hh; dqi[WS_ROWS * 16 + r * 16] = mm; dqi[2 * WS_ROWS * 16 + r * 16] = ll;
        dqsum += sdq[i];
      }
      return;
    }
    orl_split4(s0[i] * dqs, h, l);
    *(hx4*)(gh + o) = h;
    *(hx4*)(gl + o) = l;
    if (TAILS) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { tacc[j] += sdq[i] * s1[i][j]; bacc[j] += s1[i][j] > 0.f ? sdq[i] : 0.f; }
      dqsum += sdq[i];
    } else if (kq == 0) {                                            // this row's dq into the bias-gradient operand block
      hx_t hh, ll;
      orl_split1(dqs, hh, ll);
      hx_t* dqi = dqimg + (long)buf * 2 * WS_ROWS * 16;
      dqi[r * 16] = hh; dqi[WS_ROWS * 16 + r * 16] = ll;
      if (MODE == 2) dqsum += sdq[i];
    }
  };
  auto store_group = [&](int buf) __attribute__((always_inline)) {
    store_mask(buf);
#pragma unroll
    for (int i = 0; i < 4; ++i) store_piece(buf, i);
  };
  if (!TAILS) for (int e = tid; e < 2 * DQP * WS_ROWS * 16 / 2; e += WS_NT) ((unsigned int*)dqimg)[e] = 0u;   // columns 1..15 stay zero
  if (PLAIN) {                                                     // db = dZ^T 1: column 0 of the hi block = 1.0 for every row, both buffers
    __syncthreads();
    if (tid < 2 * WS_ROWS) (dqimg + (long)(tid >> 5) * 2 * WS_ROWS * 16)[(tid & 31) * 16] = (hx_t)1.0f;
  }
  if (tid < 16) mlut[tid] = (u32x2_t){((tid & 1u) | ((tid & 2u) << 15)) * ORL_HX_ONE_BITS, (((tid >> 2) & 1u) | ((tid & 8u) << 13)) * ORL_HX_ONE_BITS};
  if (RECOMP) for (int e = tid; e < 2 * WS_ROWS * WS_XLP; e += WS_NT) Xl[e] = 0.f;      // columns >= x0_pitch stay zero
  __syncthreads();

  const int g0 = blockIdx.x, gs = gridDim.x;
  WS_STAMP(1);
  // RECOMP: the narrow rows of iteration j live in Xl[j & 1]; iteration `it` produces h0 of iteration it + 1 into the other image and stages
  // the rows of iteration it + 2 where iteration it - 1 read its own
  if (RECOMP && g0 < p.groups) {
    loadX(g0);
    storeX(0);
    if (g0 + gs < p.groups) loadX(g0 + gs);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      prod_x(0, s);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) prod_block(0, s, cb);
    }
    if (g0 + gs < p.groups) storeX(1);
  }
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
    const hx_t* mi = img + (long)buf * NPL * WW_IMG;
    const hx_t* ml = mi + WW_IMG;                                    // PLAIN: lo plane of dZ
    const hx_t* gh = mi + (P3 ? 1 : NPL - 2) * WW_IMG;
    const hx_t* gl = gh + (P3 ? 2 : 1) * WW_IMG;
    const hx_t* gm = gh + WW_IMG;                                    // P3: the middle plane
    const hx_t* dqi = dqimg + (long)buf * DQP * WS_ROWS * 16;
    {
      // one v_mfma_f32_16x16x32_bf16 covers the whole 32-row group: its 8 k-values per lane are the two transposed reads of
      // rows 4 lq .. + 3 and 16 + 4 lq .. + 3 (the k order is free as long as A and B agree)
      auto cat = [](s16x4 x, s16x4 y) __attribute__((always_inline)) {
        hx8 r;
        *(s16x4*)&r = x; *((s16x4*)&r + 1) = y;
        return r;
      };
      hx8 bh[2], bl[2], bm[2];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        bh[nb] = cat(ww_tr(gh, 0, ncol0 + 16 * nb, lane), ww_tr(gh, 16, ncol0 + 16 * nb, lane));
        bl[nb] = cat(ww_tr(gl, 0, ncol0 + 16 * nb, lane), ww_tr(gl, 16, ncol0 + 16 * nb, lane));
        if (P3) bm[nb] = cat(ww_tr(gm, 0, ncol0 + 16 * nb, lane), ww_tr(gm, 16, ncol0 + 16 * nb, lane));
      }
      const int dro0 = (4 * lq + (li >> 2)) * 16 + 4 * (li & 3), dro1 = dro0 + 16 * 16;
      typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
#pragma unroll
      for (int kp = 0; kp < 8; ++kp) {                               // two 16-row k blocks per trip: dependent MFMAs are 4 apart
        const int kb0 = 2 * kp, kb1 = kb0 + 1;
        const hx8 a0 = cat(ww_tr(mi, 0, 16 * kb0, lane), ww_tr(mi, 16, 16 * kb0, lane));   // A[i = k][kk = m] = mask[m][k]   (PLAIN: dZ hi)
        const hx8 a1 = cat(ww_tr(mi, 0, 16 * kb1, lane), ww_tr(mi, 16, 16 * kb1, lane));
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[kb0][nb] = ORL_MFMA_16x16x32(a0, bl[nb], acc[kb0][nb]);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[kb1][nb] = ORL_MFMA_16x16x32(a1, bl[nb], acc[kb1][nb]);
        if (P3) {                                                    // mask * mid(G)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) acc[kb0][nb] = ORL_MFMA_16x16x32(a0, bm[nb], acc[kb0][nb]);
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) acc[kb1][nb] = ORL_MFMA_16x16x32(a1, bm[nb], acc[kb1][nb]);
        }
        if (PLAIN) {                                                 // lo(dZ) * hi(H)
          const hx8 l0 = cat(ww_tr(ml, 0, 16 * kb0, lane), ww_tr(ml, 16, 16 * kb0, lane));
          const hx8 l1 = cat(ww_tr(ml, 0, 16 * kb1, lane), ww_tr(ml, 16, 16 * kb1, lane));
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) acc[kb0][nb] = ORL_MFMA_16x16x32(l0, bh[nb], acc[kb0][nb]);
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) acc[kb1][nb] = ORL_MFMA_16x16x32(l1, bh[nb], acc[kb1][nb]);
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[kb0][nb] = ORL_MFMA_16x16x32(a0, bh[nb], acc[kb0][nb]);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[kb1][nb] = ORL_MFMA_16x16x32(a1, bh[nb], acc[kb1][nb]);
        if (!TAILS && kp == 7) {                                     // this wave's share of db1: k blocks 2 wave, 2 wave + 1 (own reads: no branch)
          // (the dq block's fragments are read here, next to their only use: fetched at the top of the iteration they were 8 - 12 more live registers -- mode 2: 4 -> 0 spilled registers, 519 -> 510 us; mode 5: 21 -> 2, 776 -> 706 us at 1 x 128, one-call A/B)
          hx8 bdh, bdl, bdm;
          bdh = cat(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(dqi + dro0)), __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(dqi + dro1)));
          bdl = cat(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(dqi + (DQP - 1) * WS_ROWS * 16 + dro0)),
                    __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(dqi + (DQP - 1) * WS_ROWS * 16 + dro1)));
          if (P3) bdm = cat(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(dqi + WS_ROWS * 16 + dro0)),
                            __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(dqi + WS_ROWS * 16 + dro1)));
          const hx8 c0 = cat(ww_tr(mi, 0, 32 * wave, lane), ww_tr(mi, 16, 32 * wave, lane));
          const hx8 c1 = cat(ww_tr(mi, 0, 32 * wave + 16, lane), ww_tr(mi, 16, 32 * wave + 16, lane));
          if (PLAIN) {                                               // the ones block has no lo plane; dZ has
            const hx8 e0 = cat(ww_tr(ml, 0, 32 * wave, lane), ww_tr(ml, 16, 32 * wave, lane));
            const hx8 e1 = cat(ww_tr(ml, 0, 32 * wave + 16, lane), ww_tr(ml, 16, 32 * wave + 16, lane));
            accb[0] = ORL_MFMA_16x16x32(e0, bdh, accb[0]);
            accb[1] = ORL_MFMA_16x16x32(e1, bdh, accb[1]);
          } else {
            accb[0] = ORL_MFMA_16x16x32(c0, bdl, accb[0]);
            accb[1] = ORL_MFMA_16x16x32(c1, bdl, accb[1]);
            if (P3) {
              accb[0] = ORL_MFMA_16x16x32(c0, bdm, accb[0]);
              accb[1] = ORL_MFMA_16x16x32(c1, bdm, accb[1]);
            }
          }
          accb[0] = ORL_MFMA_16x16x32(c0, bdh, accb[0]);
          accb[1] = ORL_MFMA_16x16x32(c1, bdh, accb[1]);
        }
        // each staging register is written to LDS and refilled at the same point of every iteration: a full iteration in flight
        if (kp < 4) {
          if (more) store_piece(buf ^ 1, kp);
          if (more2) load_piece(g + 2 * gs, kp);
        }
        if (RECOMP) {                                                // h0 of the next group: one block per trip behind the staging stores
          if (kp == 0 && more2) loadX(g + 2 * gs);
          if (kp >= 4 && more) {
            if (!(kp & 1)) prod_x(buf ^ 1, (kp - 4) >> 1);
            prod_block(buf ^ 1, (kp - 4) >> 1, kp & 1);
          }
          if (kp == 7 && more2) storeX(buf);
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
  WS_STAMP(2);
  for (; MODE != 1 && g + 2 * gs < p.groups; g += gs, ++it) iteration(g, it, true);    // (the h1-streaming variant measured slower that way)
  for (; g < p.groups; g += gs, ++it) iteration(g, it, false);
  WS_STAMP(3);

  ww_finish<(MODE == 4 ? 3 : (MODE == 5 ? 2 : MODE))>(p, ws_smem, acc, accb, tacc, bacc, dqsum, wtg, z0, z1, ncol0, 1.0f / gsc);
  WS_STAMP(4);
}

// ---- exact-fp32 flavour (precision 0): the same output-stationary structure on v_mfma_f32_16x16x4_f32 ----
// A[i = k][kk = m] = mask[m][k] and B[kk = m][j = n] = G[m][n] are single floats per lane, read with ds_read_b32 from row-major fp32
// images [32 rows][WW32_P]: a lane group lq reads 16 consecutive floats of row 4 step + lq, and the row pitch of 272 floats (16 mod 64
// banks) puts the four rows of one read on disjoint banks.  32 rows = 8 MFMA k steps; two 16-row k blocks of the output are in flight
// so that dependent MFMAs stay four instructions apart.
enum { WW32_P = 272, WW32_IMG = WS_ROWS * WW32_P };
static constexpr size_t ws_wgrad32_lds_bytes(bool recompute = false) {
  return sizeof(float) * ((size_t)2 * 2 * WW32_IMG + 2 * WS_ROWS + (recompute ? 2 * WS_ROWS * WS_XLP : 0)) > sizeof(float) * 17 * WS_K
             ? sizeof(float) * ((size_t)2 * 2 * WW32_IMG + 2 * WS_ROWS + (recompute ? 2 * WS_ROWS * WS_XLP : 0)) : sizeof(float) * 17 * WS_K;
}

template <int MODE>
__global__ __launch_bounds__(WS_NT) void ws_wgrad32_kernel(const WsWgradP p) {
  static_assert(WS_NW == 8 && WS_ROWS == 32, "8 waves x 32 columns, 32-row groups");
  constexpr bool TAILS = (MODE == 1), PLAIN = (MODE == 3 || MODE == 4), RECOMP = (MODE == 4);   // PLAIN: a materialised dZ as the A image, B = H0 itself; RECOMP: H0 rebuilt from the narrow input (ws_gemm.h)
  extern __shared__ __attribute__((aligned(16))) float ws_smem[];
  float* img = ws_smem;                                             // [buf][{mask, G}][32][WW32_P]   (PLAIN: {dZ, H0})
  float* dqs = img + 2 * 2 * WW32_IMG;                              // [buf][32]
  float* Xl = dqs + 2 * WS_ROWS;                                    // RECOMP: [buf][32][WS_XLP] narrow input rows (ones column at in0)
  __shared__ f32x4 mlut[16];                                        // 4 mask bits -> 4 floats (0.0 / 1.0)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lq = lane >> 4;
  const int z = blockIdx.z, z0 = z / p.nz1, z1 = z - z0 * p.nz1;
  const unsigned int* __restrict__ ab = PLAIN ? nullptr : p.abits + z0 * p.ab_s0 + z1 * p.ab_s1;
  const float* __restrict__ dqg = PLAIN ? nullptr : p.dq + z0 * p.dq_s0 + z1 * p.dq_s1;
  const float* __restrict__ H0g = p.H0 + z0 * p.h0_s0 + z1 * p.h0_s1;
  const float* __restrict__ wtg = PLAIN ? nullptr : p.wt + z0 * p.wt_s0 + z1 * p.wt_s1;
  const float* __restrict__ H1g = TAILS ? p.H1 + z0 * p.h1_s0 + z1 * p.h1_s1 : (PLAIN ? p.dZ + z0 * p.dz_s0 + z1 * p.dz_s1 : nullptr);
  const int h1_pitch = PLAIN ? p.dz_pitch : p.h1_pitch;
  const int ncol0 = 32 * wave;

  f32x4 acc[16][2], accb[2];
#pragma unroll
  for (int kb = 0; kb < 16; ++kb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) acc[kb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  accb[0] = accb[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  f32x4 s0[4];
  f32x4 s1[(TAILS || PLAIN) ? 4 : 1];
  f32x4 tacc = (f32x4){0.f, 0.f, 0.f, 0.f}, bacc = (f32x4){0.f, 0.f, 0.f, 0.f};
  float dqsum = 0.f;
  float sdq[4];
  unsigned int sm_word;
  // ---- RECOMP: first-layer fragments (k = 16 t + 4 lq + e) of this wave's columns and the narrow-input staging (ws_fwd_kernel<., L0, ., ., F32>) ----
  const float* __restrict__ X0g = RECOMP ? p.X0 + z0 * p.x0_s0 + z1 * p.x0_s1 : nullptr;
  f32x4 b0w[2][2];
  if (RECOMP) {
    const float* __restrict__ W0g = p.W0 + z0 * p.w0_s0 + z1 * p.w0_s1;
    const float* __restrict__ b0g = p.b0 + z0 * p.b0_s0 + z1 * p.b0_s1;
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int n = ncol0 + 16 * cb + li;
      const float bn = b0g[n];
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
  const int xe = RECOMP ? WS_ROWS * p.x0_pitch : 0;
  int xr[2], xc[2];
  float sx[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + WS_NT * i;
    xr[i] = RECOMP ? e / (RECOMP ? p.x0_pitch : 1) : 0; xc[i] = RECOMP ? e - xr[i] * p.x0_pitch : 0;
    if (RECOMP && e >= xe) { xr[i] = 0; xc[i] = 32; }               // surplus threads: pad column 32 (rows are consumed as 32 columns of the 36-float pitch)
  }
  auto loadX = [&](int g) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int e = tid + WS_NT * i; sx[i] = X0g[(long)g * xe + (e < xe ? e : xe - 1)]; }
  };
  auto storeX = [&](int xbuf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) Xl[(xbuf * WS_ROWS + xr[i]) * WS_XLP + xc[i]] = (xc[i] == p.in0) ? 1.0f : sx[i];
  };
  f32x4 fx32[2];
  auto prod_x = [&](int xbuf, int s) __attribute__((always_inline)) {
    const float* xrow = Xl + (xbuf * WS_ROWS + 16 * s + li) * WS_XLP + 4 * lq;
    fx32[0] = *(const f32x4*)xrow; fx32[1] = *(const f32x4*)(xrow + 16);
  };
  auto prod_block = [&](int buf, int s, int cb) __attribute__((always_inline)) {
    float* gi = img + (long)buf * 2 * WW32_IMG + WW32_IMG;
    f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) v = __builtin_amdgcn_mfma_f32_16x16x4f32(b0w[cb][t][e], fx32[t][e], v, 0, 0, 0);
    orl_relu_mask4(v);
    *(f32x4*)(gi + (16 * s + li) * WW32_P + ncol0 + 16 * cb + 4 * lq) = v;      // the lane holds C[m = 16 s + li][n = ncol0 + 16 cb + 4 lq ..]
  };
  // (uniform part of every global address on the scalar ALU, per-thread part computed once: see ws_wgrad_kernel)
  const unsigned int vo_h0 = (unsigned int)((tid >> 6) * p.h0_pitch + 4 * (tid & 63));
  const unsigned int vo_h1 = (TAILS || PLAIN) ? (unsigned int)((tid >> 6) * h1_pitch + 4 * (tid & 63)) : 0u;
  const unsigned int vo_dq = (unsigned int)((tid >> 6) * (int)p.dq_sm);
  const unsigned int vo_ab = (unsigned int)((tid >> 4) * p.ab_g + ((tid & 15) >> 1));
  auto load_piece = [&](int g, int i) __attribute__((always_inline)) {
    const long row0 = (long)g * WS_ROWS + 8 * i;
    if (!RECOMP) s0[i] = *(const f32x4*)&(H0g + row0 * p.h0_pitch)[vo_h0];
    if (TAILS || PLAIN) s1[i] = *(const f32x4*)&(H1g + row0 * h1_pitch)[vo_h1];
    if (!PLAIN) sdq[i] = (dqg + row0 * p.dq_sm)[vo_dq];
  };
  auto load_mask = [&](int g) __attribute__((always_inline)) {
    if (!PLAIN) sm_word = (ab + (long)g * WS_ROWS * p.ab_g)[vo_ab];
  };
  auto load_group = [&](int g) __attribute__((always_inline)) {
    load_mask(g);
#pragma unroll
    for (int i = 0; i < 4; ++i) load_piece(g, i);
  };
  auto store_mask = [&](int buf) __attribute__((always_inline)) {      // thread (row r, half-word hw): 16 mask bits -> 16 floats
    if (PLAIN) return;
    float* mi = img + (long)buf * 2 * WW32_IMG;
    const int r = tid >> 4, hw = tid & 15;
    const unsigned int bits = (sm_word >> (16 * (hw & 1))) & 0xFFFFu;
#pragma unroll
    for (int j = 0; j < 4; ++j) *(f32x4*)(mi + r * WW32_P + 16 * hw + 4 * j) = mlut[(bits >> (4 * j)) & 15u];      // 4 bits -> 4 floats: LDS table
  };
  auto store_piece = [&](int buf, int i) __attribute__((always_inline)) {
    float* gi = img + (long)buf * 2 * WW32_IMG + WW32_IMG;
    const int idx = tid + WS_NT * i, r = idx >> 6, kq = idx & 63;
    if (PLAIN) {
      if (!RECOMP) *(f32x4*)(gi + r * WW32_P + 4 * kq) = s0[i];
      *(f32x4*)(gi - WW32_IMG + r * WW32_P + 4 * kq) = s1[i];          // the A image: dZ
      return;
    }
    *(f32x4*)(gi + r * WW32_P + 4 * kq) = s0[i] * sdq[i];
    if (TAILS) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { tacc[j] += sdq[i] * s1[i][j]; bacc[j] += s1[i][j] > 0.f ? sdq[i] : 0.f; }
      dqsum += sdq[i];
    } else if (kq == 0) {
      dqs[buf * WS_ROWS + r] = sdq[i];
      if (MODE == 2) dqsum += sdq[i];
    }
  };
  auto store_group = [&](int buf) __attribute__((always_inline)) {
    store_mask(buf);
#pragma unroll
    for (int i = 0; i < 4; ++i) store_piece(buf, i);
  };

  if (threadIdx.x < 16) mlut[threadIdx.x] = (f32x4){(float)(threadIdx.x & 1u), (float)((threadIdx.x >> 1) & 1u), (float)((threadIdx.x >> 2) & 1u), (float)(threadIdx.x >> 3)};
  if (RECOMP) for (int e = tid; e < 2 * WS_ROWS * WS_XLP; e += WS_NT) Xl[e] = 0.f;      // columns >= x0_pitch stay zero
  __syncthreads();
  const int g0 = blockIdx.x, gs = gridDim.x;
  if (RECOMP && g0 < p.groups) {                                    // (Xl[j & 1] holds the rows of iteration j: see ws_wgrad_kernel)
    loadX(g0);
    storeX(0);
    if (g0 + gs < p.groups) loadX(g0 + gs);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      prod_x(0, s);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) prod_block(0, s, cb);
    }
    if (g0 + gs < p.groups) storeX(1);
  }
  if (g0 < p.groups) {
    load_group(g0);
    store_group(0);
    if (g0 + gs < p.groups) load_group(g0 + gs);
  }
  __syncthreads();
  auto iteration = [&](int g, int it, bool steady) __attribute__((always_inline)) {
    const int buf = it & 1;
    const bool more = steady || g + gs < p.groups, more2 = steady || g + 2 * gs < p.groups;
    const float* mi = img + (long)buf * 2 * WW32_IMG;
    const float* gi = mi + WW32_IMG;
    const float* dqb = dqs + buf * WS_ROWS;
    const int ro = lq * WW32_P + li;                                  // row 4 step + lq, column (block) + li
    float bg[2][8], bd[8];
#pragma unroll
    for (int st = 0; st < 8; ++st) {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) bg[nb][st] = gi[4 * st * WW32_P + ro + ncol0 + 16 * nb];
      if (!TAILS) { const float d = PLAIN ? 1.0f : dqb[4 * st + lq]; bd[st] = li == 0 ? d : 0.f; }
    }
#pragma unroll
    for (int kp = 0; kp < 8; ++kp) {
      const int kb0 = 2 * kp, kb1 = kb0 + 1;
      float a0[8], a1[8];
#pragma unroll
      for (int st = 0; st < 8; ++st) { a0[st] = mi[4 * st * WW32_P + ro + 16 * kb0]; a1[st] = mi[4 * st * WW32_P + ro + 16 * kb1]; }
#pragma unroll
      for (int st = 0; st < 8; ++st) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[kb0][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[st], bg[nb][st], acc[kb0][nb], 0, 0, 0);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) acc[kb1][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[st], bg[nb][st], acc[kb1][nb], 0, 0, 0);
      }
      if (!TAILS && kp == 7) {                                        // this wave's share of db1: k blocks 2 wave, 2 wave + 1 (own reads: no branch)
        float c0[8], c1[8];
#pragma unroll
        for (int st = 0; st < 8; ++st) { c0[st] = mi[4 * st * WW32_P + ro + 32 * wave]; c1[st] = mi[4 * st * WW32_P + ro + 32 * wave + 16]; }
#pragma unroll
        for (int st = 0; st < 8; ++st) {
          accb[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(c0[st], bd[st], accb[0], 0, 0, 0);
          accb[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(c1[st], bd[st], accb[1], 0, 0, 0);
        }
      }
      if (kp < 4) {
        if (more) store_piece(buf ^ 1, kp);
        if (more2) load_piece(g + 2 * gs, kp);
      }
      if (RECOMP) {
        if (kp == 0 && more2) loadX(g + 2 * gs);
        if (kp >= 4 && more) {
          if (!(kp & 1)) prod_x(buf ^ 1, (kp - 4) >> 1);
          prod_block(buf ^ 1, (kp - 4) >> 1, kp & 1);
        }
        if (kp == 7 && more2) storeX(buf);
      }
      if (kp == 5) {
        if (more) store_mask(buf ^ 1);
        if (more2) load_mask(g + 2 * gs);
      }
    }
    __syncthreads();
  };
  int g = g0, it = 0;
  for (; MODE != 1 && g + 2 * gs < p.groups; g += gs, ++it) iteration(g, it, true);
  for (; g < p.groups; g += gs, ++it) iteration(g, it, false);
  ww_finish<(MODE == 4 ? 3 : MODE)>(p, ws_smem, acc, accb, tacc, bacc, dqsum, wtg, z0, z1, ncol0);
}

hipError_t launch_ws_wgrad(WsWgradP p, int nz, int per_z, hipStream_t st) {
  p.groups = p.M / WS_ROWS;
  static const hipError_t attr_err = [] {
    hipError_t e = hipFuncSetAttribute((const void*)ws_wgrad_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ws_wgrad_lds_bytes());
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_wgrad_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ws_wgrad_lds_bytes());
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_wgrad_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ws_wgrad_lds_bytes());
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_wgrad32_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ws_wgrad32_lds_bytes());
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_wgrad32_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ws_wgrad32_lds_bytes());
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_wgrad32_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ws_wgrad32_lds_bytes());
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_wgrad_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ws_wgrad_lds_bytes(true));
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_wgrad32_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ws_wgrad32_lds_bytes());
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_wgrad_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ws_wgrad_lds_bytes(true, true));
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_wgrad32_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ws_wgrad32_lds_bytes(true));
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ws_wgrad_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ws_wgrad_lds_bytes(false, false, true));
    return e;
  }();
  if (attr_err != hipSuccess) return attr_err;
  const dim3 grid(per_z, 1, nz), block(WS_NT);
  if (p.dZ && p.X0) {                              // ... with the layer's input activation recomputed from the net's narrow input
    if (p.f32) hipLaunchKernelGGL(ws_wgrad32_kernel<4>, grid, block, ws_wgrad32_lds_bytes(true), st, p);
    else hipLaunchKernelGGL(ws_wgrad_kernel<4>, grid, block, ws_wgrad_lds_bytes(true, true), st, p);
    return hipGetLastError();
  }
  if (p.dZ) {                                      // plain (materialised) gradient: a hidden layer below the top one
    if (p.f32) hipLaunchKernelGGL(ws_wgrad32_kernel<3>, grid, block, ws_wgrad32_lds_bytes(), st, p);
    else hipLaunchKernelGGL(ws_wgrad_kernel<3>, grid, block, ws_wgrad_lds_bytes(true), st, p);
    return hipGetLastError();
  }
  if (p.np3) {                                     // precision 2: three G planes (the derived-tail flavour only: ws_wgrad_supported)
    hipLaunchKernelGGL(ws_wgrad_kernel<5>, grid, block, ws_wgrad_lds_bytes(false, false, true), st, p);
    return hipGetLastError();
  }
  if (p.f32) {
    if (p.H1) hipLaunchKernelGGL(ws_wgrad32_kernel<1>, grid, block, ws_wgrad32_lds_bytes(), st, p);
    else if (p.W1) hipLaunchKernelGGL(ws_wgrad32_kernel<2>, grid, block, ws_wgrad32_lds_bytes(), st, p);
    else hipLaunchKernelGGL(ws_wgrad32_kernel<0>, grid, block, ws_wgrad32_lds_bytes(), st, p);
    return hipGetLastError();
  }
  if (p.H1) hipLaunchKernelGGL(ws_wgrad_kernel<1>, grid, block, ws_wgrad_lds_bytes(), st, p);
  else if (p.W1) hipLaunchKernelGGL(ws_wgrad_kernel<2>, grid, block, ws_wgrad_lds_bytes(), st, p);
  else hipLaunchKernelGGL(ws_wgrad_kernel<0>, grid, block, ws_wgrad_lds_bytes(), st, p);
  return hipGetLastError();
}

}  // namespace orl
