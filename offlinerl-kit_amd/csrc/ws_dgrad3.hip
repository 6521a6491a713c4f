// ws_dgrad3.hip — precision 2 (three fp16 planes, fp32-class arithmetic) flavour of ws_dgrad_w0_kernel<W0 = true, STORE = false>:
// backward through the top hidden layer of a single-output net from mask bits, fused with the layer-0 weight gradient
// (interface and design notes: ws_gemm.h; the two-plane kernel this follows: ws_dgrad.hip).
//
// Three planes of the resident operand B'[k][n] = w_tail[k] W1[k][n] for 32 columns would be 192 VGPRs per lane; a workgroup therefore owns
// HALF of the net's 256 columns (blockIdx.y = the half, wave w: columns 128 half + 16 w .. + 15: 96 VGPRs) and two workgroups stream the
// same row groups.  The A operand is the 0 / 1 ReLU mask of h1 (exact in one plane): three products per block instead of two.  The second
// stage dW0^T += X^T dz0 multiplies three planes of X^T by three planes of dz0 (six products on v_mfma_f32_16x16x16_f16).  The two halves of
// a (net, slab) write disjoint columns of the same split-K slab.
// Reference: autograd of modules/critic_module.py:17-28 over nets/mlp.py:9-33 (mm / addmm / threshold_backward), as ws_dgrad.hip.
#include "ws_device.h"

namespace orl {

// W0 = true: dz0 stays in registers and feeds the layer-0 weight gradient (one slab per workgroup pair).  W0 = false (STORE): dz0 is written
// to WsDgradP::C (the critic backward of an actor loss, where dz0 feeds dL/da; an actor's own backward) -- no dynamic scale then (the stored
// values are the true ones), no X^T image, no second stage.
// PLAIN (W0 only): the incoming gradient is a materialised matrix dz1 (WsDgradP::Z; a hidden layer below the top one) -- the A image holds its
// three planes (staged from HBM by both halves), B' = W1 itself: six products per block, no dq in the epilogue.
template <bool W0, bool PLAIN = false>
__global__ __launch_bounds__(WS_NT) void ws_dgrad3_w0_kernel(const WsDgradP p) {
  static_assert(!PLAIN || W0, "the plain variant exists for the fused layer-0 gradient only");
  constexpr int APL = PLAIN ? 3 : 1;                                // planes of the A image
  static_assert(WS_NW == 8 && WS_ROWS == 32, "16 columns per wave, 32-row groups");
  extern __shared__ __attribute__((aligned(16))) float ws_smem[];
  hx_t* Ah = (hx_t*)ws_smem;                                   // [buf][row][256] 0/1 mask, swizzled
  hx_t* XT = Ah + 2 * APL * WS_ROWS * WS_PITCH;                  // [buf][hi, mid, lo][c = 32][WD_XP]: X^T of the row group
  float* EO = (float*)(XT + 2 * 3 * 32 * WD_XP);                   // [buf][dq[32] | h0 mask words [word = 8][row = 32]]: epilogue operands
  __shared__ u32x2_t mlut[16];                                     // 4 mask bits -> 4 fp16 values
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lq = lane >> 4;
  const int half = blockIdx.y;
  const int z = blockIdx.z, z0 = z / p.nz1, z1 = z - z0 * p.nz1;
  const unsigned int* __restrict__ ab = PLAIN ? nullptr : p.abits + z0 * p.ab_s0 + z1 * p.ab_s1;
  const float* __restrict__ Zg = PLAIN ? p.Z + z0 * p.z_s0 + z1 * p.z_s1 : nullptr;
  const unsigned int* __restrict__ xb = p.xbits + z0 * p.xb_s0 + z1 * p.xb_s1;
  const float* __restrict__ dqg = PLAIN ? nullptr : p.dq + z0 * p.dq_s0 + z1 * p.dq_s1;
  const float* __restrict__ Wg = p.W + z0 * p.w_s0 + z1 * p.w_s1;
  const float* __restrict__ wtg = PLAIN ? nullptr : p.wt + z0 * p.wt_s0 + z1 * p.wt_s1;
  const float* __restrict__ Xg = W0 ? p.X + z0 * p.x_s0 + z1 * p.x_s1 : nullptr;
  float* __restrict__ Cg = W0 ? nullptr : p.C + z0 * p.c_s0 + z1 * p.c_s1;
  const int ncol0 = 128 * half + 16 * wave;
  // dq enters scaled by the run's dynamic gradient scale, the resident products by ORL_WWSCALE; dz0 carries gs * that scale into the second stage
  const float gsc = (W0 && p.gscale) ? p.gscale[z0] : 1.f;
  constexpr float BSC = PLAIN ? ORL_WSCALE : ORL_WWSCALE;             // static scale of the resident operand
  const float dq_sc = W0 ? gsc : 1.0f / ORL_WWSCALE;                 // factor applied to dq (PLAIN: dz1) when it is staged
  const float out_inv = 1.0f / (gsc * BSC);

  // resident B' fragments: lane (li, lq) supplies B'[k = 32 ks + 8 lq + j][n = ncol0 + li] = w_tail[k] * W1[k][n]
  hx8 bh[8], bm[8], bl[8];
  {
    f32x4 raw[8][2];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int n = ncol0 + li, k0 = 32 * ks + 8 * lq;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        raw[ks][0][j] = Wg[(long)n * p.w_sn + (long)(k0 + j) * p.w_sk];
        raw[ks][1][j] = Wg[(long)n * p.w_sn + (long)(k0 + 4 + j) * p.w_sk];
      }
    }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int k0 = 32 * ks + 8 * lq;
      f32x4 t0 = (f32x4){1.f, 1.f, 1.f, 1.f}, t1 = t0;
      if (!PLAIN) { t0 = *(const f32x4*)&wtg[k0]; t1 = *(const f32x4*)&wtg[k0 + 4]; }
      ws_split8x3((t0 * BSC) * raw[ks][0], (t1 * BSC) * raw[ks][1], bh[ks], bm[ks], bl[ks]);
    }
  }
  if (W0) for (int e = tid; e < 2 * 3 * 32 * WD_XP / 2; e += WS_NT) ((unsigned int*)XT)[e] = 0u;      // rows c >= x_pitch are never written again
  if (tid < 16) mlut[tid] = (u32x2_t){((tid & 1u) | ((tid & 2u) << 15)) * ORL_HX_ONE_BITS, (((tid >> 2) & 1u) | ((tid & 8u) << 13)) * ORL_HX_ONE_BITS};
  __syncthreads();

  // ---- staging of one row group (as in ws_dgrad_w0_kernel): thread (row r = t >> 4, half-word hw = t & 15) expands 16 mask bits; X^T elements ----
  unsigned int sm_word;
  float sx[2];
  const int xe = W0 ? WS_ROWS * p.x_pitch : 0;                       // X elements of a row group (<= 1024)
  int xo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + WS_NT * i;
    int rr = W0 ? e / (W0 ? p.x_pitch : 1) : 0, c = W0 ? e - rr * p.x_pitch : 0;
    if (e >= xe) { rr = 32; c = 0; }                                 // pad slot, never read
    xo[i] = ((c == p.in0) ? (1 << 16) : 0) | (c * WD_XP + rr);        // bit 16: the ones column (bias gradient)
  }
  float sdq;
  unsigned int sxw;
  f32x4 sz[PLAIN ? 4 : 1];                                          // PLAIN: row (tid >> 6) + 8 i, columns 4 (tid & 63) .. of the dz1 row group
  const unsigned int vo_z = PLAIN ? (unsigned int)((tid >> 6) * p.z_pitch + 4 * (tid & 63)) : 0u;
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
      hx_t* dh = Ah + (long)buf * 3 * WS_ROWS * WS_PITCH;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int r = (tid >> 6) + 8 * i, kq = tid & 63;              // columns 4 kq ..: half (kq & 1) of the 16-byte chunk kq >> 1
        hx4 h, mm, l;
        orl_split4x3(sz[i] * gsc, h, mm, l);
        const int o = r * WS_PITCH + ((((kq >> 1) ^ (r & 15)) << 3) | ((kq & 1) << 2));
        *(hx4*)(dh + o) = h;
        *(hx4*)(dh + WS_ROWS * WS_PITCH + o) = mm;
        *(hx4*)(dh + 2 * WS_ROWS * WS_PITCH + o) = l;
      }
    } else {
    const int r = tid >> 4, hw = tid & 15;
    const unsigned int bits = (sm_word >> (16 * (hw & 1))) & 0xFFFFu;
    const u32x2_t q0 = mlut[bits & 15u], q1 = mlut[(bits >> 4) & 15u], q2 = mlut[(bits >> 8) & 15u], q3 = mlut[bits >> 12];
    const u32x4 c0 = (u32x4){q0[0], q0[1], q1[0], q1[1]}, c1 = (u32x4){q2[0], q2[1], q3[0], q3[1]};
    hx_t* d = Ah + (long)buf * WS_ROWS * WS_PITCH + r * WS_PITCH;
    *(u32x4*)(d + (((2 * hw) ^ (r & 15)) << 3)) = c0;
    *(u32x4*)(d + (((2 * hw + 1) ^ (r & 15)) << 3)) = c1;
    eo[tid & 31] = sdq * dq_sc;                                      // (replicated writes of identical values)
    }
    ((unsigned int*)eo)[WS_ROWS + (tid & 7) * WS_ROWS + ((tid >> 3) & 31)] = sxw;
    hx_t* xt = XT + (long)buf * 3 * 32 * WD_XP;
#pragma unroll
    for (int i = 0; i < 2 && W0; ++i) {
      const float x = (xo[i] >> 16) ? 1.0f : sx[i];
      hx_t hh, mm, ll;
      orl_split1x3(x, hh, mm, ll);
      xt[xo[i] & 0xFFFF] = hh;
      xt[32 * WD_XP + (xo[i] & 0xFFFF)] = mm;
      xt[2 * 32 * WD_XP + (xo[i] & 0xFFFF)] = ll;
    }
  };

  f32x4 d2[2];                                                       // dW0^T blocks [c block] of this wave's 16 columns, accumulated over all groups
  d2[0] = d2[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int xword = 4 * half + (wave >> 1), xshift = 16 * (wave & 1) + li;   // the h0 mask word / bit of column ncol0 + li

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
    f32x4 dq4[WS_SUB];
    unsigned int xw[WS_SUB][4];
    const float* eo = EO + buf * (WS_ROWS + WS_NW * WS_ROWS);
#pragma unroll
    for (int s = 0; s < WS_SUB; ++s) {
      dq4[s] = PLAIN ? (f32x4){1.f, 1.f, 1.f, 1.f} : *(const f32x4*)&eo[16 * s + 4 * lq];
      const u32x4 w4 = *(const u32x4*)&((const unsigned int*)eo)[WS_ROWS + xword * WS_ROWS + 16 * s + 4 * lq];
#pragma unroll
      for (int r = 0; r < 4; ++r) xw[s][r] = w4[r];
    }
    const hx_t* ah = Ah + (long)buf * APL * WS_ROWS * WS_PITCH;
    f32x4 acc[WS_SUB];
#pragma unroll
    for (int s = 0; s < WS_SUB; ++s) acc[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      hx8 fa[WS_SUB];
#pragma unroll
      for (int s = 0; s < WS_SUB; ++s) fa[s] = *(const hx8*)&ah[(16 * s + li) * WS_PITCH + (((4 * ks + lq) ^ li) << 3)];
      // D[m][n]: lane holds rows 4 lq + r of column li; smallest terms first
#pragma unroll
      for (int s = 0; s < WS_SUB; ++s) acc[s] = ORL_MFMA_16x16x32(fa[s], bl[ks], acc[s]);
      if constexpr (PLAIN) {                                           // the planes of dz1 below the first: lo * hi, mid * mid, mid * hi
        hx8 fm[WS_SUB], fl[WS_SUB];
#pragma unroll
        for (int s = 0; s < WS_SUB; ++s) {
          const int o = (16 * s + li) * WS_PITCH + (((4 * ks + lq) ^ li) << 3);
          fm[s] = *(const hx8*)&ah[WS_ROWS * WS_PITCH + o]; fl[s] = *(const hx8*)&ah[2 * WS_ROWS * WS_PITCH + o];
        }
#pragma unroll
        for (int s = 0; s < WS_SUB; ++s) acc[s] = ORL_MFMA_16x16x32(fl[s], bh[ks], acc[s]);
#pragma unroll
        for (int s = 0; s < WS_SUB; ++s) acc[s] = ORL_MFMA_16x16x32(fm[s], bm[ks], acc[s]);
#pragma unroll
        for (int s = 0; s < WS_SUB; ++s) acc[s] = ORL_MFMA_16x16x32(fm[s], bh[ks], acc[s]);
      }
#pragma unroll
      for (int s = 0; s < WS_SUB; ++s) acc[s] = ORL_MFMA_16x16x32(fa[s], bm[ks], acc[s]);
#pragma unroll
      for (int s = 0; s < WS_SUB; ++s) acc[s] = ORL_MFMA_16x16x32(fa[s], bh[ks], acc[s]);
    }
    if constexpr (!W0) {
      // STORE: dz0 = 1[h0 > 0] (.) dq (.) acc -> C; the lane holds rows 16 s + 4 lq + r of column ncol0 + li
#pragma unroll
      for (int s = 0; s < WS_SUB; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          Cg[((long)g * WS_ROWS + 16 * s + 4 * lq + r) * p.c_pitch + ncol0 + li] = ((xw[s][r] >> xshift) & 1u) ? acc[s][r] * dq4[s][r] : 0.f;
    } else {
      // dz0 block -> three fp16 planes = B operand of the 16x16x16 MFMA; A operand = X^T rows c, columns m = 16 s + 4 lq ..
      const hx_t* xt = XT + (long)buf * 3 * 32 * WD_XP;
  #pragma unroll
      for (int s = 0; s < WS_SUB; ++s) {
        s16x4 xh[2], xm[2], xl[2];
  #pragma unroll
        for (int cbk = 0; cbk < 2; ++cbk) {
          const int o = (16 * cbk + li) * WD_XP + 16 * s + 4 * lq;
          xh[cbk] = *(const s16x4*)&xt[o];
          xm[cbk] = *(const s16x4*)&xt[32 * WD_XP + o];
          xl[cbk] = *(const s16x4*)&xt[2 * 32 * WD_XP + o];
        }
        hx4 zh, zm, zl;
        // (the four values split together: orl_split4x3 is 14 vector instructions, four scalar splits ~36; A/B 988 - 996 vs 1004 - 1016 us)
        {
          f32x4 v4;
  #pragma unroll
          for (int r = 0; r < 4; ++r) v4[r] = ((xw[s][r] >> xshift) & 1u) ? acc[s][r] * dq4[s][r] : 0.f;
          orl_split4x3(v4, zh, zm, zl);
        }
        const s16x4 bzh = *(const s16x4*)&zh, bzm = *(const s16x4*)&zm, bzl = *(const s16x4*)&zl;
  #pragma unroll
        for (int cbk = 0; cbk < 2; ++cbk) {
          d2[cbk] = ORL_MFMA_16x16x16(xl[cbk], bzh, d2[cbk]);
          d2[cbk] = ORL_MFMA_16x16x16(xh[cbk], bzl, d2[cbk]);
          d2[cbk] = ORL_MFMA_16x16x16(xm[cbk], bzm, d2[cbk]);
          d2[cbk] = ORL_MFMA_16x16x16(xm[cbk], bzh, d2[cbk]);
          d2[cbk] = ORL_MFMA_16x16x16(xh[cbk], bzm, d2[cbk]);
          d2[cbk] = ORL_MFMA_16x16x16(xh[cbk], bzh, d2[cbk]);
        }
      }
    }
    if (g + gs < p.groups) store_group(buf ^ 1);
    if (g + 2 * gs < p.groups) load_group(g + 2 * gs);
    __syncthreads();
  }
  if (!W0) return;
  // one slab per (workgroup pair): lane (li, lq) holds dW0^T[c = 16 cbk + 4 lq + r][n = ncol0 + li]
  float* wo = p.w0_out + z0 * p.o_s0 + z1 * p.o_s1 + (long)blockIdx.x * p.o_ks;
  float* bo = p.b0_out + z0 * p.o_s0 + z1 * p.ob_s1 + (long)blockIdx.x * p.o_ks;
#pragma unroll
  for (int cbk = 0; cbk < 2; ++cbk)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = 16 * cbk + 4 * lq + r, n = ncol0 + li;
      if (c < p.in0) wo[(long)n * p.o_sr + (long)c * p.o_sc] = d2[cbk][r] * out_inv;
      else if (c == p.in0) bo[n] = d2[cbk][r] * out_inv;
    }
}

hipError_t launch_ws_dgrad3_w0(WsDgradP p, int nz, int per_z, hipStream_t st) {
  p.groups = p.M / WS_ROWS;
  if (p.Z) {
    static const hipError_t attr_err = hipFuncSetAttribute((const void*)ws_dgrad3_w0_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ws_dgrad3_lds_bytes(true));
    if (attr_err != hipSuccess) return attr_err;
    hipLaunchKernelGGL((ws_dgrad3_w0_kernel<true, true>), dim3(per_z, 2, nz), dim3(WS_NT), ws_dgrad3_lds_bytes(true), st, p);
    return hipGetLastError();
  }
  if (p.w0_out) hipLaunchKernelGGL(ws_dgrad3_w0_kernel<true>, dim3(per_z, 2, nz), dim3(WS_NT), ws_dgrad3_lds_bytes(), st, p);
  else hipLaunchKernelGGL(ws_dgrad3_w0_kernel<false>, dim3(per_z, 2, nz), dim3(WS_NT), ws_dgrad3_lds_bytes(), st, p);
  return hipGetLastError();
}

}  // namespace orl
