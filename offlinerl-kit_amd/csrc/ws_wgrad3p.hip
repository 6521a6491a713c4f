// ws_wgrad3p.hip — precision 2 (three fp16 planes, fp32-class arithmetic) flavour of ws_wgrad_kernel<3>: the weight gradient of a hidden layer
// BELOW the top one of a many-row batch (a three-layer net's middle layer), dW[k][n] = sum_m dZ[m][k] H0[m][n], db[k] = sum_m dZ[m][k], with
// dZ a materialised matrix (interface and design notes: ws_gemm.h; the two-plane kernel this follows: ws_wgrad.hip).
//
// Both operands are streamed, so three planes of both would be 2 x 6 x 16 KB = 192 KB of LDS.  A workgroup therefore owns HALF of the
// OUTPUT ROWS k (blockIdx.y = the half: 128 columns of dZ): its A images are [32][128] (3 x 8 KB), the H images stay [32][256] (3 x 16 KB),
// 72 KB per buffer; two workgroups stream the same H0 rows.  Output-stationary as before: wave w owns columns n in [32 w, 32 w + 32) and the
// half's 128 rows k (8 x 2 blocks = 64 accumulator VGPRs); six products per block (lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi); both MFMA
// operands through transposing LDS reads.  The two halves write disjoint rows of the same split-K slab.
// Reference: autograd of nets/mlp.py:9-33 (mm of the transposed gradient with the layer input; sum over the batch for the bias).
#include "ws_device.h"

namespace orl {

enum { W3_ZK = 128, W3_ZIMG = WS_ROWS * W3_ZK, W3_HIMG = WS_ROWS * WS_K };

// fp16 offset of the 8-byte piece (16-byte chunk `chunk`, half `half`) of row r of an image with `pitch` columns: chunks XOR-swizzled with 2 (r & 7)
// (ws_wgrad.hip: ww_off)
__device__ inline int w3_off(int r, int chunk, int half, int pitch) { return r * pitch + ((chunk ^ (2 * (r & 7))) << 3) + (half << 2); }

__device__ inline s16x4 w3_tr(const hx_t* img, int pitch, int row0, int col0, int lane) {
  // lane li of 16-lane group lq receives column col0 + li of rows row0 + 4 lq .. + 3 (the 16x16x16 operand layout of the image's transpose)
  const int li = lane & 15, lq = lane >> 4, row = row0 + 4 * lq + (li >> 2), col = col0 + 4 * (li & 3);
  const hx_t* a = img + w3_off(row, col >> 3, (col >> 2) & 1, pitch);
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)a);
}

__global__ __launch_bounds__(WS_NT) void ws_wgrad3p_kernel(const WsWgradP p) {
  static_assert(WS_NW == 8 && WS_ROWS == 32, "8 waves x 32 columns, 32-row groups");
  extern __shared__ __attribute__((aligned(16))) float ws_smem[];
  hx_t* img = (hx_t*)ws_smem;                                   // [buf][Z hi, Z mid, Z lo ([32][128]) | H hi, H mid, H lo ([32][256])]
  constexpr int BUF = 3 * W3_ZIMG + 3 * W3_HIMG;                     // fp16 elements per buffer
  hx_t* ones = img + 2 * BUF;                                      // [32 rows][16]: column 0 = 1.0 (bias-gradient operand), others 0
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lq = lane >> 4;
  const int half = blockIdx.y;
  const int z = blockIdx.z, z0 = z / p.nz1, z1 = z - z0 * p.nz1;
  const float* __restrict__ H0g = p.H0 + z0 * p.h0_s0 + z1 * p.h0_s1;
  const float* __restrict__ Zg = p.dZ + z0 * p.dz_s0 + z1 * p.dz_s1 + W3_ZK * half;      // this half's 128 columns of dZ
  const int ncol0 = 32 * wave;
  const float gsc = p.gscale ? p.gscale[z0] : 1.f;                   // dZ enters times the run's dynamic gradient scale, divided out of the slab

  f32x4 acc[8][2], accb;
#pragma unroll
  for (int kb = 0; kb < 8; ++kb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) acc[kb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  accb = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- staging: thread t moves float4 #(t + 512 i) of the H0 row group (i = 0..3) and of the half-width dZ row group (i = 0..1) ----
  f32x4 sh[4], sz[2];
  const unsigned int vo_h = (unsigned int)((tid >> 6) * p.h0_pitch + 4 * (tid & 63));
  const unsigned int vo_z = (unsigned int)((tid >> 5) * p.dz_pitch + 4 * (tid & 31));
  auto load_h = [&](int g, int i) __attribute__((always_inline)) {
    sh[i] = *(const f32x4*)&(H0g + ((long)g * WS_ROWS + 8 * i) * p.h0_pitch)[vo_h];
  };
  auto load_z = [&](int g, int i) __attribute__((always_inline)) {
    sz[i] = *(const f32x4*)&(Zg + ((long)g * WS_ROWS + 16 * i) * p.dz_pitch)[vo_z];
  };
  auto store_h = [&](int buf, int i) __attribute__((always_inline)) {
    hx_t* hp = img + (long)buf * BUF + 3 * W3_ZIMG;
    const int idx = tid + WS_NT * i, r = idx >> 6, kq = idx & 63;
    hx4 h, m, l;
    orl_split4x3(sh[i], h, m, l);
    const int o = w3_off(r, kq >> 1, kq & 1, WS_K);
    *(hx4*)(hp + o) = h;
    *(hx4*)(hp + W3_HIMG + o) = m;
    *(hx4*)(hp + 2 * W3_HIMG + o) = l;
  };
  auto store_z = [&](int buf, int i) __attribute__((always_inline)) {
    hx_t* zp = img + (long)buf * BUF;
    const int idx = tid + WS_NT * i, r = idx >> 5, kq = idx & 31;
    hx4 h, m, l;
    orl_split4x3(sz[i] * gsc, h, m, l);
    const int o = w3_off(r, kq >> 1, kq & 1, W3_ZK);
    *(hx4*)(zp + o) = h;
    *(hx4*)(zp + W3_ZIMG + o) = m;
    *(hx4*)(zp + 2 * W3_ZIMG + o) = l;
  };
  for (int e = tid; e < WS_ROWS * 16 / 2; e += WS_NT) ((unsigned int*)ones)[e] = 0u;
  __syncthreads();
  if (tid < WS_ROWS) ones[tid * 16] = (hx_t)1.0f;

  const int g0 = blockIdx.x, gs = gridDim.x;
  if (g0 < p.groups) {
#pragma unroll
    for (int i = 0; i < 4; ++i) load_h(g0, i);
#pragma unroll
    for (int i = 0; i < 2; ++i) load_z(g0, i);
#pragma unroll
    for (int i = 0; i < 4; ++i) store_h(0, i);
#pragma unroll
    for (int i = 0; i < 2; ++i) store_z(0, i);
    if (g0 + gs < p.groups) {
#pragma unroll
      for (int i = 0; i < 4; ++i) load_h(g0 + gs, i);
#pragma unroll
      for (int i = 0; i < 2; ++i) load_z(g0 + gs, i);
    }
  }
  __syncthreads();
  auto cat = [](s16x4 x, s16x4 y) __attribute__((always_inline)) {
    hx8 r;
    *(s16x4*)&r = x; *((s16x4*)&r + 1) = y;
    return r;
  };
  auto iteration = [&](int g, int it, bool steady) __attribute__((always_inline)) {
    const int buf = it & 1;
    const bool more = steady || g + gs < p.groups, more2 = steady || g + 2 * gs < p.groups;
    const hx_t* zh = img + (long)buf * BUF;
    const hx_t* zm = zh + W3_ZIMG;
    const hx_t* zl = zm + W3_ZIMG;
    const hx_t* hh = zh + 3 * W3_ZIMG;
    const hx_t* hm = hh + W3_HIMG;
    const hx_t* hl = hm + W3_HIMG;
    // one v_mfma_f32_16x16x32_f16 covers the whole 32-row group: its 8 k-values per lane are the transposed reads of rows 4 lq .. + 3 and 16 + 4 lq .. + 3
    hx8 bh[2], bm[2], bl[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      bh[nb] = cat(w3_tr(hh, WS_K, 0, ncol0 + 16 * nb, lane), w3_tr(hh, WS_K, 16, ncol0 + 16 * nb, lane));
      bm[nb] = cat(w3_tr(hm, WS_K, 0, ncol0 + 16 * nb, lane), w3_tr(hm, WS_K, 16, ncol0 + 16 * nb, lane));
      bl[nb] = cat(w3_tr(hl, WS_K, 0, ncol0 + 16 * nb, lane), w3_tr(hl, WS_K, 16, ncol0 + 16 * nb, lane));
    }
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
      const hx8 ah = cat(w3_tr(zh, W3_ZK, 0, 16 * kb, lane), w3_tr(zh, W3_ZK, 16, 16 * kb, lane));      // A[i = k][kk = m] = dZ[m][k]
      const hx8 am = cat(w3_tr(zm, W3_ZK, 0, 16 * kb, lane), w3_tr(zm, W3_ZK, 16, 16 * kb, lane));
      const hx8 al = cat(w3_tr(zl, W3_ZK, 0, 16 * kb, lane), w3_tr(zl, W3_ZK, 16, 16 * kb, lane));
      // smallest terms first; dependent MFMAs on one accumulator are two apart (the chain runs at the pipe's rate: MI355X_MICROARCH.md)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) acc[kb][nb] = ORL_MFMA_16x16x32(al, bh[nb], acc[kb][nb]);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) acc[kb][nb] = ORL_MFMA_16x16x32(ah, bl[nb], acc[kb][nb]);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) acc[kb][nb] = ORL_MFMA_16x16x32(am, bm[nb], acc[kb][nb]);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) acc[kb][nb] = ORL_MFMA_16x16x32(am, bh[nb], acc[kb][nb]);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) acc[kb][nb] = ORL_MFMA_16x16x32(ah, bm[nb], acc[kb][nb]);
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) acc[kb][nb] = ORL_MFMA_16x16x32(ah, bh[nb], acc[kb][nb]);
      if (kb == 7) {                                                   // this wave's share of db: k block `wave` of the half (own reads: no branch)
        typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
        const int dro0 = (4 * lq + (li >> 2)) * 16 + 4 * (li & 3), dro1 = dro0 + 16 * 16;
        const hx8 b1 = cat(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ones + dro0)), __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(ones + dro1)));
        const hx8 ch = cat(w3_tr(zh, W3_ZK, 0, 16 * wave, lane), w3_tr(zh, W3_ZK, 16, 16 * wave, lane));
        const hx8 cm = cat(w3_tr(zm, W3_ZK, 0, 16 * wave, lane), w3_tr(zm, W3_ZK, 16, 16 * wave, lane));
        const hx8 cl = cat(w3_tr(zl, W3_ZK, 0, 16 * wave, lane), w3_tr(zl, W3_ZK, 16, 16 * wave, lane));
        accb = ORL_MFMA_16x16x32(cl, b1, accb);
        accb = ORL_MFMA_16x16x32(cm, b1, accb);
        accb = ORL_MFMA_16x16x32(ch, b1, accb);
      }
      // each staging register is written to LDS and refilled at the same point of every iteration: a full iteration in flight
      if (kb < 4) {
        if (more) store_h(buf ^ 1, kb);
        if (more2) load_h(g + 2 * gs, kb);
      } else if (kb < 6) {
        if (more) store_z(buf ^ 1, kb - 4);
        if (more2) load_z(g + 2 * gs, kb - 4);
      }
    }
    __syncthreads();
  };
  int g = g0, it = 0;
  for (; g + 2 * gs < p.groups; g += gs, ++it) iteration(g, it, true);
  for (; g < p.groups; g += gs, ++it) iteration(g, it, false);

  // ---- one slab per workgroup pair: this half's rows k = 128 half + 16 kb + 4 lq + r, columns n = ncol0 + 16 nb + li ----
  const float inv = 1.0f / gsc;
  const long so = z0 * p.o_s0 + (long)blockIdx.x * p.o_ks;
  float* dW = p.dW + so + z1 * p.o_s1w;
  float* db = p.db + so + z1 * p.o_s1b;
#pragma unroll
  for (int kb = 0; kb < 8; ++kb)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) dW[(long)(W3_ZK * half + 16 * kb + 4 * lq + r) * WS_N + ncol0 + 16 * nb + li] = acc[kb][nb][r] * inv;
  if (li == 0) {
#pragma unroll
    for (int r = 0; r < 4; ++r) db[W3_ZK * half + 16 * wave + 4 * lq + r] = accb[r] * inv;
  }
}

hipError_t launch_ws_wgrad3p(WsWgradP p, int nz, int per_z, hipStream_t st) {
  p.groups = p.M / WS_ROWS;
  static const hipError_t attr_err = hipFuncSetAttribute((const void*)ws_wgrad3p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ws_wgrad3p_lds_bytes());
  if (attr_err != hipSuccess) return attr_err;
  hipLaunchKernelGGL(ws_wgrad3p_kernel, dim3(per_z, 2, nz), dim3(WS_NT), ws_wgrad3p_lds_bytes(), st, p);
  return hipGetLastError();
}

}  // namespace orl
