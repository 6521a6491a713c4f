// small_fwd.hip — fused forward of a [in0 -> 256 -> 256 -> out] net for few batched rows (interface and design notes: small_fwd.h).
#include "small_fwd.h"

namespace orl {

// LDS layout (bytes).  Split planes: 16-bit elements; F32: floats.
//   A    : the 32 rows' current activation, all 256 k   (split: hi + lo planes [32][SF_AP], F32: [32][SF_AF])
//   W    : two buffers of one 32-wide k chunk of the layer's weights for all 256 output units (split: hi + lo [256][SF_WP], F32: [256][SF_WF])
//   X    : the 32 input rows, K padded to 32, ones column at in0 (split: hi + lo [32][SF_WP], F32: [32][SF_WF])
//   WT   : tail weights [16][SF_TWP] fp32, b1 [256], bt [16]
//   the cross-wave reduction of the tail reuses the W buffers after the last chunk
//   QG mode, backward pass: the W buffers hold 32-ROW chunks of W1 (rows = output units n of the layer = the contraction index of the dgrad,
//   all 256 input units j): split planes [32][256] 16-bit with 16-byte chunks XOR-swizzled per row (sf_toff: the transposing LDS reads and
//   the 8-byte staging stores are both conflict free, as in ws_wgrad.hip), F32 [32][SF_TF] floats (pitch = 4 mod 64 banks: the four rows
//   4 lq + e of one scalar fragment read sit 16 banks apart)
enum { SF_AP = 264, SF_WP = 40, SF_AF = 260, SF_WF = 36, SF_TF = 260, SF_TWP = 260 };
template <bool F32> static constexpr size_t sf_a_bytes() { return F32 ? (size_t)SF_ROWS * SF_AF * 4 : (size_t)2 * SF_ROWS * SF_AP * 2; }
template <bool F32> static constexpr size_t sf_w_bytes() { return F32 ? (size_t)2 * SF_N * SF_WF * 4 : (size_t)2 * 2 * SF_N * SF_WP * 2; }
template <bool F32> static constexpr size_t sf_x_bytes() { return F32 ? (size_t)SF_ROWS * SF_WF * 4 : (size_t)2 * SF_ROWS * SF_WP * 2; }
template <bool F32> static constexpr size_t sf_lds_bytes() {
  return sf_a_bytes<F32>() + sf_w_bytes<F32>() + sf_x_bytes<F32>() + (size_t)(SF_MAXOUT * SF_TWP + SF_N + SF_MAXOUT) * 4;
}
static_assert(sf_w_bytes<false>() >= (size_t)32 * SF_ROWS * SF_MAXOUT * 4 && sf_w_bytes<true>() >= (size_t)32 * SF_ROWS * SF_MAXOUT * 4, "tail reduction fits the chunk buffers");
static_assert(sf_w_bytes<false>() >= (size_t)2 * 2 * SF_ROWS * SF_N * 2 && sf_w_bytes<true>() >= (size_t)2 * SF_ROWS * SF_TF * 4, "transposed chunks fit the chunk buffers");
static_assert(sf_x_bytes<false>() >= (size_t)32 * SF_ROWS * 4 && sf_x_bytes<true>() >= (size_t)32 * SF_ROWS * 4, "QG: the q partial sums fit the input image");

// 16-bit element offset of the 8-byte piece (16-byte chunk `chunk`, half `half`) of row r of a [32][256] transposed-chunk plane
__device__ __forceinline__ int sf_toff(int r, int chunk, int half) { return r * SF_N + ((chunk ^ (2 * (r & 7))) << 3) + (half << 2); }
// transposing LDS read (gfx950 ds_read_b64_tr_b16): lane li of 16-lane group lq receives column col0 + li of rows row0 + 4 lq .. + 3
__device__ __forceinline__ s16x4 sf_tr(const hx_t* img, int row0, int col0, int lane) {
  const int li = lane & 15, lq = lane >> 4, row = row0 + 4 * lq + (li >> 2), col = col0 + 4 * (li & 3);
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(img + sf_toff(row, col >> 3, (col >> 2) & 1)));
}
__device__ __forceinline__ hx8 sf_cat(s16x4 x, s16x4 y) {
  hx8 r;
  *(s16x4*)&r = x; *((s16x4*)&r + 1) = y;
  return r;
}

#ifdef SB_LAB_CLOCK
#define SF_STAMP(i) do { if (p.lab_clk && threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.z == 0) p.lab_clk[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SF_STAMP(i) do { } while (0)
#endif

template <bool F32, bool QG>
__global__ __launch_bounds__(SF_NT) void small_fwd_kernel(const SmallFwdP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sf_smem[];
  unsigned char* sA = sf_smem;
  unsigned char* sW = sA + sf_a_bytes<F32>();
  unsigned char* sX = sW + sf_w_bytes<F32>();
  float* sWT = (float*)(sX + sf_x_bytes<F32>());
  float* sB1 = sWT + SF_MAXOUT * SF_TWP;
  float* sBT = sB1 + SF_N;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lq = lane >> 4;
  const int g = blockIdx.x, z = blockIdx.z, z0 = z / p.nz1, z1 = z - z0 * p.nz1;
  const int ncol0 = 32 * wave;
  const float* __restrict__ Xg = p.X + z0 * p.x_s0 + z1 * p.x_s1 + (long)g * SF_ROWS * p.x_pitch;
  const float* __restrict__ W0g = p.W0 + z0 * p.w0_s0 + z1 * p.w0_s1;
  const float* __restrict__ b0g = p.b0 + z0 * p.b0_s0 + z1 * p.b0_s1;
  const float* __restrict__ W1g = p.W1 + z0 * p.w1_s0 + z1 * p.w1_s1;
  const float* __restrict__ b1g = p.b1 + z0 * p.b1_s0 + z1 * p.b1_s1;
  const float* __restrict__ Wtg = p.Wt + z0 * p.wt_s0 + z1 * p.wt_s1;
  const float* __restrict__ btg = p.bt + z0 * p.bt_s0 + z1 * p.bt_s1;

  SF_STAMP(0);
  // ---- one 32-wide k chunk of weights: thread t moves the float4 (unit n = (t + 512 i) >> 3, k = 4 ((t + 512 i) & 7) ..), i = 0..3 ----
  // ALL nine chunks are requested up front (144 VGPRs): a workgroup has one 32-row group to do, so there is exactly one memory latency to
  // hide and nothing to hide it behind -- with one chunk in flight per iteration the launch took 15 - 19 us, nine exposed round trips
  f32x4 wr[9][4];
  auto load_chunk = [&](int c) __attribute__((always_inline)) {      // c = 0: [W0 | b0 | 0] (layer 0, K = 32); c = 1..8: columns 32 (c - 1) .. of W1
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + SF_NT * i, n = e >> 3, q = e & 7;
      if (c == 0) {
        // clamped addresses and a multiply by 0 / 1, not guarded loads or selects: the compiler sinks a load whose value is only selected
        // under a condition back into a branch of its own, each with its own wait (16 serialized round trips per thread, ~10 us per launch).
        // (0 * w keeps the padding exact for finite weights; a diverged run's Inf / NaN would spread into the zero columns -- of a net that
        // is already lost)
        const float bn = b0g[n];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = 4 * q + j;
          const float w = W0g[(long)n * p.in0 + (k < p.in0 ? k : p.in0 - 1)];
          wr[c][i][j] = (k < p.in0 ? 1.f : 0.f) * w + (k == p.in0 ? 1.f : 0.f) * bn;
        }
      } else wr[c][i] = *(const f32x4*)&W1g[(long)n * SF_N + 32 * (c - 1) + 4 * q];
    }
  };
  auto store_chunk = [&](int c) __attribute__((always_inline)) {     // chunk c -> LDS buffer c & 1
    const int buf = c & 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + SF_NT * i, n = e >> 3, q = e & 7;
      if constexpr (F32) *(f32x4*)((float*)sW + ((long)buf * SF_N + n) * SF_WF + 4 * q) = wr[c][i];
      else {
        hx_t* wh = (hx_t*)sW + (long)buf * 2 * SF_N * SF_WP;
        hx4 h, l;
        orl_split4(wr[c][i] * ORL_WSCALE, h, l);                     // static weight scale, divided out in the epilogues
        *(hx4*)(wh + n * SF_WP + 4 * q) = h;
        *(hx4*)(wh + SF_N * SF_WP + n * SF_WP + 4 * q) = l;
      }
    }
  };

  // QG: chunk c of the backward pass = rows 32 c .. 32 c + 31 of W1 (1 KB each); thread t moves the float4 (row (t + 512 i) >> 6, columns
  // 4 ((t + 512 i) & 63) ..).  It is requested into wr[c + 1] as soon as the forward has stored that register set to LDS, so the whole
  // second pass over W1 (L2 hits) is in flight while the forward still computes.
  auto load_chunkT = [&](int c) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + SF_NT * i;
      wr[c + 1][i] = *(const f32x4*)&W1g[(long)(32 * c + (e >> 6)) * SF_N + 4 * (e & 63)];
    }
  };
  auto store_chunkT = [&](int c) __attribute__((always_inline)) {
    const int buf = c & 1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + SF_NT * i, r = e >> 6, col = 4 * (e & 63);
      if constexpr (F32) *(f32x4*)((float*)sW + ((long)buf * SF_ROWS + r) * SF_TF + col) = wr[c + 1][i];
      else {
        hx_t* th = (hx_t*)sW + (long)buf * 2 * SF_ROWS * SF_N;
        hx4 h, l;
        orl_split4(wr[c + 1][i] * ORL_WSCALE, h, l);
        const int o = sf_toff(r, col >> 3, (col >> 2) & 1);
        *(hx4*)(th + o) = h;
        *(hx4*)(th + SF_ROWS * SF_N + o) = l;
      }
    }
  };

  // ---- prologue: EVERY global load of the launch is issued before the first LDS store (one exposed memory latency): input rows (ones
  // column at in0, zero beyond), tail constants, all nine weight chunks ----
  float xs[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + SF_NT * i, r = e >> 5, k = e & 31;
    const float x = Xg[(long)r * p.x_pitch + (k < p.in0 ? k : p.in0 - 1)];
    xs[i] = (k < p.in0 ? 1.f : 0.f) * x + (k == p.in0 ? 1.0f : 0.f);
  }
  const int nwt = p.out_dim * SF_N;                                // tail weights: up to two float4 per thread
  const bool wt_vec = (((uintptr_t)Wtg) & 15) == 0;
  f32x4 wts[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = 4 * (tid + SF_NT * i);
    const int ec = e < nwt ? e : nwt - 4;                        // clamped (rows beyond out_dim are never read back)
    if (wt_vec) wts[i] = *(const f32x4*)&Wtg[ec];
    else {
#pragma unroll
      for (int j = 0; j < 4; ++j) wts[i][j] = Wtg[ec + j];
    }
  }
  const float b1v = b1g[tid & (SF_N - 1)], btv = btg[tid < p.out_dim ? tid : 0];
  float* sW0A = sWT + SF_TWP;                        // QG: [256][8] first-layer weights of the gn differentiated input columns (out_dim == 1: the tail weights use one row of sWT)
  float w0a[8];
  if constexpr (QG) {
#pragma unroll
    for (int a = 0; a < 8; ++a) w0a[a] = (a < p.gn ? 1.f : 0.f) * W0g[(long)(tid & (SF_N - 1)) * p.in0 + p.gc0 + (a < p.gn ? a : 0)];
  }
#pragma unroll
  for (int c = 0; c <= 8; ++c) load_chunk(c);
  // sampling epilogue: thread (head row lr = tid >> 4, half = (tid >> 3) & 1, action a = tid & 7) takes output rows jr = half + 2 it of
  // its head row (no index divisions); the noise values are requested HERE, behind every other load of the launch (inside the epilogue
  // each of up to 11 iterations waited out its own round trip: 16 000 clocks for the three jobs of the [s; s'] pass)
  constexpr int SF_EIT = 8;
  auto sample_item = [&](const SampleJob& jb, int it, long& j) __attribute__((always_inline)) {
    const int jr = ((tid >> 3) & 1) + 2 * it;
    const int hb = g * SF_ROWS + (tid >> 4) - jb.head_row0;                 // base row relative to the job's first head row
    j = (long)hb * jb.rep + jr;
    return hb >= 0 && jr < jb.rep && j < jb.rows && (tid & 7) < p.A;
  };
  float ev[3][SF_EIT];
  if constexpr (!QG) {
#pragma unroll
    for (int ji = 0; ji < 3; ++ji)
#pragma unroll
      for (int it = 0; it < SF_EIT; ++it) {
        ev[ji][it] = 0.f;
        if (ji < p.njobs && 2 * it < p.job[ji].rep) {                       // (uniform)
          long j;
          const bool on = sample_item(p.job[ji], it, j);
          if (p.job[ji].eps) ev[ji][it] = (on ? 1.f : 0.f) * p.job[ji].eps[z0 * p.job[ji].eps_rs + (on ? j * p.A + (tid & 7) : 0)];
        }
      }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = tid + SF_NT * i, r = e >> 5, k = e & 31;
    if constexpr (F32) ((float*)sX)[r * SF_WF + k] = xs[i];
    else {
      hx_t hh, ll;
      orl_split1(xs[i], hh, ll);
      ((hx_t*)sX)[r * SF_WP + k] = hh;
      ((hx_t*)sX)[SF_ROWS * SF_WP + r * SF_WP + k] = ll;
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int e = 4 * (tid + SF_NT * i);
    if (e < nwt) *(f32x4*)&sWT[(e >> 8) * SF_TWP + (e & (SF_N - 1))] = wts[i];
  }
  if (tid < SF_N) sB1[tid] = b1v;
  if (tid < p.out_dim) sBT[tid] = btv;
  if constexpr (QG) {
    if (tid < SF_N) {
#pragma unroll
      for (int a = 0; a < 8; ++a) sW0A[tid * 8 + a] = w0a[a];
    }
  }
  SF_STAMP(1);
  store_chunk(0);
  __syncthreads();
  SF_STAMP(2);

  f32x4 acc[2][2];
  auto zero_acc = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) acc[s][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
  };
  zero_acc();
  constexpr float inv_sc = F32 ? 1.0f : 1.0f / ORL_WSCALE;
  float* __restrict__ H0g = p.H0 ? p.H0 + z0 * p.h0_s0 + z1 * p.h0_s1 + (long)g * SF_ROWS * SF_N : nullptr;
  float* __restrict__ H1g = p.H1 ? p.H1 + z0 * p.h1_s0 + z1 * p.h1_s1 + (long)g * SF_ROWS * SF_N : nullptr;

  // products of chunk c: operands swapped (weights first), lane (li, lq) then holds C[m = 16 s + li][n = ncol0 + 16 cb + 4 lq + r]
  auto compute = [&](int c) __attribute__((always_inline)) {
    const int buf = c & 1;
    if constexpr (F32) {
      const float* wf = (const float*)sW + (long)buf * SF_N * SF_WF;
      const float* af = c == 0 ? (const float*)sX : (const float*)sA + 32 * (c - 1);
      const int apitch = c == 0 ? SF_WF : SF_AF;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x4 fa[2], fw[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) fa[s] = *(const f32x4*)&af[(16 * s + li) * apitch + 16 * t + 4 * lq];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) fw[cb] = *(const f32x4*)&wf[(ncol0 + 16 * cb + li) * SF_WF + 16 * t + 4 * lq];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) acc[s][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[cb][e], fa[s][e], acc[s][cb], 0, 0, 0);
      }
    } else {
      const hx_t* wh = (const hx_t*)sW + (long)buf * 2 * SF_N * SF_WP;
      const hx_t* wl = wh + SF_N * SF_WP;
      const hx_t* ah = c == 0 ? (const hx_t*)sX : (const hx_t*)sA + 32 * (c - 1);
      const int apitch = c == 0 ? SF_WP : SF_AP;
      const hx_t* al = ah + (c == 0 ? SF_ROWS * SF_WP : SF_ROWS * SF_AP);
      hx8 fah[2], fal[2], fwh[2], fwl[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        fah[s] = *(const hx8*)&ah[(16 * s + li) * apitch + 8 * lq];
        fal[s] = *(const hx8*)&al[(16 * s + li) * apitch + 8 * lq];
      }
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        fwh[cb] = *(const hx8*)&wh[(ncol0 + 16 * cb + li) * SF_WP + 8 * lq];
        fwl[cb] = *(const hx8*)&wl[(ncol0 + 16 * cb + li) * SF_WP + 8 * lq];
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

  unsigned int m0 = 0;                               // QG: ReLU mask of this lane's 16 first-layer outputs, bit 4 (2 s + cb) + j
#pragma unroll
  for (int c = 0; c <= 8; ++c) {
    compute(c);
    if (c + 1 <= 8) {
      store_chunk(c + 1);                            // that buffer was last read by compute(c - 1): every wave has passed the barrier since
      if constexpr (QG) load_chunkT(c);              // ... and its registers take chunk c of the backward pass
    }
    if (c == 0) {
      // layer-0 epilogue (the bias came in through the ones column): ReLU, optional store, the A image of layer 1
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          f32x4 v = acc[s][cb] * inv_sc;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if constexpr (QG) m0 |= (v[j] > 0.f ? 1u : 0u) << (4 * (2 * s + cb) + j);
            v[j] = v[j] > 0.f ? v[j] : 0.f;
          }
          const int m = 16 * s + li, n = ncol0 + 16 * cb + 4 * lq;
          if (H0g) *(f32x4*)&H0g[(long)m * SF_N + n] = v;
          if constexpr (F32) *(f32x4*)((float*)sA + m * SF_AF + n) = v;
          else {
            hx4 h, l;
            orl_split4(v, h, l);
            *(hx4*)((hx_t*)sA + m * SF_AP + n) = h;
            *(hx4*)((hx_t*)sA + SF_ROWS * SF_AP + m * SF_AP + n) = l;
          }
        }
      zero_acc();
    }
    __syncthreads();
  }

  SF_STAMP(3);
  // ---- layer-1 epilogue: bias, ReLU, optional store; tail = dot products of the wave's 32 columns, reduced across the eight waves ----
  float* red = (float*)sW;                           // [wave][lane group][row][SF_MAXOUT]; the chunk buffers are dead (barrier above)
  f32x4 v[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const int m = 16 * s + li, n = ncol0 + 16 * cb + 4 * lq;
      f32x4 x = acc[s][cb] * inv_sc + *(const f32x4*)&sB1[n];
#pragma unroll
      for (int j = 0; j < 4; ++j) x[j] = x[j] > 0.f ? x[j] : 0.f;
      if (H1g) *(f32x4*)&H1g[(long)m * SF_N + n] = x;
      v[s][cb] = x;
    }
  if constexpr (QG) {
    // ================= QG: q = h1 . w_tail + b, then the backward pass for a unit seed dL/dq = 1 =================
    // q partial sums (this lane's eight columns) -> [wave * 4 + lq][row] in the dead input image; dz1 = w_tail (.) 1[h1 > 0] -> the A image
    // (every wave has finished reading h0 from it: barrier at the end of the last chunk)
    float* qred = (float*)sX;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float pd = 0.f;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const int m = 16 * s + li, n = ncol0 + 16 * cb + 4 * lq;
        const f32x4 w = *(const f32x4*)&sWT[n];
        pd += (v[s][cb][0] * w[0] + v[s][cb][1] * w[1]) + (v[s][cb][2] * w[2] + v[s][cb][3] * w[3]);
        f32x4 d;
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = v[s][cb][j] > 0.f ? w[j] : 0.f;
        if constexpr (F32) *(f32x4*)((float*)sA + m * SF_AF + n) = d;
        else {
          hx4 h, l;
          orl_split4(d * ORL_WSCALE, h, l);          // w_tail is weight-sized: the static weight scale
          *(hx4*)((hx_t*)sA + m * SF_AP + n) = h;
          *(hx4*)((hx_t*)sA + SF_ROWS * SF_AP + m * SF_AP + n) = l;
        }
      }
      qred[(wave * 4 + lq) * SF_ROWS + 16 * s + li] = pd;
    }
    SF_STAMP(4);
    store_chunkT(0);
    __syncthreads();
    SF_STAMP(5);
    zero_acc();
    // dz0[m][j] = sum_n dz1[m][n] W1[n][j]: chunk c covers n = 32 c .. 32 c + 31.  Same accumulator layout as the forward (lane holds
    // C[m = 16 s + li][j = ncol0 + 16 cb + 4 lq + r]), so the first layer's mask bits apply to the registers as they are.
    auto computeT = [&](int c) __attribute__((always_inline)) {
      const int buf = c & 1;
      if constexpr (F32) {
        const float* tf = (const float*)sW + (long)buf * SF_ROWS * SF_TF;
        const float* af = (const float*)sA + 32 * c;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          f32x4 fa[2];
          float fw[2][4];
#pragma unroll
          for (int s = 0; s < 2; ++s) fa[s] = *(const f32x4*)&af[(16 * s + li) * SF_AF + 16 * t + 4 * lq];
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int e = 0; e < 4; ++e) fw[cb][e] = tf[(16 * t + 4 * lq + e) * SF_TF + ncol0 + 16 * cb + li];
#pragma unroll
          for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
              for (int cb = 0; cb < 2; ++cb) acc[s][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(fw[cb][e], fa[s][e], acc[s][cb], 0, 0, 0);
        }
      } else {
        const hx_t* th = (const hx_t*)sW + (long)buf * 2 * SF_ROWS * SF_N;
        const hx_t* tl = th + SF_ROWS * SF_N;
        const hx_t* ah = (const hx_t*)sA + 32 * c;
        const hx_t* al = ah + SF_ROWS * SF_AP;
        // the 8 contraction values of a lane: n = 32 c + 4 lq + {0..3} and 32 c + 16 + 4 lq + {0..3} (what two transposing reads deliver;
        // the k order inside one MFMA is free as long as both operands use the same one)
        hx8 fah[2], fal[2], fwh[2], fwl[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int o = (16 * s + li) * SF_AP + 4 * lq;
          fah[s] = __builtin_shufflevector(*(const hx4*)&ah[o], *(const hx4*)&ah[o + 16], 0, 1, 2, 3, 4, 5, 6, 7);
          fal[s] = __builtin_shufflevector(*(const hx4*)&al[o], *(const hx4*)&al[o + 16], 0, 1, 2, 3, 4, 5, 6, 7);
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          fwh[cb] = sf_cat(sf_tr(th, 0, ncol0 + 16 * cb, lane), sf_tr(th, 16, ncol0 + 16 * cb, lane));
          fwl[cb] = sf_cat(sf_tr(tl, 0, ncol0 + 16 * cb, lane), sf_tr(tl, 16, ncol0 + 16 * cb, lane));
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
      computeT(c);
      if (c + 1 < 8) store_chunkT(c + 1);
      __syncthreads();
    }
    SF_STAMP(6);
    // dz0 = (.) 1[h0 > 0]; G[m][a] = sum_j dz0[m][j] W0[j][gc0 + a]: per-lane partial sums over its eight columns, reduced in a fixed order
    constexpr float inv_sc2 = F32 ? 1.0f : 1.0f / (ORL_WSCALE * ORL_WSCALE);
    float* gred = (float*)sW;                        // [wave * 4 + lq][row][8]; the chunk buffers are dead (barrier above)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float pg[8];
#pragma unroll
      for (int a = 0; a < 8; ++a) pg[a] = 0.f;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float d = ((m0 >> (4 * (2 * s + cb) + j)) & 1u) ? acc[s][cb][j] * inv_sc2 : 0.f;
          const float* w = &sW0A[(ncol0 + 16 * cb + 4 * lq + j) * 8];
          const f32x4 wa = *(const f32x4*)w, wb = *(const f32x4*)(w + 4);
#pragma unroll
          for (int a = 0; a < 4; ++a) { pg[a] += d * wa[a]; pg[4 + a] += d * wb[a]; }
        }
      float* o = &gred[(((wave * 4 + lq) * SF_ROWS) + 16 * s + li) * 8];
      *(f32x4*)o = (f32x4){pg[0], pg[1], pg[2], pg[3]};
      *(f32x4*)(o + 4) = (f32x4){pg[4], pg[5], pg[6], pg[7]};
    }
    __syncthreads();
    {
      const int r = tid >> 4, a = tid & 15;
      if (a < p.gn) {
        float x = 0.f;
#pragma unroll
        for (int w = 0; w < 32; ++w) x += gred[(w * SF_ROWS + r) * 8 + a];      // fixed order
        (p.G + z0 * p.g_s0 + z1 * p.g_s1)[((long)g * SF_ROWS + r) * p.g_pitch + a] = x;
      } else if (a == 8) {
        float x = sBT[0];
#pragma unroll
        for (int w = 0; w < 32; ++w) x += qred[w * SF_ROWS + r];                // fixed order
        (p.OUT + z0 * p.o_s0 + z1 * p.o_s1)[((long)g * SF_ROWS + r) * p.o_pitch] = x;
      }
    }
    SF_STAMP(7);
    return;
  }
  float* shead = (float*)sA;                         // [SF_ROWS][SF_MAXOUT] tail outputs of this row group (the A image is dead: barriers above)
  if (p.out_dim > 1) {
    // multi-output head (the actor's [mu | log sigma]): out[m][o] = sum_n h1[m][n] Wt[o][n] on the exact fp32 MFMA.  The wave's own 32
    // columns are the contraction: step (cb, r) takes k slot lq <-> n = ncol0 + 16 cb + 4 lq + r, so the B operand B[k][j = m] is this lane's
    // own accumulator register v[s][cb][r] and A[i = o][k] one float of the padded tail-weight image.  16 MFMAs per wave; the eight waves'
    // partial sums meet in LDS.  (As per-lane dot products the 12 outputs took 8 200 clocks: 48 16-byte LDS reads per lane.)
    f32x4 tacc[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) tacc[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = sWT[li * SF_TWP + ncol0 + 16 * cb + 4 * lq + r];
#pragma unroll
        for (int s = 0; s < 2; ++s) tacc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, v[s][cb][r], tacc[s], 0, 0, 0);
      }
    // lane holds D[o = 4 lq + r][m = 16 s + li]
#pragma unroll
    for (int s = 0; s < 2; ++s) *(f32x4*)&red[((wave * SF_ROWS) + 16 * s + li) * SF_MAXOUT + 4 * lq] = tacc[s];
    SF_STAMP(4);
    __syncthreads();
    const int r = tid >> 4, o = tid & 15;
    if (o < p.out_dim) {
      float a = sBT[o];
#pragma unroll
      for (int w = 0; w < 8; ++w) a += red[(w * SF_ROWS + r) * SF_MAXOUT + o];       // fixed order
      (p.OUT + z0 * p.o_s0 + z1 * p.o_s1)[((long)g * SF_ROWS + r) * p.o_pitch + o] = a;
      shead[r * SF_MAXOUT + o] = a;
    }
  } else {
    // single output: every lane leaves the partial sum of its eight columns in LDS; 32 partials per row = 8 waves x 4 lane groups, summed
    // in a fixed order below
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float pd = 0.f;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        const f32x4 w = *(const f32x4*)&sWT[ncol0 + 16 * cb + 4 * lq];
        pd += (v[s][cb][0] * w[0] + v[s][cb][1] * w[1]) + (v[s][cb][2] * w[2] + v[s][cb][3] * w[3]);
      }
      red[((wave * 4 + lq) * SF_ROWS) + 16 * s + li] = pd;
    }
    SF_STAMP(4);
    __syncthreads();
    if (tid < SF_ROWS) {
      float a = sBT[0];
#pragma unroll
      for (int w = 0; w < 32; ++w) a += red[w * SF_ROWS + tid];                       // fixed order
      (p.OUT + z0 * p.o_s0 + z1 * p.o_s1)[((long)g * SF_ROWS + tid) * p.o_pitch] = a;
    }
  }
  SF_STAMP(5);
  if (p.njobs == 0) return;                          // (kernel-uniform)
  // ---- sampling epilogue (small_fwd.h): every job that draws from head rows of this group; k_tanh_sample's arithmetic and summation order ----
  if constexpr (!QG) {
    __syncthreads();
    const int A = p.A;
#pragma unroll
    for (int ji = 0; ji < 3; ++ji) {
      if (ji >= p.njobs) break;                                             // (uniform)
      const SampleJob& jb = p.job[ji];
#pragma unroll
      for (int it = 0; it < SF_EIT; ++it) {
        if (2 * it >= jb.rep) break;                                        // (uniform) whole 8-lane groups enter or leave together
        const int lr = tid >> 4, a = tid & 7;
        long j;
        const bool on = sample_item(jb, it, j);
        float term = 0.f;
        if (on) {
          float act;
          term = orl_tanh_sample(shead[lr * SF_MAXOUT + a], shead[lr * SF_MAXOUT + A + a], ev[ji][it], act);
          jb.dst[z0 * jb.dst_rs + (jb.dst_row0 + j) * jb.dst_pitch + jb.dst_col + a] = act;
        }
        for (int o = 4; o > 0; o >>= 1) term += __shfl_down(term, o, 8);
        if (on && a == 0 && jb.logp) jb.logp[z0 * jb.logp_rs + j] = term;
      }
    }
  }
  SF_STAMP(6);
}

hipError_t launch_small_fwd(const SmallFwdP& p, int nz, hipStream_t st) {
  static const hipError_t attr_err = [] {
    hipError_t e = hipFuncSetAttribute((const void*)small_fwd_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sf_lds_bytes<false>());
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)small_fwd_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sf_lds_bytes<true>());
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)small_fwd_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sf_lds_bytes<false>());
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)small_fwd_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sf_lds_bytes<true>());
    return e;
  }();
  if (attr_err != hipSuccess) return attr_err;
  const dim3 grid(p.M / SF_ROWS, 1, nz), block(SF_NT);
  if (p.G) {
    if (p.f32) hipLaunchKernelGGL((small_fwd_kernel<true, true>), grid, block, sf_lds_bytes<true>(), st, p);
    else hipLaunchKernelGGL((small_fwd_kernel<false, true>), grid, block, sf_lds_bytes<false>(), st, p);
  } else if (p.f32) hipLaunchKernelGGL((small_fwd_kernel<true, false>), grid, block, sf_lds_bytes<true>(), st, p);
  else hipLaunchKernelGGL((small_fwd_kernel<false, false>), grid, block, sf_lds_bytes<false>(), st, p);
  return hipGetLastError();
}

}  // namespace orl
