// kernels.h — non-GEMM device kernels of the update engine: replay gather + device RNG, tanh-Gaussian
// sampling and its backward, loss / gradient-seed kernels, fused Adam + split-K slab reduce + Polyak.
// Every kernel is batched over runs (and nets where it applies) through blockIdx.y / blockIdx.z.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sample.h"
#include "scalars.h"

namespace orl {

#ifndef ORL_HEALTH_NONFINITE_LOSS      // (include/orl_engine.h defines the same values for the C ABI)
#define ORL_HEALTH_NONFINITE_LOSS 1
#define ORL_HEALTH_NONFINITE_GRAD 2
#define ORL_HEALTH_SPLIT_RANGE 4
#endif

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 counter RNG (device sampling of indices / noise in orl_learn_n)
// ------------------------------------------------------------------------------------------------
struct Philox {
  uint32_t k0, k1;
  __device__ Philox(uint64_t seed) : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)) {}
  __device__ void operator()(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t (&out)[4]) const {
    uint32_t a = k0, b = k1;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
      const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ a, n1 = (uint32_t)p1;
      const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ b, n3 = (uint32_t)p0;
      c0 = n0; c1 = n1; c2 = n2; c3 = n3;
      a += 0x9E3779B9u; b += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
  }
};
__device__ inline float u01(uint32_t x) { return ((x >> 8) + 0.5f) * (1.0f / 16777216.0f); }  // (0,1)
__device__ inline void box_muller(uint32_t x, uint32_t y, float& n0, float& n1) {
  const float r = sqrtf(-2.0f * logf(u01(x)));
  const float t = 6.28318530717958647692f * u01(y);
  n0 = r * cosf(t); n1 = r * sinf(t);
}

// ------------------------------------------------------------------------------------------------
// device step counter: Adam's t and the Philox offsets derive from it
// ------------------------------------------------------------------------------------------------
__global__ void k_tick(unsigned long long* gstep) { *gstep += 1ull; }

// ------------------------------------------------------------------------------------------------
// split precision: dynamic power-of-two scale of a backward pass.  The fp16 hi + lo planes of the split multiply cover 22 significand bits
// only while hi = half(x * s) is a normal fp16 number, so the gradient matrices of one backward pass (its seed dL/d(tail output) and every
// dz derived from it) enter the MFMAs times s = 2^(3 - floor(log2(max |seed|))): the seed's largest entry lands in [8, 16), which leaves
// 2^12 of headroom for the growth of a dz through the layers below and 2^-17 of room below before an entry's hi plane turns subnormal
// (it then still carries 2^-24 absolute = 2^-27 of the largest entry).  One workgroup per run; max over every net's seed; exact.
// ------------------------------------------------------------------------------------------------
// block-wide max over 256 threads; every thread receives it (NaNs are dropped by fmaxf)
__device__ inline float block_max256(float v, float* sh4) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(sh4[0], sh4[1]), fmaxf(sh4[2], sh4[3]));
}
// Seed kernels that run as ONE workgroup per run publish the scale themselves (`gs_out` [R], null = off): no extra launch per
// backward pass (each k_grad_scale node costs the ~4 us every dependent kernel node costs: 10 % of a step at one run per engine).
__device__ inline void grad_scale_publish(float amax_thread, float* sh4, float* gs_out, int r) {
  if (!gs_out) return;                                          // uniform
  const float a = block_max256(amax_thread, sh4);
  if (threadIdx.x == 0) gs_out[r] = orl_pow2_scale(a);
}
struct GradScaleP { const float* seed; long rs, cs; int rows, cols, pitch, nets; float* out; };
__global__ void k_grad_scale(GradScaleP p) {
  __shared__ float sh[256];
  const int r = blockIdx.x;
  const float* base = p.seed + (long)r * p.rs;
  const long per_net = (long)p.rows * p.cols, n = per_net * p.nets;
  float m = 0.f;
  for (long e = threadIdx.x; e < n; e += 256) {
    const long z = e / per_net, w = e - z * per_net;
    const long row = w / p.cols, c = w - row * p.cols;
    m = fmaxf(m, fabsf(base[z * p.cs + row * p.pitch + c]));      // fmaxf drops NaNs: a diverged run keeps scale 1 and shows its NaNs in the losses
  }
  sh[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) sh[threadIdx.x] = fmaxf(sh[threadIdx.x], sh[threadIdx.x + o]);
    __syncthreads();
  }
  if (threadIdx.x == 0) p.out[r] = orl_pow2_scale(sh[0]);
}

// ------------------------------------------------------------------------------------------------
// replay gather (buffer.py:96-106).  Dataset = SoA in HBM, rows padded to 16 B (obs/next_obs pitch OP,
// act pitch AP).  idx == nullptr -> Philox indices (np.random.randint(0, size, B) restated on device).
// One thread per (row, column) of the widest array; consecutive lanes read consecutive floats of a row
// (coalesced within the row) and rows are independent random 64-128 B segments.
// grid (ceil(B*W/256), R) with W = max(OP, AP).
// ------------------------------------------------------------------------------------------------
struct GatherP {
  const float *obs, *nobs, *act, *rew, *term;  // dataset [n][OP], [n][OP], [n][AP], [n], [n]
  long n;
  int OP, AP, od, ad, B, W;
  const long long* idx; long idx_rs;           // [R][B] or null
  float *b_obs, *b_nobs, *b_act, *b_rew, *b_term;  // destinations
  long obs_rs, nobs_rs, act_rs, rew_rs, term_rs;   // run strides of the destinations
  int d_op, d_ap;                                  // destination row pitches (>= od / ad; extra columns zeroed)
  unsigned long long seed;
  const unsigned long long* gstep;
  unsigned long long counter;                      // used when gstep == nullptr
};
__global__ void k_gather(GatherP p) {
  const int r = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int row = t / p.W, c = t - row * p.W;
  if (row >= p.B) return;
  long j;
  if (p.idx) j = p.idx[(long)r * p.idx_rs + row];
  else {
    const unsigned long long ctr = p.gstep ? *p.gstep : p.counter;
    Philox ph(p.seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(r + 1));
    uint32_t o[4];
    ph((uint32_t)row, 0x51u, (uint32_t)ctr, 0x1D5u ^ (uint32_t)(ctr >> 32), o);
    j = (long)(((unsigned long long)o[0] * (unsigned long long)p.n) >> 32);
  }
  if (c < p.d_op) {
    const float vo = c < p.od ? p.obs[j * p.OP + c] : 0.f, vn = c < p.od ? p.nobs[j * p.OP + c] : 0.f;
    p.b_obs[(long)r * p.obs_rs + (long)row * p.d_op + c] = vo;
    p.b_nobs[(long)r * p.nobs_rs + (long)row * p.d_op + c] = vn;
  }
  if (c < p.d_ap) p.b_act[(long)r * p.act_rs + (long)row * p.d_ap + c] = c < p.ad ? p.act[j * p.AP + c] : 0.f;
  if (c == 0) { p.b_rew[(long)r * p.rew_rs + row] = p.rew[j]; p.b_term[(long)r * p.term_rs + row] = p.term[j]; }
}

// normalize_obs (buffer.py:88-94): per-column mean / population std in double, then in-place scaling
__global__ void k_colstats(const float* x, long n, int pitch, int dim, double* sums /*[2*dim]*/) {
  __shared__ double sh[2][256];
  const int c = blockIdx.y;
  double s = 0.0, q = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const double v = x[i * pitch + c];
    s += v; q += v * v;
  }
  sh[0][threadIdx.x] = s; sh[1][threadIdx.x] = q;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) { sh[0][threadIdx.x] += sh[0][threadIdx.x + o]; sh[1][threadIdx.x] += sh[1][threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { atomicAdd(&sums[c], sh[0][0]); atomicAdd(&sums[dim + c], sh[1][0]); }
}
__global__ void k_normalize(float* x, float* y, long n, int pitch, int dim, const float* mean, const float* stdv) {
  const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long i = t / dim; const int c = (int)(t - i * dim);
  if (i >= n) return;
  x[i * pitch + c] = (x[i * pitch + c] - mean[c]) / stdv[c];
  y[i * pitch + c] = (y[i * pitch + c] - mean[c]) / stdv[c];
}

__global__ void k_fill(float* p, long n, float v) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}

// device noise: fills `n` floats per run with N(0,1) (kind 0), U[lo,hi) (kind 1) or dropout keep masks 1[U(0,1) < lo] (kind 2, lo = 1 - p)
__global__ void k_noise(float* out, long n_per_run, int kind, float lo, float hi, unsigned long long seed,
                        const unsigned long long* gstep, uint32_t stream_id) {
  const int r = blockIdx.y;
  const long i4 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 >= n_per_run) return;
  Philox ph(seed);
  uint32_t o[4];
  ph((uint32_t)(i4 >> 2), (uint32_t)r | (stream_id << 16), (uint32_t)(*gstep), 0xA5u ^ (uint32_t)((*gstep) >> 32), o);
  float v[4];
  if (kind == 0) { box_muller(o[0], o[1], v[0], v[1]); box_muller(o[2], o[3], v[2], v[3]); }
  else if (kind == 1) { for (int k = 0; k < 4; ++k) v[k] = lo + (hi - lo) * u01(o[k]); }
  else { for (int k = 0; k < 4; ++k) v[k] = u01(o[k]) < lo ? 1.0f : 0.0f; }
  float* dst = out + (long)r * n_per_run;
  for (int k = 0; k < 4; ++k) if (i4 + k < n_per_run) dst[i4 + k] = v[k];
}

// nn.Dropout in training mode (nets/mlp.py:22-23) on a hidden activation, in place: h = keep ? h / (1 - p) : 0 with the keep mask of the
// step (0 / 1 floats [R][rows][cols]: teacher-forced from the reference's draws or k_noise kind 2); `mask == nullptr`: plain scaling by `s`
// (the 1 / (1 - p) factor of the backward pass on a masked dz).  grid (ceil(rows * cols / 256), nets, R)
__global__ void k_dropout(float* h, long h_rs, long h_cs, int pitch, const float* mask, long m_rs, int rows, int cols, float s) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)rows * cols) return;
  const int row = (int)(e / cols), c = (int)(e - (long)row * cols);
  float* x = h + blockIdx.z * h_rs + blockIdx.y * h_cs + (long)row * pitch + c;
  const float k = mask ? mask[blockIdx.z * m_rs + e] : 1.0f;
  *x = *x * k * s;
}

// ------------------------------------------------------------------------------------------------
// k_prepare: ONE launch that materialises every input of a step — replay gather (batch slots), critic-input
// matrices X = (obs | act | 0-pad) with repeated rows, and all noise arrays — instead of 1 gather + N noise +
// M assemble launches.  A job table (<= 20 entries, passed by value) describes the outputs; one thread per
// output element (four elements for the noise jobs: one Philox4x32 call each).  Observation-like sources are read either
// straight from the HBM dataset through the step's minibatch indices (sampling mode: every consumer draws its row's index from the same Philox counter, orl_draw_index) or from the
// batch slots (teacher-forced mode).
// ------------------------------------------------------------------------------------------------
enum { PS_OBS = 0, PS_NOBS = 1, PS_ACT = 2, PS_REW = 3, PS_TERM = 4, PS_NORMAL = 5, PS_UNIFORM = 6, PS_BUF = 7, PS_ZERO = 8 };
struct PrepJob {
  int block_end;        // exclusive end of this job's range of 256-thread blocks (per run): a block serves ONE job, so the job lookup
                        // is a scalar loop; a thread's unit = one element, or four for the noise jobs
  int units;            // work units of the job
  int vec4;             // gather jobs whose source rows and destination rows are 16-byte aligned: a unit = four consecutive columns
  float* dst; long dst_rs; int dst_pitch, dst_row0, dst_col0;
  int rows, width;      // output region: rows x width
  int src;              // PS_*
  int rep, mod;         // source row = src_row0 + (mod ? row % mod : row) / rep
  int src_row0;
  int ncopy;            // columns [0, ncopy) come from the source, the rest of `width` is zero
  const float* buf; long buf_rs; int buf_pitch;   // PS_BUF source
  unsigned stream_id;   // noise stream
};
struct PrepP {
  PrepJob job[20];
  int njobs;
  int blocks;           // blocks per run
  // dataset (sampling mode) or null -> batch slots
  const float *d_obs, *d_nobs, *d_act, *d_rew, *d_term; long n; int OP, AP;
  const long long* idx; long idx_rs;                  // the minibatch indices of this step [R][B] (host-supplied), or, with `draw`:
  int draw;                                           // indices are drawn here (orl_draw_index: one Philox call per consumer thread, same value for
  long long* idx_out;                                 // every consumer of a batch row) and recorded in idx_out [R][B] by the rewards job
  const float *b_obs, *b_nobs, *b_act, *b_rew, *b_term; long bo_rs, ba_rs, br_rs; int b_op, b_ap;
  int B;
  unsigned long long seed; const unsigned long long* gstep; float lo, hi;
  // step counter without a k_tick node (CQL): this kernel reads `gstep` = the PRE cell, which the step's loss kernel advances once every
  // reader of the step is behind it (k_cql_loss_rows), and publishes it to the cell every later kernel of the step reads.  null: off.
  unsigned long long* gstep_publish;
};
// np.random.randint(0, size, B) (buffer.py:98) on the device: one Philox call per (run, batch row), drawn ONCE per step and shared by
// every consumer of that row (batch slots, actor / critic input rows and their N-fold repeats)
__device__ inline long long orl_draw_index(unsigned long long seed, int r, int b, unsigned long long ctr, long n) {
  Philox ph(seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(r + 1));
  uint32_t o[4];
  ph((uint32_t)b, 0x51u, (uint32_t)ctr, 0x1D5u ^ (uint32_t)(ctr >> 32), o);
  return (long long)(((unsigned long long)o[0] * (unsigned long long)n) >> 32);
}
__global__ void k_prepare(PrepP p) {
  const int r = blockIdx.y;
  if (p.gstep_publish && blockIdx.x == 0 && r == 0 && threadIdx.x == 0) *p.gstep_publish = *p.gstep;
  int ji = 0;
  while (ji < p.njobs - 1 && (int)blockIdx.x >= p.job[ji].block_end) ++ji;       // block-uniform: scalar compares
  const PrepJob& jb = p.job[ji];
  const int u = ((int)blockIdx.x - (ji ? p.job[ji - 1].block_end : 0)) * 256 + (int)threadIdx.x;
  if (u >= jb.units) return;
  if (jb.src == PS_NORMAL || jb.src == PS_UNIFORM) {
    // one Philox call yields the four values of elements 4u .. 4u + 3 of the job
    const unsigned long long ctr = p.gstep ? *p.gstep : 0ull;
    Philox ph(p.seed);
    uint32_t o[4];
    ph((uint32_t)u, (uint32_t)r | (jb.stream_id << 16), (uint32_t)ctr, 0xA5u ^ (uint32_t)(ctr >> 32), o);
    float v[4];
    if (jb.src == PS_NORMAL) { box_muller(o[0], o[1], v[0], v[1]); box_muller(o[2], o[3], v[2], v[3]); }
    else {
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = p.lo + (p.hi - p.lo) * u01(o[k]);
    }
    const int n_el = jb.rows * jb.width, e0 = 4 * u;
    int row = e0 / jb.width, col = e0 - row * jb.width;
    float* d = jb.dst + (long)r * jb.dst_rs + (long)jb.dst_row0 * jb.dst_pitch + jb.dst_col0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (e0 + k < n_el) d[(long)row * jb.dst_pitch + col] = v[k];
      if (++col == jb.width) { col = 0; ++row; }
    }
    return;
  }
  if (jb.vec4) {
    // four columns per thread: one 16-byte load from the (zero-padded, 16-byte aligned) source row, one 16-byte store -- or scalar
    // stores for the chunk that holds the row's last columns (the columns behind them belong to another job)
    const int chunks = (jb.width + 3) >> 2;
    const int row = u / chunks, c4 = (u - row * chunks) << 2;
    const int b = jb.src_row0 + (jb.mod ? row % jb.mod : row) / jb.rep;
    const float* srow;
    if (p.d_obs) {
      const long j = p.draw ? (long)orl_draw_index(p.seed, r, b, *p.gstep, p.n) : (long)p.idx[(long)r * p.idx_rs + b];
      srow = (jb.src == PS_OBS ? p.d_obs : (jb.src == PS_NOBS ? p.d_nobs : p.d_act)) + j * (jb.src == PS_ACT ? p.AP : p.OP);
    } else {
      srow = jb.src == PS_ACT ? p.b_act + (long)r * p.ba_rs + (long)b * p.b_ap
                              : (jb.src == PS_OBS ? p.b_obs : p.b_nobs) + (long)r * p.bo_rs + (long)b * p.b_op;
    }
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 v = *(const f4*)(srow + c4);
#pragma unroll
    for (int k = 0; k < 4; ++k) if (c4 + k >= jb.ncopy) v[k] = 0.f;
    float* d = jb.dst + (long)r * jb.dst_rs + (long)(jb.dst_row0 + row) * jb.dst_pitch + jb.dst_col0 + c4;
    if (c4 + 4 <= jb.width) *(f4*)d = v;
    else {
#pragma unroll
      for (int k = 0; k < 4; ++k) if (c4 + k < jb.width) d[k] = v[k];
    }
    return;
  }
  const int row = u / jb.width, col = u - row * jb.width;
  float v = 0.f;
  if (jb.src == PS_BUF) {
    if (col < jb.ncopy) v = jb.buf[(long)r * jb.buf_rs + (long)(jb.src_row0 + (jb.mod ? row % jb.mod : row) / jb.rep) * jb.buf_pitch + col];
  } else if (jb.src != PS_ZERO && col < jb.ncopy) {
    const int b = jb.src_row0 + (jb.mod ? row % jb.mod : row) / jb.rep;
    if (p.d_obs) {
      const long j = p.draw ? (long)orl_draw_index(p.seed, r, b, *p.gstep, p.n) : (long)p.idx[(long)r * p.idx_rs + b];
      switch (jb.src) {
        case PS_OBS: v = p.d_obs[j * p.OP + col]; break;
        case PS_NOBS: v = p.d_nobs[j * p.OP + col]; break;
        case PS_ACT: v = p.d_act[j * p.AP + col]; break;
        case PS_REW: v = p.d_rew[j]; if (p.draw && p.idx_out) p.idx_out[(long)r * p.idx_rs + b] = j; break;
        default: v = p.d_term[j]; break;
      }
    } else {
      switch (jb.src) {
        case PS_OBS: v = p.b_obs[(long)r * p.bo_rs + (long)b * p.b_op + col]; break;
        case PS_NOBS: v = p.b_nobs[(long)r * p.bo_rs + (long)b * p.b_op + col]; break;
        case PS_ACT: v = p.b_act[(long)r * p.ba_rs + (long)b * p.b_ap + col]; break;
        case PS_REW: v = p.b_rew[(long)r * p.br_rs + b]; break;
        default: v = p.b_term[(long)r * p.br_rs + b]; break;
      }
    }
  }
  jb.dst[(long)r * jb.dst_rs + (long)(jb.dst_row0 + row) * jb.dst_pitch + jb.dst_col0 + col] = v;
}

// ------------------------------------------------------------------------------------------------
// critic-input assembly: X[row] = (obs_src[row / rep], act_src[row]) with pitch XP (zero padded)
// grid (ceil(rows/256), R)
// ------------------------------------------------------------------------------------------------
struct AssembleP {
  const float* obs; long obs_rs; int OP, od;    // [R][B][OP]
  const float* act; long act_rs; int apitch, ad; // [R][rows][apitch] or null (leave action cols untouched)
  float* X; long x_rs; int XP;                   // [R][rows_total][XP]; writes rows [row0, row0+rows)
  int row0, rows, rep;
};
__global__ void k_assemble(AssembleP p) {
  const int r = blockIdx.y;
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= p.rows) return;
  const float* o = p.obs + (long)r * p.obs_rs + (long)(row / p.rep) * p.OP;
  float* x = p.X + (long)r * p.x_rs + (long)(p.row0 + row) * p.XP;
  for (int c = 0; c < p.od; ++c) x[c] = o[c];
  if (p.act) {
    const float* a = p.act + (long)r * p.act_rs + (long)row * p.apitch;
    for (int c = 0; c < p.ad; ++c) x[p.od + c] = a[c];
  }
  for (int c = p.od + p.ad; c < p.XP; ++c) x[c] = 0.f;
}

// ------------------------------------------------------------------------------------------------
// tanh-Gaussian sampling (dist_module.py:117-127, :17-42): head = [mu | log_sigma_raw] per base row.
// One thread per output row; up to 3 jobs per launch (blockIdx.y), runs in blockIdx.z.
// ------------------------------------------------------------------------------------------------
struct SampleP {
  const float* head; long head_rs; int A;   // [R][rows_head][2A]
  SampleJob job[3];
};
__global__ void k_tanh_sample(SampleP p) {
  // one thread per (output row, action): the A actions of a row sit in AG = 8 / 16 / 32 consecutive lanes (lanes >= A idle) and the
  // row's log-probability is a shuffle reduction over that lane group (one thread per row looped over A with four transcendental
  // calls per action and used a sixth of the lanes' parallelism)
  const SampleJob& jb = p.job[blockIdx.y];
  const int r = blockIdx.z;
  const int A = p.A;
  const int AG = A <= 8 ? 8 : (A <= 16 ? 16 : 32);
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = t / AG, a = t - j * AG;
  const bool on = j < jb.rows && a < A;
  float term = 0.f;
  if (on) {
    const float* h = p.head + (long)r * p.head_rs + (long)(jb.head_row0 + j / jb.rep) * (2 * A);
    float act;
    term = orl_tanh_sample(h[a], h[A + a], jb.eps ? jb.eps[(long)r * jb.eps_rs + (long)j * A + a] : 0.f, act);
    jb.dst[(long)r * jb.dst_rs + (long)(jb.dst_row0 + j) * jb.dst_pitch + jb.dst_col + a] = act;
  }
  // fixed-order tree over the lane group (the reference sums the A terms of logp and of the Jacobian separately, dist_module.py:27-31;
  // the difference is fp32 reassociation of <= 32 terms)
  for (int o = AG >> 1; o > 0; o >>= 1) term += __shfl_down(term, o, AG);
  if (on && a == 0 && jb.logp) jb.logp[(long)r * jb.logp_rs + j] = term;
}

// block-wide sum over 256 threads (4 waves of 64)
__device__ inline float block_sum256(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  const float t = sh[0] + sh[1] + sh[2] + sh[3];
  return t;
}

// ------------------------------------------------------------------------------------------------
// SAC-style actor loss + temperature step (cql.py:92-106, edac.py:96-110)
//   L = mean(alpha*logp - min_c q_c) ; dq_c = -1/B routed to the min ; alpha Adam step.
// grid (R), block 256, loops over B.  K critics (2 for CQL).
// ------------------------------------------------------------------------------------------------
struct ActorLossP {
  const float* qa; long qa_rs, qa_cs;      // [R][K][B]
  float* dqa;                              // same layout
  const float* logp; long logp_rs;         // [R][B]
  int B, K;
  RunScalars* sc; const Hyper* hy;
  int auto_alpha; float fixed_alpha, target_entropy; int clamp_alpha01;
  float b1, b2, eps;
  const unsigned long long* gstep;
  float* metrics_last; float* metrics_sum; int nm;  // [R][nm]
  int m_actor, m_alpha_loss, m_alpha;              // metric slots (-1 = none)
};
__global__ void k_actor_loss(ActorLossP p) {
  __shared__ float sh[4];
  const int r = blockIdx.x;
  RunScalars& sc = p.sc[r];
  const float alpha = p.auto_alpha ? sc.alpha : p.fixed_alpha;
  float s_loss = 0.f, s_lp = 0.f;
  const float gq = -1.0f / (float)p.B;
  for (int b = threadIdx.x; b < p.B; b += 256) {
    const float* q = p.qa + (long)r * p.qa_rs + b;
    float qmin = q[0];
    for (int c = 1; c < p.K; ++c) qmin = fminf(qmin, q[(long)c * p.qa_cs]);
    int nmin = 0;
    for (int c = 0; c < p.K; ++c) nmin += (q[(long)c * p.qa_cs] == qmin);
    float* dq = p.dqa + (long)r * p.qa_rs + b;
    for (int c = 0; c < p.K; ++c) dq[(long)c * p.qa_cs] = (q[(long)c * p.qa_cs] == qmin) ? gq / (float)nmin : 0.f;
    const float lp = p.logp[(long)r * p.logp_rs + b];
    s_loss += alpha * lp - qmin;
    s_lp += lp;
  }
  s_loss = block_sum256(s_loss, sh);
  s_lp = block_sum256(s_lp, sh);
  if (threadIdx.x == 0) {
    float* ml = p.metrics_last + (long)r * p.nm;
    float* ms = p.metrics_sum + (long)r * p.nm;
    const float loss = s_loss / (float)p.B;
    ml[p.m_actor] = loss; ms[p.m_actor] += loss;
    sc.alpha_bwd = alpha;
    if (p.auto_alpha) {
      // alpha_loss = -(log_alpha * (logp + target_entropy)).mean()
      const float mean_t = s_lp / (float)p.B + p.target_entropy;
      const float aloss = -(sc.log_alpha * mean_t);
      adam_scalar(sc.log_alpha, sc.la_m, sc.la_v, -mean_t, p.hy->lr[2], p.b1, p.b2, p.eps, *p.gstep + 1ull);
      float na = expf(sc.log_alpha);
      if (p.clamp_alpha01) na = fminf(fmaxf(na, 0.f), 1.f);
      sc.alpha = na;
      ml[p.m_alpha_loss] = aloss; ms[p.m_alpha_loss] += aloss;
      ml[p.m_alpha] = na; ms[p.m_alpha] += na;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward of the tanh-Gaussian head for the actor loss (rsample path; oracle/nn.py tanh_gauss_bwd)
//   da = sum_c dx_c[:, action cols] ; dlogp = alpha/B
// grid (ceil(B/256), R)
// ------------------------------------------------------------------------------------------------
struct HeadBwdP {
  const float* dxa; long dxa_rs, dxa_cs; int dxa_pitch; int K;  // [R][K][B][pitch] action-column grads
  const float* dqa; long dqa_rs, dqa_cs;                           // non-null: dxa holds UNIT-seed gradients dq_c / da (the one-launch forward +
                                                                   // backward, small_fwd.h QG mode); da = sum_c dqa[c][b] dxa[c][b]
  const float* head; long head_rs;                                // [R][B][2A]
  const float* eps; long eps_rs;                                  // [R][B][A]
  const float* xa; long xa_rs; int XP, od;                        // actions live in Xa[:, od:]
  float* dhead; long dhead_rs;                                    // [R][B][2A]
  const RunScalars* sc; int auto_alpha; float fixed_alpha;
  int B, A;
  float* gs_out;                                                  // split precision, B <= 256 (one block per run): dynamic scale of dhead
};
__global__ void k_head_bwd(HeadBwdP p) {
  __shared__ float sh[4];
  const int r = blockIdx.y;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  float amax = 0.f;
  if (b < p.B) {
  const int A = p.A;
  const float alpha = p.auto_alpha ? p.sc[r].alpha_bwd : p.fixed_alpha;
  const float dlogp = alpha / (float)p.B;
  const float* h = p.head + (long)r * p.head_rs + (long)b * 2 * A;
  const float* e = p.eps + (long)r * p.eps_rs + (long)b * A;
  const float* x = p.xa + (long)r * p.xa_rs + (long)b * p.XP + p.od;
  float* dh = p.dhead + (long)r * p.dhead_rs + (long)b * 2 * A;
  for (int a = 0; a < A; ++a) {
    float da = 0.f;
    for (int c = 0; c < p.K; ++c)
      da += (p.dqa ? p.dqa[(long)r * p.dqa_rs + (long)c * p.dqa_cs + b] : 1.0f) * p.dxa[(long)r * p.dxa_rs + (long)c * p.dxa_cs + (long)b * p.dxa_pitch + a];
    const float lsr = h[A + a];
    const float sg = expf(fminf(fmaxf(lsr, -5.0f), 2.0f));
    const float act = x[a];
    const float om = 1.0f - act * act;
    const float t = 2.0f * act * om / (om + 1e-6f);
    const float du = da * om + dlogp * t;
    const float dls = du * sg * e[a] - dlogp;
    const float dl2 = (lsr >= -5.0f && lsr <= 2.0f) ? dls : 0.f;
    dh[a] = du;
    dh[A + a] = dl2;
    amax = fmaxf(amax, fmaxf(fabsf(du), fabsf(dl2)));
  }
  }
  grad_scale_publish(amax, sh, p.gs_out, r);
}

// ------------------------------------------------------------------------------------------------
// CQL critic loss + gradient seeds (cql.py:108-190, oracle/cql.py), two launches:
//   k_cql_loss_rows: grid (nblk, 2 critics, R): logsumexp rows -> dq + partial sums; block 0 also does the B data
//                    rows (TD target, dq, sums).  cons_scale (= pre-step cql_alpha) is known up front.
//   k_cql_loss_fin : one block per run: reduces the partials in fixed order -> metrics, Lagrange Adam step.
//   rows of q[c]: [0,B) data, [B,B+BN) pi, [B+BN,B+2BN) next-pi, [B+2BN,B+3BN) random
// ------------------------------------------------------------------------------------------------
struct CqlLossP {
  const float* q; long q_rs, q_cs;         // [R][2][Mc]
  float* dq;                               // [R][2][Mc]
  const float* qt; long qt_rs, qt_cs;      // target critics [R][2][Bt]  (Bt = B or B*N with max_q_backup)
  const float* rew; const float* term; long bt_rs;  // [R][B]
  const float* logp_next; long lpn_rs;     // [R][B]   (stochastic backup)
  const float* logp_pi; const float* logp_npi; long lpp_rs;   // [R][BN]
  float* target_q; long tq_rs;             // [R][B] (tap)
  float* part; int nblk;                   // partial sums [R][2][nblk][4] = (s_td, s_q, s_lse, max |dq|)
  unsigned int* ticket;                    // [R] arrival counters (zero between launches: the last arriver resets its own)
  float* gs_out;                           // split precision: dynamic scale of dq [R], or null
  unsigned long long* gstep_next;          // the PRE cell of the step counter (PrepP::gstep_publish): run 0's finishing lane writes gstep + 1; null: k_tick does it
  int B, N, A;
  int Bc, Br;                              // COMBO: conservative rows repeat Bc batch rows; the -w mean Q term runs over the first Br rows
  float gamma, w, T, thr;
  int max_q_backup, det_backup, with_lagrange, auto_alpha; float fixed_alpha;
  RunScalars* sc; const Hyper* hy; float b1, b2, eps;
  const unsigned long long* gstep;
  float* metrics_last; float* metrics_sum; int nm;
  int m_c1, m_c2, m_cqla_loss, m_cqla;
};
__global__ void k_cql_loss_rows(CqlLossP p) {
  __shared__ float sh[4];
  const int blk = blockIdx.x, c = blockIdx.y, r = blockIdx.z;
  const RunScalars& sc = p.sc[r];
  const int B = p.B, BN = p.Bc * p.N;
  const float alpha = p.auto_alpha ? sc.alpha : p.fixed_alpha;
  float cs = 1.0f;
  if (p.with_lagrange) cs = fminf(fmaxf(expf(sc.cql_log_alpha), 0.f), 1e6f);
  const float log_rand = logf(powf(0.5f, (float)p.A));
  const float* q = p.q + (long)r * p.q_rs + (long)c * p.q_cs;
  float* dq = p.dq + (long)r * p.q_rs + (long)c * p.q_cs;
  float s_td = 0.f, s_q = 0.f, s_lse = 0.f, amax = 0.f;
  if (blk == 0) {
    const float* t0 = p.qt + (long)r * p.qt_rs;
    const float* t1 = t0 + p.qt_cs;
    for (int b = threadIdx.x; b < B; b += 256) {
      float nq;
      if (p.max_q_backup) {
        float m0 = -INFINITY, m1 = -INFINITY;
        for (int n = 0; n < p.N; ++n) { m0 = fmaxf(m0, t0[b * p.N + n]); m1 = fmaxf(m1, t1[b * p.N + n]); }
        nq = fminf(m0, m1);
      } else {
        nq = fminf(t0[b], t1[b]);
        if (!p.det_backup) nq -= alpha * p.logp_next[(long)r * p.lpn_rs + b];
      }
      const float y = p.rew[(long)r * p.bt_rs + b] + p.gamma * (1.0f - p.term[(long)r * p.bt_rs + b]) * nq;
      if (c == 0) p.target_q[(long)r * p.tq_rs + b] = y;
      const float d = q[b] - y;
      s_td += d * d;
      if (b < p.Br) s_q += q[b];
      const float g = 2.0f * d / (float)B - (b < p.Br ? cs * p.w / (float)p.Br : 0.f);
      dq[b] = g;
      amax = fmaxf(amax, fabsf(g));
    }
  }
  const float* lpp = p.logp_pi + (long)r * p.lpp_rs;
  const float* lpn = p.logp_npi + (long)r * p.lpp_rs;
  const float gs = cs * p.w / (float)BN;
  const int per = (BN + p.nblk - 1) / p.nblk;
  const int j1 = min(BN, (blk + 1) * per);
  for (int j = blk * per + threadIdx.x; j < j1; j += 256) {
    const float v0 = (q[B + j] - lpp[j]) / p.T, v1 = (q[B + BN + j] - lpn[j]) / p.T, v2 = (q[B + 2 * BN + j] - log_rand) / p.T;
    const float mx = fmaxf(v0, fmaxf(v1, v2));
    const float e0 = expf(v0 - mx), e1 = expf(v1 - mx), e2 = expf(v2 - mx);
    const float se = e0 + e1 + e2;
    s_lse += logf(se) + mx;
    dq[B + j] = gs * (e0 / se); dq[B + BN + j] = gs * (e1 / se); dq[B + 2 * BN + j] = gs * (e2 / se);
    amax = fmaxf(amax, gs * fmaxf(e0, fmaxf(e1, e2)) / se);
  }
  s_td = block_sum256(s_td, sh);
  s_q = block_sum256(s_q, sh);
  s_lse = block_sum256(s_lse, sh);
  amax = block_max256(amax, sh);
  if (threadIdx.x != 0) return;
  // One lane per workgroup publishes its partial sums with write-through (sc1) stores, drains them, and takes a ticket with an agent-scope
  // atomic add; the workgroup whose add came last (told by the value the add returned) reads every partial with sc1 loads and finishes the
  // run: metrics, Lagrange step, the dq scale.  (MI355X_MICROARCH.md, "Valid forms": sc1 stores + vmcnt(0) + agent atomic; last arriver by the
  // returned value; sc1 loads.)  This replaces the separate one-thread k_cql_loss_fin launch -- one kernel node less per step.
  {
    float* o = p.part + (((long)r * 2 + c) * p.nblk + blk) * 4;
    __hip_atomic_store(o + 0, s_td, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(o + 1, s_q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(o + 2, s_lse, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(o + 3, amax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const unsigned int total = 2u * (unsigned int)p.nblk;
  const unsigned int t = __hip_atomic_fetch_add(p.ticket + r, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (t != total - 1u) return;
  __hip_atomic_store(p.ticket + r, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch (stream order)
  RunScalars& scw = p.sc[r];
  float e_cla = 0.f;
  cs = 1.0f;
  if (p.with_lagrange) { e_cla = expf(scw.cql_log_alpha); cs = fminf(fmaxf(e_cla, 0.f), 1e6f); }
  float raw[2];
  float amax_all = 0.f;
  for (int cc = 0; cc < 2; ++cc) {
    float s_td = 0.f, s_q = 0.f, s_lse = 0.f;
    const float* o = p.part + ((long)r * 2 + cc) * p.nblk * 4;
    for (int k = 0; k < p.nblk; ++k) {
      s_td += __hip_atomic_load(o + k * 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_q += __hip_atomic_load(o + k * 4 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_lse += __hip_atomic_load(o + k * 4 + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      amax_all = fmaxf(amax_all, __hip_atomic_load(o + k * 4 + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    }
    const int c = cc;
    float cons = (s_lse / (float)BN) * p.w * p.T - (s_q / (float)p.Br) * p.w;
    raw[c] = cons - p.thr;
    if (p.with_lagrange) cons = cs * raw[c];
    const float loss = s_td / (float)B + cons;
    const int slot = c == 0 ? p.m_c1 : p.m_c2;
    p.metrics_last[(long)r * p.nm + slot] = loss; p.metrics_sum[(long)r * p.nm + slot] += loss;
  }
  if (p.gs_out) p.gs_out[r] = orl_pow2_scale(amax_all);
  if (p.gstep_next && r == 0) *p.gstep_next = *p.gstep + 1ull;      // (k_prepare of this step is done, the next one has not started: stream order)
  if (p.with_lagrange) {
    const float l = -(cs * raw[0] + cs * raw[1]) * 0.5f;
    const float gate = (e_cla >= 0.f && e_cla <= 1e6f) ? 1.f : 0.f;
    const float g = -(raw[0] + raw[1]) * 0.5f * e_cla * gate;
    adam_scalar(scw.cql_log_alpha, scw.cla_m, scw.cla_v, g, p.hy->lr[3], p.b1, p.b2, p.eps, *p.gstep + 1ull);
    p.metrics_last[(long)r * p.nm + p.m_cqla_loss] = l; p.metrics_sum[(long)r * p.nm + p.m_cqla_loss] += l;
    p.metrics_last[(long)r * p.nm + p.m_cqla] = cs; p.metrics_sum[(long)r * p.nm + p.m_cqla] += cs;
  }
}

// ------------------------------------------------------------------------------------------------
// fused Adam (torch.optim.Adam defaults) + split-K slab reduce + optional Polyak
// (sac.py:60-64).  grid (ceil(P/256), nz1, nz0): z1 = net, z0 = run.
// ------------------------------------------------------------------------------------------------
struct AdamP {
  float* params; long p_s0, p_s1;     // [z0][z1][P]
  float* m; float* v;                 // same strides as params
  const float* g; long g_s0, g_s1, g_ks;
  int nseg; long seg_end[12]; int seg_nslab[12];   // per-tensor-group split-K slab counts
  float* target; long t_s0, t_s1;     // Polyak target or null
  long P;
  int lr_slot; const Hyper* hy;
  float b1, b2, eps, tau;
  const unsigned long long* gstep; unsigned long long t_div;  // t = gstep / t_div + 1
  unsigned int* health;               // [z0] sticky per-run flags (ORL_HEALTH_*): a non-finite summed gradient sets ORL_HEALTH_NONFINITE_GRAD
};
// b^t for an integer step count by repeated squaring (a handful of double multiplies; the library pow() on one lane of every
// workgroup was a measurable part of the launch)
__device__ inline double ipow_u64(double b, unsigned long long t) {
  double r = 1.0;
  while (t) { if (t & 1ull) r *= b; b *= b; t >>= 1; }
  return r;
}
// One thread updates four consecutive parameters (16-byte loads / stores of p, m, v, the slabs and the target) when they belong to
// one tensor group; the few groups that straddle a boundary (a scalar bias at the end of a net) fall back to single elements.
__device__ inline float adam_slab_sum(const float* g, long ks, int nslab) {
  if (nslab == 1) return g[0];
  float q[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};            // fixed order, eight independent partial sums
  int s = 0;
  for (; s + 8 <= nslab; s += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] += g[(long)(s + j) * ks];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) if (s + j < nslab) q[j] += g[(long)(s + j) * ks];
  return ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
}
__global__ void k_adam(AdamP p) {
  __shared__ float s_step, s_bc2s;
  if (threadIdx.x == 0) {
    const unsigned long long t = *p.gstep / p.t_div + 1ull;
    const double bc1 = 1.0 - ipow_u64((double)p.b1, t), bc2 = 1.0 - ipow_u64((double)p.b2, t);
    s_step = (float)((double)p.hy->lr[p.lr_slot] / bc1);
    s_bc2s = (float)sqrt(bc2);
  }
  __syncthreads();
  const long i0 = 4 * ((long)blockIdx.x * blockDim.x + threadIdx.x);
  if (i0 >= p.P) return;
  const int z1 = blockIdx.y, z0 = blockIdx.z;
  const float* g = p.g + z0 * p.g_s0 + z1 * p.g_s1;
  const long o = z0 * p.p_s0 + z1 * p.p_s1;
  float* tg = p.target ? p.target + z0 * p.t_s0 + z1 * p.t_s1 : nullptr;
  int sg = 0;
  while (sg < p.nseg - 1 && i0 >= p.seg_end[sg]) ++sg;
  const bool vec = (i0 + 4 <= p.P) && (i0 + 3 < p.seg_end[sg] || sg == p.nseg - 1) && ((o & 3) == 0) && ((p.g_ks & 3) == 0) &&
                   (((p.g_s0 | p.g_s1) & 3) == 0) && (!tg || (((p.t_s0 | p.t_s1) & 3) == 0));
  if (vec) {
    const int nslab = p.seg_nslab[sg];
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 gs;
    if (nslab == 1) gs = *(const f4*)&g[i0];
    else {
      // fixed order, eight partial sums; eight slabs requested per round trip (a net with few runs per engine has up to 32 slabs of a
      // many-row weight gradient: 4 at a time were 8 exposed latencies of the launch)
      f4 q[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) q[j] = (f4){0.f, 0.f, 0.f, 0.f};
      int s = 0;
      for (; s + 8 <= nslab; s += 8) {
        f4 t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = *(const f4*)&g[i0 + (long)(s + j) * p.g_ks];
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] += t[j];
      }
      if (s < nslab) {                                   // (uniform: the slab count belongs to the tensor group)
        f4 t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = *(const f4*)&g[i0 + (long)(s + j < nslab ? s + j : nslab - 1) * p.g_ks];      // clamped, not guarded
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] += (s + j < nslab ? 1.f : 0.f) * t[j];
      }
      gs = ((q[0] + q[1]) + (q[2] + q[3])) + ((q[4] + q[5]) + (q[6] + q[7]));
    }
    // a diverging run shows here even when the forward pass scrubbed its NaNs (the integer-view ReLU maps sign-bit NaNs to +0, gemm.h):
    // one compare per gradient in a kernel that waits for HBM
    if (!(fabsf(gs[0]) <= 3.4e38f) || !(fabsf(gs[1]) <= 3.4e38f) || !(fabsf(gs[2]) <= 3.4e38f) || !(fabsf(gs[3]) <= 3.4e38f))
      atomicOr(p.health + z0, (unsigned int)ORL_HEALTH_NONFINITE_GRAD);
    f4 m = *(const f4*)&p.m[o + i0], v = *(const f4*)&p.v[o + i0], w = *(const f4*)&p.params[o + i0];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      m[j] = m[j] + (gs[j] - m[j]) * (1.0f - p.b1);
      v[j] = v[j] * p.b2 + (1.0f - p.b2) * gs[j] * gs[j];
      w[j] -= s_step * (m[j] / (sqrtf(v[j]) / s_bc2s + p.eps));
    }
    *(f4*)&p.m[o + i0] = m; *(f4*)&p.v[o + i0] = v; *(f4*)&p.params[o + i0] = w;
    if (tg) {
      f4 t = *(const f4*)&tg[i0];
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j] = t[j] * (1.0f - p.tau) + w[j] * p.tau;
      *(f4*)&tg[i0] = t;
    }
    return;
  }
  for (long i = i0; i < i0 + 4 && i < p.P; ++i) {
    int nslab = p.seg_nslab[0];
    for (int k = 1; k < p.nseg; ++k) if (i >= p.seg_end[k - 1]) nslab = p.seg_nslab[k];
    const float gs = adam_slab_sum(g + i, p.g_ks, nslab);
    if (!(fabsf(gs) <= 3.4e38f)) atomicOr(p.health + z0, (unsigned int)ORL_HEALTH_NONFINITE_GRAD);
    float m = p.m[o + i], v = p.v[o + i], w = p.params[o + i];
    m = m + (gs - m) * (1.0f - p.b1);
    v = v * p.b2 + (1.0f - p.b2) * gs * gs;
    w -= s_step * (m / (sqrtf(v) / s_bc2s + p.eps));
    p.m[o + i] = m; p.v[o + i] = v; p.params[o + i] = w;
    if (tg) tg[i] = tg[i] * (1.0f - p.tau) + w * p.tau;
  }
}

// Range scan (orl_health_check; run by orl_step / orl_learn_n themselves when a split-precision run turns non-finite): one workgroup per
// run walks a matrix that enters the MFMAs as fp16 hi + lo planes -- an input, a stored hidden activation (x 1) or the parameters
// (x 2^6) -- and raises ORL_HEALTH_SPLIT_RANGE when an entry reaches `limit` (65504 over the operand scale): beyond it the hi plane is +-inf,
// the lo plane -+inf, and every product that touches the element is NaN.  Not on the step's path.
__global__ void k_range_scan(const float* x, long rs, long cs, int nets, int rows, int cols, int pitch, float limit, unsigned int* health) {
  const int r = blockIdx.x;
  const long per_net = (long)rows * cols, n = per_net * nets;
  bool hit = false;
  for (long e = threadIdx.x; e < n; e += blockDim.x) {
    const long z = e / per_net, w = e - z * per_net;
    const long row = w / cols, c = w - row * cols;
    hit |= fabsf(x[(long)r * rs + z * cs + row * pitch + c]) >= limit;      // (NaN compares false: a diverged run is reported as non-finite, not as out of range)
  }
  if (__syncthreads_or(hit) && threadIdx.x == 0) atomicOr(health + r, (unsigned int)ORL_HEALTH_SPLIT_RANGE);
}
// max |x| of a dataset array (orl_engine_attach_buffer at precision 1: observations / actions beyond the split range are refused up front)
__global__ void k_absmax(const float* x, long n, unsigned int* out_bits) {
  float m = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float a = fabsf(x[i]);
    m = (a > m || a != a) ? a : m;                                            // a NaN sticks
  }
  atomicMax(out_bits, __float_as_uint(m));                                    // non-negative floats (and NaNs above them) order like their bit patterns
}

// q[m] += part[0][m] + part[1][m] + ... (column-tile partial sums of a tail fused into the last hidden layer's epilogue)
struct TailAddP { float* out; long o_s0, o_s1, o_sm; const float* part; long p_s0, p_s1, p_ts; int nparts, M, nz1; };
__global__ void k_tail_add(TailAddP p) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= p.M) return;
  const int z0 = blockIdx.y / p.nz1, z1 = blockIdx.y - z0 * p.nz1;
  float* o = p.out + z0 * p.o_s0 + z1 * p.o_s1 + (long)m * p.o_sm;
  const float* q = p.part + z0 * p.p_s0 + z1 * p.p_s1 + m;
  float a = *o;
  for (int t = 0; t < p.nparts; ++t) a += q[(long)t * p.p_ts];
  *o = a;
}

// Polyak only (targets whose nets were updated earlier in the step)
__global__ void k_polyak(float* target, long t_s0, long t_s1, const float* src, long s_s0, long s_s1, long P, float tau) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  float* t = target + blockIdx.z * t_s0 + blockIdx.y * t_s1 + i;
  const float w = src[blockIdx.z * s_s0 + blockIdx.y * s_s1 + i];
  *t = *t * (1.0f - tau) + w * tau;
}


// ================================================================================================
// IQL (iql.py:86-139; oracle/iql.py)
// ================================================================================================
struct MetricsP { float* last; float* sum; int nm; };
__device__ inline void metric_set(const MetricsP& m, int r, int slot, float v) {
  m.last[(long)r * m.nm + slot] = v; m.sum[(long)r * m.nm + slot] += v;
}

// value loss: diff = min(q1_old,q2_old) - v ; w = diff>0 ? tau_e : 1-tau_e ; L = mean(w diff^2) ; dv = -2 w diff / B
struct IqlVP {
  const float* qo; long qo_rs, qo_cs;   // [R][2][B] target critics at (s,a)
  const float* v; long v_rs;            // [R][B]
  float* dv;                            // [R][B]
  float* qmin; long qmin_rs;            // [R][B] keeps min(q_old) for the actor weights
  int B; float expectile; MetricsP m; int slot;
  float* gs_out;                        // split precision: dynamic scale of dv [R], or null
};
__global__ void k_iql_v_loss(IqlVP p) {
  __shared__ float sh[4];
  const int r = blockIdx.x;
  float s = 0.f, amax = 0.f;
  for (int b = threadIdx.x; b < p.B; b += 256) {
    const float* q = p.qo + (long)r * p.qo_rs + b;
    const float qm = fminf(q[0], q[p.qo_cs]);
    const float d = qm - p.v[(long)r * p.v_rs + b];
    const float w = d > 0.f ? p.expectile : 1.0f - p.expectile;
    s += w * d * d;
    const float g = -2.0f * w * d / (float)p.B;
    p.dv[(long)r * p.v_rs + b] = g;
    amax = fmaxf(amax, fabsf(g));
    p.qmin[(long)r * p.qmin_rs + b] = qm;
  }
  s = block_sum256(s, sh);
  if (threadIdx.x == 0) metric_set(p.m, r, p.slot, s / (float)p.B);
  grad_scale_publish(amax, sh, p.gs_out, r);
}

// Q losses + advantage weights: y = r + gamma (1-d) V_new(s') ; dq_i = 2 (q_i - y)/B ;
// exp_a = min(exp((min q_old - V_new(s)) * beta), 100)
struct IqlQP {
  const float* q; long q_rs, q_cs;      // [R][2][B]
  float* dq;
  const float* v2; long v2_rs;          // [R][2B]: rows [0,B) V_new(s), [B,2B) V_new(s')
  const float* qmin; long qmin_rs;
  const float* rew; const float* term; long bt_rs;
  float* exp_a; long ea_rs;             // [R][B]
  float* target_q; long tq_rs;
  int B; float gamma, beta; MetricsP m; int slot_q1, slot_q2;
  float* gs_out;                        // split precision: dynamic scale of dq [R], or null
};
__global__ void k_iql_q_loss(IqlQP p) {
  __shared__ float sh[4];
  const int r = blockIdx.x;
  float s1 = 0.f, s2 = 0.f, amax = 0.f;
  for (int b = threadIdx.x; b < p.B; b += 256) {
    const float* v2 = p.v2 + (long)r * p.v2_rs;
    const float y = p.rew[(long)r * p.bt_rs + b] + p.gamma * (1.0f - p.term[(long)r * p.bt_rs + b]) * v2[p.B + b];
    p.target_q[(long)r * p.tq_rs + b] = y;
    const float* q = p.q + (long)r * p.q_rs + b;
    float* dq = p.dq + (long)r * p.q_rs + b;
    const float d1 = q[0] - y, d2 = q[p.q_cs] - y;
    s1 += d1 * d1; s2 += d2 * d2;
    dq[0] = 2.0f * d1 / (float)p.B; dq[p.q_cs] = 2.0f * d2 / (float)p.B;
    amax = fmaxf(amax, 2.0f * fmaxf(fabsf(d1), fabsf(d2)) / (float)p.B);
    p.exp_a[(long)r * p.ea_rs + b] = fminf(expf((p.qmin[(long)r * p.qmin_rs + b] - v2[b]) * p.beta), 100.0f);
  }
  s1 = block_sum256(s1, sh);
  s2 = block_sum256(s2, sh);
  if (threadIdx.x == 0) { metric_set(p.m, r, p.slot_q1, s1 / (float)p.B); metric_set(p.m, r, p.slot_q2, s2 / (float)p.B); }
  grad_scale_publish(amax, sh, p.gs_out, r);
}

// actor: mu = tanh(m_raw), sigma = exp(sigma_param); L = -mean(exp_a * logp(a_data)); writes d m_raw and the
// sigma_param gradient (one slab) directly.
struct IqlAP {
  const float* mraw; long mraw_rs;      // [R][B][A]
  const float* act; long act_rs; int apitch;
  const float* exp_a; long ea_rs;
  const float* sigma_param; long sp_rs; // [R][A] (inside the actor parameter block)
  float* dmraw;                         // [R][B][A]
  float* g_sigma; long gs_rs;           // [R][A] gradient slab 0 of sigma_param
  int B, A; MetricsP m; int slot;
  float* gs_out;                        // split precision: dynamic scale of dmraw [R], or null
};
__global__ void k_iql_actor_loss(IqlAP p) {
  __shared__ float sh[4];
  __shared__ float gsig[64];
  const int r = blockIdx.x;
  const int A = p.A;
  for (int a = threadIdx.x; a < A; a += 256) gsig[a] = 0.f;
  __syncthreads();
  float s = 0.f, amax = 0.f;
  const float* sp = p.sigma_param + (long)r * p.sp_rs;
  for (int b = threadIdx.x; b < p.B; b += 256) {
    const float ea = p.exp_a[(long)r * p.ea_rs + b];
    const float dlogp = -ea / (float)p.B;
    float lp = 0.f;
    for (int a = 0; a < A; ++a) {
      const float mu = tanhf(p.mraw[(long)r * p.mraw_rs + (long)b * A + a]);
      const float ls = sp[a], sg = expf(ls), var = sg * sg;
      const float d = p.act[(long)r * p.act_rs + (long)b * p.apitch + a] - mu;
      lp += -(d * d) / (2.0f * var) - ls - ORL_LOG_SQRT_2PI;
      const float g = dlogp * d / var * (1.0f - mu * mu);
      p.dmraw[(long)r * p.mraw_rs + (long)b * A + a] = g;
      amax = fmaxf(amax, fabsf(g));
    }
    s += ea * lp;
  }
  s = block_sum256(s, sh);
  // d(loss)/d(sigma_param[a]): one fixed-order block reduction per action dimension (shared-memory float atomics summed in arrival order:
  // two runs fed identical inputs drifted apart in the last bit after ~150 steps)
  for (int a = 0; a < A; ++a) {
    const float ls = sp[a], sg = expf(ls), var = sg * sg;
    float part = 0.f;
    for (int b = threadIdx.x; b < p.B; b += 256) {
      const float dlogp = -p.exp_a[(long)r * p.ea_rs + b] / (float)p.B;
      const float mu = tanhf(p.mraw[(long)r * p.mraw_rs + (long)b * A + a]);
      const float d = p.act[(long)r * p.act_rs + (long)b * p.apitch + a] - mu;
      part += dlogp * (d * d / var - 1.0f);
    }
    part = block_sum256(part, sh);
    if (threadIdx.x == 0) gsig[a] = part;
  }
  __syncthreads();
  for (int a = threadIdx.x; a < A; a += 256) p.g_sigma[(long)r * p.gs_rs + a] = gsig[a];
  if (threadIdx.x == 0) metric_set(p.m, r, p.slot, -s / (float)p.B);
  grad_scale_publish(amax, sh, p.gs_out, r);
}

// ================================================================================================
// TD3+BC (td3bc.py:83-124; oracle/td3bc.py)
// ================================================================================================
// deterministic actor output: a = max_action * tanh(m_raw) (+ clipped target-policy noise), written into a
// critic-input matrix.  grid (ceil(B*A/256), R)
struct DetActP {
  const float* mraw; long mraw_rs;       // [R][B][A]
  const float* eps; long eps_rs;         // [R][B][A] or null
  float* X; long x_rs; int XP, od;       // actions -> X[b*XP + od + a]
  int B, A; float max_action, policy_noise, noise_clip;
};
__global__ void k_det_action(DetActP p) {
  const int r = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= p.B * p.A) return;
  const int b = t / p.A, a = t - b * p.A;
  float v = p.max_action * tanhf(p.mraw[(long)r * p.mraw_rs + t]);
  if (p.eps) {
    const float nz = fminf(fmaxf(p.eps[(long)r * p.eps_rs + t] * p.policy_noise, -p.noise_clip), p.noise_clip);
    v = fminf(fmaxf(v + nz, -p.max_action), p.max_action);
  }
  p.X[(long)r * p.x_rs + (long)b * p.XP + p.od + a] = v;
}

// twin-critic TD loss (also IQL-free algorithms): y = r + gamma (1-d) (min_c qt_c - alpha*logp_next) ; dq_c = 2 (q_c - y)/B
struct TdLossP {
  const float* q; long q_rs, q_cs; float* dq;    // [R][K][B]
  const float* qt; long qt_rs, qt_cs; int Kt;    // target critics [R][Kt][Bt]
  const float* rew; const float* term; long bt_rs;
  const float* logp_next; long lpn_rs;           // or null
  float* target_q; long tq_rs;
  int B, K, rep;                                 // rep > 1: max over `rep` repeated next actions before the min
  float gamma; int sum_over_k;                   // EDAC: one metric = sum_k mean_b ; else metric slot per critic
  const RunScalars* sc; int use_alpha; int auto_alpha; float fixed_alpha;
  MetricsP m; int slot0; float last_actor_loss_slot_unused;
  float* gs_out;                                 // split precision: dynamic scale of dq [R], or null
};
__global__ void k_td_loss(TdLossP p) {
  __shared__ float sh[4];
  const int r = blockIdx.x;
  const int B = p.B;
  const float alpha = p.auto_alpha ? p.sc[r].alpha : p.fixed_alpha;
  float* tq = p.target_q + (long)r * p.tq_rs;
  for (int b = threadIdx.x; b < B; b += 256) {
    float nq = INFINITY;
    for (int c = 0; c < p.Kt; ++c) {
      const float* t = p.qt + (long)r * p.qt_rs + (long)c * p.qt_cs;
      float v = -INFINITY;
      for (int n = 0; n < p.rep; ++n) v = fmaxf(v, t[b * p.rep + n]);
      nq = fminf(nq, v);
    }
    if (p.use_alpha && p.logp_next) nq -= alpha * p.logp_next[(long)r * p.lpn_rs + b];
    tq[b] = p.rew[(long)r * p.bt_rs + b] + p.gamma * (1.0f - p.term[(long)r * p.bt_rs + b]) * nq;
  }
  __syncthreads();
  float total = 0.f, amax = 0.f;
  for (int c = 0; c < p.K; ++c) {
    const float* q = p.q + (long)r * p.q_rs + (long)c * p.q_cs;
    float* dq = p.dq + (long)r * p.q_rs + (long)c * p.q_cs;
    float s = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
      const float d = q[b] - tq[b];
      s += d * d;
      dq[b] = 2.0f * d / (float)B;
      amax = fmaxf(amax, 2.0f * fabsf(d) / (float)B);
    }
    s = block_sum256(s, sh);
    if (p.sum_over_k) total += s / (float)B;
    else if (threadIdx.x == 0) metric_set(p.m, r, p.slot0 + c, s / (float)B);
  }
  if (p.sum_over_k && threadIdx.x == 0) metric_set(p.m, r, p.slot0, total);
  grad_scale_publish(amax, sh, p.gs_out, r);
}

// TD3BC actor objective pieces: lambda = alpha / mean|q| ; L = -lambda mean(q) + mean((a_pi - a)^2) ; dq = -lambda/B
struct Td3ActorP {
  const float* q; long q_rs;             // [R][B] critic1(s, pi(s))
  float* dq;
  const float* xa; long xa_rs; int XP, od;  // pi(s) in xa[:, od:]
  const float* act; long act_rs; int apitch;
  int B, A; float alpha; RunScalars* sc; MetricsP m; int slot;
  float* gs_out;                         // split precision: dynamic scale of dq (= lambda / B on every row) [R], or null
};
__global__ void k_td3_actor_loss(Td3ActorP p) {
  __shared__ float sh[4];
  const int r = blockIdx.x;
  float sq = 0.f, sabs = 0.f, sbc = 0.f;
  for (int b = threadIdx.x; b < p.B; b += 256) {
    const float q = p.q[(long)r * p.q_rs + b];
    sq += q; sabs += fabsf(q);
    for (int a = 0; a < p.A; ++a) {
      const float d = p.xa[(long)r * p.xa_rs + (long)b * p.XP + p.od + a] - p.act[(long)r * p.act_rs + (long)b * p.apitch + a];
      sbc += d * d;
    }
  }
  sq = block_sum256(sq, sh); sabs = block_sum256(sabs, sh); sbc = block_sum256(sbc, sh);
  const float lmbda = p.alpha / (sabs / (float)p.B);
  for (int b = threadIdx.x; b < p.B; b += 256) p.dq[(long)r * p.q_rs + b] = -lmbda / (float)p.B;
  if (threadIdx.x == 0) {
    const float loss = -lmbda * (sq / (float)p.B) + sbc / (float)(p.B * p.A);
    p.sc[r].last_actor_loss = loss;
    metric_set(p.m, r, p.slot, loss);
    if (p.gs_out) p.gs_out[r] = orl_pow2_scale(fabsf(lmbda) / (float)p.B);
  }
}
// critic-only steps report the last actor loss (td3bc.py:120)
__global__ void k_td3_report_last(const RunScalars* sc, MetricsP m, int slot) {
  const int r = blockIdx.x;
  if (threadIdx.x == 0) metric_set(m, r, slot, sc[r].last_actor_loss);
}
// d m_raw = (dQ/da + 2 (a_pi - a)/(B A)) * max_action * (1 - tanh^2)
struct Td3ActorBwdP {
  const float* dxa; long dxa_rs; int dxa_pitch;   // [R][B][A]
  const float* xa; long xa_rs; int XP, od;
  const float* act; long act_rs; int apitch;
  float* dmraw; long dm_rs;
  int B, A; float max_action;
  float* gs_out;                                   // split precision: dynamic scale of the actor's backward pass [R], or null
};
// one workgroup per run (B * A elements: a few per thread), so that the seed kernel can publish the backward pass's gradient scale itself
__global__ void k_td3_actor_bwd(Td3ActorBwdP p) {
  __shared__ float sh[4];
  const int r = blockIdx.x;
  float amax = 0.f;
  for (int t = threadIdx.x; t < p.B * p.A; t += 256) {
    const int b = t / p.A, a = t - b * p.A;
    const float ap = p.xa[(long)r * p.xa_rs + (long)b * p.XP + p.od + a];
    const float da = p.dxa[(long)r * p.dxa_rs + (long)b * p.dxa_pitch + a] + 2.0f * (ap - p.act[(long)r * p.act_rs + (long)b * p.apitch + a]) / (float)(p.B * p.A);
    const float th = ap / p.max_action;
    const float g = da * p.max_action * (1.0f - th * th);
    p.dmraw[(long)r * p.dm_rs + t] = g;
    amax = fmaxf(amax, fabsf(g));
  }
  grad_scale_publish(amax, sh, p.gs_out, r);
}

// ================================================================================================
// MCQ (mcq.py:48-126; oracle/mcq.py): VAE behaviour policy (nets/vae.py) + in-distribution / OOD critic targets
// ================================================================================================
// encoder head -> latent: ehead = [mean | log_std_raw] (B x 2Z); ls = clamp(raw, -4, 15); std = exp(ls); z = mean + std * eps
// writes z into the decoder input rows xd[:, od:od+Z] and keeps std.  grid (ceil(B*Z/256), R)
struct VaeLatentP {
  const float* ehead; long eh_rs;      // [R][B][2Z]
  const float* eps; long eps_rs;       // [R][B][Z]
  float* xd; long xd_rs; int xd_pitch, od;
  float* stdv; long std_rs;            // [R][B][Z]
  int B, Z;
};
__global__ void k_vae_latent(VaeLatentP p) {
  const int r = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= p.B * p.Z) return;
  const int b = t / p.Z, j = t - b * p.Z;
  const float* h = p.ehead + (long)r * p.eh_rs + (long)b * 2 * p.Z;
  const float sd = expf(fminf(fmaxf(h[p.Z + j], -4.0f), 15.0f));
  p.stdv[(long)r * p.std_rs + t] = sd;
  p.xd[(long)r * p.xd_rs + (long)b * p.xd_pitch + p.od + j] = h[j] + sd * p.eps[(long)r * p.eps_rs + t];
}
// loss = mse(max_a tanh(m3), a) + KL (vae.py:40-53, mcq.py:53-56): metric + d m3.  grid (R), block 256
struct VaeLossP {
  const float* m3; long m3_rs;         // [R][B][A] decoder output before tanh
  const float* act; long act_rs; int apitch;
  const float* ehead; long eh_rs;      // mean | log_std_raw
  const float* stdv; long std_rs;
  float* dm3;                          // [R][B][A]
  int B, A, Z; float max_action; MetricsP m; int slot;
};
__global__ void k_vae_loss(VaeLossP p) {
  __shared__ float sh[4];
  const int r = blockIdx.x;
  float s_rec = 0.f, s_kl = 0.f;
  for (int t = threadIdx.x; t < p.B * p.A; t += 256) {
    const int b = t / p.A, a = t - b * p.A;
    const float th = tanhf(p.m3[(long)r * p.m3_rs + t]);
    const float d = p.max_action * th - p.act[(long)r * p.act_rs + (long)b * p.apitch + a];
    s_rec += d * d;
    p.dm3[(long)r * p.m3_rs + t] = 2.0f * d / (float)(p.B * p.A) * p.max_action * (1.0f - th * th);
  }
  for (int t = threadIdx.x; t < p.B * p.Z; t += 256) {
    const int b = t / p.Z, j = t - b * p.Z;
    const float mean = p.ehead[(long)r * p.eh_rs + (long)b * 2 * p.Z + j];
    const float sd = p.stdv[(long)r * p.std_rs + t];
    s_kl += 1.0f + logf(sd * sd) - mean * mean - sd * sd;
  }
  s_rec = block_sum256(s_rec, sh);
  s_kl = block_sum256(s_kl, sh);
  if (threadIdx.x == 0) metric_set(p.m, r, p.slot, s_rec / (float)(p.B * p.A) - 0.5f * (s_kl / (float)(p.B * p.Z)));
}
// gradient of the encoder head: d mean = dz + mean / (B Z) ; d ls_raw = (dz std eps + (std^2 - 1) / (B Z)) gated by the clamp
struct VaeHeadBwdP {
  const float* dz; long dz_rs;         // [R][B][Z] gradient w.r.t. the latent columns of the decoder input
  const float* ehead; long eh_rs; const float* stdv; long std_rs; const float* eps; long eps_rs;
  float* dehead;                       // [R][B][2Z]
  int B, Z;
};
__global__ void k_vae_head_bwd(VaeHeadBwdP p) {
  const int r = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= p.B * p.Z) return;
  const int b = t / p.Z, j = t - b * p.Z;
  const float* h = p.ehead + (long)r * p.eh_rs + (long)b * 2 * p.Z;
  const float dz = p.dz[(long)r * p.dz_rs + t], sd = p.stdv[(long)r * p.std_rs + t], lsr = h[p.Z + j];
  const float inv = 1.0f / (float)(p.B * p.Z);
  float* dh = p.dehead + (long)r * p.eh_rs + (long)b * 2 * p.Z;
  dh[j] = dz + h[j] * inv;
  dh[p.Z + j] = (lsr >= -4.0f && lsr <= 15.0f) ? dz * sd * p.eps[(long)r * p.eps_rs + t] + (sd * sd - 1.0f) * inv : 0.f;
}
// latent columns of the OOD decoder input: xdo[:, od:od+Z] = clamp(z, -0.5, 0.5) (vae.py:57-58).  grid (ceil(rows*Z/256), R)
__global__ void k_clamp_latent(const float* z, long z_rs, float* xd, long xd_rs, int xd_pitch, int od, int rows, int Z) {
  const int r = blockIdx.y, t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= rows * Z) return;
  const int b = t / Z, j = t - b * Z;
  xd[(long)r * xd_rs + (long)b * xd_pitch + od + j] = fminf(fmaxf(z[(long)r * z_rs + t], -0.5f), 0.5f);
}
// critic losses over the 3B rows [in-distribution B ; OOD 2B] (mcq.py:62-94):
//   y_in = r + gamma (1 - d) (min_c qt_c - alpha logp') ; y_ood[j] = min_c max_n qto_c[j N + n]
//   L_c = lambda mse(q_c[:B] - y_in) + (1 - lambda) mse(q_c[B:] - y_ood) ; dq accordingly.  grid (R), block 256
struct McqLossP {
  const float* q; long q_rs, q_cs; float* dq;        // [R][2][3B]
  const float* qt; long qt_rs, qt_cs;                // target critics on (s', a') [R][2][B]
  const float* qto; long qto_rs, qto_cs;             // target critics on the 2B*N sampled pairs [R][2][2BN]
  const float* rew; const float* term; long bt_rs; const float* logp_next; long lpn_rs;
  float* target_q; long tq_rs; float* target_ood; long to_rs;     // [R][B], [R][2B]
  int B, N; float gamma, lambda;
  const RunScalars* sc; int auto_alpha; float fixed_alpha; MetricsP m; int slot0;
};
__global__ void k_mcq_loss(McqLossP p) {
  __shared__ float sh[4];
  const int r = blockIdx.x, B = p.B;
  const float alpha = p.auto_alpha ? p.sc[r].alpha : p.fixed_alpha;
  float* yi = p.target_q + (long)r * p.tq_rs;
  float* yo = p.target_ood + (long)r * p.to_rs;
  const float* t0 = p.qt + (long)r * p.qt_rs;
  const float* o0 = p.qto + (long)r * p.qto_rs;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float nq = fminf(t0[b], t0[p.qt_cs + b]) - alpha * p.logp_next[(long)r * p.lpn_rs + b];
    yi[b] = p.rew[(long)r * p.bt_rs + b] + p.gamma * (1.0f - p.term[(long)r * p.bt_rs + b]) * nq;
  }
  for (int j = threadIdx.x; j < 2 * B; j += 256) {
    float m0 = -INFINITY, m1 = -INFINITY;
    for (int n = 0; n < p.N; ++n) { m0 = fmaxf(m0, o0[(long)j * p.N + n]); m1 = fmaxf(m1, o0[p.qto_cs + (long)j * p.N + n]); }
    yo[j] = fminf(m0, m1);
  }
  __syncthreads();
  for (int c = 0; c < 2; ++c) {
    const float* q = p.q + (long)r * p.q_rs + (long)c * p.q_cs;
    float* dq = p.dq + (long)r * p.q_rs + (long)c * p.q_cs;
    float s_in = 0.f, s_ood = 0.f;
    for (int b = threadIdx.x; b < B; b += 256) {
      const float d = q[b] - yi[b];
      s_in += d * d;
      dq[b] = p.lambda * 2.0f * d / (float)B;
    }
    for (int j = threadIdx.x; j < 2 * B; j += 256) {
      const float d = q[B + j] - yo[j];
      s_ood += d * d;
      dq[B + j] = (1.0f - p.lambda) * 2.0f * d / (float)(2 * B);
    }
    s_in = block_sum256(s_in, sh);
    s_ood = block_sum256(s_ood, sh);
    if (threadIdx.x == 0) metric_set(p.m, r, p.slot0 + c, p.lambda * (s_in / (float)B) + (1.0f - p.lambda) * (s_ood / (float)(2 * B)));
  }
}

// ================================================================================================
// EDAC gradient-diversity term (edac.py:136-149; oracle/edac.py): from g[k][b][:] = dQ_k/da
//   L_g = mean_b sum_{i!=j} <g^_i, g^_j> / (K-1) ; gamma = d(eta L_g)/dg.   One thread per batch row.
// grid (ceil(B/64), R), block 64; K*A <= 640 values per row kept in registers/LDS-free loops.
// ================================================================================================
// EDAC's adjoint sweep, first step (edac.py:136-149 restated analytically): t_0[b][j] = 1[h0[b][j] > 0] * sum_a gamma[b][a] * W0[od + a][j].
// A product over the A (6) action inputs: as a GEMM it is all epilogue (217 us tiled); here every thread writes four consecutive j of one
// row -- the launch is its 336 MB of output at the HBM rate.  fp32 FMAs in both precisions (exact-fp32 class arithmetic).
struct EdacT0P {
  const float* gamma; long g_rs, g_cs; int gpitch;      // [R][K][B][A]
  const float* W; long w_rs, w_cs; int wpitch;          // action rows of the ensemble's first layer: W[a * wpitch + j], (in, out)-major
  const float* h0; long h_rs, h_cs; int hpitch;         // activation values (mask source when no bits)
  const unsigned int* bits; long b_rs, b_cs; int bg;    // packed mask words of h0, or null
  float* out; long o_rs, o_cs; int opitch;
  int B, A, N, K;
};
// one wave = all N columns (four per lane) of EDAC_T0_RB consecutive rows: the A weight rows stay in registers across the rows (loading them
// per output element made the launch L1-bound: 13 loads per 16 bytes written, 1.6 TB/s)
enum { EDAC_T0_RB = 8, EDAC_T0_AMAX = 8 };
__global__ void k_edac_t0(EdacT0P p) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int z1 = blockIdx.y, z0 = blockIdx.z;
  typedef float f4 __attribute__((ext_vector_type(4)));
  const float* w = p.W + z0 * p.w_rs + z1 * p.w_cs;
  const float* gbase = p.gamma + z0 * p.g_rs + z1 * p.g_cs;
  float* out = p.out + z0 * p.o_rs + z1 * p.o_cs;
  // The 8 x 8 gamma values of the wave's row block are ONE vector load (lane l holds row l / 8, action l % 8; clamped index, factor 0
  // beyond A), read back below with v_readlane -- as wave-uniform scalars each of them was its own load -> wait -> readfirstlane round
  // trip (80 per thread in the first version).  Loaded OUTSIDE the column loop: every lane must hold its value, also the lanes whose columns
  // lie beyond N.
  static_assert(EDAC_T0_RB * EDAC_T0_AMAX == 64, "one lane per (row, action) of the block");
  const int b0 = EDAC_T0_RB * (blockIdx.x * (blockDim.x >> 6) + wave);
  const int gi = lane >> 3, ga = lane & 7, gb = b0 + gi < p.B ? b0 + gi : p.B - 1;
  float gl = (ga < p.A ? 1.f : 0.f) * gbase[(long)gb * p.gpitch + (ga < p.A ? ga : p.A - 1)];
  asm volatile("" : "+v"(gl));      // materialise it HERE, in every lane: the optimiser otherwise sinks the load into the column loop, where the
                                    // lanes beyond N are masked off -- and v_readlane below would read their stale registers
  for (int j = 4 * lane; j < p.N; j += 256) {
    f4 wr[EDAC_T0_AMAX];
#pragma unroll
    for (int a = 0; a < EDAC_T0_AMAX; ++a) wr[a] = *(const f4*)&w[(long)(a < p.A ? a : p.A - 1) * p.wpitch + j];      // (rows >= A: their gamma factor is 0)
    // every load of the row block is issued before the first use
    unsigned int mw[EDAC_T0_RB];
    f4 hv[EDAC_T0_RB];
    const bool use_bits = p.bits != nullptr;                       // uniform
#pragma unroll
    for (int i = 0; i < EDAC_T0_RB; ++i) {
      const int b = b0 + i < p.B ? b0 + i : p.B - 1;
      mw[i] = use_bits ? (p.bits + z0 * p.b_rs + z1 * p.b_cs)[(long)b * p.bg + (j >> 5)] >> (j & 31) : 0u;
    }
    if (!use_bits) {
#pragma unroll
      for (int i = 0; i < EDAC_T0_RB; ++i) {
        const int b = b0 + i < p.B ? b0 + i : p.B - 1;
        hv[i] = *(const f4*)&(p.h0 + z0 * p.h_rs + z1 * p.h_cs)[(long)b * p.hpitch + j];
        mw[i] = (hv[i][0] > 0.f ? 1u : 0u) | (hv[i][1] > 0.f ? 2u : 0u) | (hv[i][2] > 0.f ? 4u : 0u) | (hv[i][3] > 0.f ? 8u : 0u);
      }
    }
#pragma unroll
    for (int i = 0; i < EDAC_T0_RB; ++i) {
      const int b = b0 + i;
      f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int a = 0; a < EDAC_T0_AMAX; ++a)
        acc += __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, gl), 8 * i + a)) * wr[a];
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = ((mw[i] >> r) & 1u) ? acc[r] : 0.f;
      if (b < p.B) *(f4*)&out[(long)b * p.opitch + j] = acc;
    }
  }
}

struct EdacGP {
  const float* g; long g_rs, g_cs; int gpitch;   // [R][K][B][A]
  float* gamma;                                   // same layout
  int B, K, A; float eta; MetricsP m; int slot;
  float* gs_out;                                  // split precision: dynamic scale of gamma [R], or null
};
__global__ void k_edac_gamma(EdacGP p) {
  __shared__ float sh[4];
  const int r = blockIdx.x;
  float sl = 0.f, amax = 0.f;
  for (int b = threadIdx.x; b < p.B; b += 256) {
    float S[32];
    for (int a = 0; a < p.A; ++a) S[a] = 0.f;
    float sum_sq = 0.f;
    for (int k = 0; k < p.K; ++k) {
      const float* gk = p.g + (long)r * p.g_rs + (long)k * p.g_cs + (long)b * p.gpitch;
      float n2 = 0.f;
      for (int a = 0; a < p.A; ++a) n2 += gk[a] * gk[a];
      const float nk = sqrtf(n2) + 1e-10f;
      for (int a = 0; a < p.A; ++a) S[a] += gk[a] / nk;
      sum_sq += n2 / (nk * nk);
    }
    float ss = 0.f;
    for (int a = 0; a < p.A; ++a) ss += S[a] * S[a];
    sl += ss - sum_sq;
    const float cf = p.eta * 2.0f / (float)((p.K - 1) * p.B);
    for (int k = 0; k < p.K; ++k) {
      const float* gk = p.g + (long)r * p.g_rs + (long)k * p.g_cs + (long)b * p.gpitch;
      float* ok = p.gamma + (long)r * p.g_rs + (long)k * p.g_cs + (long)b * p.gpitch;
      float n2 = 0.f;
      for (int a = 0; a < p.A; ++a) n2 += gk[a] * gk[a];
      const float nrm = sqrtf(n2), nk = nrm + 1e-10f;
      float gc = 0.f;
      for (int a = 0; a < p.A; ++a) gc += gk[a] * cf * (S[a] - gk[a] / nk);
      const float safe = nrm > 0.f ? nrm : 1.0f;
      for (int a = 0; a < p.A; ++a) {
        const float v = cf * (S[a] - gk[a] / nk) / nk - gk[a] * (gc / (nk * nk * safe));
        ok[a] = v;
        amax = fmaxf(amax, fabsf(v));
      }
    }
  }
  sl = block_sum256(sl, sh);
  grad_scale_publish(amax, sh, p.gs_out, r);
  if (threadIdx.x == 0) {
    const float lg = p.eta * (sl / (float)p.B) / (float)(p.K - 1);
    p.m.last[(long)r * p.m.nm + p.slot] += lg; p.m.sum[(long)r * p.m.nm + p.slot] += lg;
  }
}

}  // namespace orl
