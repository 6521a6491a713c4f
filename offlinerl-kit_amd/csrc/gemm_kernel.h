// gemm_kernel.h — device code of the run-batched MFMA GEMM (see gemm.h for the interface) and the launch dispatch.
// Included only by the gemm_inst_*.hip translation units.
#pragma once
#include "gemm.h"
#include <cstdlib>

namespace orl {

// ------------------------------------------------------------------------------------------------
// operand tile loader: ROWS x TK tile -> registers -> LDS [row][k]
// ------------------------------------------------------------------------------------------------
template <int ROWS, int TK, int NT, int PITCH, int LMODE, bool IS_A, int PRO>
struct TileLoader {
  static constexpr int ELEMS = ROWS * TK;
  static constexpr int SLOT_ELEMS = (LMODE == L_SCALAR) ? 1 : ((LMODE == L_VECK || LMODE == L_VECKU) ? 4 : 16);
  static constexpr int NSLOTS = ELEMS / SLOT_ELEMS;
  static constexpr int PER_THREAD = (NSLOTS + NT - 1) / NT;
  static constexpr bool EXACT = (NSLOTS % NT) == 0;
  float reg[PER_THREAD * SLOT_ELEMS];
  // loop-invariant per-slot state, computed once by init(): the K loop only adds a uniform k offset
  int goff[PER_THREAD];     // element offset of the slot's first element at k0 = 0 (row already clamped)
  int loff[PER_THREAD];     // LDS offset of the slot's first element
  int kin[PER_THREAD];      // k of the slot inside a chunk
  int grow[PER_THREAD];     // clamped global row (rank-1 prologue) 
  int ones[PER_THREAD];     // PB_ONES: bit rr set -> row (r + rr) is the virtual ones row
  long sk_;
  float tacc[(LMODE == L_BLK4 && IS_A && PRO == PA_RANK1) ? PER_THREAD * 4 : 1];   // fused tail-weight gradient partials
  float bacc;

  __device__ static inline void slot_rk(int e, bool k_contig, int& r, int& k) {
    if (LMODE == L_VECK || LMODE == L_VECKU) { constexpr int Q = TK / 4; k = 4 * (e % Q); r = e / Q; }
    else if (LMODE == L_BLK4) { constexpr int Q = ROWS / 4; r = 4 * (e % Q); k = 4 * (e / Q); }
    else if (k_contig) { k = e % TK; r = e / TK; }
    else { r = e % ROWS; k = e / ROWS; }
  }

  __device__ inline void init(const GemmP& p, int row0, int tid) {
    const long sr = IS_A ? p.a_sr : p.b_sr, sk = IS_A ? p.a_sk : p.b_sk;
    sk_ = sk;
    // rows that exist in memory: the wgrad "ones" row (bias-gradient column) is virtual and must never be read
    const int nrows = IS_A ? p.M : ((PRO == PB_ONES && p.ones_row < p.N) ? p.ones_row : p.N);
    const int rlim = IS_A ? p.a_rlim : p.b_rlim;
    const bool k_contig = (sk == 1);
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      const int e = tid + i * NT;
      int r = 0, k = 0;
      slot_rk(EXACT ? e : (e < NSLOTS ? e : 0), k_contig, r, k);
      int gr = row0 + r;
      if (LMODE == L_BLK4) gr = gr <= rlim - 4 ? gr : rlim - 4;
      else gr = gr < nrows ? gr : nrows - 1;
      grow[i] = gr;
      kin[i] = k;
      loff[i] = r * PITCH + k;
      goff[i] = (LMODE == L_BLK4) ? (int)((long)k * sk + gr) : (int)((long)gr * sr + (long)k * sk);
      int om = 0;
      if (PRO == PB_ONES && !IS_A) {
        if (LMODE == L_BLK4) { for (int rr = 0; rr < 4; ++rr) om |= (row0 + r + rr == p.ones_row) ? (1 << rr) : 0; }
        else om = (row0 + r == p.ones_row) ? 1 : 0;
      }
      ones[i] = om;
    }
    bacc = 0.f;
    for (int i = 0; i < (int)(sizeof(tacc) / sizeof(float)); ++i) tacc[i] = 0.f;
  }

  // TAIL = false: the whole chunk [k0, k0+TK) is inside K (no k checks)
  template <bool TAIL>
  __device__ inline void load(const GemmP& p, const float* __restrict__ g, const float* __restrict__ rowv,
                              const float* __restrict__ colv, int k0, int tid) {
    const long sk = sk_;
    const float* __restrict__ gk0 = g + (long)k0 * sk;      // uniform per chunk
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      if (!EXACT && tid + i * NT >= NSLOTS) break;
      const int gk = k0 + kin[i];
      float* o = &reg[i * SLOT_ELEMS];
      if (LMODE == L_SCALAR) {
        const bool kv = !TAIL || gk < p.K;
        float v = kv ? gk0[goff[i]] : g[goff[i] - (long)kin[i] * sk];          // clamped in-bounds address
        if (PRO == PA_RANK1 && IS_A) {
          const int kk = kv ? gk : 0;
          const int mm = p.a_trans ? kk : grow[i], nn = p.a_trans ? grow[i] : kk;
          v = v > 0.f ? rowv[mm] * colv[nn] : 0.f;
        }
        if (PRO == PB_ONES && !IS_A) v = ones[i] ? 1.f : v;
        o[0] = kv ? v : 0.f;
      } else if (LMODE == L_VECK) {
        const bool kv = !TAIL || gk < p.K;               // K % 4 == 0 or zero-padded rows (host guarantees)
        if (PRO == PA_RANK1B && IS_A) {                  // `g` = packed mask words of this z; A itself is never read
          const int kk = kv ? gk : 0;
          const unsigned int w = ((const unsigned int*)g)[(long)grow[i] * p.ab_g + (kk >> 5)] >> (kk & 31);
          const float rv = rowv[grow[i]];
          const f32x4 cv = *(const f32x4*)&colv[kk];
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = (kv && ((w >> j) & 1u)) ? rv * cv[j] : 0.f;
          continue;
        }
        const float* src = kv ? gk0 + goff[i] : g + goff[i] - kin[i];
        f32x4 v = *(const f32x4*)src;
        if (PRO == PA_RANK1 && IS_A) {
          const float rv = rowv[grow[i]];
          const f32x4 cv = *(const f32x4*)&colv[kv ? gk : 0];
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? rv * cv[j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = kv ? v[j] : 0.f;
      } else if (LMODE == L_VECKU) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool kv = !TAIL || (gk + j) < p.K;
          float v = kv ? gk0[goff[i] + j] : 0.f;
          if (PRO == PA_RANK1 && IS_A) {
            const int kk = kv ? gk + j : 0;
            const int mm = p.a_trans ? kk : grow[i], nn = p.a_trans ? grow[i] : kk;
            v = (kv && v > 0.f) ? rowv[mm] * colv[nn] : 0.f;
          }
          o[j] = v;
        }
      } else {  // L_BLK4
        // all global loads of the slot are issued before anything consumes them (a branch between a load and its use
        // would otherwise serialise the four row fetches)
        f32x4 vv[4];
        float rvv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool kv = !TAIL || (gk + j) < p.K;
          const float* src = kv ? gk0 + goff[i] + (long)j * sk : g + goff[i] - (long)kin[i] * sk;
          vv[j] = *(const f32x4*)src;
          if (PRO == PA_RANK1 && IS_A) rvv[j] = rowv[kv ? gk + j : 0];
        }
        f32x4 cv;
        if (PRO == PA_RANK1 && IS_A) cv = *(const f32x4*)&colv[grow[i]];
        if (PRO == PA_RANK1 && IS_A && p.tail_w_out) {       // uniform branch: raw vv = post-ReLU activation (>= 0)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool kv = !TAIL || (gk + j) < p.K;
            const float rv = kv ? rvv[j] : 0.f;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) tacc[i * 4 + rr] += rv * vv[j][rr];
            if (((tid + i * NT) % (ROWS / 4)) == 0) bacc += rv;
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool kv = !TAIL || (gk + j) < p.K;
          f32x4 v = vv[j];
          if (PRO == PA_RANK1 && IS_A) {
            const float rv = rvv[j];
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) v[rr] = v[rr] > 0.f ? rv * cv[rr] : 0.f;
          }
          if (PRO == PB_ONES && !IS_A) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) v[rr] = ((ones[i] >> rr) & 1) ? 1.f : v[rr];
          }
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) o[rr * 4 + j] = kv ? v[rr] : 0.f;    // transpose: o[row][k]
        }
      }
    }
  }

  // split staging: (element * sc) -> hi plane at lds_h, lo plane ROWS*PITCH elements further
  __device__ static inline void split4(const float* o, float sc, hx4& h, hx4& l) {
    const f32x4 v = {o[0] * sc, o[1] * sc, o[2] * sc, o[3] * sc};
    orl_split4(v, h, l);
  }
  __device__ inline void store_split(hx_t* __restrict__ lds_h, int tid, float sc) const {
    hx_t* lds_l = lds_h + ROWS * PITCH;
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      if (!EXACT && tid + i * NT >= NSLOTS) break;
      const float* o = &reg[i * SLOT_ELEMS];
      if (LMODE == L_SCALAR) {
        hx_t hh, ll; orl_split1(o[0] * sc, hh, ll);
        lds_h[loff[i]] = hh; lds_l[loff[i]] = ll;
      } else if (LMODE == L_VECK || LMODE == L_VECKU) {
        hx4 h, l; split4(o, sc, h, l);
        *(hx4*)(lds_h + loff[i]) = h; *(hx4*)(lds_l + loff[i]) = l;
      } else {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          hx4 h, l; split4(o + rr * 4, sc, h, l);
          *(hx4*)(lds_h + loff[i] + rr * PITCH) = h; *(hx4*)(lds_l + loff[i] + rr * PITCH) = l;
        }
      }
    }
  }

  // three planes (P_SPLIT3): hi at lds_h, mid ROWS*PITCH further, lo 2*ROWS*PITCH further
  __device__ inline void store_split3(hx_t* __restrict__ lds_h, int tid, float sc) const {
    hx_t* lds_m = lds_h + ROWS * PITCH;
    hx_t* lds_l = lds_m + ROWS * PITCH;
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      if (!EXACT && tid + i * NT >= NSLOTS) break;
      const float* o = &reg[i * SLOT_ELEMS];
      if (LMODE == L_SCALAR) {
        hx_t hh, mm, ll; orl_split1x3(o[0] * sc, hh, mm, ll);
        lds_h[loff[i]] = hh; lds_m[loff[i]] = mm; lds_l[loff[i]] = ll;
      } else if (LMODE == L_VECK || LMODE == L_VECKU) {
        hx4 h, m, l;
        orl_split4x3((f32x4){o[0] * sc, o[1] * sc, o[2] * sc, o[3] * sc}, h, m, l);
        *(hx4*)(lds_h + loff[i]) = h; *(hx4*)(lds_m + loff[i]) = m; *(hx4*)(lds_l + loff[i]) = l;
      } else {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          hx4 h, m, l;
          orl_split4x3((f32x4){o[rr * 4] * sc, o[rr * 4 + 1] * sc, o[rr * 4 + 2] * sc, o[rr * 4 + 3] * sc}, h, m, l);
          *(hx4*)(lds_h + loff[i] + rr * PITCH) = h; *(hx4*)(lds_m + loff[i] + rr * PITCH) = m; *(hx4*)(lds_l + loff[i] + rr * PITCH) = l;
        }
      }
    }
  }

  __device__ inline void store(float* __restrict__ lds, int tid) const {
#pragma unroll
    for (int i = 0; i < PER_THREAD; ++i) {
      if (!EXACT && tid + i * NT >= NSLOTS) break;
      const float* o = &reg[i * SLOT_ELEMS];
      float* d = lds + loff[i];
      if (LMODE == L_SCALAR) d[0] = o[0];
      else if (LMODE == L_VECK || LMODE == L_VECKU) *(f32x4*)d = (f32x4){o[0], o[1], o[2], o[3]};
      else {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) *(f32x4*)(d + rr * PITCH) = (f32x4){o[rr * 4], o[rr * 4 + 1], o[rr * 4 + 2], o[rr * 4 + 3]};
      }
    }
  }
};

template <class CFG, int LA, int LB, int PA, int PB, int EPI, int PREC>
__global__ __launch_bounds__(CFG::NT) void gemm16_kernel(const GemmP p) {
  constexpr int TM = CFG::TM, TN = CFG::TN, TK = CFG::kTK, NT = CFG::NT;
  constexpr int PITCH = (PREC == P_F32) ? CFG::PITCH : CFG::PITCH_H;   // LDS row pitch in elements of the plane type
  constexpr int MA = CFG::kMA, NB = CFG::kNB;
  static_assert(PREC == P_F32 || TK % 32 == 0, "the 16-bit MFMA consumes 32 k per instruction");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  // fp32: As [2][TM][PITCH] floats, Bs [2][TN][PITCH].   split: As [2][hi,lo][TM][PITCH] 16-bit planes, Bs likewise
  float* As = smem;
  float* Bs = smem + 2 * TM * PITCH;
  hx_t* Ah = (hx_t*)smem;
  constexpr int NPL = (PREC == P_SPLIT3) ? 3 : 2;                       // 16-bit planes per operand
  hx_t* Bh = Ah + 2 * NPL * TM * PITCH;

  const int tid = threadIdx.x;
  const int z = p.zmajor ? (int)(blockIdx.z * 8 + (blockIdx.x & 7)) : (int)blockIdx.z;
  if (p.zmajor && z >= p.nz_total) return;
  const int z0 = z / p.nz1, z1 = z - z0 * p.nz1;
  // XCD-aware tile mapping (MI355X: 8 XCDs with private L2s; workgroups are dealt round-robin over them by linear
  // id, and gridDim.x is padded to a multiple of 8 so the XCD of a block is blockIdx.x % 8 for every z).  Logical work
  // items (tile, k-split) are numbered tile-fastest and each XCD takes a CONTIGUOUS range of them, so the tiles that
  // share an operand panel (all column tiles of one row tile; all tiles of one k-split in wgrad) hit the same L2
  // instead of each fetching the panel from HBM.  Only speed depends on the placement, never correctness.
  const int tiles_n = (p.N + TN - 1) / TN;
  const int tiles = ((p.M + TM - 1) / TM) * tiles_n;
  const int total = tiles * p.ksplit;
  // Batched problems with FEW tiles each (the 256-row products of many nets: 4 tiles of 128 x 128) use the z-major mapping instead
  // (GemmP::zmajor): with the item-major one the 4 tiles of a net landed on 4 different XCDs and every tile fetched its operand panels
  // from HBM -- PMC on EDAC's edac.t / edac.wgrad / critic.bwd.wgrad launches: 1.85 / 1.35 GB read against 0.7 - 1.0 GB algorithmic.
  const int per_xcd = gridDim.x >> 3;
  const int item = p.zmajor ? (int)(blockIdx.x >> 3) : (int)((blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3));
  if (item >= total) return;
  const int ks = item / tiles, tile = item - ks * tiles;
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * TM, n0 = tn * TN;

  const float* __restrict__ Ag = (PA == PA_RANK1B) ? (const float*)(p.a_bits + z0 * p.ab_s0 + z1 * p.ab_s1) : p.A.at(z0, z1);
  const float* __restrict__ Bg = p.B.at(z0, z1);
  const float* __restrict__ rowv = p.rowv.at(z0, z1);
  const float* __restrict__ colv = p.colv.at(z0, z1);
  // P_SPLIT: power-of-two operand scales (GemmP::a_scale ..), uniform per workgroup
  float sc_a = 1.f, sc_b = 1.f;
  if (PREC != P_F32) {
    sc_a = (p.a_scale != 0.f ? p.a_scale : 1.f) * (p.a_dscale ? p.a_dscale[z0] : 1.f);
    sc_b = (p.b_scale != 0.f ? p.b_scale : 1.f) * (p.b_dscale ? p.b_dscale[z0] : 1.f);
  }

  // K range of this split (chunk aligned); only the globally last chunk can be partial
  const int kchunks = (p.K + TK - 1) / TK;
  const int per = (kchunks + p.ksplit - 1) / p.ksplit;
  const int kc_begin = ks * per;
  int kc_end = kc_begin + per;
  if (kc_end > kchunks) kc_end = kchunks;
  const int kfull = p.K / TK;             // chunks [0, kfull) need no k checks

  TileLoader<TM, TK, NT, PITCH, LA, true, PA> la;
  TileLoader<TN, TK, NT, PITCH, LB, false, PB> lb;
  la.init(p, m0, tid);
  lb.init(p, n0, tid);

  auto load_chunk = [&](int kc) {
    if (kc < kfull) { la.template load<false>(p, Ag, rowv, colv, kc * TK, tid); lb.template load<false>(p, Bg, nullptr, nullptr, kc * TK, tid); }
    else { la.template load<true>(p, Ag, rowv, colv, kc * TK, tid); lb.template load<true>(p, Bg, nullptr, nullptr, kc * TK, tid); }
  };

  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave / CFG::kWN, wn = wave - wm * CFG::kWN;
  const int li = lane & 15, lq = lane >> 4;
  const int wrow0 = wm * MA * 16, wcol0 = wn * NB * 16;

  f32x4 acc[MA][NB];
#pragma unroll
  for (int a = 0; a < MA; ++a)
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // E_WGRAD with bias_out: the bias gradient db[m] = sum_k A[m][k] is accumulated by the first column tile with one
  // extra MFMA per row block whose other operand is all ones (no extra B column, so N stays tile-aligned)
  const bool want_bias = (EPI == E_WGRAD) && (p.bias_out != nullptr) && (tn == 0) && (wn == 0);
  f32x4 accb[MA];
#pragma unroll
  for (int a = 0; a < MA; ++a) accb[a] = (f32x4){0.f, 0.f, 0.f, 0.f};

  auto store_chunk = [&](int buf) {
    if (PREC == P_F32) { la.store(As + buf * TM * PITCH, tid); lb.store(Bs + buf * TN * PITCH, tid); }
    else if (PREC == P_SPLIT3) { la.store_split3(Ah + buf * 3 * TM * PITCH, tid, sc_a); lb.store_split3(Bh + buf * 3 * TN * PITCH, tid, sc_b); }
    else { la.store_split(Ah + buf * 2 * TM * PITCH, tid, sc_a); lb.store_split(Bh + buf * 2 * TN * PITCH, tid, sc_b); }
  };
  if (kc_begin < kc_end) {
    load_chunk(kc_begin);
    store_chunk(0);
  }
  __syncthreads();
  for (int kc = kc_begin; kc < kc_end; ++kc) {
    const int buf = (kc - kc_begin) & 1;
    const bool more = kc + 1 < kc_end;
    if (more) load_chunk(kc + 1);
    if (PREC == P_F32) {
      const float* as = As + buf * TM * PITCH;
      const float* bs = Bs + buf * TN * PITCH;
#pragma unroll
      for (int kk = 0; kk < TK; kk += 16) {
        f32x4 fa[MA], fb[NB];
#pragma unroll
        for (int a = 0; a < MA; ++a) fa[a] = *(const f32x4*)&as[(wrow0 + a * 16 + li) * PITCH + kk + 4 * lq];
#pragma unroll
        for (int b = 0; b < NB; ++b) fb[b] = *(const f32x4*)&bs[(wcol0 + b * 16 + li) * PITCH + kk + 4 * lq];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int a = 0; a < MA; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[b][s], fa[a][s], acc[a][b], 0, 0, 0);   // transposed tile: D[n][m]
        if (EPI == E_WGRAD && want_bias) {
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int a = 0; a < MA; ++a) accb[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f, fa[a][s], accb[a], 0, 0, 0);
        }
      }
    } else if (PREC == P_SPLIT3) {
      // three planes per operand: lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi (smallest terms first)
      const hx_t* ah = Ah + buf * 3 * TM * PITCH;
      const hx_t* am = ah + TM * PITCH;
      const hx_t* al = am + TM * PITCH;
      const hx_t* bh = Bh + buf * 3 * TN * PITCH;
      const hx_t* bm = bh + TN * PITCH;
      const hx_t* bl = bm + TN * PITCH;
#pragma unroll
      for (int kk = 0; kk < TK; kk += 32) {
        hx8 fah[MA], fam[MA], fal[MA], fbh[NB], fbm[NB], fbl[NB];
#pragma unroll
        for (int a = 0; a < MA; ++a) {
          const int o = (wrow0 + a * 16 + li) * PITCH + kk + 8 * lq;
          fah[a] = *(const hx8*)&ah[o]; fam[a] = *(const hx8*)&am[o]; fal[a] = *(const hx8*)&al[o];
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const int o = (wcol0 + b * 16 + li) * PITCH + kk + 8 * lq;
          fbh[b] = *(const hx8*)&bh[o]; fbm[b] = *(const hx8*)&bm[o]; fbl[b] = *(const hx8*)&bl[o];
        }
#pragma unroll
        for (int a = 0; a < MA; ++a)
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            acc[a][b] = ORL_MFMA_16x16x32(fbl[b], fah[a], acc[a][b]);
            acc[a][b] = ORL_MFMA_16x16x32(fbh[b], fal[a], acc[a][b]);
            acc[a][b] = ORL_MFMA_16x16x32(fbm[b], fam[a], acc[a][b]);
            acc[a][b] = ORL_MFMA_16x16x32(fbm[b], fah[a], acc[a][b]);
            acc[a][b] = ORL_MFMA_16x16x32(fbh[b], fam[a], acc[a][b]);
            acc[a][b] = ORL_MFMA_16x16x32(fbh[b], fah[a], acc[a][b]);
          }
        if (EPI == E_WGRAD && want_bias) {
          hx8 one;
#pragma unroll
          for (int j = 0; j < 8; ++j) one[j] = (hx_t)1.0f;
#pragma unroll
          for (int a = 0; a < MA; ++a) {
            accb[a] = ORL_MFMA_16x16x32(one, fal[a], accb[a]);
            accb[a] = ORL_MFMA_16x16x32(one, fam[a], accb[a]);
            accb[a] = ORL_MFMA_16x16x32(one, fah[a], accb[a]);
          }
        }
      }
    } else {
      // lane (li, lq) supplies 8 consecutive k (8*lq ..) of row li for both operands of the 16x16x32 16-bit MFMA
      const hx_t* ah = Ah + buf * 2 * TM * PITCH;
      const hx_t* al = ah + TM * PITCH;
      const hx_t* bh = Bh + buf * 2 * TN * PITCH;
      const hx_t* bl = bh + TN * PITCH;
#pragma unroll
      for (int kk = 0; kk < TK; kk += 32) {
        hx8 fah[MA], fal[MA], fbh[NB], fbl[NB];
#pragma unroll
        for (int a = 0; a < MA; ++a) {
          const int o = (wrow0 + a * 16 + li) * PITCH + kk + 8 * lq;
          fah[a] = *(const hx8*)&ah[o]; fal[a] = *(const hx8*)&al[o];
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          const int o = (wcol0 + b * 16 + li) * PITCH + kk + 8 * lq;
          fbh[b] = *(const hx8*)&bh[o]; fbl[b] = *(const hx8*)&bl[o];
        }
#pragma unroll
        for (int a = 0; a < MA; ++a)
#pragma unroll
          for (int b = 0; b < NB; ++b) {
            acc[a][b] = ORL_MFMA_16x16x32(fbl[b], fah[a], acc[a][b]);
            acc[a][b] = ORL_MFMA_16x16x32(fbh[b], fal[a], acc[a][b]);
            acc[a][b] = ORL_MFMA_16x16x32(fbh[b], fah[a], acc[a][b]);
          }
        if (EPI == E_WGRAD && want_bias) {
          hx8 one;
#pragma unroll
          for (int j = 0; j < 8; ++j) one[j] = (hx_t)1.0f;
#pragma unroll
          for (int a = 0; a < MA; ++a) {
            accb[a] = ORL_MFMA_16x16x32(one, fal[a], accb[a]);
            accb[a] = ORL_MFMA_16x16x32(one, fah[a], accb[a]);
          }
        }
      }
    }
    if (more) store_chunk(buf ^ 1);
    __syncthreads();
  }

  if (PREC != P_F32) {                  // divide the operand scales out (exact: powers of two)
    const float inv = 1.0f / (sc_a * sc_b), inv_a = 1.0f / sc_a;
#pragma unroll
    for (int a = 0; a < MA; ++a) {
#pragma unroll
      for (int b = 0; b < NB; ++b) acc[a][b] *= inv;
      accb[a] *= inv_a;
    }
  }
  // ---- epilogue.  The MFMA was issued with the operands swapped (W-tile as A, X-tile as B), so the 16x16 block
  // lives transposed in the accumulators: lane (li, lq) holds C[m = li][n = 4*lq + reg], i.e. four CONSECUTIVE
  // output columns of one row -> one 16-byte store (and 16-byte bias / mask loads) per block instead of four
  // dword stores. ----
  float* Cg = p.C + z0 * p.c_s0 + z1 * p.c_s1 + (long)ks * p.c_ks;
  const float* __restrict__ bias = p.bias.at(z0, z1);
  const float* __restrict__ aux = p.aux.at(z0, z1);
  float* bo = (EPI == E_WGRAD && p.bias_out) ? p.bias_out + z0 * p.bo_s0 + z1 * p.bo_s1 + (long)ks * p.bo_ks : nullptr;
  const bool vec_ok = (p.c_sn == 1) && ((p.c_sr & 3) == 0) && ((((uintptr_t)Cg) & 15) == 0) &&
                      (EPI != E_MASK || p.aux_bits != nullptr || (((p.aux_sr & 3) == 0) && ((((uintptr_t)aux) & 15) == 0))) &&
                      ((EPI != E_BIAS && EPI != E_BIAS_RELU) || ((((uintptr_t)bias) & 15) == 0));
  if (EPI == E_WGRAD && PA == PA_RANK1 && LA == L_BLK4) {
    if (p.tail_w_out && tn == 0) {
      // deterministic reduction of the per-thread partials: sred[k-slot][row], summed in k-slot order
      constexpr int KQ = TK / 4, Q = TM / 4;
      float* sred = smem;                                   // the staging tiles are dead after the last barrier
      float* sb = smem + KQ * TM;
#pragma unroll
      for (int i = 0; i < decltype(la)::PER_THREAD; ++i) {
        const int e = tid + i * NT;
        if (e < decltype(la)::NSLOTS) {
          const int r = 4 * (e % Q), kq = e / Q;
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) sred[kq * TM + r + rr] = la.tacc[i * 4 + rr];
        }
      }
      sb[tid] = la.bacc;                                     // zero for threads that own no first-row-block slot
      __syncthreads();
      float* tw = p.tail_w_out + z0 * p.tw_s0 + z1 * p.tw_s1 + (long)ks * p.c_ks;
      for (int r = tid; r < TM; r += NT) {
        const int m = m0 + r;
        if (m < p.M) {
          float sacc = 0.f;
#pragma unroll
          for (int kq = 0; kq < KQ; ++kq) sacc += sred[kq * TM + r];
          tw[m] = sacc;
        }
      }
      if (tm == 0 && tid == 0 && p.tail_b_out) {
        float sacc = 0.f;
        for (int t = 0; t < NT; ++t) sacc += sb[t];
        p.tail_b_out[z0 * p.tw_s0 + z1 * p.tb_s1 + (long)ks * p.c_ks] = sacc;
      }
    }
  }
  if (EPI == E_WGRAD && want_bias && lq == 0) {
#pragma unroll
    for (int a = 0; a < MA; ++a) {
      const int m = m0 + wrow0 + a * 16 + li;
      if (m < p.M) bo[m] = accb[a][0];      // every n-row of the ones-block holds the same sum; lane (li, 0) reg 0 = D[0][m]
    }
  }
  // Coalesced path: the accumulator blocks hold 16 rows x 64 B each, so direct stores would write (and the mask would read)
  // 64-byte pieces of 16 different rows per instruction.  Stage the TM x TN tile through LDS (the operand buffers are dead)
  // and let every wave move whole row segments: 64 lanes x 16 B = two 512-B runs (TN = 128) per instruction.
  constexpr int CP = TN + 4;                                   // LDS pitch of the staged C tile (floats)
  constexpr bool LDS_EPI_FITS = (size_t)TM * CP * sizeof(float) <= CFG::epi_lds_limit(PREC) && (NT % (TN / 4)) == 0;
  constexpr bool W0_CAP = (EPI == E_MASK) && (TN / (NT / 64) == 32) && (TM % 16 == 0);
  const bool w0 = W0_CAP && (p.w0_out != nullptr);             // uniform; the host only asks when the LDS path below is taken
  // Weight gradients stored (in, out)-major (EnsembleLinear: c_sr == 1, consecutive m are consecutive in memory): stage the tile
  // TRANSPOSED and store whole m-runs, instead of 64-byte pieces of 16 different rows per instruction in the direct path.
  constexpr bool TR_CAP = (EPI == E_WGRAD) && (TM == TN) && LDS_EPI_FITS && (NT % (TM / 4)) == 0;
  const bool tr_ok = TR_CAP && p.c_sr == 1 && p.c_sn != 1 && (p.c_sn & 3) == 0 && (p.c_s0 & 3) == 0 && (p.c_s1 & 3) == 0 && (p.c_ks & 3) == 0 &&
                     ((((uintptr_t)p.C) & 15) == 0) && (p.M & 3) == 0;                    // uniform per launch
  if (TR_CAP && tr_ok) {
    float* cs = smem;
    if (PA == PA_RANK1 && LA == L_BLK4) __syncthreads();      // the tail-gradient reduction above used smem
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
      for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int j = 0; j < 4; ++j) cs[(wcol0 + b * 16 + 4 * lq + j) * CP + wrow0 + a * 16 + li] = acc[a][b][j];
    __syncthreads();
    constexpr int C4 = TM / 4, RPP = NT / C4, NPASS = (TN + RPP - 1) / RPP;   // float4 pieces per n-row, n-rows per pass
    const int c4 = tid % C4, r0 = tid / C4;
    const int m = m0 + 4 * c4;
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int r = r0 + i * RPP, n = n0 + r;
      if (r < TN && n < p.N && m < p.M) *(f32x4*)&Cg[(long)n * p.c_sn + m] = *(const f32x4*)&cs[r * CP + 4 * c4];
    }
  } else if (LDS_EPI_FITS && (vec_ok || (w0 && p.C == nullptr)) && (p.N & 3) == 0) {             // uniform per workgroup
    float* cs = smem;
    constexpr int C4 = TN / 4, RPP = NT / C4, NPASS = (TM + RPP - 1) / RPP;   // float4 columns per row, rows per pass
    constexpr bool MB_CAP = (C4 % 8) == 0;                                     // eight lanes of a row own one 32-column mask word
    const int c4 = tid % C4, r0 = tid / C4;
    const int n = n0 + 4 * c4;
    const bool n_ok = n < p.N;
    // the mask tile is fetched first (clamped addresses, no branches) so its latency hides behind the LDS staging
    f32x4 hv[EPI == E_MASK ? NPASS : 1];
    const bool xbits = (EPI == E_MASK) && (p.aux_bits != nullptr);           // uniform: the mask comes as packed bits
    if (EPI == E_MASK) {
      const unsigned int* xb = p.aux_bits + z0 * p.xb_s0 + z1 * p.xb_s1;
#pragma unroll
      for (int i = 0; i < NPASS; ++i) {
        int m = m0 + r0 + i * RPP;
        m = m < p.M ? m : p.M - 1;
        if (xbits) hv[i][0] = __uint_as_float(xb[(long)m * p.xb_g + ((n_ok ? n : 0) >> 5)]);
        else hv[i] = *(const f32x4*)&aux[(long)m * p.aux_sr + (n_ok ? n : 0)];
      }
    }
    // fused layer-0 weight gradient: this thread's pieces of the X tile [TM][W0_XP], fetched now for the same reason
    constexpr int XQ = W0_XP / 4, XPT = W0_CAP ? (TM * XQ + NT - 1) / NT : 1;
    f32x4 xv[XPT];
    if (W0_CAP && w0) {
      const float* __restrict__ xg = p.w0_x.at(z0, z1);
      const int xq = (int)(p.w0_xsr >> 2);                     // float4 per X row in memory (<= XQ)
#pragma unroll
      for (int i = 0; i < XPT; ++i) {
        const int e = tid + i * NT, r = e / XQ, q = e - r * XQ;
        int m = m0 + r; m = m < p.M ? m : p.M - 1;
        xv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (q < xq && e < TM * XQ) xv[i] = *(const f32x4*)&xg[(long)m * p.w0_xsr + 4 * q];
#pragma unroll
        for (int j = 0; j < 4; ++j) if (4 * q + j == p.w0_in) xv[i][j] = 1.0f;     // ones column -> bias gradient
      }
    }
    f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (EPI == E_BIAS || EPI == E_BIAS_RELU) bv = *(const f32x4*)&bias[n_ok ? n : 0];
    const bool tq = (EPI == E_BIAS_RELU) && (p.tq_out != nullptr);       // uniform
    f32x4 tw = (f32x4){0.f, 0.f, 0.f, 0.f};
    float tq_bias = 0.f;
    if (EPI == E_BIAS_RELU && tq) {
      if (n_ok) tw = *(const f32x4*)&(p.tq_w.at(z0, z1)[n]);
      if (tn == 0) tq_bias = p.tq_b.at(z0, z1)[0];
    }
    if (EPI == E_WGRAD && PA == PA_RANK1 && LA == L_BLK4) __syncthreads();   // the tail-gradient reduction above used smem
#pragma unroll
    for (int a = 0; a < MA; ++a)
#pragma unroll
      for (int b = 0; b < NB; ++b)
        *(f32x4*)&cs[(wrow0 + a * 16 + li) * CP + wcol0 + b * 16 + 4 * lq] = acc[a][b];
    if (W0_CAP && w0) {
      float* xs = cs + TM * CP;                                // X tile behind the C tile
#pragma unroll
      for (int i = 0; i < XPT; ++i) {
        const int e = tid + i * NT;
        if (e < TM * XQ) *(f32x4*)&xs[(e / XQ) * W0_XP + 4 * (e % XQ)] = xv[i];
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NPASS; ++i) {
      const int r = r0 + i * RPP;
      const int m = m0 + r;
      if (r < TM) {
        f32x4 v = *(const f32x4*)&cs[r * CP + 4 * c4];
        if (EPI == E_BIAS || EPI == E_BIAS_RELU) v += bv;
        if (EPI == E_BIAS_RELU) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : 0.f;
        }
        if (EPI == E_MASK) {
          if (xbits) {
            const unsigned int w = __float_as_uint(hv[i][0]) >> (n & 31);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = ((w >> j) & 1u) ? v[j] : 0.f;
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = hv[i][j] > 0.f ? v[j] : 0.f;
          }
        }
        if (EPI == E_BIAS_RELU && MB_CAP && p.mb_out) {        // uniform: emit the packed mask of this tile's rows
          // 4 bits per lane, 8 consecutive lanes of a row make one 32-bit word (OR-reduction over lane bits 0..2)
          unsigned int w = 0;
          if (n_ok) w = ((v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u)) << (4 * (c4 & 7));
          w |= __shfl_xor(w, 1); w |= __shfl_xor(w, 2); w |= __shfl_xor(w, 4);
          if ((c4 & 7) == 0 && n_ok && m < p.M) p.mb_out[z0 * p.mb_s0 + z1 * p.mb_s1 + (long)m * p.mb_g + (n >> 5)] = w;
        }
        if (n_ok && m < p.M && (!W0_CAP || p.C != nullptr)) *(f32x4*)&Cg[(long)m * p.c_sr + n] = v;
        if (EPI == E_BIAS_RELU && tq) {                        // row m of the tile is spread over C4 consecutive lanes
          float pd = (v[0] * tw[0] + v[1] * tw[1]) + (v[2] * tw[2] + v[3] * tw[3]);
#pragma unroll
          for (int o = C4 / 2; o > 0; o >>= 1) pd += __shfl_xor(pd, o);
          if (c4 == 0 && m < p.M) {
            if (tn == 0) p.tq_out[z0 * p.tq_s0 + z1 * p.tq_s1 + (long)m * p.tq_sm] = pd + tq_bias;
            else p.tq_part[z0 * p.tq_ps0 + z1 * p.tq_ps1 + (long)(tn - 1) * p.tq_ts + m] = pd;
          }
        }
        if (W0_CAP && w0) {                                    // masked values (zero beyond M / N) back into the staged tile
          if (!(n_ok && m < p.M)) v = (f32x4){0.f, 0.f, 0.f, 0.f};
          *(f32x4*)&cs[r * CP + 4 * c4] = v;
        }
      }
    }
    if (W0_CAP && w0) {
      const float* xs = cs + TM * CP;
      __syncthreads();
      // wave w owns columns [32w, 32w+32) of the tile: D[n][c] = sum_r cs[r][n] * xs[r][c], 2 x 2 blocks of 16 x 16
      f32x4 d[2][2];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) d[nb][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const int ncol0 = 32 * wave;
      const int c1 = (16 + li) < W0_XP ? 16 + li : W0_XP - 1;   // columns >= W0_XP are never stored
#pragma unroll 2
      for (int k0 = 0; k0 < TM; k0 += 16) {
        float av[2][4], bw[2][4];
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
          const int r = k0 + 4 * lq + sidx;
          av[0][sidx] = cs[r * CP + ncol0 + li];
          av[1][sidx] = cs[r * CP + ncol0 + 16 + li];
          bw[0][sidx] = xs[r * W0_XP + li];
          bw[1][sidx] = xs[r * W0_XP + c1];
        }
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
              d[nb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[nb][sidx], bw[cb][sidx], d[nb][cb], 0, 0, 0);
      }
      // lane (li, lq) holds D[n = 4*lq + reg][c = li] of each block
      float* wo = p.w0_out + z0 * p.w0_s0 + z1 * p.w0_s1 + (long)tm * p.w0_ks;
      float* bo0 = p.w0_bias + z0 * p.w0_s0 + z1 * p.w0_bs1 + (long)tm * p.w0_ks;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            const int nn = n0 + ncol0 + 16 * nb + 4 * lq + rg, c = 16 * cb + li;
            if (nn < p.N) {
              if (c < p.w0_in) wo[(long)nn * p.w0_sr + c] = d[nb][cb][rg];
              else if (c == p.w0_in) bo0[nn] = d[nb][cb][rg];
            }
          }
    }
  } else {
#pragma unroll
    for (int a = 0; a < MA; ++a) {
      const int m = m0 + wrow0 + a * 16 + li;
  #pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int nb = n0 + wcol0 + b * 16 + 4 * lq;
        if (m >= p.M || nb >= p.N) continue;
        f32x4 v = acc[a][b];
        const int nlim = p.N;
        if (vec_ok && nb + 4 <= nlim) {
          if (EPI == E_BIAS || EPI == E_BIAS_RELU) { const f32x4 bv = *(const f32x4*)&bias[nb]; v += bv; }
          if (EPI == E_BIAS_RELU) {
  #pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = v[r] > 0.f ? v[r] : 0.f;
          }
          if (EPI == E_MASK) {
            const f32x4 hv = *(const f32x4*)&aux[(long)m * p.aux_sr + nb];
  #pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = hv[r] > 0.f ? v[r] : 0.f;
          }
          *(f32x4*)&Cg[(long)m * p.c_sr + nb] = v;
        } else {
  #pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int n = nb + r;
            if (n >= p.N) continue;
            float x = v[r];
            if (EPI == E_BIAS) x += bias[n];
            if (EPI == E_BIAS_RELU) { x += bias[n]; x = x > 0.f ? x : 0.f; }
            if (EPI == E_MASK) x = aux[(long)m * p.aux_sr + n] > 0.f ? x : 0.f;
            Cg[(long)m * p.c_sr + (long)n * p.c_sn] = x;
          }
        }
      }
    }
}
}

template <class CFG, int LA, int LB, int PA, int PB, int EPI, int PREC = P_F32>
static inline hipError_t launch_inst(const GemmP& p, int nz, hipStream_t st) {
  const int tiles = ((p.M + CFG::TM - 1) / CFG::TM) * ((p.N + CFG::TN - 1) / CFG::TN);
  dim3 grid((tiles * p.ksplit + 7) & ~7, 1, nz), block(CFG::NT);     // padded to the 8 XCDs (see the kernel's tile mapping)
  GemmP q = p;
  static const int zm_max = [] { const char* f = getenv("ORL_GEMM_ZMAJOR_MAX"); return f ? atoi(f) : 16; }();   // 0 disables (A/B runs)
  if (nz >= 8 && tiles * p.ksplit > 1 && tiles * p.ksplit <= zm_max) {   // few tiles per problem, many problems: keep a problem on one XCD
    q.zmajor = 1; q.nz_total = nz;
    grid = dim3(8 * tiles * p.ksplit, 1, (nz + 7) / 8);
  }
  size_t lds = CFG::lds_bytes(PREC);
  if (CFG::epi_lds_bytes() <= CFG::epi_lds_limit(PREC)) lds = std::max(lds, CFG::epi_lds_bytes());   // staged C tile (kernel: LDS_EPI_FITS)
  if (EPI == E_MASK && p.w0_out) lds = std::max(lds, sizeof(float) * ((size_t)CFG::TM * (CFG::TN + 4) + (size_t)CFG::TM * W0_XP));
  auto kern = gemm16_kernel<CFG, LA, LB, PA, PB, EPI, PREC>;
  if (lds > 64 * 1024) {
    static std::atomic<size_t> raised_to{0};   // per instantiation; the request can grow (fused layer-0 gradient), engines may launch from several threads
    if (lds > raised_to.load(std::memory_order_acquire)) {
      hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      raised_to.store(lds, std::memory_order_release);
    }
  }
  hipLaunchKernelGGL(kern, grid, block, lds, st, q);
  return hipGetLastError();
}

template <class CFG, int PA, int PB, int EPI, int PREC>
static inline hipError_t launch_cfg(const GemmP& p, int la, int lb, int nz, hipStream_t st) {
  // supported loader pairs; anything else falls back to the scalar loaders
  if (la == L_VECK && lb == L_VECK) return launch_inst<CFG, L_VECK, L_VECK, PA, PB, EPI, PREC>(p, nz, st);
  if (la == L_VECK && lb == L_BLK4) return launch_inst<CFG, L_VECK, L_BLK4, PA, PB, EPI, PREC>(p, nz, st);
  if (la == L_BLK4 && lb == L_BLK4) return launch_inst<CFG, L_BLK4, L_BLK4, PA, PB, EPI, PREC>(p, nz, st);
  if (la == L_VECK && lb == L_VECKU) return launch_inst<CFG, L_VECK, L_VECKU, PA, PB, EPI, PREC>(p, nz, st);
  if (la == L_VECK) return launch_inst<CFG, L_VECK, L_SCALAR, PA, PB, EPI, PREC>(p, nz, st);
  return launch_inst<CFG, L_SCALAR, L_SCALAR, PA, PB, EPI, PREC>(p, nz, st);
}
template <class CFG, int PA, int PB, int EPI>
static inline hipError_t launch_cfg_prec(const GemmP& p, int la, int lb, int nz, hipStream_t st, int prec) {
  if (prec == P_BF16X3) return launch_cfg<CFG, PA, PB, EPI, P_BF16X3>(p, la, lb, nz, st);
  if (prec == P_SPLIT3) {
    if constexpr (CFG::lds_bytes(P_SPLIT3) <= (size_t)160 * 1024) return launch_cfg<CFG, PA, PB, EPI, P_SPLIT3>(p, la, lb, nz, st);
    else return launch_cfg<CfgSq, PA, PB, EPI, P_SPLIT3>(p, la, lb, nz, st);      // (the 256 x 128 tile's three-plane buffers exceed the LDS: 128 x 128 instead)
  }
  return launch_cfg<CFG, PA, PB, EPI, P_F32>(p, la, lb, nz, st);
}

template <int PA, int PB, int EPI>
hipError_t launch_gemm(int cfg, const GemmP& p, int nz, hipStream_t st, bool a_kpad, bool force_scalar, int prec) {
  if (cfg == CFG_AUTO) cfg = pick_cfg(p.M, p.N, p.K, nz);
  int la = L_SCALAR, lb = L_SCALAR;
  if (!force_scalar) {
    la = pick_loader(p.A, p.a_sr, p.a_sk, p.K, a_kpad, p.a_rlim);
    lb = pick_loader(p.B, p.b_sr, p.b_sk, p.K, false, p.b_rlim);
    if (PA == PA_RANK1) {   // the rank-1 prologue reads colv / rowv with the operand's vector shape
      if (la == L_VECK && (p.a_trans != 0 || !aligned16(p.colv.p) || (p.colv.s0 & 3) || (p.colv.s1 & 3))) la = L_SCALAR;
      if (la == L_BLK4 && (p.a_trans != 1 || !aligned16(p.colv.p) || (p.colv.s0 & 3) || (p.colv.s1 & 3))) la = L_SCALAR;
    }
  }
  switch (cfg) {
    case CFG_BIG: return launch_cfg_prec<CfgBig, PA, PB, EPI>(p, la, lb, nz, st, prec);
    case CFG_MID: return launch_cfg_prec<CfgMid, PA, PB, EPI>(p, la, lb, nz, st, prec);
    case CFG_SMALL: return launch_cfg_prec<CfgSmall, PA, PB, EPI>(p, la, lb, nz, st, prec);
    case CFG_SQ: return launch_cfg_prec<CfgSq, PA, PB, EPI>(p, la, lb, nz, st, prec);
    case CFG_SQ8: return launch_cfg_prec<CfgSq8, PA, PB, EPI>(p, la, lb, nz, st, prec);
    case CFG_WG: return launch_cfg_prec<CfgWg, PA, PB, EPI>(p, la, lb, nz, st, prec);
    default: return launch_cfg_prec<CfgTall, PA, PB, EPI>(p, la, lb, nz, st, prec);
  }
}

template <int EPI>
hipError_t launch_gemm_rank1_bits(int cfg, const GemmP& p, int nz, hipStream_t st, int prec) {
  if (!rank1_bits_supported(cfg, p, false)) return hipErrorInvalidValue;
  const int lb = pick_loader(p.B, p.b_sr, p.b_sk, p.K, false, p.b_rlim);
#define ORL_RB(CFG, LB, PREC) launch_inst<CFG, L_VECK, LB, PA_RANK1B, PB_PLAIN, EPI, PREC>(p, nz, st)
  if (cfg == CFG_SQ) {
    if (prec == P_BF16X3) return lb == L_BLK4 ? ORL_RB(CfgSq, L_BLK4, P_BF16X3) : ORL_RB(CfgSq, L_VECK, P_BF16X3);
    if (prec == P_SPLIT3) return lb == L_BLK4 ? ORL_RB(CfgSq, L_BLK4, P_SPLIT3) : ORL_RB(CfgSq, L_VECK, P_SPLIT3);
    return lb == L_BLK4 ? ORL_RB(CfgSq, L_BLK4, P_F32) : ORL_RB(CfgSq, L_VECK, P_F32);
  }
  if (prec == P_BF16X3) return lb == L_BLK4 ? ORL_RB(CfgBig, L_BLK4, P_BF16X3) : ORL_RB(CfgBig, L_VECK, P_BF16X3);
  if (prec == P_SPLIT3) return lb == L_BLK4 ? ORL_RB(CfgBig, L_BLK4, P_SPLIT3) : ORL_RB(CfgBig, L_VECK, P_SPLIT3);
  return lb == L_BLK4 ? ORL_RB(CfgBig, L_BLK4, P_F32) : ORL_RB(CfgBig, L_VECK, P_F32);
#undef ORL_RB
}

#ifdef ORL_GEMM_TUNE_TU      // defined by gemm_inst_tune.hip only (non-template entry point: one definition)
// ---- tuning tap (orl_debug_gemm_time): the three hot kernel kinds on the main + two extra tile shapes ----
typedef GemmCfg<1, 4, 4, 4, 32> CfgT7;    // 64 x 256, 4 waves
typedef GemmCfg<2, 2, 2, 4, 32> CfgT11;   // 64 x 128
template <class CFG, int PREC>
static inline hipError_t launch_tune_kind_p(int kind, const GemmP& p, int nz, hipStream_t st) {
  if (kind == 0) return launch_inst<CFG, L_VECK, L_VECK, PA_PLAIN, PB_PLAIN, E_BIAS_RELU, PREC>(p, nz, st);
  if (kind == 1) return launch_inst<CFG, L_VECK, L_BLK4, PA_RANK1, PB_PLAIN, E_MASK, PREC>(p, nz, st);
  return launch_inst<CFG, L_BLK4, L_BLK4, PA_RANK1, PB_PLAIN, E_WGRAD, PREC>(p, nz, st);
}
template <class CFG>
static inline hipError_t launch_tune_kind(int kind, const GemmP& p, int nz, hipStream_t st, int prec) {
  if (prec == P_BF16X3) return launch_tune_kind_p<CFG, P_BF16X3>(kind, p, nz, st);
  return launch_tune_kind_p<CFG, P_F32>(kind, p, nz, st);
}
hipError_t launch_tune(int cfg_in, int kind, const GemmP& p, int nz, hipStream_t st) {
  const int prec = (cfg_in & 32) ? P_BF16X3 : P_F32;      // bit 5 selects the split-bf16 multiply
  switch (cfg_in & 31) {
    case 0: return launch_tune_kind<CfgBig>(kind, p, nz, st, prec);
    case 1: return launch_tune_kind<CfgMid>(kind, p, nz, st, prec);
    case 2: return launch_tune_kind<CfgSmall>(kind, p, nz, st, prec);
    case 3: return launch_tune_kind<CfgTall>(kind, p, nz, st, prec);
    case 4: return launch_tune_kind<CfgSq>(kind, p, nz, st, prec);
    case 7: return launch_tune_kind<CfgT7>(kind, p, nz, st, prec);
    default: return launch_tune_kind<CfgT11>(kind, p, nz, st, prec);
  }
}

#endif

}  // namespace orl
