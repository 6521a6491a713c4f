// engine.hip — MI355X-native update engine: host side (arenas, launch schedule) + C ABI.
// Build: hipcc -O3 --offload-arch=gfx950 -fPIC -shared engine.hip -o liborlengine.so
#include "engine.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>

namespace orl {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }

static int fail(const std::string& msg) {
  set_error(msg);
  return -1;
}

static inline int rup(int x, int m) { return (x + m - 1) / m * m; }

// ---------------------------------------------------------------------------------------------
// layouts (reference state_dict key names: SURVEY Appendix B)
// ---------------------------------------------------------------------------------------------
static void add_tensor(NetLayout& l, const std::string& name, long off, std::initializer_list<long> shape) {
  TensorInfo t;
  t.name = name;
  t.off = off;
  t.ndim = (int)shape.size();
  int i = 0;
  for (long s : shape) t.shape[i++] = s;
  l.tensors.push_back(t);
}

enum TailKind { TAIL_CRITIC, TAIL_TANH_GAUSS, TAIL_GAUSS, TAIL_DET };

static NetLayout make_mlp_layout(int in_dim, const int* hidden, int L, TailKind tail, int act_dim) {
  NetLayout l;
  l.present = true;
  l.in_dim = in_dim;
  l.L = L;
  long off = 0;
  int d = in_dim;
  for (int i = 0; i < L; ++i) {
    l.H[i] = hidden[i];
    l.w_off[i] = off;
    add_tensor(l, "backbone.model." + std::to_string(2 * i) + ".weight", off, {hidden[i], d});
    off += (long)hidden[i] * d;
    l.b_off[i] = off;
    add_tensor(l, "backbone.model." + std::to_string(2 * i) + ".bias", off, {hidden[i]});
    off += hidden[i];
    d = hidden[i];
  }
  if (tail == TAIL_CRITIC) {
    l.out_dim = 1;
    l.w_off[L] = off; add_tensor(l, "last.weight", off, {1, d}); off += d;
    l.b_off[L] = off; add_tensor(l, "last.bias", off, {1}); off += 1;
  } else if (tail == TAIL_TANH_GAUSS) {
    // head = [mu ; sigma] stored as one (2A x H) matrix so a single GEMM yields both
    l.out_dim = 2 * act_dim;
    l.w_off[L] = off;
    add_tensor(l, "dist_net.mu.weight", off, {act_dim, d});
    add_tensor(l, "dist_net.sigma.weight", off + (long)act_dim * d, {act_dim, d});
    off += 2L * act_dim * d;
    l.b_off[L] = off;
    add_tensor(l, "dist_net.mu.bias", off, {act_dim});
    add_tensor(l, "dist_net.sigma.bias", off + act_dim, {act_dim});
    off += 2 * act_dim;
  } else if (tail == TAIL_GAUSS) {
    l.out_dim = act_dim;
    l.w_off[L] = off; add_tensor(l, "dist_net.mu.weight", off, {act_dim, d}); off += (long)act_dim * d;
    l.b_off[L] = off; add_tensor(l, "dist_net.mu.bias", off, {act_dim}); off += act_dim;
    l.extra_off = off; add_tensor(l, "dist_net.sigma_param", off, {act_dim, 1}); off += act_dim;
  } else {
    l.out_dim = act_dim;
    l.w_off[L] = off; add_tensor(l, "last.weight", off, {act_dim, d}); off += (long)act_dim * d;
    l.b_off[L] = off; add_tensor(l, "last.bias", off, {act_dim}); off += act_dim;
  }
  l.size = off;
  return l;
}

static int build_layouts(const orl_config& c, NetLayout* lay, long* net_off, bool* is_tgt, long* P_train, long* P_tgt) {
  for (int i = 0; i < ORL_NUM_NETS; ++i) { lay[i] = NetLayout(); net_off[i] = 0; is_tgt[i] = false; }
  if (c.n_hidden < 1 || c.n_hidden > ORL_MAX_HIDDEN) return fail("n_hidden must be in [1,4]");
  if (c.obs_dim < 1 || c.act_dim < 1 || c.batch_size < 1 || c.n_runs < 1) return fail("bad dims");
  if (c.algo == ORL_ALGO_CQL) {
    lay[ORL_NET_ACTOR] = make_mlp_layout(c.obs_dim, c.hidden, c.n_hidden, TAIL_TANH_GAUSS, c.act_dim);
    NetLayout cr = make_mlp_layout(c.obs_dim + c.act_dim, c.hidden, c.n_hidden, TAIL_CRITIC, c.act_dim);
    lay[ORL_NET_CRITIC1] = lay[ORL_NET_CRITIC2] = lay[ORL_NET_CRITIC1_OLD] = lay[ORL_NET_CRITIC2_OLD] = cr;
    long o = 0;
    net_off[ORL_NET_ACTOR] = o; o += lay[ORL_NET_ACTOR].size;
    net_off[ORL_NET_CRITIC1] = o; o += cr.size;
    net_off[ORL_NET_CRITIC2] = o; o += cr.size;
    *P_train = o;
    net_off[ORL_NET_CRITIC1_OLD] = 0; net_off[ORL_NET_CRITIC2_OLD] = cr.size;
    is_tgt[ORL_NET_CRITIC1_OLD] = is_tgt[ORL_NET_CRITIC2_OLD] = true;
    *P_tgt = 2 * cr.size;
    return 0;
  }
  return fail("algorithm not built into this engine yet (CQL only)");
}

// ---------------------------------------------------------------------------------------------
// Engine
// ---------------------------------------------------------------------------------------------
Engine::~Engine() {
  if (graph_exec) hipGraphExecDestroy(graph_exec);
  if (graph) hipGraphDestroy(graph);
  for (auto& e : ev_pool) hipEventDestroy(e);
  for (void* p : allocs) hipFree(p);
  if (stream) hipStreamDestroy(stream);
}

float* Engine::raw_alloc(size_t bytes) {
  void* p = nullptr;
  if (bytes == 0) bytes = 16;
  if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
  hipMemsetAsync(p, 0, bytes, stream);
  allocs.push_back(p);
  return (float*)p;
}

Mat Engine::alloc(const std::string& name, long rows, int pitch, int nets) {
  Mat m;
  const long per_net = rows * pitch;
  const long per_run = per_net * nets;
  m.p = raw_alloc(sizeof(float) * per_run * R);
  m.rs = per_run;
  m.cs = per_net;
  m.pitch = pitch;
  ws[name] = m;
  ws_len[name] = per_run;
  return m;
}

float* Engine::net_ptr(int run, int net) const {
  if (net < 0 || net >= ORL_NUM_NETS || !lay[net].present) return nullptr;
  if (net_is_target[net]) return arena + (long)R * P_train + (long)run * P_tgt + net_off[net];
  return arena + (long)run * P_train + net_off[net];
}

NetRef Engine::net_ref(int net) const {
  NetRef r;
  r.base = net_ptr(0, net);
  r.rs = net_is_target[net] ? P_tgt : P_train;
  r.cs = lay[net].size;
  r.lay = &lay[net];
  return r;
}

void Engine::prof_begin(const char* name, double flops) {
  if (!prof_on) return;
  while (ev_pool.size() < ev_used + 2) {
    hipEvent_t e;
    hipEventCreate(&e);
    ev_pool.push_back(e);
  }
  ProfEntry pe;
  pe.name = name;
  pe.flops = flops;
  pe.a = ev_pool[ev_used++];
  pe.b = ev_pool[ev_used++];
  hipEventRecord(pe.a, stream);
  prof.push_back(pe);
}
void Engine::prof_end() {
  if (!prof_on) return;
  hipEventRecord(prof.back().b, stream);
}

template <int PA, int PB, int EPI>
static int run_gemm(Engine* e, int cfg, const GemmP& p, int nz, const char* tag) {
  const double flops = 2.0 * p.M * (double)p.N * p.K * nz;
  e->prof_begin(tag, flops);
  hipError_t err = launch_gemm<PA, PB, EPI>(cfg, p, nz, e->stream);
  e->prof_end();
  if (err != hipSuccess) return fail(std::string("gemm launch ") + tag + ": " + hipGetErrorString(err));
  return 0;
}

int Engine::linear_fwd(const Mat& X, int M, const NetRef& nr, int layer, const Mat& Y, bool relu, int nz1, const char* tag) {
  const NetLayout& l = *nr.lay;
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.A = {X.p, X.rs, X.cs};
  p.a_sr = X.pitch; p.a_sk = 1;
  p.B = {nr.base + l.w_off[layer], nr.rs, nr.cs};
  p.b_sr = l.layer_in(layer); p.b_sk = 1;
  p.C = Y.p; p.c_s0 = Y.rs; p.c_s1 = Y.cs; p.c_sr = Y.pitch;
  p.M = M; p.N = l.layer_out(layer); p.K = l.layer_in(layer);
  p.nz1 = nz1; p.ksplit = 1;
  p.bias = {nr.base + l.b_off[layer], nr.rs, nr.cs};
  const int nz = R * nz1;
  if (relu) return run_gemm<PA_PLAIN, PB_PLAIN, E_BIAS_RELU>(this, CFG_AUTO, p, nz, tag);
  return run_gemm<PA_PLAIN, PB_PLAIN, E_BIAS>(this, CFG_AUTO, p, nz, tag);
}

int Engine::linear_dgrad(const Mat& dY, int M, const NetRef& nr, int layer, int col0, int ncols, const Mat* maskH,
                         const Mat& dX, int nz1, const char* tag, const Mat* rank1_H, const Mat* rowv) {
  const NetLayout& l = *nr.lay;
  GemmP p;
  memset(&p, 0, sizeof(p));
  if (rank1_H) {
    p.A = {rank1_H->p, rank1_H->rs, rank1_H->cs};
    p.a_sr = rank1_H->pitch; p.a_sk = 1; p.a_trans = 0;
    p.rowv = {rowv->p, rowv->rs, rowv->cs};
    p.colv = {nr.base + l.w_off[l.L], nr.rs, nr.cs};
  } else {
    p.A = {dY.p, dY.rs, dY.cs};
    p.a_sr = dY.pitch; p.a_sk = 1;
  }
  p.B = {nr.base + l.w_off[layer] + col0, nr.rs, nr.cs};
  p.b_sr = 1; p.b_sk = l.layer_in(layer);
  p.C = dX.p; p.c_s0 = dX.rs; p.c_s1 = dX.cs; p.c_sr = dX.pitch;
  p.M = M; p.N = ncols; p.K = l.layer_out(layer);
  p.nz1 = nz1; p.ksplit = 1;
  if (maskH) { p.aux = {maskH->p, maskH->rs, maskH->cs}; p.aux_sr = maskH->pitch; }
  const int nz = R * nz1;
  if (rank1_H) {
    if (maskH) return run_gemm<PA_RANK1, PB_PLAIN, E_MASK>(this, CFG_AUTO, p, nz, tag);
    return run_gemm<PA_RANK1, PB_PLAIN, E_PLAIN>(this, CFG_AUTO, p, nz, tag);
  }
  if (maskH) return run_gemm<PA_PLAIN, PB_PLAIN, E_MASK>(this, CFG_AUTO, p, nz, tag);
  return run_gemm<PA_PLAIN, PB_PLAIN, E_PLAIN>(this, CFG_AUTO, p, nz, tag);
}

// split-K factor for a weight gradient: enough workgroups to fill 256 CUs, chunk aligned
static int wgrad_ksplit(int Mout, int Nout, int Krows, int nz, int max_slab) {
  const int cfg = pick_cfg(Mout, Nout, Krows);
  int TM, TN, TK;
  switch (cfg) {
    case CFG_BIG: TM = CfgBig::TM; TN = CfgBig::TN; TK = CfgBig::kTK; break;
    case CFG_MID: TM = CfgMid::TM; TN = CfgMid::TN; TK = CfgMid::kTK; break;
    case CFG_SMALL: TM = CfgSmall::TM; TN = CfgSmall::TN; TK = CfgSmall::kTK; break;
    default: TM = CfgTall::TM; TN = CfgTall::TN; TK = CfgTall::kTK; break;
  }
  const int tiles = ((Mout + TM - 1) / TM) * ((Nout + TN - 1) / TN) * nz;
  const int kchunks = (Krows + TK - 1) / TK;
  int ks = (512 + tiles - 1) / tiles;
  ks = std::max(1, std::min(ks, std::min(max_slab, kchunks)));
  // at least 2 chunks per split
  while (ks > 1 && (kchunks + ks - 1) / ks < 2) --ks;
  return ks;
}

int Engine::linear_wgrad(const Mat& dY, const Mat& X, int M, const NetRef& nr, int layer, long g_net_off, int ksplit,
                         int nz1, const char* tag, const Mat* rank1_H, const Mat* rowv) {
  const NetLayout& l = *nr.lay;
  GemmP p;
  memset(&p, 0, sizeof(p));
  if (rank1_H) {
    p.A = {rank1_H->p, rank1_H->rs, rank1_H->cs};
    p.a_sr = 1; p.a_sk = rank1_H->pitch; p.a_trans = 1;
    p.rowv = {rowv->p, rowv->rs, rowv->cs};
    p.colv = {nr.base + l.w_off[l.L], nr.rs, nr.cs};
  } else {
    p.A = {dY.p, dY.rs, dY.cs};
    p.a_sr = 1; p.a_sk = dY.pitch;
  }
  p.B = {X.p, X.rs, X.cs};
  p.b_sr = 1; p.b_sk = X.pitch;
  p.ones_row = l.layer_in(layer);
  p.M = l.layer_out(layer); p.N = l.layer_in(layer) + 1; p.K = M;
  p.nz1 = nz1; p.ksplit = ksplit;
  const long g_rs = (long)max_slab * P_train;
  p.C = grads + g_net_off + l.w_off[layer];
  p.c_s0 = g_rs; p.c_s1 = nr.cs; p.c_sr = l.layer_in(layer); p.c_ks = P_train;
  p.bias_out = grads + g_net_off + l.b_off[layer];
  p.bo_s0 = g_rs; p.bo_s1 = nr.cs; p.bo_ks = P_train;
  const int nz = R * nz1;
  if (rank1_H) return run_gemm<PA_RANK1, PB_ONES, E_WGRAD>(this, CFG_AUTO, p, nz, tag);
  return run_gemm<PA_PLAIN, PB_ONES, E_WGRAD>(this, CFG_AUTO, p, nz, tag);
}

int Engine::adam(int net, int nnets, int lr_slot, const std::vector<std::pair<long, int>>& segs, bool polyak,
                 int target_net, unsigned long long t_div) {
  AdamP a;
  memset(&a, 0, sizeof(a));
  const NetLayout& l = lay[net];
  a.params = net_ptr(0, net); a.p_s0 = P_train; a.p_s1 = l.size;
  a.m = adam_m + net_off[net]; a.v = adam_v + net_off[net];
  a.g = grads + net_off[net]; a.g_s0 = (long)max_slab * P_train; a.g_s1 = l.size; a.g_ks = P_train;
  a.nseg = (int)segs.size();
  if (a.nseg > 8) return fail("too many adam segments");
  for (int i = 0; i < a.nseg; ++i) { a.seg_end[i] = segs[i].first; a.seg_nslab[i] = segs[i].second; }
  if (polyak) { a.target = net_ptr(0, target_net); a.t_s0 = P_tgt; a.t_s1 = l.size; }
  a.P = l.size; a.lr_slot = lr_slot; a.hy = hyper;
  a.b1 = cfg.adam_beta1; a.b2 = cfg.adam_beta2; a.eps = cfg.adam_eps; a.tau = cfg.tau;
  a.gstep = gstep; a.t_div = t_div;
  dim3 grid((unsigned)((l.size + 255) / 256), nnets, R);
  prof_begin("adam", 0);
  hipLaunchKernelGGL(k_adam, grid, dim3(256), 0, stream, a);
  prof_end();
  if (hipGetLastError() != hipSuccess) return fail("adam launch failed");
  return 0;
}

// generic backward through an MLP family.  dTail: [M x out_dim] gradient w.r.t. the tail output.
struct BwdOut { std::vector<std::pair<long, int>> segs; };
static int mlp_backward(Engine* e, const NetRef& nr, const Mat& X, const std::vector<Mat>& hs, int M, int nz1,
                        const Mat& dTail, std::vector<Mat>& dz, bool want_w, long g_net_off, bool want_dx, int dx_col0,
                        int dx_ncols, const Mat* dX, const char* tag, BwdOut* out) {
  const NetLayout& l = *nr.lay;
  const int L = l.L;
  const bool rank1 = (l.out_dim == 1);
  const int nz = e->R * nz1;
  std::vector<int> ks(L + 1, 1);
  std::string t = tag;
  if (want_w) {
    for (int i = 0; i <= L; ++i) ks[i] = wgrad_ksplit(l.layer_out(i), l.layer_in(i) + 1, M, nz, e->max_slab);
    if (e->linear_wgrad(dTail, hs[L - 1], M, nr, L, g_net_off, ks[L], nz1, (t + ".wgrad_tail").c_str())) return -1;
  }
  const Mat* curH = nullptr;   // rank-1 virtual dz
  Mat cur;
  if (rank1) curH = &hs[L - 1];
  else {
    if (e->linear_dgrad(dTail, M, nr, L, 0, l.layer_in(L), &hs[L - 1], dz[L - 1], nz1, (t + ".dgrad_tail").c_str())) return -1;
    cur = dz[L - 1];
  }
  for (int i = L - 1; i >= 0; --i) {
    const Mat& xin = (i == 0) ? X : hs[i - 1];
    if (want_w) {
      if (e->linear_wgrad(cur, xin, M, nr, i, g_net_off, ks[i], nz1, (t + ".wgrad" + std::to_string(i)).c_str(), curH, &dTail)) return -1;
    }
    if (i > 0) {
      if (e->linear_dgrad(cur, M, nr, i, 0, l.layer_in(i), &hs[i - 1], dz[i - 1], nz1, (t + ".dgrad" + std::to_string(i)).c_str(), curH, &dTail)) return -1;
      cur = dz[i - 1];
      curH = nullptr;
    } else if (want_dx) {
      if (e->linear_dgrad(cur, M, nr, 0, dx_col0, dx_ncols, nullptr, *dX, nz1, (t + ".dgrad_x").c_str(), curH, &dTail)) return -1;
    }
  }
  if (out) {
    out->segs.clear();
    for (int i = 0; i <= L; ++i) out->segs.push_back({l.b_off[i] + l.layer_out(i), ks[i]});
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------
// init
// ---------------------------------------------------------------------------------------------
int Engine::init(const orl_config& c) {
  cfg = c;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("no HIP device available: the update engine needs an MI355X (gfx950)");
  if (c.device < 0 || c.device >= ndev) return fail("bad device ordinal");
  dev = c.device;
  ORL_HIP(hipSetDevice(dev));
  if (c.precision != 0) return fail("precision=1 (split-bf16 MFMA) is not built yet; use precision=0 (fp32 MFMA)");
  if (build_layouts(c, lay, net_off, net_is_target, &P_train, &P_tgt)) return -1;
  ORL_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  R = c.n_runs; B = c.batch_size; od = c.obs_dim; ad = c.act_dim;
  N = c.num_repeat_actions > 0 ? c.num_repeat_actions : 1;
  OP = rup(od, 4); AP = rup(ad, 4); XP = rup(od + ad, 4); L = c.n_hidden; Hlast = c.hidden[L - 1];
  const long arena_floats = (long)R * (P_train + P_tgt);
  if (c.external_arena) { arena = c.external_arena; arena_owned = false; }
  else { arena = raw_alloc(sizeof(float) * arena_floats); arena_owned = true; if (!arena) return fail("hipMalloc arena"); }
  adam_m = raw_alloc(sizeof(float) * R * P_train);
  adam_v = raw_alloc(sizeof(float) * R * P_train);
  grads = raw_alloc(sizeof(float) * (size_t)R * max_slab * P_train);
  scalars = (RunScalars*)raw_alloc(sizeof(RunScalars) * R);
  hyper = (Hyper*)raw_alloc(sizeof(Hyper));
  gstep = (unsigned long long*)raw_alloc(sizeof(unsigned long long));
  if (!adam_m || !adam_v || !grads || !scalars || !hyper || !gstep) return fail("hipMalloc state");
  memset(&hyper_host, 0, sizeof(hyper_host));
  hyper_host.lr[ORL_OPT_ACTOR] = c.actor_lr;
  hyper_host.lr[ORL_OPT_CRITIC] = c.critic_lr;
  hyper_host.lr[ORL_OPT_ALPHA] = c.alpha_lr;
  hyper_host.lr[ORL_OPT_CQL_ALPHA] = c.cql_alpha_lr;
  hyper_host.lr[ORL_OPT_CRITIC_V] = c.critic_v_lr;
  ORL_HIP(hipMemcpyAsync(hyper, &hyper_host, sizeof(Hyper), hipMemcpyHostToDevice, stream));
  std::vector<RunScalars> sc(R);
  for (auto& s : sc) { memset(&s, 0, sizeof(s)); s.alpha = c.auto_alpha ? 1.0f : c.alpha; s.alpha_bwd = s.alpha; s.cons_scale = 1.f; }
  ORL_HIP(hipMemcpyAsync(scalars, sc.data(), sizeof(RunScalars) * R, hipMemcpyHostToDevice, stream));
  ORL_HIP(hipStreamSynchronize(stream));
  if (c.algo == ORL_ALGO_CQL) { if (cql_build()) return -1; }
  metrics_last = raw_alloc(sizeof(float) * R * nm);
  metrics_sum = raw_alloc(sizeof(float) * R * nm);
  d_idx = (long long*)raw_alloc(sizeof(long long) * R * B);
  ORL_HIP(hipStreamSynchronize(stream));
  return 0;
}

// ---------------------------------------------------------------------------------------------
// CQL (cql.py:87-207; oracle/cql.py)
// ---------------------------------------------------------------------------------------------
int Engine::cql_build() {
  const int A = ad, BN = B * N, Mc = B + 3 * BN, Bt = cfg.max_q_backup ? BN : B;
  metric_names = {"loss/actor", "loss/critic1", "loss/critic2"};
  if (cfg.auto_alpha) { metric_names.push_back("loss/alpha"); metric_names.push_back("alpha"); }
  if (cfg.with_lagrange) { metric_names.push_back("loss/cql_alpha"); metric_names.push_back("cql_alpha"); }
  nm = (int)metric_names.size();
  // batch slots: obs and next_obs adjacent so [obs; next_obs] is one 2B-row matrix
  alloc("b_obs2", 2 * B, OP);
  alloc("b_act", B, AP); alloc("b_rew", B, 1); alloc("b_term", B, 1);
  alloc("n_eps_actor", B, A); alloc("n_eps_next", Bt, A); alloc("n_urand", BN, A);
  alloc("n_eps_pi", BN, A); alloc("n_eps_npi", BN, A);
  for (int i = 0; i < L; ++i) {
    const int h = cfg.hidden[i];
    alloc("ah" + std::to_string(i), B, h);
    alloc("dah" + std::to_string(i), B, h);
    alloc("ah2_" + std::to_string(i), 2 * B, h);
    alloc("ca" + std::to_string(i), B, h, 2);
    alloc("dca" + std::to_string(i), B, h, 2);
    alloc("ct" + std::to_string(i), Bt, h, 2);
    alloc("ch" + std::to_string(i), Mc, h, 2);
    if (i < L - 1) alloc("dch" + std::to_string(i), Mc, h, 2);
  }
  alloc("head", B, 2 * A); alloc("head2", 2 * B, 2 * A); alloc("dhead", B, 2 * A);
  alloc("xa", B, XP); alloc("xt", Bt, XP); alloc("xc", Mc, XP);
  alloc("logp_a", B, 1); alloc("logp_next", Bt, 1); alloc("logp_pi", BN, 1); alloc("logp_npi", BN, 1);
  alloc("qa", B, 1, 2); alloc("dqa", B, 1, 2); alloc("dxa", B, A, 2);
  alloc("qt", Bt, 1, 2); alloc("q", Mc, 1, 2); alloc("dq", Mc, 1, 2); alloc("target_q", B, 1);
  taps["q1"] = {ws["q"].net(0), B, 1};
  taps["q2"] = {ws["q"].net(1), B, 1};
  taps["q1_all"] = {ws["q"].net(0), Mc, 1};
  taps["q2_all"] = {ws["q"].net(1), Mc, 1};
  taps["q1a"] = {ws["qa"].net(0), B, 1};
  taps["q2a"] = {ws["qa"].net(1), B, 1};
  taps["logp_a"] = {ws["logp_a"], B, 1};
  taps["target_q"] = {ws["target_q"], B, 1};
  taps["xc"] = {ws["xc"], Mc, XP};
  taps["dq1"] = {ws["dq"].net(0), Mc, 1};
  return 0;
}

static int launch_assemble(Engine* e, const Mat& obs, int OP, int od, const Mat* act, int ad, const Mat& X, int XP, int row0, int rows, int rep) {
  AssembleP a;
  memset(&a, 0, sizeof(a));
  a.obs = obs.p; a.obs_rs = obs.rs; a.OP = OP; a.od = od;
  if (act) { a.act = act->p; a.act_rs = act->rs; a.apitch = act->pitch; }
  a.ad = ad;
  a.X = X.p; a.x_rs = X.rs; a.XP = XP; a.row0 = row0; a.rows = rows; a.rep = rep;
  e->prof_begin("assemble", 0);
  hipLaunchKernelGGL(k_assemble, dim3((rows + 255) / 256, e->R), dim3(256), 0, e->stream, a);
  e->prof_end();
  return hipGetLastError() == hipSuccess ? 0 : fail("assemble launch");
}

int Engine::cql_step() {
  const int A = ad, BN = B * N, Mc = B + 3 * BN, Bt = cfg.max_q_backup ? BN : B;
  auto W = [&](const std::string& n) -> Mat& { return ws[n]; };
  const NetRef actor = net_ref(ORL_NET_ACTOR), crit = net_ref(ORL_NET_CRITIC1), tgt = net_ref(ORL_NET_CRITIC1_OLD);
  Mat obs = W("b_obs2"), nobs = W("b_obs2").rows(B), obs2 = W("b_obs2");
  std::vector<Mat> ah, dah, ah2, ca, dca, ct, ch, dch;
  for (int i = 0; i < L; ++i) {
    const std::string s = std::to_string(i);
    ah.push_back(W("ah" + s)); dah.push_back(W("dah" + s)); ah2.push_back(W("ah2_" + s));
    ca.push_back(W("ca" + s)); dca.push_back(W("dca" + s)); ct.push_back(W("ct" + s)); ch.push_back(W("ch" + s));
    if (i < L - 1) dch.push_back(W("dch" + s)); else dch.push_back(Mat());
  }
  Mat xa = W("xa"), xt = W("xt"), xc = W("xc");
  xa.cs = 0; xt.cs = 0; xc.cs = 0;   // shared by both critics

  // ---------------- phase A: actor update (cql.py:92-106) ----------------
  for (int i = 0; i < L; ++i)
    if (linear_fwd(i == 0 ? obs : ah[i - 1], B, actor, i, ah[i], true, 1, "actor.fwd")) return -1;
  if (linear_fwd(ah[L - 1], B, actor, L, W("head"), false, 1, "actor.head")) return -1;
  if (launch_assemble(this, obs, OP, od, nullptr, ad, xa, XP, 0, B, 1)) return -1;
  {
    SampleP sp; memset(&sp, 0, sizeof(sp));
    sp.head = W("head").p; sp.head_rs = W("head").rs; sp.A = A;
    SampleJob& j = sp.job[0];
    j.head_row0 = 0; j.rows = B; j.rep = 1; j.eps = W("n_eps_actor").p; j.eps_rs = W("n_eps_actor").rs;
    j.dst = xa.p; j.dst_rs = xa.rs; j.dst_pitch = XP; j.dst_col = od; j.dst_row0 = 0;
    j.logp = W("logp_a").p; j.logp_rs = W("logp_a").rs;
    prof_begin("tanh_sample", 0);
    hipLaunchKernelGGL(k_tanh_sample, dim3((B + 255) / 256, 1, R), dim3(256), 0, stream, sp);
    prof_end();
  }
  for (int i = 0; i < L; ++i)
    if (linear_fwd(i == 0 ? xa : ca[i - 1], B, crit, i, ca[i], true, 2, "critic.fwd_a")) return -1;
  if (linear_fwd(ca[L - 1], B, crit, L, W("qa"), false, 2, "critic.head_a")) return -1;
  {
    ActorLossP p; memset(&p, 0, sizeof(p));
    p.qa = W("qa").p; p.qa_rs = W("qa").rs; p.qa_cs = W("qa").cs; p.dqa = W("dqa").p;
    p.logp = W("logp_a").p; p.logp_rs = W("logp_a").rs; p.B = B; p.K = 2;
    p.sc = scalars; p.hy = hyper; p.auto_alpha = cfg.auto_alpha; p.fixed_alpha = cfg.alpha;
    p.target_entropy = cfg.target_entropy; p.clamp_alpha01 = 0;
    p.b1 = cfg.adam_beta1; p.b2 = cfg.adam_beta2; p.eps = cfg.adam_eps; p.gstep = gstep;
    p.metrics_last = metrics_last; p.metrics_sum = metrics_sum; p.nm = nm;
    p.m_actor = 0; p.m_alpha_loss = 3; p.m_alpha = 4;
    prof_begin("actor_loss", 0);
    hipLaunchKernelGGL(k_actor_loss, dim3(R), dim3(256), 0, stream, p);
    prof_end();
  }
  {
    Mat dxa = W("dxa");
    if (mlp_backward(this, crit, xa, ca, B, 2, W("dqa"), dca, false, 0, true, od, A, &dxa, "critic.bwd_a", nullptr)) return -1;
  }
  {
    HeadBwdP p; memset(&p, 0, sizeof(p));
    p.dxa = W("dxa").p; p.dxa_rs = W("dxa").rs; p.dxa_cs = W("dxa").cs; p.dxa_pitch = A; p.K = 2;
    p.head = W("head").p; p.head_rs = W("head").rs; p.eps = W("n_eps_actor").p; p.eps_rs = W("n_eps_actor").rs;
    p.xa = xa.p; p.xa_rs = xa.rs; p.XP = XP; p.od = od; p.dhead = W("dhead").p; p.dhead_rs = W("dhead").rs;
    p.sc = scalars; p.auto_alpha = cfg.auto_alpha; p.fixed_alpha = cfg.alpha; p.B = B; p.A = A;
    prof_begin("head_bwd", 0);
    hipLaunchKernelGGL(k_head_bwd, dim3((B + 255) / 256, R), dim3(256), 0, stream, p);
    prof_end();
  }
  BwdOut bo;
  if (mlp_backward(this, actor, obs, ah, B, 1, W("dhead"), dah, true, net_off[ORL_NET_ACTOR], false, 0, 0, nullptr, "actor.bwd", &bo)) return -1;
  if (adam(ORL_NET_ACTOR, 1, ORL_OPT_ACTOR, bo.segs, false, -1)) return -1;

  // ---------------- phase T: targets + repeated-action sampling with the UPDATED actor ----------------
  for (int i = 0; i < L; ++i)
    if (linear_fwd(i == 0 ? obs2 : ah2[i - 1], 2 * B, actor, i, ah2[i], true, 1, "actor.fwd2")) return -1;
  if (linear_fwd(ah2[L - 1], 2 * B, actor, L, W("head2"), false, 1, "actor.head2")) return -1;
  // critic input rows: [0,B) (obs, a_data) ; [B,B+BN) (obs rep, a_pi) ; next BN (obs rep, a_next_pi) ; last BN (obs rep, u_rand)
  {
    Mat act = W("b_act");
    if (launch_assemble(this, obs, OP, od, &act, ad, xc, XP, 0, B, 1)) return -1;
    if (launch_assemble(this, obs, OP, od, nullptr, ad, xc, XP, B, BN, N)) return -1;
    if (launch_assemble(this, obs, OP, od, nullptr, ad, xc, XP, B + BN, BN, N)) return -1;
    Mat ur = W("n_urand");
    if (launch_assemble(this, obs, OP, od, &ur, ad, xc, XP, B + 2 * BN, BN, N)) return -1;
    if (launch_assemble(this, nobs, OP, od, nullptr, ad, xt, XP, 0, Bt, cfg.max_q_backup ? N : 1)) return -1;
  }
  {
    SampleP sp; memset(&sp, 0, sizeof(sp));
    sp.head = W("head2").p; sp.head_rs = W("head2").rs; sp.A = A;
    SampleJob& j0 = sp.job[0];   // next actions for the TD target (cql.py:108-130)
    j0.head_row0 = B; j0.rows = Bt; j0.rep = cfg.max_q_backup ? N : 1; j0.eps = W("n_eps_next").p; j0.eps_rs = W("n_eps_next").rs;
    j0.dst = xt.p; j0.dst_rs = xt.rs; j0.dst_pitch = XP; j0.dst_col = od; j0.dst_row0 = 0;
    j0.logp = W("logp_next").p; j0.logp_rs = W("logp_next").rs;
    SampleJob& j1 = sp.job[1];   // a ~ pi(tmp_obss)  (cql.py:149)
    j1.head_row0 = 0; j1.rows = BN; j1.rep = N; j1.eps = W("n_eps_pi").p; j1.eps_rs = W("n_eps_pi").rs;
    j1.dst = xc.p; j1.dst_rs = xc.rs; j1.dst_pitch = XP; j1.dst_col = od; j1.dst_row0 = B;
    j1.logp = W("logp_pi").p; j1.logp_rs = W("logp_pi").rs;
    SampleJob& j2 = sp.job[2];   // a ~ pi(tmp_next_obss), evaluated at tmp_obss (cql.py:150)
    j2.head_row0 = B; j2.rows = BN; j2.rep = N; j2.eps = W("n_eps_npi").p; j2.eps_rs = W("n_eps_npi").rs;
    j2.dst = xc.p; j2.dst_rs = xc.rs; j2.dst_pitch = XP; j2.dst_col = od; j2.dst_row0 = B + BN;
    j2.logp = W("logp_npi").p; j2.logp_rs = W("logp_npi").rs;
    prof_begin("tanh_sample", 0);
    hipLaunchKernelGGL(k_tanh_sample, dim3((BN + 255) / 256, 3, R), dim3(256), 0, stream, sp);
    prof_end();
  }
  for (int i = 0; i < L; ++i)
    if (linear_fwd(i == 0 ? xt : ct[i - 1], Bt, tgt, i, ct[i], true, 2, "target.fwd")) return -1;
  if (linear_fwd(ct[L - 1], Bt, tgt, L, W("qt"), false, 2, "target.head")) return -1;

  // ---------------- phase C: critics (cql.py:132-190) ----------------
  for (int i = 0; i < L; ++i)
    if (linear_fwd(i == 0 ? xc : ch[i - 1], Mc, crit, i, ch[i], true, 2, "critic.fwd")) return -1;
  if (linear_fwd(ch[L - 1], Mc, crit, L, W("q"), false, 2, "critic.head")) return -1;
  {
    CqlLossP p; memset(&p, 0, sizeof(p));
    p.q = W("q").p; p.q_rs = W("q").rs; p.q_cs = W("q").cs; p.dq = W("dq").p;
    p.qt = W("qt").p; p.qt_rs = W("qt").rs; p.qt_cs = W("qt").cs;
    p.rew = W("b_rew").p; p.term = W("b_term").p; p.bt_rs = W("b_rew").rs;
    p.logp_next = W("logp_next").p; p.lpn_rs = W("logp_next").rs;
    p.logp_pi = W("logp_pi").p; p.logp_npi = W("logp_npi").p; p.lpp_rs = W("logp_pi").rs;
    p.target_q = W("target_q").p; p.tq_rs = W("target_q").rs;
    p.B = B; p.N = N; p.A = A; p.gamma = cfg.gamma; p.w = cfg.cql_weight; p.T = cfg.temperature; p.thr = cfg.lagrange_threshold;
    p.max_q_backup = cfg.max_q_backup; p.det_backup = cfg.deterministic_backup; p.with_lagrange = cfg.with_lagrange;
    p.auto_alpha = cfg.auto_alpha; p.fixed_alpha = cfg.alpha;
    p.sc = scalars; p.hy = hyper; p.b1 = cfg.adam_beta1; p.b2 = cfg.adam_beta2; p.eps = cfg.adam_eps; p.gstep = gstep;
    p.metrics_last = metrics_last; p.metrics_sum = metrics_sum; p.nm = nm;
    p.m_c1 = 1; p.m_c2 = 2; p.m_cqla_loss = cfg.auto_alpha ? 5 : 3; p.m_cqla = cfg.auto_alpha ? 6 : 4;
    prof_begin("cql_loss", 0);
    hipLaunchKernelGGL(k_cql_loss, dim3(R), dim3(256), 0, stream, p);
    prof_end();
  }
  BwdOut bc;
  if (mlp_backward(this, crit, xc, ch, Mc, 2, W("dq"), dch, true, net_off[ORL_NET_CRITIC1], false, 0, 0, nullptr, "critic.bwd", &bc)) return -1;
  if (adam(ORL_NET_CRITIC1, 2, ORL_OPT_CRITIC, bc.segs, true, ORL_NET_CRITIC1_OLD)) return -1;
  return 0;
}

// ---------------------------------------------------------------------------------------------
// sampling / noise / step driver
// ---------------------------------------------------------------------------------------------
int Engine::enqueue_sample(const long long* idx_dev) {
  if (!d_obs) return fail("no dataset loaded (orl_buffer_load)");
  GatherP g; memset(&g, 0, sizeof(g));
  g.obs = d_obs; g.nobs = d_nobs; g.act = d_act; g.rew = d_rew; g.term = d_term; g.n = n_data;
  g.OP = OP; g.AP = AP; g.od = od; g.ad = ad; g.B = B; g.idx = idx_dev;
  Mat o2 = ws["b_obs2"];
  g.b_obs = o2.p; g.b_nobs = o2.p + (long)B * OP;
  g.b_act = ws["b_act"].p; g.b_rew = ws["b_rew"].p; g.b_term = ws["b_term"].p;
  g.seed = cfg.seed; g.gstep = gstep;
  // batch slot run strides: b_obs2 has 2B rows per run -> the kernel indexes dst = r*B+row, so launch per run
  // with explicit run offsets instead (keeps the gather kernel simple)
  for (int r = 0; r < R; ++r) {
    GatherP gr = g;
    gr.b_obs = o2.p + r * o2.rs; gr.b_nobs = gr.b_obs + (long)B * OP;
    gr.b_act = ws["b_act"].p + r * ws["b_act"].rs; gr.b_rew = ws["b_rew"].p + r * ws["b_rew"].rs;
    gr.b_term = ws["b_term"].p + r * ws["b_term"].rs;
    gr.idx = idx_dev ? idx_dev + (long)r * B : nullptr;
    gr.seed = cfg.seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(r + 1);
    prof_begin("gather", 0);
    hipLaunchKernelGGL(k_gather, dim3((B + 63) / 64, 1), dim3(256), 0, stream, gr);
    prof_end();
  }
  return hipGetLastError() == hipSuccess ? 0 : fail("gather launch");
}

int Engine::enqueue_noise() {
  if (cfg.algo != ORL_ALGO_CQL) return fail("noise: algorithm not built");
  struct { const char* name; int kind; } slots[] = {{"n_eps_actor", 0}, {"n_eps_next", 0}, {"n_urand", 1}, {"n_eps_pi", 0}, {"n_eps_npi", 0}};
  uint32_t sid = 1;
  for (auto& s : slots) {
    const long n = ws_len[s.name];
    prof_begin("noise", 0);
    hipLaunchKernelGGL(k_noise, dim3((unsigned)((n / 4 + 256) / 256), R), dim3(256), 0, stream, ws[s.name].p, n, s.kind,
                       cfg.act_low, cfg.act_high, cfg.seed, gstep, sid++);
    prof_end();
  }
  return hipGetLastError() == hipSuccess ? 0 : fail("noise launch");
}

int Engine::enqueue_step() {
  int rc = -1;
  if (cfg.algo == ORL_ALGO_CQL) rc = cql_step();
  else return fail("algorithm not built");
  if (rc) return rc;
  hipLaunchKernelGGL(k_tick, dim3(1), dim3(1), 0, stream, gstep);
  return hipGetLastError() == hipSuccess ? 0 : fail("tick launch");
}

}  // namespace orl

// =================================================================================================
// C ABI
// =================================================================================================
using namespace orl;

struct orl_engine {
  Engine e;
};

extern "C" {

const char* orl_last_error(void) { return g_err.c_str(); }
const char* orl_version(void) { return "orl-engine 0.1 (gfx950, fp32 MFMA)"; }

void orl_config_default(orl_config* c, int32_t algo) {
  memset(c, 0, sizeof(*c));
  c->algo = algo;
  c->obs_dim = 17; c->act_dim = 6;
  c->n_hidden = 2; c->hidden[0] = c->hidden[1] = 256;
  c->batch_size = 256; c->n_runs = 1; c->device = 0; c->precision = 0; c->seed = 0;
  c->gamma = 0.99f; c->tau = 0.005f;
  c->adam_beta1 = 0.9f; c->adam_beta2 = 0.999f; c->adam_eps = 1e-8f;
  c->auto_alpha = 1; c->alpha = 0.2f; c->target_entropy = -6.0f; c->alpha_lr = 1e-4f;
  c->actor_lr = 1e-4f; c->critic_lr = 3e-4f;
  c->cql_weight = 5.0f; c->temperature = 1.0f; c->max_q_backup = 0; c->deterministic_backup = 1; c->with_lagrange = 0;
  c->lagrange_threshold = 10.0f; c->cql_alpha_lr = 3e-4f; c->num_repeat_actions = 10; c->act_low = -1.0f; c->act_high = 1.0f;
  c->expectile = 0.7f; c->iql_temperature = 3.0f; c->critic_v_lr = 3e-4f;
  c->policy_noise = 0.2f; c->noise_clip = 0.5f; c->td3bc_alpha = 2.5f; c->max_action = 1.0f; c->update_actor_freq = 2;
  c->num_critics = 10; c->eta = 1.0f;
  if (algo == ORL_ALGO_IQL || algo == ORL_ALGO_TD3BC) { c->actor_lr = 3e-4f; }
  if (algo == ORL_ALGO_EDAC) { c->n_hidden = 3; c->hidden[2] = 256; c->deterministic_backup = 0; }
}

int64_t orl_arena_floats(const orl_config* cfg) {
  NetLayout lay[ORL_NUM_NETS];
  long off[ORL_NUM_NETS], pt = 0, pg = 0;
  bool tg[ORL_NUM_NETS];
  if (build_layouts(*cfg, lay, off, tg, &pt, &pg)) return -1;
  return (int64_t)cfg->n_runs * (pt + pg);
}

int orl_engine_create(const orl_config* cfg, orl_engine** out) {
  if (!cfg || !out) return fail("null argument");
  orl_engine* h = new orl_engine();
  if (h->e.init(*cfg)) { delete h; *out = nullptr; return -1; }
  *out = h;
  return 0;
}

void orl_engine_destroy(orl_engine* h) { if (h) { hipSetDevice(h->e.dev); hipStreamSynchronize(h->e.stream); delete h; } }

int orl_engine_sync(orl_engine* h) { ORL_HIP(hipStreamSynchronize(h->e.stream)); return 0; }

int orl_net_present(orl_engine* h, int net) { return net >= 0 && net < ORL_NUM_NETS && h->e.lay[net].present; }
int64_t orl_net_floats(orl_engine* h, int net) { return orl_net_present(h, net) ? h->e.lay[net].size : -1; }
int orl_net_num_tensors(orl_engine* h, int net) { return orl_net_present(h, net) ? (int)h->e.lay[net].tensors.size() : -1; }
int orl_net_tensor(orl_engine* h, int net, int idx, char* name, int name_cap, int64_t* offset, int32_t* ndim, int64_t shape[4]) {
  if (!orl_net_present(h, net)) return fail("net not present");
  const auto& ts = h->e.lay[net].tensors;
  if (idx < 0 || idx >= (int)ts.size()) return fail("tensor index out of range");
  snprintf(name, name_cap, "%s", ts[idx].name.c_str());
  *offset = ts[idx].off; *ndim = ts[idx].ndim;
  for (int i = 0; i < 4; ++i) shape[i] = i < ts[idx].ndim ? ts[idx].shape[i] : 1;
  return 0;
}
float* orl_net_ptr(orl_engine* h, int run, int net) {
  if (run < 0 || run >= h->e.R) return nullptr;
  return h->e.net_ptr(run, net);
}
int orl_net_set(orl_engine* h, int run, int net, const float* host, int64_t n) {
  float* d = orl_net_ptr(h, run, net);
  if (!d || n != h->e.lay[net].size) return fail("orl_net_set: bad net/run/size");
  ORL_HIP(hipMemcpyAsync(d, host, sizeof(float) * n, hipMemcpyHostToDevice, h->e.stream));
  ORL_HIP(hipStreamSynchronize(h->e.stream));
  return 0;
}
int orl_net_get(orl_engine* h, int run, int net, float* host, int64_t n) {
  float* d = orl_net_ptr(h, run, net);
  if (!d || n != h->e.lay[net].size) return fail("orl_net_get: bad net/run/size");
  ORL_HIP(hipStreamSynchronize(h->e.stream));
  ORL_HIP(hipMemcpy(host, d, sizeof(float) * n, hipMemcpyDeviceToHost));
  return 0;
}
int orl_scalar_set(orl_engine* h, int run, int which, float v) {
  Engine& e = h->e;
  if (run < 0 || run >= e.R) return fail("bad run");
  RunScalars s;
  ORL_HIP(hipStreamSynchronize(e.stream));
  ORL_HIP(hipMemcpy(&s, e.scalars + run, sizeof(s), hipMemcpyDeviceToHost));
  if (which == ORL_SCALAR_LOG_ALPHA) { s.log_alpha = v; if (e.cfg.auto_alpha) { s.alpha = expf(v); if (e.cfg.algo == ORL_ALGO_EDAC) s.alpha = fminf(fmaxf(s.alpha, 0.f), 1.f); } }
  else if (which == ORL_SCALAR_CQL_LOG_ALPHA) s.cql_log_alpha = v;
  else return fail("scalar not settable");
  ORL_HIP(hipMemcpy(e.scalars + run, &s, sizeof(s), hipMemcpyHostToDevice));
  return 0;
}
int orl_scalar_get(orl_engine* h, int run, int which, float* v) {
  Engine& e = h->e;
  if (run < 0 || run >= e.R) return fail("bad run");
  RunScalars s;
  ORL_HIP(hipStreamSynchronize(e.stream));
  ORL_HIP(hipMemcpy(&s, e.scalars + run, sizeof(s), hipMemcpyDeviceToHost));
  if (which == ORL_SCALAR_LOG_ALPHA) *v = s.log_alpha;
  else if (which == ORL_SCALAR_CQL_LOG_ALPHA) *v = s.cql_log_alpha;
  else if (which == ORL_SCALAR_ALPHA) *v = e.cfg.auto_alpha ? s.alpha : e.cfg.alpha;
  else return fail("unknown scalar");
  return 0;
}
int orl_set_lr(orl_engine* h, int opt, float lr) {
  if (opt < 0 || opt >= 8) return fail("bad optimizer id");
  h->e.hyper_host.lr[opt] = lr;
  ORL_HIP(hipMemcpyAsync(h->e.hyper, &h->e.hyper_host, sizeof(Hyper), hipMemcpyHostToDevice, h->e.stream));
  ORL_HIP(hipStreamSynchronize(h->e.stream));
  return 0;
}
int orl_reset_optimizers(orl_engine* h) {
  Engine& e = h->e;
  ORL_HIP(hipMemsetAsync(e.adam_m, 0, sizeof(float) * e.R * e.P_train, e.stream));
  ORL_HIP(hipMemsetAsync(e.adam_v, 0, sizeof(float) * e.R * e.P_train, e.stream));
  ORL_HIP(hipMemsetAsync(e.gstep, 0, sizeof(unsigned long long), e.stream));
  e.step_host = 0;
  ORL_HIP(hipStreamSynchronize(e.stream));
  return 0;
}

// ---- replay buffer ----
static int upload_padded(Engine& e, float** dst, const float* src, long n, int dim, int pitch) {
  *dst = e.raw_alloc(sizeof(float) * n * pitch);
  if (!*dst) return fail("hipMalloc dataset");
  ORL_HIP(hipMemcpy2DAsync(*dst, sizeof(float) * pitch, src, sizeof(float) * dim, sizeof(float) * dim, n, hipMemcpyHostToDevice, e.stream));
  return 0;
}
int orl_buffer_load(orl_engine* h, const float* obs, const float* act, const float* next_obs, const float* rew, const float* term, int64_t n) {
  Engine& e = h->e;
  if (n <= 0) return fail("empty dataset");
  if (e.d_obs) return fail("dataset already loaded");
  if (upload_padded(e, &e.d_obs, obs, n, e.od, e.OP)) return -1;
  if (upload_padded(e, &e.d_nobs, next_obs, n, e.od, e.OP)) return -1;
  if (upload_padded(e, &e.d_act, act, n, e.ad, e.AP)) return -1;
  if (upload_padded(e, &e.d_rew, rew, n, 1, 1)) return -1;
  if (upload_padded(e, &e.d_term, term, n, 1, 1)) return -1;
  e.n_data = n;
  ORL_HIP(hipStreamSynchronize(e.stream));
  return 0;
}
int64_t orl_buffer_size(orl_engine* h) { return h->e.n_data; }
int orl_buffer_normalize_obs(orl_engine*, float, float*, float*) { return fail("normalize_obs: not built yet"); }
int orl_buffer_sample(orl_engine* h, const int64_t* idx, orl_batch* out) {
  Engine& e = h->e;
  if (idx) {
    for (long i = 0; i < (long)e.R * e.B; ++i) if (idx[i] < 0 || idx[i] >= e.n_data) return fail("sample index out of range");
    ORL_HIP(hipMemcpyAsync(e.d_idx, idx, sizeof(long long) * e.R * e.B, hipMemcpyHostToDevice, e.stream));
  }
  if (e.enqueue_sample(idx ? e.d_idx : nullptr)) return -1;
  ORL_HIP(hipStreamSynchronize(e.stream));
  if (out) {
    out->observations = e.ws["b_obs2"].p; out->next_observations = e.ws["b_obs2"].p + (long)e.B * e.OP;
    out->actions = e.ws["b_act"].p; out->rewards = e.ws["b_rew"].p; out->terminals = e.ws["b_term"].p; out->on_device = 1;
  }
  return 0;
}

// ---- hot path ----
static int copy_rows(Engine& e, const Mat& dst, const float* src, int rows, int dim, bool on_device, long row0 = 0) {
  // src: [R][rows][dim] packed; dst: per-run padded rows
  if (!src) return fail("null input array");
  for (int r = 0; r < e.R; ++r) {
    ORL_HIP(hipMemcpy2DAsync(dst.p + r * dst.rs + row0 * dst.pitch, sizeof(float) * dst.pitch, src + (long)r * rows * dim,
                             sizeof(float) * dim, sizeof(float) * dim, rows, on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, e.stream));
  }
  return 0;
}

int orl_step(orl_engine* h, const orl_batch* b, const orl_noise* nz, float* metrics) {
  Engine& e = h->e;
  ORL_HIP(hipSetDevice(e.dev));
  const int B = e.B, A = e.ad, BN = e.B * e.N, Bt = e.cfg.max_q_backup ? BN : B;
  if (b) {
    const bool dv = b->on_device != 0;
    if (copy_rows(e, e.ws["b_obs2"], b->observations, B, e.od, dv, 0)) return -1;
    if (copy_rows(e, e.ws["b_obs2"], b->next_observations, B, e.od, dv, B)) return -1;
    if (copy_rows(e, e.ws["b_act"], b->actions, B, e.ad, dv)) return -1;
    if (copy_rows(e, e.ws["b_rew"], b->rewards, B, 1, dv)) return -1;
    if (copy_rows(e, e.ws["b_term"], b->terminals, B, 1, dv)) return -1;
  }
  if (nz) {
    const bool dv = nz->on_device != 0;
    if (e.cfg.algo == ORL_ALGO_CQL) {
      if (copy_rows(e, e.ws["n_eps_actor"], nz->slot[0], B, A, dv)) return -1;
      if (copy_rows(e, e.ws["n_eps_next"], nz->slot[1], Bt, A, dv)) return -1;
      if (copy_rows(e, e.ws["n_urand"], nz->slot[2], BN, A, dv)) return -1;
      if (copy_rows(e, e.ws["n_eps_pi"], nz->slot[3], BN, A, dv)) return -1;
      if (copy_rows(e, e.ws["n_eps_npi"], nz->slot[4], BN, A, dv)) return -1;
    }
  } else {
    if (e.enqueue_noise()) return -1;
  }
  if (e.enqueue_step()) return -1;
  e.step_host++;
  ORL_HIP(hipStreamSynchronize(e.stream));
  if (metrics) {
    std::vector<float> m(e.R * e.nm);
    ORL_HIP(hipMemcpy(m.data(), e.metrics_last, sizeof(float) * m.size(), hipMemcpyDeviceToHost));
    for (int r = 0; r < e.R; ++r)
      for (int k = 0; k < ORL_MAX_METRICS; ++k) metrics[r * ORL_MAX_METRICS + k] = k < e.nm ? m[r * e.nm + k] : 0.f;
  }
  return 0;
}

int orl_learn_n(orl_engine* h, int n_steps, float* metrics_mean, float* elapsed_ms) {
  Engine& e = h->e;
  ORL_HIP(hipSetDevice(e.dev));
  if (n_steps <= 0) return fail("n_steps must be positive");
  ORL_HIP(hipMemsetAsync(e.metrics_sum, 0, sizeof(float) * e.R * e.nm, e.stream));
  const bool graphable = e.use_graph && !e.prof_on;
  if (graphable && !e.graph_exec) {
    ORL_HIP(hipStreamBeginCapture(e.stream, hipStreamCaptureModeThreadLocal));
    int rc = e.enqueue_sample(nullptr) || e.enqueue_noise() || e.enqueue_step();
    hipError_t ce = hipStreamEndCapture(e.stream, &e.graph);
    if (rc) return -1;
    if (ce != hipSuccess) return fail(std::string("graph capture: ") + hipGetErrorString(ce));
    ORL_HIP(hipGraphInstantiate(&e.graph_exec, e.graph, nullptr, nullptr, 0));
  }
  hipEvent_t t0, t1;
  ORL_HIP(hipEventCreate(&t0));
  ORL_HIP(hipEventCreate(&t1));
  if (e.prof_on) { e.prof.clear(); e.ev_used = 0; }
  ORL_HIP(hipEventRecord(t0, e.stream));
  for (int s = 0; s < n_steps; ++s) {
    if (graphable) { ORL_HIP(hipGraphLaunch(e.graph_exec, e.stream)); }
    else {
      if (e.enqueue_sample(nullptr)) return -1;
      if (e.enqueue_noise()) return -1;
      if (e.enqueue_step()) return -1;
    }
  }
  ORL_HIP(hipEventRecord(t1, e.stream));
  ORL_HIP(hipStreamSynchronize(e.stream));
  e.step_host += n_steps;
  float ms = 0.f;
  ORL_HIP(hipEventElapsedTime(&ms, t0, t1));
  hipEventDestroy(t0); hipEventDestroy(t1);
  if (elapsed_ms) *elapsed_ms = ms;
  if (metrics_mean) {
    std::vector<float> m(e.R * e.nm);
    ORL_HIP(hipMemcpy(m.data(), e.metrics_sum, sizeof(float) * m.size(), hipMemcpyDeviceToHost));
    for (int r = 0; r < e.R; ++r)
      for (int k = 0; k < ORL_MAX_METRICS; ++k) metrics_mean[r * ORL_MAX_METRICS + k] = k < e.nm ? m[r * e.nm + k] / n_steps : 0.f;
  }
  return 0;
}

int orl_num_metrics(orl_engine* h) { return h->e.nm; }
const char* orl_metric_name(orl_engine* h, int idx) { return (idx >= 0 && idx < h->e.nm) ? h->e.metric_names[idx].c_str() : ""; }
int64_t orl_step_count(orl_engine* h) { return (int64_t)h->e.step_host; }

int64_t orl_debug_read(orl_engine* h, int run, const char* name, float* host, int64_t cap) {
  Engine& e = h->e;
  auto it = e.taps.find(name);
  if (it == e.taps.end()) { fail(std::string("unknown tap ") + name); return -1; }
  if (run < 0 || run >= e.R) { fail("bad run"); return -1; }
  const Engine::Tap& t = it->second;
  const int64_t n = t.rows * t.cols;
  if (cap < n) { fail("tap buffer too small"); return -1; }
  if (hipStreamSynchronize(e.stream) != hipSuccess) { fail("sync"); return -1; }
  if (hipMemcpy2D(host, sizeof(float) * t.cols, t.m.p + run * t.m.rs, sizeof(float) * t.m.pitch, sizeof(float) * t.cols, t.rows,
                  hipMemcpyDeviceToHost) != hipSuccess) { fail("tap copy"); return -1; }
  return n;
}

int orl_profile_enable(orl_engine* h, int on) { h->e.prof_on = on != 0; return 0; }
int orl_profile_query(orl_engine* h, int idx, char* name, int name_cap, double* total_ms, int64_t* launches, double* flops_per_launch) {
  Engine& e = h->e;
  if (hipStreamSynchronize(e.stream) != hipSuccess) return fail("sync");
  std::map<std::string, std::tuple<double, int64_t, double>> agg;
  for (auto& p : e.prof) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.a, p.b) != hipSuccess) continue;
    auto& a = agg[p.name];
    std::get<0>(a) += ms; std::get<1>(a) += 1; std::get<2>(a) = p.flops;
  }
  std::vector<std::pair<double, std::string>> order;
  for (auto& kv : agg) order.push_back({-std::get<0>(kv.second), kv.first});
  std::sort(order.begin(), order.end());
  if (idx < 0 || idx >= (int)order.size()) return 1;
  const auto& a = agg[order[idx].second];
  snprintf(name, name_cap, "%s", order[idx].second.c_str());
  *total_ms = std::get<0>(a); *launches = std::get<1>(a); *flops_per_launch = std::get<2>(a);
  return 0;
}

// kernel unit test: C = op(A) op(B) through one tile configuration.
//  mode 0: forward   C[M,N] = A[M,K] B[N,K]^T + v0[N] (bias), relu
//  mode 1: dgrad     C[M,N] = A[M,K] Bm[K,N], masked by v0 viewed [M,N] (>0)
//  mode 2: wgrad     C[M,N+1]: C[:, :N] = A[K,M]^T B[K,N], C[:, N] = column sums of A ; split-K slabs summed on host
//  mode 3: rank-1 dgrad  A_eff[m,k] = A[m,k]>0 ? v0[m]*v1[k] : 0 ; C = A_eff Bm[K,N]
//  mode 4: rank-1 wgrad  A_eff[k,m] as above (A is [K,M]) ; C[M,N+1] like mode 2
int orl_debug_gemm(int cfg, int mode, int M, int N, int K, const float* A, const float* Bh, const float* v0, const float* v1,
                   float* C, int ksplit, int precision) {
  if (precision != 0) return fail("precision 1 not built");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("no HIP device");
  hipStream_t st = nullptr;
  float *dA = nullptr, *dB = nullptr, *d0 = nullptr, *d1 = nullptr, *dC = nullptr;
  const bool wg = (mode == 2 || mode == 4);
  const long nA = (long)M * K, nB = (long)N * K;
  const long nC = wg ? (long)M * (N + 1) : (long)M * N;
  const long n0 = (mode == 0) ? N : (mode == 1 ? (long)M * N : (mode == 3 ? M : (mode == 4 ? K : 0)));
  const long n1 = (mode == 3) ? K : (mode == 4 ? M : 0);
  if (ksplit < 1) ksplit = 1;
  ORL_HIP(hipMalloc(&dA, sizeof(float) * nA));
  ORL_HIP(hipMalloc(&dB, sizeof(float) * nB));
  ORL_HIP(hipMalloc(&dC, sizeof(float) * nC * ksplit));
  ORL_HIP(hipMemcpy(dA, A, sizeof(float) * nA, hipMemcpyHostToDevice));
  ORL_HIP(hipMemcpy(dB, Bh, sizeof(float) * nB, hipMemcpyHostToDevice));
  ORL_HIP(hipMemset(dC, 0, sizeof(float) * nC * ksplit));
  if (n0) { ORL_HIP(hipMalloc(&d0, sizeof(float) * n0)); ORL_HIP(hipMemcpy(d0, v0, sizeof(float) * n0, hipMemcpyHostToDevice)); }
  if (n1) { ORL_HIP(hipMalloc(&d1, sizeof(float) * n1)); ORL_HIP(hipMemcpy(d1, v1, sizeof(float) * n1, hipMemcpyHostToDevice)); }
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.nz1 = 1; p.ksplit = ksplit; p.C = dC; p.c_ks = nC;
  hipError_t err = hipSuccess;
  if (mode == 0) {
    p.A = {dA, 0, 0}; p.a_sr = K; p.a_sk = 1; p.B = {dB, 0, 0}; p.b_sr = K; p.b_sk = 1;
    p.M = M; p.N = N; p.K = K; p.c_sr = N; p.bias = {d0, 0, 0};
    err = launch_gemm<PA_PLAIN, PB_PLAIN, E_BIAS_RELU>(cfg, p, 1, st);
  } else if (mode == 1 || mode == 3) {
    p.A = {dA, 0, 0}; p.a_sr = K; p.a_sk = 1; p.B = {dB, 0, 0}; p.b_sr = 1; p.b_sk = N;
    p.M = M; p.N = N; p.K = K; p.c_sr = N;
    if (mode == 1) { p.aux = {d0, 0, 0}; p.aux_sr = N; err = launch_gemm<PA_PLAIN, PB_PLAIN, E_MASK>(cfg, p, 1, st); }
    else { p.rowv = {d0, 0, 0}; p.colv = {d1, 0, 0}; p.a_trans = 0; err = launch_gemm<PA_RANK1, PB_PLAIN, E_PLAIN>(cfg, p, 1, st); }
  } else {
    p.A = {dA, 0, 0}; p.a_sr = 1; p.a_sk = M; p.B = {dB, 0, 0}; p.b_sr = 1; p.b_sk = N;
    p.M = M; p.N = N + 1; p.K = K; p.c_sr = N; p.ones_row = N;
    p.bias_out = dC + (long)M * N; p.bo_ks = nC;
    if (mode == 2) err = launch_gemm<PA_PLAIN, PB_ONES, E_WGRAD>(cfg, p, 1, st);
    else { p.rowv = {d0, 0, 0}; p.colv = {d1, 0, 0}; p.a_trans = 1; err = launch_gemm<PA_RANK1, PB_ONES, E_WGRAD>(cfg, p, 1, st); }
  }
  if (err != hipSuccess) return fail(std::string("debug gemm launch: ") + hipGetErrorString(err));
  ORL_HIP(hipDeviceSynchronize());
  std::vector<float> tmp(nC * ksplit);
  ORL_HIP(hipMemcpy(tmp.data(), dC, sizeof(float) * nC * ksplit, hipMemcpyDeviceToHost));
  for (long i = 0; i < nC; ++i) {
    float s = 0.f;
    for (int k = 0; k < ksplit; ++k) s += tmp[k * nC + i];
    C[i] = s;
  }
  hipFree(dA); hipFree(dB); hipFree(dC); if (d0) hipFree(d0); if (d1) hipFree(d1);
  return 0;
}

}  // extern "C"
