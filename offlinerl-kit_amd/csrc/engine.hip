// engine.hip — MI355X-native update engine: host side (arenas, launch helpers, ABI).
// Algorithm schedules live in algo_*.inc (same translation unit).
// Build: hipcc -O3 --offload-arch=gfx950 -fPIC -shared engine.hip -o liborlengine.so
#include "engine.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <tuple>

namespace orl {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int fail(const std::string& msg) {
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

// seq_step: distance of consecutive Linear layers in the backbone's nn.Sequential -- 2 for [Linear, ReLU], 3 when every ReLU is followed by
// nn.Dropout (nets/mlp.py:20-23): the state_dict keys are backbone.model.{0, 3, 6, ...} then
static NetLayout make_mlp_layout(int in_dim, const int* hidden, int L, TailKind tail, int act_dim, int seq_step = 2) {
  NetLayout l;
  l.present = true;
  l.in_dim = in_dim;
  l.L = L;
  long off = 0;
  int d = in_dim;
  for (int i = 0; i < L; ++i) {
    l.H[i] = hidden[i];
    l.w_off[i] = off;
    add_tensor(l, "backbone.model." + std::to_string(seq_step * i) + ".weight", off, {hidden[i], d});
    off += (long)hidden[i] * d;
    l.b_off[i] = off;
    add_tensor(l, "backbone.model." + std::to_string(seq_step * i) + ".bias", off, {hidden[i]});
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
  for (int i = 0; i <= L; ++i) l.w_ms[i] = l.b_ms[i] = l.stride();
  return l;
}

// MCQ's behaviour policy (nets/vae.py:8-66) as two MLP families whose tensors carry the reference's parameter names:
//   encoder e1, e2 (ReLU) + tail [mean ; log_std] stored as one (2Z x VH) matrix;  decoder d1, d2 (ReLU) + tail d3
static NetLayout make_vae_layout(bool encoder, int in_dim, int vh, int out_dim) {
  NetLayout l;
  l.present = true;
  l.in_dim = in_dim; l.L = 2; l.out_dim = encoder ? 2 * out_dim : out_dim;
  const char* n0 = encoder ? "e1" : "d1";
  const char* n1 = encoder ? "e2" : "d2";
  long off = 0;
  int d = in_dim;
  const char* names[2] = {n0, n1};
  for (int i = 0; i < 2; ++i) {
    l.H[i] = vh;
    l.w_off[i] = off; add_tensor(l, std::string(names[i]) + ".weight", off, {vh, d}); off += (long)vh * d;
    l.b_off[i] = off; add_tensor(l, std::string(names[i]) + ".bias", off, {vh}); off += vh;
    d = vh;
  }
  l.w_off[2] = off;
  if (encoder) {
    add_tensor(l, "mean.weight", off, {out_dim, d});
    add_tensor(l, "log_std.weight", off + (long)out_dim * d, {out_dim, d});
    off += 2L * out_dim * d;
    l.b_off[2] = off;
    add_tensor(l, "mean.bias", off, {out_dim});
    add_tensor(l, "log_std.bias", off + out_dim, {out_dim});
    off += 2 * out_dim;
  } else {
    add_tensor(l, "d3.weight", off, {out_dim, d}); off += (long)out_dim * d;
    l.b_off[2] = off; add_tensor(l, "d3.bias", off, {out_dim}); off += out_dim;
  }
  l.size = off;
  for (int i = 0; i <= 2; ++i) l.w_ms[i] = l.b_ms[i] = l.stride();
  return l;
}

// EnsembleCritic (modules/ensemble_critic_module.py:11-31): model.{0,2,..}.weight (K,in,out), .bias (K,1,out)
static NetLayout make_ensemble_layout(int in_dim, const int* hidden, int L, int K) {
  NetLayout l;
  l.present = true; l.ens = true; l.members = K;
  l.in_dim = in_dim; l.L = L; l.out_dim = 1;
  long off = 0;
  int d = in_dim;
  for (int i = 0; i <= L; ++i) {
    const int o = (i == L) ? 1 : hidden[i];
    if (i < L) l.H[i] = hidden[i];
    l.w_off[i] = off; l.w_ms[i] = (long)d * o;
    add_tensor(l, "model." + std::to_string(2 * i) + ".weight", off, {K, d, o});
    off += (long)K * d * o;
    l.b_off[i] = off; l.b_ms[i] = o;
    add_tensor(l, "model." + std::to_string(2 * i) + ".bias", off, {K, 1, o});
    off += (long)K * o;
    d = o;
  }
  l.size = off;
  return l;
}

static int build_layouts(const orl_config& c, NetLayout* lay, long* net_off, bool* is_tgt, long* P_train, long* P_tgt) {
  for (int i = 0; i < ORL_NUM_NETS; ++i) { lay[i] = NetLayout(); net_off[i] = 0; is_tgt[i] = false; }
  if (c.n_hidden < 1 || c.n_hidden > ORL_MAX_HIDDEN) return fail("n_hidden must be in [1,4]");
  if (c.obs_dim < 1 || c.act_dim < 1 || c.batch_size < 1 || c.n_runs < 1) return fail("bad dims");
  if (c.act_dim > 32) return fail("act_dim > 32 not supported");
  const int od = c.obs_dim, ad = c.act_dim, L = c.n_hidden;
  const NetLayout crit = make_mlp_layout(od + ad, c.hidden, L, TAIL_CRITIC, ad);
  long o = 0, t = 0;
  auto train = [&](int id, const NetLayout& l) { lay[id] = l; net_off[id] = o; o += l.stride(); };
  auto target = [&](int id, const NetLayout& l) { lay[id] = l; net_off[id] = t; is_tgt[id] = true; t += l.stride(); };
  if (c.algo == ORL_ALGO_CQL || c.algo == ORL_ALGO_SAC || c.algo == ORL_ALGO_MCQ) {
    train(ORL_NET_ACTOR, make_mlp_layout(od, c.hidden, L, TAIL_TANH_GAUSS, ad));
    train(ORL_NET_CRITIC1, crit); train(ORL_NET_CRITIC2, crit);
    target(ORL_NET_CRITIC1_OLD, crit); target(ORL_NET_CRITIC2_OLD, crit);
    if (c.algo == ORL_ALGO_MCQ) {
      if (c.vae_hidden < 1 || c.vae_latent < 1 || c.vae_latent > 64) return fail("MCQ: vae_hidden / vae_latent out of range");
      train(ORL_NET_VAE_ENC, make_vae_layout(true, od + ad, c.vae_hidden, c.vae_latent));
      train(ORL_NET_VAE_DEC, make_vae_layout(false, od + c.vae_latent, c.vae_hidden, ad));
    }
  } else if (c.algo == ORL_ALGO_IQL) {
    train(ORL_NET_ACTOR, make_mlp_layout(od, c.hidden, L, TAIL_GAUSS, ad, c.actor_dropout > 0.f ? 3 : 2));
    train(ORL_NET_CRITIC1, crit); train(ORL_NET_CRITIC2, crit);
    train(ORL_NET_CRITIC_V, make_mlp_layout(od, c.hidden, L, TAIL_CRITIC, ad));
    target(ORL_NET_CRITIC1_OLD, crit); target(ORL_NET_CRITIC2_OLD, crit);
  } else if (c.algo == ORL_ALGO_TD3BC) {
    const NetLayout act = make_mlp_layout(od, c.hidden, L, TAIL_DET, ad);
    train(ORL_NET_ACTOR, act);
    train(ORL_NET_CRITIC1, crit); train(ORL_NET_CRITIC2, crit);
    target(ORL_NET_CRITIC1_OLD, crit); target(ORL_NET_CRITIC2_OLD, crit);
    target(ORL_NET_ACTOR_OLD, act);
  } else if (c.algo == ORL_ALGO_EDAC) {
    if (c.num_critics < 2 || c.num_critics > 64) return fail("num_critics must be in [2,64]");
    train(ORL_NET_ACTOR, make_mlp_layout(od, c.hidden, L, TAIL_TANH_GAUSS, ad));
    const NetLayout ens = make_ensemble_layout(od + ad, c.hidden, L, c.num_critics);
    train(ORL_NET_CRITIC1, ens);
    target(ORL_NET_CRITIC1_OLD, ens);
  } else {
    return fail("unknown algorithm id");
  }
  *P_train = o; *P_tgt = t;
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Buffer
// ---------------------------------------------------------------------------------------------
Buffer::~Buffer() {
  hipSetDevice(dev);
  for (float* p : {obs, nobs, act, rew, term}) if (p) hipFree(p);
  if (idx) hipFree(idx);
}

// ---------------------------------------------------------------------------------------------
// Engine basics
// ---------------------------------------------------------------------------------------------
Engine::~Engine() {
  for (int i = 0; i < 2; ++i) {
    if (graph_exec[i]) hipGraphExecDestroy(graph_exec[i]);
    if (graph[i]) hipGraphDestroy(graph[i]);
  }
  for (auto& e : ev_pool) hipEventDestroy(e);
  for (void* p : allocs) hipFree(p);
  if (forked) std::swap(stream, side_stream);
  if (ev_fork) hipEventDestroy(ev_fork);
  if (ev_join) hipEventDestroy(ev_join);
  if (side_stream) hipStreamDestroy(side_stream);
  if (stream) hipStreamDestroy(stream);
}

void Engine::drop_graphs() {
  for (int i = 0; i < 2; ++i) {
    if (graph_exec[i]) { hipGraphExecDestroy(graph_exec[i]); graph_exec[i] = nullptr; }
    if (graph[i]) { hipGraphDestroy(graph[i]); graph[i] = nullptr; }
  }
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
  if (pitch >= 128 && (pitch & 127) == 0) {          // hidden-activation shaped: room for the packed ReLU masks
    m.bg = pitch / 32;                                 // one 32-bit word per 32 columns
    m.bcs = rows * m.bg; m.brs = m.bcs * nets;
    m.bits = (unsigned int*)raw_alloc(sizeof(unsigned int) * m.brs * R);
  }
  ws[name] = m;
  ws_len[name] = per_run;
  return m;
}

float* Engine::net_ptr(int run, int net) const {
  if (net < 0 || net >= ORL_NUM_NETS || !lay[net].present) return nullptr;
  if (net_is_target[net]) return arena + (long)R * P_train + (long)run * P_tgt + net_off[net];
  return arena + (long)run * P_train + net_off[net];
}

NetRef Engine::net_ref(int net, int members) const {
  NetRef r;
  r.base = net_ptr(0, net);
  r.rs = net_is_target[net] ? P_tgt : P_train;
  r.g_off = net_off[net];
  r.lay = &lay[net];
  r.nz1 = members;
  return r;
}

void Engine::prof_begin(const char* name, double flops, double bytes) {
  if (!prof_on) return;
  while (ev_pool.size() < ev_used + 2) {
    hipEvent_t e;
    hipEventCreate(&e);
    ev_pool.push_back(e);
  }
  ProfEntry pe;
  pe.name = name;
  pe.flops = flops;
  pe.bytes = bytes;
  pe.a = ev_pool[ev_used++];
  pe.b = ev_pool[ev_used++];
  hipEventRecord(pe.a, stream);
  prof.push_back(pe);
}
void Engine::prof_end() {
  if (!prof_on) return;
  hipEventRecord(prof.back().b, stream);
}

int Engine::fork_side() {
  forked = false;
  if (!fork_on || prof_on) return 0;                 // (the per-tag timing of a profiled run brackets launches on the main stream)
  if (hipEventRecord(ev_fork, stream) != hipSuccess || hipStreamWaitEvent(side_stream, ev_fork, 0) != hipSuccess) return fail("fork: event");
  std::swap(stream, side_stream);
  forked = true;
  return 0;
}
int Engine::fork_join() {
  if (!forked) return 0;
  forked = false;                                    // (fork_main() has swapped the streams back: side_stream holds the side work)
  if (hipEventRecord(ev_join, side_stream) != hipSuccess || hipStreamWaitEvent(stream, ev_join, 0) != hipSuccess) return fail("join: event");
  return 0;
}

float* Engine::gscale_slot() {
  if (!split_scales()) return nullptr;
  if (gscale_next >= GSCALE_SLOTS) { fail("grad_scale: out of slots"); return nullptr; }
  return gscale_buf + (long)(gscale_next++) * R;
}

// split precision: choose the dynamic scale of a backward pass from its seed (kernels.h: k_grad_scale); null at precision 0
const float* Engine::grad_scale(const Mat& seed, int rows, int cols, int nets, const char* tag) {
  if (!split_scales()) return nullptr;
  if (gscale_next >= GSCALE_SLOTS) { fail("grad_scale: out of slots"); return nullptr; }
  float* out = gscale_buf + (long)(gscale_next++) * R;
  GradScaleP g;
  g.seed = seed.p; g.rs = seed.rs; g.cs = seed.cs; g.rows = rows; g.cols = cols; g.pitch = seed.pitch; g.nets = nets; g.out = out;
  prof_begin((std::string(tag) + ".gscale").c_str(), 0);
  hipLaunchKernelGGL(k_grad_scale, dim3(R), dim3(256), 0, stream, g);
  prof_end();
  return out;
}

void Engine::watch_range(const Mat& m, int rows, int cols, int nets, const char* what) {
  if (!split_scales() || !m.p) return;
  RangeWatch w{m, rows, cols, nets, what ? what : ""};
  range_watch[m.p] = w;
}

// every matrix the last enqueued step feeds to the split-precision MFMAs at operand scale 1 (dead ones -- never stored by the forward pass --
// skipped), then the parameters of every net (operand scale ORL_WSCALE)
int Engine::range_scan() {
  if (!split_scales()) return 0;
  const float lim = 65504.0f;
  for (auto& kv : range_watch) {
    const RangeWatch& w = kv.second;
    if (vals_dead.count(w.m.p)) continue;
    hipLaunchKernelGGL(k_range_scan, dim3(R), dim3(256), 0, stream, (const float*)w.m.p, w.m.rs, w.m.cs, w.nets, w.rows, w.cols, w.m.pitch, lim, health);
  }
  hipLaunchKernelGGL(k_range_scan, dim3(R), dim3(256), 0, stream, (const float*)arena, P_train, 0L, 1, 1, (int)P_train, (int)P_train, lim / ORL_WSCALE, health);
  if (P_tgt > 0)
    hipLaunchKernelGGL(k_range_scan, dim3(R), dim3(256), 0, stream, (const float*)(arena + (long)R * P_train), P_tgt, 0L, 1, 1, (int)P_tgt, (int)P_tgt, lim / ORL_WSCALE, health);
  return hipGetLastError() == hipSuccess ? 0 : fail("range scan launch");
}

// Called after the stream has been synchronised: folds what the host just read (non-finite metrics) and what the kernels raised into the
// sticky per-run flags.  A split-precision engine gets the operands of its last step scanned for the fp16-plane range when a run turned
// non-finite (so the report can name the cause) and every RANGE_SCAN_EVERY steps: an activation beyond 65504 does NOT surface by itself --
// its hi / lo planes are +inf / -inf, the MFMA adds them to the NaN 0xffc00000 (tools/probes/nan_sign_probe.hip, measured on gfx950), and
// a ReLU (integer view or v_max_f32 alike) turns a sign-bit NaN into +0: the row's next layer is silently all zero.  The scan is a pass
// over the workspaces (~0.1 % of 256 steps); steps_done < 0: scan now (orl_health_check).
// *any = OR over the runs.  Sets the error string (not the return code) when a flag is up.
int Engine::health_update(const float* metrics, long steps_done, unsigned int* any) {
  std::vector<unsigned int> dv(R);
  ORL_HIP(hipMemcpy(dv.data(), health, sizeof(unsigned int) * R, hipMemcpyDeviceToHost));
  bool fresh = false;
  for (int r = 0; r < R; ++r) {
    unsigned int f = dv[r];
    if (metrics)
      for (int k = 0; k < nm; ++k) if (!std::isfinite(metrics[(long)r * nm + k])) f |= ORL_HEALTH_NONFINITE_LOSS;
    if (f & ~health_host[r]) fresh = true;
    health_host[r] |= f;
  }
  steps_since_scan += steps_done > 0 ? steps_done : 0;
  if (split_scales() && (fresh || steps_done < 0 || steps_since_scan >= RANGE_SCAN_EVERY)) {
    steps_since_scan = 0;
    if (range_scan()) return -1;
    ORL_HIP(hipStreamSynchronize(stream));
    ORL_HIP(hipMemcpy(dv.data(), health, sizeof(unsigned int) * R, hipMemcpyDeviceToHost));
    for (int r = 0; r < R; ++r) health_host[r] |= dv[r];
  }
  unsigned int all = 0;
  for (int r = 0; r < R; ++r) all |= health_host[r];
  if (any) *any = all;
  if (all) {
    std::string msg = "unhealthy run(s):";
    int listed = 0;
    for (int r = 0; r < R && listed < 8; ++r) {
      if (!health_host[r]) continue;
      msg += " run " + std::to_string(r) + " [";
      if (health_host[r] & ORL_HEALTH_NONFINITE_LOSS) msg += " non-finite loss";
      if (health_host[r] & ORL_HEALTH_NONFINITE_GRAD) msg += " non-finite gradient";
      if (health_host[r] & ORL_HEALTH_SPLIT_RANGE) msg += " operand beyond the split-precision range (|x| >= 65504 or |w| >= 1023.5: use precision 0 or normalise the inputs)";
      msg += " ]";
      ++listed;
    }
    set_error(msg);
  }
  return 0;
}

#define ORL_LAUNCH(tag, kernel, grid, block, ...)                                      \
  do {                                                                                 \
    prof_begin(tag, 0);                                                                \
    hipLaunchKernelGGL(kernel, grid, block, 0, stream, __VA_ARGS__);                   \
    prof_end();                                                                        \
    if (hipGetLastError() != hipSuccess) return fail(std::string(tag) + ": launch failed"); \
  } while (0)

template <int PA, int PB, int EPI>
static int run_gemm(Engine* e, int cfg, const GemmP& p, int nz, const char* tag, bool a_kpad = false) {
  const double flops = 2.0 * p.M * (double)p.N * p.K * nz;
  // algorithmic bytes: each operand read once, the result written once (split-K: one slab per split), mask read once
  const double bytes = 4.0 * nz * ((double)p.M * p.K + (double)p.N * p.K + (double)p.M * p.N * (EPI == E_WGRAD ? p.ksplit : 1) +
                                   (EPI == E_MASK ? (double)p.M * p.N / (p.aux_bits ? 32.0 : 1.0) : 0.0));
  double bytes_adj = bytes;
  if (EPI == E_MASK && p.w0_out) bytes_adj += 4.0 * nz * ((double)p.M * p.w0_xsr - (p.C ? 0.0 : (double)p.M * p.N));
  e->prof_begin(tag, flops + (EPI == E_MASK && p.w0_out ? 2.0 * p.M * (double)p.N * (p.w0_in + 1) * nz : 0.0), bytes_adj);
  hipError_t err = launch_gemm<PA, PB, EPI>(cfg, p, nz, e->stream, a_kpad, e->force_scalar, e->mm_prec());
  e->prof_end();
  if (err != hipSuccess) return fail(std::string("gemm launch ") + tag + ": " + hipGetErrorString(err));
  return 0;
}

// gradient w.r.t. a layer output [M x out].  rank1: dz = (H > 0) ? rowv[m] * w_tail[n] : 0 (never materialised)
struct DY {
  bool rank1 = false;
  Mat m;     // plain: the gradient matrix ; rank1: post-ReLU activation H of the top hidden layer
  Mat rowv;  // rank1: dLoss/dq per row
  static DY plain(const Mat& m) { DY d; d.m = m; return d; }
  static DY virt(const Mat& H, const Mat& dq) { DY d; d.rank1 = true; d.m = H; d.rowv = dq; return d; }
};

// fuse_X0 (layer == 1 only): X = hs[0] has not been computed yet; it is relu(fuse_X0 W0^T + b0).  The weight-stationary kernel
// produces it inside this launch (stores it to X for the backward pass); any other path first runs layer 0 on its own.
int Engine::linear_fwd(const Mat& X, int M, const NetRef& nr, int layer, const Mat& Y, int epi, const Mat* maskH,
                       const char* tag, int in_row0, int in_rows, const Mat* tail_out, bool* tail_fused, const Mat* fuse_X0, const char* tag0,
                       const float* x_dscale) {
  const NetLayout& l = *nr.lay;
  const int in = l.layer_in(layer), out = l.layer_out(layer);
  if (in_rows < 0) in_rows = in;
  if (!x_dscale) {                                    // an input / activation operand (scale 1): orl_health_check scans it for the fp16-plane range
    watch_range(X, M, in_rows, X.cs ? nr.nz1 : 1, tag);
    if (fuse_X0) watch_range(*fuse_X0, M, l.layer_in(0), fuse_X0->cs ? nr.nz1 : 1, tag0 ? tag0 : tag);
  }
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.b_scale = ORL_WSCALE; p.a_dscale = x_dscale;      // split precision: static scale of the weight operand, dynamic one of a gradient-like input
  p.A = {X.p, X.rs, X.cs};
  p.a_sr = X.pitch; p.a_sk = 1;
  if (l.ens) { p.B = {nr.base + l.w_off[layer] + (long)in_row0 * out, nr.rs, l.w_ms[layer]}; p.b_sr = 1; p.b_sk = out; p.b_rlim = out & ~3; }
  else { p.B = {nr.base + l.w_off[layer] + in_row0, nr.rs, l.w_ms[layer]}; p.b_sr = in; p.b_sk = 1; }
  p.C = Y.p; p.c_s0 = Y.rs; p.c_s1 = Y.cs; p.c_sr = Y.pitch; p.c_sn = 1;
  p.M = M; p.N = out; p.K = in_rows;
  const bool a_kpad = X.pitch >= ((in_rows + 3) & ~3);   // input matrices are zero-padded to 16 B rows
  p.nz1 = nr.nz1; p.ksplit = 1;
  vals_dead.erase(Y.p);
  p.bias = {nr.base + l.b_off[layer], nr.rs, l.b_ms[layer]};
  if (maskH) {
    p.aux = {maskH->p, maskH->rs, maskH->cs}; p.aux_sr = maskH->pitch;
    // the ReLU mask as packed bits (1/32 of the bytes) when the forward pass that produced the activation left them behind
    if (epi == E_MASK && maskH->bits && bits_live.count(maskH->bits) && out == maskH->pitch && aligned16(maskH->p) && (maskH->pitch & 3) == 0) {
      p.aux_bits = maskH->bits; p.xb_s0 = maskH->brs; p.xb_s1 = maskH->bcs; p.xb_g = maskH->bg;
    }
  }
  const int nz = R * nr.nz1;
  int cfg = pick_cfg(p.M, p.N, p.K, nz);
  if (cfg == CFG_SQ) cfg = CFG_SQ8;          // forward products: the 8-wave flavour of the 128 x 128 tile measured faster
  // weight-stationary row-streaming kernel (csrc/ws_gemm.h) for the many-row 256 x 256 hidden layers in split-bf16 precision
  if (epi == E_BIAS_RELU && ws_precision_ok() && !no_ws && !force_scalar && in_row0 == 0 && in_rows == in &&
      Y.bits && Y.pitch == out && (long)M * nz >= ws_fwd_min_rows) {   // measured faster than the 16x64 tiles from 16 x 256 rows up
    WsFwdP w;
    memset(&w, 0, sizeof(w));
    w.X = X.p; w.x_s0 = X.rs; w.x_s1 = X.cs; w.x_pitch = X.pitch;
    w.W = nr.base + l.w_off[layer]; w.w_s0 = nr.rs; w.w_s1 = l.w_ms[layer];
    if (l.ens) { w.w_sn = 1; w.w_sk = out; } else { w.w_sn = in; w.w_sk = 1; }      // EnsembleLinear keeps (in, out)-major weights
    w.bias = nr.base + l.b_off[layer]; w.b_s0 = nr.rs; w.b_s1 = l.b_ms[layer];
    w.Y = Y.p; w.y_s0 = Y.rs; w.y_s1 = Y.cs; w.y_pitch = Y.pitch;
    w.mb = Y.bits; w.mb_s0 = Y.brs; w.mb_s1 = Y.bcs; w.mb_g = Y.bg;
    w.M = M; w.nz1 = nr.nz1; w.f32 = this->ws_f32();
    const bool want_tail = tail_out && tail_fused && layer == l.L - 1 && l.out_dim == 1;
    if (want_tail) {
      w.tw = nr.base + l.w_off[l.L]; w.tw_s0 = nr.rs; w.tw_s1 = l.w_ms[l.L];
      w.tb = nr.base + l.b_off[l.L]; w.tb_s0 = nr.rs; w.tb_s1 = l.b_ms[l.L];
      w.tq = tail_out->p; w.tq_s0 = tail_out->rs; w.tq_s1 = tail_out->cs; w.tq_sm = tail_out->pitch;
    }
    // With the single-output tail folded in, the backward pass of a many-row batch needs only the mask bits of this activation
    // (ws_dgrad_w0 / ws_wgrad's derived tail gradients): the activation itself then never goes to HBM.
    const bool elide = want_tail && layer >= 1 && (fwd_only || (elide_top && !l.ens && ws_wgrad_rows_ok(M, nz)));
    if (elide) w.Y = nullptr;
    const bool ws_ok = ws_fwd_supported(w, in, out);
    bool fused0 = false;
    if (ws_ok && fuse_X0 && layer == 1 && X.bits && X.pitch == in) {
      w.X0 = fuse_X0->p; w.x0_s0 = fuse_X0->rs; w.x0_s1 = fuse_X0->cs; w.x0_pitch = fuse_X0->pitch; w.in0 = l.layer_in(0);
      w.W0 = nr.base + l.w_off[0]; w.w0_s0 = nr.rs; w.w0_s1 = l.w_ms[0];
      if (l.ens) { w.w0_sn = 1; w.w0_sk = l.layer_out(0); } else { w.w0_sn = l.layer_in(0); w.w0_sk = 1; }
      w.b0 = nr.base + l.b_off[0]; w.b0_s0 = nr.rs; w.b0_s1 = l.b_ms[0];
      w.mb0 = X.bits; w.mb0_s0 = X.brs; w.mb0_s1 = X.bcs; w.mb0_g = X.bg;
      fused0 = ws_fwd01_supported(w) && aligned16(fuse_X0->p);
      if (!fused0) w.X0 = nullptr;
      else if (fwd_only || (!want_tail && recompute_h0_ok(l, layer, M, nz))) w.x0_discard = 1;      // (a three-layer net's middle-layer wgrad rebuilds h0 from the input rows)
    }
    if (fuse_X0 && !fused0) {      // layer 0 on its own, then this layer
      if (linear_fwd(*fuse_X0, M, nr, 0, X, E_BIAS_RELU, nullptr, tag0 ? tag0 : tag)) return -1;
      fuse_X0 = nullptr;
    }
    if (ws_ok) {
      // precision 2: the fused first + second layer + tail forward of a many-row single-output net has a three-plane flavour (ws_fwd3.hip);
      // its two column halves leave two tail partial sums
      bool fwd3 = false;
      if (p3(1) && (fused0 || !w.X0) && ws_dump && (!want_tail || ws.count("tq_scratch"))) {
        bool room = true;
        if (want_tail) {
          const Mat& sc = ws.at("tq_scratch");
          room = (long)M <= sc.cs && nr.nz1 <= tq_scratch_nets;
          if (room) { w.tq2 = sc.p; w.tq2_s0 = sc.rs; w.tq2_s1 = sc.cs; }
        }
        if (room) {
          w.np3 = 1; w.dump = ws_dump;
          fwd3 = ws_fwd3_supported(w, in, out);
          if (!fwd3) { w.np3 = 0; w.tq2 = nullptr; w.dump = nullptr; }
        }
      }
      const double f0 = fused0 ? 2.0 * M * (double)in * (w.in0 + 1) * nz : 0.0;
      prof_begin(fwd3 ? (std::string(tag) + "@p3").c_str() : tag, f0 + 2.0 * M * (double)out * (in + (want_tail ? 1 : 0)) * nz,
                 4.0 * nz * ((fused0 ? (double)M * w.x0_pitch : 0.0) + (w.x0_discard ? (double)M * in / 32 : (double)M * in) + (double)out * in +
                             (elide ? (double)M * out / 32 : (double)M * out)));
      if (fused0) bits_live.insert(X.bits);
      if (w.x0_discard) vals_dead.insert(X.p);
      else if (fused0) vals_dead.erase(X.p);
      if (M >= 1024 && !fwd_only) w.lab_clk = (unsigned long long*)(aloss_part + (long)R * SB_MAXGROUPS * 2) + 64;      // (lab builds: the critic pass)
      hipError_t err = fwd3 ? launch_ws_fwd3(w, nz, ws_blocks_per_problem(M / WS_ROWS, 2 * nz, 10, 1 << 20, ws_geo), stream) : launch_ws_fwd(w, nz, stream, ws_geo);
      prof_end();
      if (err != hipSuccess) return fail(std::string("ws_fwd launch ") + tag + ": " + hipGetErrorString(err));
      if (fwd3 && want_tail) {
        TailAddP t;
        t.out = w.tq; t.o_s0 = w.tq_s0; t.o_s1 = w.tq_s1; t.o_sm = w.tq_sm;
        t.part = w.tq2; t.p_s0 = w.tq2_s0; t.p_s1 = w.tq2_s1; t.p_ts = M; t.nparts = 1; t.M = M; t.nz1 = nr.nz1;
        prof_begin((std::string(tag) + ".tail_add").c_str(), 0);
        hipLaunchKernelGGL(k_tail_add, dim3((M + 255) / 256, nz), dim3(256), 0, stream, t);
        prof_end();
        if (hipGetLastError() != hipSuccess) return fail("tail_add launch");
      }
      bits_live.insert(Y.bits);
      if (elide) vals_dead.insert(Y.p);
      if (tail_fused) *tail_fused = want_tail;
      return 0;
    }
  }
  if (fuse_X0) {                   // no fused path: layer 0 first
    if (linear_fwd(*fuse_X0, M, nr, 0, X, E_BIAS_RELU, nullptr, tag0 ? tag0 : tag)) return -1;
  }
  if (Y.bits) {
    if (epi == E_BIAS_RELU && !force_scalar && out == Y.pitch && mb_supported(cfg, p)) {
      p.mb_out = Y.bits; p.mb_s0 = Y.brs; p.mb_s1 = Y.bcs; p.mb_g = Y.bg;
      bits_live.insert(Y.bits);
    } else bits_live.erase(Y.bits);
  }
  if (tail_fused) *tail_fused = false;
  int tq_parts = 0;
  if (tail_out && tail_fused && epi == E_BIAS_RELU && layer == l.L - 1 && l.out_dim == 1 && !force_scalar && ws.count("tq_scratch")) {
    const float* tw = nr.base + l.w_off[l.L];
    tq_parts = tq_fused_parts(cfg, p, tw, nr.rs, l.w_ms[l.L]);
    const Mat& sc = ws.at("tq_scratch");
    if (tq_parts >= 1 && (long)(tq_parts - 1) * M <= sc.cs && nr.nz1 <= tq_scratch_nets) {
      p.tq_w = {tw, nr.rs, l.w_ms[l.L]};
      p.tq_b = {nr.base + l.b_off[l.L], nr.rs, l.b_ms[l.L]};
      p.tq_out = tail_out->p; p.tq_s0 = tail_out->rs; p.tq_s1 = tail_out->cs; p.tq_sm = tail_out->pitch;
      p.tq_part = sc.p; p.tq_ps0 = sc.rs; p.tq_ps1 = sc.cs; p.tq_ts = M;
      *tail_fused = true;
    } else tq_parts = 0;
  }
  if (tq_parts >= 1) {
    if (run_gemm<PA_PLAIN, PB_PLAIN, E_BIAS_RELU>(this, cfg, p, nz, tag, a_kpad)) return -1;
    if (tq_parts > 1) {
      TailAddP t;
      t.out = p.tq_out; t.o_s0 = p.tq_s0; t.o_s1 = p.tq_s1; t.o_sm = p.tq_sm;
      t.part = p.tq_part; t.p_s0 = p.tq_ps0; t.p_s1 = p.tq_ps1; t.p_ts = p.tq_ts; t.nparts = tq_parts - 1; t.M = M; t.nz1 = nr.nz1;
      prof_begin((std::string(tag) + ".tail_add").c_str(), 0);
      hipLaunchKernelGGL(k_tail_add, dim3((M + 255) / 256, nz), dim3(256), 0, stream, t);
      prof_end();
      if (hipGetLastError() != hipSuccess) return fail("tail_add launch");
    }
    return 0;
  }
  switch (epi) {
    case E_BIAS_RELU: return run_gemm<PA_PLAIN, PB_PLAIN, E_BIAS_RELU>(this, cfg, p, nz, tag, a_kpad);
    case E_BIAS: return run_gemm<PA_PLAIN, PB_PLAIN, E_BIAS>(this, CFG_AUTO, p, nz, tag, a_kpad);
    case E_MASK: return run_gemm<PA_PLAIN, PB_PLAIN, E_MASK>(this, CFG_AUTO, p, nz, tag, a_kpad);
    default: return run_gemm<PA_PLAIN, PB_PLAIN, E_PLAIN>(this, CFG_AUTO, p, nz, tag, a_kpad);
  }
}

int Engine::linear_dgrad(const DY& dy, int M, const NetRef& nr, int layer, int col0, int ncols, const Mat* maskH,
                         const Mat& dX, const char* tag, const Mat* w0_X, bool store_dx, int* w0_slabs) {
  const NetLayout& l = *nr.lay;
  const int in = l.layer_in(layer), out = l.layer_out(layer);
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.a_dscale = cur_gscale; p.b_scale = ORL_WSCALE;     // split precision: A = a gradient matrix of the current backward pass, B = weights
  if (mm_prec() == P_SPLIT3) p.a_scale = ORL_GSCALE3;   // three planes: 2^5 more headroom above fp16's 2^-24 grid for the matrix's small elements
  p.A = {dy.m.p, dy.m.rs, dy.m.cs};
  p.a_sr = dy.m.pitch; p.a_sk = 1;
  if (dy.rank1) {
    p.a_trans = 0;
    p.rowv = {dy.rowv.p, dy.rowv.rs, dy.rowv.cs};
    p.colv = {nr.base + l.w_off[l.L], nr.rs, l.w_ms[l.L]};
  }
  if (l.ens) { p.B = {nr.base + l.w_off[layer] + (long)col0 * out, nr.rs, l.w_ms[layer]}; p.b_sr = out; p.b_sk = 1; }
  else { p.B = {nr.base + l.w_off[layer] + col0, nr.rs, l.w_ms[layer]}; p.b_sr = 1; p.b_sk = in; p.b_rlim = (in - col0) & ~3; }
  p.C = dX.p; p.c_s0 = dX.rs; p.c_s1 = dX.cs; p.c_sr = dX.pitch; p.c_sn = 1;
  p.M = M; p.N = ncols; p.K = out;
  p.nz1 = nr.nz1; p.ksplit = 1;
  if (maskH) { p.aux = {maskH->p, maskH->rs, maskH->cs}; p.aux_sr = maskH->pitch; }
  const int nz = R * nr.nz1;
  if (w0_slabs) *w0_slabs = 0;
  if (maskH && maskH->bits && bits_live.count(maskH->bits) && ncols == maskH->pitch && aligned16(maskH->p) && (maskH->pitch & 3) == 0) {
    p.aux_bits = maskH->bits; p.xb_s0 = maskH->brs; p.xb_s1 = maskH->bcs; p.xb_g = maskH->bg;
  }
  // weight-stationary fused kernel (csrc/ws_gemm.h): top-layer dgrad from mask bits + layer-0 weight gradient, nothing stored
  if (w0_X && w0_slabs && !store_dx && maskH && dy.rank1 && layer == 1 && col0 == 0 && !l.ens && !force_scalar &&
      ws_precision_ok() && p.aux_bits && dy.m.bits && bits_live.count(dy.m.bits) && out == dy.m.pitch &&
      ncols == in && (long)M * nz >= ws_bwd_min_rows) {
    WsDgradP w;
    memset(&w, 0, sizeof(w));
    w.abits = dy.m.bits; w.ab_s0 = dy.m.brs; w.ab_s1 = dy.m.bcs; w.ab_g = dy.m.bg;
    w.xbits = maskH->bits; w.xb_s0 = maskH->brs; w.xb_s1 = maskH->bcs; w.xb_g = maskH->bg;
    w.dq = dy.rowv.p; w.dq_s0 = dy.rowv.rs; w.dq_s1 = dy.rowv.cs; w.dq_sm = dy.rowv.pitch;
    w.wt = nr.base + l.w_off[l.L]; w.wt_s0 = nr.rs; w.wt_s1 = l.w_ms[l.L];
    w.W = nr.base + l.w_off[layer]; w.w_s0 = nr.rs; w.w_s1 = l.w_ms[layer]; w.w_sn = 1; w.w_sk = in;
    w.X = w0_X->p; w.x_s0 = w0_X->rs; w.x_s1 = w0_X->cs; w.x_pitch = w0_X->pitch; w.in0 = l.layer_in(0);
    float* g = grads + nr.g_off;
    w.w0_out = g + l.w_off[0]; w.b0_out = g + l.b_off[0];
    w.o_s0 = (long)max_slab * P_train; w.o_s1 = l.w_ms[0]; w.ob_s1 = l.b_ms[0]; w.o_ks = P_train; w.o_sr = l.layer_in(0); w.o_sc = 1;
    w.M = M; w.nz1 = nr.nz1; w.f32 = this->ws_f32(); w.gscale = cur_gscale;
    if (ws_dgrad_supported(w, out, in)) {
      const bool d3 = p3(2) && ws_dgrad3_supported(w, out, in);      // precision 2: three planes, two workgroups (column halves) per slab
      const int per_z = ws_dgrad_blocks(M, d3 ? 2 * nz : nz, max_slab, ws_geo);
      prof_begin(d3 ? (std::string(tag) + "@p3").c_str() : tag, 2.0 * M * (double)in * (out + l.layer_in(0) + 1) * nz,
                 nz * (4.0 * in * out + M * (double)(in + out) / 8 + 4.0 * M * (w0_X->pitch + 1) + 4.0 * per_z * in * (l.layer_in(0) + 1)));
      hipError_t err = d3 ? launch_ws_dgrad3_w0(w, nz, per_z, stream) : launch_ws_dgrad_w0(w, nz, per_z, stream);
      prof_end();
      if (err != hipSuccess) return fail(std::string("ws_dgrad launch ") + tag + ": " + hipGetErrorString(err));
      *w0_slabs = per_z;
      return 0;
    }
  }
  // same kernel, storing variant (no layer-0 gradient): e.g. the critic backward of the actor loss, where dz0 feeds dL/da
  if (!(w0_X && w0_slabs) && maskH && dy.rank1 && col0 == 0 && !force_scalar && ws_precision_ok() && p.aux_bits &&
      dy.m.bits && bits_live.count(dy.m.bits) && out == dy.m.pitch && ncols == in && dX.pitch >= in && (long)M * nz >= ws_bwd_min_rows) {
    WsDgradP w;
    memset(&w, 0, sizeof(w));
    w.abits = dy.m.bits; w.ab_s0 = dy.m.brs; w.ab_s1 = dy.m.bcs; w.ab_g = dy.m.bg;
    w.xbits = maskH->bits; w.xb_s0 = maskH->brs; w.xb_s1 = maskH->bcs; w.xb_g = maskH->bg;
    w.dq = dy.rowv.p; w.dq_s0 = dy.rowv.rs; w.dq_s1 = dy.rowv.cs; w.dq_sm = dy.rowv.pitch;
    w.wt = nr.base + l.w_off[l.L]; w.wt_s0 = nr.rs; w.wt_s1 = l.w_ms[l.L];
    w.W = nr.base + l.w_off[layer]; w.w_s0 = nr.rs; w.w_s1 = l.w_ms[layer];
    if (l.ens) { w.w_sn = out; w.w_sk = 1; } else { w.w_sn = 1; w.w_sk = in; }
    w.C = dX.p; w.c_s0 = dX.rs; w.c_s1 = dX.cs; w.c_pitch = dX.pitch;
    w.M = M; w.nz1 = nr.nz1; w.f32 = this->ws_f32();
    if (ws_dgrad_supported(w, out, in)) {
      const bool d3 = p3(2) && ws_dgrad3_supported(w, out, in);      // precision 2: three planes, two workgroups (column halves) per net
      const int per_z = ws_dgrad_blocks(M, d3 ? 2 * nz : nz, 1 << 20, ws_geo);
      prof_begin(d3 ? (std::string(tag) + "@p3").c_str() : tag, 2.0 * M * (double)in * out * nz, nz * (4.0 * in * out + M * (double)(in + out) / 8 + 4.0 * M * (in + 1)));
      hipError_t err = d3 ? launch_ws_dgrad3_w0(w, nz, per_z, stream) : launch_ws_dgrad_w0(w, nz, per_z, stream);
      prof_end();
      if (err != hipSuccess) return fail(std::string("ws_dgrad launch ") + tag + ": " + hipGetErrorString(err));
      return 0;
    }
  }
  // the same fused kernel fed with a MATERIALISED gradient (a three-layer net's middle layer): dz0 = 1[h0 > 0] (dz1 W1) never leaves the
  // registers, dW0 / db0 come out as one slab per workgroup
  if (w0_X && w0_slabs && !store_dx && maskH && !dy.rank1 && layer == 1 && col0 == 0 && !force_scalar && ws_precision_ok() &&
      p.aux_bits && dy.m.pitch == out && ncols == in && (long)M * nz >= ws_dgrad_plain_min_rows) {
    WsDgradP w;
    memset(&w, 0, sizeof(w));
    w.Z = dy.m.p; w.z_s0 = dy.m.rs; w.z_s1 = dy.m.cs; w.z_pitch = dy.m.pitch;
    w.xbits = maskH->bits; w.xb_s0 = maskH->brs; w.xb_s1 = maskH->bcs; w.xb_g = maskH->bg;
    w.W = nr.base + l.w_off[layer]; w.w_s0 = nr.rs; w.w_s1 = l.w_ms[layer];
    w.X = w0_X->p; w.x_s0 = w0_X->rs; w.x_s1 = w0_X->cs; w.x_pitch = w0_X->pitch; w.in0 = l.layer_in(0);
    float* g = grads + nr.g_off;
    w.w0_out = g + l.w_off[0]; w.b0_out = g + l.b_off[0];
    w.o_s0 = (long)max_slab * P_train; w.o_s1 = l.w_ms[0]; w.ob_s1 = l.b_ms[0]; w.o_ks = P_train;
    // nn.Linear keeps (out, in)-major weights, EnsembleLinear (in, out)-major ones: the same holds for the gradient slabs
    if (l.ens) { w.w_sn = out; w.w_sk = 1; w.o_sr = 1; w.o_sc = l.layer_out(0); }
    else { w.w_sn = 1; w.w_sk = in; w.o_sr = l.layer_in(0); w.o_sc = 1; }
    w.M = M; w.nz1 = nr.nz1; w.f32 = this->ws_f32(); w.gscale = cur_gscale;
    if (ws_dgrad_supported(w, out, in)) {
      const bool d3 = p3(2) && ws_dgrad3_supported(w, out, in);      // precision 2: three planes of dz1 and of W1
      const int per_z = ws_dgrad_blocks(M, d3 ? 2 * nz : nz, max_slab, ws_geo);
      prof_begin(d3 ? (std::string(tag) + "@p3").c_str() : tag, 2.0 * M * (double)in * (out + l.layer_in(0) + 1) * nz,
                 nz * (4.0 * in * out + 4.0 * M * (double)out + M * (double)in / 8 + 4.0 * M * (w0_X->pitch + 1) + 4.0 * per_z * in * (l.layer_in(0) + 1)));
      hipError_t err = d3 ? launch_ws_dgrad3_w0(w, nz, per_z, stream) : launch_ws_dgrad_w0(w, nz, per_z, stream);
      prof_end();
      if (err != hipSuccess) return fail(std::string("ws_dgrad launch ") + tag + ": " + hipGetErrorString(err));
      *w0_slabs = per_z;
      return 0;
    }
  }
  if (w0_X && w0_slabs && maskH && layer == 1 && col0 == 0 && !l.ens && !force_scalar) {
    // fuse the layer-0 weight / bias gradient into this launch's epilogue (one slab per row tile)
    const int slabs = w0_fused_slabs(p, nz, l.layer_in(0), w0_X->pitch, w0_X->p, w0_X->rs, w0_X->cs, max_slab);
    if (slabs > 0) {
      float* g = grads + nr.g_off;
      p.w0_x = {w0_X->p, w0_X->rs, w0_X->cs}; p.w0_xsr = w0_X->pitch; p.w0_in = l.layer_in(0);
      p.w0_out = g + l.w_off[0]; p.w0_bias = g + l.b_off[0];
      p.w0_s0 = (long)max_slab * P_train; p.w0_s1 = l.w_ms[0]; p.w0_bs1 = l.b_ms[0]; p.w0_ks = P_train; p.w0_sr = l.layer_in(0);
      if (!store_dx) p.C = nullptr;
      *w0_slabs = slabs;
    }
  }
  // plain (materialised) dz through a 256 x 256 layer with the ReLU mask of the receiving activation: the weight-stationary kernel
  // in gradient mode (B = the weights viewed transposed, epilogue = mask from bits)
  if (!dy.rank1 && !p.w0_out && maskH && p.aux_bits && col0 == 0 && ncols == in && ws_precision_ok() && !force_scalar &&
      dy.m.pitch == out && dX.pitch == in && (long)M * nz >= ws_bwd_min_rows) {
    WsFwdP w;
    memset(&w, 0, sizeof(w));
    w.X = dy.m.p; w.x_s0 = dy.m.rs; w.x_s1 = dy.m.cs; w.x_pitch = dy.m.pitch;
    w.W = nr.base + l.w_off[layer]; w.w_s0 = nr.rs; w.w_s1 = l.w_ms[layer];
    // B[n = input unit][k = output unit]: nn.Linear W (out, in) -> element (k, n) at k * in + n; EnsembleLinear (in, out) -> n * out + k
    if (l.ens) { w.w_sn = out; w.w_sk = 1; } else { w.w_sn = 1; w.w_sk = in; }
    w.Y = dX.p; w.y_s0 = dX.rs; w.y_s1 = dX.cs; w.y_pitch = dX.pitch;
    w.dmask = maskH->bits; w.dm_s0 = maskH->brs; w.dm_s1 = maskH->bcs; w.dm_g = maskH->bg;
    w.M = M; w.nz1 = nr.nz1; w.f32 = this->ws_f32(); w.gscale = cur_gscale;
    if (ws_fwd_supported(w, out, in)) {
      const bool f3 = p3(2) && ws_fwd3_supported(w, out, in);       // precision 2: three planes of dz and of the weights (column halves)
      prof_begin(f3 ? (std::string(tag) + "@p3").c_str() : tag, 2.0 * M * (double)in * out * nz, nz * (4.0 * M * out + 4.0 * in * out + 4.0 * M * in + M * (double)in / 8));
      hipError_t err = f3 ? launch_ws_fwd3(w, nz, ws_blocks_per_problem(M / WS_ROWS, 2 * nz, 10, 1 << 20, ws_geo), stream) : launch_ws_fwd(w, nz, stream, ws_geo);
      prof_end();
      if (err != hipSuccess) return fail(std::string("ws dgrad launch ") + tag + ": " + hipGetErrorString(err));
      return 0;
    }
  }
  if (dy.rank1) {
    if (maskH && dy.m.bits && bits_live.count(dy.m.bits) && out == dy.m.pitch) {
      // the virtual dz of the top hidden layer from its packed ReLU mask: the activation matrix is not read at all
      GemmP q = p;
      q.a_bits = dy.m.bits; q.ab_s0 = dy.m.brs; q.ab_s1 = dy.m.bcs; q.ab_g = dy.m.bg;
      const int tcfg = pick_cfg(q.M, q.N, q.K, nz);
      if (rank1_bits_supported(tcfg, q, force_scalar)) {
        const double flops = 2.0 * q.M * (double)q.N * q.K * nz + (q.w0_out ? 2.0 * q.M * (double)q.N * (q.w0_in + 1) * nz : 0.0);
        const double bytes = nz * (4.0 * q.N * q.K + q.M * (double)q.K / 8 + (q.C ? 4.0 * q.M * q.N : 0.0) +
                                   (q.aux_bits ? q.M * (double)q.N / 8 : 4.0 * q.M * q.N) + (q.w0_out ? 4.0 * q.M * q.w0_xsr : 0.0));
        prof_begin(tag, flops, bytes);
        hipError_t err = launch_gemm_rank1_bits<E_MASK>(tcfg, q, nz, stream, this->mm_prec());
        prof_end();
        if (err != hipSuccess) return fail(std::string("gemm launch ") + tag + ": " + hipGetErrorString(err));
        return 0;
      }
    }
    if (vals_dead.count(dy.m.p)) return fail(std::string("dgrad ") + tag + ": the activation values were not stored by the forward pass");
    if (maskH) return run_gemm<PA_RANK1, PB_PLAIN, E_MASK>(this, CFG_AUTO, p, nz, tag);
    return run_gemm<PA_RANK1, PB_PLAIN, E_PLAIN>(this, CFG_AUTO, p, nz, tag);
  }
  if (maskH) return run_gemm<PA_PLAIN, PB_PLAIN, E_MASK>(this, CFG_AUTO, p, nz, tag);
  return run_gemm<PA_PLAIN, PB_PLAIN, E_PLAIN>(this, CFG_AUTO, p, nz, tag);
}

// split-K factor for a weight gradient: enough workgroups to fill 256 CUs, chunk aligned
static int wgrad_ksplit(int Mout, int Nout, int Krows, int nz, int cap) {
  const int cfg = pick_cfg(Mout, Nout, Krows, nz);
  int TM, TN, TK;
  switch (cfg) {
    case CFG_SQ: TM = CfgSq::TM; TN = CfgSq::TN; TK = CfgSq::kTK; break;
    case CFG_WG: TM = CfgWg::TM; TN = CfgWg::TN; TK = CfgWg::kTK; break;
    case CFG_BIG: TM = CfgBig::TM; TN = CfgBig::TN; TK = CfgBig::kTK; break;
    case CFG_MID: TM = CfgMid::TM; TN = CfgMid::TN; TK = CfgMid::kTK; break;
    case CFG_SMALL: TM = CfgSmall::TM; TN = CfgSmall::TN; TK = CfgSmall::kTK; break;
    default: TM = CfgTall::TM; TN = CfgTall::TN; TK = CfgTall::kTK; break;
  }
  const int tiles = ((Mout + TM - 1) / TM) * ((Nout + TN - 1) / TN) * nz;
  const int kchunks = (Krows + TK - 1) / TK;
  static const int target = [] { const char* f = getenv("ORL_WGRAD_WG_TARGET"); return (f && atoi(f) > 0) ? atoi(f) : 512; }();
  static const int long_min = [] { const char* f = getenv("ORL_WGRAD_LONG_MIN"); return (f && atoi(f) > 0) ? atoi(f) : 4; }();
  int ks = (target + tiles - 1) / tiles;
  // many batched nets fill the CUs without split-K, but workgroups that all stream thousands of rows from the same offset of
  // equally strided matrices run 2x slower (measured at 256 nets x 7936 rows, fp32: 12.8 -> 5.4 ms): keep >= 4 k-ranges
  if (Krows >= 4096) ks = std::max(ks, long_min);
  static const int small_ks = [] { const char* f = getenv("ORL_WGRAD_SMALL_KS"); return (f && atoi(f) > 0) ? atoi(f) : 1; }();
  // 256-row wgrads of many nets: round 2 measured two k-ranges 1.5x faster than one -- with the item-major workgroup mapping, where the four
  // tiles of a net sat on four XCDs.  With the z-major mapping (gemm_kernel.h) one range wins: EDAC 15.4k -> 16.9k steps/s, IQL +2.5 %,
  // TD3+BC +2.5 % at 128 runs (Adam reads half the slabs).  ORL_WGRAD_SMALL_KS=2 restores the old rule for A/B runs.
  if (cfg == CFG_SQ && Krows < 1024 && kchunks >= 8) ks = std::max(ks, small_ks);
  ks = std::max(1, std::min(ks, std::min(cap, kchunks)));
  while (ks > 1 && (kchunks + ks - 1) / ks < 2) --ks;
  return ks;
}

// slabs_out: number of split-K slabs actually written (== ksplit unless the weight-stationary kernel chose its own decomposition)
int Engine::linear_wgrad(const DY& dy, const Mat& X, int M, const NetRef& nr, int layer, int ksplit, int slab0, bool with_bias,
                         const char* tag, int in_row0, int in_rows, bool* fuse_tail, int* slabs_out, const float* x_dscale, const Mat* recompute_X0) {
  const NetLayout& l = *nr.lay;
  const int in = l.layer_in(layer), out = l.layer_out(layer);
  if (in_rows < 0) in_rows = in;
  if (slab0 + ksplit > max_slab) return fail("wgrad: slab budget exceeded");
  const bool x_dead = vals_dead.count(X.p) > 0;
  if (x_dead && !(recompute_X0 && layer == 1)) return fail(std::string("wgrad ") + tag + ": the input activation was not stored by the forward pass");
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.a_dscale = cur_gscale; p.b_dscale = x_dscale;      // split precision: A = dY^T of the current backward pass; B = X (an activation, or a gradient-like matrix with its own scale)
  if (mm_prec() == P_SPLIT3) { p.a_scale = ORL_GSCALE3; if (x_dscale) p.b_scale = ORL_GSCALE3; }
  p.A = {dy.m.p, dy.m.rs, dy.m.cs};
  p.a_sr = 1; p.a_sk = dy.m.pitch;
  if (out == 1 && dy.m.pitch == 1 && !dy.rank1) p.a_sr = 4;   // a [1 x M] row vector: k-contiguous, any row stride -> vector loads
  if (dy.rank1) {
    p.a_trans = 1;
    p.rowv = {dy.rowv.p, dy.rowv.rs, dy.rowv.cs};
    p.colv = {nr.base + l.w_off[l.L], nr.rs, l.w_ms[l.L]};
  }
  p.B = {X.p, X.rs, X.cs};
  p.b_sr = 1; p.b_sk = X.pitch;
  p.a_rlim = dy.m.pitch & ~3; p.b_rlim = X.pitch & ~3;
  p.ones_row = 1 << 30;
  p.M = out; p.N = in_rows; p.K = M;
  p.nz1 = nr.nz1; p.ksplit = ksplit;
  const long g_rs = (long)max_slab * P_train;
  float* g = grads + nr.g_off + (long)slab0 * P_train;
  if (l.ens) { p.C = g + l.w_off[layer] + (long)in_row0 * out; p.c_sr = 1; p.c_sn = out; }
  else { p.C = g + l.w_off[layer] + in_row0; p.c_sr = in; p.c_sn = 1; }
  p.c_s0 = g_rs; p.c_s1 = l.w_ms[layer]; p.c_ks = P_train;
  if (with_bias) { p.bias_out = g + l.b_off[layer]; p.bo_s0 = g_rs; p.bo_s1 = l.b_ms[layer]; p.bo_ks = P_train; }
  const int nz = R * nr.nz1;
  if (slabs_out) *slabs_out = ksplit;
  // output-stationary kernel (csrc/ws_gemm.h): dz^T from the packed ReLU mask, G = dq (.) X and the activation itself streamed once;
  // the tail layer's gradients ride along (*fuse_tail = true, same slabs)
  if (slabs_out && fuse_tail && dy.rank1 && with_bias && slab0 == 0 && ws_precision_ok() && !force_scalar && !l.ens &&
      dy.m.bits && bits_live.count(dy.m.bits) && out == dy.m.pitch && in_row0 == 0 && in_rows == in && X.pitch == in &&
      ws_wgrad_rows_ok(M, nz)) {
    const bool derived = vals_dead.count(dy.m.p) > 0;
    WsWgradP w;
    memset(&w, 0, sizeof(w));
    w.abits = dy.m.bits; w.ab_s0 = dy.m.brs; w.ab_s1 = dy.m.bcs; w.ab_g = dy.m.bg;
    w.dq = dy.rowv.p; w.dq_s0 = dy.rowv.rs; w.dq_s1 = dy.rowv.cs; w.dq_sm = dy.rowv.pitch;
    w.H0 = X.p; w.h0_s0 = X.rs; w.h0_s1 = X.cs; w.h0_pitch = X.pitch;
    w.wt = nr.base + l.w_off[l.L]; w.wt_s0 = nr.rs; w.wt_s1 = l.w_ms[l.L];
    w.dW = g + l.w_off[layer]; w.db = g + l.b_off[layer];
    w.o_s0 = g_rs; w.o_s1w = l.w_ms[layer]; w.o_s1b = l.b_ms[layer]; w.o_ks = P_train;
    // the activation values are streamed through registers as well: tail-layer gradients (and db) from the same pass, same slabs
    // (or, when the forward pass did not store h1, derived from the accumulators: see WsWgradP)
    if (derived) {
      w.W1 = nr.base + l.w_off[layer]; w.w1_s0 = nr.rs; w.w1_s1 = l.w_ms[layer];
      w.b1 = nr.base + l.b_off[layer]; w.b1_s0 = nr.rs; w.b1_s1 = l.b_ms[layer];
    } else { w.H1 = dy.m.p; w.h1_s0 = dy.m.rs; w.h1_s1 = dy.m.cs; w.h1_pitch = dy.m.pitch; }
    w.dwt = g + l.w_off[l.L]; w.dbt = g + l.b_off[l.L]; w.o_s1wt = l.w_ms[l.L]; w.o_s1bt = l.b_ms[l.L];
    w.M = M; w.nz1 = nr.nz1; w.f32 = this->ws_f32(); w.gscale = cur_gscale;
    w.np3 = p3(4) && derived;                                        // precision 2: three planes of G = dq (.) h0 (ws_wgrad_kernel<5>)
    if (ws_wgrad_supported(w, out, in)) {
      const int per_z = ws_dgrad_blocks(M, nz, ws_wgrad_slab_cap, ws_geo, 1 << 20);      // one round: the slab write + derived tail gradients per workgroup cost more than idle CUs (4 slabs at 192 nets: 540 us either way, and Adam then reads 4 slabs)
      prof_begin(w.np3 ? (std::string(tag) + "@p3").c_str() : tag, 2.0 * M * (double)in * (out + 2) * nz,
                 nz * ((derived ? M * (double)out / 8 : 4.0 * M * (double)out) + 4.0 * M * (in + 1) + 4.0 * per_z * out * (in + 2)));
      w.lab_clk = (unsigned long long*)(aloss_part + (long)R * SB_MAXGROUPS * 2) + 72;
      hipError_t err = launch_ws_wgrad(w, nz, per_z, stream);
      prof_end();
      if (err != hipSuccess) return fail(std::string("ws_wgrad launch ") + tag + ": " + hipGetErrorString(err));
      *fuse_tail = true;
      *slabs_out = per_z;
      return 0;
    }
  }
  // the same output-stationary kernel for a hidden layer BELOW the top one of a many-row batch: dZ is a materialised matrix (three products
  // per block instead of the rank-1 form's two, no mask / w_tail); one slab per workgroup
  if (slabs_out && !dy.rank1 && with_bias && slab0 == 0 && ws_precision_ok() && !force_scalar && !l.ens && !x_dscale && in_row0 == 0 &&
      in_rows == in && X.pitch == in && dy.m.pitch == out && ws_wgrad_rows_ok(M, nz)) {
    WsWgradP w;
    memset(&w, 0, sizeof(w));
    w.dZ = dy.m.p; w.dz_s0 = dy.m.rs; w.dz_s1 = dy.m.cs; w.dz_pitch = dy.m.pitch;
    w.H0 = X.p; w.h0_s0 = X.rs; w.h0_s1 = X.cs; w.h0_pitch = X.pitch;
    if (x_dead) {                                  // h0 = relu(X0 W0^T + b0) is rebuilt inside the launch
      w.X0 = recompute_X0->p; w.x0_s0 = recompute_X0->rs; w.x0_s1 = recompute_X0->cs; w.x0_pitch = recompute_X0->pitch; w.in0 = l.layer_in(0);
      w.W0 = nr.base + l.w_off[0]; w.w0_s0 = nr.rs; w.w0_s1 = l.w_ms[0]; w.w0_sn = l.layer_in(0); w.w0_sk = 1;
      w.b0 = nr.base + l.b_off[0]; w.b0_s0 = nr.rs; w.b0_s1 = l.b_ms[0];
    }
    w.dW = g + l.w_off[layer]; w.db = g + l.b_off[layer];
    w.o_s0 = g_rs; w.o_s1w = l.w_ms[layer]; w.o_s1b = l.b_ms[layer]; w.o_ks = P_train;
    w.M = M; w.nz1 = nr.nz1; w.f32 = this->ws_f32(); w.gscale = cur_gscale;
    if (ws_wgrad_supported(w, out, in)) {
      const bool w3 = p3(4) && ws_wgrad3p_supported(w, out, in);      // precision 2: three planes of dZ and of H0, two workgroups (halves of the output rows) per slab
      const int per_z = ws_dgrad_blocks(M, w3 ? 2 * nz : nz, max_slab, ws_geo, 1 << 20);
      prof_begin(w3 ? (std::string(tag) + "@p3").c_str() : tag, 2.0 * M * (double)in * (out + 1 + (x_dead ? w.in0 + 1 : 0)) * nz,
                 nz * (4.0 * M * (double)((x_dead ? w.x0_pitch : in) + out) + 4.0 * per_z * out * (in + 1)));
      hipError_t err = w3 ? launch_ws_wgrad3p(w, nz, per_z, stream) : launch_ws_wgrad(w, nz, per_z, stream);
      prof_end();
      if (err != hipSuccess) return fail(std::string("ws_wgrad launch ") + tag + ": " + hipGetErrorString(err));
      *slabs_out = per_z;
      return 0;
    }
  }
  if (x_dead) return fail(std::string("wgrad ") + tag + ": the input activation was not stored and the recomputing kernel does not serve this shape");
  if (dy.rank1 && vals_dead.count(dy.m.p)) return fail(std::string("wgrad ") + tag + ": the activation values were not stored by the forward pass");
  if (fuse_tail) {
    // the rank-1 kernel streams h and dq anyway: let it also emit the tail layer's dw / db (same split-K slabs)
    *fuse_tail = dy.rank1 && rank1_wgrad_is_fast(p, force_scalar);
    if (*fuse_tail) {
      p.tail_w_out = g + l.w_off[l.L]; p.tail_b_out = g + l.b_off[l.L];
      p.tw_s0 = g_rs; p.tw_s1 = l.w_ms[l.L]; p.tb_s1 = l.b_ms[l.L];
    }
  }
  if (dy.rank1) return run_gemm<PA_RANK1, PB_PLAIN, E_WGRAD>(this, CFG_AUTO, p, nz, tag);
  return run_gemm<PA_PLAIN, PB_PLAIN, E_WGRAD>(this, CFG_AUTO, p, nz, tag);
}

int Engine::adam(int net, int nnets, int lr_slot, const std::vector<std::pair<long, int>>& segs, int target_net, unsigned long long t_div) {
  AdamP a;
  memset(&a, 0, sizeof(a));
  const NetLayout& l = lay[net];
  a.params = net_ptr(0, net); a.p_s0 = P_train; a.p_s1 = l.stride();
  a.m = adam_m + net_off[net]; a.v = adam_v + net_off[net];
  a.g = grads + net_off[net]; a.g_s0 = (long)max_slab * P_train; a.g_s1 = l.stride(); a.g_ks = P_train;
  a.nseg = (int)segs.size();
  if (a.nseg > 12) return fail("too many adam segments");
  for (int i = 0; i < nnets && net + i < ORL_NUM_NETS; ++i) last_segs[net + i] = segs;
  for (int i = 0; i < a.nseg; ++i) { a.seg_end[i] = segs[i].first; a.seg_nslab[i] = segs[i].second; }
  if (target_net >= 0) { a.target = net_ptr(0, target_net); a.t_s0 = P_tgt; a.t_s1 = l.stride(); }
  a.P = l.size; a.lr_slot = lr_slot; a.hy = hyper;
  a.b1 = cfg.adam_beta1; a.b2 = cfg.adam_beta2; a.eps = cfg.adam_eps; a.tau = cfg.tau;
  a.gstep = gstep; a.t_div = t_div; a.health = health;
  // algorithmic bytes: every gradient slab read once, m / v / parameter read and written, the Polyak target read and written
  double bytes = 0.0;
  for (int i = 0; i < a.nseg; ++i) bytes += (double)(a.seg_end[i] - (i ? a.seg_end[i - 1] : 0)) * (4.0 * a.seg_nslab[i] + 24.0 + (a.target ? 8.0 : 0.0));
  prof_begin("adam", 0, bytes * nnets * R);
  hipLaunchKernelGGL(k_adam, dim3((unsigned)((l.size + 1023) / 1024), nnets, R), dim3(256), 0, stream, a);    // four parameters per thread
  prof_end();
  if (hipGetLastError() != hipSuccess) return fail("adam: launch failed");
  return 0;
}

int Engine::polyak(int target_net, int src_net, int nnets) {
  const long P = lay[src_net].size, PS = lay[src_net].stride();
  prof_begin("polyak", 0, 12.0 * P * nnets * R);
  hipLaunchKernelGGL(k_polyak, dim3((unsigned)((P + 255) / 256), nnets, R), dim3(256), 0, stream, net_ptr(0, target_net), P_tgt, PS,
                     (const float*)net_ptr(0, src_net), P_train, PS, P, cfg.tau);
  prof_end();
  if (hipGetLastError() != hipSuccess) return fail("polyak: launch failed");
  return 0;
}

int Engine::assemble(const Mat& obs, const Mat* act, const Mat& X, int row0, int rows, int rep) {
  AssembleP a;
  memset(&a, 0, sizeof(a));
  a.obs = obs.p; a.obs_rs = obs.rs; a.OP = obs.pitch; a.od = od;
  if (act) { a.act = act->p; a.act_rs = act->rs; a.apitch = act->pitch; }
  a.ad = ad;
  a.X = X.p; a.x_rs = X.rs; a.XP = X.pitch; a.row0 = row0; a.rows = rows; a.rep = rep;
  ORL_LAUNCH("assemble", k_assemble, dim3((rows + 255) / 256, R), dim3(256), a);
  return 0;
}

int Engine::mlp_forward(const Mat& X, int M, const NetRef& nr, std::vector<Mat>& hs, const Mat& out, const char* tag,
                        const SampleJob* jobs, int njobs, bool* jobs_done) {
  const NetLayout& l = *nr.lay;
  const int Ln = l.L;
  std::string t = tag;
  if (jobs_done) *jobs_done = false;
  // few batched rows (one to a few runs per engine): the whole pass as ONE launch (small_fwd.h) instead of layer 0 + layer 1 + tail
  if (small_fwd_on && Ln == 2 && !l.ens && !no_ws && !force_scalar && l.H[0] == SF_N && l.H[1] == SF_N && hs[0].pitch == SF_N && hs[1].pitch == SF_N &&
      (long)M * R * nr.nz1 <= small_fwd_max_rows) {
    SmallFwdP w;
    memset(&w, 0, sizeof(w));
    w.X = X.p; w.x_s0 = X.rs; w.x_s1 = X.cs; w.x_pitch = X.pitch; w.in0 = l.layer_in(0);
    w.W0 = nr.base + l.w_off[0]; w.w0_s0 = nr.rs; w.w0_s1 = l.w_ms[0];
    w.b0 = nr.base + l.b_off[0]; w.b0_s0 = nr.rs; w.b0_s1 = l.b_ms[0];
    w.W1 = nr.base + l.w_off[1]; w.w1_s0 = nr.rs; w.w1_s1 = l.w_ms[1];
    w.b1 = nr.base + l.b_off[1]; w.b1_s0 = nr.rs; w.b1_s1 = l.b_ms[1];
    w.Wt = nr.base + l.w_off[2]; w.wt_s0 = nr.rs; w.wt_s1 = l.w_ms[2];
    w.bt = nr.base + l.b_off[2]; w.bt_s0 = nr.rs; w.bt_s1 = l.b_ms[2];
    if (!fwd_only) {
      w.H0 = hs[0].p; w.h0_s0 = hs[0].rs; w.h0_s1 = hs[0].cs;
      w.H1 = hs[1].p; w.h1_s0 = hs[1].rs; w.h1_s1 = hs[1].cs;
    }
    w.OUT = out.p; w.o_s0 = out.rs; w.o_s1 = out.cs; w.o_pitch = out.pitch; w.out_dim = l.out_dim;
    w.M = M; w.nz1 = nr.nz1; w.f32 = ws_f32();
    if (fuse_small && jobs && jobs_done && njobs >= 1 && njobs <= 3 && nr.nz1 == 1 && l.out_dim == 2 * ad && ad <= 8 && out.pitch == l.out_dim) {
      w.njobs = njobs; w.A = ad;
      for (int i = 0; i < njobs; ++i) w.job[i] = jobs[i];
      if (!small_fwd_supported(w)) w.njobs = 0;      // (e.g. more than 16 repeated actions per row: the pass stays one launch, the sampling its own)
    }
    if (small_fwd_supported(w)) {
      const int nz = R * nr.nz1;
      if (lab_slot < 4) w.lab_clk = (unsigned long long*)(aloss_part + (long)R * SB_MAXGROUPS * 2) + 16 + 12 * lab_slot++;
      watch_range(X, M, w.in0, X.cs ? nr.nz1 : 1, tag);
      if (!fwd_only) watch_range(hs[0], M, SF_N, nr.nz1, tag);      // (h1 meets the tail weights in fp32 vector arithmetic)
      prof_begin(tag, 2.0 * M * (double)nz * (SF_N * (double)(w.in0 + 1) + (double)SF_N * SF_N + (double)SF_N * l.out_dim),
                 4.0 * nz * (M * (double)(X.pitch + (fwd_only ? 0 : 2 * SF_N) + l.out_dim) + (double)SF_N * (w.in0 + SF_N + l.out_dim + 2)));
      hipError_t err = launch_small_fwd(w, nz, stream);
      prof_end();
      if (err != hipSuccess) return fail(std::string("small_fwd launch ") + tag + ": " + hipGetErrorString(err));
      if (w.njobs) *jobs_done = true;
      for (int i = 0; i < 2; ++i) {
        if (hs[i].bits) bits_live.erase(hs[i].bits);      // no packed masks from this path: the backward reads 1[h > 0] from the values
        if (fwd_only) vals_dead.insert(hs[i].p); else vals_dead.erase(hs[i].p);
      }
      return 0;
    }
  }
  bool tail_done = false;
  for (int i = 0; i < Ln; ++i) {
    if (i == 0 && Ln >= 2) continue;         // layer 0 is issued together with layer 1 (fused into it when the ws kernel applies)
    if (linear_fwd(i == 0 ? X : hs[i - 1], M, nr, i, hs[i], E_BIAS_RELU, nullptr, (t + ".fwd" + std::to_string(i)).c_str(), 0, -1,
                   i == Ln - 1 ? &out : nullptr, i == Ln - 1 ? &tail_done : nullptr, i == 1 ? &X : nullptr, (t + ".fwd0").c_str())) return -1;
  }
  if (tail_done) return 0;                 // single-output tail folded into the last hidden layer's epilogue
  return linear_fwd(hs[Ln - 1], M, nr, Ln, out, E_BIAS, nullptr, (t + ".tail").c_str());
}

int Engine::mlp_qgrad(const Mat& X, int M, const NetRef& nr, const Mat& q, const Mat& G, int gc0, int gn, const char* tag, bool* done) {
  const NetLayout& l = *nr.lay;
  *done = false;
  if (!(fuse_small && small_fwd_on && l.L == 2 && !l.ens && !no_ws && !force_scalar && l.H[0] == SF_N && l.H[1] == SF_N && l.out_dim == 1 &&
        (long)M * R * nr.nz1 <= small_fwd_max_rows)) return 0;
  SmallFwdP w;
  memset(&w, 0, sizeof(w));
  w.X = X.p; w.x_s0 = X.rs; w.x_s1 = X.cs; w.x_pitch = X.pitch; w.in0 = l.layer_in(0);
  w.W0 = nr.base + l.w_off[0]; w.w0_s0 = nr.rs; w.w0_s1 = l.w_ms[0];
  w.b0 = nr.base + l.b_off[0]; w.b0_s0 = nr.rs; w.b0_s1 = l.b_ms[0];
  w.W1 = nr.base + l.w_off[1]; w.w1_s0 = nr.rs; w.w1_s1 = l.w_ms[1];
  w.b1 = nr.base + l.b_off[1]; w.b1_s0 = nr.rs; w.b1_s1 = l.b_ms[1];
  w.Wt = nr.base + l.w_off[2]; w.wt_s0 = nr.rs; w.wt_s1 = l.w_ms[2];
  w.bt = nr.base + l.b_off[2]; w.bt_s0 = nr.rs; w.bt_s1 = l.b_ms[2];
  w.OUT = q.p; w.o_s0 = q.rs; w.o_s1 = q.cs; w.o_pitch = q.pitch; w.out_dim = 1;
  w.G = G.p; w.g_s0 = G.rs; w.g_s1 = G.cs; w.g_pitch = G.pitch; w.gc0 = gc0; w.gn = gn;
  w.M = M; w.nz1 = nr.nz1; w.f32 = ws_f32();
  if (!small_fwd_supported(w)) return 0;
  const int nz = R * nr.nz1;
  if (lab_slot < 4) w.lab_clk = (unsigned long long*)(aloss_part + (long)R * SB_MAXGROUPS * 2) + 16 + 12 * lab_slot++;
  watch_range(X, M, w.in0, X.cs ? nr.nz1 : 1, tag);
  // forward (layer 0, layer 1, tail) + the unit-seed backward through layer 1 and the gn input columns of layer 0
  prof_begin(tag, 2.0 * M * (double)nz * (SF_N * (double)(w.in0 + 1) + 2.0 * SF_N * SF_N + SF_N + (double)SF_N * gn),
             4.0 * nz * (M * (double)(X.pitch + 1 + gn) + 2.0 * SF_N * SF_N + (double)SF_N * (w.in0 + 3)));
  hipError_t err = launch_small_fwd(w, nz, stream);
  prof_end();
  if (err != hipSuccess) return fail(std::string("small_qgrad launch ") + tag + ": " + hipGetErrorString(err));
  *done = true;
  return 0;
}

int Engine::mlp_forward_only(const Mat& X, int M, const NetRef& nr, std::vector<Mat>& hs, const Mat& out, const char* tag,
                             const SampleJob* jobs, int njobs, bool* jobs_done) {
  struct Scope { Engine* e; ~Scope() { e->fwd_only = false; } } scope{this};
  fwd_only = true;
  return mlp_forward(X, M, nr, hs, out, tag, jobs, njobs, jobs_done);
}

int Engine::scale_inplace(const Mat& m, int rows, int cols, int nets, float s, const Mat* mask, const char* tag) {
  ORL_LAUNCH(tag, k_dropout, dim3((unsigned)(((long)rows * cols + 255) / 256), nets, R), dim3(256), m.p, m.rs, m.cs, m.pitch,
             (const float*)(mask ? mask->p : nullptr), mask ? mask->rs : 0L, rows, cols, s);
  return 0;
}

int Engine::mlp_forward_dropout(const Mat& X, int M, const NetRef& nr, std::vector<Mat>& hs, const Mat& out, const char* tag, float p,
                                const std::vector<Mat>& masks) {
  const NetLayout& l = *nr.lay;
  const std::string t = tag;
  const float inv_keep = 1.0f / (1.0f - p);
  struct Scope { Engine* e; ~Scope() { e->no_ws = false; } } scope{this};
  no_ws = true;                                            // one launch per layer: the dropout sits between them
  for (int i = 0; i < l.L; ++i) {
    if (linear_fwd(i == 0 ? X : hs[i - 1], M, nr, i, hs[i], E_BIAS_RELU, nullptr, (t + ".fwd" + std::to_string(i)).c_str())) return -1;
    if (scale_inplace(hs[i], M, l.layer_out(i), nr.nz1, inv_keep, &masks[i], (t + ".dropout" + std::to_string(i)).c_str())) return -1;
    if (hs[i].bits) bits_live.erase(hs[i].bits);           // the packed ReLU mask predates the dropout: the backward reads 1[h > 0] from the values
  }
  return linear_fwd(hs[l.L - 1], M, nr, l.L, out, E_BIAS, nullptr, (t + ".tail").c_str());
}

// Adam segment table: one (end offset, slab count) pair per weight tensor and per bias tensor
static std::vector<std::pair<long, int>> make_segs(const NetLayout& l, const std::vector<int>& ksW, const std::vector<int>& ksB) {
  std::vector<std::pair<long, int>> s;
  for (int i = 0; i <= l.L; ++i) {
    s.push_back({l.b_off[i], ksW[i]});   // weights of layer i end where its bias starts
    const long bend = l.b_off[i] + (l.ens ? (long)l.members * l.layer_out(i) : l.layer_out(i));
    s.push_back({bend, ksB[i]});
  }
  if (l.extra_off >= 0) s.push_back({l.size, 1});
  return s;
}

// generic backward through an MLP family.  dTail: [M x out_dim] gradient w.r.t. the tail output.
struct BwdOut { std::vector<int> ks; };
static int mlp_backward(Engine* e, const NetRef& nr, const Mat& X, const std::vector<Mat>& hs, int M, const Mat& dTail,
                        std::vector<Mat>& dz, bool want_w, bool want_dx, int dx_col0, int dx_ncols, const Mat* dX,
                        const char* tag, BwdOut* out, const float* gscale_given = nullptr) {
  const NetLayout& l = *nr.lay;
  const int L = l.L;
  const bool rank1 = (l.out_dim == 1);
  const int nz = e->R * nr.nz1;
  std::vector<int> ks(L + 1, 1);
  std::string t = tag;
  // split precision: every gradient matrix of this pass enters the MFMAs times one dynamic power-of-two scale per run, chosen from the seed
  struct ScaleScope { Engine* e; const float* prev; ~ScaleScope() { e->cur_gscale = prev; } } scope{e, e->cur_gscale};
  e->cur_gscale = gscale_given ? gscale_given : e->grad_scale(dTail, M, l.out_dim, nr.nz1, tag);
  if (e->split_scales() && !e->cur_gscale) return -1;
  if (want_w) {
    for (int i = 0; i <= L; ++i) ks[i] = wgrad_ksplit(l.layer_out(i), l.layer_in(i), M, nz, e->ksplit_cap);
    if (!rank1 && e->linear_wgrad(DY::plain(dTail), hs[L - 1], M, nr, L, ks[L], 0, true, (t + ".wgrad_tail").c_str())) return -1;
  }
  DY cur;
  bool w0_done = false;
  if (rank1) cur = DY::virt(hs[L - 1], dTail);
  else {
    if (e->linear_dgrad(DY::plain(dTail), M, nr, L, 0, l.layer_in(L), &hs[L - 1], dz[L - 1], (t + ".dgrad_tail").c_str())) return -1;
    if (e->bwd_scale != 1.0f && e->scale_inplace(dz[L - 1], M, l.layer_out(L - 1), nr.nz1, e->bwd_scale, nullptr, (t + ".dropout_bwd").c_str())) return -1;
    cur = DY::plain(dz[L - 1]);
  }
  for (int i = L - 1; i >= 0; --i) {
    const Mat& xin = (i == 0) ? X : hs[i - 1];
    if (want_w && !(i == 0 && w0_done)) {
      bool fused = false;
      const bool top = rank1 && (i == L - 1);
      int slabs = ks[i];
      if (e->linear_wgrad(cur, xin, M, nr, i, ks[i], 0, true, (t + ".wgrad" + std::to_string(i)).c_str(), 0, -1, top ? &fused : nullptr, &slabs, nullptr,
                          i == 1 ? &X : nullptr)) return -1;
      ks[i] = slabs;
      if (top) {
        if (fused) ks[L] = ks[i];           // the tail gradients were written into the same split-K slabs
        else if (e->linear_wgrad(DY::plain(dTail), hs[L - 1], M, nr, L, ks[L], 0, true, (t + ".wgrad_tail").c_str())) return -1;
      }
    }
    if (i > 0) {
      // the dgrad that produces dz0 can also produce dW0 / db0 (and then dz0 is only stored when dX is wanted)
      int w0_slabs = 0;
      if (e->linear_dgrad(cur, M, nr, i, 0, l.layer_in(i), &hs[i - 1], dz[i - 1], (t + ".dgrad" + std::to_string(i)).c_str(),
                          (want_w && i == 1) ? &X : nullptr, want_dx, &w0_slabs)) return -1;
      if (w0_slabs > 0) { w0_done = true; ks[0] = w0_slabs; }
      if (e->bwd_scale != 1.0f && e->scale_inplace(dz[i - 1], M, l.layer_out(i - 1), nr.nz1, e->bwd_scale, nullptr, (t + ".dropout_bwd").c_str())) return -1;
      cur = DY::plain(dz[i - 1]);
    } else if (want_dx) {
      if (e->linear_dgrad(cur, M, nr, 0, dx_col0, dx_ncols, nullptr, *dX, (t + ".dgrad_x").c_str())) return -1;
    }
  }
  if (out) out->ks = ks;
  return 0;
}

// tanh-Gaussian sampling launch (up to 3 jobs)
static int launch_sample(Engine* e, const Mat& head, int A, const SampleJob* jobs, int njobs) {
  SampleP sp;
  memset(&sp, 0, sizeof(sp));
  sp.head = head.p; sp.head_rs = head.rs; sp.A = A;
  int maxrows = 0;
  for (int i = 0; i < njobs; ++i) { sp.job[i] = jobs[i]; maxrows = std::max(maxrows, jobs[i].rows); }
  e->prof_begin("tanh_sample", 0);
  const int AG = A <= 8 ? 8 : (A <= 16 ? 16 : 32);      // lanes per output row (k_tanh_sample)
  hipLaunchKernelGGL(k_tanh_sample, dim3((unsigned)(((long)maxrows * AG + 255) / 256), njobs, e->R), dim3(256), 0, e->stream, sp);
  e->prof_end();
  return hipGetLastError() == hipSuccess ? 0 : fail("tanh_sample launch");
}
static SampleJob make_job(int head_row0, int rows, int rep, const Mat& eps, const Mat& dst, int dst_col, int dst_row0, const Mat& logp) {
  SampleJob j;
  memset(&j, 0, sizeof(j));
  j.head_row0 = head_row0; j.rows = rows; j.rep = rep; j.eps = eps.p; j.eps_rs = eps.rs;
  j.dst = dst.p; j.dst_rs = dst.rs; j.dst_pitch = dst.pitch; j.dst_col = dst_col; j.dst_row0 = dst_row0;
  j.logp = logp.p; j.logp_rs = logp.rs;
  return j;
}

}  // namespace orl

#include "algo_cql.inc"
#include "algo_iql.inc"
#include "algo_td3bc.inc"
#include "algo_edac.inc"
#include "algo_sac.inc"
#include "algo_mcq.inc"

namespace orl {

// ---------------------------------------------------------------------------------------------
// init / step driver
// ---------------------------------------------------------------------------------------------
int Engine::build_common() {
  // batch slots: obs and next_obs adjacent so [obs; next_obs] is one 2B-row matrix
  alloc("b_obs2", 2 * B, OP);
  alloc("b_act", B, AP); alloc("b_rew", B, 1); alloc("b_term", B, 1);
  alloc("ones", std::max(B, 16), 1);
  d_idx = (long long*)raw_alloc(sizeof(long long) * (size_t)R * B);
  if (!d_idx) return fail("hipMalloc minibatch indices");
  taps["b_obs"] = {W("b_obs2"), B, od};
  taps["b_nobs"] = {W("b_obs2").rows(B), B, od};
  taps["b_act"] = {W("b_act"), B, ad};
  taps["b_rew"] = {W("b_rew"), B, 1};
  taps["b_term"] = {W("b_term"), B, 1};
  // column-tile partial sums of fused single-output tails (linear_fwd): up to 3 extra parts of the longest row batch
  tq_scratch_nets = std::max(2, K);
  alloc("tq_scratch", 3L * (B + 3L * B * N), 1, tq_scratch_nets);
  if (cfg.algo == ORL_ALGO_CQL) {
    if (cfg.cql_cons_row0 < 0 || cfg.cql_cons_rows < 0 || cfg.cql_real_rows < 0 || cfg.cql_cons_row0 + cfg.cql_cons_rows > B || cfg.cql_real_rows > B)
      return fail("cql_cons_row0 / cql_cons_rows / cql_real_rows out of the batch");
  }
  return 0;
}

int Engine::init(const orl_config& c) {
  cfg = c;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("no HIP device available: the update engine needs an MI355X (gfx950)");
  if (c.device < 0 || c.device >= ndev) return fail("bad device ordinal");
  dev = c.device;
  ORL_HIP(hipSetDevice(dev));
  if (c.precision < 0 || c.precision > 2) return fail("precision must be 0 (fp32 MFMA), 1 (split-precision MFMA: two fp16 planes) or 2 (three fp16 planes where a kernel has them, fp32 MFMA elsewhere)");
  if (c.actor_dropout != 0.f && (c.algo != ORL_ALGO_IQL || !(c.actor_dropout > 0.f && c.actor_dropout < 1.f)))
    return fail("actor_dropout: supported for IQL only (run_iql.py --dropout_rate), 0 < p < 1");
  if (build_layouts(c, lay, net_off, net_is_target, &P_train, &P_tgt)) return -1;
  ORL_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  ORL_HIP(hipStreamCreateWithFlags(&side_stream, hipStreamNonBlocking));
  ORL_HIP(hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming));
  ORL_HIP(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
  { const char* f = getenv("ORL_FORK"); if (f) fork_on = atoi(f) != 0; }
  R = c.n_runs; B = c.batch_size; od = c.obs_dim; ad = c.act_dim;
  N = c.num_repeat_actions > 0 ? c.num_repeat_actions : 1;
  OP = rup(od, 4); AP = rup(ad, 4); XP = rup(od + ad, 4); L = c.n_hidden;
  K = (c.algo == ORL_ALGO_EDAC) ? c.num_critics : 2;
  const long arena_floats = (long)R * (P_train + P_tgt);
  if (c.external_arena) arena = c.external_arena;
  else { arena = raw_alloc(sizeof(float) * arena_floats); if (!arena) return fail("hipMalloc arena"); }
  adam_m = raw_alloc(sizeof(float) * R * P_train);
  adam_v = raw_alloc(sizeof(float) * R * P_train);
  grads = raw_alloc(sizeof(float) * (size_t)R * max_slab * P_train);
  scalars = (RunScalars*)raw_alloc(sizeof(RunScalars) * R);
  hyper = (Hyper*)raw_alloc(sizeof(Hyper));
  gstep = (unsigned long long*)raw_alloc(2 * sizeof(unsigned long long));
  gstep_pre = gstep ? gstep + 1 : nullptr;
  gscale_buf = raw_alloc(sizeof(float) * (size_t)GSCALE_SLOTS * R);
  gscale_inv_b = raw_alloc(sizeof(float) * (size_t)R);
  cql_ticket = (unsigned int*)raw_alloc(sizeof(unsigned int) * (size_t)R);
  aloss_part = raw_alloc(sizeof(float) * ((size_t)R * SB_MAXGROUPS * 2 + 192));
  health = (unsigned int*)raw_alloc(sizeof(unsigned int) * (size_t)R);
  health_host.assign(R, 0u);
  if (c.precision == 2) {
    ws_dump = raw_alloc(sizeof(float) * (size_t)WS_DUMP_SLOTS * WS_N);
    if (!ws_dump) return fail("hipMalloc scratch lines");
  }
  if (!adam_m || !adam_v || !grads || !scalars || !hyper || !gstep || !gscale_buf || !gscale_inv_b || !cql_ticket || !health || !aloss_part) return fail("hipMalloc state");
  {
    std::vector<float> inv(R, orl_pow2_scale(1.0f / (float)c.batch_size));
    ORL_HIP(hipMemcpyAsync(gscale_inv_b, inv.data(), sizeof(float) * R, hipMemcpyHostToDevice, stream));
    ORL_HIP(hipMemsetAsync(cql_ticket, 0, sizeof(unsigned int) * R, stream));
    ORL_HIP(hipStreamSynchronize(stream));
  }
  memset(&hyper_host, 0, sizeof(hyper_host));
  hyper_host.lr[ORL_OPT_ACTOR] = c.actor_lr;
  hyper_host.lr[ORL_OPT_CRITIC] = c.critic_lr;
  hyper_host.lr[ORL_OPT_ALPHA] = c.alpha_lr;
  hyper_host.lr[ORL_OPT_CQL_ALPHA] = c.cql_alpha_lr;
  hyper_host.lr[ORL_OPT_CRITIC_V] = c.critic_v_lr;
  hyper_host.lr[ORL_OPT_VAE] = c.behavior_lr;
  ORL_HIP(hipMemcpyAsync(hyper, &hyper_host, sizeof(Hyper), hipMemcpyHostToDevice, stream));
  std::vector<RunScalars> sc(R);
  for (auto& s : sc) { memset(&s, 0, sizeof(s)); s.alpha = c.auto_alpha ? 1.0f : c.alpha; s.alpha_bwd = s.alpha; s.cons_scale = 1.f; }
  ORL_HIP(hipMemcpyAsync(scalars, sc.data(), sizeof(RunScalars) * R, hipMemcpyHostToDevice, stream));
  ORL_HIP(hipStreamSynchronize(stream));
  { const char* f = getenv("ORL_WS_WGRAD_MIN"); if (f && atol(f) > 0) ws_wgrad_min_rows = atol(f); }
  { const char* f = getenv("ORL_WS_WGRAD_MIN_M"); if (f && atoi(f) > 0) ws_wgrad_min_m = atoi(f); }
  { const char* f = getenv("ORL_WS_WGRAD_SLABS"); if (f && atoi(f) > 0) ws_wgrad_slab_cap = std::min(atoi(f), max_slab); }
  { const char* f = getenv("ORL_WS_DGRAD_PLAIN_MIN"); if (f && atol(f) > 0) ws_dgrad_plain_min_rows = atol(f); }
  { const char* f = getenv("ORL_WS_FWD_MIN"); if (f && atol(f) > 0) ws_fwd_min_rows = atol(f); }
  { const char* f = getenv("ORL_SMALL_FWD"); if (f) small_fwd_on = atoi(f) != 0; }
  { const char* f = getenv("ORL_FUSE_SMALL"); if (f) fuse_small = atoi(f) != 0; }
  { const char* f = getenv("ORL_WS_RECOMPUTE_H0"); if (f) recompute_h0 = atoi(f) != 0; }
  { const char* f = getenv("ORL_P3"); if (f) p3_mask = atoi(f); }
  { const char* f = getenv("ORL_SMALL_FWD_MAX"); if (f && atol(f) > 0) small_fwd_max_rows = atol(f); }
  { const char* f = getenv("ORL_WS_BWD_MIN"); if (f && atol(f) > 0) ws_bwd_min_rows = atol(f); }
  { const char* f = getenv("ORL_WS_KEEP_H1"); elide_top = !(f && atoi(f) != 0); }
  { const char* f = getenv("ORL_WS"); use_ws = !(f && atoi(f) == 0); }
  { const char* f = getenv("ORL_WS32"); use_ws32 = !(f && atoi(f) == 0); }
  // launch geometry of the weight-stationary kernels: config fields, overridden once (here) by the environment
  ws_geo.one_round = c.ws_one_round != 0;
  ws_geo.cus = (c.ws_cus >= 8 && c.ws_cus <= 256) ? c.ws_cus : 256;
  { const char* f = getenv("ORL_WS_ONE_ROUND"); if (f) ws_geo.one_round = atoi(f) != 0; }
  { const char* f = getenv("ORL_WS_CUS"); const int x = f ? atoi(f) : 0; if (x >= 8 && x <= 256) ws_geo.cus = x; }
  if (build_common()) return -1;
  int rc = -1;
  switch (c.algo) {
    case ORL_ALGO_CQL: rc = cql_build(); break;
    case ORL_ALGO_IQL: rc = iql_build(); break;
    case ORL_ALGO_TD3BC: rc = td3bc_build(); break;
    case ORL_ALGO_EDAC: rc = edac_build(); break;
    case ORL_ALGO_SAC: rc = sac_build(); break;
    case ORL_ALGO_MCQ: rc = mcq_build(); break;
  }
  if (rc) return rc;
  { Mat lc; lc.p = aloss_part + (long)R * SB_MAXGROUPS * 2; lc.pitch = 192; taps["lab_clk"] = {lc, 1, 192}; }      // shader-clock stamps of lab builds (small_bwd.hip)
  for (auto& ns : noise_slots) taps[ns.name] = {W(ns.name), ns.rows, ns.cols ? ns.cols : ad};      // the noise arrays of the last step
  nm = (int)metric_names.size();
  if (nm > ORL_MAX_METRICS) return fail("too many metrics");
  metrics_last = raw_alloc(sizeof(float) * R * nm);
  metrics_sum = raw_alloc(sizeof(float) * R * nm);
  {
    Mat ones = W("ones");
    hipLaunchKernelGGL(k_fill, dim3((unsigned)((ones.rs * R + 255) / 256)), dim3(256), 0, stream, ones.p, ones.rs * R, 1.0f);
  }
  ORL_HIP(hipStreamSynchronize(stream));
  return 0;
}

int Engine::enqueue_sample() {
  if (!buf || !buf->obs) return fail("no replay buffer attached (orl_engine_attach_buffer)");
  GatherP g;
  memset(&g, 0, sizeof(g));
  g.obs = buf->obs; g.nobs = buf->nobs; g.act = buf->act; g.rew = buf->rew; g.term = buf->term; g.n = buf->n;
  g.OP = buf->OP; g.AP = buf->AP; g.od = od; g.ad = ad; g.B = B; g.W = std::max(OP, AP);
  Mat o2 = W("b_obs2");
  g.b_obs = o2.p; g.b_nobs = o2.p + (long)B * OP; g.obs_rs = o2.rs; g.nobs_rs = o2.rs; g.d_op = OP;
  g.b_act = W("b_act").p; g.act_rs = W("b_act").rs; g.d_ap = AP;
  g.b_rew = W("b_rew").p; g.rew_rs = W("b_rew").rs; g.b_term = W("b_term").p; g.term_rs = W("b_term").rs;
  g.seed = cfg.seed; g.gstep = gstep;
  ORL_LAUNCH("gather", k_gather, dim3((unsigned)(((long)B * g.W + 255) / 256), R), dim3(256), g);
  return 0;
}

int Engine::enqueue_noise() {
  uint32_t sid = 1;
  for (auto& s : noise_slots) {
    const long n = ws_len.at(s.name);
    ORL_LAUNCH("noise", k_noise, dim3((unsigned)((n / 4 + 256) / 256), R), dim3(256), W(s.name).p, n, s.kind,
               s.kind == 2 ? 1.0f - cfg.actor_dropout : cfg.act_low, cfg.act_high, cfg.seed, (const unsigned long long*)gstep, sid);
    ++sid;
  }
  return 0;
}

void Engine::add_prep(const Mat& dst, int row0, int col0, int rows, int width, int src, int rep, int mod, int ncopy, const Mat* buf,
                      unsigned stream_id, int need_sampling, int need_devnoise, int src_row0) {
  PrepSpec s;
  memset(&s, 0, sizeof(s));
  s.job.dst = dst.p; s.job.dst_rs = dst.rs; s.job.dst_pitch = dst.pitch; s.job.dst_row0 = row0; s.job.dst_col0 = col0;
  s.job.rows = rows; s.job.width = width; s.job.src = src; s.job.rep = rep; s.job.mod = mod; s.job.ncopy = ncopy;
  if (buf) { s.job.buf = buf->p; s.job.buf_rs = buf->rs; s.job.buf_pitch = buf->pitch; }
  s.job.stream_id = stream_id; s.job.src_row0 = src_row0;
  s.need_sampling = need_sampling; s.need_devnoise = need_devnoise;
  prep.push_back(s);
}

int Engine::enqueue_prepare(bool sampling, bool devnoise) {
  PrepP p;
  memset(&p, 0, sizeof(p));
  int blocks = 0;
  for (auto& s : prep) {
    if (s.need_sampling >= 0 && s.need_sampling != (int)sampling) continue;
    if (s.need_devnoise >= 0 && s.need_devnoise != (int)devnoise) continue;
    if (p.njobs >= 20) return fail("too many prepare jobs");
    PrepJob j = s.job;
    const long n_el = (long)j.rows * j.width;
    if (n_el > (1L << 30)) return fail("prepare job too large");
    // 16-byte units for gathers of >= 4 columns between 16-byte aligned rows (dataset / batch-slot rows are zero-padded to OP / AP)
    const bool gather = j.src == PS_OBS || j.src == PS_NOBS || j.src == PS_ACT;
    j.vec4 = gather && j.width >= 4 && (j.dst_col0 & 3) == 0 && (j.dst_pitch & 3) == 0 && (j.dst_rs & 3) == 0 && aligned16(j.dst) &&
             ((j.width + 3) & ~3) <= (j.src == PS_ACT ? AP : OP);
    j.units = (int)((j.src == PS_NORMAL || j.src == PS_UNIFORM) ? (n_el + 3) / 4 : (j.vec4 ? (long)j.rows * ((j.width + 3) / 4) : n_el));
    blocks += (j.units + 255) / 256;
    j.block_end = blocks;
    p.job[p.njobs++] = j;
  }
  if (p.njobs == 0) return tick_folded ? fail("prepare: no job to carry the step counter") : 0;
  p.blocks = blocks;
  if (sampling) {
    if (!buf || !buf->obs) return fail("no replay buffer attached (orl_engine_attach_buffer)");
    p.d_obs = buf->obs; p.d_nobs = buf->nobs; p.d_act = buf->act; p.d_rew = buf->rew; p.d_term = buf->term; p.n = buf->n;
    p.OP = buf->OP; p.AP = buf->AP;
    // the minibatch indices (np.random.randint(0, size, B), buffer.py:98) are drawn inside k_prepare by every consumer of a batch row
    // (same Philox counter -> same index) and recorded once in d_idx: no separate index-drawing node in front of the step
    p.idx = d_idx; p.idx_rs = B; p.draw = 1; p.idx_out = d_idx;
  }
  Mat o2 = W("b_obs2");
  p.b_obs = o2.p; p.b_nobs = o2.p + (long)B * OP; p.bo_rs = o2.rs; p.b_op = OP;
  p.b_act = W("b_act").p; p.ba_rs = W("b_act").rs; p.b_ap = AP;
  p.b_rew = W("b_rew").p; p.b_term = W("b_term").p; p.br_rs = W("b_rew").rs;
  p.B = B; p.seed = cfg.seed; p.gstep = gstep; p.lo = cfg.act_low; p.hi = cfg.act_high;
  if (tick_folded) { p.gstep = gstep_pre; p.gstep_publish = gstep; }
  ORL_LAUNCH("prepare", k_prepare, dim3((unsigned)blocks, R), dim3(256), p);
  return 0;
}

int Engine::n_variants() const { return cfg.algo == ORL_ALGO_TD3BC ? 2 : 1; }
int Engine::step_variant() const {
  if (cfg.algo != ORL_ALGO_TD3BC) return 0;
  const int f = cfg.update_actor_freq > 0 ? cfg.update_actor_freq : 1;
  return (step_host % f == 0) ? 1 : 0;   // 1 = actor + target-sync step (td3bc.py:107)
}

int Engine::enqueue_step(int variant) {
  int rc = -1;
  gscale_next = 0; cur_gscale = nullptr; lab_slot = 0;
  switch (cfg.algo) {
    case ORL_ALGO_CQL: rc = cql_step(); break;
    case ORL_ALGO_IQL: rc = iql_step(); break;
    case ORL_ALGO_TD3BC: rc = td3bc_step(variant == 1); break;
    case ORL_ALGO_EDAC: rc = edac_step(); break;
    case ORL_ALGO_SAC: rc = sac_step(); break;
    case ORL_ALGO_MCQ: rc = mcq_step(); break;
  }
  if (rc) return rc;
  if (tick_folded) return 0;               // the step's own kernels advanced the counter
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
struct orl_buffer {
  Buffer b;
};

extern "C" {

const char* orl_last_error(void) { return g_err.c_str(); }
#ifdef ORL_SPLIT_BF16
const char* orl_version(void) { return "orl-engine 0.5 (gfx950; fp32 MFMA + split-bf16 MFMA; CQL IQL TD3BC EDAC SAC(MOPO) COMBO MCQ)"; }
#else
const char* orl_version(void) { return "orl-engine 0.5 (gfx950; fp32 MFMA + split-fp16 MFMA; CQL IQL TD3BC EDAC SAC(MOPO) COMBO MCQ)"; }
#endif
int orl_split_bits(void) { return ORL_SPLIT_BITS; }

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
  if (algo == ORL_ALGO_SAC) { c->actor_lr = 1e-4f; c->critic_lr = 3e-4f; c->deterministic_backup = 0; }   /* run_mopo.py:33-34 */
  c->vae_hidden = 750; c->vae_latent = 2 * c->act_dim; c->mcq_lambda = 0.9f; c->behavior_lr = 1e-3f;   /* run_mcq.py:34-36, 93-99 */
  if (algo == ORL_ALGO_MCQ) { c->hidden[0] = c->hidden[1] = 400; c->actor_lr = c->critic_lr = c->alpha_lr = 3e-4f; c->target_entropy = -(float)c->act_dim; }
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
  ORL_HIP(hipSetDevice(h->e.dev));
  ORL_HIP(hipMemcpyAsync(d, host, sizeof(float) * n, hipMemcpyHostToDevice, h->e.stream));
  ORL_HIP(hipStreamSynchronize(h->e.stream));
  return 0;
}
int orl_net_get(orl_engine* h, int run, int net, float* host, int64_t n) {
  float* d = orl_net_ptr(h, run, net);
  if (!d || n != h->e.lay[net].size) return fail("orl_net_get: bad net/run/size");
  ORL_HIP(hipSetDevice(h->e.dev));
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
  if (which == ORL_SCALAR_LOG_ALPHA) { s.log_alpha = v; if (e.cfg.auto_alpha) s.alpha = expf(v); }   // sac.py:46 / edac.py:45
  else if (which == ORL_SCALAR_CQL_LOG_ALPHA) s.cql_log_alpha = v;
  else if (which == ORL_SCALAR_ALPHA) s.alpha = v;       // resume: EDAC / SAC keep a clamped alpha between steps (edac.py:110)
  else if (which == ORL_SCALAR_LOG_ALPHA_M) s.la_m = v;
  else if (which == ORL_SCALAR_LOG_ALPHA_V) s.la_v = v;
  else if (which == ORL_SCALAR_CQL_LOG_ALPHA_M) s.cla_m = v;
  else if (which == ORL_SCALAR_CQL_LOG_ALPHA_V) s.cla_v = v;
  else if (which == ORL_SCALAR_LAST_ACTOR_LOSS) s.last_actor_loss = v;
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
  else if (which == ORL_SCALAR_LOG_ALPHA_M) *v = s.la_m;
  else if (which == ORL_SCALAR_LOG_ALPHA_V) *v = s.la_v;
  else if (which == ORL_SCALAR_CQL_LOG_ALPHA_M) *v = s.cla_m;
  else if (which == ORL_SCALAR_CQL_LOG_ALPHA_V) *v = s.cla_v;
  else if (which == ORL_SCALAR_LAST_ACTOR_LOSS) *v = s.last_actor_loss;
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
  ORL_HIP(hipMemsetAsync(e.gstep, 0, 2 * sizeof(unsigned long long), e.stream));
  e.step_host = 0;
  ORL_HIP(hipStreamSynchronize(e.stream));
  return 0;
}
static int adam_xfer(orl_engine* h, int run, int net, float* m, float* v, int64_t n, bool to_host) {
  Engine& e = h->e;
  if (run < 0 || run >= e.R || net < 0 || net >= ORL_NUM_NETS || !e.lay[net].present || e.net_is_target[net]) return fail("orl_adam: bad run / net");
  if (n != e.lay[net].size || !m || !v) return fail("orl_adam: bad size");
  ORL_HIP(hipSetDevice(e.dev));
  ORL_HIP(hipStreamSynchronize(e.stream));
  const long off = (long)run * e.P_train + e.net_off[net];
  if (to_host) {
    ORL_HIP(hipMemcpy(m, e.adam_m + off, sizeof(float) * n, hipMemcpyDeviceToHost));
    ORL_HIP(hipMemcpy(v, e.adam_v + off, sizeof(float) * n, hipMemcpyDeviceToHost));
  } else {
    ORL_HIP(hipMemcpy(e.adam_m + off, m, sizeof(float) * n, hipMemcpyHostToDevice));
    ORL_HIP(hipMemcpy(e.adam_v + off, v, sizeof(float) * n, hipMemcpyHostToDevice));
  }
  return 0;
}
int orl_adam_get(orl_engine* h, int run, int net, float* m, float* v, int64_t n) { return adam_xfer(h, run, net, m, v, n, true); }
int orl_adam_set(orl_engine* h, int run, int net, const float* m, const float* v, int64_t n) {
  return adam_xfer(h, run, net, const_cast<float*>(m), const_cast<float*>(v), n, false);
}
int orl_set_step_count(orl_engine* h, int64_t steps) {
  Engine& e = h->e;
  if (steps < 0) return fail("negative step count");
  ORL_HIP(hipSetDevice(e.dev));
  ORL_HIP(hipStreamSynchronize(e.stream));
  const unsigned long long s[2] = {(unsigned long long)steps, (unsigned long long)steps};      // (both cells: gstep and gstep_pre)
  ORL_HIP(hipMemcpy(e.gstep, s, sizeof(s), hipMemcpyHostToDevice));
  e.step_host = s[0];
  return 0;
}

// ---- replay buffer ----
int orl_buffer_create(int32_t obs_dim, int32_t act_dim, int32_t device, orl_buffer** out) {
  if (!out || obs_dim < 1 || act_dim < 1) return fail("orl_buffer_create: bad arguments");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("no HIP device available: the replay buffer lives in MI355X HBM");
  if (device < 0 || device >= ndev) return fail("bad device ordinal");
  orl_buffer* h = new orl_buffer();
  h->b.dev = device; h->b.od = obs_dim; h->b.ad = act_dim; h->b.OP = rup(obs_dim, 4); h->b.AP = rup(act_dim, 4);
  *out = h;
  return 0;
}
void orl_buffer_destroy(orl_buffer* h) { delete h; }
static int upload_padded(float** dst, const float* src, long n, int dim, int pitch) {
  ORL_HIP(hipMalloc((void**)dst, sizeof(float) * n * pitch));
  ORL_HIP(hipMemset(*dst, 0, sizeof(float) * n * pitch));
  ORL_HIP(hipMemcpy2D(*dst, sizeof(float) * pitch, src, sizeof(float) * dim, sizeof(float) * dim, n, hipMemcpyHostToDevice));
  return 0;
}
int orl_buffer_load(orl_buffer* h, const float* obs, const float* act, const float* next_obs, const float* rew, const float* term, int64_t n) {
  Buffer& b = h->b;
  if (n <= 0) return fail("empty dataset");
  ORL_HIP(hipSetDevice(b.dev));
  for (float** p : {&b.obs, &b.nobs, &b.act, &b.rew, &b.term}) if (*p) { hipFree(*p); *p = nullptr; }
  if (upload_padded(&b.obs, obs, n, b.od, b.OP)) return -1;
  if (upload_padded(&b.nobs, next_obs, n, b.od, b.OP)) return -1;
  if (upload_padded(&b.act, act, n, b.ad, b.AP)) return -1;
  if (upload_padded(&b.rew, rew, n, 1, 1)) return -1;
  if (upload_padded(&b.term, term, n, 1, 1)) return -1;
  b.n = n;
  b.gen++;               // the arrays moved: engines that captured them re-capture (orl_learn_n)
  return 0;
}
int64_t orl_buffer_size(orl_buffer* h) { return h->b.n; }
int orl_buffer_normalize_obs(orl_buffer* h, float eps, float* mean_out, float* std_out) {
  Buffer& b = h->b;
  if (!b.obs) return fail("normalize_obs: empty buffer");
  ORL_HIP(hipSetDevice(b.dev));
  double* sums = nullptr;
  ORL_HIP(hipMalloc((void**)&sums, sizeof(double) * 2 * b.od));
  ORL_HIP(hipMemset(sums, 0, sizeof(double) * 2 * b.od));
  hipLaunchKernelGGL(k_colstats, dim3(128, b.od), dim3(256), 0, 0, (const float*)b.obs, b.n, b.OP, b.od, sums);
  std::vector<double> hs(2 * b.od);
  ORL_HIP(hipMemcpy(hs.data(), sums, sizeof(double) * 2 * b.od, hipMemcpyDeviceToHost));
  hipFree(sums);
  std::vector<float> mean(b.od), sd(b.od);
  for (int c = 0; c < b.od; ++c) {
    const double m = hs[c] / (double)b.n;
    double var = hs[b.od + c] / (double)b.n - m * m;
    if (var < 0) var = 0;
    mean[c] = (float)m;
    sd[c] = (float)sqrt(var) + eps;        // buffer.py:90: std + eps
  }
  float *dm = nullptr, *ds = nullptr;
  ORL_HIP(hipMalloc((void**)&dm, sizeof(float) * b.od));
  ORL_HIP(hipMalloc((void**)&ds, sizeof(float) * b.od));
  ORL_HIP(hipMemcpy(dm, mean.data(), sizeof(float) * b.od, hipMemcpyHostToDevice));
  ORL_HIP(hipMemcpy(ds, sd.data(), sizeof(float) * b.od, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_normalize, dim3((unsigned)((b.n * b.od + 255) / 256)), dim3(256), 0, 0, b.obs, b.nobs, b.n, b.OP, b.od,
                     (const float*)dm, (const float*)ds);
  ORL_HIP(hipDeviceSynchronize());
  hipFree(dm); hipFree(ds);
  b.absmax_gen = ~0ull;                    // the values changed in place (same arrays: captured graphs stay valid, the cached range does not)
  if (mean_out) memcpy(mean_out, mean.data(), sizeof(float) * b.od);
  if (std_out) memcpy(std_out, sd.data(), sizeof(float) * b.od);
  return 0;
}
int orl_buffer_sample(orl_buffer* h, const int64_t* idx, int32_t batch, uint64_t seed, float* obs_out, float* act_out,
                      float* next_obs_out, float* rew_out, float* term_out) {
  Buffer& b = h->b;
  if (!b.obs) return fail("sample: empty buffer");
  if (batch < 1 || !obs_out || !act_out || !next_obs_out || !rew_out || !term_out) return fail("sample: bad arguments");
  ORL_HIP(hipSetDevice(b.dev));
  if (idx) {
    for (int i = 0; i < batch; ++i) if (idx[i] < 0 || idx[i] >= b.n) return fail("sample index out of range");
    if (b.idx_cap < batch) {
      if (b.idx) hipFree(b.idx);
      ORL_HIP(hipMalloc((void**)&b.idx, sizeof(long long) * batch));
      b.idx_cap = batch;
    }
    ORL_HIP(hipMemcpy(b.idx, idx, sizeof(long long) * batch, hipMemcpyHostToDevice));
  }
  GatherP g;
  memset(&g, 0, sizeof(g));
  g.obs = b.obs; g.nobs = b.nobs; g.act = b.act; g.rew = b.rew; g.term = b.term; g.n = b.n;
  g.OP = b.OP; g.AP = b.AP; g.od = b.od; g.ad = b.ad; g.B = batch; g.W = std::max(b.od, b.ad);
  g.idx = idx ? b.idx : nullptr; g.idx_rs = batch;
  g.b_obs = obs_out; g.b_nobs = next_obs_out; g.b_act = act_out; g.b_rew = rew_out; g.b_term = term_out;
  g.d_op = b.od; g.d_ap = b.ad;
  g.seed = seed; g.gstep = nullptr; g.counter = b.counter++;
  hipLaunchKernelGGL(k_gather, dim3((unsigned)(((long)batch * g.W + 255) / 256), 1), dim3(256), 0, 0, g);
  ORL_HIP(hipGetLastError());
  ORL_HIP(hipStreamSynchronize(0));
  return 0;
}
int orl_engine_attach_buffer(orl_engine* h, orl_buffer* b) {
  if (!b) { h->e.buf = nullptr; h->e.drop_graphs(); return 0; }
  if (b->b.od != h->e.od || b->b.ad != h->e.ad) return fail("attach_buffer: obs/act dims differ from the engine's");
  if (b->b.dev != h->e.dev) return fail("attach_buffer: buffer lives on another device");
  if (h->e.split_scales() && b->b.obs) {
    // split precision multiplies fp16 hi + lo planes: an observation or action component of 65504 or more is +-inf there.  The dataset is
    // checked once per load (a 40 us reduction over the HBM arrays), not per sampled batch.
    Buffer& bb = b->b;
    if (bb.absmax_gen != bb.gen) {
      ORL_HIP(hipSetDevice(bb.dev));
      unsigned int* d = nullptr;
      ORL_HIP(hipMalloc((void**)&d, sizeof(unsigned int)));
      ORL_HIP(hipMemset(d, 0, sizeof(unsigned int)));
      hipLaunchKernelGGL(k_absmax, dim3(1024), dim3(256), 0, 0, (const float*)bb.obs, bb.n * bb.OP, d);
      hipLaunchKernelGGL(k_absmax, dim3(1024), dim3(256), 0, 0, (const float*)bb.nobs, bb.n * bb.OP, d);
      hipLaunchKernelGGL(k_absmax, dim3(512), dim3(256), 0, 0, (const float*)bb.act, bb.n * bb.AP, d);
      unsigned int bits = 0;
      ORL_HIP(hipMemcpy(&bits, d, sizeof(bits), hipMemcpyDeviceToHost));
      hipFree(d);
      memcpy(&bb.absmax, &bits, sizeof(float));
      bb.absmax_gen = bb.gen;
    }
    if (!(bb.absmax < 65504.0f)) {
      char msg[256];
      snprintf(msg, sizeof(msg), "attach_buffer: the dataset's observations / actions reach |x| = %g, beyond the operand range of precision 1 "
               "(fp16 hi + lo planes, |x| < 65504): normalise the observations (ReplayBuffer.normalize_obs) or use precision 0", (double)bb.absmax);
      return fail(msg);
    }
  }
  h->e.buf = &b->b;
  h->e.buf_gen = b->b.gen;
  h->e.drop_graphs();             // captured graphs hold the old dataset pointers
  return 0;
}

// ---- hot path ----
static int copy_rows(Engine& e, const Mat& dst, const float* src, int rows, int dim, bool on_device, long row0 = 0) {
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
  const int B = e.B;
  if (b) {
    const bool dv = b->on_device != 0;
    if (copy_rows(e, e.W("b_obs2"), b->observations, B, e.od, dv, 0)) return -1;
    if (copy_rows(e, e.W("b_obs2"), b->next_observations, B, e.od, dv, B)) return -1;
    if (copy_rows(e, e.W("b_act"), b->actions, B, e.ad, dv)) return -1;
    if (copy_rows(e, e.W("b_rew"), b->rewards, B, 1, dv)) return -1;
    if (copy_rows(e, e.W("b_term"), b->terminals, B, 1, dv)) return -1;
  }
  if (nz) {
    const bool dv = nz->on_device != 0;
    for (size_t i = 0; i < e.noise_slots.size(); ++i)
      if (copy_rows(e, e.W(e.noise_slots[i].name), nz->slot[i], e.noise_slots[i].rows, e.noise_slots[i].cols ? e.noise_slots[i].cols : e.ad, dv)) return -1;
  }
  if (!e.prep.empty()) { if (e.enqueue_prepare(false, nz == nullptr)) return -1; }
  else if (!nz) { if (e.enqueue_noise()) return -1; }
  if (e.enqueue_step(e.step_variant())) return -1;
  e.step_host++;
  ORL_HIP(hipStreamSynchronize(e.stream));
  std::vector<float> m(e.R * e.nm);
  ORL_HIP(hipMemcpy(m.data(), e.metrics_last, sizeof(float) * m.size(), hipMemcpyDeviceToHost));
  if (metrics) {
    for (int r = 0; r < e.R; ++r)
      for (int k = 0; k < ORL_MAX_METRICS; ++k) metrics[r * ORL_MAX_METRICS + k] = k < e.nm ? m[r * e.nm + k] : 0.f;
  }
  unsigned int bad = 0;
  if (e.health_update(m.data(), 1, &bad)) return -1;
  return bad ? ORL_RC_UNHEALTHY : 0;
}

int orl_learn_n(orl_engine* h, int n_steps, float* metrics_mean, float* elapsed_ms) {
  Engine& e = h->e;
  ORL_HIP(hipSetDevice(e.dev));
  if (n_steps <= 0) return fail("n_steps must be positive");
  if (!e.buf || !e.buf->obs) return fail("orl_learn_n: no replay buffer attached");
  ORL_HIP(hipMemsetAsync(e.metrics_sum, 0, sizeof(float) * e.R * e.nm, e.stream));
  if (e.buf_gen != e.buf->gen) {       // the buffer was reloaded since the graphs were captured: they hold freed pointers and the old size
    ORL_HIP(hipStreamSynchronize(e.stream));
    e.drop_graphs();
    e.buf_gen = e.buf->gen;
  }
  const bool graphable = e.use_graph && !e.prof_on;
  if (graphable) {
    for (int v = 0; v < e.n_variants(); ++v) {
      if (e.graph_exec[v]) continue;
      ORL_HIP(hipStreamBeginCapture(e.stream, hipStreamCaptureModeRelaxed));
      int rc = (e.prep.empty() ? (e.enqueue_sample() || e.enqueue_noise()) : e.enqueue_prepare(true, true)) || e.enqueue_step(v);
      hipError_t ce = hipStreamEndCapture(e.stream, &e.graph[v]);
      if (rc) return -1;
      if (ce != hipSuccess) return fail(std::string("graph capture: ") + hipGetErrorString(ce));
      ORL_HIP(hipGraphInstantiate(&e.graph_exec[v], e.graph[v], nullptr, nullptr, 0));
    }
  }
  // (the two timing events are released on every exit path, early error returns included)
  struct EventPair {
    hipEvent_t a = nullptr, b = nullptr;
    ~EventPair() { if (a) hipEventDestroy(a); if (b) hipEventDestroy(b); }
  } ev;
  ORL_HIP(hipEventCreate(&ev.a));
  ORL_HIP(hipEventCreate(&ev.b));
  const hipEvent_t t0 = ev.a, t1 = ev.b;
  if (e.prof_on) { e.prof.clear(); e.ev_used = 0; }
  ORL_HIP(hipEventRecord(t0, e.stream));
  for (int s = 0; s < n_steps; ++s) {
    const int v = e.step_variant();
    if (graphable) { ORL_HIP(hipGraphLaunch(e.graph_exec[v], e.stream)); }
    else {
      if (e.prep.empty()) { if (e.enqueue_sample()) return -1; if (e.enqueue_noise()) return -1; }
      else if (e.enqueue_prepare(true, true)) return -1;
      if (e.enqueue_step(v)) return -1;
    }
    e.step_host++;
  }
  ORL_HIP(hipEventRecord(t1, e.stream));
  ORL_HIP(hipStreamSynchronize(e.stream));
  float ms = 0.f;
  ORL_HIP(hipEventElapsedTime(&ms, t0, t1));
  if (elapsed_ms) *elapsed_ms = ms;
  std::vector<float> m(e.R * e.nm);
  ORL_HIP(hipMemcpy(m.data(), e.metrics_sum, sizeof(float) * m.size(), hipMemcpyDeviceToHost));      // (a non-finite metric of ANY of the n steps stays in its sum)
  if (metrics_mean) {
    for (int r = 0; r < e.R; ++r)
      for (int k = 0; k < ORL_MAX_METRICS; ++k) metrics_mean[r * ORL_MAX_METRICS + k] = k < e.nm ? m[r * e.nm + k] / n_steps : 0.f;
  }
  unsigned int bad = 0;
  if (e.health_update(m.data(), n_steps, &bad)) return -1;
  return bad ? ORL_RC_UNHEALTHY : 0;
}

int orl_health(orl_engine* h, uint32_t* flags_out) {
  Engine& e = h->e;
  unsigned int all = 0;
  for (int r = 0; r < e.R; ++r) { all |= e.health_host[r]; if (flags_out) flags_out[r] = e.health_host[r]; }
  return (int)all;
}
int orl_health_check(orl_engine* h, uint32_t* flags_out) {
  Engine& e = h->e;
  if (hipSetDevice(e.dev) != hipSuccess) { fail("hipSetDevice"); return -1; }
  if (hipStreamSynchronize(e.stream) != hipSuccess) { fail("sync"); return -1; }
  unsigned int all = 0;
  if (e.health_update(nullptr, -1, &all)) return -1;
  if (flags_out) for (int r = 0; r < e.R; ++r) flags_out[r] = e.health_host[r];
  return (int)all;
}
int orl_health_clear(orl_engine* h) {
  Engine& e = h->e;
  ORL_HIP(hipSetDevice(e.dev));
  ORL_HIP(hipMemsetAsync(e.health, 0, sizeof(unsigned int) * e.R, e.stream));
  ORL_HIP(hipStreamSynchronize(e.stream));
  e.health_host.assign(e.R, 0u);
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

// packed ReLU-mask words of a hidden activation of the LAST step (one 32-bit word per (row, 32 columns), all members of the family):
// returns the number of words written, or < 0 (unknown workspace / no mask / the producing launch did not emit bits)
int64_t orl_debug_read_bits(orl_engine* h, int run, const char* name, uint32_t* host, int64_t cap) {
  Engine& e = h->e;
  auto it = e.ws.find(name);
  if (it == e.ws.end() || !it->second.bits) { fail(std::string("no mask bits for workspace ") + name); return -1; }
  if (run < 0 || run >= e.R) { fail("bad run"); return -1; }
  const Mat& m = it->second;
  if (!e.bits_live.count(m.bits)) { fail(std::string("mask bits of ") + name + " were not emitted by the last step's kernels"); return -1; }
  if (cap < m.brs) { fail("bits buffer too small"); return -1; }
  if (hipStreamSynchronize(e.stream) != hipSuccess) { fail("sync"); return -1; }
  if (hipMemcpy(host, m.bits + (long)run * m.brs, sizeof(uint32_t) * m.brs, hipMemcpyDeviceToHost) != hipSuccess) { fail("bits copy"); return -1; }
  return m.brs;
}

// gradient of the LAST step w.r.t. every parameter of `net` (state_dict order, orl_net_floats values): the split-K slabs the
// backward kernels wrote, summed per tensor group (double accumulation on the host; k_adam sums the same slabs in fp32)
int orl_debug_grads(orl_engine* h, int run, int net, float* host, int64_t n) {
  Engine& e = h->e;
  if (run < 0 || run >= e.R) return fail("bad run");
  if (net < 0 || net >= ORL_NUM_NETS || !e.lay[net].present || e.net_is_target[net]) return fail("orl_debug_grads: not a trainable net");
  const NetLayout& l = e.lay[net];
  if (n != l.size) return fail("orl_debug_grads: bad size");
  const auto& segs = e.last_segs[net];
  if (segs.empty()) return fail("orl_debug_grads: no optimizer step has run for this net yet");
  int maxs = 1;
  for (auto& sg : segs) maxs = std::max(maxs, sg.second);
  ORL_HIP(hipSetDevice(e.dev));
  ORL_HIP(hipStreamSynchronize(e.stream));
  std::vector<float> tmp((size_t)maxs * l.size);
  const float* src = e.grads + (long)run * e.max_slab * e.P_train + e.net_off[net];
  ORL_HIP(hipMemcpy2D(tmp.data(), sizeof(float) * l.size, src, sizeof(float) * e.P_train, sizeof(float) * l.size, maxs, hipMemcpyDeviceToHost));
  long i = 0;
  for (auto& sg : segs) {
    for (; i < sg.first && i < l.size; ++i) {
      double a = 0.0;
      for (int k = 0; k < sg.second; ++k) a += tmp[(size_t)k * l.size + i];
      host[i] = (float)a;
    }
  }
  for (; i < l.size; ++i) host[i] = tmp[i];
  return 0;
}

int orl_profile_enable(orl_engine* h, int on) { h->e.prof_on = on != 0; return 0; }
int orl_profile_query(orl_engine* h, int idx, char* name, int name_cap, double* total_ms, int64_t* launches, double* flops_per_launch,
                      double* bytes_per_launch) {
  Engine& e = h->e;
  if (hipStreamSynchronize(e.stream) != hipSuccess) return fail("sync");
  std::map<std::string, std::tuple<double, int64_t, double, double>> agg;
  for (auto& p : e.prof) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.a, p.b) != hipSuccess) continue;
    auto& a = agg[p.name];
    std::get<0>(a) += ms; std::get<1>(a) += 1; std::get<2>(a) += p.flops; std::get<3>(a) += p.bytes;
  }
  std::vector<std::pair<double, std::string>> order;
  for (auto& kv : agg) order.push_back({-std::get<0>(kv.second), kv.first});
  std::sort(order.begin(), order.end());
  if (idx < 0 || idx >= (int)order.size()) return 1;
  const auto& a = agg[order[idx].second];
  snprintf(name, name_cap, "%s", order[idx].second.c_str());
  *total_ms = std::get<0>(a); *launches = std::get<1>(a); *flops_per_launch = std::get<2>(a) / std::get<1>(a);      // means over the tag's launches
  if (bytes_per_launch) *bytes_per_launch = std::get<3>(a) / std::get<1>(a);
  return 0;
}

// kernel unit test: C = op(A) op(B) through one tile configuration.
//  mode 0: forward   C[M,N] = relu(A[M,K] B[N,K]^T + v0[N])
//  mode 1: dgrad     C[M,N] = A[M,K] Bm[K,N], masked by v0 viewed [M,N] (>0)
//  mode 2: wgrad     A is [K,M], B is [K,N]: C = [A^T B (M*N) | column sums of A (M)]; split-K slabs summed on host
//  mode 3: rank-1 dgrad  A_eff[m,k] = A[m,k]>0 ? v0[m]*v1[k] : 0 ; C = A_eff Bm[K,N]
//  mode 4: rank-1 wgrad  A_eff[k,m] = A[k,m]>0 ? v0[k]*v1[m] : 0 ; output like mode 2
int orl_debug_gemm(int cfg, int mode, int M, int N, int K, const float* A, const float* Bh, const float* v0, const float* v1,
                   float* C, int ksplit, int precision) {
  if (precision < 0 || precision > 2) return fail("precision must be 0, 1 or 2");
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
  p.nz1 = 1; p.ksplit = ksplit; p.C = dC; p.c_ks = nC; p.c_sn = 1;
  const bool fs = (cfg & 16) != 0;      // bit 4 of cfg forces the scalar loaders (unit tests cover both paths)
  cfg &= 15;
  p.a_rlim = M & ~3; p.b_rlim = N & ~3;
  hipError_t err = hipSuccess;
  if (mode == 0) {
    p.A = {dA, 0, 0}; p.a_sr = K; p.a_sk = 1; p.B = {dB, 0, 0}; p.b_sr = K; p.b_sk = 1;
    p.M = M; p.N = N; p.K = K; p.c_sr = N; p.bias = {d0, 0, 0};
    err = launch_gemm<PA_PLAIN, PB_PLAIN, E_BIAS_RELU>(cfg, p, 1, st, false, fs, precision);
  } else if (mode == 1 || mode == 3) {
    p.A = {dA, 0, 0}; p.a_sr = K; p.a_sk = 1; p.B = {dB, 0, 0}; p.b_sr = 1; p.b_sk = N;
    p.M = M; p.N = N; p.K = K; p.c_sr = N;
    if (mode == 1) { p.aux = {d0, 0, 0}; p.aux_sr = N; err = launch_gemm<PA_PLAIN, PB_PLAIN, E_MASK>(cfg, p, 1, st, false, fs, precision); }
    else { p.rowv = {d0, 0, 0}; p.colv = {d1, 0, 0}; p.a_trans = 0; err = launch_gemm<PA_RANK1, PB_PLAIN, E_PLAIN>(cfg, p, 1, st, false, fs, precision); }
  } else {
    p.A = {dA, 0, 0}; p.a_sr = 1; p.a_sk = M; p.B = {dB, 0, 0}; p.b_sr = 1; p.b_sk = N;
    p.M = M; p.N = N; p.K = K; p.c_sr = N; p.ones_row = 1 << 30;
    p.bias_out = dC + (long)M * N; p.bo_ks = nC;
    if (mode == 2) err = launch_gemm<PA_PLAIN, PB_PLAIN, E_WGRAD>(cfg, p, 1, st, false, fs, precision);
    else { p.rowv = {d0, 0, 0}; p.colv = {d1, 0, 0}; p.a_trans = 1; err = launch_gemm<PA_RANK1, PB_PLAIN, E_WGRAD>(cfg, p, 1, st, false, fs, precision); }
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

// GEMM tuning tap: times `reps` launches of one tile configuration on random data of the hot-path shapes.
//  kind 0: forward  (VECK,VECK, bias+relu)   C[M,N]   = relu(A[M,K] W[N,K]^T + b)
//  kind 1: dgrad    (VECK,BLK4, rank-1, mask) C[M,N]   = ((H>0) dq w) W[K,N]  masked
//  kind 2: wgrad    (BLK4,BLK4, rank-1, ones) C[M,N+1] = ((H>0) dq w)^T X     (M = out, K = rows), split-K
int orl_debug_gemm_time(int cfg, int kind, int M, int N, int K, int nz, int ksplit, int reps, float* ms_out) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("no HIP device");
  const long nA = (kind == 2) ? (long)K * M : (long)M * K;
  const long nB = (kind == 0) ? (long)N * K : (kind == 1 ? (long)K * N : (long)K * N);
  const long nC = (kind == 2) ? (long)M * (N + 1) * ksplit : (long)M * N;
  float *dA, *dB, *dC, *dv0, *dv1, *dH;
  ORL_HIP(hipMalloc(&dA, sizeof(float) * nA * nz));
  ORL_HIP(hipMalloc(&dB, sizeof(float) * nB * nz));
  ORL_HIP(hipMalloc(&dC, sizeof(float) * nC * nz));
  ORL_HIP(hipMalloc(&dH, sizeof(float) * (long)M * N * nz));
  ORL_HIP(hipMalloc(&dv0, sizeof(float) * (M + K + N) * nz));
  ORL_HIP(hipMalloc(&dv1, sizeof(float) * (M + K + N) * nz));
  std::vector<float> h(std::max(std::max(nA, nB), std::max((long)M * N, (long)(M + K + N))) * nz);
  unsigned s = 12345u;
  auto fill = [&](float* d, long n) { for (long i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) / 8388608.0f) - 1.0f; } return hipMemcpy(d, h.data(), sizeof(float) * n, hipMemcpyHostToDevice); };
  ORL_HIP(fill(dA, nA * nz)); ORL_HIP(fill(dB, nB * nz)); ORL_HIP(fill(dH, (long)M * N * nz)); ORL_HIP(fill(dv0, (long)(M + K + N) * nz)); ORL_HIP(fill(dv1, (long)(M + K + N) * nz));
  GemmP p;
  memset(&p, 0, sizeof(p));
  p.nz1 = nz; p.ksplit = ksplit; p.C = dC; p.c_sn = 1; p.c_s1 = nC;
  p.A = {dA, 0, nA}; p.B = {dB, 0, nB};
  p.rowv = {dv0, 0, (long)(M + K + N)}; p.colv = {dv1, 0, (long)(M + K + N)};
  p.bias = {dv0, 0, (long)(M + K + N)}; p.aux = {dH, 0, (long)M * N}; p.aux_sr = N;
  hipStream_t st = nullptr;
  hipEvent_t e0, e1;
  ORL_HIP(hipEventCreate(&e0)); ORL_HIP(hipEventCreate(&e1));
  hipError_t err = hipSuccess;
  for (int it = 0; it < reps + 3; ++it) {
    if (it == 3) ORL_HIP(hipEventRecord(e0, st));
    if (kind == 0) {
      p.a_sr = K; p.a_sk = 1; p.b_sr = K; p.b_sk = 1; p.M = M; p.N = N; p.K = K; p.c_sr = N;
      err = launch_tune(cfg, 0, p, nz, st);
    } else if (kind == 1) {
      p.a_sr = K; p.a_sk = 1; p.b_sr = 1; p.b_sk = N; p.b_rlim = N & ~3; p.M = M; p.N = N; p.K = K; p.c_sr = N; p.a_trans = 0;
      err = launch_tune(cfg, 1, p, nz, st);
    } else {
      p.a_sr = 1; p.a_sk = M; p.b_sr = 1; p.b_sk = N; p.a_rlim = M & ~3; p.b_rlim = N & ~3; p.M = M; p.N = N; p.K = K; p.c_sr = N;
      p.ones_row = 1 << 30; p.a_trans = 1; p.c_ks = (long)M * (N + 1); p.c_s1 = nC; p.bias_out = dC + (long)M * N; p.bo_s1 = nC; p.bo_ks = (long)M * (N + 1);
      err = launch_tune(cfg, 2, p, nz, st);
    }
    if (err != hipSuccess) return fail(std::string("gemm_time launch: ") + hipGetErrorString(err));
  }
  ORL_HIP(hipEventRecord(e1, st));
  ORL_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  ORL_HIP(hipEventElapsedTime(&ms, e0, e1));
  *ms_out = ms / reps;
  hipEventDestroy(e0); hipEventDestroy(e1);
  hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dH); hipFree(dv0); hipFree(dv1);
  return 0;
}

}  // extern "C"
