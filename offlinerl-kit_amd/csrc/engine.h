// engine.h — host-side engine state: parameter arenas, workspaces, launch helpers.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../include/orl_engine.h"
#include "gemm.h"
#include "kernels.h"
#include "ws_gemm.h"
#include "small_fwd.h"
#include "small_bwd.h"

namespace orl {

void set_error(const std::string& msg);
int fail(const std::string& msg);
#define ORL_HIP(expr)                                                                            \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess) {                                                                      \
      ::orl::set_error(std::string(#expr) + ": " + hipGetErrorString(_e));                       \
      return -1;                                                                                 \
    }                                                                                            \
  } while (0)

struct TensorInfo {
  std::string name;
  long off;
  int ndim;
  long shape[4];
};

// A network (family) = MLP of L ReLU layers + one linear tail (critic last / actor head).
// `members` identical nets are batched through blockIdx.z.  Two storage layouts:
//   ens = false : nn.Linear, W (out,in), every member is its own contiguous block (member stride = size)
//   ens = true  : EnsembleLinear (nets/ensemble_linear.py:9-41), W (K,in,out), b (K,1,out): member k of layer l
//                 lives at w_off[l] + k*in*out
struct NetLayout {
  bool present = false;
  bool ens = false;
  int members = 1;
  int in_dim = 0, L = 0, out_dim = 0;
  int H[ORL_MAX_HIDDEN + 1] = {0};
  long w_off[ORL_MAX_HIDDEN + 1] = {0};  // layers 0..L-1 hidden, layer L = tail
  long b_off[ORL_MAX_HIDDEN + 1] = {0};
  long w_ms[ORL_MAX_HIDDEN + 1] = {0};   // member strides
  long b_ms[ORL_MAX_HIDDEN + 1] = {0};
  long extra_off = -1;                   // IQL sigma_param
  long size = 0;                         // floats of ONE net (ens: of the whole ensemble)
  long stride() const { return (size + 3) & ~3L; }   // arena stride: keeps every net 16-B aligned
  std::vector<TensorInfo> tensors;
  int layer_in(int l) const { return l == 0 ? in_dim : H[l - 1]; }
  int layer_out(int l) const { return l == L ? out_dim : H[l]; }
};

// matrix / vector views batched over (run, member)
struct Mat {
  float* p = nullptr;
  long rs = 0, cs = 0;  // run stride, member stride (elements)
  int pitch = 0;
  // packed ReLU-mask bits of a hidden activation (gemm.h, GemmP::mb_out): one 32-bit word per (row, 32 columns); strides in words
  unsigned int* bits = nullptr;
  long brs = 0, bcs = 0;
  int bg = 0;
  Mat rows(long r0) const { Mat m = *this; m.p = p + r0 * pitch; if (bits) m.bits = bits + r0 * bg; return m; }
  Mat cols(int c0) const { Mat m = *this; m.p = p + c0; m.bits = nullptr; return m; }
  Mat net(int c) const { Mat m = *this; m.p = p + c * cs; if (bits) m.bits = bits + c * bcs; return m; }
  Mat shared() const { Mat m = *this; m.cs = 0; m.bcs = 0; return m; }
};
struct NetRef {
  float* base = nullptr;  // params of run 0, member 0
  long rs = 0;            // run stride
  long g_off = 0;         // offset of the family's gradients inside a run's gradient slab
  const NetLayout* lay = nullptr;
  int nz1 = 1;            // members driven by a launch
};

struct ProfEntry {
  std::string name;
  double flops;
  double bytes = 0;   // algorithmic HBM bytes of the launch (operands read once + results written once)
  hipEvent_t a, b;
};

struct Buffer {             // HBM-resident replay buffer (buffer/buffer.py)
  int dev = 0, od = 0, ad = 0, OP = 0, AP = 0;
  long n = 0;
  float *obs = nullptr, *nobs = nullptr, *act = nullptr, *rew = nullptr, *term = nullptr;
  long long* idx = nullptr; long idx_cap = 0;
  unsigned long long counter = 0;
  float absmax = -1.f; unsigned long long absmax_gen = ~0ull;   // max |obs|, |next_obs|, |act| of generation absmax_gen (engines at precision 1 ask for it)
  unsigned long long gen = 0;   // bumped by every orl_buffer_load: engines re-capture graphs that hold the old dataset pointers / size
  ~Buffer();
};

struct DY;  // backward seed description (engine.hip)

struct Engine {
  orl_config cfg;
  int dev = 0;
  hipStream_t stream = nullptr;
  // A second stream for launches that do not depend on each other (CQL: the target critics' forward next to the critics' forward, both fed
  // by the actor pass before them): fork_side() makes `stream` the side stream until fork_main(); fork_join() makes the main stream wait for
  // the side work.  Inside a graph capture the three calls become a fork / join of the captured graph.  OFF by default (ORL_FORK=1 enables it):
  // measured 10 us per step SLOWER at 1 - 8 runs per engine -- a fork / join of graph branches costs more than the 12 us launch it hides.
  hipStream_t side_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool fork_on = false, forked = false;
  int fork_side();
  void fork_main() { if (forked) std::swap(stream, side_stream); }
  int fork_join();
  int R = 1, B = 0, N = 0, od = 0, ad = 0, OP = 0, AP = 0, XP = 0, L = 0, K = 2;
  NetLayout lay[ORL_NUM_NETS];
  float* arena = nullptr;      // [R][P_train] then [R][P_tgt]
  long P_train = 0, P_tgt = 0;
  long net_off[ORL_NUM_NETS];
  bool net_is_target[ORL_NUM_NETS];
  float* adam_m = nullptr;
  float* adam_v = nullptr;
  float* grads = nullptr;      // [R][max_slab][P_train]
  int max_slab = 128;          // split-K slabs per parameter (the fused layer-0 weight gradient writes one per row tile)
  int ksplit_cap = 32;         // cap of the split-K factor chosen for a stand-alone wgrad launch
  RunScalars* scalars = nullptr;
  Hyper* hyper = nullptr;
  Hyper hyper_host;
  unsigned long long* gstep = nullptr;
  // CQL advances the counter without a k_tick node: k_prepare reads gstep_pre and copies it to gstep (read by every later kernel of the
  // step), the loss kernel's finishing lane writes gstep_pre = gstep + 1 (kernels.h: PrepP::gstep_publish, CqlLossP::gstep_next)
  unsigned long long* gstep_pre = nullptr;
  bool tick_folded = false;
  bool fuse_small = true;      // few-runs fusions (sampling epilogue of the one-launch forward, step counter without k_tick, fused actor phase); ORL_FUSE_SMALL=0: the separate launches (A/B runs, cross-check tests)
  unsigned long long step_host = 0;
  float *metrics_last = nullptr, *metrics_sum = nullptr;
  int nm = 0;
  std::vector<std::string> metric_names;
  Buffer* buf = nullptr;       // attached replay buffer (not owned)
  long long* d_idx = nullptr;  // [R][B] minibatch indices of the current step (recorded by k_prepare's rewards job)
  unsigned long long buf_gen = 0;   // Buffer::gen the captured graphs were built against
  void drop_graphs();
  // split-K slab table of the last adam() launch per net (orl_debug_grads sums the slabs the way k_adam does)
  std::vector<std::pair<long, int>> last_segs[ORL_NUM_NETS];
  std::vector<void*> allocs;
  std::map<std::string, Mat> ws;
  std::map<std::string, long> ws_len;
  std::set<const void*> vals_dead;   // activations whose VALUES the forward pass did not store (mask bits only); consumers of the values fail
  std::set<const void*> bits_live;   // mask-bit buffers whose producer launch emitted them this step (decided on the host)
  struct Tap { Mat m; long rows; int cols; };
  std::map<std::string, Tap> taps;
  struct NoiseSlot { std::string name; int kind; int rows; int cols = 0; };      // cols 0 = act_dim
  std::vector<NoiseSlot> noise_slots;
  // k_prepare job list (algorithms that use it skip the separate gather / noise / assemble launches)
  struct PrepSpec { PrepJob job; int need_sampling; int need_devnoise; };   // -1 any, 0 no, 1 yes
  std::vector<PrepSpec> prep;
  void add_prep(const Mat& dst, int row0, int col0, int rows, int width, int src, int rep, int mod, int ncopy, const Mat* buf,
                unsigned stream_id, int need_sampling, int need_devnoise, int src_row0 = 0);
  int enqueue_prepare(bool sampling, bool devnoise);
  bool prof_on = false;
  std::vector<ProfEntry> prof;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  hipGraphExec_t graph_exec[2] = {nullptr, nullptr};
  hipGraph_t graph[2] = {nullptr, nullptr};
  bool use_graph = true;
  bool force_scalar = false;   // debug: disable the vector loaders
  int loss_nblk = 1;
  bool elide_top = true;       // many-row single-output nets: keep the top hidden activation out of HBM (ORL_WS_KEEP_H1=1 stores it)
  bool small_fwd_on = true;      // up to small_fwd_max_rows batched rows a [in -> 256 -> 256 -> out] forward is ONE launch (small_fwd.h; ORL_SMALL_FWD=0: tiled / weight-stationary launches)
  long small_fwd_max_rows = 8192;   // measured: 8 / 16 runs per engine +4 % against the weight-stationary launch at 4096 .. 8192 rows, 32 runs (16384 rows) -1 % (ORL_SMALL_FWD_MAX)
  long ws_fwd_min_rows = 4096, ws_bwd_min_rows = 4096;   // batched rows from which the weight-stationary forward / dgrad kernels replace the tiled launches (ORL_WS_FWD_MIN / ORL_WS_BWD_MIN)
  long ws_dgrad_plain_min_rows = 40000;   // batched rows from which a middle layer's dgrad (+ dW0) runs on the plain weight-stationary kernel (ORL_WS_DGRAD_PLAIN_MIN overrides)
  long ws_wgrad_min_rows = 15000;   // batched rows from which the output-stationary wgrad kernel is used (ORL_WS_WGRAD_MIN overrides; round 4: 40 000 -> 15 000, the 15 872 critic rows of ONE run: 1 run +4.3 %, 2 runs +1.7 %, fp32 +7.4 % / +0.7 %)
  int ws_wgrad_min_m = 1024;        // ... and rows PER NET: a 256-row net is 8 row groups behind a 256 KB slab write -- the tiled wgrad is faster there
                                    // (IQL / TD3+BC at 128 runs: +1.2 %; ORL_WS_WGRAD_MIN_M overrides)
  int ws_wgrad_slab_cap = 128;      // slabs (= workgroups per net) of the output-stationary wgrad: Adam reads every one of them (ORL_WS_WGRAD_SLABS)
  bool ws_wgrad_rows_ok(int M, int nz) const { return (long)M * nz >= ws_wgrad_min_rows && M >= ws_wgrad_min_m; }
  bool use_ws = true;          // weight-stationary kernels (csrc/ws_gemm.h); ORL_WS=0 keeps everything on the tiled kernels (tests)
  bool use_ws32 = true;        // ... and their exact-fp32 variants at precision 0 (ORL_WS32=0: tiled fp32 kernels only)
  WsGeom ws_geo;               // workgroups per net / CUs per launch of the weight-stationary kernels (orl_config::ws_one_round, ws_cus)
  bool ws_precision_ok() const { return use_ws && (cfg.precision == 1 || use_ws32); }
  // precision 2 = fp32-class arithmetic at more than the fp32 MFMA rate: the launches that have a three-plane flavour (the fused first +
  // second layer forward of a two-hidden-layer net from 4096 batched rows, the top-layer dgrad from mask bits -- fused with the layer-0 wgrad or
  // storing --, the output-stationary top-layer wgrad with derived tail gradients, and for deeper nets the forward of a layer fed from HBM and
  // the plain dgrad / wgrad of a middle layer: ws_fwd3 / ws_dgrad3 / ws_wgrad_kernel<5> / ws_wgrad3p) multiply three fp16 planes per operand
  // (six resp. three products), every other launch is the exact-fp32 kernel of precision 0.  The gradient scales and the
  // fp16 range watch of precision 1 apply (the planes are fp16); fp32 kernels ignore the scales.
  bool split_scales() const { return cfg.precision >= 1; }
  // precision of a tiled launch (gemm.h: P_F32 / P_SPLIT / P_SPLIT3) and the exact-fp32 flag of a weight-stationary / few-rows launch WITHOUT
  // a three-plane flavour.  The tiled kernel has the three-plane multiply as a template flavour (ORL_P3 bit 3), but precision 2 keeps its tiled
  // launches on the exact-fp32 MFMA by default: they are staging- and latency-bound, three planes cost them LDS space and split work and
  // measured SLOWER than fp32 (128 runs: CQL 33.2k -> 30.8k, IQL 137.8k -> 132.1k, TD3+BC 184.7k -> 179.5k, EDAC 14.1k -> 13.5k).
  int mm_prec() const { return cfg.precision == 2 ? ((p3_mask & 8) ? 2 : 0) : cfg.precision; }
  int ws_f32() const { return cfg.precision != 1; }
  int p3_mask = 7;                                                  // ORL_P3: bit 0 forward, 1 dgrad, 2 wgrad, (3: tiled launches, off by default) (lab: single kernels against their fp32 twins)
  bool p3(int bit) const { return cfg.precision == 2 && (p3_mask & bit); }
  float* ws_dump = nullptr;                                         // precision 2: scratch lines of ws_fwd3_kernel (WS_DUMP_SLOTS x WS_N floats)
  // split precision: per-run dynamic power-of-two scale of the gradient matrices of the backward pass being enqueued (k_grad_scale);
  // one slot per backward pass of a step, fixed order, so a captured graph replays with the same slots
  float* gscale_buf = nullptr;       // [GSCALE_SLOTS][R]
  enum { GSCALE_SLOTS = 24 };
  int gscale_next = 0;
  const float* cur_gscale = nullptr; // scale array [R] of the current backward pass (null: precision 0 or outside a backward pass)
  const float* grad_scale(const Mat& seed, int rows, int cols, int nets, const char* tag);
  float* gscale_slot();              // next slot for a seed kernel that publishes the scale itself (GradScaleP-free path); null at precision 0
  float* gscale_inv_b = nullptr;     // [R] constant scale of seeds whose entries are +-1/B (actor-loss dq)
  unsigned int* cql_ticket = nullptr;  // [R] arrival counters of k_cql_loss_rows
  int lab_slot = 0;                    // lab builds: which stamp block the next one-launch forward writes (reset per step)
  float* aloss_part = nullptr;         // [R][SB_MAXGROUPS][2] per-row-group loss sums of the fused actor update (small_bwd.h)
  // health (include/orl_engine.h: ORL_HEALTH_*): device words raised by kernels (k_adam: non-finite gradient; k_range_scan), the sticky
  // host copy that also holds what the host finds in the metrics it reads back, and the matrices of the last enqueued step that enter
  // the MFMAs as fp16 planes with operand scale 1 (registered by linear_fwd / mlp_forward while the step is enqueued or captured)
  unsigned int* health = nullptr;      // [R]
  std::vector<unsigned int> health_host;
  struct RangeWatch { Mat m; int rows, cols, nets; std::string what; };
  std::map<const float*, RangeWatch> range_watch;
  void watch_range(const Mat& m, int rows, int cols, int nets, const char* what);
  int range_scan();                    // enqueues k_range_scan for every watched matrix and the parameters (precision 1)
  int health_update(const float* metrics, long steps_done, unsigned int* any);   // after a stream sync; metrics: [R][nm] values just read back (or null)
  long steps_since_scan = 0;           // precision 1: the operands of the last step are range-scanned every RANGE_SCAN_EVERY steps (and when a run turns non-finite)
  enum { RANGE_SCAN_EVERY = 256 };

  ~Engine();
  int init(const orl_config& c);
  Mat alloc(const std::string& name, long rows, int pitch, int nets = 1);
  Mat& W(const std::string& n) { return ws.at(n); }
  float* raw_alloc(size_t bytes);
  float* net_ptr(int run, int net) const;
  NetRef net_ref(int net, int members) const;
  MetricsP mp() const { return MetricsP{metrics_last, metrics_sum, nm}; }

  // launch helpers (enqueue on stream)
  int linear_fwd(const Mat& X, int M, const NetRef& nr, int layer, const Mat& Y, int epi, const Mat* maskH, const char* tag,
                 int in_row0 = 0, int in_rows = -1, const Mat* tail_out = nullptr, bool* tail_fused = nullptr,
                 const Mat* fuse_X0 = nullptr, const char* tag0 = nullptr, const float* x_dscale = nullptr);
  int tq_scratch_nets = 2;
  int linear_dgrad(const DY& dy, int M, const NetRef& nr, int layer, int col0, int ncols, const Mat* maskH, const Mat& dX, const char* tag,
                   const Mat* w0_X = nullptr, bool store_dx = true, int* w0_slabs = nullptr);
  // recompute_X0 (layer 1 only): X = hs[0] was not stored by the forward pass (vals_dead); the plain output-stationary kernel rebuilds it
  // from the net's input rows (WsWgradP::X0)
  int linear_wgrad(const DY& dy, const Mat& X, int M, const NetRef& nr, int layer, int ksplit, int slab0, bool with_bias,
                   const char* tag, int in_row0 = 0, int in_rows = -1, bool* fuse_tail = nullptr, int* slabs_out = nullptr,
                   const float* x_dscale = nullptr, const Mat* recompute_X0 = nullptr);
  // Three-layer nets on the weight-stationary kernels: do not store the first hidden activation, let the middle layer's wgrad rebuild it from
  // the 24-column input (ws_wgrad_kernel<4>, ORL_WS_RECOMPUTE_H0=1).  OFF by default -- measured at 128 runs (round 4, one call): the forward
  // drops 957 -> 790 us, but the wgrad's row loop is not HBM-bound enough to absorb 12 MFMAs + the ReLU / split of a 32 x 32 block per wave
  // and group: 770 -> 1225 us (split, 42 spilled VGPRs next to the 128 accumulators; exact fp32, no spills: step 14.68 -> 14.86 ms);
  // whole step 23.46k -> 22.78k steps/s.
  bool recompute_h0 = false;
  bool recompute_h0_ok(const NetLayout& l, int layer, int M, int nz) const {
    return recompute_h0 && layer == 1 && l.L >= 3 && !l.ens && ws_precision_ok() && ws_wgrad_rows_ok(M, nz);
  }
  int adam(int net, int nnets, int lr_slot, const std::vector<std::pair<long, int>>& segs, int target_net, unsigned long long t_div = 1);
  int polyak(int target_net, int src_net, int nnets);
  void prof_begin(const char* name, double flops, double bytes = 0);
  void prof_end();
  int assemble(const Mat& obs, const Mat* act, const Mat& X, int row0, int rows, int rep);
  // jobs (optional): tanh-Gaussian sampling jobs on this pass's head rows; the one-launch forward runs them as its epilogue (*jobs_done = true),
  // any other path leaves them to the caller's k_tanh_sample launch
  int mlp_forward(const Mat& X, int M, const NetRef& nr, std::vector<Mat>& hs, const Mat& out, const char* tag,
                  const SampleJob* jobs = nullptr, int njobs = 0, bool* jobs_done = nullptr);
  // q = net(X) and G = dq / dX[:, gc0 : gc0 + gn] for a unit seed in ONE launch (small_fwd.h, QG mode); *done = false when the shape is not
  // served (the caller then runs forward + loss + backward launches)
  int mlp_qgrad(const Mat& X, int M, const NetRef& nr, const Mat& q, const Mat& G, int gc0, int gn, const char* tag, bool* done);
  // a pass nobody differentiates (target nets, the actor's action proposals for the critic loss): the weight-stationary launches keep
  // hidden activations they do not need themselves out of HBM (they are marked dead: a later reader fails loudly)
  int mlp_forward_only(const Mat& X, int M, const NetRef& nr, std::vector<Mat>& hs, const Mat& out, const char* tag,
                       const SampleJob* jobs = nullptr, int njobs = 0, bool* jobs_done = nullptr);
  bool fwd_only = false;       // set while mlp_forward_only enqueues
  // the same with nn.Dropout(p) behind every hidden ReLU (keep masks in `masks[i]`, [R][M][H_i]); layer by layer on the tiled kernels
  int mlp_forward_dropout(const Mat& X, int M, const NetRef& nr, std::vector<Mat>& hs, const Mat& out, const char* tag, float p,
                          const std::vector<Mat>& masks);
  int scale_inplace(const Mat& m, int rows, int cols, int nets, float s, const Mat* mask, const char* tag);
  bool no_ws = false;          // set while a dropout forward is enqueued: its layers must exist one by one
  float bwd_scale = 1.0f;      // mlp_backward: every masked dz of the pass is multiplied by this (1 / (1 - p) of a dropout backbone)

  int enqueue_sample();
  int enqueue_noise();
  int enqueue_step(int variant);
  int step_variant() const;
  int n_variants() const;
  int build_common();
  int cql_build(); int cql_step();
  int iql_build(); int iql_step();
  int td3bc_build(); int td3bc_step(bool actor_step);
  int edac_build(); int edac_step();
  int sac_build(); int sac_step();
  int mcq_build(); int mcq_step();
};

}  // namespace orl
