// engine.h — host-side engine state: parameter arenas, workspaces, launch helpers.
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/orl_engine.h"
#include "gemm.h"
#include "kernels.h"

namespace orl {

void set_error(const std::string& msg);
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

// A network = MLP backbone of L ReLU layers + one linear "tail" (critic last / actor head).
struct NetLayout {
  bool present = false;
  int in_dim = 0, L = 0, out_dim = 0;
  int H[ORL_MAX_HIDDEN + 1] = {0};       // hidden widths
  long w_off[ORL_MAX_HIDDEN + 1] = {0};  // layers 0..L-1 hidden, layer L = tail
  long b_off[ORL_MAX_HIDDEN + 1] = {0};
  long extra_off = -1;                   // IQL sigma_param
  long size = 0;
  std::vector<TensorInfo> tensors;
  int layer_in(int l) const { return l == 0 ? in_dim : H[l - 1]; }
  int layer_out(int l) const { return l == L ? out_dim : H[l]; }
};

// matrix / vector views batched over (run, net)
struct Mat {
  float* p = nullptr;
  long rs = 0, cs = 0;  // run stride, net stride (elements)
  int pitch = 0;
  Mat rows(long r0) const { Mat m = *this; m.p = p + r0 * pitch; return m; }
  Mat cols(int c0) const { Mat m = *this; m.p = p + c0; return m; }
  Mat net(int c) const { Mat m = *this; m.p = p + c * cs; return m; }
};
struct NetRef {           // parameters of a (run, net)-batched family with identical layout
  float* base = nullptr;  // params of run 0, net 0
  long rs = 0, cs = 0;
  const NetLayout* lay = nullptr;
};

struct ProfEntry {
  std::string name;
  double flops;
  hipEvent_t a, b;
};

struct Engine {
  orl_config cfg;
  int dev = 0;
  hipStream_t stream = nullptr;
  int R = 1, B = 0, N = 0, od = 0, ad = 0, OP = 0, AP = 0, XP = 0, L = 0, Hlast = 0;
  NetLayout lay[ORL_NUM_NETS];
  // arenas
  float* arena = nullptr;      // [R][P_train] then [R][P_tgt]
  bool arena_owned = false;
  long P_train = 0, P_tgt = 0;
  long net_off[ORL_NUM_NETS];  // offset inside the run's trainable (or target) block
  bool net_is_target[ORL_NUM_NETS];
  float* adam_m = nullptr;     // [R][P_train]
  float* adam_v = nullptr;
  float* grads = nullptr;      // [R][max_slab][P_train]
  int max_slab = 32;
  RunScalars* scalars = nullptr;
  Hyper* hyper = nullptr;
  Hyper hyper_host;
  unsigned long long* gstep = nullptr;
  unsigned long long step_host = 0;
  float *metrics_last = nullptr, *metrics_sum = nullptr;
  int nm = 0;
  std::vector<std::string> metric_names;
  // replay buffer (HBM-resident SoA)
  float *d_obs = nullptr, *d_nobs = nullptr, *d_act = nullptr, *d_rew = nullptr, *d_term = nullptr;
  long n_data = 0;
  long long* d_idx = nullptr;  // [R][B]
  // workspace
  std::vector<void*> allocs;
  std::map<std::string, Mat> ws;      // named buffers
  std::map<std::string, long> ws_len; // floats per run
  // debug taps: name -> (matrix, rows, cols)
  struct Tap { Mat m; long rows; int cols; };
  std::map<std::string, Tap> taps;
  // profiling
  bool prof_on = false;
  std::vector<ProfEntry> prof;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  // graph
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  bool use_graph = true;

  ~Engine();
  int init(const orl_config& c);
  Mat alloc(const std::string& name, long rows, int pitch, int nets = 1);
  float* raw_alloc(size_t bytes);
  float* net_ptr(int run, int net) const;
  NetRef net_ref(int net) const;  // family starting at `net` (critic1 -> {critic1,critic2})

  // launch helpers (enqueue on stream)
  int linear_fwd(const Mat& X, int M, const NetRef& nr, int layer, const Mat& Y, bool relu, int nz1, const char* tag);
  int linear_dgrad(const Mat& dY, int M, const NetRef& nr, int layer, int col0, int ncols, const Mat* maskH,
                   const Mat& dX, int nz1, const char* tag, const Mat* rank1_H = nullptr, const Mat* rowv = nullptr);
  int linear_wgrad(const Mat& dY, const Mat& X, int M, const NetRef& nr, int layer, long g_net_off, int ksplit,
                   int nz1, const char* tag, const Mat* rank1_H = nullptr, const Mat* rowv = nullptr);
  int adam(int net, int nnets, int lr_slot, const std::vector<std::pair<long, int>>& segs, bool polyak, int target_net,
           unsigned long long t_div = 1);
  void prof_begin(const char* name, double flops);
  void prof_end();

  int enqueue_sample(const long long* idx_dev);
  int enqueue_noise();
  int enqueue_step();
  int cql_build();
  int cql_step();
};

}  // namespace orl
