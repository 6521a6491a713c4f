// scalars.h -- per-run device scalars and the scalar Adam step, shared by the loss kernels (kernels.h) and the fused actor update
// (small_bwd.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace orl {

// per-run scalars kept on the device so captured graphs replay without host patching
struct RunScalars {
  float log_alpha, la_m, la_v;          // SAC temperature + its Adam moments (run_cql.py:102-103)
  float cql_log_alpha, cla_m, cla_v;    // CQL Lagrange multiplier (cql.py:57-58)
  float alpha;                          // alpha used by the NEXT actor loss (sac.py:46 / cql.py:106)
  float alpha_bwd;                      // alpha the current actor backward must use (pre-update value)
  float cons_scale;                     // cql_alpha seen by the critic gradients (cql.py:170-178)
  float last_actor_loss;                // TD3BC _last_actor_loss (td3.py:59)
  float pad[6];
};

struct Hyper {                          // mutable hyper-parameters (orl_set_lr)
  float lr[8];
};

__device__ __host__ inline float orl_pow2_scale(float a) {     // 2^(3 - floor(log2 a)); 1 for 0 / non-finite
  if (!(a > 0.f) || !(a < 3.0e38f)) return 1.f;
  int e = ilogbf(a);
  e = e < -100 ? -100 : (e > 100 ? 100 : e);
  return ldexpf(1.0f, 3 - e);
}
// scalar Adam (log_alpha, cql_log_alpha): torch.optim.Adam single-tensor semantics
__device__ inline void adam_scalar(float& p, float& m, float& v, float g, float lr, float b1, float b2, float eps,
                                   unsigned long long t) {
  m = m + (g - m) * (1.0f - b1);
  v = v * b2 + (1.0f - b2) * g * g;
  const double bc1 = 1.0 - pow((double)b1, (double)t), bc2 = 1.0 - pow((double)b2, (double)t);
  const float step = (float)((double)lr / bc1), bc2s = (float)sqrt(bc2);
  p -= step * (m / (sqrtf(v) / bc2s + eps));
}

}  // namespace orl
