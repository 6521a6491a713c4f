// sample.h -- tanh-Gaussian sampling shared by k_tanh_sample (kernels.h) and the sampling epilogue of small_fwd_kernel (small_fwd.h).
// Reference: modules/dist_module.py:117-127 (TanhDiagGaussian.forward: mu, clamp(log sigma, -5, 2)), :17-42 (TanhNormalWrapper.rsample /
// log_prob: tanh(u), log N(u; mu, sigma) - log(1 - tanh(u)^2 + 1e-6)).
#pragma once
#include <hip/hip_runtime.h>

namespace orl {

#define ORL_LOG_SQRT_2PI 0.91893853320467274178f
struct SampleJob {
  int head_row0;     // first base row inside head
  int rows;          // output rows
  int rep;           // output row j uses base row head_row0 + j / rep
  const float* eps;  long eps_rs;   // [R][rows][A] or null (deterministic)
  float* dst;        long dst_rs; int dst_pitch, dst_col, dst_row0;  // actions -> dst[(dst_row0+j)*pitch + col + a]
  float* logp;       long logp_rs;  // [R][rows] or null
};

// one action component: a = tanh(mu + sigma eps) and its term of the row's log-probability (the caller sums the A terms)
__device__ __forceinline__ float orl_tanh_sample(float mu, float ls_raw, float eps, float& act) {
  const float ls = fminf(fmaxf(ls_raw, -5.0f), 2.0f);
  const float sg = expf(ls);
  const float u = mu + sg * eps;
  act = tanhf(u);
  const float dm = u - mu;
  return (-(dm * dm) / (2.0f * (sg * sg)) - ls - ORL_LOG_SQRT_2PI) - logf((1.0f - act * act) + 1e-6f);
}

}  // namespace orl
