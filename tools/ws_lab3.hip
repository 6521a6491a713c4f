// ws_lab3.hip — correctness + timing lab for ws_wgrad_kernel (csrc/ws_gemm.h); not part of the product.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "ws_wgrad_experiment.h"
using namespace orl;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static unsigned rs = 11;
static float rnd() { rs = rs * 1664525u + 1013904223u; return ((rs >> 8) / 8388608.0f) - 1.0f; }
int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 1088, nz = argc > 2 ? atoi(argv[2]) : 2, SL = 128;
  const bool check = M * (long)nz <= 40000;
  const long P = 65536 + 256 + 256 + 1, nH = (long)M * 256;
  std::vector<float> h0(nH * nz), h1(nH * nz), hdq((long)M * nz), hwt(256 * nz);
  std::vector<unsigned> hab((long)M * 8 * nz, 0u);
  for (auto& v : h0) { const float t = rnd(); v = t > 0.f ? t : 0.f; }
  for (long i = 0; i < nH * nz; ++i) { const float t = rnd(); h1[i] = t > 0.f ? t : 0.f; if (h1[i] > 0.f) hab[(i / 256) * 8 + ((i % 256) >> 5)] |= 1u << (i & 31); }
  for (auto& v : hdq) v = rnd() * 0.01f; for (auto& v : hwt) v = rnd() * 0.0625f;
  unsigned* dab; float *d0, *d1, *ddq, *dwt, *dG;
  CK(hipMalloc(&dab, 4L * M * 8 * nz)); CK(hipMalloc(&d0, 4 * nH * nz)); CK(hipMalloc(&d1, 4 * nH * nz)); CK(hipMalloc(&ddq, 4L * M * nz)); CK(hipMalloc(&dwt, 4 * 256 * nz));
  CK(hipMalloc(&dG, 4L * SL * P * nz)); CK(hipMemset(dG, 0, 4L * SL * P * nz));
  CK(hipMemcpy(dab, hab.data(), 4L * M * 8 * nz, hipMemcpyHostToDevice)); CK(hipMemcpy(d0, h0.data(), 4 * nH * nz, hipMemcpyHostToDevice));
  CK(hipMemcpy(d1, h1.data(), 4 * nH * nz, hipMemcpyHostToDevice)); CK(hipMemcpy(ddq, hdq.data(), 4L * M * nz, hipMemcpyHostToDevice));
  CK(hipMemcpy(dwt, hwt.data(), 4 * 256 * nz, hipMemcpyHostToDevice));
  WsWgradP p; memset(&p, 0, sizeof(p));
  p.abits = dab; p.ab_s1 = (long)M * 8; p.ab_g = 8; p.dq = ddq; p.dq_s1 = M; p.dq_sm = 1;
  p.H0 = d0; p.h0_s1 = nH; p.h0_pitch = 256; p.wt = dwt; p.wt_s1 = 256;
  const bool tails = argc > 3 && atoi(argv[3]) == 1;
  if (tails) { p.H1 = d1; p.h1_s1 = nH; p.h1_pitch = 256; p.dwt = dG + 65536 + 256; p.dbt = dG + 65536 + 512; p.o_s1wt = p.o_s1bt = (long)SL * P; }
  p.dW = dG; p.db = dG + 65536;
  p.o_s1w = p.o_s1b = (long)SL * P; p.o_ks = P; p.M = M; p.nz1 = nz;
  if (!ws_wgrad_supported(p, 256, 256)) { printf("not supported\n"); return 1; }
  const int per_z = ws_dgrad_blocks(M, nz, SL);
  CK(launch_ws_wgrad(p, nz, per_z, 0)); CK(hipDeviceSynchronize());
  if (check) {
    std::vector<float> G((long)SL * P * nz); CK(hipMemcpy(G.data(), dG, 4L * SL * P * nz, hipMemcpyDeviceToHost));
    double e[4] = {0, 0, 0, 0}, sc[4] = {0, 0, 0, 0};
    for (int z = 0; z < nz; ++z) {
      std::vector<double> ref(P, 0.0);
      for (int m = 0; m < M; ++m) {
        const double dq = hdq[(long)z * M + m];
        const float* a0 = &h0[z * nH + (long)m * 256]; const float* a1 = &h1[z * nH + (long)m * 256];
        for (int k = 0; k < 256; ++k) {
          ref[65536 + 256 + k] += dq * a1[k];
          if (a1[k] > 0.f) { const double dz = dq * hwt[z * 256 + k]; ref[65536 + k] += dz; for (int n = 0; n < 256; ++n) ref[k * 256 + n] += dz * a0[n]; }
        }
        ref[65536 + 512] += dq;
      }
      for (long i = 0; i < P; ++i) {
        double got = 0; for (int s = 0; s < per_z; ++s) got += G[((long)z * SL + s) * P + i];
        const int w = i < 65536 ? 0 : (i < 65536 + 256 ? 1 : (i < 65536 + 512 ? 2 : 3));
        e[w] = std::max(e[w], std::fabs(got - ref[i])); sc[w] = std::max(sc[w], std::fabs(ref[i]));
      }
    }
    printf("M=%d nz=%d blocks/z=%d  rel err: dW1 %.2e  db1 %.2e  dw_tail %.2e  db_tail %.2e\n", M, nz, per_z, e[0] / sc[0], e[1] / sc[1], e[2] / sc[2], e[3] / sc[3]);
  }
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) CK(launch_ws_wgrad(p, nz, per_z, 0));
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < 20; ++i) CK(launch_ws_wgrad(p, nz, per_z, 0));
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double us = ms / 20 * 1e3;
  printf("ws_wgrad M=%d nz=%d: %.1f us  %.1f TF(alg)  %.0f GB/s (h0 + h1 read)\n", M, nz, us, 2.0 * M * 65536.0 * nz / us * 1e-6, 8.0 * nH * nz / us * 1e-3);
  return 0;
}
