// ws_lab2.hip — correctness + timing lab for ws_dgrad_w0_kernel (csrc/ws_gemm.h); not part of the product.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../offlinerl-kit_amd/csrc/ws_gemm.h"
using namespace orl;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static unsigned rs = 7;
static float rnd() { rs = rs * 1664525u + 1013904223u; return ((rs >> 8) / 8388608.0f) - 1.0f; }
static unsigned rndu() { rs = rs * 1664525u + 1013904223u; unsigned a = rs; rs = rs * 1664525u + 1013904223u; return (a & 0xffff0000u) | (rs >> 16); }

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 1088, nz = argc > 2 ? atoi(argv[2]) : 2, xp = 24, in0 = 23, SL = 128;
  const bool check = M * (long)nz <= 40000;
  const long P = 256 * in0 + 256;     // slab size: dW0 then db0
  std::vector<unsigned> hab((long)M * 8 * nz), hxb((long)M * 8 * nz);
  std::vector<float> hdq((long)M * nz), hwt(256 * nz), hW(65536L * nz), hX((long)M * xp * nz);
  for (auto& v : hab) v = rndu(); for (auto& v : hxb) v = rndu();
  for (auto& v : hdq) v = rnd() * 0.01f; for (auto& v : hwt) v = rnd() * 0.0625f; for (auto& v : hW) v = rnd() * 0.0625f;
  for (long i = 0; i < (long)M * nz; ++i) for (int c = 0; c < xp; ++c) hX[i * xp + c] = c < in0 ? rnd() : 0.f;
  unsigned *dab, *dxb; float *ddq, *dwt, *dW, *dX, *dG;
  CK(hipMalloc(&dab, 4L * M * 8 * nz)); CK(hipMalloc(&dxb, 4L * M * 8 * nz)); CK(hipMalloc(&ddq, 4L * M * nz)); CK(hipMalloc(&dwt, 4 * 256 * nz));
  CK(hipMalloc(&dW, 4 * 65536L * nz)); CK(hipMalloc(&dX, 4L * M * xp * nz)); CK(hipMalloc(&dG, 4L * SL * P * nz)); CK(hipMemset(dG, 0, 4L * SL * P * nz));
  CK(hipMemcpy(dab, hab.data(), 4L * M * 8 * nz, hipMemcpyHostToDevice)); CK(hipMemcpy(dxb, hxb.data(), 4L * M * 8 * nz, hipMemcpyHostToDevice));
  CK(hipMemcpy(ddq, hdq.data(), 4L * M * nz, hipMemcpyHostToDevice)); CK(hipMemcpy(dwt, hwt.data(), 4 * 256 * nz, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW, hW.data(), 4 * 65536L * nz, hipMemcpyHostToDevice)); CK(hipMemcpy(dX, hX.data(), 4L * M * xp * nz, hipMemcpyHostToDevice));
  WsDgradP p; memset(&p, 0, sizeof(p));
  p.abits = dab; p.ab_s1 = (long)M * 8; p.ab_g = 8; p.xbits = dxb; p.xb_s1 = (long)M * 8; p.xb_g = 8;
  p.dq = ddq; p.dq_s1 = M; p.dq_sm = 1; p.wt = dwt; p.wt_s1 = 256; p.W = dW; p.w_s1 = 65536; p.w_sn = 1; p.w_sk = 256; p.X = dX; p.x_s1 = (long)M * xp; p.x_pitch = xp; p.in0 = in0;
  p.w0_out = dG; p.b0_out = dG + 256 * in0; p.o_s1 = (long)SL * P; p.ob_s1 = (long)SL * P; p.o_ks = P; p.o_sr = in0; p.M = M; p.nz1 = nz;
  if (!ws_dgrad_supported(p, 256, 256)) { printf("not supported\n"); return 1; }
  const int per_z = ws_dgrad_blocks(M, nz, SL);
  CK(launch_ws_dgrad_w0(p, nz, per_z, 0)); CK(hipDeviceSynchronize());
  if (check) {
    std::vector<float> G((long)SL * P * nz); CK(hipMemcpy(G.data(), dG, 4L * SL * P * nz, hipMemcpyDeviceToHost));
    double emax = 0, smax = 0;
    for (int z = 0; z < nz; ++z) {
      std::vector<double> ref(P, 0.0), Bp(65536);
      for (int k = 0; k < 256; ++k) for (int n = 0; n < 256; ++n) Bp[k * 256 + n] = (double)hwt[z * 256 + k] * hW[z * 65536L + k * 256 + n];
      for (int m = 0; m < M; ++m) {
        const unsigned* a = &hab[((long)z * M + m) * 8]; const unsigned* x = &hxb[((long)z * M + m) * 8];
        for (int n = 0; n < 256; ++n) {
          if (!((x[n >> 5] >> (n & 31)) & 1u)) continue;
          double s = 0; for (int k = 0; k < 256; ++k) if ((a[k >> 5] >> (k & 31)) & 1u) s += Bp[k * 256 + n];
          const double dz = s * hdq[(long)z * M + m];
          for (int c = 0; c < in0; ++c) ref[n * in0 + c] += dz * hX[((long)z * M + m) * xp + c];
          ref[256 * in0 + n] += dz;
        }
      }
      for (long i = 0; i < P; ++i) {
        double got = 0; for (int s = 0; s < per_z; ++s) got += G[((long)z * SL + s) * P + i];
        emax = std::max(emax, std::fabs(got - ref[i])); smax = std::max(smax, std::fabs(ref[i]));
      }
    }
    printf("M=%d nz=%d blocks/z=%d  max|d(dW0,db0)|=%.3e (scale %.3e)  rel %.2e\n", M, nz, per_z, emax, smax, emax / smax);
  }
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) CK(launch_ws_dgrad_w0(p, nz, per_z, 0));
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < 20; ++i) CK(launch_ws_dgrad_w0(p, nz, per_z, 0));
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double us = ms / 20 * 1e3;
  printf("ws_dgrad_w0 M=%d nz=%d: %.1f us  %.1f TF(alg, dgrad+w0)\n", M, nz, us, 2.0 * M * (65536.0 + 256 * 24) * nz / us * 1e-6);
  return 0;
}
