// ws_lab0.hip — correctness + timing lab for ws_fwd0_kernel (csrc/ws_gemm.h); not part of the product.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "ws_fwd0_experiment.h"
using namespace orl;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
static unsigned rs = 5;
static float rnd() { rs = rs * 1664525u + 1013904223u; return ((rs >> 8) / 8388608.0f) - 1.0f; }
int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 1088, nz = argc > 2 ? atoi(argv[2]) : 2, xp = 24, in0 = 23;
  const bool check = M * (long)nz <= 40000;
  std::vector<float> hX((long)M * xp * nz), hW(256L * in0 * nz), hb(256 * nz);
  for (long i = 0; i < (long)M * nz; ++i) for (int c = 0; c < xp; ++c) hX[i * xp + c] = c < in0 ? rnd() : 0.f;
  for (auto& v : hW) v = rnd() * 0.2f; for (auto& v : hb) v = rnd() * 0.2f;
  float *dX, *dW, *db, *dY; unsigned* dmb;
  CK(hipMalloc(&dX, 4L * M * xp * nz)); CK(hipMalloc(&dW, 4 * 256L * in0 * nz)); CK(hipMalloc(&db, 4 * 256 * nz)); CK(hipMalloc(&dY, 4L * M * 256 * nz)); CK(hipMalloc(&dmb, 4L * M * 8 * nz));
  CK(hipMemcpy(dX, hX.data(), 4L * M * xp * nz, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hW.data(), 4 * 256L * in0 * nz, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), 4 * 256 * nz, hipMemcpyHostToDevice));
  WsFwd0P p; memset(&p, 0, sizeof(p));
  p.X = dX; p.x_s1 = (long)M * xp; p.x_pitch = xp; p.in0 = in0; p.W = dW; p.w_s1 = 256L * in0; p.bias = db; p.b_s1 = 256;
  p.Y = dY; p.y_s1 = (long)M * 256; p.y_pitch = 256; p.mb = dmb; p.mb_s1 = (long)M * 8; p.mb_g = 8; p.M = M; p.nz1 = nz;
  if (!ws_fwd0_supported(p, 256)) { printf("not supported\n"); return 1; }
  CK(launch_ws_fwd0(p, nz, 0)); CK(hipDeviceSynchronize());
  if (check) {
    std::vector<float> Y((long)M * 256 * nz); std::vector<unsigned> mb((long)M * 8 * nz);
    CK(hipMemcpy(Y.data(), dY, 4L * M * 256 * nz, hipMemcpyDeviceToHost)); CK(hipMemcpy(mb.data(), dmb, 4L * M * 8 * nz, hipMemcpyDeviceToHost));
    double e = 0, sc = 0; long bad = 0;
    for (int z = 0; z < nz; ++z) for (int m = 0; m < M; ++m) for (int n = 0; n < 256; ++n) {
      double a = hb[z * 256 + n];
      for (int k = 0; k < in0; ++k) a += (double)hX[((long)z * M + m) * xp + k] * hW[((long)z * 256 + n) * in0 + k];
      const double y = a > 0 ? a : 0; const float got = Y[((long)z * M + m) * 256 + n];
      e = std::max(e, std::fabs(got - y)); sc = std::max(sc, y);
      if ((((mb[((long)z * M + m) * 8 + (n >> 5)] >> (n & 31)) & 1u) != 0) != (got > 0.f)) ++bad;
    }
    printf("M=%d nz=%d  max|dY|=%.3e (scale %.3f)  mask bits inconsistent: %ld\n", M, nz, e, sc, bad);
  }
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) CK(launch_ws_fwd0(p, nz, 0));
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < 20; ++i) CK(launch_ws_fwd0(p, nz, 0));
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double us = ms / 20 * 1e3;
  printf("ws_fwd0 M=%d nz=%d: %.1f us  %.0f GB/s written\n", M, nz, us, 4.0 * M * 256 * nz / us * 1e-3);
  return 0;
}
