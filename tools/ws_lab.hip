// ws_lab.hip — correctness + timing lab for csrc/ws_gemm.h (not part of the product).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/ws_lab.hip -o tools/ws_lab_a
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../offlinerl-kit_amd/csrc/ws_gemm.h"
using namespace orl;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static unsigned rs = 1;
static float rnd() { rs = rs * 1664525u + 1013904223u; return ((rs >> 8) / 8388608.0f) - 1.0f; }

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 1088, nz = argc > 2 ? atoi(argv[2]) : 2;
  const bool check = M * (long)nz <= 40000;
  const long nX = (long)M * 256, nW = 65536;
  std::vector<float> hX(nX * nz), hW(nW * nz), hb(256 * nz), htw(256 * nz), htb(nz);
  for (auto& v : hX) v = rnd() > 0.f ? rnd() : 0.f;     // post-ReLU-like input
  for (auto& v : hW) v = rnd() * 0.0625f;
  for (auto& v : hb) v = rnd() * 0.1f;
  for (auto& v : htw) v = rnd() * 0.0625f;
  for (auto& v : htb) v = rnd();
  float *dX, *dW, *db, *dY, *dtw, *dtb, *dq; unsigned* dmb;
  CK(hipMalloc(&dX, 4 * nX * nz)); CK(hipMalloc(&dW, 4 * nW * nz)); CK(hipMalloc(&db, 4 * 256 * nz)); CK(hipMalloc(&dY, 4 * nX * nz));
  CK(hipMalloc(&dtw, 4 * 256 * nz)); CK(hipMalloc(&dtb, 4 * nz)); CK(hipMalloc(&dq, 4L * M * nz)); CK(hipMalloc(&dmb, 4L * M * 8 * nz));
  CK(hipMemcpy(dX, hX.data(), 4 * nX * nz, hipMemcpyHostToDevice)); CK(hipMemcpy(dW, hW.data(), 4 * nW * nz, hipMemcpyHostToDevice));
  CK(hipMemcpy(db, hb.data(), 4 * 256 * nz, hipMemcpyHostToDevice)); CK(hipMemcpy(dtw, htw.data(), 4 * 256 * nz, hipMemcpyHostToDevice));
  CK(hipMemcpy(dtb, htb.data(), 4 * nz, hipMemcpyHostToDevice)); CK(hipMemset(dmb, 0, 4L * M * 8 * nz));
  WsFwdP p; memset(&p, 0, sizeof(p));
  p.X = dX; p.x_s1 = nX; p.x_pitch = 256; p.W = dW; p.w_s1 = nW; p.bias = db; p.b_s1 = 256; p.Y = dY; p.y_s1 = nX; p.y_pitch = 256;
  p.mb = dmb; p.mb_s1 = (long)M * 8; p.mb_g = 8; p.tw = dtw; p.tw_s1 = 256; p.tb = dtb; p.tb_s1 = 1; p.tq = dq; p.tq_s1 = M; p.tq_sm = 1;
  p.M = M; p.nz1 = nz;
  if (!ws_fwd_supported(p, 256, 256)) { printf("not supported\n"); return 1; }
  CK(launch_ws_fwd(p, nz, 0)); CK(hipDeviceSynchronize());
  if (check) {
    std::vector<float> Y(nX * nz), q((long)M * nz); std::vector<unsigned> mb((long)M * 8 * nz);
    CK(hipMemcpy(Y.data(), dY, 4 * nX * nz, hipMemcpyDeviceToHost)); CK(hipMemcpy(q.data(), dq, 4L * M * nz, hipMemcpyDeviceToHost));
    CK(hipMemcpy(mb.data(), dmb, 4L * M * 8 * nz, hipMemcpyDeviceToHost));
    double eY = 0, eq = 0, sY = 0, sq = 0; long badbits = 0, nearzero = 0;
    for (int z = 0; z < nz; ++z)
      for (int m = 0; m < M; ++m) {
        double qa = htb[z];
        for (int n = 0; n < 256; ++n) {
          double a = hb[z * 256 + n];
          for (int k = 0; k < 256; ++k) a += (double)hX[z * nX + (long)m * 256 + k] * hW[z * nW + n * 256 + k];
          const double y = a > 0 ? a : 0;
          const float got = Y[z * nX + (long)m * 256 + n];
          eY = std::max(eY, std::fabs(got - y)); sY = std::max(sY, std::fabs(y));
          qa += y * htw[z * 256 + n];
          const unsigned bit = (mb[z * (long)M * 8 + (long)m * 8 + (n >> 5)] >> (n & 31)) & 1u;
          if (bit != (got > 0.f ? 1u : 0u)) ++badbits;
          if (std::fabs(a) < 1e-5) ++nearzero;
        }
        eq = std::max(eq, std::fabs(q[z * (long)M + m] - qa)); sq = std::max(sq, std::fabs(qa));
      }
    printf("M=%d nz=%d  max|dY|=%.3e (scale %.3f)  max|dq|=%.3e (scale %.3f)  mask bits inconsistent with stored Y: %ld\n", M, nz, eY, sY, eq, sq, badbits);
  }
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) CK(launch_ws_fwd(p, nz, 0));
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < 20; ++i) CK(launch_ws_fwd(p, nz, 0));
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  const double us = ms / 20 * 1e3;
  printf("ws_fwd M=%d nz=%d: %.1f us  %.1f TF(alg)  %.0f GB/s (X read + Y written)\n", M, nz, us, 2.0 * M * 65536 * nz / us * 1e-6, 8.0 * nX * nz / us * 1e-3);
  return 0;
}
