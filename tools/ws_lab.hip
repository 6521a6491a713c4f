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
  const bool l0 = argc > 3 && atoi(argv[3]) == 1;
  const int xp = 24, in0 = 23;
  const long nX = (long)M * 256, nW = 65536;
  std::vector<float> hX(nX * nz), hW(nW * nz), hb(256 * nz), htw(256 * nz), htb(nz);
  for (auto& v : hX) v = rnd() > 0.f ? rnd() : 0.f;     // post-ReLU-like input
  std::vector<float> hX0((long)M * xp * nz), hW0(256L * in0 * nz), hb0(256 * nz);
  for (long i = 0; i < (long)M * nz; ++i) for (int c = 0; c < xp; ++c) hX0[i * xp + c] = c < in0 ? rnd() : 0.f;
  for (auto& v : hW0) v = rnd() * 0.2f; for (auto& v : hb0) v = rnd() * 0.2f;
  if (l0) for (int z = 0; z < nz; ++z) for (int m = 0; m < M; ++m) for (int n = 0; n < 256; ++n) {   // reference h0 (fp64 -> fp32)
    double a = hb0[z * 256 + n]; for (int k = 0; k < in0; ++k) a += (double)hX0[((long)z * M + m) * xp + k] * hW0[((long)z * 256 + n) * in0 + k];
    hX[z * nX + (long)m * 256 + n] = a > 0 ? (float)a : 0.f; }
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
  p.X = dX; p.x_s1 = nX; p.x_pitch = 256; p.W = dW; p.w_s1 = nW; p.w_sn = 256; p.w_sk = 1; p.bias = db; p.b_s1 = 256; p.Y = dY; p.y_s1 = nX; p.y_pitch = 256;
  p.mb = dmb; p.mb_s1 = (long)M * 8; p.mb_g = 8; p.tw = dtw; p.tw_s1 = 256; p.tb = dtb; p.tb_s1 = 1; p.tq = dq; p.tq_s1 = M; p.tq_sm = 1;
  p.M = M; p.nz1 = nz;
  float *dX0, *dW0, *db0; unsigned* dmb0;
  CK(hipMalloc(&dX0, 4L * M * xp * nz)); CK(hipMalloc(&dW0, 4 * 256L * in0 * nz)); CK(hipMalloc(&db0, 4 * 256 * nz)); CK(hipMalloc(&dmb0, 4L * M * 8 * nz));
  CK(hipMemcpy(dX0, hX0.data(), 4L * M * xp * nz, hipMemcpyHostToDevice)); CK(hipMemcpy(dW0, hW0.data(), 4 * 256L * in0 * nz, hipMemcpyHostToDevice));
  CK(hipMemcpy(db0, hb0.data(), 4 * 256 * nz, hipMemcpyHostToDevice));
  if (l0) { CK(hipMemset(dX, 0, 4 * nX * nz)); p.X0 = dX0; p.x0_s1 = (long)M * xp; p.x0_pitch = xp; p.in0 = in0; p.W0 = dW0; p.w0_s1 = 256L * in0; p.w0_sn = in0; p.w0_sk = 1; p.b0 = db0; p.b0_s1 = 256;
            p.mb0 = dmb0; p.mb0_s1 = (long)M * 8; p.mb0_g = 8; if (!ws_fwd01_supported(p)) { printf("l0 not supported\n"); return 1; } }
  if (!ws_fwd_supported(p, 256, 256)) { printf("not supported\n"); return 1; }
  CK(launch_ws_fwd(p, nz, 0)); CK(hipDeviceSynchronize());
  const bool noY = argc > 4 && atoi(argv[4]) == 1;      // time (and re-check q / mask bits of) the variant that does not store Y
  std::vector<float> q; std::vector<unsigned> mb;
  if (check) {
    std::vector<float> Y(nX * nz); q.resize((long)M * nz); mb.resize((long)M * 8 * nz);
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
    if (l0) { std::vector<float> H0(nX * nz); std::vector<unsigned> m0((long)M * 8 * nz); CK(hipMemcpy(H0.data(), dX, 4 * nX * nz, hipMemcpyDeviceToHost));
      CK(hipMemcpy(m0.data(), dmb0, 4L * M * 8 * nz, hipMemcpyDeviceToHost)); double e0 = 0; long bad0 = 0;
      for (long i = 0; i < nX * nz; ++i) { e0 = std::max(e0, (double)std::fabs(H0[i] - hX[i])); const long row = i / 256; const int n = i % 256;
        if ((((m0[row * 8 + (n >> 5)] >> (n & 31)) & 1u) != 0) != (H0[i] > 0.f)) ++bad0; }
      printf("fused layer 0: max|dh0|=%.3e  h0 mask bits inconsistent: %ld\n", e0, bad0); }
    printf("M=%d nz=%d  max|dY|=%.3e (scale %.3f)  max|dq|=%.3e (scale %.3f)  mask bits inconsistent with stored Y: %ld\n", M, nz, eY, sY, eq, sq, badbits);
  }
  if (noY) {
    p.Y = nullptr;
    if (check) {
      CK(hipMemset(dq, 0, 4L * M * nz)); CK(hipMemset(dmb, 0, 4L * M * 8 * nz));
      CK(launch_ws_fwd(p, nz, 0)); CK(hipDeviceSynchronize());
      std::vector<float> q2((long)M * nz); std::vector<unsigned> mb2((long)M * 8 * nz);
      CK(hipMemcpy(q2.data(), dq, 4L * M * nz, hipMemcpyDeviceToHost)); CK(hipMemcpy(mb2.data(), dmb, 4L * M * 8 * nz, hipMemcpyDeviceToHost));
      printf("no-Y variant: q identical %d, mask bits identical %d\n", (int)(memcmp(q.data(), q2.data(), 4L * M * nz) == 0), (int)(memcmp(mb.data(), mb2.data(), 4L * M * 8 * nz) == 0));
    }
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
