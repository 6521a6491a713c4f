// gemm_lab.hip — standalone timing lab for the GEMM kernels (compiles in seconds; not part of the product).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DORL_LAB_FAKE_SPLIT] tools/gemm_lab.hip -o /tmp/gemm_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../offlinerl-kit_amd/csrc/gemm.h"
using namespace orl;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <class F> float time_it(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) f();
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps * 1e3f;
}

int main(int argc, char** argv) {
  const int M = 7936, N = 256, K = 256, nz = argc > 1 ? atoi(argv[1]) : 32;
  const long nA = (long)M * K, nB = (long)N * K, nC = (long)M * N;
  float *dA, *dB, *dC, *dH, *dv0, *dv1;
  CK(hipMalloc(&dA, 4 * nA * nz)); CK(hipMalloc(&dB, 4 * nB * nz)); CK(hipMalloc(&dC, 4 * nC * nz * 2)); CK(hipMalloc(&dH, 4 * nC * nz));
  CK(hipMalloc(&dv0, 4 * (M + N + K) * nz)); CK(hipMalloc(&dv1, 4 * (M + N + K) * nz));
  std::vector<float> h(nC * nz);
  unsigned s = 1;
  auto fill = [&](float* d, long n) { for (long i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) / 8388608.0f) - 1.0f; } CK(hipMemcpy(d, h.data(), 4 * n, hipMemcpyHostToDevice)); };
  fill(dA, nA * nz); fill(dB, nB * nz); fill(dH, nC * nz); fill(dv0, (long)(M + N + K) * nz); fill(dv1, (long)(M + N + K) * nz);
  GemmP p; memset(&p, 0, sizeof(p));
  p.nz1 = nz; p.ksplit = 1; p.C = dC; p.c_sn = 1; p.c_s1 = nC; p.A = {dA, 0, nA}; p.B = {dB, 0, nB};
  p.rowv = {dv0, 0, (long)(M + N + K)}; p.colv = {dv1, 0, (long)(M + N + K)}; p.bias = {dv0, 0, (long)(M + N + K)};
  p.aux = {dH, 0, nC}; p.aux_sr = N; p.M = M; p.N = N; p.K = K; p.c_sr = N;
  const double gf = 2.0 * M * N * K * nz / 1e9;
  auto rep = [&](const char* name, float us) { printf("%-44s %8.1f us  %7.1f TF(alg)\n", name, us, gf / us * 1e-3); };
  auto rep2 = [&](const char* name, float us) { printf("%-52s %8.1f us  %7.1f TF(alg)\n", name, us, gf / us * 1e-3); fflush(stdout); };
#define FWD(CFG, PREC) { p.a_sr = K; p.a_sk = 1; p.b_sr = K; p.b_sk = 1; \
    rep2("fwd   " #CFG " " #PREC, time_it([&] { (void)launch_inst<CFG, L_VECK, L_VECK, PA_PLAIN, PB_PLAIN, E_BIAS_RELU, PREC>(p, nz, 0); }, 20)); }
#define DGR(CFG, PREC) { p.a_sr = K; p.a_sk = 1; p.b_sr = 1; p.b_sk = N; p.b_rlim = N; p.a_trans = 0; \
    rep2("dgrad " #CFG " " #PREC, time_it([&] { (void)launch_inst<CFG, L_VECK, L_BLK4, PA_RANK1, PB_PLAIN, E_MASK, PREC>(p, nz, 0); }, 20)); }
  typedef GemmCfg<2, 2, 4, 4, 32> C_2244;
  typedef GemmCfg<2, 4, 4, 2, 32> C_2442;
  typedef GemmCfg<2, 2, 2, 2, 32> C_2222;
  typedef GemmCfg<2, 4, 8, 4, 32> C_2484;
  typedef GemmCfg<4, 2, 4, 8, 32> C_4248;
  typedef GemmCfg<4, 2, 4, 4, 32> C_4244;
  typedef GemmCfg<2, 4, 4, 4, 32> C_2444;
  typedef GemmCfg<4, 4, 2, 4, 32> C_4424;
  // wgrad-like: dW[out x in] = dz^T h over `rows` rows, rank-1 virtual dz; both operands row-contiguous
  const int rows = M;
  float* dG; CK(hipMalloc(&dG, 4L * 32 * (256 * 256 + 1024) * nz)); 
#define WGR(CFG, PREC, KS, TAILG) { GemmP w = p; w.M = 256; w.N = 256; w.K = rows; w.a_sr = 1; w.a_sk = 256; w.b_sr = 1; w.b_sk = 256; \
    w.a_rlim = 256; w.b_rlim = 256; w.a_trans = 1; w.ksplit = KS; w.ones_row = 1 << 30; w.A = {dA, 0, nA}; w.B = {dH, 0, nC}; \
    w.C = dG; w.c_sr = 256; w.c_sn = 1; w.c_s1 = 32L * (65536 + 1024); w.c_ks = 65536 + 1024; w.bias_out = dG + 65536; w.bo_s1 = w.c_s1; w.bo_ks = w.c_ks; \
    if (TAILG) { w.tail_w_out = dG + 65536 + 256; w.tail_b_out = dG + 65536 + 512; w.tw_s1 = w.c_s1; w.tb_s1 = w.c_s1; } \
    rep2("wgrad " #CFG " " #PREC " ks=" #KS " tail=" #TAILG, time_it([&] { (void)launch_inst<CFG, L_BLK4, L_BLK4, PA_RANK1, PB_PLAIN, E_WGRAD, PREC>(w, nz, 0); }, 20)); \
    if (!TAILG) rep2("wgrad(plain A) " #CFG " " #PREC " ks=" #KS, time_it([&] { (void)launch_inst<CFG, L_BLK4, L_BLK4, PA_PLAIN, PB_PLAIN, E_WGRAD, PREC>(w, nz, 0); }, 20)); }
#ifdef ORL_LAB_STAMPS
  unsigned long long* dst; CK(hipMalloc(&dst, 8 * 16 * 128)); CK(hipMemset(dst, 0, 8 * 16 * 128));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_lab_stamps), &dst, sizeof(dst)));
  auto dump = [&](const char* name) {
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> st(16 * 128); CK(hipMemcpy(st.data(), dst, 8 * 16 * 128, hipMemcpyDeviceToHost));
    printf("%s stamps (cycles after the kernel-entry stamp; 1 init done, 2 first chunk staged, 12 loop done, 13 epilogue done)\n", name);
    for (int b : {0, 1, 8, 60, 100}) { printf("  blk %3d:", b); for (int i : {1, 2, 12, 13}) printf(" %6lld", st[b * 16 + i] ? (long long)(st[b * 16 + i] - st[b * 16]) : -1LL); printf("\n"); }
  };
  WGR(C_2244, P_BF16X3, 4, 1) dump("wgrad");
#else
  FWD(C_2442, P_BF16X3) FWD(C_2444, P_BF16X3) FWD(C_4244, P_BF16X3)
  DGR(C_2244, P_BF16X3) DGR(C_2444, P_BF16X3) DGR(C_4244, P_BF16X3)
#endif
  CK(hipDeviceSynchronize());
  printf("done\n");
  return 0;
}
