// Micro-probe (run on the GPU box): is an fp16 hi + lo operand split usable on gfx950's f16 MFMA?
//   hipcc -O3 --offload-arch=gfx950 tools/fp16_split_probe.hip -o gpurun_out/fp16_split_probe && gpurun_out/fp16_split_probe
// (1) fp16 SUBNORMAL operands: does v_mfma_f32_16x16x32_f16 / 32x32x16_f16 multiply them exactly or flush them to zero?
//     lo = x - fp16(x) is ~2^-12 |x| and falls below fp16's smallest normal (6.1e-5) for |x| < 0.25.
// (2) does v_cvt_pk_f16_f32 round subnormal results (round-to-nearest-even onto the 2^-24 grid) or flush them?
// (3) a 256-deep dot product through the three products hi*hi + hi*lo + lo*hi, fp32 accumulation in the MFMA, against float64:
//     bf16 split vs fp16 split (operands O(1) unscaled, and weight-sized operands with and without a power-of-two scale).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void k_subnormal(float* out) {
  // every A element = 2^-20 (fp16 subnormal: 16 * 2^-24), every B element = 1: D = 32 * 2^-20 = 2^-15 if honoured, 0 if flushed
  h8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)9.5367431640625e-07f; b[i] = (_Float16)1.0f; }
  f4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  f16v d; for (int i = 0; i < 16; ++i) d[i] = 0.f;
  d = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, d, 0, 0, 0);
  // subnormal x subnormal: 2^-20 * 2^-20 * 32 = 2^-35 (an fp32 normal)
  f4 e = {0, 0, 0, 0};
  e = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, a, e, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = c[0]; out[1] = d[0]; out[2] = e[0]; }
  // conversion: 3 * 2^-25 is halfway between 2^-24 and 2^-23 on the subnormal grid -> RNE gives 2^-23; 2^-26 -> 0; 1e-6 -> 17 * 2^-24
  if (threadIdx.x == 0) {
    volatile float x0 = 8.940696716308594e-08f, x1 = 1.4901161193847656e-08f, x2 = 1.0e-6f;
    out[3] = (float)(_Float16)x0; out[4] = (float)(_Float16)x1; out[5] = (float)(_Float16)x2;
  }
}

// one wave: D[16][16] = A[16][32*KT] B^T, three split products; MODE 0 bf16, 1 fp16.  sa / sb: power-of-two operand scales (folded back).
template <int MODE>
__global__ void k_dot(const float* A, const float* B, float* D, int K, float sa, float sb) {
  const int lane = threadIdx.x, row = lane & 15, kq = lane >> 4;
  f4 acc = {0, 0, 0, 0};
  for (int k0 = 0; k0 < K; k0 += 32) {
    float av[8], bv[8];
    for (int i = 0; i < 8; ++i) { av[i] = A[row * K + k0 + kq * 8 + i] * sa; bv[i] = B[row * K + k0 + kq * 8 + i] * sb; }
    if (MODE == 0) {
      b8 ah, al, bh, bl;
      for (int i = 0; i < 8; ++i) {
        ah[i] = (__bf16)av[i]; al[i] = (__bf16)(av[i] - (float)ah[i]);
        bh[i] = (__bf16)bv[i]; bl[i] = (__bf16)(bv[i] - (float)bh[i]);
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
    } else {
      h8 ah, al, bh, bl;
      for (int i = 0; i < 8; ++i) {
        ah[i] = (_Float16)av[i]; al[i] = (_Float16)(av[i] - (float)ah[i]);
        bh[i] = (_Float16)bv[i]; bl[i] = (_Float16)(bv[i] - (float)bh[i]);
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc, 0, 0, 0);
    }
  }
  const float inv = 1.0f / (sa * sb);
  // D layout of 16x16: lane holds column (lane & 15) of B-rows, rows 4 * (lane >> 4) + j of A
  for (int j = 0; j < 4; ++j) D[(4 * kq + j) * 16 + row] = acc[j] * inv;
}

static double frand() { return rand() / (double)RAND_MAX; }
static double nrand() { return sqrt(-2.0 * log(frand() + 1e-12)) * cos(6.283185307179586 * frand()); }

static void dot_case(const char* what, double amp_a, bool relu_a, double amp_b, float sa, float sb) {
  const int K = 256;
  float hA[16 * K], hB[16 * K], hD[256];
  for (int i = 0; i < 16 * K; ++i) {
    double a = nrand() * amp_a; if (relu_a && a < 0) a = 0;
    hA[i] = (float)a; hB[i] = (float)((2 * frand() - 1) * amp_b);
  }
  double ex[256], scale = 0;
  for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
    double s = 0; for (int k = 0; k < K; ++k) s += (double)hA[m * K + k] * hB[n * K + k];
    ex[m * 16 + n] = s; if (fabs(s) > scale) scale = fabs(s);
  }
  float *dA, *dB, *dD;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  for (int mode = 0; mode < 2; ++mode) {
    if (mode == 0) hipLaunchKernelGGL(k_dot<0>, dim3(1), dim3(64), 0, 0, dA, dB, dD, K, 1.0f, 1.0f);
    else hipLaunchKernelGGL(k_dot<1>, dim3(1), dim3(64), 0, 0, dA, dB, dD, K, sa, sb);
    hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    double mx = 0, ss = 0;
    for (int i = 0; i < 256; ++i) { double e = fabs(hD[i] - ex[i]); if (e > mx) mx = e; ss += e * e; }
    printf("%-44s %-22s max err / scale %.2e  rms %.2e\n", what, mode == 0 ? "bf16 hi+lo" : "fp16 hi+lo", mx / scale, sqrt(ss / 256) / scale);
  }
  // plain fp32 reference on the host (sequential float accumulation)
  double mx = 0;
  for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) {
    float s = 0; for (int k = 0; k < K; ++k) s = fmaf(hA[m * K + k], hB[n * K + k], s);
    double e = fabs(s - ex[m * 16 + n]); if (e > mx) mx = e;
  }
  printf("%-44s %-22s max err / scale %.2e\n", what, "fp32 fmaf chain (host)", mx / scale);
  hipFree(dA); hipFree(dB); hipFree(dD);
}

int main() {
  float* out; hipMalloc(&out, 64);
  hipLaunchKernelGGL(k_subnormal, dim3(1), dim3(64), 0, 0, out);
  float h[6]; hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  printf("subnormal fp16 A (2^-20) x 1.0, K = 32 : 16x16x32_f16 -> %.6e, 32x32x16_f16 (K = 16) -> %.6e   (honoured: 3.051758e-05 / 1.525879e-05, flushed: 0)\n", h[0], h[1]);
  printf("subnormal x subnormal (2^-40 * 32)      : %.6e   (honoured: 2.910383e-11)\n", h[2]);
  printf("v_cvt f32 -> f16 of 3*2^-25, 2^-26, 1e-6 : %.6e %.6e %.6e   (RNE on the subnormal grid: 1.192093e-07 0 1.013279e-06)\n", h[3], h[4], h[5]);
  srand(1);
  dot_case("relu(N(0,0.5)) x U(-1/16,1/16), fp16 unscaled", 0.5, true, 1.0 / 16, 1.0f, 1.0f);
  dot_case("relu(N(0,0.5)) x U(-1/16,1/16), fp16 B*2^6", 0.5, true, 1.0 / 16, 1.0f, 64.0f);
  dot_case("N(0,1e-3) x relu-less N(0,0.5), fp16 unscaled", 1e-3, false, 0.5, 1.0f, 1.0f);
  dot_case("N(0,1e-3) x N(0,0.5), fp16 A*2^10", 1e-3, false, 0.5, 1024.0f, 1.0f);
  return 0;
}
