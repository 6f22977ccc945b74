// Prints the lane/register -> (row, col) map of v_mfma_f64_16x16x4_f64 on the device it runs on,
// and a crude issue-rate estimate.  Diagnostic only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe(double* out, int kq) {
  int l = threadIdx.x;
  int i = l & 15, k = l >> 4;  // A[i][k], B[k][j=l&15]
  double a = 1.0 + i + 16.0 * k;
  double b = (k == kq) ? (double)((l & 15) + 1) * 1000.0 : 0.0;
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
__global__ void rate(double* out, int iters) {
  int l = threadIdx.x;
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
  for (int it = 0; it < iters; ++it) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + l] = c0[0] + c1[1] + c2[2] + c3[3];
}
__global__ void rate_dep(double* out, int iters) {
  int l = threadIdx.x;
  d4 c0 = {0, 0, 0, 0};
  double a = 1.0 + l * 1e-3, b = 1.0 - l * 1e-3;
  for (int it = 0; it < iters; ++it) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
  }
  out[blockIdx.x * blockDim.x + l] = c0[0];
}
__global__ void rate_fma(double* out, int iters) {
  int l = threadIdx.x;
  double x0 = l, x1 = l + 1, x2 = l + 2, x3 = l + 3, x4 = 1, x5 = 2, x6 = 3, x7 = 4;
  double a = 1.0000001, b = 1e-9;
  for (int it = 0; it < iters; ++it) {
    x0 = fma(x0, a, b); x1 = fma(x1, a, b); x2 = fma(x2, a, b); x3 = fma(x3, a, b);
    x4 = fma(x4, a, b); x5 = fma(x5, a, b); x6 = fma(x6, a, b); x7 = fma(x7, a, b);
  }
  out[blockIdx.x * blockDim.x + l] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
int main() {
  double* d;
  hipMalloc(&d, 1 << 24);
  std::vector<double> h(256);
  for (int kq = 0; kq < 4; ++kq) {
    probe<<<1, 64>>>(d, kq);
    hipMemcpy(h.data(), d, 256 * 8, hipMemcpyDeviceToHost);
    int bad_guide = 0;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 4; ++r) {
        double v = h[l * 4 + r] / 1000.0;  // (1 + i + 16 kq) * (j + 1)
        int j = l & 15;
        double ai = v / (j + 1) - 1 - 16 * kq;
        int i_guide = (l >> 4) + 4 * r;
        if ((int)(ai + 0.5) != i_guide) bad_guide++;
        if (kq == 0 && (l == 0 || l == 17 || l == 35 || l == 63)) printf("lane %2d reg %d -> row %g col %d\n", l, r, ai, j);
      }
    printf("kq=%d: entries not matching row=(lane>>4)+4*reg, col=lane&15 : %d\n", kq, bad_guide);
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, blocks = 256 * 4;  // 1 wave per block -> ~1 wave per SIMD over the chip
  for (int rep = 0; rep < 2; ++rep) {
    float ms;
    hipEventRecord(e0); rate<<<blocks, 64>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * iters * 4 * 2048;
    printf("mfma_f64 4 indep acc, 1 wave/SIMD : %.3f ms  %.2f TFLOP/s\n", ms, flops / ms * 1e-9);
    hipEventRecord(e0); rate_dep<<<blocks, 64>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("mfma_f64 dependent chain, 1 wave/SIMD: %.3f ms  %.2f TFLOP/s\n", ms, flops / ms * 1e-9);
    hipEventRecord(e0); rate<<<blocks * 2, 64>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("mfma_f64 4 indep acc, 2 waves/SIMD: %.3f ms  %.2f TFLOP/s\n", ms, 2 * flops / ms * 1e-9);
    hipEventRecord(e0); rate_fma<<<blocks * 4, 64>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("v_fma_f64 8 indep, 4 waves/SIMD   : %.3f ms  %.2f TFLOP/s\n", ms, (double)blocks * 4 * iters * 8 * 128 / ms * 1e-9);
  }
  return 0;
}
