// Stage timings of k_aem_inverse (debug tool, not part of the library): the kernel is instantiated with parts switched
// off (wrong results) to see where a launch spends its time.
// Build on the GPU box: hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -w -Itinyda_amd/csrc -Iinclude -o /tmp/aip tools/aem_inverse_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#include "tinyda_amd.h"
#include "tda_kernels_ml.h"
using namespace tda;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int SKIP, int NWV = 8>
static float time_variant(const AemInvArgs& a, size_t lds, int reps) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k_aem_inverse<SKIP, NWV>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k_aem_inverse<SKIP, NWV>), dim3((unsigned)a.N), dim3(64 * NWV), lds, 0, a);
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_aem_inverse<SKIP, NWV>), dim3((unsigned)a.N), dim3(64 * NWV), lds, 0, a);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / reps;
}

int main(int argc, char** argv) {
  const int m = argc > 1 ? atoi(argv[1]) : 128;
  const int64_t N = argc > 2 ? atoll(argv[2]) : 4096;
  const int MP = m <= 64 ? 64 : 128, nb = (m + 15) / 16;
  std::mt19937_64 g(5);
  std::normal_distribution<double> nd;
  std::vector<double> cov((size_t)MP * MP, 0.0), sig((size_t)MP * MP, 0.0), x(m);
  for (int i = 0; i < m; ++i) cov[(size_t)i * MP + i] = 0.01;
  for (int s = 0; s < 20; ++s) {  // rank-20 bias covariance
    for (auto& v : x) v = 0.05 * nd(g);
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < m; ++j) sig[(size_t)i * MP + j] += x[i] * x[j] / 20;
  }
  double *dcov, *dsig, *dP;
  CK(hipMalloc(&dcov, cov.size() * 8));
  CK(hipMalloc(&dsig, (size_t)N * MP * MP * 8));
  CK(hipMalloc(&dP, (size_t)N * MP * MP * 8));
  CK(hipMemcpy(dcov, cov.data(), cov.size() * 8, hipMemcpyHostToDevice));
  for (int64_t c = 0; c < N; ++c) CK(hipMemcpy(dsig + (size_t)c * MP * MP, sig.data(), sig.size() * 8, hipMemcpyHostToDevice));
  AemInvArgs a{};
  a.N = N; a.m = m; a.MP = MP; a.nb = nb; a.nsum = 1; a.cov = dcov; a.sig[0] = dsig; a.P = dP;
  const size_t lds = (size_t)(nb * (nb + 1) / 2) * AEM_BS * sizeof(double);
  printf("m=%d N=%lld nb=%d lds=%zu B\n", m, (long long)N, nb, lds);
  printf("full, 4 waves per chain      %9.1f us\n", time_variant<0, 4>(a, lds, 10));
  printf("full                         %9.1f us\n", time_variant<0>(a, lds, 10));
  printf("staging only                 %9.1f us\n", time_variant<2>(a, lds, 10));
  printf("no diagonal blocks           %9.1f us\n", time_variant<1>(a, lds, 10));
  printf("no P = W^T W                 %9.1f us\n", time_variant<4>(a, lds, 10));
  printf("P not stored                 %9.1f us\n", time_variant<16>(a, lds, 10));
  printf("no tri-inverse, no P         %9.1f us\n", time_variant<12>(a, lds, 10));
  printf("no diag, no tri-inv, no P    %9.1f us\n", time_variant<13>(a, lds, 10));
  // check of the full variant against a host inverse of chain 0
  hipLaunchKernelGGL((k_aem_inverse<0, 8>), dim3((unsigned)N), dim3(512), lds, 0, a);
  std::vector<double> P((size_t)MP * MP);
  CK(hipMemcpy(P.data(), dP, P.size() * 8, hipMemcpyDeviceToHost));
  double err = 0;
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < m; ++j) {
      double s = 0;
      for (int k = 0; k < m; ++k) s += (cov[(size_t)i * MP + k] + sig[(size_t)i * MP + k]) * P[(size_t)k * MP + j];
      err = fmax(err, fabs(s - (i == j ? 1.0 : 0.0)));
    }
  printf("max |M P - I| = %.2e\n", err);
  return 0;
}
