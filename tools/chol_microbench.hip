// Stand-alone timing / equality harness for the per-chain Cholesky kernel (debug tool, not part of the library).
// Measured on MI355X (4096 chains, d = 64): library k_chol 105 us; LDS-broadcast variant 114 us (bitwise equal); forcing 3 waves/SIMD
// 157 us (spills); reciprocal-sqrt pivots 126-135 us; two lanes per row 1.2 ms; round 2: k_chol_loop (a real column loop, shifting
// registers) 134 us, 115 without any HBM access.  None replaced the library kernel (tools/experimental/README.md has the table).
// Usage: /tmp/cmb [d [chains]]
// Build on the GPU box: hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -w -Itinyda_amd/csrc -Iinclude -Itools -o /tmp/cmb tools/chol_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#include <random>
#include "tinyda_amd.h"
#include "tda_kernels_mh.h"
#include "experimental/tda_kernels_chol_x.h"
using namespace tda;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)


int main(int argc, char** argv) {
  constexpr int D = 64;
  const int d = argc > 1 ? atoi(argv[1]) : 64;
  const int64_t N = argc > 2 ? atoll(argv[2]) : 4096;
  constexpr int NTL = am_tiles<D>();
  std::mt19937_64 g(3);
  std::normal_distribution<double> nd;
  std::vector<double> sig((size_t)N * NTL * 256, 0.0);
  std::vector<double> B((size_t)d * d), S((size_t)d * d);
  for (int64_t c = 0; c < N; ++c) {
    for (auto& v : B) v = nd(g);
    for (int i = 0; i < d; ++i)
      for (int j = 0; j <= i; ++j) {
        double s = (i == j) ? 1e-3 : 0.0;
        for (int k = 0; k < d; ++k) s += B[i * d + k] * B[j * d + k] / d;
        sig[(size_t)c * NTL * 256 + am_sigma_offset(i, j)] = s;
      }
  }
  double *dsig, *L1, *L2;
  int32_t* flags;
  CK(hipMalloc(&dsig, sig.size() * 8));
  CK(hipMemcpy(dsig, sig.data(), sig.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&L1, (size_t)N * D * D * 8));
  CK(hipMalloc(&L2, (size_t)N * D * D * 8));
  CK(hipMalloc(&flags, N * 4));
  CK(hipMemset(flags, 0, N * 4));
  CK(hipMemset(L1, 0xff, (size_t)N * D * D * 8));
  CK(hipMemset(L2, 0xff, (size_t)N * D * D * 8));
  CholArgs a{};
  a.N = N;
  a.d = d;
  a.am_sigma = dsig;
  a.flags = flags;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const char* names[] = {"k_chol (library: readlane)", "k_chol_x<rs=0,wpe=2> (LDS broadcast)", "k_chol_x<rs=0,wpe=3>", "k_chol_x<rs=1,wpe=2>", "k_chol_x<rs=1,wpe=3>",
                         "k_chol_2t<rs=0>", "k_chol_2t<rs=1>", "k_chol_loop", "k_chol_loop, no loads", "k_chol_loop, no stores", "k_chol_loop, neither"};
  std::vector<double> h1((size_t)N * D * D), h2((size_t)N * D * D);
  for (int which = 0; which < 11; ++which) {
    a.Lk = which ? L2 : L1;
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipEventRecord(e0));
      const dim3 g4((unsigned)((N + 3) / 4)), g2((unsigned)((N + 1) / 2));
      switch (which) {
        case 0: hipLaunchKernelGGL(k_chol<D>, dim3((unsigned)N), dim3(64), 0, 0, a); break;
        case 1: hipLaunchKernelGGL((k_chol_x<D, 0, 2>), g4, dim3(256), 0, 0, a); break;
        case 2: hipLaunchKernelGGL((k_chol_x<D, 0, 3>), g4, dim3(256), 0, 0, a); break;
        case 3: hipLaunchKernelGGL((k_chol_x<D, 1, 2>), g4, dim3(256), 0, 0, a); break;
        case 4: hipLaunchKernelGGL((k_chol_x<D, 1, 3>), g4, dim3(256), 0, 0, a); break;
        case 5: hipLaunchKernelGGL((k_chol_2t<D, 0>), g2, dim3(256), 0, 0, a); break;
        case 6: hipLaunchKernelGGL((k_chol_2t<D, 1>), g2, dim3(256), 0, 0, a); break;
        case 7: hipLaunchKernelGGL((k_chol_loop<D, 0>), dim3((unsigned)N), dim3(64), 0, 0, a); break;
        case 8: hipLaunchKernelGGL((k_chol_loop<D, 1>), dim3((unsigned)N), dim3(64), 0, 0, a); break;
        case 9: hipLaunchKernelGGL((k_chol_loop<D, 2>), dim3((unsigned)N), dim3(64), 0, 0, a); break;
        default: hipLaunchKernelGGL((k_chol_loop<D, 3>), dim3((unsigned)N), dim3(64), 0, 0, a); break;
      }
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best) best = ms;
    }
    CK(hipGetLastError());
    printf("%-26s %.1f us per launch (%lld chains, d = %d)", names[which], best * 1e3, (long long)N, d);
    if (which == 0) {
      CK(hipMemcpy(h1.data(), L1, h1.size() * 8, hipMemcpyDeviceToHost));
      printf("\n");
    } else {
      CK(hipMemcpy(h2.data(), L2, h2.size() * 8, hipMemcpyDeviceToHost));
      double worst = 0.0;
      for (size_t i = 0; i < h1.size(); ++i) {
        const double df = fabs(h1[i] - h2[i]);
        const double sc = fabs(h1[i]) > 1e-300 ? df / fabs(h1[i]) : df;
        if (sc > worst) worst = sc;
      }
      printf("   max rel. deviation from k_chol: %.2e\n", worst);
    }
  }
  std::vector<int32_t> hf(N);
  CK(hipMemcpy(hf.data(), flags, N * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (auto f : hf) bad += f != 0;
  printf("flagged chains: %d\n", bad);
  return 0;
}
