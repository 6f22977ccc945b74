// Cycle count of the diagonal-tile step of k_aem_refresh (aemr_diag, tda_kernels_aemr.h) in isolation: one wave per workgroup,
// `reps` factorisations of the same tile back to back, s_memtime around them.  Debug tool, not part of the library.
// Build: hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -w -Itinyda_amd/csrc -Iinclude -o /tmp/adp tools/aem_diag_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "tinyda_amd.h"
#include "tda_kernels_aemr.h"
using namespace tda;

template <int VARIANT>
__global__ void __launch_bounds__(64) k_probe(const double* tile, double* out, long long* cyc, int reps) {
  const int lane = threadIdx.x, lc = lane & 15, hi = lane >> 4;
  double C0[4];
  for (int r = 0; r < 4; ++r) C0[r] = tile[(hi + 4 * r) * 16 + lc];
  double acc = 0.0;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < reps; ++it) {
    double C[4], Vd[4], Vt[4];
    for (int r = 0; r < 4; ++r) C[r] = C0[r] + acc * 1e-300;  // (serialises the repetitions)
    aemr_diag(C, Vd, Vt, lc, hi);  // (VARIANT 1 was the DPP / permlane form, see the note in tda_kernels_aemr.h)
    acc += Vt[0] + Vt[1] + Vt[2] + Vt[3] + Vd[0];
  }
  const long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cyc[blockIdx.x] = t1 - t0;
  out[blockIdx.x * 64 + lane] = acc;
}

int main(int argc, char** argv) {
  const int blocks = argc > 1 ? atoi(argv[1]) : 1024, reps = 50;
  std::vector<double> t(256);
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) t[i * 16 + j] = (i == j ? 2.0 : 0.0) + 0.01 * ((i * 7 + j * 3) % 5 + (j * 7 + i * 3) % 5);
  double *dt, *dout;
  long long* dc;
  hipMalloc(&dt, 256 * 8);
  hipMalloc(&dout, (size_t)blocks * 64 * 8);
  hipMalloc(&dc, (size_t)blocks * 8);
  hipMemcpy(dt, t.data(), 256 * 8, hipMemcpyHostToDevice);
  std::vector<long long> c(blocks);
  std::vector<double> o0(64), o1(64);
  for (int v = 0; v < 1; ++v) {
    for (int rep = 0; rep < 2; ++rep) {
      if (v == 0) hipLaunchKernelGGL(k_probe<0>, dim3(blocks), dim3(64), 0, 0, dt, dout, dc, reps);
      else hipLaunchKernelGGL(k_probe<1>, dim3(blocks), dim3(64), 0, 0, dt, dout, dc, reps);
      hipDeviceSynchronize();
    }
    hipMemcpy(c.data(), dc, (size_t)blocks * 8, hipMemcpyDeviceToHost);
    hipMemcpy(v == 0 ? o0.data() : o1.data(), dout, 64 * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto x : c) s += (double)x;
    printf("variant %d: %.0f counter ticks per tile (mean over %d waves; s_memtime ticks at 100 MHz: x24 for 2.4 GHz core cycles)\n", v, s / blocks / reps, blocks);
  }
  double d = 0;
  for (int i = 0; i < 64; ++i) d = fmax(d, fabs(o0[i] - o1[i]) / (fabs(o0[i]) + 1e-300));
  printf("max rel difference of the two variants' sums: %.2e\n", d);
  return 0;
}
