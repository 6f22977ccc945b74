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

// variant 2: rows exchanged UNSCALED (the ds_bpermute round does not wait for the reciprocal square root), update with 1 / d,
// next pivot formed ahead of the update from two v_readlane
template <int KL>
__device__ __forceinline__ void pivot2(double (&C)[4], double (&Vd)[4], double& dkk, int lc, int hi) {
  constexpr int kh = KL & 3, kr = KL >> 2;
  double ucol[4];
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) ucol[rr] = rr >= kr ? aemr_pick(C[kr], kh * 16 + hi + 4 * rr) : 0.0;
  const double urc = aemr_pick(C[kr], kh * 16 + lc);
  const double urv = KL > 0 ? aemr_pick(Vd[kr], kh * 16 + lc) : ((lc == 0) ? 1.0 : 0.0);
  const double y0 = __builtin_amdgcn_rsq(dkk);
  const double e0 = fma(-dkk * y0, y0, 1.0);
  const double inv = fma(y0 * e0, fma(0.375, e0, 0.5), y0);
  const double inv2 = inv * inv;
  if constexpr (KL < 15) {
    constexpr int nh = (KL + 1) & 3, nr = (KL + 1) >> 2;
    const double c01 = bcast_lane64(C[kr], kh * 16 + KL + 1);
    const double c11 = bcast_lane64(C[nr], nh * 16 + KL + 1);
    dkk = fma(-(c01 * c01), inv2, c11);
  }
#pragma unroll
  for (int rr = 0; rr < 4; ++rr) {
    if (rr < kr) continue;
    const bool live = rr > kr || hi > kh;
    const double uc = fma(-(ucol[rr] * urc), inv2, C[rr]);
    const double uv = fma(-(ucol[rr] * urv), inv2, Vd[rr]);
    C[rr] = live ? uc : C[rr];
    Vd[rr] = live ? uv : Vd[rr];
  }
  const double fac = (hi == kh) ? inv : 1.0;
  C[kr] *= fac;
  Vd[kr] *= fac;
}
__device__ __forceinline__ void diag2(double (&C)[4], double (&Vd)[4], double (&Vt)[4], int lc, int hi) {
#pragma unroll
  for (int r = 0; r < 4; ++r) Vd[r] = (hi + 4 * r == lc) ? 1.0 : 0.0;
  double dkk = bcast_lane64(C[0], 0);
  aemr_static_for<16>([&](auto kc) { pivot2<decltype(kc)::value>(C, Vd, dkk, lc, hi); });
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int src = (lc & 3) * 16 + hi + 4 * r;
    const double t0 = aemr_pick(Vd[0], src), t1 = aemr_pick(Vd[1], src), t2 = aemr_pick(Vd[2], src), t3 = aemr_pick(Vd[3], src);
    const int sr = lc >> 2;
    Vt[r] = sr == 0 ? t0 : (sr == 1 ? t1 : (sr == 2 ? t2 : t3));
  }
}
// variant 3: the factorisation alone (no V part, no transpose): how the time scales with the instruction count
__device__ __forceinline__ void diag3(double (&C)[4], double (&Vd)[4], double (&Vt)[4], int lc, int hi) {
#pragma unroll
  for (int kl = 0; kl < 16; ++kl) {
    const int kh = kl & 3, kr = kl >> 2;
    const double dkk = bcast_lane64(C[kr], kh * 16 + kl);
    const double y0 = __builtin_amdgcn_rsq(dkk);
    const double e0 = fma(-dkk * y0, y0, 1.0);
    const double inv = fma(y0 * e0, fma(0.375, e0, 0.5), y0);
    const double fac = (hi == kh) ? inv : 1.0;
    C[kr] *= fac;
    double ucol[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) ucol[rr] = rr >= kr ? aemr_pick(C[kr], kh * 16 + hi + 4 * rr) : 0.0;
    const double urc = aemr_pick(C[kr], kh * 16 + lc);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      if (rr < kr) continue;
      const bool live = rr > kr || hi > kh;
      const double uc = fma(-ucol[rr], urc, C[rr]);
      C[rr] = live ? uc : C[rr];
    }
  }
  for (int r = 0; r < 4; ++r) Vd[r] = Vt[r] = C[r];
}

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
    if constexpr (VARIANT == 0) aemr_diag(C, Vd, Vt, lc, hi);  // (the DPP / permlane form: see the note in tda_kernels_aemr.h)
    else if constexpr (VARIANT == 1) diag2(C, Vd, Vt, lc, hi);
    else diag3(C, Vd, Vt, lc, hi);
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
  for (int v = 0; v < 3; ++v) {
    for (int rep = 0; rep < 2; ++rep) {
      if (v == 0) hipLaunchKernelGGL(k_probe<0>, dim3(blocks), dim3(64), 0, 0, dt, dout, dc, reps);
      else if (v == 1) hipLaunchKernelGGL(k_probe<1>, dim3(blocks), dim3(64), 0, 0, dt, dout, dc, reps);
      else hipLaunchKernelGGL(k_probe<2>, dim3(blocks), dim3(64), 0, 0, dt, dout, dc, reps);
      hipDeviceSynchronize();
    }
    hipMemcpy(c.data(), dc, (size_t)blocks * 8, hipMemcpyDeviceToHost);
    if (v < 2) hipMemcpy(v == 0 ? o0.data() : o1.data(), dout, 64 * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto x : c) s += (double)x;
    printf("variant %d: %.0f counter ticks per tile (mean over %d waves; s_memtime ticks at 100 MHz: x24 for 2.4 GHz core cycles)\n", v, s / blocks / reps, blocks);
  }
  double d = 0;
  for (int i = 0; i < 64; ++i) d = fmax(d, fabs(o0[i] - o1[i]) / (fabs(o0[i]) + 1e-300));
  printf("max rel difference of the two variants' sums: %.2e\n", d);
  return 0;
}
