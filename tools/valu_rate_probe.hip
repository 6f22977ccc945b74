// Issue rate of the fp64 vector instructions on gfx950 (debug tool): cycles per wave64 instruction for independent chains of
// v_fma_f64 / v_mul_f64 / v_add_f64 / v_mov_b64 and the k_adapt mix (5 mul : 3 add : 1 mov), with 1, 2 and 4 waves per SIMD.
// Build on the GPU box: hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -w -o /tmp/vrp tools/valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int NV = 16;     // independent accumulators per lane
constexpr int ITER = 20000;  // loop trips; NV instructions of the kind per trip

template <int KIND>
__global__ void __launch_bounds__(64) k_rate(double* out, long long* cyc, double seed) {
  double v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) v[i] = seed + 1e-3 * (threadIdx.x + 64 * i);
  const double m = 1.0 + 1e-9 * seed, b = 1e-12 * seed;
  const long long t0 = (long long)__builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (KIND == 0) v[i] = fma(v[i], m, b);
      else if (KIND == 1) v[i] = v[i] * m;
      else if (KIND == 2) v[i] = v[i] + b;
      else if (KIND == 3) asm volatile("v_mov_b64 %0, %1" : "=v"(v[i]) : "v"(v[(i + 1) % NV]));
      else {  // the k_adapt element: tp = t1 * (pr * pc); M = (TM - tp) + xr * xc; Sg = ca * Sg + cb * M; TM = tp  (5 mul, 3 add)
        const double tp = m * (v[i] * m);
        const double M = (v[(i + 1) % NV] - tp) + v[i] * b;
        v[i] = m * v[i] + b * M;
      }
    }
  }
  const long long t1 = (long long)__builtin_amdgcn_s_memtime();
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += v[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  const int maxb = 256 * 4 * 4;
  double* out; long long* cyc;
  CK(hipMalloc(&out, (size_t)maxb * 64 * 8));
  CK(hipMalloc(&cyc, (size_t)maxb * 8));
  const char* names[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_mov_b64", "k_adapt element (5 mul + 3 add)"};
  const int per_trip[] = {NV, NV, NV, NV, 8 * NV};
  for (int kind = 0; kind < 5; ++kind)
    for (int wps : {1, 2, 4}) {
      const int nb = 256 * 4 * wps;
      hipEvent_t e0, e1;
      CK(hipEventCreate(&e0));
      CK(hipEventCreate(&e1));
      float ms = 0.f;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        switch (kind) {
          case 0: hipLaunchKernelGGL(k_rate<0>, dim3(nb), dim3(64), 0, 0, out, cyc, 1.0); break;
          case 1: hipLaunchKernelGGL(k_rate<1>, dim3(nb), dim3(64), 0, 0, out, cyc, 1.0); break;
          case 2: hipLaunchKernelGGL(k_rate<2>, dim3(nb), dim3(64), 0, 0, out, cyc, 1.0); break;
          case 3: hipLaunchKernelGGL(k_rate<3>, dim3(nb), dim3(64), 0, 0, out, cyc, 1.0); break;
          default: hipLaunchKernelGGL(k_rate<4>, dim3(nb), dim3(64), 0, 0, out, cyc, 1.0); break;
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
      }
      std::vector<long long> h(nb);
      CK(hipMemcpy(h.data(), cyc, (size_t)nb * 8, hipMemcpyDeviceToHost));
      std::sort(h.begin(), h.end());
      const double med = (double)h[nb / 2];
      // cycles of the SIMD per instruction = wave cycles / instructions / waves sharing the SIMD
      // and from the wall time of the launch (all waves resident at once) at 2.4 GHz
      printf("%-34s %d wave(s)/SIMD: %7.2f cycles per instruction and wave, %6.2f SIMD cycles per instruction; launch %.1f us -> %.2f\n",
             names[kind], wps, med / ((double)ITER * per_trip[kind]), med / ((double)ITER * per_trip[kind]) / wps, ms * 1e3,
             ms * 1e-3 * 2.4e9 / ((double)ITER * per_trip[kind] * wps));
    }
  return 0;
}
