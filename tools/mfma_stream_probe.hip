// How fast can ONE wave per SIMD stream 16x16x4 f64 MFMAs when its A operand comes from L2 and the B operand from
// registers?  (debug probe for the design of the step kernel; not part of the library)
//   variant 0: fragments resident in registers (no loads), one dependent accumulator chain per block
//   variant 1: fragments loaded per block (8 x dwordx4, prefetch distance 2 blocks), one chain
//   variant 2: as 1, two blocks interleaved (two independent accumulator chains)
//   variant 3: as 0 with two chains
// hipcc -O3 --offload-arch=gfx950 -w -o /tmp/msp tools/mfma_stream_probe.hip && /tmp/msp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__device__ __forceinline__ double4_t mf(double a, double b, double4_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

template <int VAR, int WAVES>
__global__ void __launch_bounds__(64 * WAVES, WAVES / 4) k_probe(const double2* __restrict__ A, int ncb, int iters, double* out, long long* cyc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double th[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) th[k] = 1.0 + 0.001 * (lane + k);
  const double2* base = A + lane;
  double2 f0[8], f1[8], f2[8];
  auto load = [&](double2 (&f)[8], int cb) {
    const double2* p = base + (size_t)(cb % ncb) * 8 * 64;
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = p[k * 64];
  };
  load(f0, wave); load(f1, wave + WAVES); load(f2, wave + 2 * WAVES);
  double4_t s = {0, 0, 0, 0};
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    for (int cb = wave; cb < ncb; cb += 3 * WAVES) {
      double4_t a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0};
      if (VAR == 0 || VAR == 1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { a0 = mf(f0[k].x, th[2 * k], a0); a0 = mf(f0[k].y, th[2 * k + 1], a0); if (VAR == 1 && k == 7) load(f0, cb + 3 * WAVES); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 8; ++k) { a1 = mf(f1[k].x, th[2 * k], a1); a1 = mf(f1[k].y, th[2 * k + 1], a1); if (VAR == 1 && k == 7) load(f1, cb + 4 * WAVES); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 8; ++k) { a2 = mf(f2[k].x, th[2 * k], a2); a2 = mf(f2[k].y, th[2 * k + 1], a2); if (VAR == 1 && k == 7) load(f2, cb + 5 * WAVES); }
        __builtin_amdgcn_sched_barrier(0);
      } else {
        // three blocks, chains interleaved: a0, a1, a2 round robin
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          a0 = mf(f0[k].x, th[2 * k], a0); a1 = mf(f1[k].x, th[2 * k], a1); a2 = mf(f2[k].x, th[2 * k], a2);
          a0 = mf(f0[k].y, th[2 * k + 1], a0); a1 = mf(f1[k].y, th[2 * k + 1], a1); a2 = mf(f2[k].y, th[2 * k + 1], a2);
        }
        if (VAR == 2) { load(f0, cb + 3 * WAVES); load(f1, cb + 4 * WAVES); load(f2, cb + 5 * WAVES); }
        __builtin_amdgcn_sched_barrier(0);
      }
      s += a0 + a1 + a2;
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0 && blockIdx.x == 0) cyc[wave] = t1 - t0;
  out[(size_t)blockIdx.x * 64 * WAVES + threadIdx.x] = s[0] + s[1] + s[2] + s[3] + f0[0].x + f1[0].x + f2[0].x;
}

template <int VAR, int WAVES>
int run(const double2* A, int ncb, double* out, long long* cyc) {
  const int iters = 200;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_probe<VAR, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, A, ncb, 5, out, cyc);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_probe<VAR, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, A, ncb, iters, out, cyc);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  long long h[8]; CK(hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost));
  const double blocks_per_simd = (double)iters * ncb / 4.0;
  const double flops = 256.0 * iters * ncb * 16 * 2048.0;
  printf("variant %d, %d waves/CU: %.3f ms  %.1f TFLOP/s   %.0f ticks per block per SIMD (1024 = 64 per MFMA)\n", VAR, WAVES, ms, flops / ms * 1e-9,
         (double)h[0] / blocks_per_simd);
  return 0;
}

int main() {
  const int ncb = 64 * 3 / 3 * 1;  // 64 blocks = m 1024 ... use 96 so that 4 waves x 3 sets divide evenly
  const int NCB = 96;
  std::vector<double> h((size_t)NCB * 8 * 64 * 2);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 1e-3 * (double)(i % 977);
  double2* A; double* out; long long* cyc;
  CK(hipMalloc((void**)&A, h.size() * 8)); CK(hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc((void**)&out, 256 * 512 * 8)); CK(hipMalloc((void**)&cyc, 64));
  (void)ncb;
  if (run<0, 4>(A, NCB, out, cyc)) return 1;
  if (run<3, 4>(A, NCB, out, cyc)) return 1;
  if (run<1, 4>(A, NCB, out, cyc)) return 1;
  if (run<2, 4>(A, NCB, out, cyc)) return 1;
  if (run<0, 8>(A, NCB, out, cyc)) return 1;
  if (run<1, 8>(A, NCB, out, cyc)) return 1;
  if (run<2, 8>(A, NCB, out, cyc)) return 1;
  return 0;
}
