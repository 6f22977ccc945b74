// Does a small-register RNG kernel run in the shadow of the MFMA-bound step kernel?  (debug tool)
// k_mh_steps<64,8> takes 2 x 224 of the 512 registers of a SIMD; a kernel limited to 64 registers can share the SIMD and
// use the VALU slots the matrix phase leaves idle.  Build on the GPU box:
//   hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -w -Itinyda_amd/csrc -o /tmp/ovl tools/overlap_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
#include "tda_kernels_mh.h"
using namespace tda;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
template <class T> T* dev(const std::vector<T>& h) {
  T* p = nullptr;
  if (hipMalloc((void**)&p, h.size() * sizeof(T)) != hipSuccess) return nullptr;
  (void)hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
  return p;
}

// normals for S steps x N chains x 64 dims: wave per (chain, 16-step group), lane = (step lc, quarter hi) as in k_propose
__global__ void __launch_bounds__(64, 8) k_rng_probe(double* z, int64_t N, int S, uint64_t seed) {
  const int lane = threadIdx.x, lc = lane & 15, hi = lane >> 4;
  const int64_t c = blockIdx.x;
  const int s = blockIdx.y * 16 + lc;
  if (s >= S) return;
#pragma unroll 1
  for (int q = 0; q < 8; ++q) {
    const int p = 4 * q + hi;
    double z0, z1;
    normal_pair(seed, (uint32_t)c, (uint32_t)s, STREAM_PROPOSAL, (uint32_t)p, z0, z1);
    *reinterpret_cast<double2*>(&z[((size_t)s * N + c) * 64 + 2 * p]) = double2{z0, z1};
  }
}

int main(int argc, char** argv) {
  const int m = 1024;
  constexpr int D = 64;
  const int64_t N = 4096, NP = 4096;
  const int S = 100, ncb = m / 16;
  std::mt19937_64 g(1);
  std::normal_distribution<double> nd;
  std::uniform_real_distribution<double> ud(1e-12, 1.0);
  std::vector<double> Apk((size_t)m * D), ytil(m), theta((size_t)NP * D), lp(NP, 0.0), ll(NP, -1e300), scal(NP, 1.0);
  for (auto& v : Apk) v = nd(g) / 8;
  for (auto& v : ytil) v = nd(g);
  for (auto& v : theta) v = nd(g);
  std::vector<double> inc((size_t)S * NP * D), u((size_t)S * NP), lu((size_t)S * NP), zeros(D, 0.0), ones(D, 1.0);
  for (auto& v : inc) v = 0.01 * nd(g);
  for (size_t i = 0; i < u.size(); ++i) { u[i] = ud(g); lu[i] = std::log(u[i]); }
  StepArgs a{};
  a.lv.Apk = dev(Apk); a.lv.ytil = dev(ytil); a.lv.ncb = ncb; a.lv.m_pad = m; a.lv.noise_kind = 0; a.lv.var = 0.01;
  a.pr.mean = dev(zeros); a.pr.pinv = dev(ones); a.pr.kind = PRIOR_STANDARD; a.pr.logconst = D * std::log(2 * M_PI);
  a.N = N; a.NP = NP; a.d = D; a.S = S; a.mode = MODE_STEP; a.prop_kind = 0;
  a.theta = dev(theta); a.lp = dev(lp); a.ll = dev(ll); a.scaling = dev(scal);
  std::vector<int32_t> ac(NP, 0);
  a.acc_count = dev(ac);
  a.inc = dev(inc); a.u = dev(u); a.logu = dev(lu);
  double *rp, *rs, *zb; uint8_t* ra;
  CK(hipMalloc((void**)&rp, (size_t)S * N * D * 8)); CK(hipMalloc((void**)&rs, (size_t)S * N * 3 * 8)); CK(hipMalloc((void**)&ra, (size_t)S * N));
  CK(hipMalloc((void**)&zb, (size_t)S * N * D * 8));
  a.rec_params = rp; a.rec_stats = rs; a.rec_acc = ra;
  const size_t lds = (16 * (D + 2) + 256 + 2 * D + m) * 8;
  hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  hipEvent_t e0, e1, f0, f1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
  auto steps = [&](hipStream_t st) { hipLaunchKernelGGL((k_mh_steps<D, 8>), dim3(NP / 16), dim3(512), lds, st, a); };
  auto rng = [&](hipStream_t st) { hipLaunchKernelGGL(k_rng_probe, dim3(N, (S + 15) / 16), dim3(64), 0, st, zb, N, S, 7ull); };
  for (int i = 0; i < 3; ++i) { steps(s1); rng(s1); }
  CK(hipDeviceSynchronize());
  const int R = 10;
  float ms_steps, ms_rng, ms_both_a, ms_both_b, ms_wall;
  CK(hipEventRecord(e0, s1)); for (int i = 0; i < R; ++i) steps(s1); CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms_steps, e0, e1));
  CK(hipEventRecord(e0, s1)); for (int i = 0; i < R; ++i) rng(s1); CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms_rng, e0, e1));
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, s1)); CK(hipEventRecord(f0, s2));
  for (int i = 0; i < R; ++i) { steps(s1); rng(s2); }
  CK(hipEventRecord(e1, s1)); CK(hipEventRecord(f1, s2));
  CK(hipEventSynchronize(e1)); CK(hipEventSynchronize(f1));
  CK(hipEventElapsedTime(&ms_both_a, e0, e1)); CK(hipEventElapsedTime(&ms_both_b, f0, f1));
  CK(hipEventElapsedTime(&ms_wall, e0, f1));
  // one pair, both released by the same event: when does the rng kernel finish relative to the step kernel?
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, s1));
    CK(hipStreamWaitEvent(s2, e0, 0));
    steps(s1);
    rng(s2);
    CK(hipEventRecord(e1, s1));
    CK(hipEventRecord(f1, s2));
    CK(hipEventSynchronize(e1));
    CK(hipEventSynchronize(f1));
    float ta, tb;
    CK(hipEventElapsedTime(&ta, e0, e1));
    CK(hipEventElapsedTime(&tb, e0, f1));
    printf("released together: steps done at %.3f ms, rng done at %.3f ms\n", ta, tb);
  }
  printf("alone: steps %.3f ms  rng %.3f ms  (sum %.3f)\n", ms_steps / R, ms_rng / R, (ms_steps + ms_rng) / R);
  printf("two streams: steps stream %.3f ms  rng stream %.3f ms per pair\n", ms_both_a / R, ms_both_b / R);
  return 0;
}
