// Stand-alone timing harness for the fused MH-step kernels (debug tool, not part of the library or the bench contract).
// Build on the GPU box:  hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -w -Itinyda_amd/csrc -Itools \
//                              [-DTDA_EXP_...] -o /tmp/smb tools/steps_microbench.hip
//        (-DTDA_STEP_TRACE additionally prints where the cycles of a step go; the stamps themselves cost a few percent)
// Run:  /tmp/smb [kernel: 0 = k_mh_steps<64,8>, 1 = k_mh_steps_frag<64,false,1>, 2 = the 4-wave gapless k_mh_steps_frag<64,false,1,4>, 3 = k_mh_steps_lin<64,false,1>] [records: 0 none, 1 stats only, 2 all] [m]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <random>
#include <algorithm>
#include "experimental/tda_kernels_mh_lin.h"
using namespace tda;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <class T> T* dev(const std::vector<T>& h) {
  T* p = nullptr;
  if (hipMalloc((void**)&p, h.size() * sizeof(T)) != hipSuccess) return nullptr;
  (void)hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
  return p;
}

int main(int argc, char** argv) {
  const int which = argc > 1 ? atoi(argv[1]) : 1;
  const int recs = argc > 2 ? atoi(argv[2]) : 2;
  const int m = argc > 3 ? atoi(argv[3]) : 1024;
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
  a.lv.Apk = dev(Apk); a.lv.ytil = dev(ytil); a.lv.w = nullptr; a.lv.Ppk = nullptr; a.lv.ncb = ncb; a.lv.m_pad = m;
  a.lv.noise_kind = 0; a.lv.var = 0.01;
  a.pr.mean = dev(zeros); a.pr.pinv = dev(ones); a.pr.Wpk = nullptr; a.pr.wmu = nullptr; a.pr.ncb = 0; a.pr.kind = PRIOR_STANDARD;
  a.pr.logconst = D * std::log(2 * M_PI);
  a.N = N; a.NP = NP; a.d = D; a.S = S; a.mode = MODE_STEP; a.prop_kind = 0;
  a.theta = dev(theta); a.lp = dev(lp); a.ll = dev(ll); a.scaling = dev(scal);
  std::vector<int32_t> ac(NP, 0);
  a.acc_count = dev(ac);
  a.inc = dev(inc); a.u = dev(u); a.logu = dev(lu);
  double *rp = nullptr, *rs = nullptr; uint8_t* ra = nullptr;
  CK(hipMalloc((void**)&rp, (size_t)S * N * D * 8)); CK(hipMalloc((void**)&rs, (size_t)S * N * 3 * 8)); CK(hipMalloc((void**)&ra, (size_t)S * N));
  a.rec_params = recs >= 2 ? rp : nullptr; a.rec_stats = recs >= 1 ? rs : nullptr; a.rec_acc = recs >= 1 ? ra : nullptr;
  a.trace = nullptr;
  size_t lds; const void* fn;
  if (which == 0) { lds = (16 * (D + 2) + 256 + 2 * D + m) * 8; fn = (const void*)&k_mh_steps<D, 8>; }
  else if (which == 1) { lds = steps_frag_lds_doubles<D>(m, false) * 8; fn = (const void*)&k_mh_steps_frag<D, false, 1>; }
  else if (which == 2) { lds = steps_frag_lds_doubles<D>(m, false) * 8; fn = (const void*)&k_mh_steps_frag<D, false, 1, 4>; }
  else { lds = steps_lin_lds_doubles<D>(m, false) * 8; fn = (const void*)&k_mh_steps_lin<D, false, 1>; }
  if (lds > 64 * 1024) CK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&]() {
    if (which == 0) hipLaunchKernelGGL((k_mh_steps<D, 8>), dim3(NP / 16), dim3(512), lds, 0, a);
    else if (which == 1) hipLaunchKernelGGL((k_mh_steps_frag<D, false, 1>), dim3(NP / 16), dim3(512), lds, 0, a);
    else if (which == 2) hipLaunchKernelGGL((k_mh_steps_frag<D, false, 1, 4>), dim3(NP / 16), dim3(256), lds, 0, a);
    else hipLaunchKernelGGL((k_mh_steps_lin<D, false, 1>), dim3(NP / 16), dim3(512), lds, 0, a);
  };
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  const int R = 20;
  CK(hipEventRecord(e0));
  for (int i = 0; i < R; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<int32_t> hac(NP);
  CK(hipMemcpy(hac.data(), a.acc_count, NP * 4, hipMemcpyDeviceToHost));
  long tot = 0; for (auto v : hac) tot += v;
  const double ns = ms * 1e6 / ((double)R * S * N);
#ifdef TDA_STEP_TRACE
  {  // one more launch with cycle stamps of tile 0: median interval between consecutive stamps, per wave
    long long* tr = nullptr;
    CK(hipMalloc((void**)&tr, ((size_t)S * 64 + 8) * 8));
    CK(hipMemset(tr, 0, ((size_t)S * 64 + 8) * 8));
    a.trace = tr;
    // the clock the chip holds under this kernel (MI355X_MICROARCH.md, DVFS give-back (6)): stamped after >= 2 s of
    // back-to-back launches; shader cycles per 100 MHz tick over the step loop of tile 0
    if (which == 0) {
      const int warm = (int)(2000.0 / (ms / R)) + 1;
      for (int i = 0; i < warm; ++i) launch();
      CK(hipDeviceSynchronize());
    }
    launch();
    CK(hipDeviceSynchronize());
    std::vector<long long> h((size_t)S * 64 + 8);
    CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
    const int ord_frag[8] = {0, 2, 3, 1, 4, 5, 6, 7}, ord_tm[8] = {0, 1, 2, 3, 4, 5, 6, 7}, ord_lin[8] = {0, 2, 3, 1, 4, 5, 6, 7};
    const int* ord = which == 3 ? ord_lin : (which >= 1 ? ord_frag : ord_tm);
    const int nwk = which == 2 ? 4 : 8;
    auto med = [&](int w, int ia, int ib, int shift) {
      std::vector<long long> v;
      for (int s = 5; s + shift < S - 5; ++s) v.push_back(h[((size_t)(s + shift) * nwk + w) * 8 + ib] - h[((size_t)s * nwk + w) * 8 + ia]);
      std::sort(v.begin(), v.end());
      return v[v.size() / 2];
    };
    printf("  stamp order within a step:");
    for (int i = 0; i < 8; ++i) printf(" %d", ord[i]);
    printf("   (cycles; columns = waves 0..7 of tile 0)\n  step period   ");
    for (int w = 0; w < nwk; ++w) printf(" %6lld", med(w, 0, 0, 1));
    printf("\n");
    for (int i = 0; i < 7; ++i) {
      printf("  %d -> %d        ", ord[i], ord[i + 1]);
      for (int w = 0; w < nwk; ++w) printf(" %6lld", med(w, ord[i], ord[i + 1], 0));
      printf("\n");
    }
    if (which == 0) {
      const long long* cs = h.data() + (size_t)S * 8 * 8;
      const double cyc = (double)(cs[2] - cs[0]), ticks = (double)(cs[3] - cs[1]);
      printf("  in-kernel clock: %.0f shader cycles in %.0f ticks of 100 MHz = %.3f GHz (%.0f cycles per step)\n", cyc, ticks, cyc / ticks * 0.1, cyc / S);
    }
    printf("  7 -> next 0   ");
    for (int w = 0; w < nwk; ++w) printf(" %6lld", med(w, 7, 0, 1));
    printf("\n");
  }
#endif
  printf("kernel %d records %d m %d: %.3f ms/launch  %.3f ns/eval  %.1f TFLOP/s   (acc rate %.3f)\n", which, recs, m, ms / R, ns,
         (2.0 * m * D + 3.0 * m + 2 * D) / ns * 1e-3, (double)tot / ((double)(R + 3) * S * N));
  return 0;
}
