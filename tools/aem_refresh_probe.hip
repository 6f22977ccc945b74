// k_aem_refresh (tda_kernels_aemr.h) on its own: correctness against a host Cholesky (factor form) / triangular inverse, and launch time at
// 4096 chains.  Debug tool, not part of the library.
// Build: hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -w -Itinyda_amd/csrc -Iinclude -o /tmp/arp tools/aem_refresh_probe.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include <string>
#include "tinyda_amd.h"
#include "tda_kernels_aemr.h"
using namespace tda;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int T>
static void launch(const AemRefreshArgs& a) {
  if (a.nsum == 1) hipLaunchKernelGGL((k_aem_refresh<T, 1>), dim3((unsigned)a.N), dim3(64), 0, 0, a);
  else if (a.nsum == 2) hipLaunchKernelGGL((k_aem_refresh<T, 2>), dim3((unsigned)a.N), dim3(64), 0, 0, a);
  else hipLaunchKernelGGL((k_aem_refresh<T, 3>), dim3((unsigned)a.N), dim3(64), 0, 0, a);
}

template <int T>
static void launch_big(const AemRefreshArgs& a) {
  if (a.nsum == 1) hipLaunchKernelGGL((k_aem_refresh_big<T, 1>), dim3((unsigned)a.N), dim3(64), 0, 0, a);
  else if (a.nsum == 2) hipLaunchKernelGGL((k_aem_refresh_big<T, 2>), dim3((unsigned)a.N), dim3(64), 0, 0, a);
  else hipLaunchKernelGGL((k_aem_refresh_big<T, 3>), dim3((unsigned)a.N), dim3(64), 0, 0, a);
}
static bool g_force_big = false;  // argv[2] = "big": the run-time-loop kernel at every width (at 64 / 128 it must write the register kernel's bits)

static int run_case(int m, int64_t N, int nsum, int reps) {
  const int MP = m <= 64 ? 64 : (m <= 128 ? 128 : 256), T = MP / 16;
  const int nlev = 3, k = 0;
  std::mt19937_64 g(5 + m + nsum);
  std::normal_distribution<double> nd;
  const int NV = 4;  // distinct matrices, dealt round-robin
  std::vector<double> cov((size_t)MP * MP, 0.0);
  for (int i = 0; i < m; ++i) cov[(size_t)i * MP + i] = 0.01;
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < i; ++j) cov[(size_t)i * MP + j] = cov[(size_t)j * MP + i] = 1e-4 * nd(g) / m;
  std::vector<std::vector<double>> sig(nsum * NV, std::vector<double>((size_t)MP * MP, 0.0));
  std::vector<double> x(m);
  for (auto& sgm : sig)
    for (int s = 0; s < 20; ++s) {  // (x x^T accumulated in one order: bitwise symmetric)
      for (auto& v : x) v = 0.05 * nd(g);
      for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) sgm[(size_t)i * MP + j] += x[i] * x[j] / 20;
    }
  std::vector<double> rv((size_t)NV * MP, 0.0);
  for (int v = 0; v < NV; ++v)
    for (int i = 0; i < m; ++i) rv[(size_t)v * MP + i] = 0.1 * nd(g);
  // the update of tracker 0 (state-independent form), t = 7
  const double tt = 7.0;
  std::vector<double> upd((size_t)NV * 3 * MP, 0.0);
  for (int v = 0; v < NV; ++v)
    for (int i = 0; i < m; ++i) {
      upd[((size_t)v * 3 + 0) * MP + i] = 0.05 * nd(g);
      upd[((size_t)v * 3 + 1) * MP + i] = 0.02 * nd(g);
      upd[((size_t)v * 3 + 2) * MP + i] = (1.0 / (tt + 1.0)) * (tt * upd[((size_t)v * 3 + 1) * MP + i] + upd[((size_t)v * 3 + 0) * MP + i]);
    }
  std::vector<std::vector<double>> sig0new(NV, std::vector<double>((size_t)MP * MP, 0.0));  // what the kernel must leave in tracker 0
  for (int v = 0; v < NV; ++v)
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < m; ++j) {
        const double* u = &upd[(size_t)v * 3 * MP];
        const double M = (tt * (u[MP + i] * u[MP + j]) - (tt + 1.0) * (u[2 * MP + i] * u[2 * MP + j])) + u[i] * u[j];
        sig0new[v][(size_t)i * MP + j] = (tt - 1.0) / tt * sig[0 * NV + v][(size_t)i * MP + j] + 1.0 / tt * M;
      }
  double *dcov, *dsig[3] = {nullptr, nullptr, nullptr}, *dV, *drv, *dll, *dS, *dupd;
  int64_t* dsid;
  const size_t VD = aemr_v_doubles(MP);
  auto tiles_of = [&](const std::vector<double>& full, bool ident) {  // symmetric row-major -> upper tiles
    std::vector<double> t(VD, 0.0);
    for (int i = 0; i < MP; ++i)
      for (int j = i; j < MP; ++j) t[aemr_u_offset(MP, i, j)] = (i < m && j < m) ? full[(size_t)i * MP + j] : ((ident && i == j) ? 1.0 : 0.0);
    return t;
  };
  CK(hipMalloc(&dcov, VD * 8));
  CK(hipMemcpy(dcov, tiles_of(cov, true).data(), VD * 8, hipMemcpyHostToDevice));
  for (int s = 0; s < nsum; ++s) CK(hipMalloc(&dsig[s], (size_t)N * VD * 8));
  auto reset_trackers = [&]() {
    for (int s = 0; s < nsum; ++s)
      for (int64_t c = 0; c < N; ++c) hipMemcpy(dsig[s] + (size_t)c * VD, tiles_of(sig[s * NV + c % NV], false).data(), VD * 8, hipMemcpyHostToDevice);
  };
  reset_trackers();
  CK(hipMalloc(&dupd, (size_t)N * 3 * MP * 8));
  for (int64_t c = 0; c < N; ++c) CK(hipMemcpy(dupd + (size_t)c * 3 * MP, upd.data() + (size_t)(c % NV) * 3 * MP, 3 * MP * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&dV, (size_t)N * VD * 8));
  CK(hipMemset(dV, 0, (size_t)N * VD * 8));
  CK(hipMalloc(&drv, (size_t)N * MP * 8));
  for (int64_t c = 0; c < N; ++c) CK(hipMemcpy(drv + (size_t)c * MP, rv.data() + (size_t)(c % NV) * MP, MP * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&dll, (size_t)nlev * N * 8));
  CK(hipMalloc(&dS, (size_t)3 * 2 * N * 8));
  CK(hipMalloc(&dsid, (size_t)nlev * N * 8));
  CK(hipMemset(dsid, 0, (size_t)nlev * N * 8));
  AemRefreshArgs a{};
  a.N = N; a.NP = N; a.m = m; a.MP = MP; a.nsum = nsum; a.cov = dcov;
  for (int s = 0; s < nsum; ++s) a.sig[s] = dsig[s];
  a.V = dV; a.rvec = drv; a.ll = dll; a.Sst = dS; a.sid = dsid; a.nlev = nlev; a.k = k;
  a.upd = nullptr; a.b_t = (int64_t)tt;
  auto go = [&]() {
    if (T == 16) launch_big<16>(a);
    else if (g_force_big) { if (T == 4) launch_big<4>(a); else launch_big<8>(a); }
    else if (T == 4) launch<4>(a);
    else launch<8>(a);
  };
  // timing: no update (the trackers stay what they are over the repetitions; the arithmetic of the update runs all the same)
  go();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) go();
  hipEventRecord(e1, 0);
  CK(hipEventSynchronize(e1));
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  // the checked launch: with the update of tracker 0
  a.upd = dupd;
  go();
  CK(hipDeviceSynchronize());
#ifdef AEMR_TRACE
  {
    long long st[64];
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(g_aemr_trace), sizeof st));
    printf("  trace of chain 0 (cycles): load+precheck %lld;", st[1] - st[0]);
    long long tot[6] = {0, 0, 0, 0, 0, 0};
    for (int q = 0; q < T; ++q) {
      const long long* b = st + 2 + 6 * q;
      const long long prev = q == 0 ? st[1] : st[2 + 6 * (q - 1) + 5];
      tot[0] += b[0] - prev; tot[1] += b[1] - b[0]; tot[2] += b[2] - b[1]; tot[3] += b[3] - b[2]; tot[4] += b[4] - b[3]; tot[5] += b[5] - b[4];
    }
    printf(" row sums+update %lld, left-looking U %lld, diagonal tiles %lld, row scaling %lld, factor out + z %lld, (unused) %lld; total %lld\n",
           tot[0], tot[1], tot[2], tot[3], tot[4], tot[5], st[2 + 6 * T] - st[0]);
  }
#endif
  // check chains 0 .. NV-1 and the last one
  double worstV = 0, worstL = 0;
  std::vector<double> V(VD), ll(N);
  CK(hipMemcpy(ll.data(), dll, N * 8, hipMemcpyDeviceToHost));
  for (int64_t c : {(int64_t)0, (int64_t)1, (int64_t)2, (int64_t)3, N - 1}) {
    if (c >= N) continue;
    CK(hipMemcpy(V.data(), dV + (size_t)c * VD, VD * 8, hipMemcpyDeviceToHost));
    std::vector<long double> S((size_t)m * m), L((size_t)m * m, 0.0L), W((size_t)m * m, 0.0L);
    for (int i = 0; i < m; ++i)
      for (int j = 0; j < m; ++j) {
        double sb = sig0new[c % NV][(size_t)i * MP + j];
        for (int s = 1; s < nsum; ++s) sb += sig[s * NV + c % NV][(size_t)i * MP + j];
        S[(size_t)i * m + j] = cov[(size_t)i * MP + j] + sb;
      }
    for (int j = 0; j < m; ++j) {
      long double d = S[(size_t)j * m + j];
      for (int k2 = 0; k2 < j; ++k2) d -= L[(size_t)j * m + k2] * L[(size_t)j * m + k2];
      L[(size_t)j * m + j] = sqrtl(d);
      for (int i = j + 1; i < m; ++i) {
        long double s = S[(size_t)i * m + j];
        for (int k2 = 0; k2 < j; ++k2) s -= L[(size_t)i * m + k2] * L[(size_t)j * m + k2];
        L[(size_t)i * m + j] = s / L[(size_t)j * m + j];
      }
    }
    for (int j = 0; j < m; ++j) {  // column j of W = L^-1
      W[(size_t)j * m + j] = 1.0L / L[(size_t)j * m + j];
      for (int i = j + 1; i < m; ++i) {
        long double s = 0;
        for (int k2 = j; k2 < i; ++k2) s += L[(size_t)i * m + k2] * W[(size_t)k2 * m + j];
        W[(size_t)i * m + j] = -s / L[(size_t)i * m + i];
      }
    }
    // the factor form (round 5): off-diagonal tiles hold L (as U = L^T tiles), the diagonal tiles the inverses of L's diagonal tiles
    std::vector<long double> Dinv((size_t)MP * 16, 0.0L);  // [row][col within the row's diagonal tile]
    for (int qb = 0; qb * 16 < m; ++qb)
      for (int j = 0; j < 16 && qb * 16 + j < m; ++j) {
        const int J = qb * 16 + j;
        Dinv[(size_t)J * 16 + j] = 1.0L / L[(size_t)J * m + J];
        for (int i = j + 1; i < 16 && qb * 16 + i < m; ++i) {
          const int I = qb * 16 + i;
          long double s = 0;
          for (int k2 = j; k2 < i; ++k2) s += L[(size_t)I * m + qb * 16 + k2] * Dinv[(size_t)(qb * 16 + k2) * 16 + j];
          Dinv[(size_t)I * 16 + j] = -s / L[(size_t)I * m + I];
        }
      }
    long double q = 0;
    for (int i = 0; i < m; ++i) {
      long double z = 0;
      for (int j = 0; j <= i; ++j) {
        z += W[(size_t)i * m + j] * rv[(size_t)(c % NV) * MP + j];
        const bool same = (i >> 4) == (j >> 4);
        const double got = same ? V[aemr_v_offset(i, j)] : V[aemr_w_offset_offdiag(i, j)];
        const double ref = same ? (double)Dinv[(size_t)i * 16 + (j & 15)] : (double)L[(size_t)i * m + j];
        worstV = fmax(worstV, fabs(got - ref) / (fabs(ref) + 1.0));
      }
      q += z * z;
    }
    const double llref = (double)(-0.5L * q);
    worstL = fmax(worstL, fabs(ll[c] - llref) / fabs(llref));
    // tracker 0 after the launch: the reference's arithmetic, bit for bit
    std::vector<double> T0(VD);
    CK(hipMemcpy(T0.data(), dsig[0] + (size_t)c * VD, VD * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < m; ++i)
      for (int j = i; j < m; ++j)
        if (T0[aemr_u_offset(MP, i, j)] != sig0new[c % NV][(size_t)i * MP + j]) worstV = 1.0;
  }
  const double us = ms * 1000.0 / reps;
  const double flops = (double)N * (2.0 / 3.0) * pow((double)MP, 3);
  unsigned long long hsh = 1469598103934665603ull;  // FNV-1a over chain 0's factor and log-likelihood: equal bits <=> equal hash across builds / kernels
  {
    CK(hipMemcpy(V.data(), dV, VD * 8, hipMemcpyDeviceToHost));
    const unsigned char* b = reinterpret_cast<const unsigned char*>(V.data());
    for (size_t i = 0; i < VD * 8; ++i) hsh = (hsh ^ b[i]) * 1099511628211ull;
    b = reinterpret_cast<const unsigned char*>(ll.data());
    for (size_t i = 0; i < 8; ++i) hsh = (hsh ^ b[i]) * 1099511628211ull;
  }
  printf("m=%3d MP=%3d nsum=%d N=%lld: %8.1f us / launch  (%.1f TFLOP/s at 2/3 m^3)   max rel err V %.2e, ll %.2e   bits %016llx\n", m, MP, nsum, (long long)N, us,
         flops / (us * 1e-6) / 1e12, worstV, worstL, hsh);
  hipFree(dcov); for (int s = 0; s < nsum; ++s) hipFree(dsig[s]);
  hipFree(dV); hipFree(drv); hipFree(dll); hipFree(dS); hipFree(dsid); hipFree(dupd);
  return (worstV < 1e-10 && worstL < 1e-10) ? 0 : 2;
}

int main(int argc, char** argv) {
  const int64_t N = argc > 1 ? atoll(argv[1]) : 4096;
  g_force_big = argc > 2 && std::string(argv[2]) == "big";
  int rc = 0;
  rc |= run_case(256, N, 2, 3);
  rc |= run_case(200, N, 1, 3);
  rc |= run_case(130, N, 3, 2);
  rc |= run_case(128, N, 2, 10);
  rc |= run_case(128, N, 1, 10);
  rc |= run_case(100, N, 3, 5);
  rc |= run_case(64, N, 2, 10);
  rc |= run_case(40, N, 1, 5);
  rc |= run_case(8, N, 1, 5);
  printf(rc ? "FAILED\n" : "all ok\n");
  return rc;
}
