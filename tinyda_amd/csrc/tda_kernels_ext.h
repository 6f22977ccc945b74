// Batched host-callback forward models: the engine hands ALL chains' proposals of a step to one host function and takes
// all model outputs back, so an arbitrary (vectorised) Python / C model runs behind the device engine's proposals,
// acceptance, adaptation and records.  The reference evaluates a Python callable once per chain and step
// (posterior.py:95-96; umbridge.py:56-80 does the same over HTTP); here the per-step unit is the [N][d] proposal matrix.
//
//   k_ext_propose  theta' = theta + scaling * inc[s]   (pCN: sqrt(1 - beta^2) theta + beta inc[s])  -> prop[N][d]
//   (host)         F[N][m] = callback(prop)
//   k_ext_accept   log-likelihood of F rows, log-prior of prop rows, accept test, state + record update
//
// One wave per chain in both kernels: lane j owns parameter j, the lanes stride over the m outputs.
#pragma once
#include <hip/hip_runtime.h>

namespace tda {

struct ExtArgs {
  long long N, NP;
  int d, DP, m, s, mode, prop_kind;  // s = step within the block; mode 1 = evaluate the current states (initial link);
                                     // prop_kind 1 = pCN (TDA_PROP_PCN)
  double* theta;
  double* lp;
  double* ll;
  const double* scaling;
  int* acc_count;
  const double* inc;  // [S][NP][DP] block of increments L z
  const double* u;    // [S][NP]
  double* prop;       // [N][d]  proposals as the host sees them
  const double* F;    // [N][m]  model outputs from the host
  const double* data;
  const double* w;  // 1 / diag(noise) or null (isotropic)
  double var;
  const double* pr_mean;
  const double* pr_pinv;
  const double* pr_lo;
  const double* pr_hi;
  double logconst;
  double* rec_params;
  double* rec_stats;
  unsigned char* rec_acc;
};

constexpr int EXT_WAVES = 4;

__global__ void __launch_bounds__(64 * EXT_WAVES) k_ext_propose(const ExtArgs a) {
  const int lane = threadIdx.x & 63;
  const long long c = (long long)blockIdx.x * EXT_WAVES + (threadIdx.x >> 6);
  if (c >= a.N || lane >= a.d) return;
  const double cur = a.theta[c * a.DP + lane];
  double prp = cur;
  if (a.mode != 1) {  // proposal.py:249-251 / :351-355
    const double scal = a.scaling[c];
    const double sx = scal * a.inc[((size_t)a.s * a.NP + c) * a.DP + lane];
    prp = a.prop_kind == 1 ? sqrt(1.0 - scal * scal) * cur + sx : cur + sx;
  }
  a.prop[c * a.d + lane] = prp;
}

__device__ __forceinline__ double ext_wave_sum(double v) {
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

__global__ void __launch_bounds__(64 * EXT_WAVES) k_ext_accept(const ExtArgs a) {
  const int lane = threadIdx.x & 63;
  const long long c = (long long)blockIdx.x * EXT_WAVES + (threadIdx.x >> 6);
  if (c >= a.N) return;  // whole waves leave together
  const bool lj = lane < a.d, eval = a.mode == 1;
  const double prp = lj ? a.prop[c * a.d + lane] : 0.0;
  double sse = 0.0;  // distributions.py:295-326
  const double* Fc = a.F + (size_t)c * a.m;
  for (int o = lane; o < a.m; o += 64) {
    const double r = Fc[o] - a.data[o];
    double sq = r * r;
    if (a.w) sq *= a.w[o];
    sse += sq;
  }
  sse = ext_wave_sum(sse);
  double pj = 0.0;
  if (lj) {
    const double dv = prp - a.pr_mean[lane];
    pj = dv * dv * a.pr_pinv[lane];
    if (a.pr_lo && (prp < a.pr_lo[lane] || prp > a.pr_hi[lane])) pj = __builtin_inf();  // uniform prior components
  }
  const double maha = ext_wave_sum(pj);
  const double ll_n = a.w ? -0.5 * sse : -0.5 * sse / a.var;
  const double lp_n = -0.5 * (a.logconst + maha);  // scipy MVN logpdf, posterior.py:92
  const double post_n = lp_n + ll_n;               // link.py:48
  double lp = a.lp[c], ll = a.ll[c];
  bool acc = true;
  if (!eval) {  // chain.py:112
    const double delta = a.prop_kind == 1 ? ll_n - ll : post_n - (lp + ll);
    double alpha = exp(delta);
    if (post_n != post_n) alpha = 0.0;
    acc = a.u[(size_t)a.s * a.NP + c] < alpha;
  }
  double cur = lj ? a.theta[c * a.DP + lane] : 0.0;
  if (acc) {
    lp = lp_n;
    ll = ll_n;
    cur = prp;
    if (lj) a.theta[c * a.DP + lane] = cur;
    if (lane == 0) {
      a.lp[c] = lp;
      a.ll[c] = ll;
    }
  }
  if (!eval) {
    const size_t r = (size_t)a.s * a.N + c;
    if (lane == 0) {
      if (acc && a.acc_count) a.acc_count[c] += 1;
      if (a.rec_stats) {
        a.rec_stats[r * 3 + 0] = lp;
        a.rec_stats[r * 3 + 1] = ll;
        a.rec_stats[r * 3 + 2] = lp + ll;
      }
      if (a.rec_acc) a.rec_acc[r] = acc ? 1 : 0;
    }
    if (a.rec_params && lj) a.rec_params[r * a.d + lane] = cur;
  }
}

}  // namespace tda
