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

#include "tda_philox.h"

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
  const double* Pd; // dense noise: Sigma^-1 [m][m] (DefaultGaussianLogLike, distributions.py:295-298) or null; needs m doubles of dynamic LDS per wave
  double var;
  const double* pr_mean;
  const double* pr_pinv;
  const double* pr_lo;
  const double* pr_hi;
  double logconst;
  double* rec_params;
  double* rec_stats;
  unsigned char* rec_acc;
  int* anyacc;  // multi-level: set when this (base-level) step accepted -- "the subchain moved" (chain.py:357-364); may be null
  unsigned char* ring;  // multi-level: the base proposal's `accepted` list as a ring [ring_P][NP] (scaling adaptation window,
  int ring_P;           // proposal.py:236; it also receives the alignment entries of the levels above); may be null
  long long ring_pos;   // absolute list position of this step's entry
  // IndependenceSampler (prop_kind 4): theta' = q_mean + inc, alpha = exp(post' - post + lq - qz) with log q up to its constant
  const double* q_mean;  // [DP]
  const double* qz;      // [S][NP]  -|z|^2 / 2 of the proposal
  double* lq;            // [NP]     the same for the current state
  // OperatorWeightedCrankNicolson (prop_kind 5): theta' = S theta + inc, likelihood-ratio acceptance
  const double* SopT;    // [DP][DP] transposed state operator
  int theta_ld;         // k_ext_propose, mode 1: row stride of `theta` (0 = DP); the promoted states of a randomised subchain live in ysnap
  // Delayed Acceptance with randomize_subchain_length (chain.py:525-527): the state after step `pick[c]` of the running
  // subchain is the one promoted to the fine level; k_ext_accept snapshots it (parameters, log-prior, log-likelihood)
  const int* pick;      // [NP] or null
  int cnt;              // position of this step inside the subchain
  double* ysnap;        // [NP][DP + 2]
};

constexpr int EXT_WAVES = 4;

__global__ void __launch_bounds__(64 * EXT_WAVES) k_ext_propose(const ExtArgs a) {
  const int lane0 = threadIdx.x & 63;
  const long long c = (long long)blockIdx.x * EXT_WAVES + (threadIdx.x >> 6);
  if (c >= a.N) return;
  // lane = parameter (and parameter + 64 at 65 .. 128 parameters: GaussianRandomWalk / CrankNicolson / AdaptiveMetropolis only, whose
  // proposals are elementwise -- tda_engine_init refuses the other kinds there)
  for (int lane = lane0; lane < a.d; lane += 64) {
    const double cur = a.theta[c * (a.theta_ld ? a.theta_ld : a.DP) + lane];
    double prp = cur;
    if (a.mode != 1) {  // proposal.py:249-251 / :351-355 / :113-115 / :592-598
      const double scal = a.scaling[c];
      const double inc = a.inc[((size_t)a.s * a.NP + c) * a.DP + lane];
      const double sx = scal * inc;
      if (a.prop_kind == 4) {
        prp = a.q_mean[lane] + inc;
      } else if (a.prop_kind == 5) {  // (S theta)[lane] = sum_j S^T[j][lane] theta[j] (d <= 64: every source j is a lane of this wave)
        double st = 0.0;
        for (int j = 0; j < a.d; ++j) st = fma(a.SopT[(size_t)j * a.DP + lane], __shfl(cur, j), st);
        prp = st + inc;
      } else {
        prp = a.prop_kind == 1 ? sqrt(1.0 - scal * scal) * cur + sx : cur + sx;
      }
    }
    a.prop[c * a.d + lane] = prp;
  }
}

__device__ __forceinline__ double ext_wave_sum(double v) {
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// sum of squared residuals of one chain's model outputs, weighted as the likelihood says: isotropic / diagonal, or the dense
// quadratic form r^T Sigma^-1 r (residual staged in this wave's LDS slice, lane = column of Sigma^-1, rows streamed)
__device__ __forceinline__ double ext_weighted_sse(const double* __restrict__ Fc, const double* __restrict__ data,
                                                   const double* __restrict__ w, const double* __restrict__ Pd, int m, int lane,
                                                   double* __restrict__ sr) {
  double sse = 0.0;
  if (Pd) {
    for (int o = lane; o < m; o += 64) sr[o] = Fc[o] - data[o];
    __builtin_amdgcn_wave_barrier();
    for (int j = lane; j < m; j += 64) {
      double t = 0.0;
      for (int o = 0; o < m; ++o) t = fma(Pd[(size_t)o * m + j], sr[o], t);
      sse += sr[j] * t;
    }
  } else {
    for (int o = lane; o < m; o += 64) {
      const double r = Fc[o] - data[o];
      double sq = r * r;
      if (w) sq *= w[o];
      sse += sq;
    }
  }
  return ext_wave_sum(sse);
}

__global__ void __launch_bounds__(64 * EXT_WAVES) k_ext_accept(const ExtArgs a) {
  const int lane = threadIdx.x & 63;
  const long long c = (long long)blockIdx.x * EXT_WAVES + (threadIdx.x >> 6);
  if (c >= a.N) return;  // whole waves leave together
  const bool lj = lane < a.d, lj2 = lane + 64 < a.d, eval = a.mode == 1;  // (lj2: 65 .. 128 parameters, a second parameter per lane)
  const double prp = lj ? a.prop[c * a.d + lane] : 0.0, prp2 = lj2 ? a.prop[c * a.d + lane + 64] : 0.0;
  extern __shared__ double ext_dyn_lds[];  // dense noise only: [EXT_WAVES][m]
  const double sse = ext_weighted_sse(a.F + (size_t)c * a.m, a.data, a.w, a.Pd, a.m, lane, ext_dyn_lds + (size_t)(threadIdx.x >> 6) * a.m);  // distributions.py:295-326
  double pj = 0.0;
  if (lj) {
    const double dv = prp - a.pr_mean[lane];
    pj = dv * dv * a.pr_pinv[lane];
    if (a.pr_lo && (prp < a.pr_lo[lane] || prp > a.pr_hi[lane])) pj = __builtin_inf();  // uniform prior components
  }
  if (lj2) {
    const double dv = prp2 - a.pr_mean[lane + 64];
    pj += dv * dv * a.pr_pinv[lane + 64];
    if (a.pr_lo && (prp2 < a.pr_lo[lane + 64] || prp2 > a.pr_hi[lane + 64])) pj = __builtin_inf();
  }
  const double maha = ext_wave_sum(pj);
  const double ll_n = (a.w || a.Pd) ? -0.5 * sse : -0.5 * sse / a.var;
  const double lp_n = -0.5 * (a.logconst + maha);  // scipy MVN logpdf, posterior.py:92
  const double post_n = lp_n + ll_n;               // link.py:48
  double lp = a.lp[c], ll = a.ll[c];
  bool acc = true;
  const double qzs = (!eval && a.prop_kind == 4) ? a.qz[(size_t)a.s * a.NP + c] : 0.0;
  if (!eval) {  // chain.py:112; proposal.py:253-258 / :357-362 / :117-123
    double delta = (a.prop_kind == 1 || a.prop_kind == 5) ? ll_n - ll : post_n - (lp + ll);
    if (a.prop_kind == 4) delta = (delta + a.lq[c]) - qzs;
    double alpha = exp(delta);
    if (post_n != post_n) alpha = 0.0;
    acc = a.u[(size_t)a.s * a.NP + c] < alpha;
    if (acc && a.prop_kind == 4 && lane == 0) a.lq[c] = qzs;
  }
  double cur = lj ? a.theta[c * a.DP + lane] : 0.0, cur2 = lj2 ? a.theta[c * a.DP + lane + 64] : 0.0;
  if (acc) {
    lp = lp_n;
    ll = ll_n;
    cur = prp;
    cur2 = prp2;
    if (lj) a.theta[c * a.DP + lane] = cur;
    if (lj2) a.theta[c * a.DP + lane + 64] = cur2;
    if (lane == 0) {
      a.lp[c] = lp;
      a.ll[c] = ll;
    }
  }
  if (!eval) {
    const size_t r = (size_t)a.s * a.N + c;
    if (lane == 0) {
      if (acc && a.acc_count) a.acc_count[c] += 1;
      if (acc && a.anyacc) a.anyacc[c] = 1;
      if (a.ring) a.ring[(size_t)(a.ring_pos % a.ring_P) * a.NP + c] = acc ? 1 : 0;
      if (a.rec_stats) {
        a.rec_stats[r * 3 + 0] = lp;
        a.rec_stats[r * 3 + 1] = ll;
        a.rec_stats[r * 3 + 2] = lp + ll;
      }
      if (a.rec_acc) a.rec_acc[r] = acc ? 1 : 0;
    }
    if (a.rec_params && lj) a.rec_params[r * a.d + lane] = cur;
    if (a.rec_params && lj2) a.rec_params[r * a.d + lane + 64] = cur2;
    if (a.pick && a.cnt == a.pick[c]) {  // the promoted state of this subchain
      double* ys = a.ysnap + (size_t)c * (a.DP + 2);
      if (lane < a.DP) ys[lane] = lj ? cur : 0.0;
      if (lane + 64 < a.DP) ys[lane + 64] = lj2 ? cur2 : 0.0;
      if (lane == 0) {
        ys[a.DP] = lp;
        ys[a.DP + 1] = ll;
      }
    }
  }
}

// promoted index of the subchain that starts now (chain.py:525-527): replayed reference index in [-L, -1] or Philox
__global__ void k_ext_pick(long long N, int L, unsigned long long seed, long long chain_offset, long long step,
                           const double* __restrict__ ridx_rep, int* __restrict__ pick) {
  const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  int p;
  if (ridx_rep) {
    const double r = ridx_rep[c];
    p = (r != r) ? L - 1 : (int)r + L;
  } else {
    const u32x4 x = philox4x32_10(u32x4{0u, (uint32_t)step, (uint32_t)(chain_offset + c), 3u /* STREAM_INDEX */}, (uint32_t)seed,
                                  (uint32_t)(seed >> 32));
    p = (int)(((uint64_t)x.x * (uint64_t)L) >> 32);
  }
  pick[c] = p;
}

// DREAM(Z) over a callback / source-defined model (proposal.py:663-852): the draws of a block come from k_dreamz_draw as
// usual; per step the jump is applied here (gathering the archive rows unless the draw kernel already did), the model is
// evaluated outside, k_ext_accept decides, and k_dz_ext_append grows the archive.  One wave per chain.
struct DzExtArgs {
  long long N, NP;
  int d, DP, delta, s, shared, jump_ready;
  long long M_base, cap;
  double* arch;           // per chain [NP][cap][DP] / shared [cap][DP]
  const double* theta;    // [NP][DP]
  const double* coef;     // [S][NP][DP]
  const double* epsm;
  const int* ridx;        // [S][NP][2 * 4]
  double* prop;           // [N][d]
  double* blk_states;     // [S][NP][DP] (shared archive: the block's states) or null
};

__global__ void __launch_bounds__(64 * EXT_WAVES) k_dz_ext_propose(const DzExtArgs a) {
  const int lane = threadIdx.x & 63;
  const long long c = (long long)blockIdx.x * EXT_WAVES + (threadIdx.x >> 6);
  if (c >= a.N || lane >= a.d) return;
  const size_t o = ((size_t)a.s * a.NP + c) * a.DP + lane;
  double jump;
  if (a.jump_ready) {
    jump = a.coef[o];
  } else {  // proposal.py:823-826, :850-852
    const double* arch_c = a.shared ? a.arch : a.arch + (size_t)c * a.cap * a.DP;
    double z1 = 0.0, z2 = 0.0;
    for (int i = 0; i < a.delta; ++i) {
      const int r1 = a.ridx[((size_t)a.s * a.NP + c) * 8 + 2 * i + 0];
      const int r2 = a.ridx[((size_t)a.s * a.NP + c) * 8 + 2 * i + 1];
      z1 += arch_c[(size_t)r1 * a.DP + lane];
      z2 += arch_c[(size_t)r2 * a.DP + lane];
    }
    jump = a.coef[o] * (z1 - z2) + a.epsm[o];
  }
  a.prop[c * a.d + lane] = a.theta[c * a.DP + lane] + jump;
}

__global__ void __launch_bounds__(64 * EXT_WAVES) k_dz_ext_append(const DzExtArgs a) {  // proposal.py:794-795
  const int lane = threadIdx.x & 63;
  const long long c = (long long)blockIdx.x * EXT_WAVES + (threadIdx.x >> 6);
  if (c >= a.NP || lane >= a.DP) return;
  const double cur = a.theta[c * a.DP + lane];
  if (!a.shared && c < a.N) a.arch[((size_t)c * a.cap + a.M_base + a.s) * a.DP + lane] = cur;
  if (a.blk_states) a.blk_states[((size_t)a.s * a.NP + c) * a.DP + lane] = cur;
}

// accept uniforms of one level-q step for all chains (accept_uniform of the RNG contract), for level kernels that are
// compiled with a user's model source and know nothing of the engine's Philox
__global__ void k_ext_level_uniforms(long long N, unsigned long long seed, long long chain_offset, long long step, int level,
                                     const double* __restrict__ u_rep, double* __restrict__ u) {
  const long long c = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N) return;
  u[c] = u_rep ? u_rep[c] : accept_uniform(seed, (uint32_t)(chain_offset + c), (uint32_t)step, (uint32_t)level);
}

// A linear level inside a host-sequenced hierarchy (e.g. a linear surrogate below a non-linear model): F = A prop for all
// chains (F = A prop + b), one wave per chain, the lanes stride over the outputs
__global__ void __launch_bounds__(64 * EXT_WAVES) k_ext_linear_eval(long long N, int d, int m, const double* __restrict__ A,
                                                                    const double* __restrict__ bvec,
                                                                    const double* __restrict__ prop, double* __restrict__ F) {
  __shared__ double s_th[EXT_WAVES][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long c = (long long)blockIdx.x * EXT_WAVES + wv;
  if (c >= N) return;  // whole waves leave together
  s_th[wv][lane] = lane < d ? prop[c * d + lane] : 0.0;
  __builtin_amdgcn_wave_barrier();
  for (int o = lane; o < m; o += 64) {
    const double* __restrict__ Ao = A + (size_t)o * d;
    double f = 0.0;
    for (int j = 0; j < d; ++j) f = fma(Ao[j], s_th[wv][j], f);
    F[c * m + o] = f + bvec[o];
  }
}

// ------------------------------------------------------------------------------------------------
// Delayed Acceptance / MLDA with callback models: one step of level q >= 1 for every chain once the subchain of level
// q - 1 has finished (DAChain.sample, chain.py:353-402; MLDA.make_mlda_proposal, proposal.py:1515-1545; MLDAChain.sample,
// chain.py:711-737).  The host has evaluated level q's model at the states of level q - 1 (F); this kernel does what
// the cascade of k_ml_steps does for linear levels: two-stage acceptance with the densities kept from the subchain
// start, the skip rule (nothing accepted below -> rejection), alignment of the levels below, records.
// ------------------------------------------------------------------------------------------------
struct ExtLevelArgs {
  long long N, NP, chain_offset;
  int d, DP, m, nlev, q;
  unsigned long long seed;
  long long step;       // index of this level-q step (RNG / replay row)
  const double* F;      // [N][m] level-q model at theta_{q-1}
  const double* data;   // [m]
  const double* w;      // 1 / diag(noise) or null
  const double* Pd;     // dense Sigma^-1 [m][m] or null (m doubles of dynamic LDS per wave)
  double var;
  double* theta;        // [nlev][NP][DP]
  double* lp;           // [nlev][NP]
  double* ll;
  double* Sst;          // [npairs][2][NP], pair (j, q) at q (q - 1) / 2 + j
  int* anyacc;          // [nlev][NP]
  const double* u_rep;  // [N] replay uniforms of this step (NaN = none drawn) or null
  double* rec_params;   // row of this step (may be null)
  double* rec_stats;
  unsigned char* rec_acc;
  unsigned char* ring;  // alignment entry in the base proposal's accepted window (chain.py:363,389,397; proposal.py:1486); may be null
  int ring_P;
  long long ring_pos;
  const double* ysnap;  // randomised subchain (DA): the promoted state [NP][DP + 2] instead of level q - 1's last state; or null
};

__global__ void __launch_bounds__(64 * EXT_WAVES) k_ext_level_action(const ExtLevelArgs a) {
  const int lane = threadIdx.x & 63;
  const long long c = (long long)blockIdx.x * EXT_WAVES + (threadIdx.x >> 6);
  if (c >= a.N) return;  // whole waves leave together
  const int q = a.q, k = a.q - 1;
  const bool lj = lane < a.d;
  auto TH = [&](int lev) { return a.theta + ((size_t)lev * a.NP + c) * a.DP; };
  auto PI = [](int j, int qq) { return qq * (qq - 1) / 2 + j; };
  extern __shared__ double ext_dyn_lds[];  // dense noise only: [EXT_WAVES][m]
  const double sse = ext_weighted_sse(a.F + (size_t)c * a.m, a.data, a.w, a.Pd, a.m, lane, ext_dyn_lds + (size_t)(threadIdx.x >> 6) * a.m);
  const double lln = (a.w || a.Pd) ? -0.5 * sse : -0.5 * sse / a.var;
  const double* ys = a.ysnap ? a.ysnap + (size_t)c * (a.DP + 2) : nullptr;
  const bool lj2 = lane + 64 < a.d;  // (65 .. 128 parameters: a second parameter per lane)
  const double yj = lj ? (ys ? ys[lane] : TH(k)[lane]) : 0.0, xj = lj ? TH(q)[lane] : 0.0;
  const double yj2 = lj2 ? (ys ? ys[lane + 64] : TH(k)[lane + 64]) : 0.0, xj2 = lj2 ? TH(q)[lane + 64] : 0.0;
  const double y_lp = ys ? ys[a.DP] : a.lp[(size_t)k * a.NP + c], y_ll = ys ? ys[a.DP + 1] : a.ll[(size_t)k * a.NP + c];
  const double x_lp = a.lp[(size_t)q * a.NP + c], x_ll = a.ll[(size_t)q * a.NP + c];
  const int pkq = PI(k, q);
  const double st_lp = a.Sst[((size_t)pkq * 2 + 0) * a.NP + c], st_ll = a.Sst[((size_t)pkq * 2 + 1) * a.NP + c];
  const bool any = a.anyacc[(size_t)k * a.NP + c] != 0;
  const double lpn = y_lp;  // same prior, same parameters (posterior.py:92)
  const double alpha = exp(((lpn + lln) - (x_lp + x_ll)) + (st_lp + st_ll) - (y_lp + y_ll));  // chain.py:475-483, proposal.py:1615-1624
  double u;
  if (a.u_rep) u = a.u_rep[c];
  else u = accept_uniform(a.seed, (uint32_t)(a.chain_offset + c), (uint32_t)a.step, (uint32_t)q);
  const bool acc = any && (u < alpha);
  // alignment (chain.py:357-398; proposal.py:1469-1493): accept -> level q takes y; reject -> the levels below return to theta_q
  if (acc) {  // level q takes y -- and level q - 1 too, if y is a promoted intermediate state
    if (lane < a.DP) TH(q)[lane] = lj ? yj : 0.0;
    if (lane + 64 < a.DP) TH(q)[lane + 64] = lj2 ? yj2 : 0.0;
    if (ys && lane < a.DP) TH(k)[lane] = lj ? yj : 0.0;
    if (ys && lane + 64 < a.DP) TH(k)[lane + 64] = lj2 ? yj2 : 0.0;
  } else {
    for (int j = 0; j < q; ++j) {
      if (lane < a.DP) TH(j)[lane] = lj ? xj : 0.0;
      if (lane + 64 < a.DP) TH(j)[lane + 64] = lj2 ? xj2 : 0.0;
    }
  }
  if (lane == 0) {
    if (acc) {
      a.lp[(size_t)q * a.NP + c] = lpn;
      a.ll[(size_t)q * a.NP + c] = lln;
      a.lp[(size_t)k * a.NP + c] = y_lp;
      a.ll[(size_t)k * a.NP + c] = y_ll;
    } else {
      for (int j = 0; j < q; ++j) {
        const int p = PI(j, q);
        a.lp[(size_t)j * a.NP + c] = a.Sst[((size_t)p * 2 + 0) * a.NP + c];
        a.ll[(size_t)j * a.NP + c] = a.Sst[((size_t)p * 2 + 1) * a.NP + c];
      }
    }
    for (int j = 0; j < q; ++j)
      for (int q2 = j + 1; q2 <= q; ++q2) {
        const int p = PI(j, q2);
        a.Sst[((size_t)p * 2 + 0) * a.NP + c] = a.lp[(size_t)j * a.NP + c];
        a.Sst[((size_t)p * 2 + 1) * a.NP + c] = a.ll[(size_t)j * a.NP + c];
      }
    a.anyacc[(size_t)k * a.NP + c] = 0;
    if (q < a.nlev - 1 && acc) a.anyacc[(size_t)q * a.NP + c] = 1;
    if (a.rec_stats) {
      const double l1 = a.lp[(size_t)q * a.NP + c], l2 = a.ll[(size_t)q * a.NP + c];
      a.rec_stats[c * 3 + 0] = l1;
      a.rec_stats[c * 3 + 1] = l2;
      a.rec_stats[c * 3 + 2] = l1 + l2;
    }
    if (a.rec_acc) a.rec_acc[c] = acc ? 1 : 0;
    if (a.ring) a.ring[(size_t)(a.ring_pos % a.ring_P) * a.NP + c] = acc ? 1 : 0;
  }
  if (a.rec_params && lj) a.rec_params[c * a.d + lane] = acc ? yj : xj;
  if (a.rec_params && lj2) a.rec_params[c * a.d + lane + 64] = acc ? yj2 : xj2;
}

}  // namespace tda
