// Diagonal state-independent adaptive error model (extension; SURVEY.md §7 / §8(f)2: "AEM as written cannot scale").
//
// tinyDA's error model (Cui et al. 2019; chain.py:268-305, 485-499, 739-765; proposal.py:1404-1467, 1547-1578) keeps, per
// chain and per pair of adjacent levels, the running mean and the full m x m covariance of the model difference, and re-inverts
// Sigma_e + Sigma_bias after every step of the finer level (distributions.py:385-402): m^2 doubles per tracker and chain and
// m^3 flops per level step -- 512 MiB per matrix and 74 % of the run time at 4096 chains and m = 128, impossible at m = 2048.
// This variant keeps only the DIAGONAL of every tracker (element by element the same recursion, utils.py:113-122, so each
// variance is bit for bit the diagonal entry of the reference's matrix) and a diagonal Sigma_e: the corrected likelihood is
//     -1/2 sum_o (F_o + b_o - y_o)^2 / (sigma_o^2 + s_o^2),
// 3 m doubles per tracker and chain, O(m) per level step, no inversion kernel, any m.  Everything else is the reference's
// protocol: biases stack upwards (level k is corrected by the trackers of all levels above it), Delayed Acceptance feeds the
// current pair every fine step, MLDA refreshes the difference on acceptance only, set_bias keeps the previous inverse
// while every entry of the bias covariance is below 1e-9, the latest link one level down is re-evaluated (update_link).
//
// The hierarchy is sequenced by the host (model outputs of a level for all chains in F[N][m]: linear levels through
// k_ext_linear_eval, callback and source-defined levels through their own paths); these kernels do the rest of a step.
// One wave per chain, the lanes stride over the m outputs.
#pragma once
#include <hip/hip_runtime.h>

#include "tda_kernels_ext.h"
#include "tda_kernels_ml.h"

namespace tda {

struct AemdArgs {
  long long N, NP, chain_offset;
  int d, DP, m, nlev, q, is_da, s, prop_kind;
  unsigned long long seed;
  long long step;              // local step index of level q (accept uniform of the RNG contract)
  const double* F;             // [N][m] fresh model outputs: level 0 at the proposals (accept) / level q at theta_{q-1} (action)
  const double* prop;          // [N][d] base-level proposals (accept)
  const double* u0;            // [S][NP] base-level uniforms (accept)
  const double* data[AEM_MAXLEV];  // [m]
  const double* sig2[AEM_MAXLEV];  // [m] diagonal of Sigma_e (adaptive levels)
  const double* wfin;          // [m] 1 / diagonal of the finest level's noise, or null (isotropic: var_finest)
  double var_finest;
  double* Fcur[AEM_MAXLEV];        // [N][m] model output of every level's current link
  double* Fst;                 // [npairs][N][m] output of level j at theta_q (companion of Sst)
  double* theta;               // [nlev][NP][DP]
  double* lp;                  // [nlev][NP]
  double* ll;
  double* Sst;                 // [npairs][2][NP]
  int* anyacc;                 // [nlev][NP]
  long long* sid;              // [nlev][NP]
  long long sid_value;
  double* bias[AEM_MAXLEV];        // [N][m] total bias of adaptive level k
  double* w[AEM_MAXLEV];           // [N][m] 1 / (sigma^2 + total bias variance) in use at level k
  double* mu[AEM_MAXLEV];          // trackers of levels >= 1: running mean, diagonal variance, last difference fed
  double* var[AEM_MAXLEV];
  double* md[AEM_MAXLEV];
  long long b_t;               // recursion counter of level q's tracker before this update
  const double* pr_mean;
  const double* pr_pinv;
  const double* pr_lo;
  const double* pr_hi;
  double logconst;
  const double* u_rep;         // [N] replay uniform of this level step or null
  unsigned char* ring;
  int ring_P;
  long long ring_pos;
  double* rec_params;
  double* rec_stats;
  unsigned char* rec_acc;
};

__device__ __forceinline__ bool aemd_wave_all(bool v) { return __ballot(v) == ~0ull; }

// trackers, biases, weights and corrected likelihoods at theta0 (chain.py:268-305, :643-678; proposal.py:1404-1467): every
// level's current output is F_k(theta0); mu_q = last difference = F_q - F_{q-1}, variance 0; the inverse stays Sigma_e^-1
// (every bias variance is < 1e-9); the initial links of the adaptive levels are re-evaluated under the stacked biases
__global__ void __launch_bounds__(64 * EXT_WAVES) k_aemd_init(const AemdArgs a) {
  const int lane = threadIdx.x & 63;
  const long long c = (long long)blockIdx.x * EXT_WAVES + (threadIdx.x >> 6);
  if (c >= a.N) return;
  const int nl = a.nlev, m = a.m;
  for (int q = 1; q < nl; ++q)
    for (int o = lane; o < m; o += 64) {
      const double df = a.Fcur[q][c * m + o] - a.Fcur[q - 1][c * m + o];
      a.mu[q][c * m + o] = df;
      a.md[q][c * m + o] = df;
      a.var[q][c * m + o] = 0.0;
    }
  for (int k = 0; k < nl - 1; ++k) {
    double sq = 0.0;
    for (int o = lane; o < m; o += 64) {
      double bt = 0.0;
      for (int p = k + 1; p < nl; ++p) bt += a.Fcur[p][c * m + o] - a.Fcur[p - 1][c * m + o];
      const double wv = 1.0 / a.sig2[k][o];
      a.bias[k][c * m + o] = bt;
      a.w[k][c * m + o] = wv;
      const double r = (a.Fcur[k][c * m + o] + bt) - a.data[k][o];
      sq += wv * (r * r);
    }
    const double llk = -0.5 * ext_wave_sum(sq);
    if (lane == 0) {
      a.ll[(size_t)k * a.NP + c] = llk;
      for (int q2 = k + 1; q2 < nl; ++q2) a.Sst[((size_t)pair_index(k, q2) * 2 + 1) * a.NP + c] = llk;
    }
  }
  for (int q2 = 1; q2 < nl; ++q2)
    for (int j = 0; j < q2; ++j)
      for (int o = lane; o < m; o += 64) a.Fst[((size_t)pair_index(j, q2) * a.N + c) * m + o] = a.Fcur[j][c * m + o];
}

// base-level step under the corrected diagonal likelihood (the twin of k_ext_aem_accept)
__global__ void __launch_bounds__(64 * EXT_WAVES) k_aemd_accept(const AemdArgs a) {
  const int lane = threadIdx.x & 63;
  const long long c = (long long)blockIdx.x * EXT_WAVES + (threadIdx.x >> 6);
  if (c >= a.N) return;
  const int m = a.m;
  const bool lj = lane < a.d;
  double sq = 0.0;
  for (int o = lane; o < m; o += 64) {
    const double r = (a.F[c * m + o] + a.bias[0][c * m + o]) - a.data[0][o];
    sq += a.w[0][c * m + o] * (r * r);
  }
  const double ll_n = -0.5 * ext_wave_sum(sq);
  const double prp = lj ? a.prop[c * a.d + lane] : 0.0;
  double pj = 0.0;
  if (lj) {
    const double dv = prp - a.pr_mean[lane];
    pj = dv * dv * a.pr_pinv[lane];
    if (a.pr_lo && (prp < a.pr_lo[lane] || prp > a.pr_hi[lane])) pj = __builtin_inf();
  }
  const double lp_n = -0.5 * (a.logconst + ext_wave_sum(pj));
  const double post_n = lp_n + ll_n;
  double lp = a.lp[c], ll = a.ll[c];
  const double delta = a.prop_kind == 1 ? ll_n - ll : post_n - (lp + ll);
  double alpha = exp(delta);
  if (post_n != post_n) alpha = 0.0;
  const bool acc = a.u0[(size_t)a.s * a.NP + c] < alpha;
  double cur = lj ? a.theta[c * a.DP + lane] : 0.0;
  if (acc) {
    lp = lp_n;
    ll = ll_n;
    cur = prp;
    if (lj) a.theta[c * a.DP + lane] = cur;
    for (int o = lane; o < m; o += 64) a.Fcur[0][c * m + o] = a.F[c * m + o];
    if (lane == 0) {
      a.lp[c] = lp;
      a.ll[c] = ll;
      a.anyacc[c] = 1;
      a.sid[c] = a.sid_value;
    }
  }
  const size_t rr = (size_t)a.s * a.N + c;
  if (lane == 0) {
    if (a.ring) a.ring[(size_t)(a.ring_pos % a.ring_P) * a.NP + c] = acc ? 1 : 0;
    if (a.rec_stats) {
      a.rec_stats[rr * 3 + 0] = lp;
      a.rec_stats[rr * 3 + 1] = ll;
      a.rec_stats[rr * 3 + 2] = lp + ll;
    }
    if (a.rec_acc) a.rec_acc[rr] = acc ? 1 : 0;
  }
  if (a.rec_params && lj) a.rec_params[rr * a.d + lane] = cur;
}

// one step of level q >= 1 for every chain: decision (chain.py:475-483, proposal.py:1615-1624), alignment, output book-keeping,
// tracker update, stacked bias and inverse variances of level q - 1, update_link of its latest link -- one launch, no inversion
__global__ void __launch_bounds__(64 * EXT_WAVES) k_aemd_action(const AemdArgs a) {
  const int lane = threadIdx.x & 63;
  const long long c = (long long)blockIdx.x * EXT_WAVES + (threadIdx.x >> 6);
  if (c >= a.N) return;
  const int q = a.q, k = a.q - 1, nl = a.nlev, m = a.m, d = a.d;
  const bool lj = lane < d;
  auto TH = [&](int lev) { return a.theta + ((size_t)lev * a.NP + c) * a.DP; };
  auto FS = [&](int j, int qq) { return a.Fst + ((size_t)pair_index(j, qq) * a.N + c) * m; };

  // ---- log-likelihood of level q at y = theta_k (its fresh outputs are in F) ----
  double sq = 0.0;
  if (q == nl - 1) {
    for (int o = lane; o < m; o += 64) {
      const double r = a.F[c * m + o] - a.data[q][o];
      sq += a.wfin ? a.wfin[o] * (r * r) : r * r;
    }
  } else {
    for (int o = lane; o < m; o += 64) {
      const double r = (a.F[c * m + o] + a.bias[q][c * m + o]) - a.data[q][o];
      sq += a.w[q][c * m + o] * (r * r);
    }
  }
  sq = ext_wave_sum(sq);
  const double lln = (q == nl - 1 && !a.wfin) ? -0.5 * sq / a.var_finest : -0.5 * sq;
  const double yj = lj ? TH(k)[lane] : 0.0, xj = lj ? TH(q)[lane] : 0.0;
  const double y_lp = a.lp[(size_t)k * a.NP + c], y_ll = a.ll[(size_t)k * a.NP + c];
  const double x_lp = a.lp[(size_t)q * a.NP + c], x_ll = a.ll[(size_t)q * a.NP + c];
  const int pkq = pair_index(k, q);
  const double st_lp = a.Sst[((size_t)pkq * 2 + 0) * a.NP + c], st_ll = a.Sst[((size_t)pkq * 2 + 1) * a.NP + c];
  const bool any = a.anyacc[(size_t)k * a.NP + c] != 0;
  const double lpn = y_lp;  // same prior, same parameters (posterior.py:92)
  const double alpha = exp(((lpn + lln) - (x_lp + x_ll)) + (st_lp + st_ll) - (y_lp + y_ll));
  double u;
  if (a.u_rep) u = a.u_rep[c];
  else u = accept_uniform(a.seed, (uint32_t)(a.chain_offset + c), (uint32_t)a.step, (uint32_t)q);
  const bool acc = any && (u < alpha);

  // ---- alignment (chain.py:357-398; proposal.py:1469-1493) ----
  if (acc) {
    if (lane < a.DP) TH(q)[lane] = lj ? yj : 0.0;
  } else {
    for (int j = 0; j < q; ++j)
      if (lane < a.DP) TH(j)[lane] = lj ? xj : 0.0;
  }
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) {
    if (acc) {
      a.lp[(size_t)q * a.NP + c] = lpn;
      a.ll[(size_t)q * a.NP + c] = lln;
      a.sid[(size_t)q * a.NP + c] = a.sid[(size_t)k * a.NP + c];
    } else {
      for (int j = 0; j < q; ++j) {
        const int p = pair_index(j, q);
        a.lp[(size_t)j * a.NP + c] = a.Sst[((size_t)p * 2 + 0) * a.NP + c];
        a.ll[(size_t)j * a.NP + c] = a.Sst[((size_t)p * 2 + 1) * a.NP + c];
        a.sid[(size_t)j * a.NP + c] = a.sid[(size_t)q * a.NP + c];
      }
    }
    for (int j = 0; j < q; ++j)
      for (int q2 = j + 1; q2 <= q; ++q2) {
        const int p = pair_index(j, q2);
        a.Sst[((size_t)p * 2 + 0) * a.NP + c] = a.lp[(size_t)j * a.NP + c];
        a.Sst[((size_t)p * 2 + 1) * a.NP + c] = a.ll[(size_t)j * a.NP + c];
      }
    a.anyacc[(size_t)k * a.NP + c] = 0;
    if (q < nl - 1) a.anyacc[(size_t)q * a.NP + c] |= acc ? 1 : 0;
    if (a.ring) a.ring[(size_t)(a.ring_pos % a.ring_P) * a.NP + c] = acc ? 1 : 0;
    if (a.rec_stats) {
      const double l1 = a.lp[(size_t)q * a.NP + c], l2 = a.ll[(size_t)q * a.NP + c];
      a.rec_stats[c * 3 + 0] = l1;
      a.rec_stats[c * 3 + 1] = l2;
      a.rec_stats[c * 3 + 2] = l1 + l2;
    }
    if (a.rec_acc) a.rec_acc[c] = acc ? 1 : 0;
  }
  if (a.rec_params && lj) a.rec_params[c * d + lane] = acc ? yj : xj;

  // ---- model outputs of the aligned links, tracker of the pair (k, q), stacked bias / inverse variances of level k,
  //      update_link of level k's latest link (posterior.py:112-134) ----
  // Two outputs per lane and pass, every load of a pass before its first store: the arrays may alias as far as the compiler can
  // tell, so a load written after a store waits for it -- in the one-output-at-a-time form a pass was a chain of ~15 dependent
  // trips to memory (25 us per action at m = 128, 156 us at m = 1024).
  const double t = (double)a.b_t;
  const double ca = (t - 1.0) / t, cb = 1.0 / t;
  bool small = true;  // every entry of the total bias variance below 1e-9: set_bias keeps the inverse (distributions.py:399-402)
  for (int o0 = lane; o0 < m; o0 += 128) {
    double fnew[2], fq_old[2], fj_old[AEM_MAXLEV - 1][2], fs[AEM_MAXLEV - 1][2], md_old[2], mu_p[AEM_MAXLEV][2], var_p[AEM_MAXLEV][2];
    bool in[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int o = o0 + 64 * u;
      in[u] = o < m;
      const size_t i = (size_t)c * m + (in[u] ? o : o0);
      fnew[u] = a.F[i];
      fq_old[u] = a.Fcur[q][i];
      md_old[u] = a.md[q][i];
#pragma unroll
      for (int j = 0; j < AEM_MAXLEV - 1; ++j)  // (constant trip counts + predicates: the per-level values stay in registers)
        if (j < q) {
          fj_old[j][u] = a.Fcur[j][i];
          fs[j][u] = FS(j, q)[in[u] ? o : o0];
        } else {
          fj_old[j][u] = fs[j][u] = 0.0;
        }
#pragma unroll
      for (int p = 1; p < AEM_MAXLEV; ++p)
        if (p >= q && p < nl) {
          mu_p[p][u] = a.mu[p][i];
          var_p[p][u] = a.var[p][i];
        } else {
          mu_p[p][u] = var_p[p][u] = 0.0;
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (!in[u]) continue;
      const int o = o0 + 64 * u;
      const size_t i = (size_t)c * m + o;
      const double fq_cur = acc ? fnew[u] : fq_old[u];
      double fk_cur = 0.0;  // acc ? Fcur[k] : FS(k, q): the aligned output of level k, picked up in the loop below
      if (acc) a.Fcur[q][i] = fnew[u];
#pragma unroll
      for (int j = 0; j < AEM_MAXLEV - 1; ++j)
        if (j < q) {
          const double fj = acc ? fj_old[j][u] : fs[j][u];  // the output of level j's current link after the alignment
          if (j == k) fk_cur = fj;
          if (!acc) a.Fcur[j][i] = fj;
          for (int q2 = j + 1; q2 <= q; ++q2) FS(j, q2)[o] = fj;
        }
      // RecursiveSampleMoments.update restricted to the diagonal (utils.py:113-122 with sd = 1, eps = 0)
      const double diff_new = fq_cur - fk_cur;
      const double dm = (a.is_da || acc) ? diff_new : md_old[u];  // MLDA refreshes the difference on accept only
      a.md[q][i] = dm;
      double mu_o = 0.0, var_o = 0.0;
#pragma unroll
      for (int p = 1; p < AEM_MAXLEV; ++p)
        if (p == q) {
          mu_o = mu_p[p][u];
          var_o = var_p[p][u];
        }
      const double mu_n = (1.0 / (t + 1.0)) * (t * mu_o + dm);
      const double M = (t * (mu_o * mu_o) - (t + 1.0) * (mu_n * mu_n)) + dm * dm;
      const double var_n = ca * var_o + cb * M;
      a.var[q][i] = var_n;
      a.mu[q][i] = mu_n;
      double bt = 0.0, s2 = 0.0;
#pragma unroll
      for (int p = 1; p < AEM_MAXLEV; ++p)  // ascending p, as the reference sums the trackers (proposal.py:1563-1569)
        if (p >= q && p < nl) {
          bt += p == q ? mu_n : mu_p[p][u];
          s2 += p == q ? var_n : var_p[p][u];
        }
      a.bias[k][i] = bt;
      small = small && (s2 < 1e-9);
    }
  }
  const bool keep = aemd_wave_all(small);
  double sk = 0.0;
  for (int o0 = lane; o0 < m; o0 += 128) {
    double wv[2], s2[2], fk[2], bk[2], dk[2], sg[2];
    bool in[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int o = o0 + 64 * u;
      in[u] = o < m;
      const int oo = in[u] ? o : o0;
      const size_t i = (size_t)c * m + oo;
      wv[u] = a.w[k][i];
      s2[u] = 0.0;
      if (!keep)
        for (int p = q; p < nl; ++p) s2[u] += a.var[p][i];  // (loads only: no per-level array needed)
      sg[u] = a.sig2[k][oo];
      fk[u] = a.Fcur[k][i];
      bk[u] = a.bias[k][i];
      dk[u] = a.data[k][oo];
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (!in[u]) continue;
      const size_t i = (size_t)c * m + o0 + 64 * u;
      double w_ = wv[u];
      if (!keep) {
        w_ = 1.0 / (sg[u] + s2[u]);
        a.w[k][i] = w_;
      }
      const double r = (fk[u] + bk[u]) - dk[u];
      sk += w_ * (r * r);
    }
  }
  const double llk = -0.5 * ext_wave_sum(sk);
  if (lane == 0) {
    a.ll[(size_t)k * a.NP + c] = llk;
    const long long idk = a.sid[(size_t)k * a.NP + c];
    for (int q2 = q; q2 < nl; ++q2)
      if (a.sid[(size_t)q2 * a.NP + c] == idk) a.Sst[((size_t)pair_index(k, q2) * 2 + 1) * a.NP + c] = llk;
  }
}

}  // namespace tda
