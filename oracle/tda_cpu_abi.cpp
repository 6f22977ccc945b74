/*
 * tda_cpu_abi.cpp -- "libtda_cpu.so": the C-ABI of include/tinyda_amd.h compiled a second time, for the CPU.
 * TEST / BASELINE INFRASTRUCTURE, NOT PRODUCT CODE: only tests/ and bench.py's cpu_baseline leg load
 * this library, explicitly and by path; tinyda_amd never falls back to it (a lowerable problem without the HIP library raises).
 *
 * What it is for: (i) the same ctypes binding and Engine wrapper that drive the GPU library can be exercised on a box without a
 * GPU against the reference's golden traces (tests/test_cpu_abi.py); (ii) a CPU baseline behind the very interface the GPU
 * engine is measured through.  It covers the headline path: single-level chains with a linear forward model, isotropic or
 * diagonal noise, a multivariate-normal prior, GaussianRandomWalk / CrankNicolson / AdaptiveMetropolis (with the global
 * scaling adaptation), recorded or Philox variates (the RNG contract of the header), host buffers.  Everything else returns
 * TDA_ERR_UNSUPPORTED.  One chain per OpenMP thread, one sequential loop per chain in the reference's order of operations
 * (paths relative to /root/reference):
 *   proposal   theta' = theta + scaling L z / sqrt(1 - b^2) theta + b L z     tinyDA/proposal.py:247-251, :349-355
 *   link       prior.logpdf (scipy MVN, normalised), F = A theta' + b, loglike tinyDA/posterior.py:78-110, distributions.py:304-329
 *   accept     u < exp(post' - post) (pCN: likelihood ratio), NaN -> reject    tinyDA/proposal.py:253-258, :357-362; chain.py:112
 *   adapt      t += 1, scaling, RecursiveSampleMoments.update, C <- Sigma      proposal.py:228-245, :502-512; utils.py:113-124
 * The factor of C is cached between swaps (NumPy's multivariate_normal re-runs an SVD per draw), so this is faster than the
 * reference itself -- a reported baseline, not a target.
 *
 * Round 5 ("an honest CPU baseline"): the loops are written so that `-O3 -march=native` vectorises them WITHOUT
 * changing one rounding (-ffp-contract=off stays): the forward model runs over a transposed copy of A, a block of outputs per
 * vector register, each output's sum still taken in the order j = 0 .. d-1 (the scalar loop's order: results are bit-identical to
 * the -O2 build, which a test holds); the proposal's L z likewise over a transposed factor; the moment recursion with its two
 * loop-invariant quotients formed once per step.  Two builds exist: libtda_cpu.so (-O2, portable: the parity / sanitizer tests)
 * and libtda_cpu_native.so (-O3 -march=native, built on the machine that runs it: bench.py's cpu_baseline).
 */
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "tinyda_amd.h"

#include "../tinyda_amd/csrc/tda_philox.h"

namespace {

thread_local std::string g_err;
int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

bool cholesky(const double* C, int d, std::vector<double>& L) {
  L.assign((size_t)d * d, 0.0);
  for (int k = 0; k < d; ++k)
    for (int i = k; i < d; ++i) {
      double s = C[(size_t)i * d + k];
      for (int p = 0; p < k; ++p) s -= L[(size_t)i * d + p] * L[(size_t)k * d + p];
      if (i == k) {
        if (!(s > 0.0)) return false;
        L[(size_t)k * d + k] = std::sqrt(s);
      } else {
        L[(size_t)i * d + k] = s / L[(size_t)k * d + k];
      }
    }
  return true;
}

struct ChainState {
  std::vector<double> theta, L, Lt, mu, sigma;  // Lt[k][i] = L[i][k] (the draw runs over it)
  void set_factor(const std::vector<double>& Ln, int d) {
    L = Ln;
    Lt.assign((size_t)d * d, 0.0);
    for (int i = 0; i < d; ++i)
      for (int k = 0; k <= i; ++k) Lt[(size_t)k * d + i] = Ln[(size_t)i * d + k];
  }
  double lp = 0.0, ll = 0.0, scaling = 1.0;
  int flags = 0;
};

}  // namespace

struct tda_engine {
  tda_config cfg{};
  int d = 0, m = 0;
  int64_t N = 0;
  bool prior_set = false, level_set = false, prop_set = false, inited = false;
  std::vector<double> pmean, pcov, pL, pW;  // prior: mean, covariance, Cholesky factor, its inverse (whitening)
  double plogconst = 0.0;
  std::vector<double> A, b, data, w;
  std::vector<double> At;  // [d][mpad]: A transposed, outputs contiguous (padding rows are zero)
  int mpad = 0;
  int noise_kind = 0;
  double var = 1.0;
  tda_proposal_params pp{};
  std::vector<double> C0;
  double am_sd = 1.0;
  std::vector<ChainState> ch;
  std::vector<std::vector<uint8_t>> window;  // accepted[-period:] of every chain since the last boundary
  int64_t t = 0, k_adapt = 0;
  const double* z_rep = nullptr;
  const double* u_rep = nullptr;
  int64_t rep_steps = 0, rep_pos = 0;
  double* z_exp = nullptr;
  double* u_exp = nullptr;
  int64_t exp_steps = 0, exp_pos = 0;
  std::vector<double> z_rep_own, u_rep_own;

  void evaluate(const double* th, double& lp, double& ll) const {
    // scipy.stats.multivariate_normal.logpdf: -1/2 (d log 2 pi + log det + |W (x - mean)|^2)
    double maha = 0.0;
    for (int i = 0; i < d; ++i) {
      double s = 0.0;
      for (int j = 0; j <= i; ++j) s += pW[(size_t)i * d + j] * (th[j] - pmean[j]);
      maha += s * s;
    }
    lp = -0.5 * (plogconst + maha);
    // F = A theta: OB outputs at a time, each output's sum in the order j = 0 .. d-1 (what `f += a[j] * th[j]` per output does);
    // the sum of squares in the order o = 0 .. m-1
    constexpr int OB = 32;
    double ss = 0.0;
    const double* __restrict__ at = At.data();
    for (int o0 = 0; o0 < m; o0 += OB) {
      double f[OB];
      for (int k = 0; k < OB; ++k) f[k] = 0.0;
      for (int j = 0; j < d; ++j) {
        const double tj = th[j];
        const double* __restrict__ col = at + (size_t)j * mpad + o0;
        for (int k = 0; k < OB; ++k) f[k] += col[k] * tj;
      }
      const int nk = m - o0 < OB ? m - o0 : OB;
      for (int k = 0; k < nk; ++k) {
        const int o = o0 + k;
        const double r = (f[k] + b[o]) - data[o];
        ss += noise_kind == TDA_NOISE_DIAG ? r * r * w[o] : r * r;
      }
    }
    if (noise_kind == TDA_NOISE_DIAG) {
      ll = -0.5 * ss;
    } else {  // -0.5 * np.linalg.norm(r) ** 2 / variance (distributions.py:326)
      const double nrm = std::sqrt(ss);
      ll = -0.5 * (nrm * nrm) / var;
    }
  }
};

extern "C" {

const char* tda_last_error(void) { return g_err.c_str(); }
const char* tda_version(void) { return "tinyda_amd 0.5 (cpu twin of the C-ABI: test / baseline infrastructure)"; }
int64_t tda_release_cached_memory(void) { return 0; }

int tda_engine_create(const tda_config* cfg, tda_engine** out) {
  if (!cfg || !out) return fail(TDA_ERR_INVALID, "null argument");
  if (cfg->struct_size != sizeof(tda_config)) return fail(TDA_ERR_INVALID, "tda_config.struct_size mismatch");
  if (cfg->dim < 1 || cfg->n_chains < 1) return fail(TDA_ERR_INVALID, "dim and n_chains must be positive");
  if (cfg->n_levels != 1) return fail(TDA_ERR_UNSUPPORTED, "the CPU twin runs single-level chains");
  tda_engine* e = new tda_engine();
  e->cfg = *cfg;
  e->d = cfg->dim;
  e->N = cfg->n_chains;
  *out = e;
  return TDA_OK;
}

void tda_engine_destroy(tda_engine* e) { delete e; }

int tda_engine_set_prior(tda_engine* e, const double* mean, const double* cov) {
  if (!e || !mean || !cov) return fail(TDA_ERR_INVALID, "null argument");
  const int d = e->d;
  e->pmean.assign(mean, mean + d);
  e->pcov.assign(cov, cov + (size_t)d * d);
  if (!cholesky(cov, d, e->pL)) return fail(TDA_ERR_NUMERIC, "prior covariance is not positive definite");
  e->pW.assign((size_t)d * d, 0.0);  // inverse of the lower factor by forward substitution
  for (int c = 0; c < d; ++c)
    for (int i = c; i < d; ++i) {
      double s = i == c ? 1.0 : 0.0;
      for (int k = c; k < i; ++k) s -= e->pL[(size_t)i * d + k] * e->pW[(size_t)k * d + c];
      e->pW[(size_t)i * d + c] = s / e->pL[(size_t)i * d + i];
    }
  double logdet = 0.0;
  for (int i = 0; i < d; ++i) logdet += 2.0 * std::log(e->pL[(size_t)i * d + i]);
  e->plogconst = d * std::log(2.0 * M_PI) + logdet;
  e->prior_set = true;
  e->inited = false;
  return TDA_OK;
}

int tda_engine_set_level(tda_engine* e, int level, int m, const double* A, const double* b, const double* data, int noise_kind,
                         const double* noise) {
  if (!e || !A || !data || !noise) return fail(TDA_ERR_INVALID, "null argument");
  if (level != 0 || m < 1) return fail(TDA_ERR_INVALID, "level / m out of range");
  if (noise_kind != TDA_NOISE_ISO && noise_kind != TDA_NOISE_DIAG) return fail(TDA_ERR_UNSUPPORTED, "the CPU twin knows isotropic and diagonal noise");
  e->m = m;
  e->A.assign(A, A + (size_t)m * e->d);
  e->mpad = (m + 31) / 32 * 32;
  e->At.assign((size_t)e->d * e->mpad, 0.0);
  for (int o = 0; o < m; ++o)
    for (int j = 0; j < e->d; ++j) e->At[(size_t)j * e->mpad + o] = A[(size_t)o * e->d + j];
  e->b.assign(m, 0.0);
  if (b) e->b.assign(b, b + m);
  e->data.assign(data, data + m);
  e->noise_kind = noise_kind;
  e->var = noise[0];
  e->w.assign(m, 0.0);
  if (noise_kind == TDA_NOISE_DIAG)
    for (int o = 0; o < m; ++o) e->w[o] = 1.0 / noise[o];
  e->level_set = true;
  e->inited = false;
  return TDA_OK;
}

int tda_engine_set_proposal(tda_engine* e, const tda_proposal_params* p) {
  if (!e || !p) return fail(TDA_ERR_INVALID, "null argument");
  if (p->struct_size != sizeof(tda_proposal_params)) return fail(TDA_ERR_INVALID, "tda_proposal_params.struct_size mismatch");
  if (p->kind != TDA_PROP_GRW && p->kind != TDA_PROP_PCN && p->kind != TDA_PROP_AM)
    return fail(TDA_ERR_UNSUPPORTED, "the CPU twin knows GaussianRandomWalk, CrankNicolson and AdaptiveMetropolis");
  if (p->kind != TDA_PROP_PCN && !p->C) return fail(TDA_ERR_INVALID, "proposal covariance missing");
  if (p->period < 1) return fail(TDA_ERR_INVALID, "period must be >= 1");
  e->pp = *p;
  if (p->C) e->C0.assign(p->C, p->C + (size_t)e->d * e->d);
  e->pp.C = nullptr;
  e->am_sd = p->sd > 0.0 ? p->sd : std::fmin(1.0, 2.4 * 2.4 / e->d);  // proposal.py:465-468
  e->prop_set = true;
  e->inited = false;
  return TDA_OK;
}

int tda_engine_init(tda_engine* e, const double* theta0) {
  if (!e) return fail(TDA_ERR_INVALID, "null engine");
  if (!e->prior_set || !e->level_set || !e->prop_set) return fail(TDA_ERR_STATE, "set_prior, set_level and set_proposal must precede init");
  const int d = e->d;
  std::vector<double> L;
  if (!cholesky(e->pp.kind == TDA_PROP_PCN ? e->pcov.data() : e->C0.data(), d, L))
    return fail(TDA_ERR_NUMERIC, "proposal covariance is not positive definite");
  e->ch.assign(e->N, ChainState());
  e->window.assign(e->N, std::vector<uint8_t>());
  for (int64_t c = 0; c < e->N; ++c) {
    ChainState& s = e->ch[c];
    s.theta.resize(d);
    if (theta0) {
      std::copy(theta0 + (size_t)c * d, theta0 + (size_t)(c + 1) * d, s.theta.begin());
    } else {  // theta0 ~ prior from RNG stream 2 (sampler.py:209)
      std::vector<double> z(d + 1);
      for (int bb = 0; bb < (d + 1) / 2; ++bb)
        tda::normal_pair(e->cfg.seed, (uint32_t)(e->cfg.chain_offset + c), 0u, tda::STREAM_INIT, (uint32_t)bb, z[2 * bb], z[2 * bb + 1]);
      for (int i = 0; i < d; ++i) {
        double acc = 0.0;
        for (int k = 0; k <= i; ++k) acc = std::fma(e->pL[(size_t)i * d + k], z[k], acc);
        s.theta[i] = e->pmean[i] + acc;
      }
    }
    s.set_factor(L, d);
    s.scaling = e->pp.kind == TDA_PROP_AM ? 1.0 : e->pp.scaling;
    if (e->pp.kind == TDA_PROP_AM) {  // RecursiveSampleMoments(mu0 = theta0, sigma0 = 0) (proposal.py:495-500)
      s.mu = s.theta;
      s.sigma.assign((size_t)d * d, 0.0);
    }
    e->evaluate(s.theta.data(), s.lp, s.ll);
  }
  e->t = 0;
  e->k_adapt = 0;
  e->rep_pos = 0;
  e->exp_pos = 0;
  e->inited = true;
  return TDA_OK;
}

int tda_engine_set_replay(tda_engine* e, const double* z, const double* u, int64_t n_steps) {
  if (!e) return fail(TDA_ERR_INVALID, "null engine");
  e->rep_steps = e->rep_pos = 0;
  if (!z || !u || n_steps <= 0) return TDA_OK;
  e->z_rep_own.assign(z, z + (size_t)n_steps * e->N * e->d);  // the GPU engine copies them too: callers may free theirs
  e->u_rep_own.assign(u, u + (size_t)n_steps * e->N);
  e->z_rep = e->z_rep_own.data();
  e->u_rep = e->u_rep_own.data();
  e->rep_steps = n_steps;
  return TDA_OK;
}

int tda_engine_set_export(tda_engine* e, double* z, double* u, int64_t n_steps) {
  if (!e) return fail(TDA_ERR_INVALID, "null engine");
  e->z_exp = z;
  e->u_exp = u;
  e->exp_steps = (z && u && n_steps > 0) ? n_steps : 0;
  e->exp_pos = 0;
  return TDA_OK;
}

int tda_engine_run(tda_engine* e, int64_t n_iter, const tda_outputs* out) {
  if (!e || !e->inited) return fail(TDA_ERR_STATE, "engine not initialised");
  if (n_iter < 0) return fail(TDA_ERR_INVALID, "n_iterations < 0");
  if (out && out->struct_size != sizeof(tda_outputs)) return fail(TDA_ERR_INVALID, "tda_outputs.struct_size mismatch");
  if (out && (out->params || out->stats || out->accepted) && (int64_t)out->rows < n_iter)
    return fail(TDA_ERR_INVALID, "tda_outputs[0].rows = %u but this run() produces %lld records for that level", out->rows, (long long)n_iter);
  if (e->rep_steps && e->rep_pos + n_iter > e->rep_steps) return fail(TDA_ERR_INVALID, "replay buffer too short");
  if (e->exp_steps && e->exp_pos + n_iter > e->exp_steps) return fail(TDA_ERR_INVALID, "export buffer too small");
  const int d = e->d, period = e->pp.period, kind = e->pp.kind;
  const bool is_am = kind == TDA_PROP_AM, adaptive = e->pp.adaptive != 0;
  const int64_t N = e->N, t_start = e->t;
  int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
  for (int64_t c = 0; c < N; ++c) {
    ChainState& s = e->ch[c];
    std::vector<uint8_t>& win = e->window[c];
    std::vector<double> z(d + 1), prop(d), inc(d), mu_new(d), Lnew;
    int64_t k_ad = e->k_adapt;
    for (int64_t it = 0; it < n_iter; ++it) {
      const int64_t t = t_start + it;  // proposal.t before this step
      double u;
      if (e->rep_steps) {
        const double* zr = e->z_rep + ((size_t)(e->rep_pos + it) * N + c) * d;
        std::copy(zr, zr + d, z.begin());
        u = e->u_rep[(size_t)(e->rep_pos + it) * N + c];
      } else {
        for (int bb = 0; bb < (d + 1) / 2; ++bb)
          tda::normal_pair(e->cfg.seed, (uint32_t)(e->cfg.chain_offset + c), (uint32_t)t, tda::STREAM_PROPOSAL, (uint32_t)bb, z[2 * bb], z[2 * bb + 1]);
        u = tda::accept_uniform(e->cfg.seed, (uint32_t)(e->cfg.chain_offset + c), (uint32_t)t, 0u);
      }
      if (e->exp_steps) {
        std::copy(z.begin(), z.begin() + d, e->z_exp + ((size_t)(e->exp_pos + it) * N + c) * d);
        e->u_exp[(size_t)(e->exp_pos + it) * N + c] = u;
      }
      // inc = L z, every element's sum in the order k = 0 .. i (Lt is zero above the diagonal of L: those terms are skipped by
      // starting row k at i = k)
      for (int i = 0; i < d; ++i) inc[i] = 0.0;
      for (int k = 0; k < d; ++k) {
        const double zk = z[k];
        const double* __restrict__ lt = s.Lt.data() + (size_t)k * d;
        double* __restrict__ ip = inc.data();
        for (int i = k; i < d; ++i) ip[i] += lt[i] * zk;
      }
      const double keep = kind == TDA_PROP_PCN ? std::sqrt(1.0 - s.scaling * s.scaling) : 1.0;
      for (int i = 0; i < d; ++i) prop[i] = kind == TDA_PROP_PCN ? keep * s.theta[i] + s.scaling * inc[i] : s.theta[i] + s.scaling * inc[i];
      double lp_n, ll_n;
      e->evaluate(prop.data(), lp_n, ll_n);
      const double post_n = lp_n + ll_n;
      double alpha = kind == TDA_PROP_PCN ? std::exp(ll_n - s.ll) : std::exp(post_n - (s.lp + s.ll));
      if (post_n != post_n) alpha = 0.0;
      const bool acc = u < alpha;
      if (acc) {
        s.theta = prop;
        s.lp = lp_n;
        s.ll = ll_n;
      }
      win.push_back(acc ? 1 : 0);
      if (out) {
        const size_t r = (size_t)it * N + c;
        if (out->params) std::copy(s.theta.begin(), s.theta.end(), out->params + r * d);
        if (out->stats) {
          out->stats[r * 3 + 0] = s.lp;
          out->stats[r * 3 + 1] = s.ll;
          out->stats[r * 3 + 2] = s.lp + s.ll;
        }
        if (out->accepted) out->accepted[r] = acc ? 1 : 0;
      }
      // ---- adapt (proposal.py:228-245, :502-512) ----
      const int64_t tn = t + 1;
      const bool boundary = tn % period == 0;
      if (adaptive && boundary) {
        int hits = 0;
        const size_t n = win.size();
        for (size_t i = n >= (size_t)period ? n - period : 0; i < n; ++i) hits += win[i];
        const double rate = (double)hits / (double)period;
        const double target = 0.24;
        s.scaling = std::exp(std::log(s.scaling) + std::pow(e->pp.gamma, -(double)k_ad) * (rate - target));
        k_ad += 1;
      }
      if (boundary) win.clear();
      if (is_am) {  // RecursiveSampleMoments.update (utils.py:113-122): recursion counter = adapt calls + 1
        const double tt = (double)tn, t1 = tt + 1.0, ca = (tt - 1.0) / tt, cb = e->am_sd / tt, eps = e->pp.epsilon;
        const double* __restrict__ mu = s.mu.data();
        const double* __restrict__ th = s.theta.data();
        double* __restrict__ mn = mu_new.data();
        for (int i = 0; i < d; ++i) mn[i] = (1.0 / t1) * (tt * mu[i] + th[i]);
        for (int i = 0; i < d; ++i) {
          double* __restrict__ sg = s.sigma.data() + (size_t)i * d;
          const double mi = mu[i], ni = mn[i], ti = th[i];
          for (int j = 0; j < d; ++j) {
            const double M = ((tt * (mi * mu[j]) - t1 * (ni * mn[j])) + ti * th[j]) + (i == j ? eps : 0.0);
            sg[j] = ca * sg[j] + cb * M;
          }
        }
        s.mu = mu_new;
        if (tn >= e->pp.t0 && boundary) {
          if (cholesky(s.sigma.data(), d, Lnew)) s.set_factor(Lnew, d);
          else {
            s.flags |= 1;
            bad |= 1;
          }
        }
      }
    }
  }
  if (adaptive) e->k_adapt += (t_start + n_iter) / period - t_start / period;
  e->t += n_iter;
  if (e->rep_steps) e->rep_pos += n_iter;
  if (e->exp_steps) e->exp_pos += n_iter;
  (void)bad;
  return TDA_OK;
}

int tda_engine_sync(tda_engine* e) { return e ? TDA_OK : fail(TDA_ERR_INVALID, "null engine"); }

int tda_engine_get_current(tda_engine* e, double* theta, double* stats) {
  if (!e || !e->inited) return fail(TDA_ERR_STATE, "engine not initialised");
  for (int64_t c = 0; c < e->N; ++c) {
    if (theta) std::copy(e->ch[c].theta.begin(), e->ch[c].theta.end(), theta + (size_t)c * e->d);
    if (stats) {
      stats[c * 3 + 0] = e->ch[c].lp;
      stats[c * 3 + 1] = e->ch[c].ll;
      stats[c * 3 + 2] = e->ch[c].lp + e->ch[c].ll;
    }
  }
  return TDA_OK;
}

int tda_engine_get_level_state(tda_engine* e, int level, double* theta, double* stats) {
  if (level != 0) return fail(TDA_ERR_INVALID, "level out of range");
  return tda_engine_get_current(e, theta, stats);
}

int tda_engine_get_proposal_state(tda_engine* e, double* scaling, double* C, double* am_mu, double* am_sigma, int64_t* counters) {
  if (!e || !e->inited) return fail(TDA_ERR_STATE, "engine not initialised");
  const int d = e->d;
  if ((am_mu || am_sigma) && e->pp.kind != TDA_PROP_AM) return fail(TDA_ERR_STATE, "proposal has no running moments");
  for (int64_t c = 0; c < e->N; ++c) {
    const ChainState& s = e->ch[c];
    if (scaling) scaling[c] = s.scaling;
    if (C)
      for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) {
          double acc = 0.0;
          for (int k = 0; k <= (i < j ? i : j); ++k) acc += s.L[(size_t)i * d + k] * s.L[(size_t)j * d + k];
          C[((size_t)c * d + i) * d + j] = acc;
        }
    if (am_mu) std::copy(s.mu.begin(), s.mu.end(), am_mu + (size_t)c * d);
    if (am_sigma) std::copy(s.sigma.begin(), s.sigma.end(), am_sigma + (size_t)c * d * d);
  }
  if (counters) {
    counters[0] = e->t;
    counters[1] = e->k_adapt;
  }
  return TDA_OK;
}

int tda_engine_get_flags(tda_engine* e, int32_t* flags) {
  if (!e || !flags) return fail(TDA_ERR_INVALID, "null argument");
  for (int64_t c = 0; c < e->N; ++c) flags[c] = e->inited ? e->ch[c].flags : 0;
  return TDA_OK;
}

int tda_engine_evaluate(tda_engine* e, int level, const double* theta, int64_t n, double* stats) {
  if (!e || !theta || !stats) return fail(TDA_ERR_INVALID, "null argument");
  if (level != 0 || !e->prior_set || !e->level_set) return fail(TDA_ERR_STATE, "level not set");
  for (int64_t c = 0; c < n; ++c) {
    double lp, ll;
    e->evaluate(theta + (size_t)c * e->d, lp, ll);
    stats[c * 3 + 0] = lp;
    stats[c * 3 + 1] = ll;
    stats[c * 3 + 2] = lp + ll;
  }
  return TDA_OK;
}

int tda_engine_rng_probe(tda_engine* e, int64_t step, double* z, double* u) {
  if (!e || !z || !u) return fail(TDA_ERR_INVALID, "null argument");
  const int d = e->d;
  std::vector<double> zz(d + 1);
  for (int64_t c = 0; c < e->N; ++c) {
    for (int bb = 0; bb < (d + 1) / 2; ++bb)
      tda::normal_pair(e->cfg.seed, (uint32_t)(e->cfg.chain_offset + c), (uint32_t)step, tda::STREAM_PROPOSAL, (uint32_t)bb, zz[2 * bb], zz[2 * bb + 1]);
    std::copy(zz.begin(), zz.begin() + d, z + (size_t)c * d);
    u[c] = tda::accept_uniform(e->cfg.seed, (uint32_t)(e->cfg.chain_offset + c), (uint32_t)step, 0u);
  }
  return TDA_OK;
}

int tda_rng_philox(int device, const uint32_t* counter, const uint32_t* key, uint32_t* out) {
  if (!counter || !key || !out) return fail(TDA_ERR_INVALID, "null argument");
  if (device >= 0) return fail(TDA_ERR_UNSUPPORTED, "the CPU twin has no device");
  const tda::u32x4 r = tda::philox4x32_10(tda::u32x4{counter[0], counter[1], counter[2], counter[3]}, key[0], key[1]);
  out[0] = r.x;
  out[1] = r.y;
  out[2] = r.z;
  out[3] = r.w;
  return TDA_OK;
}

int tda_engine_set_profiling(tda_engine*, int) { return TDA_OK; }
int tda_engine_get_profile(tda_engine* e, tda_profile* p) {
  if (!e || !p) return fail(TDA_ERR_INVALID, "null argument");
  const uint32_t sz = p->struct_size;
  if (sz != sizeof(tda_profile) && sz != 48) return fail(TDA_ERR_INVALID, "tda_profile.struct_size mismatch");
  std::memset(p, 0, sz);  // (a 0.3 caller's struct is 48 bytes)
  p->struct_size = sz;
  return TDA_OK;
}

// ---- the rest of the header: declared, exported, not available on the CPU twin ----
#define TDA_CPU_UNSUPPORTED(what) return fail(TDA_ERR_UNSUPPORTED, what " is not part of the CPU twin of the ABI")
int tda_engine_set_record_thinning(tda_engine*, int32_t thin) { if (thin == 1) return TDA_OK; TDA_CPU_UNSUPPORTED("record thinning"); }
int tda_engine_set_progress(tda_engine*, int) { TDA_CPU_UNSUPPORTED("progress reporting"); }
int tda_engine_set_proposal_spectrum(tda_engine*, const double*, const double*) { TDA_CPU_UNSUPPORTED("operator-weighted pCN"); }
int tda_engine_detach_proposal_state(tda_engine*, tda_proposal_snapshot**) { TDA_CPU_UNSUPPORTED("proposal snapshots"); }
int tda_proposal_snapshot_read(tda_proposal_snapshot*, double*, double*, double*, double*, int64_t*) { TDA_CPU_UNSUPPORTED("proposal snapshots"); }
void tda_proposal_snapshot_destroy(tda_proposal_snapshot*) {}
int tda_engine_get_progress(tda_engine*, int64_t*, int64_t*, double*) { TDA_CPU_UNSUPPORTED("progress reporting"); }
int tda_engine_set_proposal_dreamz(tda_engine*, const tda_dreamz_params*) { TDA_CPU_UNSUPPORTED("DREAM(Z)"); }
int tda_engine_set_proposal_operators(tda_engine*, const double*, const double*) { TDA_CPU_UNSUPPORTED("OperatorWeightedCrankNicolson"); }
int tda_engine_set_archive(tda_engine*, const double*) { TDA_CPU_UNSUPPORTED("DREAM(Z)"); }
int tda_engine_set_level_rosenbrock(tda_engine*, int, double, double, double, double) { TDA_CPU_UNSUPPORTED("the Rosenbrock level"); }
int tda_engine_set_subchains(tda_engine*, const int32_t*, int) { TDA_CPU_UNSUPPORTED("Delayed Acceptance / MLDA"); }
int tda_engine_set_error_model(tda_engine*, int) { TDA_CPU_UNSUPPORTED("the adaptive error model"); }
int tda_engine_get_error_model(tda_engine*, int, double*, double*) { TDA_CPU_UNSUPPORTED("the adaptive error model"); }
int tda_engine_set_replay_level(tda_engine*, int, const double*, int64_t) { TDA_CPU_UNSUPPORTED("Delayed Acceptance / MLDA"); }
int tda_engine_set_replay_dreamz(tda_engine*, const int32_t*, const int32_t*, const double*, const int32_t*, const double*, const double*,
                                 const double*, int64_t) {
  TDA_CPU_UNSUPPORTED("DREAM(Z)");
}
int tda_engine_get_dreamz_state(tda_engine*, double*, int64_t*) { TDA_CPU_UNSUPPORTED("DREAM(Z)"); }
int tda_engine_archive_take(tda_engine*, double*, int64_t*) { TDA_CPU_UNSUPPORTED("DREAM"); }
int tda_engine_archive_append(tda_engine*, const double*, int64_t) { TDA_CPU_UNSUPPORTED("DREAM"); }
int tda_engine_archive_ipc_handle(tda_engine*, void*) { TDA_CPU_UNSUPPORTED("DREAM"); }
int tda_engine_archive_pointer(tda_engine*, void**) { TDA_CPU_UNSUPPORTED("DREAM"); }
int tda_engine_set_archive_peers(tda_engine*, int, int, const void*, const double* const*) { TDA_CPU_UNSUPPORTED("DREAM"); }
int tda_engine_archive_local_sums(tda_engine*, double*) { TDA_CPU_UNSUPPORTED("DREAM"); }
int tda_engine_archive_publish(tda_engine*, const double*) { TDA_CPU_UNSUPPORTED("DREAM"); }
int tda_engine_set_archive_auto_append(tda_engine*, int) { TDA_CPU_UNSUPPORTED("DREAM"); }
int tda_engine_reduce_moments(tda_engine*, const double*, int64_t, double*) { TDA_CPU_UNSUPPORTED("pooled moments"); }
int tda_engine_set_proposal_covariance(tda_engine*, const double*) { TDA_CPU_UNSUPPORTED("pooled moments"); }
int64_t tda_engine_state_size(tda_engine*) { return fail(TDA_ERR_UNSUPPORTED, "checkpoints are not part of the CPU twin of the ABI"); }
int tda_engine_get_state(tda_engine*, void*, int64_t) { TDA_CPU_UNSUPPORTED("checkpoints"); }
int tda_engine_set_state(tda_engine*, const void*, int64_t) { TDA_CPU_UNSUPPORTED("checkpoints"); }
int tda_engine_set_prior_joint(tda_engine*, const int32_t*, const double*, const double*) { TDA_CPU_UNSUPPORTED("JointPrior"); }
int tda_engine_set_level_source(tda_engine*, int, const char*, int32_t, const double*, int32_t, const double*) {
  TDA_CPU_UNSUPPORTED("source-defined models");
}
int tda_engine_set_level_callback(tda_engine*, int, tda_forward_batch_fn, void*, int32_t, const double*, int32_t, const double*) {
  TDA_CPU_UNSUPPORTED("callback models");
}
int tda_diag_ess_rhat(int, void*, const double*, int64_t, int64_t, int32_t, int64_t, double*, double*) {
  TDA_CPU_UNSUPPORTED("device diagnostics");
}

}  // extern "C"
