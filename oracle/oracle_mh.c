/*
 * oracle_mh.c -- scalar C restatement of the reference's single-level MH + AdaptiveMetropolis path.
 * TEST / BASELINE INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg load this library.  The shipped engine never calls it.
 *
 * One chain = one sequential loop, exactly the reference's order of operations per step
 * (paths relative to /root/reference):
 *   proposal   theta' = theta + scaling * L z            tinyDA/proposal.py:247-251 (L = chol C, see gen_golden.py)
 *   link       prior.logpdf, F = A theta', loglike(F)    tinyDA/posterior.py:78-110, distributions.py:324-326
 *   accept     u < exp(post' - post), NaN -> 0           tinyDA/proposal.py:253-258, chain.py:112
 *   adapt      t++, scaling, RecursiveSampleMoments.update, C <- Sigma   proposal.py:228-245,502-512; utils.py:113-124
 * Chains run in parallel over OpenMP threads (the reference's Ray mode runs one process per chain).
 * Unlike the reference, the factor of C is cached between swaps instead of re-derived on every draw
 * (np.random.multivariate_normal repeats an SVD per call) - so this baseline is faster than the reference.
 *
 * Pinned: tests/test_oracle_c.py checks it against the NumPy oracle and the reference's golden vectors.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int chol(const double* C, int d, double* L) {
  memset(L, 0, sizeof(double) * d * d);
  for (int k = 0; k < d; ++k)
    for (int i = k; i < d; ++i) {
      double s = C[i * d + k];
      for (int p = 0; p < k; ++p) s -= L[i * d + p] * L[k * d + p];
      if (i == k) {
        if (!(s > 0.0)) return -1;
        L[k * d + k] = sqrt(s);
      } else
        L[i * d + k] = s / L[k * d + k];
    }
  return 0;
}

static void evaluate(int d, int m, const double* A, const double* data, double noise_var, const double* pmean,
                     const double* pvar, double logconst, const double* th, double* lp, double* ll) {
  double maha = 0.0;
  for (int j = 0; j < d; ++j) {
    const double dv = th[j] - pmean[j];
    maha += dv * dv / pvar[j];
  }
  *lp = -0.5 * (logconst + maha);
  double ss = 0.0;
  for (int o = 0; o < m; ++o) {
    const double* a = A + (size_t)o * d;
    double f = 0.0;
    for (int j = 0; j < d; ++j) f += a[j] * th[j];
    const double r = f - data[o];
    ss += r * r;
  }
  const double nrm = sqrt(ss); /* np.linalg.norm(...)**2, distributions.py:326 */
  *ll = -0.5 * (nrm * nrm) / noise_var;
}

/* Layouts follow the engine: z [T][N][d], u [T][N]; outputs stats [T][N][3], accepted [T][N], theta_out [N][d].
 * prior: independent normal N(pmean_j, pvar_j) (identity covariance in BASELINE configs).
 * kind: 0 = GaussianRandomWalk (C fixed), 2 = AdaptiveMetropolis.  Returns 0, or -1 if a Cholesky failed. */
int oracle_mh_run(int n_chains, int d, int m, int T, const double* A, const double* data, double noise_var,
                  const double* pmean, const double* pvar, int kind, const double* C0, double scaling0, int adaptive,
                  double gamma, int period, double sd, double eps, int t0, const double* theta0, const double* z,
                  const double* u, double* stats, uint8_t* accepted, double* theta_out, double* sigma_out,
                  int n_threads) {
  double logconst = d * log(2.0 * M_PI);
  for (int j = 0; j < d; ++j) logconst += log(pvar[j]);
  int status = 0;
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
  for (int c = 0; c < n_chains; ++c) {
    double* th = (double*)malloc(sizeof(double) * d * 5);
    double *prop = th + d, *mu = th + 2 * d, *mun = th + 3 * d, *inc = th + 4 * d;
    double* L = (double*)malloc(sizeof(double) * d * d * 2);
    double* Sg = L + d * d;
    memcpy(th, theta0 + (size_t)c * d, sizeof(double) * d);
    memcpy(mu, th, sizeof(double) * d);
    memset(Sg, 0, sizeof(double) * d * d);
    if (chol(C0, d, L)) status = -1;
    double lp, ll, scaling = (kind == 2) ? 1.0 : scaling0;
    evaluate(d, m, A, data, noise_var, pmean, pvar, logconst, th, &lp, &ll);
    int t = 0, k = 0, acc_in_period = 0;
    for (int s = 0; s < T; ++s) {
      const double* zs = z + ((size_t)s * n_chains + c) * d;
      for (int i = 0; i < d; ++i) {
        double a = 0.0;
        for (int p = 0; p <= i; ++p) a += L[i * d + p] * zs[p];
        inc[i] = a;
      }
      for (int i = 0; i < d; ++i) prop[i] = th[i] + scaling * inc[i];
      double lpn, lln;
      evaluate(d, m, A, data, noise_var, pmean, pvar, logconst, prop, &lpn, &lln);
      const double postn = lpn + lln, posto = lp + ll;
      double alpha = exp(postn - posto);
      if (postn != postn) alpha = 0.0;
      const int acc = u[(size_t)s * n_chains + c] < alpha;
      if (acc) {
        memcpy(th, prop, sizeof(double) * d);
        lp = lpn;
        ll = lln;
      }
      acc_in_period += acc;
      const size_t r = (size_t)s * n_chains + c;
      if (stats) {
        stats[r * 3] = lp;
        stats[r * 3 + 1] = ll;
        stats[r * 3 + 2] = lp + ll;
      }
      if (accepted) accepted[r] = (uint8_t)acc;
      /* adapt */
      t += 1;
      if (t % period == 0) {
        if (adaptive) {
          const double rate = (double)acc_in_period / (double)period;
          scaling = exp(log(scaling) + pow(gamma, -(double)k) * (rate - 0.24));
          k += 1;
        }
        acc_in_period = 0;
      }
      if (kind == 2) {
        const double tt = (double)t; /* recursor.t before the update */
        for (int i = 0; i < d; ++i) mun[i] = (1.0 / (tt + 1.0)) * (tt * mu[i] + th[i]);
        const double ca = (tt - 1.0) / tt, cb = sd / tt;
        for (int i = 0; i < d; ++i)
          for (int j = 0; j < d; ++j) {
            double M = (tt * (mu[i] * mu[j]) - (tt + 1.0) * (mun[i] * mun[j])) + th[i] * th[j];
            if (i == j) M = M + eps;
            Sg[i * d + j] = ca * Sg[i * d + j] + cb * M;
          }
        memcpy(mu, mun, sizeof(double) * d);
        if (t >= t0 && t % period == 0) {
          if (chol(Sg, d, L)) status = -1;
        }
      }
    }
    if (theta_out) memcpy(theta_out + (size_t)c * d, th, sizeof(double) * d);
    if (sigma_out) memcpy(sigma_out + (size_t)c * d * d, Sg, sizeof(double) * d * d);
    free(th);
    free(L);
  }
  return status;
}

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
