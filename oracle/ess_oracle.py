"""Rank-normalised split bulk ESS and R-hat of Vehtari, Gelman, Simpson, Carpenter, Buerkner (2021), "Rank-normalization,
folding, and localization: an improved R-hat for assessing convergence of MCMC" -- the estimators `arviz.summary` applies in
the reference's notebooks (tinyDA/diagnostics.py:6-111 hands chains to ArviZ; examples/Basic Sampler.ipynb cell 17).
TEST INFRASTRUCTURE, NOT PRODUCT CODE: the checker of tinyda_amd.summaries (NumPy) and of tda_diag_ess_rhat (HIP).

PARITY UNPINNED against ArviZ itself: arviz 0.18.0 (poetry.lock:252) is neither vendored in the reference nor importable here
and the reference's tests pin no value.  This restatement follows the paper's equations directly -- autocovariances by
explicit lagged sums (no FFT), ranks by sorting with tie averaging, the normal quantile from scipy -- so that it shares no
code path with the implementations it checks; tests/test_diagnostics.py additionally pins all three on processes whose
effective sample size is known in closed form.

Paper, section 3-4 (notation: M chains of N draws after splitting):
  z = Phi^-1((r - 3/8) / (S + 1/4)),  r = average ranks over all S = M N draws                    (eq. 14)
  W = mean of within-chain variances, B / N = variance of chain means, var+ = (N-1)/N W + B/N      (eq. 2-3)
  R-hat = sqrt(var+ / W); reported: max(R-hat of z, R-hat of folded z)                            (eq. 4, sec. 4.2)
  rho_t = 1 - (W - mean_m acov_m(t)) / var+                                                       (eq. 10)
  tau = -1 + 2 sum_k P_k, P_k = rho_2k + rho_2k+1 truncated at the first negative pair and made monotone (Geyer 1992),
  ESS = M N / tau                                                                                 (eq. 11-13)
(the conventions Stan and ArviZ add are kept: the autocovariances use the biased 1/N normalisation, W in rho_t is the
1/(N-1) variance, the positive part of the next even lag is added after truncation, tau >= 1 / log10(M N).)
"""
import numpy as np
from scipy.stats import norm


def split_chains(x):
    """[chains, draws] -> [2 chains, draws // 2]: first and last halves (an odd middle draw is dropped)"""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[1] // 2
    return np.vstack([x[:, :n], x[:, x.shape[1] - n:]])


def average_ranks(v):
    """1-based ranks of a flat array, ties sharing the mean of their positions"""
    order = np.argsort(v, kind="mergesort")
    ranks = np.empty(v.size)
    s = v[order]
    i = 0
    while i < s.size:
        j = i
        while j + 1 < s.size and s[j + 1] == s[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    return ranks


def normal_scores(x):
    r = average_ranks(x.reshape(-1)).reshape(x.shape)
    return norm.ppf((r - 0.375) / (x.size + 0.25))


def _rhat_plain(z):
    m, n = z.shape
    means = z.mean(axis=1)
    W = np.mean([np.sum((z[c] - means[c]) ** 2) / (n - 1.0) for c in range(m)])
    B_over_n = np.sum((means - means.mean()) ** 2) / (m - 1.0)
    return float(np.sqrt(((n - 1.0) / n * W + B_over_n) / W))


def rhat(x):
    s = split_chains(x)
    folded = np.abs(s - np.median(s))
    return max(_rhat_plain(normal_scores(s)), _rhat_plain(normal_scores(folded)))


def _ess_of(z):
    """ESS of already split (and, for the bulk estimate, rank-normalised) chains z [M][N] (eq. 10-13)"""
    m, n = z.shape
    if n < 4:
        return float("nan")
    means = z.mean(axis=1)
    zc = z - means[:, None]
    acov = np.empty((m, n))
    for t in range(n):  # biased autocovariance, explicit lagged sums
        acov[:, t] = np.sum(zc[:, :n - t] * zc[:, t:], axis=1) / n
    W = np.mean(acov[:, 0]) * n / (n - 1.0)
    var_plus = W * (n - 1.0) / n
    if m > 1:
        var_plus += np.sum((means - means.mean()) ** 2) / (m - 1.0)
    if not var_plus > 0:
        return float("nan")
    rho = 1.0 - (W - acov.mean(axis=0)) / var_plus
    rho[0] = 1.0
    tau = -1.0
    prev = np.inf
    k = 0
    while 2 * k + 1 < n:
        pair = rho[2 * k] + rho[2 * k + 1]
        if pair < 0:
            break
        prev = min(prev, pair)
        tau += 2.0 * prev
        k += 1
    if 2 * k + 1 < n:
        tau += max(rho[2 * k], 0.0)
    tau = max(tau, 1.0 / np.log10(m * n))
    return float(m * n / tau)


def ess_bulk(x):
    return _ess_of(normal_scores(split_chains(x)))


def ess_tail(x):
    """tail ESS (paper sec. 4.3): the smaller of the ESS of the indicators I(x <= q) at the 5 % and 95 % quantiles of all
    draws, computed on the split chains without rank-normalisation"""
    x = np.asarray(x, dtype=np.float64)
    out = []
    for prob in (0.05, 0.95):
        q = np.quantile(x, prob)
        out.append(_ess_of(split_chains((x <= q).astype(np.float64))))
    return float(min(out))


def ess_mean(x):
    """ESS of the mean: split chains, no rank-normalisation (what the Monte Carlo standard error of the mean uses)"""
    return _ess_of(split_chains(x))


def mcse_mean(x):
    """Monte Carlo standard error of the posterior mean: sd / sqrt(ESS of the mean)"""
    x = np.asarray(x, dtype=np.float64)
    return float(np.std(x, ddof=1) / np.sqrt(ess_mean(x)))


def hdi(x, prob=0.94):
    """highest density interval: the narrowest interval containing `prob` of the pooled draws (unimodal convention)"""
    v = np.sort(np.asarray(x, dtype=np.float64).ravel())
    n = v.size
    k = int(np.floor(prob * n))
    widths = [v[i + k] - v[i] for i in range(n - k)]
    i = int(np.argmin(widths))
    return float(v[i]), float(v[i + k])
