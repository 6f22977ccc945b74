"""Round-5 paths against closed forms (conjugate linear-Gaussian targets; one JSON line per check):
  * 128 parameters, single level: a random walk with the FIXED covariance 2.4^2/d x posterior covariance started in stationarity
    (k_mh_steps<128>, k_rng<128>, k_wide_apply): variances 1.000 of the closed form, acceptance 0.234; AdaptiveMetropolis from a
    pilot (k_wide_adapt, the covariance swap on k_aem_refresh<8,1>): the reference's finite-adaptation bias, as at 64 parameters;
  * Delayed Acceptance at 128 parameters and six-level MLDA (k_ml_steps<128, 2>, k_ml_steps<32, 6>): the FINEST posterior stays
    invariant (chains started from exact draws of it, pCN base proposals, nothing adapts);
  * Delayed Acceptance under the dense state-independent error model at 256 outputs (k_aem_refresh_big, k_aem_action<256>,
    k_aem_base_steps<16>): the finest posterior stays invariant whatever the error model learns."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from tinyda_amd import _lib
from tinyda_amd.engine import Engine

_lib.load()
SIGMA = 0.1


def conjugate(A, y, d):
    cov = np.linalg.inv(A.T @ A / SIGMA ** 2 + np.eye(d))
    return cov @ (A.T @ y / SIGMA ** 2), cov


def summarise(name, P, mean_post, cov_post, extra):
    T, N, d = P.shape
    half = P[T // 2:]
    r = half.reshape(-1, d).var(axis=0) / np.diag(cov_post)
    cm = half.mean(axis=0)
    z = (cm.mean(axis=0) - mean_post) / (cm.std(axis=0, ddof=1) / np.sqrt(N))
    print(json.dumps(dict(check=name, recorded_iterations=int(T), chains=int(N), var_ratio_min=float(r.min()), var_ratio_max=float(r.max()),
                          var_ratio_mean=float(r.mean()), max_abs_z_of_mean=float(np.abs(z).max()), rms_z_of_mean=float(np.sqrt((z ** 2).mean())), **extra)), flush=True)


def single_level_128():
    d, m, N = 128, 1024, 4096
    rng = np.random.default_rng(3)
    A = rng.standard_normal((m, d)) / np.sqrt(d)
    y = A @ rng.standard_normal(d) + SIGMA * rng.standard_normal(m)
    mean_post, cov_post = conjugate(A, y, d)
    start = mean_post + (np.linalg.cholesky(cov_post) @ rng.standard_normal((d, N))).T
    e = Engine(N, d, seed=21)
    e.set_prior(np.zeros(d), np.eye(d))
    e.set_level(0, A, y, 0, SIGMA ** 2)
    e.set_proposal(0, (2.4 ** 2 / d) * cov_post)
    e.init(start)
    e.run(2000, None, None, None)
    P, _, acc = e.run_host(1200)
    e.close()
    summarise("128 parameters: GaussianRandomWalk with the fixed covariance 2.4^2/d x posterior covariance, started in stationarity", P, mean_post,
              cov_post, dict(acceptance=float(acc.mean())))
    e = Engine(N, d, seed=22)
    e.set_prior(np.zeros(d), np.eye(d))
    e.set_level(0, A, y, 0, SIGMA ** 2)
    e.set_proposal(2, (2.4 ** 2 / d) * 0.5 * np.diag(np.diag(cov_post)), t0=200, period=200)
    e.init(start)
    e.run(20000, None, None, None)
    P, _, acc = e.run_host(1200)
    e.close()
    summarise("128 parameters: AdaptiveMetropolis(t0 = 200, period 200) from stationarity, after 20000 iterations (per-chain covariances: the reference's finite-adaptation bias)",
              P, mean_post, cov_post, dict(acceptance=float(acc.mean())))


def hierarchy(name, d, ms, sl, n_fine, beta, N=4096, error_model=None, perturb=0.0):
    rng = np.random.default_rng(7)
    truth = rng.standard_normal(d)
    lv = []
    base = rng.standard_normal((ms[-1], d)) / np.sqrt(d)
    for k, m in enumerate(ms):
        if error_model:  # a common output dimension: the coarse model is a perturbed copy of the fine one
            A = base + (perturb * (len(ms) - 1 - k)) * rng.standard_normal((m, d)) / np.sqrt(d)
            lv.append((A, None))
        else:
            A = rng.standard_normal((m, d)) / np.sqrt(d)
            lv.append((A, A @ truth + SIGMA * rng.standard_normal(m)))
    if error_model:
        yf = lv[-1][0] @ truth + SIGMA * rng.standard_normal(ms[-1])
        lv = [(A, yf) for A, _ in lv]
    Af, yf = lv[-1]
    mean_post, cov_post = conjugate(Af, yf, d)
    e = Engine(N, d, seed=31, n_levels=len(ms))
    e.set_prior(np.zeros(d), np.eye(d))
    for k, (A, y) in enumerate(lv):
        if error_model and k < len(ms) - 1:
            e.set_level(k, A, y, 3, SIGMA ** 2 * np.eye(len(y)))
        else:
            e.set_level(k, A, y, 0, SIGMA ** 2)
    e.set_proposal(1, None, scaling=beta)
    e.set_subchains(sl)
    if error_model:
        e.set_error_model(error_model)
    e.init(mean_post + (np.linalg.cholesky(cov_post) @ rng.standard_normal((d, N))).T)
    outs = e.run_levels_host(n_fine)
    e.close()
    summarise(name, outs[-1][0], mean_post, cov_post, dict(acceptance_per_level=[float(o[2].mean()) for o in outs]))


if __name__ == "__main__":
    which = sys.argv[1:] or ["single", "da128", "mlda6", "aem256"]
    if "single" in which:
        single_level_128()
    if "da128" in which:
        hierarchy("Delayed Acceptance at 128 parameters, pCN(0.004), 256 / 2048 observations, subchain 10 (k_ml_steps<128, 2>)", 128, (256, 2048), [10], 160, 0.004)
    if "mlda6" in which:
        hierarchy("six-level MLDA at 32 parameters, pCN(0.01), 32 ... 1024 observations, subchains [2, 2, 2, 2, 2] (k_ml_steps<32, 6>)", 32,
                  (32, 64, 128, 256, 512, 1024), [2, 2, 2, 2, 2], 160, 0.01)
    if "aem256" in which:
        hierarchy("Delayed Acceptance under the dense state-independent error model, 256 outputs on both levels, 32 parameters, pCN(0.02), subchain 4", 32,
                  (256, 256), [4], 160, 0.02, N=2048, error_model="state-independent", perturb=0.03)
