"""Does the engine sample the right distribution?  BASELINE config 2a has a conjugate posterior (Gaussian prior, linear model, Gaussian
noise): after the bench's set-up (pilot + AdaptiveMetropolis burn-in) 20 000 recorded iterations of 4096 chains are compared with the
closed form -- posterior mean in units of its Monte Carlo standard error, covariance entry by entry.  A second run is the control:
a random walk with a FIXED, optimally scaled covariance started in stationarity (no adaptation anywhere), whose variances must come
out at 1.000 and whose acceptance rate at 0.234.  Two JSON lines."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from tinyda_amd import _lib
from tinyda_amd.engine import Engine
from tinyda_amd.summaries import ess_rhat_device

D, M, N, SIGMA = 64, 1024, 4096, 0.1
_lib.load()
rng = np.random.default_rng(1)
A = rng.standard_normal((M, D)) / 8
truth = rng.standard_normal(D)
y = A @ truth + SIGMA * rng.standard_normal(M)
cov_post = np.linalg.inv(A.T @ A / SIGMA ** 2 + np.eye(D))
mean_post = cov_post @ (A.T @ y / SIGMA ** 2)


def engine(kind, **kw):
    e = Engine(N, D, seed=11)
    e.set_prior(np.zeros(D), np.eye(D))
    e.set_level(0, A, y, 0, SIGMA ** 2)
    e.set_proposal(kind, 1e-4 * np.eye(D), **kw)
    return e


pe = engine(0)
pe.init(None)
pe.run(10000, None, None, None)
start, _ = pe.current()
pe.close()
e = engine(2, t0=100, period=100)
e.init(start)
e.run(80000, None, None, None)
T = 2000
p = torch.empty((T, N, D), dtype=torch.float64, device="cuda")
sums = torch.zeros(D, dtype=torch.float64, device="cuda")
outer = torch.zeros((D, D), dtype=torch.float64, device="cuda")
ess = np.zeros(D)
n_rec = 0
for blk in range(10):  # 20 000 iterations in ten recorded windows (the record of one is 4 GiB)
    e.run(T, p, None, None)
    flat = p.reshape(-1, D)
    sums += flat.sum(0)
    outer += flat.T @ flat
    n_rec += flat.shape[0]
    ess += ess_rhat_device(p)["ess"]
e.close()
mean = (sums / n_rec).cpu().numpy()
cov = (outer / n_rec).cpu().numpy() - np.outer(mean, mean)
sd = np.sqrt(np.diag(cov_post))
z = (mean - mean_post) / (sd / np.sqrt(ess))  # ess: sum of the windows' bulk ESS (a lower bound of the whole run's)
rel = np.abs(cov - cov_post) / np.sqrt(np.outer(np.diag(cov_post), np.diag(cov_post)))
print(json.dumps(dict(check="C2a conjugate posterior, 4096 chains x 20000 recorded iterations after pilot + burn-in", draws=n_rec,
                      min_bulk_ess=float(ess.min()), max_abs_z_of_mean=float(np.abs(z).max()), rms_z_of_mean=float(np.sqrt((z ** 2).mean())),
                      max_cov_error_relative_to_sd_products=float(rel.max()), var_ratio_min=float((np.diag(cov) / np.diag(cov_post)).min()),
                      var_ratio_max=float((np.diag(cov) / np.diag(cov_post)).max()))))

# ---- control: no adaptation ----
e = Engine(N, D, seed=12)
e.set_prior(np.zeros(D), np.eye(D))
e.set_level(0, A, y, 0, SIGMA ** 2)
e.set_proposal(0, (2.4 ** 2 / D) * cov_post)   # fixed, optimally scaled random walk: no adaptation at all
Lp = np.linalg.cholesky(cov_post)
e.init(mean_post + (Lp @ rng.standard_normal((D, N))).T)  # start in stationarity
e.run(5000, None, None, None)
T = 2000
p = torch.empty((T, N, D), dtype=torch.float64, device="cuda")
a = torch.empty((T, N), dtype=torch.uint8, device="cuda")
sums = torch.zeros(D, dtype=torch.float64, device="cuda"); outer = torch.zeros((D, D), dtype=torch.float64, device="cuda"); n = 0; ess = np.zeros(D)
for blk in range(10):
    e.run(T, p, None, a)
    f = p.reshape(-1, D); sums += f.sum(0); outer += f.T @ f; n += f.shape[0]; ess += ess_rhat_device(p)["ess"]
mean = (sums / n).cpu().numpy(); cov = (outer / n).cpu().numpy() - np.outer(mean, mean)
r = np.diag(cov) / np.diag(cov_post)
print(json.dumps(dict(check="control: GaussianRandomWalk with the fixed covariance 2.4^2/d * posterior covariance, started in stationarity, 4096 chains x 20000 recorded iterations", acc=float(a.float().mean()), ess_min=float(ess.min()), var_ratio_min=float(r.min()), var_ratio_max=float(r.max()), var_ratio_mean=float(r.mean()))))

# ---- the pooled-moments extension from the same pilot states: one covariance from all chains, refreshed every 100 steps ----
from tinyda_amd.distributed import PooledAdaptiveMetropolis

e = engine(0)
e.init(start)
pam = PooledAdaptiveMetropolis(e, 1e-4 * np.eye(D), t0=100, period=100)
for blk in range(10):
    pam.run(T, p, None, a)
sums.zero_(); outer.zero_(); n = 0
for blk in range(10):
    pam.run(T, p, None, a)
    f = p.reshape(-1, D); sums += f.sum(0); outer += f.T @ f; n += f.shape[0]
e.close()
mean = (sums / n).cpu().numpy(); cov = (outer / n).cpu().numpy() - np.outer(mean, mean)
r = np.diag(cov) / np.diag(cov_post)
print(json.dumps(dict(check="PooledAdaptiveMetropolis (extension) from the pilot states, 4096 chains x 20000 recorded iterations after 20000", acc=float(a.float().mean()),
                      var_ratio_min=float(r.min()), var_ratio_max=float(r.max()), var_ratio_mean=float(r.mean()))))
