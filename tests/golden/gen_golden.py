"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE (tinyDA) itself.

Run in the build container only (needs /root/reference):

    python tests/golden/gen_golden.py            # all fixtures
    python tests/golden/gen_golden.py g2_am_small  # one fixture

Every fixture is plain data: the problem definition (A, data, noise, prior, proposal settings),
the random variates the reference consumed, and what the reference produced (per-step
parameters / log-prior / log-likelihood / accept flags / adaptation state).  No reference
source text is stored.

How variates are pinned (SURVEY.md §8(c), §7 "hard parts"): tinyDA looks up np.random.* at call
time, so the functions it uses are replaced by a tap that
  * draws d legacy standard normals `z` and returns `mean + chol(C) @ z` for
    np.random.multivariate_normal (NumPy's own map is SVD based and not reproducible elsewhere;
    the Cholesky map has the same law), recording z;
  * records every np.random.random / uniform / normal / choice / randint result.
Chains are run one after the other exactly as tinyDA's sequential sampler does
(sampler.py:295-309, 335-368, 441-473) but each chain gets deep-copied posteriors so the
sequential-mode likelihood leak is not baked into the vectors.
"""
import copy
import os
import sys

import numpy as np
import scipy.stats as stats

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from _refload import load_reference  # noqa: E402

tda = load_reference()


# --------------------------------------------------------------------------------------
# variate tap
# --------------------------------------------------------------------------------------
class Tap:
    """Replaces the np.random entry points tinyDA calls; records what was handed out."""

    NAMES = ("multivariate_normal", "random", "uniform", "normal", "choice", "randint")

    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)
        self.log = []  # list of (kind, value)
        self.ulog = []  # (level, uniform) in consumption order
        self._saved = {}

    # -- the replacement functions -------------------------------------------------
    def multivariate_normal(self, mean, cov, size=None):
        assert size is None
        mean = np.asarray(mean, dtype=float)
        z = self.rs.standard_normal(mean.shape[0])
        L = np.linalg.cholesky(np.asarray(cov, dtype=float))
        self.log.append(("z", z.copy()))
        return mean + L @ z

    def random(self, size=None):
        assert size is None
        u = self.rs.random_sample()
        self.log.append(("u", u))
        # which sampler level asked?  (DAChain._sample_coarse -> 0, DAChain.sample -> 1,
        # MLDA.make_base_proposal -> 0, MLDA.make_mlda_proposal -> its level, MLDAChain.sample -> finest)
        fr = sys._getframe(1)
        name, owner = fr.f_code.co_name, fr.f_locals.get("self")
        if name in ("_sample_coarse", "make_base_proposal"):
            lvl = 0
        elif name == "make_mlda_proposal":
            lvl = owner.level
        elif name == "sample" and hasattr(owner, "chain_fine"):
            lvl = 1
        elif name == "sample" and hasattr(owner, "level"):
            lvl = owner.level
        else:
            lvl = 0
        self.ulog.append((lvl, u))
        return u

    def uniform(self, low=0.0, high=1.0, size=None):
        x = self.rs.random_sample(size)
        self.log.append(("uniform01", np.array(x, copy=True)))
        return low + (high - low) * x

    def normal(self, loc=0.0, scale=1.0, size=None):
        x = self.rs.standard_normal(size)
        self.log.append(("normal01", np.array(x, copy=True)))
        return loc + scale * x

    def choice(self, a, size=None, replace=True, p=None):
        r = self.rs.choice(a, size=size, replace=replace, p=p)
        self.log.append(("choice", np.array(r, copy=True)))
        return r

    def randint(self, low, high=None, size=None):
        r = self.rs.randint(low, high, size)
        self.log.append(("randint", np.array(r, copy=True)))
        return r

    # -- context manager -------------------------------------------------------------
    def __enter__(self):
        for n in self.NAMES:
            self._saved[n] = getattr(np.random, n)
            setattr(np.random, n, getattr(self, n))
        return self

    def __exit__(self, *exc):
        for n, f in self._saved.items():
            setattr(np.random, n, f)

    def take(self, kind):
        return [v for k, v in self.log if k == kind]


# --------------------------------------------------------------------------------------
# problem builders
# --------------------------------------------------------------------------------------
def linear_problem(seed, d, m, sigma=0.1, a_scale=None):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((m, d)) / (a_scale if a_scale else np.sqrt(d))
    theta_true = rng.standard_normal(d)
    y = A @ theta_true + sigma * rng.standard_normal(m)
    return A, theta_true, y


def make_model(A, b=None):
    if b is None:
        return lambda th: A @ th
    return lambda th: A @ th + b


def chain_trace(chain_links):
    th = np.array([l.parameters for l in chain_links])
    pr = np.array([l.prior for l in chain_links])
    ll = np.array([l.likelihood for l in chain_links])
    po = np.array([l.posterior for l in chain_links])
    return th, pr, ll, po


def run_mh(posterior, proposal, theta0, iterations, n_chains, seed, snapshot=None, zkind="z"):
    """Run tinyDA.Chain for each chain; returns stacked traces [chain][step]."""
    out = {k: [] for k in ("theta", "logprior", "loglike", "logpost", "accepted", "z", "u")}
    snaps = []
    for c in range(n_chains):
        post = copy.deepcopy(posterior)
        prop = copy.deepcopy(proposal)
        with Tap(seed + 1000 * c) as tap:
            ch = tda.Chain(post, prop, theta0[c].copy())
            if snapshot is None:
                ch.sample(iterations, progressbar=False)
            else:
                period = snapshot["period"]
                done = 0
                snap_c = []
                while done < iterations:
                    n = min(period, iterations - done)
                    ch.sample(n, progressbar=False)
                    done += n
                    snap_c.append(snapshot["fn"](ch.proposal))
                snaps.append(snap_c)
        th, pr, ll, po = chain_trace(ch.chain)
        out["theta"].append(th)
        out["logprior"].append(pr)
        out["loglike"].append(ll)
        out["logpost"].append(po)
        out["accepted"].append(np.array(ch.accepted, dtype=np.uint8))
        out["z"].append(np.array(tap.take(zkind)))
        out["u"].append(np.array(tap.take("u")))
    res = {k: np.array(v) for k, v in out.items()}
    return res, snaps


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %-28s %7.1f KiB" % (name + ".npz", os.path.getsize(path) / 1024))


# --------------------------------------------------------------------------------------
# fixtures
# --------------------------------------------------------------------------------------
def g1_basic_sampler():
    """BASELINE config 1 (examples/Basic Sampler.ipynb): 2-parameter linear regression,
    GaussianRandomWalk(C=I, scaling=0.1, adaptive=True), isotropic likelihood."""
    rs = np.random.RandomState(0)
    x = np.linspace(0, 1, 50)
    y = 1.0 + 2.0 * x + rs.normal(0, 0.2, 50)
    A = np.stack([np.ones_like(x), x], axis=1)
    prior = stats.multivariate_normal(np.zeros(2), np.eye(2))
    like = tda.GaussianLogLike(y, 0.04 * np.eye(50))
    assert type(like).__name__ == "IsotropicGaussianLogLike"
    post = tda.Posterior(prior, like, make_model(A))
    prop = tda.GaussianRandomWalk(C=np.eye(2), scaling=0.1, adaptive=True, gamma=1.01, period=50)
    theta0 = np.array([[0.3, -0.2], [1.5, 0.7]])
    res, snaps = run_mh(post, prop, theta0, 300, 2, seed=11,
                        snapshot={"period": 50, "fn": lambda p: (p.scaling, p.k, p.t)})
    save("g1_basic_sampler", A=A, data=y, noise_var=np.array(0.04), prior_mean=np.zeros(2),
         prior_cov=np.eye(2), C=np.eye(2), scaling0=np.array(0.1), gamma=np.array(1.01),
         period=np.array(50), theta0=theta0, scaling_hist=np.array(snaps)[:, :, 0],
         **res)


def g2_am(name, d, m, n_chains, iters, t0, period, seed, c0=1e-2, adaptive=False,
          noise="iso", prior_kind="identity", sd=None, eps=1e-6, sigma=0.1):
    A, theta_true, y = linear_problem(seed, d, m, sigma=sigma)
    rng = np.random.default_rng(seed + 7)
    if prior_kind == "identity":
        pm, pc = np.zeros(d), np.eye(d)
    else:
        B = rng.standard_normal((d, d)) / np.sqrt(d)
        pc = B @ B.T + 0.5 * np.eye(d)
        pm = 0.1 * rng.standard_normal(d)
    prior = stats.multivariate_normal(pm, pc)
    if noise == "iso":
        cov = sigma ** 2 * np.eye(m)
    elif noise == "diag":
        cov = np.diag(sigma ** 2 * (0.5 + rng.random(m)))
    else:
        Lc = sigma * np.eye(m) + 0.01 * np.tril(rng.standard_normal((m, m)))
        cov = Lc @ Lc.T
    like = tda.GaussianLogLike(y, cov)
    post = tda.Posterior(prior, like, make_model(A))
    C0 = c0 * np.eye(d)
    prop = tda.AdaptiveMetropolis(C0=C0, sd=sd, epsilon=eps, t0=t0, period=period,
                                  adaptive=adaptive, gamma=1.01)
    theta0 = theta_true[None, :] + 0.05 * rng.standard_normal((n_chains, d))

    def snap(p):
        return (p.C.copy(), p.AM_recursor.get_mu().copy(), p.AM_recursor.get_sigma().copy(),
                float(p.scaling), int(p.t))

    res, snaps = run_mh(post, prop, theta0, iters, n_chains, seed=seed + 100,
                        snapshot={"period": period, "fn": snap})
    C_hist = np.array([[s[0] for s in sc] for sc in snaps])
    mu_hist = np.array([[s[1] for s in sc] for sc in snaps])
    sig_hist = np.array([[s[2] for s in sc] for sc in snaps])
    scal_hist = np.array([[s[3] for s in sc] for sc in snaps])
    save(name, A=A, data=y, noise_kind=np.array(noise), noise_cov=cov if noise == "dense" else np.diag(cov),
         prior_mean=pm, prior_cov=pc, C0=C0, sd=np.array(prop.sd), epsilon=np.array(eps),
         t0=np.array(t0), period=np.array(period), adaptive=np.array(adaptive), gamma=np.array(1.01),
         theta0=theta0, C_hist=C_hist, mu_hist=mu_hist, sigma_hist=sig_hist,
         scaling_hist=scal_hist, like_class=np.array(type(like).__name__), **res)


def g2b_pcn():
    d, m, n_chains, iters = 8, 16, 4, 200
    A, theta_true, y = linear_problem(21, d, m, sigma=0.2)
    rng = np.random.default_rng(22)
    B = rng.standard_normal((d, d)) / np.sqrt(d)
    pc = B @ B.T + 0.5 * np.eye(d)
    pm = np.zeros(d)
    prior = stats.multivariate_normal(pm, pc)
    like = tda.GaussianLogLike(y, 0.04 * np.eye(m))
    post = tda.Posterior(prior, like, make_model(A))
    prop = tda.CrankNicolson(scaling=0.15, adaptive=True, gamma=1.02, period=40)
    theta0 = 0.3 * rng.standard_normal((n_chains, d))
    res, snaps = run_mh(post, prop, theta0, iters, n_chains, seed=230,
                        snapshot={"period": 40, "fn": lambda p: float(p.scaling)})
    save("g2b_pcn", A=A, data=y, noise_var=np.array(0.04), prior_mean=pm, prior_cov=pc,
         scaling0=np.array(0.15), gamma=np.array(1.02), period=np.array(40), theta0=theta0,
         scaling_hist=np.array(snaps), **res)


def g13_owcn(name, adaptive):
    """OperatorWeightedCrankNicolson (proposal.py:515-605) on a linear-Gaussian posterior with a general prior covariance."""
    d, m, n_chains, iters = 8, 16, 4, 150
    A, theta_true, y = linear_problem(131, d, m, sigma=0.2)
    rng = np.random.default_rng(132)
    R = rng.standard_normal((d, d)) / np.sqrt(d)
    pc = R @ R.T + 0.5 * np.eye(d)
    pm = np.zeros(d)
    Q, _ = np.linalg.qr(rng.standard_normal((d, d)))
    Bop = Q @ np.diag(np.linspace(0.05, 0.6, d)) @ Q.T  # symmetric, spectrum inside (0, 1 / scaling)
    prior = stats.multivariate_normal(pm, pc)
    post = tda.Posterior(prior, tda.GaussianLogLike(y, 0.04 * np.eye(m)), make_model(A))
    prop = tda.OperatorWeightedCrankNicolson(Bop, scaling=0.4, adaptive=adaptive, gamma=1.02, period=30)
    theta0 = 0.3 * rng.standard_normal((n_chains, d))
    res, snaps = run_mh(post, prop, theta0, iters, n_chains, seed=1330,
                        snapshot={"period": 30, "fn": lambda p: float(p.scaling)})
    save(name, A=A, data=y, noise_var=np.array(0.04), prior_mean=pm, prior_cov=pc, B=Bop, scaling0=np.array(0.4),
         adaptive=np.array(adaptive), gamma=np.array(1.02), period=np.array(30), theta0=theta0, scaling_hist=np.array(snaps), **res)


class _LinearModelWithGradient:
    """Forward model with the `gradient(parameters, sensitivity)` method MALA looks for (proposal.py:938-943, :996-998)."""

    def __init__(self, A):
        self.A = A

    def __call__(self, th):
        return self.A @ th

    def gradient(self, th, sensitivity):
        return self.A.T @ sensitivity


def g14_mala(name, adaptive, cov_kind="iso"):
    """MALA (proposal.py:861-1005) with the exact gradient of a linear-Gaussian posterior, general prior covariance."""
    d, m, n_chains, iters = 8, 16, 4, 150
    A, theta_true, y = linear_problem(141, d, m, sigma=0.2)
    rng = np.random.default_rng(142)
    R = rng.standard_normal((d, d)) / np.sqrt(d)
    pc = R @ R.T + 0.5 * np.eye(d)
    pm = 0.1 * rng.standard_normal(d)
    if cov_kind == "iso":
        cov = 0.04 * np.eye(m)
    else:
        Ln = 0.2 * np.eye(m) + 0.03 * np.tril(rng.standard_normal((m, m)))
        cov = Ln @ Ln.T
    prior = stats.multivariate_normal(pm, pc)
    post = tda.Posterior(prior, tda.GaussianLogLike(y, cov), _LinearModelWithGradient(A))
    prop = tda.MALA(scaling=0.12, adaptive=adaptive, gamma=1.02, period=30)
    theta0 = theta_true[None] + 0.1 * rng.standard_normal((n_chains, d))
    res, snaps = run_mh(post, prop, theta0, iters, n_chains, seed=1430, zkind="normal01",
                        snapshot={"period": 30, "fn": lambda p: float(p.scaling)})
    save(name, A=A, data=y, noise_cov=cov, prior_mean=pm, prior_cov=pc, scaling0=np.array(0.12), adaptive=np.array(adaptive),
         gamma=np.array(1.02), period=np.array(30), theta0=theta0, scaling_hist=np.array(snaps), **res)


def g3_loglike_kats():
    rng = np.random.default_rng(31)
    m = 12
    data = rng.standard_normal(m)
    X = rng.standard_normal((6, m))
    iso = tda.GaussianLogLike(data, 0.3 * np.eye(m))
    dg = np.diag(0.1 + rng.random(m))
    diag = tda.GaussianLogLike(data, dg)
    Lc = 0.5 * np.eye(m) + 0.2 * np.tril(rng.standard_normal((m, m)))
    dense_cov = Lc @ Lc.T
    dense = tda.GaussianLogLike(data, dense_cov)
    names = np.array([type(o).__name__ for o in (iso, diag, dense)])
    ada = tda.AdaptiveGaussianLogLike(data, dense_cov)
    out_ada0 = np.array([ada.loglike(x) for x in X])
    bias = 0.1 * rng.standard_normal(m)
    Bc = 0.05 * rng.standard_normal((m, m))
    bias_cov = Bc @ Bc.T
    ada.set_bias(bias, bias_cov)
    out_ada1 = np.array([ada.loglike(x) for x in X])
    custom = 0.2 * rng.standard_normal(m)
    out_ada_custom = np.array([ada.loglike_custom_bias(x, custom) for x in X])
    # threshold case: every entry < 1e-9 -> inverse NOT refreshed (distributions.py:399-402)
    ada2 = tda.AdaptiveGaussianLogLike(data, dense_cov)
    tiny_cov = 1e-10 * np.ones((m, m))
    ada2.set_bias(bias, tiny_cov)
    out_ada_tiny = np.array([ada2.loglike(x) for x in X])
    # mixed case: one entry above threshold -> refreshed
    mixed_cov = tiny_cov.copy()
    mixed_cov[0, 0] = 1e-3
    ada2.set_bias(bias, mixed_cov)
    out_ada_mixed = np.array([ada2.loglike(x) for x in X])
    save("g3_loglike_kats", data=data, X=X, iso_var=np.array(0.3), diag_cov=dg, dense_cov=dense_cov,
         class_names=names,
         out_iso=np.array([iso.loglike(x) for x in X]),
         out_diag=np.array([diag.loglike(x) for x in X]),
         out_dense=np.array([dense.loglike(x) for x in X]),
         grad_iso=np.array([iso.grad_loglike(x) for x in X]),
         grad_diag=np.array([diag.grad_loglike(x) for x in X]),
         grad_dense=np.array([dense.grad_loglike(x) for x in X]),
         bias=bias, bias_cov=bias_cov, custom_bias=custom, tiny_cov=tiny_cov, mixed_cov=mixed_cov,
         out_ada0=out_ada0, out_ada1=out_ada1, out_ada_custom=out_ada_custom,
         out_ada_tiny=out_ada_tiny, out_ada_mixed=out_ada_mixed)


def g7_moments():
    rng = np.random.default_rng(71)
    d, n = 5, 40
    X = rng.standard_normal((n, d)) * np.array([1.0, 0.5, 2.0, 0.1, 1.5]) + 0.3
    r = tda.RecursiveSampleMoments(X[0].copy(), np.zeros((d, d)), sd=0.7, epsilon=1e-6)
    mus, sigs = [], []
    for x in X[1:]:
        r.update(x)
        mus.append(r.get_mu().copy())
        sigs.append(r.get_sigma().copy())
    r0 = tda.RecursiveSampleMoments(X[0].copy(), np.zeros((d, d)))
    for x in X[1:]:
        r0.update(x)
    z = tda.chain.ZeroMeanRecursiveSampleMoments(np.zeros((d, d)))
    zs = []
    for x in X:
        z.update(x)
        zs.append(z.get_sigma().copy())
    save("g7_moments", X=X, sd=np.array(0.7), epsilon=np.array(1e-6), mu_hist=np.array(mus),
         sigma_hist=np.array(sigs), plain_mu=r0.get_mu(), plain_sigma=r0.get_sigma(),
         np_cov=np.cov(X.T), zero_mean_hist=np.array(zs))


def g9_mvn_logpdf():
    rng = np.random.default_rng(91)
    d = 7
    B = rng.standard_normal((d, d))
    cov = B @ B.T / d + 0.2 * np.eye(d)
    mean = rng.standard_normal(d)
    X = rng.standard_normal((9, d))
    frozen = stats.multivariate_normal(mean, cov)
    ident = stats.multivariate_normal(np.zeros(d), np.eye(d))
    save("g9_mvn_logpdf", mean=mean, cov=cov, X=X, logpdf=frozen.logpdf(X),
         logpdf_identity=ident.logpdf(X))


def _ml_problem(seed, d, ms, sigma, prior_kind="identity"):
    """Levels share theta*, each has its own operator (fine operator + level-dependent perturbation) and data."""
    rng = np.random.default_rng(seed)
    theta_true = rng.standard_normal(d)
    Afine = rng.standard_normal((max(ms), d)) / np.sqrt(d)
    As, ys = [], []
    for k, m in enumerate(ms):
        pert = 0.05 * (len(ms) - 1 - k) * rng.standard_normal((m, d)) / np.sqrt(d)
        A = Afine[:m] + pert
        As.append(A)
        ys.append(Afine[:m] @ theta_true + sigma * rng.standard_normal(m))
    if prior_kind == "identity":
        pm, pc = np.zeros(d), np.eye(d)
    else:
        B = rng.standard_normal((d, d)) / np.sqrt(d)
        pm, pc = np.zeros(d), B @ B.T + 0.5 * np.eye(d)
    return As, ys, theta_true, pm, pc


def _split_uniforms(ulog, n_levels, counts):
    """per-level arrays of the uniforms each level consumed, NaN where a step consumed none is handled by
    the caller (which knows from the accept history which steps evaluated)."""
    out = [[] for _ in range(n_levels)]
    for lvl, u in ulog:
        out[lvl].append(u)
    return out


def _place(consumed, evaluated):
    """consumed: uniforms in order; evaluated: bool per step -> array with NaN at steps that drew nothing."""
    arr = np.full(len(evaluated), np.nan)
    it = iter(consumed)
    for i, ev in enumerate(evaluated):
        if ev:
            arr[i] = next(it)
    assert next(it, None) is None
    return arr


def _level_covs(seed, ms, sigma, noise):
    """observation covariance per level: noise[k] in 'iso' | 'diag' | 'dense' (the dense ones as g2_am builds them)"""
    rng = np.random.default_rng(seed + 31)
    covs = []
    for m, kind in zip(ms, noise):
        if kind == "iso":
            covs.append(sigma ** 2 * np.eye(m))
        elif kind == "diag":
            covs.append(np.diag(sigma ** 2 * (0.5 + rng.random(m))))
        else:
            Lc = sigma * np.eye(m) + 0.02 * np.tril(rng.standard_normal((m, m)))
            covs.append(Lc @ Lc.T)
    return covs


def g4_da(name, proposal_kind, d=6, ms=(10, 24), L=4, iters=60, n_chains=4, seed=401, randomize=False,
          adaptive=False, period=7, prior_kind="identity", noise=None):
    sigma = 0.2
    As, ys, theta_true, pm, pc = _ml_problem(seed, d, ms, sigma, prior_kind)
    prior = stats.multivariate_normal(pm, pc)
    covs = _level_covs(seed, ms, sigma, noise or ["iso"] * len(ms))
    posts = [tda.Posterior(prior, tda.GaussianLogLike(y, cv), make_model(A)) for A, y, cv in zip(As, ys, covs)]
    rng = np.random.default_rng(seed + 1)
    theta0 = theta_true[None] + 0.2 * rng.standard_normal((n_chains, d))
    if proposal_kind == "pcn":
        prop = tda.CrankNicolson(scaling=0.25, adaptive=adaptive, gamma=1.02, period=period)
        pcfg = dict(kind="pcn", scaling=0.25, adaptive=adaptive, gamma=1.02, period=period)
    elif proposal_kind == "grw":
        prop = tda.GaussianRandomWalk(0.02 * np.eye(d), scaling=1.0, adaptive=adaptive, gamma=1.02, period=period)
        pcfg = dict(kind="grw", C=0.02 * np.eye(d), scaling=1.0, adaptive=adaptive, gamma=1.02, period=period)
    else:
        prop = tda.AdaptiveMetropolis(0.02 * np.eye(d), t0=10, period=period, adaptive=adaptive, gamma=1.02)
        pcfg = dict(kind="am", C0=0.02 * np.eye(d), t0=10, period=period, adaptive=adaptive, gamma=1.02,
                    sd=float(prop.sd), epsilon=1e-6)
    out = {k: [] for k in ("z", "u0", "u1", "ridx", "th0", "lp0", "ll0", "acc0", "th1", "lp1", "ll1", "acc1", "scaling")}
    for c in range(n_chains):
        with Tap(seed + 50 * c) as tap:
            ch = tda.DAChain(copy.deepcopy(posts[0]), copy.deepcopy(posts[1]), copy.deepcopy(prop), L,
                             randomize_subchain_length=randomize, initial_parameters=theta0[c].copy())
            ch.sample(iters, progressbar=False)
        loc = np.array(ch.is_coarse, dtype=bool)
        cth, clp, cll, _ = chain_trace(ch.chain_coarse)
        fth, flp, fll, _ = chain_trace(ch.chain_fine)
        acc_c = np.array(ch.accepted_coarse, dtype=np.uint8)
        out["th0"].append(cth[loc]); out["lp0"].append(clp[loc]); out["ll0"].append(cll[loc]); out["acc0"].append(acc_c[loc])
        out["th1"].append(fth); out["lp1"].append(flp); out["ll1"].append(fll)
        out["acc1"].append(np.array(ch.accepted_fine, dtype=np.uint8))
        us = _split_uniforms(tap.ulog, 2, None)
        out["z"].append(np.array(tap.take("z")))
        out["u0"].append(np.array(us[0]))
        evaluated = acc_c[loc].reshape(iters, L).sum(axis=1) > 0  # fine level draws only after a coarse accept
        out["u1"].append(_place(us[1], evaluated))
        r = tap.take("randint")
        out["ridx"].append(_place([int(x) for x in r], evaluated) if randomize else np.full(iters, -1.0))
        out["scaling"].append(float(ch.proposal.scaling))
    arrays = {k: np.array(v) for k, v in out.items()}
    flat = {}
    for k, v in pcfg.items():
        flat["prop_" + k] = np.array(v)
    extra = {"cov%d" % k: cv for k, cv in enumerate(covs)} if noise else {}
    save(name, A0=As[0], A1=As[1], y0=ys[0], y1=ys[1], noise_var=np.array(sigma ** 2), prior_mean=pm, prior_cov=pc,
         theta0=theta0, subchain_length=np.array(L), randomize=np.array(randomize), **extra, **flat, **arrays)


def g5_mlda(name, proposal_kind, d=6, ms=(8, 14, 24), sl=(3, 2), iters=40, n_chains=4, seed=501, adaptive=False,
            period=7, noise=None):
    sigma = 0.2
    As, ys, theta_true, pm, pc = _ml_problem(seed, d, ms, sigma)
    prior = stats.multivariate_normal(pm, pc)
    covs = _level_covs(seed, ms, sigma, noise or ["iso"] * len(ms))
    posts = [tda.Posterior(prior, tda.GaussianLogLike(y, cv), make_model(A)) for A, y, cv in zip(As, ys, covs)]
    rng = np.random.default_rng(seed + 1)
    theta0 = theta_true[None] + 0.2 * rng.standard_normal((n_chains, d))
    if proposal_kind == "grw":
        prop = tda.GaussianRandomWalk(0.02 * np.eye(d), scaling=1.0, adaptive=adaptive, gamma=1.02, period=period)
        pcfg = dict(kind="grw", C=0.02 * np.eye(d), scaling=1.0, adaptive=adaptive, gamma=1.02, period=period)
    else:
        prop = tda.AdaptiveMetropolis(0.02 * np.eye(d), t0=10, period=period, adaptive=adaptive, gamma=1.02)
        pcfg = dict(kind="am", C0=0.02 * np.eye(d), t0=10, period=period, adaptive=adaptive, gamma=1.02,
                    sd=float(prop.sd), epsilon=1e-6)
    nl = len(ms)
    out = {"z": [], "scaling": []}
    for k in range(nl):
        for key in ("u", "th", "lp", "ll", "acc"):
            out["%s%d" % (key, k)] = []
    for c in range(n_chains):
        with Tap(seed + 50 * c) as tap:
            ch = tda.MLDAChain([copy.deepcopy(p) for p in posts], copy.deepcopy(prop), list(sl),
                               initial_parameters=theta0[c].copy())
            ch.sample(iters, progressbar=False)
        us = _split_uniforms(tap.ulog, nl, None)
        out["z"].append(np.array(tap.take("z")))
        objs = [ch]
        cur = ch.proposal
        while True:
            objs.append(cur)
            if cur.level == 0:
                break
            cur = cur.proposal
        objs = objs[::-1]  # level 0 first
        local_acc = []
        for k, ob in enumerate(objs):
            th, lp, ll, _ = chain_trace(ob.chain)
            acc = np.array(ob.accepted, dtype=np.uint8)
            if k < nl - 1:
                loc = np.array(ob.is_local, dtype=bool)
                th, lp, ll, acc = th[loc], lp[loc], ll[loc], acc[loc]
            out["th%d" % k].append(th); out["lp%d" % k].append(lp); out["ll%d" % k].append(ll); out["acc%d" % k].append(acc)
            local_acc.append(acc)
        out["u0"].append(np.array(us[0]))
        for k in range(1, nl):
            below = local_acc[k - 1]
            nsteps = len(local_acc[k]) - (1 if k == nl - 1 else 0)
            evaluated = below.reshape(nsteps, sl[k - 1]).sum(axis=1) > 0
            out["u%d" % k].append(_place(us[k], evaluated))
        out["scaling"].append(float(objs[0].proposal.scaling))
    arrays = {k: np.array(v) for k, v in out.items()}
    flat = {"prop_" + k: np.array(v) for k, v in pcfg.items()}
    lv = {}
    for k in range(nl):
        lv["A%d" % k] = As[k]
        lv["y%d" % k] = ys[k]
        if noise:
            lv["cov%d" % k] = covs[k]
    save(name, noise_var=np.array(sigma ** 2), prior_mean=pm, prior_cov=pc, theta0=theta0,
         subchain_lengths=np.array(sl), n_levels=np.array(nl), **lv, **flat, **arrays)


def rosenbrock_model(a=1.0, b=10.0):
    """d-dimensional chain of the notebook's 2-D Rosenbrock 'forward model' (examples/MALA Rosenbrock.ipynb):
    one scalar output, data [0], unit variance -> loglike = -f(theta)^2 / 2."""
    return lambda th: np.array([np.sum((a - th[:-1]) ** 2 + b * (th[1:] - th[:-1] ** 2) ** 2)])


def g6_dreamz(name, problem, d, M0, delta, nCR, adaptive, period, iters, n_chains, seed, b=5e-2, b_star=1e-6, m=12, noise="iso"):
    rng = np.random.default_rng(seed)
    pm, pc = np.zeros(d), np.eye(d)
    prior = stats.multivariate_normal(pm, pc)
    extra = {}
    if problem == "linear":
        A, theta_true, y = linear_problem(seed, d, m, sigma=0.3)
        cov = _level_covs(seed, [m], 0.3, [noise])[0]
        like = tda.GaussianLogLike(y, cov)
        post = tda.Posterior(prior, like, make_model(A))
        extra = dict(A=A, data=y, noise_var=np.array(0.09))
        if noise != "iso":
            extra["noise_cov"] = cov
    else:
        like = tda.GaussianLogLike(np.zeros(1), np.eye(1))
        post = tda.Posterior(prior, like, rosenbrock_model(1.0, 10.0))
        extra = dict(rosen_a=np.array(1.0), rosen_b=np.array(10.0), data=np.zeros(1), noise_var=np.array(1.0))
    theta0 = 0.5 * rng.standard_normal((n_chains, d))
    keys = ("Z0", "r", "mcr", "sub_u", "forced", "e_u", "eps_n", "u", "theta", "logprior", "loglike", "accepted", "pCR",
            "scaling")
    out = {k: [] for k in keys}
    for c in range(n_chains):
        np.random.seed(seed + 7 * c)  # scipy's prior.rvs inside setup_proposal draws from the global stream
        prop = tda.DREAMZ(M0, delta=delta, b=b, b_star=b_star, Z_method="random", nCR=nCR, adaptive=adaptive,
                          gamma=1.02, period=period)
        with Tap(seed + 50 * c) as tap:
            ch = tda.Chain(copy.deepcopy(post), prop, theta0[c].copy())
            out["Z0"].append(np.array(ch.proposal.Z, copy=True))
            ch.sample(iters, progressbar=False)
        # parse the sequential log: per step  delta x choice(2) , choice() , uniform(d) , [choice()] , uniform(d) , normal(d) , u
        it = iter(tap.log)
        R, MC, SU, FO, EU, EN, U = [], [], [], [], [], [], []
        for _ in range(iters):
            rr = []
            for _i in range(delta):
                kind, v = next(it)
                assert kind == "choice" and np.size(v) == 2
                rr.append(np.array(v))
            kind, v = next(it)
            assert kind == "choice" and np.size(v) == 1
            MC.append(int(v))
            kind, v = next(it)
            assert kind == "uniform01"
            SU.append(v)
            kind, v = next(it)
            if kind == "choice":
                FO.append(int(v))
                kind, v = next(it)
            else:
                FO.append(-1)
            assert kind == "uniform01"
            EU.append(v)
            kind, v = next(it)
            assert kind == "normal01"
            EN.append(v)
            kind, v = next(it)
            assert kind == "u"
            U.append(v)
            R.append(np.array(rr))
        assert next(it, None) is None
        th, lp, ll, _ = chain_trace(ch.chain)
        for k, v in zip(("r", "mcr", "sub_u", "forced", "e_u", "eps_n", "u", "theta", "logprior", "loglike", "accepted", "pCR", "scaling"),
                        (R, MC, SU, FO, EU, EN, U, th, lp, ll, np.array(ch.accepted, dtype=np.uint8), ch.proposal.pCR,
                         float(ch.proposal.scaling))):
            out[k].append(np.array(v))
    save(name, problem=np.array(problem), prior_mean=pm, prior_cov=pc, theta0=theta0, M0=np.array(M0), delta=np.array(delta),
         nCR=np.array(nCR), adaptive=np.array(adaptive), period=np.array(period), gamma=np.array(1.02), b=np.array(b),
         b_star=np.array(b_star), **extra, **{k: np.array(v) for k, v in out.items()})



def g15_hier_dreamz(name, ms, sl, d=5, M0=24, delta=1, nCR=3, adaptive=True, period=6, iters=30, n_chains=3, seed=1501, randomize=False,
                    aem=None):
    """DREAMZ as the base proposal of Delayed Acceptance (2 levels, DAChain) / MLDA (3 levels, MLDAChain): the configuration of
    the reference's own MLDA notebook (examples/Multilevel Delayed Acceptance.ipynb cells 20-23)."""
    sigma = 0.2
    nl = len(ms)
    bs = None
    if aem is None:
        As, ys, theta_true, pm, pc = _ml_problem(seed, d, ms, sigma)
        prior = stats.multivariate_normal(pm, pc)
        posts = [tda.Posterior(prior, tda.GaussianLogLike(y, sigma ** 2 * np.eye(len(y))), make_model(A)) for A, y in zip(As, ys)]
    else:  # adaptive error model: the levels share the observation vector, the coarse likelihoods are adaptive (chain.py:268-305)
        assert len(set(ms)) == 1
        As, bs, y, theta_true = _aem_problem(seed, d, ms[0], nl, sigma)
        ys = [y] * nl
        pm, pc = np.zeros(d), np.eye(d)
        prior = stats.multivariate_normal(pm, pc)
        cov = sigma ** 2 * np.eye(ms[0])
        posts = [tda.Posterior(prior, tda.AdaptiveGaussianLogLike(y, cov) if k < nl - 1 else tda.GaussianLogLike(y, cov),
                               make_model(As[k], bs[k])) for k in range(nl)]
    rng = np.random.default_rng(seed + 1)
    theta0 = theta_true[None] + 0.2 * rng.standard_normal((n_chains, d))
    out = {k: [] for k in ("Z0", "r", "mcr", "sub_u", "forced", "e_u", "eps_n", "u0", "pCR", "scaling", "ridx")}
    for k in range(nl):
        for key in ("th", "lp", "ll", "acc"):
            out["%s%d" % (key, k)] = []
        if k:
            out["u%d" % k] = []
    for c in range(n_chains):
        np.random.seed(seed + 7 * c)  # prior.rvs(M0) inside DREAMZ.setup_proposal draws from the global stream
        prop = tda.DREAMZ(M0, delta=delta, Z_method="random", nCR=nCR, adaptive=adaptive, gamma=1.02, period=period)
        with Tap(seed + 50 * c) as tap:
            if nl == 2:
                ch = tda.DAChain(copy.deepcopy(posts[0]), copy.deepcopy(posts[1]), prop, sl[0], randomize_subchain_length=randomize,
                                 initial_parameters=theta0[c].copy(), adaptive_error_model=aem)
            else:
                ch = tda.MLDAChain([copy.deepcopy(p) for p in posts], prop, list(sl), initial_parameters=theta0[c].copy(),
                                   adaptive_error_model=aem)
            base = ch.proposal if nl == 2 else None
            ch.sample(iters, progressbar=False)
        if nl == 2:
            base_prop = ch.proposal
            loc = np.array(ch.is_coarse, dtype=bool)
            cth, clp, cll, _ = chain_trace(ch.chain_coarse)
            acc_c = np.array(ch.accepted_coarse, dtype=np.uint8)
            traces = [(cth[loc], clp[loc], cll[loc], acc_c[loc])]
            fth, flp, fll, _ = chain_trace(ch.chain_fine)
            traces.append((fth, flp, fll, np.array(ch.accepted_fine, dtype=np.uint8)))
        else:
            objs = [ch]
            cur = ch.proposal
            while True:
                objs.append(cur)
                if cur.level == 0:
                    break
                cur = cur.proposal
            objs = objs[::-1]
            base_prop = objs[0].proposal
            traces = []
            for k, ob in enumerate(objs):
                th, lp, ll, _ = chain_trace(ob.chain)
                acc = np.array(ob.accepted, dtype=np.uint8)
                if k < nl - 1:
                    lc = np.array(ob.is_local, dtype=bool)
                    th, lp, ll, acc = th[lc], lp[lc], ll[lc], acc[lc]
                traces.append((th, lp, ll, acc))
        out["Z0"].append(np.array(base_prop.Z[:M0], copy=True))
        # the sequential log: a base step is  delta x choice(2), choice(), uniform(d), [choice()], uniform(d), normal(d), u ;
        # a bare u between base steps belongs to an upper level (kept per level in tap.ulog)
        it = iter(tap.log)
        R, MC, SU, FO, EU, EN, U = [], [], [], [], [], [], []
        for kind, v in it:
            if kind in ("u", "randint"):  # an upper level's uniform / the promoted index of a randomised subchain (chain.py:525-527)
                continue
            assert kind == "choice" and np.size(v) == 2
            rr = [np.array(v)]
            for _i in range(delta - 1):
                kind, v = next(it)
                rr.append(np.array(v))
            kind, v = next(it)
            assert kind == "choice" and np.size(v) == 1
            MC.append(int(v))
            kind, v = next(it)
            assert kind == "uniform01"
            SU.append(v)
            kind, v = next(it)
            if kind == "choice":
                FO.append(int(v))
                kind, v = next(it)
            else:
                FO.append(-1)
            assert kind == "uniform01"
            EU.append(v)
            kind, v = next(it)
            assert kind == "normal01"
            EN.append(v)
            kind, v = next(it)
            assert kind == "u"
            U.append(v)
            R.append(np.array(rr))
        n_base = iters * int(np.prod(sl))
        assert len(U) == n_base, (len(U), n_base)
        us = _split_uniforms(tap.ulog, nl, None)
        assert np.array_equal(np.array(us[0]), np.array(U))
        for key, v in zip(("r", "mcr", "sub_u", "forced", "e_u", "eps_n", "u0"), (R, MC, SU, FO, EU, EN, U)):
            out[key].append(np.array(v))
        for k, (th, lp, ll, acc) in enumerate(traces):
            out["th%d" % k].append(th); out["lp%d" % k].append(lp); out["ll%d" % k].append(ll); out["acc%d" % k].append(acc)
        for k in range(1, nl):
            below = traces[k - 1][3]
            nsteps = len(traces[k][3]) - (1 if k == nl - 1 else 0)
            evaluated = below.reshape(nsteps, sl[k - 1]).sum(axis=1) > 0
            out["u%d" % k].append(_place(us[k], evaluated))
            if k == 1:
                rr_ = tap.take("randint")
                out["ridx"].append(_place([int(x) for x in rr_], evaluated) if randomize else np.full(nsteps, -1.0))
        out["pCR"].append(np.array(base_prop.pCR))
        out["scaling"].append(float(base_prop.scaling))
    lv = {}
    for k in range(nl):
        lv["A%d" % k] = As[k]
        lv["y%d" % k] = ys[k]
        if bs is not None:
            lv["b%d" % k] = bs[k]
    if aem is not None:
        lv["aem"] = np.array(aem)
    save(name, noise_var=np.array(sigma ** 2), prior_mean=pm, prior_cov=pc, theta0=theta0, subchain_lengths=np.array(sl),
         n_levels=np.array(nl), M0=np.array(M0), delta=np.array(delta), nCR=np.array(nCR), adaptive=np.array(adaptive),
         period=np.array(period), gamma=np.array(1.02), b=np.array(5e-2), b_star=np.array(1e-6), randomize=np.array(randomize), **lv,
         **{k: np.array(v) for k, v in out.items()})


def _aem_problem(seed, d, m, n_levels, sigma):
    """Levels share the observation vector (AEM subtracts model outputs of adjacent levels, chain.py:274-276)."""
    rng = np.random.default_rng(seed)
    theta_true = rng.standard_normal(d)
    Afine = rng.standard_normal((m, d)) / np.sqrt(d)
    y = Afine @ theta_true + sigma * rng.standard_normal(m)
    As = [Afine + 0.15 * (n_levels - 1 - k) * rng.standard_normal((m, d)) / np.sqrt(d) for k in range(n_levels)]
    bs = [0.1 * (n_levels - 1 - k) * rng.standard_normal(m) for k in range(n_levels)]
    return As, bs, y, theta_true


def g8_da_aem(name, aem, proposal_kind="grw", d=4, m=8, L=3, iters=50, n_chains=4, seed=801, prop_var=0.05, beta=0.3):
    sigma = 0.3
    As, bs, y, theta_true = _aem_problem(seed, d, m, 2, sigma)
    pm, pc = np.zeros(d), np.eye(d)
    prior = stats.multivariate_normal(pm, pc)
    cov = sigma ** 2 * np.eye(m)
    post_c = tda.Posterior(prior, tda.AdaptiveGaussianLogLike(y, cov), make_model(As[0], bs[0]))
    post_f = tda.Posterior(prior, tda.GaussianLogLike(y, cov), make_model(As[1], bs[1]))
    rng = np.random.default_rng(seed + 1)
    theta0 = theta_true[None] + 0.2 * rng.standard_normal((n_chains, d))
    if proposal_kind == "grw":
        prop = tda.GaussianRandomWalk(prop_var * np.eye(d), scaling=1.0)
        pcfg = dict(kind="grw", C=prop_var * np.eye(d), scaling=1.0, adaptive=False, gamma=1.01, period=100)
    else:
        prop = tda.CrankNicolson(scaling=beta)
        pcfg = dict(kind="pcn", scaling=beta, adaptive=False, gamma=1.01, period=100)
    out = {k: [] for k in ("z", "u0", "u1", "th0", "lp0", "ll0", "acc0", "th1", "lp1", "ll1", "acc1", "bias_mu", "bias_sigma")}
    for c in range(n_chains):
        with Tap(seed + 50 * c) as tap:
            ch = tda.DAChain(copy.deepcopy(post_c), copy.deepcopy(post_f), copy.deepcopy(prop), L,
                             initial_parameters=theta0[c].copy(), adaptive_error_model=aem)
            ch.sample(iters, progressbar=False)
        loc = np.array(ch.is_coarse, dtype=bool)
        cth, clp, cll, _ = chain_trace(ch.chain_coarse)
        fth, flp, fll, _ = chain_trace(ch.chain_fine)
        acc_c = np.array(ch.accepted_coarse, dtype=np.uint8)
        out["th0"].append(cth[loc]); out["lp0"].append(clp[loc]); out["ll0"].append(cll[loc]); out["acc0"].append(acc_c[loc])
        out["th1"].append(fth); out["lp1"].append(flp); out["ll1"].append(fll)
        out["acc1"].append(np.array(ch.accepted_fine, dtype=np.uint8))
        us = _split_uniforms(tap.ulog, 2, None)
        out["z"].append(np.array(tap.take("z")))
        out["u0"].append(np.array(us[0]))
        evaluated = acc_c[loc].reshape(iters, L).sum(axis=1) > 0
        out["u1"].append(_place(us[1], evaluated))
        out["bias_mu"].append(np.array(ch.posterior_coarse.likelihood.bias, copy=True))
        out["bias_sigma"].append(np.array(ch.bias.get_sigma(), copy=True))
    flat = {"prop_" + k: np.array(v) for k, v in pcfg.items()}
    save(name, A0=As[0], A1=As[1], b0=bs[0], b1=bs[1], y0=y, y1=y, noise_var=np.array(sigma ** 2), prior_mean=pm, prior_cov=pc,
         theta0=theta0, subchain_length=np.array(L), aem=np.array(aem), **flat, **{k: np.array(v) for k, v in out.items()})


def g8_mlda_aem(name, d=4, m=8, sl=(3, 2), iters=30, n_chains=4, seed=811, prop_var=0.05):
    sigma = 0.3
    nl = len(sl) + 1
    As, bs, y, theta_true = _aem_problem(seed, d, m, nl, sigma)
    pm, pc = np.zeros(d), np.eye(d)
    prior = stats.multivariate_normal(pm, pc)
    cov = sigma ** 2 * np.eye(m)
    posts = [tda.Posterior(prior, tda.AdaptiveGaussianLogLike(y, cov) if k < nl - 1 else tda.GaussianLogLike(y, cov),
                           make_model(As[k], bs[k])) for k in range(nl)]
    rng = np.random.default_rng(seed + 1)
    theta0 = theta_true[None] + 0.2 * rng.standard_normal((n_chains, d))
    prop = tda.GaussianRandomWalk(prop_var * np.eye(d), scaling=1.0)
    pcfg = dict(kind="grw", C=prop_var * np.eye(d), scaling=1.0, adaptive=False, gamma=1.01, period=100)
    out = {"z": []}
    for k in range(nl):
        for key in ("u", "th", "lp", "ll", "acc"):
            out["%s%d" % (key, k)] = []
    for c in range(n_chains):
        with Tap(seed + 50 * c) as tap:
            ch = tda.MLDAChain([copy.deepcopy(p) for p in posts], copy.deepcopy(prop), list(sl), initial_parameters=theta0[c].copy(),
                               adaptive_error_model="state-independent")
            ch.sample(iters, progressbar=False)
        us = _split_uniforms(tap.ulog, nl, None)
        out["z"].append(np.array(tap.take("z")))
        objs = [ch]
        cur = ch.proposal
        while True:
            objs.append(cur)
            if cur.level == 0:
                break
            cur = cur.proposal
        objs = objs[::-1]
        local_acc = []
        for k, ob in enumerate(objs):
            th, lp, ll, _ = chain_trace(ob.chain)
            acc = np.array(ob.accepted, dtype=np.uint8)
            if k < nl - 1:
                loc = np.array(ob.is_local, dtype=bool)
                th, lp, ll, acc = th[loc], lp[loc], ll[loc], acc[loc]
            out["th%d" % k].append(th); out["lp%d" % k].append(lp); out["ll%d" % k].append(ll); out["acc%d" % k].append(acc)
            local_acc.append(acc)
        out["u0"].append(np.array(us[0]))
        for k in range(1, nl):
            nsteps = len(local_acc[k]) - (1 if k == nl - 1 else 0)
            evaluated = local_acc[k - 1].reshape(nsteps, sl[k - 1]).sum(axis=1) > 0
            out["u%d" % k].append(_place(us[k], evaluated))
    flat = {"prop_" + k: np.array(v) for k, v in pcfg.items()}
    lv = {}
    for k in range(nl):
        lv["A%d" % k], lv["b%d" % k], lv["y%d" % k] = As[k], bs[k], y
    save(name, noise_var=np.array(sigma ** 2), prior_mean=pm, prior_cov=pc, theta0=theta0, subchain_lengths=np.array(sl),
         n_levels=np.array(nl), aem=np.array("state-independent"), **lv, **flat, **{k: np.array(v) for k, v in out.items()})


def g10_jointprior():
    """JointPrior (distributions.py:8-100) of scalar normal and uniform components under a random-walk chain whose
    proposals leave the uniform supports now and then (log-prior -inf -> rejected), plus log-density known answers."""
    d, m = 5, 12
    A, theta_true, y = linear_problem(1001, d, m, sigma=0.2)
    kinds = np.array([0, 1, 0, 1, 1])                      # 0 = norm(loc, scale), 1 = uniform(loc, scale)
    loc = np.array([0.3, -1.0, -0.5, -2.0, 0.0])
    scale = np.array([1.5, 2.0, 0.7, 4.0, 1.0])
    comps = [stats.norm(loc[j], scale[j]) if kinds[j] == 0 else stats.uniform(loc[j], scale[j]) for j in range(d)]
    prior = tda.JointPrior(comps)
    like = tda.GaussianLogLike(y, 0.04 * np.eye(m))
    post = tda.Posterior(prior, like, make_model(A))
    prop = tda.GaussianRandomWalk(C=0.3 * np.eye(d), scaling=1.0, adaptive=True, gamma=1.01, period=25)
    rs = np.random.RandomState(5)
    n_chains = 4
    theta0 = np.array([[loc[j] + (0.2 + 0.6 * rs.rand()) * scale[j] if kinds[j] else loc[j] + 0.3 * rs.randn() for j in range(d)]
                       for _ in range(n_chains)])
    res, snaps = run_mh(post, prop, theta0, 150, n_chains, seed=1011, snapshot={"period": 25, "fn": lambda p: (p.scaling, p.k, p.t)})
    pts = np.concatenate([theta0, theta0 + rs.randn(n_chains, d) * 1.5, loc[None, :] + scale[None, :] * np.array([[0.0, 0.0, 1.0, 1.0, 0.5]])])
    kat = np.array([prior.logpdf(x) for x in pts])
    assert np.isinf(res["logprior"]).sum() == 0 and (res["accepted"] == 0).sum() > 0
    save("g10_jointprior", A=A, data=y, noise_var=np.array(0.04), kinds=kinds, loc=loc, scale=scale, C=0.3 * np.eye(d),
         scaling0=np.array(1.0), gamma=np.array(1.01), period=np.array(25), theta0=theta0,
         scaling_hist=np.array(snaps)[:, :, 0], kat_points=pts, kat_logpdf=kat, **res)


class _MVNq:
    """q for IndependenceSampler: draws through np.random.multivariate_normal (so the tap records the normals and maps
    them through chol(cov)), log-density from scipy."""

    def __init__(self, mean, cov):
        self.mean, self.cov = np.asarray(mean, dtype=float), np.asarray(cov, dtype=float)
        self._dist = stats.multivariate_normal(self.mean, self.cov)

    def rvs(self, n):
        return np.random.multivariate_normal(self.mean, self.cov)[None, :]

    def logpdf(self, x):
        return self._dist.logpdf(x)


def g12_independence():
    """IndependenceSampler (proposal.py:65-129) with a Gaussian q centred near the posterior."""
    d, m = 4, 10
    A, theta_true, y = linear_problem(1201, d, m, sigma=0.3)
    prior = stats.multivariate_normal(np.zeros(d), np.eye(d))
    like = tda.GaussianLogLike(y, 0.09 * np.eye(m))
    post = tda.Posterior(prior, like, make_model(A))
    cov_post = np.linalg.inv(A.T @ A / 0.09 + np.eye(d))
    mean_post = cov_post @ (A.T @ y / 0.09)
    rs = np.random.RandomState(12)
    B = rs.randn(d, d) * 0.15
    q_mean = mean_post + 0.1 * rs.randn(d)
    q_cov = 2.0 * cov_post + B @ B.T
    prop = tda.IndependenceSampler(_MVNq(q_mean, q_cov))
    n_chains = 4
    theta0 = mean_post[None, :] + 0.3 * rs.randn(n_chains, d)
    res, _ = run_mh(post, prop, theta0, 120, n_chains, seed=1211)
    assert 0.1 < res["accepted"][:, 1:].mean() < 0.9
    save("g12_independence", A=A, data=y, noise_var=np.array(0.09), prior_mean=np.zeros(d), prior_cov=np.eye(d),
         q_mean=q_mean, q_cov=q_cov, theta0=theta0, **res)


def g16_get_samples():
    """Layout pin for diagnostics.get_samples (diagnostics.py:114-209): the REFERENCE's sample() result dicts (MH, Delayed
    Acceptance, 3-level MLDA; a model returning (output, qoi)) flattened to arrays, and what the reference's get_samples makes
    of them for every attribute / level / burn-in the notebooks use.  Stored: the per-link traces of every chain key of the
    result dict (`<case>__res__<key>__<field>`) and every get_samples output (`<case>__gs<j>__<key>`), scalars as 0-d arrays."""
    import contextlib
    import io

    sigma, d = 0.2, 4
    As, ys, theta_true, pm, pc = _ml_problem(1601, d, (6, 9, 12), sigma)
    prior = stats.multivariate_normal(pm, pc)
    w = np.linspace(0.5, 1.5, d)

    def model_of(A):
        return lambda th: (A @ th, np.array([float(w @ th), float(th[0] * th[1])]))  # (output, qoi): posterior.py:97-101

    posts = [tda.Posterior(prior, tda.GaussianLogLike(y, sigma ** 2 * np.eye(len(y))), model_of(A)) for A, y in zip(As, ys)]
    rng = np.random.default_rng(1602)
    theta0 = [theta_true + 0.2 * rng.standard_normal(d) for _ in range(3)]
    cases = {
        "mh": dict(posteriors=posts[2], proposal=tda.GaussianRandomWalk(0.02 * np.eye(d)), iterations=24, n_chains=3),
        "da": dict(posteriors=posts[1:], proposal=tda.GaussianRandomWalk(0.02 * np.eye(d)), iterations=12, n_chains=3, subchain_length=3),
        "mlda": dict(posteriors=posts, proposal=tda.GaussianRandomWalk(0.02 * np.eye(d)), iterations=8, n_chains=3, subchain_length=[3, 2]),
    }
    queries = {
        "mh": [dict(), dict(burnin=5), dict(attribute="stats", burnin=3), dict(attribute="model_output"), dict(attribute="qoi", burnin=2),
               dict(attribute="likelihood", burnin=4)],
        "da": [dict(), dict(level="coarse", burnin=4), dict(attribute="stats", level="coarse"), dict(attribute="model_output", level="fine", burnin=2),
               dict(attribute="qoi", level="coarse", burnin=1), dict(attribute="posterior", level="fine")],
        "mlda": [dict(level=2), dict(level=1, burnin=3), dict(level=0, burnin=7), dict(attribute="stats", level=1), dict(attribute="model_output", level=0, burnin=2),
                 dict(attribute="qoi", level=2, burnin=1)],
    }
    arrays = {}
    for name, kw in cases.items():
        with Tap(1610 + len(name)), contextlib.redirect_stdout(io.StringIO()):
            res = tda.sample(initial_parameters=[t.copy() for t in theta0], force_sequential=True, **kw)
        for key, val in res.items():
            if key.startswith("chain_"):
                arrays["%s__res__%s__parameters" % (name, key)] = np.array([l.parameters for l in val])
                arrays["%s__res__%s__prior" % (name, key)] = np.array([l.prior for l in val])
                arrays["%s__res__%s__likelihood" % (name, key)] = np.array([l.likelihood for l in val])
                arrays["%s__res__%s__model_output" % (name, key)] = np.array([l.model_output for l in val])
                arrays["%s__res__%s__qoi" % (name, key)] = np.array([l.qoi for l in val])
            else:
                arrays["%s__res__%s" % (name, key)] = np.array(val)
        for j, q in enumerate(queries[name]):
            gs = tda.get_samples(res, **q)
            arrays["%s__gs%d__query" % (name, j)] = np.array(repr(sorted(q.items())))
            for key, val in gs.items():
                arrays["%s__gs%d__%s" % (name, j, key)] = np.array(val)
    save("g16_get_samples", **arrays)


FIXTURES = {
    "g1_basic_sampler": g1_basic_sampler,
    "g2_am_small": lambda: g2_am("g2_am_small", d=8, m=16, n_chains=8, iters=128, t0=16, period=16, seed=201),
    "g2_am_small_adaptive": lambda: g2_am("g2_am_small_adaptive", d=8, m=16, n_chains=4, iters=128,
                                          t0=0, period=16, seed=202, adaptive=True),
    "g2_am_diag_genprior": lambda: g2_am("g2_am_diag_genprior", d=6, m=20, n_chains=4, iters=96, t0=16,
                                         period=16, seed=203, noise="diag", prior_kind="general"),
    "g2_am_dense": lambda: g2_am("g2_am_dense", d=6, m=20, n_chains=4, iters=96, t0=16, period=16,
                                 seed=204, noise="dense", prior_kind="general"),
    "g2_am_c2": lambda: g2_am("g2_am_c2", d=64, m=1024, n_chains=2, iters=300, t0=100, period=100,
                              seed=1, c0=1e-4),
    # round 5: 96 parameters (the device's 65 .. 128-parameter path, tda_kernels_wide.h), two covariance swaps
    "g2_am_d96": lambda: g2_am("g2_am_d96", d=96, m=256, n_chains=2, iters=300, t0=100, period=100, seed=96, c0=1e-4),
    "g2b_pcn": g2b_pcn,
    "g3_loglike_kats": g3_loglike_kats,
    "g4_da_pcn": lambda: g4_da("g4_da_pcn", "pcn"),
    "g4_da_grw_adaptive": lambda: g4_da("g4_da_grw_adaptive", "grw", adaptive=True, period=7, seed=402),
    "g4_da_am_random": lambda: g4_da("g4_da_am_random", "am", randomize=True, period=10, seed=403, prior_kind="general"),
    "g4_da_pcn_adaptive_c3shape": lambda: g4_da("g4_da_pcn_adaptive_c3shape", "pcn", d=16, ms=(32, 96), L=10, iters=30,
                                                adaptive=True, period=25, seed=404),
    # dense observation covariances inside a hierarchy (DefaultGaussianLogLike at the fine level / at every level; round 4)
    "g4_da_pcn_dense_fine": lambda: g4_da("g4_da_pcn_dense_fine", "pcn", seed=405, noise=["iso", "dense"]),
    "g4_da_am_dense_both": lambda: g4_da("g4_da_am_dense_both", "am", d=9, ms=(20, 40), L=3, iters=40, period=10, seed=406,
                                         prior_kind="general", noise=["dense", "dense"]),
    "g5_mlda_am_dense": lambda: g5_mlda("g5_mlda_am_dense", "am", period=10, seed=504, noise=["diag", "dense", "dense"]),
    "g6_dreamz_linear": lambda: g6_dreamz("g6_dreamz_linear", "linear", d=6, M0=20, delta=1, nCR=3, adaptive=False, period=25,
                                          iters=150, n_chains=4, seed=601),
    "g6_dreamz_rosen_adaptive": lambda: g6_dreamz("g6_dreamz_rosen_adaptive", "rosenbrock", d=4, M0=30, delta=2, nCR=3,
                                                  adaptive=True, period=25, iters=200, n_chains=4, seed=602),
    "g6_dreamz_empty_subspace": lambda: g6_dreamz("g6_dreamz_empty_subspace", "linear", d=3, M0=12, delta=1, nCR=3,
                                                  adaptive=True, period=20, iters=120, n_chains=3, seed=603, m=7),
    "g6_dreamz_linear_dense": lambda: g6_dreamz("g6_dreamz_linear_dense", "linear", d=6, M0=20, delta=2, nCR=3, adaptive=True, period=25,
                                                iters=120, n_chains=4, seed=604, m=20, noise="dense"),
    "g8_da_aem_indep": lambda: g8_da_aem("g8_da_aem_indep", "state-independent"),
    "g8_da_aem_dep": lambda: g8_da_aem("g8_da_aem_dep", "state-dependent", seed=802),
    "g8_da_aem_dep_pcn": lambda: g8_da_aem("g8_da_aem_dep_pcn", "state-dependent", proposal_kind="pcn", L=1, seed=803),
    "g8_mlda_aem": lambda: g8_mlda_aem("g8_mlda_aem"),
    # error models beyond 64 outputs (two waves per chain in k_aem_action): ragged 72 / 100 and the full 128
    "g8_da_aem_indep_m72": lambda: g8_da_aem("g8_da_aem_indep_m72", "state-independent", d=5, m=72, L=3, iters=16, n_chains=2,
                                             seed=821, prop_var=0.004),
    "g8_da_aem_dep_pcn_m128": lambda: g8_da_aem("g8_da_aem_dep_pcn_m128", "state-dependent", proposal_kind="pcn", d=6, m=128, L=1,
                                                iters=24, n_chains=2, seed=822, beta=0.08),
    "g8_mlda_aem_m100": lambda: g8_mlda_aem("g8_mlda_aem_m100", d=5, m=100, sl=(3, 2), iters=10, n_chains=2, seed=823,
                                            prop_var=0.004),
    # ... and beyond 128 (round 5: k_aem_refresh_big, k_aem_action<256>, k_aem_base_steps<16>): ragged 200 / 160 and the full 256
    "g8_da_aem_indep_m200": lambda: g8_da_aem("g8_da_aem_indep_m200", "state-independent", d=6, m=200, L=3, iters=14, n_chains=2,
                                              seed=831, prop_var=0.003),
    "g8_da_aem_dep_pcn_m256": lambda: g8_da_aem("g8_da_aem_dep_pcn_m256", "state-dependent", proposal_kind="pcn", d=6, m=256, L=1,
                                                iters=20, n_chains=2, seed=832, beta=0.06),
    "g8_mlda_aem_m160": lambda: g8_mlda_aem("g8_mlda_aem_m160", d=5, m=160, sl=(3, 2), iters=8, n_chains=2, seed=833,
                                            prop_var=0.003),
    # error models at 65 .. 128 parameters (round 5: k_ml_steps<128, 1>, k_aem_action<128 | 256> with a parameter per thread)
    "g8_da_aem_indep_d80": lambda: g8_da_aem("g8_da_aem_indep_d80", "state-independent", d=80, m=72, L=3, iters=14, n_chains=2,
                                             seed=841, prop_var=0.0004),
    "g8_da_aem_dep_pcn_d96": lambda: g8_da_aem("g8_da_aem_dep_pcn_d96", "state-dependent", proposal_kind="pcn", d=96, m=100, L=1,
                                               iters=20, n_chains=2, seed=842, beta=0.05),
    "g8_mlda_aem_d72": lambda: g8_mlda_aem("g8_mlda_aem_d72", d=72, m=40, sl=(3, 2), iters=8, n_chains=2, seed=843,
                                           prop_var=0.0004),
    "g5_mlda_am": lambda: g5_mlda("g5_mlda_am", "am", period=10),
    "g5_mlda_grw_adaptive": lambda: g5_mlda("g5_mlda_grw_adaptive", "grw", adaptive=True, period=7, seed=502),
    "g5_mlda_4level": lambda: g5_mlda("g5_mlda_4level", "am", ms=(6, 10, 16, 24), sl=(3, 2, 2), iters=20, period=10, seed=503),
    # five and six levels (round 5: MAXLEV = 6, the generic level kernel)
    "g5_mlda_5level": lambda: g5_mlda("g5_mlda_5level", "grw", ms=(5, 8, 12, 16, 24), sl=(2, 2, 2, 2), iters=10, adaptive=True, period=8, seed=504),
    "g5_mlda_6level": lambda: g5_mlda("g5_mlda_6level", "am", ms=(4, 6, 9, 12, 16, 24), sl=(2, 2, 2, 2, 2), iters=8, period=12, seed=505),
    "g15_da_dreamz": lambda: g15_hier_dreamz("g15_da_dreamz", ms=(10, 24), sl=(3,), seed=1501),
    "g15_mlda_dreamz": lambda: g15_hier_dreamz("g15_mlda_dreamz", ms=(8, 14, 24), sl=(3, 2), iters=20, seed=1502),
    "g15_da_dreamz_random": lambda: g15_hier_dreamz("g15_da_dreamz_random", ms=(10, 24), sl=(4,), seed=1503, randomize=True),
    "g15_da_dreamz_aem": lambda: g15_hier_dreamz("g15_da_dreamz_aem", ms=(8, 8), sl=(3,), seed=1504, aem="state-independent"),
    "g15_da_dreamz_aem_dep": lambda: g15_hier_dreamz("g15_da_dreamz_aem_dep", ms=(8, 8), sl=(2,), seed=1505, aem="state-dependent"),
    "g15_da_dreamz_aem_m160": lambda: g15_hier_dreamz("g15_da_dreamz_aem_m160", ms=(160, 160), sl=(3,), iters=14, n_chains=2, seed=1507, aem="state-independent"),
    "g15_mlda_dreamz_aem": lambda: g15_hier_dreamz("g15_mlda_dreamz_aem", ms=(8, 8, 8), sl=(3, 2), iters=20, seed=1506, aem="state-independent"),
    "g7_moments": g7_moments,
    "g16_get_samples": g16_get_samples,
    "g9_mvn_logpdf": g9_mvn_logpdf,
    "g10_jointprior": g10_jointprior,
    "g12_independence": g12_independence,
    "g13_owcn": lambda: g13_owcn("g13_owcn", False),
    "g13_owcn_adaptive": lambda: g13_owcn("g13_owcn_adaptive", True),
    "g14_mala": lambda: g14_mala("g14_mala", False),
    "g14_mala_adaptive_dense": lambda: g14_mala("g14_mala_adaptive_dense", True, cov_kind="dense"),
}

if __name__ == "__main__":
    names = sys.argv[1:] or list(FIXTURES)
    for n in names:
        FIXTURES[n]()
