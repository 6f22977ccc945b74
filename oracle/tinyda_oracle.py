"""CPU oracle for the tinyDA many-chain MH hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The shipped path (tinyda_amd/) never does; it fails loudly when the HIP library is missing.

What this is: a NumPy restatement of the reference algorithm (mikkelbue/tinyDA, pure Python)
for the path tda.sample() -> Chain.sample -> Proposal / Posterior / GaussianLogLike, written
batch-of-chains (arrays are [chain, ...]) so that N chains advance in lock-step exactly like
the device engine, but with every formula kept in the reference's own arithmetic form.
Each function cites the reference file:line it follows (paths relative to /root/reference).

Pinned: tests/test_oracle_golden.py checks this module against tests/golden/*.npz, which were
produced by running the reference itself (tests/golden/gen_golden.py) on recorded variates.

Random variates: the reference uses the global MT19937 stream; parity is defined on identical
*variates* (SURVEY.md §7).  Every driver here takes explicit arrays z[chain, step, d] and
u[chain, step]; `PhiloxStream` restates the engine's counter-based stream so the device RNG can
be checked word for word.
"""
import math

import numpy as np

LOG_2PI = math.log(2.0 * math.pi)

# ----------------------------------------------------------------------------------------
# Philox4x32-10 stream (the engine's RNG contract, include/tinyda_amd.h "RNG stream")
# ----------------------------------------------------------------------------------------
_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = np.uint32(0x9E3779B9)
_W1 = np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)

STREAM_PROPOSAL = 0
STREAM_ACCEPT = 1
STREAM_INIT = 2
STREAM_DREAM = 3


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox-4x32 with 10 rounds (Salmon et al. 2011), vectorised over equal-shaped counters."""
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint32).copy() for c in np.broadcast_arrays(c0, c1, c2, c3)]
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = c0.astype(np.uint64) * _M0
            p1 = c2.astype(np.uint64) * _M1
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & _MASK).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & _MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32(k0 + _W0)
            k1 = np.uint32(k1 + _W1)
    return c0, c1, c2, c3


def u53(xa, xb):
    """53-bit uniform in [0, 1): ((xa >> 5) * 2^26 + (xb >> 6)) * 2^-53."""
    return ((xa >> np.uint32(5)).astype(np.float64) * 67108864.0 + (xb >> np.uint32(6)).astype(np.float64)) * (
        1.0 / 9007199254740992.0
    )


class PhiloxStream:
    """key = (seed lo, seed hi); counter = (block, step, global chain id, stream tag)."""

    def __init__(self, seed):
        self.k0 = np.uint32(seed & 0xFFFFFFFF)
        self.k1 = np.uint32((seed >> 32) & 0xFFFFFFFF)

    def words(self, chain, step, stream, block):
        return philox4x32_10(block, step, chain, stream, self.k0, self.k1)

    def normals(self, chains, step, d, stream=STREAM_PROPOSAL):
        """[len(chains), d] standard normals: block b -> (z[2b], z[2b+1]) by Box-Muller,
        r = sqrt(-2 ln(1-u1)), angle = 2 pi u2."""
        chains = np.asarray(chains, dtype=np.uint32)
        nb = (d + 1) // 2
        b = np.arange(nb, dtype=np.uint32)[None, :]
        x0, x1, x2, x3 = self.words(chains[:, None], np.uint32(step), np.uint32(stream), b)
        u1 = u53(x0, x1)
        u2 = u53(x2, x3)
        r = np.sqrt(-2.0 * np.log(1.0 - u1))
        ang = 2.0 * np.pi * u2
        z = np.empty((chains.shape[0], 2 * nb))
        z[:, 0::2] = r * np.cos(ang)
        z[:, 1::2] = r * np.sin(ang)
        return z[:, :d]

    def uniform(self, chains, step, level=0):
        chains = np.asarray(chains, dtype=np.uint32)
        x0, x1, _, _ = self.words(chains, np.uint32(step), np.uint32(STREAM_ACCEPT), np.uint32(level))
        return u53(x0, x1)


# ----------------------------------------------------------------------------------------
# densities
# ----------------------------------------------------------------------------------------
class MVNPrior:
    """scipy.stats.multivariate_normal(mean, cov).logpdf as called at posterior.py:92.

    scipy (1.15, _multivariate.py `_logpdf` / `_PSD`) evaluates
        -0.5 * (rank*log(2 pi) + log_pdet + || (x-mean) @ U ||^2),  U = eigvec / sqrt(eigval)
    from a symmetric eigendecomposition; restated here with numpy.linalg.eigh.
    """

    def __init__(self, mean, cov):
        self.mean = np.asarray(mean, dtype=float)
        self.cov = np.asarray(cov, dtype=float)
        s, v = np.linalg.eigh(self.cov)
        self.U = v / np.sqrt(s)
        self.log_pdet = float(np.sum(np.log(s)))
        self.rank = self.mean.shape[0]

    def logpdf(self, x):
        dev = np.atleast_2d(x) - self.mean
        maha = np.sum(np.square(dev @ self.U), axis=-1)
        return -0.5 * (self.rank * LOG_2PI + self.log_pdet + maha)


def loglike_isotropic(F, data, var):
    """distributions.py:324-326  -0.5 * ||F - data||^2 / var  (norm, then squared)."""
    return -0.5 * np.linalg.norm(F - data, axis=-1) ** 2 / var


def loglike_diagonal(F, data, diag_cov):
    """distributions.py:310-312  -0.5 * sum((F - data)^2 / diag)."""
    return -0.5 * ((F - data) ** 2 / diag_cov).sum(axis=-1)


def loglike_dense(F, data, cov_inverse, bias=0.0):
    """distributions.py:295-298 (bias = 0) and :419-425 / :444-446 (bias-corrected)."""
    r = F + bias - data
    return -0.5 * np.einsum("...i,ij,...j->...", r, cov_inverse, r)


def make_loglike(kind, data, noise):
    """GaussianLogLike factory outcome (distributions.py:203-243) as a closure over F[chain, m].
    kind: 'iso' (noise = variance), 'diag' (noise = diagonal), 'dense' (noise = covariance)."""
    data = np.asarray(data, dtype=float)
    # .grad = grad_loglike with respect to the model output (distributions.py:300-301, :314-315, :328-329), used by MALA
    if kind == "iso":
        var = float(noise)
        fn = lambda F: loglike_isotropic(F, data, var)
        fn.grad = lambda F: 1 / var * (data - F)
        return fn
    if kind == "diag":
        dg = np.asarray(noise, dtype=float)
        fn = lambda F: loglike_diagonal(F, data, dg)
        fn.grad = lambda F: 1 / dg * (data - F)
        return fn
    if kind == "dense":
        inv = np.linalg.inv(np.asarray(noise, dtype=float))  # distributions.py:280
        fn = lambda F: loglike_dense(F, data, inv)
        fn.grad = lambda F: (data - F) @ inv.T
        return fn
    raise ValueError(kind)


def classify_covariance(cov):
    """Which class the factory returns (distributions.py:237-243)."""
    cov = np.asarray(cov)
    if np.count_nonzero(cov - np.diag(np.diag(cov))) == 0:
        if np.all(np.diag(cov) == cov[0, 0]):
            return "iso"
        return "diag"
    return "dense"


def noise_of(cov):
    """(kind, noise) arguments of make_loglike for a covariance matrix, as the factory classifies it (distributions.py:237-243)"""
    cov = np.asarray(cov, dtype=float)
    kind = classify_covariance(cov)
    return kind, (float(cov[0, 0]) if kind == "iso" else (np.diag(cov).copy() if kind == "diag" else cov))


class AdaptiveLogLike:
    """AdaptiveGaussianLogLike (distributions.py:332-449), one instance per chain batch entry."""

    def __init__(self, data, cov):
        self.data = np.asarray(data, dtype=float)
        self.cov = np.asarray(cov, dtype=float)
        self.cov_inverse = np.linalg.inv(self.cov)
        self.bias = np.zeros(self.data.shape[0])

    def set_bias(self, mean_bias, cov_bias):
        self.bias = mean_bias
        if not np.all(cov_bias < 1e-9):  # distributions.py:399-402
            self.cov_inverse = np.linalg.inv(self.cov + cov_bias)

    def loglike(self, F):
        return loglike_dense(F, self.data, self.cov_inverse, self.bias)

    def loglike_custom_bias(self, F, bias):
        return loglike_dense(F, self.data, self.cov_inverse, bias)


# ----------------------------------------------------------------------------------------
# running moments  (utils.py:104-124, :189-201)
# ----------------------------------------------------------------------------------------
def moments_update(mu, sigma, t, x, sd=1.0, epsilon=0.0):
    """One RecursiveSampleMoments.update for a batch: mu[N,d], sigma[N,d,d], x[N,d]; t is the
    recursion counter BEFORE the update (starts at 1).  Same operation order as utils.py:113-122."""
    d = mu.shape[-1]
    mu_prev = mu
    mu_new = (1 / (t + 1)) * (t * mu_prev + x)
    outer = lambda a: a[..., :, None] * a[..., None, :]
    sigma_new = (t - 1) / t * sigma + sd / t * (
        t * outer(mu_prev) - (t + 1) * outer(mu_new) + outer(x) + epsilon * np.eye(d)
    )
    return mu_new, sigma_new


def zero_mean_moments_update(sigma, t, x):
    """ZeroMeanRecursiveSampleMoments.update (utils.py:189-201)."""
    return (t - 1) / t * sigma + 1 / t * (x[..., :, None] * x[..., None, :])


# ----------------------------------------------------------------------------------------
# single-level Metropolis-Hastings (chain.py:37-129) with GRW / pCN / AM proposals
# ----------------------------------------------------------------------------------------
class JointPriorOracle:
    """JointPrior.logpdf (distributions.py:45-58): the sum over components, in parameter order, of scipy's scalar
    log-densities; kinds[j] 0 = norm(loc, scale), 1 = uniform(loc, scale)."""

    def __init__(self, kinds, loc, scale):
        import scipy.stats as stats

        self.kinds, self.loc, self.scale = np.asarray(kinds), np.asarray(loc, dtype=float), np.asarray(scale, dtype=float)
        self.comps = [stats.norm(l, s) if k == 0 else stats.uniform(l, s) for k, l, s in zip(self.kinds, self.loc, self.scale)]
        self.mean = np.where(self.kinds == 0, self.loc, self.loc + 0.5 * self.scale)
        self.cov = np.diag(np.where(self.kinds == 0, self.scale ** 2, self.scale ** 2 / 12.0))

    def logpdf(self, theta):
        theta = np.atleast_2d(theta)
        out = np.zeros(theta.shape[0])
        with np.errstate(divide="ignore"):
            for j, c in enumerate(self.comps):  # sum([...]) in the reference: left to right from 0
                out = out + c.logpdf(theta[:, j])
        return out


class LinearGaussianLevel:
    """Posterior.create_link (posterior.py:78-110) for model F = A theta (+ b)."""

    def __init__(self, A, data, noise_kind, noise, prior, b=None):
        self.A = np.asarray(A, dtype=float)
        self.b = None if b is None else np.asarray(b, dtype=float)
        self.prior = prior
        self.loglike = make_loglike(noise_kind, data, noise)

    def forward(self, theta):
        F = theta @ self.A.T
        return F if self.b is None else F + self.b

    def evaluate(self, theta):
        lp = self.prior.logpdf(theta)
        F = self.forward(theta)
        ll = self.loglike(F)
        return lp, ll, F

    def grad_logpost(self, theta, F):
        """MALA._compute_gradient (proposal.py:986-998) for a model whose `gradient(theta, s)` returns A^T s:
        grad_log_p (utils.py:273-280) + A^T grad_log_l (utils.py:283-287)."""
        g_prior = (self.prior.mean[None, :] - theta) @ np.linalg.inv(self.prior.cov).T
        return g_prior + self.loglike.grad(F) @ self.A


class CallableGaussianLevel:
    """Posterior.create_link (posterior.py:78-110) for an arbitrary forward model given as a batched Python callable
    theta[N, d] -> F[N, m] (checker for the engine's source-defined models)."""

    def __init__(self, fn, data, noise_kind, noise, prior):
        self.fn = fn
        self.prior = prior
        self.loglike = make_loglike(noise_kind, data, noise)

    def forward(self, theta):
        return np.asarray(self.fn(theta), dtype=float)

    def evaluate(self, theta):
        F = self.forward(theta)
        return self.prior.logpdf(theta), self.loglike(F), F


def _acceptance(kind, lp_new, ll_new, lp_old, ll_old):
    """proposal.py:253-258 (GRW/AM/DREAMZ: posterior ratio) and :357-362 (pCN: likelihood ratio).
    posterior = prior + likelihood as in link.py:48; NaN posterior -> 0."""
    post_new = lp_new + ll_new
    post_old = lp_old + ll_old
    with np.errstate(over="ignore", invalid="ignore"):
        if kind in ("pcn", "owcn"):  # OperatorWeightedCrankNicolson inherits CrankNicolson.get_acceptance
            alpha = np.exp(ll_new - ll_old)
        else:
            alpha = np.exp(post_new - post_old)
    return np.where(np.isnan(post_new), 0.0, alpha)


def owcn_operators(B, scaling):
    """state / noise operators of OperatorWeightedCrankNicolson (proposal.py:576-580): real parts of the principal
    matrix square roots of I - scaling B and scaling B (scipy.linalg.sqrtm, as the reference calls it)."""
    from scipy.linalg import sqrtm

    d = B.shape[0]
    return np.real(sqrtm(np.eye(d) - scaling * B)), np.real(sqrtm(scaling * B))


def run_mh(level, proposal, theta0, z, u):
    """N chains x T steps of Chain.sample (chain.py:95-125) on recorded variates.

    proposal: dict with 'kind' in {'grw','pcn','am','owcn','mala','indep'} and
        mala: scaling(sigma), adaptive, gamma, period           (proposal.py:861-1005)
        owcn: B[d,d], scaling, adaptive, gamma, period; C = prior covariance (proposal.py:515-605)
        grw: C[d,d], scaling, adaptive, gamma, period            (proposal.py:171-258)
        pcn: scaling(beta), adaptive, gamma, period; C = prior covariance (proposal.py:302-362)
        am : C0[d,d], sd, epsilon, t0, period, adaptive, gamma   (proposal.py:416-512)
    theta0[N,d]; z[N,T,d] standard normals mapped through chol(C) (see gen_golden.py); u[N,T].
    Returns dict of traces with the initial link at index 0, like tinyDA's chain lists.
    """
    theta0 = np.asarray(theta0, dtype=float)
    N, d = theta0.shape
    T = z.shape[1]
    kind = proposal["kind"]
    adaptive = bool(proposal.get("adaptive", False))
    gamma = float(proposal.get("gamma", 1.01))
    period = int(proposal.get("period", 100))
    alpha_star = 0.24  # proposal.py:169

    if kind == "grw":
        C = np.broadcast_to(np.asarray(proposal["C"], dtype=float), (N, d, d)).copy()
        scaling = np.full(N, float(proposal.get("scaling", 1.0)))
    elif kind == "pcn":
        C = np.broadcast_to(level.prior.cov, (N, d, d)).copy()  # proposal.py:336-341
        scaling = np.full(N, float(proposal.get("scaling", 0.1)))
    elif kind == "am":
        C = np.broadcast_to(np.asarray(proposal["C0"], dtype=float), (N, d, d)).copy()
        scaling = np.ones(N)  # proposal.py:462
        sd = proposal.get("sd")
        sd = min(1.0, 2.4 ** 2 / d) if sd is None else float(sd)  # proposal.py:465-468
        eps = float(proposal.get("epsilon", 1e-6))
        t0 = int(proposal.get("t0", 0))
        am_mu = theta0.copy()  # proposal.py:495-500
        am_sigma = np.zeros((N, d, d))
    elif kind == "owcn":  # OperatorWeightedCrankNicolson (proposal.py:515-605): C = prior covariance, operators from B
        C = np.broadcast_to(level.prior.cov, (N, d, d)).copy()
        scaling = np.full(N, float(proposal.get("scaling", 1.0)))
        B = np.asarray(proposal["B"], dtype=float)
        state_op, noise_op = zip(*[owcn_operators(B, sc) for sc in scaling])  # proposal.py:576-580
        state_op, noise_op = np.array(state_op), np.array(noise_op)
    elif kind == "mala":  # MALA (proposal.py:861-1005): unit-covariance normals, drift along the posterior gradient
        C = np.broadcast_to(np.eye(d), (N, d, d)).copy()
        scaling = np.full(N, float(proposal.get("scaling", 0.1)))
        alpha_star = 0.57  # proposal.py:899
    elif kind == "indep":  # IndependenceSampler with q = N(q_mean, q_cov)  (proposal.py:65-129); never adapts
        q = MVNPrior(proposal["q_mean"], proposal["q_cov"])
        C = np.broadcast_to(q.cov, (N, d, d)).copy()
        scaling = np.ones(N)
        adaptive = False
    else:
        raise ValueError(kind)
    L = np.linalg.cholesky(C)

    theta = theta0.copy()
    lp, ll, F0 = level.evaluate(theta)
    lq = q.logpdf(theta) if kind == "indep" else None
    grad = level.grad_logpost(theta, F0) if kind == "mala" else None
    out_theta = np.empty((N, T + 1, d))
    out_lp = np.empty((N, T + 1))
    out_ll = np.empty((N, T + 1))
    out_acc = np.ones((N, T + 1), dtype=np.uint8)
    out_theta[:, 0], out_lp[:, 0], out_ll[:, 0] = theta, lp, ll
    scaling_hist, C_hist = [], []
    t = 0  # proposal.t: number of adapt() calls
    k = 0  # diminishing-adaptation counter

    for s in range(T):
        inc = np.einsum("nij,nj->ni", L, z[:, s])
        if kind == "pcn":
            prop = np.sqrt(1 - scaling ** 2)[:, None] * theta + scaling[:, None] * inc  # proposal.py:351-355
        elif kind == "mala":  # proposal.py:945-956
            sg = scaling[:, None]
            prop = theta + 0.5 * sg ** 2 * grad + sg * inc
        elif kind == "owcn":  # proposal.py:592-598
            prop = np.einsum("nij,nj->ni", state_op, theta) + np.einsum("nij,nj->ni", noise_op, inc)
        elif kind == "indep":
            prop = q.mean[None, :] + inc  # q.rvs (proposal.py:113-115) through the Cholesky map of the variate tap
        else:
            prop = theta + scaling[:, None] * inc  # proposal.py:249-251
        lp_n, ll_n, F_n = level.evaluate(prop)
        if kind == "mala":  # proposal.py:958-984
            grad_n = level.grad_logpost(prop, F_n)
            sg = scaling[:, None]
            q_x_y = -0.5 / scaling ** 2 * np.linalg.norm(theta - prop - 0.5 * sg ** 2 * grad_n, axis=1) ** 2
            q_y_x = -0.5 / scaling ** 2 * np.linalg.norm(prop - theta - 0.5 * sg ** 2 * grad, axis=1) ** 2
            with np.errstate(over="ignore", invalid="ignore"):
                alpha = np.exp((lp_n + ll_n) - (lp + ll) + q_x_y - q_y_x)
            alpha = np.where(np.isnan(lp_n + ll_n), 0.0, alpha)
        elif kind == "indep":  # proposal.py:117-123: exp(post' - post + q(prev) - q(prop))
            lq_n = q.logpdf(prop)
            with np.errstate(over="ignore", invalid="ignore"):
                alpha = np.exp((lp_n + ll_n) - (lp + ll) + lq - lq_n)
        else:
            alpha = _acceptance(kind, lp_n, ll_n, lp, ll)
        acc = u[:, s] < alpha  # chain.py:112
        if kind == "indep":
            lq = np.where(acc, lq_n, lq)
        if kind == "mala":
            grad = np.where(acc[:, None], grad_n, grad)
        theta = np.where(acc[:, None], prop, theta)
        lp = np.where(acc, lp_n, lp)
        ll = np.where(acc, ll_n, ll)
        out_theta[:, s + 1], out_lp[:, s + 1], out_ll[:, s + 1], out_acc[:, s + 1] = theta, lp, ll, acc

        # ---- adapt (proposal.py:228-245; AM :502-512) ----
        t += 1
        if adaptive and t % period == 0:
            rate = out_acc[:, : s + 2][:, -period:].mean(axis=1)
            scaling = np.exp(np.log(scaling) + gamma ** -k * (rate - alpha_star))
            k += 1
            if kind == "owcn":  # proposal.py:582-590
                state_op, noise_op = zip(*[owcn_operators(B, sc) for sc in scaling])
                state_op, noise_op = np.array(state_op), np.array(noise_op)
        if kind == "am":
            am_mu, am_sigma = moments_update(am_mu, am_sigma, t, theta, sd, eps)  # recursor.t == t here
            if t >= t0 and t % period == 0:
                C = am_sigma.copy()
                L = np.linalg.cholesky(C)
        if t % period == 0:
            scaling_hist.append(scaling.copy())
            C_hist.append(C.copy())

    res = dict(theta=out_theta, logprior=out_lp, loglike=out_ll, logpost=out_lp + out_ll, accepted=out_acc,
               scaling=scaling, scaling_hist=np.array(scaling_hist).T if scaling_hist else None,
               C_hist=np.swapaxes(np.array(C_hist), 0, 1) if C_hist else None, C=C, L=L)
    if kind == "am":
        res.update(am_mu=am_mu, am_sigma=am_sigma)
    return res


# ----------------------------------------------------------------------------------------
# Delayed Acceptance (chain.py:185-530) and MLDA (chain.py:534-769, proposal.py:1285-1624)
# ----------------------------------------------------------------------------------------
class _BaseProposalState:
    """The coarsest-level proposal of a batch of chains: GRW / pCN / AM state + adapt(), with the
    `accepted` list the reference hands to adapt() kept per chain (it contains the alignment entries that
    DAChain / MLDA.align_chain append, chain.py:363,389,397 and proposal.py:1486, so the scaling window
    of proposal.py:236 sees them)."""

    def __init__(self, proposal, theta0, prior_cov):
        N, d = theta0.shape
        self.kind = proposal["kind"]
        self.adaptive = bool(proposal.get("adaptive", False))
        self.gamma = float(proposal.get("gamma", 1.01))
        self.period = int(proposal.get("period", 100))
        if self.kind == "grw":
            C = np.asarray(proposal["C"], dtype=float)
            self.scaling = np.full(N, float(proposal.get("scaling", 1.0)))
        elif self.kind == "pcn":
            C = np.asarray(prior_cov, dtype=float)
            self.scaling = np.full(N, float(proposal.get("scaling", 0.1)))
        else:
            C = np.asarray(proposal["C0"], dtype=float)
            self.scaling = np.ones(N)
            sd = proposal.get("sd")
            self.sd = min(1.0, 2.4 ** 2 / d) if sd is None else float(sd)
            self.eps = float(proposal.get("epsilon", 1e-6))
            self.t0 = int(proposal.get("t0", 0))
            self.mu = theta0.copy()
            self.sigma = np.zeros((N, d, d))
        self.L = np.linalg.cholesky(np.broadcast_to(C, (N, d, d)).copy())
        self.t = 0
        self.k = 0
        self.accepted = [np.ones(N, dtype=bool)]  # chain.py:256 / proposal.py:1380

    def propose(self, theta, z):
        inc = np.einsum("nij,nj->ni", self.L, z)
        if self.kind == "pcn":
            return np.sqrt(1 - self.scaling ** 2)[:, None] * theta + self.scaling[:, None] * inc
        return theta + self.scaling[:, None] * inc

    def adapt(self, theta):
        self.t += 1
        if self.adaptive and self.t % self.period == 0:
            rate = np.mean(np.array(self.accepted[-self.period:]), axis=0)
            self.scaling = np.exp(np.log(self.scaling) + self.gamma ** -self.k * (rate - 0.24))
            self.k += 1
        if self.kind == "am":
            self.mu, self.sigma = moments_update(self.mu, self.sigma, self.t, theta, self.sd, self.eps)
            if self.t >= self.t0 and self.t % self.period == 0:
                self.L = np.linalg.cholesky(self.sigma)


def run_multilevel(levels, proposal, subchain_lengths, theta0, z, u_levels, n_fine, ridx=None):
    """DAChain.sample (2 levels) / MLDAChain.sample (>2) for N chains in lock-step on recorded variates.

    levels: LinearGaussianLevel list, coarsest first.  subchain_lengths[k] = steps of level k per step of
    level k+1 (sampler.py:260-264).  z [N, T0, d] base-level normals, u_levels[k] [N, n_k] uniforms of level
    k (NaN where the reference drew none), ridx [N, n_fine] DA promoted index in [-L, -1] (chain.py:525-527)
    or None for the fixed last state.
    Invariant used: after a step of level q completes, every level j < q sits at theta_q
    and S[j][q] holds level j's (log-prior, log-like) there; a rejection at level q restores those.
    Returns per-level traces of *local* steps (what sampler.py:421-427 / :535-538 return), the finest level
    with its initial link first.
    """
    theta0 = np.asarray(theta0, dtype=float)
    N, d = theta0.shape
    nl = len(levels)
    sl = list(subchain_lengths)
    prop = _BaseProposalState(proposal, theta0, levels[0].prior.cov)
    th = [theta0.copy() for _ in range(nl)]
    lp, ll = [None] * nl, [None] * nl
    for k in range(nl):
        lp[k], ll[k], _ = levels[k].evaluate(theta0)
    S = {(j, q): (lp[j].copy(), ll[j].copy()) for q in range(nl) for j in range(q)}
    rec = [dict(theta=[], logprior=[], loglike=[], accepted=[]) for _ in range(nl)]
    rec[nl - 1]["theta"].append(th[nl - 1].copy())
    rec[nl - 1]["logprior"].append(lp[nl - 1].copy())
    rec[nl - 1]["loglike"].append(ll[nl - 1].copy())
    rec[nl - 1]["accepted"].append(np.ones(N, dtype=bool))
    cnt = [0] * nl  # local steps done per level
    promoted = {}

    def record(k, acc):
        rec[k]["theta"].append(th[k].copy())
        rec[k]["logprior"].append(lp[k].copy())
        rec[k]["loglike"].append(ll[k].copy())
        rec[k]["accepted"].append(acc.copy())

    def step(k, snap_at=None):
        """one local step of level k; returns its accept mask"""
        if k == 0:
            t = cnt[0]
            cand = prop.propose(th[0], z[:, t])
            lpn, lln, _ = levels[0].evaluate(cand)
            alpha = _acceptance(prop.kind, lpn, lln, lp[0], ll[0])
            acc = u_levels[0][:, t] < alpha
            th[0] = np.where(acc[:, None], cand, th[0])
            lp[0] = np.where(acc, lpn, lp[0])
            ll[0] = np.where(acc, lln, ll[0])
            prop.accepted.append(acc.copy())
            prop.adapt(th[0])  # proposal.py:1607-1611 / chain.py:440-444
            record(0, acc)
            cnt[0] += 1
            return acc
        L = sl[k - 1]
        it = cnt[k]
        start_lp, start_ll = S[(k - 1, k)]  # level k-1 at the subchain start = at theta_k
        any_acc = np.zeros(N, dtype=bool)
        # DA only: which coarse state is promoted (chain.py:369-375); -1 = last
        pick = None
        if ridx is not None and nl == 2 and k == 1:
            pick = ridx[:, it]
            pick = np.where(np.isnan(pick), -1, pick).astype(int) + L  # 0-based step index whose result is promoted
            y_th, y_lp, y_ll = th[0].copy(), lp[0].copy(), ll[0].copy()
        for i in range(L):
            a = step(k - 1)
            any_acc |= a
            if pick is not None:
                sel = pick == i
                y_th[sel], y_lp[sel], y_ll[sel] = th[0][sel], lp[0][sel], ll[0][sel]
        if pick is None:
            y_th, y_lp, y_ll = th[k - 1], lp[k - 1], ll[k - 1]
        lpn, lln, _ = levels[k].evaluate(y_th)  # only meaningful where any_acc (skip-eval rule)
        with np.errstate(over="ignore", invalid="ignore"):
            alpha = np.exp((lpn + lln) - (lp[k] + ll[k]) + (start_lp + start_ll) - (y_lp + y_ll))  # chain.py:475-483
        acc = any_acc & (u_levels[k][:, it] < alpha)
        # accept: level k takes y; with a promoted intermediate state the coarse chain restarts from it
        th[k] = np.where(acc[:, None], y_th, th[k])
        lp[k] = np.where(acc, lpn, lp[k])
        ll[k] = np.where(acc, lln, ll[k])
        for j in range(k):
            if j == k - 1:
                tj, lj, ljl = y_th, y_lp, y_ll
            else:
                tj, lj, ljl = th[j], lp[j], ll[j]
            th[j] = np.where(acc[:, None], tj, th[k])  # reject: everything below reverts to theta_k
            lp[j] = np.where(acc, lj, S[(j, k)][0])
            ll[j] = np.where(acc, ljl, S[(j, k)][1])
        for j in range(k):
            for q in range(j + 1, k + 1):
                S[(j, q)] = (lp[j].copy(), ll[j].copy())
        prop.accepted.append(acc.copy())  # alignment entry on the base list (chain.py:363,389,397; proposal.py:1486)
        record(k, acc)
        cnt[k] += 1
        return acc

    for _ in range(n_fine):
        step(nl - 1)
    out = []
    for k in range(nl):
        r = rec[k]
        lpk, llk = np.array(r["logprior"]).T, np.array(r["loglike"]).T
        out.append(dict(theta=np.swapaxes(np.array(r["theta"]), 0, 1), logprior=lpk, loglike=llk, logpost=lpk + llk,
                        accepted=np.array(r["accepted"]).T.astype(np.uint8)))
    return out, prop


# ----------------------------------------------------------------------------------------
# DREAM(Z) (proposal.py:608-852) inside the single-level chain (chain.py:95-125)
# ----------------------------------------------------------------------------------------
def rosenbrock_forward(theta, a=1.0, b=10.0):
    """d-dimensional chain of examples/MALA Rosenbrock.ipynb's scalar 'forward model'; returns [N, 1]."""
    t = np.atleast_2d(theta)
    return np.sum((a - t[:, :-1]) ** 2 + b * (t[:, 1:] - t[:, :-1] ** 2) ** 2, axis=1, keepdims=True)


class RosenbrockLevel:
    """Posterior with the Rosenbrock forward model, data [0], isotropic unit-variance likelihood."""

    def __init__(self, prior, a=1.0, b=10.0, data=0.0, var=1.0):
        self.prior, self.a, self.b, self.data, self.var = prior, a, b, data, var

    def evaluate(self, theta):
        F = rosenbrock_forward(theta, self.a, self.b)
        return self.prior.logpdf(theta), loglike_isotropic(F, np.array([self.data]), self.var), F


def dreamz_jump(theta, Zr1, Zr2, mcr, sub_u, forced, e_u, eps_n, scaling, nCR, delta, b, b_star):
    """DREAMZ.make_proposal (proposal.py:811-852) for a batch, given the variates it would have drawn:
    Zr1 / Zr2 [N, d] = sums of the selected archive rows, mcr [N] crossover index, sub_u [N, d] subspace
    uniforms, forced [N] index used when the subspace came out empty, e_u [N, d] uniforms mapped to (-b, b),
    eps_n [N, d] standard normals scaled by b_star."""
    N, d = theta.shape
    CR = (mcr + 1) / nCR
    ind = (sub_u < CR[:, None]).astype(float)
    empty = ind.sum(axis=1) == 0
    rows = np.nonzero(empty)[0]
    ind[rows, forced[rows].astype(int)] = 1.0
    gamma = scaling * 2.38 / np.sqrt(2 * delta * ind.sum(axis=1))
    e = -b + (b - (-b)) * e_u
    eps = 0.0 + b_star * eps_n
    return theta + ind * ((np.ones(d) + e) * gamma[:, None] * (Zr1 - Zr2) + eps)


def run_dreamz(level, cfg, theta0, Z0, var):
    """N chains x T steps of Chain.sample with a DREAMZ proposal on recorded variates.
    cfg: M0, delta, nCR, adaptive, period, gamma, b, b_star.  Z0 [N, M0, d] initial archives.
    var: r [N,T,delta,2], mcr [N,T], sub_u [N,T,d], forced [N,T], e_u [N,T,d], eps_n [N,T,d], u [N,T]."""
    theta0 = np.asarray(theta0, dtype=float)
    N, d = theta0.shape
    T = var["u"].shape[1]
    M0, delta, nCR = int(cfg["M0"]), int(cfg["delta"]), int(cfg["nCR"])
    adaptive, period, gam = bool(cfg["adaptive"]), int(cfg["period"]), float(cfg["gamma"])
    b, b_star = float(cfg["b"]), float(cfg["b_star"])
    Z = np.empty((N, M0 + T, d))
    Z[:, :M0] = Z0
    M = M0
    scaling = np.ones(N)  # proposal.py:715
    pCR = np.full((N, nCR), 1.0 / nCR)
    LCR = np.zeros((N, nCR))
    DeltaCR = np.ones((N, nCR))
    theta = theta0.copy()
    lp, ll, _ = level.evaluate(theta)
    out_theta = np.empty((N, T + 1, d))
    out_lp, out_ll = np.empty((N, T + 1)), np.empty((N, T + 1))
    out_acc = np.ones((N, T + 1), dtype=np.uint8)
    out_theta[:, 0], out_lp[:, 0], out_ll[:, 0] = theta, lp, ll
    t = k = 0
    rows = np.arange(N)
    for s in range(T):
        r = var["r"][:, s].astype(int)  # [N, delta, 2]
        Zr1 = sum(Z[rows, r[:, i, 0]] for i in range(delta))
        Zr2 = sum(Z[rows, r[:, i, 1]] for i in range(delta))
        mcr = var["mcr"][:, s].astype(int)
        prop = dreamz_jump(theta, Zr1, Zr2, mcr, var["sub_u"][:, s], var["forced"][:, s], var["e_u"][:, s],
                           var["eps_n"][:, s], scaling, nCR, delta, b, b_star)
        lpn, lln, _ = level.evaluate(prop)
        alpha = _acceptance("grw", lpn, lln, lp, ll)
        acc = var["u"][:, s] < alpha
        prev = theta
        theta = np.where(acc[:, None], prop, theta)
        lp, ll = np.where(acc, lpn, lp), np.where(acc, lln, ll)
        out_theta[:, s + 1], out_lp[:, s + 1], out_ll[:, s + 1], out_acc[:, s + 1] = theta, lp, ll, acc
        # ---- adapt (proposal.py:790-809 after :228-245) ----
        t += 1
        boundary = adaptive and t % period == 0
        if boundary:
            rate = out_acc[:, : s + 2][:, -period:].mean(axis=1)
            scaling = np.exp(np.log(scaling) + gam ** -k * (rate - 0.24))
            k += 1
        Z[:, M] = theta
        M += 1
        if boundary:
            jd = theta - prev
            inc = (jd ** 2 / np.var(Z[:, :M], axis=1)).sum(axis=1)
            DeltaCR[rows, mcr] += inc
            LCR[rows, mcr] += 1
            ok = np.all(LCR > 0, axis=1)
            mean = DeltaCR / np.where(LCR > 0, LCR, 1.0)
            pCR = np.where(ok[:, None], mean / mean.sum(axis=1, keepdims=True), pCR)
    return dict(theta=out_theta, logprior=out_lp, loglike=out_ll, logpost=out_lp + out_ll, accepted=out_acc,
                scaling=scaling, pCR=pCR, archive=Z)


# ----------------------------------------------------------------------------------------
# Adaptive error model (Cui et al. 2019) on top of DA / MLDA
#   DAChain: chain.py:268-305 (set-up), :446-473 (state-dependent acceptance), :485-523 (update)
#   MLDAChain / MLDA: chain.py:643-678, :739-765; proposal.py:1407-1467, :1547-1578
# ----------------------------------------------------------------------------------------
def run_multilevel_aem(levels, proposal, subchain_lengths, theta0, z, u_levels, n_fine, aem, diagonal=False):
    """Like run_multilevel, with tinyDA's adaptive error model.

    levels[k] = dict(A, b, y, cov | var): every level but the finest has an AdaptiveGaussianLogLike (dense `cov`,
    distributions.py:332-449), the finest an isotropic one (`var`).  All levels share the output dimension.
    aem = 'state-independent' (DA and MLDA) or 'state-dependent' (DA only).
    diagonal=True (extension, not in tinyDA): set_bias receives only the DIAGONAL of the bias covariance -- the scalable
    variant of the error model (include/tinyda_amd.h, TDA_AEM_STATE_INDEPENDENT_DIAGONAL).  The trackers follow the
    reference's full recursion; their off-diagonal entries simply never reach the likelihood.
    Book-keeping that differs from run_multilevel: a Link's likelihood can be refreshed later (update_link), and
    align_chain looks links up by *identity* of their parameter array (proposal.py:1481-1483), so every state
    carries an id and S[j][q] means "the latest level-j link whose parameters are theta_q".
    """
    theta0 = np.asarray(theta0, dtype=float)
    N, d = theta0.shape
    nl = len(levels)
    sl = list(subchain_lengths)
    m = len(levels[0]["y"])  # a level is linear (A, optional b) or any batched callable theta[N, d] -> F[N, m] under "fn"
    prior = levels[0]["prior"]
    prop = _BaseProposalState(proposal, theta0, prior.cov)
    dependent = aem == "state-dependent"
    assert not dependent or nl == 2
    is_da = nl == 2

    def forward(k, theta):
        if levels[k].get("fn") is not None:
            return np.asarray(levels[k]["fn"](theta), dtype=float)
        F = theta @ levels[k]["A"].T
        return F if levels[k].get("b") is None else F + levels[k]["b"]

    cov_inv = [np.broadcast_to(np.linalg.inv(levels[k]["cov"]), (N, m, m)).copy() for k in range(nl - 1)]
    bias_tot = [np.zeros((N, m)) for _ in range(nl - 1)]

    def loglike(k, F, bias=None):
        if k == nl - 1:
            return loglike_isotropic(F, levels[k]["y"], levels[k]["var"])
        r = F + (bias_tot[k] if bias is None else bias) - levels[k]["y"]
        return -0.5 * np.einsum("ni,nij,nj->n", r, cov_inv[k], r)

    def set_bias(k, mu, sigma):  # distributions.py:385-402, per chain
        bias_tot[k] = mu.copy()
        if diagonal:
            sigma = sigma * np.eye(m)
        refresh = ~np.all(sigma < 1e-9, axis=(1, 2))
        if refresh.any():
            cov_inv[k][refresh] = np.linalg.inv(levels[k]["cov"] + sigma[refresh])

    th = [theta0.copy() for _ in range(nl)]
    F = [forward(k, theta0) for k in range(nl)]
    lp = [prior.logpdf(theta0) for _ in range(nl)]
    ll = [loglike(k, F[k]) for k in range(nl)]
    sid = [np.zeros(N, dtype=np.int64) for _ in range(nl)]
    # error-model trackers of levels q >= 1
    mdiff = {q: F[q] - F[q - 1] for q in range(1, nl)}
    b_mu = {q: mdiff[q].copy() for q in range(1, nl)}
    b_sig = {q: np.zeros((N, m, m)) for q in range(1, nl)}
    b_t = {q: 1 for q in range(1, nl)}
    for q in range(nl - 1, 0, -1):  # chain.py:286-305, :659-678; proposal.py:1442-1467
        if dependent:
            set_bias(q - 1, mdiff[q], b_sig[q])
        else:
            set_bias(q - 1, sum(b_mu[p] for p in range(q, nl)), sum(b_sig[p] for p in range(q, nl)))
        ll[q - 1] = loglike(q - 1, F[q - 1])
    S = {(j, q): (lp[j].copy(), ll[j].copy(), F[j].copy()) for q in range(nl) for j in range(q)}
    rec = [dict(theta=[], logprior=[], loglike=[], accepted=[]) for _ in range(nl)]
    rec[nl - 1]["theta"].append(th[nl - 1].copy())
    rec[nl - 1]["logprior"].append(lp[nl - 1].copy())
    rec[nl - 1]["loglike"].append(ll[nl - 1].copy())
    rec[nl - 1]["accepted"].append(np.ones(N, dtype=bool))
    rec_slot = [[] for _ in range(nl)]  # per record: the state id, to apply later likelihood refreshes of stored links
    cnt = [0] * nl
    next_id = [1]

    def record(k, acc):
        rec[k]["theta"].append(th[k].copy())
        rec[k]["logprior"].append(lp[k].copy())
        rec[k]["loglike"].append(ll[k].copy())
        rec[k]["accepted"].append(acc.copy())

    def pcn_q(x, y):  # proposal.py:364-369
        beta = prop.scaling
        dev = y - np.sqrt(1 - beta ** 2)[:, None] * x
        U = prior.U
        maha = np.sum(np.square(dev @ U), axis=1) / beta ** 2
        return -0.5 * (d * LOG_2PI + prior.log_pdet + d * np.log(beta ** 2) + maha)

    def step(k):
        if k == 0:
            t = cnt[0]
            cand = prop.propose(th[0], z[:, t])
            Fn = forward(0, cand)
            lpn, lln = prior.logpdf(cand), loglike(0, Fn)
            alpha = _acceptance(prop.kind, lpn, lln, lp[0], ll[0])
            acc = u_levels[0][:, t] < alpha
            th[0] = np.where(acc[:, None], cand, th[0])
            F[0] = np.where(acc[:, None], Fn, F[0])
            lp[0], ll[0] = np.where(acc, lpn, lp[0]), np.where(acc, lln, ll[0])
            sid[0] = np.where(acc, next_id[0], sid[0])
            next_id[0] += 1
            prop.accepted.append(acc.copy())
            prop.adapt(th[0])
            record(0, acc)
            cnt[0] += 1
            return acc
        L = sl[k - 1]
        it = cnt[k]
        start_lp, start_ll, start_F = S[(k - 1, k)]
        any_acc = np.zeros(N, dtype=bool)
        for _ in range(L):
            any_acc |= step(k - 1)
        y_th, y_lp, y_ll, y_F, y_id = th[k - 1], lp[k - 1], ll[k - 1], F[k - 1], sid[k - 1]
        Fq = forward(k, y_th)
        lpn, lln = prior.logpdf(y_th), loglike(k, Fq)
        with np.errstate(over="ignore", invalid="ignore"):
            if dependent:  # chain.py:446-473
                bias_next = Fq - y_F
                ll_biased = loglike(k - 1, start_F, bias_next)
                if prop.kind == "pcn":
                    q_xy, q_yx = pcn_q(th[k], y_th), pcn_q(y_th, th[k])
                else:
                    q_xy = q_yx = 0.0
                alpha = np.exp(np.minimum(lpn + lln + q_yx, start_lp + ll_biased + q_xy)
                               - np.minimum(lp[k] + ll[k] + q_xy, y_lp + y_ll + q_yx))
            else:
                alpha = np.exp((lpn + lln) - (lp[k] + ll[k]) + (start_lp + start_ll) - (y_lp + y_ll))
        acc = any_acc & (u_levels[k][:, it] < alpha)
        th[k] = np.where(acc[:, None], y_th, th[k])
        F[k] = np.where(acc[:, None], Fq, F[k])
        lp[k], ll[k] = np.where(acc, lpn, lp[k]), np.where(acc, lln, ll[k])
        sid[k] = np.where(acc, y_id, sid[k])
        for j in range(k):  # reject: everything below returns to the latest link holding theta_k
            th[j] = np.where(acc[:, None], th[j], th[k])
            F[j] = np.where(acc[:, None], F[j], S[(j, k)][2])
            lp[j] = np.where(acc, lp[j], S[(j, k)][0])
            ll[j] = np.where(acc, ll[j], S[(j, k)][1])
            sid[j] = np.where(acc, sid[j], sid[k])
        for j in range(k):
            for q in range(j + 1, k + 1):
                S[(j, q)] = (lp[j].copy(), ll[j].copy(), F[j].copy())
        prop.accepted.append(acc.copy())
        # ---- error model update (chain.py:485-523 for DA; :739-765 and proposal.py:1547-1578 for MLDA) ----
        new_diff = F[k] - F[k - 1]
        if dependent:
            corrected = F[k] - (F[k - 1] + mdiff[k])  # chain.py:505-507
            mdiff[k] = new_diff
            b_sig[k] = zero_mean_moments_update(b_sig[k], b_t[k], corrected)
            b_t[k] += 1
            set_bias(k - 1, mdiff[k], b_sig[k])
        else:
            mdiff[k] = new_diff if is_da else np.where(acc[:, None], new_diff, mdiff[k])  # MLDA refreshes on accept only
            b_mu[k], b_sig[k] = moments_update(b_mu[k], b_sig[k], b_t[k], mdiff[k])
            b_t[k] += 1
            set_bias(k - 1, sum(b_mu[p] for p in range(k, nl)), sum(b_sig[p] for p in range(k, nl)))
        ll[k - 1] = loglike(k - 1, F[k - 1])  # update_link of the latest link one level down
        for q in range(k, nl):  # every "latest link with parameters theta_q" that is this very link
            same = sid[q] == sid[k - 1]
            o = S[(k - 1, q)]
            S[(k - 1, q)] = (o[0], np.where(same, ll[k - 1], o[1]), o[2])
        record(k, acc)
        cnt[k] += 1
        return acc

    for _ in range(n_fine):
        step(nl - 1)
    out = []
    for k in range(nl):
        r = rec[k]
        lpk, llk = np.array(r["logprior"]).T, np.array(r["loglike"]).T
        out.append(dict(theta=np.swapaxes(np.array(r["theta"]), 0, 1), logprior=lpk, loglike=llk, logpost=lpk + llk,
                        accepted=np.array(r["accepted"]).T.astype(np.uint8)))
    return out, dict(bias=bias_tot, b_mu=b_mu, b_sigma=b_sig, cov_inv=cov_inv)


# ----------------------------------------------------------------------------------------
# the reference's cost profile (bench.py cpu_baseline): ONE chain at a time, every call the reference
# makes per step -- scipy's frozen multivariate_normal.logpdf (posterior.py:92), the forward model, the isotropic
# log-likelihood (distributions.py:295-298), np.random.multivariate_normal with its SVD of the proposal covariance on every
# draw (proposal.py:249-251), the accept test (chain.py:104-119) and the RecursiveSampleMoments update with its three outer
# products (utils.py:113-122; proposal.py:509-510 for the swap).  Timing aid, not a parity path.
# ----------------------------------------------------------------------------------------
def reference_shaped_am_chain(A, y, sigma2, theta0, n_steps, C0, t0=100, period=100, epsilon=1e-6, seed=0):
    import scipy.stats as stats

    d = A.shape[1]
    rs = np.random.RandomState(seed)
    prior = stats.multivariate_normal(np.zeros(d), np.eye(d))
    sd = min(1.0, 2.4 ** 2 / d)
    theta = np.array(theta0, dtype=float)
    r = A @ theta - y
    logpost = prior.logpdf(theta) + -0.5 * (r @ r) / sigma2
    C = np.array(C0, dtype=float)
    mu, sigma, t = theta.copy(), np.zeros((d, d)), 1
    accepted = 0
    for k in range(n_steps):
        prop = theta + rs.multivariate_normal(np.zeros(d), C)
        r = A @ prop - y
        lp = prior.logpdf(prop) + -0.5 * (r @ r) / sigma2
        with np.errstate(over="ignore"):
            alpha = np.exp(lp - logpost)
        if rs.random_sample() < alpha:
            theta, logpost = prop, lp
            accepted += 1
        mu_new = (1 / (t + 1)) * (t * mu + theta)
        sigma = (t - 1) / t * sigma + sd / t * (t * np.outer(mu, mu) - (t + 1) * np.outer(mu_new, mu_new) + np.outer(theta, theta) + epsilon * np.eye(d))
        mu = mu_new
        t += 1
        if (k + 1) >= t0 and (k + 1) % period == 0:
            C = sigma.copy()
    return theta, accepted / max(n_steps, 1)
