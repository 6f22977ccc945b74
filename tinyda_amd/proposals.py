"""Transition kernels with the reference's constructor signatures and proposal protocol
(tinyDA/proposal.py:17-41): setup_proposal / adapt / make_proposal / get_acceptance / get_q.

On the device path `sample()` never calls these methods per step: `_lowering()` hands the constructor
arguments to the HIP engine, which runs propose -> evaluate -> accept -> adapt fused.  The methods exist so
that the same objects also drive the host loop for opaque Python models (BASELINE config 1), and so user
code written against tinyDA keeps working.
"""
import numpy as np
import scipy.stats as stats

from . import _lib
from .moments import RecursiveSampleMoments


class Proposal:
    is_symmetric = False

    def setup_proposal(self, **kwargs):
        pass

    def adapt(self, **kwargs):
        pass

    def make_proposal(self, link):
        pass

    def get_acceptance(self, proposal_link, previous_link):
        pass

    def get_q(self, x_link, y_link):
        pass


class IndependenceSampler(Proposal):
    """Independent proposals from a fixed distribution q with .rvs() and .logpdf() (proposal.py:65-129).  On the device
    path q must be Gaussian (a frozen scipy.stats.multivariate_normal, or any object with `mean` and `cov`)."""

    def __init__(self, q):
        self.q = q
        self.q.logpdf(self.q.rvs(1))  # the reference's interface check (proposal.py:101-105)

    def make_proposal(self, link):
        return self.q.rvs(1).flatten()

    def get_acceptance(self, proposal_link, previous_link):
        q_proposal = self.get_q(None, proposal_link)
        q_previous = self.get_q(None, previous_link)
        return np.exp(proposal_link.posterior - previous_link.posterior + q_previous - q_proposal)

    def get_q(self, x_link, y_link):
        return self.q.logpdf(y_link.parameters)

    def _lowering(self):
        mean, cov = getattr(self.q, "mean", None), getattr(self.q, "cov", None)
        if mean is None or cov is None:
            return None
        return dict(kind=_lib.PROP_INDEPENDENCE, C_=np.atleast_2d(np.asarray(cov, dtype=np.float64)),
                    q_mean=np.atleast_1d(np.asarray(mean, dtype=np.float64)))


def _require_square(C, name):
    # same checks and messages as proposal.py:189-196 / :443-450
    if not isinstance(C, np.ndarray):
        raise TypeError("%s must be a numpy array" % name)
    if C.ndim == 1:
        if C.shape[0] != 1:
            raise ValueError("%s must be an NxN array" % name)
    elif C.shape[0] != C.shape[1]:
        raise ValueError("%s must be an NxN array" % name)


class GaussianRandomWalk(Proposal):
    """theta' = theta + scaling * N(0, C), optional global scaling adaptation (proposal.py:132-258)."""

    is_symmetric = True
    alpha_star = 0.24

    def __init__(self, C, scaling=1, adaptive=False, gamma=1.01, period=100):
        _require_square(C, "C")
        self.C = C
        self.d = C.shape[0]
        self._init_scaling(scaling, adaptive, gamma, period)

    def _init_scaling(self, scaling, adaptive, gamma, period):
        self.scaling = scaling
        self.adaptive = adaptive
        self.gamma = gamma
        self.period = period
        self.k = 0  # completed scaling adaptations (diminishing adaptation)
        self.t = 0  # adapt() calls
        self._factor_of = None

    def _draw(self):
        """N(0, C) as chol(C) z.  (The reference asks NumPy, which re-runs an SVD of C on every call;
        the law is the same.)"""
        if self._factor_of is not self.C:
            self._L = np.linalg.cholesky(np.atleast_2d(self.C))
            self._factor_of = self.C
        return self._L @ np.random.standard_normal(self.d)

    def adapt(self, **kwargs):
        self.t += 1
        if self.adaptive and self.t % self.period == 0:
            rate = np.mean(kwargs["accepted"][-self.period:])
            self.scaling = np.exp(np.log(self.scaling) + self.gamma ** -self.k * (rate - self.alpha_star))
            self.k += 1

    def make_proposal(self, link):
        return link.parameters + self.scaling * self._draw()

    def get_acceptance(self, proposal_link, previous_link):
        if np.isnan(proposal_link.posterior):
            return 0
        return np.exp(proposal_link.posterior - previous_link.posterior)

    def _lowering(self):
        return dict(kind=_lib.PROP_GRW, C_=np.atleast_2d(self.C), scaling=float(self.scaling),
                    adaptive=bool(self.adaptive), gamma=float(self.gamma), period=int(self.period))


class CrankNicolson(GaussianRandomWalk):
    """pCN: theta' = sqrt(1 - beta^2) theta + beta N(0, C_prior); acceptance on the likelihood ratio
    (proposal.py:261-369).  The prior mean is ignored, as in the reference."""

    is_symmetric = False

    def __init__(self, scaling=0.1, adaptive=False, gamma=1.01, period=100):
        self._init_scaling(scaling, adaptive, gamma, period)

    def setup_proposal(self, **kwargs):
        prior = kwargs["posterior"].prior
        self.C = prior.cov if hasattr(prior, "cov") else prior.cov_object.covariance
        self.d = self.C.shape[0]

    def make_proposal(self, link):
        return np.sqrt(1 - self.scaling ** 2) * link.parameters + self.scaling * self._draw()

    def get_acceptance(self, proposal_link, previous_link):
        if np.isnan(proposal_link.posterior):
            return 0
        return np.exp(proposal_link.likelihood - previous_link.likelihood)

    def get_q(self, x_link, y_link):
        return stats.multivariate_normal.logpdf(
            y_link.parameters, mean=np.sqrt(1 - self.scaling ** 2) * x_link.parameters, cov=self.scaling ** 2 * self.C
        )

    def _lowering(self):
        return dict(kind=_lib.PROP_PCN, C_=None, scaling=float(self.scaling), adaptive=bool(self.adaptive),
                    gamma=float(self.gamma), period=int(self.period))


class OperatorWeightedCrankNicolson(CrankNicolson):
    """Operator-weighted pCN (Law 2014): theta' = sqrtm(I - scaling B) theta + sqrtm(scaling B) N(0, C_prior), acceptance on
    the likelihood ratio (proposal.py:515-605).  The operators are recomputed when the adaptive scaling changes.  On the
    device path the fixed operators are handed to the engine; with adaptive=True and a symmetric B the engine gets B's spectrum and
    works the per-chain operators out from each chain's own scaling (single-level chains, linear forward model)."""

    def __init__(self, B, scaling=1.0, adaptive=False, gamma=1.01, period=100):
        self.B = B
        super().__init__(scaling, adaptive, gamma, period)

    def _operators(self):
        from scipy.linalg import sqrtm

        d = np.atleast_2d(self.B).shape[0]
        self.state_operator = np.real(sqrtm(np.eye(d) - self.scaling * self.B))
        self.noise_operator = np.real(sqrtm(self.scaling * self.B))

    def setup_proposal(self, **kwargs):
        super().setup_proposal(**kwargs)
        self._operators()

    def adapt(self, **kwargs):
        super().adapt(**kwargs)
        if self.adaptive and self.t % self.period == 0:
            self._operators()

    def make_proposal(self, link):
        return np.dot(self.state_operator, link.parameters) + np.dot(self.noise_operator, self._draw())

    def get_q(self, x_link, y_link):
        return stats.multivariate_normal.logpdf(
            y_link.parameters, mean=np.dot(self.state_operator, x_link.parameters), cov=np.dot(self.scaling * self.B, self.C)
        )

    def _lowering(self):
        if self.adaptive:
            # per-chain operators (every chain adapts its own scaling): for a symmetric B they are functions of its spectrum,
            # which is what the engine takes (tda_engine_set_proposal_spectrum); any other B: host protocol
            B = np.atleast_2d(np.asarray(self.B, dtype=np.float64))
            if B.shape[0] != B.shape[1] or not np.allclose(B, B.T, rtol=1e-12, atol=1e-14):
                return None
            lam, V = np.linalg.eigh(0.5 * (B + B.T))
            return dict(kind=_lib.PROP_OWCN, C_=None, scaling=float(self.scaling), adaptive=True, gamma=float(self.gamma),
                        period=int(self.period), spectrum=(np.ascontiguousarray(V), np.ascontiguousarray(lam)))
        self._operators()
        return dict(kind=_lib.PROP_OWCN, C_=None, scaling=1.0, adaptive=False, gamma=float(self.gamma), period=int(self.period),
                    state_operator=np.ascontiguousarray(self.state_operator, dtype=np.float64),
                    noise_operator=np.ascontiguousarray(self.noise_operator, dtype=np.float64))


def _grad_log_prior(x, prior):
    """utils.py:273-280 for a frozen scipy multivariate normal (anything with mean / cov); finite differences otherwise."""
    cov = getattr(prior, "cov", None)
    if cov is None and hasattr(prior, "cov_object"):
        cov = prior.cov_object.covariance
    if cov is not None and hasattr(prior, "mean"):
        return np.dot(np.linalg.inv(np.atleast_2d(cov)), (prior.mean - x))
    from scipy.optimize import approx_fprime

    return approx_fprime(x, prior.logpdf)


class MALA(GaussianRandomWalk):
    """Metropolis-adjusted Langevin: theta' = theta + scaling^2/2 grad log post(theta) + scaling N(0, I), acceptance with
    the two transition densities (proposal.py:861-1005).  The gradient is exact when the model has a
    `gradient(parameters, sensitivity)` method and the likelihood a `grad_loglike`, finite differences of
    `posterior.logpdf` otherwise.  On the device path (linear model, Gaussian prior) the gradient is c - H theta with
    H = Sigma_prior^-1 + A^T Sigma_e^-1 A precomputed once: a d x d product per step instead of a second pass over the
    observations."""

    is_symmetric = False
    alpha_star = 0.57

    def __init__(self, scaling=0.1, adaptive=False, gamma=1.01, period=100):
        self._init_scaling(scaling, adaptive, gamma, period)

    def setup_proposal(self, **kwargs):
        self.posterior = kwargs["posterior"]
        self.d = np.asarray(self.posterior.prior.rvs()).size
        exact = callable(getattr(self.posterior.model, "gradient", None)) and hasattr(self.posterior.likelihood, "grad_loglike")
        self.compute_gradient = self._compute_gradient if exact else self._compute_gradient_approx

    def _compute_gradient(self, link):
        sens = self.posterior.likelihood.grad_loglike(link.model_output)
        return _grad_log_prior(link.parameters, self.posterior.prior) + self.posterior.model.gradient(link.parameters, sens)

    def _compute_gradient_approx(self, link):
        from scipy.optimize import approx_fprime

        return approx_fprime(link.parameters, self.posterior.logpdf)

    def make_proposal(self, link):
        if not hasattr(link, "gradient"):
            link.gradient = self.compute_gradient(link)
        return link.parameters + 0.5 * self.scaling ** 2 * link.gradient + self.scaling * np.random.standard_normal(self.d)

    def get_acceptance(self, proposal_link, previous_link):
        if np.isnan(proposal_link.posterior):
            return 0
        if not hasattr(proposal_link, "gradient"):
            proposal_link.gradient = self.compute_gradient(proposal_link)
        q_x_y = self.get_q(previous_link, proposal_link)
        q_y_x = self.get_q(proposal_link, previous_link)
        return np.exp(proposal_link.posterior - previous_link.posterior + q_x_y - q_y_x)

    def get_q(self, x_link, y_link):
        return -0.5 / self.scaling ** 2 * np.linalg.norm(
            x_link.parameters - y_link.parameters - 0.5 * self.scaling ** 2 * y_link.gradient) ** 2

    def _lowering(self):
        return dict(kind=_lib.PROP_MALA, C_=None, scaling=float(self.scaling), adaptive=bool(self.adaptive),
                    gamma=float(self.gamma), period=int(self.period))


class AdaptiveMetropolis(GaussianRandomWalk):
    """Haario et al. (2001): proposal covariance <- running sample covariance every `period` adapt calls
    once t >= t0 (proposal.py:372-512).

    `block_moments` (extension, device path only, default off): update the running covariance once per block of steps
    in closed form on the matrix cores instead of following RecursiveSampleMoments.update operation for operation --
    algebraically identical, ~8x cheaper, differs from the reference recursion by that recursion's rounding error."""

    def __init__(self, C0, sd=None, epsilon=1e-6, t0=0, period=100, adaptive=False, gamma=1.01, block_moments=False):
        _require_square(C0, "C0")
        self.block_moments = bool(block_moments)
        self.C = C0
        self.d = C0.shape[0]
        self._init_scaling(1, adaptive, gamma, period)
        self.sd = sd if sd is not None else min(1, 2.4 ** 2 / self.d)
        self.epsilon = epsilon
        self.t0 = t0

    def setup_proposal(self, **kwargs):
        self.AM_recursor = RecursiveSampleMoments(
            kwargs["parameters"], np.zeros((self.d, self.d)), sd=self.sd, epsilon=self.epsilon
        )

    def adapt(self, **kwargs):
        super().adapt(**kwargs)
        self.AM_recursor.update(kwargs["parameters"])
        if self.t >= self.t0 and self.t % self.period == 0:
            self.C = self.AM_recursor.get_sigma()

    def _lowering(self):
        return dict(kind=_lib.PROP_AM, C_=np.atleast_2d(self.C), adaptive=bool(self.adaptive), gamma=float(self.gamma),
                    period=int(self.period), sd=float(self.sd), epsilon=float(self.epsilon), t0=int(self.t0),
                    block_moments=self.block_moments)


class DREAMZ(GaussianRandomWalk):
    """DREAM(Z): differential-evolution jumps from an archive of past states (proposal.py:608-852).
    Host protocol = reference semantics (per-chain archive, np.random global stream); on the device path the
    constructor arguments are lowered and the archive lives in HBM."""

    def __init__(self, M0, delta=1, b=5e-2, b_star=1e-6, Z_method="random", nCR=3, adaptive=False, gamma=1.01, period=100):
        self.M = M0
        self.M0 = M0
        self._init_scaling(1, adaptive, gamma, period)
        self.delta, self.b, self.b_star = delta, b, b_star
        self.Z_method = Z_method
        self.nCR = nCR
        self.mCR = None
        self.pCR = np.array(nCR * [1 / nCR])

    def setup_proposal(self, **kwargs):
        prior = kwargs["posterior"].prior
        self.d = prior.rvs().size
        if self.adaptive:
            self.LCR = np.zeros(self.nCR)
            self.DeltaCR = np.ones(self.nCR)
        if self.Z_method == "lhs":
            import scipy.stats as st

            u = st.qmc.LatinHypercube(d=self.d).random(n=self.M)
            if hasattr(prior, "ppf"):
                self.Z = prior.ppf(u)
                return
            if hasattr(prior, "mean") and (hasattr(prior, "cov") or hasattr(prior, "cov_object")):
                cov = prior.cov if hasattr(prior, "cov") else prior.cov_object.covariance
                self.Z = st.norm(loc=np.asarray(prior.mean), scale=np.sqrt(np.diag(cov))).ppf(u)
                return
        self.Z = prior.rvs(self.M)

    def adapt(self, **kwargs):
        GaussianRandomWalk.adapt(self, **kwargs)
        self.Z = np.vstack((self.Z, kwargs["parameters"]))
        self.M = self.Z.shape[0]
        if self.adaptive and self.t % self.period == 0:
            jump = kwargs["parameters"] - kwargs["parameters_previous"]
            self.DeltaCR[self.mCR] += (jump ** 2 / np.var(self.Z, axis=0)).sum()
            self.LCR[self.mCR] += 1
            if np.all(self.LCR > 0):
                mean = self.DeltaCR / self.LCR
                self.pCR = mean / mean.sum()

    def make_proposal(self, link, Z=None):
        Z = self.Z if Z is None else Z
        M = Z.shape[0]
        up, down = np.zeros(self.d), np.zeros(self.d)
        for _ in range(self.delta):
            r1, r2 = np.random.choice(M, 2, replace=False)
            up += Z[r1]
            down += Z[r2]
        self.mCR = np.random.choice(self.nCR, p=self.pCR)
        mask = (np.random.uniform(size=self.d) < (self.mCR + 1) / self.nCR).astype(float)
        if mask.sum() == 0:
            mask[np.random.choice(self.d)] = 1
        gamma = self.scaling * 2.38 / np.sqrt(2 * self.delta * mask.sum())
        e = np.random.uniform(-self.b, self.b, size=self.d)
        eps = np.random.normal(0, self.b_star, size=self.d)
        return link.parameters + mask * ((1 + e) * gamma * (up - down) + eps)

    _shared = False

    def _lowering(self):
        if self.Z_method not in ("random", "lhs"):
            return None
        return dict(kind=_lib.PROP_DREAMZ, Z_method=self.Z_method, M0=int(self.M0), delta=int(self.delta), b=float(self.b), b_star=float(self.b_star),
                    nCR=int(self.nCR), adaptive=bool(self.adaptive), gamma=float(self.gamma), period=int(self.period),
                    shared=self._shared)


class DREAM(DREAMZ):
    """DREAM with one archive shared by all chains (proposal.py:1627-1656; ray.py:365-384 ArchiveManager).
    The reference pushes rows to a Ray actor fire-and-forget, so what a proposal sees is timing dependent; the
    device engine synchronises the archive at block boundaries (RCCL all-gather across GPUs)."""

    _shared = True
