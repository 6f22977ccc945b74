"""Gaussian log-likelihoods with the reference's API (tinyDA/distributions.py:203-449).

All are *unnormalised* (-1/2 r^T Sigma^-1 r).  `GaussianLogLike` is the reference's factory: a diagonal
covariance with equal entries becomes isotropic, a diagonal one diagonal, anything else dense
(distributions.py:237-243).  Each class also reports how the device engine should see it (`_lowering`).
"""
import numpy as np

from . import _lib


def _check_covariance(data, covariance):
    # same exceptions as distributions.py:227-235 / :370-378
    if not isinstance(covariance, np.ndarray) or covariance.ndim != 2:
        raise TypeError("Covariance must be a 2-D numpy array.")
    if covariance.shape[0] != data.shape[0]:
        raise ValueError("Dimensions of data and covariance do not match.")
    if covariance.shape[0] != covariance.shape[1]:
        raise ValueError("Covariance must be an NxN array.")


class DefaultGaussianLogLike:
    """Dense covariance; the inverse is formed once (distributions.py:280)."""

    def __init__(self, data, covariance):
        self.data = data
        self.cov = covariance
        self.cov_inverse = np.linalg.inv(covariance)

    def _residual(self, x):
        return x - self.data

    def loglike(self, x):
        r = self._residual(x)
        return -0.5 * np.linalg.multi_dot((r.T, self.cov_inverse, r))

    def grad_loglike(self, x):
        return np.dot(self.cov_inverse, -self._residual(x))

    def _lowering(self):
        return _lib.NOISE_DENSE, np.asarray(self.cov, dtype=np.float64)


class DiagonalGaussianLogLike(DefaultGaussianLogLike):
    def __init__(self, data, covariance):
        self.data = data
        self.cov = np.diag(covariance)

    def loglike(self, x):
        return -0.5 * (self._residual(x) ** 2 / self.cov).sum()

    def grad_loglike(self, x):
        return 1 / self.cov * -self._residual(x)

    def _lowering(self):
        return _lib.NOISE_DIAG, np.asarray(self.cov, dtype=np.float64)


class IsotropicGaussianLogLike(DefaultGaussianLogLike):
    def __init__(self, data, variance):
        self.data = data
        self.var = variance

    def loglike(self, x):
        return -0.5 * np.linalg.norm(self._residual(x)) ** 2 / self.var

    def grad_loglike(self, x):
        return 1 / self.var * -self._residual(x)

    def _lowering(self):
        return _lib.NOISE_ISO, np.array([float(self.var)])


class AdaptiveGaussianLogLike(DefaultGaussianLogLike):
    """Bias-corrected dense likelihood for the adaptive error model (distributions.py:332-449)."""

    def __init__(self, data, covariance):
        _check_covariance(data, covariance)
        super().__init__(data, covariance)
        self.bias = np.zeros(self.data.shape[0])

    def set_bias(self, mean_bias, covariance_bias):
        self.bias = mean_bias
        self.cov_bias = covariance_bias
        # the reference leaves the inverse untouched while every entry is below 1e-9 (:399-402)
        if not np.all(self.cov_bias < 1e-9):
            self.cov_inverse = np.linalg.inv(self.cov + self.cov_bias)

    def _residual(self, x):
        return x + self.bias - self.data

    def loglike_custom_bias(self, x, bias):
        r = x + bias - self.data
        return -0.5 * np.linalg.multi_dot((r.T, self.cov_inverse, r))

    def _lowering(self):
        return _lib.NOISE_ADAPTIVE, np.asarray(self.cov, dtype=np.float64)


def GaussianLogLike(data, covariance):
    """Factory with the reference's dispatch and error behaviour (distributions.py:203-243)."""
    _check_covariance(data, covariance)
    diagonal = np.diag(covariance)
    if np.count_nonzero(covariance - np.diag(diagonal)) == 0:
        if np.all(diagonal == covariance[0, 0]):
            return IsotropicGaussianLogLike(data, covariance[0, 0])
        return DiagonalGaussianLogLike(data, covariance)
    return DefaultGaussianLogLike(data, covariance)


class JointPrior:
    """A list of independent scalar priors, one per parameter, in parameter order (tinyDA/distributions.py:8-100)."""

    def __init__(self, distributions):
        self.distributions = distributions
        self.dim = len(distributions)

    def logpdf(self, x):
        # independent components: the joint log-density is the sum of the marginals, accumulated in parameter order
        total = 0.0
        for dist, xi in zip(self.distributions, np.asarray(x).reshape(-1)):
            total = total + dist.logpdf(xi)
        return total

    def rvs(self, n_samples=1):
        # one column per component, drawn in parameter order (the order fixes which variates of the global stream go where)
        draws = np.column_stack([dist.rvs(size=n_samples) for dist in self.distributions])
        return draws[0] if n_samples == 1 else draws

    def ppf(self, x):
        # quantile transform of a [n, dim] array of uniforms (Latin hypercube archives), column by column
        return np.column_stack([dist.ppf(col) for dist, col in zip(self.distributions, np.asarray(x).T)])

    def _lowering(self):
        """(kinds, loc, scale) when every component is a frozen scipy norm or uniform, else None."""
        kinds, loc, scale = [], [], []
        for dist in self.distributions:
            name = getattr(getattr(dist, "dist", None), "name", None)
            if name not in ("norm", "uniform"):
                return None
            _, l, s = dist.dist._parse_args(*dist.args, **dist.kwds)
            kinds.append(0 if name == "norm" else 1)
            loc.append(float(l))
            scale.append(float(s))
        return np.array(kinds, dtype=np.int32), np.array(loc), np.array(scale)
