"""Running sample moments with the reference's update order (tinyDA/utils.py:9-201), host versions.
The device twin of the recursion is k_adapt in csrc/tda_kernels.h; the error-model trackers live in k_aem_action."""
import numpy as np


class _RunningMoments:
    """state (mu, sigma, t) plus the accessor protocol tinyDA's callers use"""

    mu = None

    def get_mu(self):
        return self.mu

    def get_sigma(self):
        return self.sigma


class RecursiveSampleMoments(_RunningMoments):
    """mean / covariance recursion used by AdaptiveMetropolis (with sd, epsilon) and by the state-independent error
    model (sd = 1, epsilon = 0); the counter starts at 1 because the initial point counts as the first sample."""

    def __init__(self, mu0, sigma0, t=1, sd=1, epsilon=0):
        self.mu, self.sigma, self.t = mu0, sigma0, t
        self.d = mu0.shape[0]
        self.sd, self.epsilon = sd, epsilon

    def __call__(self):
        return self.mu, self.sigma

    def update(self, x):
        t, old = self.t, self.mu
        new = (1 / (t + 1)) * (t * old + x)
        spread = t * np.outer(old, old) - (t + 1) * np.outer(new, new) + np.outer(x, x) + self.epsilon * np.eye(self.d)
        self.sigma = (t - 1) / t * self.sigma + self.sd / t * spread
        self.mu, self.t = new, t + 1


class ZeroMeanRecursiveSampleMoments(_RunningMoments):
    """second moment about zero, used by the state-dependent error model"""

    def __init__(self, sigma0, t=1):
        self.sigma, self.t = sigma0, t
        self.d = sigma0.shape[0]

    def __call__(self):
        return self.sigma

    def update(self, x):
        self.sigma = (self.t - 1) / self.t * self.sigma + 1 / self.t * np.outer(x, x)
        self.t += 1
