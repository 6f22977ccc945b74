"""Running sample moments (tinyDA/utils.py:9-201), host versions with the reference's update order.
The device twin of `RecursiveSampleMoments.update` is k_adapt in csrc/tda_kernels.h."""
import numpy as np


class RecursiveSampleMoments:
    def __init__(self, mu0, sigma0, t=1, sd=1, epsilon=0):
        self.mu = mu0
        self.d = mu0.shape[0]
        self.sigma = sigma0
        self.t = t
        self.sd = sd
        self.epsilon = epsilon

    def __call__(self):
        return self.mu, self.sigma

    def get_mu(self):
        return self.mu

    def get_sigma(self):
        return self.sigma

    def update(self, x):
        t, old = self.t, self.mu
        new = (1 / (t + 1)) * (t * old + x)
        spread = t * np.outer(old, old) - (t + 1) * np.outer(new, new) + np.outer(x, x) + self.epsilon * np.eye(self.d)
        self.sigma = (t - 1) / t * self.sigma + self.sd / t * spread
        self.mu = new
        self.t = t + 1


class ZeroMeanRecursiveSampleMoments(RecursiveSampleMoments):
    def __init__(self, sigma0, t=1):
        self.sigma = sigma0
        self.d = sigma0.shape[0]
        self.t = t

    def __call__(self):
        return self.sigma

    def get_mu(self):
        return None

    def update(self, x):
        self.sigma = (self.t - 1) / self.t * self.sigma + 1 / self.t * np.outer(x, x)
        self.t += 1
