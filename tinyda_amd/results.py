"""Array-backed chain containers.

tinyDA returns `chain_i` as a Python list of Link objects (sampler.py:305-309).  With thousands of chains
that is millions of objects, so the device path returns `DeviceChain`: a read-only sequence over the
engine's record arrays that materialises a Link (including its model output A theta + b) only when indexed.
`get_samples` reads the arrays directly.
"""
from collections.abc import Sequence

import numpy as np

from .link import Link


class DeviceChain(Sequence):
    def __init__(self, parameters, stats, accepted, model=None):
        self.parameters = parameters  # [T+1, d]
        self.stats = stats  # [T+1, 3] log-prior, log-likelihood, log-posterior
        self.accepted = accepted  # [T+1] (entry 0 is the initial link, True as in chain.py:71)
        self._model = model

    def __len__(self):
        return self.parameters.shape[0]

    def __getitem__(self, i):
        if isinstance(i, slice):
            return DeviceChain(self.parameters[i], self.stats[i], self.accepted[i], self._model)
        theta = np.array(self.parameters[i])
        out = self._model(theta) if self._model is not None else None
        return Link(theta, float(self.stats[i, 0]), out, float(self.stats[i, 1]))
