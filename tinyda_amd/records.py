"""Sample records: `Link` (one MCMC sample, attribute names of tinyDA/link.py:23-48) and `DeviceChain`, the
array-backed sequence the device path returns instead of a Python list of Links."""
from collections.abc import Sequence

import numpy as np

_FIELDS = ("parameters", "prior", "model_output", "likelihood", "qoi")


class Link:
    """parameters, log-prior, model output, log-likelihood, optional quantity of interest; `posterior` is the sum of a
    *normalised* log-prior and an *unnormalised* Gaussian log-likelihood, exactly as the reference combines them."""

    __slots__ = _FIELDS + ("posterior", "gradient")  # `gradient` is attached lazily by MALA (proposal.py:948-949)

    def __init__(self, parameters, prior, model_output, likelihood, qoi=None):
        for name, value in zip(_FIELDS, (parameters, prior, model_output, likelihood, qoi)):
            object.__setattr__(self, name, value)
        object.__setattr__(self, "posterior", prior + likelihood)

    def __repr__(self):
        return "Link(posterior=%r, parameters=%r)" % (self.posterior, self.parameters)


class DeviceChain(Sequence):
    """tinyDA returns `chain_i` as a list of Link objects (sampler.py:305-309).  With thousands of chains that is
    millions of objects, so the device path returns this read-only view over the engine's record arrays; a Link
    (including its model output) is materialised only when indexed.  `get_samples` reads the arrays directly."""

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
        out, qoi = out if isinstance(out, tuple) else (out, None)  # posterior.py:97-101
        return Link(theta, float(self.stats[i, 0]), out, float(self.stats[i, 1]), qoi)
