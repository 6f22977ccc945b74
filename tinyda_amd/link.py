"""Sample record of the sampler.  Mirrors tinyDA/link.py:23-48 (same attribute names)."""


class Link:
    """One MCMC sample: parameters, log-prior, model output, log-likelihood, optional QoI.

    `posterior` is prior + likelihood with a *normalised* prior and an *unnormalised* Gaussian
    likelihood, exactly as the reference combines them (link.py:48).
    """

    __slots__ = ("parameters", "prior", "model_output", "likelihood", "qoi", "posterior")

    def __init__(self, parameters, prior, model_output, likelihood, qoi=None):
        self.parameters = parameters
        self.prior = prior
        self.model_output = model_output
        self.likelihood = likelihood
        self.qoi = qoi
        self.posterior = prior + likelihood

    def __repr__(self):
        return "Link(posterior=%r, parameters=%r)" % (self.posterior, self.parameters)
