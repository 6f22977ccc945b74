"""Posterior = prior x likelihood x forward model (tinyDA/posterior.py:41-151)."""
import numpy as np

from .records import Link
from .models import BatchedModel, DeviceModel, LinearModel, Rosenbrock


class Posterior:
    def __init__(self, prior, likelihood, model=None):
        self.prior = prior
        self.likelihood = likelihood
        # subclasses may provide evaluate_model instead of passing a callable (posterior.py:57-61)
        self.model = self.evaluate_model if model is None else model

    def evaluate_model(self, parameters):  # pragma: no cover - to be overridden
        raise NotImplementedError("pass model= or override evaluate_model")

    def create_link(self, parameters):
        """prior.logpdf -> model -> loglike -> Link (posterior.py:78-110)."""
        log_prior = self.prior.logpdf(parameters)
        result = self.model(parameters)
        output, qoi = result if isinstance(result, tuple) else (result, None)
        if not isinstance(output, np.ndarray):
            raise TypeError("Model output must be a numpy array!")
        return Link(parameters, log_prior, output, self.likelihood.loglike(output), qoi)

    def update_link(self, link, bias=None):
        """Re-evaluate only the likelihood of an existing link (posterior.py:112-134)."""
        if bias is None:
            log_like = self.likelihood.loglike(link.model_output)
        else:
            log_like = self.likelihood.loglike_custom_bias(link.model_output, bias)
        return Link(link.parameters, link.prior, link.model_output, log_like, link.qoi)

    def logpdf(self, parameters):
        return self.create_link(parameters).posterior

    __call__ = logpdf

    # ---- device lowering ---------------------------------------------------------------
    def _lowering(self):
        """What the HIP engine needs, or None when this posterior only runs through the host protocol
        (opaque Python model, non-Gaussian prior, ...)."""
        prior = self.prior
        joint = None
        if hasattr(prior, "distributions") and hasattr(prior, "_lowering"):  # JointPrior of scalar norm / uniform components
            joint = prior._lowering()
            if joint is None:
                return None
            kinds, loc, scale = joint
            low = self._lowering_with(np.where(kinds == 0, loc, loc + 0.5 * scale),
                                      np.diag(np.where(kinds == 0, scale ** 2, scale ** 2 / 12.0)))
            if low is not None:
                low["prior_joint"] = joint
            return low
        mean = getattr(prior, "mean", None)
        cov = getattr(prior, "cov", None)
        if cov is None and hasattr(prior, "cov_object"):
            cov = prior.cov_object.covariance
        if mean is None or cov is None or not hasattr(prior, "logpdf"):
            return None
        return self._lowering_with(mean, cov)

    def _lowering_with(self, mean, cov):
        if not isinstance(self.model, (LinearModel, Rosenbrock, DeviceModel, BatchedModel)) or not hasattr(self.likelihood, "_lowering"):
            return None
        mean = np.atleast_1d(np.asarray(mean, dtype=np.float64))
        cov = np.atleast_2d(np.asarray(cov, dtype=np.float64))
        if isinstance(self.model, Rosenbrock):
            kind, noise = self.likelihood._lowering()
            data = np.atleast_1d(np.asarray(self.likelihood.data, dtype=np.float64))
            if data.shape != (1,) or np.size(noise) != 1 or mean.shape[0] < 2:
                return None
            return dict(prior_mean=mean, prior_cov=cov, rosenbrock=(self.model.a, self.model.b), A=None, b=None, data=data,
                        noise_kind=kind, noise=np.asarray(noise, dtype=np.float64).reshape(1))
        if isinstance(self.model, DeviceModel):
            kind, noise = self.likelihood._lowering()
            data = np.atleast_1d(np.asarray(self.likelihood.data, dtype=np.float64))
            if data.shape != (self.model.n_outputs,):
                return None
            return dict(prior_mean=mean, prior_cov=cov, source=self.model.source, A=None, b=None, data=data,
                        noise_kind=kind, noise=np.asarray(noise, dtype=np.float64))
        if isinstance(self.model, BatchedModel):
            kind, noise = self.likelihood._lowering()
            data = np.atleast_1d(np.asarray(self.likelihood.data, dtype=np.float64))
            if data.shape != (self.model.n_outputs,):
                return None
            return dict(prior_mean=mean, prior_cov=cov, batched=self.model.batch, A=None, b=None, data=data,
                        noise_kind=kind, noise=np.asarray(noise, dtype=np.float64))
        if self.model.A.shape[1] != mean.shape[0]:
            return None
        kind, noise = self.likelihood._lowering()
        return dict(prior_mean=mean, prior_cov=cov, A=self.model.A, b=self.model.b,
                    data=np.asarray(self.likelihood.data, dtype=np.float64), noise_kind=kind, noise=noise)
