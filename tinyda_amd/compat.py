"""The reference's pre- / post-processing helpers and deprecated names, so that scripts written against tinyDA import unchanged.
None of this is on the hot path (SURVEY.md §2 rows 4, 6b, 7b, 10 mark it out of scope for the kernels); it is host Python over
the same Posterior / JointPrior / DREAMZ objects the engine lowers.

  get_MAP, get_ML            tinyDA/utils.py:204-269      scipy.optimize over Posterior.create_link
  grad_log_p, grad_log_l     tinyDA/utils.py:272-287      gradients MALA uses
  to_xarray                  tinyDA/diagnostics.py:72-111 get_samples() dict -> Dataset (xarray when installed)
  LinkFactory, BlackBoxLinkFactory  tinyDA/posterior.py:154-179   deprecated spellings of Posterior
  CompositePrior             tinyDA/distributions.py:103-106     deprecated spelling of JointPrior
  SingleDreamZ               tinyDA/proposal.py:855-858          deprecated spelling of DREAMZ
  DAChain, MLDAChain         tinyDA/chain.py:132-530, :534-769   the reference's constructors and attribute names over
                             hostloop.HierarchyChain (ONE chain on the host; `sample()` is what runs many on the GPU)
"""
import warnings

import numpy as np

from .hostloop import HierarchyChain
from .likelihoods import DefaultGaussianLogLike, JointPrior
from .proposals import DREAMZ, _grad_log_prior
from .summaries import DatasetLite
from .target import Posterior


def _optimise(objective, posterior, kwargs):
    from scipy.optimize import differential_evolution, minimize

    method = kwargs.pop("method", None)
    if method == "differential_evolution":
        return differential_evolution(objective, **kwargs)["x"]
    start = kwargs.pop("initial_parameters", None)
    if start is None:
        start = np.ravel(posterior.prior.rvs())  # the reference's default: one draw from the prior
    return minimize(objective, start, method=method, **kwargs)["x"]


def get_MAP(posterior, **kwargs):
    """Maximum a posteriori point: scipy.optimize.minimize (or differential_evolution with method='differential_evolution')
    on -create_link(theta).posterior; `initial_parameters` and every other keyword as in utils.py:204-235."""
    return _optimise(lambda parameters: -posterior.create_link(parameters).posterior, posterior, kwargs)


def get_ML(posterior, **kwargs):
    """Maximum likelihood point, the same over -create_link(theta).likelihood (utils.py:238-269)."""
    return _optimise(lambda parameters: -posterior.create_link(parameters).likelihood, posterior, kwargs)


def grad_log_p(x, dist):
    """Gradient of a prior's log-density: Sigma^-1 (mean - x) for a Gaussian, finite differences of logpdf otherwise
    (utils.py:272-280; the reference's fallback passes the bound method instead of calling it and raises -- here it works)."""
    return _grad_log_prior(np.asarray(x, dtype=np.float64), dist)


def grad_log_l(x, dist):
    """Gradient of a log-likelihood with respect to the model output (utils.py:283-287)."""
    if isinstance(dist, DefaultGaussianLogLike):
        return dist.grad_loglike(x)
    from scipy.optimize import approx_fprime

    return approx_fprime(np.asarray(x, dtype=np.float64), dist.loglike)


def to_xarray(samples, keys):
    """get_samples() dict -> one Dataset with dimensions (chain, draw) and the variables named by `keys`
    (diagnostics.py:72-111): xarray.Dataset when xarray is installed, the DatasetLite stand-in otherwise."""
    n_chains, n_draws = samples["n_chains"], samples["iterations"]
    data = {keys[i]: np.array([np.asarray(samples["chain_{}".format(j)])[:, i] for j in range(n_chains)])
            for i in range(samples["dimension"])}
    try:
        import xarray as xr
    except ImportError:
        return DatasetLite(data, n_chains, n_draws)
    return xr.Dataset({k: (["chain", "draw"], v) for k, v in data.items()},
                      coords=dict(chain=("chain", list(range(n_chains))), draw=("draw", list(range(n_draws)))))


class LinkFactory(Posterior):
    """Deprecated spelling of Posterior (posterior.py:154-163): subclass it and provide evaluate_model."""

    def __init__(self, prior, likelihood):
        warnings.warn("LinkFactory is deprecated and will be removed in the next version. Please use Posterior instead", stacklevel=2)
        super().__init__(prior, likelihood)


class BlackBoxLinkFactory(Posterior):
    """Deprecated spelling of Posterior with the model first (posterior.py:165-179); get_qoi has no effect."""

    def __init__(self, model, prior, likelihood, get_qoi=False):
        warnings.warn("BlackBoxLinkFactory is deprecated and will be removed in the next version. Please use Posterior instead.", stacklevel=2)
        if get_qoi:
            warnings.warn("The argument get_qoi has been removed and has no effect. If a quantity of interest is required, "
                          "the call to the forward model must return a tuple of (model_output, qoi)", stacklevel=2)
        super().__init__(prior, likelihood, model)


def CompositePrior(*args, **kwargs):
    warnings.warn("CompositePrior has been deprecated. Please use JointPrior.")
    return JointPrior(*args, **kwargs)


def SingleDreamZ(*args, **kwargs):
    warnings.warn("SingleDreamZ has been deprecated. Please use DREAMZ.")
    return DREAMZ(*args, **kwargs)


class DAChain(HierarchyChain):
    """tinyDA.DAChain's constructor (chain.py:185-195) and result attributes over the host hierarchy driver."""

    def __init__(self, posterior_coarse, posterior_fine, proposal, subchain_length, randomize_subchain_length=False,
                 initial_parameters=None, adaptive_error_model=None, store_coarse_chain=True):
        self.posterior_coarse, self.posterior_fine = posterior_coarse, posterior_fine
        self.subchain_length = subchain_length
        self.randomize_subchain_length = randomize_subchain_length
        self.adaptive_error_model = adaptive_error_model
        self.store_coarse_chain = store_coarse_chain
        super().__init__([posterior_coarse, posterior_fine], proposal, [subchain_length], initial_parameters, adaptive_error_model,
                         store_coarse_chain, randomize_subchain_length)

    chain_fine = property(lambda self: self.rungs[1].links)
    accepted_fine = property(lambda self: self.rungs[1].took)
    chain_coarse = property(lambda self: self.rungs[0].links)
    accepted_coarse = property(lambda self: self.rungs[0].took)
    is_coarse = property(lambda self: self.rungs[0].own)
    promoted_coarse = property(lambda self: self.promoted)
    subchain_lengths = property(lambda self: self.effective_lengths)
    bias = property(lambda self: self.trackers[1])


class MLDAChain(HierarchyChain):
    """tinyDA.MLDAChain's constructor (chain.py:570-578): `chain` / `accepted` are the finest level's, `levels[k]` the rungs
    below it (the reference reaches those through nested `proposal.proposal...` objects; here they are one flat list)."""

    def __init__(self, posteriors, proposal, subchain_lengths, initial_parameters=None, adaptive_error_model=None,
                 store_coarse_chain=True):
        self.posteriors = list(posteriors)
        self.posterior = self.posteriors[-1]
        self.subchain_lengths = list(subchain_lengths)
        self.adaptive_error_model = adaptive_error_model
        self.store_coarse_chain = store_coarse_chain
        super().__init__(self.posteriors, proposal, self.subchain_lengths, initial_parameters, adaptive_error_model, store_coarse_chain)

    chain = property(lambda self: self.rungs[-1].links)
    accepted = property(lambda self: self.rungs[-1].took)
    levels = property(lambda self: self.rungs)
    biases = property(lambda self: [t for t in reversed(self.trackers[1:])])
