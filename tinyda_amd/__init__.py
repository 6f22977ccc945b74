"""tinyda_amd: MI355X-native many-chain MH engine behind tinyDA's sampling API.

Drop-in for the hot path tda.sample() -> Chain.sample -> Proposal / Posterior / GaussianLogLike
(tinyDA/sampler.py, chain.py, proposal.py, posterior.py, distributions.py); see DESIGN.md for the scope.
"""
__version__ = "0.5.0"

from ._lib import EngineError  # noqa: F401
from .hostloop import Chain  # noqa: F401
from .summaries import ess_bulk, ess_summary, ess_tail, get_samples, hdi, mcse_mean, rhat, to_inference_data  # noqa: F401
from .likelihoods import (  # noqa: F401
    JointPrior,
    AdaptiveGaussianLogLike,
    DefaultGaussianLogLike,
    DiagonalGaussianLogLike,
    GaussianLogLike,
    IsotropicGaussianLogLike,
)
from .engine import Engine  # noqa: F401
from .records import Link  # noqa: F401
from .models import BatchedModel, DeviceModel, LinearModel, Rosenbrock  # noqa: F401
from .target import Posterior  # noqa: F401
from .proposals import (  # noqa: F401
    DREAM, DREAMZ, MALA, AdaptiveMetropolis, CrankNicolson, GaussianRandomWalk, IndependenceSampler,
    OperatorWeightedCrankNicolson, Proposal)
from .records import DeviceChain  # noqa: F401
from .api import HostFallbackWarning, release_cached_memory, sample  # noqa: F401
from .moments import RecursiveSampleMoments, ZeroMeanRecursiveSampleMoments  # noqa: F401
