"""tinyda_amd: MI355X-native many-chain MH / DA / MLDA engine behind tinyDA's sampling API.

Drop-in for the hot path tda.sample() -> Chain.sample (tinyDA/sampler.py, chain.py); see DESIGN.md.
"""
__version__ = "0.1.0"

from ._lib import EngineError  # noqa: F401
from .engine import Engine  # noqa: F401
