"""ESS / R-hat: ArviZ is not importable here and the reference pins no values ('parity
unpinned'), so the estimators are pinned on processes with known answers."""
import numpy as np

import tinyda_amd as tda


def _ar1(rho, chains, n, seed):
    rng = np.random.default_rng(seed)
    x = np.empty((chains, n))
    x[:, 0] = rng.standard_normal(chains)
    e = rng.standard_normal((chains, n)) * np.sqrt(1 - rho ** 2)
    for t in range(1, n):
        x[:, t] = rho * x[:, t - 1] + e[:, t]
    return x


def test_ess_iid_and_ar1():
    x = _ar1(0.0, 8, 2000, 1)
    assert abs(tda.ess_bulk(x) / x.size - 1) < 0.1
    for rho in (0.5, 0.9):
        x = _ar1(rho, 16, 4000, 2)
        expect = x.size * (1 - rho) / (1 + rho)
        assert abs(tda.ess_bulk(x) / expect - 1) < 0.15, (rho, tda.ess_bulk(x), expect)
    assert abs(tda.rhat(_ar1(0.3, 8, 2000, 3)) - 1) < 0.01
    shifted = _ar1(0.3, 4, 1000, 4)
    shifted[0] += 3.0
    assert tda.rhat(shifted) > 1.2


def test_ess_summary_layout():
    x = np.stack([_ar1(0.5, 6, 800, s).T for s in range(3)], axis=2)  # [draws, chains, dim]
    s = tda.ess_summary(x, burnin=100)
    assert s["ess"].shape == (3,) and s["ess_min"] <= s["ess_median"]
