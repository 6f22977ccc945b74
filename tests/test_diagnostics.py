"""ESS / R-hat.  ArviZ is not importable here and the reference pins no values ('parity unpinned'), so the
estimators are pinned two ways: (1) the product's NumPy implementation (tinyda_amd.summaries; the HIP one in
tests/test_gpu_diag.py) against oracle/ess_oracle.py, an independent restatement of Vehtari et al. (2021) that shares no code
path with it (explicit lagged sums instead of FFTs, its own tie-averaged ranks), on the paper's stress cases -- heavy tails,
antithetic chains, chains that disagree, a chain stuck on ties; (2) both against processes whose effective sample size is
known in closed form."""
import numpy as np
import pytest

import tinyda_amd as tda
from oracle import ess_oracle as eo


def _ar1(rho, chains, n, seed):
    rng = np.random.default_rng(seed)
    x = np.empty((chains, n))
    x[:, 0] = rng.standard_normal(chains)
    e = rng.standard_normal((chains, n)) * np.sqrt(1 - rho ** 2)
    for t in range(1, n):
        x[:, t] = rho * x[:, t - 1] + e[:, t]
    return x


def cases():
    rng = np.random.default_rng(11)
    out = {}
    out["iid"] = _ar1(0.0, 4, 600, 1)
    out["ar1_0.9"] = _ar1(0.9, 6, 900, 2)
    out["antithetic_-0.7"] = _ar1(-0.7, 4, 800, 3)  # negative autocorrelation: ESS above the number of draws (paper sec. 3.2)
    out["cauchy"] = np.tan(np.pi * (rng.random((4, 700)) - 0.5))  # infinite variance: only rank-normalisation makes sense of it (sec. 4.1)
    heavy = _ar1(0.6, 4, 700, 4)
    out["heavy_ar1"] = np.sign(heavy) * np.abs(heavy) ** 3
    shifted = _ar1(0.3, 4, 500, 5)
    shifted[0] += 2.0
    out["one_chain_off"] = shifted  # chains disagree in location (sec. 4.2)
    scaled = _ar1(0.3, 4, 500, 6)
    scaled[1] *= 4.0
    out["one_chain_wide"] = scaled  # same location, different scale: the folded R-hat catches it
    sticky = _ar1(0.5, 3, 401, 7)
    sticky[:, 100:160] = sticky[:, 99:100]  # a run of rejections: exact ties, odd draw count
    out["ties_odd"] = sticky
    out["short"] = _ar1(0.2, 2, 9, 8)
    return out


@pytest.mark.parametrize("name", list(cases()))
def test_product_numpy_matches_the_oracle(name):
    x = cases()[name]
    np.testing.assert_allclose(tda.ess_bulk(x), eo.ess_bulk(x), rtol=1e-9)
    np.testing.assert_allclose(tda.rhat(x), eo.rhat(x), rtol=1e-10)


@pytest.mark.parametrize("name", list(cases()))
def test_tail_ess_mcse_hdi_match_the_oracle(name):
    """the other az.summary columns: tail ESS (indicator chains at the 5 % / 95 % quantiles), Monte Carlo standard error of the
    mean, 94 % highest density interval"""
    x = cases()[name]
    if x.shape[1] < 8:
        return
    np.testing.assert_allclose(tda.ess_tail(x), eo.ess_tail(x), rtol=1e-9)
    np.testing.assert_allclose(tda.mcse_mean(x), eo.mcse_mean(x), rtol=1e-9)
    np.testing.assert_allclose(tda.hdi(x), eo.hdi(x), rtol=0, atol=0)


def test_tail_ess_and_mcse_known_answers():
    """iid draws: tail ESS ~ S and mcse ~ sd / sqrt(S); AR(1): mcse^2 S / var ~ (1 + rho) / (1 - rho); HDI of a standard normal"""
    x = _ar1(0.0, 8, 2000, 21)
    assert abs(tda.ess_tail(x) / x.size - 1) < 0.15
    assert abs(tda.mcse_mean(x) * np.sqrt(x.size) / x.std(ddof=1) - 1) < 0.1
    lo, hi = tda.hdi(x)
    assert abs(lo + 1.88) < 0.12 and abs(hi - 1.88) < 0.12  # 94 % of N(0, 1)
    y = _ar1(0.8, 16, 4000, 22)
    infl = (tda.mcse_mean(y) ** 2) * y.size / y.var(ddof=1)
    assert abs(infl / 9.0 - 1) < 0.25
    assert tda.ess_tail(y) < 0.5 * y.size


def test_known_answers():
    """AR(1): ESS = S (1 - rho) / (1 + rho); iid heavy tails: ESS ~ S after rank normalisation; disagreement shows in R-hat"""
    for est in (tda.ess_bulk, eo.ess_bulk):
        x = _ar1(0.0, 8, 2000, 1)
        assert abs(est(x) / x.size - 1) < 0.1
        for rho in (0.5, 0.9):
            x = _ar1(rho, 16, 4000 if est is tda.ess_bulk else 1500, 2)
            expect = x.size * (1 - rho) / (1 + rho)
            assert abs(est(x) / expect - 1) < 0.2, (rho, est(x), expect)
        x = _ar1(-0.5, 8, 1500, 9)
        assert est(x) > 1.8 * x.size  # antithetic: (1 + 0.5) / (1 - 0.5) = 3 in the limit, capped by S log10 S
        c = cases()["cauchy"]
        assert abs(est(c) / c.size - 1) < 0.15
    for r in (tda.rhat, eo.rhat):
        assert abs(r(_ar1(0.3, 8, 2000, 3)) - 1) < 0.01
        assert r(cases()["one_chain_off"]) > 1.2
        assert r(cases()["one_chain_wide"]) > 1.05


def test_ess_summary_layout():
    x = np.stack([_ar1(0.5, 6, 800, s).T for s in range(3)], axis=2)  # [draws, chains, dim]
    s = tda.ess_summary(x, burnin=100)
    assert s["ess"].shape == (3,) and s["ess_min"] <= s["ess_median"]
    np.testing.assert_allclose(s["ess"][1], eo.ess_bulk(x[100:, :, 1].T), rtol=1e-9)
