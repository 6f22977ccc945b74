"""The oracle's DREAM(Z) restatement against tinyDA's DREAMZ traces (all variates recorded)."""
import numpy as np
import pytest

from oracle import tinyda_oracle as orc


def dreamz_inputs(g):
    prior = orc.MVNPrior(g["prior_mean"], g["prior_cov"])
    if str(g["problem"]) == "linear":
        noise = orc.noise_of(g["noise_cov"]) if "noise_cov" in g.files else ("iso", float(g["noise_var"]))
        level = orc.LinearGaussianLevel(g["A"], g["data"], *noise, prior)
    else:
        level = orc.RosenbrockLevel(prior, float(g["rosen_a"]), float(g["rosen_b"]))
    cfg = {k: g[k].item() for k in ("M0", "delta", "nCR", "adaptive", "period", "gamma", "b", "b_star")}
    var = {k: g[k] for k in ("r", "mcr", "sub_u", "forced", "e_u", "eps_n", "u")}
    return level, cfg, var


@pytest.mark.parametrize("name", ["g6_dreamz_linear", "g6_dreamz_rosen_adaptive", "g6_dreamz_empty_subspace", "g6_dreamz_linear_dense"])
def test_dreamz(golden, name):
    g = golden(name)
    level, cfg, var = dreamz_inputs(g)
    res = orc.run_dreamz(level, cfg, g["theta0"], g["Z0"], var)
    assert np.array_equal(res["accepted"], g["accepted"])
    np.testing.assert_allclose(res["logpost"], g["logprior"] + g["loglike"], rtol=1e-10)
    np.testing.assert_allclose(res["theta"], g["theta"], rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(res["scaling"], g["scaling"], rtol=1e-12)
    np.testing.assert_allclose(res["pCR"], g["pCR"], rtol=1e-9)
    assert (g["forced"] >= 0).any()
