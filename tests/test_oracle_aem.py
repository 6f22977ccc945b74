"""The oracle's adaptive-error-model restatement against tinyDA's DAChain / MLDAChain traces with AEM on."""
import numpy as np
import pytest

from oracle import tinyda_oracle as orc
from tests.test_oracle_multilevel import _prop


def aem_levels(g, nl):
    prior = orc.MVNPrior(g["prior_mean"], g["prior_cov"])
    m = g["A0"].shape[0]
    lv = []
    for k in range(nl):
        spec = dict(A=g["A%d" % k], b=g["b%d" % k], y=g["y%d" % k], prior=prior)
        if k < nl - 1:
            spec["cov"] = float(g["noise_var"]) * np.eye(m)
        else:
            spec["var"] = float(g["noise_var"])
        lv.append(spec)
    return lv


def _compare(res, g, nl, accepted_only=False):
    """The reference stores Link objects; a coarse link's likelihood in the returned chain is the value it had when it
    was appended (later update_link calls create new Link objects), so traces are compared as recorded."""
    for k in range(nl):
        assert np.array_equal(res[k]["accepted"], g["acc%d" % k]), "level %d accept masks differ" % k
        np.testing.assert_allclose(res[k]["theta"], g["th%d" % k], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(res[k]["logprior"], g["lp%d" % k], rtol=1e-10)
        if k == nl - 1:
            np.testing.assert_allclose(res[k]["loglike"], g["ll%d" % k], rtol=1e-10)


@pytest.mark.parametrize("name", ["g8_da_aem_indep", "g8_da_aem_dep", "g8_da_aem_dep_pcn", "g8_da_aem_indep_m72", "g8_da_aem_dep_pcn_m128",
                                  "g8_da_aem_indep_m200", "g8_da_aem_dep_pcn_m256",
                                  "g8_da_aem_indep_d80", "g8_da_aem_dep_pcn_d96"])
def test_da_with_error_model(golden, name):
    g = golden(name)
    L = int(g["subchain_length"])
    n_fine = g["th1"].shape[1] - 1
    res, st = orc.run_multilevel_aem(aem_levels(g, 2), _prop(g), [L], g["theta0"], g["z"], [g["u0"], g["u1"]], n_fine, str(g["aem"]))
    _compare(res, g, 2)
    np.testing.assert_allclose(st["bias"][0], g["bias_mu"], rtol=1e-9, atol=1e-12)
    key = "b_sigma"
    np.testing.assert_allclose(st[key][1], g["bias_sigma"], rtol=1e-8, atol=1e-12)


@pytest.mark.parametrize("name", ["g8_mlda_aem", "g8_mlda_aem_m100", "g8_mlda_aem_m160", "g8_mlda_aem_d72"])
def test_mlda_with_error_model(golden, name):
    g = golden(name)
    nl = int(g["n_levels"])
    n_fine = g["th%d" % (nl - 1)].shape[1] - 1
    res, _ = orc.run_multilevel_aem(aem_levels(g, nl), _prop(g), list(g["subchain_lengths"]), g["theta0"], g["z"],
                                    [g["u%d" % k] for k in range(nl)], n_fine, "state-independent")
    _compare(res, g, nl)
