"""The oracle's Delayed-Acceptance / MLDA state machine against traces produced by tinyDA's DAChain / MLDAChain."""
import numpy as np
import pytest

from oracle import tinyda_oracle as orc

RTOL = 1e-10


def _prop(g):
    p = {k[5:]: g[k] for k in g.files if k.startswith("prop_")}
    out = {"kind": str(p["kind"])}
    for k, v in p.items():
        if k == "kind":
            continue
        out[k] = v if v.ndim else v.item()
    return out


def _levels(g, n):
    prior = orc.MVNPrior(g["prior_mean"], g["prior_cov"])
    if "cov0" in g.files:  # per-level covariances (dense / diagonal / isotropic): what GaussianLogLike's factory makes of each
        return [orc.LinearGaussianLevel(g["A%d" % k], g["y%d" % k], *orc.noise_of(g["cov%d" % k]), prior) for k in range(n)]
    return [orc.LinearGaussianLevel(g["A%d" % k], g["y%d" % k], "iso", float(g["noise_var"]), prior) for k in range(n)]


def _check_level(res, g, k, finest):
    assert np.array_equal(res["accepted"], g["acc%d" % k]), "level %d accept masks differ" % k
    np.testing.assert_allclose(res["logprior"] + res["loglike"], g["lp%d" % k] + g["ll%d" % k], rtol=RTOL)
    np.testing.assert_allclose(res["loglike"], g["ll%d" % k], rtol=1e-9)
    np.testing.assert_allclose(res["theta"], g["th%d" % k], rtol=1e-9, atol=1e-11)


@pytest.mark.parametrize("name", ["g4_da_pcn", "g4_da_grw_adaptive", "g4_da_am_random", "g4_da_pcn_adaptive_c3shape",
                                  "g4_da_pcn_dense_fine", "g4_da_am_dense_both"])
def test_delayed_acceptance(golden, name):
    g = golden(name)
    L = int(g["subchain_length"])
    n_fine = g["th1"].shape[1] - 1
    ridx = g["ridx"] if bool(g["randomize"]) else None
    res, prop = orc.run_multilevel(_levels(g, 2), _prop(g), [L], g["theta0"], g["z"], [g["u0"], g["u1"]], n_fine, ridx)
    _check_level(res[0], g, 0, False)
    _check_level(res[1], g, 1, True)
    np.testing.assert_allclose(prop.scaling, g["scaling"], rtol=1e-12)


@pytest.mark.parametrize("name", ["g5_mlda_am", "g5_mlda_grw_adaptive", "g5_mlda_4level", "g5_mlda_am_dense", "g5_mlda_5level", "g5_mlda_6level"])
def test_mlda(golden, name):
    g = golden(name)
    nl = int(g["n_levels"])
    n_fine = g["th%d" % (nl - 1)].shape[1] - 1
    res, prop = orc.run_multilevel(_levels(g, nl), _prop(g), list(g["subchain_lengths"]), g["theta0"], g["z"],
                                   [g["u%d" % k] for k in range(nl)], n_fine)
    for k in range(nl):
        _check_level(res[k], g, k, k == nl - 1)
    np.testing.assert_allclose(prop.scaling, g["scaling"], rtol=1e-12)
