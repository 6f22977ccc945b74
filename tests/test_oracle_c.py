"""The C oracle (CPU baseline) against the reference's golden vectors and the NumPy oracle."""
import numpy as np
import pytest

from oracle import tinyda_oracle as orc


@pytest.fixture(scope="module")
def oc():
    import __graft_entry__ as g

    g.build()
    from oracle import oracle_c

    oracle_c.load()
    return oracle_c


@pytest.mark.parametrize("name", ["g2_am_small", "g2_am_small_adaptive", "g2_am_c2"])
def test_c_oracle_matches_reference_golden(oc, golden, name):
    g = golden(name)
    d = g["theta"].shape[2]
    res = oc.run_mh(g["A"], g["data"], float(g["noise_cov"][0]), g["prior_mean"], np.diag(g["prior_cov"]), 2, g["C0"],
                    g["theta0"], np.swapaxes(g["z"], 0, 1), np.swapaxes(g["u"], 0, 1), adaptive=bool(g["adaptive"]),
                    gamma=float(g["gamma"]), period=int(g["period"]), sd=float(g["sd"]), eps=float(g["epsilon"]),
                    t0=int(g["t0"]))
    assert np.array_equal(res["accepted"], np.swapaxes(g["accepted"][:, 1:], 0, 1))
    np.testing.assert_allclose(res["stats"][:, :, 2], np.swapaxes(g["logpost"][:, 1:], 0, 1), rtol=1e-10)
    np.testing.assert_allclose(res["sigma"], g["sigma_hist"][:, -1], rtol=1e-9, atol=1e-12)


def test_c_oracle_matches_numpy_oracle_grw(oc, golden):
    g = golden("g1_basic_sampler")
    res = oc.run_mh(g["A"], g["data"], float(g["noise_var"]), g["prior_mean"], np.diag(g["prior_cov"]), 0, g["C"],
                    g["theta0"], np.swapaxes(g["z"], 0, 1), np.swapaxes(g["u"], 0, 1), scaling0=float(g["scaling0"]),
                    adaptive=True, gamma=float(g["gamma"]), period=int(g["period"]))
    assert np.array_equal(res["accepted"], np.swapaxes(g["accepted"][:, 1:], 0, 1))
    np.testing.assert_allclose(res["stats"][:, :, 2], np.swapaxes(g["logpost"][:, 1:], 0, 1), rtol=1e-10)
