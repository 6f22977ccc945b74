"""On-device bulk ESS / R-hat (tda_diag_ess_rhat) against oracle/ess_oracle.py, the independent restatement of Vehtari et al.
(2021) that also checks the package's NumPy implementation (tests/test_diagnostics.py)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _chains(T, N, d, seed, rho, sticky):
    """AR(1) chains with repeated values (rejections produce exact ties), different means per chain for d > 1"""
    rng = np.random.default_rng(seed)
    x = np.zeros((T, N, d))
    cur = rng.standard_normal((N, d))
    for t in range(T):
        move = rng.random((N, 1)) > sticky
        prop = rho * cur + np.sqrt(1 - rho ** 2) * rng.standard_normal((N, d))
        cur = np.where(move, prop, cur)
        x[t] = cur
    x[:, :, -1] += 3.0 * np.arange(N)[None, :] / N  # a parameter whose chains disagree: R-hat > 1
    return x


@pytest.mark.parametrize("T,N,d,burnin", [(400, 16, 3, 0), (257, 5, 2, 31), (1200, 64, 4, 200)])
def test_device_ess_rhat_matches_the_oracle(T, N, d, burnin):
    import torch

    from oracle import ess_oracle as eo
    from tinyda_amd import summaries as sm

    x = _chains(T, N, d, seed=T + N, rho=0.9, sticky=0.6)
    dev = torch.tensor(x, dtype=torch.float64, device="cuda")
    out = sm.ess_rhat_device(dev, burnin=burnin)
    for j in range(d):
        xs = x[burnin:, :, j].T
        np.testing.assert_allclose(out["ess"][j], eo.ess_bulk(xs), rtol=1e-8)
        np.testing.assert_allclose(out["rhat"][j], eo.rhat(xs), rtol=1e-10)
    assert out["rhat"][-1] > out["rhat"][0]


def test_device_ess_rhat_on_the_stress_cases():
    """heavy tails, antithetic chains, disagreeing chains, runs of ties with an odd number of draws (tests/test_diagnostics.py)"""
    import torch

    from oracle import ess_oracle as eo
    from tests.test_diagnostics import cases
    from tinyda_amd import summaries as sm

    for name, x in cases().items():
        if x.shape[1] < 16:
            continue
        dev = torch.tensor(np.ascontiguousarray(x.T[:, :, None]), dtype=torch.float64, device="cuda")  # [draws, chains, 1]
        out = sm.ess_rhat_device(dev)
        np.testing.assert_allclose(out["ess"][0], eo.ess_bulk(x), rtol=1e-8, err_msg=name)
        np.testing.assert_allclose(out["rhat"][0], eo.rhat(x), rtol=1e-10, err_msg=name)


def test_device_diag_rejects_host_pointers():
    from tinyda_amd import _lib, summaries as sm

    class Fake:
        shape = (100, 4, 2)

        def __init__(self):
            self.a = np.zeros(self.shape)

        def data_ptr(self):
            return self.a.ctypes.data

    with pytest.raises(_lib.EngineError):
        sm.ess_rhat_device(Fake())
