"""tinyda_amd.hostloop.HierarchyChain (the host Delayed-Acceptance / MLDA driver for what the device engine does not lower)
replays traces recorded from tinyDA's DAChain / MLDAChain: the reference's variates are fed in the reference's order, so
accept masks must be equal and log-densities agree to 1e-10 at every level."""
import copy

import numpy as np
import pytest
import scipy.stats as stats

import tinyda_amd as tda
from tinyda_amd.hostloop import HierarchyChain


class _Fed(HierarchyChain):
    """a HierarchyChain whose uniforms / promoted indices come from per-level queues and whose proposal normals come from z"""

    def feed(self, z, u_levels, ridx=None):
        self._z, self._iz = z, 0
        self._u = [list(u[~np.isnan(u)]) for u in u_levels]
        self._r = None if ridx is None else [int(v) for v in ridx[~np.isnan(ridx)]]

    def _uniform(self, level):
        return self._u[level].pop(0)

    def _promoted_index(self, length):
        return self._r.pop(0)


class _Normals:
    def __init__(self, chain):
        self.chain = chain

    def __enter__(self):
        self.saved = np.random.standard_normal
        np.random.standard_normal = self._next
        return self

    def __exit__(self, *a):
        np.random.standard_normal = self.saved

    def _next(self, n):
        self.chain._iz += 1
        return self.chain._z[self.chain._iz - 1]


def _proposal(g):
    kind = str(g["prop_kind"])
    get = lambda k, default=None: (g["prop_" + k].item() if g["prop_" + k].ndim == 0 else g["prop_" + k]) if "prop_" + k in g.files else default
    if kind == "pcn":
        return tda.CrankNicolson(scaling=float(get("scaling")), adaptive=bool(get("adaptive", False)), gamma=float(get("gamma", 1.01)),
                                 period=int(get("period", 100)))
    if kind == "grw":
        return tda.GaussianRandomWalk(get("C"), scaling=float(get("scaling", 1.0)), adaptive=bool(get("adaptive", False)),
                                      gamma=float(get("gamma", 1.01)), period=int(get("period", 100)))
    return tda.AdaptiveMetropolis(get("C0"), sd=float(get("sd")), epsilon=float(get("epsilon", 1e-6)), t0=int(get("t0", 0)),
                                  period=int(get("period", 100)), adaptive=bool(get("adaptive", False)), gamma=float(get("gamma", 1.01)))


def _model(A, b=None):
    return (lambda th: A @ th) if b is None else (lambda th: A @ th + b)


def _trace(links):
    return (np.array([ln.parameters for ln in links]), np.array([ln.prior for ln in links]), np.array([ln.likelihood for ln in links]))


def _run(g, nl, lengths, posteriors_for_chain, aem=None, randomize=False):
    out = []
    for c in range(g["theta0"].shape[0]):
        ch = _Fed.__new__(_Fed)
        ch.feed(g["z"][c], [g["u%d" % k][c] for k in range(nl)], g["ridx"][c] if randomize else None)
        with _Normals(ch):
            HierarchyChain.__init__(ch, posteriors_for_chain(), _proposal(g), lengths, initial_parameters=g["theta0"][c].copy(),
                                    adaptive_error_model=aem, randomize_subchain_length=randomize)
            ch.sample(g["th%d" % (nl - 1)].shape[1] - 1)
        assert all(len(q) == 0 for q in ch._u), "uniforms left over: the driver drew fewer than the reference"
        assert ch._iz == g["z"].shape[1]
        out.append(ch)
    return out


def _check(ch, g, c, nl, with_like=True):
    for k in range(nl):
        links = ch.level_chain(k)
        th, lp, ll = _trace(links)
        took = np.array([t for t, own in zip(ch.rungs[k].took, ch.rungs[k].own) if own] if k < nl - 1 else ch.rungs[k].took, dtype=np.uint8)
        assert np.array_equal(took, g["acc%d" % k][c]), "level %d accept flags differ" % k
        np.testing.assert_allclose(th, g["th%d" % k][c], rtol=1e-10, atol=1e-12)
        np.testing.assert_allclose(lp, g["lp%d" % k][c], rtol=1e-10)
        if with_like or k == nl - 1:
            np.testing.assert_allclose(ll, g["ll%d" % k][c], rtol=1e-10)


@pytest.mark.parametrize("name", ["g4_da_pcn", "g4_da_grw_adaptive", "g4_da_am_random", "g4_da_pcn_adaptive_c3shape"])
def test_delayed_acceptance_replay(golden, name):
    g = golden(name)
    prior = stats.multivariate_normal(g["prior_mean"], g["prior_cov"])
    var = float(g["noise_var"])

    def posts():
        return [tda.Posterior(prior, tda.GaussianLogLike(g["y%d" % k], var * np.eye(len(g["y%d" % k]))), _model(g["A%d" % k])) for k in range(2)]

    chains = _run(g, 2, [int(g["subchain_length"])], posts, randomize=bool(g["randomize"]))
    for c, ch in enumerate(chains):
        _check(ch, g, c, 2)
        np.testing.assert_allclose(ch.proposal.scaling, g["scaling"][c], rtol=1e-12)


@pytest.mark.parametrize("name", ["g5_mlda_am", "g5_mlda_grw_adaptive", "g5_mlda_4level"])
def test_mlda_replay(golden, name):
    g = golden(name)
    nl = int(g["n_levels"])
    prior = stats.multivariate_normal(g["prior_mean"], g["prior_cov"])
    var = float(g["noise_var"])

    def posts():
        return [tda.Posterior(prior, tda.GaussianLogLike(g["y%d" % k], var * np.eye(len(g["y%d" % k]))), _model(g["A%d" % k])) for k in range(nl)]

    chains = _run(g, nl, [int(v) for v in g["subchain_lengths"]], posts)
    for c, ch in enumerate(chains):
        _check(ch, g, c, nl)
        np.testing.assert_allclose(ch.proposal.scaling, g["scaling"][c], rtol=1e-12)


def _aem_posts(g, nl):
    prior = stats.multivariate_normal(g["prior_mean"], g["prior_cov"])
    var = float(g["noise_var"])
    m = g["A0"].shape[0]

    def posts():
        out = []
        for k in range(nl):
            like = tda.AdaptiveGaussianLogLike(g["y%d" % k], var * np.eye(m)) if k < nl - 1 else tda.GaussianLogLike(g["y%d" % k], var * np.eye(m))
            out.append(tda.Posterior(prior, like, _model(g["A%d" % k], g["b%d" % k])))
        return out

    return posts


@pytest.mark.parametrize("name", ["g8_da_aem_indep", "g8_da_aem_dep", "g8_da_aem_dep_pcn"])
def test_delayed_acceptance_with_error_model_replay(golden, name):
    g = golden(name)
    chains = _run(g, 2, [int(g["subchain_length"])], _aem_posts(g, 2), aem=str(g["aem"]))
    for c, ch in enumerate(chains):
        _check(ch, g, c, 2, with_like=False)  # coarse likelihoods are recorded before later update_link calls replace them


def test_mlda_with_error_model_replay(golden):
    g = golden("g8_mlda_aem")
    nl = int(g["n_levels"])
    chains = _run(g, nl, [int(v) for v in g["subchain_lengths"]], _aem_posts(g, nl), aem="state-independent")
    for c, ch in enumerate(chains):
        _check(ch, g, c, nl, with_like=False)


class _DreamzTap:
    """feeds the host DREAMZ class the draws recorded from the reference (np.random.choice / uniform / normal)"""

    def __init__(self, g, c, delta):
        self.r = [list(map(int, pair)) for step in g["r"][c] for pair in step]
        self.mcr = [int(v) for v in g["mcr"][c]]
        self.forced = [int(v) for v in g["forced"][c] if v >= 0]
        self.sub_u, self.e_u, self.eps = list(g["sub_u"][c]), list(g["e_u"][c]), list(g["eps_n"][c])
        self._phase = 0

    def __enter__(self):
        self.saved = (np.random.choice, np.random.uniform, np.random.normal)
        np.random.choice, np.random.uniform, np.random.normal = self.choice, self.uniform, self.normal
        return self

    def __exit__(self, *a):
        np.random.choice, np.random.uniform, np.random.normal = self.saved

    def choice(self, a, size=None, replace=True, p=None):
        if size == 2:
            return np.array(self.r.pop(0))
        if p is not None:
            return self.mcr.pop(0)
        return self.forced.pop(0)

    def uniform(self, low=0.0, high=1.0, size=None):
        if low == 0.0 and high == 1.0:
            return self.sub_u.pop(0)
        return low + (high - low) * self.e_u.pop(0)

    def normal(self, loc=0.0, scale=1.0, size=None):
        return loc + scale * self.eps.pop(0)


@pytest.mark.parametrize("name", ["g15_da_dreamz", "g15_mlda_dreamz", "g15_da_dreamz_random", "g15_da_dreamz_aem", "g15_da_dreamz_aem_dep",
                                  "g15_mlda_dreamz_aem", "g15_da_dreamz_aem_m160"])
def test_dreamz_below_a_hierarchy_replay(golden, name):
    """DREAMZ as the base proposal of Delayed Acceptance / MLDA (the reference's MLDA notebook configuration) on the host driver;
    round 3: also with randomised subchains and with the state-independent / state-dependent error model"""
    g = golden(name)
    nl = int(g["n_levels"])
    prior = stats.multivariate_normal(g["prior_mean"], g["prior_cov"])
    var = float(g["noise_var"])
    aem = str(g["aem"]) if "aem" in g.files else None
    randomize = bool(g["randomize"])
    for c in range(g["theta0"].shape[0]):
        Z0 = g["Z0"][c]

        class Seeded(tda.DREAMZ):
            def setup_proposal(self, **kw):
                super().setup_proposal(**kw)
                self.Z = Z0.copy()

        posts = _aem_posts(g, nl)() if aem else [tda.Posterior(prior, tda.GaussianLogLike(g["y%d" % k], var * np.eye(len(g["y%d" % k]))),
                                                               _model(g["A%d" % k])) for k in range(nl)]
        prop = Seeded(int(g["M0"]), delta=int(g["delta"]), nCR=int(g["nCR"]), adaptive=bool(g["adaptive"]), gamma=float(g["gamma"]),
                      period=int(g["period"]))
        ch = _Fed.__new__(_Fed)
        ch.feed(np.zeros((0, 1)), [g["u0"][c]] + [g["u%d" % k][c] for k in range(1, nl)], g["ridx"][c] if randomize else None)
        with _DreamzTap(g, c, int(g["delta"])) as tap:
            HierarchyChain.__init__(ch, posts, prop, [int(v) for v in g["subchain_lengths"]], initial_parameters=g["theta0"][c].copy(),
                                    adaptive_error_model=aem, randomize_subchain_length=randomize)
            ch.sample(g["th%d" % (nl - 1)].shape[1] - 1)
        assert not tap.r and not tap.mcr and not tap.eps and all(len(q) == 0 for q in ch._u)
        _check(ch, g, c, nl, with_like=aem is None)
        np.testing.assert_allclose(ch.proposal.scaling, g["scaling"][c], rtol=1e-12)
        np.testing.assert_allclose(ch.proposal.pCR, g["pCR"][c], rtol=1e-10)


class _FedByIndex(_Fed):
    """upper-level uniforms by step index (a different model takes different decisions than the recorded run, so the
    'no uniform drawn here' pattern of a fixture does not apply): full arrays, one value per step of every level"""

    def _uniform(self, level):
        if level == 0:
            return self._u[0].pop(0)
        rung = self.rungs[level]
        done = len(rung.took) - 1 if level == len(self.rungs) - 1 else sum(rung.own)
        return self._u_full[level][done]


@pytest.mark.parametrize("name,nl", [("g8_da_aem_indep", 2), ("g8_mlda_aem", 3)])
def test_diagonal_error_model_host_driver_equals_oracle(golden, name, nl):
    """error_model_covariance='diagonal' (extension: only the diagonal of a bias covariance reaches the likelihood): the host
    driver with the product's AdaptiveGaussianLogLike and the oracle's restatement agree on the same variates -- accept flags
    of every level equal, states and finest densities to 1e-10; and the model differs from the dense one."""
    from oracle import tinyda_oracle as orc
    from tests.test_oracle_aem import aem_levels
    from tests.test_oracle_multilevel import _prop

    g = golden(name)
    sl = [int(g["subchain_length"])] if nl == 2 else [int(v) for v in g["subchain_lengths"]]
    n_fine = g["th%d" % (nl - 1)].shape[1] - 1
    rng = np.random.default_rng(17)
    us = [g["u0"]] + [rng.random(g["u%d" % k].shape) for k in range(1, nl)]  # a uniform for every step of every level
    ref, _ = orc.run_multilevel_aem(aem_levels(g, nl), _prop(g), sl, g["theta0"], g["z"], us, n_fine, "state-independent", diagonal=True)
    dense, _ = orc.run_multilevel_aem(aem_levels(g, nl), _prop(g), sl, g["theta0"], g["z"], us, n_fine, "state-independent")
    assert not np.allclose(ref[nl - 1]["loglike"], dense[nl - 1]["loglike"], rtol=1e-6)
    posts = _aem_posts(g, nl)
    for c in range(g["theta0"].shape[0]):
        ch = _FedByIndex.__new__(_FedByIndex)
        ch.feed(g["z"][c], [us[0][c]] + [np.zeros(0) for _ in range(1, nl)])
        ch._u_full = [None] + [us[k][c] for k in range(1, nl)]
        with _Normals(ch):
            HierarchyChain.__init__(ch, posts(), _proposal(g), sl, initial_parameters=g["theta0"][c].copy(),
                                    adaptive_error_model="state-independent", error_model_covariance="diagonal")
            ch.sample(n_fine)
        for k in range(nl):
            links = ch.level_chain(k)
            took = np.array([t for t, own in zip(ch.rungs[k].took, ch.rungs[k].own) if own] if k < nl - 1 else ch.rungs[k].took, dtype=np.uint8)
            assert np.array_equal(took, ref[k]["accepted"][c]), "level %d" % k
            np.testing.assert_allclose(np.array([ln.parameters for ln in links]), ref[k]["theta"][c], rtol=1e-10, atol=1e-12)
            if k == nl - 1:
                np.testing.assert_allclose([ln.posterior for ln in links], ref[k]["logpost"][c], rtol=1e-10)
