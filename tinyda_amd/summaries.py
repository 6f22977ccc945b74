"""Post-processing: get_samples (tinyDA/diagnostics.py:114-209) and the ESS / R-hat used for ESS/s.

ArviZ / xarray are not importable in this image: `to_inference_data` returns real ArviZ objects when they are installed
and a light container with the same groups / names / layout otherwise; `ess_bulk` / `rhat` implement the rank-normalised split-chain estimators of Vehtari et al. (2021), the
definition ArviZ's `summary` (called in the reference's notebooks) uses.
"""
import numpy as np
import scipy.stats as stats

from .records import DeviceChain


def _attribute_rows(chain, attribute, burnin):
    if isinstance(chain, DeviceChain):
        if attribute == "parameters":
            return np.asarray(chain.parameters[burnin:])
        if attribute == "stats":
            return np.asarray(chain.stats[burnin:])
        return np.array([getattr(chain[i], attribute) for i in range(burnin, len(chain))])
    if attribute == "stats":
        return np.array([[l.prior, l.likelihood, l.posterior] for l in chain[burnin:]])
    return np.array([getattr(l, attribute) for l in chain[burnin:]])


def _bulk_rows(chains, attribute, burnin):
    """All chains of a device result at once: when every value is a DeviceChain over the SAME device records, chain i being column
    i, the history crosses PCIe in one chunked pass (DeviceRecords.all_chains_host) instead of one strided gather per chain.
    Returns [n_chains, rows, width] or None (anything else: chain by chain)."""
    first = chains[0]
    if attribute not in ("parameters", "stats") or not isinstance(first, DeviceChain) or first._records is None or burnin < 0:
        return None
    recs = first._records
    if recs.n_chains != len(chains):
        return None
    for i, c in enumerate(chains):
        if not isinstance(c, DeviceChain) or c._records is not recs or c._chain != i or c._rows != slice(None):
            return None
    return recs.all_chains_host(attribute, start=burnin)


def get_samples(chain, attribute="parameters", level="fine", burnin=0):
    """Result dict of sample() -> per-chain (iterations - burnin, dim) arrays keyed chain_i (diagnostics.py:114-209)."""
    out = {"sampler": chain["sampler"], "n_chains": chain["n_chains"], "attribute": attribute}
    if chain["sampler"] == "MH":
        key = "chain_{}"
    elif chain["sampler"] == "DA":
        out["subchain_length"] = chain["subchain_length"]
        out["level"] = level
        key = "chain_" + str(level) + "_{}"
    elif chain["sampler"] == "MLDA":
        out["subchain_lengths"] = chain["subchain_lengths"]
        out["level"] = level
        key = "chain_l" + str(level) + "_{}"
    else:
        raise ValueError("unknown sampler %r" % chain["sampler"])
    chains = [chain[key.format(i)] for i in range(chain["n_chains"])]
    bulk = _bulk_rows(chains, attribute, burnin)
    for i in range(chain["n_chains"]):
        rows = bulk[i] if bulk is not None else _attribute_rows(chains[i], attribute, burnin)
        out["chain_{}".format(i)] = rows[..., np.newaxis] if rows.ndim == 1 else rows
    out["iterations"] = out["chain_0"].shape[0]
    out["dimension"] = out["chain_0"].shape[1]
    return out


class DatasetLite:
    """Stand-in for the xarray.Dataset of one InferenceData group when xarray is not installed: `data_vars` maps a
    variable name to its (chain, draw) array, `coords` holds the two index vectors (diagnostics.py:73-111)."""

    dims = ("chain", "draw")

    def __init__(self, data_vars, n_chains, n_draws):
        self.data_vars = dict(data_vars)
        self.coords = {"chain": np.arange(n_chains), "draw": np.arange(n_draws)}

    def __getitem__(self, name):
        return self.data_vars[name]

    def __contains__(self, name):
        return name in self.data_vars

    def __iter__(self):
        return iter(self.data_vars)

    def __len__(self):
        return len(self.data_vars)

    def to_array(self):
        """[variable, chain, draw]"""
        return np.stack([self.data_vars[k] for k in self.data_vars]) if self.data_vars else np.empty((0, 0, 0))


class InferenceDataLite:
    """What to_inference_data returns when ArviZ is not installed: the same four groups under the same names
    (`posterior`, `posterior_predictive`, `qoi`, `sample_stats`; diagnostics.py:60-66) as DatasetLite objects, and
    `summary()` with the columns of `arviz.summary` this package can compute itself (mean, sd, ess_bulk, r_hat)."""

    def __init__(self, **groups):
        self._groups = list(groups)
        for k, v in groups.items():
            setattr(self, k, v)

    def groups(self):
        return list(self._groups)

    def summary(self, group="posterior"):
        ds = getattr(self, group)
        rows = {}
        for name in ds:
            x = np.asarray(ds[name], dtype=np.float64)
            split_ok = x.shape[1] >= 4
            lo, hi = hdi(x) if x.size > 1 else (float("nan"), float("nan"))
            rows[name] = {"mean": float(x.mean()), "sd": float(x.std(ddof=1)) if x.size > 1 else float("nan"),
                          "hdi_3%": lo, "hdi_97%": hi,
                          "mcse_mean": mcse_mean(x) if split_ok else float("nan"),
                          "ess_bulk": float(ess_bulk(x)) if split_ok else float("nan"),
                          "ess_tail": ess_tail(x) if split_ok else float("nan"),
                          "r_hat": float(rhat(x)) if split_ok else float("nan")}
        return rows


def to_inference_data(chain, level="fine", burnin=0, parameter_names=None):
    """Result dict of sample() -> arviz.InferenceData with the groups and variable names of diagnostics.py:6-70; without
    ArviZ / xarray (not installable here) an InferenceDataLite with the same groups, names and (chain, draw) layout.
    Groups whose attribute the chain does not carry (no model output kept, no qoi) are left empty."""
    try:
        import arviz as az
        import xarray as xr
    except ImportError:
        az = xr = None
    groups = []
    for attr in ("parameters", "model_output", "qoi", "stats"):
        try:
            s = get_samples(chain, attr, level, burnin)
            dim = s["dimension"]
            first = np.asarray(s["chain_0"])
            if first.dtype == object or first.size == 0:  # every link carries None (no qoi / no model output)
                raise TypeError
        except (TypeError, ValueError, AttributeError):
            s, dim = None, 0
        if attr == "parameters":
            names = list(parameter_names) if parameter_names is not None else ["x{}".format(i) for i in range(dim)]
        elif attr == "stats":
            names = ["prior", "likelihood", "posterior"]
        else:
            names = ["{}_{}".format("obs" if attr == "model_output" else "qoi", i) for i in range(dim)]
        n_chains = chain["n_chains"]
        n_draws = s["iterations"] if s is not None else 0
        data = {}
        if s is not None:
            for i, n in enumerate(names[:dim]):
                data[n] = np.array([np.asarray(s["chain_{}".format(j)], dtype=np.float64)[:, i] for j in range(n_chains)])
        if xr is not None:
            groups.append(xr.Dataset({n: (["chain", "draw"], v) for n, v in data.items()},
                                     coords=dict(chain=list(range(n_chains)), draw=list(range(n_draws)))))
        else:
            groups.append(DatasetLite(data, n_chains, n_draws))
    if az is not None:
        return az.InferenceData(posterior=groups[0], posterior_predictive=groups[1], qoi=groups[2], sample_stats=groups[3])
    return InferenceDataLite(posterior=groups[0], posterior_predictive=groups[1], qoi=groups[2], sample_stats=groups[3])


# ---------------------------------------------------------------------------------------------
# ESS / R-hat
# ---------------------------------------------------------------------------------------------
def _split(x):
    """[chains, draws] -> [2*chains, draws//2]"""
    n = x.shape[1] // 2
    return np.concatenate([x[:, :n], x[:, x.shape[1] - n:]], axis=0)


def _rank_normalise(x):
    r = stats.rankdata(x, method="average").reshape(x.shape)
    return stats.norm.ppf((r - 0.375) / (x.size + 0.25))


def _autocov(x):
    """Biased autocovariance of every row via FFT."""
    n = x.shape[1]
    m = 1 << int(np.ceil(np.log2(2 * n)))
    xc = x - x.mean(axis=1, keepdims=True)
    f = np.fft.rfft(xc, n=m, axis=1)
    return np.fft.irfft(f * np.conj(f), n=m, axis=1)[:, :n] / n


def _ess_raw(x):
    """ESS of [chains, draws] with Geyer's initial monotone sequence on the multi-chain autocorrelation."""
    m, n = x.shape
    if n < 4:
        return np.nan
    acov = _autocov(x)
    mean_var = acov[:, 0].mean() * n / (n - 1.0)
    var_plus = mean_var * (n - 1.0) / n
    if m > 1:
        var_plus += x.mean(axis=1).var(ddof=1)
    if not var_plus > 0:
        return np.nan
    rho = 1.0 - (mean_var - acov.mean(axis=0)) / var_plus
    rho[0] = 1.0
    # pair sums P_k = rho[2k] + rho[2k+1]; stop at the first negative pair, enforce monotone decrease
    npair = n // 2
    pairs = rho[0:2 * npair:2] + rho[1:2 * npair:2]
    neg = np.nonzero(pairs < 0)[0]
    kmax = neg[0] if neg.size else npair
    pairs = np.minimum.accumulate(pairs[:kmax])
    tau = -1.0 + 2.0 * pairs.sum()
    if kmax < npair:  # add the positive part of the next even lag, as Stan / ArviZ do
        tau += max(rho[2 * kmax], 0.0)
    tau = max(tau, 1.0 / np.log10(m * n))
    return m * n / tau


def ess_bulk(x):
    """Bulk ESS of one scalar quantity: x [chains, draws] -> float."""
    x = np.asarray(x, dtype=np.float64)
    return _ess_raw(_rank_normalise(_split(x)))


def ess_tail(x):
    """Tail ESS (Vehtari et al. 2021, sec. 4.3; the `ess_tail` column of az.summary): the smaller ESS of the indicators
    I(x <= q05), I(x <= q95) on the split chains."""
    x = np.asarray(x, dtype=np.float64)
    return float(min(_ess_raw(_split((x <= np.quantile(x, prob)).astype(np.float64))) for prob in (0.05, 0.95)))


def mcse_mean(x):
    """Monte Carlo standard error of the mean: sd / sqrt(ESS of the mean) (`mcse_mean` of az.summary)."""
    x = np.asarray(x, dtype=np.float64)
    return float(x.std(ddof=1) / np.sqrt(_ess_raw(_split(x))))


def hdi(x, prob=0.94):
    """Highest density interval of the pooled draws (az.summary's hdi_3% / hdi_97% at the default 0.94): the narrowest interval
    holding `prob` of them."""
    v = np.sort(np.asarray(x, dtype=np.float64).ravel())
    k = int(np.floor(prob * v.size))
    w = v[k:] - v[:v.size - k]
    i = int(np.argmin(w))
    return float(v[i]), float(v[i + k])


def rhat(x):
    """Rank-normalised split R-hat (max of bulk and folded)."""
    x = np.asarray(x, dtype=np.float64)

    def _r(z):
        m, n = z.shape
        w = z.var(axis=1, ddof=1).mean()
        b = n * z.mean(axis=1).var(ddof=1)
        return np.sqrt(((n - 1.0) / n * w + b / n) / w)

    s = _split(x)
    return max(_r(_rank_normalise(s)), _r(_rank_normalise(np.abs(s - np.median(s)))))


def ess_rhat_device(params, burnin=0, device=0, stream=None):
    """Bulk ESS and R-hat of every parameter of a DEVICE history [draws, chains, dim] (a torch tensor or anything with
    data_ptr()/shape), computed on the GPU (tda_diag_ess_rhat: hipCUB radix sort + hipFFT).  Same estimators as
    ess_bulk / rhat above, which are the checker.  Returns dict(ess=[dim], rhat=[dim])."""
    import ctypes as C

    from . import _lib

    lib = _lib.load()
    T, N, d = (int(v) for v in params.shape)
    if hasattr(params, "is_contiguous") and not params.is_contiguous():
        raise ValueError("params must be contiguous [draws, chains, dim]")
    ess = np.empty(d)
    rh = np.empty(d)
    _lib.check(lib.tda_diag_ess_rhat(int(device), C.c_void_p(stream), C.c_void_p(params.data_ptr()), T, N, d, int(burnin),
                                     ess.ctypes.data_as(C.c_void_p), rh.ctypes.data_as(C.c_void_p)))
    return dict(ess=ess, rhat=rh)


def ess_summary(samples, burnin=0):
    """min / median bulk ESS over parameters for a parameters array [draws, chains, dim] (engine layout)."""
    arr = np.asarray(samples)[burnin:]
    vals = np.array([ess_bulk(arr[:, :, j].T) for j in range(arr.shape[2])])
    return dict(ess_min=float(np.nanmin(vals)), ess_median=float(np.nanmedian(vals)), ess=vals)
