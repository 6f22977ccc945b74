"""Post-processing: get_samples (tinyDA/diagnostics.py:114-209) and the ESS / R-hat used for ESS/s.

ArviZ / xarray are not importable in this image, so `to_inference_data` raises unless they are installed;
`ess_bulk` / `rhat` implement the rank-normalised split-chain estimators of Vehtari et al. (2021), the
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


def get_samples(chain, attribute="parameters", level="fine", burnin=0):
    """Result dict of sample() -> per-chain (iterations - burnin, dim) arrays keyed chain_i."""
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
    for i in range(chain["n_chains"]):
        rows = _attribute_rows(chain[key.format(i)], attribute, burnin)
        out["chain_{}".format(i)] = rows[..., np.newaxis] if rows.ndim == 1 else rows
    out["iterations"] = out["chain_0"].shape[0]
    out["dimension"] = out["chain_0"].shape[1]
    return out


def to_inference_data(chain, level="fine", burnin=0, parameter_names=None):
    try:
        import arviz as az
        import xarray as xr
    except ImportError as exc:
        raise ImportError("to_inference_data needs arviz and xarray; use get_samples / ess_bulk / rhat") from exc
    groups = []
    for attr in ("parameters", "model_output", "qoi", "stats"):
        s = get_samples(chain, attr, level, burnin)
        if attr == "parameters":
            names = parameter_names or ["x{}".format(i) for i in range(s["dimension"])]
        elif attr == "stats":
            names = ["prior", "likelihood", "posterior"]
        else:
            names = ["{}_{}".format("obs" if attr == "model_output" else "qoi", i) for i in range(s["dimension"])]
        data = {n: (["chain", "draw"], np.array([s["chain_{}".format(j)][:, i] for j in range(s["n_chains"])]))
                for i, n in enumerate(names)}
        groups.append(xr.Dataset(data, coords=dict(chain=list(range(s["n_chains"])), draw=list(range(s["iterations"])))))
    return az.InferenceData(posterior=groups[0], posterior_predictive=groups[1], qoi=groups[2], sample_stats=groups[3])


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
