"""Soak check of the level kernels: Delayed Acceptance (C3 shape) and 3-level MLDA (C5 shape) on the engine's own Philox stream over
MANY launches against the NumPy oracle (accept masks of every level, log-posterior): what the short forward-mode tests do, long
enough for anything that accumulates across blocks -- e.g. the linearly carried model outputs of k_da_steps -- to show."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import tinyda_amd.engine as eng_mod
from oracle import tinyda_oracle as orc
import tests.test_gpu_multilevel as T

def run(case, n_fine, N):
    rng = np.random.default_rng(78)
    if case == "mlda3":
        d, ms, sl = 64, (128, 512, 2048), [5, 3]
        prop = dict(kind="am", C0=1e-4 * np.eye(d), t0=50, period=50)
    else:
        d, ms, sl = 64, (256, 2048), [10]
        prop = dict(kind="pcn", scaling=0.02, adaptive=True, gamma=1.01, period=40)
    truth = rng.standard_normal(d)
    As = [rng.standard_normal((m, d)) / 8 for m in ms]
    ys = [A @ truth + 0.1 * rng.standard_normal(len(A)) for A in As]
    theta0 = truth + 0.02 * rng.standard_normal((N, d))
    nl = len(ms); seed = 4242
    e = eng_mod.Engine(N, d, seed=seed, n_levels=nl)
    e.set_prior(np.zeros(d), np.eye(d))
    for k in range(nl): e.set_level(k, As[k], ys[k], 0, 0.01)
    if prop["kind"] == "pcn": e.set_proposal(1, None, scaling=prop["scaling"], adaptive=True, gamma=prop["gamma"], period=prop["period"])
    else: e.set_proposal(2, prop["C0"], t0=prop["t0"], period=prop["period"], adaptive=False)
    e.set_subchains(sl, False)
    e.init(theta0)
    rows = e.rows_per_level(n_fine)
    z, _ = e.set_export(rows[0])
    outs = e.run_levels_host(n_fine)
    us, ridx = T._oracle_uniforms(seed, N, rows, sl, None)
    prior = orc.MVNPrior(np.zeros(d), np.eye(d))
    levels = [orc.LinearGaussianLevel(As[k], ys[k], "iso", 0.01, prior) for k in range(nl)]
    t0 = time.time()
    res, pstate = orc.run_multilevel(levels, prop, sl, theta0, np.swapaxes(z, 0, 1), us, n_fine, ridx)
    flips = 0; worst = 0.0
    for k in range(nl):
        ref = res[k]; sk = slice(1, None) if k == nl - 1 else slice(None)
        flips += int((outs[k][2] != ref["accepted"][:, sk].T).sum())
        a, b = outs[k][1][:, :, 2], ref["logpost"][:, sk].T
        worst = max(worst, float(np.max(np.abs(a - b) / np.abs(b))))
    print(case, "fine iterations", n_fine, "chains", N, "rows", rows, "accept flips", flips, "max rel dlogpost %.2e" % worst, "oracle %.0fs" % (time.time() - t0), flush=True)
    e.close()

run("da_c3", 60, 48)
run("mlda3", 40, 32)
