"""Throughput of the other BASELINE configurations (parity-test cases, not the bench line): C3 (2-level DA, pCN,
256/2048 obs, subsampling 10) and C5-literal (3-level MLDA 128/512/2048 obs, AM, no error model), 4096 chains."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tinyda_amd.engine import Engine

def levels(ms, d=64, seed=2, sigma=0.1):
    rng = np.random.default_rng(seed)
    truth = rng.standard_normal(d)
    out = []
    for m in ms:
        A = rng.standard_normal((m, d)) / 8
        out.append((A, A @ truth + sigma * rng.standard_normal(m)))
    return out

def run(name, ms, sl, prop, n_fine, N=4096, d=64):
    lv = levels(ms)
    e = Engine(N, d, seed=9, n_levels=len(ms))
    e.set_prior(np.zeros(d), np.eye(d))
    for k, (A, y) in enumerate(lv):
        e.set_level(k, A, y, 0, 0.01)
    e.set_proposal(**prop)
    e.set_subchains(sl)
    e.init(None)
    rows = e.rows_per_level(n_fine)
    outs = [(torch.empty((r, N, d), dtype=torch.float64, device="cuda"), torch.empty((r, N, 3), dtype=torch.float64, device="cuda"),
             torch.empty((r, N), dtype=torch.uint8, device="cuda")) for r in rows]
    e.run_levels(max(1, n_fine // 10), outs)  # warm-up
    e.set_profiling(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e.run_levels(n_fine, outs)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    p = e.profile()
    res = dict(config=name, chains=N, fine_iterations=n_fine, seconds=dt, coarse_evals_per_s=N * rows[0] / dt,
               finest_iterations_per_s=N * n_fine / dt, acceptance=[float(o[2].float().mean().item()) for o in outs], kernel_ms=p)
    print(json.dumps(res))
    e.close()

def run_c5_aem(N=4096, d=64, m=64, n_fine=20, diagonal=False):
    """C5 with the state-independent adaptive error model: levels share the output dimension (m = 128 is SURVEY
    §8(d)'s C5, the device error-model limit); levels 0/1 AdaptiveGaussianLogLike, level 2 isotropic; AM; subchains [5, 3]."""
    rng = np.random.default_rng(6)
    truth = rng.standard_normal(d)
    Af = rng.standard_normal((m, d)) / 8
    y = Af @ truth + 0.1 * rng.standard_normal(m)
    e = Engine(N, d, seed=10, n_levels=3)
    e.set_prior(np.zeros(d), np.eye(d))
    for k in range(3):
        A = Af + 0.02 * (2 - k) * rng.standard_normal((m, d)) / 8
        if k < 2 and not diagonal:
            e.set_level(k, A, y, 3, 0.01 * np.eye(m))
        else:
            e.set_level(k, A, y, 0, 0.01)
    e.set_proposal(2, 1e-4 * np.eye(d), t0=100, period=100)
    e.set_subchains([5, 3])
    e.set_error_model("state-independent-diagonal" if diagonal else "state-independent")
    e.init(None)
    rows = e.rows_per_level(n_fine)
    outs = [(torch.empty((r, N, d), dtype=torch.float64, device="cuda"), torch.empty((r, N, 3), dtype=torch.float64, device="cuda"),
             torch.empty((r, N), dtype=torch.uint8, device="cuda")) for r in rows]
    e.run_levels(2, outs)
    e.set_profiling(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e.run_levels(n_fine, outs)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps(dict(config="C5 + state-independent AEM (%s), common m=%d, AM, subchains [5,3]" % ("diagonal extension" if diagonal else "dense, reference", m), chains=N, kernel_ms=e.profile(), fine_iterations=n_fine, seconds=dt,
                          coarse_evals_per_s=N * rows[0] / dt, finest_iterations_per_s=N * n_fine / dt,
                          acceptance=[float(o[2].float().mean().item()) for o in outs])))
    e.close()


def run_mala(N=4096, d=64, m=1024, T=2000):
    """C2a target (d = 64, m = 1024, iso noise, prior N(0, I), theta0 ~ prior) sampled with MALA, adaptive scaling: evals/s and
    bulk ESS/s of the second half computed on the device, next to AdaptiveMetropolis on the same target."""
    from tinyda_amd import summaries
    rng = np.random.default_rng(1)
    A = rng.standard_normal((m, d)) / 8
    y = A @ rng.standard_normal(d) + 0.1 * rng.standard_normal(m)
    for name, prop in (("MALA(scaling=0.02, adaptive, period=50)", dict(kind=6, scaling=0.02, adaptive=True, gamma=1.01, period=50)),
                       ("AdaptiveMetropolis(C0=1e-4 I, t0=100, period=100)", dict(kind=2, C_=1e-4 * np.eye(d), t0=100, period=100))):
        e = Engine(N, d, seed=1)
        e.set_prior(np.zeros(d), np.eye(d)); e.set_level(0, A, y, 0, 0.01)
        e.set_proposal(**prop)
        e.init(None)
        p = torch.empty((T, N, d), dtype=torch.float64, device="cuda"); s = torch.empty((T, N, 3), dtype=torch.float64, device="cuda")
        a = torch.empty((T, N), dtype=torch.uint8, device="cuda")
        e.run(200, p[:200], s[:200], a[:200])
        e.set_profiling(True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e.run(T, p, s, a)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        pr = e.profile()
        di = summaries.ess_rhat_device(p, burnin=T // 2)
        ess, rhat = di["ess"], di["rhat"]
        print(json.dumps(dict(config="C2a target, " + name, chains=N, steps=T, evals_per_s=N * T / dt,
                              steps_kernel_ns_per_eval=pr["ms_steps"] * 1e6 / (N * T), acceptance_second_half=float(a[T // 2:].float().mean().item()),
                              min_bulk_ess=float(np.min(ess)), ess_per_s=float(np.min(ess)) / dt, max_rhat=float(np.max(rhat)),
                              scaling_mean=float(np.mean(e.proposal_state()["scaling"])))))
        e.close()


def run_c2b(N=4096, d=64, m=1024, T=300):
    """C2b: dense data covariance (DefaultGaussianLogLike), AM, 4096 chains."""
    rng = np.random.default_rng(1)
    A = rng.standard_normal((m, d)) / 8
    y = A @ rng.standard_normal(d) + 0.1 * rng.standard_normal(m)
    Lc = 0.1 * np.eye(m) + 0.01 * np.tril(rng.standard_normal((m, m)))
    e = Engine(N, d, seed=1)
    e.set_prior(np.zeros(d), np.eye(d)); e.set_level(0, A, y, 2, Lc @ Lc.T)
    e.set_proposal(2, 1e-4 * np.eye(d), t0=100, period=100)
    e.init(None)
    p = torch.empty((T, N, d), dtype=torch.float64, device="cuda"); s = torch.empty((T, N, 3), dtype=torch.float64, device="cuda")
    a = torch.empty((T, N), dtype=torch.uint8, device="cuda")
    e.run(100, p[:100], s[:100], a[:100])
    e.set_profiling(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e.run(T, p, s, a)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    pr = e.profile()
    flops = 2 * m * d + 2 * m * m + 3 * m
    print(json.dumps(dict(config="C2b: dense Sigma (m=1024), AM", chains=N, steps=T, evals_per_s=N * T / dt,
                          steps_kernel_ns_per_eval=pr["ms_steps"] * 1e6 / (N * T),
                          mfma_tflops_survey_accounting=flops * N * T / (pr["ms_steps"] * 1e-3) / 1e12, mfma_peak_tflops=78.6)))
    e.close()


def run_c4(N=8192, d=32, T=400, M0=320, K=16, peer=False, lag=False):
    """C4 on one GPU: d=32 Rosenbrock chain, DREAM (shared archive of M0 prior rows + every chain's states, synchronised
    every K steps), 8192 chains/GPU."""
    from tinyda_amd import distributed as tdist
    e = Engine(N, d, seed=4)
    e.set_prior(np.zeros(d), np.eye(d))
    e.set_level_rosenbrock(0, 1.0, 10.0, 0.0, 1.0)
    e.set_proposal_dreamz(M0, delta=1, nCR=3, adaptive=True, period=100, shared=True, sync_every=K, capacity=M0 + (T + 50) * N)
    e.set_archive(None)
    e.init(None)
    params = torch.empty((T, N, d), dtype=torch.float64, device="cuda")
    stats = torch.empty((T, N, 3), dtype=torch.float64, device="cuda")
    acc = torch.empty((T, N), dtype=torch.uint8, device="cuda")
    if peer:  # the distributed archive's block-wise publish protocol with the one rank a box has (what it costs on the host side)
        tdist.setup_peer_archive(e)
        tdist.run_peer_dream(e, 48, K, params, stats, acc, period=100, lag=lag)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tdist.run_peer_dream(e, T, K, params, stats, acc, period=100, lag=lag)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(json.dumps(dict(config="C4 with the distributed archive protocol (one rank%s), sync every %d" % (", lagged publish" if lag else "", K), chains=N, steps=T, seconds=dt,
                              evals_per_s=N * T / dt, archive_rows=e.dreamz_state()["archive_rows"])))
        e.close()
        return
    tdist.run_shared_dream(e, 48, K, params, stats, acc)  # warm-up
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tdist.run_shared_dream(e, T, K, params, stats, acc)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(json.dumps(dict(config="C4: DREAM (shared archive, sync every %d) on 32-dim Rosenbrock" % K, chains=N, steps=T, seconds=dt,
                          evals_per_s=N * T / dt, acceptance=float(acc.float().mean().item()), archive_rows=e.dreamz_state()["archive_rows"],
                          alg_bytes_per_eval=1500, hbm_equiv_GBps=N * T / dt * 1500 / 1e9)))
    e.close()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] in ("c5aem", "c5aemd"):  # python tools/bench_configs.py c5aem 128 [n_fine]
        run_c5_aem(m=int(sys.argv[2]), n_fine=int(sys.argv[3]) if len(sys.argv) > 3 else 20, diagonal=sys.argv[1] == "c5aemd")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "c3":
        run("C3: DA pCN(0.02) 256/2048 obs, subsampling_rate=10", (256, 2048), [10], dict(kind=1, scaling=0.02), 200)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] in ("c4", "c4peer", "c4peerlag"):
        run_c4(K=int(sys.argv[2]) if len(sys.argv) > 2 else 16, peer=sys.argv[1] != "c4", lag=sys.argv[1] == "c4peerlag")
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "c5":
        run("C5-literal: MLDA AM 128/512/2048 obs, subchains [5,3], no AEM", (128, 512, 2048), [5, 3],
            dict(kind=2, C_=1e-4 * np.eye(64), t0=100, period=100), 60)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "mala":
        run_mala()
        sys.exit(0)
    run_c2b()
    run_c5_aem()
    run_c4()
    run("C3: DA pCN(0.02) 256/2048 obs, subsampling_rate=10", (256, 2048), [10], dict(kind=1, scaling=0.02), 200)
    run("C5-literal: MLDA AM 128/512/2048 obs, subchains [5,3], no AEM", (128, 512, 2048), [5, 3],
        dict(kind=2, C_=1e-4 * np.eye(64), t0=100, period=100), 60)
