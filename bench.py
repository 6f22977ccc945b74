#!/usr/bin/env python3
"""Headline benchmark: proposal evaluations / s of the many-chain MH hot path (BASELINE.json configs[1]):
64-dim Gaussian-linear posterior, 1024 observations, isotropic noise, AdaptiveMetropolis(t0=100, period=100),
4096 chains per GPU, synthetic data of SURVEY.md §8(d) "C2 synthetic input".

    python bench.py --gpus N --steps K --warmup W

A "step" = one Metropolis-Hastings step of every chain on the GPU (propose -> forward model -> log-likelihood
-> log alpha -> accept -> record, with the AdaptiveMetropolis recursion and covariance swaps included).
For N > 1 launch under torch.distributed.run; chains are sharded by global id, no data-path collective.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL (set before the HIP runtime loads)

D, M, SIGMA = 64, 1024, 0.1
CHAINS_PER_GPU = 4096
HBM_PEAK = 8.0e12        # B/s, MI355X_MICROARCH.md "HBM3E peak BW"
FP64_MFMA_PEAK = 78.6e12  # FLOP/s dense fp64 matrix (AMD MI355X spec; tools/mfma_probe measures 75-77 on the box)
B_ALG_STEP_SYNC = 40 * D + 25 + 24 * D * D + 8 * (M * D + M) / CHAINS_PER_GPU  # 101 019 B / eval
FLOPS_STEPS_KERNEL = 2 * M * D + 3 * M + 2 * D  # forward + isotropic log-like + identity prior


def c2_problem(seed=1):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((M, D)) / 8
    theta_true = rng.standard_normal(D)
    y = A @ theta_true + SIGMA * rng.standard_normal(M)
    return A, theta_true, y


def cpu_baseline(A, y, target_seconds=12.0):
    """The C oracle (oracle/oracle_mh.c, kind 'port') on all host cores, bounded sample of the same workload."""
    from oracle import oracle_c

    oracle_c.load()
    cores = os.cpu_count() or 1
    rng = np.random.default_rng(7)

    def run(n_chains, T):
        theta0 = rng.standard_normal((n_chains, D))
        z = rng.standard_normal((T, n_chains, D))
        u = rng.random((T, n_chains))
        t0 = time.perf_counter()
        oracle_c.run_mh(A, y, SIGMA ** 2, np.zeros(D), np.ones(D), 2, 1e-4 * np.eye(D), theta0, z, u, period=100, t0=100,
                        n_threads=cores, want_records=False)
        return time.perf_counter() - t0

    T = 250
    t_cal = run(cores, 50)
    rate = cores * 50 / t_cal
    n_chains = int(max(cores, min(4096, round(rate * target_seconds / T / cores) * cores)))
    dt = run(n_chains, T)
    return {"value": n_chains * T / dt, "unit": "evals/s", "cores": cores, "kind": "port",
            "sample": "%d chains x %d steps of the same workload (C oracle, OpenMP over chains, %.1f s)" % (n_chains, T, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--chains", type=int, default=CHAINS_PER_GPU, help="chains per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ess", action="store_true")
    args = ap.parse_args()

    import torch

    from tinyda_amd import summaries as diagnostics
    from tinyda_amd import distributed as tdist
    from tinyda_amd.engine import Engine

    # TINYDA_BENCH_ONE_GPU=1 (testing the N > 1 code path on a one-GPU box): every rank on cuda:0, gloo instead of RCCL
    one_gpu = os.environ.get("TINYDA_BENCH_ONE_GPU") == "1"
    rank, local_rank, world = tdist.init_process_group("gloo" if one_gpu else None)
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    N, K, W = args.chains, args.steps, args.warmup

    A, _, y = c2_problem()
    eng = Engine(N, D, seed=2026, device=local_rank, chain_offset=rank * N)  # weak scaling: N chains per GPU
    eng.set_prior(np.zeros(D), np.eye(D))
    eng.set_level(0, A, y, 0, SIGMA ** 2)
    eng.set_proposal(2, 1e-4 * np.eye(D), t0=100, period=100, sd=None, epsilon=1e-6)
    eng.init(None)  # theta0 ~ prior, Philox stream 2 keyed by global chain id

    params = torch.empty((max(K, W), N, D), dtype=torch.float64, device=dev)
    stats = torch.empty((max(K, W), N, 3), dtype=torch.float64, device=dev)
    acc = torch.empty((max(K, W), N), dtype=torch.uint8, device=dev)
    if W > 0:
        eng.run(W, params[:W], stats[:W], acc[:W])
    eng.set_profiling(True)  # HIP events around every kernel launch, on the engine's stream
    torch.cuda.synchronize()
    tdist.barrier()
    t0 = time.perf_counter()
    eng.run(K, params[:K], stats[:K], acc[:K], sync=True)
    torch.cuda.synchronize()
    tdist.barrier()
    dt = tdist.reduce_scalar(time.perf_counter() - t0, "max", dev)
    prof = eng.profile()

    evals = world * N * K
    out = {
        "metric": "proposal evals/sec (node), 64-dim AdaptiveMetropolis, 4096 chains/GPU",
        "value": evals / dt,
        "unit": "evals/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": dt / K * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[1] / SURVEY C2a: d=64 Gaussian-linear posterior, m=1024 obs, isotropic noise, "
                               "prior N(0,I), AdaptiveMetropolis(C0=1e-4 I, t0=100, period=100), theta0 ~ prior",
                   "chains_per_gpu": N, "dim": D, "observations": M, "records": "params+stats+accepted to HBM every step"},
    }
    if rank == 0:
        ev_rank = N * K
        kern = {"k_mh_steps": (prof["ms_steps"], prof["n_launch_steps"]),
                "k_adapt": (prof["ms_adapt"], prof["n_launch_adapt"]),
                "k_apply": (prof["ms_propose"], prof["n_launch_propose"])}  # k_rng runs under k_mh_steps on a 2nd stream
        ms_st, n_st = kern["k_mh_steps"]
        avg_launch_s = ms_st * 1e-3 / max(n_st, 1)
        evals_per_launch = ev_rank / max(n_st, 1)
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_i_pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get("k_mh_steps_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        achieved = FLOPS_STEPS_KERNEL * evals_per_launch / avg_launch_s
        out["roofline"] = {"kernel": "k_mh_steps", "bound": "mfma", "achieved": achieved / 1e12, "peak": FP64_MFMA_PEAK / 1e12,
                           "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK, "traffic": traffic,
                           "flops_per_eval": FLOPS_STEPS_KERNEL, "evals_per_launch": evals_per_launch,
                           "avg_launch_ms": avg_launch_s * 1e3}
        # the figure north_star quotes: step-synchronous algorithmic bytes against HBM peak, whole pipeline
        rate_gpu = ev_rank / dt
        out["roofline_hbm_step_synchronous"] = {"bound": "hbm", "bytes_per_eval": B_ALG_STEP_SYNC,
                                                "achieved": rate_gpu * B_ALG_STEP_SYNC / 1e9, "peak": HBM_PEAK / 1e9,
                                                "unit": "GB/s", "frac": rate_gpu * B_ALG_STEP_SYNC / HBM_PEAK,
                                                "note": "period-blocked pipeline keeps Sigma/L traffic off the per-step path; >1 means the "
                                                        "step-synchronous HBM model no longer binds"}
        out["kernel_ms"] = {k: {"total_ms": v[0], "launches": v[1], "ns_per_eval": v[0] * 1e6 / ev_rank} for k, v in kern.items()}
        out["acceptance_rate"] = float(acc[:K].float().mean().item())
        if not args.no_ess:
            # ESS/s: min-over-parameters bulk ESS (rank-normalised, split chains) of the second half of the timed draws of
            # ALL chains of this GPU, computed on the device (tda_diag_ess_rhat: hipCUB sort + hipFFT); chains are
            # independent and identically set up on every GPU, so the node figure is world x the rank-0 figure.
            dd = diagnostics.ess_rhat_device(params[K // 2:K], device=local_rank)
            ess_min, ess_med = float(np.nanmin(dd["ess"])), float(np.nanmedian(dd["ess"]))
            out["ess_per_sec"] = ess_min * world / dt
            out["ess"] = {"min_bulk_ess_node": ess_min * world, "median_bulk_ess_node": ess_med * world,
                          "max_rhat": float(np.nanmax(dd["rhat"])), "chains_used": N, "draws_per_chain": K - K // 2,
                          "computed": "on device, all chains"}
        extras = world == 1  # the side measurements below are single-GPU figures (and the pooled one holds a collective)
        if extras and not args.no_ess:
            # The C2 recipe starts every chain from a prior draw with C0 = 1e-4 I, so the timed window is still burn-in
            # and its ESS is dominated by between-chain variance.  For reference, the same kernel pipeline started in
            # stationarity (theta0 ~ exact conjugate posterior, C0 = sd * posterior covariance): ESS/s of the sampler
            # itself.  Short separate run, outside the timed region, 512 chains' worth of draws scaled like above.
            cov_post = np.linalg.inv(A.T @ A / SIGMA ** 2 + np.eye(D))
            mean_post = cov_post @ (A.T @ y / SIGMA ** 2)
            rs = np.random.default_rng(3)
            th_st = mean_post + rs.standard_normal((N, D)) @ np.linalg.cholesky(cov_post).T
            e2 = Engine(N, D, seed=77, device=local_rank)
            e2.set_prior(np.zeros(D), np.eye(D))
            e2.set_level(0, A, y, 0, SIGMA ** 2)
            e2.set_proposal(2, min(1.0, 2.4 ** 2 / D) * cov_post, t0=100, period=100)
            e2.init(th_st)
            Ks = min(K, 2000)
            e2.run(100, params[:100], stats[:100], acc[:100])  # first run() of an engine allocates its block buffers
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            e2.run(Ks, params[:Ks], stats[:Ks], acc[:Ks], sync=True)
            torch.cuda.synchronize()
            dts = time.perf_counter() - t1
            ess2 = {"ess_min": float(np.nanmin(diagnostics.ess_rhat_device(params[:Ks], device=local_rank)["ess"]))}
            sub = N
            out["ess_stationary_start"] = {"ess_per_sec_per_gpu": ess2["ess_min"] * (N / sub) / dts, "min_bulk_ess": ess2["ess_min"] * (N / sub),
                                           "steps": Ks, "seconds": dts, "acceptance_rate": float(acc[:Ks].float().mean().item()),
                                           "note": "per-chain AM (reference semantics): 100 draws cannot estimate a 64x64 covariance, mixing is slow for any implementation"}
            e2.close()
            # extension: ONE covariance pooled over all chains (and GPUs, one all_reduce of 1+d+d^2 doubles per period),
            # started from C0 = 1e-4 I in stationarity: after the first period it has the posterior covariance
            from tinyda_amd.distributed import PooledAdaptiveMetropolis

            e3 = Engine(N, D, seed=78, device=local_rank, chain_offset=rank * N)
            e3.set_prior(np.zeros(D), np.eye(D))
            e3.set_level(0, A, y, 0, SIGMA ** 2)
            e3.set_proposal(0, 1e-4 * np.eye(D))
            e3.init(th_st)  # same stationary start; the pooled covariance is learnt from the chain cloud itself
            pam = PooledAdaptiveMetropolis(e3, 1e-4 * np.eye(D), t0=100, period=100)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            pam.run(Ks, params[:Ks], stats[:Ks], acc[:Ks])
            torch.cuda.synchronize()
            dtp = time.perf_counter() - t2
            ess3 = {"ess_min": float(np.nanmin(diagnostics.ess_rhat_device(params[Ks // 2:Ks], device=local_rank)["ess"]))}
            out["ess_pooled_am_extension"] = {"ess_per_sec_per_gpu": ess3["ess_min"] * (N / sub) / dtp, "min_bulk_ess_second_half": ess3["ess_min"] * (N / sub),
                                              "evals_per_sec_per_gpu": N * Ks / dtp, "steps": Ks, "seconds": dtp,
                                              "acceptance_rate_second_half": float(acc[Ks // 2:Ks].float().mean().item())}
            e3.close()
        if extras and not args.no_ess:
            # the same target and start (theta0 ~ prior, same seed) sampled with the reference's MALA proposal
            # (proposal.py:861-1005), adaptive scaling: what the choice of proposal does to ESS/s on this engine.  Not the headline.
            e5 = Engine(N, D, seed=2026, device=local_rank, chain_offset=rank * N)
            e5.set_prior(np.zeros(D), np.eye(D))
            e5.set_level(0, A, y, 0, SIGMA ** 2)
            e5.set_proposal(6, None, scaling=0.02, adaptive=True, gamma=1.01, period=50)
            e5.init(None)
            if W > 0:
                e5.run(W, params[:W], stats[:W], acc[:W])
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            e5.run(K, params[:K], stats[:K], acc[:K], sync=True)
            torch.cuda.synchronize()
            dtm = time.perf_counter() - t4
            dm = diagnostics.ess_rhat_device(params[K // 2:K], device=local_rank)
            out["mala_same_target"] = {"evals_per_sec_per_gpu": N * K / dtm, "ess_per_sec_per_gpu": float(np.nanmin(dm["ess"])) / dtm,
                                       "min_bulk_ess": float(np.nanmin(dm["ess"])), "max_rhat": float(np.nanmax(dm["rhat"])),
                                       "acceptance_rate_second_half": float(acc[K // 2:K].float().mean().item()), "seconds": dtm,
                                       "proposal": "MALA(scaling=0.02, adaptive=True, period=50), exact gradient c - H theta on the device"}
            e5.close()
        # extension, not the headline: AdaptiveMetropolis(block_moments=True) -- the running covariance as one rank-S update
        # per block on the matrix cores instead of the reference's elementwise recursion (parity 1e-8 instead of 1e-10)
        if extras:
            e4 = Engine(N, D, seed=2026, device=local_rank, chain_offset=rank * N)
            e4.set_prior(np.zeros(D), np.eye(D))
            e4.set_level(0, A, y, 0, SIGMA ** 2)
            e4.set_proposal(2, 1e-4 * np.eye(D), t0=100, period=100, block_moments=True)
            e4.init(None)
            if W > 0:
                e4.run(W, params[:W], stats[:W], acc[:W])
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            e4.run(K, params[:K], stats[:K], acc[:K], sync=True)
            torch.cuda.synchronize()
            dtb = time.perf_counter() - t3
            out["block_moments_extension"] = {"evals_per_sec_per_gpu": N * K / dtb, "seconds": dtb,
                                              "acceptance_rate": float(acc[:K].float().mean().item())}
            e4.close()
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(A, y)
        print(json.dumps(out))
    eng.close()
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
