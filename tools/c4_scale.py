"""BASELINE config 4 (DREAM, shared archive, 32-dim Rosenbrock chain, 8192 chains per GPU) under N ranks, one per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/c4_scale.py --mode M

M: replicated-blocking | replicated-overlapped | distributed-sync | distributed-lagged  (tinyda_amd/distributed.py: run_shared_dream
/ run_peer_dream).  Rank 0 prints ONE JSON line: whole-job evals/s (max-over-ranks time, barrier + synchronize on both sides), every
rank's own rate, rccl_ranks.  TINYDA_BENCH_ONE_GPU=1: every rank on cuda:0 over gloo (rehearsal of the code path, not a measurement).
A distributed mode whose peer mapping fails falls back to the matching replicated mode on every rank and says so in the line."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import numpy as np  # noqa: E402
import torch  # noqa: E402

from tinyda_amd import distributed as tdist  # noqa: E402
from tinyda_amd.engine import Engine  # noqa: E402

MODES = ("replicated-blocking", "replicated-overlapped", "distributed-sync", "distributed-lagged")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=MODES, default=MODES[0])
    ap.add_argument("--chains", type=int, default=8192, help="chains per GPU")
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--sync", type=int, default=16, help="exchange interval (steps)")
    ap.add_argument("--dim", type=int, default=32)
    args = ap.parse_args()
    one_gpu = os.environ.get("TINYDA_BENCH_ONE_GPU") == "1"
    rank, local_rank, world = tdist.init_process_group("gloo" if one_gpu else None)
    if world > torch.cuda.device_count() and not one_gpu:
        raise SystemExit("%d ranks but %d device(s): refusing (TINYDA_BENCH_ONE_GPU=1 rehearses on one GPU)" % (world, torch.cuda.device_count()))
    devi = 0 if one_gpu else local_rank
    torch.cuda.set_device(devi)
    dev = torch.device("cuda", devi)
    N, d, T, K, M0 = args.chains, args.dim, args.steps, args.sync, 320
    warm = 3 * K
    mode, note = args.mode, None
    lag = mode.endswith("overlapped") or mode.endswith("lagged")

    def make(distributed_archive):
        stream = torch.cuda.Stream(device=dev) if lag else None
        e = Engine(N, d, seed=4, device=devi, chain_offset=rank * N, stream=None if stream is None else stream.cuda_stream)
        e.set_prior(np.zeros(d), np.eye(d))
        e.set_level_rosenbrock(0, 1.0, 10.0, 0.0, 1.0)
        rows = M0 + (T + warm) * N * (1 if distributed_archive else world)
        e.set_proposal_dreamz(M0, delta=1, nCR=3, adaptive=True, period=100, shared=True, sync_every=K, capacity=rows)
        e.set_archive(None)
        e.init(None)
        return e, stream

    dist_mode = mode.startswith("distributed")
    e, stream = make(dist_mode)
    if dist_mode:
        try:
            tdist.setup_peer_archive(e)
        except tdist.PeerArchiveUnavailable as exc:
            note = "FELL BACK to the replicated archive: %s" % exc
            sys.stderr.write("[c4_scale] rank %d: %s\n" % (rank, note))
            e.close()
            dist_mode = False
            e, stream = make(False)
    params = torch.empty((T, N, d), dtype=torch.float64, device=dev)
    stats = torch.empty((T, N, 3), dtype=torch.float64, device=dev)
    acc = torch.empty((T, N), dtype=torch.uint8, device=dev)

    def go(n):
        if dist_mode:
            tdist.run_peer_dream(e, n, K, params[:n], stats[:n], acc[:n], period=100, lag=lag, stream=stream)
        else:
            tdist.run_shared_dream(e, n, K, params[:n], stats[:n], acc[:n], overlap=lag, stream=stream)
        if stream is not None:
            stream.synchronize()

    go(warm)
    torch.cuda.synchronize()
    tdist.barrier()
    t0 = time.perf_counter()
    go(T)
    torch.cuda.synchronize()
    mine = time.perf_counter() - t0
    tdist.barrier()
    dt = tdist.reduce_scalar(time.perf_counter() - t0, "max", dev if not one_gpu else torch.device("cpu"))
    rates = [None] * world
    if world > 1:
        import torch.distributed as dist

        dist.all_gather_object(rates, N * T / mine)
    else:
        rates = [N * T / mine]
    rows = e.dreamz_state()["archive_rows"]
    if rank == 0:
        print(json.dumps({"config": "C4: DREAM shared archive, %d-dim Rosenbrock, %d chains per GPU, exchange every %d steps" % (d, N, K),
                          "mode": args.mode, "ran_as": ("distributed" if dist_mode else "replicated") + ("-lagged" if lag else "-blocking"), "note": note,
                          "n_gpus": world, "rccl_ranks": world if (world > 1 and not one_gpu) else (0 if world == 1 else "gloo rehearsal on one GPU"),
                          "steps": T, "evals_per_s": world * N * T / dt, "per_rank_evals_per_s": rates, "seconds": dt,
                          "archive_rows_rank0": rows, "acceptance_rank0": float(acc.float().mean().item())}), flush=True)
    e.close()
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
