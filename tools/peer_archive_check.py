"""The distributed shared archive across PROCESSES on the one GPU a box has: two ranks under torch.distributed.run (gloo for the
control collectives), each with an engine for half of the chains on cuda:0, the segments exchanged as IPC handles and mapped with
hipIpcOpenMemHandle -- everything the 8-GPU deployment does except that the peer mapping crosses xGMI there.  Rank 0 also runs the
whole problem on one engine with the replicated archive and compares (bit for bit: no adaptation; to rounding with it).
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tools/peer_archive_check.py
Prints one JSON line on rank 0."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist

from tinyda_amd import distributed as tdist
from tinyda_amd.engine import Engine

D, N, T, M0, K = 8, 128, 47, 24, 5  # (128 chains: 16 per rank with eight ranks; adjusted in main() when the world does not divide it)


DEV = 0  # set in main(): the rank's own GPU on a multi-GPU node, cuda:0 for every rank on a one-GPU box


def make(n, off, theta0, Z0, adaptive):
    e = Engine(n, D, seed=321, chain_offset=off, device=DEV)
    e.set_prior(np.zeros(D), np.eye(D))
    e.set_level_rosenbrock(0, 1.0, 10.0, 0.0, 1.0)
    e.set_proposal_dreamz(M0, delta=2, nCR=3, adaptive=adaptive, period=20, gamma=1.02, shared=True, sync_every=K, capacity=M0 + T * N)
    e.set_archive(Z0)
    e.init(theta0)
    return e


def _lagged_reference(e, P, S, A):
    """one engine, replicated archive, rows of block b appended before block b + 2 (by hand: no process group involved)"""
    e.set_archive_auto_append(False)
    pend, done = [], 0
    while done < T:
        k = min(K, T - done)
        if len(pend) == 2:
            e.archive_append(pend.pop(0))
        e.run(k, P[done:done + k], S[done:done + k], A[done:done + k])
        buf = torch.empty((k, N, D), dtype=torch.float64, device=P.device)
        e.archive_take(buf)
        pend.append(buf.reshape(-1, D))
        done += k
    for rows in pend:
        e.archive_append(rows)
    e.sync()


def main():
    dist.init_process_group("gloo")
    global DEV, N
    rank, world = dist.get_rank(), dist.get_world_size()
    if N % (16 * world):  # a rank's share is a multiple of the 16-chain tile (3, 5, 6 ... ranks)
        N = 16 * world * max(1, N // (16 * world))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    DEV = local_rank if torch.cuda.device_count() > local_rank else 0
    torch.cuda.set_device(DEV)
    rng = np.random.default_rng(5)
    Z0 = rng.standard_normal((M0, D))
    theta0 = 0.3 * rng.standard_normal((N, D))
    h = N // world
    out = {"world": world}
    for adaptive, lag in ((False, False), (True, False), (False, True)):
        e = make(h, rank * h, theta0[rank * h:(rank + 1) * h], Z0, adaptive)
        assert tdist.setup_peer_archive(e) == world
        dev = torch.device("cuda", DEV)
        p = torch.zeros((T, h, D), dtype=torch.float64, device=dev)
        s = torch.zeros((T, h, 3), dtype=torch.float64, device=dev)
        a = torch.zeros((T, h), dtype=torch.uint8, device=dev)
        tdist.run_peer_dream(e, T, K, p, s, a, period=20 if adaptive else None, lag=lag)
        e.sync()
        rows = e.dreamz_state()["archive_rows"]
        mine = [p.cpu(), s.cpu(), a.cpu()]
        gathered = [[torch.empty_like(x) for _ in range(world)] for x in mine]
        for g, x in zip(gathered, mine):
            dist.all_gather(g, x)
        dist.barrier()  # nobody unmaps a segment a peer may still read
        e.close()
        if rank == 0:
            one = make(N, 0, theta0, Z0, adaptive)
            if lag:  # the replicated archive with the same one-block lag
                P_ = torch.zeros((T, N, D), dtype=torch.float64, device=dev)
                S_ = torch.zeros((T, N, 3), dtype=torch.float64, device=dev)
                A_ = torch.zeros((T, N), dtype=torch.uint8, device=dev)
                _lagged_reference(one, P_, S_, A_)
                P, S, A = P_.cpu().numpy(), S_.cpu().numpy(), A_.cpu().numpy()
            else:
                P, S, A = one.run_host(T)
            one.close()
            jp, js, ja = (torch.cat(g, dim=1).numpy() for g in gathered)
            tag = "lagged" if lag else ("adaptive" if adaptive else "plain")
            out[tag + "_rows_ok"] = bool(rows == M0 + T * N)
            out[tag + "_accept_equal"] = bool(np.array_equal(ja, A))
            out[tag + "_max_rel_logpost"] = float(np.max(np.abs(js[:, :, 2] - S[:, :, 2]) / np.abs(S[:, :, 2])))
            out[tag + "_params_equal"] = bool(np.array_equal(jp, P))
    dist.barrier()
    if rank == 0:
        out["ok"] = bool(out["plain_accept_equal"] and out["plain_params_equal"] and out["plain_max_rel_logpost"] == 0.0 and
                         out["adaptive_accept_equal"] and out["adaptive_max_rel_logpost"] < 1e-9 and out["plain_rows_ok"] and out["adaptive_rows_ok"] and
                         out["lagged_accept_equal"] and out["lagged_params_equal"] and out["lagged_max_rel_logpost"] == 0.0 and out["lagged_rows_ok"])
        print(json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
