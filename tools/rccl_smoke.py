"""The collectives of the N > 1 path on real RCCL with the one GPU a box has: a process group of ONE rank, backend nccl,
TINYDA_FORCE_COLLECTIVES=1 so that tinyda_amd.distributed issues every call it would issue among eight ranks (all_gather of
the DREAM archive rows -- blocking and asynchronous under the next block --, all_reduce of pooled moments and of the timing
scalar, barrier) instead of short-cutting them.  Each result is compared with the same run without a process group: with one
rank the collectives must be the identity, bit for bit.  Start it as
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/rccl_smoke.py
Prints one JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TINYDA_FORCE_COLLECTIVES"] = "1"
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np
import torch
import torch.distributed as dist

from tinyda_amd import distributed as tdist
from tinyda_amd.engine import Engine


def dream_engine(stream=None):
    d, N, M0, K, T = 8, 64, 24, 5, 33
    rng = np.random.default_rng(4)
    A = rng.standard_normal((12, d)) / np.sqrt(d)
    y = rng.standard_normal(12)
    e = Engine(N, d, seed=77, stream=stream)
    e.set_prior(np.zeros(d), np.eye(d))
    e.set_level(0, A, y, 0, 0.25)
    e.set_proposal_dreamz(M0, delta=2, nCR=3, adaptive=True, period=20, gamma=1.02, shared=True, sync_every=K, capacity=M0 + T * N)
    e.set_archive(rng.standard_normal((M0, d)))
    e.init(0.3 * rng.standard_normal((N, d)))
    return e, T, K


def run_dream(overlap):
    ts = torch.cuda.Stream() if overlap else None
    e, T, K = dream_engine(ts.cuda_stream if ts else None)
    dev = torch.device("cuda", 0)
    p = torch.zeros((T, e.n_chains, e.dim), dtype=torch.float64, device=dev)
    a = torch.zeros((T, e.n_chains), dtype=torch.uint8, device=dev)
    tdist.run_shared_dream(e, T, K, p, None, a, overlap=overlap, stream=ts)
    if ts:
        ts.synchronize()
    e.sync()
    rows = e.dreamz_state()["archive_rows"]
    e.close()
    return p.cpu().numpy(), a.cpu().numpy(), rows


def run_peer(lag):
    """the distributed-archive protocol (one rank: its own segment only): barrier per block, lagged = stream-ordered async"""
    ts = torch.cuda.Stream()
    e, T, K = dream_engine(ts.cuda_stream)
    dev = torch.device("cuda", 0)
    p = torch.zeros((T, e.n_chains, e.dim), dtype=torch.float64, device=dev)
    a = torch.zeros((T, e.n_chains), dtype=torch.uint8, device=dev)
    tdist.setup_peer_archive(e)
    tdist.run_peer_dream(e, T, K, p, None, a, period=20, lag=lag, stream=ts)
    ts.synchronize()
    rows = e.dreamz_state()["archive_rows"]
    e.close()
    return p.cpu().numpy(), a.cpu().numpy(), rows


def run_pooled():
    d, m, N, T = 8, 24, 128, 200
    rng = np.random.default_rng(8)
    A = rng.standard_normal((m, d)) / 3
    y = rng.standard_normal(m)
    e = Engine(N, d, seed=21)
    e.set_prior(np.zeros(d), np.eye(d))
    e.set_level(0, A, y, 0, 0.04)
    e.set_proposal(0, 1e-3 * np.eye(d))
    e.init(np.zeros((N, d)))
    params = torch.empty((T, N, d), dtype=torch.float64, device="cuda")
    pam = tdist.PooledAdaptiveMetropolis(e, 1e-3 * np.eye(d), t0=50, period=50)
    pam.run(T, params)
    out = params.cpu().numpy(), pam.sums.cpu().numpy()
    e.close()
    return out


def main():
    torch.cuda.set_device(0)
    out = {}
    # 1. without a process group: the collectives short-cut (reference results)
    assert not tdist._collectives_active()
    ref = {"dream": run_dream(False), "dream_overlap": run_dream(True), "pooled": run_pooled(), "peer": run_peer(False), "peer_lag": run_peer(True)}
    # 2. RCCL, one rank
    rank, local_rank, world = tdist.init_process_group()
    assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1, (dist.is_initialized(), world)
    assert tdist._collectives_active()
    out["backend"] = dist.get_backend()
    tdist.barrier()
    out["reduce_scalar_max"] = tdist.reduce_scalar(1.25, "max")
    assert out["reduce_scalar_max"] == 1.25
    got = {"dream": run_dream(False), "dream_overlap": run_dream(True), "pooled": run_pooled(), "peer": run_peer(False), "peer_lag": run_peer(True)}
    # the distributed archive with one rank is the replicated one: same visibility of the rows, block-synchronous or lagged
    for a_, b_ in (("peer", "dream"), ("peer_lag", "dream_overlap")):
        same = all(np.array_equal(np.asarray(x), np.asarray(y)) for x, y in ((got[a_][0], got[b_][0]), (got[a_][1], got[b_][1])))
        out["%s_equals_%s" % (a_, b_)] = bool(same)
    for k in ref:
        for i, (r, g) in enumerate(zip(ref[k], got[k])):
            same = np.array_equal(np.asarray(r), np.asarray(g))
            out["%s[%d]_identical" % (k, i)] = bool(same)
            assert same, "%s[%d]: a one-rank collective changed the result" % (k, i)
    n, mu, m2 = tdist.gather_moments(10.0, torch.arange(4, dtype=torch.float64, device="cuda"), torch.eye(4, dtype=torch.float64, device="cuda"))
    assert n == 10.0 and torch.equal(mu.cpu(), torch.arange(4, dtype=torch.float64))
    tdist.barrier()
    dist.destroy_process_group()
    out["ok"] = True
    print(json.dumps(out))


if __name__ == "__main__":
    main()
