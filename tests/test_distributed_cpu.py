"""The N>1 plumbing on CPU: world_size-2 gloo process group (sharding, barrier, max-over-ranks, pooled moments)."""
import os
import subprocess
import sys

import numpy as np

from tinyda_amd import distributed as tdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r"""
import os, sys, json
sys.path.insert(0, %(root)r)
import torch
from tinyda_amd import distributed as tdist
rank, local_rank, world = tdist.init_process_group("gloo")
off, cnt = tdist.shard_chains(10, rank, world)
tdist.barrier()
mx = tdist.reduce_scalar(1.0 + rank, "max")
sm = tdist.reduce_scalar(cnt, "sum")
g = torch.Generator().manual_seed(5)
x = torch.randn(10, 3, generator=g, dtype=torch.float64)[off:off + cnt]
mean = x.mean(0)
m2 = (x - mean).T @ (x - mean)
n, mu, M2 = tdist.gather_moments(cnt, mean, m2)
# DREAM archive exchange: rank r holds chains [5r, 5r+5) of 3 steps; value encodes (step, global chain)
loc = torch.tensor([[[100.0 * s + off + c, -1.0] for c in range(cnt)] for s in range(3)], dtype=torch.float64)
rows = tdist.gather_archive_rows(loc)
# the overlapped exchange (run_shared_dream(overlap=True)) with a stand-in engine on CPU tensors: what is appended, and when
class FakeEngine:
    n_chains, dim, device = cnt, 2, 0
    def __init__(self):
        self.log, self.step, self.pending = [], 0, []
    def set_archive_auto_append(self, on):
        self.log.append(("auto", bool(on)))
    def run(self, k, params=None, stats=None, accepted=None, sync=True):
        self.log.append(("run", k))
        self.pending = [[[100.0 * (self.step + s) + off + c, float(rank)] for c in range(cnt)] for s in range(k)]
        self.step += k
    def archive_take(self, buf):
        buf.copy_(torch.tensor(self.pending, dtype=torch.float64))
        return len(self.pending)
    def archive_append(self, rows):
        self.log.append(("append", rows[:, 0].tolist()))
    def sync(self):
        self.log.append(("sync",))
_dev = torch.device
torch.device = lambda *a, **k: _dev("cpu")  # the staging buffers of the pipeline live where the engine does: here the CPU
fe = FakeEngine()
tdist.run_shared_dream(fe, 7, 3, overlap=True)
torch.device = _dev
# the distributed-archive protocol (setup_peer_archive / run_peer_dream) with a stand-in engine: handles exchanged once, per block
# run -> collective -> publish (block-synchronous) or at most two unpublished blocks (lagged); sums only at adaptation boundaries
import numpy as np
class PeerEngine:
    n_chains, dim, device = 16, 2, 0
    _dz = dict(M0=4, capacity=100)
    def __init__(self):
        self.log, self.unpub, self.steps = [], 0, 0
    def counters(self):
        return self.steps, 0
    def archive_ipc_handle(self):
        return bytes([rank]) * 64
    def set_archive_peers(self, n, me, handles=None, pointers=None):
        self.log.append(("peers", n, me, [h[0] for h in handles]))
    def run(self, k, params=None, stats=None, accepted=None, sync=True):
        assert self.unpub < 2
        self.unpub += 1
        self.steps += k
        self.log.append(("run", k, self.unpub))
    def archive_local_sums(self):
        return np.full((2, 2), float(rank + 1) * self.steps)
    def archive_publish(self, total=None):
        assert self.unpub >= 1
        self.unpub -= 1
        self.log.append(("publish", None if total is None else float(total[0, 0])))
    def sync(self):
        pass
logs = {}
for lag in (False, True):
    pe = PeerEngine()
    assert tdist.setup_peer_archive(pe) == world
    tdist.run_peer_dream(pe, 11, 3, period=6, lag=lag)
    assert pe.unpub == 0
    logs["lag" if lag else "sync"] = pe.log
# ranks that disagree in their chain count (or hold a count that is not a multiple of the tile) must ALL raise before any of them
# maps a segment or waits in a barrier (the advisor's hang: one rank fails a local check, its peers wait for it forever)
mismatch = []
for n_bad in (16 + 16 * rank, 24):
    bad = PeerEngine()
    bad.n_chains = n_bad
    try:
        tdist.setup_peer_archive(bad)
        mismatch.append("no error")
    except ValueError as exc:
        mismatch.append("ValueError" if not bad.log else "mapped before raising")
logs["mismatch"] = mismatch
with open(os.path.join(%(out)r, "rank%%d.json" %% rank), "w") as fh:
    json.dump(dict(rank=rank, off=off, cnt=cnt, mx=mx, sm=sm, n=n, mu=mu.tolist(), M2=M2.tolist(), rows=rows[:, 0].tolist(),
                   pipeline=fe.log, peer=logs), fh)
"""


def test_shard_chains_partition():
    for n, w in ((4096, 8), (10, 3), (5, 8), (65536, 8)):
        parts = [tdist.shard_chains(n, r, w) for r in range(w)]
        assert sum(c for _, c in parts) == n
        assert all(parts[i][0] + parts[i][1] == parts[i + 1][0] for i in range(w - 1))
        assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


def test_world_size_2_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER % {"root": ROOT, "out": str(tmp_path)})
    import socket

    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    for attempt in range(2):  # (the port is free when picked, not necessarily a moment later: one retry with another)
        sock = socket.socket()
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
        sock.close()
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                            "127.0.0.1", "--master-port", str(port), str(script)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           text=True, env=env, timeout=300)
        if r.returncode == 0 or "address already in use" not in r.stdout.lower():
            break
    assert r.returncode == 0, r.stdout[-3000:]
    import json

    res = [json.load(open(tmp_path / ("rank%d.json" % k))) for k in range(2)]
    assert [d["off"] for d in res] == [0, 5] and [d["cnt"] for d in res] == [5, 5]
    assert all(d["mx"] == 2.0 and d["sm"] == 10.0 and d["n"] == 10.0 for d in res)
    expect = [100.0 * s_ + c for s_ in range(3) for c in range(10)]  # step-major, global chain minor, on every rank
    assert res[0]["rows"] == expect and res[1]["rows"] == expect
    # overlapped exchange: blocks of 3, 3, 1 steps; the rows of block b are appended (canonical order, all ranks alike)
    # only after block b + 1 has been queued -- i.e. before block b + 2 -- and everything has landed at the end
    block = lambda s0, k: [100.0 * s_ + c for s_ in range(s0, s0 + k) for c in range(10)]
    want = [["auto", False], ["run", 3], ["run", 3], ["append", block(0, 3)], ["run", 1], ["append", block(3, 3)],
            ["append", block(6, 1)], ["sync"]]
    assert res[0]["pipeline"] == want and res[1]["pipeline"] == want
    import torch

    x = torch.randn(10, 3, generator=torch.Generator().manual_seed(5), dtype=torch.float64).numpy()
    np.testing.assert_allclose(res[0]["mu"], x.mean(0), rtol=1e-12)
    np.testing.assert_allclose(res[1]["M2"], (x - x.mean(0)).T @ (x - x.mean(0)), rtol=1e-10)
    # distributed archive: handles of both ranks everywhere; steps 11 = blocks 3, 3 (adaptation boundary at 6), 3, 2.
    # block-synchronous: every block is published before the next runs; the boundary block's publish carries the total of the ranks'
    # sums (rank r reports (r + 1) * steps done: 6 + 12 = 18)
    for r_ in range(2):
        sync_log, lag_log = res[r_]["peer"]["sync"], res[r_]["peer"]["lag"]
        assert sync_log[0] == ["peers", 2, r_, [0, 1]]
        assert sync_log[1:] == [["run", 3, 1], ["publish", None], ["run", 3, 1], ["publish", 18.0], ["run", 3, 1], ["publish", None],
                                ["run", 2, 1], ["publish", None]]
        # lagged: a second block runs before the first is published; the boundary block publishes the OLDEST unpublished block with
        # the sums; everything is published at the end
        assert lag_log[1:] == [["run", 3, 1], ["run", 3, 2], ["publish", 18.0], ["run", 3, 2], ["publish", None], ["run", 2, 2],
                               ["publish", None], ["publish", None]]
        # unequal chain counts (16 / 32) and a count that is not a multiple of the tile (24 on both): ValueError on EVERY rank
        assert res[r_]["peer"]["mismatch"] == ["ValueError", "ValueError"]
