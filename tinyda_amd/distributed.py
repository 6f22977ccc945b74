"""Multi-GPU plumbing: one process per GPU, chains sharded by global id, no data-path collective.

tinyDA's only parallelism is one Ray actor per chain (tinyDA/ray.py:12-91); chains never interact for
GaussianRandomWalk / CrankNicolson / AdaptiveMetropolis (sampler.py:176 deep-copies the proposal per chain),
so MH shards trivially: rank r owns global chains [offset_r, offset_r + n_r) and the RNG is keyed by the
global id.  torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" on CPU) is used only for the barrier /
max-over-ranks timing and for gathering summary statistics.
"""
import os

import numpy as np


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_chains(n_total, rank, world):
    """Contiguous block partition: returns (offset, count) with counts differing by at most one."""
    base, extra = divmod(int(n_total), int(world))
    count = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return offset, count


def _collectives_active():
    """True when the collectives of this module should actually run: a process group of more than one rank, or of ONE rank
    with TINYDA_FORCE_COLLECTIVES=1 (tools/rccl_smoke.py: on a one-GPU box that is the only way to put the very RCCL calls
    of the N > 1 path -- all_gather, all_reduce, stream ordering of async work -- on real hardware)."""
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("TINYDA_FORCE_COLLECTIVES") == "1"


def init_process_group(backend=None):
    import torch
    import torch.distributed as dist

    rank, local_rank, world = env_rank_world()
    # the host driver of this pool only supports dmabuf IPC: without it RCCL fails with hipIpcGetMemHandle: invalid argument
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    forced = os.environ.get("TINYDA_FORCE_COLLECTIVES") == "1" and "WORLD_SIZE" in os.environ
    if (world > 1 or forced) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":  # RCCL binds a rank's communicator to the current device: one GPU per local rank, chosen first
            torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def barrier():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def reduce_scalar(value, op="max", device=None):
    """max / sum of a Python float over ranks (identity when not distributed)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
    return float(t.item())


def gather_moments(count, mean, m2):
    """Pool per-rank sample moments (n, mean [d], centred second moment [d,d]) with one all_reduce each
    (Chan et al. parallel update expressed through sums); used for pooled summaries across GPUs."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        return count, mean, m2
    dev = mean.device
    n = torch.tensor([float(count)], dtype=torch.float64, device=dev)
    s1 = mean * count
    s2 = m2 + count * torch.outer(mean, mean)
    for t in (n, s1, s2):
        dist.all_reduce(t)
    tot = float(n.item())
    mu = s1 / tot
    return tot, mu, s2 - tot * torch.outer(mu, mu)


def gather_archive_rows(local_rows):
    """DREAM shared archive: local_rows [steps, n_local, d] (same n_local on every rank) -> all ranks' rows in the
    canonical order the engine appends them, [steps * n_total, d] = step-major, global chain id minor.  One
    all_gather per synchronisation (RCCL over xGMI on GPUs, gloo on CPU); identity when not distributed."""
    import torch
    import torch.distributed as dist

    if not _collectives_active():
        return local_rows.reshape(-1, local_rows.shape[-1]).contiguous()
    parts = [torch.empty_like(local_rows) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, local_rows.contiguous())
    return torch.cat(parts, dim=1).reshape(-1, local_rows.shape[-1]).contiguous()


def _canonical_rows(gathered, world, k, n_local, dim):
    """[world][k][n_local][dim] as gathered by rank -> [k * world * n_local][dim] in canonical order (step major, global chain
    id minor: rank r owns the contiguous global ids [r n_local, (r + 1) n_local))"""
    return gathered.reshape(world, k, n_local, dim).permute(1, 0, 2, 3).reshape(-1, dim).contiguous()


def run_shared_dream(engine, n_iterations, sync_every, params=None, stats=None, accepted=None, overlap=False, stream=None):
    """Drive a shared-archive DREAM engine under a process group.

    overlap=False: `sync_every` steps, then one all_gather of the new rows, appended identically on every rank before the next
    block starts (tests/test_gpu_dreamz.py checks the result is independent of sharding); the exchange runs serialised with
    the compute.

    overlap=True: the exchange of block b runs UNDER the compute of block b + 1.  The rows block b produced become visible to
    the proposals of block b + 2 (one block later than above; the reference's own archive is updated fire-and-forget by a Ray
    actor, ray.py:365-384, so what a proposal sees there depends on timing -- here the lag is fixed and the same for any
    number of ranks, 1 included, so results stay independent of the sharding).  Per block: append the gathered rows of block
    b - 2 (a stream-side wait on that collective, no host wait), run block b, copy its rows into a staging buffer on the
    engine's stream, start the all_gather asynchronously.  `stream`: the torch.cuda.Stream the engine was created on
    (Engine(..., stream=s.cuda_stream)); the collective is ordered behind the engine's work on it."""
    import contextlib

    import torch
    import torch.distributed as dist

    multi = _collectives_active()
    world = dist.get_world_size() if multi else 1
    N, d = engine.n_chains, engine.dim
    if not overlap:
        if not multi:
            # one process: its rows are all rows; the engine appends each `sync_every`-step block in place (same archive
            # contents and order as the exchange below produces with one rank), no copies, no host synchronisation
            # ONE call: the engine cuts the run into `sync_every`-step blocks itself (tda_engine_set_proposal_dreamz), and a
            # call per block from here left the GPU idle a third of the time (C4: 29 us of host work per 117 us block)
            engine.set_archive_auto_append(True)
            # (the engine's exchange points are the call-relative multiples of its interval, a block cut by an adaptation boundary
            # is exchanged whole -- exactly what the loop further down does with several ranks -- so one call is the same run)
            if getattr(engine, "_dz", {}).get("sync_every") == sync_every:
                engine.run(n_iterations, params, stats, accepted)
                return
            for done in range(0, n_iterations, sync_every):  # an interval other than the engine's own: block by block
                sl = slice(done, min(done + sync_every, n_iterations))
                engine.run(sl.stop - sl.start, None if params is None else params[sl], None if stats is None else stats[sl],
                           None if accepted is None else accepted[sl])
            return
        engine.set_archive_auto_append(False)
        done = 0
        # archive_take queues its copy on the engine's stream: the collective must be ordered behind it, so it is issued with
        # that stream current (an engine on the default stream: legacy null-stream ordering does the same)
        with (torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()):
            while done < n_iterations:
                k = min(sync_every, n_iterations - done)
                sl = slice(done, done + k)
                engine.run(k, None if params is None else params[sl], None if stats is None else stats[sl],
                           None if accepted is None else accepted[sl])
                buf = torch.empty((k, N, d), dtype=torch.float64, device=torch.device("cuda", engine.device))
                engine.archive_take(buf)
                if stream is None:
                    engine.sync()  # no torch stream to order the collective behind: wait for the copy on the host
                engine.archive_append(gather_archive_rows(buf))
                done += k
        return

    # ---- overlapped exchange, fixed lag of one block ----
    dev = torch.device("cuda", engine.device)
    ctx = torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()
    engine.set_archive_auto_append(False)
    stage = [torch.empty((sync_every, N, d), dtype=torch.float64, device=dev) for _ in range(3)]
    recv = [[torch.empty((sync_every, N, d), dtype=torch.float64, device=dev) for _ in range(world)] for _ in range(3)] if multi else None
    in_flight = []  # (work or None, block index, steps)

    def land(entry):
        work, b, k = entry
        if multi:
            work.wait()  # orders the current (= the engine's) stream behind the collective
            rows = _canonical_rows(torch.stack([t[:k] for t in recv[b % 3]]), world, k, N, d)
        else:
            rows = stage[b % 3][:k].reshape(-1, d)
        engine.archive_append(rows)

    with ctx:
        done, b = 0, 0
        while done < n_iterations:
            k = min(sync_every, n_iterations - done)
            sl = slice(done, done + k)
            if len(in_flight) == 2:
                land(in_flight.pop(0))
            engine.run(k, None if params is None else params[sl], None if stats is None else stats[sl],
                       None if accepted is None else accepted[sl], sync=False)
            mine = stage[b % 3][:k]
            engine.archive_take(mine)
            work = None
            if multi:
                outs = [t[:k] for t in recv[b % 3]] if k == sync_every else None
                if outs is None:  # ragged last block: exact-size views are required by the collective
                    recv[b % 3] = [torch.empty((k, N, d), dtype=torch.float64, device=dev) for _ in range(world)]
                    outs = recv[b % 3]
                work = dist.all_gather(outs, mine.contiguous(), async_op=True)
            in_flight.append((work, b, k))
            done += k
            b += 1
        for entry in in_flight:  # the archive ends complete
            land(entry)
        engine.sync()


class PeerArchiveUnavailable(RuntimeError):
    """raised on EVERY rank when any rank could not map its peers' archive segments (hipIpcOpenMemHandle across devices refused:
    no peer access, an IPC mode the driver does not support, processes on different nodes)"""


def setup_peer_archive(engine):
    """Distributed shared archive (include/tinyda_amd.h: tda_engine_set_archive_peers): exchange the IPC handles of the ranks'
    archive segments once and map them.  One process per GPU of one node (peer access over xGMI); with one rank it is a no-op
    apart from switching the engine to the block-wise publish protocol.  If any rank fails to map a segment, every rank raises
    PeerArchiveUnavailable (agreed through one more small collective, so nobody is left waiting in a barrier); callers fall back
    to the replicated archive."""
    import torch.distributed as dist

    if not _collectives_active():
        engine.set_archive_peers(1, 0, pointers=[engine.archive_pointer()])
        return 1
    world, rank = dist.get_world_size(), dist.get_rank()
    # k_dreamz_draw<., true> maps a global archive row to (owner, local row) by dividing by ONE chain count, and reads the owner's
    # segment up to ONE capacity: every rank must hold the same number of chains (a multiple of the 16-chain tile), dimension,
    # M0 and segment capacity.  Checked on every rank from the same gathered list, BEFORE any rank maps a segment or enters a
    # barrier, so a mismatch raises everywhere instead of hanging the ranks that happened to pass a local check.
    dz = getattr(engine, "_dz", {})
    mine = (engine.n_chains, engine.dim, dz.get("M0"), dz.get("capacity"))
    entries = [None] * world
    dist.all_gather_object(entries, (mine, engine.archive_ipc_handle()))
    shapes = [e[0] for e in entries]
    if any(sh != shapes[0] for sh in shapes) or shapes[0][0] % 16 != 0:
        raise ValueError("distributed DREAM archive: every rank needs the same (n_chains, dim, M0, capacity) and n_chains a multiple of 16; "
                         "got %s" % (shapes,))
    err = None
    try:
        engine.set_archive_peers(world, rank, handles=[e[1] for e in entries])
    except Exception as exc:  # EngineError: hipIpcOpenMemHandle / peer access refused
        err = "rank %d: %s" % (rank, exc)
    errs = [None] * world
    dist.all_gather_object(errs, err)
    if any(errs):
        raise PeerArchiveUnavailable("; ".join(x for x in errs if x))
    return world


def run_peer_dream(engine, n_iterations, sync_every, params=None, stats=None, accepted=None, period=None, lag=False, stream=None):
    """Drive a DREAM engine whose shared archive is DISTRIBUTED (setup_peer_archive): per exchange interval one run(), then ONE
    small collective -- a barrier; at adaptation boundaries the all-gather of the ranks' column sums of the visible rows, 2 d
    doubles per rank -- then publish.  No row ever travels unless a proposal reads it.  `period`: the proposal's adaptation
    period (a run() call must not cross an adaptation boundary); None = no adaptation.

    lag=False: block b's rows are visible to block b + 1; every block ends in a host round trip (stream sync, collective, publish).

    lag=True: block b's rows are visible from block b + 2 (the same fixed lag as run_shared_dream(overlap=True), for any number
    of ranks): the barrier of block b is issued behind the engine's work on a side stream and runs under block b + 1; the engine's
    stream waits for it (no host wait) before block b + 2 draws.  Only adaptation boundaries synchronise with the host.  `stream`:
    the torch.cuda.Stream the engine was created on.  With a CPU backend (gloo) the barrier cannot be stream-ordered and is taken
    right away -- the same results, none of the overlap."""
    import contextlib

    import torch
    import torch.distributed as dist

    multi = _collectives_active()
    world = dist.get_world_size() if multi else 1
    on_gpu = multi and dist.get_backend() == "nccl"
    dev = torch.device("cuda", engine.device)
    backend_dev = dev if on_gpu else torch.device("cpu")
    ctx = torch.cuda.stream(stream) if stream is not None else contextlib.nullcontext()
    side = torch.cuda.Stream(device=dev) if (lag and on_gpu) else None
    token = torch.zeros(1, device=dev) if on_gpu else None
    in_flight = []  # per unpublished block: the async work of its barrier (None = already taken)

    def meet():
        """every rank has finished everything it has queued so far: returns the async work (GPU backend, lagged) or None"""
        if not multi:
            return None
        if side is not None:
            side.wait_stream(stream if stream is not None else torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                return dist.all_reduce(token, async_op=True)
        engine.sync()
        dist.barrier()
        return None

    def publish_oldest(total=None):
        work = in_flight.pop(0)
        if work is not None:
            work.wait()  # orders the current (= the engine's) stream behind the collective, no host wait
        engine.archive_publish(total)

    done, t = 0, engine.counters()[0]  # steps the engine has taken so far, whoever drove them (adaptation boundaries count from 0)
    with ctx:
        while done < n_iterations:
            k = min(sync_every, n_iterations - done)
            if period:
                k = min(k, period - (t % period))
            sl = slice(done, done + k)
            if len(in_flight) == (2 if lag else 1):
                publish_oldest()
            engine.run(k, None if params is None else params[sl], None if stats is None else stats[sl],
                       None if accepted is None else accepted[sl], sync=False)
            boundary = bool(period) and (t + k) % period == 0
            if boundary:
                # the crossover adaptation needs the column sums of the archive the block proposed from: the ranks' partial sums
                # (2 d doubles each), added in rank order -- the same bits on every rank; the collective is a barrier as well
                local = torch.from_numpy(engine.archive_local_sums()).to(backend_dev)  # (waits for the engine's stream)
                if multi:
                    parts = [torch.empty_like(local) for _ in range(world)]
                    dist.all_gather(parts, local)
                    total = torch.stack(parts).sum(dim=0)
                else:
                    total = local
                in_flight.append(None)  # this block is complete on every rank
                for i in range(len(in_flight)):  # ... and so is everything before it
                    in_flight[i] = None
                publish_oldest(total.cpu().numpy())  # adds the sums, adapts, publishes the oldest unpublished block
            else:
                in_flight.append(meet() if lag else None)
                if not lag:
                    engine.sync()
                    if multi:
                        dist.barrier()
            done += k
            t += k
        while in_flight:  # the archive ends complete
            publish_oldest()
    engine.sync()


class PooledAdaptiveMetropolis:
    """Extension (not in tinyDA, whose AdaptiveMetropolis is strictly per chain): one proposal covariance for all
    chains on all GPUs, C = sd (Cov_pooled + eps I), refreshed every `period` steps once t >= t0.  Each refresh costs
    one device reduction over the new records and ONE all_reduce of 1 + d + d^2 doubles (33 KB at d = 64, latency
    bound over xGMI), never per step.  Drives a GaussianRandomWalk engine."""

    def __init__(self, engine, C0, sd=None, epsilon=1e-6, t0=0, period=100):
        import torch

        self.e, self.d = engine, engine.dim
        self.sd = min(1.0, 2.4 ** 2 / self.d) if sd is None else sd
        self.eps, self.t0, self.period = epsilon, t0, period
        self.t = 0
        self.sums = torch.zeros(1 + self.d + self.d * self.d, dtype=torch.float64, device=torch.device("cuda", engine.device))
        self.C = np.asarray(C0, dtype=np.float64)

    def absorb(self, rows):
        """rows: device tensor [..., d] of chain states to add to the pooled moments (local part; all-reduced here)."""
        import torch
        import torch.distributed as dist

        part = torch.empty_like(self.sums)
        self.e.reduce_moments(rows, part)
        if _collectives_active():
            dist.all_reduce(part)  # RCCL: {n, sum x, sum x x^T}
        self.sums += part

    def covariance(self):
        s = self.sums.cpu().numpy()
        n, s1, s2 = s[0], s[1:1 + self.d], s[1 + self.d:].reshape(self.d, self.d)
        mean = s1 / n
        cov = (s2 - n * np.outer(mean, mean)) / (n - 1.0)
        return self.sd * (cov + self.eps * np.eye(self.d))

    def run(self, n_iterations, params, stats=None, accepted=None):
        """params: device tensor [n_iterations, n_chains, d] (the records are the moment source)."""
        done = 0
        while done < n_iterations:
            k = min(self.period - self.t % self.period, n_iterations - done)
            sl = slice(done, done + k)
            self.e.run(k, params[sl], None if stats is None else stats[sl], None if accepted is None else accepted[sl])
            self.absorb(params[sl])
            self.t += k
            done += k
            if self.t >= self.t0 and self.t % self.period == 0:
                self.C = self.covariance()
                self.e.set_proposal_covariance(self.C)
