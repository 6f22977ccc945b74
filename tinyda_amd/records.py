"""Sample records: `Link` (one MCMC sample, attribute names of tinyDA/link.py:23-48), `DeviceRecords` (the record arrays of one
level of one run, resident in HBM) and `DeviceChain`, the array-backed sequence the device path returns instead of a Python list
of Links (sampler.py:305-309).

Nothing crosses PCIe when sample() returns: a chain's rows are copied to the host when that chain (or get_samples) asks for
them, the whole history of all chains in one pass chunked through a small reusable page-locked buffer."""
import threading
from collections.abc import Sequence

import numpy as np

_FIELDS = ("parameters", "prior", "model_output", "likelihood", "qoi")


class Link:
    """parameters, log-prior, model output, log-likelihood, optional quantity of interest; `posterior` is the sum of a
    *normalised* log-prior and an *unnormalised* Gaussian log-likelihood, exactly as the reference combines them."""

    __slots__ = _FIELDS + ("posterior", "gradient")  # `gradient` is attached lazily by MALA (proposal.py:948-949)

    def __init__(self, parameters, prior, model_output, likelihood, qoi=None):
        for name, value in zip(_FIELDS, (parameters, prior, model_output, likelihood, qoi)):
            object.__setattr__(self, name, value)
        object.__setattr__(self, "posterior", prior + likelihood)

    def __repr__(self):
        return "Link(posterior=%r, parameters=%r)" % (self.posterior, self.parameters)


_STAGE_BYTES = 64 << 20  # per page-locked staging buffer (two per process and device)
_stage = {}
_stage_lock = threading.Lock()
_transfer_locks = {}  # device -> lock held for a whole all_chains_host pass (the two staging slots are per process and device)
_COPY_THREADS = 8
_copy_pool = None


def _parallel_copy(dst, src):
    """dst[...] = src for large host arrays, split over a few threads along the first axis: NumPy releases the GIL while it
    copies, and the destination is freshly allocated pageable memory whose first touch (one page fault per 4 KiB) is what a
    single thread spends most of its time on (measured: 2 GB/s with one thread)."""
    global _copy_pool
    n = dst.shape[0]
    if dst.nbytes < (8 << 20) or n < 2:
        dst[...] = src
        return
    if _copy_pool is None:
        from concurrent.futures import ThreadPoolExecutor

        _copy_pool = ThreadPoolExecutor(max_workers=_COPY_THREADS, thread_name_prefix="tda-copy")
    k = min(_COPY_THREADS, n)
    bounds = [n * i // k for i in range(k + 1)]

    def part(i):
        dst[bounds[i]:bounds[i + 1]] = src[bounds[i]:bounds[i + 1]]

    list(_copy_pool.map(part, range(k)))


def _staging(device):
    """the process's two page-locked staging buffers for `device` (allocated once: page-locking is the slow part)"""
    import torch

    with _stage_lock:
        if device not in _stage:
            _stage[device] = [torch.empty(_STAGE_BYTES, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
        return _stage[device]


def _transfer_lock(device):
    with _stage_lock:
        return _transfer_locks.setdefault(device, threading.Lock())


def _is_torch(a):
    return hasattr(a, "data_ptr") and hasattr(a, "device")


class DeviceRecords:
    """Record arrays of one level of one run, as the engine wrote them: parameters [R, N, d], stats [R, N, 3] (log-prior,
    log-likelihood, log-posterior), accepted [R, N] -- torch tensors in device memory (or NumPy arrays: the same interface over
    host records).  Row 0 of the finest level is the initial link (chain.py:70-71)."""

    FIELDS = ("parameters", "stats", "accepted")

    def __init__(self, parameters, stats, accepted):
        self.parameters, self.stats, self.accepted = parameters, stats, accepted
        self.on_device = _is_torch(parameters) and parameters.device.type != "cpu"

    @property
    def n_rows(self):
        return int(self.parameters.shape[0])

    @property
    def n_chains(self):
        return int(self.parameters.shape[1])

    def to_host(self):
        """move the record arrays to host memory and let go of the HBM (a C2-size result holds 4.4 GB there, the coarse levels of
        a long MLDA run far more; the lazy views keep working, from host arrays).  Returns self."""
        if self.on_device:
            for f in self.FIELDS:
                setattr(self, f, getattr(self, f).cpu())
            self.on_device = False
        return self

    def chain_host(self, field, chain, rows=slice(None)):
        """rows of ONE chain as a NumPy array ([R', d], [R', 3] or [R']); one strided device gather + one copy"""
        a = getattr(self, field)
        if not _is_torch(a):
            return np.asarray(a[rows, chain])
        return a[rows, chain].contiguous().cpu().numpy()

    def all_chains_host(self, field, start=0):
        """rows [start:] of EVERY chain, chain-major: NumPy [N, R - start, width].  Device records are transposed on the GPU
        a group of chains at a time and cross PCIe through two page-locked staging buffers (the copy of group g + 1 runs under
        the host-side unpacking of group g); the result is pageable memory owned by the caller."""
        a = getattr(self, field)
        if a.ndim == 2:
            a = a[:, :, None]
        R, N, w = int(a.shape[0]) - int(start), int(a.shape[1]), int(a.shape[2])
        if not _is_torch(a):
            return np.ascontiguousarray(np.swapaxes(np.asarray(a[start:]), 0, 1))
        out = np.empty((N, max(R, 0), w), dtype={8: np.float64, 1: np.uint8}[a.element_size()])
        if R <= 0 or N == 0:
            return out
        if not self.on_device:
            out[...] = a[start:].permute(1, 0, 2).numpy()
            return out
        import torch

        per_chain = R * w * a.element_size()
        group = max(1, _STAGE_BYTES // per_chain)
        if per_chain > _STAGE_BYTES:  # one chain's history exceeds a staging buffer: torch's own pageable path
            for c in range(N):
                out[c] = a[start:, c].contiguous().cpu().numpy()
            return out
        stage = _staging(a.device.index or 0)
        events = [torch.cuda.Event(), torch.cuda.Event()]
        # two threads reading results at once would interleave their copies in the same two slots: one pass at a time per device
        with _transfer_lock(a.device.index or 0), torch.cuda.device(a.device):
            pending = None
            for g, c0 in enumerate(range(0, N, group)):
                c1 = min(N, c0 + group)
                slot = g & 1
                blockdev = a[start:, c0:c1].permute(1, 0, 2).contiguous()  # [chains, rows, width] on the device
                view = stage[slot][:blockdev.numel() * a.element_size()].view(a.dtype).view(blockdev.shape)
                view.copy_(blockdev, non_blocking=True)
                events[slot].record()
                if pending is not None:
                    ps, p0, p1, pview = pending
                    events[ps].synchronize()
                    _parallel_copy(out[p0:p1], pview.numpy())
                pending = (slot, c0, c1, view)
            ps, p0, p1, pview = pending
            events[ps].synchronize()
            _parallel_copy(out[p0:p1], pview.numpy())
        return out


class DeviceChain(Sequence):
    """tinyDA returns `chain_i` as a list of Link objects (sampler.py:305-309).  With thousands of chains that is millions of
    objects, so the device path returns this read-only view of one chain of the engine's record arrays; `parameters`, `stats`
    and `accepted` are fetched from the device the first time they are touched (and kept), a Link -- including its model output --
    is materialised only when indexed.  `get_samples` reads all chains of a result in one pass (DeviceRecords.all_chains_host).

    DeviceChain(records, chain_index, model) over DeviceRecords, or DeviceChain(parameters, stats, accepted, model) over host
    arrays of one chain.

    A result holds one of these per chain and level (12 288 at BASELINE config 5): no instance dictionary, no cache object until
    something is fetched -- the interpreter's cyclic collector walks every tracked object, and a result three times the size
    pushed a full collection (70 ms) into every third `sample()` call."""

    __slots__ = ("_records", "_chain", "_rows", "_model", "_cache")

    def __init__(self, parameters, stats=None, accepted=None, model=None, rows=slice(None)):
        if isinstance(parameters, DeviceRecords):
            self._records, self._chain, self._rows = parameters, int(stats), rows
            self._model = accepted if model is None else model
            self._cache = None
        else:
            self._records = None
            self._rows = self._chain = None
            self._cache = {"parameters": parameters, "stats": stats, "accepted": accepted}
            self._model = model

    def _get(self, field):
        if self._cache is None:
            self._cache = {}
        if field not in self._cache:
            self._cache[field] = self._records.chain_host(field, self._chain, self._rows)
        return self._cache[field]

    @property
    def parameters(self):  # [T+1, d]
        return self._get("parameters")

    @property
    def stats(self):  # [T+1, 3] log-prior, log-likelihood, log-posterior
        return self._get("stats")

    @property
    def accepted(self):  # [T+1] (entry 0 is the initial link, True as in chain.py:71)
        return self._get("accepted")

    def __len__(self):
        if self._records is not None and (self._cache is None or "parameters" not in self._cache):
            return len(range(*self._rows.indices(self._records.n_rows)))
        return self.parameters.shape[0]

    def __getitem__(self, i):
        if isinstance(i, slice):
            if self._records is not None and not self._cache:
                r = range(*self._rows.indices(self._records.n_rows))[i]
                if r.step > 0:  # still nothing fetched: compose the row ranges
                    return DeviceChain(self._records, self._chain, self._model, rows=slice(r.start, r.stop, r.step))
            return DeviceChain(self.parameters[i], self.stats[i], self.accepted[i], self._model)
        theta = np.array(self.parameters[i])
        out = self._model(theta) if self._model is not None else None
        out, qoi = out if isinstance(out, tuple) else (out, None)  # posterior.py:97-101
        return Link(theta, float(self.stats[i, 0]), out, float(self.stats[i, 1]), qoi)
