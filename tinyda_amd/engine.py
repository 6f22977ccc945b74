"""Thin object wrapper over the C-ABI (include/tinyda_amd.h); numpy / torch arrays in, arrays out."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check as _check


def _ptr(a):
    """Pointer of a numpy array (host) or a torch tensor (device or host)."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return a.ctypes.data_as(C.c_void_p)
    if hasattr(a, "data_ptr"):
        assert a.is_contiguous()
        return C.c_void_p(a.data_ptr())
    raise TypeError("expected numpy array or torch tensor")


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def pinned_empty(shape, dtype=np.float64):
    """Host array in page-locked memory when a GPU is present (torch is the allocator): records written there are copied
    by a second stream while the next block computes (tda_engine_run); plain np.empty otherwise."""
    try:
        import torch

        if torch.cuda.is_available():
            tdt = {np.dtype(np.float64): torch.float64, np.dtype(np.uint8): torch.uint8}[np.dtype(dtype)]
            return torch.empty(tuple(int(v) for v in shape), dtype=tdt, pin_memory=True).numpy()
    except Exception:  # no torch / no GPU / allocation refused: pageable memory works too, only slower
        pass
    return np.empty(shape, dtype=dtype)


class ProposalSnapshot:
    """Proposal state taken over from a finished engine (tda_engine_detach_proposal_state): device buffers, read on demand."""

    def __init__(self, lib, handle):
        self.lib, self.h = lib, handle

    def read(self, n_chains, dim, want_am=False):
        N, d = n_chains, dim
        sc, Cm, cnt = np.empty(N), np.empty((N, d, d)), np.zeros(2, dtype=np.int64)
        mu = np.empty((N, d)) if want_am else None
        sg = np.empty((N, d, d)) if want_am else None
        _check(self.lib.tda_proposal_snapshot_read(self.h, _ptr(sc), _ptr(Cm), _ptr(mu), _ptr(sg), _ptr(cnt)), self.lib)
        return dict(scaling=sc, C=Cm, am_mu=mu, am_sigma=sg, t=int(cnt[0]), k=int(cnt[1]))

    def close(self):
        if getattr(self, "h", None):
            self.lib.tda_proposal_snapshot_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    """One many-chain MH engine on one GPU (one per process under torch.distributed)."""

    def __init__(self, n_chains, dim, seed=0, device=0, chain_offset=0, n_levels=1, block_steps=0, stream=None, lib=None):
        # lib: another library exporting the same C-ABI (_lib.load_from): tests and the CPU baseline drive the CPU twin of
        # the ABI through this very class; the package itself never passes it
        self.lib = _lib.load() if lib is None else lib
        self.n_chains, self.dim, self.n_levels = int(n_chains), int(dim), int(n_levels)
        self.device = int(device)
        self.subchain_lengths = []
        cfg = _lib.tda_config(C.sizeof(_lib.tda_config), device, n_chains, chain_offset, dim, n_levels, seed,
                              stream, block_steps, 0)
        h = C.c_void_p()
        self._ck(self.lib.tda_engine_create(C.byref(cfg), C.byref(h)))
        self.h = h
        self._keep = []

    def _ck(self, rc):
        _check(rc, self.lib)

    def close(self):
        if getattr(self, "h", None):
            self.lib.tda_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- problem ------------------------------------------------------------------------
    def set_prior(self, mean, cov):
        mean, cov = _f64(mean), _f64(cov)
        assert mean.shape == (self.dim,) and cov.shape == (self.dim, self.dim)
        self._ck(self.lib.tda_engine_set_prior(self.h, _ptr(mean), _ptr(cov)))

    def set_level(self, level, A, data, noise_kind, noise, b=None):
        A, data, noise = _f64(A), _f64(data), _f64(np.atleast_1d(noise))
        m = A.shape[0]
        assert A.shape == (m, self.dim) and data.shape == (m,)
        b = None if b is None else _f64(b)
        self._ck(self.lib.tda_engine_set_level(self.h, level, m, _ptr(A), _ptr(b), _ptr(data), noise_kind, _ptr(noise)))

    def set_proposal(self, kind, C_=None, scaling=1.0, adaptive=False, gamma=1.01, period=100, sd=None,
                     epsilon=1e-6, t0=0, block_moments=False, q_mean=None, state_operator=None, noise_operator=None, spectrum=None):
        Cm = None if C_ is None else _f64(C_)
        p = _lib.tda_proposal_params(C.sizeof(_lib.tda_proposal_params), kind, scaling, int(adaptive), period, gamma,
                                     _ptr(Cm), -1.0 if sd is None else sd, epsilon, t0, int(block_moments),
                                     _ptr(None if q_mean is None else _f64(q_mean)))
        self._ck(self.lib.tda_engine_set_proposal(self.h, C.byref(p)))
        if state_operator is not None:  # OperatorWeightedCrankNicolson (kind 5): theta' = S theta + N chol(C_prior) z
            So, No = _f64(state_operator), _f64(noise_operator)
            assert So.shape == (self.dim, self.dim) and No.shape == (self.dim, self.dim)
            self._ck(self.lib.tda_engine_set_proposal_operators(self.h, _ptr(So), _ptr(No)))
        if spectrum is not None:  # OperatorWeightedCrankNicolson with per-chain operators: B = V diag(lam) V^T, V's columns the eigenvectors
            V, lam = _f64(spectrum[0]), _f64(spectrum[1])
            assert V.shape == (self.dim, self.dim) and lam.shape == (self.dim,)
            self._ck(self.lib.tda_engine_set_proposal_spectrum(self.h, _ptr(V), _ptr(lam)))

    def set_prior_joint(self, kinds, loc, scale):
        """JointPrior of scalar components: kinds[j] 0 = norm(loc, scale), 1 = uniform(loc, scale)"""
        kinds = np.ascontiguousarray(np.asarray(kinds, dtype=np.int32))
        loc, scale = _f64(loc), _f64(scale)
        assert kinds.shape == loc.shape == scale.shape == (self.dim,)
        self._ck(self.lib.tda_engine_set_prior_joint(self.h, _ptr(kinds), _ptr(loc), _ptr(scale)))

    def set_level_source(self, level, source, data, noise_kind, noise):
        """forward model as HIP source defining `__device__ double tda_forward(const double* theta, int dim, int o)`"""
        data = _f64(np.atleast_1d(data))
        noise = _f64(np.atleast_1d(noise))
        self._ck(self.lib.tda_engine_set_level_source(self.h, level, source.encode(), data.size, _ptr(data), noise_kind, _ptr(noise)))

    def set_level_callback(self, level, fn, data, noise_kind, noise, inplace=False):
        """forward model behind a batched host callback: fn maps the (n_chains, dim) proposals of a step to the
        (n_chains, m) model outputs; everything else of the step stays on the device.  inplace: fn(thetas, out) writes
        the outputs into the engine's page-locked buffer itself (saves an allocation and a copy per step)"""
        data = _f64(np.atleast_1d(data))
        noise = _f64(np.atleast_1d(noise))
        m = data.size
        self._cb_exc = None

        def trampoline(_user, theta_p, F_p, n, dim, m_):
            try:
                theta = np.ctypeslib.as_array(theta_p, shape=(n, dim))
                F = np.ctypeslib.as_array(F_p, shape=(n, m_))
                if inplace:
                    fn(theta, F)
                    return 0
                out = np.asarray(fn(theta), dtype=np.float64)
                if out.shape != (n, m_):
                    raise ValueError("the batched model returned shape %s, expected %s" % (out.shape, (n, m_)))
                F[...] = out
                return 0
            except BaseException as exc:  # an exception must not unwind through the C frames
                self._cb_exc = exc
                return 1

        if not hasattr(self, "_cbs"):
            self._cbs = {}
        self._cbs[level] = _lib.FORWARD_BATCH_FN(trampoline)  # keep the thunks alive as long as the engine
        self._ck(self.lib.tda_engine_set_level_callback(self.h, level, C.cast(self._cbs[level], C.c_void_p), None, m, _ptr(data),
                                                     noise_kind, _ptr(noise)))

    def _check_run(self, rc):
        exc = getattr(self, "_cb_exc", None)
        if rc == _lib.TDA_ERR_CALLBACK and exc is not None:
            self._cb_exc = None
            raise exc
        self._ck(rc)

    def set_level_rosenbrock(self, level, a=1.0, b=10.0, data=0.0, noise_var=1.0):
        self._ck(self.lib.tda_engine_set_level_rosenbrock(self.h, level, a, b, data, noise_var))

    def set_proposal_dreamz(self, M0, delta=1, b=5e-2, b_star=1e-6, nCR=3, adaptive=False, gamma=1.01, period=100,
                            shared=False, sync_every=0, capacity=None):
        cap = int(capacity if capacity is not None else M0)
        p = _lib.tda_dreamz_params(C.sizeof(_lib.tda_dreamz_params), M0, delta, nCR, b, b_star, int(adaptive), period, gamma,
                                   int(shared), sync_every, cap)
        self._dz = dict(M0=M0, delta=delta, nCR=nCR, shared=bool(shared), sync_every=int(sync_every), adaptive=bool(adaptive),
                        period=int(period), capacity=cap)
        self._ck(self.lib.tda_engine_set_proposal_dreamz(self.h, C.byref(p)))

    def set_archive(self, Z0=None):
        if Z0 is not None:
            Z0 = _f64(Z0)
            want = (self._dz["M0"], self.dim) if self._dz["shared"] else (self.n_chains, self._dz["M0"], self.dim)
            assert Z0.shape == want, (Z0.shape, want)
        self._ck(self.lib.tda_engine_set_archive(self.h, _ptr(Z0)))

    def set_replay_dreamz(self, r, mcr, sub_u, forced, e_u, eps_n, u):
        """all arrays step-major: r [T,N,delta,2], mcr/forced/u [T,N], sub_u/e_u/eps_n [T,N,d]"""
        i32 = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.int32))
        r, mcr, forced = i32(r), i32(mcr), i32(forced)
        sub_u, e_u, eps_n, u = _f64(sub_u), _f64(e_u), _f64(eps_n), _f64(u)
        T = u.shape[0]
        assert r.shape == (T, self.n_chains, self._dz["delta"], 2) and sub_u.shape == (T, self.n_chains, self.dim)
        self._ck(self.lib.tda_engine_set_replay_dreamz(self.h, _ptr(r), _ptr(mcr), _ptr(sub_u), _ptr(forced), _ptr(e_u),
                                                    _ptr(eps_n), _ptr(u), T))

    def dreamz_state(self):
        pcr = np.empty((self.n_chains, self._dz["nCR"]))
        rows = np.zeros(1, dtype=np.int64)
        self._ck(self.lib.tda_engine_get_dreamz_state(self.h, _ptr(pcr), _ptr(rows)))
        return dict(pCR=pcr, archive_rows=int(rows[0]))

    def set_archive_auto_append(self, on):
        self._ck(self.lib.tda_engine_set_archive_auto_append(self.h, int(on)))

    # -- distributed shared archive (every rank keeps its own rows; include/tinyda_amd.h) ------------
    def archive_ipc_handle(self):
        h = (C.c_ubyte * 64)()
        self._ck(self.lib.tda_engine_archive_ipc_handle(self.h, C.byref(h)))
        return bytes(h)

    def archive_pointer(self):
        p = C.c_void_p()
        self._ck(self.lib.tda_engine_archive_pointer(self.h, C.byref(p)))
        return p.value

    def set_archive_peers(self, n_ranks, my_rank, handles=None, pointers=None):
        """handles: list of n_ranks 64-byte IPC handles (other processes) or pointers: list of n_ranks device addresses (engines
        of this process); the entry of my_rank is ignored"""
        hb = pb = None
        if handles is not None:
            assert len(handles) == n_ranks and all(len(x) == 64 for x in handles)
            hb = (C.c_ubyte * (64 * n_ranks)).from_buffer_copy(b"".join(handles))
        if pointers is not None:
            assert len(pointers) == n_ranks
            pb = (C.c_void_p * n_ranks)(*[C.c_void_p(int(x)) for x in pointers])
        self._ck(self.lib.tda_engine_set_archive_peers(self.h, n_ranks, my_rank, hb, pb))
        self._dz["dist_ranks"] = n_ranks

    def archive_local_sums(self):
        out = np.zeros((2, self.dim))
        self._ck(self.lib.tda_engine_archive_local_sums(self.h, _ptr(out)))
        return out

    def archive_publish(self, sums_total=None):
        if sums_total is not None:
            sums_total = _f64(sums_total)
            assert sums_total.shape == (2, self.dim)
        self._ck(self.lib.tda_engine_archive_publish(self.h, _ptr(sums_total)))

    def archive_take(self, rows=None):
        """shared archive: number of pending steps; if `rows` ([steps, chains, dim] array / tensor) is given it is filled"""
        n = np.zeros(1, dtype=np.int64)
        self._ck(self.lib.tda_engine_archive_take(self.h, _ptr(rows), _ptr(n)))
        return int(n[0])

    def archive_append(self, rows):
        """rows: [n_rows, dim] numpy array or torch tensor (device ok)"""
        n = rows.shape[0]
        self._ck(self.lib.tda_engine_archive_append(self.h, _ptr(rows), n))

    def set_subchains(self, lengths, randomize=False):
        arr = np.ascontiguousarray(np.asarray(lengths, dtype=np.int32))
        assert arr.shape == (self.n_levels - 1,)
        self.subchain_lengths = [int(x) for x in arr]
        self._ck(self.lib.tda_engine_set_subchains(self.h, _ptr(arr), int(randomize)))

    def reduce_moments(self, rows, out=None):
        """[count, sum x, sum x x^T] over a device record buffer [..., dim]; `out` torch tensor (device) or None -> numpy"""
        n = 1
        for sdim in rows.shape[:-1]:
            n *= int(sdim)
        if out is None:
            out = np.empty(1 + self.dim + self.dim * self.dim)
        self._ck(self.lib.tda_engine_reduce_moments(self.h, _ptr(rows), n, _ptr(out)))
        return out

    def set_proposal_covariance(self, Cm):
        Cm = _f64(Cm)
        assert Cm.shape == (self.dim, self.dim)
        self._ck(self.lib.tda_engine_set_proposal_covariance(self.h, _ptr(Cm)))

    def get_state(self):
        """checkpoint: opaque bytes (numpy uint8) holding chain, proposal and counter state"""
        n = int(self.lib.tda_engine_state_size(self.h))
        if n < 0:
            self._ck(n)
        blob = np.empty(n, dtype=np.uint8)
        self._ck(self.lib.tda_engine_get_state(self.h, _ptr(blob), n))
        return blob

    def set_state(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        self._ck(self.lib.tda_engine_set_state(self.h, _ptr(blob), blob.size))
        self._t_py = None  # the step counter came out of the blob: ask the engine once

    def counters(self):
        """(t, k): adapt() calls so far = steps taken (proposal.py:228, :509) and the diminishing-adaptation counter, from the engine"""
        cnt = np.zeros(2, dtype=np.int64)
        self._ck(self.lib.tda_engine_get_proposal_state(self.h, None, None, None, None, _ptr(cnt)))
        return int(cnt[0]), int(cnt[1])

    def set_error_model(self, kind):
        code = {None: 0, "state-independent": 1, "state-dependent": 2, "state-independent-diagonal": 3}[kind]
        self._ck(self.lib.tda_engine_set_error_model(self.h, code))

    def error_model_state(self, level, m, covariance=True):
        """bias [chains, m] and (Sigma_e + Sigma_bias)^-1 [chains, m, m] of one level (covariance=False: bias only, None)"""
        bias = np.empty((self.n_chains, m))
        P = np.empty((self.n_chains, m, m)) if covariance else None
        self._ck(self.lib.tda_engine_get_error_model(self.h, level, _ptr(bias), _ptr(P)))
        return bias, P

    def init(self, theta0=None):
        if theta0 is not None and isinstance(theta0, np.ndarray):
            theta0 = _f64(theta0)
            assert theta0.shape == (self.n_chains, self.dim)
        self._check_run(self.lib.tda_engine_init(self.h, _ptr(theta0)))
        self._t_py = 0  # single-level step counter kept on this side: run(sync=False) must not ask the engine (that synchronises)

    # -- variates -----------------------------------------------------------------------
    def set_replay(self, z, u):
        """z [steps, chains, dim], u [steps, chains]"""
        if z is None:
            self._ck(self.lib.tda_engine_set_replay(self.h, None, None, 0))
            return
        z, u = _f64(z), _f64(u)
        assert z.shape[1:] == (self.n_chains, self.dim) and u.shape == z.shape[:2]
        self._ck(self.lib.tda_engine_set_replay(self.h, _ptr(z), _ptr(u), z.shape[0]))

    def set_replay_level(self, level, u):
        """uniforms of level >= 1 as [steps of that level, chains]; level = -1: DA promoted index in [-L, -1]"""
        if u is None:
            self._ck(self.lib.tda_engine_set_replay_level(self.h, level, None, 0))
            return
        u = _f64(u)
        assert u.shape[1] == self.n_chains
        self._ck(self.lib.tda_engine_set_replay_level(self.h, level, _ptr(u), u.shape[0]))

    def set_export(self, n_steps):
        z = np.zeros((n_steps, self.n_chains, self.dim))
        u = np.zeros((n_steps, self.n_chains))
        self._keep = [z, u]
        self._ck(self.lib.tda_engine_set_export(self.h, _ptr(z), _ptr(u), n_steps))
        return z, u

    # -- running ------------------------------------------------------------------------
    def _outputs(self, level, need, params, stats, accepted):
        """tda_outputs for one level after checking every record buffer: dtype, trailing shape and at least `need` rows
        (the library checks `rows` again and the extent of device allocations; a short buffer is an error, not a fault)"""
        want = (("params", params, (self.n_chains, self.dim), "float64"), ("stats", stats, (self.n_chains, 3), "float64"),
                ("accepted", accepted, (self.n_chains,), "uint8"))
        rows = None
        for name, a, tail, dt in want:
            if a is None:
                continue
            shape = tuple(int(v) for v in a.shape)
            if str(a.dtype).replace("torch.", "") != dt:
                raise ValueError("%s buffer of level %d must be %s, got %s" % (name, level, dt, a.dtype))
            if len(shape) != len(tail) + 1 or shape[1:] != tail:
                raise ValueError("%s buffer of level %d must have shape (rows, %s), got %s" % (name, level, ", ".join(map(str, tail)), shape))
            if shape[0] < need:
                raise ValueError("%s buffer of level %d holds %d records, this run produces %d" % (name, level, shape[0], need))
            rows = shape[0] if rows is None else min(rows, shape[0])
        return _lib.tda_outputs(C.sizeof(_lib.tda_outputs), min(rows or 0, 0xFFFFFFFF), _ptr(params), _ptr(stats), _ptr(accepted))

    def run(self, n_iterations, params=None, stats=None, accepted=None, sync=True):
        need = int(n_iterations)
        thin = getattr(self, "_thin", 1)
        if thin > 1 and not (params is None and stats is None and accepted is None):
            if getattr(self, "_t_py", None) is None:
                self._t_py = self.counters()[0]
            t = self._t_py
            need = (t + need) // thin - t // thin  # records that reach the caller (include/tinyda_amd.h, record thinning)
        out = self._outputs(0, need, params, stats, accepted)
        self._check_run(self.lib.tda_engine_run(self.h, n_iterations, C.byref(out)))
        if getattr(self, "_t_py", None) is not None:
            self._t_py += int(n_iterations)
        if sync:
            self.sync()

    def rows_per_level(self, n_iterations):
        """records each level produces for n_iterations finest-level steps (include/tinyda_amd.h, tda_outputs)"""
        rows = [n_iterations]
        for L in reversed(self.subchain_lengths):
            rows.insert(0, rows[0] * L)
        return rows

    def run_levels(self, n_iterations, outputs=None, sync=True):
        """Multi-level run.  outputs: list (coarsest first) of (params, stats, accepted) arrays / tensors or None."""
        arr = (_lib.tda_outputs * self.n_levels)()
        need = self.rows_per_level(int(n_iterations))
        for k in range(self.n_levels):
            p, s_, a = outputs[k] if outputs and outputs[k] is not None else (None, None, None)
            arr[k] = self._outputs(k, need[k], p, s_, a)
        self._check_run(self.lib.tda_engine_run(self.h, n_iterations, arr))
        if sync:
            self.sync()

    def run_levels_host(self, n_iterations):
        N, d = self.n_chains, self.dim
        outs = [(pinned_empty((r, N, d)), pinned_empty((r, N, 3)), pinned_empty((r, N), dtype=np.uint8))
                for r in self.rows_per_level(n_iterations)]
        self.run_levels(n_iterations, outs)
        return outs

    def level_state(self, level):
        th, st = np.empty((self.n_chains, self.dim)), np.empty((self.n_chains, 3))
        self._ck(self.lib.tda_engine_get_level_state(self.h, level, _ptr(th), _ptr(st)))
        return th, st

    def run_host(self, n_iterations):
        """Convenience: run and return numpy records (params [T,N,d], stats [T,N,3], accepted [T,N])."""
        T, N, d = n_iterations, self.n_chains, self.dim
        params, stats = pinned_empty((T, N, d)), pinned_empty((T, N, 3))
        acc = pinned_empty((T, N), dtype=np.uint8)
        self.run(T, params, stats, acc)
        return params, stats, acc

    def sync(self):
        self._ck(self.lib.tda_engine_sync(self.h))

    def set_record_thinning(self, thin):
        """only iterations with (t + 1) % thin == 0 reach the record buffers (a run of n from t = 0 gives n // thin records)"""
        self._ck(self.lib.tda_engine_set_record_thinning(self.h, int(thin)))
        self._thin = int(thin)

    def set_progress(self, on=True):
        self._ck(self.lib.tda_engine_set_progress(self.h, int(on)))

    def progress(self):
        """(iterations completed on the device, iterations queued, mean accept flag of the last completed block or -1): a read of
        page-locked memory, no synchronisation -- callable while run(sync=False) work is in flight"""
        done, queued, rate = C.c_int64(0), C.c_int64(0), C.c_double(-1.0)
        self._ck(self.lib.tda_engine_get_progress(self.h, C.byref(done), C.byref(queued), C.byref(rate)))
        return done.value, queued.value, rate.value

    def current_into(self, theta, stats):
        """current states into caller arrays / DEVICE tensors ([chains, dim], [chains, 3])"""
        self._ck(self.lib.tda_engine_get_current(self.h, _ptr(theta), _ptr(stats)))

    def level_state_into(self, level, theta, stats):
        self._ck(self.lib.tda_engine_get_level_state(self.h, level, _ptr(theta), _ptr(stats)))

    def detach_proposal_state(self):
        """hand the proposal buffers to a ProposalSnapshot (no copy); the engine can only be closed afterwards"""
        h = C.c_void_p()
        self._ck(self.lib.tda_engine_detach_proposal_state(self.h, C.byref(h)))
        return ProposalSnapshot(self.lib, h)

    def current(self):
        th, st = np.empty((self.n_chains, self.dim)), np.empty((self.n_chains, 3))
        self._ck(self.lib.tda_engine_get_current(self.h, _ptr(th), _ptr(st)))
        return th, st

    def proposal_state(self, want_am=False):
        N, d = self.n_chains, self.dim
        sc, Cm, cnt = np.empty(N), np.empty((N, d, d)), np.zeros(2, dtype=np.int64)
        mu = np.empty((N, d)) if want_am else None
        sg = np.empty((N, d, d)) if want_am else None
        self._ck(self.lib.tda_engine_get_proposal_state(self.h, _ptr(sc), _ptr(Cm), _ptr(mu), _ptr(sg), _ptr(cnt)))
        return dict(scaling=sc, C=Cm, am_mu=mu, am_sigma=sg, t=int(cnt[0]), k=int(cnt[1]))

    def proposal_state_scaling(self):
        sc = np.empty(self.n_chains)
        self._ck(self.lib.tda_engine_get_proposal_state(self.h, _ptr(sc), None, None, None, None))
        return sc

    def flags(self):
        f = np.zeros(self.n_chains, dtype=np.int32)
        self._ck(self.lib.tda_engine_get_flags(self.h, _ptr(f)))
        return f

    def evaluate(self, theta, level=0):
        theta = _f64(theta)
        n = theta.shape[0]
        st = np.empty((n, 3))
        self._ck(self.lib.tda_engine_evaluate(self.h, level, _ptr(theta), n, _ptr(st)))
        return st

    def rng_probe(self, step):
        z, u = np.empty((self.n_chains, self.dim)), np.empty(self.n_chains)
        self._ck(self.lib.tda_engine_rng_probe(self.h, step, _ptr(z), _ptr(u)))
        return z, u

    def set_profiling(self, on):
        self._ck(self.lib.tda_engine_set_profiling(self.h, int(on)))

    def profile(self):
        p = _lib.tda_profile(C.sizeof(_lib.tda_profile))
        self._ck(self.lib.tda_engine_get_profile(self.h, C.byref(p)))
        return {k: getattr(p, k) for k, _ in p._fields_ if k != "struct_size"}
