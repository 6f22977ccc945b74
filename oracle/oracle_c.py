"""ctypes wrapper of oracle/oracle_mh.c (TEST / BASELINE INFRASTRUCTURE; see the C file's header)."""
import ctypes as C
import os

import numpy as np

_SO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "liboracle_mh.so")
_lib = None


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise RuntimeError("oracle C library not built: run __graft_entry__.build()")
        _lib = C.CDLL(_SO)
        _lib.oracle_mh_run.restype = C.c_int
        _lib.oracle_num_threads.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def run_mh(A, data, noise_var, prior_mean, prior_var, kind, C0, theta0, z, u, scaling0=1.0, adaptive=False,
           gamma=1.01, period=100, sd=None, eps=1e-6, t0=0, n_threads=0, want_records=True):
    """z [T, N, d], u [T, N] (engine layout).  Returns dict(stats [T,N,3], accepted [T,N], theta [N,d], sigma)."""
    lib = load()
    A = np.ascontiguousarray(A, dtype=np.float64)
    m, d = A.shape
    T, N = u.shape
    f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
    data, pm, pv, C0, theta0, z, u = f(data), f(prior_mean), f(prior_var), f(C0), f(theta0), f(z), f(u)
    sd = min(1.0, 2.4 ** 2 / d) if sd is None else sd
    stats = np.empty((T, N, 3)) if want_records else None
    acc = np.empty((T, N), dtype=np.uint8) if want_records else None
    th = np.empty((N, d))
    sg = np.empty((N, d, d))
    rc = lib.oracle_mh_run(C.c_int(N), C.c_int(d), C.c_int(m), C.c_int(T), _p(A), _p(data), C.c_double(noise_var),
                           _p(pm), _p(pv), C.c_int(kind), _p(C0), C.c_double(scaling0), C.c_int(int(adaptive)),
                           C.c_double(gamma), C.c_int(period), C.c_double(sd), C.c_double(eps), C.c_int(t0),
                           _p(theta0), _p(z), _p(u), _p(stats), _p(acc), _p(th), _p(sg), C.c_int(n_threads))
    if rc != 0:
        raise RuntimeError("oracle_mh_run: Cholesky failed")
    return dict(stats=stats, accepted=acc, theta=th, sigma=sg)
