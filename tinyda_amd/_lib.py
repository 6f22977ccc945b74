"""ctypes binding of libtinyda_hip.so (C-ABI: include/tinyda_amd.h).

There is no CPU compute fallback behind this module: if the shared library is missing or cannot be
loaded, `load()` raises and every device code path of the package raises with it.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libtinyda_hip.so")

TDA_OK = 0
TDA_ERR_INVALID, TDA_ERR_HIP, TDA_ERR_STATE, TDA_ERR_UNSUPPORTED, TDA_ERR_NUMERIC, TDA_ERR_CALLBACK = -1, -2, -3, -4, -5, -6
NOISE_ISO, NOISE_DIAG, NOISE_DENSE, NOISE_ADAPTIVE = 0, 1, 2, 3
AEM_NONE, AEM_STATE_INDEPENDENT, AEM_STATE_DEPENDENT, AEM_STATE_INDEPENDENT_DIAGONAL = 0, 1, 2, 3
PROP_GRW, PROP_PCN, PROP_AM, PROP_DREAMZ, PROP_INDEPENDENCE, PROP_OWCN, PROP_MALA = 0, 1, 2, 3, 4, 5, 6


class EngineError(RuntimeError):
    pass


class tda_config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("device", C.c_int32),
        ("n_chains", C.c_int64),
        ("chain_offset", C.c_int64),
        ("dim", C.c_int32),
        ("n_levels", C.c_int32),
        ("seed", C.c_uint64),
        ("stream", C.c_void_p),
        ("block_steps", C.c_int32),
        ("reserved", C.c_int32),
    ]


class tda_proposal_params(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("kind", C.c_int32),
        ("scaling", C.c_double),
        ("adaptive", C.c_int32),
        ("period", C.c_int32),
        ("gamma", C.c_double),
        ("C", C.c_void_p),
        ("sd", C.c_double),
        ("epsilon", C.c_double),
        ("t0", C.c_int32),
        ("block_moments", C.c_int32),
        ("q_mean", C.c_void_p),
    ]


class tda_dreamz_params(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("M0", C.c_int32),
        ("delta", C.c_int32),
        ("nCR", C.c_int32),
        ("b", C.c_double),
        ("b_star", C.c_double),
        ("adaptive", C.c_int32),
        ("period", C.c_int32),
        ("gamma", C.c_double),
        ("shared", C.c_int32),
        ("sync_every", C.c_int32),
        ("capacity", C.c_int64),
    ]


class tda_outputs(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("rows", C.c_uint32),
        ("params", C.c_void_p),
        ("stats", C.c_void_p),
        ("accepted", C.c_void_p),
    ]


class tda_profile(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("n_launch_propose", C.c_uint32),
        ("n_launch_steps", C.c_uint32),
        ("n_launch_adapt", C.c_uint32),
        ("ms_propose", C.c_double),
        ("ms_steps", C.c_double),
        ("ms_adapt", C.c_double),
        ("ms_total", C.c_double),
        ("n_launch_aem", C.c_uint32),
        ("reserved0", C.c_uint32),
        ("ms_aem", C.c_double),
    ]


# every symbol include/tinyda_amd.h declares: name -> (restype, argtypes)
_P = C.c_void_p
# int (*tda_forward_batch_fn)(void* user, const double* theta, double* F, int64_t n_chains, int32_t dim, int32_t m)
FORWARD_BATCH_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64, C.c_int32, C.c_int32)
SYMBOLS = {
    "tda_last_error": (C.c_char_p, []),
    "tda_version": (C.c_char_p, []),
    "tda_release_cached_memory": (C.c_int64, []),
    "tda_engine_create": (C.c_int, [C.POINTER(tda_config), C.POINTER(_P)]),
    "tda_engine_destroy": (None, [_P]),
    "tda_engine_set_prior": (C.c_int, [_P, _P, _P]),
    "tda_engine_set_level": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, C.c_int, _P]),
    "tda_engine_set_proposal": (C.c_int, [_P, C.POINTER(tda_proposal_params)]),
    "tda_engine_set_proposal_dreamz": (C.c_int, [_P, C.POINTER(tda_dreamz_params)]),
    "tda_engine_set_proposal_operators": (C.c_int, [_P, _P, _P]),
    "tda_engine_set_proposal_spectrum": (C.c_int, [_P, _P, _P]),
    "tda_engine_set_archive": (C.c_int, [_P, _P]),
    "tda_engine_set_level_rosenbrock": (C.c_int, [_P, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double]),
    "tda_engine_set_replay_dreamz": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, C.c_int64]),
    "tda_engine_get_dreamz_state": (C.c_int, [_P, _P, _P]),
    "tda_engine_archive_take": (C.c_int, [_P, _P, _P]),
    "tda_engine_archive_append": (C.c_int, [_P, _P, C.c_int64]),
    "tda_engine_set_archive_auto_append": (C.c_int, [_P, C.c_int]),
    "tda_engine_archive_ipc_handle": (C.c_int, [_P, _P]),
    "tda_engine_archive_pointer": (C.c_int, [_P, _P]),
    "tda_engine_set_archive_peers": (C.c_int, [_P, C.c_int, C.c_int, _P, _P]),
    "tda_engine_archive_local_sums": (C.c_int, [_P, _P]),
    "tda_engine_archive_publish": (C.c_int, [_P, _P]),
    "tda_engine_reduce_moments": (C.c_int, [_P, _P, C.c_int64, _P]),
    "tda_engine_set_proposal_covariance": (C.c_int, [_P, _P]),
    "tda_engine_set_prior_joint": (C.c_int, [_P, _P, _P, _P]),
    "tda_engine_set_level_source": (C.c_int, [_P, C.c_int, C.c_char_p, C.c_int32, _P, C.c_int32, _P]),
    "tda_engine_set_level_callback": (C.c_int, [_P, C.c_int, _P, _P, C.c_int32, _P, C.c_int32, _P]),
    "tda_diag_ess_rhat": (C.c_int, [C.c_int, _P, _P, C.c_int64, C.c_int64, C.c_int32, C.c_int64, _P, _P]),
    "tda_engine_state_size": (C.c_int64, [_P]),
    "tda_engine_get_state": (C.c_int, [_P, _P, C.c_int64]),
    "tda_engine_set_state": (C.c_int, [_P, _P, C.c_int64]),
    "tda_engine_set_subchains": (C.c_int, [_P, _P, C.c_int]),
    "tda_engine_set_error_model": (C.c_int, [_P, C.c_int]),
    "tda_engine_get_error_model": (C.c_int, [_P, C.c_int, _P, _P]),
    "tda_engine_set_replay_level": (C.c_int, [_P, C.c_int, _P, C.c_int64]),
    "tda_engine_get_level_state": (C.c_int, [_P, C.c_int, _P, _P]),
    "tda_engine_init": (C.c_int, [_P, _P]),
    "tda_engine_get_current": (C.c_int, [_P, _P, _P]),
    "tda_engine_set_replay": (C.c_int, [_P, _P, _P, C.c_int64]),
    "tda_engine_set_export": (C.c_int, [_P, _P, _P, C.c_int64]),
    "tda_engine_run": (C.c_int, [_P, C.c_int64, C.POINTER(tda_outputs)]),
    "tda_engine_sync": (C.c_int, [_P]),
    "tda_engine_set_record_thinning": (C.c_int, [_P, C.c_int32]),
    "tda_engine_set_progress": (C.c_int, [_P, C.c_int]),
    "tda_engine_get_progress": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "tda_engine_get_proposal_state": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "tda_engine_detach_proposal_state": (C.c_int, [_P, C.POINTER(_P)]),
    "tda_proposal_snapshot_read": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "tda_proposal_snapshot_destroy": (None, [_P]),
    "tda_engine_get_flags": (C.c_int, [_P, _P]),
    "tda_engine_evaluate": (C.c_int, [_P, C.c_int, _P, C.c_int64, _P]),
    "tda_engine_rng_probe": (C.c_int, [_P, C.c_int64, _P, _P]),
    "tda_rng_philox": (C.c_int, [C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "tda_engine_set_profiling": (C.c_int, [_P, C.c_int]),
    "tda_engine_get_profile": (C.c_int, [_P, C.POINTER(tda_profile)]),
}

_lib = None


def _bind_single_hip_runtime():
    """One HIP runtime per process, and torch's libraries mapped BEFORE that runtime is initialised.

    The PyTorch-ROCm wheel bundles its own libamdhip64 (SONAME libamdhip64.so.7, the same as /opt/rocm's).  If this library
    pulled in /opt/rocm's copy first and torch loaded its own afterwards, the second runtime would find no GPU.  And when torch is
    imported AFTER the first HIP call of the process, mapping its libraries takes ~10 s instead of ~0.7 s (measured on MI355X,
    profiles/r03_api.json: the runtime then loads every code object of libtorch_hip eagerly instead of on first use) -- which is
    what made the first tda.sample() of a script that had not imported torch itself take 11 s.  So when torch is installed it is
    imported here, before libtinyda_hip.so is mapped and before any HIP call; without torch the library binds to the system
    runtime named in its RUNPATH."""
    import sys

    if "torch" in sys.modules:
        return
    try:
        import importlib.util

        if importlib.util.find_spec("torch") is not None:
            import torch  # noqa: F401  (device memory, streams and torch.distributed are this package's plumbing anyway)
    except Exception:  # a broken torch install: fall back to the system runtime
        pass


def load():
    """Load the HIP engine library; raises EngineError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EngineError(
            "libtinyda_hip.so not found at %s: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(the MH engine has no CPU fallback)" % LIB_PATH
        )
    _bind_single_hip_runtime()
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as exc:  # missing ROCm runtime etc.
        raise EngineError("cannot load %s: %s" % (LIB_PATH, exc)) from exc
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def load_from(path):
    """Bind ANOTHER library exporting the same C-ABI (tests and bench.py's cpu_baseline load oracle/_build/libtda_cpu.so,
    the CPU twin of the ABI, this way).  Never used by the package itself: nothing here falls back to it."""
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


def check(rc, lib=None):
    if rc != TDA_OK:
        raise EngineError("tinyda_amd engine error %d: %s" % (rc, (lib or load()).tda_last_error().decode()))
