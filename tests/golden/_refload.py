"""Import the upstream reference (tinyDA) in THIS container only, to generate golden vectors.

The reference cannot travel to the GPU box; nothing under tests/ imports this module at test
time.  It is used by gen_golden.py (committed next to the fixtures it produced).

Recipe: `ray`, `xarray` and `arviz` are not installed, so they are replaced
by inert stand-in modules *in sys.modules only* before `import tinyDA`; no bytecode is written
into /root/reference.
"""
import sys
import types

REFERENCE_ROOT = "/root/reference"


def load_reference():
    sys.dont_write_bytecode = True
    if "tinyDA" in sys.modules:
        return sys.modules["tinyDA"]

    ray = types.ModuleType("ray")

    def _remote(obj=None, **_kw):
        class _Stub:
            def __init__(self, *a, **k):
                raise RuntimeError("ray stand-in: remote actors are not available")

            @classmethod
            def remote(cls, *a, **k):
                raise RuntimeError("ray stand-in: remote actors are not available")

        return _Stub if obj is not None else (lambda o: _Stub)

    ray.remote = _remote
    ray.init = lambda *a, **k: None
    ray.get = lambda x: x
    sys.modules["ray"] = ray
    sys.modules["xarray"] = types.ModuleType("xarray")
    sys.modules["arviz"] = types.ModuleType("arviz")
    sys.path.insert(0, REFERENCE_ROOT)
    import tinyDA  # noqa: E402

    return tinyDA
