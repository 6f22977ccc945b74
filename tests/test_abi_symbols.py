"""The C-ABI library builds, loads, and exports exactly what include/tinyda_amd.h declares (no GPU needed)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g

    g.build()
    from tinyda_amd import _lib

    return _lib


def _declared():
    text = open(os.path.join(ROOT, "include", "tinyda_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(tda_[a-z_]+)\s*\(", text))


def test_header_and_binding_agree(lib):
    assert _declared() == set(lib.SYMBOLS), "ctypes table and header drifted apart"


def test_library_exports_every_declared_symbol(lib):
    out = subprocess.run(["nm", "-D", "--defined-only", lib.LIB_PATH], stdout=subprocess.PIPE, text=True).stdout
    exported = set(re.findall(r"\bT (tda_[a-z_]+)", out))
    assert _declared() <= exported, _declared() - exported
    loaded = lib.load()
    assert loaded.tda_version().startswith(b"tinyda_amd")


def test_struct_sizes_match_header(lib):
    # natural alignment on x86-64; guards against field drift between header and ctypes
    import ctypes as C

    assert C.sizeof(lib.tda_config) == 56
    assert C.sizeof(lib.tda_proposal_params) == 72  # + q_mean (independence sampler)
    assert C.sizeof(lib.tda_outputs) == 32
    assert C.sizeof(lib.tda_profile) == 64


def test_product_fails_loudly_without_gpu(lib):
    """No CPU fallback: on a box without a GPU the device path raises instead of computing elsewhere."""
    import numpy as np
    import scipy.stats as st

    import tinyda_amd as tda

    try:
        import torch

        if torch.cuda.is_available():
            pytest.skip("GPU present")
    except ImportError:
        pass
    A = np.eye(3)
    post = tda.Posterior(st.multivariate_normal(np.zeros(3), np.eye(3)), tda.GaussianLogLike(np.zeros(3), np.eye(3)),
                         tda.LinearModel(A))
    with pytest.raises(tda.EngineError):
        tda.sample(post, tda.GaussianRandomWalk(np.eye(3)), 10, n_chains=4)
