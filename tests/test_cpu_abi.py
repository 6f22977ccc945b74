"""libtda_cpu.so -- the C-ABI of include/tinyda_amd.h compiled for the CPU (oracle/tda_cpu_abi.cpp; test / baseline
infrastructure) -- driven through the SAME ctypes binding and Engine wrapper as the GPU library, against traces recorded from
tinyDA itself.  No GPU needed: this is how the binding layer is exercised in the CPU test tier."""
import os
import re
import subprocess

import numpy as np
import pytest

from oracle import tinyda_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


_SANITIZED = "TINYDA_CPU_ABI_SO" in os.environ  # (one build, named by the parent test, under the sanitizers)


@pytest.fixture(scope="module", params=["portable"] if _SANITIZED else ["portable", "native"])
def cpu(request):
    """both builds of oracle/tda_cpu_abi.cpp: libtda_cpu.so (-O2) and libtda_cpu_native.so (-O3 -march=native for this host:
    what bench.py's cpu_baseline leg times, round 5)"""
    import __graft_entry__ as g

    g.build()
    from tinyda_amd import _lib

    # tests/test_cpu_twin_sanitized.py re-runs this module in a child process against an ASan + UBSan build of the same source
    if _SANITIZED:
        path = os.environ["TINYDA_CPU_ABI_SO"]
    else:
        path = g.CPU_ABI_SO if request.param == "portable" else g.build_cpu_native()
        if path is None:
            pytest.skip("no compiler for the -march=native build")
    return _lib.load_from(path), path


def test_cpu_twin_exports_the_whole_abi(cpu):
    lib, path = cpu
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "tinyda_amd.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(tda_[a-z_]+)\s*\(", text))
    out = subprocess.run(["nm", "-D", "--defined-only", path], stdout=subprocess.PIPE, text=True).stdout
    assert declared <= set(re.findall(r"\bT (tda_[a-z_]+)", out))
    assert b"cpu twin" in lib.tda_version()


@pytest.mark.parametrize("name", ["g1_basic_sampler", "g2_am_small", "g2_am_small_adaptive", "g2b_pcn", "g2_am_c2", "g2_am_d96"])
def test_cpu_twin_replays_reference_traces(cpu, golden, name):
    from tinyda_amd.engine import Engine

    lib, _ = cpu
    g = golden(name)
    N, T1, d = g["theta"].shape
    T = T1 - 1
    e = Engine(N, d, seed=1, lib=lib)
    e.set_prior(g["prior_mean"], g["prior_cov"])
    if "noise_var" in g.files:
        e.set_level(0, g["A"], g["data"], 0, float(g["noise_var"]))
    else:
        e.set_level(0, g["A"], g["data"], 0, float(g["noise_cov"][0]))
    if name.startswith("g1"):
        e.set_proposal(0, g["C"], scaling=float(g["scaling0"]), adaptive=True, gamma=float(g["gamma"]), period=int(g["period"]))
    elif name.startswith("g2b"):
        e.set_proposal(1, None, scaling=float(g["scaling0"]), adaptive=True, gamma=float(g["gamma"]), period=int(g["period"]))
    else:
        e.set_proposal(2, g["C0"], sd=float(g["sd"]), epsilon=float(g["epsilon"]), t0=int(g["t0"]), period=int(g["period"]),
                       adaptive=bool(g["adaptive"]), gamma=float(g["gamma"]))
    e.init(g["theta0"])
    e.set_replay(np.swapaxes(g["z"], 0, 1), np.swapaxes(g["u"], 0, 1))
    params, stats, acc = np.empty((T, N, d)), np.empty((T, N, 3)), np.empty((T, N), dtype=np.uint8)
    half = T // 2  # a run cut in two continues where it stopped
    e.run(half, params[:half], stats[:half], acc[:half])
    e.run(T - half, params[half:], stats[half:], acc[half:])
    assert np.array_equal(acc, np.swapaxes(g["accepted"][:, 1:], 0, 1))
    np.testing.assert_allclose(stats[:, :, 2], np.swapaxes(g["logpost"][:, 1:], 0, 1), rtol=1e-10)
    # (after a covariance swap the proposals carry the last-bit differences between this Cholesky and LAPACK's)
    np.testing.assert_allclose(params, np.swapaxes(g["theta"][:, 1:], 0, 1), rtol=1e-9, atol=1e-11)
    st = e.proposal_state(want_am=name.startswith("g2_"))
    if "scaling" in g.files:
        np.testing.assert_allclose(st["scaling"], np.atleast_1d(g["scaling"])[-N:] if np.ndim(g["scaling"]) == 1 else g["scaling"][:, -1], rtol=1e-12)
    if name.startswith("g2_"):
        np.testing.assert_allclose(st["am_sigma"], g["sigma_hist"][:, -1], rtol=1e-9, atol=1e-12)
    with pytest.raises(ValueError):  # the wrapper's capacity check
        e.run(5, params[:3], None, None)
    e.close()


def test_cpu_twin_philox_stream_is_the_contract(cpu):
    """engine-generated variates (RNG contract of the header) = the oracle's independent restatement; theta0 ~ prior"""
    from tinyda_amd.engine import Engine

    lib, _ = cpu
    N, d, T = 5, 7, 9
    rng = np.random.default_rng(0)
    A = rng.standard_normal((11, d))
    e = Engine(N, d, seed=(7 << 32) | 12345, chain_offset=3, lib=lib)
    e.set_prior(np.zeros(d), np.eye(d))
    e.set_level(0, A, rng.standard_normal(11), 1, 0.5 + rng.random(11))
    e.set_proposal(2, 0.05 * np.eye(d), t0=4, period=4)
    e.init(None)
    z, u = e.set_export(T)
    e.run(T)
    st = orc.PhiloxStream((7 << 32) | 12345)
    chains = np.arange(3, 3 + N)
    for t in range(T):
        np.testing.assert_allclose(z[t], st.normals(chains, t, d), rtol=1e-13, atol=1e-15)
        assert np.array_equal(u[t], st.uniform(chains, t))
    zz, uu = e.rng_probe(4)
    np.testing.assert_allclose(zz, z[4], rtol=0, atol=0)
    e.close()
    with pytest.raises(Exception, match="CPU twin"):
        Engine(4, 3, n_levels=2, lib=lib)


def test_native_build_rounds_like_the_portable_one():
    """-O3 -march=native vectorises the forward model, the draw and the moment recursion over outputs / columns; every sum keeps
    the scalar loop's order, so the two builds agree BITWISE on an engine-driven AdaptiveMetropolis run with a covariance swap"""
    import __graft_entry__ as g

    g.build()
    from tinyda_amd import _lib
    from tinyda_amd.engine import Engine

    native = os.environ["TINYDA_CPU_ABI_SO"] if _SANITIZED else g.build_cpu_native()  # (sanitized: that build against the -O2 one)
    if native is None:
        pytest.skip("no compiler for the -march=native build")
    rng = np.random.default_rng(4)
    d, m, N, T = 13, 45, 6, 70
    A = rng.standard_normal((m, d)) / 3
    y = rng.standard_normal(m)
    res = []
    for path in (g.CPU_ABI_SO, native):
        e = Engine(N, d, seed=99, lib=_lib.load_from(path))
        e.set_prior(0.1 * np.ones(d), np.eye(d) + 0.1)
        e.set_level(0, A, y, 1, 0.5 + rng.random(m) * 0 + 0.25)
        e.set_proposal(2, 0.01 * np.eye(d), t0=20, period=20, adaptive=True)
        e.init(None)
        params, stats, acc = np.empty((T, N, d)), np.empty((T, N, 3)), np.empty((T, N), dtype=np.uint8)
        e.run(T, params, stats, acc)
        st = e.proposal_state(want_am=True)
        res.append((params, stats, acc, st["am_sigma"], st["scaling"]))
        e.close()
    for a, b in zip(*res):
        assert np.array_equal(a, b)
    assert 0.02 < res[0][2].mean() < 0.98


def test_bench_counts_the_cores_it_may_use():
    """bench.py's cpu_baseline reports `cores` = the threads it started = min(CPU count, affinity mask, cgroup quota), and per_core"""
    import importlib.util

    spec = importlib.util.spec_from_file_location("tda_bench_for_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n = bench.effective_cores()
    assert 1 <= n <= (os.cpu_count() or 1) and n <= len(os.sched_getaffinity(0))
    assert (bench.CPU_SAMPLE_CHAINS, bench.CPU_SAMPLE_ITERATIONS) == (4096, 1000)
