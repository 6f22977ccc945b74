"""bench.py as the driver runs it (fresh subprocess, small arguments), its self-launching multi-rank path rehearsed on one GPU,
and the record-buffer capacity contract of tda_engine_run (a short buffer is an error code, never an out-of-bounds write)."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(argv, env_extra=None, timeout=900):
    env = dict(os.environ)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, cwd=ROOT, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert r.returncode == 0, "bench.py %s failed (rc %d):\n%s\n%s" % (argv, r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "expected ONE JSON line, got %d" % len(lines)
    return json.loads(lines[0])


def test_bench_with_the_drivers_arguments():
    """`python bench.py --gpus 1 --steps 20 --warmup 5` (round 1 died here with a GPU memory fault)"""
    out = _run_bench(["--gpus", "1", "--steps", "20", "--warmup", "5"])
    assert out["n_gpus"] == 1 and out["steps"] == 20 and out["warmup"] == 5
    assert out["unit"] == "evals/s" and out["dtype"] == "f64" and out["vs_baseline"] is None and out["scaling"] == "weak"
    assert "configs[1]" in out["config"]["workload"]
    per_step = out["config"]["mh_iterations_per_step"] * out["config"]["chains_per_gpu"]
    np.testing.assert_allclose(out["value"], per_step / (out["ms_per_step"] * 1e-3), rtol=1e-9)
    assert out["value"] > 5e7  # north_star's per-GPU figure is 4.75e7
    rf = out["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and 0.2 < rf["frac"] < 1.0
    np.testing.assert_allclose(rf["frac"], rf["achieved"] / rf["peak"], rtol=1e-9)
    assert 0.0 < rf["whole_pipeline_frac"] <= rf["frac"]
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
    np.testing.assert_allclose(cb["per_core"], cb["value"] / cb["cores"], rtol=1e-12)  # round 5: the job's CPU share, not the host's thread count
    assert "-march=native" in cb["flags"] and "4096 chains" in cb["sample"]
    assert cb["reference_shaped"].get("value", 0) > 0, cb["reference_shaped"]  # the reference's cost profile, one core
    assert "roofline_hbm_step_synchronous" not in out
    assert "ess" in out and "max_rhat" in out["ess"], out.get("ess_error")
    if out["ess"]["valid"]:
        assert out["ess_per_sec"] > 0 and out["ess"]["max_rhat"] < 1.05
    else:
        assert out["ess_per_sec"] is None
    # the other BASELINE configurations ride in the same line, full chain counts, each with its own roofline
    cfgs = out.get("configs")
    assert cfgs, out.get("configs_error")
    assert [c["tag"] for c in cfgs] == ["C2b", "C3", "C4/16", "C4/128", "C5-literal", "C5+AEM128", "C5+AEM256"]
    for c in cfgs:
        assert "error" not in c, c
        for key in ("name", "evals_per_s", "finest_it_per_s", "dominant_kernel", "bound", "flops_or_bytes_per_eval", "frac", "kernel_ms"):
            assert key in c, (c["tag"], key)
        assert c["bound"] in ("mfma", "hbm") and 0.0 < c["pipeline_frac"] <= c["frac"] < 1.0, c
        assert c["chains"] == (8192 if c["tag"].startswith("C4") else 4096)
        # round 5: median of >= 5 repetitions of a window of >= 0.1 s, spread reported
        assert c["repetitions"] >= 5 and len(c["window_seconds"]) == c["repetitions"] and c["calls_per_window"] >= 1, c["tag"]
        assert c["window_s"] >= 0.09 and 0.0 <= c["spread"] < 0.5, (c["tag"], c["window_s"], c["spread"])
    c4 = [c for c in cfgs if c["tag"].startswith("C4")]
    assert all("k_dreamz_draw" in c["dominant_kernel"] for c in c4)  # (round 5: draws and steps in one launch; before: priced on the SUM of the two)
    assert out["configs_seconds"] < 120.0
    # round 5 (item 2a): the matrix-core counters of the committed PMC pass ride beside the algorithmic frac, stale-stamped
    assert "mfma_util_counter" in rf and "mfma_flops_counted" in rf and "mfma_counter_stale" in rf


def test_bench_survives_a_hung_cpu_baseline_and_flags_stale_traffic():
    """the CPU legs run in a child process under a wall-clock cap: when it is exceeded the line still carries the GPU result and says
    what happened; `roofline.traffic` comes from a committed PMC pass and `traffic_stale` says whether the kernel source has changed
    since that pass"""
    out = _run_bench(["--steps", "2", "--warmup", "1", "--chains", "512", "--pilot", "200", "--burnin", "300", "--ess-iterations", "0", "--no-configs"],
                     env_extra={"TINYDA_CPU_BASELINE_CAP_S": "0.3"})
    assert out["value"] > 0 and "cpu_baseline" not in out and "timeout" in out["cpu_baseline_error"]
    assert isinstance(out["roofline"]["traffic_stale"], bool) and out["roofline"]["traffic_source"].endswith("pmc_traffic.json")
    import hashlib

    data = open(os.path.join(ROOT, "tinyda_amd", "csrc", "tda_kernels_mh.h"), "rb").read()
    blob = hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()
    stamped = json.load(open(os.path.join(ROOT, "profiles", out["roofline"]["traffic_source"]))).get("kernel_source_blob")
    assert out["roofline"]["traffic_stale"] == (stamped != blob)


def test_bench_refuses_more_ranks_than_devices():
    """--gpus 2 on a one-GPU box without the rehearsal switch: no line, a non-zero exit and the reason"""
    import torch

    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a box with fewer than 2 devices")
    env = {k: v for k, v in os.environ.items() if k != "TINYDA_BENCH_ONE_GPU"}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--chains", "256",
                        "--pilot", "100", "--burnin", "100", "--ess-iterations", "0", "--no-cpu-baseline"], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "refusing to report a multi-GPU figure" in (r.stdout + r.stderr)


@pytest.mark.parametrize("argv", [["--steps", "1", "--warmup", "0"], ["--steps", "2", "--warmup", "7"]])
def test_bench_odd_small_arguments(argv):
    """warm-up longer than the timed run, no warm-up at all: every run() stays inside its buffers"""
    out = _run_bench(argv + ["--chains", "512", "--pilot", "200", "--burnin", "300", "--ess-iterations", "0", "--no-cpu-baseline"])
    assert out["steps"] == int(argv[1]) and out["value"] > 0 and "roofline" in out


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` from a bare shell: two ranks on this one GPU (gloo rehearsal of the RCCL path)"""
    env = {"TINYDA_BENCH_ONE_GPU": "1"}
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        os.environ.pop(k, None)
    small = ["--steps", "3", "--warmup", "1", "--chains", "1024", "--pilot", "200", "--burnin", "500", "--ess-iterations", "0", "--no-cpu-baseline"]
    two = _run_bench(["--gpus", "2"] + small, env)
    assert two["n_gpus"] == 2 and two["config"]["chains_per_gpu"] == 1024 and two["value"] > 0
    one = _run_bench(["--gpus", "1"] + small)
    assert one["n_gpus"] == 1
    # the two ranks share one GPU here, so the pair cannot be faster than twice a single run nor much slower than one
    assert 0.4 * one["value"] < two["value"] < 2.2 * one["value"]


def test_scale_script_rehearsal_on_one_gpu(tmp_path):
    """tools/r05_scale.sh -- the one command for an 8-GPU node (bench at 1/2/4/8 ranks, config 4 in the four archive modes, the peer
    archive check) -- rehearsed with two ranks on this one GPU over gloo: every stage runs, every line carries its rank count."""
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        os.environ.pop(k, None)
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "r05_scale.sh"), str(tmp_path)], cwd=ROOT, env=dict(os.environ, REHEARSE="1"),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-4000:]
    rep = json.load(open(tmp_path / "summary.json"))
    assert [b["n_gpus"] for b in rep["bench"]] == [1, 2] and all(b["value"] > 0 and b["efficiency"] > 0 for b in rep["bench"])
    assert rep["bench"][1]["rccl_ranks"] == "gloo rehearsal on one GPU"
    modes = [(c["mode"], c["n_gpus"]) for c in rep["c4"]]
    assert modes == [(m, n) for n in (1, 2) for m in ("replicated-blocking", "replicated-overlapped", "distributed-sync", "distributed-lagged")]
    for c in rep["c4"]:
        assert c["evals_per_s"] > 0 and len(c["per_rank_evals_per_s"]) == c["n_gpus"] and c["efficiency"] > 0
        assert c["ran_as"].startswith(c["mode"].split("-")[0]) and c["note"] is None  # IPC mapping works between processes on one GPU
        assert c["archive_rows_rank0"] == 320 + (64 + 48) * 512 * c["n_gpus"]  # every chain's state of every step, whoever stores it
    assert rep["peer_archive_check"] and rep["peer_archive_check"][0]["world"] == 2
    # round 4: the other configurations per rank count (tools/configs_scale.py; rehearsal sizes)
    assert [(c["tag"], c["n_gpus"]) for c in rep["configs"]] == [("C3", 1), ("C5+AEM", 1), ("C3", 2), ("C5+AEM", 2)]
    assert all(c["evals_per_s"] > 0 and len(c["per_rank_evals_per_s"]) == c["n_gpus"] and c["efficiency"] > 0 for c in rep["configs"])


def _small_engine(eng_mod, N=32, d=8, m=16, n_levels=1):
    rng = np.random.default_rng(3)
    e = eng_mod.Engine(N, d, seed=5, n_levels=n_levels)
    e.set_prior(np.zeros(d), np.eye(d))
    for k in range(n_levels):
        A = rng.standard_normal((m * (k + 1), d)) / 3
        e.set_level(k, A, rng.standard_normal(m * (k + 1)), 0, 0.5)
    e.set_proposal(2, 0.05 * np.eye(d), t0=10, period=10)
    if n_levels > 1:
        e.set_subchains([3] * (n_levels - 1))
    e.init(np.zeros((N, d)))
    return e


def test_short_record_buffers_are_refused():
    import torch

    from tinyda_amd import _lib, engine

    N, d = 32, 8
    e = _small_engine(engine, N, d)
    dev = torch.device("cuda", 0)
    p = torch.zeros((20, N, d), dtype=torch.float64, device=dev)
    s = torch.zeros((20, N, 3), dtype=torch.float64, device=dev)
    a = torch.zeros((20, N), dtype=torch.uint8, device=dev)
    th0, st0 = e.current()
    # the round-1 bench bug: a slice that silently stays 20 rows, 100 iterations requested
    with pytest.raises(ValueError, match="holds 20 records"):
        e.run(100, p[:100], s[:100], a[:100])
    with pytest.raises(ValueError, match="float64"):
        e.run(10, p.float(), s, a)
    with pytest.raises(ValueError, match="shape"):
        e.run(10, p[:, :, :4].contiguous(), s, a)
    with pytest.raises(ValueError, match="holds 5 records"):
        e.run(10, None, np.zeros((5, N, 3)), None)
    # straight through the C-ABI: rows too small, and rows overstated against the real device allocation
    lib = e.lib
    out = _lib.tda_outputs(C.sizeof(_lib.tda_outputs), 20, C.c_void_p(p.data_ptr()), C.c_void_p(s.data_ptr()), C.c_void_p(a.data_ptr()))
    assert lib.tda_engine_run(e.h, 21, C.byref(out)) == _lib.TDA_ERR_INVALID
    assert b"rows" in lib.tda_last_error()
    out0 = _lib.tda_outputs(C.sizeof(_lib.tda_outputs), 0, C.c_void_p(p.data_ptr()), None, None)
    assert lib.tda_engine_run(e.h, 1, C.byref(out0)) == _lib.TDA_ERR_INVALID
    raw = torch.cuda.caching_allocator_alloc(4096, 0)
    try:
        lie = _lib.tda_outputs(C.sizeof(_lib.tda_outputs), 1000000, C.c_void_p(raw), None, None)
        assert lib.tda_engine_run(e.h, 1000000, C.byref(lie)) == _lib.TDA_ERR_INVALID
        assert b"allocation" in lib.tda_last_error()
    finally:
        torch.cuda.caching_allocator_delete(raw)
    # nothing ran: the chains are where they were, and a correct call still works
    th1, st1 = e.current()
    assert np.array_equal(th0, th1) and np.array_equal(st0, st1)
    e.run(20, p, s, a)
    assert a.sum().item() > 0
    e.close()


def test_short_multilevel_buffers_are_refused():
    import torch

    from tinyda_amd import engine

    N, d = 32, 8
    e = _small_engine(engine, N, d, n_levels=2)
    dev = torch.device("cuda", 0)

    def bufs(rows):
        return (torch.zeros((rows, N, d), dtype=torch.float64, device=dev), torch.zeros((rows, N, 3), dtype=torch.float64, device=dev),
                torch.zeros((rows, N), dtype=torch.uint8, device=dev))

    assert e.rows_per_level(10) == [30, 10]
    with pytest.raises(ValueError, match="level 0 holds 10 records"):
        e.run_levels(10, [bufs(10), bufs(10)])  # the coarse level needs 30 rows
    e.run_levels(10, [bufs(30), bufs(10)])
    e.close()


def test_collectives_on_rccl_with_one_rank():
    """tools/rccl_smoke.py under torch.distributed.run, backend nccl (= RCCL), world size 1: every collective the N > 1 path
    issues (archive all_gather blocking and overlapped, pooled-moment all_reduce, timing all_reduce, barrier) runs on the real
    library and leaves the one-rank results bit-identical"""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TINYDA_BENCH_ONE_GPU"):
        env.pop(k, None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    import socket

    sock = socket.socket()  # a port nobody holds (a fixed one can be busy on a shared box: the rendezvous then waits minutes)
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tools", "rccl_smoke.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["ok"] and out["backend"] == "nccl"
    assert all(v for k, v in out.items() if k.endswith("_identical"))


def test_distributed_archive_across_processes():
    """tools/peer_archive_check.py: two processes on this GPU, archive segments exchanged as IPC handles and mapped into each
    other's address space, block-wise publish protocol driven by distributed.run_peer_dream over gloo; the union of the two
    ranks' records equals the one-engine run with the replicated archive"""
    import socket

    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "TINYDA_BENCH_ONE_GPU", "TINYDA_FORCE_COLLECTIVES"):
        env.pop(k, None)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tools", "peer_archive_check.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["ok"], out
