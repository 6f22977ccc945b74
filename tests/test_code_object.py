"""The shipped gfx950 code object, read without a GPU: the kernels of the BASELINE configurations must not spill registers.

A spilled register in a step loop is not two instructions: on gfx9 a scratch reload counts on the same in-order counter as
global loads and stores, so its use is `s_waitcnt vmcnt(0)` -- every prefetch in flight is waited for (DESIGN.md 5, round 3:
three reloads per coarse step cost `k_da_steps` 2 ms of 15).  The AMDGPU metadata note of the code object states spilled
registers and scratch bytes per kernel; `profiles/r03_scratch_counts.txt` is the same table for every instance."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

LLVM_BIN = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tinyda_amd", "lib", "libtinyda_hip.so")

# kernel instance (mangled prefix) -> what launches it
HOT = {
    "_ZN3tda10k_mh_stepsILi64ELi8ELb0ELi0EEE": "C2a step kernel",
    "_ZN3tda7k_adaptILi64ELb1EEE": "C2a moment recursion (diagonal blocks as circulant slots, round 5)",
    "_ZN3tda7k_adaptILi64ELb0EEE": "C2a moment recursion, full diagonal tiles (TINYDA_ADAPT_CIRC=0)",
    "_ZN3tda16k_chol_apply_blkILi64ELb1EEE": "C2a covariance swap + increments",
    "_ZN3tda7k_applyILi64EEE": "C2a increments",
    "_ZN3tda10k_da_stepsILi64ELi2ELb1ELi0ELi2EEE": "C3 Delayed Acceptance (replay mode, non-identity proposal factors)",
    "_ZN3tda15k_da_steps_r224ILi64ELi2ELb1ELi0ELi2EEE": "C3 Delayed Acceptance, 224-register entry point (draws of the next block beside it)",
    "_ZN3tda10k_da_stepsILi64ELi1ELb0ELi0ELi3EEE": "C5-literal MLDA",
    "_ZN3tda12k_rng_directILi64EEE": "C3 proposal normals",
    "_ZN3tda9k_proposeILi64EEE": "C5 proposal increments",
    "_ZN3tda19k_dreamz_steps_waveILi32EEE": "C4 DREAM steps",
    "_ZN3tda10k_ml_stepsILi64ELi1ELi4ELb0EEE": "dense error model base steps, step-by-step path (TINYDA_AEM_BASE=0, dense priors)",
    "_ZN3tda16k_aem_base_stepsILi8EEE": "C5 + dense error model base subchain (one pass over V per launch)",
    "_ZN3tda12k_aem_actionILi128EEE": "C5 + dense error model level decision",
    "_ZN3tda13k_aem_refreshILi8ELi1EEE": "C5 + dense error model refresh of level 1 (one tracker)",
    "_ZN3tda13k_aem_refreshILi8ELi2EEE": "C5 + dense error model refresh of level 0 (two trackers)",
    "_ZN3tda18k_adapt_chol_applyILi64ELb1EEE": "C2a period boundary in one launch (diagonal blocks as circulant slots, round 5)",
    "_ZN3tda18k_adapt_chol_applyILi64ELb0EEE": "C2a period boundary in one launch, full diagonal tiles (TINYDA_ADAPT_CIRC=0)",
    "_ZN3tda13k_dreamz_drawILi32ELb0ELb0EEE": "C4 DREAM draws (two-kernel path, TINYDA_DZ_FUSED=0)",
    "_ZN3tda13k_dreamz_drawILi32ELb0ELb1EEE": "C4 DREAM block, draws and steps in one launch (round 5)",
}

# Every OTHER instance of the code object must be spill-free too, except the ones listed here with the number of spilled registers
# they are known to have (a cap per instance, so that neither a new spiller nor a worse one goes unnoticed).
#   (k_aem_refresh<8, *> left this list in round 5: the factor form -- no V = L^-1 beside U -- needs 412 - 452 of the 512 registers)
#   k_ml_steps<64,4,4,false>   the generic level kernel with four levels in one launch (C5 runs k_da_steps): 3 (round 4: 10 / 7 with three /
#                              four levels, 46 / 117 before the upper levels' state waited in LDS; round 5: its error-model evaluation
#                              solves in place in LDS, aem_quad_factor_inplace);
#   k_ml_steps<128,3|4,4,false> the level kernel at 65 .. 128 parameters with three / four levels (one observation block in flight, no prior
#                              in registers: with them the TWO-level instance spilled 385)
#   k_ml_steps<64,*,4,true>    its instances for hierarchies with a dense observation covariance on some level (round 4)
KNOWN_SPILLERS = {
    "_ZN3tda13k_adapt_splitILi64EEE": 8,  # (TINYDA_ADAPT_SPLIT=1, an A/B switch: the moment recursion on two waves per chain at three waves per SIMD)
    "_ZN3tda10k_ml_stepsILi64ELi4ELi4ELb0EEE": 8,
    "_ZN3tda10k_ml_stepsILi64ELi5ELi4ELb0EEE": 56, "_ZN3tda10k_ml_stepsILi64ELi6ELi4ELb0EEE": 100,  # (five / six levels, round 5: 51 / 95; no spill at 8 .. 32 parameters)
    "_ZN3tda10k_ml_stepsILi128ELi3ELi4ELb0EEE": 4, "_ZN3tda10k_ml_stepsILi128ELi4ELi4ELb0EEE": 48,  # (MLDA above 64 parameters, round 5: 2 / 44; the two-level instance spills nothing)
    "_ZN3tda10k_ml_stepsILi64ELi2ELi4ELb1EEE": 24, "_ZN3tda10k_ml_stepsILi64ELi3ELi4ELb1EEE": 64, "_ZN3tda10k_ml_stepsILi64ELi4ELi4ELb1EEE": 104,
}


@pytest.fixture(scope="module")
def kernel_metadata():
    tools = [os.path.join(LLVM_BIN, t) for t in ("llvm-objdump", "llvm-readelf")]
    if not all(os.path.exists(t) for t in tools):
        pytest.skip("ROCm LLVM tools not installed")
    if not os.path.exists(LIB):
        pytest.skip("libtinyda_hip.so not built (python -c 'import __graft_entry__ as g; g.build()')")
    tmp = tempfile.mkdtemp(prefix="tda_co_")
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(LIB, so)
        subprocess.run([tools[0], "--offloading", so], cwd=tmp, check=True, capture_output=True)  # writes lib.so.0.hipv4-...gfx950
        co = [f for f in os.listdir(tmp) if "gfx950" in f]
        assert co, "no gfx950 code object in the library"
        notes = subprocess.run([tools[1], "--notes", os.path.join(tmp, co[0])], check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    meta = {}
    for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", notes):
        meta[m.group(1)] = dict(scratch_bytes=int(m.group(2)), vgprs=int(m.group(3)), spilled=int(m.group(4)))
    assert len(meta) > 100, "kernel metadata not parsed"
    return meta


@pytest.mark.parametrize("prefix", sorted(HOT))
def test_hot_kernels_do_not_spill(kernel_metadata, prefix):
    hits = {k: v for k, v in kernel_metadata.items() if k.startswith(prefix)}
    assert len(hits) == 1, "%s (%s): %d instances in the code object" % (prefix, HOT[prefix], len(hits))
    (name, md), = hits.items()
    assert md["spilled"] == 0, "%s (%s) spills %d registers" % (name, HOT[prefix], md["spilled"])
    # (a frame of the out-of-line accept_exact is allowed: 32 bytes, no spill)
    assert md["scratch_bytes"] <= 32, "%s (%s) uses %d bytes of scratch" % (name, HOT[prefix], md["scratch_bytes"])


def test_every_instance_is_spill_free_or_listed(kernel_metadata):
    """ALL instances of the code object, not a hand-picked list: zero spilled registers, or on KNOWN_SPILLERS within its cap"""
    bad = []
    for name, md in sorted(kernel_metadata.items()):
        cap = max([c for p, c in KNOWN_SPILLERS.items() if name.startswith(p)] or [0])
        if md["spilled"] > cap:
            bad.append("%s: %d spilled registers (allowed %d), %d bytes of scratch" % (name, md["spilled"], cap, md["scratch_bytes"]))
    assert not bad, "\n".join(bad)
    stale = [p for p in KNOWN_SPILLERS if not any(n.startswith(p) and md["spilled"] > 0 for n, md in kernel_metadata.items())]
    assert not stale, "no longer spilling (take them off KNOWN_SPILLERS): %s" % stale


def test_rng_kernel_fits_beside_the_step_kernel(kernel_metadata):
    """k_rng runs on the second stream UNDER k_mh_steps: two step waves and one generator wave share a SIMD's 512 registers"""
    step = [v for k, v in kernel_metadata.items() if k.startswith("_ZN3tda10k_mh_stepsILi64ELi8ELb0ELi0EEE")][0]
    rng = [v for k, v in kernel_metadata.items() if k.startswith("_ZN3tda5k_rngILi64EEE")][0]
    gran = lambda n: (n + 7) // 8 * 8  # allocation granule
    assert 2 * gran(step["vgprs"]) + gran(rng["vgprs"]) <= 512, (step, rng)


def test_generator_fits_beside_the_two_level_kernel(kernel_metadata):
    """run_multilevel draws block b + 1 on the second stream under block b when k_rng_direct's waves fit beside a resident tile of the
    level kernel (two waves per SIMD): the 224-register entry point of C3's instance and the one-block two-level instances as
    compiled must leave 64 registers, without spilling"""
    gran = lambda n: (n + 7) // 8 * 8
    rng = [v for k, v in kernel_metadata.items() if k.startswith("_ZN3tda12k_rng_directILi64EEE")][0]
    assert gran(rng["vgprs"]) <= 64 and rng["spilled"] == 0, rng
    fit = ["_ZN3tda15k_da_steps_r224ILi64ELi2ELb1ELi0ELi2EEE", "_ZN3tda15k_da_steps_r224ILi64ELi2ELb0ELi0ELi2EEE",
           "_ZN3tda10k_da_stepsILi64ELi1ELb1ELi0ELi2EEE", "_ZN3tda10k_da_stepsILi64ELi1ELb0ELi0ELi2EEE",
           "_ZN3tda10k_da_stepsILi64ELi1ELb1ELi1ELi2EEE", "_ZN3tda10k_da_stepsILi64ELi1ELb0ELi1ELi2EEE"]
    for pre in fit:
        md = [v for k, v in kernel_metadata.items() if k.startswith(pre)]
        assert md, pre
        assert md[0]["spilled"] == 0 and 2 * gran(md[0]["vgprs"]) + 64 <= 512, (pre, md[0])


# ---- loop-level gate (round 5): the STEP LOOP of each hot kernel, located in the disassembly ----
# kernel instance -> (cap on `s_waitcnt vmcnt(0)` inside its step loop, what the ones that remain are).  The caps are the counts of
# the shipped build (tools/loop_audit.py prints them); a new full wait in a step loop -- the signature of a spill reload, of an LDS
# pointer demoted to a flat access, or of a load whose count the compiler lost -- fails here, on the CPU tier.
STEP_LOOP_VMCNT0 = {
    # 32 of the 39 sit in the dense-prior branch (two loads per parameter block, never taken with a diagonal prior: BASELINE);
    # on the C2a path: the drain behind the last operator block of the step (2, either exit), the increments + uniform requested a
    # whole step earlier in front of the step's barrier (1) -- true dependences -- and the alternatives of those for the other
    # operator-block counts (4)
    "_ZN3tda10k_mh_stepsILi64ELi8ELb0ELi0EEE": (39, "C2a"),
    "_ZN3tda10k_mh_stepsILi64ELi4ELb0ELi0EEE": (41, "C2b (34 of them in the dense-prior branch)"),
    # round 5: 72 -> 12.  Rounds 3-4 read the level action's states through a flat pointer (tda_kernels_mh.h block_sse_frag):
    # eight full waits per observation block of every level action
    "_ZN3tda15k_da_steps_r224ILi64ELi1ELb0ELi0ELi3EEE": (12, "C5-literal"),
    "_ZN3tda15k_da_steps_r224ILi64ELi2ELb1ELi0ELi2EEE": (6, "C3"),
    "_ZN3tda19k_dreamz_steps_waveILi32EEE": (10, "C4 steps"),
    "_ZN3tda16k_aem_base_stepsILi8EEE": (2, "C5 + dense error model, base subchain"),
}


@pytest.fixture(scope="module")
def loop_audit():
    import importlib.util

    if not os.path.exists(os.path.join(LLVM_BIN, "llvm-objdump")):
        pytest.skip("ROCm LLVM tools not installed")
    if not os.path.exists(LIB):
        pytest.skip("libtinyda_hip.so not built")
    spec = importlib.util.spec_from_file_location("tda_loop_audit", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "loop_audit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.audit(["k_mh_steps", "k_da_steps", "k_dreamz_steps_wave", "k_aem_base_steps", "k_ml_steps", "k_aem_action", "k_adapt"])


@pytest.mark.parametrize("prefix", sorted(STEP_LOOP_VMCNT0))
def test_step_loops_hold_their_full_wait_count(loop_audit, prefix):
    cap, what = STEP_LOOP_VMCNT0[prefix]
    hits = {k: v for k, v in loop_audit.items() if k.startswith(prefix)}
    assert len(hits) == 1, (prefix, list(hits))
    (name, r), = hits.items()
    assert r["step"] is not None, "%s: no step loop found in the disassembly" % name
    st = r["step"][2]
    assert st["instructions"] > 200, (name, st)  # it IS the step loop, not a staging loop
    assert st["scratch"] == 0, "%s (%s): %d scratch accesses inside the step loop" % (name, what, st["scratch"])
    assert st["flat"] == 0, "%s (%s): %d flat accesses inside the step loop (an LDS / global pointer lost its address space)" % (name, what, st["flat"])
    assert st["vmcnt0"] <= cap, "%s (%s): %d `s_waitcnt vmcnt(0)` inside the step loop, %d allowed" % (name, what, st["vmcnt0"], cap)


@pytest.mark.parametrize("prefix", ["_ZN3tda7k_adaptILi64ELb1EEE", "_ZN3tda18k_adapt_chol_applyILi64ELb1EEE"])
def test_moment_recursion_state_loop(loop_audit, prefix):
    """the state loop of the AdaptiveMetropolis moment recursion (a third of C2a's period, the loop with the most vector
    instructions of its kernel; it stores nothing, so step_loop() does not find it): two states per pass, 264 operations per state
    with the diagonal blocks as circulant slots and no register copies -- at most 640 vector instructions per pass (605 shipped; the tile
    version: 2 x 394), nothing spilled, no flat access, the full waits are the state prefetch (one per state) + the coefficient refresh"""
    hits = {k: v for k, v in loop_audit.items() if k.startswith(prefix)}
    assert len(hits) == 1, (prefix, list(hits))
    (name, r), = hits.items()
    lo, hi, depth, st = max(r["loops"], key=lambda l: (l[3]["valu"], -l[2]))
    assert st["mfma"] == 0 and st["lds"] >= 40, (name, st)  # it IS the state loop (operand exchange through LDS, no matrix instruction)
    assert st["valu"] <= 640, "%s: %d vector instructions per pass of two states" % (name, st["valu"])
    assert st["scratch"] == 0 and st["flat"] == 0, (name, st)
    assert st["vmcnt0"] <= 4, (name, st)


def test_no_level_kernel_reads_memory_through_flat_pointers(loop_audit):
    """every k_da_steps / k_ml_steps / k_mh_steps instance: no flat_load / flat_store anywhere (rounds 3-4 shipped 64 per three-level
    k_da_steps instance: LDS reads behind `s_waitcnt vmcnt(0) lgkmcnt(0)`)"""
    bad = ["%s: %d" % (k, v["whole"]["flat"]) for k, v in sorted(loop_audit.items())
           if v["whole"].get("flat", 0) and ("k_da_steps" in k or "k_ml_steps" in k or "k_mh_steps" in k)]
    assert not bad, "\n".join(bad)
