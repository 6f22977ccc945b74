"""The kernels an environment switch selects between must agree: every switch in README.md that changes which kernel runs is
exercised both ways on the same seeded problem (fresh processes: the switches are read once) -- accept flags equal, states and
log-densities to rounding (the lean kernels sum in a different order), bitwise where the arithmetic is the same."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _probe(what, env_extra, tmp_path, tag):
    out = str(tmp_path / ("%s_%s.npz" % (what, tag)))
    env = dict(os.environ)
    for k in ("TINYDA_DA_LEAN", "TINYDA_DZ_WAVE", "TINYDA_DZ_PIPELINE", "TINYDA_AEMD_FUSED", "TINYDA_FUSE_CHOL_APPLY", "TINYDA_CHOL_BLOCKED",
              "TINYDA_FUSE_ADAPT_CHOL", "TINYDA_ADAPT_SPLIT", "TINYDA_ADAPT_CIRC", "TINYDA_AEM_BASE", "TINYDA_DZ_FUSED", "TINYDA_ML_SPLIT", "TINYDA_DA_R224", "TINYDA_AM_DEFER", "TINYDA_ML_PREDRAW", "TINYDA_AEM_PRE"):
        env.pop(k, None)
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "switch_probe.py"), what, out], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return dict(np.load(out))


@pytest.mark.parametrize("what", ["mlda3", "da2", "mlda3_ragged", "da2_ragged"])
def test_pipelined_level_kernel_equals_the_generic_one(what, tmp_path):
    """k_da_steps (two levels, and three with a coarse operator of <= 128 observations) against k_ml_steps (TINYDA_DA_LEAN=0)"""
    lean, generic = _probe(what, {}, tmp_path, "lean"), _probe(what, {"TINYDA_DA_LEAN": "0"}, tmp_path, "generic")
    for k in lean:
        if k.startswith("acc"):
            assert np.array_equal(lean[k], generic[k]), "%s: %d accept flips" % (k, int((lean[k] != generic[k]).sum()))
        else:
            np.testing.assert_allclose(lean[k], generic[k], rtol=1e-10, atol=1e-12, err_msg=k)
    assert 0.02 < lean["acc0"].mean() < 0.98


@pytest.mark.parametrize("what", ["da2", "da2_ragged", "mlda3", "mlda3_ragged"])
def test_draws_on_the_second_stream_change_nothing(what, tmp_path):
    """run_multilevel draws block b + 1 under block b's level kernel when the generator fits beside it (the 224-register entry
    points of k_da_steps) -- against TINYDA_ML_SPLIT=0: one stream, k_propose in front of every block, the kernel as compiled.
    pCN with the identity factor (da2): the same arithmetic, so the same bits.  AdaptiveMetropolis (mlda3): increments by k_apply
    / the swap launch instead of k_propose, the same products in the same order."""
    on, off = _probe(what, {}, tmp_path, "split"), _probe(what, {"TINYDA_ML_SPLIT": "0"}, tmp_path, "one_stream")
    for k in on:
        if k.startswith("acc") or what.startswith("da2"):
            assert np.array_equal(on[k], off[k]), "%s differs (%s)" % (k, what)
        else:
            np.testing.assert_allclose(on[k], off[k], rtol=1e-10, atol=1e-12, err_msg=k)
    assert 0.02 < on["acc0"].mean() < 0.98
    # the 224-register entry point against the kernel as compiled, both with the draws on the second stream
    plain = _probe(what, {"TINYDA_DA_R224": "0", "TINYDA_ML_SPLIT": "1"}, tmp_path, "plain_kernel")
    for k in on:
        assert np.array_equal(on[k], plain[k]), "%s differs between the two entry points (%s)" % (k, what)


@pytest.mark.parametrize("what", ["am", "am_ragged"])
def test_fused_swap_and_increments_equal_the_two_launches(what, tmp_path):
    """k_chol_apply (covariance swap + the next block's increments in one launch, the factor read back from L2 by the wave that
    wrote it) against k_chol followed by k_apply (TINYDA_FUSE_CHOL_APPLY=0): the same arithmetic, bit for bit, over five swaps"""
    off = {"TINYDA_CHOL_BLOCKED": "0"}  # the row-per-lane factorisation in both: the same arithmetic, so the same bits
    fused, split = _probe(what, off, tmp_path, "fused"), _probe(what, dict(off, TINYDA_FUSE_CHOL_APPLY="0"), tmp_path, "split")
    for k in fused:
        assert np.array_equal(fused[k], split[k]), k
    assert 0.02 < fused["acc0"].mean() < 0.98 and np.abs(fused["C"]).max() > 0
    # the blocked factorisation (default at 64 parameters) sums in a different order and takes its pivots' reciprocal square roots
    # from the hardware estimate + one correction: the same factor to rounding, the same decisions
    blk = _probe(what, {}, tmp_path, "blocked")
    assert np.array_equal(blk["acc0"], fused["acc0"])
    # (the probe swaps after 60 steps in 64 dimensions: its first covariances are rank-deficient up to the eps I term, so their
    # factors -- and the states proposed from them -- move by the rounding error times a large condition number)
    for k, tol in (("params0", 1e-8), ("stats0", 1e-6), ("C", 1e-8), ("sigma", 1e-8), ("scaling", 1e-12)):
        np.testing.assert_allclose(blk[k], fused[k], rtol=tol, atol=tol * np.abs(fused[k]).max(), err_msg=k)
    blk2 = _probe(what, {"TINYDA_FUSE_CHOL_APPLY": "0"}, tmp_path, "blocked_split")  # k_chol_apply_blk<., false> + k_apply
    assert np.array_equal(blk2["acc0"], blk["acc0"])
    np.testing.assert_allclose(blk2["stats0"], blk["stats0"], rtol=1e-6)
    # round 4: the moment recursion, the swap and the increments in ONE launch (k_adapt_chol_apply, the default at 64 parameters;
    # Sigma goes from the recursion's registers into the factorisation through a register transpose) against k_adapt followed by
    # k_chol_apply_blk (TINYDA_FUSE_ADAPT_CHOL=0): the same arithmetic on the same values -- every record and the final state bitwise
    two = _probe(what, {"TINYDA_FUSE_ADAPT_CHOL": "0"}, tmp_path, "two_launches")
    for k in blk:
        assert np.array_equal(blk[k], two[k]), "one-launch boundary changed %s" % k
    # round 5: the recursion's ten tiles dealt to two waves per chain (k_adapt_split, TINYDA_ADAPT_SPLIT=1): the same operations per
    # element -- bitwise
    spl = _probe(what, {"TINYDA_FUSE_ADAPT_CHOL": "0", "TINYDA_ADAPT_SPLIT": "1"}, tmp_path, "split_recursion")
    for k in blk:
        assert np.array_equal(blk[k], spl[k]), "k_adapt_split changed %s" % k
    # round 5, late: the diagonal blocks of Sigma as circulant slots (adapt_am_chain_c64, the default at 64 parameters; 264 instead of
    # 320 operations per state) against full diagonal tiles (TINYDA_ADAPT_CIRC=0), in the one-launch boundary and in k_adapt alone: the
    # same operations per element -- bitwise
    full = _probe(what, {"TINYDA_ADAPT_CIRC": "0"}, tmp_path, "full_diagonal_tiles")
    full2 = _probe(what, {"TINYDA_ADAPT_CIRC": "0", "TINYDA_FUSE_ADAPT_CHOL": "0"}, tmp_path, "full_diagonal_tiles_two_launches")
    for k in blk:
        assert np.array_equal(blk[k], full[k]), "circulant diagonal blocks changed %s" % k
        assert np.array_equal(two[k], full2[k]), "circulant diagonal blocks (k_adapt) changed %s" % k


@pytest.mark.parametrize("what", ["am_d40", "am_d33_ragged"])
def test_circulant_diagonal_blocks_with_padded_parameters(what, tmp_path):
    """the moment recursion with the diagonal blocks of Sigma as circulant slots (the default at 33 .. 64 parameters) against
    full diagonal tiles (TINYDA_ADAPT_CIRC=0) where the 64-parameter instance is padded: the same operations per element -- every
    record, Sigma, C, the scaling bitwise; in the one-launch boundary and in k_adapt alone"""
    for extra in ({}, {"TINYDA_FUSE_ADAPT_CHOL": "0"}):
        circ = _probe(what, dict(extra), tmp_path, "circ%d" % len(extra))
        full = _probe(what, dict(extra, TINYDA_ADAPT_CIRC="0"), tmp_path, "full%d" % len(extra))
        for k in circ:
            assert np.array_equal(circ[k], full[k]), "circulant diagonal blocks changed %s (%s, %s)" % (k, what, extra)
        assert 0.02 < circ["acc0"].mean() < 0.98 and np.abs(circ["C"]).max() > 0


@pytest.mark.parametrize("what", ["aem_dense", "aem_dense_ragged", "aem_dense_da_pcn", "aem_dense_m200", "aem_dense_m200_ragged", "aem_dense_chunks"])
def test_error_model_base_subchain_kernels_agree(what, tmp_path):
    """dense error model over linear levels: the base subchain on k_aem_base_steps (one pass over each chain's factor V per launch,
    linear update of V r) against k_ml_steps (TINYDA_AEM_BASE=0: -1/2 |V r'|^2 evaluated per step): same decisions, log-densities
    to rounding"""
    one, per_step = _probe(what, {}, tmp_path, "one_pass"), _probe(what, {"TINYDA_AEM_BASE": "0"}, tmp_path, "per_step")
    for k in one:
        if k.startswith("acc"):
            assert np.array_equal(one[k], per_step[k]), "%s: %d accept flips" % (k, int((one[k] != per_step[k]).sum()))
        else:
            np.testing.assert_allclose(one[k], per_step[k], rtol=1e-10, atol=1e-12, err_msg=k)
    assert 0.02 < one["acc0"].mean() < 0.98


@pytest.mark.parametrize("what", ["aem_dense", "aemd_lean", "mlda3_short"])
def test_deferred_moment_recursion_is_bitwise_the_per_block_one(what, tmp_path):
    """AdaptiveMetropolis below an error model (blocks of one base subchain) or with blocks shorter than the period: the period's
    states are folded into the moments in one launch at the boundary instead of block by block (TINYDA_AM_DEFER=0) -- the same
    recursion over the same states in the same order"""
    a, b = _probe(what, {}, tmp_path, "deferred"), _probe(what, {"TINYDA_AM_DEFER": "0"}, tmp_path, "per_block")
    for k in a:
        assert np.array_equal(a[k], b[k]), "%s differs (%s)" % (k, what)
    assert 0.02 < a["acc0"].mean() < 0.98


@pytest.mark.parametrize("what", ["aem_dense", "aemd_lean", "aem_dense_da_pcn", "aemd"])
def test_window_of_draws_is_bitwise_the_per_block_draws(what, tmp_path):
    """error-model hierarchies: proposal increments and uniforms of a whole window of steps in one launch, the blocks (one base
    subchain each) walking through it -- against one k_propose per block (TINYDA_ML_PREDRAW=0): the same counters, the same draws"""
    a, b = _probe(what, {}, tmp_path, "window"), _probe(what, {"TINYDA_ML_PREDRAW": "0"}, tmp_path, "per_block")
    for k in a:
        assert np.array_equal(a[k], b[k]), "%s differs (%s)" % (k, what)
    assert 0.02 < a["acc0"].mean() < 0.98


@pytest.mark.parametrize("what", ["aem_dense", "aem_dense_ragged", "aem_dense_da_pcn", "aem_dense_m200"])
def test_error_model_outputs_on_the_matrix_cores_agree(what, tmp_path):
    """dense error model over linear levels: the model outputs k_aem_action needs for all chains from one k_linear_outputs_multi launch
    (an operator fragment serves a 16-chain tile) against three matrix-vector products per chain inside the kernel
    (TINYDA_AEM_PRE=0): the same products summed in another order -- decisions equal, densities to rounding"""
    a, b = _probe(what, {}, tmp_path, "pre"), _probe(what, {"TINYDA_AEM_PRE": "0"}, tmp_path, "gemv")
    for k in a:
        if k.startswith("acc"):
            assert np.array_equal(a[k], b[k]), "%s: %d accept flips" % (k, int((a[k] != b[k]).sum()))
        else:
            np.testing.assert_allclose(a[k], b[k], rtol=1e-10, atol=1e-12, err_msg=k)
    assert 0.02 < a["acc0"].mean() < 0.98


@pytest.mark.parametrize("what", ["dream", "dream_ragged"])
def test_dream_kernel_choices_agree(what, tmp_path):
    """the fused block (draws + steps in one launch) against k_dreamz_draw -> k_dreamz_steps_wave (TINYDA_DZ_FUSED=0): bitwise;
    k_dreamz_steps_wave against the 16-chain tile kernel (TINYDA_DZ_WAVE=0), and the draw-ahead pipeline
    (TINYDA_DZ_PIPELINE=1: same sums in the same order, bitwise)"""
    fused = _probe(what, {}, tmp_path, "fused")  # (512 chains: draws and steps of a block in ONE launch, k_dreamz_draw<32, false, true>)
    wave = _probe(what, {"TINYDA_DZ_FUSED": "0"}, tmp_path, "wave")
    tile = _probe(what, {"TINYDA_DZ_FUSED": "0", "TINYDA_DZ_WAVE": "0"}, tmp_path, "tile")
    pipe = _probe(what, {"TINYDA_DZ_FUSED": "0", "TINYDA_DZ_PIPELINE": "1"}, tmp_path, "pipe")
    for k in ("acc0", "stats0", "params0", "pCR"):
        assert np.array_equal(fused[k], wave[k]), "the fused block changed %s" % k
    assert np.array_equal(wave["acc0"], tile["acc0"])
    np.testing.assert_allclose(wave["stats0"], tile["stats0"], rtol=1e-10)
    np.testing.assert_allclose(wave["params0"], tile["params0"], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(wave["pCR"], tile["pCR"], rtol=1e-9)
    for k in ("acc0", "stats0", "params0", "pCR"):
        assert np.array_equal(wave[k], pipe[k]), "draw-ahead pipeline changed %s" % k
    assert 0.01 < wave["acc0"].mean() < 0.9


@pytest.mark.parametrize("what,other", [("aemd", {"TINYDA_AEMD_FUSED": "0"}), ("aemd_lean", {"TINYDA_AEMD_FUSED": "0"}),
                                        ("aemd_lean", {"TINYDA_DA_LEAN": "0"}), ("aemd_ragged", {"TINYDA_AEMD_FUSED": "0"}),
                                        ("aemd_lean_ragged", {"TINYDA_AEMD_FUSED": "0"})])
def test_fused_and_per_step_diagonal_error_model_agree(what, other, tmp_path):
    """diagonal error model over three linear levels: base subchains in the fused level kernels (k_ml_steps; k_da_steps at <= 128
    outputs) against one propose / outputs / accept triple per base step (TINYDA_AEMD_FUSED=0) and against each other"""
    fused, steps = _probe(what, {}, tmp_path, "fused"), _probe(what, other, tmp_path, "other")
    for k in fused:
        if k.startswith("acc"):
            assert np.array_equal(fused[k], steps[k]), "%s: %d accept flips" % (k, int((fused[k] != steps[k]).sum()))
        else:
            np.testing.assert_allclose(fused[k], steps[k], rtol=1e-9, atol=1e-11, err_msg=k)
    assert 0.02 < fused["acc0"].mean() < 0.98 and 0.02 < fused["acc2"].mean() <= 1.0


def test_long_base_subchains_of_the_dense_error_model_take_the_step_by_step_kernel(tmp_path):
    """k_aem_base_steps keeps its S + 2 product vectors in LDS; a base subchain whose set exceeds 64 KB (beyond ~58 steps at
    128 outputs) runs on the level kernel instead of asking for more LDS than a workgroup may have -- and agrees with TINYDA_AEM_BASE=0"""
    a, b = _probe("aem_dense_long", {}, tmp_path, "auto"), _probe("aem_dense_long", {"TINYDA_AEM_BASE": "0"}, tmp_path, "per_step")
    for k in a:
        if k.startswith("acc"):
            assert np.array_equal(a[k], b[k]), k
        else:
            np.testing.assert_allclose(a[k], b[k], rtol=1e-10, atol=1e-12, err_msg=k)
