"""The data-parallel step on a real GPU, through the PRODUCT's multi-rank code: two worker processes (tests/dp_worker.py,
started by conftest.py before this process touched HIP) each run half of a global batch — engine.StepPlan with
capture(split_optimizer, overlap) + parallel.GradReducer, and Trainer with WORLD_SIZE=2 — sharing the box's card with
gloo carrying the all-reduce; this process runs the same global batch alone and compares (shard-invariant
eps, normalisation by the global batch, identical weights on every rank)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _close_after_adam(a, b, lr, steps):
    """weights after `steps` Adam updates from gradients that differ in summation order only: equal to fp32 noise
    except where a near-zero gradient flips sign (Adam's update is ~lr * sign(g) whatever |g| is)"""
    d = np.abs(a - b)
    assert d.max() <= 2.1 * lr * steps, d.max()
    assert (d > 2e-5).mean() < 0.02, (d > 2e-5).mean()


def test_engine_two_ranks_match_the_single_process_global_batch(gpu, dp_results):
    import dp_worker as W
    ref = W.run_engine(0, 1)  # the whole global batch in this process: one graph, no collective
    r0, r1 = dp_results
    per = W.B_GLOBAL // 2
    assert int(r0["eng_three_graphs"]) == 1 and int(r1["eng_three_graphs"]) == 1  # the overlapped schedule ran
    assert int(r0["eng_steps"]) == int(r1["eng_steps"]) == ref["steps"] == W.STEPS
    # eps is drawn per GLOBAL sample index: the shards see exactly the rows a single process draws
    assert np.array_equal(r0["eng_eps1"], ref["eps1"][:per]) and np.array_equal(r1["eng_eps1"], ref["eps1"][per:])
    # per-sample losses of step 1 are the single-process ones, bit for bit (same kernels on the same rows)
    np.testing.assert_allclose(np.concatenate([r0["eng_total1"], r1["eng_total1"]]), ref["total1"], rtol=1e-5)
    # the all-reduced bucket = the global-batch gradient; both ranks hold the same bytes
    assert np.array_equal(r0["eng_g1"], r1["eng_g1"])
    scale = np.abs(ref["g1"]).max()
    assert np.abs(r0["eng_g1"] - ref["g1"]).max() <= 2e-4 * scale, np.abs(r0["eng_g1"] - ref["g1"]).max() / scale
    # weights stay identical across ranks and follow the single-process trajectory
    assert np.array_equal(r0["eng_w"], r1["eng_w"])
    _close_after_adam(r0["eng_w"], ref["w"], W.LR, W.STEPS)


def test_wide_engine_two_ranks_share_the_card_with_the_one_launch_tails(gpu, dp_results):
    """the same comparison at the headline configuration's widths: both ranks run the one-launch position-0 tails (a grid barrier among
    16 workgroups that must land on one XCD) and the attention launch with the projection inside WHILE SHARING THE CARD — the
    co-tenancy the step guard exists for. No step may have been skipped, the tails must still be in use, and the two ranks' run must
    equal the single-process run of the global batch."""
    import dp_worker as W
    ref = W.run_engine(0, 1, wide=True)
    r0, r1 = dp_results
    for r in (r0, r1):
        assert int(r["wide_skipped"]) == 0 and int(r["wide_tail_fused"]) == 1 and int(r["wide_tails"]) == 2
        assert int(r["wide_three_graphs"]) == 1 and int(r["wide_steps"]) == W.STEPS
    assert ref["tails"] == 2 and ref["skipped"] == 0
    per = W.B_GLOBAL // 2
    assert np.array_equal(r0["wide_eps1"], ref["eps1"][:per]) and np.array_equal(r1["wide_eps1"], ref["eps1"][per:])
    # At these widths the feed-forward launches are the fused ones, whose workgroups walk the hidden chunks in an order rotated by
    # their position in the launch (gemm_nt.hip MST_FFN_ROT: 8 us per step): the fp32 sum of FFN2 runs in another order for a row
    # block that sits elsewhere in the launch, i.e. a shard equals the global batch to a rounding of the activation type (most
    # samples bit for bit, a few elements one bf16 ulp apart), not bit for bit as at the narrow widths above
    np.testing.assert_allclose(np.concatenate([r0["wide_total1"], r1["wide_total1"]]), ref["total1"], rtol=2e-3)
    assert (np.concatenate([r0["wide_total1"], r1["wide_total1"]]) == ref["total1"]).mean() >= 0.5
    assert np.array_equal(r0["wide_g1"], r1["wide_g1"]) and np.array_equal(r0["wide_w"], r1["wide_w"])
    scale = np.abs(ref["g1"]).max()
    assert np.abs(r0["wide_g1"] - ref["g1"]).max() <= 1e-2 * scale, np.abs(r0["wide_g1"] - ref["g1"]).max() / scale
    # (Adam's update is ~lr * sign(g): gradients that agree to a rounding still flip the sign of the near-zero ones)
    d = np.abs(r0["wide_w"] - ref["w"])
    assert d.max() <= 2.1 * W.LR * W.STEPS and (d > 0.5 * W.LR).mean() < 0.05, (d.max(), (d > 0.5 * W.LR).mean())


def test_trainer_two_ranks_match_the_single_rank_trainer(gpu, dp_results):
    import dp_worker as W
    for k in ("WORLD_SIZE", "RANK"):
        assert os.environ.get(k) in (None, "1", "0"), "this process must be a single-rank run"
    ref = W.run_trainer()
    r0, r1 = dp_results
    assert int(r0["tr_steps"]) == int(r1["tr_steps"]) == ref["steps"] == 3
    assert np.array_equal(r0["tr_w"], r1["tr_w"])
    _close_after_adam(r0["tr_w"], ref["w"], W.LR, 3)
    # collect_metrics all-reduces the sums: every rank reports the global-batch means
    for k in ("total_loss", "kl_loss"):
        assert abs(float(r0["tr_" + k]) - float(r1["tr_" + k])) <= 1e-6 * abs(float(r0["tr_" + k]))
        assert abs(float(r0["tr_" + k]) - ref[k]) <= 2e-3 * abs(ref[k]), (k, float(r0["tr_" + k]), ref[k])
