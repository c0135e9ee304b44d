"""BASELINE.json's remaining single-GPU shapes, end to end through the C-ABI against the oracle:
  configs[2]  multi-instrument piano-roll, 16 tracks x 128 pitches (P = 2048), latent 256, batch 64, T 256, bf16
  configs[4]  long sequence T = 1024, fp16 MFMA path, 32 samples per GPU (256 over 8 GPUs): the streaming attention
              kernels (a 1024-long sequence does not fit the resident ones) inside a whole step
and the ELBO tolerance of north_star (1e-3 relative) at the RAW Xavier initialisation, with weight rounding separated
from kernel error (the oracle is fed the weights as the kernels read them)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

WIDTHS = (256, 2, 8, 128, 1, 8)  # scripts/train-vae.sh: encoder 256 x 2 layers x 8 heads, decoder 128 x 1 x 8
CFG1 = (128, 128, 2, 64) + WIDTHS
CFG2 = (2048, 2048, 2, 256) + WIDTHS


def test_config2_multi_instrument_step(gpu):
    """configs[2]: P = 16 x 128, Z = 256 — forward, losses, every gradient and the Adam update against the oracle"""
    from test_step_gpu import _compare_step
    _compare_step(gpu, "pianoroll", CFG2, B=64, T=256, seed=2048, steps=1, lr=3e-4)


def test_config4_long_sequence_fp16_step(gpu):
    """configs[4]'s shape with a batch the oracle's autograd handles in seconds (B 8 of the 32 per GPU): T = 1024, fp16"""
    from test_step_gpu import _compare_step
    # (max_err: the top encoder layer's row-wise weight gradients sum over B = 8 rows only — measured 10 % of the largest element)
    _compare_step(gpu, "pianoroll", CFG1, B=8, T=1024, seed=1024, steps=1, lr=3e-4, dtype=torch.float16, max_err=0.2)


def test_config4_full_per_gpu_batch(gpu):
    """configs[4] at its full per-GPU size (32 x 1024 frames, fp16): one whole training step runs; ELBO / KL / reconstruction
    of that step against the oracle's forward pass, every gradient finite, and the captured graph replays it"""
    from test_step_gpu import _setup
    O, E, ocfg, ecfg, params, batch, eps = _setup("pianoroll", CFG1, 32, 1024, 4096)
    store = E.ParamStore(ecfg, gpu, torch.float16, params_np=params)
    plan = E.StepPlan(store, 32, 1024, lr=3e-4, clip_gradient=1.0)
    plan.load_batch(batch["x"], batch["seq_lens"], batch["classes"], batch["labels"], eps)
    with torch.no_grad():
        P = O.to_torch_params(params, requires_grad=False)
        loss, recon, kl, _, means, stds = O.step_losses(P, ocfg, batch, torch.from_numpy(eps))
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        plan.step_kernels(True)
        st.synchronize()
        rel = lambda a, b: abs(float(a) - float(b)) / abs(float(b))
        assert rel(plan.total.mean().item(), loss.mean()) <= 1e-3, (plan.total.mean().item(), float(loss.mean()))
        assert rel(plan.kl.mean().item(), kl.mean()) <= 1e-3
        assert rel(plan.recon.mean().item(), recon.mean()) <= 1e-3
        assert np.sqrt(((plan.sigma.cpu().numpy() - stds.numpy()) ** 2).mean()) <= 3e-3
        g = store.g.cpu().numpy()
        assert np.isfinite(g).all() and np.abs(g).max() > 0
        w1 = store.w.clone()
        plan.capture(True)
        plan.run()
        st.synchronize()
        assert int(store.step_state[0].item()) == 2
        assert torch.isfinite(store.w).all() and not torch.equal(store.w, w1)
        # one Adam step later (every weight moved by ~lr in its gradient's direction: the oracle's ELBO drops from 53.6 to
        # 22.5 on a B 8 x T 128 slice of this setup): finite and lower
        second = plan.total.mean().item()
        assert np.isfinite(second) and 0.0 < second < float(loss.mean())


def _raw_init_forward(gpu, seed, dtype):
    """forward pass + losses of configs[1] at the PLAIN Xavier initialisation (what bench.py and train-vae.sh start from),
    on the GPU and in the oracle fed the weights as the kernels read them (ParamStore.as_consumed_numpy)"""
    from test_step_gpu import _setup
    O, E, ocfg, ecfg, params, batch, eps = _setup("pianoroll", CFG1, 64, 256, seed, sigma_bias=0.0, ragged=False)
    store = E.ParamStore(ecfg, gpu, dtype, params_np=params)
    plan = E.StepPlan(store, 64, 256)
    plan.load_batch(batch["x"], batch["seq_lens"], batch["classes"], batch["labels"], eps)
    plan.forward()
    plan.losses(with_grad=False)
    torch.cuda.synchronize()
    with torch.no_grad():
        P = O.to_torch_params(store.as_consumed_numpy(), requires_grad=False)
        loss, recon, kl, _, means, stds = O.step_losses(P, ocfg, batch, torch.from_numpy(eps))
    got = dict(total=plan.total.cpu().numpy(), recon=plan.recon.cpu().numpy(), kl=plan.kl.cpu().numpy(), mu=plan.mu.cpu().numpy(),
               sigma=plan.sigma.cpu().numpy())
    ref = dict(total=loss.numpy(), recon=recon.numpy(), kl=kl.numpy(), mu=means.numpy(), sigma=stds.numpy())
    return got, ref


def _rel(a, b):
    return abs(float(a) - float(b)) / abs(float(b))


@pytest.mark.parametrize("seed", [1234, 99, 7])
def test_full_size_config1_raw_init_bf16(gpu, seed):
    """configs[1] in bf16 at the raw Xavier init, weight rounding taken out (the oracle reads the same rounded GEMM weights).
    sigma is a raw linear output that straddles 0 (model.py:100-103) and KL = 0.5 * sum(sigma^2 + mu^2 - 1 - log sigma^2)
    has no epsilon (loss.py:9): ~25 of the 4096 sigma values of a batch have |sigma| < 1e-2, where -log sigma^2 turns the
    5e-3 rms error that ANY bf16 evaluation of the encoder leaves on sigma (tests/diag_rounding.py: rounding layer 0's or
    the top layer's activations alone gives 4e-3) into O(1) errors of those terms. So:
      * reconstruction loss, and mu / sigma themselves: tight;
      * KL over the elements away from the singularity (|sigma| >= 0.05, 96 % of them): within 1e-3 relative;
      * the full ELBO: within 4e-3 — the singular elements' share, not kernel error (the fp16 path, whose sigma error is
        5x smaller, meets 1e-3 on the full ELBO: next test)."""
    got, ref = _raw_init_forward(gpu, seed, torch.bfloat16)
    assert _rel(got["recon"].mean(), ref["recon"].mean()) <= 1e-3
    for k in ("mu", "sigma"):
        d = got[k] - ref[k]
        assert np.sqrt((d ** 2).mean()) <= 8e-3 and np.abs(d).max() <= 6e-2, (k, np.sqrt((d ** 2).mean()), np.abs(d).max())
    kl_terms = lambda mu, s: 0.5 * (s * s + mu * mu - 1.0 - np.log(s * s))
    away = np.abs(ref["sigma"]) >= 0.05
    assert away.mean() > 0.9
    a, b = kl_terms(got["mu"], got["sigma"])[away].sum(), kl_terms(ref["mu"], ref["sigma"])[away].sum()
    assert _rel(a, b) <= 1e-3, (a, b, _rel(a, b))
    assert _rel(got["total"].mean(), ref["total"].mean()) <= 4e-3, (got["total"].mean(), ref["total"].mean())
    assert _rel(got["kl"].mean(), ref["kl"].mean()) <= 4e-3


@pytest.mark.parametrize("seed", [1234, 99, 7])
def test_full_size_config1_raw_init_fp16_elbo(gpu, seed):
    """the same comparison on the fp16 path (11 significand bits instead of 8): north_star's ELBO tolerance of 1e-3 relative
    holds on the FULL ELBO at the raw Xavier init"""
    got, ref = _raw_init_forward(gpu, seed, torch.float16)
    for k in ("total", "kl", "recon"):
        assert _rel(got[k].mean(), ref[k].mean()) <= 1e-3, (k, got[k].mean(), ref[k].mean())
    assert np.sqrt(((got["sigma"] - ref["sigma"]) ** 2).mean()) <= 2e-3
