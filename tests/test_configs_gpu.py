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
    _compare_step(gpu, "pianoroll", CFG1, B=8, T=1024, seed=1024, steps=1, lr=3e-4, dtype=torch.float16)


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
        assert abs(plan.total.mean().item() - float(loss.mean())) <= 0.05 * abs(float(loss.mean()))  # one lr = 3e-4 step later


@pytest.mark.parametrize("seed", [1234, 99, 7])
def test_full_size_config1_elbo_raw_init(gpu, seed):
    """configs[1] at the PLAIN Xavier initialisation (what bench.py and train-vae.sh start from: sigma straddles 0, where
    KL = ... - log sigma^2 is singular, loss.py:9). With the oracle reading the same bf16-rounded GEMM weights the kernels
    read, what is left is the kernels' own error: ELBO, KL and reconstruction within 1e-3 relative."""
    from test_step_gpu import _compare_step
    _compare_step(gpu, "pianoroll", CFG1, B=64, T=256, seed=seed, steps=1, lr=3e-4, sigma_bias=0.0, ragged=False,
                  check_grads=False, consumed_weights=True)


def test_full_size_config1_elbo_raw_init_fp32_weights_bound(gpu):
    """the same comparison against the oracle on the fp32 master weights: weight rounding included, the documented
    looser bound (rounding only the weights to bf16 already moves the batch-mean KL by up to 9e-4)"""
    from test_step_gpu import _compare_step
    _compare_step(gpu, "pianoroll", CFG1, B=64, T=256, seed=1234, steps=1, lr=3e-4, sigma_bias=0.0, ragged=False,
                  check_grads=False, elbo_tol=4e-3)
